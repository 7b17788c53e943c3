"""Plain configuration records of the model API.

`RVmeta` and `NetConf` mirror the two odin config records the reference passes
around (sisua/train.py:75-89; sisua/models/single_cell_model.py:74-81): they are
data only.  `ModelConfig` is the flattened form handed to the C-ABI
(include/sisua_hip.h: smx_config).
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

SOFTPLUS_INV_1 = float(np.log(np.expm1(1.0)))   # softplus(SOFTPLUS_INV_1) == 1
LIKELIHOODS = ("nb", "zinb", "nbd", "zinbd")
OUTPUT_POSTERIORS = LIKELIHOODS + ("mse",)   # 'mse': deterministic output, -log_prob(x) = mean squared error (tests/test_singlecell_models.py:82-91)


def label_planes(llk: str, P: int = 0) -> int:
  """Raw head outputs per label dimension: 'nb' 2 (log total_count, logits), 'onehot' 1, 'mixnbC' 3 C (C mixture
  logits, C log total_counts, C logits: MISA's mixture-of-NB labels, sisua/models/vae.py:47-98), 'mixgaussC' 3 C (C mixture
  logits, C locations, C raw scales: its mixture-of-Gaussians labels for continuous variables, vae.py:86-92), 'mixtrilC'
  C (2 + P) (the docstring example of vae.py:58: C full-covariance Gaussians over the whole label vector -- C planes whose
  first column is a component's mixture logit, C planes of locations, per component P planes = the columns of its
  lower-triangular scale factor; the other entries are inert)."""
  if llk.startswith("mixtril"):
    if P <= 0:
      raise ValueError("label_planes('mixtrilC') needs the label dimension")
    return mixture_components(llk) * (2 + P)
  if llk.startswith("mixzinb"):   # MISA(zero_inflated=True, vae.py:76-84): the 'mixnb' planes + C zero-inflation gate logits
    return 4 * mixture_components(llk)
  if llk in ("nbd", "zinb", "zinbd"):   # the other count posteriors as heads (vae.py:30): planes as for the gene output
    return 2 if llk == "nbd" else 3
  return 2 if llk == "nb" else 1 if llk == "onehot" else 3 * mixture_components(llk)


def mixture_components(llk: str) -> int:
  """C of 'mixnbC' / 'mixgaussC' (0: not a mixture head)."""
  return int(llk[-1]) if llk.startswith("mix") else 0


@dataclass
class RVmeta:
  """Random-variable description: RVmeta(event_shape, posterior, projection, name)
  (sisua/train.py:79-89; data/_single_cell_base.py:518-533)."""
  event_shape: int = 10
  posterior: str = "diag"
  projection: bool = True
  name: Optional[str] = None
  kwargs: dict = field(default_factory=dict)

  def __post_init__(self):
    if isinstance(self.event_shape, (tuple, list)):
      self.event_shape = int(np.prod(self.event_shape))
    self.event_shape = int(self.event_shape)
    self.posterior = str(self.posterior).lower()

  @property
  def is_zero_inflated(self):
    return self.posterior in ("zinb", "zinbd")

  @property
  def is_deterministic(self):
    return self.posterior in ("relu", "linear", "identity", "mse")

  def copy(self):
    return dataclasses.replace(self, kwargs=dict(self.kwargs))


@dataclass
class NetConf:
  """MLP description: NetConf(units, batchnorm, dropout, input_dropout)
  (configs/base.yaml:10-17; single_cell_model.py:78-81).  Block order is
  Dense -> BatchNorm -> ReLU -> Dropout (frozen third-party semantics)."""
  units: Sequence[int] = (64, 64)
  batchnorm: bool = True
  dropout: float = 0.0
  input_dropout: float = 0.0
  activation: str = "relu"
  name: Optional[str] = None

  def __post_init__(self):
    self.units = tuple(int(u) for u in (self.units if isinstance(self.units, (tuple, list)) else [self.units]))
    assert self.activation == "relu", "only relu hidden activations are built"

  def copy(self):
    return dataclasses.replace(self)


@dataclass(frozen=True)
class ModelConfig:
  """Flat model description; field names and defaults equal
  oracle/sisua_oracle.py:Spec so parity tests can build both from one dict."""
  model: str = "vae"
  n_genes: int = 0
  likelihood: str = "zinb"
  enc_units: Tuple[int, ...] = (64, 64)
  dec_units: Tuple[int, ...] = (64, 64)
  latent_dim: int = 10
  encl_units: Tuple[int, ...] = (64,)
  labels: Tuple[Tuple[int, str], ...] = ()
  # outputs[1:] of the reference's constructors (tests/test_singlecell_models.py:129-141; scvi.py:168-169): further fully observed
  # output variables -- heads on the decoder output with weight 1 and no label mask.  Heads are ordered extra outputs, then labels.
  extra_outputs: Tuple[Tuple[int, str], ...] = ()
  # scvi.py:55-56,66-86: 'full' = a Dense head; 'share' = one trainable per-gene vector (out1/b resp. out2/b without a kernel); 'single' = one scalar
  dispersion: str = "full"
  inflation: str = "full"
  batchnorm: bool = True
  dropout_enc: float = 0.1
  dropout_dec: float = 0.1
  input_dropout: float = 0.0
  log_norm: bool = True
  beta: float = 1.0
  alpha: float = 10.0
  latent_activation: str = "relu"
  clip_library: float = 1e3
  bn_momentum: float = 0.99
  bn_eps: float = 1e-3
  lr: float = 1e-3
  adam_beta1: float = 0.9
  adam_beta2: float = 0.999
  adam_eps: float = 1e-7
  clipnorm: float = 100.0
  seed: int = 8
  n_components: int = 10   # model 'scale': components of the Gaussian-mixture prior
  tie_mixtures: bool = False   # scale.py:29-33: uniform fixed mixture weights / one location / one scale vector for every component
  tie_loc: bool = False
  tie_scale: bool = False
  # scale.py:28,35 `covariance` of the mixture's components: 'none' / 'diag' (diagonal, prior/scale [C, D]) or 'tril' / 'full' (a
  # lower-triangular factor per component, prior/scale [C D, D]; diag = softplus(raw) + 1e-5; not with the tie_* options)
  covariance: str = "none"
  # scale.py:26,38-47 read literally: q(z|x) itself a mixture of n_components diagonal Gaussians (a (1 + 2 C) D-wide latent head: logits in
  # the first C columns of plane 0, C location planes, C raw-scale planes), standard-normal prior, Monte-Carlo KL; no prior/* tensors
  latent_mixture: bool = False
  # model 'fvae' (sisua/models/fvae.py:9-18; odin factorVAE defaults): the total-correlation discriminator
  disc_units: int = 1000
  disc_layers: int = 5
  gamma: float = 6.0
  disc_leak: float = 0.2

  @property
  def k(self) -> int:
    return 1 if self.likelihood == "mse" else 3 if self.likelihood in ("zinb", "zinbd") else 2

  @property
  def stochastic(self) -> bool:
    return self.model != "dca"

  @property
  def scale_tril(self) -> bool:
    if self.covariance not in ("none", "diag", "tril", "full"):
      raise ValueError(f"covariance must be 'none' / 'diag' or 'tril' / 'full', given: {self.covariance}")
    tril = self.model == "scale" and self.covariance in ("tril", "full")
    if self.latent_mixture:
      if self.model != "scale" or tril or self.tie_mixtures or self.tie_loc or self.tie_scale:
        raise ValueError("the mixture-density posterior (latent_mixture) is built for model 'scale' with covariance='none' and no tied parameters")
      if not 2 <= self.n_components <= min(self.latent_dim, 8):
        raise ValueError("the mixture-density posterior takes 2 .. min(latent_dim, 8) components")
    if tril and (self.tie_mixtures or self.tie_loc or self.tie_scale):
      raise ValueError("tied mixture parameters are built for diagonal components only (covariance='none')")
    return tril

  @property
  def disc_outputs(self) -> int:
    return sum(P for P, _ in self.labels) if self.labels else 1   # (SemiFVAE: the classes of every label variable, one variable behind the other)

  @property
  def head_labels(self):
    """(dim, kind) of every head on the decoder output: the extra outputs, then the label variables (SemiFVAE's one-hot labels are
    classified by the discriminator: no head)."""
    return self.extra_outputs + (() if self.model == "fvae" else self.labels)

  @property
  def targets(self):
    """(dim, kind) of the target arrays a dataset carries beside the counts, in upload order: extra outputs, then labels."""
    return self.extra_outputs + self.labels

  def head_plane(self, c: int) -> bool:
    """scvi: whether plane c of the gene output (0 MeanScale, 1 Dispersion, 2 DropoutLogits) is a Dense head (scvi.py:66-86)."""
    if self.dispersion not in ("full", "share", "single") or self.inflation not in ("full", "share", "single"):
      raise ValueError(f"dispersion / inflation must be 'full', 'share' or 'single', given: {self.dispersion} / {self.inflation}")
    if self.model != "scvi" and (self.dispersion != "full" or self.inflation != "full"):
      raise ValueError("dispersion / inflation are options of scvi (scvi.py:55-56)")
    return c == 0 or (c == 1 and self.dispersion == "full") or (c == 2 and self.inflation == "full")

  def plane_single(self, c: int) -> bool:
    """scvi: plane c is ONE trainable scalar for every cell and gene ('single': tensor out{c}/b of shape (1,))."""
    return (c == 1 and self.dispersion == "single") or (c == 2 and self.inflation == "single")

  def to_dict(self):
    return dataclasses.asdict(self)


def manifest(cfg: ModelConfig) -> List[Tuple[str, Tuple[int, ...]]]:
  """Trainable tensors in library order (names, logical shapes); checked
  against smx_tensor_info at model creation."""
  out = []

  def mlp(prefix, n_in, units):
    for i, u in enumerate(units):
      out.append((f"{prefix}{i}/W", (n_in, u)))
      if cfg.batchnorm:
        out.append((f"{prefix}{i}/gamma", (u,)))
        out.append((f"{prefix}{i}/beta", (u,)))
      else:
        out.append((f"{prefix}{i}/b", (u,)))
      n_in = u
    return n_in

  G, D = cfg.n_genes, cfg.latent_dim
  h = mlp("enc", G, cfg.enc_units)
  _ = cfg.scale_tril, cfg.head_plane(0)   # (validates the SCALE / scvi options)
  nl = (1 + 2 * cfg.n_components) * D if cfg.latent_mixture else (2 * D if cfg.stochastic else D)
  out += [("lat/W", (h, nl)), ("lat/b", (nl,))]
  if cfg.model == "scale" and not cfg.latent_mixture:
    C = cfg.n_components
    out += [("prior/logits", (C,)), ("prior/loc", (C, D)), ("prior/scale", (C * D, D) if cfg.scale_tril else (C, D))]
  if cfg.model == "scvi":
    hl = mlp("encl", G, cfg.encl_units)
    out += [("latl/W", (hl, 2)), ("latl/b", (2,))]
  hd = mlp("dec", D, cfg.dec_units)
  if cfg.model == "fvae":   # discriminator on z: Dense + bias, never BatchNorm
    n_in = D
    for i in range(cfg.disc_layers):
      out += [(f"disc{i}/W", (n_in, cfg.disc_units)), (f"disc{i}/b", (cfg.disc_units,))]
      n_in = cfg.disc_units
    out += [("discout/W", (n_in, cfg.disc_outputs)), ("discout/b", (cfg.disc_outputs,))]
  if cfg.model == "scvi":
    for c in range(cfg.k):
      out += ([(f"out{c}/W", (hd, G))] if cfg.head_plane(c) else []) + [(f"out{c}/b", (1,) if cfg.plane_single(c) else (G,))]
  else:
    out += [("out/W", (hd, cfg.k * G)), ("out/b", (cfg.k * G,))]
  for j, (P, llk) in enumerate(cfg.head_labels):
    ky = label_planes(llk, P)
    out += [(f"lab{j}/W", (hd, ky * P)), (f"lab{j}/b", (ky * P,))]
  return out


def init_params(cfg: ModelConfig, seed: Optional[int] = None) -> Dict[str, np.ndarray]:
  """Glorot-uniform kernels, zero biases, gamma = 1, beta = 0 (Keras defaults),
  drawn with numpy default_rng(seed) in manifest order."""
  rng = np.random.default_rng(cfg.seed if seed is None else seed)
  params = {}
  for name, shape in manifest(cfg):
    kind = name.split("/")[1]
    if kind == "W":
      limit = np.sqrt(6.0 / (shape[0] + shape[1]))
      params[name] = rng.uniform(-limit, limit, size=shape).astype(np.float32)
    elif kind == "gamma":
      params[name] = np.ones(shape, dtype=np.float32)
    elif name == "prior/loc":   # scale: the mixture components must not start identical
      params[name] = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
      if cfg.tie_loc:
        params[name][:] = 0.0
    elif name == "prior/scale" and cfg.scale_tril:   # L_c = I: softplus(log(e - 1)) = 1 on the diagonals
      params[name] = np.zeros(shape, dtype=np.float32)
      params[name][np.arange(shape[0]), np.arange(shape[0]) % shape[1]] = SOFTPLUS_INV_1
    else:
      params[name] = np.zeros(shape, dtype=np.float32)
  return params
