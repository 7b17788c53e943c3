"""Shared builders for the parity tests: one dict -> oracle Spec + ModelConfig,
plus small synthetic datasets with the reference datasets' sparsity."""
import numpy as np

from oracle import sisua_oracle as so


def make_pair(**kw):
  from sisua_amd.config import ModelConfig
  return so.Spec(**kw), ModelConfig(**kw)


def synth_counts(n, g, sparsity=0.9, seed=0, max_count=None):
  rng = np.random.default_rng(seed)
  mg = rng.normal(-0.5, 1.2, size=g)
  sc = rng.lognormal(0.0, 0.3, size=n)
  lam = sc[:, None] * np.exp(mg)[None, :] * rng.gamma(2.0, 0.5, size=(n, g))
  x = rng.poisson(lam).astype(np.float32)
  thr = np.quantile(rng.uniform(size=(n, g)), 1 - sparsity)
  keep = rng.uniform(size=(n, g)) > sparsity
  x = x * keep
  if max_count:
    x[rng.integers(0, n), rng.integers(0, g)] = max_count
  x[:, 0] = np.maximum(x[:, 0], 1.0)  # no all-zero cell (library size is log of the total)
  return x.astype(np.float32)


def synth_labels(n, labels, seed=1):
  rng = np.random.default_rng(seed)
  ys = []
  for P, kind in labels:
    if kind == "onehot":
      ys.append(np.eye(P, dtype=np.float32)[rng.integers(0, P, n)])
    elif kind.startswith("mixtril"):   # continuous and CORRELATED across the label dimensions: two populations with their own covariance
      on = rng.uniform(size=(n, 1)) < 0.4
      A0, A1 = rng.normal(size=(P, P)) * 0.4 + np.eye(P) * 0.6, rng.normal(size=(P, P)) * 0.3 + np.eye(P) * 0.5
      e = rng.normal(size=(n, P))
      ys.append(np.where(on, 2.0 + e @ A1.T, -0.5 + e @ A0.T).astype(np.float32))
    elif kind.startswith("mixgauss"):   # continuous, bimodal (log-normalised protein levels: a background and a signal mode)
      on = rng.uniform(size=(n, P)) < 0.4
      ys.append(np.where(on, rng.normal(2.5, 0.5, size=(n, P)), rng.normal(-0.5, 0.8, size=(n, P))).astype(np.float32))
    elif kind.startswith("mixzinb"):   # the bimodal counts below with dropouts: a third of the entries zeroed
      on = rng.uniform(size=(n, P)) < 0.4
      yy = np.where(on, rng.poisson(30.0, size=(n, P)), rng.poisson(8.0, size=(n, P)))   # (neither mode puts mass on zero: the zeros need the gate)
      ys.append((yy * (rng.uniform(size=(n, P)) > 0.33)).astype(np.float32))
    elif kind.startswith("mixnb"):   # bimodal ADT-like counts: a background and a signal population per protein
      on = rng.uniform(size=(n, P)) < 0.4
      ys.append(np.where(on, rng.poisson(30.0, size=(n, P)), rng.poisson(2.0, size=(n, P))).astype(np.float32))
    else:
      ys.append(np.clip(rng.lognormal(0.6, 0.6, size=(n, P)), 0.5, 9.1).astype(np.float32))
  return ys


def perturbed_params(spec, scale=0.05, seed=3):
  """Oracle init moved off the symmetric point (gamma=1, beta=0, b=0), fp32-representable."""
  rng = np.random.default_rng(seed)
  p = so.init_params(spec)
  out = {k: (v + scale * rng.normal(size=v.shape)).astype(np.float32).astype(np.float64) for k, v in p.items()}
  for name, on in (("prior/loc", getattr(spec, "tie_loc", False)), ("prior/scale", getattr(spec, "tie_scale", False))):
    if on and name in out:   # SCALE's tied tensors keep identical rows
      out[name] = np.broadcast_to(out[name][:1], out[name].shape).copy()
  if getattr(spec, "scale_tril", False):   # full-covariance components: off-diagonals that matter (the 0.05 above leaves L ~ I)
    out["prior/scale"] = (out["prior/scale"] + 0.3 * rng.normal(size=out["prior/scale"].shape)).astype(np.float32).astype(np.float64)
  if getattr(spec, "tie_mixtures", False) and "prior/logits" in out:
    out["prior/logits"] = np.zeros_like(out["prior/logits"])
  return out


def rel_l2(a, b, floor=1e-5):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  d = np.linalg.norm(a - b)
  # floor: a gradient that is analytically zero (e.g. a bias feeding BatchNorm) is fp32 noise on the GPU
  return d / max(np.linalg.norm(b), floor)


def grad_errors(got, ref, floor_frac=1e-3):
  """Per-tensor relative L2 error; tensors whose true gradient is below `floor_frac` of the largest
  gradient norm are judged on that absolute scale (fp32 rounding of an exact zero)."""
  top = max(np.linalg.norm(np.asarray(v, np.float64)) for v in ref.values())
  return {k: rel_l2(got[k], ref[k], floor=max(floor_frac * top, 1e-5)) for k in got}


def adam_state_errors(engine, opt, floor_frac=1e-3):
  """Per-tensor relative L2 error of the optimiser's first and second moments on the device (smx_get_tensor which = 2 / 3)
  against the oracle's.  m is LINEAR in the gradients and v quadratic, so -- unlike the post-update weights, where Adam turns
  a rounding-level gradient into a full-size step -- they can be held to the gradients' own tolerance.
  Returns (worst m error, worst v error, their tensors)."""
  em = grad_errors(engine.get_params(which=2), opt["m"], floor_frac)
  ev = grad_errors(engine.get_params(which=3), opt["v"], floor_frac)
  km, kv = max(em, key=em.get), max(ev, key=ev.get)
  return em[km], ev[kv], (km, kv)


def masked_move_error(after, before, ref_after, grad, lr, frac=1e-2):
  """Relative L2 error of one tensor's update (after - before vs the oracle's) over the elements whose gradient is above
  `frac` of the tensor's largest: where |g| is far above float32 rounding the Adam step is a smooth function of g.  None when
  no element qualifies (a tensor whose gradient is rounding noise throughout)."""
  g = np.abs(np.asarray(grad, np.float64))
  big = g > frac * g.max() if g.size and g.max() > 0 else np.zeros(g.shape, bool)
  if not big.any():
    return None
  return rel_l2((np.asarray(after, np.float64) - before)[big], (ref_after - before)[big], floor=1e-3 * lr * np.sqrt(big.sum()))
