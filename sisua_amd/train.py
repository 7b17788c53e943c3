"""Counterpart of the reference's experiment driver for the training path
(sisua/train.py:61-147: on_load_data -> on_create_model -> on_train) without the
odin/hydra Experimenter: a plain dict / YAML file with the keys of configs/base.yaml.

    python -m sisua_amd.train model.name=vae dataset.name=8kly train.epochs=5

Datasets are the synthetic stand-ins of sisua_amd.data (the real ones need network
downloads); everything numeric runs in libsisua_hip.so.
"""
from __future__ import annotations

import copy
import inspect
import os
import sys
from functools import partial
from typing import Dict

import numpy as np

from sisua_amd import data
from sisua_amd.config import NetConf, RVmeta
from sisua_amd.models import get_model

DEFAULT_CONFIG = {  # configs/base.yaml:1-62
    "verbose": False,
    "model": {"name": "dca", "log_norm": True, "alpha": 10.0, "beta": 1.0, "gamma": 6.0, "lamda": 1.0,
              "encoder": {"units": [64, 64], "batchnorm": True, "dropout": 0.1},
              "decoder": {"units": [64, 64], "batchnorm": True, "dropout": 0.1}},
    "dataset": {"name": "cortex", "train_percent": 0.8, "labels_percent": 0.1, "batch_size": 64, "dropout_rate": 0.2,
                "retain_rate": 0.2},
    "variables": {"latents": {"event_shape": 12, "posterior": "diag", "kwargs": {}},
                  "transcriptomic": {"posterior": "zinbd", "kwargs": {}},
                  "chromatin": {"posterior": "zinbd", "kwargs": {}},
                  "proteomic": {"posterior": "nb", "kwargs": {}},
                  "celltype": {"posterior": "onehot", "kwargs": {}}},
    "train": {"optimizer": "adam", "learning_rate": 1e-3, "valid_freq": 500, "valid_interval": 0, "clipnorm": 100,
              "epochs": 500, "max_iter": -1, "sample_shape": [], "logging_interval": 2, "earlystop_threshold": 0.001,
              "earlystop_progress_length": 0, "earlystop_patience": 20, "earlystop_min_epoch": -1,
              "terminate_on_nan": True, "allow_rollback": False, "allow_none_gradients": False,
              "track_gradient_norms": False},
}


def get_dataset(name: str) -> data.SingleCellOMIC:
  """Synthetic stand-ins keyed like the reference registry (data/__init__.py:220-224)."""
  name = str(name).lower()
  if name in ("8kly", "pbmc8kly", "pbmc8k_ly"):
    x, y = data.synthetic_8kly()
    return data.SingleCellOMIC(x, name="8kly").add_omic("proteomic", y)
  if name in ("eccly", "pbmceccly", "pbmcecc_ly"):
    x, y = data.synthetic_eccly()
    return data.SingleCellOMIC(x, name="eccly").add_omic("proteomic", y)
  if name == "cortex":
    x, y = data.synthetic_cortex()
    return data.SingleCellOMIC(x, name="cortex").add_omic("celltype", y)
  raise ValueError(f"unknown dataset '{name}' (available: cortex, 8kly, eccly)")


def _from_config(cfg: Dict, fn, overrides=None):
  """train.py:31-43: keep only the keys the callee's signature accepts."""
  assert callable(fn)
  spec = inspect.getfullargspec(fn)
  ok = lambda k: k in spec.args or k in spec.kwonlyargs or spec.varkw is not None
  kw = {k: v for k, v in cfg.items() if ok(k)}
  kw.update({k: v for k, v in (overrides or {}).items() if ok(k)})
  return fn(**kw)


def config_hash(cfg: Dict, exclude_keys=("train", "verbose"), hash_length: int = 5) -> str:
  """Identity of an experiment: md5 of the configuration without the keys that do not change WHAT is trained
  (`Experimenter(exclude_keys=["train", "verbose"], hash_length=5)`, sisua/train.py:51-55)."""
  import hashlib
  import json
  core = {k: v for k, v in cfg.items() if k not in exclude_keys}
  return hashlib.md5(json.dumps(core, sort_keys=True, default=str).encode()).hexdigest()[:hash_length]


class Experiment:
  """on_load_data -> on_create_model -> on_train of sisua/train.py:61-147.  With `save_path` every configuration gets
  its own directory `<save_path>/exp_<5-char hash>/` (train.py:49-59): the model ALWAYS tries `load_weights` from
  there when it is created (train.py:107-108), so re-running an interrupted experiment resumes it -- the training
  schedule is a pure function of the restored step count (SingleCellModel._fit), the remaining epochs are
  `train.epochs` minus the epochs already done."""

  def __init__(self, cfg: Dict = None, model_dir: str = None, save_path: str = None):
    self.cfg = copy.deepcopy(DEFAULT_CONFIG)
    for k, v in (cfg or {}).items():
      if isinstance(v, dict) and isinstance(self.cfg.get(k), dict):
        _deep_update(self.cfg[k], v)
      else:
        self.cfg[k] = v
    self.hash = config_hash(self.cfg)
    if model_dir is None and save_path is not None:
      # SISUA_EXP names the experiments' ROOT as in the reference (train.py:49-59: one `exp_<hash>` directory per
      # configuration).  A directory that already holds a checkpoint itself (SISUA_EXP meant the model directory before
      # round 2) keeps being found: it is used as it is.
      legacy = any(os.path.exists(os.path.join(save_path, f)) for f in ("model.npz", "model.metamodel"))
      model_dir = save_path if legacy else os.path.join(save_path, f"exp_{self.hash}")
    self.model_dir = model_dir
    self.resumed_from = 0

  def on_load_data(self):
    ds = self.cfg["dataset"]
    self.sco = get_dataset(ds["name"])
    self.train, self.test = self.sco.split(train_percent=ds["train_percent"])

  def on_create_model(self):
    model = self.cfg["model"]
    cls = get_model(model["name"])
    encoder = _from_config(model["encoder"], NetConf)
    decoder = _from_config(model["decoder"], NetConf)
    omics = {o: self.sco.get_dim(o) for o in self.sco.omics}
    rv = {k: _from_config(v, RVmeta, overrides=dict(event_shape=omics[k] if k in omics else v.get("event_shape"),
                                                     projection=True, name=k))
          for k, v in self.cfg["variables"].items() if k in omics or k == "latents"}
    overrides = dict(outputs=rv["transcriptomic"], latents=rv["latents"], encoder=encoder, decoder=decoder)
    if "labels" in inspect.getfullargspec(cls.__init__).args:
      overrides["labels"] = [rv[o] for o in self.sco.omics[1:] if o in rv]
    self.model = _from_config({k: v for k, v in model.items() if k not in ("name", "encoder", "decoder")}, cls, overrides)
    if self.model_dir:
      self.model.load_weights(os.path.join(self.model_dir, "model"), verbose=self.cfg["verbose"])
      self.resumed_from = int(self.model.step)
    self.omics = [l.name for l in self.model.output_layers] + [l.name for l in self.model.labels]

  def on_train(self):
    ds, tr = self.cfg["dataset"], dict(self.cfg["train"])
    self.model.set_metadata(self.sco)
    train, valid = self.train.split(0.9)
    train.corrupt(dropout_rate=ds["dropout_rate"], retain_rate=ds["retain_rate"], inplace=True)
    mk = lambda s: s.create_dataset(self.omics, labels_percent=ds["labels_percent"], batch_size=ds["batch_size"],
                                    drop_remainder=True, shuffle=1000)
    fn_save = partial(self.model.save_weights, filepath=os.path.join(self.model_dir, "model")) if self.model_dir else None
    tr["sample_shape"] = tuple(tr.get("sample_shape", ()))
    # re-running an interrupted experiment finishes ITS schedule: `train.epochs` is then the total, not an increment
    tr["epochs_are_total"] = bool(self.resumed_from)
    self.model.fit(mk(train), valid=mk(valid), checkpoint=fn_save, log_tag=f"{self.cfg['model']['name']}-{ds['name']}",
                   verbose=self.cfg["verbose"], **tr)
    return self.model

  def run(self):
    self.on_load_data()
    self.on_create_model()
    return self.on_train()


def _deep_update(dst, src):
  for k, v in src.items():
    if isinstance(v, dict) and isinstance(dst.get(k), dict):
      _deep_update(dst[k], v)
    else:
      dst[k] = v


def _parse_overrides(argv):
  import yaml
  cfg = {}
  for a in argv:
    if "=" not in a:
      continue
    key, val = a.split("=", 1)
    node = cfg
    parts = key.split(".")
    for p in parts[:-1]:
      node = node.setdefault(p, {})
    node[parts[-1]] = yaml.safe_load(val)
  return cfg


if __name__ == "__main__":
  exp = Experiment(_parse_overrides(sys.argv[1:]), save_path=os.environ.get("SISUA_EXP"))
  m = exp.run()
  print(m, "final loss", (m.train_history.get("loss") or [None])[-1], "val", m.valid_history.get("val_loss", [None])[-1])
