"""A spread of configurations nobody wrote a test for one by one (tools/dev/matrix_probe.py: model family x likelihood x gene-panel width x
hidden widths x minibatch raggedness x BatchNorm / dropout): each against the oracle (ELBO terms of three steps, every gradient of the first,
the Adam moments) and across the three resident stores, which must agree bit for bit on a multi-step call, a single step, an evaluation, a
forward pass and a scoring call.  The suite runs every second configuration of the probe's list (minibatches of up to 300 cells included) (both wide widths of every family among
them); the whole list: `python tools/dev/matrix_probe.py` on the GPU box (profiles/r06_matrix_probe.txt)."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location("matrix_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dev", "matrix_probe.py"))
matrix_probe = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(matrix_probe)

CASES = matrix_probe.configurations()[::2]


@pytest.fixture(scope="module")
def Engine():
  from sisua_amd import build
  build.build(verbose=False)
  from sisua_amd.engine import Engine
  return Engine


def test_the_probes_list_covers_every_family_at_every_width():
  seen = {(c["model"], c["n_genes"]) for c in CASES}
  assert len(CASES) == 36 and len(seen) >= 24
  assert {c["model"] for c in CASES} == {"vae", "dca", "scvi", "sisua", "scale", "fvae"}
  assert len({matrix_probe.name_of(c) for c in matrix_probe.configurations()}) == 72 and max(c["B"] for c in CASES) == 300


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[matrix_probe.name_of(c) for c in CASES])
def test_configuration_against_the_oracle_and_across_the_stores(Engine, case):
  problems = matrix_probe.run_one(case)
  assert not problems, problems
