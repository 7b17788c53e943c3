"""Parity of single HIP kernels against the oracle, through the C-ABI
(smx_k_* entry points).  Needs a real MI355X."""
import numpy as np
import pytest

from oracle import sisua_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
  from sisua_amd import build
  build.build(verbose=False)
  from sisua_amd import engine
  return engine


def test_philox_dropout_bit_exact_and_normals(eng):
  ids = np.array([0, 1, 5, 4096, 2 ** 31 - 1, 123456789], dtype=np.int64)
  for stream, step, sample, p, width in ((16, 0, 0, 0.1, 130), (48, 77, 3, 0.3, 64), (0, 2 ** 31, 0, 0.25, 37)):
    mult, nrm = eng.k_noise(8, stream, step, ids, width, p=p, sample=sample)
    ref_m = so.philox_dropout_mask(8, stream, step, ids, width, p, sample)
    assert np.array_equal(mult, ref_m.astype(np.float32))          # integer-exact decisions
    ref_n = so.philox_normal(8, stream, step, ids, width, sample)
    assert np.allclose(nrm, ref_n, rtol=2e-5, atol=1e-5)   # fast v_log / v_sin / v_cos forms
  # 64-bit seed reaches both key words
  m1, _ = eng.k_noise((7 << 32) | 8, 16, 0, ids, 16, p=0.5)
  assert np.array_equal(m1, so.philox_dropout_mask((7 << 32) | 8, 16, 0, ids, 16, 0.5).astype(np.float32))


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
def test_gemm_all_layouts_and_tiles(eng, ta, tb, tile):
  rng = np.random.default_rng(tile * 7 + ta * 2 + tb)
  for (M, N, K, S) in ((128, 96, 64, 1), (37, 32, 200, 1), (130, 256, 515, 4), (5, 64, 33, 2)):
    if tile == 3 and N % 128:
      continue
    if tile in (2,) and N % 64:
      continue
    A = rng.normal(size=(K, M) if ta else (M, K)).astype(np.float32)
    B = rng.normal(size=(N, K) if tb else (K, N)).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    out = eng.k_gemm(A, B, ta, tb, split_k=S, tile=tile)
    assert np.allclose(out, ref, rtol=2e-5, atol=2e-4 * np.sqrt(K)), (M, N, K, S, np.abs(out - ref).max())


def test_gemm_layout_is_not_transposed(eng):
  # A = I with an asymmetric B catches a swapped C/D register map
  B = np.arange(64 * 96, dtype=np.float32).reshape(64, 96)
  assert np.array_equal(eng.k_gemm(np.eye(64, dtype=np.float32), B), B)


def _edge_grid():
  rng = np.random.default_rng(0)
  x = np.concatenate([np.zeros(24), np.arange(1, 13), rng.integers(1, 200, 20), [181, 1000, 10738, 8, 9]]).astype(np.float32)
  r_log = np.log(np.array([1e-4, 1e-2, 1.0, 7.9, 8.1, 50.0, 1e4]))
  l = np.array([-30.0, -5.0, -0.5, 0.0, 0.7, 6.0, 30.0])
  X, A, L = np.meshgrid(x, r_log, l, indexing="ij")
  n = X.size
  G = 64
  pad = (-n) % G
  def f(v, fill):
    return np.concatenate([v.ravel(), np.full(pad, fill)]).reshape(-1, G).astype(np.float32)
  g = rng.uniform(-6, 6, size=n + pad).reshape(-1, G).astype(np.float32)
  return f(X, 0), f(A, 0), f(L, 0), g


@pytest.mark.parametrize("lk", so.LIKELIHOODS)
def test_count_llk_edge_grid(eng, lk):
  """SURVEY 8c KAT grid: x in {0..12, 181, 1000, 10738}, r in {1e-4..1e4}, |logits| <= 30."""
  x, a, l, g = _edge_grid()
  if lk in ("nbd", "zinbd"):
    l = np.clip(l, -10, 10)       # second plane is the dispersion pre-activation here
    a = np.clip(a, -8, 9.2)
  planes = [a, l, g][: so.n_params_per_gene(lk)]
  llk, grads = eng.k_count_llk(lk, x, np.stack(planes))
  ref_e, ref_g = so.count_llk(x.astype(np.float64), [p.astype(np.float64) for p in planes], lk)
  ref = ref_e.sum(1)
  assert np.allclose(llk, ref, rtol=1e-4, atol=1e-2)
  for i in range(len(planes)):
    err = np.abs(grads[i] - ref_g[i])
    tol = 1e-4 * np.abs(ref_g[i]) + 1e-4 * (1 + np.abs(x))
    assert (err <= tol).all(), (lk, i, err.max(), np.unravel_index(np.argmax(err - tol), err.shape))


@pytest.mark.parametrize("lk", ["nb", "zinb"])
def test_count_llk_tiny_total_count_stays_finite(eng, lk):
  """log total_count down to -69 (r = 1e-30, the kernel's clamp) with counts from 9 to 1e5: the Stirling branch's shift
  prod (r+i)/(x+r+i) is ~1e-42 there, below the smallest normal float, where v_log_f32 returns -inf -- its
  numerator and denominator are kept apart, and the result must agree with the float64 oracle."""
  xs = np.array([9, 50, 181, 1000, 10738, 65535, 100000], np.float32)
  a = np.array([-69.0, -60.0, -46.0, -30.0, -12.0], np.float32)
  l = np.array([-3.0, 0.0, 2.0], np.float32)
  X, A, L = [v.ravel() for v in np.meshgrid(xs, a, l, indexing="ij")]
  n = X.size
  G = 32
  pad = (-n) % G
  f = lambda v: np.concatenate([v, np.zeros(pad, np.float32)]).reshape(-1, G).astype(np.float32)
  x, pa, pl = f(X), f(A), f(L)
  planes = [pa, pl, np.zeros_like(pa)][: so.n_params_per_gene(lk)]
  llk, grads = eng.k_count_llk(lk, x, np.stack(planes))
  ref_e, ref_g = so.count_llk(x.astype(np.float64), [p.astype(np.float64) for p in planes], lk)
  assert np.isfinite(llk).all() and np.isfinite(grads).all()
  assert np.allclose(llk, ref_e.sum(1), rtol=1e-4, atol=1e-2)
  for i in range(len(planes)):
    assert (np.abs(grads[i] - ref_g[i]) <= 1e-4 * np.abs(ref_g[i]) + 1e-4 * (1 + np.abs(x))).all(), (lk, i)


@pytest.mark.parametrize("lk", ["nbd", "zinbd"])
def test_count_llk_direct_mode(eng, lk):
  rng = np.random.default_rng(4)
  B, G = 9, 203   # ragged G: exercises the padded tail
  x = rng.poisson(1.0, size=(B, G)).astype(np.float32) * (rng.uniform(size=(B, G)) < 0.3)
  mu = rng.lognormal(0, 1.5, size=(B, G)).astype(np.float32)
  th = rng.lognormal(0, 1.0, size=(B, G)).astype(np.float32)
  gate = rng.normal(size=(B, G)).astype(np.float32)
  planes = [mu, th, gate][: so.n_params_per_gene(lk)]
  llk, grads = eng.k_count_llk(lk, x, np.stack(planes), direct=True)
  ref_e, ref_g = so.count_llk(x.astype(np.float64), [p.astype(np.float64) for p in planes], lk, direct=True)
  assert np.allclose(llk, ref_e.sum(1), rtol=1e-5, atol=1e-3)
  for i in range(len(planes)):
    assert np.allclose(grads[i], ref_g[i], rtol=2e-4, atol=2e-5)


def test_count_llk_sizes_and_empty_rows(eng):
  rng = np.random.default_rng(5)
  for B, G in ((1, 1), (3, 31), (2, 1024), (2, 1025), (4, 1998), (1, 4100)):
    x = (rng.poisson(2.0, size=(B, G)) * (rng.uniform(size=(B, G)) < 0.07)).astype(np.float32)
    x[0, :] = 0                     # an all-zero cell
    planes = np.stack([rng.normal(size=(B, G)), rng.normal(size=(B, G)), rng.normal(size=(B, G))]).astype(np.float32)
    llk, grads = eng.k_count_llk("zinb", x, planes)
    ref_e, ref_g = so.count_llk(x.astype(np.float64), list(planes.astype(np.float64)), "zinb")
    assert np.allclose(llk, ref_e.sum(1), rtol=1e-5, atol=1e-4)
    assert np.allclose(grads, np.stack(ref_g), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,G,lk", [(260, 1998, "zinb"), (300, 1501, "nbd"), (70, 130000, "zinb"), (66, 126001, "nb")])
def test_count_llk_wide_access_forms(eng, B, G, lk):
  """The loss kernel switches to 8-byte accesses from ~0.4 M elements and 16-byte from ~8 M: same results,
  ragged gene counts included (the full-size property: the sum of the per-cell values equals the sum of the
  elementwise oracle)."""
  rng = np.random.default_rng(B + G)
  x = (rng.poisson(3.0, size=(B, G)) * (rng.uniform(size=(B, G)) < 0.1)).astype(np.float32)
  k = 3 if lk.startswith("zi") else 2
  planes = rng.normal(size=(k, B, G)).astype(np.float32)
  llk, grads = eng.k_count_llk(lk, x, planes)
  ref_e, ref_g = so.count_llk(x.astype(np.float64), list(planes.astype(np.float64)), lk)
  assert np.allclose(llk, ref_e.sum(1), rtol=1e-5, atol=1e-3)
  assert np.allclose(grads, np.stack(ref_g), rtol=1e-4, atol=1e-5)


def test_the_noise_function_is_hiprands_philox_generator():
  """BASELINE north_star: "fused Gaussian reparameterisation sampling from a hiprand state per wavefront".  The kernels evaluate
  Philox4x32-10 as a pure function of (column block, cell id, step, stream | sample) under the model's seed; this IS a
  `hiprandStatePhilox4_32_10_t` set up on the fly -- `hiprand_init(seed, subsequence = (step, stream), offset = 4 (column block, cell id))`
  followed by one `hiprand4()` -- bit for bit, for random counters and the corners, without a state ever being stored or loaded."""
  import ctypes as C
  from sisua_amd import _hip, build
  build.build(verbose=False)
  lib = _hip.require_gpu(0)
  rng = np.random.default_rng(7)
  n = 4096
  c = rng.integers(0, 2**32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
  c[:, 1] &= (1 << 30) - 1                                   # cell ids below 2^30 (the offset 4 * (c0 | c1 << 32) is 64 bits)
  c[:8] = [[0, 0, 0, 0], [1, 0, 0, 0], [0xFFFFFFFF, 0, 0, 0], [0xFFFFFFFF, (1 << 30) - 1, 0, 0], [0, 0, 0xFFFFFFFF, 0xFFFFFFFF],
           [3, 1000, 7, 64], [0, 1, 2, 64 | (5 << 8)], [499, 3380, 299, 48]]
  for seed in (8, 0, 2**63 + 12345):
    ours, theirs = np.empty((n, 4), np.uint32), np.empty((n, 4), np.uint32)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    _hip.check(lib.smx_k_hiprand(C.c_uint64(seed), n, p(np.ascontiguousarray(c)), p(ours), p(theirs)))
    assert np.array_equal(ours, theirs), (seed, np.nonzero((ours != theirs).any(1))[0][:5])
    # ... and it is the function the oracle implements (Random123 known-answer vectors: tests/test_oracle_rng.py)
    ref = so.philox4x32_10(c[:64, 0], c[:64, 1], c[:64, 2], c[:64, 3], np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32))
    assert np.array_equal(np.stack(ref, 1).astype(np.uint32), ours[:64])


@pytest.mark.parametrize("lk,u16", [("nbd", True), ("nbd", False), ("zinbd", True), ("zinbd", False), ("zinb", True), ("nb", True)])
def test_head_fused_is_the_same_over_thousands_of_launches(eng, lk, u16):
  """the two timing-dependent wrong results of head_fused_kernel (a dP term lost in lanes 48-63 of waves 4-7 =
  the last gene of a unit; garbage in zinbd's table of split operands) were hardware behaviour beside OTHER waves' instructions, so a
  single launch rarely shows them.  2000 launches at the configs[4] width, every wave of every workgroup busy, each compared on the
  device bit for bit with the first; the first launch's last-gene-of-a-unit columns against float64."""
  B, G = 128, 20000
  rng = np.random.default_rng(11)
  k = so.n_params_per_gene(lk)
  x = (rng.poisson(3.0, size=(B, G)) * (rng.uniform(size=(B, G)) < 0.3)).astype(np.float32)
  d = np.maximum(rng.normal(size=(B, 128)), 0).astype(np.float32)
  W = (rng.normal(size=(128, k, G)) * 0.08).astype(np.float32)
  bias = (rng.normal(size=(k, G)) * 0.3).astype(np.float32)
  scale = -1.0 / B
  n_differ, first = eng.k_head_fused_stress(lk, x, d, W, bias, launches=2000, grad_scale=scale, u16=u16)
  assert n_differ == 0, (n_differ, first)
  got = eng.k_head_fused(lk, x, d, W, bias, grad_scale=scale, u16=u16)
  last = np.arange(15, G, 16)                                   # gene 15 of every unit: accumulator row 15 = lanes 48-63, element 3
  P = np.einsum("bh,hkg->kbg", d.astype(np.float64), W[:, :, last].astype(np.float64)) + bias[:, last].astype(np.float64)[:, None, :]
  _, ref_g = so.count_llk(x[:, last].astype(np.float64), list(P), lk)
  db = scale * np.stack(ref_g).sum(1)
  hi = scale * np.stack(ref_g)[:, 64:].sum(1)                    # ... of the cells of waves 4-7 alone: what a lost term would move
  err = np.abs(got["db"][:, last] - db)
  assert (err <= 2e-5 * np.abs(db) + 1e-6 * np.abs(hi).max()).all(), (err.max(), np.abs(db).max())


@pytest.mark.parametrize("lk,B,G,u16", [("zinb", 128, 4128, True), ("zinb", 100, 4100, False), ("nb", 128, 4096, False),
                                         ("nbd", 37, 4130, True), ("zinbd", 128, 8000, False), ("zinb", 1, 4096, True),
                                         # more than 128 cells: one launch per 128, the later ones adding their dW / db (round 5)
                                         ("zinb", 256, 4128, True), ("nb", 200, 4100, False), ("zinbd", 129, 4096, True),
                                         # three units per workgroup, every cell row in use (the nbd / uint16 build once lost a term of dP in
                                         # waves 4-7 here: a packed-f32 op_sel beside the partner wave's MFMAs, tools/isa_lint.py rule R1)
                                         ("nbd", 128, 12000, True), ("nb", 128, 12000, True), ("zinbd", 128, 12000, True), ("zinb", 128, 12000, False)])
def test_head_fused_matches_float64(eng, lk, B, G, u16):
  """smx_headfused.hip: output product + likelihood + dW / db / d d in one launch against the float64 arithmetic of the oracle
  (P = d W + b; count_llk; dP = scale * d llk / d P; dW = d^T dP; db = colsum dP; dd = dP W^T) -- ragged minibatches, gene
  counts that are not multiples of 32, both count stores, every likelihood."""
  rng = np.random.default_rng(B * 7 + G)
  k = so.n_params_per_gene(lk)
  x = (rng.poisson(3.0, size=(B, G)) * (rng.uniform(size=(B, G)) < 0.1)).astype(np.float32)
  x[0, :7] = [9, 40, 181, 1000, 10738, 12, 8]     # counts beyond the rising-factorial path
  d = np.maximum(rng.normal(size=(B, 128)), 0).astype(np.float32) * (rng.uniform(size=(B, 128)) < 0.9)
  W = (rng.normal(size=(128, k, G)) * 0.08).astype(np.float32)
  bias = (rng.normal(size=(k, G)) * 0.3).astype(np.float32)
  scale = -1.0 / B
  got = eng.k_head_fused(lk, x, d, W, bias, grad_scale=scale, u16=u16)
  d64, W64 = d.astype(np.float64), W.astype(np.float64)
  P = np.einsum("bh,hkg->kbg", d64, W64) + bias.astype(np.float64)[:, None, :]
  ref_e, ref_g = so.count_llk(x.astype(np.float64), list(P), lk)
  dP = scale * np.stack(ref_g)                                  # [k][B][G]
  dW = np.einsum("bh,kbg->hkg", d64, dP)
  db = dP.sum(1)
  dd = np.einsum("kbg,hkg->bh", dP, W64)
  rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
  assert np.allclose(got["llk"], ref_e.sum(1), rtol=2e-5, atol=2e-2), np.abs(got["llk"] - ref_e.sum(1)).max()
  assert rel(got["dW"], dW) < 2e-5, rel(got["dW"], dW)
  assert rel(got["db"], db) < 2e-5, rel(got["db"], db)
  assert rel(got["dd"], dd) < 2e-5, rel(got["dd"], dd)
  assert np.abs(got["dW"] - dW).max() <= 1e-4 * np.abs(dW).max()
  assert abs(got["sumsq"] - (dW ** 2).sum()) <= 1e-4 * (dW ** 2).sum()
