"""Tile / split-K sweep of the dense products of one C2 step (diagnostic; knob kgemm_reps prints timings)."""
import os, sys
os.environ.setdefault("SMX_TUNING", "kgemm_reps=200")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sisua_amd import engine

rng = np.random.default_rng(0)
G = int(os.environ.get("SWEEP_G", "1998"))
B, H, D, GP = int(os.environ.get("SWEEP_B", "128")), 128, 32, 3 * G
shapes = [
  ("lat_fwd  h[B,H] W[H,2D]", (B, H), (H, 2 * D), False, False, [1]),
  ("dec_fwd  z[B,D] W[D,H]", (B, D), (D, H), False, False, [1]),
  ("out_fwd  h[B,H] W[H,kG]", (B, H), (H, GP), False, False, [1]),
  ("out_dW   h^T[B,H] dP[B,kG]", (B, H), (B, GP), True, False, [1]),
  ("out_dH   dP[B,kG] W^T[H,kG]", (B, GP), (H, GP), False, True, [int(v) for v in os.environ.get("SWEEP_SPLITS", "8,16,24,32,47").split(",")]),
  ("enc_dW   x^T[B,G] dpre[B,H]", (B, G), (B, H), True, False, [1]),
  ("mid_dW   h^T[B,H] d[B,2D]", (B, H), (B, 2 * D), True, False, [1]),
  ("mid_dX   d[B,H] W^T[D,H]", (B, H), (D, H), False, True, [1]),
]
for name, sa, sb, ta, tb, splits in shapes:
  A = rng.standard_normal(sa).astype(np.float32); Bm = rng.standard_normal(sb).astype(np.float32)
  print("##", name, file=sys.stderr, flush=True)
  for s in splits:
    for tile in (1, 2, 3, 4, 5):
      try:
        engine.k_gemm(A, Bm, trans_a=ta, trans_b=tb, split_k=s, tile=tile)
      except Exception as e:
        print("  tile", tile, "split", s, "->", str(e)[:80], file=sys.stderr, flush=True)
