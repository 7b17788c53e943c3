"""One wide product (M=128, N=60000, K=128; the C5-width output head) launched repeatedly: target of PMC passes."""
import os, sys
os.environ.setdefault("SMX_TUNING", "kgemm_reps=50")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sisua_amd import engine
rng = np.random.default_rng(0)
M, N, K = 128, int(os.environ.get("PMC_N", "60000")), 128
A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((K, N)).astype(np.float32)
engine.k_gemm(A, B, split_k=1, tile=int(os.environ.get("PMC_TILE", "1")))
