// smx_dgemm.hip -- mid-size dense products with a deep contraction axis (FactorVAE's discriminator: 256 x 1024 x 1024,
// sisua/models/fvae.py:9-18) from bf16 MFMAs on three-way split operands (smx_device.h), K split over the waves of a
// workgroup that owns one 32 x 32 tile:
//
//   C = A Bm      A [M][K] (k contiguous), Bm [K][N] (n contiguous)      a layer's forward product (+ bias + leaky ReLU)
//   C = A Bm^T    A [M][K], Bm [N][K] (k contiguous)                      its input gradient (* the activation's derivative)
//
// The LDS-tiled kernel (smx_gemm.hip) walks K in serial load -> LDS -> barrier -> MFMA rounds, one tile of loads in flight per
// workgroup, with f32 MFMAs that hold the vector pipe: 11-14 us for 0.54 GFLOP with 256 workgroups (a quarter of the MFMA
// rate).  Here:
//   * a workgroup owns one 32 x 32 tile of C and splits K over its 8 waves, 32 k per wave and round; the waves run their
//     rounds independently (no workgroup barrier until the partial tiles meet);
//   * a k-contiguous operand is read COALESCED -- 8 lanes per 128-byte line of a row, 4 loads of 16 bytes per lane and round
//     -- and turned into MFMA operand vectors (row i, 8 consecutive k per lane) through a wave-private LDS tile.  (Each lane
//     reading its own row's line in 16-byte pieces, the form of smx_headbwd.hip role 1, looked simpler and measured no faster
//     than the kernel it replaced: 64 lines per load instruction, 8 instructions per line, 16 waves per CU -- the lines do not
//     survive in the 32 KB L1 between their pieces and every piece goes to L2 again);
//   * an n-contiguous Bm needs no such turn: lane i reads column n0 + i of 8 consecutive rows, coalesced over the lanes
//     (raw buffer loads: one lane offset + a scalar row offset; rows beyond K read as 0);
//   * split three ways in registers, 2 steps x 6 bf16 MFMAs per round and wave;
//   * the 8 partial tiles meet in LDS (over the operand tiles), every wave finishes two accumulator registers: bias,
//     activation (or its derivative from the sign of the forward output), coalesced stores;
//   * blocks 8 apart share an XCD: they take the row tiles of ONE column tile, so a Bm tile is fetched into one L2 only.
#include <stdlib.h>

#include "smx_internal.h"
#include "smx_panel.h"
#include "smx_dgemm.h"
#include "../../include/sisua_hip.h"

namespace smx {

template <int B_KC>
__global__ __launch_bounds__(512) void dgemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[SMX_DG_SMEM_FLOATS(B_KC)];
  dgemm_body<B_KC>(g, (int)blockIdx.x, smem);
}

bool dgemm_supported(const GemmArgs& g) {
  return !g.a_kmajor && !g.use_xform && g.epi == 0 && !g.colsum && !g.sq_part && g.split_k <= 1 && g.M > 0 && g.N % 32 == 0 && g.K % 32 == 0 &&
         g.K >= 512 && (long)g.K * g.ldb * 4 < 0xFFFFFFFFL && (g.lda % 4) == 0 && (!g.b_nmajor || (g.ldb % 4) == 0) && g.A && g.B && g.C &&
         (g.act != 2 || (g.act_out && g.act_ld >= g.N));
}

int launch_dgemm(hipStream_t st, const GemmArgs& g) {
  if (!dgemm_supported(g)) { set_error("dgemm: unsupported problem"); return SMX_ERR_INVALID; }
  const int n_mt = (g.M + 31) / 32, n_nt = g.N / 32;
  dim3 grid((unsigned)(n_mt * ((n_nt + 7) / 8 * 8)));
  if (g.b_nmajor) hipLaunchKernelGGL(dgemm_kernel<1>, grid, dim3(512), 0, st, g);
  else hipLaunchKernelGGL(dgemm_kernel<0>, grid, dim3(512), 0, st, g);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx
