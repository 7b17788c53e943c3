// smx_panel.h -- weight gradients of a WIDE gene panel whose contraction axis is the minibatch (K = B cells), one workgroup
// per 32-entry tile of the wide axis with EVERY tile of the narrow axis (SURVEY.md 8 rows a-16 / a-17 at BASELINE.json
// configs[4]: 20 000 genes, 128 cells per step):
//
//   role 0   C[wide][n]  = sum_c big[c][wide] S[c][n]     the first encoder layer: big = log1p of the gathered count rows,
//                                                          S = d pre-activation [cells][H]
//   role 1   C[h][wide]  = sum_c S[c][h] big[c][wide]     the output head: big = dP (one plane per workgroup), S = the decoder's
//                                                          output d [cells][H]; + db = column sums of big
//
// What the 32 x 32-tile kernels of smx_headbwd.hip leave on the table at this width is vector-pipe time, not arithmetic: a
// workgroup per (wide tile, narrow tile) transforms and splits the same panel elements once per narrow tile (4x at H = 128),
// gives each of its 8 waves ONE MFMA step and then sends eight partial tiles through LDS -- ~250 vector instructions per wave
// for two outputs per lane (21.7 us for the encoder's 0.66 GFLOP, 26.4 us for the head's 1.97 GFLOP at 128 x 20 000).  Here
//   * the panel tile [128 cells][32 entries] is loaded ONCE: 8 consecutive cells of one entry per lane (coalesced over the
//     entries), transformed, split three ways into bf16 (smx_device.h) and left in LDS as MFMA operand vectors (16 B per lane
//     and term: the readers' ds_read_b128 are conflict-free);
//   * 8 waves = 4 narrow tiles x 2 halves of the cells: a wave's narrow operand (64 cells x 32 columns of S, the same for every
//     workgroup: L2) goes straight into registers and is split while the panel loads are in flight; 4 MFMA steps x 6 products;
//   * the two halves meet in LDS, every wave finishes half a tile: 8 registers per lane, stores coalesced along the output's
//     contiguous axis, sum of squares for clipnorm (8 slots per workgroup), column sums for the bias gradients.
// More than 128 cells: the same in chunks of 128, accumulating.
#ifndef SMX_PANEL_H_
#define SMX_PANEL_H_
#include "smx_device.h"
#include "smx_internal.h"

namespace smx {

#define SMX_PANEL_SMEM_FLOATS (6144 + 4096 + 512 + 512)   // operand image 24 KB | exchange 16 KB | panel column sums | S column sums

// a raw buffer descriptor over [p, p + bytes): loads take ONE 32-bit lane offset + a scalar offset per instruction (no 64-bit
// address pair per load in vector registers -- 40 loads in flight per lane here)
__device__ inline __amdgpu_buffer_rsrc_t panel_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : bytes), 0x00020000);
}

// MODE: the panel is 0 a float32 [cells][wide] buffer of this step (no gather); 1 rows of the resident float32 count store
// gathered by row id; 2 the same from the uint16 store.
// A workgroup walks the units first, first + stride, ... (unit = wide tile x plane): its narrow operand is loaded and split
// ONCE (minibatches of at most 128 cells; per chunk otherwise), and the next unit's panel loads are in flight while the
// current one is multiplied, exchanged and stored.
// ONE: the minibatch is at most 128 cells (one chunk; a launch-time choice, so that neither form carries the other's registers)
template <int ROLE, int MODE, int ONE>
__device__ inline void panel_body(const PanelProblem& P, const int first, const int stride, float* smem) {
  preload(P.big, P.ld_big, P.big_mode, P.log1p, P.rows, P.sub_stride, P.n_sub, P.S, P.ldS, P.n_st, P.out, P.ld_out, P.big_colsum, P.s_colsum, P.sq_part, P.n_wt);   // (one batch of scalar loads: smx_device.h)
  smx_bf16x8* img = reinterpret_cast<smx_bf16x8*>(smem);   // [3 terms][16 cell blocks][32 entries]
  float* ex = smem + 6144;                                  // [2 senders][4 narrow tiles][8 registers][64 lanes]
  float* cs = smem + 10240;                                 // [16 cell blocks][32 entries]
  float* scs = smem + 10752;                                // [8 waves][64 lanes]
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (w in a scalar register: uniform branches)
  const int i = lane & 31, hh = lane >> 5;
  const int t = w & 3, kh = w >> 2;
  const bool tile_on = t < P.n_st;   // wave-uniform: narrow axes under 128 leave waves without a tile (they still load the panel)
  const int cb = 2 * w + hh;         // this lane's cell block of the panel tile
  const int n_units = P.n_wt * P.n_sub;
  constexpr bool one_chunk = ONE != 0;
  const bool want_ss = P.s_colsum && first == 0;     // block-uniform: the workgroup of unit 0 leaves the narrow operand's column sums
  float ssum = 0.f;
  const __amdgpu_buffer_rsrc_t rs = panel_rsrc(P.S, (long)P.B * P.ldS * 4);
  const int vo_s = (8 * hh * P.ldS + 32 * t + i) * 4;
  const int vo_b = (int)(8 * hh * P.ld_big + i) * 4;

  // the narrow operand of chunk kc: 64 cells x 32 columns per wave, straight into registers, split three ways.  Its descriptor
  // ends with the minibatch's last row, and a raw buffer load beyond num_records returns 0 (voffset + soffset is what the
  // hardware checks: tools/bufoob.hip): a ragged minibatch needs neither clamping nor masking
  Split8 ss[4];
  auto load_narrow = [&](int kc, bool count) {
    if (!tile_on) return;
    float sv[4][8];
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int s = 0; s < 8; ++s)
        sv[st][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo_s, (kc + 64 * kh + 16 * st + s) * P.ldS * 4, 0));
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      if (count) {
#pragma unroll
        for (int s = 0; s < 8; ++s) ssum += sv[st][s];
      }
      ss[st] = split3x8(sv[st]);
    }
  };
  // this lane's 8 cells of one entry of the panel tile of (unit, chunk): issued early, consumed by stage()
  // the lane's 8 gathered rows, once (one chunk): read inside load_panel they are not hoisted -- the stores of the loop may alias them for all
  // the compiler knows -- and every unit's panel loads then wait for a round trip of row ids first
  long rowoff[8];
  if (MODE && one_chunk) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {   // (no row ids: the sparse store's dense tile of this minibatch -- cell c is row c)
      const int c = min(8 * cb + s, P.B - 1);
      rowoff[s] = (long)(P.rows ? P.rows[c] : c) * P.ld_big;
    }
  }
  auto load_panel = [&](int unit, int kc, float (&bv)[8]) {
    const int wt = P.n_sub == 1 ? unit : unit / P.n_sub, sub = unit - wt * P.n_sub;   // (one plane: no integer division per unit)
    const long col0 = (long)sub * P.sub_stride + wt * 32;
    if (MODE) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int c = min(kc + 8 * cb + s, P.B - 1);
        const long ro = one_chunk ? rowoff[s] : (long)(P.rows ? P.rows[c] : c) * P.ld_big;
        if (MODE == 2) bv[s] = (float)reinterpret_cast<const uint16_t*>(P.big)[ro + col0 + i];
        else bv[s] = reinterpret_cast<const float*>(P.big)[ro + col0 + i];
      }
    } else {
      // (a panel without gather is this step's [cells][wide] buffer: byte offsets below 2^32)
      const __amdgpu_buffer_rsrc_t rb = panel_rsrc(reinterpret_cast<const float*>(P.big) + col0, ((long)P.B * P.ld_big - col0) * 4);
#pragma unroll
      for (int s = 0; s < 8; ++s)
        bv[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, vo_b, (int)((kc + 16 * w + s) * P.ld_big * 4), 0));
    }
  };

  if (first >= n_units) return;
  float bv[8];
  load_panel(first, 0, bv);
  if (one_chunk) load_narrow(0, want_ss);
  for (int unit = first; unit < n_units; unit += stride) {
    const int wt = P.n_sub == 1 ? unit : unit / P.n_sub, sub = unit - wt * P.n_sub;   // (one plane: no integer division per unit)
    const int w0 = wt * 32;
    const long boff = (long)sub * P.sub_stride + w0 + i;
    smx_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    for (int kc = 0; kc < (ONE ? 1 : P.B); kc += 128) {
      if (!one_chunk) load_narrow(kc, want_ss && unit == first);
      if (MODE) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          float v = bv[s];
          if (P.log1p) v = log1p_count(v);
          bv[s] = (kc + 8 * cb + s < P.B) ? v : 0.f;   // K is the ragged minibatch axis (the gathered row of a cell beyond it was a valid one)
        }
      }
      if (P.big_colsum) {
#pragma unroll
        for (int s = 0; s < 8; ++s) bsum += bv[s];
      }
      const Split8 sb = split3x8(bv);
      if (kc) __syncthreads();   // the previous chunk's image has been read (the previous unit's: its exchange barrier below)
      img[(0 * 16 + cb) * 32 + i] = sb.t0;
      img[(1 * 16 + cb) * 32 + i] = sb.t1;
      img[(2 * 16 + cb) * 32 + i] = sb.t2;
      // the next panel tile's loads go out now: this chunk's next one, or the next unit's first
      {
        const bool more = kc + 128 < P.B;
        const int nu = more ? unit : unit + stride;
        if (nu < n_units) load_panel(nu, more ? kc + 128 : 0, bv);
      }
      __syncthreads();
      if (tile_on) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const int cbr = 8 * kh + 2 * st + hh;   // cells kc + 64 kh + 16 st + 8 hh + s: the narrow operand's of this step
          Split8 bb;
          bb.t0 = img[(0 * 16 + cbr) * 32 + i];
          bb.t1 = img[(1 * 16 + cbr) * 32 + i];
          bb.t2 = img[(2 * 16 + cbr) * 32 + i];
          acc = ROLE == 0 ? mfma_bf16x3(bb, ss[st], acc) : mfma_bf16x3(ss[st], bb, acc);
        }
      }
    }
    // ---- the two halves of the cells meet: wave (t, kh) hands registers 8 (1 - kh) .. + 7 over and finishes 8 kh .. + 7
    // (kh is wave-uniform: two straight-line copies under a scalar branch, never a run-time index into the accumulator) ----
    if (tile_on) {
      if (kh) {
#pragma unroll
        for (int r = 0; r < 8; ++r) ex[((4 + t) * 8 + r) * 64 + lane] = acc[r];
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) ex[(t * 8 + r) * 64 + lane] = acc[8 + r];
      }
    }
    if (P.big_colsum) cs[cb * 32 + i] = bsum;
    __syncthreads();   // (also: every wave is done with this unit's image -- the next unit may overwrite it)
    float sq = 0.f;
    auto put = [&](int rr, float v) {
      const int row = (rr & 3) + 8 * (rr >> 2) + 4 * hh;   // accumulator register rr of a 32 x 32 tile is this row, column i
      if (ROLE == 0) P.out[(long)(w0 + row) * P.ld_out + 32 * t + i] = v;
      else P.out[(long)(32 * t + row) * P.ld_out + boff] = v;   // (padded rows and entries are zero by construction)
      sq += v * v;
    };
    if (tile_on) {
      if (kh) {
#pragma unroll
        for (int r = 0; r < 8; ++r) put(8 + r, ex[(t * 8 + r) * 64 + lane] + acc[8 + r]);   // cells 0..63 first, whichever wave adds
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) put(r, acc[r] + ex[((4 + t) * 8 + r) * 64 + lane]);
      }
    }
    if (P.sq_part) {
      sq = wave_sum(sq);
      if (lane == 0) P.sq_part[(long)unit * 8 + w] = sq;
    }
    if (P.big_colsum && w == 0 && lane < 32) {
      float c = 0.f;
#pragma unroll
      for (int b = 0; b < 16; ++b) c += cs[b * 32 + lane];
      P.big_colsum[(long)sub * P.sub_stride + w0 + lane] = c;
    }
    // (the exchange and the column sums of this unit are read before anyone can pass the next unit's image barrier)
  }
  if (want_ss) {
    scs[w * 64 + lane] = ssum;
    __syncthreads();
    if (tile_on && kh == 0 && lane < 32)
      P.s_colsum[32 * t + lane] = (scs[t * 64 + lane] + scs[t * 64 + 32 + lane]) + (scs[(t + 4) * 64 + lane] + scs[(t + 4) * 64 + 32 + lane]);
  }
}

}  // namespace smx
#endif  // SMX_PANEL_H_
