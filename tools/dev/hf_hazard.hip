// hf_hazard.hip -- run ISA-level variants of head_fused_kernel (hand-edited `hipcc -S` output, assembled into code objects by
// tools/dev/hf_hazard.sh) on the same inputs, many launches each, and compare what they leave bit for bit with the first code
// object's first launch.  Used to find the wait states a timing-dependent wrong result goes away with.
//   hf_hazard <LK 0..3> <u16 0|1> <launches> ref.hsaco variant.hsaco ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../sisua_amd/csrc/smx_internal.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: hf_hazard LK u16 launches ref.hsaco [variant.hsaco ...]\n"); return 1; }
  const int LK = atoi(argv[1]), u16 = atoi(argv[2]), launches = atoi(argv[3]);
  const int NP = (LK == 1 || LK == 3) ? 3 : 2, B = 128, G = 20000, Gp = 20000, H = 128;
  const int units = Gp / 16, per_wg = (units + 255) / 256, grid = (units + per_wg - 1) / per_wg;
  std::mt19937 rng(7);
  std::normal_distribution<float> nrm;
  std::uniform_real_distribution<float> uni;
  std::poisson_distribution<int> poi(3.0);
  std::vector<float> hD((size_t)B * H), hW((size_t)H * NP * Gp), hb((size_t)NP * Gp), hXf((size_t)B * Gp);
  std::vector<uint16_t> hX((size_t)B * Gp);
  for (auto& v : hD) v = std::max(nrm(rng), 0.f);
  for (auto& v : hW) v = 0.08f * nrm(rng);
  for (auto& v : hb) v = 0.3f * nrm(rng);
  for (size_t i = 0; i < hX.size(); ++i) { const int c = uni(rng) < 0.3f ? poi(rng) : 0; hX[i] = (uint16_t)c; hXf[i] = (float)c; }
  float *dD, *dW, *db_, *dGW, *dGb, *dPart, *dLl, *dSq; void *dX, *dTab;
  CK(hipMalloc(&dD, hD.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db_, hb.size() * 4)); CK(hipMalloc(&dX, hX.size() * 4));
  CK(hipMalloc(&dGW, hW.size() * 4)); CK(hipMalloc(&dGb, hb.size() * 4)); CK(hipMalloc(&dPart, (size_t)grid * B * H * 4)); CK(hipMalloc(&dLl, (size_t)B * grid * 4));
  CK(hipMalloc(&dSq, (size_t)grid * 8 * 4)); CK(hipMalloc(&dTab, SMX_HEAD_FUSED_TAB_BYTES));
  CK(hipMemcpy(dD, hD.data(), hD.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db_, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  if (u16) CK(hipMemcpy(dX, hX.data(), hX.size() * 2, hipMemcpyHostToDevice)); else CK(hipMemcpy(dX, hXf.data(), hXf.size() * 4, hipMemcpyHostToDevice));
  smx::HeadFusedArgs a;
  a.D = dD; a.ldd = H; a.W = dW; a.ldw = (long)NP * Gp; a.bias = db_; a.X = dX; a.ldx = Gp; a.x_u16 = u16;
  a.dW = dGW; a.db = dGb; a.part = dPart; a.slab_stride = (long)B * H; a.llk_part = dLl; a.sq_part = dSq; a.dtab = dTab;
  a.B = B; a.G = G; a.Gp = Gp; a.likelihood = LK; a.grad_scale = -1.f / B; a.n_gt = units; a.per_wg = per_wg; a.n_chunks = grid;
  const std::string name = "_ZN3smx17head_fused_kernelILi" + std::to_string(LK) + "ELi" + std::to_string(u16) + "ELi0EEEvNS_13HeadFusedArgsE";
  const unsigned lds = 4 * 3 * NP * 4096 + 2 * 8 * 16 * NP * 4;
  std::vector<float> ref_b(hb.size()), ref_w(hW.size()), ref_p((size_t)grid * B * H), got_b(hb.size()), got_w(hW.size()), got_p(ref_p.size());
  int rc = 0;
  for (int v = 4; v < argc; ++v) {
    hipModule_t mod; hipFunction_t fn;
    CK(hipModuleLoad(&mod, argv[v])); CK(hipModuleGetFunction(&fn, mod, name.c_str()));
    size_t sz = sizeof(a);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    long bad_launches = 0, bad_genes = 0, hist_pos[16] = {0}, hist_wave[8] = {0}, hist_lanegroup[4] = {0};
    bool full_diff = false;
    for (int it = 0; it < launches; ++it) {
      CK(hipMemsetAsync(dGb, 0xFF, hb.size() * 4, nullptr));
      CK(hipModuleLaunchKernel(fn, grid, 1, 1, 512, 1, 1, lds, nullptr, nullptr, cfg));
      CK(hipMemcpy(got_b.data(), dGb, hb.size() * 4, hipMemcpyDeviceToHost));
      if (v == 4 && it == 0) {
        ref_b = got_b;
        CK(hipMemcpy(ref_w.data(), dGW, hW.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ref_p.data(), dPart, ref_p.size() * 4, hipMemcpyDeviceToHost));
        continue;
      }
      if (!memcmp(got_b.data(), ref_b.data(), hb.size() * 4)) continue;
      ++bad_launches;
      CK(hipMemcpy(got_p.data(), dPart, got_p.size() * 4, hipMemcpyDeviceToHost));
      for (int p = 0; p < NP; ++p)
        for (int n = 0; n < Gp; ++n) {
          if (!memcmp(&got_b[(size_t)p * Gp + n], &ref_b[(size_t)p * Gp + n], 4)) continue;
          ++bad_genes; ++hist_pos[n & 15];
          const int wg = (n / 16) / per_wg;
          // the cells whose rows of this workgroup's d d slab moved: the cell's wave is cell / 16
          for (int c = 0; c < B; ++c)
            if (memcmp(&got_p[((size_t)wg * B + c) * H], &ref_p[((size_t)wg * B + c) * H], H * 4)) {
              ++hist_wave[c >> 4];
              if (bad_genes <= 6) printf("    launch %d plane %d gene %d (unit position %d, workgroup %d) cell %d: db %.9g instead of %.9g, count %d\n", it, p, n, n & 15, wg, c,
                                         got_b[(size_t)p * Gp + n], ref_b[(size_t)p * Gp + n], (int)hX[(size_t)c * Gp + n]);
            }
          (void)hist_lanegroup;
        }
    }
    if (bad_launches && getenv("HF_TABLE")) {   // workgroup 0 of the last launch: counts at unit positions 14, 15 of the cells of waves 4-7; '*' = the cell's slab row moved, '!' = the gene's db moved
      for (int u = 0; u < per_wg; ++u) printf("  unit %d (gene %d): db %s\n", u, 16 * u + 15, memcmp(&got_b[16 * u + 15], &ref_b[16 * u + 15], 4) ? "MOVED" : "same");
      for (int c = 64; c < B; ++c) {
        printf("  cell %3d%s:", c, memcmp(&got_p[(size_t)c * H], &ref_p[(size_t)c * H], H * 4) ? "*" : " ");
        for (int u = 0; u < per_wg; ++u) printf("  (%d,%d)", (int)hX[(size_t)c * Gp + 16 * u + 14], (int)hX[(size_t)c * Gp + 16 * u + 15]);
        printf("\n");
      }
    }
    if (v > 4 || launches > 1) {   // the last launch's dW and slabs in full
      CK(hipMemcpy(got_w.data(), dGW, hW.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(got_p.data(), dPart, got_p.size() * 4, hipMemcpyDeviceToHost));
      full_diff = memcmp(got_w.data(), ref_w.data(), hW.size() * 4) || memcmp(got_p.data(), ref_p.data(), got_p.size() * 4);
    }
    printf("%s: %ld of %d launches differ from the reference (%ld gene entries of db; by position in the unit:", argv[v], bad_launches, launches, bad_genes);
    for (int i = 0; i < 16; ++i) printf(" %ld", hist_pos[i]);
    printf("; by wave of the cell:");
    for (int i = 0; i < 8; ++i) printf(" %ld", hist_wave[i]);
    printf(")%s\n", full_diff ? "  [last launch: dW or d d slabs differ]" : "");
    fflush(stdout);
    if (bad_launches) rc = 2;
    CK(hipModuleUnload(mod));
  }
  return rc;
}
