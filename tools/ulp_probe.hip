// ulp_probe.hip -- accuracy of the single-instruction transcendental forms the kernels use (smx_device.h) against
// float64 on the device: max and mean relative error of v_log_f32 (as ln), v_exp_f32, v_rcp_f32, v_sqrt_f32 over the
// argument ranges that occur in the step.   hipcc --offload-arch=gfx950 -O3 tools/ulp_probe.hip -o tools/ulp_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

__global__ void probe(int kind, float lo, float hi, int n, double* max_rel, double* sum_rel, double* max_abs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // log-uniform sweep of [lo, hi]
  const float x = lo * powf(hi / lo, (float)i / (float)(n - 1));
  float got; double ref;
  switch (kind) {
    case 0: got = __builtin_amdgcn_logf(x) * 0.6931471805599453f; ref = log((double)x); break;
    case 1: got = __expf(-x); ref = exp(-(double)x); break;
    case 2: got = __builtin_amdgcn_rcpf(x); ref = 1.0 / (double)x; break;
    case 3: got = __builtin_amdgcn_sqrtf(x); ref = sqrt((double)x); break;
    case 4: got = __builtin_amdgcn_logf(1.0f + x) * 0.6931471805599453f; ref = log1p((double)x); break;   // log(1 + e)
    default: got = __logf(x); ref = log((double)x); break;
  }
  const double ab = fabs((double)got - ref), rel = ref != 0.0 ? ab / fabs(ref) : ab;
  atomicMax((unsigned long long*)max_rel, (unsigned long long)__double_as_longlong(rel));
  atomicMax((unsigned long long*)max_abs, (unsigned long long)__double_as_longlong(ab));
  atomicAdd(sum_rel, rel);
}

int main() {
  double* d; hipMalloc(&d, 3 * sizeof(double));
  struct { int kind; float lo, hi; const char* name; } cases[] = {
      {0, 1e-30f, 1e30f, "ln x      = v_log * ln2, x in [1e-30, 1e30]"}, {0, 0.5f, 2.0f, "ln x      near 1, x in [0.5, 2]"},
      {0, 0.97f, 1.03f, "ln x      x in [0.97, 1.03]"},
      {4, 0.03125f, 1.0f, "ln(1 + e) e in [1/32, 1] (log1p_small above its series)"},
      {1, 1e-6f, 80.f, "exp(-x)   = v_exp(x log2e), x in [1e-6, 80]"}, {2, 1e-30f, 1e30f, "1 / x     = v_rcp"},
      {3, 1e-30f, 1e30f, "sqrt x    = v_sqrt"}, {5, 0.5f, 2.0f, "__logf    near 1 (the 14-instruction form)"}};
  const int n = 1 << 22;
  for (auto& c : cases) {
    hipMemset(d, 0, 3 * sizeof(double));
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, c.kind, c.lo, c.hi, n, d, d + 1, d + 2);
    double h[3]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s max rel %.3e  mean rel %.3e  max abs %.3e\n", c.name, h[0], h[1] / n, h[2]);
  }
  return 0;
}
