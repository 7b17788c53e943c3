// bufoob.hip -- does a raw buffer load on gfx950 range-check voffset + soffset against num_records (returning 0 beyond it)?
// (calibration for smx_panel.h, which leans on it for the ragged last chunk of a minibatch)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* p, float* o, int n_rec_bytes, int so) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, n_rec_bytes, 0x00020000);
  o[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, threadIdx.x * 4, so, 0));
}
int main() {
  float *p, *o; hipMalloc(&p, 4096 * 4); hipMalloc(&o, 64 * 4);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i < 1000 ? 1.f : 7.f;
  hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice);
  for (int so : {0, 3800, 3900, 4000, 8000}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o, 1000 * 4, so);
    float r[64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("soffset %d bytes (element %d + lane): lanes", so, so / 4);
    for (int l = 0; l < 64; l += 7) printf(" [%d]=%g", l, r[l]);
    printf("\n");
  }
  return 0;
}
