// Host <-> device transfer rates that shape predict(): pinned vs pageable, contiguous vs pitched, pin / register cost.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const size_t B = 128, G = 1998, Gp = 2016, k = 3;
  const size_t n = B * k * Gp;   // one batch of planes
  float *d, *pin, *pin2; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
  CK(hipHostMalloc(&pin, n * 4)); CK(hipHostMalloc(&pin2, n * 4));
  std::vector<float> page(n), page2(n);
  hipStream_t st; CK(hipStreamCreate(&st));
  auto rate = [&](const char* what, auto fn, double bytes) {
    fn(); double t = now(); const int R = 20; for (int i = 0; i < R; ++i) fn(); t = (now() - t) / R;
    printf("%-46s %8.1f us  %6.2f GB/s\n", what, t * 1e6, bytes / t / 1e9);
  };
  rate("D2H pinned contiguous 3.1 MB", [&] { hipMemcpyAsync(pin, d, n * 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }, n * 4.0);
  rate("D2H pageable contiguous 3.1 MB", [&] { hipMemcpy(page.data(), d, n * 4, hipMemcpyDeviceToHost); }, n * 4.0);
  rate("D2H pageable pitched (k*B rows of G)", [&] { hipMemcpy2D(page.data(), G * 4, d, Gp * 4, G * 4, k * B, hipMemcpyDeviceToHost); }, k * B * G * 4.0);
  rate("D2H pinned pitched (k*B rows of G)", [&] { hipMemcpy2DAsync(pin, G * 4, d, Gp * 4, G * 4, k * B, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }, k * B * G * 4.0);
  rate("H2D pinned contiguous 1.0 MB", [&] { hipMemcpyAsync(d, pin, B * Gp * 4, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); }, B * Gp * 4.0);
  rate("H2D pageable contiguous 1.0 MB", [&] { hipMemcpy(d, page.data(), B * Gp * 4, hipMemcpyHostToDevice); }, B * Gp * 4.0);
  rate("H2D pageable pitched (B rows of G)", [&] { hipMemcpy2D(d, Gp * 4, page.data(), G * 4, G * 4, B, hipMemcpyHostToDevice); }, B * G * 4.0);
  rate("host memcpy pinned -> pageable 3.1 MB", [&] { memcpy(page2.data(), pin, n * 4); }, n * 4.0);
  rate("host memcpy rows (k*B rows of G) pinned -> pageable", [&] { for (size_t r = 0; r < k * B; ++r) memcpy(page2.data() + r * G, pin + r * Gp, G * 4); }, k * B * G * 4.0);
  for (size_t mb : {8, 32, 128}) {
    double t = now(); void* p; CK(hipHostMalloc(&p, mb << 20)); double t1 = now(); CK(hipHostFree(p)); double t2 = now();
    printf("hipHostMalloc %zu MB: %.2f ms, free %.2f ms\n", mb, (t1 - t) * 1e3, (t2 - t1) * 1e3);
    std::vector<char> v(mb << 20, 1);
    t = now(); CK(hipHostRegister(v.data(), mb << 20, hipHostRegisterDefault)); t1 = now(); CK(hipHostUnregister(v.data())); t2 = now();
    printf("hipHostRegister %zu MB: %.2f ms, unregister %.2f ms\n", mb, (t1 - t) * 1e3, (t2 - t1) * 1e3);
  }
  return 0;
}
