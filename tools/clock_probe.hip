// What shader clock does an fp32-MFMA-bound kernel run at?  Each workgroup reads the shader-cycle counter (clock64, s_memtime)
// and the constant 100 MHz counter (wall_clock64, s_memrealtime) around an LDS-free MFMA loop; f = d(cycles) / d(wall) * 100 MHz.
// Launched back to back for ~0.3 s to show the clock once power management has settled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 2) void k(long long* out, int iters, float* sink) {
  const long long c0 = clock64(), w0 = wall_clock64();
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = c1 - c0;
    out[blockIdx.x * 2 + 1] = w1 - w0;
  }
}
int main() {
  const int blocks = 512, iters = 18;  // the SRB convolution's shape: 512 workgroups of 8 waves, 18 x 16 MFMAs per wave
  long long* d;
  float* sink;
  (void)hipMalloc(&d, blocks * 2 * sizeof(long long));
  (void)hipMalloc(&sink, 4);
  std::vector<long long> h(blocks * 2);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int phase = 0; phase < 6; ++phase) {
    const int n = phase == 0 ? 1 : 2000;
    (void)hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> f;
    double cyc = 0;
    for (int b = 0; b < blocks; ++b) {
      if (h[2 * b + 1] > 0) f.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
      cyc += (double)h[2 * b];
    }
    std::sort(f.begin(), f.end());
    const double flops = (double)blocks * 8 * iters * 16 * 4096.0;
    printf("phase %d: %4d launches, %.2f us/launch, %.1f TFLOP/s | last launch: shader clock median %.0f MHz (min %.0f, max %.0f), %.0f cycles per workgroup\n",
           phase, n, ms * 1e3 / n, flops * n / ms * 1e-9, f[f.size() / 2], f.front(), f.back(), cyc / blocks);
  }
  return 0;
}
