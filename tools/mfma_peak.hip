// Measures the sustained v_mfma_f32_32x32x2_f32 rate of the device (no memory traffic): the practical ceiling for the fp32 GEMMs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int i = 0; i < 16; ++i) s += acc[a][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
  printf("nacc=%d blocks=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, iters, ms, fl / ms * 1e-9);
}
int main() {
  float* d;
  hipMalloc(&d, 4096 * 256 * 4);
  run<1>(256, 20000, d);
  run<2>(256, 10000, d);
  run<2>(512, 10000, d);
  run<2>(768, 10000, d);
  run<4>(1024, 5000, d);
  run<2>(256, 200000, d);
  return 0;
}
