// Experiment: which part of the GEMM main loop keeps the MFMA pipe idle?  Synthetic loop with the same shape as gemm_kernel
// (8 waves, 128x64x32 tile, 16 MFMAs per wave per tile), pieces switched on one at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 64, BK = 32;
template <bool BARRIER, bool DSWRITE, bool GLOAD, int DIST, int NVALU = 0>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ X, float* out, int ntiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5, wm = wave & 3, wn = wave >> 2;
  const int kq = tid & 7, r0 = tid >> 3;
  for (int i = tid; i < 2 * (BM + BN) * BK; i += 512) lds[i] = i * 1e-6f;
  __syncthreads();
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float4 r[2][3];
  for (int s = 0; s < 2; ++s)
    for (int j = 0; j < 3; ++j) r[s][j] = make_float4(1.f, 2.f, 3.f, 4.f);
  const float* xp = X + ((long)blockIdx.x * 192 + r0) * 64 + kq * 4;  // rows of 64 floats, tile = 192 rows x 32 k
  auto gload = [&](int t, float4 (&rr)[3]) {
    if constexpr (GLOAD) {
#pragma unroll
      for (int j = 0; j < 3; ++j) rr[j] = *reinterpret_cast<const float4*>(xp + (long)j * 64 * 64 + (t & 1) * 32 + (long)(t >> 1) * 0);
    }
  };
  auto dswrite = [&](int buf, const float4 (&rr)[3]) {
    if constexpr (DSWRITE) {
      float* b = lds + buf * (BM + BN) * BK;
      const int sw = (kq ^ (r0 & 7)) * 4;
#pragma unroll
      for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(b + (r0 + 64 * j) * BK + sw) = rr[j];
    }
  };
  unsigned junk = tid;
  auto valu = [&]() {
#pragma unroll
    for (int i = 0; i < NVALU; ++i) junk = junk * 1664525u + (unsigned)i + (junk >> 7);  // 3 dependent VALU ops each
  };
  auto compute = [&](int buf) {
    const float* As = lds + buf * (BM + BN) * BK;
    const float* Bs = As + BM * BK;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      float4 a = *reinterpret_cast<const float4*>(As + (wm * 32 + li) * BK + (((2 * kg + lh) ^ (li & 7)) << 2));
      float4 b = *reinterpret_cast<const float4*>(Bs + (wn * 32 + li) * BK + (((2 * kg + lh) ^ (li & 7)) << 2));
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  };
  if (DIST == 2) { gload(1, r[1]); gload(2, r[0]); } else { gload(1, r[1]); }
  for (int t = 0; t < ntiles; t += 2) {
    valu();
    if (DIST == 2) { dswrite(1, r[1]); gload(t + 3, r[1]); compute(0); }
    else { compute(0); dswrite(1, r[1]); gload(t + 2, r[0]); }
    if (BARRIER) __syncthreads();
    valu();
    if (DIST == 2) { dswrite(0, r[0]); gload(t + 4, r[0]); compute(1); }
    else { compute(1); dswrite(0, r[0]); gload(t + 3, r[1]); }
    if (BARRIER) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 512 + tid] = s + r[0][0].x + r[1][2].y + (float)junk;
}
template <bool BARRIER, bool DSWRITE, bool GLOAD, int DIST, int NVALU = 0>
void run(const char* name, const float* X, float* d, int blocks, bool SHORT = false) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int ntiles = 18 * 20;
  hipLaunchKernelGGL((k<BARRIER, DSWRITE, GLOAD, DIST, NVALU>), dim3(blocks), dim3(512), 0, 0, X, d, 18);
  (void)hipEventRecord(e0);
  if (SHORT) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<BARRIER, DSWRITE, GLOAD, DIST, NVALU>), dim3(blocks), dim3(512), 0, 0, X, d, 18);
  } else {
    hipLaunchKernelGGL((k<BARRIER, DSWRITE, GLOAD, DIST, NVALU>), dim3(blocks), dim3(512), 0, 0, X, d, ntiles);
  }
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 8 * ntiles * 16 * 4096.0;
  printf("%-44s blocks=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, fl / ms * 1e-9);
}
int main() {
  float *d, *X;
  (void)hipMalloc(&d, 2048 * 512 * 4);
  (void)hipMalloc(&X, (size_t)2048 * 192 * 64 * 4);
  (void)hipMemset(X, 0, (size_t)2048 * 192 * 64 * 4);
  for (int blocks : {512, 768, 1024}) {
    run<false, false, false, 1>("lds-read + mfma", X, d, blocks);
    run<true, false, false, 1>("+ barrier", X, d, blocks);
    run<true, true, false, 1>("+ ds_write", X, d, blocks);
    run<true, true, true, 1>("+ global loads, dist 1", X, d, blocks);
    run<true, true, true, 2>("+ global loads, dist 2", X, d, blocks);
    run<true, true, true, 2>("same, 20 launches of 18 tiles", X, d, blocks, true);
    run<true, true, true, 2, 10>("dist 2 + 30 VALU/tile", X, d, blocks);
    run<true, true, true, 2, 20>("dist 2 + 60 VALU/tile", X, d, blocks);
    run<true, true, true, 2, 40>("dist 2 + 120 VALU/tile", X, d, blocks);
    run<true, true, true, 2, 60>("dist 2 + 180 VALU/tile", X, d, blocks);
    run<false, false, false, 1>("lds+mfma only, 20 launches of 18 tiles", X, d, blocks, true);
  }
  return 0;
}
