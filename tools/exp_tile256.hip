// Experiment: would a 256x64x32 tile (two 32x32 accumulators per wave, one 8-wave workgroup per CU, 80 KB of LDS) beat the
// 128x64x32 tile (one accumulator, two workgroups per CU) of gemm_kernel on the SRB convolution's 18-tile K loop?
// Same synthetic loop as exp_mfma_lds.hip (LDS-fed MFMAs, ds_write + global loads two tiles ahead, one barrier per tile).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BN = 64, BK = 32;
template <int MT, bool REAL = false>
__global__ __launch_bounds__(512, (MT == 1 ? 2 : 1)) void k(const float* __restrict__ X, float* out, int ntiles) {
  constexpr int BM = 128 * MT, NA = 2 * MT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5, wm = wave & 3, wn = wave >> 2;
  const int kq = tid & 7, r0 = tid >> 3;
  for (int i = tid; i < 2 * (BM + BN) * BK; i += 512) lds[i] = i * 1e-6f;
  __syncthreads();
  f32x16 acc[MT];
  for (int s = 0; s < MT; ++s)
    for (int i = 0; i < 16; ++i) acc[s][i] = 0.f;
  float4 r[2][NA + 1];
  for (int s = 0; s < 2; ++s)
    for (int j = 0; j < NA + 1; ++j) r[s][j] = make_float4(1.f, 2.f, 3.f, 4.f);
  const float* xp = X + ((long)blockIdx.x * (BM + BN) + r0) * 64 + kq * 4;
  // REAL: the im2col access pattern of the 3x3 convolution on [64,16,64,64] NHWC: tile t = tap t/2, channel half t%2; rows are the
  // workgroup's 128 (256) pixels shifted by the tap (clamped instead of zero-filled), weights from a [64][576] array behind X.
  const long R = 65536;
  const float* wbase = X + R * 64 + (long)r0 * 576 + kq * 4;
  auto gload = [&](int t, float4 (&rr)[NA + 1]) {
    if constexpr (REAL) {
      const int tap = (t >> 1) % 9, half = t & 1;
      const int shift = (tap / 3 - 1) * 64 + (tap % 3 - 1);
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        long m = (long)blockIdx.x * BM + r0 + 64 * j + shift;
        m = m < 0 ? 0 : (m >= R ? R - 1 : m);
        rr[j] = *reinterpret_cast<const float4*>(X + m * 64 + kq * 4 + half * 32);
      }
      rr[NA] = *reinterpret_cast<const float4*>(wbase + tap * 64 + half * 32);
    } else {
#pragma unroll
      for (int j = 0; j < NA + 1; ++j) rr[j] = *reinterpret_cast<const float4*>(xp + (long)j * 64 * 64 + (t & 1) * 32);
    }
  };
  auto dswrite = [&](int buf, const float4 (&rr)[NA + 1]) {
    float* b = lds + buf * (BM + BN) * BK;
    const int sw = (kq ^ (r0 & 7)) * 4;
#pragma unroll
    for (int j = 0; j < NA + 1; ++j) *reinterpret_cast<float4*>(b + (r0 + 64 * j) * BK + sw) = rr[j];
  };
  auto compute = [&](int buf) {
    const float* As = lds + buf * (BM + BN) * BK;
    const float* Bs = As + BM * BK;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      float4 b = *reinterpret_cast<const float4*>(Bs + (wn * 32 + li) * BK + (((2 * kg + lh) ^ (li & 7)) << 2));
      float4 a[MT];
#pragma unroll
      for (int s = 0; s < MT; ++s) a[s] = *reinterpret_cast<const float4*>(As + ((wm * MT + s) * 32 + li) * BK + (((2 * kg + lh) ^ (li & 7)) << 2));
#pragma unroll
      for (int s = 0; s < MT; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].x, b.x, acc[s], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < MT; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].y, b.y, acc[s], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < MT; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].z, b.z, acc[s], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < MT; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].w, b.w, acc[s], 0, 0, 0);
    }
  };
  gload(0, r[0]);
  gload(1, r[1]);
  dswrite(0, r[0]);
  gload(2, r[0]);
  __syncthreads();
  for (int t = 0; t < ntiles; t += 2) {
    dswrite(1, r[1]); gload(t + 3, r[1]); compute(0);
    __syncthreads();
    dswrite(0, r[0]); gload(t + 4, r[0]); compute(1);
    __syncthreads();
  }
  // epilogue like the GEMM's: 16 strided stores per accumulator
  float* o = out + ((long)blockIdx.x * BM + wm * MT * 32) * 64 + wn * 32 + li;
  for (int s = 0; s < MT; ++s)
    for (int i = 0; i < 16; ++i) o[(long)(s * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh) * 64] = acc[s][i] + r[0][0].x + r[1][NA].y;
}
template <int MT, bool REAL = false>
void run(const char* name, const float* X, float* d, int blocks, bool SHORT) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int lds = 2 * (128 * MT + BN) * BK * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MT, REAL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int ntiles = 18 * 20;
  hipLaunchKernelGGL((k<MT, REAL>), dim3(blocks), dim3(512), lds, 0, X, d, 18);
  (void)hipEventRecord(e0);
  if (SHORT) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MT, REAL>), dim3(blocks), dim3(512), lds, 0, X, d, 18);
  } else {
    hipLaunchKernelGGL((k<MT, REAL>), dim3(blocks), dim3(512), lds, 0, X, d, ntiles);
  }
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 8 * ntiles * 16 * MT * 4096.0;
  printf("%-44s blocks=%d: %.3f ms  %.1f TFLOP/s%s\n", name, blocks, ms, fl / ms * 1e-9, hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
}
int main() {
  float *d, *X;
  (void)hipMalloc(&d, (size_t)65536 * 2 * 64 * 4);
  (void)hipMalloc(&X, (size_t)1024 * 320 * 64 * 4);
  (void)hipMemset(X, 0, (size_t)1024 * 320 * 64 * 4);
  for (int rep = 0; rep < 2; ++rep) {
    run<1>("128x64, long loop", X, d, 512, false);
    run<2>("256x64, long loop", X, d, 256, false);
    run<1>("128x64, 20 launches of 18 tiles", X, d, 512, true);
    run<2>("256x64, 20 launches of 18 tiles", X, d, 256, true);
    run<1, true>("128x64 im2col pattern, 20 x 18 tiles", X, d, 512, true);
    run<2, true>("256x64 im2col pattern, 20 x 18 tiles", X, d, 256, true);
    run<1>("128x64, 20 launches of 18 tiles", X, d, 1024, true);
    run<2>("256x64, 20 launches of 18 tiles", X, d, 512, true);
  }
  return 0;
}
