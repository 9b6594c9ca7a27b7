// Experiment: the GEMM main loop (128x64x32 tile, swizzled LDS double buffer, two tiles in flight) with 8 waves x one 32x32 accumulator
// (gemm_kernel today) against 4 waves x two accumulators (wave tile 32x64: one A fragment feeds two B fragments).  fp32 MFMA shares the
// SIMD's issue with everything else, so instructions per MFMA are what the variants differ in.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/exp_wave2acc tools/exp_wave2acc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 64, BK = 32;
template <int NW>  // 8: wave tile 32x32; 4: wave tile 32x64
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 3) void k(const float* __restrict__ X, float* out, int ntiles) {
  constexpr int NT = 64 * NW, NA = BM * BK / 4 / NT, NB = BN * BK / 4 / NT, NACC = NW == 8 ? 1 : 2;
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave & 3, wn = NW == 8 ? wave >> 2 : 0;
  const int kq = tid & 7, r0 = tid >> 3;  // NT/8 rows per pass
  for (int i = tid; i < 2 * (BM + BN) * BK; i += NT) lds[i] = i * 1e-6f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  float4 r[2][NA + NB];
  for (int s = 0; s < 2; ++s)
    for (int j = 0; j < NA + NB; ++j) r[s][j] = make_float4(1.f, 2.f, 3.f, 4.f);
  const float* xp = X + ((long)blockIdx.x * 192 + r0) * 64 + kq * 4;
  auto gload = [&](int t, float4 (&rr)[NA + NB]) {
#pragma unroll
    for (int j = 0; j < NA + NB; ++j) rr[j] = *reinterpret_cast<const float4*>(xp + (long)j * (NT / 8) * 64 + (t & 1) * 32);
  };
  auto dswrite = [&](int buf, const float4 (&rr)[NA + NB]) {
    float* b = lds + buf * (BM + BN) * BK;
    const int sw = (kq ^ (r0 & 7)) * 4;
#pragma unroll
    for (int j = 0; j < NA + NB; ++j) *reinterpret_cast<float4*>(b + (r0 + (NT / 8) * j) * BK + sw) = rr[j];
  };
  auto compute = [&](int buf) {
    const float* As = lds + buf * (BM + BN) * BK;
    const float* Bs = As + BM * BK;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int sw = ((2 * kg + lh) ^ (li & 7)) << 2;
      float4 a = *reinterpret_cast<const float4*>(As + (wm * 32 + li) * BK + sw);
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        float4 b = *reinterpret_cast<const float4*>(Bs + ((NW == 8 ? wn : n) * 32 + li) * BK + sw);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[n], 0, 0, 0);
      }
    }
  };
  gload(1, r[1]);
  gload(2, r[0]);
  for (int t = 0; t < ntiles; t += 2) {
    dswrite(1, r[1]);
    gload(t + 3, r[1]);
    compute(0);
    __syncthreads();
    dswrite(0, r[0]);
    gload(t + 4, r[0]);
    compute(1);
    __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int i = 0; i < 16; ++i) s += acc[a][i];
  out[blockIdx.x * NT + tid] = s + r[0][0].x + r[1][2].y;
}
template <int NW>
void run(const char* name, const float* X, float* d, int blocks, bool SHORT) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int ntiles = 18 * 20;
  hipLaunchKernelGGL((k<NW>), dim3(blocks), dim3(64 * NW), 0, 0, X, d, 18);
  (void)hipEventRecord(e0);
  if (SHORT) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<NW>), dim3(blocks), dim3(64 * NW), 0, 0, X, d, 18);
  } else {
    hipLaunchKernelGGL((k<NW>), dim3(blocks), dim3(64 * NW), 0, 0, X, d, ntiles);
  }
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * ntiles * 2.0 * BM * BN * BK;
  printf("%-28s %-10s blocks=%d: %.3f ms  %.1f TFLOP/s  (%.1f us per 18-tile launch)\n", name, SHORT ? "20x18" : "1x360", blocks, ms, fl / ms * 1e-9, ms * 50);
}
// 64x64 tile, 4 waves x one 32x32 accumulator: twice as many workgroups (two rounds on the chip), B tile re-staged per 64 rows
__global__ __launch_bounds__(256, 4) void k64(const float* __restrict__ X, float* out, int ntiles) {
  constexpr int M2 = 64;
  __shared__ __attribute__((aligned(16))) float lds[2 * (M2 + BN) * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int kq = tid & 7, r0 = tid >> 3;  // 32 rows per pass
  for (int i = tid; i < 2 * (M2 + BN) * BK; i += 256) lds[i] = i * 1e-6f;
  __syncthreads();
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float4 r[2][4];
  for (int s = 0; s < 2; ++s)
    for (int j = 0; j < 4; ++j) r[s][j] = make_float4(1.f, 2.f, 3.f, 4.f);
  const float* xp = X + ((long)blockIdx.x * 128 + r0) * 64 + kq * 4;
  auto gload = [&](int t, float4 (&rr)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) rr[j] = *reinterpret_cast<const float4*>(xp + (long)j * 32 * 64 + (t & 1) * 32);
  };
  auto dswrite = [&](int buf, const float4 (&rr)[4]) {
    float* b = lds + buf * (M2 + BN) * BK;
    const int sw = (kq ^ (r0 & 7)) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(b + (r0 + 32 * j) * BK + sw) = rr[j];
  };
  auto compute = [&](int buf) {
    const float* As = lds + buf * (M2 + BN) * BK;
    const float* Bs = As + M2 * BK;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int sw = ((2 * kg + lh) ^ (li & 7)) << 2;
      float4 a = *reinterpret_cast<const float4*>(As + (wm * 32 + li) * BK + sw);
      float4 b = *reinterpret_cast<const float4*>(Bs + (wn * 32 + li) * BK + sw);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  };
  gload(1, r[1]);
  gload(2, r[0]);
  for (int t = 0; t < ntiles; t += 2) {
    dswrite(1, r[1]);
    gload(t + 3, r[1]);
    compute(0);
    __syncthreads();
    dswrite(0, r[0]);
    gload(t + 4, r[0]);
    compute(1);
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + tid] = s + r[0][0].x + r[1][2].y;
}
void run64(const float* X, float* d, int blocks, bool SHORT) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int ntiles = 18 * 20;
  hipLaunchKernelGGL(k64, dim3(blocks), dim3(256), 0, 0, X, d, 18);
  (void)hipEventRecord(e0);
  if (SHORT) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k64, dim3(blocks), dim3(256), 0, 0, X, d, 18);
  } else {
    hipLaunchKernelGGL(k64, dim3(blocks), dim3(256), 0, 0, X, d, ntiles);
  }
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * ntiles * 2.0 * 64 * BN * BK;
  printf("%-28s %-10s blocks=%d: %.3f ms  %.1f TFLOP/s  (%.1f us per 18-tile launch)\n", "64x64 tile, 4 waves", SHORT ? "20x18" : "1x360", blocks, ms, fl / ms * 1e-9, ms * 50);
}
// 256x64 tile, SIXTEEN waves x one 32x32 accumulator, one workgroup per CU: four waves per SIMD that advance in lockstep (one barrier
// per tile) -- two co-resident 8-wave workgroups do not: the older one gets the matrix pipe first, ends at 33 us and leaves the
// younger one alone (two waves per SIMD cannot keep the pipe full) until 44 us (per-workgroup timestamps of the product kernel)
__global__ __launch_bounds__(1024, 1) void k16(const float* __restrict__ X, float* out, int ntiles) {
  constexpr int M2 = 256;
  __shared__ __attribute__((aligned(16))) float lds[2 * (M2 + BN) * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave & 7, wn = wave >> 3;
  const int kq = tid & 7, r0 = tid >> 3;  // 128 rows per pass
  for (int i = tid; i < 2 * (M2 + BN) * BK; i += 1024) lds[i] = i * 1e-6f;
  __syncthreads();
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float4 r[2][3];
  for (int s = 0; s < 2; ++s)
    for (int j = 0; j < 3; ++j) r[s][j] = make_float4(1.f, 2.f, 3.f, 4.f);
  const float* xp = X + ((long)blockIdx.x * 384 + r0) * 64 + kq * 4;
  const bool hasb = r0 < 64;  // wave-uniform: waves 0..7 also stage the B tile
  auto gload = [&](int t, float4 (&rr)[3]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) rr[j] = *reinterpret_cast<const float4*>(xp + (long)j * 128 * 64 + (t & 1) * 32);
    if (hasb) rr[2] = *reinterpret_cast<const float4*>(xp + (long)2 * 128 * 64 + (t & 1) * 32);
  };
  auto dswrite = [&](int buf, const float4 (&rr)[3]) {
    float* b = lds + buf * (M2 + BN) * BK;
    const int sw = (kq ^ (r0 & 7)) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(b + (r0 + 128 * j) * BK + sw) = rr[j];
    if (hasb) *reinterpret_cast<float4*>(b + (r0 + 256) * BK + sw) = rr[2];
  };
  auto compute = [&](int buf) {
    const float* As = lds + buf * (M2 + BN) * BK;
    const float* Bs = As + M2 * BK;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int sw = ((2 * kg + lh) ^ (li & 7)) << 2;
      float4 a = *reinterpret_cast<const float4*>(As + (wm * 32 + li) * BK + sw);
      float4 b = *reinterpret_cast<const float4*>(Bs + (wn * 32 + li) * BK + sw);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  };
  gload(1, r[1]);
  gload(2, r[0]);
  for (int t = 0; t < ntiles; t += 2) {
    dswrite(1, r[1]);
    gload(t + 3, r[1]);
    compute(0);
    __syncthreads();
    dswrite(0, r[0]);
    gload(t + 4, r[0]);
    compute(1);
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 1024 + tid] = s + r[0][0].x + r[1][1].y;
}
void run16(const float* X, float* d, int blocks, bool SHORT) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int ntiles = 18 * 20;
  hipLaunchKernelGGL(k16, dim3(blocks), dim3(1024), 0, 0, X, d, 18);
  (void)hipEventRecord(e0);
  if (SHORT) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k16, dim3(blocks), dim3(1024), 0, 0, X, d, 18);
  } else {
    hipLaunchKernelGGL(k16, dim3(blocks), dim3(1024), 0, 0, X, d, ntiles);
  }
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * ntiles * 2.0 * 256 * BN * BK;
  printf("%-28s %-10s blocks=%d: %.3f ms  %.1f TFLOP/s  (%.1f us per 18-tile launch)\n", "256x64 tile, 16 waves", SHORT ? "20x18" : "1x360", blocks, ms, fl / ms * 1e-9, ms * 50);
}
int main() {
  float *d, *X;
  (void)hipMalloc(&d, 2048 * 512 * 4);
  (void)hipMalloc(&X, (size_t)2048 * 192 * 64 * 4);
  (void)hipMemset(X, 0, (size_t)2048 * 192 * 64 * 4);
  for (int blocks : {512, 1024}) {
    for (int s = 0; s < 2; ++s) {
      run<8>("8 waves x 1 acc", X, d, blocks, s);
      run<4>("4 waves x 2 acc", X, d, blocks, s);
    }
  }
  for (int blocks : {256, 512}) {
    run16(X, d, blocks, false);
    run16(X, d, blocks, true);
  }
  for (int blocks : {1024, 2048}) {
    run64(X, d, blocks, false);
    run64(X, d, blocks, true);
  }
  return 0;
}
