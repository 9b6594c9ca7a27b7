// EXPERIMENT (end of round 3), not built into the library: parity-green (fp64 product 2e-6, tiled GEMM 2e-6) and measured 39.0 / 24.7 us
// against 40.5 / 26.5 us for tatt_gemm_cat at K = 128 / 64, B = 64 -- the output, not the multiply, is what these launches wait for.
// Knock-outs (K = 128): LDS reads + MFMA only 27.5 us (ideal 20.5), the W preload 4.4 us, the 48 dword stores per lane and tile 6.6 us;
// no staging piece (global loads, LDS stores, barriers) moves it.  The store cost is per wave instruction (~20 cycles each through the
// CU's store path, 384 of them per tile), additive to the MFMA time because every wave of the CU stores at the same point of the loop:
// behind the barrier, spread three per k-step over the next k-tile, or as float4 runs of four channels per pixel (MFMA roles swapped:
// partial lines, slower: 42.7 us) -- none hides it.  What would: full-line float4 stores after a transpose through LDS, for which the
// K = 128 form has no LDS left.  DESIGN.md section 8.
// Input projection of a GruBlock with the 1x1 convolution folded in (model/tsrn.py:1082-1088; gru_fold.hip makes W_f, b_f):
//   gi[R,192] = [A1 | A2][R,K] W_f[192,K]^T + b_f,   K = 64 (one 64-channel input) or 128 (conv features | text-prior map),
// on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation in k order).
//
// Why a kernel of its own: in the tiled GEMM (gemm_conv.hip) this shape is 1536 workgroups of two or four k-tiles each -- load
// latency, LDS staging and the 32 KB output tile are most of a workgroup's life, the matrix pipe runs at 57-78 TF, and in the forward
// pass nothing runs beside these ten launches (tools/alone_time.py: 367 us of the step with exactly this kernel on the GPU).
// Here W_f stays in LDS for the life of a persistent workgroup (one per CU) and the rows of A stream through a double buffer:
//   * 8 waves = 4 (32-row slabs of a 128-row tile) x 2 (96-column halves of the 192 outputs): three accumulators per wave, so one A
//     fragment feeds three MFMAs and three independent accumulation chains interleave;
//   * W_f as [k][n] with a row stride of 224 floats (= 32 mod 64 banks): the two k rows that the halves of a wave read lie 32 banks
//     apart, conflict-free; A as [m][k] with a row stride of 34 floats: rows land on distinct even banks, the k+1 half on the odd ones;
//   * the tile's next k-tile (or the next tile's first) is in registers while the current one is multiplied, the one after that in
//     flight; one barrier per k-tile; the 96 KB of output per tile drain while the next tile is multiplied.
#include "common.h"

namespace {

constexpr int GP_THR = 512, GP_BM = 128, GP_N = 192, GP_BK = 32, GP_AS = 34, GP_WS = 224;

template <int KT>  // K = 32 * KT
__global__ __launch_bounds__(GP_THR) void gru_proj_k(const float* __restrict__ A1, const float* __restrict__ A2, long lda1, long lda2,
                                                     const float* __restrict__ Wf, const float* __restrict__ bias, float* __restrict__ C,
                                                     long ldc, int ntiles) {
  constexpr int K = KT * GP_BK;
  __shared__ float Ws[K * GP_WS];
  __shared__ __attribute__((aligned(16))) float As[2][GP_BM * GP_AS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5, wm = wave & 3, wn = wave >> 2;
  // W_f [192][K] -> Ws[k][n]: consecutive lanes take consecutive n of one k quad (conflict-free stores; the loads come from L2)
  for (int i = tid; i < GP_N * (K / 4); i += GP_THR) {
    const int kq = i / GP_N, n = i - kq * GP_N;
    const float4 v = *reinterpret_cast<const float4*>(Wf + (long)n * K + kq * 4);
    Ws[(kq * 4 + 0) * GP_WS + n] = v.x;
    Ws[(kq * 4 + 1) * GP_WS + n] = v.y;
    Ws[(kq * 4 + 2) * GP_WS + n] = v.z;
    Ws[(kq * 4 + 3) * GP_WS + n] = v.w;
  }
  // this workgroup's tiles: blockIdx.x, blockIdx.x + gridDim.x, ...; work items s = (tile index, k-tile), k-tile fastest
  const int mine = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int total = mine * KT;
  const int kq = tid & 7, r0 = tid >> 3;  // A loader: rows r0 and r0 + 64 of the tile, k quad kq of the k-tile
  auto gload = [&](int s, float4 (&v)[2]) {
    const int it = s / KT, kt = s - it * KT;
    const long row = (long)(blockIdx.x + it * gridDim.x) * GP_BM + r0;
    const bool second = KT == 4 && kt >= 2;
    const float* src = second ? A2 : A1;
    const long ld = second ? lda2 : lda1;
    const int col = (kt & 1) * GP_BK + kq * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) v[j] = *reinterpret_cast<const float4*>(src + (row + 64 * j) * ld + col);
  };
  auto sstore = [&](int buf, const float4 (&v)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* d = &As[buf][(r0 + 64 * j) * GP_AS + kq * 4];  // 8-byte aligned: 136 r + 16 kq
      *reinterpret_cast<float2*>(d) = make_float2(v[j].x, v[j].y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(v[j].z, v[j].w);
    }
  };
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  float bv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) bv[j] = bias ? bias[wn * 96 + j * 32 + li] : 0.f;

  auto multiply = [&](int s) {
    const int it = s / KT, kt = s - it * KT;
    const float* as = &As[s & 1][(wm * 32 + li) * GP_AS + lh];
    const float* ws = &Ws[(kt * GP_BK + lh) * GP_WS + wn * 96 + li];
#pragma unroll
    for (int kk = 0; kk < GP_BK; kk += 2) {
      const float a = as[kk];
      const float b0 = ws[kk * GP_WS], b1 = ws[kk * GP_WS + 32], b2 = ws[kk * GP_WS + 64];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
    }
  };
  // The finished tile is stored AFTER the next k-tile has gone to LDS and the barrier: stores count in vmcnt like loads, in order, so a
  // wait for prefetched rows that follows the 48 stores waits for the stores' round trip too (measured 3.3 us per tile with the stores
  // in front of that wait); behind it they have the whole next k-tile to drain.
  auto finish_tile = [&](int s) {
    const int it = s / KT;
    const long row0 = (long)(blockIdx.x + it * gridDim.x) * GP_BM + wm * 32 + 4 * lh;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float* c = C + row0 * ldc + wn * 96 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        c[(long)((r & 3) + 8 * (r >> 2)) * ldc] = acc[j][r] + bv[j];
        acc[j][r] = 0.f;
      }
    }
  };

  if (total == 0) return;
  float4 ra[2], rb[2];
  gload(0, ra);
  gload(1, rb);  // total is even (KT is)
  sstore(0, ra);
  __syncthreads();  // Ws and the first k-tile are in place
  for (int s = 0; s < total; s += 2) {
    // even item: registers `ra` are free (stored), `rb` hold item s + 1
    if (s + 2 < total) gload(s + 2, ra);
    multiply(s);
    sstore(1, rb);
    __syncthreads();
    // odd item: `rb` are free, `ra` hold item s + 2
    if (s + 3 < total) gload(s + 3, rb);
    multiply(s + 1);
    if (s + 2 < total) sstore(0, ra);
    __syncthreads();
    if (((s + 1) % KT) == KT - 1) finish_tile(s + 1);  // (KT is even: a tile ends on an odd item)
  }
}

}  // namespace

extern "C" {

// gi[R,192] = [A1 | A2] W_f^T + b_f.  K = 64 (A2 unused) or 128 (A1, A2: 64 columns each); R a multiple of 128; 16-byte aligned
// operands, row strides multiples of 4.  Returns TATT_ERR_ARG for anything else (callers fall back to tatt_gemm_cat).
int tatt_gru_proj(const float* A1, const float* A2, long lda1, long lda2, const float* Wf, const float* bias, float* C, long ldc, int R,
                  int K, hipStream_t st) {
  TATT_CHECK_ARG(A1 && Wf && C && R > 0 && R % GP_BM == 0 && (K == 64 || K == 128) && (K == 64 || A2) && lda1 % 4 == 0 && lda1 >= 64 &&
                     (K == 64 || (lda2 % 4 == 0 && lda2 >= 64)) && ldc >= GP_N && ((((uintptr_t)A1) | ((uintptr_t)A2) | ((uintptr_t)Wf)) & 15) == 0,
                 "tatt_gru_proj: need K in {64, 128}, R %% 128 == 0, 16-byte aligned operands (R %d, K %d)", R, K);
  const int ntiles = R / GP_BM;
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus = n;
    if (cus <= 0) cus = 256;
  }
  // one persistent workgroup per CU; a grid that divides the tile count keeps every workgroup at the same number of tiles
  const int most = ntiles < cus ? ntiles : cus;
  int grid = most;
  while (ntiles % grid) --grid;
  if (grid * 2 < most) grid = most;  // no divisor near the CU count: accept a ragged last round
  if (K == 64)
    hipLaunchKernelGGL((gru_proj_k<2>), dim3(grid), dim3(GP_THR), 0, st, A1, A2, lda1, lda2, Wf, bias, C, ldc, ntiles);
  else
    hipLaunchKernelGGL((gru_proj_k<4>), dim3(grid), dim3(GP_THR), 0, st, A1, A2, lda1, lda2, Wf, bias, C, ldc, ntiles);
  TATT_LAUNCH_CHECK("gru_proj");
  return TATT_OK;
}

}  // extern "C"
