// Row-streaming GEMM for the tall, short-K products of the path: C[M, N] (+)= A[M, K] B^T + bias with M = B*H*W = 65,536 rows and
// K, N <= 192 -- the folded GRU input projection ([x | tp_map] -> 192 gates) and its input gradient, every Linear of the TP
// interpreter's decoder and their input gradients (model/tsrn.py:1082-1088, model/transformer_v2.py:805-832).
//
// Why a second GEMM kernel: with K = 64..192 the tiled kernel (gemm_conv.hip) runs two to six k-tiles per workgroup, so a launch is
// "everybody loads, everybody multiplies, everybody stores": measured on [R,64]x[64,64], R = 16k..262k: 3.6 TB/s in the steady state,
// i.e. time = MFMA time + HBM time, not their maximum.  Here the loop runs over ROWS instead:
//   * the whole B operand (<= 96 KB) is staged into LDS once per workgroup, in the k-tile-major swizzled layout the MFMA reads;
//   * every wave streams its own 32-row groups: A sub-tiles (32 rows x 32 k) go global -> registers -> a wave-PRIVATE LDS double
//     buffer -> MFMA fragments, two tiles in flight in registers, the sequence (group, k-tile) flattened so that the first tiles of
//     the next group are requested while the current one is multiplied and stored;
//   * no workgroup barrier after the B panel is in place: waves drift apart, loads, MFMAs and stores of different waves overlap;
//   * a wave owns 32 rows x N columns (N/32 accumulators = independent MFMA chains), so A is read once for all of N.
// v_mfma_f32_32x32x2_f32, fp32 fma chains in k order 0..K-1 per output element: the same sums as the tiled kernel.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4s __attribute__((ext_vector_type(4)));
constexpr int SW = 4;  // waves per workgroup

struct StreamArgs {
  const float* A;
  const float* A2;  // second source for k >= ksplit (null: none)
  int ksplit;
  long sam;
  const float* B;
  long sbn, sbk;
  float* C;
  float* C2;  // columns >= nsplit (null: none)
  int nsplit;
  long ldc, ldc2;
  int acc1, acc2, relu;
  const float* bias;
  int M;
  // BatchNorm-backward epilogue on the first destination (EPI 2)
  const float* bn_x;
  long bn_ldx;
  const float* bn_stat;
  const float* bn_gamma;
  const float* bn_beta;
  int bn_act, bn_C;
  float* partial;  // [gridDim.x * SW][2][bn_C]
  int dbg;
};

__device__ __forceinline__ float act_grad_s(float t, int act) {
  if (act == 1) return t > 0.f ? 1.f : 0.f;
  if (act == 2) return mish_gradf_(t);
  return 1.f;
}

template <int NT, int KT, int EPI>
// waves per SIMD the register allocation must leave room for: three workgroups per CU for the small panels, two otherwise (left
// alone the scheduler hoists every B fragment read and takes 250-450 VGPRs)
__global__ __launch_bounds__(64 * SW) __attribute__((amdgpu_waves_per_eu(NT <= 2 ? 3 : 2))) void gemm_stream_kernel(StreamArgs a) {
  constexpr int N = NT * 32, K = KT * 32;
  __shared__ __attribute__((aligned(16))) float Bs[KT * N * 32];       // [k-tile][n][32 k], 16-byte groups swizzled by n & 7
  __shared__ __attribute__((aligned(16))) float As[SW][2][32 * 32];    // per wave: two [32 rows][32 k] images, swizzled by row & 7
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // ---- B panel -> LDS (once) ----
  if (a.sbk == 1) {  // k-contiguous rows of B(n, .)
    for (int i = tid; i < N * (K / 4); i += 64 * SW) {
      const int n = i / (K / 4), c = i - n * (K / 4);
      const float4 v = *reinterpret_cast<const float4*>(a.B + (long)n * a.sbn + c * 4);
      *reinterpret_cast<float4*>(Bs + ((c >> 3) * N + n) * 32 + (((c & 7) ^ (n & 7)) << 2)) = v;
    }
  } else {  // n-contiguous rows of B(., k): transposed on the way in
    for (int i = tid; i < K * (N / 4); i += 64 * SW) {
      const int k = i / (N / 4), n4 = (i - k * (N / 4)) * 4;
      const float4 v = *reinterpret_cast<const float4*>(a.B + (long)k * a.sbk + n4);
      const float e[4] = {v.x, v.y, v.z, v.w};
      const int kk = k & 31, cc = kk >> 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) Bs[((k >> 5) * N + n4 + j) * 32 + ((cc ^ ((n4 + j) & 7)) << 2) + (kk & 3)] = e[j];
    }
  }
  __syncthreads();

  // ---- this wave's row groups: g = gw, gw + W, ... (all waves sweep one dense window of the matrix together) ----
  const int W = gridDim.x * SW, gw = blockIdx.x * SW + wave;
  const int ngroups = a.M >> 5;
  const int mygroups = gw < ngroups ? (ngroups - gw + W - 1) / W : 0;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)0x80000000, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A2 ? a.A2 : a.A), 0, (int)0x80000000, 0x00020000);
  // loader: lane -> (row r + 8 j, 16-byte k group c): eight lanes cover one row's 128-byte k-tile
  const int lr = lane >> 3, lc = lane & 7;
  int vo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) vo[j] = (int)(((long)(lr + 8 * j) * a.sam + lc * 4) * 4);
  const int st_off = lr * 32 + ((lc ^ (lr & 7)) << 2);  // rows r and r + 8 j share r & 7
  // wave-uniform position of the next tile to request
  int t_g = gw, t_kt = 0;
  auto load_tile = [&](float4 (&ra)[4]) {
    const int k0 = t_kt * 32;
    const bool second = k0 >= a.ksplit;
    const __amdgpu_buffer_rsrc_t rs = second ? rA2 : rA;
    const int so = (int)(((long)t_g * 32 * a.sam + (second ? k0 - a.ksplit : k0)) * 4);
    const bool live = t_g < ngroups && !(a.dbg & 4);  // past the end: nothing is requested (the values are never used)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      i32x4s v = __builtin_amdgcn_raw_buffer_load_b128(rs, live ? vo[j] : (int)0x80000000, so, 0);
      ra[j] = *reinterpret_cast<float4*>(&v);  // (component-wise bit_casts of the result compile to a one-dword load on this toolchain:
      // the timings in DESIGN.md were taken with that defect, i.e. with a quarter of the A traffic, and were still not faster)
    }
    if (++t_kt == KT) {
      t_kt = 0;
      t_g += W;
    }
  };
  float* Aw = &As[wave][0][0];
  auto store_tile = [&](int buf, const float4 (&ra)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(Aw + buf * 1024 + st_off + j * 8 * 32) = ra[j];
  };

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

  float s1[NT], s2[NT];  // EPI 2: running (sum g, sum g xhat) of this lane's column over all of the wave's rows
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.f;

  // Epilogue addressing: element (row0 + rr(r) + 4 lh, n) with rr(r) = (r & 3) + 8 (r >> 2).  Everything but the lane's own part
  // ((4 lh) ld + li) is wave-uniform and rides in the scalar offset of a buffer access; the column tile nt is an immediate.  (Plain
  // pointers cost a 64-bit address per (row, tile): 250-450 VGPRs and scratch spills.)
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)a.C, 0, (int)0x80000000, 0x00020000);
  const __amdgpu_buffer_rsrc_t rC2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.C2 ? a.C2 : a.C), 0, (int)0x80000000, 0x00020000);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 2 ? a.bn_x : a.A), 0, (int)0x80000000, 0x00020000);
  const int vC = (int)((4 * lh * a.ldc + li) * 4), vC2 = (int)((4 * lh * a.ldc2 + li) * 4), vX = (int)((4 * lh * a.bn_ldx + li) * 4);
  const int ldc4 = (int)a.ldc * 4, ldc24 = (int)a.ldc2 * 4, ldx4 = (int)a.bn_ldx * 4;
  auto epilogue = [&](int g) {
    const int row0 = g * 32;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 32 + li;
      const bool second = nt * 32 >= a.nsplit;  // uniform
      const __amdgpu_buffer_rsrc_t rd = second ? rC2 : rC;
      const int ld4 = second ? ldc24 : ldc4;
      const int vd = (second ? vC2 - a.nsplit * 4 : vC) + nt * 128;
      const bool accum = second ? a.acc2 : a.acc1;
      const float bv = a.bias ? a.bias[n] : 0.f;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[nt][r] + bv;
      if (EPI == 2 && !second && nt * 32 < a.bn_C) {
        float xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          xv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, vX + nt * 128, (row0 + (r & 3) + 8 * (r >> 2)) * ldx4, 0));
        const float mean = a.bn_stat[n], inv = a.bn_stat[a.bn_C + n], ga = a.bn_gamma[n], be = a.bn_beta[n];
        auto pass = [&](auto actc) {
          constexpr int ACT = decltype(actc)::value;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float xh = (xv[r] - mean) * inv;
            const float gq = v[r] * act_grad_s(xh * ga + be, ACT);
            s1[nt] += gq;
            s2[nt] += gq * xh;
            v[r] = gq;
          }
        };
        if (a.bn_act == 2) pass(std::integral_constant<int, 2>{});
        else if (a.bn_act == 1) pass(std::integral_constant<int, 1>{});
        else pass(std::integral_constant<int, 0>{});
      }
      if (accum) {
        float prev[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          prev[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, vd, (row0 + (r & 3) + 8 * (r >> 2)) * ld4, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += prev[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float o = v[r];
        if (a.relu) o = fmaxf(o, 0.f);
        if (!(a.dbg & 2)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rd, vd, (row0 + (r & 3) + 8 * (r >> 2)) * ld4, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    }
  };

  auto compute = [&](int buf, int kt) {  // kt is a constant after unrolling
    const float* At = Aw + buf * 1024;
    const float* Bt = Bs + kt * N * 32;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int sw = ((2 * kg + lh) ^ (li & 7)) << 2;
      const float4 av = *reinterpret_cast<const float4*>(At + li * 32 + sw);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float4 bv = *reinterpret_cast<const float4*>(Bt + (nt * 32 + li) * 32 + sw);
        if (a.dbg & 1) continue;
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[nt], 0, 0, 0);
      }
    }
  };

  // Two register sets, two LDS images: tiles t+1 and t+2 are in flight while tile t is multiplied.  The body of the group loop is
  // straight-line (KT is even and a compile-time constant; the epilogue runs unconditionally at its end): a conditional path with
  // memory operations inside the loop would cost the exact vmcnt bookkeeping that keeps the second tile in flight.  Requests past
  // the wave's last tile are predicated off.
  float4 ra0[4], ra1[4];
  load_tile(ra0);
  load_tile(ra1);
  store_tile(0, ra0);
  load_tile(ra0);
  int c_g = gw;
  for (int gi = 0; gi < mygroups; ++gi) {
#pragma unroll
    for (int kt = 0; kt < KT; kt += 2) {
      store_tile(1, ra1);
      load_tile(ra1);
      compute(0, kt);
      store_tile(0, ra0);
      load_tile(ra0);
      compute(1, kt + 1);
    }
    epilogue(c_g);
    c_g += W;
  }

  if constexpr (EPI == 2) {
    // one partial per wave: its sums over every row it owned (any partition of the rows gives the same totals)
    float* p = a.partial + (long)gw * 2 * a.bn_C;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt * 32 >= a.bn_C || nt * 32 >= a.nsplit) continue;
      const float t1 = s1[nt] + __shfl_xor(s1[nt], 32, 64), t2 = s2[nt] + __shfl_xor(s2[nt], 32, 64);
      if (lh == 0) {
        p[nt * 32 + li] = t1;
        p[a.bn_C + nt * 32 + li] = t2;
      }
    }
  }
}

template <int NT, int KT, int EPI>
int launch_stream(const StreamArgs& a, int grid, hipStream_t st) {
  hipLaunchKernelGGL((gemm_stream_kernel<NT, KT, EPI>), dim3(grid), dim3(64 * SW), 0, st, a);
  TATT_LAUNCH_CHECK("gemm_stream_kernel");
  return TATT_OK;
}

bool stream_enabled() {
  static const bool on = [] {
    const char* e = getenv("TATT_GEMM_STREAM");
    return !(e && e[0] == '0');
  }();
  return on;
}

}  // namespace

extern "C" {

// Number of workgroups (of SW waves) the streaming kernel is launched with for an [M, N, K] product: as many as stay resident at once
// (LDS: the B panel + 32 KB of wave-private A images per workgroup), at most one wave per 32-row group.
int tatt_gemm_stream_grid(int M, int N, int K) {
  static const int gpw = [] {  // row groups per wave aimed at (experiments: TATT_STREAM_GPW)
    const char* e = getenv("TATT_STREAM_GPW");
    const int v = e ? atoi(e) : 2;
    return v < 1 ? 1 : v;
  }();
  const int lds = (N * K + SW * 2 * 1024) * 4;
  int per_cu = (160 * 1024) / lds;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  const int cap = 256 * per_cu;                                   // workgroups that stay resident at once
  int grid = ((M >> 5) + SW * gpw - 1) / (SW * gpw);              // gpw groups per wave
  if (grid < 256) grid = 256;                                     // ... but never fewer workgroups than CUs
  if (grid > cap) grid = cap;
  const int need = ((M >> 5) + SW - 1) / SW;
  return grid < need ? grid : need;
}

// Is this product one the streaming kernel takes?  (Shapes instantiated: N in {64, 128, 192} x K in {64, 128, 192}.)
bool tatt_gemm_stream_ok(const float* A, const float* A2, int ksplit, const float* B, const float* C, const float* C2, int nsplit, int M,
                         int N, int K, long sam, long sbn, long sbk, long ldc, long ldc2, bool bn) {
  if (!stream_enabled()) return false;
  auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (M < 4096 || (M & 31) || (N != 64 && N != 128 && N != 192) || (K != 64 && K != 128 && K != 192)) return false;
  if (N * K > 128 * 192) return false;  // the B panel + 32 KB of A images must fit the 160 KB of LDS
  if (bn && N == 192) return false;
  if (!al16(A) || !al16(B) || sam % 4 || ((long)M * sam + K) * 4 >= (1L << 31)) return false;
  if (((long)M * ldc + N) * 4 >= (1L << 31) || (C2 && ((long)M * ldc2 + N) * 4 >= (1L << 31))) return false;  // 32-bit buffer offsets
  if (A2 && (!al16(A2) || ksplit % 32 || ksplit <= 0 || ksplit >= K)) return false;
  if (!((sbk == 1 && sbn % 4 == 0) || (sbn == 1 && sbk % 4 == 0))) return false;
  if (C2 && (nsplit % 32 || nsplit <= 0 || nsplit >= N)) return false;
  (void)C;
  return true;
}

int tatt_gemm_stream_launch(const float* A, const float* A2, int ksplit, const float* B, float* C, float* C2, int nsplit, const float* bias,
                            int M, int N, int K, long sam, long sbn, long sbk, long ldc, long ldc2, int accumulate, int accumulate2, int relu,
                            const float* bn_x, long bn_ldx, const float* bn_stat, const float* bn_gamma, const float* bn_beta, int bn_act,
                            int bn_C, float* partial, hipStream_t st) {
  StreamArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.A2 = A2; a.ksplit = A2 ? ksplit : 0x7fffffff; a.sam = sam;
  a.B = B; a.sbn = sbn; a.sbk = sbk;
  a.C = C; a.C2 = C2; a.nsplit = C2 ? nsplit : 0x7fffffff; a.ldc = ldc; a.ldc2 = ldc2;
  a.acc1 = accumulate; a.acc2 = accumulate2; a.relu = relu; a.bias = bias; a.M = M;
  a.bn_x = bn_x; a.bn_ldx = bn_ldx; a.bn_stat = bn_stat; a.bn_gamma = bn_gamma; a.bn_beta = bn_beta; a.bn_act = bn_act; a.bn_C = bn_C;
  a.partial = partial;
  { const char* e = getenv("TATT_STREAM_DBG"); a.dbg = e ? atoi(e) : 0; }
  const int grid = tatt_gemm_stream_grid(M, N, K);
  const int nt = N / 32, kt = K / 32;
#define TS(NTv, KTv)                                                   \
  if (nt == NTv && kt == KTv) {                                        \
    if (partial) {                                                     \
      if constexpr (NTv <= 4) return launch_stream<NTv, KTv, 2>(a, grid, st); \
    } else {                                                           \
      return launch_stream<NTv, KTv, 0>(a, grid, st);                  \
    }                                                                  \
  }
  TS(2, 2) TS(2, 4) TS(2, 6) TS(4, 2) TS(4, 4) TS(4, 6) TS(6, 2) TS(6, 4)
#undef TS
  tatt_set_error("tatt_gemm_stream_launch: shape not instantiated");
  return TATT_ERR_ARG;
}

}  // extern "C"
