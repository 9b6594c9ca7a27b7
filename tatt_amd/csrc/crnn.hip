// CRNN text-prior generator (model/crnn/crnn.py:45-119), forward pieces that the SR kernels do not already provide:
//   * crnn_prep      : parse_crnn_data (interfaces/base.py:441-459) = bicubic resize to 32x100 + gray, NCHW -> NHWC (C padded to 4)
//   * maxpool_gen    : MaxPool2d with kernel/stride/padding (the (2,2),(2,1),(0,1) pools of the CRNN trunk), NHWC
//   * lstm_step_fwd  : one launch per time step of BidirectionalLSTM's nn.LSTM (hidden 256), same scheme as rpe_step_fwd
//   * linear_f64     : the `embedding` Linear that the reference evaluates in float64 (crnn.py:31-33)
// The 3x3 convolutions, BatchNorm and the input projections run on the GEMM/conv/norm kernels of the SR path.
#include "common.h"
#include <cstdint>

namespace {

// ---- bicubic (torch upsample_bicubic2d, align_corners=False, A = -0.75) + gray ------------------------------------------
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;
  float x = t + 1.f;
  w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  x = 1.f - t;
  w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

// x [B,Ctot,H,W] NCHW (channels 0..2 used) -> y [B,OH,OW,4] NHWC, y[...,0] = 0.299 R + 0.587 G + 0.114 B of the resized image
__global__ void crnn_prep_k(const float* __restrict__ x, float* __restrict__ y, int B, int Ctot, int H, int W, int OH, int OW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * OH * OW) return;
  const int ox = i % OW, oy = (i / OW) % OH, b = i / (OW * OH);
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  const float ry = sh * (oy + 0.5f) - 0.5f, rx = sw * (ox + 0.5f) - 0.5f;
  const int iy = (int)floorf(ry), ix = (int)floorf(rx);
  float wy[4], wx[4];
  cubic_coeffs(ry - iy, wy);
  cubic_coeffs(rx - ix, wx);
  float ch[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* p = x + ((long)b * Ctot + c) * H * W;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int yy = min(max(iy - 1 + r, 0), H - 1);
      float row = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) row += wx[q] * p[yy * W + min(max(ix - 1 + q, 0), W - 1)];
      acc += wy[r] * row;
    }
    ch[c] = acc;
  }
  *reinterpret_cast<float4*>(y + (long)i * 4) = make_float4(0.299f * ch[0] + 0.587f * ch[1] + 0.114f * ch[2], 0.f, 0.f, 0.f);
}

// F.interpolate(x, (OH, OW), mode="bicubic") on P = B*C planes (the LR image brought to SR size for the psnr_lr / ssim_lr
// metrics, interfaces/super_resolution_eng.py:486)
__global__ void bicubic_planes_k(const float* __restrict__ x, float* __restrict__ y, int P, int H, int W, int OH, int OW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)P * OH * OW) return;
  const int ox = i % OW, oy = (i / OW) % OH;
  const long pl = i / ((long)OW * OH);
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  const float ry = sh * (oy + 0.5f) - 0.5f, rx = sw * (ox + 0.5f) - 0.5f;
  const int iy = (int)floorf(ry), ix = (int)floorf(rx);
  float wy[4], wx[4];
  cubic_coeffs(ry - iy, wy);
  cubic_coeffs(rx - ix, wx);
  const float* p = x + pl * H * W;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int yy = min(max(iy - 1 + r, 0), H - 1);
    float row = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) row += wx[q] * p[yy * W + min(max(ix - 1 + q, 0), W - 1)];
    acc += wy[r] * row;
  }
  y[i] = acc;
}

// ---- MaxPool2d(kernel, stride, padding), NHWC; padded positions never win (torch pads with -inf) -----------------------
__global__ void maxpool_gen_k(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int KH, int KW, int SH,
                              int SW, int PH, int PW, int OH, int OW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * OH * OW * C) return;
  const int c = i % C;
  const int ox = (i / C) % OW, oy = (i / ((long)C * OW)) % OH, b = i / ((long)C * OW * OH);
  float m = -INFINITY;
  for (int dy = 0; dy < KH; ++dy) {
    const int yy = oy * SH - PH + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = 0; dx < KW; ++dx) {
      const int xx = ox * SW - PW + dx;
      if (xx < 0 || xx >= W) continue;
      m = fmaxf(m, x[(((long)b * H + yy) * W + xx) * C + c]);
    }
  }
  y[i] = m;
}

// ---- out[r][n] = float( b[n] + sum_k double(x[r][k]) * double(W[n][k]) ) ----------------------------------------------------
// v_mfma_f64_16x16x4_f64: a wave owns a 16 x 16 output tile (four waves = 64 rows x 16 columns per workgroup); lane (i = lane & 15,
// q = lane >> 4) loads four consecutive k (k0 + 4 q ..) of row / column i as one float4 and feeds element e to the e-th of four
// MFMAs -- A and B use the same (q, e) -> k map, so 16 k are consumed per trip with one load per operand.  (The first version was
// one thread per output with two LDS reads and two conversions per fp64 FMA: VALU-issue bound, 47 us for the 1664 x 256 x 512 call.)
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4k(const float* __restrict__ p, long row_off, int k, int K, bool row_ok, bool vec) {
  if (!row_ok) return make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec && k + 3 < K) return *reinterpret_cast<const float4*>(p + row_off + k);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) v.x = p[row_off + k];
  if (k + 1 < K) v.y = p[row_off + k + 1];
  if (k + 2 < K) v.z = p[row_off + k + 2];
  if (k + 3 < K) v.w = p[row_off + k + 3];
  return v;
}
__global__ __launch_bounds__(256) void linear_f64_k(const float* __restrict__ x, long ldx, const float* __restrict__ Wt,
                                                    const float* __restrict__ bias, float* __restrict__ out, long ldo, int R, int N,
                                                    int K) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * 64 + wv * 16, n0 = blockIdx.y * 16;
  const bool xv = (ldx % 4 == 0) && ((((uintptr_t)x) & 15) == 0), wvec = (K % 4 == 0) && ((((uintptr_t)Wt) & 15) == 0);
  const bool rok = r0 + i < R, nok = n0 + i < N;
  const long xo = (long)(r0 + i) * ldx, wo = (long)(n0 + i) * K;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  float4 a = ld4k(x, xo, 4 * q, K, rok, xv), b = ld4k(Wt, wo, 4 * q, K, nok, wvec);
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float4 an = ld4k(x, xo, k0 + 16 + 4 * q, K, rok, xv), bn = ld4k(Wt, wo, k0 + 16 + 4 * q, K, nok, wvec);  // next trip's operands
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a.x, (double)b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a.y, (double)b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a.z, (double)b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a.w, (double)b.w, acc, 0, 0, 0);
    a = an;
    b = bn;
  }
  // D layout of the f64 instruction: lane (i, q), register r holds row 4 r + q of column i (unlike f32 16x16x4: row 4 q + r)
  const int n = n0 + i;
  if (n >= N) return;
  const double bv = bias ? (double)bias[n] : 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + 4 * r + q;
    if (row < R) out[(long)row * ldo + n] = (float)(acc[r] + bv);
  }
}

// ---- LSTM step (hidden 256, gate order i, f, g, o; torch nn.LSTM) ---------------------------------------------------------
// gi [T*B][2*1024] = W_ih x + b_ih for both directions (time-major rows); state h, c [2][B][256]; out [T*B][512] (dir concat).
// Block (jb, dir, row block of 64) owns hidden units j0..j0+3: its 16 gate columns are one 16-wide MFMA tile; the four waves
// split K = 256 and are summed through LDS; then the gate math for (row, unit).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LH = 256;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_step_fwd_k(
    const float* __restrict__ gi, const float* __restrict__ whh_f, const float* __restrict__ whh_b, const float* __restrict__ bhh_f,
    const float* __restrict__ bhh_b, const float* __restrict__ hprev, const float* __restrict__ cprev, float* __restrict__ hout,
    float* __restrict__ cout, float* __restrict__ out, float* __restrict__ save, int s, int T, int B) {
  __shared__ float red[4][64][17];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
  const int dir = blockIdx.y, j0 = blockIdx.x * 4, rb = blockIdx.z * 64;
  const float* W = dir ? whh_b : whh_f;
  const float* hp = hprev + (long)dir * B * LH;
  const int kbase = wv * 64 + q * 4;
  float4 bf[4], af[4][4];
  {
    const float* pb = W + (long)((i >> 2) * LH + j0 + (i & 3)) * LH + kbase;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bf[e] = *reinterpret_cast<const float4*>(pb + 16 * e);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = rb + mt * 16 + i;
        af[mt][e] = row < B ? *reinterpret_cast<const float4*>(hp + (long)row * LH + kbase + 16 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  const int wl = tid >> 2, jj = tid & 3, j = j0 + jj, row = rb + wl;
  const int t = dir ? T - 1 - s : s;
  const bool live = row < B;
  const float* bh = dir ? bhh_b : bhh_f;
  float g4[4] = {0.f, 0.f, 0.f, 0.f}, cp = 0.f;
  if (live) {
    const float* g_i = gi + ((long)t * B + row) * (8 * LH) + dir * 4 * LH;
#pragma unroll
    for (int g = 0; g < 4; ++g) g4[g] = g_i[g * LH + j] + bh[g * LH + j];
    cp = cprev[((long)dir * B + row) * LH + j];
  }
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_sched_barrier(0);  // every load above is issued before the first MFMA: the step is pure latency
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e].x, bf[e].x, acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e].y, bf[e].y, acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e].z, bf[e].z, acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e].w, bf[e].w, acc[mt], 0, 0, 0);
    }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][mt * 16 + 4 * q + r][i] = acc[mt][r];
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) g4[g] += (red[0][wl][g * 4 + jj] + red[1][wl][g * 4 + jj]) + (red[2][wl][g * 4 + jj] + red[3][wl][g * 4 + jj]);
  const float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf_(g4[2]), og = sigmoidf_(g4[3]);
  const float c = fg * cp + ig * gg;
  const float h = og * tanhf_(c);
  cout[((long)dir * B + row) * LH + j] = c;
  hout[((long)dir * B + row) * LH + j] = h;
  out[((long)t * B + row) * (2 * LH) + dir * LH + j] = h;
  if (save) {  // [T*B][2][5][256]: i, f, g, o, c of this step (for lstm_step_bwd_k)
    float* sp = save + (((long)t * B + row) * 2 + dir) * 5 * LH + j;
    sp[0] = ig;
    sp[LH] = fg;
    sp[2 * LH] = gg;
    sp[3 * LH] = og;
    sp[4 * LH] = c;
  }
}

// BPTT step s (forward-order time t = s for dir 0, T-1-s for dir 1; the launches run s = T-1 .. 0).
// Block (jb, mh, dir) owns rows mh*32..+31 and hidden units jb*16..+15:
//   dh = dout[t] + dgates_{next}[.,1024] W_hh          (K = 1024 split over the four waves; `next` = the step launched before)
//   dc = dc_carry + dh o (1 - tanh^2 c);  gate gradients;  dc_carry <- dc f;  dgates[t] written for the next launch / the GEMMs
// whhT [2][256][1024] = W_hh^T; dgates [T*B][2048] (dir*1024 + gate*256 + j).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_step_bwd_k(
    const float* __restrict__ save, const float* __restrict__ whhT, const float* __restrict__ dout, float* __restrict__ dgates,
    float* __restrict__ dc_carry, int s, int T, int B) {
  __shared__ float red[4][32][17];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
  const int jb = blockIdx.x, mh = blockIdx.y, dir = blockIdx.z;
  const int t = dir ? T - 1 - s : s;
  const int tn = dir ? t - 1 : t + 1;  // time index of the step that ran after this one in the forward recurrence
  const bool has_next = s + 1 < T;
  if (has_next) {
    const float* A = dgates + ((long)tn * B + mh * 32 + i) * (8 * LH) + dir * 4 * LH + wv * 256 + q * 4;
    const float* Bt = whhT + ((long)dir * LH + jb * 16 + i) * (4 * LH) + wv * 256 + q * 4;
    float4 af[2][16], bf[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      bf[e] = *reinterpret_cast<const float4*>(Bt + 16 * e);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const bool on = mh * 32 + mt * 16 + i < B;
        af[mt][e] = on ? *reinterpret_cast<const float4*>(A + (long)mt * 16 * (8 * LH) + 16 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e].x, bf[e].x, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e].y, bf[e].y, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e].z, bf[e].z, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e].w, bf[e].w, acc[mt], 0, 0, 0);
      }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wv][mt * 16 + 4 * q + r][i] = acc[mt][r];
  }
  __syncthreads();
  const int tp = dir ? t + 1 : t - 1;  // forward-order predecessor (its c is c_prev of this step)
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int e = tid + 256 * h2, wl = e >> 4, jj = e & 15;
    const int row = mh * 32 + wl, j = jb * 16 + jj;
    if (row >= B) continue;
    float dh = dout[((long)t * B + row) * (2 * LH) + dir * LH + j];
    if (has_next) dh += (red[0][wl][jj] + red[1][wl][jj]) + (red[2][wl][jj] + red[3][wl][jj]);
    const float* sp = save + (((long)t * B + row) * 2 + dir) * 5 * LH + j;
    const float ig = sp[0], fg = sp[LH], gg = sp[2 * LH], og = sp[3 * LH], c = sp[4 * LH];
    const float cprev = (s > 0) ? save[(((long)tp * B + row) * 2 + dir) * 5 * LH + 4 * LH + j] : 0.f;
    const float tc = tanhf_(c);
    const long ci = ((long)dir * B + row) * LH + j;
    const float dc = (has_next ? dc_carry[ci] : 0.f) + dh * og * (1.f - tc * tc);
    float* o = dgates + ((long)t * B + row) * (8 * LH) + dir * 4 * LH + j;
    o[0] = dc * gg * ig * (1.f - ig);
    o[LH] = dc * cprev * fg * (1.f - fg);
    o[2 * LH] = dc * ig * (1.f - gg * gg);
    o[3 * LH] = dh * tc * og * (1.f - og);
    dc_carry[ci] = dc * fg;
  }
}

// ---- the whole recurrence of one nn.LSTM(., 256, bidirectional) as ONE launch (tatt_lstm256_chain_fwd / _bwd) --------------------
// Same scheme as the recurrent positional encoding's chains (gru.hip, rpe_chain_fwd): the batch rows are independent, so a chain =
// (block of 16 rows, direction) lives inside ONE XCD's L2: 32 persistent workgroups, each owning 8 hidden units (their 32 gate
// columns of W_hh, or 8 rows of W_hh^T in the backward pass, stay in registers for all T steps), exchange the 16 x 256 state through
// `out` itself (forward) / the gate gradients (backward) with one flag line per reader; a reader polls with one returning L2 atomic
// and then reads with ordinary first-touch loads.  B = 64 gives 4 row blocks x 2 directions = 8 chains = the 8 XCDs of the chip;
// any B <= 128 works (group k of XCD x runs chain x + n_xcd * k).  Every wait is bounded; a wait that gives up sets the sticky
// error word sync[1] and poisons what the chain writes from then on with NaN (the loss shows it).  Hardware assumptions: gru.hip.
// The workgroups are kept small (forward 256 threads, < 64 registers) so that the chains of the teacher, the student and the RPE,
// which run at the same time on different streams, always fit on an XCD together: a chain that cannot place all its workgroups
// would wait for itself until the spin limit.
constexpr unsigned LSTM_SPIN_LIMIT = 1u << 18;
constexpr int LSTM_MAX_CHAINS = 16;
constexpr int LSTM_SYNC_WORDS = 32 * 9 + LSTM_MAX_CHAINS * 1024;   // header (sync[1] sticky error), one ticket line per XCD, per chain [reader 32][writer 32]

__device__ __forceinline__ unsigned lstm_l2_read(unsigned* p) {    // a plain load could be served from the CU's cache for ever
  unsigned v;
  const unsigned zero = 0u;
  asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(zero) : "memory");
  return v;
}
// role of this workgroup: chain (row block mh, direction dir) and its 8 hidden units (jb); false: surplus, exit
__device__ __forceinline__ bool lstm_chain_role(unsigned* sync, int nxcd, int nchains, int& jb, int& mh, int& dir, unsigned*& flags) {
  __shared__ unsigned s_role[2];
  if (threadIdx.x == 0) {
    const int x = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15) % nxcd;  // HW_REG_XCC_ID[3:0]
    unsigned* line = sync + 32 * (1 + x);
    const unsigned t = __hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    s_role[0] = (unsigned)x + (unsigned)nxcd * (t >> 5);
    s_role[1] = t & 31;
  }
  __syncthreads();
  const unsigned role = s_role[0];
  jb = (int)s_role[1];
  mh = (int)(role >> 1);
  dir = (int)(role & 1);
  flags = sync + 32 * 9 + 1024 * (role & (LSTM_MAX_CHAINS - 1));
  return role < (unsigned)nchains;
}
// wait until the 32 owners of this chain have published step `need` (workgroup-uniform result: false = gave up)
__device__ __forceinline__ bool lstm_wait(unsigned* flags, int jb, unsigned need, unsigned* sync) {
  for (unsigned it = 0;; ++it) {
    unsigned f = 0xffffffffu;
    if (threadIdx.x < 32) f = lstm_l2_read(flags + jb * 32 + threadIdx.x);
    if (__syncthreads_and(f >= need)) return true;
    if (it > LSTM_SPIN_LIMIT) {
      if (threadIdx.x == 0) __hip_atomic_fetch_or(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ void lstm_publish(unsigned* flags, int jb, unsigned v) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);          // this thread's stores have reached L2
  __syncthreads();
  if (threadIdx.x < 32) *reinterpret_cast<volatile unsigned*>(flags + threadIdx.x * 32 + jb) = v;  // one copy per reader
}

// gi [T*B][2048] (dir*1024 + gate*256 + j), out [T*B][512] (dir*256 + j), save [T*B][2][5][256] or NULL: as tatt_lstm256_step_fwd.
__global__ __launch_bounds__(256, 4) void lstm_chain_fwd_k(const float* __restrict__ gi, const float* __restrict__ whh_f,
                                                           const float* __restrict__ whh_b, const float* __restrict__ bhh_f,
                                                           const float* __restrict__ bhh_b, float* out, float* __restrict__ save,
                                                           int T, int B, unsigned* sync, int nxcd) {
  __shared__ float red[4][16][33];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
  int jb, mh, dir;
  unsigned* flags;
  if (!lstm_chain_role(sync, nxcd, 2 * ((B + 15) / 16), jb, mh, dir, flags)) return;
  const float* W = dir ? whh_b : whh_f;
  const int kbase = wv * 64 + q * 4;      // the four waves split K = 256; lane (i, q): k = kbase + 16 e + 0..3
  // once for the whole chain: W_hh rows of this workgroup's 32 gate columns (two 16-wide tiles: column i of tile n = gate 2n + i/8, unit i%8)
  float4 bf[2][4];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const float* pb = W + (long)((2 * n + (i >> 3)) * LH + jb * 8 + (i & 7)) * LH + kbase;
#pragma unroll
    for (int e = 0; e < 4; ++e) bf[n][e] = *reinterpret_cast<const float4*>(pb + 16 * e);
  }
  // gate phase: thread (row wl, unit jj) for tid < 128
  const int wl = (tid >> 3) & 15, jj = tid & 7, row = mh * 16 + wl, j = jb * 8 + jj;
  const bool gate = tid < 128 && row < B;
  const float* bh = dir ? bhh_b : bhh_f;
  float bg[4] = {0.f, 0.f, 0.f, 0.f};
  if (gate) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bg[g] = bh[g * LH + j];
  }
  const int arow = mh * 16 + i;            // A operand row of this lane
  const bool aon = arow < B;
  float cp = 0.f;
  bool bad = false;
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s, tp = dir ? t + 1 : t - 1;
    float g4[4] = {0.f, 0.f, 0.f, 0.f};
    if (gate) {                            // independent of the chain: requested before the wait
      const float* g_i = gi + ((long)t * B + row) * (8 * LH) + dir * 4 * LH + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) g4[g] = g_i[g * LH] + bg[g];
    }
    if (s > 0) {
      if (!lstm_wait(flags, jb, (unsigned)s, sync)) bad = true;
      // h of the previous step: rows of `out` written by this XCD's owners, first touched by this CU now
      float4 af[4];
      const float* pa = out + ((long)tp * B + arow) * (2 * LH) + dir * LH + kbase;
#pragma unroll
      for (int e = 0; e < 4; ++e) af[e] = aon ? *reinterpret_cast<const float4*>(pa + 16 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e].x, bf[n][e].x, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e].y, bf[n][e].y, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e].z, bf[n][e].z, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e].w, bf[n][e].w, acc[n], 0, 0, 0);
        }
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][4 * q + r][n * 16 + i] = acc[n][r];
      __syncthreads();
      if (gate) {
#pragma unroll
        for (int g = 0; g < 4; ++g) g4[g] += (red[0][wl][g * 8 + jj] + red[1][wl][g * 8 + jj]) + (red[2][wl][g * 8 + jj] + red[3][wl][g * 8 + jj]);
      }
    }
    if (gate) {
      const float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf_(g4[2]), og = sigmoidf_(g4[3]);
      const float c = fg * cp + ig * gg;
      const float h = og * tanhf_(c);
      cp = c;
      out[((long)t * B + row) * (2 * LH) + dir * LH + j] = bad ? __builtin_nanf("") : h;
      if (save) {
        float* sp = save + (((long)t * B + row) * 2 + dir) * 5 * LH + j;
        sp[0] = ig;
        sp[LH] = fg;
        sp[2 * LH] = gg;
        sp[3 * LH] = og;
        sp[4 * LH] = c;
      }
    }
    lstm_publish(flags, jb, (unsigned)s + 1u);
  }
}

// BPTT chain: as T launches of lstm_step_bwd_k (s = T-1 .. 0).  dgates [T*B][2048] is both the result and the hand-over buffer;
// the carried dc belongs to one thread for the whole chain (a register).
__global__ __launch_bounds__(512, 2) void lstm_chain_bwd_k(const float* __restrict__ save, const float* __restrict__ whhT,
                                                           const float* __restrict__ dout, float* dgates, int T, int B, unsigned* sync,
                                                           int nxcd) {
  __shared__ float red[8][16][9];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
  int jb, mh, dir;
  unsigned* flags;
  if (!lstm_chain_role(sync, nxcd, 2 * ((B + 15) / 16), jb, mh, dir, flags)) return;
  // once: rows jb*8 .. +7 of W_hh^T (K = 1024 split over the eight waves); columns 8..15 of the 16-wide tile are padding
  const int kbase = wv * 128 + q * 4;
  float4 bf[8];
  {
    const float* Bt = whhT + ((long)dir * LH + jb * 8 + (i & 7)) * (4 * LH) + kbase;
#pragma unroll
    for (int e = 0; e < 8; ++e) bf[e] = i < 8 ? *reinterpret_cast<const float4*>(Bt + 16 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int wl = (tid >> 3) & 15, jj = tid & 7, row = mh * 16 + wl, j = jb * 8 + jj;
  const bool gate = tid < 128 && row < B;
  const int arow = mh * 16 + i;
  const bool aon = arow < B;
  float carry = 0.f;
  bool bad = false;
  for (int s = T - 1; s >= 0; --s) {
    const int t = dir ? T - 1 - s : s;
    const int tn = dir ? t - 1 : t + 1;   // the step differentiated just before (its gate gradients feed dh)
    const int tp = dir ? t + 1 : t - 1;   // forward-order predecessor (its c is c_prev of this step)
    float dh = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, cprev = 0.f;
    if (gate) {                            // independent of the chain: requested before the wait
      dh = dout[((long)t * B + row) * (2 * LH) + dir * LH + j];
      const float* sp = save + (((long)t * B + row) * 2 + dir) * 5 * LH + j;
      ig = sp[0]; fg = sp[LH]; gg = sp[2 * LH]; og = sp[3 * LH]; c = sp[4 * LH];
      if (s > 0) cprev = save[(((long)tp * B + row) * 2 + dir) * 5 * LH + 4 * LH + j];
    }
    if (s + 1 < T) {
      if (!lstm_wait(flags, jb, (unsigned)(T - 1 - s), sync)) bad = true;
      const float* A = dgates + ((long)tn * B + arow) * (8 * LH) + dir * 4 * LH + kbase;
      float4 af[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) af[e] = aon ? *reinterpret_cast<const float4*>(A + 16 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e].x, bf[e].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e].y, bf[e].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e].z, bf[e].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e].w, bf[e].w, acc, 0, 0, 0);
      }
      if (i < 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][4 * q + r][i] = acc[r];
      }
      __syncthreads();
      if (gate) {
#pragma unroll
        for (int v = 0; v < 8; v += 2) dh += red[v][wl][jj] + red[v + 1][wl][jj];
      }
    }
    if (gate) {
      const float tc = tanhf_(c);
      const float dc = carry + dh * og * (1.f - tc * tc);
      float* o = dgates + ((long)t * B + row) * (8 * LH) + dir * 4 * LH + j;
      const float nanv = bad ? __builtin_nanf("") : 0.f;
      o[0] = dc * gg * ig * (1.f - ig) + nanv;
      o[LH] = dc * cprev * fg * (1.f - fg);
      o[2 * LH] = dc * ig * (1.f - gg * gg);
      o[3 * LH] = dh * tc * og * (1.f - og);
      carry = dc * fg;
    }
    lstm_publish(flags, jb, (unsigned)(T - s));
  }
}

// out[c][r] = in[r][c]
__global__ void transpose_k(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float t[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) t[r][tx] = (r0 + r < rows && c0 + tx < cols) ? in[(long)(r0 + r) * cols + c0 + tx] : 0.f;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (c0 + r < cols && r0 + tx < rows) out[(long)(c0 + r) * rows + r0 + tx] = t[tx][r];
}

// MaxPool2d backward as a gather (windows overlap when stride < kernel): an input element receives dy of every window
// whose arg-max (first maximum in (dy, dx) scan order, like ATen) it is.  Deterministic, no atomics.
__global__ void maxpool_gen_bwd_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B, int H,
                                  int W, int C, int KH, int KW, int SH, int SW, int PH, int PW, int OH, int OW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * H * W * C) return;
  const int c = i % C;
  const int xx = (i / C) % W, yy = (i / ((long)C * W)) % H, b = i / ((long)C * W * H);
  float g = 0.f;
  const int oy0 = max(0, (yy + PH - KH + SH) / SH), oy1 = min(OH - 1, (yy + PH) / SH);
  const int ox0 = max(0, (xx + PW - KW + SW) / SW), ox1 = min(OW - 1, (xx + PW) / SW);
  for (int oy = oy0; oy <= oy1; ++oy)
    for (int ox = ox0; ox <= ox1; ++ox) {
      float m = -INFINITY;
      int ay = -1, ax = -1;
      for (int a = 0; a < KH; ++a) {
        const int y2 = oy * SH - PH + a;
        if (y2 < 0 || y2 >= H) continue;
        for (int d = 0; d < KW; ++d) {
          const int x2 = ox * SW - PW + d;
          if (x2 < 0 || x2 >= W) continue;
          const float v = x[(((long)b * H + y2) * W + x2) * C + c];
          if (v > m) {
            m = v;
            ay = y2;
            ax = x2;
          }
        }
      }
      if (ay == yy && ax == xx) g += dy[(((long)b * OH + oy) * OW + ox) * C + c];
    }
  dx[i] = g;
}

// ---- SemanticLoss (loss/semantic_loss.py:21-38) fused with the two softmaxes that feed it ---------------------------------
// pred = softmax(zs), gt = softmax(zt) over K classes; per row: sum |gt - pred| and sum (gt+1e-20)(log(gt+1e-20) - log(pred+1e-20));
// loss = (mean |gt - pred| + mean KL) over all R*K elements.  dzs (+)= dloss * d loss / d zs through the softmax.
__global__ void semantic_loss_k(const float* __restrict__ zs, const float* __restrict__ zt, float* __restrict__ partial,
                                float* __restrict__ dzs, float dscale, int R, int K) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  float tot = 0.f;
  if (r < R) {
    const float* a = zs + (long)r * K;
    const float* b = zt + (long)r * K;
    float ma = -INFINITY, mb = -INFINITY;
    for (int k = 0; k < K; ++k) {
      ma = fmaxf(ma, a[k]);
      mb = fmaxf(mb, b[k]);
    }
    float sa = 0.f, sb = 0.f;
    for (int k = 0; k < K; ++k) {
      sa += expf(a[k] - ma);
      sb += expf(b[k] - mb);
    }
    float gdot = 0.f;
    for (int k = 0; k < K; ++k) {
      const float p = expf(a[k] - ma) / sa, g = expf(b[k] - mb) / sb;
      tot += fabsf(g - p) + (g + 1e-20f) * (logf(g + 1e-20f) - logf(p + 1e-20f));
      if (dzs) {
        const float d = (p > g ? 1.f : (p < g ? -1.f : 0.f)) - (g + 1e-20f) / (p + 1e-20f);
        gdot += d * p;
      }
    }
    if (dzs)
      for (int k = 0; k < K; ++k) {
        const float p = expf(a[k] - ma) / sa, g = expf(b[k] - mb) / sb;
        const float d = (p > g ? 1.f : (p < g ? -1.f : 0.f)) - (g + 1e-20f) / (p + 1e-20f);
        dzs[(long)r * K + k] = dscale * p * (d - gdot);
      }
  }
  __shared__ float red[4];
  tot = wave_sum(tot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// SemanticLoss on probabilities as the reference module takes them (pred_vec, gt_vec): no softmax; dpred = dscale * d loss/d pred
__global__ void semantic_loss_probs_k(const float* __restrict__ pred, const float* __restrict__ gt, float* __restrict__ partial,
                                      float* __restrict__ dpred, float dscale, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float tot = 0.f;
  if (i < n) {
    const float p = pred[i], g = gt[i];
    tot = fabsf(g - p) + (g + 1e-20f) * (logf(g + 1e-20f) - logf(p + 1e-20f));
    if (dpred) dpred[i] = dscale * ((p > g ? 1.f : (p < g ? -1.f : 0.f)) - (g + 1e-20f) / (p + 1e-20f));
  }
  __shared__ float red[4];
  tot = wave_sum(tot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[r][:] = softmax(z[r][:]) (the text prior label_vecs = softmax(logits, -1), interfaces/super_resolution_eng.py:227)
__global__ void softmax_rows_k(const float* __restrict__ z, float* __restrict__ out, int R, int K) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* a = z + (long)r * K;
  float m = -INFINITY, s = 0.f;
  for (int k = 0; k < K; ++k) m = fmaxf(m, a[k]);
  for (int k = 0; k < K; ++k) s += expf(a[k] - m);
  for (int k = 0; k < K; ++k) out[(long)r * K + k] = expf(a[k] - m) / s;
}

__global__ void semantic_final_k(const float* __restrict__ partial, int n, float scale, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// Greedy CTC decode of get_string_crnn (utils/metrics.py:128-146): one wave per sample b.  arg[b][t] = argmax_k x[t][b][k]
// (first index on ties, as torch.max); codes[b][0..lens[b]) = the indices left after merging runs and dropping blank 0.
__global__ __launch_bounds__(64) void ctc_greedy_k(const float* __restrict__ x, int* __restrict__ arg, int* __restrict__ codes,
                                                   int* __restrict__ lens, int T, int B, int K) {
  const int b = blockIdx.x, l = threadIdx.x;
  int last = 0, n = 0;
  for (int t = 0; t < T; ++t) {
    const float* r = x + ((long)t * B + b) * K;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = l; k < K; k += 64) {
      const float v = r[k];
      if (v > bv) { bv = v; bi = k; }
    }
    for (int o = 32; o; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (bi == 0x7fffffff) bi = 0;
    if (l == 0) {
      arg[b * T + t] = bi;
      if (bi != last && bi != 0) codes[b * T + n++] = bi;
      last = bi;
    }
  }
  if (l == 0) {
    lens[b] = n;
    for (int i = n; i < T; ++i) codes[b * T + i] = 0;
  }
}

}  // namespace

extern "C" {

int tatt_bicubic_resize(const float* x, float* y, int planes, int H, int W, int OH, int OW, hipStream_t st) {
  TATT_CHECK_ARG(x && y && planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "tatt_bicubic_resize: bad arguments");
  TATT_CHECK_ARG((long)planes * OH * OW < (1L << 31) * 256, "tatt_bicubic_resize: too large");
  hipLaunchKernelGGL(bicubic_planes_k, dim3(cdiv((long)planes * OH * OW, 256)), dim3(256), 0, st, x, y, planes, H, W, OH, OW);
  TATT_LAUNCH_CHECK("bicubic_resize");
  return TATT_OK;
}

int tatt_ctc_greedy(const float* logits, int* arg, int* codes, int* lens, int T, int B, int K, hipStream_t st) {
  TATT_CHECK_ARG(logits && arg && codes && lens && T > 0 && B > 0 && K > 0, "tatt_ctc_greedy: bad arguments");
  hipLaunchKernelGGL(ctc_greedy_k, dim3(B), dim3(64), 0, st, logits, arg, codes, lens, T, B, K);
  TATT_LAUNCH_CHECK("ctc_greedy");
  return TATT_OK;
}

// parse_crnn_data: x [B,Ctot>=3,H,W] NCHW in [0,1] -> y [B,OH,OW,4] NHWC (channel 0 = gray, 1..3 = 0)
int tatt_crnn_prep(const float* x, float* y, int B, int Ctot, int H, int W, int OH, int OW, hipStream_t st) {
  TATT_CHECK_ARG(x && y && B > 0 && Ctot >= 3 && H > 0 && W > 0 && OH > 0 && OW > 0, "tatt_crnn_prep: bad arguments");
  hipLaunchKernelGGL(crnn_prep_k, dim3(cdiv((long)B * OH * OW, 256)), dim3(256), 0, st, x, y, B, Ctot, H, W, OH, OW);
  TATT_LAUNCH_CHECK("crnn_prep");
  return TATT_OK;
}

int tatt_maxpool_gen_fwd(const float* x, float* y, int B, int H, int W, int C, int KH, int KW, int SH, int SW, int PH, int PW,
                         hipStream_t st) {
  TATT_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && SH > 0 && SW > 0 && PH >= 0 && PW >= 0 &&
                     2 * PH <= KH && 2 * PW <= KW, "tatt_maxpool_gen_fwd: bad arguments");
  const int OH = (H + 2 * PH - KH) / SH + 1, OW = (W + 2 * PW - KW) / SW + 1;
  TATT_CHECK_ARG(OH > 0 && OW > 0, "tatt_maxpool_gen_fwd: empty output");
  hipLaunchKernelGGL(maxpool_gen_k, dim3(cdiv((long)B * OH * OW * C, 256)), dim3(256), 0, st, x, y, B, H, W, C, KH, KW, SH, SW, PH, PW,
                     OH, OW);
  TATT_LAUNCH_CHECK("maxpool_gen_fwd");
  return TATT_OK;
}

int tatt_linear_f64(const float* x, long ldx, const float* Wt, const float* bias, float* out, long ldo, int R, int N, int K,
                    hipStream_t st) {
  TATT_CHECK_ARG(x && Wt && out && R > 0 && N > 0 && K > 0, "tatt_linear_f64: bad arguments");
  hipLaunchKernelGGL(linear_f64_k, dim3(cdiv(R, 64), cdiv(N, 16)), dim3(256), 0, st, x, ldx, Wt, bias, out, ldo, R, N, K);
  TATT_LAUNCH_CHECK("linear_f64");
  return TATT_OK;
}

int tatt_lstm256_step_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b,
                          const float* hprev, const float* cprev, float* hout, float* cout, float* out, float* save, int s, int T, int B,
                          hipStream_t st) {
  TATT_CHECK_ARG(gi && whh_f && whh_b && bhh_f && bhh_b && hprev && cprev && hout && cout && out && s >= 0 && s < T && B > 0,
                 "tatt_lstm256_step_fwd: bad arguments");
  hipLaunchKernelGGL(lstm_step_fwd_k, dim3(LH / 4, 2, cdiv(B, 64)), dim3(256), 0, st, gi, whh_f, whh_b, bhh_f, bhh_b, hprev, cprev,
                     hout, cout, out, save, s, T, B);
  TATT_LAUNCH_CHECK("lstm_step_fwd");
  return TATT_OK;
}

int tatt_lstm256_step_bwd(const float* save, const float* whhT, const float* dout, float* dgates, float* dc_carry, int s, int T, int B,
                          hipStream_t st) {
  TATT_CHECK_ARG(save && whhT && dout && dgates && dc_carry && s >= 0 && s < T && B > 0, "tatt_lstm256_step_bwd: bad arguments");
  hipLaunchKernelGGL(lstm_step_bwd_k, dim3(LH / 16, cdiv(B, 32), 2), dim3(256), 0, st, save, whhT, dout, dgates, dc_carry, s, T, B);
  TATT_LAUNCH_CHECK("lstm_step_bwd");
  return TATT_OK;
}

static int lstm_num_xcd() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || n < 1 || n > 8) n = 8;
  return n;
}
size_t tatt_rpe_chain_supported(void);   // gru.hip: same device requirements (gfx942 / gfx950, every XCD can hold its chains)

size_t tatt_lstm256_chain_sync_words(void) { return (size_t)LSTM_SYNC_WORDS; }

// All T steps of tatt_lstm256_step_fwd in one launch (h_0 = c_0 = 0).  sync: tatt_lstm256_chain_sync_words() uint32 that the caller
// zeroes ONCE; sync[1] != 0 afterwards = a wait gave up (sticky across calls; results invalid, outputs NaN from that step on).
int tatt_lstm256_chain_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b, float* out,
                           float* save, int T, int B, uint32_t* sync, hipStream_t st) {
  TATT_CHECK_ARG(gi && whh_f && whh_b && bhh_f && bhh_b && out && sync && T > 0 && B > 0, "tatt_lstm256_chain_fwd: bad arguments");
  TATT_CHECK_ARG(B <= 8 * LSTM_MAX_CHAINS, "tatt_lstm256_chain_fwd: at most %d rows (got %d): use the per-step launches", 8 * LSTM_MAX_CHAINS, B);
  TATT_CHECK_ARG(tatt_rpe_chain_supported(), "tatt_lstm256_chain_fwd: this device cannot hold the chains (tatt_rpe_chain_supported() == 0): use the per-step launches");
  TATT_CHECK_ARG(((((uintptr_t)gi) | ((uintptr_t)whh_f) | ((uintptr_t)whh_b) | ((uintptr_t)out)) & 15) == 0, "tatt_lstm256_chain_fwd: 16-byte aligned pointers required");
  const int nxcd = lstm_num_xcd();
  if (hipMemsetAsync(sync + 8, 0, (LSTM_SYNC_WORDS - 8) * 4, st) != hipSuccess) {
    tatt_set_error("tatt_lstm256_chain_fwd: hipMemsetAsync failed");
    return TATT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(lstm_chain_fwd_k, dim3(512), dim3(256), 0, st, gi, whh_f, whh_b, bhh_f, bhh_b, out, save, T, B, sync, nxcd);
  TATT_LAUNCH_CHECK("lstm_chain_fwd");
  return TATT_OK;
}

// All T BPTT steps of tatt_lstm256_step_bwd (s = T-1 .. 0) in one launch; no dc_carry buffer (it lives in registers).  A sync buffer of its own.
int tatt_lstm256_chain_bwd(const float* save, const float* whhT, const float* dout, float* dgates, int T, int B, uint32_t* sync, hipStream_t st) {
  TATT_CHECK_ARG(save && whhT && dout && dgates && sync && T > 0 && B > 0, "tatt_lstm256_chain_bwd: bad arguments");
  TATT_CHECK_ARG(B <= 8 * LSTM_MAX_CHAINS, "tatt_lstm256_chain_bwd: at most %d rows (got %d): use the per-step launches", 8 * LSTM_MAX_CHAINS, B);
  TATT_CHECK_ARG(tatt_rpe_chain_supported(), "tatt_lstm256_chain_bwd: this device cannot hold the chains (tatt_rpe_chain_supported() == 0): use the per-step launches");
  TATT_CHECK_ARG(((((uintptr_t)whhT) | ((uintptr_t)dgates)) & 15) == 0, "tatt_lstm256_chain_bwd: 16-byte aligned pointers required");
  const int nxcd = lstm_num_xcd();
  if (hipMemsetAsync(sync + 8, 0, (LSTM_SYNC_WORDS - 8) * 4, st) != hipSuccess) {
    tatt_set_error("tatt_lstm256_chain_bwd: hipMemsetAsync failed");
    return TATT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(lstm_chain_bwd_k, dim3(512), dim3(512), 0, st, save, whhT, dout, dgates, T, B, sync, nxcd);
  TATT_LAUNCH_CHECK("lstm_chain_bwd");
  return TATT_OK;
}

int tatt_transpose(const float* in, float* out, int rows, int cols, hipStream_t st) {
  TATT_CHECK_ARG(in && out && rows > 0 && cols > 0, "tatt_transpose: bad arguments");
  hipLaunchKernelGGL(transpose_k, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, st, in, out, rows, cols);
  TATT_LAUNCH_CHECK("transpose");
  return TATT_OK;
}

int tatt_maxpool_gen_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int KH, int KW, int SH, int SW, int PH,
                         int PW, hipStream_t st) {
  TATT_CHECK_ARG(x && dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && SH > 0 && SW > 0 && PH >= 0 && PW >= 0 &&
                     2 * PH <= KH && 2 * PW <= KW, "tatt_maxpool_gen_bwd: bad arguments");
  const int OH = (H + 2 * PH - KH) / SH + 1, OW = (W + 2 * PW - KW) / SW + 1;
  hipLaunchKernelGGL(maxpool_gen_bwd_k, dim3(cdiv((long)B * H * W * C, 256)), dim3(256), 0, st, x, dy, dx, B, H, W, C, KH, KW, SH, SW,
                     PH, PW, OH, OW);
  TATT_LAUNCH_CHECK("maxpool_gen_bwd");
  return TATT_OK;
}

size_t tatt_semantic_loss_workspace(int R) { return (size_t)cdiv(R, 256); }

// SemanticLoss.forward(pred_vec, gt_vec) on probabilities (n elements): out[0] = loss, dpred (optional) = dloss * dloss/dpred.
// workspace: ceil(n / 256) floats.
int tatt_semantic_loss_probs(const float* pred, const float* gt, float* out, float* dpred, float dloss, long n, float* workspace,
                             hipStream_t st) {
  TATT_CHECK_ARG(pred && gt && out && workspace && n > 0, "tatt_semantic_loss_probs: bad arguments");
  const int nb = cdiv(n, 256);
  const float inv = 1.f / (float)n;
  hipLaunchKernelGGL(semantic_loss_probs_k, dim3(nb), dim3(256), 0, st, pred, gt, workspace, dpred, dloss * inv, n);
  TATT_LAUNCH_CHECK("semantic_loss_probs");
  hipLaunchKernelGGL(semantic_final_k, dim3(1), dim3(64), 0, st, workspace, nb, inv, out);
  TATT_LAUNCH_CHECK("semantic_final");
  return TATT_OK;
}

int tatt_softmax_rows(const float* z, float* out, int R, int K, hipStream_t st) {
  TATT_CHECK_ARG(z && out && R > 0 && K > 0, "tatt_softmax_rows: bad arguments");
  hipLaunchKernelGGL(softmax_rows_k, dim3(cdiv(R, 256)), dim3(256), 0, st, z, out, R, K);
  TATT_LAUNCH_CHECK("softmax_rows");
  return TATT_OK;
}

// out[0] = SemanticLoss(softmax(zs), softmax(zt)); if dzs: dzs = dloss * d out / d zs (zt, the teacher, takes no gradient)
int tatt_semantic_loss(const float* zs, const float* zt, float* out, float* dzs, float dloss, int R, int K, float* workspace,
                       hipStream_t st) {
  TATT_CHECK_ARG(zs && zt && out && workspace && R > 0 && K > 0, "tatt_semantic_loss: bad arguments");
  const int nb = cdiv(R, 256);
  const float inv = 1.f / ((float)R * K);
  hipLaunchKernelGGL(semantic_loss_k, dim3(nb), dim3(256), 0, st, zs, zt, workspace, dzs, dloss * inv, R, K);
  TATT_LAUNCH_CHECK("semantic_loss");
  hipLaunchKernelGGL(semantic_final_k, dim3(1), dim3(64), 0, st, workspace, nb, inv, out);
  TATT_LAUNCH_CHECK("semantic_final");
  return TATT_OK;
}

}  // extern "C"
