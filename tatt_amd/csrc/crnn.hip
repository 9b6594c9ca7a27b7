// CRNN text-prior generator (model/crnn/crnn.py:45-119), forward pieces that the SR kernels do not already provide:
//   * crnn_prep      : parse_crnn_data (interfaces/base.py:441-459) = bicubic resize to 32x100 + gray, NCHW -> NHWC (C padded to 4)
//   * maxpool_gen    : MaxPool2d with kernel/stride/padding (the (2,2),(2,1),(0,1) pools of the CRNN trunk), NHWC
//   * lstm_step_fwd  : one launch per time step of BidirectionalLSTM's nn.LSTM (hidden 256), same scheme as rpe_step_fwd
//   * linear_f64     : the `embedding` Linear that the reference evaluates in float64 (crnn.py:31-33)
// The 3x3 convolutions, BatchNorm and the input projections run on the GEMM/conv/norm kernels of the SR path.
#include "common.h"

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
__global__ __launch_bounds__(256) void linear_f64_k(const float* __restrict__ x, long ldx, const float* __restrict__ Wt,
                                                    const float* __restrict__ bias, float* __restrict__ out, long ldo, int R, int N,
                                                    int K) {
  __shared__ float xs[16][65], ws[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int r0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
  double acc = 0.0;
  for (int k0 = 0; k0 < K; k0 += 64) {
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int rr = e >> 6, kk = e & 63;
      xs[rr][kk] = (r0 + rr < R && k0 + kk < K) ? x[(long)(r0 + rr) * ldx + k0 + kk] : 0.f;
      ws[rr][kk] = (n0 + rr < N && k0 + kk < K) ? Wt[(long)(n0 + rr) * K + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 64; ++kk) acc += (double)xs[ty][kk] * (double)ws[tx][kk];
    __syncthreads();
  }
  if (r0 + ty < R && n0 + tx < N) out[(long)(r0 + ty) * ldo + n0 + tx] = (float)(acc + (bias ? (double)bias[n0 + tx] : 0.0));
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
    float* __restrict__ cout, float* __restrict__ out, int s, int T, int B) {
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
}

}  // namespace

extern "C" {

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
  hipLaunchKernelGGL(linear_f64_k, dim3(cdiv(R, 16), cdiv(N, 16)), dim3(256), 0, st, x, ldx, Wt, bias, out, ldo, R, N, K);
  TATT_LAUNCH_CHECK("linear_f64");
  return TATT_OK;
}

int tatt_lstm256_step_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b,
                          const float* hprev, const float* cprev, float* hout, float* cout, float* out, int s, int T, int B,
                          hipStream_t st) {
  TATT_CHECK_ARG(gi && whh_f && whh_b && bhh_f && bhh_b && hprev && cprev && hout && cout && out && s >= 0 && s < T && B > 0,
                 "tatt_lstm256_step_fwd: bad arguments");
  hipLaunchKernelGGL(lstm_step_fwd_k, dim3(LH / 4, 2, cdiv(B, 64)), dim3(256), 0, st, gi, whh_f, whh_b, bhh_f, bhh_b, hprev, cprev,
                     hout, cout, out, s, T, B);
  TATT_LAUNCH_CHECK("lstm_step_fwd");
  return TATT_OK;
}

}  // extern "C"
