// HBM-bound element-wise / layout / small-reduction kernels of the SR path.
#include "common.h"

namespace {

#define GRID1D(n) dim3(cdiv((n), 256)), dim3(256)

__global__ void nchw_to_nhwc_k(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C * HW) return;
  int c = (int)(i % C);
  long p = i / C;  // b*HW + hw
  long b = p / HW, hw = p - b * HW;
  y[i] = x[(b * C + c) * HW + hw];
}
__global__ void nhwc_to_nchw_k(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C * HW) return;
  long hw = i % HW;
  long bc = i / HW;
  long b = bc / C, c = bc - b * C;
  y[i] = x[(b * HW + hw) * C + c];
}

// out = alpha*a + beta*b (b optional) (+ out if accumulate)
__global__ void axpby_k(const float* __restrict__ a, float alpha, const float* __restrict__ b, float beta, float* __restrict__ out,
                        long n, int accumulate) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = alpha * a[i];
  if (b) v += beta * b[i];
  if (accumulate) v += out[i];
  out[i] = v;
}

__global__ void prelu_fwd_k(const float* __restrict__ x, const float* __restrict__ slope, float* __restrict__ y, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  y[i] = v > 0.f ? v : v * slope[0];
}
// dx = dy * (x>0 ? 1 : slope) ; partial[blk] = sum dy*x over x<=0
__global__ __launch_bounds__(256) void prelu_bwd_k(const float* __restrict__ x, const float* __restrict__ slope,
                                                   const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ partial,
                                                   long n) {
  __shared__ float red[4];
  float s = 0.f;
  const float a = slope[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i], g = dy[i];
    dx[i] = v > 0.f ? g : g * a;
    if (!(v > 0.f)) s += g * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sum_partials_k(const float* __restrict__ partial, int n, float scale, float* __restrict__ out, int accumulate) {
  // one wave: lane l sums partial[l], partial[l + 64], ... (fixed order), then a butterfly -- deterministic, and ~20x shorter
  // than one thread walking a thousand dependent L2 loads
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  s = wave_sum(s) * scale;
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}

// column sums of [R,C] (ld) : stage 1 partial[blk][C]; blockIdx.x = row block, blockIdx.y = chunk of 1024 columns
__global__ __launch_bounds__(256) void colsum_partial_k(const float* __restrict__ x, long ld, int R, int C, int rows_per_blk,
                                                        float* __restrict__ partial) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int rb = blockIdx.x * rows_per_blk, re = min(R, rb + rows_per_blk);
  const int cbeg = blockIdx.y * 1024, cend = min(C, cbeg + 1024);
  if ((C & 3) == 0 && (ld & 3) == 0 && ((((uintptr_t)x) & 15) == 0)) {
    // vector path: thread = (float4 column group, row lane); four independent accumulators keep loads in flight
    __shared__ float4 red4[256];
    const int C4 = (cend - cbeg) >> 2, CT4 = C4, RL4 = 256 / CT4;
    const int cl4 = tid % CT4, rl4 = tid / CT4;
    for (int c4 = (cbeg >> 2) + cl4; c4 < (cend >> 2); c4 += CT4) {
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
      if (rl4 < RL4) {
        int r = rb + rl4;
        float4 s2 = s0, s3 = s0;
        for (; r + 3 * RL4 < re; r += 4 * RL4) {
          float4 a = *reinterpret_cast<const float4*>(x + (long)r * ld + c4 * 4);
          float4 b = *reinterpret_cast<const float4*>(x + (long)(r + RL4) * ld + c4 * 4);
          float4 c = *reinterpret_cast<const float4*>(x + (long)(r + 2 * RL4) * ld + c4 * 4);
          float4 d = *reinterpret_cast<const float4*>(x + (long)(r + 3 * RL4) * ld + c4 * 4);
          s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
          s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
          s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
          s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
        }
        s0.x += s2.x; s0.y += s2.y; s0.z += s2.z; s0.w += s2.w;
        s1.x += s3.x; s1.y += s3.y; s1.z += s3.z; s1.w += s3.w;
        for (; r + RL4 < re; r += 2 * RL4) {
          float4 a = *reinterpret_cast<const float4*>(x + (long)r * ld + c4 * 4);
          float4 b = *reinterpret_cast<const float4*>(x + (long)(r + RL4) * ld + c4 * 4);
          s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
          s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (r < re) {
          float4 a = *reinterpret_cast<const float4*>(x + (long)r * ld + c4 * 4);
          s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
      }
      red4[tid] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
      __syncthreads();
      if (rl4 == 0) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < RL4; ++j) {
          float4 v = red4[j * CT4 + cl4];
          t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        *reinterpret_cast<float4*>(partial + (long)blockIdx.x * C + c4 * 4) = t;
      }
      __syncthreads();
    }
    return;
  }
  const int CT = cend - cbeg < 256 ? cend - cbeg : 256, RL = 256 / CT;
  const int cl = tid % CT, rl = tid / CT;
  for (int c = cbeg + cl; c < cend; c += CT) {
    float s = 0.f;
    if (rl < RL)
      for (int r = rb + rl; r < re; r += RL) s += x[(long)r * ld + c];
    red[tid] = (rl < RL) ? s : 0.f;
    __syncthreads();
    if (rl == 0) {
      float t = 0.f;
      for (int j = 0; j < RL; ++j) t += red[j * CT + cl];
      partial[(long)blockIdx.x * C + c] = t;
    }
    __syncthreads();
  }
}
__global__ void colsum_final_k(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out, int accumulate) {
  __shared__ float red[256];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  float s = block16_colsum(partial, nblk, (long)C, c, c < C, red);
  if (threadIdx.x >= 16 || c >= C) return;
  out[c] = accumulate ? out[c] + s : s;
}

// up to MC_MAX contiguous segments gathered into one buffer by one launch (blockIdx.y = segment)
constexpr int MC_MAX = 64;
struct MultiCopy {
  const float* src[MC_MAX];
  long dst_off[MC_MAX], len[MC_MAX];
};
__global__ void multi_copy_k(MultiCopy mc, float* __restrict__ dst) {
  const int sgm = blockIdx.y;
  const float* __restrict__ s = mc.src[sgm];
  float* __restrict__ d = dst + mc.dst_off[sgm];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < mc.len[sgm]; i += (long)gridDim.x * blockDim.x) d[i] = s[i];
}

// PixelShuffle(2)+mish: in NHWC [B,H,W,4C] (channel c*4 + i*2 + j) -> out NHWC [B,2H,2W,C].
// One thread per (input pixel, channel): the four sub-pixels of a channel are one aligned float4 of the input, and a wave (64
// consecutive channels) writes four contiguous 256-byte runs of the output.  (One thread per OUTPUT element read the input with a
// 16-byte stride, a quarter of every line per wave: 55 us for 134 MB.)
__global__ __launch_bounds__(256) void pixshuf_mish_fwd_k(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * H * W * C) return;
  const int c = (int)(i % C);
  const long p = i / C;  // b, yy, xx
  const int xx = (int)(p % W);
  const long q = p / W;
  const int yy = (int)(q % H);
  const long b = q / H;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  float* o = y + ((b * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + c;
  o[0] = mishf_(v.x);
  o[C] = mishf_(v.y);
  o[(long)2 * W * C] = mishf_(v.z);
  o[(long)2 * W * C + C] = mishf_(v.w);
}
// dx (layout of x) = dy(shuffled position) * mish'(x)
__global__ __launch_bounds__(256) void pixshuf_mish_bwd_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                          int B, int H, int W, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * H * W * C) return;
  const int c = (int)(i % C);
  const long p = i / C;
  const int xx = (int)(p % W);
  const long q = p / W;
  const int yy = (int)(q % H);
  const long b = q / H;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const float* g = dy + ((b * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + c;
  const float g0 = g[0], g1 = g[C], g2 = g[(long)2 * W * C], g3 = g[(long)2 * W * C + C];
  reinterpret_cast<float4*>(dx)[i] = make_float4(g0 * mish_gradf_(v.x), g1 * mish_gradf_(v.y), g2 * mish_gradf_(v.z), g3 * mish_gradf_(v.w));
}

// The same with dx STORED AS bf16 (the bf16 step: this gradient is read by the up-sampling convolution's dgrad / wgrad launches only,
// which round it to bf16 anyway) and, because the convolution's bias gradient is the column sum of the UNROUNDED values, per-block
// column sums in fp32 on the side: colpart [gridDim.x][4C] (NULL: none; tatt_colsum over it gives the bias gradient).  C = 64: a
// thread keeps one channel quad and walks PSB_PIX / 4 pixels.
constexpr int PSB_PIX = 64;
__global__ __launch_bounds__(256) void pixshuf_mish_bwd16_k(const float* __restrict__ x, const float* __restrict__ dy, uint2* __restrict__ dx,
                                                            float* __restrict__ colpart, int H, int W, long npix) {
  constexpr int C = 64;
  __shared__ float4 red[4][C];
  const int c = threadIdx.x & 63, pl = threadIdx.x >> 6;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int it = 0; it < PSB_PIX / 4; ++it) {
    const long p = (long)blockIdx.x * PSB_PIX + it * 4 + pl;
    if (p >= npix) break;
    const int xx = (int)(p % W);
    const long q = p / W;
    const int yy = (int)(q % H);
    const long b = q / H;
    const long i = p * C + c;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float* g = dy + ((b * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + c;
    const float g0 = g[0], g1 = g[C], g2 = g[(long)2 * W * C], g3 = g[(long)2 * W * C + C];
    const float o0 = g0 * mish_gradf_(v.x), o1 = g1 * mish_gradf_(v.y), o2 = g2 * mish_gradf_(v.z), o3 = g3 * mish_gradf_(v.w);
    dx[i] = tatt_pack_bf16x4(o0, o1, o2, o3);
    acc.x += o0; acc.y += o1; acc.z += o2; acc.w += o3;
  }
  if (!colpart) return;
  red[pl][c] = acc;
  __syncthreads();
  if (pl == 0) {
    const float4 a = red[0][c], b2 = red[1][c], c2 = red[2][c], d2 = red[3][c];
    reinterpret_cast<float4*>(colpart + (long)blockIdx.x * 4 * C)[c] =
        make_float4((a.x + b2.x) + (c2.x + d2.x), (a.y + b2.y) + (c2.y + d2.y), (a.z + b2.z) + (c2.z + d2.z), (a.w + b2.w) + (c2.w + d2.w));
  }
}

// sr (NCHW) = sigmoid(pre NHWC [P,C])
__global__ void sigmoid_to_nchw_k(const float* __restrict__ pre, float* __restrict__ sr, int B, int C, int HW) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C * HW) return;
  long hw = i % HW;
  long bc = i / HW;
  long b = bc / C, c = bc - b * C;
  sr[i] = sigmoidf_(pre[(b * HW + hw) * C + c]);
}
// dpre (NHWC) = dsr * sr * (1-sr)   (sr, dsr NCHW)
__global__ void sigmoid_bwd_to_nhwc_k(const float* __restrict__ sr, const float* __restrict__ dsr, float* __restrict__ dpre, int B,
                                      int C, int HW) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C * HW) return;
  int c = (int)(i % C);
  long p = i / C;
  long b = p / HW, hw = p - b * HW;
  long s = (b * C + c) * HW + hw;
  float v = sr[s];
  dpre[i] = dsr[s] * v * (1.f - v);
}

// NHWC max pool (kh x kw, stride = kernel)
__global__ void maxpool_fwd_k(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int kh, int kw) {
  const int OH = H / kh, OW = W / kw;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * OH * OW * C) return;
  int c = (int)(i % C);
  long p = i / C;
  int ox = (int)(p % OW);
  long q = p / OW;
  int oy = (int)(q % OH);
  long b = q / OH;
  float m = -INFINITY;
  for (int a = 0; a < kh; ++a)
    for (int d = 0; d < kw; ++d) m = fmaxf(m, x[((b * H + oy * kh + a) * W + ox * kw + d) * C + c]);
  y[i] = m;
}
// dx[window argmax (first in scan order)] = dy, other window elements 0
__global__ void maxpool_bwd_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W,
                              int C, int kh, int kw) {
  const int OH = H / kh, OW = W / kw;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * OH * OW * C) return;
  int c = (int)(i % C);
  long p = i / C;
  int ox = (int)(p % OW);
  long q = p / OW;
  int oy = (int)(q % OH);
  long b = q / OH;
  float m = -INFINITY;
  int am = 0;
  for (int a = 0; a < kh; ++a)
    for (int d = 0; d < kw; ++d) {
      float v = x[((b * H + oy * kh + a) * W + ox * kw + d) * C + c];
      if (v > m) { m = v; am = a * kw + d; }
    }
  float g = dy[i];
  for (int a = 0; a < kh; ++a)
    for (int d = 0; d < kw; ++d) dx[((b * H + oy * kh + a) * W + ox * kw + d) * C + c] = (a * kw + d == am) ? g : 0.f;
}

// text prior [B,K,1,L] with arbitrary element strides -> rows [B*L, K]
__global__ void tp_rows_k(const float* __restrict__ tp, long sB, long sK, long sL, float* __restrict__ out, int B, int K, int L) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * L * K) return;
  int k = (int)(i % K);
  long r = i / K;
  long b = r / L, l = r - b * L;
  out[i] = tp[b * sB + k * sK + l * sL];
}

// pos[(b*L+l)*64+c] = pe[l*64+c] * dropmask(index = same)
__global__ void pe_dropout_k(const float* __restrict__ pe, float* __restrict__ pos, int B, int L, DropCfg dc) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * L * 64) return;
  int c = (int)(i % 64);
  int l = (int)((i / 64) % L);
  pos[i] = pe[l * 64 + c] * tatt_drop_mask(tatt_eff_seed(dc), (uint32_t)i, dc.thresh, dc.scale);
}

// y = x * dropmask(i)   (in place allowed)
__global__ void dropout_mul_k(const float* __restrict__ x, float* __restrict__ y, long n, DropCfg dc) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = x[i] * tatt_drop_mask(tatt_eff_seed(dc), (uint32_t)i, dc.thresh, dc.scale);
}
// FFN hidden backward: dpre = dh * dropmask(i) * (h > 0)   (h = relu output before dropout)
__global__ void relu_drop_bwd_k(const float* __restrict__ h, const float* __restrict__ dh, float* __restrict__ out, long n,
                                DropCfg dc) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = h[i] > 0.f ? dh[i] * tatt_drop_mask(tatt_eff_seed(dc), (uint32_t)i, dc.thresh, dc.scale) : 0.f;
}

// seed[0] += 1 (one thread): part of a captured training step so that every replay draws fresh dropout masks
__global__ void seed_bump_k(uint32_t* seed) {
  if (threadIdx.x == 0 && blockIdx.x == 0) seed[0] += 1u;
}

}  // namespace

extern "C" {

int tatt_seed_bump(uint32_t* seed, hipStream_t st) {
  hipLaunchKernelGGL(seed_bump_k, dim3(1), dim3(64), 0, st, seed);
  TATT_LAUNCH_CHECK("seed_bump");
  return TATT_OK;
}


int tatt_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(nchw_to_nhwc_k, GRID1D((long)B * C * HW), 0, st, x, y, B, C, HW);
  TATT_LAUNCH_CHECK("nchw_to_nhwc");
  return TATT_OK;
}
int tatt_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(nhwc_to_nchw_k, GRID1D((long)B * C * HW), 0, st, x, y, B, C, HW);
  TATT_LAUNCH_CHECK("nhwc_to_nchw");
  return TATT_OK;
}
int tatt_axpby(const float* a, float alpha, const float* b, float beta, float* out, long n, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(axpby_k, GRID1D(n), 0, st, a, alpha, b, beta, out, n, accumulate);
  TATT_LAUNCH_CHECK("axpby");
  return TATT_OK;
}
int tatt_prelu_fwd(const float* x, const float* slope, float* y, long n, hipStream_t st) {
  hipLaunchKernelGGL(prelu_fwd_k, GRID1D(n), 0, st, x, slope, y, n);
  TATT_LAUNCH_CHECK("prelu_fwd");
  return TATT_OK;
}
// workspace: 1024 floats
int tatt_prelu_bwd(const float* x, const float* slope, const float* dy, float* dx, float* dslope, int accumulate, long n,
                   float* workspace, hipStream_t st) {
  int nblk = cdiv(n, 256 * 8);
  if (nblk > 1024) nblk = 1024;
  hipLaunchKernelGGL(prelu_bwd_k, dim3(nblk), dim3(256), 0, st, x, slope, dy, dx, workspace, n);
  TATT_LAUNCH_CHECK("prelu_bwd");
  hipLaunchKernelGGL(sum_partials_k, dim3(1), dim3(64), 0, st, workspace, nblk, 1.0f, dslope, accumulate);
  TATT_LAUNCH_CHECK("prelu_bwd_final");
  return TATT_OK;
}
int tatt_multi_copy(int n, const float* const* src, const long* dst_off, const long* len, float* dst, hipStream_t st) {
  TATT_CHECK_ARG(n > 0 && n <= MC_MAX && src && dst_off && len && dst, "tatt_multi_copy: 1..%d segments, non-null arrays", MC_MAX);
  MultiCopy mc;
  long mx = 0;
  for (int i = 0; i < n; ++i) {
    TATT_CHECK_ARG(src[i] && len[i] >= 0 && dst_off[i] >= 0, "tatt_multi_copy: bad segment %d", i);
    mc.src[i] = src[i]; mc.dst_off[i] = dst_off[i]; mc.len[i] = len[i];
    if (len[i] > mx) mx = len[i];
  }
  int bx = cdiv(mx, 256 * 4);
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(multi_copy_k, dim3(bx, n), dim3(256), 0, st, mc, dst);
  TATT_LAUNCH_CHECK("multi_copy");
  return TATT_OK;
}
size_t tatt_colsum_workspace(int C) { return (size_t)256 * C; }
int tatt_colsum(const float* x, long ld, int R, int C, float* out, int accumulate, float* workspace, hipStream_t st) {
  TATT_CHECK_ARG(R > 0 && C > 0 && (C > 256 || 256 % C == 0 || true), "tatt_colsum: bad dims");
  // up to 256 row blocks (more measured slower on [65536,64]) x chunks of 1024 columns, ~512 workgroups in all: wide matrices
  // (the RPE gate gradients are [4096,3072]) would otherwise leave a few blocks walking thousands of columns
  const int ncy = cdiv(C, 1024);
  int nbr = cdiv(512, ncy);
  if (nbr > 256) nbr = 256;
  if (nbr > cdiv(R, 16)) nbr = cdiv(R, 16);
  const int rpb = cdiv(R, nbr);
  const int nblk = cdiv(R, rpb);
  hipLaunchKernelGGL(colsum_partial_k, dim3(nblk, ncy), dim3(256), 0, st, x, ld, R, C, rpb, workspace);
  TATT_LAUNCH_CHECK("colsum_partial");
  hipLaunchKernelGGL(colsum_final_k, dim3(cdiv(C, 16)), dim3(256), 0, st, workspace, nblk, C, out, accumulate);
  TATT_LAUNCH_CHECK("colsum_final");
  return TATT_OK;
}
int tatt_pixshuf_mish_fwd(const float* x, float* y, int B, int H, int W, int C, hipStream_t st) {
  TATT_CHECK_ARG((((uintptr_t)x) & 15) == 0, "tatt_pixshuf_mish_fwd: x must be 16-byte aligned");
  hipLaunchKernelGGL(pixshuf_mish_fwd_k, GRID1D((long)B * H * W * C), 0, st, x, y, B, H, W, C);
  TATT_LAUNCH_CHECK("pixshuf_mish_fwd");
  return TATT_OK;
}
int tatt_pixshuf_mish_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, hipStream_t st) {
  TATT_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)dx)) & 15) == 0, "tatt_pixshuf_mish_bwd: x and dx must be 16-byte aligned");
  hipLaunchKernelGGL(pixshuf_mish_bwd_k, GRID1D((long)B * H * W * C), 0, st, x, dy, dx, B, H, W, C);
  TATT_LAUNCH_CHECK("pixshuf_mish_bwd");
  return TATT_OK;
}
// tatt_pixshuf_mish_bwd with dx stored as bf16; colpart (optional): tatt_pixshuf_colpart_rows(B, H, W) x 4C fp32 column sums of the
// unrounded values, one row per block of 64 pixels
size_t tatt_pixshuf_colpart_rows(int B, int H, int W) { return (size_t)cdiv((long)B * H * W, PSB_PIX); }
int tatt_pixshuf_mish_bwd_bf16(const float* x, const float* dy, uint16_t* dx, float* colpart, int B, int H, int W, int C, hipStream_t st) {
  TATT_CHECK_ARG(C == 64 && B > 0 && H > 0 && W > 0, "tatt_pixshuf_mish_bwd_bf16: built for C = 64 (got %d)", C);
  TATT_CHECK_ARG((((uintptr_t)x) & 15) == 0 && (((uintptr_t)dx) & 7) == 0 && (((uintptr_t)colpart) & 15) == 0, "tatt_pixshuf_mish_bwd_bf16: x / colpart 16-byte, dx 8-byte aligned");
  const long npix = (long)B * H * W;
  hipLaunchKernelGGL(pixshuf_mish_bwd16_k, dim3(cdiv(npix, PSB_PIX)), dim3(256), 0, st, x, dy, reinterpret_cast<uint2*>(dx), colpart, H, W, npix);
  TATT_LAUNCH_CHECK("pixshuf_mish_bwd<bf16 out>");
  return TATT_OK;
}
int tatt_sigmoid_to_nchw(const float* pre, float* sr, int B, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(sigmoid_to_nchw_k, GRID1D((long)B * C * HW), 0, st, pre, sr, B, C, HW);
  TATT_LAUNCH_CHECK("sigmoid_to_nchw");
  return TATT_OK;
}
int tatt_sigmoid_bwd_to_nhwc(const float* sr, const float* dsr, float* dpre, int B, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(sigmoid_bwd_to_nhwc_k, GRID1D((long)B * C * HW), 0, st, sr, dsr, dpre, B, C, HW);
  TATT_LAUNCH_CHECK("sigmoid_bwd_to_nhwc");
  return TATT_OK;
}
int tatt_maxpool_fwd(const float* x, float* y, int B, int H, int W, int C, int kh, int kw, hipStream_t st) {
  TATT_CHECK_ARG(H % kh == 0 && W % kw == 0, "tatt_maxpool_fwd: H/W must divide");
  hipLaunchKernelGGL(maxpool_fwd_k, GRID1D((long)B * (H / kh) * (W / kw) * C), 0, st, x, y, B, H, W, C, kh, kw);
  TATT_LAUNCH_CHECK("maxpool_fwd");
  return TATT_OK;
}
int tatt_maxpool_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int kh, int kw, hipStream_t st) {
  hipLaunchKernelGGL(maxpool_bwd_k, GRID1D((long)B * (H / kh) * (W / kw) * C), 0, st, x, dy, dx, B, H, W, C, kh, kw);
  TATT_LAUNCH_CHECK("maxpool_bwd");
  return TATT_OK;
}
int tatt_tp_rows(const float* tp, long sB, long sK, long sL, float* out, int B, int K, int L, hipStream_t st) {
  hipLaunchKernelGGL(tp_rows_k, GRID1D((long)B * L * K), 0, st, tp, sB, sK, sL, out, B, K, L);
  TATT_LAUNCH_CHECK("tp_rows");
  return TATT_OK;
}
int tatt_pe_dropout(const float* pe, float* pos, int B, int L, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, hipStream_t st) {
  DropCfg dc{seed_site, thresh, scale, seed_dev};
  hipLaunchKernelGGL(pe_dropout_k, GRID1D((long)B * L * 64), 0, st, pe, pos, B, L, dc);
  TATT_LAUNCH_CHECK("pe_dropout");
  return TATT_OK;
}
int tatt_dropout_mul(const float* x, float* y, long n, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, hipStream_t st) {
  DropCfg dc{seed_site, thresh, scale, seed_dev};
  hipLaunchKernelGGL(dropout_mul_k, GRID1D(n), 0, st, x, y, n, dc);
  TATT_LAUNCH_CHECK("dropout_mul");
  return TATT_OK;
}
int tatt_relu_drop_bwd(const float* h, const float* dh, float* out, long n, uint32_t seed_site, uint32_t thresh, float scale,
                       const uint32_t* seed_dev, hipStream_t st) {
  DropCfg dc{seed_site, thresh, scale, seed_dev};
  hipLaunchKernelGGL(relu_drop_bwd_k, GRID1D(n), 0, st, h, dh, out, n, dc);
  TATT_LAUNCH_CHECK("relu_drop_bwd");
  return TATT_OK;
}

}  // extern "C"
