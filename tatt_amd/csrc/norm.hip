// BatchNorm (training/eval, NHWC rows x C) and LayerNorm (C = 64, one wave per row) kernels, fwd + bwd.
#include "common.h"

namespace {

constexpr int BN_MAXBLK = 256;
constexpr int BN_RPT = 8;  // rows per thread of the apply kernels
inline int bn_rows_per_block(int C) {
  const int c4 = C / 4, CT = c4 < 256 ? c4 : 256, RL = 256 / CT;
  return RL * BN_RPT;
}
inline int bn_apply_grid(int R, int C) {
  const int c4 = C / 4, CT = c4 < 256 ? c4 : 256, RL = 256 / CT;
  return cdiv(R, RL * BN_RPT);
}

// ---- BatchNorm statistics: per-block (mean, M2) over a row range, merged with Chan's formula -------------
// partial layout: [nblk][2][C]
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ x, long ldx, int R, int C, int rows_per_blk,
                                                        float* __restrict__ partial) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int rb = blockIdx.x * rows_per_blk, re = min(R, rb + rows_per_blk);
  const int cnt = re - rb;
  if ((C & 3) == 0 && (ldx & 3) == 0 && ((((uintptr_t)x) & 15) == 0) && C <= 1024) {
    // vector path: thread = (float4 channel group, row lane); second pass re-reads the block's rows (L2 resident)
    __shared__ float4 red4[256];
    const int C4 = C >> 2, RL4 = 256 / C4;
    const int cl4 = tid % C4, rl4 = tid / C4;
    const bool on = rl4 < RL4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on)
      for (int r = rb + rl4; r < re; r += RL4) {
        float4 a = *reinterpret_cast<const float4*>(x + (long)r * ldx + cl4 * 4);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      }
    red4[tid] = s;
    __syncthreads();
    float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
      for (int j = 0; j < RL4; ++j) {
        float4 v = red4[j * C4 + cl4];
        mean.x += v.x; mean.y += v.y; mean.z += v.z; mean.w += v.w;
      }
      const float ic = 1.f / (float)cnt;
      mean.x *= ic; mean.y *= ic; mean.z *= ic; mean.w *= ic;
    }
    __syncthreads();
    float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on)
      for (int r = rb + rl4; r < re; r += RL4) {
        float4 a = *reinterpret_cast<const float4*>(x + (long)r * ldx + cl4 * 4);
        float dx = a.x - mean.x, dy = a.y - mean.y, dz = a.z - mean.z, dw = a.w - mean.w;
        m2.x += dx * dx; m2.y += dy * dy; m2.z += dz * dz; m2.w += dw * dw;
      }
    red4[tid] = m2;
    __syncthreads();
    if (on && rl4 == 0) {
      float4 t2 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j < RL4; ++j) {
        float4 v = red4[j * C4 + cl4];
        t2.x += v.x; t2.y += v.y; t2.z += v.z; t2.w += v.w;
      }
      *reinterpret_cast<float4*>(partial + ((long)blockIdx.x * 2 + 0) * C + cl4 * 4) = mean;
      *reinterpret_cast<float4*>(partial + ((long)blockIdx.x * 2 + 1) * C + cl4 * 4) = t2;
    }
    return;
  }
  const int CT = C < 256 ? C : 256, RL = 256 / CT;
  const int cl = tid % CT, rl = tid / CT;
  for (int c = cl; c < C; c += CT) {
    float s = 0.f;
    if (rl < RL)
      for (int r = rb + rl; r < re; r += RL) s += x[(long)r * ldx + c];
    red[tid] = (rl < RL) ? s : 0.f;
    __syncthreads();
    float tot = 0.f;
    for (int j = 0; j < RL; ++j) tot += red[j * CT + cl];
    float mean = tot / (float)cnt;
    __syncthreads();
    float m2 = 0.f;
    if (rl < RL)
      for (int r = rb + rl; r < re; r += RL) {
        float d = x[(long)r * ldx + c] - mean;
        m2 += d * d;
      }
    red[tid] = (rl < RL) ? m2 : 0.f;
    __syncthreads();
    if (rl == 0) {
      float t2 = 0.f;
      for (int j = 0; j < RL; ++j) t2 += red[j * CT + cl];
      partial[((long)blockIdx.x * 2 + 0) * C + c] = mean;
      partial[((long)blockIdx.x * 2 + 1) * C + c] = t2;
    }
    __syncthreads();
  }
}

// stat layout: [2][C] = mean, invstd.  running stats updated with momentum (unbiased var).
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
  if (nb == 0.f) return;
  const float d = mb - mean, nn = n + nb;
  mean += d * (nb / nn);
  m2 += m2b + d * d * (n * nb / nn);
  n = nn;
}

// block = 16 channels x (blockDim.x / 16) row lanes (64 with 1024 threads: the conv epilogues leave one partial per 128 rows, 512 of
// them at B = 64): each lane Chan-merges a strided subset of the row blocks, then 8 lanes merge 8 lanes each, then lane 0 those 8
// (cpb = channels per block: 16 with 16 row lanes for few row blocks, 4 with 64 row lanes for many -- always a four-wave workgroup: a
// sixteen-wave one waits for a CU with sixteen free wave slots while the weight-gradient strands and the RPE chain hold theirs)
__global__ __launch_bounds__(256) void bn_stats_final(const float* __restrict__ partial, int nblk, int R, int C, int rows_per_blk, float eps,
                                                       float momentum, float* __restrict__ stat, float* __restrict__ run_mean,
                                                       float* __restrict__ run_var, int cpb) {
  __shared__ float sn[256], sm[256], s2[256];
  const int RLN = blockDim.x / cpb;
  const int cl = threadIdx.x % cpb, rl = threadIdx.x / cpb;
  const int c = blockIdx.x * cpb + cl;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  if (c < C)
    for (int b = rl; b < nblk; b += RLN) {
      const float nb = (float)(min(R, (b + 1) * rows_per_blk) - b * rows_per_blk);
      chan_merge(n, mean, m2, nb, partial[((long)b * 2 + 0) * C + c], partial[((long)b * 2 + 1) * C + c]);
    }
  sn[threadIdx.x] = n; sm[threadIdx.x] = mean; s2[threadIdx.x] = m2;
  __syncthreads();
  if (RLN > 8) {  // second level: lanes 0..7 merge every 8th lane's result
    if (rl < 8) {
      n = 0.f; mean = 0.f; m2 = 0.f;
      for (int j = rl; j < RLN; j += 8) chan_merge(n, mean, m2, sn[j * cpb + cl], sm[j * cpb + cl], s2[j * cpb + cl]);
    }
    __syncthreads();
    if (rl < 8) { sn[threadIdx.x] = n; sm[threadIdx.x] = mean; s2[threadIdx.x] = m2; }
    __syncthreads();
  }
  if (rl != 0 || c >= C) return;
  n = 0.f; mean = 0.f; m2 = 0.f;
  const int last = RLN > 8 ? 8 : RLN;
  for (int j = 0; j < last; ++j) chan_merge(n, mean, m2, sn[j * cpb + cl], sm[j * cpb + cl], s2[j * cpb + cl]);
  float var = m2 / n;
  stat[c] = mean;
  stat[C + c] = rsqrtf(var + eps);
  if (run_mean) {
    // (explicit fma: the grouped form below must round like this one whatever the compiler contracts)
    run_mean[c] = fmaf(momentum, mean, (1.f - momentum) * run_mean[c]);
    float unb = n > 1.f ? m2 / (n - 1.f) : var;
    run_var[c] = fmaf(momentum, unb, (1.f - momentum) * run_var[c]);
  }
}

// The same merge for ng statistics groups in ONE launch (the two SR passes of a batched TSC step: one convolution launch, BatchNorm
// per pass): partial [ng][nblk][2][C], stat [ng][2][C]; the running statistics take the groups' updates one after the other, in
// group order, exactly as ng launches of bn_stats_final would leave them.
__global__ __launch_bounds__(256) void bn_stats_final_groups(const float* __restrict__ partial, int nblk, int ng, int R, int C, int rows_per_blk,
                                                              float eps, float momentum, float* __restrict__ stat, float* __restrict__ run_mean,
                                                              float* __restrict__ run_var, int cpb) {
  __shared__ float sn[256], sm[256], s2[256];
  const int RLN = blockDim.x / cpb;
  const int cl = threadIdx.x % cpb, rl = threadIdx.x / cpb;
  const int c = blockIdx.x * cpb + cl;
  const bool fin = rl == 0 && c < C;
  float rmean = 0.f, rvar = 0.f;
  if (fin && run_mean) {
    rmean = run_mean[c];
    rvar = run_var[c];
  }
  for (int g = 0; g < ng; ++g) {
    const float* __restrict__ pg = partial + (long)g * nblk * 2 * C;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (c < C)
      for (int b = rl; b < nblk; b += RLN) {
        const float nb = (float)(min(R, (b + 1) * rows_per_blk) - b * rows_per_blk);
        chan_merge(n, mean, m2, nb, pg[((long)b * 2 + 0) * C + c], pg[((long)b * 2 + 1) * C + c]);
      }
    sn[threadIdx.x] = n; sm[threadIdx.x] = mean; s2[threadIdx.x] = m2;
    __syncthreads();
    if (RLN > 8) {
      if (rl < 8) {
        n = 0.f; mean = 0.f; m2 = 0.f;
        for (int j = rl; j < RLN; j += 8) chan_merge(n, mean, m2, sn[j * cpb + cl], sm[j * cpb + cl], s2[j * cpb + cl]);
      }
      __syncthreads();
      if (rl < 8) { sn[threadIdx.x] = n; sm[threadIdx.x] = mean; s2[threadIdx.x] = m2; }
      __syncthreads();
    }
    if (fin) {
      n = 0.f; mean = 0.f; m2 = 0.f;
      const int last = RLN > 8 ? 8 : RLN;
      for (int j = 0; j < last; ++j) chan_merge(n, mean, m2, sn[j * cpb + cl], sm[j * cpb + cl], s2[j * cpb + cl]);
      const float var = m2 / n;
      stat[((long)g * 2 + 0) * C + c] = mean;
      stat[((long)g * 2 + 1) * C + c] = rsqrtf(var + eps);
      const float unb = n > 1.f ? m2 / (n - 1.f) : var;
      rmean = fmaf(momentum, mean, (1.f - momentum) * rmean);
      rvar = fmaf(momentum, unb, (1.f - momentum) * rvar);
    }
    __syncthreads();  // the scratch is reused by the next group
  }
  if (fin && run_mean) {
    run_mean[c] = rmean;
    run_var[c] = rvar;
  }
}

// eval mode: stat from running stats
__global__ void bn_stat_from_running(const float* __restrict__ rm, const float* __restrict__ rv, int C, float eps,
                                     float* __restrict__ stat) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  stat[c] = rm[c];
  stat[C + c] = rsqrtf(rv[c] + eps);
}

// act: 0 none, 1 relu, 2 mish.  y = act(bn(x)) (+ res).  ACT is a template parameter and the row tail is handled by clamping the
// load row (stores predicated): with a run-time `act` and `if (r < R)` around every load the compiler emitted a basic block per
// load with vmcnt(0) behind it -- the loads went out one at a time.
// OUT16: y is stored as bf16 (ldy in elements): the bf16 step keeps tensors that only bf16 matrix-core launches read in bf16.
template <int ACT, bool OUT16 = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ stat,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ res, long ldr, float* __restrict__ y, long ldy, int R, int C, int Rg) {
  // Rg: rows per statistics group (R for one set; a multiple of the block's rows): stat [R / Rg][2][C]
  // thread = (float4 channel group, row lane); the channel group's scale/shift is folded once and kept in registers while the
  // thread walks BN_RPT rows (one thread per float4 re-read 16 parameters for every 16 bytes of payload)
  const int c4 = C >> 2, CT = c4 < 256 ? c4 : 256, RL = 256 / CT;
  const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
  if (rl >= RL) return;
  const int rbase = blockIdx.x * (RL * BN_RPT) + rl;
  stat += (long)((blockIdx.x * (RL * BN_RPT)) / Rg) * 2 * C;
  for (int cg = cl; cg < c4; cg += CT) {
    const int c = cg * 4;
    float mu[4], sc[4], ga[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mu[e] = stat[c + e];
      sc[e] = stat[C + c + e];
      ga[e] = gamma[c + e];
      sh[e] = beta[c + e];
    }
    float4 v[BN_RPT], rr[BN_RPT];
#pragma unroll
    for (int k = 0; k < BN_RPT; ++k) v[k] = *reinterpret_cast<const float4*>(x + (long)min(rbase + k * RL, R - 1) * ldx + c);
    if (res) {
#pragma unroll
      for (int k = 0; k < BN_RPT; ++k) rr[k] = *reinterpret_cast<const float4*>(res + (long)min(rbase + k * RL, R - 1) * ldr + c);
    } else {
#pragma unroll
      for (int k = 0; k < BN_RPT; ++k) rr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < BN_RPT; ++k) {
      const int r = rbase + k * RL;
      const float in[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, ad[4] = {rr[k].x, rr[k].y, rr[k].z, rr[k].w};
      float out[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = (in[e] - mu[e]) * sc[e] * ga[e] + sh[e];
        if (ACT == 1) t = fmaxf(t, 0.f);
        else if (ACT == 2) t = mishf_(t);
        out[e] = t + ad[e];
      }
      if constexpr (OUT16) {
        if (r < R) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(y) + (long)r * ldy + c) = tatt_pack_bf16x4(out[0], out[1], out[2], out[3]);
      } else {
        if (r < R) *reinterpret_cast<float4*>(y + (long)r * ldy + c) = make_float4(out[0], out[1], out[2], out[3]);
      }
    }
  }
}

__device__ __forceinline__ float act_grad(float t, int act) {
  if (act == 1) return t > 0.f ? 1.f : 0.f;
  if (act == 2) return mish_gradf_(t);
  return 1.f;
}

// partial sums of g = dy*act'(bn(x)) and g*xhat : partial [nblk][2][C]
__global__ __launch_bounds__(256) void bn_bwd_reduce(const float* __restrict__ x, long ldx, const float* __restrict__ dy, long lddy,
                                                     const float* __restrict__ stat, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int R, int C, int act, int rows_per_blk,
                                                     float* __restrict__ partial) {
  __shared__ float red[2][256];
  const int tid = threadIdx.x;
  const int rb = blockIdx.x * rows_per_blk, re = min(R, rb + rows_per_blk);
  if ((C & 3) == 0 && (ldx & 3) == 0 && (lddy & 3) == 0 && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)dy) & 15) == 0) && C <= 1024) {
    __shared__ float4 r1[256], r2[256];
    const int C4 = C >> 2, RL4 = 256 / C4;
    const int cl4 = tid % C4, rl4 = tid / C4;
    const bool on = rl4 < RL4;
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    if (on) {
      float mean[4], inv[4], ga[4], be[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mean[e] = stat[cl4 * 4 + e]; inv[e] = stat[C + cl4 * 4 + e]; ga[e] = gamma[cl4 * 4 + e]; be[e] = beta[cl4 * 4 + e];
      }
      auto acc1 = [&](const float4& xv, const float4& gv) {
        const float xi[4] = {xv.x, xv.y, xv.z, xv.w}, gi[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xh = (xi[e] - mean[e]) * inv[e];
          float g = gi[e] * act_grad(xh * ga[e] + be[e], act);
          a1[e] += g;
          a2[e] += g * xh;
        }
      };
      int r = rb + rl4;
      for (; r + 3 * RL4 < re; r += 4 * RL4) {  // eight 16-byte loads in flight per thread (one wave per SIMD: nothing else hides them)
        float4 xv[4], gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xv[k] = *reinterpret_cast<const float4*>(x + (long)(r + k * RL4) * ldx + cl4 * 4);
          gv[k] = *reinterpret_cast<const float4*>(dy + (long)(r + k * RL4) * lddy + cl4 * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc1(xv[k], gv[k]);
      }
      for (; r < re; r += RL4)
        acc1(*reinterpret_cast<const float4*>(x + (long)r * ldx + cl4 * 4), *reinterpret_cast<const float4*>(dy + (long)r * lddy + cl4 * 4));
    }
    r1[tid] = make_float4(a1[0], a1[1], a1[2], a1[3]);
    r2[tid] = make_float4(a2[0], a2[1], a2[2], a2[3]);
    __syncthreads();
    if (on && rl4 == 0) {
      float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
      for (int j = 0; j < RL4; ++j) {
        float4 u = r1[j * C4 + cl4], v = r2[j * C4 + cl4];
        t1.x += u.x; t1.y += u.y; t1.z += u.z; t1.w += u.w;
        t2.x += v.x; t2.y += v.y; t2.z += v.z; t2.w += v.w;
      }
      *reinterpret_cast<float4*>(partial + ((long)blockIdx.x * 2 + 0) * C + cl4 * 4) = t1;
      *reinterpret_cast<float4*>(partial + ((long)blockIdx.x * 2 + 1) * C + cl4 * 4) = t2;
    }
    return;
  }
  const int CT = C < 256 ? C : 256, RL = 256 / CT;
  const int cl = tid % CT, rl = tid / CT;
  for (int c = cl; c < C; c += CT) {
    float s1 = 0.f, s2 = 0.f;
    if (rl < RL) {
      float mean = stat[c], inv = stat[C + c], ga = gamma[c], be = beta[c];
      for (int r = rb + rl; r < re; r += RL) {
        float xh = (x[(long)r * ldx + c] - mean) * inv;
        float g = dy[(long)r * lddy + c] * act_grad(xh * ga + be, act);
        s1 += g;
        s2 += g * xh;
      }
    }
    red[0][tid] = s1;
    red[1][tid] = s2;
    __syncthreads();
    if (rl == 0) {
      float t1 = 0.f, t2 = 0.f;
      for (int j = 0; j < RL; ++j) {
        t1 += red[0][j * CT + cl];
        t2 += red[1][j * CT + cl];
      }
      partial[((long)blockIdx.x * 2 + 0) * C + c] = t1;
      partial[((long)blockIdx.x * 2 + 1) * C + c] = t2;
    }
    __syncthreads();
  }
}

// sums[2][C]; dgamma/dbeta (+)=.  block = 16 channels x (blockDim.x / 16) row lanes, every lane's loads independent of one another
__global__ __launch_bounds__(256) void bn_bwd_final(const float* __restrict__ partial, int nblk, int C, float* __restrict__ sums,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, int cpb) {
  __shared__ float r1[256], r2[256];
  const int RLN = blockDim.x / cpb;
  const int cl = threadIdx.x % cpb, rl = threadIdx.x / cpb;
  const int c = blockIdx.x * cpb + cl;
  float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
  if (c < C) {
    int b = rl;
    for (; b + RLN < nblk; b += 2 * RLN) {
      a1 += partial[((long)b * 2 + 0) * C + c];
      a2 += partial[((long)b * 2 + 1) * C + c];
      b1 += partial[((long)(b + RLN) * 2 + 0) * C + c];
      b2 += partial[((long)(b + RLN) * 2 + 1) * C + c];
    }
    if (b < nblk) {
      a1 += partial[((long)b * 2 + 0) * C + c];
      a2 += partial[((long)b * 2 + 1) * C + c];
    }
  }
  r1[threadIdx.x] = a1 + b1;
  r2[threadIdx.x] = a2 + b2;
  __syncthreads();
  if (RLN > 8) {
    float t1 = 0.f, t2 = 0.f;
    if (rl < 8)
      for (int j = rl; j < RLN; j += 8) { t1 += r1[j * cpb + cl]; t2 += r2[j * cpb + cl]; }
    __syncthreads();
    if (rl < 8) { r1[threadIdx.x] = t1; r2[threadIdx.x] = t2; }
    __syncthreads();
  }
  if (rl != 0 || c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  const int last = RLN > 8 ? 8 : RLN;
  for (int j = 0; j < last; ++j) { s1 += r1[j * cpb + cl]; s2 += r2[j * cpb + cl]; }
  sums[c] = s1;
  sums[C + c] = s2;
  if (dgamma) {
    dgamma[c] = accumulate ? dgamma[c] + s2 : s2;
    dbeta[c] = accumulate ? dbeta[c] + s1 : s1;
  }
}

// dx = gamma*invstd*(g - s1/R - xhat*s2/R) in training mode; eval mode (train=0): dx = gamma*invstd*g.  Branch-free like bn_apply_kernel.
template <int ACT, bool OUT16 = false>
__global__ __launch_bounds__(256) void bn_bwd_apply(const float* __restrict__ x, long ldx, const float* __restrict__ dy, long lddy,
                                                    const float* __restrict__ stat, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ sums, float* __restrict__ dx,
                                                    long lddx, int R, int C, int train, int accumulate) {
  const int c4 = C >> 2, CT = c4 < 256 ? c4 : 256, RL = 256 / CT;
  const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
  if (rl >= RL) return;
  const int rbase = blockIdx.x * (RL * BN_RPT) + rl;
  const float invR = train ? 1.f / (float)R : 0.f;
  for (int cg = cl; cg < c4; cg += CT) {
    const int c = cg * 4;
    float mean[4], inv[4], ga[4], be[4], s1[4], s2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mean[e] = stat[c + e]; inv[e] = stat[C + c + e]; ga[e] = gamma[c + e]; be[e] = beta[c + e];
      s1[e] = sums[c + e] * invR; s2[e] = sums[C + c + e] * invR;
    }
    float4 xv[BN_RPT], gv[BN_RPT], pv[BN_RPT];
#pragma unroll
    for (int k = 0; k < BN_RPT; ++k) {
      const long r = min(rbase + k * RL, R - 1);
      xv[k] = *reinterpret_cast<const float4*>(x + r * ldx + c);
      gv[k] = *reinterpret_cast<const float4*>(dy + r * lddy + c);
    }
    if (accumulate) {
#pragma unroll
      for (int k = 0; k < BN_RPT; ++k) pv[k] = *reinterpret_cast<const float4*>(dx + (long)min(rbase + k * RL, R - 1) * lddx + c);
    } else {
#pragma unroll
      for (int k = 0; k < BN_RPT; ++k) pv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < BN_RPT; ++k) {
      const int r = rbase + k * RL;
      const float xi[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w}, gi[4] = {gv[k].x, gv[k].y, gv[k].z, gv[k].w};
      const float pi[4] = {pv[k].x, pv[k].y, pv[k].z, pv[k].w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xi[e] - mean[e]) * inv[e];
        const float g = gi[e] * act_grad(xh * ga[e] + be[e], ACT);
        o[e] = ga[e] * inv[e] * (g - s1[e] - xh * s2[e]) + pi[e];
      }
      if constexpr (OUT16) {
        if (r < R) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(dx) + (long)r * lddx + c) = tatt_pack_bf16x4(o[0], o[1], o[2], o[3]);
      } else {
        if (r < R) *reinterpret_cast<float4*>(dx + (long)r * lddx + c) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// ---- LayerNorm over 64 channels: one wave per row, lane = channel ------------------------------------------
// v = x + dropmask*res (res optional, drop index = row*64 + c) ; y = LN(v)*gamma + beta ; vout (optional) = v
// optional second output: y2 accumulation  yacc = yscale * y (+ yacc)  (used for the mean over decoder layers)
__device__ __forceinline__ float sum16(float v) {  // sum over the 16 lanes that share a row
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 16 lanes per row, 4 channels (one 16-byte access) per lane: a wave works on 4 rows at a time
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res, DropCfg dc,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ vout, float* __restrict__ y, int rows, float eps) {
  const int lane = threadIdx.x & 63, cg = lane & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + (lane >> 4);
  const bool ok = row < rows;
  const long i = (long)(ok ? row : 0) * 64 + cg * 4;
  float4 xv = *reinterpret_cast<const float4*>(x + i);
  float v[4] = {xv.x, xv.y, xv.z, xv.w};
  if (res) {
    const float4 rv = *reinterpret_cast<const float4*>(res + i);
    const float r[4] = {rv.x, rv.y, rv.z, rv.w};
    const uint32_t ss = tatt_eff_seed(dc);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += r[e] * tatt_drop_mask(ss, (uint32_t)(i + e), dc.thresh, dc.scale);
  }
  const float mean = sum16(v[0] + v[1] + v[2] + v[3]) * (1.f / 64.f);
  float d[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e] = v[e] - mean;
  const float var = sum16(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 64.f);
  const float rstd = rsqrtf(var + eps);
  if (!ok) return;
  const float4 ga = *reinterpret_cast<const float4*>(gamma + cg * 4), be = *reinterpret_cast<const float4*>(beta + cg * 4);
  if (vout) *reinterpret_cast<float4*>(vout + i) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(y + i) =
      make_float4(d[0] * rstd * ga.x + be.x, d[1] * rstd * ga.y + be.y, d[2] * rstd * ga.z + be.z, d[3] * rstd * ga.w + be.w);
}

// dv = rstd*(gamma*dy - mean(gamma*dy) - xhat*mean(gamma*dy*xhat)); dv (+)= ; partial dgamma/dbeta per block.
// dres (optional) = dv_this * dropmask  (written, not accumulated)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ v, const float* __restrict__ dy, float dyscale,
                                                     const float* __restrict__ gamma, float* __restrict__ dv, int dv_accumulate,
                                                     float* __restrict__ dres, DropCfg dc, float* __restrict__ partial, int rows,
                                                     int rows_per_blk, float eps) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int cg = lane & 15, rq = lane >> 4;
  const int rb = blockIdx.x * rows_per_blk, re = min(rows, rb + rows_per_blk);
  const float4 gav = *reinterpret_cast<const float4*>(gamma + cg * 4);
  const float ga[4] = {gav.x, gav.y, gav.z, gav.w};
  const uint32_t ss = tatt_eff_seed(dc);
  float sg[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
  for (int row0 = rb + w * 4; row0 < re; row0 += 16) {
    const int row = row0 + rq;
    const bool ok = row < re;
    const long i = (long)(ok ? row : rb) * 64 + cg * 4;
    const float4 vv4 = *reinterpret_cast<const float4*>(v + i), gy4 = *reinterpret_cast<const float4*>(dy + i);
    const float vv[4] = {vv4.x, vv4.y, vv4.z, vv4.w}, gy[4] = {gy4.x, gy4.y, gy4.z, gy4.w};
    const float mean = sum16(vv[0] + vv[1] + vv[2] + vv[3]) * (1.f / 64.f);
    float d[4], xh[4], gg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = vv[e] - mean;
    const float var = sum16(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 64.f);
    const float rstd = rsqrtf(var + eps);
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xh[e] = d[e] * rstd;
      const float g = ok ? gy[e] * dyscale : 0.f;
      sg[e] += g * xh[e];
      sb[e] += g;
      gg[e] = g * ga[e];
      p1 += gg[e];
      p2 += gg[e] * xh[e];
    }
    const float m1 = sum16(p1) * (1.f / 64.f), m2 = sum16(p2) * (1.f / 64.f);
    if (ok) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rstd * (gg[e] - m1 - xh[e] * m2);
      if (dres) {
        *reinterpret_cast<float4*>(dres + i) = make_float4(o[0] * tatt_drop_mask(ss, (uint32_t)i, dc.thresh, dc.scale),
                                                           o[1] * tatt_drop_mask(ss, (uint32_t)(i + 1), dc.thresh, dc.scale),
                                                           o[2] * tatt_drop_mask(ss, (uint32_t)(i + 2), dc.thresh, dc.scale),
                                                           o[3] * tatt_drop_mask(ss, (uint32_t)(i + 3), dc.thresh, dc.scale));
      }
      if (dv_accumulate) {
        const float4 pv = *reinterpret_cast<const float4*>(dv + i);
        o[0] += pv.x; o[1] += pv.y; o[2] += pv.z; o[3] += pv.w;
      }
      *reinterpret_cast<float4*>(dv + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {  // the four rows of the wave, then the four waves
    sg[e] += __shfl_xor(sg[e], 16, 64);
    sg[e] += __shfl_xor(sg[e], 32, 64);
    sb[e] += __shfl_xor(sb[e], 16, 64);
    sb[e] += __shfl_xor(sb[e], 32, 64);
  }
  if (rq == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][w][cg * 4 + e] = sg[e];
      red[1][w][cg * 4 + e] = sb[e];
    }
  }
  __syncthreads();
  if (w == 0) {
    partial[((long)blockIdx.x * 2 + 0) * 64 + lane] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    partial[((long)blockIdx.x * 2 + 1) * 64 + lane] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
  }
}

__global__ void ln_bwd_final(const float* __restrict__ partial, int nblk, float* __restrict__ dgamma, float* __restrict__ dbeta,
                             int accumulate) {
  __shared__ float red[256];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  float a = block16_colsum(partial, nblk, 128L, c, true, red);
  float b = block16_colsum(partial + 64, nblk, 128L, c, true, red);
  if (threadIdx.x >= 16) return;
  dgamma[c] = accumulate ? dgamma[c] + a : a;
  dbeta[c] = accumulate ? dbeta[c] + b : b;
}

inline void bn_blocking(int R, int* nblk, int* rpb) {
  int rows_per_blk = R >= 64 * BN_MAXBLK ? cdiv(R, BN_MAXBLK) : 64;
  *rpb = rows_per_blk;
  *nblk = cdiv(R, rows_per_blk);
}

// ---- BatchNorm(training) + ReLU + MaxPool of the STN head's small maps as ONE launch per direction ------------------------------
// The last four convolution stages of the STN head (model/stn_head.py:37-62: maps of 4x16 and smaller) and its fc1 BatchNorm work on
// at most 4096 rows; as statistics-partial / statistics-final / apply / max-pool launches they are four dependent launches of a
// few microseconds each on the critical path of the step, backward as many.  Channels are independent, so a workgroup that owns 4
// channels needs no other workgroup: it sums its [R,4] strip (two passes over registers: mean, then M2), updates the running
// statistics, applies gamma / beta / ReLU, writes the activation and the pooled map.  Thread = (float4 channel lane, window lane);
// a thread keeps its pooling windows in registers (at most SM_NW windows of KH*KW rows).
// Four-wave workgroups of four channels: big workgroups (1024 threads, 8-16 channels) were faster alone but waited for a CU with
// sixteen free wave slots while the RPE chain and the weight-gradient strands hold theirs -- the step got slower, not faster.
constexpr int SM_CL = 1, SM_CH = 4 * SM_CL, SM_NW = 4, SM_THR = 256, SM_WL = SM_THR / SM_CL;

__device__ __forceinline__ float4 sm_bsum4(float4 v, float4* red, int tid) {  // sum over the window lanes of one channel lane
#pragma unroll
  for (int o = SM_CL; o < 64; o <<= 1) {
    v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
  }
  if ((tid & 63) < SM_CL) red[(tid >> 6) * SM_CL + (tid & (SM_CL - 1))] = v;
  __syncthreads();
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int w = 0; w < SM_THR / 64; ++w) {
    const float4 u = red[w * SM_CL + (tid & (SM_CL - 1))];
    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
  }
  __syncthreads();
  return t;
}

template <int KH, int KW>
__global__ __launch_bounds__(SM_THR) void bn_pool_small_fwd(const float* __restrict__ x, float* __restrict__ stat, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ run_mean,
                                                            float* __restrict__ run_var, float* __restrict__ act_out,
                                                            float* __restrict__ pooled, int B, int H, int W, int C, float eps, float momentum) {
  __shared__ float4 red[SM_THR / 64 * SM_CL];
  constexpr int NR = KH * KW;
  const int tid = threadIdx.x, cl = tid & (SM_CL - 1), wl = tid / SM_CL;
  const int c = blockIdx.x * SM_CH + cl * 4;
  const int OH = H / KH, OW = W / KW, nwin = B * OH * OW;
  float4 v[SM_NW][NR];
  int row0[SM_NW];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < SM_NW; ++k) {
    const int w = wl + k * SM_WL;
    const bool on = w < nwin;
    const int wc = on ? w : 0;
    const int ox = wc % OW, q = wc / OW, oy = q % OH, b = q / OH;
    row0[k] = (b * H + oy * KH) * W + ox * KW;
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
      for (int d = 0; d < KW; ++d) {
        float4 t = *reinterpret_cast<const float4*>(x + (long)(row0[k] + a * W + d) * C + c);
        if (!on) t = make_float4(0.f, 0.f, 0.f, 0.f);
        v[k][a * KW + d] = t;
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
  }
  const float R = (float)(nwin * NR);
  s = sm_bsum4(s, red, tid);
  const float4 mean = make_float4(s.x / R, s.y / R, s.z / R, s.w / R);
  float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < SM_NW; ++k) {
    if (wl + k * SM_WL < nwin) {
#pragma unroll
      for (int e = 0; e < NR; ++e) {
        const float dx = v[k][e].x - mean.x, dy = v[k][e].y - mean.y, dz = v[k][e].z - mean.z, dw = v[k][e].w - mean.w;
        m2.x += dx * dx; m2.y += dy * dy; m2.z += dz * dz; m2.w += dw * dw;
      }
    }
  }
  m2 = sm_bsum4(m2, red, tid);
  const float4 inv = make_float4(rsqrtf(m2.x / R + eps), rsqrtf(m2.y / R + eps), rsqrtf(m2.z / R + eps), rsqrtf(m2.w / R + eps));
  if (wl == 0) {
    *reinterpret_cast<float4*>(stat + c) = mean;
    *reinterpret_cast<float4*>(stat + C + c) = inv;
    if (run_mean) {
      const float mu[4] = {mean.x, mean.y, mean.z, mean.w}, q2[4] = {m2.x, m2.y, m2.z, m2.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        run_mean[c + e] = (1.f - momentum) * run_mean[c + e] + momentum * mu[e];
        const float unb = R > 1.f ? q2[e] / (R - 1.f) : q2[e] / R;
        run_var[c + e] = (1.f - momentum) * run_var[c + e] + momentum * unb;
      }
    }
  }
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
#pragma unroll
  for (int k = 0; k < SM_NW; ++k) {
    const int w = wl + k * SM_WL;
    if (w >= nwin) continue;
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
      for (int d = 0; d < KW; ++d) {
        const float4 t = v[k][a * KW + d];
        float4 o;
        o.x = fmaxf((t.x - mean.x) * inv.x * ga.x + be.x, 0.f);
        o.y = fmaxf((t.y - mean.y) * inv.y * ga.y + be.y, 0.f);
        o.z = fmaxf((t.z - mean.z) * inv.z * ga.z + be.z, 0.f);
        o.w = fmaxf((t.w - mean.w) * inv.w * ga.w + be.w, 0.f);
        *reinterpret_cast<float4*>(act_out + (long)(row0[k] + a * W + d) * C + c) = o;
        mx.x = fmaxf(mx.x, o.x); mx.y = fmaxf(mx.y, o.y); mx.z = fmaxf(mx.z, o.z); mx.w = fmaxf(mx.w, o.w);
      }
    if (NR > 1) *reinterpret_cast<float4*>(pooled + (long)w * C + c) = mx;
  }
}

// backward of the same: dpooled [nwin,C] -> gradient of the max-pool (first maximum in scan order, as tatt_maxpool_bwd) -> ReLU
// mask (act > 0) -> BatchNorm backward in training mode; dgamma / dbeta are ACCUMULATED.  Like the forward kernel the thread keeps its
// windows in registers (g and xhat in place of the loaded activation and input): one round of loads, all of them in flight together.
template <int KH, int KW>
__global__ __launch_bounds__(SM_THR) void bn_pool_small_bwd(const float* __restrict__ x, const float* __restrict__ act, const float* __restrict__ dpooled,
                                                            const float* __restrict__ stat, const float* __restrict__ gamma, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int H, int W, int C) {
  __shared__ float4 red[SM_THR / 64 * SM_CL];
  constexpr int NR = KH * KW;
  const int tid = threadIdx.x, cl = tid & (SM_CL - 1), wl = tid / SM_CL;
  const int c = blockIdx.x * SM_CH + cl * 4;
  const int OH = H / KH, OW = W / KW, nwin = B * OH * OW;
  float4 g[SM_NW][NR], xh[SM_NW][NR], dp[SM_NW];   // loaded as (act, x, dpooled); g / xhat after the first loop
  int row0[SM_NW];
#pragma unroll
  for (int k = 0; k < SM_NW; ++k) {
    const int w = wl + k * SM_WL;
    const int wc = w < nwin ? w : 0;
    const int ox = wc % OW, q = wc / OW, oy = q % OH, b = q / OH;
    row0[k] = (b * H + oy * KH) * W + ox * KW;
    dp[k] = *reinterpret_cast<const float4*>(dpooled + (long)wc * C + c);
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
      for (int d = 0; d < KW; ++d) {
        g[k][a * KW + d] = *reinterpret_cast<const float4*>(act + (long)(row0[k] + a * W + d) * C + c);
        xh[k][a * KW + d] = *reinterpret_cast<const float4*>(x + (long)(row0[k] + a * W + d) * C + c);
      }
  }
  const float4 mean = *reinterpret_cast<const float4*>(stat + c), inv = *reinterpret_cast<const float4*>(stat + C + c);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  auto lane = [](float4& v, int e) -> float& { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
#pragma unroll
  for (int k = 0; k < SM_NW; ++k) {
    const bool on = wl + k * SM_WL < nwin;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int am = 0;
      float m = -INFINITY;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const float av = lane(g[k][i], e);
        if (av > m) { m = av; am = i; }
      }
      const float mu = lane(const_cast<float4&>(mean), e), iv = lane(const_cast<float4&>(inv), e), dpe = lane(dp[k], e);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const float gv = (on && i == am && lane(g[k][i], e) > 0.f) ? dpe : 0.f;
        const float xv = (lane(xh[k][i], e) - mu) * iv;
        lane(g[k][i], e) = gv;
        lane(xh[k][i], e) = xv;
        lane(s1, e) += gv;
        lane(s2, e) += gv * xv;
      }
    }
  }
  const float4 t1 = sm_bsum4(s1, red, tid);
  const float4 t2 = sm_bsum4(s2, red, tid);
  if (wl == 0) {
    float4 dg = *reinterpret_cast<float4*>(dgamma + c), db = *reinterpret_cast<float4*>(dbeta + c);
    dg.x += t2.x; dg.y += t2.y; dg.z += t2.z; dg.w += t2.w;
    db.x += t1.x; db.y += t1.y; db.z += t1.z; db.w += t1.w;
    *reinterpret_cast<float4*>(dgamma + c) = dg;
    *reinterpret_cast<float4*>(dbeta + c) = db;
  }
  const float invR = 1.f / (float)(nwin * NR);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
  const float4 k0 = make_float4(ga.x * inv.x, ga.y * inv.y, ga.z * inv.z, ga.w * inv.w);
  const float4 m1 = make_float4(t1.x * invR, t1.y * invR, t1.z * invR, t1.w * invR), m2 = make_float4(t2.x * invR, t2.y * invR, t2.z * invR, t2.w * invR);
#pragma unroll
  for (int k = 0; k < SM_NW; ++k) {
    if (wl + k * SM_WL >= nwin) continue;
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
      for (int d = 0; d < KW; ++d) {
        const float4 gv = g[k][a * KW + d], xv = xh[k][a * KW + d];
        *reinterpret_cast<float4*>(dx + (long)(row0[k] + a * W + d) * C + c) =
            make_float4(k0.x * (gv.x - m1.x - xv.x * m2.x), k0.y * (gv.y - m1.y - xv.y * m2.y), k0.z * (gv.z - m1.z - xv.z * m2.z),
                        k0.w * (gv.w - m1.w - xv.w * m2.w));
      }
  }
}

}  // namespace

extern "C" {

size_t tatt_bn_workspace(int C) { return (size_t)BN_MAXBLK * 2 * C + 2 * C; }

// training: batch statistics of x [R,C] -> stat[2][C] (mean, invstd), running stats updated in place.
int tatt_bn_stats(const float* x, long ldx, int R, int C, float eps, float momentum, float* stat, float* run_mean, float* run_var,
                  float* workspace, hipStream_t st) {
  TATT_CHECK_ARG(R > 0 && C > 0 && (C <= 256 || C % 256 == 0), "tatt_bn_stats: bad dims R=%d C=%d", R, C);
  int nblk, rpb;
  bn_blocking(R, &nblk, &rpb);
  hipLaunchKernelGGL(bn_stats_partial, dim3(nblk), dim3(256), 0, st, x, ldx, R, C, rpb, workspace);
  TATT_LAUNCH_CHECK("bn_stats_partial");
  hipLaunchKernelGGL(bn_stats_final, dim3(cdiv(C, nblk > 128 ? 4 : 16)), dim3(256), 0, st, workspace, nblk, R, C, rpb, eps, momentum, stat, run_mean,
                     run_var, nblk > 128 ? 4 : 16);
  TATT_LAUNCH_CHECK("bn_stats_final");
  return TATT_OK;
}

int tatt_bn_stat_from_running(const float* rm, const float* rv, int C, float eps, float* stat, hipStream_t st) {
  hipLaunchKernelGGL(bn_stat_from_running, dim3(cdiv(C, 64)), dim3(64), 0, st, rm, rv, C, eps, stat);
  TATT_LAUNCH_CHECK("bn_stat_from_running");
  return TATT_OK;
}

// rows r of group r / Rg are normalised with stat[r / Rg] ([groups][2][C]); Rg = R: one set of statistics
int tatt_bn_apply_groups(const float* x, long ldx, const float* stat, const float* gamma, const float* beta, const float* res, long ldr,
                         float* y, long ldy, int R, int C, int act, int Rg, hipStream_t st) {
  TATT_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "tatt_bn_apply: C/ld must be multiples of 4");
  TATT_CHECK_ARG(act >= 0 && act <= 2 && R > 0, "tatt_bn_apply: act must be 0 (none), 1 (relu) or 2 (mish)");
  TATT_CHECK_ARG(Rg > 0 && (Rg == R || (R % Rg == 0 && Rg % bn_rows_per_block(C) == 0)),
                 "tatt_bn_apply_groups: rows per group (%d) must divide R and be a multiple of %d", Rg, bn_rows_per_block(C));
#define BN_APPLY(A) hipLaunchKernelGGL(bn_apply_kernel<A>, dim3(bn_apply_grid(R, C)), dim3(256), 0, st, x, ldx, stat, gamma, beta, res, ldr, y, ldy, R, C, Rg)
  if (act == 0) BN_APPLY(0);
  else if (act == 1) BN_APPLY(1);
  else BN_APPLY(2);
#undef BN_APPLY
  TATT_LAUNCH_CHECK("bn_apply");
  return TATT_OK;
}
int tatt_bn_apply(const float* x, long ldx, const float* stat, const float* gamma, const float* beta, const float* res, long ldr,
                  float* y, long ldy, int R, int C, int act, hipStream_t st) {
  return tatt_bn_apply_groups(x, ldx, stat, gamma, beta, res, ldr, y, ldy, R, C, act, R, st);
}

// tatt_bn_apply with y stored as bf16 (round to nearest even; ldy in elements, y 8-byte aligned)
int tatt_bn_apply_bf16_groups(const float* x, long ldx, const float* stat, const float* gamma, const float* beta, const float* res, long ldr,
                              uint16_t* y, long ldy, int R, int C, int act, int Rg, hipStream_t st) {
  TATT_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)y) & 7) == 0, "tatt_bn_apply_bf16: C/ld must be multiples of 4, y 8-byte aligned");
  TATT_CHECK_ARG(act >= 0 && act <= 2 && R > 0, "tatt_bn_apply_bf16: act must be 0 (none), 1 (relu) or 2 (mish)");
  TATT_CHECK_ARG(Rg > 0 && (Rg == R || (R % Rg == 0 && Rg % bn_rows_per_block(C) == 0)),
                 "tatt_bn_apply_bf16_groups: rows per group (%d) must divide R and be a multiple of %d", Rg, bn_rows_per_block(C));
  float* yf = reinterpret_cast<float*>(y);
#define BN_APPLY(A) hipLaunchKernelGGL((bn_apply_kernel<A, true>), dim3(bn_apply_grid(R, C)), dim3(256), 0, st, x, ldx, stat, gamma, beta, res, ldr, yf, ldy, R, C, Rg)
  if (act == 0) BN_APPLY(0);
  else if (act == 1) BN_APPLY(1);
  else BN_APPLY(2);
#undef BN_APPLY
  TATT_LAUNCH_CHECK("bn_apply<bf16 out>");
  return TATT_OK;
}
int tatt_bn_apply_bf16(const float* x, long ldx, const float* stat, const float* gamma, const float* beta, const float* res, long ldr,
                       uint16_t* y, long ldy, int R, int C, int act, hipStream_t st) {
  return tatt_bn_apply_bf16_groups(x, ldx, stat, gamma, beta, res, ldr, y, ldy, R, C, act, R, st);
}

// Backward through y = act(bn(x)).  workspace: tatt_bn_workspace(C) floats.
int tatt_bn_bwd(const float* x, long ldx, const float* dy, long lddy, const float* stat, const float* gamma, const float* beta,
                float* dx, long lddx, float* dgamma, float* dbeta, int R, int C, int act, int train, int accumulate_params,
                int accumulate_dx, float* workspace, hipStream_t st) {
  TATT_CHECK_ARG(C % 4 == 0 && (C <= 256 || C % 256 == 0), "tatt_bn_bwd: bad C");
  int nblk, rpb;
  bn_blocking(R, &nblk, &rpb);
  float* sums = workspace + (size_t)BN_MAXBLK * 2 * C;
  hipLaunchKernelGGL(bn_bwd_reduce, dim3(nblk), dim3(256), 0, st, x, ldx, dy, lddy, stat, gamma, beta, R, C, act, rpb, workspace);
  TATT_LAUNCH_CHECK("bn_bwd_reduce");
  hipLaunchKernelGGL(bn_bwd_final, dim3(cdiv(C, nblk > 128 ? 4 : 16)), dim3(256), 0, st, workspace, nblk, C, sums, dgamma, dbeta, accumulate_params, nblk > 128 ? 4 : 16);
  TATT_LAUNCH_CHECK("bn_bwd_final");
#define BN_BWD_APPLY(A) hipLaunchKernelGGL(bn_bwd_apply<A>, dim3(bn_apply_grid(R, C)), dim3(256), 0, st, x, ldx, dy, lddy, stat, gamma, beta, sums, dx, lddx, R, C, train, accumulate_dx)
  if (act == 1) BN_BWD_APPLY(1);
  else if (act == 2) BN_BWD_APPLY(2);
  else BN_BWD_APPLY(0);
#undef BN_BWD_APPLY
  TATT_LAUNCH_CHECK("bn_bwd_apply");
  return TATT_OK;
}

// Second halves of the BatchNorm passes whose first half ran in a GEMM / convolution epilogue (tatt_conv_fwd_bn, tatt_gemm_cat_bnbwd):
// partial [nblk][2][C] holds one (mean, M2) resp. (sum g, sum g xhat) pair per block of rows_per_blk rows.
int tatt_bn_stats_from_partial(const float* partial, int nblk, int rows_per_blk, int R, int C, float eps, float momentum, float* stat,
                               float* run_mean, float* run_var, hipStream_t st) {
  TATT_CHECK_ARG(partial && stat && nblk > 0 && rows_per_blk > 0 && (long)nblk * rows_per_blk >= R && (long)(nblk - 1) * rows_per_blk < R && C > 0,
                 "tatt_bn_stats_from_partial: bad arguments");
  hipLaunchKernelGGL(bn_stats_final, dim3(cdiv(C, nblk > 128 ? 4 : 16)), dim3(256), 0, st, partial, nblk, R, C, rows_per_blk, eps, momentum, stat, run_mean,
                     run_var, nblk > 128 ? 4 : 16);
  TATT_LAUNCH_CHECK("bn_stats_final");
  return TATT_OK;
}
// ng groups of R rows each in one launch: partial [ng][nblk][2][C], stat [ng][2][C]; running statistics updated group after group
int tatt_bn_stats_from_partial_groups(const float* partial, int nblk, int rows_per_blk, int R, int C, int ng, float eps, float momentum,
                                      float* stat, float* run_mean, float* run_var, hipStream_t st) {
  TATT_CHECK_ARG(partial && stat && nblk > 0 && rows_per_blk > 0 && (long)nblk * rows_per_blk >= R && (long)(nblk - 1) * rows_per_blk < R && C > 0 &&
                     ng >= 1, "tatt_bn_stats_from_partial_groups: bad arguments");
  hipLaunchKernelGGL(bn_stats_final_groups, dim3(cdiv(C, nblk > 128 ? 4 : 16)), dim3(256), 0, st, partial, nblk, ng, R, C, rows_per_blk, eps, momentum,
                     stat, run_mean, run_var, nblk > 128 ? 4 : 16);
  TATT_LAUNCH_CHECK("bn_stats_final_groups");
  return TATT_OK;
}

// dx (+)= BatchNorm backward given g = dy * act'(bn(x)) (already multiplied by the activation's derivative) and its partial sums;
// sums: 2*C floats of scratch.
int tatt_bn_bwd_from_partial(const float* x, long ldx, const float* g, long ldg, const float* stat, const float* gamma, const float* beta,
                             const float* partial, int nblk, float* dx, long lddx, float* dgamma, float* dbeta, int R, int C, int train,
                             int accumulate_params, int accumulate_dx, float* sums, hipStream_t st) {
  TATT_CHECK_ARG(x && g && stat && gamma && beta && partial && dx && sums && nblk > 0 && C % 4 == 0 && (C <= 256 || C % 256 == 0),
                 "tatt_bn_bwd_from_partial: bad arguments");
  hipLaunchKernelGGL(bn_bwd_final, dim3(cdiv(C, nblk > 128 ? 4 : 16)), dim3(256), 0, st, partial, nblk, C, sums, dgamma, dbeta, accumulate_params, nblk > 128 ? 4 : 16);
  TATT_LAUNCH_CHECK("bn_bwd_final");
  hipLaunchKernelGGL(bn_bwd_apply<0>, dim3(bn_apply_grid(R, C)), dim3(256), 0, st, x, ldx, g, ldg, stat, gamma, beta, sums, dx, lddx, R, C,
                     train, accumulate_dx);
  TATT_LAUNCH_CHECK("bn_bwd_apply");
  return TATT_OK;
}

// tatt_bn_bwd_from_partial with dx stored as bf16 (no accumulation onto dx)
int tatt_bn_bwd_from_partial_bf16(const float* x, long ldx, const float* g, long ldg, const float* stat, const float* gamma, const float* beta,
                                  const float* partial, int nblk, uint16_t* dx, long lddx, float* dgamma, float* dbeta, int R, int C, int train,
                                  int accumulate_params, float* sums, hipStream_t st) {
  TATT_CHECK_ARG(x && g && stat && gamma && beta && partial && dx && sums && nblk > 0 && C % 4 == 0 && (C <= 256 || C % 256 == 0) && lddx % 4 == 0 &&
                     (((uintptr_t)dx) & 7) == 0, "tatt_bn_bwd_from_partial_bf16: bad arguments");
  hipLaunchKernelGGL(bn_bwd_final, dim3(cdiv(C, nblk > 128 ? 4 : 16)), dim3(256), 0, st, partial, nblk, C, sums, dgamma, dbeta, accumulate_params, nblk > 128 ? 4 : 16);
  TATT_LAUNCH_CHECK("bn_bwd_final");
  hipLaunchKernelGGL((bn_bwd_apply<0, true>), dim3(bn_apply_grid(R, C)), dim3(256), 0, st, x, ldx, g, ldg, stat, gamma, beta, sums, reinterpret_cast<float*>(dx),
                     lddx, R, C, train, 0);
  TATT_LAUNCH_CHECK("bn_bwd_apply<bf16 out>");
  return TATT_OK;
}

int tatt_ln_fwd(const float* x, const float* res, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev,
                const float* gamma,
                const float* beta, float* vout, float* y, int rows, float eps, hipStream_t st) {
  TATT_CHECK_ARG(rows > 0 && ((((uintptr_t)x) | ((uintptr_t)res) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)vout) | ((uintptr_t)y)) & 15) == 0,
                 "tatt_ln_fwd: rows > 0 and 16-byte aligned pointers required");
  DropCfg dc{seed_site, thresh, scale, seed_dev};
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(rows, 16)), dim3(256), 0, st, x, res, dc, gamma, beta, vout, y, rows, eps);
  TATT_LAUNCH_CHECK("ln_fwd");
  return TATT_OK;
}

size_t tatt_ln_workspace(int rows) { (void)rows; return (size_t)512 * 128; }

int tatt_ln_bwd(const float* v, const float* dy, float dyscale, const float* gamma, float* dv, int dv_accumulate, float* dres,
                uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, float* dgamma, float* dbeta,
                int accumulate_params, int rows,
                float eps, float* workspace, hipStream_t st) {
  TATT_CHECK_ARG(rows > 0 && ((((uintptr_t)v) | ((uintptr_t)dy) | ((uintptr_t)gamma) | ((uintptr_t)dv) | ((uintptr_t)dres)) & 15) == 0,
                 "tatt_ln_bwd: rows > 0 and 16-byte aligned pointers required");
  DropCfg dc{seed_site, thresh, scale, seed_dev};
  int nblk = cdiv(rows, 64);
  if (nblk > 512) nblk = 512;
  int rpb = cdiv(rows, nblk);
  nblk = cdiv(rows, rpb);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nblk), dim3(256), 0, st, v, dy, dyscale, gamma, dv, dv_accumulate, dres, dc, workspace, rows,
                     rpb, eps);
  TATT_LAUNCH_CHECK("ln_bwd");
  hipLaunchKernelGGL(ln_bwd_final, dim3(4), dim3(256), 0, st, workspace, nblk, dgamma, dbeta, accumulate_params);
  TATT_LAUNCH_CHECK("ln_bwd_final");
  return TATT_OK;
}

// One-launch BatchNorm (batch statistics, running-statistic update) + ReLU + max-pool (kh x kw, stride = window; 1 x 1 = none) of a
// small NHWC map x [B,H,W,C]: stat [2][C] = (mean, invstd), act_out [B,H,W,C] = relu(bn(x)), pooled [B,H/kh,W/kw,C] (unused for 1 x 1).
size_t tatt_bn_pool_small_supported(int B, int H, int W, int C, int kh, int kw) {
  const bool pool_ok = (kh == 1 && kw == 1) || (kh == 2 && kw == 2) || (kh == 1 && kw == 2);
  return pool_ok && B > 0 && H > 0 && W > 0 && C > 0 && C % SM_CH == 0 && H % kh == 0 && W % kw == 0 &&
         (long)B * (H / kh) * (W / kw) <= (long)SM_NW * SM_WL;
}

int tatt_bn_pool_small_fwd(const float* x, float* stat, const float* gamma, const float* beta, float* run_mean, float* run_var, float* act_out,
                           float* pooled, int B, int H, int W, int C, int kh, int kw, float eps, float momentum, hipStream_t st) {
  TATT_CHECK_ARG(tatt_bn_pool_small_supported(B, H, W, C, kh, kw), "tatt_bn_pool_small_fwd: need C %% 4 == 0, a 1x1 / 2x2 / 1x2 window that divides "
                 "the map and at most %d windows (got B=%d H=%d W=%d C=%d %dx%d)", SM_NW * SM_WL, B, H, W, C, kh, kw);
  TATT_CHECK_ARG(x && stat && gamma && beta && act_out && (pooled || kh * kw == 1) && ((run_mean == nullptr) == (run_var == nullptr)),
                 "tatt_bn_pool_small_fwd: null pointer");
#define SM_FWD(a, b) hipLaunchKernelGGL((bn_pool_small_fwd<a, b>), dim3(C / SM_CH), dim3(SM_THR), 0, st, x, stat, gamma, beta, run_mean, run_var, act_out, pooled, B, H, W, C, eps, momentum)
  if (kh == 2) SM_FWD(2, 2);
  else if (kw == 2) SM_FWD(1, 2);
  else SM_FWD(1, 1);
#undef SM_FWD
  TATT_LAUNCH_CHECK("bn_pool_small_fwd");
  return TATT_OK;
}

// dpooled: gradient of the pooled map (of act_out itself for a 1 x 1 window); dx = gradient of x; dgamma / dbeta are accumulated.
int tatt_bn_pool_small_bwd(const float* x, const float* act, const float* dpooled, const float* stat, const float* gamma, float* dx,
                           float* dgamma, float* dbeta, int B, int H, int W, int C, int kh, int kw, hipStream_t st) {
  TATT_CHECK_ARG(tatt_bn_pool_small_supported(B, H, W, C, kh, kw), "tatt_bn_pool_small_bwd: unsupported shape (B=%d H=%d W=%d C=%d %dx%d)", B, H, W, C,
                 kh, kw);
  TATT_CHECK_ARG(x && act && dpooled && stat && gamma && dx && dgamma && dbeta, "tatt_bn_pool_small_bwd: null pointer");
#define SM_BWD(a, b) hipLaunchKernelGGL((bn_pool_small_bwd<a, b>), dim3(C / SM_CH), dim3(SM_THR), 0, st, x, act, dpooled, stat, gamma, dx, dgamma, dbeta, B, H, W, C)
  if (kh == 2) SM_BWD(2, 2);
  else if (kw == 2) SM_BWD(1, 2);
  else SM_BWD(1, 1);
#undef SM_BWD
  TATT_LAUNCH_CHECK("bn_pool_small_bwd");
  return TATT_OK;
}

}  // extern "C"
