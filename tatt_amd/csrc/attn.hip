// Multi-head attention core for the TP interpreter: E = 64, 4 heads x 16, S = 26 keys (the text-prior length),
// Lq queries per batch element (1024 image tokens for the decoder, 26 for the encoder self-attention).
// The score matrix never leaves the CU: K/V rows arrive as wave-uniform scalar loads, one lane owns one (query, head)
// row, softmax and the dropout mask live in registers.  Q, K, V are the already-projected [rows,64] matrices
// (head h = columns 16h..16h+15); token order is (batch, position).
#include "common.h"

namespace {

constexpr int S = 26, DH = 16, E = 64, NH = 4;

// grid (ceil(Lq/64), B), block 256: wave = head, lane = query
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ Kp,
                                                       const float* __restrict__ Vp, float* __restrict__ ctx,
                                                       float* __restrict__ pw, int Lq, DropCfg dc) {
  // K / V rows are wave-uniform: scalar loads, SGPR operands of the FMAs (3.3 KB per head: the scalar cache holds them).  Measured
  // against LDS staging with 32-bit broadcasts (same time) and with 128-bit broadcasts (the scheduler hoists every row's read,
  // 474 VGPRs, one wave per SIMD: 35 us instead of 31.6).  The compiler spills part of the SGPR rows through v_writelane; the
  // kernel is bound by instruction issue (about 3,400 per wave), not by memory.
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* __restrict__ Ks = Kp + (long)b * S * E + h * DH;
  const float* __restrict__ Vs = Vp + (long)b * S * E + h * DH;
  const int q = blockIdx.x * 64 + lane;
  if (q >= Lq) return;
  const long row = (long)b * Lq + q;
  float qv[DH];
#pragma unroll
  for (int d4 = 0; d4 < DH / 4; ++d4) {
    float4 v = *reinterpret_cast<const float4*>(Q + row * E + h * DH + d4 * 4);
    qv[d4 * 4] = v.x * 0.25f; qv[d4 * 4 + 1] = v.y * 0.25f; qv[d4 * 4 + 2] = v.z * 0.25f; qv[d4 * 4 + 3] = v.w * 0.25f;
  }
  float p[S];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < S; ++k) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) s += qv[d] * Ks[k * E + d];
    p[k] = s;
    mx = fmaxf(mx, s);
  }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < S; ++k) {
    p[k] = expf(p[k] - mx);
    sum += p[k];
  }
  const float inv = 1.f / sum;
  const uint32_t ibase = (uint32_t)((((long)b * NH + h) * Lq + q) * S);
  const uint32_t ss = tatt_eff_seed(dc);
  float o[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] = 0.f;
#pragma unroll
  for (int k = 0; k < S; ++k) {
    float pk = p[k] * inv * tatt_drop_mask(ss, ibase + k, dc.thresh, dc.scale);
    if (pw) pw[(long)ibase + k] = pk;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] += pk * Vs[k * E + d];
  }
#pragma unroll
  for (int d4 = 0; d4 < DH / 4; ++d4)
    *reinterpret_cast<float4*>(ctx + row * E + h * DH + d4 * 4) = make_float4(o[d4 * 4], o[d4 * 4 + 1], o[d4 * 4 + 2], o[d4 * 4 + 3]);
}

// Backward.  grid (ceil(Lq/256), B*4), block 256: one thread per query of one (batch, head); K/V rows of that head by scalar loads
// (wave-uniform).  Every thread differentiates its own softmax row and writes dQ; dV = P^T dC and dK = dS^T Q are sums
// over the queries, reduced in two phases through ONE pair of LDS tiles ([256][27] and [256][17], 45 KB: three workgroups per CU)
// -- phase 1 holds (P, dC), phase 2 (dS, Q), both still live in the owning thread's registers -- into a partial per query chunk;
// attn_bwd_final sums the chunks in a fixed order (deterministic, no atomics).  The previous kernel walked all queries of a
// (batch, head) in ONE workgroup, 128 at a time with half of its threads idle: 256 workgroups, 105 us per decoder layer.
constexpr int QT = 256, LP = S + 1, LD = DH + 1;

__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ Kp,
                                                       const float* __restrict__ Vp, const float* __restrict__ dctx,
                                                       float* __restrict__ dQ, float* __restrict__ part, int Lq, int B, DropCfg dc) {
  __shared__ float Lt[QT * LP], Rt[QT * LD];
  const int bh = blockIdx.y, b = bh / NH, h = bh % NH, tid = threadIdx.x, chunk = blockIdx.x;
  const float* __restrict__ Ks = Kp + (long)b * S * E + h * DH;  // wave-uniform rows: scalar loads (see attn_fwd_kernel)
  const float* __restrict__ Vs = Vp + (long)b * S * E + h * DH;
  const int q = chunk * QT + tid;
  const bool on = q < Lq;
  const long row = (long)b * Lq + (on ? q : 0);
  float pd[S], ds[S], qv[DH], dcv[DH];
#pragma unroll
  for (int d4 = 0; d4 < DH / 4; ++d4) {
    float4 v = *reinterpret_cast<const float4*>(Q + row * E + h * DH + d4 * 4);
    qv[d4 * 4] = v.x * 0.25f; qv[d4 * 4 + 1] = v.y * 0.25f; qv[d4 * 4 + 2] = v.z * 0.25f; qv[d4 * 4 + 3] = v.w * 0.25f;
    float4 g = *reinterpret_cast<const float4*>(dctx + row * E + h * DH + d4 * 4);
    dcv[d4 * 4] = g.x; dcv[d4 * 4 + 1] = g.y; dcv[d4 * 4 + 2] = g.z; dcv[d4 * 4 + 3] = g.w;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < S; ++k) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) s += qv[d] * Ks[k * E + d];
    pd[k] = s;
    mx = fmaxf(mx, s);
  }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < S; ++k) {
    pd[k] = expf(pd[k] - mx);
    sum += pd[k];
  }
  const float inv = 1.f / sum;
  const uint32_t ibase = (uint32_t)((((long)b * NH + h) * Lq + q) * S);
  const uint32_t ss = tatt_eff_seed(dc);
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < S; ++k) {
    const float pk = pd[k] * inv;
    const float mk = tatt_drop_mask(ss, ibase + k, dc.thresh, dc.scale);
    float g = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) g += dcv[d] * Vs[k * E + d];
    g *= mk;  // dL/dP (pre-dropout softmax)
    dot += pk * g;
    ds[k] = g;
    pd[k] = pk;
    Lt[tid * LP + k] = on ? pk * mk : 0.f;  // phase 1 left tile: the dropped softmax
  }
  float dq[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    dq[d] = 0.f;
    Rt[tid * LD + d] = on ? dcv[d] : 0.f;  // phase 1 right tile: dC
  }
#pragma unroll
  for (int k = 0; k < S; ++k) {
    ds[k] = pd[k] * (ds[k] - dot);  // dL/dS
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] += ds[k] * Ks[k * E + d];
  }
  if (on) {
#pragma unroll
    for (int d4 = 0; d4 < DH / 4; ++d4)
      *reinterpret_cast<float4*>(dQ + row * E + h * DH + d4 * 4) =
          make_float4(dq[d4 * 4] * 0.25f, dq[d4 * 4 + 1] * 0.25f, dq[d4 * 4 + 2] * 0.25f, dq[d4 * 4 + 3] * 0.25f);
  }
  // reductions over the chunk's queries: thread t < 208 owns outputs (k = t / 8, d = 2 (t % 8), + 1)
  const int rk = tid >> 3, rd = (tid & 7) * 2;
  float* pout = part + (((long)chunk * 2) * B * S + (long)b * S) * E + h * DH;  // [chunk][which][b][k][64]
#pragma unroll
  for (int phase = 0; phase < 2; ++phase) {
    __syncthreads();
    if (tid < 8 * S) {
      float a0 = 0.f, a1 = 0.f, c0 = 0.f, c1 = 0.f;
#pragma unroll 8
      for (int qq = 0; qq < QT; qq += 2) {
        const float l0 = Lt[qq * LP + rk], l1 = Lt[(qq + 1) * LP + rk];
        a0 += l0 * Rt[qq * LD + rd];
        a1 += l0 * Rt[qq * LD + rd + 1];
        c0 += l1 * Rt[(qq + 1) * LD + rd];
        c1 += l1 * Rt[(qq + 1) * LD + rd + 1];
      }
      float* o = pout + (long)phase * B * S * E + (long)rk * E + rd;  // phase 0 -> dV partial, phase 1 -> dK partial
      o[0] = a0 + c0;
      o[1] = a1 + c1;
    }
    if (phase == 0) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < S; ++k) Lt[tid * LP + k] = on ? ds[k] : 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) Rt[tid * LD + d] = on ? qv[d] : 0.f;
    }
  }
}

// dV, dK (+)= sum over query chunks of the partials [chunk][2][B*S][64]
__global__ void attn_bwd_final(const float* __restrict__ part, int nchunk, long n, float* __restrict__ dK, float* __restrict__ dV,
                               int accumulate_kv) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n) return;
  const int which = i >= n;
  const long e = i - which * n;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += part[((long)c * 2 + which) * n + e];
  float* dst = (which ? dK : dV) + e;
  *dst = accumulate_kv ? *dst + s : s;
}

// head average of per-head weights pw [B,4,Lq,S] -> [B,Lq,S]
__global__ void attn_avg_kernel(const float* __restrict__ pw, float* __restrict__ out, int B, int Lq) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long per = (long)Lq * S;
  if (i >= (long)B * per) return;
  long b = i / per, r = i - b * per;
  const float* p = pw + b * NH * per + r;
  out[i] = 0.25f * (p[0] + p[per] + p[2 * per] + p[3 * per]);
}

}  // namespace

extern "C" {

int tatt_attn_fwd(const float* Q, const float* Kp, const float* Vp, float* ctx, float* pw, int B, int Lq, int nkeys,
                  uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, hipStream_t st) {
  TATT_CHECK_ARG(nkeys == S, "tatt_attn_fwd: built for %d keys, got %d", S, nkeys);
  TATT_CHECK_ARG(B > 0 && Lq > 0, "tatt_attn_fwd: bad dims");
  DropCfg dc{seed_site, thresh, scale, seed_dev};
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(Lq, 64), B), dim3(256), 0, st, Q, Kp, Vp, ctx, pw, Lq, dc);
  TATT_LAUNCH_CHECK("attn_fwd");
  return TATT_OK;
}

size_t tatt_attn_bwd_workspace(int B, int Lq) { return (size_t)cdiv(Lq, QT) * 2 * B * S * E; }

int tatt_attn_bwd(const float* Q, const float* Kp, const float* Vp, const float* dctx, float* dQ, float* dK, float* dV, int B,
                  int Lq, int nkeys, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, int accumulate_kv,
                  float* workspace, hipStream_t st) {
  TATT_CHECK_ARG(nkeys == S, "tatt_attn_bwd: built for %d keys, got %d", S, nkeys);
  TATT_CHECK_ARG(Q && Kp && Vp && dctx && dQ && dK && dV && workspace && B > 0 && Lq > 0, "tatt_attn_bwd: bad arguments (workspace: tatt_attn_bwd_workspace floats)");
  TATT_CHECK_ARG(((((uintptr_t)Q) | ((uintptr_t)dctx) | ((uintptr_t)dQ)) & 15) == 0, "tatt_attn_bwd: Q, dctx, dQ must be 16-byte aligned");
  DropCfg dc{seed_site, thresh, scale, seed_dev};
  const int nchunk = cdiv(Lq, QT);
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(nchunk, B * NH), dim3(256), 0, st, Q, Kp, Vp, dctx, dQ, workspace, Lq, B, dc);
  TATT_LAUNCH_CHECK("attn_bwd");
  const long n = (long)B * S * E;
  hipLaunchKernelGGL(attn_bwd_final, dim3(cdiv(2 * n, 256)), dim3(256), 0, st, workspace, nchunk, n, dK, dV, accumulate_kv);
  TATT_LAUNCH_CHECK("attn_bwd_final");
  return TATT_OK;
}

int tatt_attn_kv_final(const float* part, int nchunk, int B, float* dK, float* dV, int accumulate_kv, hipStream_t st) {
  TATT_CHECK_ARG(part && dK && dV && nchunk > 0 && B > 0, "tatt_attn_kv_final: bad arguments");
  const long n = (long)B * S * E;
  hipLaunchKernelGGL(attn_bwd_final, dim3(cdiv(2 * n, 256)), dim3(256), 0, st, part, nchunk, n, dK, dV, accumulate_kv);
  TATT_LAUNCH_CHECK("attn_kv_final");
  return TATT_OK;
}

int tatt_attn_avg(const float* pw, float* out, int B, int Lq, hipStream_t st) {
  long tot = (long)B * Lq * S;
  hipLaunchKernelGGL(attn_avg_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, pw, out, B, Lq);
  TATT_LAUNCH_CHECK("attn_avg");
  return TATT_OK;
}

}  // extern "C"
