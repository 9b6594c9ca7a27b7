// Multi-head attention core for the TP interpreter: E = 64, 4 heads x 16, S = 26 keys (the text-prior length),
// Lq queries per batch element (1024 image tokens for the decoder, 26 for the encoder self-attention).
// The score matrix never leaves the CU: K/V of one batch element sit in LDS, one lane owns one (query, head)
// row, softmax and the dropout mask live in registers.  Q, K, V are the already-projected [rows,64] matrices
// (head h = columns 16h..16h+15); token order is (batch, position).
#include "common.h"

namespace {

constexpr int S = 26, DH = 16, E = 64, NH = 4;

// grid (ceil(Lq/64), B), block 256: wave = head, lane = query
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ Kp,
                                                       const float* __restrict__ Vp, float* __restrict__ ctx,
                                                       float* __restrict__ pw, int Lq, DropCfg dc) {
  __shared__ __attribute__((aligned(16))) float Ks[S * E], Vs[S * E];
  const int b = blockIdx.y, tid = threadIdx.x, h = tid >> 6, lane = tid & 63;
  for (int i = tid; i < S * E / 4; i += 256) {
    reinterpret_cast<float4*>(Ks)[i] = reinterpret_cast<const float4*>(Kp + (long)b * S * E)[i];
    reinterpret_cast<float4*>(Vs)[i] = reinterpret_cast<const float4*>(Vp + (long)b * S * E)[i];
  }
  __syncthreads();
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
    for (int d = 0; d < DH; ++d) s += qv[d] * Ks[k * E + h * DH + d];
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
    for (int d = 0; d < DH; ++d) o[d] += pk * Vs[k * E + h * DH + d];
  }
#pragma unroll
  for (int d4 = 0; d4 < DH / 4; ++d4)
    *reinterpret_cast<float4*>(ctx + row * E + h * DH + d4 * 4) = make_float4(o[d4 * 4], o[d4 * 4 + 1], o[d4 * 4 + 2], o[d4 * 4 + 3]);
}

// grid (B*4): one block per (batch, head); 128-query tiles; dK/dV reduced inside the block (deterministic).
constexpr int QT = 128, LP = S + 1, LD = DH + 1;

__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ Kp,
                                                       const float* __restrict__ Vp, const float* __restrict__ dctx,
                                                       float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV,
                                                       int Lq, DropCfg dc, int accumulate_kv) {
  __shared__ float Ks[S * DH], Vs[S * DH];
  __shared__ float Pd_t[QT * LP], dS_t[QT * LP], dc_t[QT * LD], q_t[QT * LD];
  const int b = blockIdx.x / NH, h = blockIdx.x % NH, tid = threadIdx.x;
  for (int i = tid; i < S * DH; i += 256) {
    int k = i / DH, d = i % DH;
    Ks[i] = Kp[((long)b * S + k) * E + h * DH + d];
    Vs[i] = Vp[((long)b * S + k) * E + h * DH + d];
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int t0 = 0; t0 < Lq; t0 += QT) {
    if (tid < QT) {
      const int q = t0 + tid;
      float pd[S], ds[S], qv[DH], dcv[DH];
      if (q < Lq) {
        const long row = (long)b * Lq + q;
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
          for (int d = 0; d < DH; ++d) s += qv[d] * Ks[k * DH + d];
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
        float dot = 0.f;
        const uint32_t ss = tatt_eff_seed(dc);
#pragma unroll
        for (int k = 0; k < S; ++k) {
          float pk = pd[k] * inv;
          float mk = tatt_drop_mask(ss, ibase + k, dc.thresh, dc.scale);
          float g = 0.f;
#pragma unroll
          for (int d = 0; d < DH; ++d) g += dcv[d] * Vs[k * DH + d];
          g *= mk;            // dL/dP (pre-dropout softmax)
          dot += pk * g;
          ds[k] = g;          // finished below
          pd[k] = pk;         // plain softmax for now
          Pd_t[tid * LP + k] = pk * mk;
        }
        float dq[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) dq[d] = 0.f;
#pragma unroll
        for (int k = 0; k < S; ++k) {
          float v = pd[k] * (ds[k] - dot);
          dS_t[tid * LP + k] = v;
#pragma unroll
          for (int d = 0; d < DH; ++d) dq[d] += v * Ks[k * DH + d];
        }
#pragma unroll
        for (int d4 = 0; d4 < DH / 4; ++d4)
          *reinterpret_cast<float4*>(dQ + row * E + h * DH + d4 * 4) =
              make_float4(dq[d4 * 4] * 0.25f, dq[d4 * 4 + 1] * 0.25f, dq[d4 * 4 + 2] * 0.25f, dq[d4 * 4 + 3] * 0.25f);
#pragma unroll
        for (int d = 0; d < DH; ++d) {
          dc_t[tid * LD + d] = dcv[d];
          q_t[tid * LD + d] = qv[d];
        }
      } else {
#pragma unroll
        for (int k = 0; k < S; ++k) {
          Pd_t[tid * LP + k] = 0.f;
          dS_t[tid * LP + k] = 0.f;
        }
#pragma unroll
        for (int d = 0; d < DH; ++d) {
          dc_t[tid * LD + d] = 0.f;
          q_t[tid * LD + d] = 0.f;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int o = tid + 256 * i;
      if (o < 2 * S * DH) {
        int which = o / (S * DH), rem = o % (S * DH), k = rem / DH, d = rem % DH;
        const float* L = which ? dS_t : Pd_t;
        const float* R = which ? q_t : dc_t;
        float a = 0.f;
        for (int qq = 0; qq < QT; ++qq) a += L[qq * LP + k] * R[qq * LD + d];
        acc[i] += a;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int o = tid + 256 * i;
    if (o < 2 * S * DH) {
      int which = o / (S * DH), rem = o % (S * DH), k = rem / DH, d = rem % DH;
      float* dst = (which ? dK : dV) + ((long)b * S + k) * E + h * DH + d;
      *dst = accumulate_kv ? *dst + acc[i] : acc[i];
    }
  }
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

int tatt_attn_bwd(const float* Q, const float* Kp, const float* Vp, const float* dctx, float* dQ, float* dK, float* dV, int B,
                  int Lq, int nkeys, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, int accumulate_kv,
                  hipStream_t st) {
  TATT_CHECK_ARG(nkeys == S, "tatt_attn_bwd: built for %d keys, got %d", S, nkeys);
  DropCfg dc{seed_site, thresh, scale, seed_dev};
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * NH), dim3(256), 0, st, Q, Kp, Vp, dctx, dQ, dK, dV, Lq, dc, accumulate_kv);
  TATT_LAUNCH_CHECK("attn_bwd");
  return TATT_OK;
}

int tatt_attn_avg(const float* pw, float* out, int B, int Lq, hipStream_t st) {
  long tot = (long)B * Lq * S;
  hipLaunchKernelGGL(attn_avg_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, pw, out, B, Lq);
  TATT_LAUNCH_CHECK("attn_avg");
  return TATT_OK;
}

}  // extern "C"
