// GRU recurrences.
//  (1) small BiGRU of the SRBs (hidden 32): one half-wave (32 lanes) per (sequence, direction); lane j owns
//      hidden unit j, keeps its three W_hh rows (96 floats) in VGPRs, h is broadcast through LDS.
//      The time-parallel input projection gi = W_ih x + b_ih is a GEMM done by the caller.
//  (2) RPE GRU of the TP interpreter (hidden 512, 64 rows, recurrence over the batch axis): per step the
//      h*W_hh^T product is a GEMM launched by the caller; the element-wise gate kernels live here.
// Gate order (r, z, n):  r = s(gi_r+gh_r)  z = s(gi_z+gh_z)  n = tanh(gi_n + r*gh_n)  h' = (1-z) n + z h.
#include "common.h"

namespace {

constexpr int HID = 32;
typedef float f32x2 __attribute__((ext_vector_type(2)));  // pairs of k: the dot products run as v_pk_fma_f32 (two FMAs per lane and issue)

struct SeqMap {  // pixel index of (sequence s, step t) = (s / inner) * outer_stride + (s % inner) * inner_stride + t * tstride
  int inner;
  long outer_stride, inner_stride, tstride;
  int T;
};

__device__ __forceinline__ long seq_base(const SeqMap& m, int s) {
  int so = s / m.inner, si = s - so * m.inner;
  return (long)so * m.outer_stride + (long)si * m.inner_stride;
}

// gi [P,192] (dir*96 + gate*32 + j); whh/bhh: per-direction pointers; out [P,64]; save [P,2,4,32] or null
// RES: also out2 [P,64] = res [P,64] + out (the "x + residual" that feeds the second GruBlock of an SRB, model/tsrn.py:901): the
// residual rides the same three-set prefetch as the input projections, so the 16.8 MB add never is a launch of its own
template <bool RES>
__global__ __launch_bounds__(256) void gru32_fwd(const float* __restrict__ gi, const float* __restrict__ whh_f,
                                                 const float* __restrict__ whh_b, const float* __restrict__ bhh_f,
                                                 const float* __restrict__ bhh_b, float* __restrict__ out,
                                                 float* __restrict__ save, int nseq, SeqMap sm, const float* __restrict__ res,
                                                 float* __restrict__ out2) {
  __shared__ __attribute__((aligned(16))) float hbuf[8][HID];
  const int tid = threadIdx.x, j = tid & 31, slot = tid >> 5;
  int wi = blockIdx.x * 8 + slot;
  const bool active = wi < nseq * 2;
  if (!active) wi = nseq * 2 - 1;
  const int s = wi >> 1, dir = wi & 1;
  const float* whh = dir ? whh_b : whh_f;
  const float* bhh = dir ? bhh_b : bhh_f;
  f32x2 wr[HID / 2], wz[HID / 2], wn[HID / 2];
#pragma unroll
  for (int k = 0; k < HID / 2; ++k) {
    wr[k] = *reinterpret_cast<const f32x2*>(&whh[(0 * HID + j) * HID + 2 * k]);
    wz[k] = *reinterpret_cast<const f32x2*>(&whh[(1 * HID + j) * HID + 2 * k]);
    wn[k] = *reinterpret_cast<const f32x2*>(&whh[(2 * HID + j) * HID + 2 * k]);
  }
  const float br = bhh[j], bz = bhh[HID + j], bn = bhh[2 * HID + j];
  const long base = seq_base(sm, s);
  float h = 0.f;
  hbuf[slot][j] = 0.f;
  const int T = sm.T;
  long pix = base + (long)(dir ? T - 1 : 0) * sm.tstride;
  const long pstep = dir ? -sm.tstride : sm.tstride;
  // The horizontal pass has one wave per SIMD, so nothing hides a global load but the chain itself: the input projections
  // are fetched in sets of PF steps, two sets ahead of their use (time index clamped instead of predicated: no branches around
  // the loads; see gru32_bwd for why one set ahead is not enough).
  constexpr int PF = 4;
  const float* gcol = gi + dir * 96 + j;
  const float* ocol = RES ? res + dir * HID + j : nullptr;
  float rA[PF], zA[PF], nA[PF], rB[PF], zB[PF], nB[PF], rC[PF], zC[PF], nC[PF];
  float oA[PF], oB[PF], oC[PF];
  auto fetch = [&](int t0, float (&fr)[PF], float (&fz)[PF], float (&fn)[PF], float (&fo)[PF]) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int tc = min(t0 + u, T - 1);
      const long px = base + (long)(dir ? T - 1 - tc : tc) * sm.tstride;
      const float* q = gcol + px * 192;
      fr[u] = q[0];
      fz[u] = q[HID];
      fn[u] = q[2 * HID];
      if constexpr (RES) fo[u] = ocol[px * 64];
    }
  };
  auto steps = [&](int t0, const float (&c_r)[PF], const float (&c_z)[PF], const float (&c_n)[PF], const float (&c_o)[PF]) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (t0 + u >= T) break;
      const float g_r = c_r[u], g_z = c_z[u], g_n = c_n[u];
      __builtin_amdgcn_wave_barrier();
      f32x2 ar2 = {br, 0.f}, az2 = {bz, 0.f}, an2 = {bn, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < HID / 4; ++k4) {
        const float4 hv = *reinterpret_cast<const float4*>(&hbuf[slot][k4 * 4]);
        const f32x2 h0 = {hv.x, hv.y}, h1 = {hv.z, hv.w};
        ar2 = __builtin_elementwise_fma(wr[2 * k4], h0, ar2);
        az2 = __builtin_elementwise_fma(wz[2 * k4], h0, az2);
        an2 = __builtin_elementwise_fma(wn[2 * k4], h0, an2);
        ar2 = __builtin_elementwise_fma(wr[2 * k4 + 1], h1, ar2);
        az2 = __builtin_elementwise_fma(wz[2 * k4 + 1], h1, az2);
        an2 = __builtin_elementwise_fma(wn[2 * k4 + 1], h1, an2);
      }
      const float ar = ar2.x + ar2.y, az = az2.x + az2.y, an = an2.x + an2.y;
      float r = sigmoidf_(g_r + ar), z = sigmoidf_(g_z + az);
      float n = tanhf_(g_n + r * an);
      h = (1.f - z) * n + z * h;
      __builtin_amdgcn_wave_barrier();
      hbuf[slot][j] = h;
      if (active) {
        out[pix * 64 + dir * HID + j] = h;
        if constexpr (RES) out2[pix * 64 + dir * HID + j] = h + c_o[u];
        if (save) {
          float* sp = save + (pix * 2 + dir) * 4 * HID;
          sp[j] = r;
          sp[HID + j] = z;
          sp[2 * HID + j] = n;
          sp[3 * HID + j] = an;
        }
      }
      pix += pstep;
    }
  };
  fetch(0, rA, zA, nA, oA);
  fetch(PF, rB, zB, nB, oB);
  for (int t0 = 0; t0 < T; t0 += 3 * PF) {
    fetch(t0 + 2 * PF, rC, zC, nC, oC);
    steps(t0, rA, zA, nA, oA);
    fetch(t0 + 3 * PF, rA, zA, nA, oA);
    steps(t0 + PF, rB, zB, nB, oB);
    fetch(t0 + 4 * PF, rB, zB, nB, oB);
    steps(t0 + 2 * PF, rC, zC, nC, oC);
  }
}

// BPTT. dout [P,64]; out [P,64] (forward output, gives h_prev); save [P,2,4,32];
// writes dgi [P,192] (dr,dz,dn), dghn [P,64] (dir*32 + j: dn*r, the only gate where the gradient of gh differs from that of gi --
// the W_hh gradient reads its r / z columns from dgi), hprev [P,64].
__global__ __launch_bounds__(256) void gru32_bwd(const float* __restrict__ dout, const float* __restrict__ out,
                                                 const float* __restrict__ save, const float* __restrict__ whh_f,
                                                 const float* __restrict__ whh_b, float* __restrict__ dgi,
                                                 float* __restrict__ dghn, float* __restrict__ hprev_out, int nseq, SeqMap sm) {
  __shared__ __attribute__((aligned(16))) float gbuf[8][3 * HID];
  const int tid = threadIdx.x, j = tid & 31, slot = tid >> 5;
  int wi = blockIdx.x * 8 + slot;
  const bool active = wi < nseq * 2;
  if (!active) wi = nseq * 2 - 1;
  const int s = wi >> 1, dir = wi & 1;
  const float* whh = dir ? whh_b : whh_f;
  f32x2 wc[3 * HID / 2];  // column j of W_hh
#pragma unroll
  for (int i = 0; i < 3 * HID / 2; ++i) wc[i] = f32x2{whh[2 * i * HID + j], whh[(2 * i + 1) * HID + j]};
  const long base = seq_base(sm, s);
  const int T = sm.T;
  // forward order for dir 0 is t=0..T-1, so BPTT visits T-1..0; for dir 1 the reverse.
  long pix = base + (long)(dir ? 0 : T - 1) * sm.tstride;
  const long pstep = dir ? sm.tstride : -sm.tstride;  // towards the forward-order predecessor
  float dh = 0.f;
  // The six per-step operands are fetched in sets of PF steps, TWO sets ahead of their use (time index clamped rather than
  // predicated: no branches around the loads).  One set ahead was 4 steps = 0.8 us of cover, less than the memory latency behind
  // the step's own seven stores (they share the counter the loads are waited on): the horizontal pass (one wave per SIMD,
  // nothing else to hide it) stalled every trip, 1.1 us per step instead of 0.2.
  constexpr int PF = 4;
  const long pix0 = pix;
  float sA[PF][4], hA[PF], dA[PF], sB[PF][4], hB[PF], dB[PF], sC[PF][4], hC[PF], dC[PF];
  auto fetch = [&](int t0, float (&fs)[PF][4], float (&fhp)[PF], float (&fdo)[PF]) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int tc = min(t0 + u, T - 1), tp = min(t0 + u + 1, T - 1);
      const long pc = pix0 + tc * pstep;
      const float* sp = save + (pc * 2 + dir) * 4 * HID + j;
#pragma unroll
      for (int e = 0; e < 4; ++e) fs[u][e] = sp[e * HID];
      fhp[u] = out[(pix0 + tp * pstep) * 64 + dir * HID + j];
      fdo[u] = dout[pc * 64 + dir * HID + j];
    }
  };
  auto steps = [&](int t0, const float (&c_s)[PF][4], const float (&c_hp)[PF], const float (&c_do)[PF]) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int t = t0 + u;
      if (t >= T) break;
      const float r = c_s[u][0], z = c_s[u][1], n = c_s[u][2], ghn = c_s[u][3];
      const float hp = (t + 1 < T) ? c_hp[u] : 0.f;
      float d = dh + c_do[u];
      float dn = d * (1.f - z) * (1.f - n * n);
      float dz = d * (hp - n) * z * (1.f - z);
      float dr = dn * ghn * r * (1.f - r);
      float dgn = dn * r;
      __builtin_amdgcn_wave_barrier();
      gbuf[slot][j] = dr;
      gbuf[slot][HID + j] = dz;
      gbuf[slot][2 * HID + j] = dgn;
      if (active) {
        float* gi_o = dgi + pix * 192 + dir * 96;
        gi_o[j] = dr; gi_o[HID + j] = dz; gi_o[2 * HID + j] = dn;
        dghn[pix * 64 + dir * HID + j] = dgn;
        hprev_out[pix * 64 + dir * HID + j] = hp;
      }
      __builtin_amdgcn_wave_barrier();
      f32x2 a0 = {d * z, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
      for (int i4 = 0; i4 < 3 * HID / 4; ++i4) {
        const float4 gv = *reinterpret_cast<const float4*>(&gbuf[slot][i4 * 4]);
        a0 = __builtin_elementwise_fma(wc[2 * i4], f32x2{gv.x, gv.y}, a0);
        a1 = __builtin_elementwise_fma(wc[2 * i4 + 1], f32x2{gv.z, gv.w}, a1);
      }
      dh = (a0.x + a1.x) + (a0.y + a1.y);
      pix += pstep;
    }
  };
  fetch(0, sA, hA, dA);
  fetch(PF, sB, hB, dB);
  for (int t0 = 0; t0 < T; t0 += 3 * PF) {
    fetch(t0 + 2 * PF, sC, hC, dC);
    steps(t0, sA, hA, dA);
    fetch(t0 + 3 * PF, sA, hA, dA);
    steps(t0 + PF, sB, hB, dB);
    fetch(t0 + 4 * PF, sB, hB, dB);
    steps(t0 + 2 * PF, sC, hC, dC);
  }
}

// ---- RPE GRU (hidden 512, 64 rows w, both directions) ---------------------------------------------------------
// layouts: gi [2][64][1536] (time invariant); gh [2][64][1536] (this step, bias included); h buffers [2][64][512].
// query_pos is NHWC [B][16][64][64]; feature f = dir*512 + j -> (hh = f/64, c = f%64); time index of step s is
// s for dir 0 and B-1-s for dir 1.
constexpr int RH = 512, RW = 64;

__global__ void rpe_gate_fwd(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ bhh_f,
                             const float* __restrict__ bhh_b, const float* __restrict__ hprev, float* __restrict__ hout,
                             float* __restrict__ qp, int s, int B, long dir_stride_gh, long dir_stride_h) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;  // (dir, w, j)
  if (i >= 2 * RW * RH) return;
  int j = i % RH, w = (i / RH) % RW, dir = i / (RH * RW);
  const float* g_i = gi + ((long)dir * RW + w) * 3 * RH;
  const float* g_h = gh + dir * dir_stride_gh + (long)w * 3 * RH;
  const float* bh = dir ? bhh_b : bhh_f;
  float hp = hprev[dir * dir_stride_h + (long)w * RH + j];
  float r = sigmoidf_(g_i[j] + g_h[j] + bh[j]);
  float z = sigmoidf_(g_i[RH + j] + g_h[RH + j] + bh[RH + j]);
  float n = tanhf_(g_i[2 * RH + j] + r * (g_h[2 * RH + j] + bh[2 * RH + j]));
  float h = (1.f - z) * n + z * hp;
  hout[dir * dir_stride_h + (long)w * RH + j] = h;
  int tt = dir ? B - 1 - s : s;
  int f = dir * RH + j;
  qp[(((long)tt * 16 + (f >> 6)) * RW + w) * 64 + (f & 63)] = h;
}

// carry [2][64][512]: on entry dL/dh_s (recurrent part), on exit dh*z (the caller's GEMM then adds dgh*W_hh)
__global__ void rpe_gate_bwd(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ bhh_f,
                             const float* __restrict__ bhh_b, const float* __restrict__ hprev, const float* __restrict__ dqp, float* __restrict__ carry, float* __restrict__ dgh,
                             float* __restrict__ dgi_sum, int s, int B, long dir_stride_gh, long dir_stride_h) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * RW * RH) return;
  int j = i % RH, w = (i / RH) % RW, dir = i / (RH * RW);
  const float* g_i = gi + ((long)dir * RW + w) * 3 * RH;
  const float* g_h = gh + dir * dir_stride_gh + (long)w * 3 * RH;
  const float* bh = dir ? bhh_b : bhh_f;
  float hp = hprev[dir * dir_stride_h + (long)w * RH + j];
  float ghn = g_h[2 * RH + j] + bh[2 * RH + j];
  float r = sigmoidf_(g_i[j] + g_h[j] + bh[j]);
  float z = sigmoidf_(g_i[RH + j] + g_h[RH + j] + bh[RH + j]);
  float n = tanhf_(g_i[2 * RH + j] + r * ghn);
  int tt = dir ? B - 1 - s : s;
  int f = dir * RH + j;
  long ci = ((long)dir * RW + w) * RH + j;
  float d = carry[ci] + dqp[(((long)tt * 16 + (f >> 6)) * RW + w) * 64 + (f & 63)];
  float dn = d * (1.f - z) * (1.f - n * n);
  float dz = d * (hp - n) * z * (1.f - z);
  float dr = dn * ghn * r * (1.f - r);
  float* o = dgh + dir * dir_stride_gh + (long)w * 3 * RH;
  o[j] = dr; o[RH + j] = dz; o[2 * RH + j] = dn * r;
  float* gs = dgi_sum + ((long)dir * RW + w) * 3 * RH;
  gs[j] += dr; gs[RH + j] += dz; gs[2 * RH + j] += dn;
  carry[ci] = d * z;
}

// x[w][hh*64+c] = init_factor[(hh*64+w)*64 + c]  (transformer_v2.py:215-217 with identical rows over the batch)
__global__ void rpe_gather_x(const float* __restrict__ init_factor, float* __restrict__ x) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= RW * 1024) return;
  int f = i % 1024, w = i / 1024;
  x[i] = init_factor[((long)(f >> 6) * RW + w) * 64 + (f & 63)];
}
__global__ void rpe_scatter_dx(const float* __restrict__ dx, float* __restrict__ dinit, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= RW * 1024) return;
  int f = i % 1024, w = i / 1024;
  long o = ((long)(f >> 6) * RW + w) * 64 + (f & 63);
  dinit[o] = accumulate ? dinit[o] + dx[i] : dx[i];
}

// ---- fused RPE steps ---------------------------------------------------------------------------------------------
// The recurrence over the batch axis is a chain of 2*B dependent [64,512]x[512,1536] products: latency, not FLOPs.  One
// launch per step does the product, the cross-wave reduction and the gate math (instead of split-K GEMM + reduce + gate
// kernel), with v_mfma_f32_16x16x4_f32 tiles fed straight from global memory: lane (i = l&15, q = l>>4) loads float4s of
// row i at k = kbase + 16e + 4q (the four q-lanes of a row cover one 64-byte line per load), A and B with the same k
// order, so no LDS staging is needed.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma4(f32x4& acc, const float4& a, const float4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
}

// forward step: block (jb, dir) owns hidden units j0..j0+3 (12 gate columns, padded to one 16-wide tile) for all 64 rows;
// the NWV waves split K = 512 (eight: half the MFMA chain and half the loads per wave of four).  Writes h_{s+1}, query_pos and the raw gh (no bias) that the backward pass re-reads.
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void rpe_step_fwd(const float* __restrict__ gi, const float* __restrict__ whh_f,
                                                    const float* __restrict__ whh_b, const float* __restrict__ bhh_f,
                                                    const float* __restrict__ bhh_b, const float* __restrict__ hprev,
                                                    float* __restrict__ hout, float* __restrict__ gh_out, float* __restrict__ qp,
                                                    int s, int B, long dir_stride_gh, long dir_stride_h) {
  constexpr int NE = RH / NWV / 16;  // float4 k-steps per wave
  __shared__ float red[NWV][RW][17];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
  const int dir = blockIdx.y, j0 = blockIdx.x * 4;
  const float* W = dir ? whh_b : whh_f;
  const float* hp = hprev + dir * dir_stride_h;
  const int kbase = wv * (RH / NWV) + q * 4;  // step e: k = kbase + 16e (+c): the 4 q-lanes of a row read one 64-byte line
  float4 bf[NE], af[4][NE];
  {
    const bool on = i < 12;
    const float* pb = W + (long)((i >> 2) * RH + j0 + (i & 3)) * RH + kbase;
    const float* pa = hp + (long)i * RH + kbase;
#pragma unroll
    for (int e = 0; e < NE; ++e) {  // issued in the order the MFMAs consume them
      bf[e] = on ? *reinterpret_cast<const float4*>(pb + 16 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt][e] = *reinterpret_cast<const float4*>(pa + (long)mt * 16 * RH + 16 * e);
    }
  }
  // gate operands of this thread's (row, unit): issued before the MFMAs so that their latency hides behind them
  const int gt = tid & 255;  // gate element of this thread (threads 256.. of an 8-wave block only take part in the product)
  const int w = gt >> 2, jj = gt & 3, j = j0 + jj;
  const float* g_i = gi + ((long)dir * RW + w) * 3 * RH;
  const float* bh = dir ? bhh_b : bhh_f;
  const float gir = g_i[j], giz = g_i[RH + j], gin = g_i[2 * RH + j];
  const float br = bh[j], bz = bh[RH + j], bn = bh[2 * RH + j];
  const float hpv = hp[(long)w * RH + j];
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_sched_barrier(0);  // every load above is issued before the first MFMA: the step is pure latency
#pragma unroll
  for (int e = 0; e < NE; ++e)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) mfma4(acc[mt], af[mt][e], bf[e]);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][mt * 16 + 4 * q + r][i] = acc[mt][r];
  __syncthreads();
  if (tid >= 256) return;
  float gh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    float t = 0.f;
#pragma unroll
    for (int v = 0; v < NWV; v += 2) t += red[v][w][g * 4 + jj] + red[v + 1][w][g * 4 + jj];
    gh[g] = t;
  }
  const float r = sigmoidf_(gir + gh[0] + br);
  const float z = sigmoidf_(giz + gh[1] + bz);
  const float n = tanhf_(gin + r * (gh[2] + bn));
  const float h = (1.f - z) * n + z * hpv;
  hout[dir * dir_stride_h + (long)w * RH + j] = h;
  float* go = gh_out + dir * dir_stride_gh + (long)w * 3 * RH;
  go[j] = gh[0];
  go[RH + j] = gh[1];
  go[2 * RH + j] = gh[2];
  const int tt = dir ? B - 1 - s : s;
  const int f = dir * RH + j;
  qp[(((long)tt * 16 + (f >> 6)) * RW + w) * 64 + (f & 63)] = h;
}

// ---- the whole forward chain as ONE launch, one XCD per (row block, direction) ---------------------------------------------------
// The 64 rows of the recurrence are independent chains, and 4 row blocks of 16 x 2 directions = 8 = the number of XCDs.  XCD x runs
// the chain of (row block x & 3, direction x >> 2) entirely inside its own L2: 32 persistent workgroups (one per 16 hidden units =
// 48 gate columns), each holding its 96 KB slice of W_hh in registers for all B steps and exchanging the 16 x 512 state through L2
// (a CU's stores write through to it) so that a step costs the hand-over inside one L2 + the product
// (6.4 us measured), where a kernel boundary per step cost 9.4 us in a graph and 12 us launched eagerly (launch gap, cold caches,
// W_hh re-read every step).
//   * Which XCD a workgroup is on is read from the hardware (HW_REG_XCC_ID), not inferred from its id, and which chain an XCD runs
//     follows from that id and the device's XCD count (chain_role below): nothing depends on the dispatcher's placement order beyond
//     every XCD receiving its share of the grid.  The launch oversubscribes (grid = 2 x 256 workgroups); the ones not needed exit at once.
//   * Hand-over: an owner stores its 16 x 16 block of h_{s+1}, waits until the stores have reached L2 (vmcnt(0)), then writes "s+1" into
//     its flag word -- one copy per reader, so that every reader polls a 128-byte line of its own.  A reader polls with ONE returning
//     L2 atomic (OR of 0) on its 32 flag words -- a plain load could be served from the CU's cache for ever -- and then reads h_s with
//     ordinary loads: nobody touches those lines before they are complete, so the CU cannot hold a stale copy of them.
//     Tried on the way (tools/micro_rpe_chain.py, us per step): polling the data itself against a NaN sentinel with device-scope loads
//     6.8 (through the fabric), with L2 atomics 34, behind an L1 invalidate (`buffer_inv sc1`) 47, flag counters bumped by 256
//     same-address device atomics 34; per-step kernels in a graph 9.4; this scheme 6.4, of which 1.3 are the product itself (0.2 GFLOP
//     per step is the whole chip's fp32 matrix rate for 1.3 us) and about 2 each the flag hand-over and the h read.
//   * A workgroup only waits for the other 31 owners of ITS XCD, all of which have started (they hold tickets): no deadlock, no
//     grid-wide barrier.  The wait is bounded all the same: after RPE_SPIN_LIMIT polls a workgroup sets sync[1] and carries on
//     with what is there (wrong numbers and an error flag, never a hang) -- e.g. if an XCD ever received fewer than 32 workgroups.
//   * gh and query_pos are only read by later kernels (ordinary stores; visible at the end of the kernel like any output).
//   * Hardware behaviour this relies on (gfx942 / gfx950; outside the HIP memory model, hence the architecture gate in
//     tatt_rpe_chain_supported and the bit-compare against the per-step kernels in the tests): the CU's vector cache (TCP) is
//     write-through and does not allocate a line on a store, so a CU that half-wrote a 128-byte line of h one step earlier holds no
//     stale copy of it; `s_waitcnt vmcnt(0)` after a store returns once L2 has acknowledged it; every CU of an XCD shares that XCD's
//     L2, so data published there needs no fabric round trip.  Only time-outs are detected at run time (flag word + NaN poison).
constexpr unsigned RPE_SPIN_LIMIT = 1u << 18;  // polls of ~1-2 us each before a reader gives up
constexpr int RPE_SYNC_WORDS = 32 * 9 + 8 * 1024;          // sync[0..7]: header the calls do NOT clear -- sync[1] sticky error flags (caller zeroes the buffer once); sync[32 * (1 + x)] XCD x's ticket counter (a line of its own); then per chain [reader 32][writer 32] flag words
// read a word AT the L2 of this XCD (returning atomic OR of 0: a plain load could be served from the CU's cache for ever)
__device__ __forceinline__ unsigned l2_read(unsigned* p) {
  unsigned v;
  const unsigned zero = 0u;
  asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(zero) : "memory");
  return v;
}
// Which of the 8 chains (row block mh, direction dir) this workgroup works on and which 16 hidden units (jb); false: not needed, exit.
// Workgroups that arrive on the same XCD form groups of 32 in arrival order (a ticket from that XCD's counter, an L2 atomic); group k
// of XCD x runs chain x + nxcd * k (nxcd = hipDeviceAttributeNumberOfXccs, passed by the host).  On MI355X (8 XCDs, grid = 512
// dealt round-robin) the first group of every XCD gets that XCD's chain and the second exits; with fewer XCDs visible (another
// partition mode) an XCD runs several chains -- every chain still lives inside one L2.  (An earlier version drew the chain from a
// device-wide counter: in a graph, next to other kernels, an XCD that started late lost its chain to the SECOND group of an early
// one, two chains then shared one XCD and the step was 0.3 ms slower than with per-step launches.)
__device__ __forceinline__ bool chain_role(unsigned* sync, int nxcd, int& jb, int& mh, int& dir, unsigned*& flags) {
  __shared__ unsigned s_role[2];
  if (threadIdx.x == 0) {
    const int x = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15) % nxcd;  // HW_REG_XCC_ID[3:0]
    unsigned* line = sync + 32 * (1 + x);  // this XCD's own 128-byte line: [0] tickets
    const unsigned t = __hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    s_role[0] = (unsigned)x + (unsigned)nxcd * (t >> 5);
    s_role[1] = t & 31;
  }
  __syncthreads();
  const unsigned role = s_role[0];
  jb = (int)s_role[1];
  mh = (int)(role & 3);
  dir = (int)((role >> 2) & 1);
  flags = sync + 32 * 9 + 1024 * (role & 7);  // this chain's flag words: [reader 32][writer 32], a 128-byte line per reader
  return role < 8u;
}

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void rpe_chain_fwd(const float* __restrict__ gi, const float* __restrict__ whh_f,
                                                          const float* __restrict__ whh_b, const float* __restrict__ bhh_f,
                                                          const float* __restrict__ bhh_b, float* hstep, float* __restrict__ gh_all,
                                                          float* __restrict__ qp, int B, long dir_stride_gh, long dir_stride_h,
                                                          unsigned* sync, int nxcd) {
  constexpr int NE = RH / NWV / 16;
  __shared__ float red[NWV][16][49];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
  int jb, mh, dir;
  unsigned* flags;
  if (!chain_role(sync, nxcd, jb, mh, dir, flags)) return;
  const float* W = dir ? whh_b : whh_f;
  const int kbase = wv * (RH / NWV) + q * 4;
  // ---- once for the whole chain: the W_hh slice and the (time-invariant) gate operands ----
  float4 bf[3][NE];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const float* pb = W + (long)(g * RH + jb * 16 + i) * RH + kbase;
#pragma unroll
    for (int e = 0; e < NE; ++e) bf[g][e] = *reinterpret_cast<const float4*>(pb + 16 * e);
  }
  const int gt = tid & 255;
  const int wl = gt >> 4, jj = gt & 15, w = mh * 16 + wl, j = jb * 16 + jj;
  const float* g_i = gi + ((long)dir * RW + w) * 3 * RH;
  const float* bh = dir ? bhh_b : bhh_f;
  const float gir = g_i[j], giz = g_i[RH + j], gin = g_i[2 * RH + j];
  const float br = bh[j], bz = bh[RH + j], bn = bh[2 * RH + j];
  float hpv = hstep[dir * dir_stride_h + (long)w * RH + j];  // h_0; afterwards this thread's own previous output
  float gh_s[3] = {0.f, 0.f, 0.f};
  bool bad = false;  // (workgroup-uniform) a wait ran out: from then on query_pos is written as NaN, so that the loss, the clip norm
                     // and every later check see the failure even where nobody reads the flag word
  for (int s = 0; s < B; ++s) {
    const long hoff = dir * dir_stride_h + (long)s * RW * RH;  // h_s of this direction; h_{s+1} follows RW * RH floats later
    // ---- h_s of this row block: written by this XCD's 32 owners in step s-1, each of which then set its flag word to s ----
    if (s > 0) {
      for (unsigned it = 0;; ++it) {
        unsigned f = 0xffffffffu;
        if (tid < 32) f = l2_read(flags + jb * 32 + tid);
        if (__syncthreads_and(f >= (unsigned)s)) break;
        if (it > RPE_SPIN_LIMIT) {  // never a hang: give up, flag it, carry on with what is there
          if (tid == 0) __hip_atomic_fetch_or(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    // first touch of these lines by this CU in this launch (nobody reads h_s before it is complete): ordinary loads, served by L2
    float4 af[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) af[e] = *reinterpret_cast<const float4*>(hstep + hoff + (long)(mh * 16 + i) * RH + kbase + 16 * e);
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int g = 0; g < 3; ++g) mfma4(acc[g], af[e], bf[g][e]);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wv][4 * q + r][g * 16 + i] = acc[g][r];
    __syncthreads();
    if (tid < 256) {
      float gh[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        float t = 0.f;
#pragma unroll
        for (int v = 0; v < NWV; v += 2) t += red[v][wl][g * 16 + jj] + red[v + 1][wl][g * 16 + jj];
        gh[g] = t;
      }
      const float r = sigmoidf_(gir + gh[0] + br);
      const float z = sigmoidf_(giz + gh[1] + bz);
      const float n = tanhf_(gin + r * (gh[2] + bn));
      const float h = (1.f - z) * n + z * hpv;
      hpv = h;
      hstep[hoff + (long)RW * RH + (long)w * RH + j] = h;
      gh_s[0] = gh[0]; gh_s[1] = gh[1]; gh_s[2] = gh[2];
    }
    // publish: every thread's h store has reached L2 (vmcnt(0)), then this workgroup's flag word says "s + 1 steps done"
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid < 32) *reinterpret_cast<volatile unsigned*>(flags + tid * 32 + jb) = (unsigned)s + 1u;  // one copy per reader: each polls its own line
    if (tid < 256) {  // off the chain's critical path: what only later kernels read
      float* go = gh_all + dir * dir_stride_gh + ((long)s * RW + w) * 3 * RH;
      go[j] = gh_s[0];
      go[RH + j] = gh_s[1];
      go[2 * RH + j] = gh_s[2];
      const int tt = dir ? B - 1 - s : s;
      const int f = dir * RH + j;
      qp[(((long)tt * 16 + (f >> 6)) * RW + w) * 64 + (f & 63)] = bad ? __builtin_nanf("") : hpv;
    }
  }
}

// whhT[dir][j][i] = whh_dir[i][j]  ([1536][512] -> [512][1536]); once per backward pass
__global__ void rpe_whh_transpose(const float* __restrict__ whh_f, const float* __restrict__ whh_b, float* __restrict__ whhT) {
  __shared__ float t[32][33];
  const float* W = blockIdx.z ? whh_b : whh_f;
  float* O = whhT + (long)blockIdx.z * RH * 3 * RH;
  const int i0 = blockIdx.y * 32, jb = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) t[r][tx] = W[(long)(i0 + r) * RH + jb + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) O[(long)(jb + r) * 3 * RH + i0 + tx] = t[tx][r];
}

// backward step s: block (jb, mh, dir) owns rows mh*16*MT..+16*MT-1, units jb*16..+15 (MT = 1: 256 workgroups, one per CU; with
// MT = 2 the 128 workgroups left half of the CUs idle and each wave had twice the MFMA chain).  First the recurrent gradient arriving from
// step s+1, carry = d_{s+1} z_{s+1} (left in `carry` by that step) + dgh_{s+1} W_hh (K = 1536 split over the four waves,
// three 128-wide chunks each), then the gate backward of step s for the same (row, unit) elements.  dgh_next == nullptr
// for the first (last-in-time) step.
template <int MT, int NWV>
__global__ __launch_bounds__(64 * NWV) void rpe_step_bwd(const float* __restrict__ gi, const float* __restrict__ gh,
                                                    const float* __restrict__ bhh_f, const float* __restrict__ bhh_b,
                                                    const float* __restrict__ hprev, const float* __restrict__ dqp,
                                                    float* __restrict__ carry, const float* __restrict__ dgh_next,
                                                    float* __restrict__ dgh_out, float* __restrict__ dgi_sum,
                                                    const float* __restrict__ whhT, int s, int B, long dir_stride_gh,
                                                    long dir_stride_h) {
  constexpr int RM = 16 * MT;  // rows per workgroup
  constexpr int KW = 3 * RH / NWV, CH = KW / 3, NE = CH / 16;  // K range of a wave, in three chunks of NE float4 steps
  __shared__ float red[NWV][RM][17];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
  const int jb = blockIdx.x, mh = blockIdx.y, dir = blockIdx.z;
  // gate operands of this thread's (row, unit) elements: issued before the product so that their latency hides behind it
  // (they do not depend on it); threads 256.. of the workgroup only take part in the product
  const float* bh = dir ? bhh_b : bhh_f;
  const int tt = dir ? B - 1 - s : s;
  const int gt = tid & 255;
  float g_c[MT], g_ir[MT], g_iz[MT], g_in[MT], g_hr[MT], g_hz[MT], g_hn[MT], g_hp[MT], g_dq[MT], g_br[MT], g_bz[MT], g_bn[MT];
#pragma unroll
  for (int h2 = 0; h2 < MT; ++h2) {
    const int e = gt + 256 * h2, wl = e >> 4, jj = e & 15;
    const int w = mh * RM + wl, j = jb * 16 + jj;
    const float* g_i = gi + ((long)dir * RW + w) * 3 * RH;
    const float* g_h = gh + dir * dir_stride_gh + (long)w * 3 * RH;
    const int f = dir * RH + j;
    g_c[h2] = carry[((long)dir * RW + w) * RH + j];
    g_ir[h2] = g_i[j]; g_iz[h2] = g_i[RH + j]; g_in[h2] = g_i[2 * RH + j];
    g_hr[h2] = g_h[j]; g_hz[h2] = g_h[RH + j]; g_hn[h2] = g_h[2 * RH + j];
    g_br[h2] = bh[j]; g_bz[h2] = bh[RH + j]; g_bn[h2] = bh[2 * RH + j];
    g_hp[h2] = hprev[dir * dir_stride_h + (long)w * RH + j];
    g_dq[h2] = dqp[(((long)tt * 16 + (f >> 6)) * RW + w) * 64 + (f & 63)];
  }
  if (dgh_next) {
    const float* A = dgh_next + dir * dir_stride_gh + (long)(mh * RM + i) * 3 * RH + wv * KW + q * 4;
    const float* Bt = whhT + ((long)dir * RH + jb * 16 + i) * 3 * RH + wv * KW + q * 4;
    float4 af[3][MT][NE], bf[3][NE];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {  // issued in the order the MFMAs consume them
        bf[c][e] = *reinterpret_cast<const float4*>(Bt + c * CH + 16 * e);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[c][mt][e] = *reinterpret_cast<const float4*>(A + (long)mt * 16 * 3 * RH + c * CH + 16 * e);
      }
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);  // every load in flight before the first MFMA: the step is pure latency
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mfma4(acc[mt], af[c][mt][e], bf[c][e]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wv][mt * 16 + 4 * q + r][i] = acc[mt][r];
  }
  __syncthreads();
  if (tid >= 256) return;  // the gate phase has 256 elements per 16 rows
#pragma unroll
  for (int h2 = 0; h2 < MT; ++h2) {
    const int e = tid + 256 * h2, wl = e >> 4, jj = e & 15;
    const int w = mh * RM + wl, j = jb * 16 + jj;
    const long ci = ((long)dir * RW + w) * RH + j;
    float c = g_c[h2];
    if (dgh_next) {
#pragma unroll
      for (int v = 0; v < NWV; v += 2) c += red[v][wl][jj] + red[v + 1][wl][jj];
    }
    const float hp = g_hp[h2];
    const float ghn = g_hn[h2] + g_bn[h2];
    const float r = sigmoidf_(g_ir[h2] + g_hr[h2] + g_br[h2]);
    const float z = sigmoidf_(g_iz[h2] + g_hz[h2] + g_bz[h2]);
    const float n = tanhf_(g_in[h2] + r * ghn);
    const float d = c + g_dq[h2];
    const float dn = d * (1.f - z) * (1.f - n * n);
    const float dz = d * (hp - n) * z * (1.f - z);
    const float dr = dn * ghn * r * (1.f - r);
    float* o = dgh_out + dir * dir_stride_gh + (long)w * 3 * RH;
    o[j] = dr;
    o[RH + j] = dz;
    o[2 * RH + j] = dn * r;
    float* gs = dgi_sum + ((long)dir * RW + w) * 3 * RH;
    gs[j] += dr;
    gs[RH + j] += dz;
    gs[2 * RH + j] += dn;
    carry[ci] = d * z;
  }
}

// ---- the BPTT chain as ONE launch: same placement as rpe_chain_fwd (XCD = (row block, direction), 32 persistent owners of 16 hidden
// units each).  Per step s = B-1 .. 0 an owner needs dgh_{s+1} of its 16 rows -- written by its XCD's 32 owners one step earlier,
// handed over through the same per-reader flag lines -- multiplies it with its 96 KB slice of W_hh (transposed; in registers for the
// whole chain), and differentiates the gates of step s for its 16 x 16 elements.  The carried d*z and the running sum of dgi belong to
// one thread each for the whole chain: registers, read from / added to memory once.
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void rpe_chain_bwd(const float* __restrict__ gi, const float* __restrict__ gh_all,
                                                          const float* __restrict__ bhh_f, const float* __restrict__ bhh_b,
                                                          const float* __restrict__ hstep, const float* __restrict__ dqp,
                                                          float* __restrict__ carry, float* dgh_all, float* __restrict__ dgi_sum,
                                                          const float* __restrict__ whhT, int B, long dir_stride_gh, long dir_stride_h,
                                                          unsigned* sync, int nxcd) {
  constexpr int KW = 3 * RH / NWV, CH = KW / 3, NE = CH / 16;  // K range of a wave, in three chunks of NE float4 steps
  __shared__ float red[NWV][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
  int jb, mh, dir;
  unsigned* flags;
  if (!chain_role(sync, nxcd, jb, mh, dir, flags)) return;
  // ---- once for the whole chain: the W_hh^T slice and the time-invariant gate operands ----
  float4 bf[3][NE];
  {
    const float* Bt = whhT + ((long)dir * RH + jb * 16 + i) * 3 * RH + wv * KW + q * 4;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < NE; ++e) bf[c][e] = *reinterpret_cast<const float4*>(Bt + c * CH + 16 * e);
  }
  const int gt = tid & 255;
  const int wl = gt >> 4, jj = gt & 15, w = mh * 16 + wl, j = jb * 16 + jj;
  const float* bh = dir ? bhh_b : bhh_f;
  const float* g_i = gi + ((long)dir * RW + w) * 3 * RH;
  const float g_ir = g_i[j], g_iz = g_i[RH + j], g_in = g_i[2 * RH + j];
  const float g_br = bh[j], g_bz = bh[RH + j], g_bn = bh[2 * RH + j];
  const long ci = ((long)dir * RW + w) * RH + j;
  float cr = carry[ci];                      // d_{s+1} z_{s+1}: this thread's own, step after step
  float sum_r = 0.f, sum_z = 0.f, sum_n = 0.f;  // this element's share of dgi_sum
  bool bad = false;                             // a wait ran out: dgi_sum (-> the gradients of W_ih, init_factor) is poisoned with NaN
  for (int s = B - 1; s >= 0; --s) {
    // what this step needs from the forward pass and from the loss: independent of the chain, requested before the wait
    const float* g_h = gh_all + dir * dir_stride_gh + ((long)s * RW + w) * 3 * RH;
    const float g_hr = g_h[j], g_hz = g_h[RH + j], g_hn = g_h[2 * RH + j];
    const float hp = hstep[dir * dir_stride_h + ((long)s * RW + w) * RH + j];
    const int tt = dir ? B - 1 - s : s;
    const int f = dir * RH + j;
    const float dq = dqp[(((long)tt * 16 + (f >> 6)) * RW + w) * 64 + (f & 63)];
    float c = cr;
    if (s + 1 < B) {
      // ---- dgh_{s+1} of this row block: every owner of this XCD has set its flag word to B - 1 - s once it was stored ----
      for (unsigned it = 0;; ++it) {
        unsigned fl = 0xffffffffu;
        if (tid < 32) fl = l2_read(flags + jb * 32 + tid);
        if (__syncthreads_and(fl >= (unsigned)(B - 1 - s))) break;
        if (it > RPE_SPIN_LIMIT) {
          if (tid == 0) __hip_atomic_fetch_or(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      const float* A = dgh_all + dir * dir_stride_gh + ((long)(s + 1) * RW + mh * 16 + i) * 3 * RH + wv * KW + q * 4;
      float4 af[3][NE];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int e = 0; e < NE; ++e) af[cc][e] = *reinterpret_cast<const float4*>(A + cc * CH + 16 * e);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int e = 0; e < NE; ++e) mfma4(acc, af[cc][e], bf[cc][e]);
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wv][4 * q + r][i] = acc[r];
      __syncthreads();
      if (tid < 256) {
#pragma unroll
        for (int v = 0; v < NWV; v += 2) c += red[v][wl][jj] + red[v + 1][wl][jj];
      }
    }
    if (tid < 256) {
      const float ghn = g_hn + g_bn;
      const float r = sigmoidf_(g_ir + g_hr + g_br);
      const float z = sigmoidf_(g_iz + g_hz + g_bz);
      const float n = tanhf_(g_in + r * ghn);
      const float d = c + dq;
      const float dn = d * (1.f - z) * (1.f - n * n);
      const float dz = d * (hp - n) * z * (1.f - z);
      const float dr = dn * ghn * r * (1.f - r);
      float* o = dgh_all + dir * dir_stride_gh + ((long)s * RW + w) * 3 * RH;
      o[j] = dr;
      o[RH + j] = dz;
      o[2 * RH + j] = dn * r;
      sum_r += dr;
      sum_z += dz;
      sum_n += dn;
      cr = d * z;
    }
    // publish: the dgh stores have reached L2, then "B - s steps done" into this owner's flag word of every reader
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid < 32) *reinterpret_cast<volatile unsigned*>(flags + tid * 32 + jb) = (unsigned)(B - s);
  }
  if (tid < 256) {
    float* gs = dgi_sum + ((long)dir * RW + w) * 3 * RH;
    gs[j] += bad ? __builtin_nanf("") : sum_r;
    gs[RH + j] += sum_z;
    gs[2 * RH + j] += sum_n;
    carry[ci] = cr;
  }
}

}  // namespace

extern "C" {

int tatt_rpe_step_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b,
                      const float* hprev, float* hout, float* gh_out, float* qp, int s, int B, long dir_stride_gh,
                      long dir_stride_h, hipStream_t st) {
  TATT_CHECK_ARG(gi && whh_f && whh_b && hprev && hout && gh_out && qp && s >= 0 && s < B, "tatt_rpe_step_fwd: bad arguments");
  hipLaunchKernelGGL(rpe_step_fwd<8>, dim3(RH / 4, 2), dim3(512), 0, st, gi, whh_f, whh_b, bhh_f, bhh_b, hprev, hout, gh_out, qp, s, B,
                     dir_stride_gh, dir_stride_h);
  TATT_LAUNCH_CHECK("rpe_step_fwd");
  return TATT_OK;
}

// Words of the `sync` buffer of the chain launches (uint32; zeroed by the call itself): error flag + one ticket counter per XCD.
// XCDs of the current device (8 on MI355X in its default partition mode); no allocation, no synchronisation, legal under capture
static int rpe_num_xcd() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || n < 1 || n > 8) n = 8;
  return n;
}

// 1 if the chain kernels may be used on the current device: every XCD must be able to hold its chains' 32 workgroups each on a CU of
// their own (MI355X in its default partition mode: 8 XCDs x 32 CUs, one chain per XCD); else 0 -- use tatt_rpe_step_fwd / _bwd.
size_t tatt_rpe_chain_supported(void) {
  static signed char cached[64];  // per device: 0 unknown, 1 no, 2 yes (the answer is a property of the device: asked once)
  int dev = 0, n = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64 && cached[dev]) return (size_t)(cached[dev] == 2);
  hipDeviceProp_t prop;
  bool ok = hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            (strncmp(prop.gcnArchName, "gfx950", 6) == 0 || strncmp(prop.gcnArchName, "gfx942", 6) == 0);  // cache behaviour above
  ok = ok && hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess &&
       hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 1 && n <= 8;
  if (ok) {
    const int chains_per_xcd = (8 + n - 1) / n;
    ok = chains_per_xcd * 32 <= cus / n;
  }
  if (dev >= 0 && dev < 64) cached[dev] = ok ? 2 : 1;
  return (size_t)(ok ? 1 : 0);
}

size_t tatt_rpe_chain_sync_words(int B) { (void)B; return (size_t)RPE_SYNC_WORDS; }

// All B steps of tatt_rpe_step_fwd in one launch (see rpe_chain_fwd).  hstep [2][B+1][64][512] with h_0 = hstep[:, 0] set by the caller,
// gh_all [2][B][64][1536]; sync: tatt_rpe_chain_sync_words(B) uint32 of device memory; after the launch sync[1] != 0 means a workgroup
// gave up waiting for its predecessor step (the results are then invalid).
int tatt_rpe_chain_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b, float* hstep,
                       float* gh_all, float* qp, int B, long dir_stride_gh, long dir_stride_h, uint32_t* sync, hipStream_t st) {
  TATT_CHECK_ARG(gi && whh_f && whh_b && bhh_f && bhh_b && hstep && gh_all && qp && sync && B > 0 && B <= 4096, "tatt_rpe_chain_fwd: bad arguments");
  TATT_CHECK_ARG(2L * (B + 1) * RW * RH * 4 < (1L << 31) && dir_stride_h >= (long)(B + 1) * RW * RH, "tatt_rpe_chain_fwd: hstep too large / bad stride");
  TATT_CHECK_ARG(tatt_rpe_chain_supported(), "tatt_rpe_chain_fwd: this device cannot hold the chains (tatt_rpe_chain_supported() == 0): use the per-step kernels");
  const int nxcd = rpe_num_xcd();
  if (hipMemsetAsync(sync + 8, 0, (tatt_rpe_chain_sync_words(B) - 8) * 4, st) != hipSuccess) {
    tatt_set_error("tatt_rpe_chain_fwd: hipMemsetAsync failed");
    return TATT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(rpe_chain_fwd<8>, dim3(2 * 256), dim3(512), 0, st, gi, whh_f, whh_b, bhh_f, bhh_b, hstep, gh_all, qp, B, dir_stride_gh,
                     dir_stride_h, sync, nxcd);
  TATT_LAUNCH_CHECK("rpe_chain_fwd");
  return TATT_OK;
}

// All B steps of tatt_rpe_step_bwd (s = B-1 .. 0) in one launch: gh_all / dgh_all [2][B][64][1536], hstep [2][B+1][64][512], carry and
// dgi_sum as for the step kernel (carry in / out, dgi_sum accumulated); sync as for tatt_rpe_chain_fwd (a buffer of its own).
int tatt_rpe_chain_bwd(const float* gi, const float* gh_all, const float* bhh_f, const float* bhh_b, const float* hstep, const float* dqp,
                       float* carry, float* dgh_all, float* dgi_sum, const float* whhT, int B, long dir_stride_gh, long dir_stride_h,
                       uint32_t* sync, hipStream_t st) {
  TATT_CHECK_ARG(gi && gh_all && bhh_f && bhh_b && hstep && dqp && carry && dgh_all && dgi_sum && whhT && sync && B > 0 && B <= 4096,
                 "tatt_rpe_chain_bwd: bad arguments");
  TATT_CHECK_ARG(tatt_rpe_chain_supported(), "tatt_rpe_chain_bwd: this device cannot hold the chains (tatt_rpe_chain_supported() == 0): use the per-step kernels");
  const int nxcd = rpe_num_xcd();
  if (hipMemsetAsync(sync + 8, 0, (tatt_rpe_chain_sync_words(B) - 8) * 4, st) != hipSuccess) {
    tatt_set_error("tatt_rpe_chain_bwd: hipMemsetAsync failed");
    return TATT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(rpe_chain_bwd<8>, dim3(2 * 256), dim3(512), 0, st, gi, gh_all, bhh_f, bhh_b, hstep, dqp, carry, dgh_all, dgi_sum, whhT, B,
                     dir_stride_gh, dir_stride_h, sync, nxcd);
  TATT_LAUNCH_CHECK("rpe_chain_bwd");
  return TATT_OK;
}

int tatt_rpe_whh_transpose(const float* whh_f, const float* whh_b, float* whhT, hipStream_t st) {
  hipLaunchKernelGGL(rpe_whh_transpose, dim3(RH / 32, 3 * RH / 32, 2), dim3(256), 0, st, whh_f, whh_b, whhT);
  TATT_LAUNCH_CHECK("rpe_whh_transpose");
  return TATT_OK;
}

int tatt_rpe_step_bwd(const float* gi, const float* gh, const float* bhh_f, const float* bhh_b, const float* hprev,
                      const float* dqp, float* carry, const float* dgh_next, float* dgh_out, float* dgi_sum, const float* whhT,
                      int s, int B, long dir_stride_gh, long dir_stride_h, hipStream_t st) {
  TATT_CHECK_ARG(gi && gh && hprev && dqp && carry && dgh_out && dgi_sum && whhT && s >= 0 && s < B, "tatt_rpe_step_bwd: bad arguments");
  hipLaunchKernelGGL((rpe_step_bwd<1, 8>), dim3(RH / 16, RW / 16, 2), dim3(512), 0, st, gi, gh, bhh_f, bhh_b, hprev, dqp, carry, dgh_next, dgh_out,
                     dgi_sum, whhT, s, B, dir_stride_gh, dir_stride_h);
  TATT_LAUNCH_CHECK("rpe_step_bwd");
  return TATT_OK;
}

int tatt_gru32_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b, float* out,
                   float* save, int nseq, int T, int inner, long outer_stride, long inner_stride, long tstride, hipStream_t st) {
  TATT_CHECK_ARG(nseq > 0 && T > 0 && inner > 0, "tatt_gru32_fwd: bad dims");
  SeqMap sm{inner, outer_stride, inner_stride, tstride, T};
  hipLaunchKernelGGL(gru32_fwd<false>, dim3(cdiv(nseq * 2, 8)), dim3(256), 0, st, gi, whh_f, whh_b, bhh_f, bhh_b, out, save, nseq, sm,
                     (const float*)nullptr, (float*)nullptr);
  TATT_LAUNCH_CHECK("gru32_fwd");
  return TATT_OK;
}

int tatt_gru32_fwd_res(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b, float* out,
                       float* save, const float* res, float* out2, int nseq, int T, int inner, long outer_stride, long inner_stride,
                       long tstride, hipStream_t st) {
  TATT_CHECK_ARG(nseq > 0 && T > 0 && inner > 0 && res && out2 && out2 != out, "tatt_gru32_fwd_res: bad dims / null residual or second output");
  SeqMap sm{inner, outer_stride, inner_stride, tstride, T};
  hipLaunchKernelGGL(gru32_fwd<true>, dim3(cdiv(nseq * 2, 8)), dim3(256), 0, st, gi, whh_f, whh_b, bhh_f, bhh_b, out, save, nseq, sm, res, out2);
  TATT_LAUNCH_CHECK("gru32_fwd_res");
  return TATT_OK;
}

int tatt_gru32_bwd(const float* dout, const float* out, const float* save, const float* whh_f, const float* whh_b, float* dgi,
                   float* dghn, float* hprev, int nseq, int T, int inner, long outer_stride, long inner_stride, long tstride,
                   hipStream_t st) {
  TATT_CHECK_ARG(nseq > 0 && T > 0 && inner > 0, "tatt_gru32_bwd: bad dims");
  SeqMap sm{inner, outer_stride, inner_stride, tstride, T};
  hipLaunchKernelGGL(gru32_bwd, dim3(cdiv(nseq * 2, 8)), dim3(256), 0, st, dout, out, save, whh_f, whh_b, dgi, dghn, hprev, nseq, sm);
  TATT_LAUNCH_CHECK("gru32_bwd");
  return TATT_OK;
}

int tatt_rpe_gate_fwd(const float* gi, const float* gh, const float* bhh_f, const float* bhh_b, const float* hprev, float* hout,
                      float* qp, int s, int B, long dir_stride_gh, long dir_stride_h, hipStream_t st) {
  hipLaunchKernelGGL(rpe_gate_fwd, dim3(cdiv(2 * RW * RH, 256)), dim3(256), 0, st, gi, gh, bhh_f, bhh_b, hprev, hout, qp, s, B,
                     dir_stride_gh, dir_stride_h);
  TATT_LAUNCH_CHECK("rpe_gate_fwd");
  return TATT_OK;
}

int tatt_rpe_gate_bwd(const float* gi, const float* gh, const float* bhh_f, const float* bhh_b, const float* hprev,
                      const float* dqp, float* carry, float* dgh, float* dgi_sum, int s, int B, long dir_stride_gh,
                      long dir_stride_h, hipStream_t st) {
  hipLaunchKernelGGL(rpe_gate_bwd, dim3(cdiv(2 * RW * RH, 256)), dim3(256), 0, st, gi, gh, bhh_f, bhh_b, hprev, dqp, carry, dgh,
                     dgi_sum, s, B, dir_stride_gh, dir_stride_h);
  TATT_LAUNCH_CHECK("rpe_gate_bwd");
  return TATT_OK;
}

int tatt_rpe_gather_x(const float* init_factor, float* x, hipStream_t st) {
  hipLaunchKernelGGL(rpe_gather_x, dim3(cdiv(RW * 1024, 256)), dim3(256), 0, st, init_factor, x);
  TATT_LAUNCH_CHECK("rpe_gather_x");
  return TATT_OK;
}
int tatt_rpe_scatter_dx(const float* dx, float* dinit, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(rpe_scatter_dx, dim3(cdiv(RW * 1024, 256)), dim3(256), 0, st, dx, dinit, accumulate);
  TATT_LAUNCH_CHECK("rpe_scatter_dx");
  return TATT_OK;
}

}  // extern "C"
