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
__global__ __launch_bounds__(256) void gru32_fwd(const float* __restrict__ gi, const float* __restrict__ whh_f,
                                                 const float* __restrict__ whh_b, const float* __restrict__ bhh_f,
                                                 const float* __restrict__ bhh_b, float* __restrict__ out,
                                                 float* __restrict__ save, int nseq, SeqMap sm) {
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
  float rA[PF], zA[PF], nA[PF], rB[PF], zB[PF], nB[PF], rC[PF], zC[PF], nC[PF];
  auto fetch = [&](int t0, float (&fr)[PF], float (&fz)[PF], float (&fn)[PF]) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int tc = min(t0 + u, T - 1);
      const float* q = gcol + (base + (long)(dir ? T - 1 - tc : tc) * sm.tstride) * 192;
      fr[u] = q[0];
      fz[u] = q[HID];
      fn[u] = q[2 * HID];
    }
  };
  auto steps = [&](int t0, const float (&c_r)[PF], const float (&c_z)[PF], const float (&c_n)[PF]) {
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
  fetch(0, rA, zA, nA);
  fetch(PF, rB, zB, nB);
  for (int t0 = 0; t0 < T; t0 += 3 * PF) {
    fetch(t0 + 2 * PF, rC, zC, nC);
    steps(t0, rA, zA, nA);
    fetch(t0 + 3 * PF, rA, zA, nA);
    steps(t0 + PF, rB, zB, nB);
    fetch(t0 + 4 * PF, rB, zB, nB);
    steps(t0 + 2 * PF, rC, zC, nC);
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
  hipLaunchKernelGGL(gru32_fwd, dim3(cdiv(nseq * 2, 8)), dim3(256), 0, st, gi, whh_f, whh_b, bhh_f, bhh_b, out, save, nseq, sm);
  TATT_LAUNCH_CHECK("gru32_fwd");
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
