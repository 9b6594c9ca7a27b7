// TP-interpreter decoder (model/transformer_v2.py:805-832 TransformerDecoderLayer_TP.forward_post x 2, :368-389 TransformerDecoder
// with its final LayerNorm on every intermediate, model/tsrn.py:220-224 mean over the two layers) as ONE forward launch and one
// backward launch per layer.  Every row (image token) goes through
//     q = (tgt + query_pos) Wq^T + bq -> softmax(q K^T / 4) (26 keys, 4 heads x 16) -> dropout -> P V -> out-proj
//     -> tgt + dropout -> LayerNorm2 -> Linear-ReLU-dropout-Linear -> + dropout -> LayerNorm3 -> decoder.norm
// independently of every other row (keys / values come from the 26 text tokens of its batch element), so a wavefront keeps 32
// tokens in registers from the first projection to the last LayerNorm and nothing but the tensors the backward pass needs touches HBM.
//
// Register layout ("T layout"): a [64 features x 32 tokens] matrix lives in 2 x 16 VGPRs exactly as v_mfma_f32_32x32x2_f32 leaves an
// accumulator: lane = token + 32 * hi, register (ob, i) = feature ob*32 + (i/4)*8 + hi*4 + i%4.  All products are computed transposed,
// Y^T = W X^T: the weights are the A operand (rows = output features), and the B operand of k-step (ob, i) -- B[k = hi][col = token] --
// IS register (ob, i) of the previous result, whatever the order the k index runs in, as long as the A operand agrees: lane l supplies
// W[row][ob*32 + (i/4)*8 + (l/32)*4 + i%4].  So chained products need no LDS round trip and no shuffles; A operands are streamed from
// global memory in that "fragment-major" order (tatt_dec_pack_w / tatt_dec_pack_kv: one coalesced 16-byte load per lane and four MFMAs).
// Attention stays in the same layout: scores^T[key][token] = (K/4) Q_h (A = K rows of head h, k = the head's 16 features = 8 MFMAs;
// 26 keys padded to 32), softmax over a lane's 16 registers + its partner lane l^32, context^T = V_h^T P (A = V^T with the rows of the
// block's other head zeroed, so both heads of a 32-feature block accumulate into one accumulator).
// fp32 throughout: v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain.
#include "common.h"

namespace {

constexpr int E = 64, S = 26, NH = 4;
constexpr int NSAVE = 7;                       // qin, Q, c, v2, t2, hd, v3
constexpr int KV_SLOTS = 48;                   // float4 slots per lane and batch element: KA 8, VA 16, VB 8, KBt 16
constexpr int KA0 = 0, VA0 = 8, VB0 = 24, KB0 = 32;

struct T32 {
  f32x16 b[2];
};

struct DecLayer {
  const float4* Wf;                            // [8][1024] float4: Wq, Wo, W1, W2, Wq^T, Wo^T, W1^T, W2^T in fragment-major order
  const float4* kv;                            // [B][48][64] float4
  const float* p[8];                           // bq, bo, b1, b2, norm2.weight, norm2.bias, norm3.weight, norm3.bias
  float* sv[NSAVE];
  uint32_t site[2][4];                         // per pass: tatt_seed_site words of attention, dropout2, FFN dropout, dropout3
};

struct DecFwdArgs {
  DecLayer L[2];
  const float* tgt;
  const float* qp;
  const float* nw;
  const float* nb;
  float* tp_map;
  int Lq;
  uint32_t pass_rows;                          // rows of one pass (B * Lq): rows >= pass_rows are the second pass of a two-pass (TSC) batch
  uint32_t thresh;
  float scale;
  const uint32_t* seed_dev;
  int wide;                                    // saved tensors leave through wave-private LDS tiles as full lines (TATT_DEC_WIDE=0: 32-byte pieces)
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void t_zero(T32& x) {
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) x.b[ob][i] = 0.f;
}
// p: the lane's row (or a 64-float parameter vector); hi = lane / 32
__device__ __forceinline__ void t_load(T32& x, const float* __restrict__ p, int hi) {
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(p + ob * 32 + g * 8 + hi * 4);
      x.b[ob][g * 4] = v.x; x.b[ob][g * 4 + 1] = v.y; x.b[ob][g * 4 + 2] = v.z; x.b[ob][g * 4 + 3] = v.w;
    }
}
__device__ __forceinline__ void t_store(const T32& x, float* __restrict__ p, int hi) {
#ifdef DEC_KO_STORE
  if (x.b[0][0] != 12345.678f) return;
#endif
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(p + ob * 32 + g * 8 + hi * 4) =
          make_float4(x.b[ob][g * 4], x.b[ob][g * 4 + 1], x.b[ob][g * 4 + 2], x.b[ob][g * 4 + 3]);
}
// The same store through a wave-private LDS tile: in the accumulator layout a lane holds runs of four channels of ITS token, so every
// wave store instruction above writes 32-byte pieces of 32 different rows (partial lines); the wave's 32 tokens are one contiguous
// 8 KB block, which leaves as eight instructions of 1 KB each after a transpose.  tile: 32 rows x TW floats (TW = 68: 16-byte rows, row
// starts four banks apart); blk = the block's first element (token 0 of the wave, channel 0).
constexpr int TW = 68;
__device__ __forceinline__ void t_store_wide(const T32& x, float* __restrict__ blk, float* __restrict__ tile, int lane) {
  const int r = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(tile + r * TW + ob * 32 + g * 8 + hi * 4) =
          make_float4(x.b[ob][g * 4], x.b[ob][g * 4 + 1], x.b[ob][g * 4 + 2], x.b[ob][g * 4 + 3]);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 v = *reinterpret_cast<const float4*>(tile + (i * 4 + (lane >> 4)) * TW + (lane & 15) * 4);
    *reinterpret_cast<float4*>(blk + (i * 64 + lane) * 4) = v;
  }
}
// acc += M x  (M [64][64] in fragment-major order: slot ((obr*2+obk)*4+g), lane -> M[obr*32 + l%32][obk*32 + g*8 + (l/32)*4 + 0..3])
__device__ __forceinline__ void mm64(T32& acc, const float4* __restrict__ Wf, int lane, const T32& x) {
#pragma unroll
  for (int obk = 0; obk < 2; ++obk)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#ifdef DEC_KO_WLOAD
      const float4 a0 = make_float4(lane * 0.01f, 0.1f, 0.2f, 0.3f), a1 = make_float4(0.1f, lane * 0.02f, 0.2f, 0.3f);
#else
      const float4 a0 = Wf[((0 * 2 + obk) * 4 + g) * 64 + lane];
      const float4 a1 = Wf[((1 * 2 + obk) * 4 + g) * 64 + lane];
#endif
      acc.b[0] = mfma(a0.x, x.b[obk][g * 4 + 0], acc.b[0]);
      acc.b[1] = mfma(a1.x, x.b[obk][g * 4 + 0], acc.b[1]);
      acc.b[0] = mfma(a0.y, x.b[obk][g * 4 + 1], acc.b[0]);
      acc.b[1] = mfma(a1.y, x.b[obk][g * 4 + 1], acc.b[1]);
      acc.b[0] = mfma(a0.z, x.b[obk][g * 4 + 2], acc.b[0]);
      acc.b[1] = mfma(a1.z, x.b[obk][g * 4 + 2], acc.b[1]);
      acc.b[0] = mfma(a0.w, x.b[obk][g * 4 + 3], acc.b[0]);
      acc.b[1] = mfma(a1.w, x.b[obk][g * 4 + 3], acc.b[1]);
    }
}
// lane l <-> lane l ^ 32 (the two halves of a token's features) on v_permlane32_swap (VALU; no trip through the LDS crossbar)
__device__ __forceinline__ float pair_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pair_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ void ln_stats(const T32& v, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v.b[ob][i];
  mean = pair_sum(s) * (1.f / 64.f);
  float q = 0.f;
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float d = v.b[ob][i] - mean;
      q += d * d;
    }
  rstd = rsqrtf(pair_sum(q) * (1.f / 64.f) + 1e-5f);
}
// y = (v - mean) * rstd * gamma + beta
__device__ __forceinline__ void ln_apply(T32& y, const T32& v, float mean, float rstd, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, int hi) {
  T32 ga, be;
  t_load(ga, gamma, hi);
  t_load(be, beta, hi);
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) y.b[ob][i] = (v.b[ob][i] - mean) * rstd * ga.b[ob][i] + be.b[ob][i];
}
// element index of register (ob, i) in a [rows][64] tensor
__device__ __forceinline__ uint32_t eidx(uint32_t row, int ob, int i, int hi) {
  return row * 64u + (uint32_t)(ob * 32 + (i >> 2) * 8 + hi * 4 + (i & 3));
}
// dropout mask of an active site (thresh > 0)
__device__ __forceinline__ float dmask(uint32_t ss, uint32_t idx, uint32_t thresh, float scale) {
  return ((tatt_hash_u32(ss, idx) >> 8) >= thresh) ? scale : 0.0f;
}
__device__ __forceinline__ int key_of(int i, int hi) { return (i >> 2) * 8 + hi * 4 + (i & 3); }

// Softmax probabilities of head h for the lane's token (un-dropped), from the projected query Q (T layout): p[i] <-> key_of(i, hi)
__device__ __forceinline__ void head_probs(f32x16& p, const T32& Q, const float4* __restrict__ kvb, int h, int lane, int hi) {
  const int ob = h >> 1, hh = h & 1;
  f32x16 s;
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float4 a = kvb[(KA0 + h * 2 + g) * 64 + lane];
    s = mfma(a.x, Q.b[ob][hh * 8 + g * 4 + 0], s);
    s = mfma(a.y, Q.b[ob][hh * 8 + g * 4 + 1], s);
    s = mfma(a.z, Q.b[ob][hh * 8 + g * 4 + 2], s);
    s = mfma(a.w, Q.b[ob][hh * 8 + g * 4 + 3], s);
  }
  // keys >= 26: registers 14, 15 of every lane and 12, 13 of the hi lanes
  float mx = s[0];
#pragma unroll
  for (int i = 1; i < 12; ++i) mx = fmaxf(mx, s[i]);
  if (hi == 0) mx = fmaxf(mx, fmaxf(s[12], s[13]));
  mx = pair_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    p[i] = __expf(s[i] - mx);
    sum += p[i];
  }
  p[12] = hi == 0 ? __expf(s[12] - mx) : 0.f;
  p[13] = hi == 0 ? __expf(s[13] - mx) : 0.f;
  p[14] = 0.f;
  p[15] = 0.f;
  sum += p[12] + p[13];
  const float inv = 1.f / pair_sum(sum);
#pragma unroll
  for (int i = 0; i < 14; ++i) p[i] *= inv;
}

// ================================================================================================================================
__global__ __launch_bounds__(256, 2) void dec_fwd_kernel(DecFwdArgs A) {
  const int lane = threadIdx.x & 63, hi = lane >> 5, wv = threadIdx.x >> 6;
  const uint32_t grow = (uint32_t)(blockIdx.x * 128 + wv * 32 + (lane & 31));   // row of the batch: addresses, key / value table
  const int ps = grow >= A.pass_rows;                                           // pass (workgroup-uniform: pass_rows % 128 == 0)
  const uint32_t row = grow - (ps ? A.pass_rows : 0u);                          // row within its pass: dropout indices, query_pos
  const int b = (int)(row / (uint32_t)A.Lq), q = (int)(row - (uint32_t)b * (uint32_t)A.Lq), gb = (int)(grow / (uint32_t)A.Lq);
  const long ro = (long)grow * E, rq = (long)row * E;
  const uint32_t sd = A.seed_dev ? A.seed_dev[0] * 0x85EBCA77u : 0u;
  __shared__ __attribute__((aligned(16))) float wtile[4][32 * TW];
  const long wro = (long)(blockIdx.x * 128 + wv * 32) * E;   // the wave's 32 tokens: one contiguous block of every [R,64] tensor
  auto save = [&](const T32& x, float* base) {
    if (A.wide) t_store_wide(x, base + wro, wtile[wv], lane);
    else t_store(x, base + ro, hi);
  };
  T32 tgt, tp;
  t_load(tgt, A.tgt + ro, hi);
#pragma unroll
  for (int li = 0; li < 2; ++li) {
    const DecLayer& L = A.L[li];
    const float4* __restrict__ kvb = L.kv + (long)gb * KV_SLOTS * 64;
    T32 x, y;
    // ---- qin = tgt + query_pos ; Q = Wq qin + bq
    t_load(x, A.qp + rq, hi);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) x.b[ob][i] += tgt.b[ob][i];
    save(x, L.sv[0]);
    t_load(y, L.p[0], hi);
    mm64(y, L.Wf + 0 * 1024, lane, x);
    save(y, L.sv[1]);
    // ---- attention: x = context
    t_zero(x);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      f32x16 p;
      head_probs(p, y, kvb, h, lane, hi);
      if (A.thresh) {
        const uint32_t ib = (uint32_t)((((long)b * NH + h) * A.Lq + q) * S), ss = L.site[ps][0] + sd;
#pragma unroll
        for (int i = 0; i < 14; ++i) p[i] *= dmask(ss, ib + (uint32_t)key_of(i, hi), A.thresh, A.scale);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 a = kvb[(VA0 + h * 4 + g) * 64 + lane];
        x.b[h >> 1] = mfma(a.x, p[g * 4 + 0], x.b[h >> 1]);
        x.b[h >> 1] = mfma(a.y, p[g * 4 + 1], x.b[h >> 1]);
        x.b[h >> 1] = mfma(a.z, p[g * 4 + 2], x.b[h >> 1]);
        x.b[h >> 1] = mfma(a.w, p[g * 4 + 3], x.b[h >> 1]);
      }
    }
    save(x, L.sv[2]);
    // ---- out-proj, residual + dropout, LayerNorm2
    t_load(y, L.p[1], hi);
    mm64(y, L.Wf + 1 * 1024, lane, x);
    if (A.thresh) {
      const uint32_t ss = L.site[ps][1] + sd;
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int i = 0; i < 16; ++i) y.b[ob][i] *= dmask(ss, eidx(row, ob, i, hi), A.thresh, A.scale);
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) y.b[ob][i] += tgt.b[ob][i];
    save(y, L.sv[3]);
    float mean, rstd;
    ln_stats(y, mean, rstd);
    ln_apply(tgt, y, mean, rstd, L.p[4], L.p[5], hi);          // tgt <- t2
    save(tgt, L.sv[4]);
    // ---- FFN
    t_load(x, L.p[2], hi);
    mm64(x, L.Wf + 2 * 1024, lane, tgt);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) x.b[ob][i] = fmaxf(x.b[ob][i], 0.f);
    if (A.thresh) {
      const uint32_t ss = L.site[ps][2] + sd;
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int i = 0; i < 16; ++i) x.b[ob][i] *= dmask(ss, eidx(row, ob, i, hi), A.thresh, A.scale);
    }
    save(x, L.sv[5]);
    t_load(y, L.p[3], hi);
    mm64(y, L.Wf + 3 * 1024, lane, x);
    if (A.thresh) {
      const uint32_t ss = L.site[ps][3] + sd;
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int i = 0; i < 16; ++i) y.b[ob][i] *= dmask(ss, eidx(row, ob, i, hi), A.thresh, A.scale);
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) y.b[ob][i] += tgt.b[ob][i];
    save(y, L.sv[6]);
    ln_stats(y, mean, rstd);
    ln_apply(tgt, y, mean, rstd, L.p[6], L.p[7], hi);          // tgt <- t3 = the layer's output
    // ---- decoder.norm on the intermediate, mean over the two layers
    ln_stats(tgt, mean, rstd);
    ln_apply(y, tgt, mean, rstd, A.nw, A.nb, hi);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) tp.b[ob][i] = li == 0 ? 0.5f * y.b[ob][i] : tp.b[ob][i] + 0.5f * y.b[ob][i];
  }
  save(tp, A.tp_map);
}

// ================================================================================================================================
// Backward of one layer.  The row-local chain runs in T layout like the forward pass (products with the transposed weights); the
// gradients that contract over tokens are produced as follows:
//   * dY operands of the four weight gradients (dQ, d attn-out, d hidden, d ffn-out) are stored; tatt_tn_small_group sums dY^T X;
//   * dK / dV of the attention: per wave, P (dropped) / dS and the context gradient / Q are transposed through a wave-private LDS
//     tile and multiplied on the matrix core, out[key][feature] over the wave's 32 tokens; the four waves' partials are summed
//     through LDS and stored once per workgroup in the layout attn_bwd_final sums ([chunk][2][B*26][64], chunk = 128 tokens);
//   * LayerNorm weight / bias gradients: per-lane products reduced over the 32 token lanes by recursive halving (31 exchanges per
//     64-vector), summed over the workgroup's waves through LDS; dec_ln_final sums the per-workgroup rows.
struct DecBwdArgs {
  DecLayer L;
  const float* dt3_in;     // gradient arriving at the layer's output from the layer above (NULL: none)
  const float* dtp;        // gradient of tp_map [R0][64]
  const float* nw;         // decoder.norm.weight
  float* dtgt;             // gradient wrt the layer's input
  int dtgt_acc;
  float* dqp;              // gradient wrt query_pos
  int dqp_acc;
  float* dY[4];            // dQ, da, dh, df
  float* kvpart;           // [Lq/128][2][B*26][64]: one partial per workgroup (128 tokens)
  float* lnpart;           // [R0/128][6][64]: d norm2.weight, d norm2.bias, d norm3.weight, d norm3.bias, d decoder.norm.weight, .bias
  int Lq, B;               // B: batch elements of the whole launch (both passes)
  uint32_t pass_rows;
  uint32_t thresh;
  float scale;
  const uint32_t* seed_dev;
};

// Sum of each of the 32 registers of v over the 32 lanes of this lane's half: lane r (= l % 32) ends up with the total of register r.
__device__ __forceinline__ float lanes32_reduce(const T32& v, int lane) {
  float r16[16], r8[8], r4[4], r2[2];
  {
    const bool up = lane & 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float keep = up ? v.b[1][j] : v.b[0][j], send = up ? v.b[0][j] : v.b[1][j];
      r16[j] = keep + __shfl_xor(send, 16, 64);
    }
  }
  {
    const bool up = lane & 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float keep = up ? r16[j + 8] : r16[j], send = up ? r16[j] : r16[j + 8];
      r8[j] = keep + __shfl_xor(send, 8, 64);
    }
  }
  {
    const bool up = lane & 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float keep = up ? r8[j + 4] : r8[j], send = up ? r8[j] : r8[j + 4];
      r4[j] = keep + __shfl_xor(send, 4, 64);
    }
  }
  {
    const bool up = lane & 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float keep = up ? r4[j + 2] : r4[j], send = up ? r4[j] : r4[j + 2];
      r2[j] = keep + __shfl_xor(send, 2, 64);
    }
  }
  const bool up = lane & 1;
  const float keep = up ? r2[1] : r2[0], send = up ? r2[0] : r2[1];
  return keep + __shfl_xor(send, 1, 64);
}
// feature of register r = l % 32 for a lane of half hi
__device__ __forceinline__ int feat_of_reg(int r, int hi) { return (r >> 4) * 32 + ((r & 15) >> 2) * 8 + hi * 4 + (r & 3); }

// LayerNorm backward in T layout: xh = normalised input, g = dL/dy.  dx = rstd * (g*gamma - mean(g*gamma) - xh * mean(g*gamma*xh));
// the per-lane gradient products of gamma / beta are reduced over the tokens and stored to part[0..63] / part[64..127].
__device__ __forceinline__ void ln_bwd_t(T32& dx, const T32& xh, const T32& g, float rstd, const float* __restrict__ gamma,
                                         float* __restrict__ part, int lane, int hi) {
  T32 ga, t;
  t_load(ga, gamma, hi);
  float p1 = 0.f, p2 = 0.f;
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float gg = g.b[ob][i] * ga.b[ob][i];
      t.b[ob][i] = g.b[ob][i] * xh.b[ob][i];
      p1 += gg;
      p2 += gg * xh.b[ob][i];
      dx.b[ob][i] = gg;
    }
  const float m1 = pair_sum(p1) * (1.f / 64.f), m2 = pair_sum(p2) * (1.f / 64.f);
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) dx.b[ob][i] = rstd * (dx.b[ob][i] - m1 - xh.b[ob][i] * m2);
  const int f = feat_of_reg(lane & 31, hi);
  part[f] = lanes32_reduce(t, lane);
  part[64 + f] = lanes32_reduce(g, lane);
}

constexpr int TS = 36;                               // row stride of the wave-private transpose tiles (floats): 16-byte rows, conflict-free

// out[key][feature of block] += sum over the wave's 32 tokens of  A^T[key][tok] * Bm[tok][feature]:
// a (T layout registers: key_of(i, hi) x token) is written to tile tA, the B tile tB ([tok][32 features]) is already there;
// head mask: only the 16 features of head half hh contribute.
__device__ __forceinline__ void tok_contract(f32x16& acc, const f32x16& a, float* tA, const float* tB, int hh, int lane, int hi) {
  const int t = lane & 31;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(tA + t * TS + g * 8 + hi * 4) = make_float4(a[g * 4], a[g * 4 + 1], a[g * 4 + 2], a[g * 4 + 3]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const bool on = ((t >> 4) == hh);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float av = tA[(2 * s + hi) * TS + t];
    const float bv = tB[(2 * s + hi) * TS + t];
    acc = mfma(av, on ? bv : 0.f, acc);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// write one 32-feature block of a T-layout tensor as a [tok][32] tile
__device__ __forceinline__ void tile_put(float* tB, const f32x16& x, int lane, int hi) {
  const int t = lane & 31;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(tB + t * TS + g * 8 + hi * 4) = make_float4(x[g * 4], x[g * 4 + 1], x[g * 4 + 2], x[g * 4 + 3]);
}

__global__ __launch_bounds__(256, 2) void dec_bwd_kernel(DecBwdArgs A) {
  __shared__ __attribute__((aligned(16))) float tiles[4][3][32 * TS];   // 13,824 floats; reused at the end for the 4 x 3,328 dK / dV partials
  __shared__ float lnred[4][6 * 64];
  const int lane = threadIdx.x & 63, hi = lane >> 5, wv = threadIdx.x >> 6;
  const uint32_t grow = (uint32_t)(blockIdx.x * 128 + wv * 32 + (lane & 31));
  const int ps = grow >= A.pass_rows;
  const uint32_t row = grow - (ps ? A.pass_rows : 0u);     // row within its pass (dropout indices)
  const int b = (int)(row / (uint32_t)A.Lq), q = (int)(row - (uint32_t)b * (uint32_t)A.Lq), gb = (int)(grow / (uint32_t)A.Lq);
  const long ro = (long)grow * E;
  const uint32_t sd = A.seed_dev ? A.seed_dev[0] * 0x85EBCA77u : 0u;
  const DecLayer& L = A.L;
  const float4* __restrict__ kvb = L.kv + (long)gb * KV_SLOTS * 64;
  float* lnp = lnred[wv];
  float* tA = tiles[wv][0];
  float* tB1 = tiles[wv][1];
  float* tB2 = tiles[wv][2];

  T32 x, xh, g, d;
  float mean, rstd, mean3, rstd3;
  // ---- t3 = LayerNorm3(v3); decoder.norm backward on t3
  t_load(x, L.sv[6] + ro, hi);
  ln_stats(x, mean3, rstd3);
  {
    T32 t3;
    ln_apply(t3, x, mean3, rstd3, L.p[6], L.p[7], hi);
    ln_stats(t3, mean, rstd);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) xh.b[ob][i] = (t3.b[ob][i] - mean) * rstd;
  }
  t_load(g, A.dtp + ro, hi);
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) g.b[ob][i] *= 0.5f;
  ln_bwd_t(d, xh, g, rstd, A.nw, lnp + 4 * 64, lane, hi);          // d = dL/dt3 through the final norm
  if (A.dt3_in) {
    t_load(g, A.dt3_in + ro, hi);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) d.b[ob][i] += g.b[ob][i];
  }
  // ---- LayerNorm3 backward: x = v3
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) xh.b[ob][i] = (x.b[ob][i] - mean3) * rstd3;
  T32 dt2;
  ln_bwd_t(dt2, xh, d, rstd3, L.p[6], lnp + 2 * 64, lane, hi);     // dt2 = dv3 (residual branch)
  // ---- FFN backward
  g = dt2;
  if (A.thresh) {
    const uint32_t ss = L.site[ps][3] + sd;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) g.b[ob][i] *= dmask(ss, eidx(row, ob, i, hi), A.thresh, A.scale);
  }
  t_store(g, A.dY[3] + ro, hi);                                    // df
  t_zero(d);
  mm64(d, L.Wf + 7 * 1024, lane, g);                               // d hidden(dropped) = W2^T df
  t_load(x, L.sv[5] + ro, hi);                                     // hd = relu(.) * mask: positive exactly where both let the gradient through
  {
    const float sc = A.thresh ? A.scale : 1.f;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) d.b[ob][i] = x.b[ob][i] > 0.f ? d.b[ob][i] * sc : 0.f;
  }
  t_store(d, A.dY[2] + ro, hi);                                    // dh
  mm64(dt2, L.Wf + 6 * 1024, lane, d);                             // dt2 += W1^T dh
  // ---- LayerNorm2 backward
  t_load(x, L.sv[3] + ro, hi);
  ln_stats(x, mean, rstd);
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) xh.b[ob][i] = (x.b[ob][i] - mean) * rstd;
  T32 dtgt;
  ln_bwd_t(dtgt, xh, dt2, rstd, L.p[4], lnp, lane, hi);            // dtgt = dv2 (residual branch)
  g = dtgt;
  if (A.thresh) {
    const uint32_t ss = L.site[ps][1] + sd;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) g.b[ob][i] *= dmask(ss, eidx(row, ob, i, hi), A.thresh, A.scale);
  }
  t_store(g, A.dY[1] + ro, hi);                                    // da
  t_zero(d);
  mm64(d, L.Wf + 5 * 1024, lane, g);                               // d = d context = Wo^T da
  // ---- attention backward
  t_load(x, L.sv[1] + ro, hi);                                     // Q
  T32 dQ, dKa, dVa;
  t_zero(dQ);
  t_zero(dKa);
  t_zero(dVa);
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int ob = h >> 1, hh = h & 1;
    if (hh == 0) {                                                 // the block's [tok][32] tiles: context gradient and Q
      tile_put(tB1, d.b[ob], lane, hi);
      tile_put(tB2, x.b[ob], lane, hi);
    }
    f32x16 p, dp, m;
    head_probs(p, x, kvb, h, lane, hi);
#pragma unroll
    for (int i = 0; i < 16; ++i) dp[i] = 0.f;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const float4 a = kvb[(VB0 + h * 2 + gq) * 64 + lane];
      dp = mfma(a.x, d.b[ob][hh * 8 + gq * 4 + 0], dp);
      dp = mfma(a.y, d.b[ob][hh * 8 + gq * 4 + 1], dp);
      dp = mfma(a.z, d.b[ob][hh * 8 + gq * 4 + 2], dp);
      dp = mfma(a.w, d.b[ob][hh * 8 + gq * 4 + 3], dp);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = i < 14 ? 1.f : 0.f;
    if (A.thresh) {
      const uint32_t ib = (uint32_t)((((long)b * NH + h) * A.Lq + q) * S), ss = L.site[ps][0] + sd;
#pragma unroll
      for (int i = 0; i < 14; ++i) m[i] = dmask(ss, ib + (uint32_t)key_of(i, hi), A.thresh, A.scale);
    }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      dp[i] *= m[i];                                               // dL/dP (softmax output before dropout); p = 0 on padded keys
      dot += p[i] * dp[i];
    }
    dot = pair_sum(dot);
    f32x16 ds;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      ds[i] = p[i] * (dp[i] - dot);
      m[i] *= p[i];                                                // dropped probabilities (what multiplied V)
    }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 a = kvb[(KB0 + h * 4 + gq) * 64 + lane];
      dQ.b[ob] = mfma(a.x, ds[gq * 4 + 0], dQ.b[ob]);
      dQ.b[ob] = mfma(a.y, ds[gq * 4 + 1], dQ.b[ob]);
      dQ.b[ob] = mfma(a.z, ds[gq * 4 + 2], dQ.b[ob]);
      dQ.b[ob] = mfma(a.w, ds[gq * 4 + 3], dQ.b[ob]);
    }
    tok_contract(dVa.b[ob], m, tA, tB1, hh, lane, hi);             // dV[key][f] += P_dropped^T d context
    tok_contract(dKa.b[ob], ds, tA, tB2, hh, lane, hi);            // dK[key][f] += dS^T Q   (x 1/4 below)
  }
  t_store(dQ, A.dY[0] + ro, hi);
  // ---- query projection backward
  t_zero(g);
  mm64(g, L.Wf + 4 * 1024, lane, dQ);                              // dqin = Wq^T dQ
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) dtgt.b[ob][i] += g.b[ob][i];
  if (A.dtgt_acc) {
    t_load(x, A.dtgt + ro, hi);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) dtgt.b[ob][i] += x.b[ob][i];
  }
  t_store(dtgt, A.dtgt + ro, hi);
  if (A.dqp_acc) {
    t_load(x, A.dqp + ro, hi);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) g.b[ob][i] += x.b[ob][i];
  }
  t_store(g, A.dqp + ro, hi);
  // ---- the workgroup's partials: dK / dV (registers <-> keys, lane % 32 <-> feature of the block) and the LayerNorm rows of its four
  // waves are summed through LDS (the transpose tiles are free now) and stored once per workgroup
  __syncthreads();
  {
    float* red = &tiles[0][0][0] + wv * (2 * S * E);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = key_of(i, hi);
        if (key < S) {
          red[key * E + ob * 32 + (lane & 31)] = dVa.b[ob][i];
          red[(S + key) * E + ob * 32 + (lane & 31)] = 0.25f * dKa.b[ob][i];
        }
      }
  }
  __syncthreads();
  {
    const float* red = &tiles[0][0][0];
    const int chunk = q >> 7;                                      // workgroup-uniform: its 128 tokens share b and q / 128
    const long n = (long)A.B * S * E;
    float* pv = A.kvpart + ((long)chunk * 2) * n + (long)gb * S * E;
    for (int idx = threadIdx.x; idx < 2 * S * E; idx += 256) {
      const float v = (red[idx] + red[2 * S * E + idx]) + (red[4 * S * E + idx] + red[6 * S * E + idx]);
      const int which = idx >= S * E;
      pv[(long)which * n + (idx - which * S * E)] = v;
    }
    float* lo = A.lnpart + (long)blockIdx.x * 6 * 64;
    for (int idx = threadIdx.x; idx < 6 * 64; idx += 256)
      lo[idx] = (lnred[0][idx] + lnred[1][idx]) + (lnred[2][idx] + lnred[3][idx]);
  }
}

// LayerNorm parameter gradients: column sums of the per-wave partial rows.  part [nblk][nvec][64]; out[v] (+)= sum over blk.
struct LnFinalArgs {
  float* out[6];
};
__global__ __launch_bounds__(1024) void dec_ln_final(const float* __restrict__ part, int nblk, int nvec, LnFinalArgs o) {
  __shared__ float red[64][17];
  const int v = blockIdx.y, cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  const float* p = part + v * 64 + c;
  const long st = (long)nvec * 64;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = rl;
  for (; b + 192 < nblk; b += 256) {
    s0 += p[(long)b * st];
    s1 += p[(long)(b + 64) * st];
    s2 += p[(long)(b + 128) * st];
    s3 += p[(long)(b + 192) * st];
  }
  for (; b < nblk; b += 64) s0 += p[(long)b * st];
  red[rl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0) {
    float t = 0.f;
    for (int j = 0; j < 64; ++j) t += red[j][cl];
    o.out[v][c] += t;
  }
}

// ---- operand packs ---------------------------------------------------------------------------------------------------------
// W row-major [64][64] -> fragment-major (see mm64); transposed = 1: the fragments of W^T.  grid (4, 16 matrices), block 256
struct PackWArgs {
  const float* W[8];                           // layer 0: Wq, Wo, W1, W2; layer 1: the same
};
__global__ void dec_pack_w_kernel(PackWArgs a, float4* __restrict__ out) {
  const int mat = blockIdx.y;                  // layer * 8 + (transposed * 4 + which)
  const int layer = mat >> 3, tr = (mat >> 2) & 1, which = mat & 3;
  const float* __restrict__ W = a.W[layer * 4 + which];
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int obr = slot >> 3, obk = (slot >> 2) & 1, g = slot & 3;
  const int r = obr * 32 + (lane & 31), k = obk * 32 + g * 8 + (lane >> 5) * 4;
  float4 v;
  if (!tr) {
    v = *reinterpret_cast<const float4*>(W + r * 64 + k);
  } else {
    v = make_float4(W[k * 64 + r], W[(k + 1) * 64 + r], W[(k + 2) * 64 + r], W[(k + 3) * 64 + r]);
  }
  out[(long)mat * 1024 + slot * 64 + lane] = v;
}
// projected keys / values [B*26][64] -> the A operands of the attention products of one layer; grid (B), block 256
__global__ void dec_pack_kv_kernel(const float* __restrict__ Kp, const float* __restrict__ Vp, float4* __restrict__ out) {
  const int b = blockIdx.x;
  const float* __restrict__ K = Kp + (long)b * S * E;
  const float* __restrict__ V = Vp + (long)b * S * E;
  float4* o = out + (long)b * KV_SLOTS * 64;
  for (int idx = threadIdx.x; idx < KV_SLOTS * 64; idx += blockDim.x) {
    const int slot = idx >> 6, lane = idx & 63, r = lane & 31, hi = lane >> 5;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (slot < VA0 || (slot >= VB0 && slot < KB0)) {
      // rows = keys, k = the head's 16 features: KA (keys / 4, for the scores) and VB (values, for dL/dP)
      const bool isK = slot < VA0;
      const int s2 = isK ? slot - KA0 : slot - VB0, h = s2 >> 1, g = s2 & 1;
      if (r < S) {
        const float* src = (isK ? K : V) + r * E + h * 16 + g * 8 + hi * 4;
        const float sc = isK ? 0.25f : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = src[e] * sc;
      }
    } else {
      // rows = the 32 features of block h / 2 (those of the other head zeroed), k = keys: VA (context) and KBt (dQ, keys / 4)
      const bool isV = slot < VB0;
      const int s2 = isV ? slot - VA0 : slot - KB0, h = s2 >> 2, g = s2 & 3;
      if ((r >> 4) == (h & 1)) {
        const float* src = (isV ? V : K) + (h >> 1) * 32 + r;
        const float sc = isV ? 1.f : 0.25f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = g * 8 + hi * 4 + e;
          if (key < S) v[e] = src[key * E] * sc;
        }
      }
    }
    o[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

}  // namespace

static const int g_dec_wide = [] { const char* e = getenv("TATT_DEC_WIDE"); return (e && e[0] == '0') ? 0 : 1; }();

extern "C" {

size_t tatt_dec_packed_w_floats(void) { return (size_t)16 * 4096; }
size_t tatt_dec_kv_table_floats(int B) { return (size_t)B * KV_SLOTS * 64 * 4; }
size_t tatt_dec_bwd_kvpart_floats(int B, int Lq) { return (size_t)(Lq / 128) * 2 * B * S * E; }
size_t tatt_dec_bwd_lnpart_floats(int B, int Lq) { return (size_t)B * (Lq / 128) * 6 * 64; }

int tatt_dec_pack_w(const float* const* W, float* packed, hipStream_t st) {
  TATT_CHECK_ARG(W && packed && (((uintptr_t)packed) & 15) == 0, "tatt_dec_pack_w: bad arguments");
  PackWArgs a;
  for (int i = 0; i < 8; ++i) {
    TATT_CHECK_ARG(W[i] && (((uintptr_t)W[i]) & 15) == 0, "tatt_dec_pack_w: weight %d missing or not 16-byte aligned", i);
    a.W[i] = W[i];
  }
  hipLaunchKernelGGL(dec_pack_w_kernel, dim3(4, 16), dim3(256), 0, st, a, reinterpret_cast<float4*>(packed));
  TATT_LAUNCH_CHECK("dec_pack_w");
  return TATT_OK;
}

int tatt_dec_pack_kv(const float* Kp, const float* Vp, float* table, int B, hipStream_t st) {
  TATT_CHECK_ARG(Kp && Vp && table && B > 0 && (((uintptr_t)table) & 15) == 0, "tatt_dec_pack_kv: bad arguments");
  hipLaunchKernelGGL(dec_pack_kv_kernel, dim3(B), dim3(256), 0, st, Kp, Vp, reinterpret_cast<float4*>(table));
  TATT_LAUNCH_CHECK("dec_pack_kv");
  return TATT_OK;
}

static int fill_layer(DecLayer& L, const float* packed_w, int li, const float* kv, const float* const* params, float* const* saved,
                      const uint32_t* sites, const uint32_t* sites2, const char* who) {
  L.Wf = reinterpret_cast<const float4*>(packed_w) + (long)li * 8 * 1024;
  L.kv = reinterpret_cast<const float4*>(kv);
  TATT_CHECK_ARG(kv && (((uintptr_t)kv) & 15) == 0, "%s: layer %d key/value table missing or unaligned", who, li);
  for (int i = 0; i < 8; ++i) {
    TATT_CHECK_ARG(params[i] && (((uintptr_t)params[i]) & 15) == 0, "%s: layer %d parameter %d missing or not 16-byte aligned", who, li, i);
    L.p[i] = params[i];
  }
  for (int i = 0; i < NSAVE; ++i) {
    TATT_CHECK_ARG(saved[i] && (((uintptr_t)saved[i]) & 15) == 0, "%s: layer %d saved tensor %d missing or not 16-byte aligned", who, li, i);
    L.sv[i] = saved[i];
  }
  for (int i = 0; i < 4; ++i) {
    L.site[0][i] = sites[i];
    L.site[1][i] = sites2 ? sites2[i] : sites[i];
  }
  return TATT_OK;
}

int tatt_dec_fwd(const float* tgt, const float* qp, const float* packed_w, const float* const* kv, const float* const* params,
                 const float* norm_w, const float* norm_b, float* const* saved, float* tp_map, int B, int Lq, int npass,
                 const uint32_t* seed_sites, uint32_t thresh, float scale, const uint32_t* seed_dev, hipStream_t st) {
  TATT_CHECK_ARG(tgt && qp && packed_w && kv && params && norm_w && norm_b && saved && tp_map && seed_sites, "tatt_dec_fwd: NULL argument");
  TATT_CHECK_ARG(B > 0 && Lq > 0 && Lq % 128 == 0 && (long)B * Lq * E < (1L << 32), "tatt_dec_fwd: Lq must be a multiple of 128 and B*Lq*64 < 2^32");
  TATT_CHECK_ARG((npass == 1 || npass == 2) && B % npass == 0, "tatt_dec_fwd: npass must be 1 or 2 and divide B");
  TATT_CHECK_ARG(((((uintptr_t)tgt) | ((uintptr_t)qp) | ((uintptr_t)tp_map) | ((uintptr_t)packed_w) | ((uintptr_t)norm_w) | ((uintptr_t)norm_b)) & 15) == 0,
                 "tatt_dec_fwd: pointers must be 16-byte aligned");
  DecFwdArgs a;
  for (int li = 0; li < 2; ++li) {
    int rc = fill_layer(a.L[li], packed_w, li, kv[li], params + li * 8, saved + li * NSAVE, seed_sites + li * 4,
                        npass == 2 ? seed_sites + 8 + li * 4 : nullptr, "tatt_dec_fwd");
    if (rc) return rc;
  }
  a.tgt = tgt; a.qp = qp; a.nw = norm_w; a.nb = norm_b; a.tp_map = tp_map; a.Lq = Lq;
  a.pass_rows = (uint32_t)((long)(B / npass) * Lq);
  a.thresh = thresh; a.scale = scale; a.seed_dev = seed_dev;
  a.wide = g_dec_wide;
  hipLaunchKernelGGL(dec_fwd_kernel, dim3((unsigned)((long)B * Lq / 128)), dim3(256), 0, st, a);
  TATT_LAUNCH_CHECK("dec_fwd");
  return TATT_OK;
}

int tatt_dec_bwd(int layer, const float* dt3_in, const float* dtp, const float* packed_w, const float* kv, const float* const* params,
                 const float* norm_w, float* const* saved, float* dtgt, int dtgt_accumulate, float* dqp, int dqp_accumulate,
                 float* const* dY, float* kvpart, float* lnpart, int B, int Lq, int npass, const uint32_t* seed_sites, uint32_t thresh,
                 float scale, const uint32_t* seed_dev, hipStream_t st) {
  TATT_CHECK_ARG((layer == 0 || layer == 1) && dtp && packed_w && params && norm_w && saved && dtgt && dqp && dY && kvpart && lnpart && seed_sites,
                 "tatt_dec_bwd: bad arguments");
  TATT_CHECK_ARG(B > 0 && Lq > 0 && Lq % 128 == 0 && (long)B * Lq * E < (1L << 32), "tatt_dec_bwd: Lq must be a multiple of 128 and B*Lq*64 < 2^32");
  TATT_CHECK_ARG((npass == 1 || npass == 2) && B % npass == 0, "tatt_dec_bwd: npass must be 1 or 2 and divide B");
  TATT_CHECK_ARG(((((uintptr_t)dt3_in) | ((uintptr_t)dtp) | ((uintptr_t)dtgt) | ((uintptr_t)dqp) | ((uintptr_t)norm_w)) & 15) == 0,
                 "tatt_dec_bwd: pointers must be 16-byte aligned");
  DecBwdArgs a;
  int rc = fill_layer(a.L, packed_w, layer, kv, params, saved, seed_sites, npass == 2 ? seed_sites + 4 : nullptr, "tatt_dec_bwd");
  if (rc) return rc;
  for (int i = 0; i < 4; ++i) {
    TATT_CHECK_ARG(dY[i] && (((uintptr_t)dY[i]) & 15) == 0, "tatt_dec_bwd: dY[%d] missing or not 16-byte aligned", i);
    a.dY[i] = dY[i];
  }
  a.dt3_in = dt3_in; a.dtp = dtp; a.nw = norm_w; a.dtgt = dtgt; a.dtgt_acc = dtgt_accumulate; a.dqp = dqp; a.dqp_acc = dqp_accumulate;
  a.kvpart = kvpart; a.lnpart = lnpart; a.Lq = Lq; a.B = B; a.pass_rows = (uint32_t)((long)(B / npass) * Lq); a.thresh = thresh; a.scale = scale; a.seed_dev = seed_dev;
  hipLaunchKernelGGL(dec_bwd_kernel, dim3((unsigned)((long)B * Lq / 128)), dim3(256), 0, st, a);
  TATT_LAUNCH_CHECK("dec_bwd");
  return TATT_OK;
}

// grads[6]: d norm2.weight, d norm2.bias, d norm3.weight, d norm3.bias, d decoder.norm.weight, d decoder.norm.bias (each += )
int tatt_dec_ln_final(const float* lnpart, int B, int Lq, float* const* grads, hipStream_t st) {
  TATT_CHECK_ARG(lnpart && grads && B > 0 && Lq % 128 == 0, "tatt_dec_ln_final: bad arguments");
  LnFinalArgs o;
  for (int i = 0; i < 6; ++i) {
    TATT_CHECK_ARG(grads[i], "tatt_dec_ln_final: gradient %d missing", i);
    o.out[i] = grads[i];
  }
  hipLaunchKernelGGL(dec_ln_final, dim3(4, 6), dim3(1024), 0, st, lnpart, B * (Lq / 128), 6, o);
  TATT_LAUNCH_CHECK("dec_ln_final");
  return TATT_OK;
}

}  // extern "C"
