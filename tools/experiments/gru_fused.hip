// EXPERIMENT, not part of libtatt_hip.so (round 3; measured and left out).  GruBlock forward with the input projection fused into the
// recurrence launch.  Parity-green against tatt_gemm_cat + tatt_gru32_fwd (out and saved gates to 5e-6) and against an fp64 GRU, but
// not faster at B = 64 on MI355X (tools/experiments/micro_gru_fused.py): vertical (T = 16, Cin = 128) 84 us against 42 + 35 for the
// two launches, horizontal (T = 64, Cin = 64) 55 us against 28 + 32; the step 7.21 ms against 7.15.  Why: the recurrence is a latency
// chain of ~0.5 us per step that neither form shortens (horizontal: 2048 half-wave chains on 1024 SIMDs, one wave each, 64 steps =
// 32 us whatever feeds it), and here the workgroup's four waves do the matrix phase and the recurrence phase one after the other,
// whereas the GEMM launch runs at full-chip occupancy.  What would pay is a shorter step (h hand-over without the LDS round trip),
// not less traffic.
// GruBlock forward (model/tsrn.py:1082-1092: Conv2d 1x1 -> nn.GRU(64, 32, bidirectional)) as ONE launch: the input projection
//     gi = Wf x + bf        (Wf = W_ih W_c [192][Cin], the 1x1 conv folded into the GRU's input weights: tatt_gru_fold_w)
// is computed on the matrix core for 16 time steps of four sequences at a time and handed to the recurrence through LDS, so the
// [R,192] tensor of input projections (50 MB at B = 64, written by a GEMM and read back by tatt_gru32_fwd) never exists in HBM.
//   * projection: transposed product gi^T = Wf x^T with the activations as the B operand in the accumulator layout of
//     v_mfma_f32_32x32x2_f32 (lane = pixel, 16-byte loads of four consecutive features; see dec_fused.hip) and Wf as the A operand in
//     fragment-major order (tatt_gru_pack_wf).  Wave w computes the 96 gate pre-activations of direction w / 2 for 32 pixels (two
//     sequences x 16 steps; the reverse direction takes its 16 steps from the far end of the sequence) = 48 or 96 MFMAs.
//   * recurrence: as tatt_gru32_fwd -- one half-wave per (sequence, direction), lane j owns hidden unit j with its three W_hh rows in
//     registers, h broadcast through LDS -- with gi read from the LDS tile instead of global memory.
// The concatenation cat(conv features, tp_map) of the first GruBlock (model/tsrn.py:893-895) is two source pointers (Cin = 128).
#include "common.h"

namespace {

constexpr int HID = 32, TC = 16, SEQS = 4;
constexpr int TST = 100;                                 // floats per pixel slot of the gi tile (96 + pad: conflict-free 16-byte stores)
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct SeqMap {  // pixel index of (sequence s, step t) = (s / inner) * outer_stride + (s % inner) * inner_stride + t * tstride
  int inner;
  long outer_stride, inner_stride, tstride;
  int T;
};
__device__ __forceinline__ long seq_base(const SeqMap& m, int s) {
  const int so = s / m.inner, si = s - so * m.inner;
  return (long)so * m.outer_stride + (long)si * m.inner_stride;
}
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// Wf [192][Cin] row-major -> fragment-major: slot ((rb * KB + kb) * 4 + g), lane -> Wf[rb*32 + l%32][kb*32 + g*8 + (l/32)*4 + 0..3]
struct PackGroup {
  const float* Wf[16];
  float* out[16];
  int Cin[16];
};
__global__ __launch_bounds__(256) void gru_pack_wf_k(PackGroup p) {
  const int b = blockIdx.y, Cin = p.Cin[b], KB = Cin / 32;
  const int idx = blockIdx.x * 256 + threadIdx.x;        // float4 index
  if (idx >= 192 * Cin / 4) return;
  const int lane = idx & 63, slot = idx >> 6, g = slot & 3, kb = (slot >> 2) % KB, rb = (slot >> 2) / KB;
  const float4 v = *reinterpret_cast<const float4*>(p.Wf[b] + (long)(rb * 32 + (lane & 31)) * Cin + kb * 32 + g * 8 + (lane >> 5) * 4);
  reinterpret_cast<float4*>(p.out[b])[idx] = v;
}

template <int NIN>
__global__ __launch_bounds__(256, 2) void gru32_proj_fwd_k(const float* __restrict__ x1, const float* __restrict__ x2,
                                                        const float4* __restrict__ Wp, const float* __restrict__ bf,
                                                        const float* __restrict__ whh_f, const float* __restrict__ whh_b,
                                                        const float* __restrict__ bhh_f, const float* __restrict__ bhh_b,
                                                        float* __restrict__ out, float* __restrict__ save, SeqMap sm) {
  constexpr int KB = 2 * NIN;
  __shared__ __attribute__((aligned(16))) float tile[2][SEQS * TC][TST];
  __shared__ __attribute__((aligned(16))) float hbuf[8][HID];
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, wv = tid >> 6;
  const int T = sm.T;
  // ---- recurrence role: half-wave (sequence sl, direction dir), lane j
  const int j = tid & 31, slot8 = tid >> 5, sl = slot8 >> 1, dir = slot8 & 1;
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
  const long base = seq_base(sm, blockIdx.x * SEQS + sl);
  float h = 0.f;
  hbuf[slot8][j] = 0.f;
  // ---- projection role: wave wv -> direction pdir, sequences 2*psp and 2*psp+1; lane -> (sequence, step of the chunk)
  const int pdir = wv >> 1, psp = wv & 1, lp = lane & 31;
  const long pbase = seq_base(sm, blockIdx.x * SEQS + 2 * psp + (lp >> 4));
  const int pu = lp & 15;
  for (int c = 0; c < T / TC; ++c) {
    // ---- gi of this chunk: 32 pixels x 96 gate units per wave
    {
      const int t = pdir ? T - 1 - (c * TC + pu) : c * TC + pu;
      const long pix = pbase + (long)t * sm.tstride;
      f32x16 xr[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const float* src = (kb < 2 ? x1 : x2) + pix * 64 + (kb & 1) * 32 + hi * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *reinterpret_cast<const float4*>(src + g * 8);
          xr[kb][g * 4] = v.x; xr[kb][g * 4 + 1] = v.y; xr[kb][g * 4 + 2] = v.z; xr[kb][g * 4 + 3] = v.w;
        }
      }
      // (pointers laundered per chunk: the fragment / bias loads are invariant across chunks, and hoisted out of the loop they are
      //  spilled to scratch -- 96 to 192 registers that the recurrence's W_hh rows need)
      const float4* Wq = Wp;
      const float* bq = bf;
      asm volatile("" : "+s"(Wq), "+s"(bq));
      f32x16 acc[3];                                       // start from the bias
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *reinterpret_cast<const float4*>(bq + (3 * pdir + r) * 32 + g * 8 + hi * 4);
          acc[r][g * 4] = v.x; acc[r][g * 4 + 1] = v.y; acc[r][g * 4 + 2] = v.z; acc[r][g * 4 + 3] = v.w;
        }
      // A fragments one group (four k-steps x three row blocks) ahead of their MFMAs; the scheduling barriers keep the compiler
      // from hoisting all 48 / 96 fragment registers to the top (it then spills the recurrence's W_hh rows)
      float4 a[3], an[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) a[r] = Wq[((3 * pdir + r) * KB * 4) * 64 + lane];
#pragma unroll
      for (int q = 0; q < KB * 4; ++q) {
        const int kb = q >> 2, g = q & 3;
        if (q + 1 < KB * 4) {
#pragma unroll
          for (int r = 0; r < 3; ++r) an[r] = Wq[((3 * pdir + r) * KB * 4 + q + 1) * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[r] = mfma(a[r].x, xr[kb][g * 4 + 0], acc[r]);
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[r] = mfma(a[r].y, xr[kb][g * 4 + 1], acc[r]);
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[r] = mfma(a[r].z, xr[kb][g * 4 + 2], acc[r]);
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[r] = mfma(a[r].w, xr[kb][g * 4 + 3], acc[r]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 3; ++r) a[r] = an[r];
      }
      float* dst = &tile[pdir][psp * 32 + lp][hi * 4];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(dst + r * 32 + g * 8) = make_float4(acc[r][g * 4], acc[r][g * 4 + 1], acc[r][g * 4 + 2], acc[r][g * 4 + 3]);
    }
    __syncthreads();
    // ---- 16 steps of the recurrence
#pragma unroll 4
    for (int u = 0; u < TC; ++u) {
      const int t = dir ? T - 1 - (c * TC + u) : c * TC + u;
      const long pix = base + (long)t * sm.tstride;
      const float* gq = &tile[dir][sl * TC + u][j];
      const float g_r = gq[0], g_z = gq[HID], g_n = gq[2 * HID];
      __builtin_amdgcn_wave_barrier();
      f32x2 ar2 = {br, 0.f}, az2 = {bz, 0.f}, an2 = {bn, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < HID / 4; ++k4) {
        const float4 hv = *reinterpret_cast<const float4*>(&hbuf[slot8][k4 * 4]);
        const f32x2 h0 = {hv.x, hv.y}, h1 = {hv.z, hv.w};
        ar2 = __builtin_elementwise_fma(wr[2 * k4], h0, ar2);
        az2 = __builtin_elementwise_fma(wz[2 * k4], h0, az2);
        an2 = __builtin_elementwise_fma(wn[2 * k4], h0, an2);
        ar2 = __builtin_elementwise_fma(wr[2 * k4 + 1], h1, ar2);
        az2 = __builtin_elementwise_fma(wz[2 * k4 + 1], h1, az2);
        an2 = __builtin_elementwise_fma(wn[2 * k4 + 1], h1, an2);
      }
      const float ar = ar2.x + ar2.y, az = az2.x + az2.y, an = an2.x + an2.y;
      const float r = sigmoidf_(g_r + ar), z = sigmoidf_(g_z + az);
      const float n = tanhf_(g_n + r * an);
      h = (1.f - z) * n + z * h;
      __builtin_amdgcn_wave_barrier();
      hbuf[slot8][j] = h;
      out[pix * 64 + dir * HID + j] = h;
      if (save) {
        float* sp = save + (pix * 2 + dir) * 4 * HID;
        sp[j] = r;
        sp[HID + j] = z;
        sp[2 * HID + j] = n;
        sp[3 * HID + j] = an;
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" {

// n <= 16 folded input weights Wf[i] [192][Cin[i]] (tatt_gru_fold_w) -> out[i] (same size) in MFMA-fragment order.  HOST arrays.
int tatt_gru_pack_wf(int n, const float* const* Wf, float* const* out, const int* Cin, hipStream_t st) {
  TATT_CHECK_ARG(n > 0 && n <= 16 && Wf && out && Cin, "tatt_gru_pack_wf: 1..16 blocks, non-null arrays");
  PackGroup p;
  memset(&p, 0, sizeof(p));
  int mx = 0;
  for (int i = 0; i < n; ++i) {
    TATT_CHECK_ARG(Wf[i] && out[i] && (Cin[i] == 64 || Cin[i] == 128) && ((((uintptr_t)Wf[i]) | ((uintptr_t)out[i])) & 15) == 0,
                   "tatt_gru_pack_wf: block %d: Cin must be 64 or 128, pointers 16-byte aligned", i);
    p.Wf[i] = Wf[i]; p.out[i] = out[i]; p.Cin[i] = Cin[i];
    if (Cin[i] > mx) mx = Cin[i];
  }
  hipLaunchKernelGGL(gru_pack_wf_k, dim3(cdiv(192 * mx / 4, 256), n), dim3(256), 0, st, p);
  TATT_LAUNCH_CHECK("gru_pack_wf");
  return TATT_OK;
}

// out [P,64], save [P,2,4,32] (or NULL) as tatt_gru32_fwd; x1 (and x2 for Cin = 128) [P,64] are the block's input(s), Wp / bf the packed
// folded input projection.  nseq % 4 == 0, T % 16 == 0.
int tatt_gru32_proj_fwd(const float* x1, const float* x2, const float* Wp, const float* bf, const float* whh_f, const float* whh_b,
                        const float* bhh_f, const float* bhh_b, float* out, float* save, int nseq, int T, int inner, long outer_stride,
                        long inner_stride, long tstride, hipStream_t st) {
  TATT_CHECK_ARG(x1 && Wp && bf && whh_f && whh_b && bhh_f && bhh_b && out, "tatt_gru32_proj_fwd: NULL argument");
  TATT_CHECK_ARG(nseq > 0 && nseq % SEQS == 0 && T > 0 && T % TC == 0 && inner > 0, "tatt_gru32_proj_fwd: nseq %% 4 == 0 and T %% 16 == 0 required (nseq %d, T %d)", nseq, T);
  TATT_CHECK_ARG(((((uintptr_t)x1) | ((uintptr_t)x2) | ((uintptr_t)Wp) | ((uintptr_t)bf)) & 15) == 0, "tatt_gru32_proj_fwd: x1, x2, Wp, bf must be 16-byte aligned");
  SeqMap sm{inner, outer_stride, inner_stride, tstride, T};
  if (x2)
    hipLaunchKernelGGL(gru32_proj_fwd_k<2>, dim3(nseq / SEQS), dim3(256), 0, st, x1, x2, reinterpret_cast<const float4*>(Wp), bf, whh_f, whh_b,
                       bhh_f, bhh_b, out, save, sm);
  else
    hipLaunchKernelGGL(gru32_proj_fwd_k<1>, dim3(nseq / SEQS), dim3(256), 0, st, x1, x2, reinterpret_cast<const float4*>(Wp), bf, whh_f, whh_b,
                       bhh_f, bhh_b, out, save, sm);
  TATT_LAUNCH_CHECK("gru32_proj_fwd");
  return TATT_OK;
}

}  // extern "C"
