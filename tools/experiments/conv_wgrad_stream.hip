// EXPERIMENT, NOT BUILT INTO libtatt_hip.so (kept for the next round).
// Measured on MI355X (B=64, 3x3 64->64, 256 pixel ranges): 66 us per call incl. the slab reduction, against 75 us for the tiled
// split-K GEMM form that the library uses -- but the graph-replayed step did not move (9.97-10.00 ms either way), so it was
// left out.  Lessons that cost time: (1) a buffer descriptor whose base depends on threadIdx (wave = tid >> 6) is treated as
// divergent and every load gets a readfirstlane "waterfall" loop -- take the wave index through readfirstlane; (2) a negative
// scalar offset breaks the raw-buffer range check (valid lanes came back as zero) -- fold shifts into the descriptor base;
// (3) `a && b ? off : OOB` became divergent branches with a load in each arm -- build the offset with masks.
//
// Weight gradient of the 3x3, 64 -> 64 convolutions of the SRBs as a streaming reduction over the pixels:
//     dW[co][ci][ty][tx] += sum_p dY[p][co] * X[p + (ty-1, tx-1)][ci]
// The tiled GEMM form (gemm_kernel<OP_IM2COL_M, OP_ROWCONTIG>) stages both operands through LDS in row-contiguous images and
// needs 150-way split-K to fill the GPU; here nothing is staged.  For v_mfma_f32_32x32x2_f32 lane (i = l&31, h = l>>5) supplies
// A[i][k=h] and B[k=h][i]; with NHWC operands that is "lanes 0-31 read 32 consecutive channels of pixel p, lanes 32-63 of pixel
// p+1": coalesced 128-byte loads straight into the MFMA operands (the scheme of tn_small.hip), the tap only shifts the pixel
// row that X is read from.  One wave owns one (tap, 32-wide co half): 2 accumulators, per pixel pair 2 X loads + 1 dY load feed
// 2 MFMAs; the three taps of one kernel row x two co halves make a 6-wave workgroup (blockIdx.y = kernel row), so the waves of a
// workgroup re-read each other's lines from L1.  No LDS, no barriers.  Every workgroup streams its own pixel range and writes one
// partial slab; a second kernel sums the slabs in a fixed order (deterministic, no atomics) into OIHW.
#include "common.h"

namespace {

constexpr int OOB = (int)0x80000000;  // buffer offset beyond num_records: the load returns 0
constexpr int CIN = 64, COUT = 64, T9 = 9;

__global__ __launch_bounds__(384) void conv3x3_wgrad_stream(const float* __restrict__ X, long ldx, const float* __restrict__ dY, long lddy,
                                                            float* __restrict__ slabs, int H, int W, int R, int rows_per_blk) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: the X descriptor below depends on it
  const int li = lane & 31, lh = lane >> 5;
  const int dxi = wave >> 1, nt = wave & 1;  // tap column 0..2, co half
  const int dy = (int)blockIdx.y - 1, dx = dxi - 1;
  const int rb = blockIdx.x * rows_per_blk, re = min(R, rb + rows_per_blk);  // rb and rows_per_blk are even, W is even
  f32x16 acc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  // the tap's pixel shift goes into the descriptor base (it may point ahead of the tensor: lanes whose tap pixel does not
  // exist are switched off by their voffset and never form an address), so that the scalar offset below is never negative
  const int shift = dy * W + dx;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (long)shift * ldx), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, OOB, 0x00020000);
  // per-lane byte offsets (column + the lane half's pixel); the pixel position is the wave-uniform soffset
  const int cx0 = (int)((lh * ldx + li) * 4);
  const int cy = (int)((lh * lddy + nt * 32 + li) * 4);
  const int sx = (int)ldx * 4, sy = (int)lddy * 4;
  // (x, y) of the next pixel pair to load; a pair never straddles an image row (x even, W even)
  int px = rb % W, py = (rb / W) % H;
  constexpr int U = 8;
  float a0[U][2], b0[U], a1[U][2], b1[U];
  auto load = [&](int r, float (&av)[U][2], float (&bv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rw = r + 2 * u;
      // masks, not conditions: with `a && b ? x : y` the compiler builds divergent branches around the loads
      const int rokm = -(int)(rw + lh < re);
      const int okm = rokm & -((int)((unsigned)(px + lh + dx) < (unsigned)W) & (int)((unsigned)(py + dy) < (unsigned)H));
      const int vx = (cx0 & okm) | (OOB & ~okm), vy = (cy & rokm) | (OOB & ~rokm);
      const int so = rw * sx;
      av[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, vx, so, 0));
      av[u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, vx + 32 * 4, so, 0));  // OOB + 128 is still out of range
      bv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rY, vy, rw * sy, 0));
      px += 2;
      if (px >= W) {
        px -= W;
        py = py + 1 == H ? 0 : py + 1;
      }
    }
  };
  auto mma = [&](const float (&av)[U][2], const float (&bv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][0], bv[u], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][1], bv[u], acc[1], 0, 0, 0);
    }
  };
  // two register sets: the loads of the next set are in flight while the current one is multiplied; straight-line, rows past
  // `re` are predicated to zero (no early exit: the s_waitcnt bookkeeping stays exact)
  load(rb, a0, b0);
  for (int r = rb; r < re; r += 4 * U) {
    load(r + 2 * U, a1, b1);
    mma(a0, b0);
    load(r + 4 * U, a0, b0);
    mma(a1, b1);
  }
  // slab [(tap, ci)][co]: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int tap = ((int)blockIdx.y) * 3 + dxi;
  float* slab = slabs + (long)blockIdx.x * (T9 * CIN * COUT) + (long)tap * CIN * COUT + nt * 32 + li;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) slab[(a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * COUT] = acc[a][e];
}

// dW (OIHW) (+)= sum over slabs; 16 output quads x 16 slab lanes per workgroup (a thread per output would walk nslab dependent
// loads: latency bound)
__global__ __launch_bounds__(256) void conv3x3_wgrad_stream_reduce(const float* __restrict__ slabs, int nslab, float* __restrict__ dW,
                                                                   int accumulate) {
  __shared__ float4 red[256];
  const int ql = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const int q = blockIdx.x * 16 + ql;  // quad of consecutive co for one (tap, ci)
  constexpr int PER = T9 * CIN * COUT;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = zl; z < nslab; z += 16) {
    const float4 v = *reinterpret_cast<const float4*>(slabs + (long)z * PER + q * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  red[zl * 16 + ql] = s;
  __syncthreads();
  if (zl != 0) return;
  for (int j = 1; j < 16; ++j) {
    const float4 v = red[j * 16 + ql];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int m = (q * 4) / COUT, co = (q * 4) % COUT;  // m = tap * CIN + ci
  const int tap = m / CIN, ci = m % CIN;
  const float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float* d = dW + ((long)(co + e) * CIN + ci) * T9 + tap;
    *d = accumulate ? *d + o[e] : o[e];
  }
}

}  // namespace

// Called by tatt_conv_wgrad for 3x3, Cin = Cout = 64, even W and even pixel count (fp32 mode): nslab pixel ranges, workspace
// of nslab * 9*64*64 floats (what tatt_conv_wgrad_workspace(64, 64, 3, 3, nslab) returns).
int tatt_conv3x3_wgrad_stream(const float* X, const float* dY, float* dW, int B, int H, int W, long ldx, long lddy, int nslab,
                              float* workspace, int accumulate, hipStream_t st) {
  const int R = B * H * W;
  TATT_CHECK_ARG(nslab >= 1 && workspace && W % 2 == 0 && ((long)R + W + 2) * ldx * 4 < (1L << 31) && ((long)R + 2) * lddy * 4 < (1L << 31),
                 "tatt_conv_wgrad (streaming 3x3): even W and operands below 2 GiB required");
  int rpb = (cdiv(R, nslab) + 1) & ~1;
  if (rpb < 16) rpb = 16;
  const int nblk = cdiv(R, rpb);
  hipLaunchKernelGGL(conv3x3_wgrad_stream, dim3(nblk, 3), dim3(384), 0, st, X, ldx, dY, lddy, workspace, H, W, R, rpb);
  TATT_LAUNCH_CHECK("conv3x3_wgrad_stream");
  hipLaunchKernelGGL(conv3x3_wgrad_stream_reduce, dim3(T9 * CIN * COUT / 4 / 16), dim3(256), 0, st, workspace, nblk, dW, accumulate);
  TATT_LAUNCH_CHECK("conv3x3_wgrad_stream_reduce");
  return TATT_OK;
}
