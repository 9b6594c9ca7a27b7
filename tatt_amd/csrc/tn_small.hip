// dW[M,N] (+)= dY[R,M]^T * X[R,N]  and  db[M] (+)= column sums of dY, for the many weight gradients whose output is
// tiny (M <= 96, N <= 64) and whose reduction runs over all R = B*H*W rows: an HBM-bound streaming reduction.
// No LDS staging: for v_mfma_f32_32x32x2_f32 lane (i = l&31, h = l>>5) supplies A[i][k=h] and B[k=h][i], which for row-major
// [R, C] operands is exactly "lanes 0-31 read 32 consecutive floats of row r, lanes 32-63 of row r+1": coalesced 128-B
// loads straight into the MFMA operands.  Each wave streams its own row range (deep unroll keeps loads in flight), the
// eight waves of a block are summed by a three-level tree through LDS, every block (at most one per CU, so at most 256
// slabs to write and re-read) writes one partial slab, a second kernel sums the slabs in a fixed order (deterministic;
// no atomics).  The bias gradient rides along on the already loaded dY fragments.
#include "common.h"
#include <algorithm>

namespace {

constexpr int TN_WAVES = 8, TN_MAXBLK = 256;
constexpr int OOB = (int)0x80000000;  // buffer offset beyond num_records: the load returns 0

template <int MT, int NT>
__device__ __forceinline__ void tn_small_body(const float* __restrict__ dY, long lddy, const float* __restrict__ dY2, long lddy2, int msplit,
                                              const float* __restrict__ X, long ldx, int M, int N, int R, int rows_per_blk,
                                              float* __restrict__ slabs, int blk, float* red) {
  constexpr int SLOT = MT * 32 * NT * 32 + MT * 32;  // one wave's partial: the MxN tile + the bias column sums
  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform by construction; said so to the compiler: a VGPR-derived soffset wraps every buffer load in a readfirstlane
  // waterfall loop (a basic block per load: nothing is batched or overlapped with the MFMAs)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int rb = blk * rows_per_blk, re = min(R, rb + rows_per_blk);
  const int rows_per_wave = ((rows_per_blk + 2 * TN_WAVES - 1) / (2 * TN_WAVES)) * 2;  // even
  const int r0 = rb + wave * rows_per_wave, r1 = min(re, r0 + rows_per_wave);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  float asum[MT];
  bool mok[MT], nok[NT];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    asum[a] = 0.f;
    mok[a] = a * 32 + li < M;
  }
#pragma unroll
  for (int b = 0; b < NT; ++b) nok[b] = b * 32 + li < N;

  // Two register sets of U k-pairs each: the loads of the next set are issued before the MFMAs of the current one, so the
  // HBM latency and the matrix work (0.8 GFLOP for a [96,64] output, as long as the 42 MB stream itself) overlap inside a
  // wave instead of adding up.  (Knock-outs on the six-problem GruBlock group, 50.9 us: without the MFMAs 40.9 us, without the
  // loads 45.0 us -- the launch runs at the pace of its matrix instructions at the clock the chip sustains.)  Straight-line, two sets per trip.  (A ring of four sets, loads three sets ahead, measured no
  // faster: the kernel is not latency-bound once the loop is branch-free.)
  constexpr int U = 4;
  float av0[U][MT], bv0[U][NT], av1[U][MT], bv1[U][NT];
  // Branch-free predication: raw buffer loads whose descriptor ends at this wave's last row (num_records = r1 rows), the whole
  // offset in the VGPR operand -- rows past r1 and columns past M / N (offset pushed beyond 2 GiB) come back as 0 with no
  // compare, select or branch in the loop.  (The scalar offset operand is not range-checked, hence the VALU add per load.)
  // Columns m >= msplit of dY come from a second matrix (dY2, column m - msplit): the W_hh gradient of a GRU direction reads its
  // r / z gate columns from dgi and only the n gate from its own [R,32] block.  Descriptor, row stride and lane offset are chosen
  // per 32-column tile before the loop (block-uniform selects on scalar registers).
  const int sx = (int)ldx * 4;
  const int r1u = __builtin_amdgcn_readfirstlane(r1 > 0 ? r1 : 0);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, r1u * sx, 0x00020000);
  __amdgpu_buffer_rsrc_t rYa[MT];
  int sya[MT], cy[MT], cx[NT];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const bool second = a * 32 >= msplit;
    const long ld = second ? lddy2 : lddy;
    sya[a] = (int)ld * 4;
    rYa[a] = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? dY2 : dY), 0, r1u * sya[a], 0x00020000);
    cy[a] = mok[a] ? (int)((lh * ld + a * 32 - (second ? msplit : 0) + li) * 4) : OOB;
  }
#pragma unroll
  for (int b = 0; b < NT; ++b) cx[b] = nok[b] ? (int)((lh * ldx + b * 32 + li) * 4) : OOB;
  auto load = [&](int r, float (&av)[U][MT], float (&bv)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rw = r + 2 * u;
#pragma unroll
      for (int a = 0; a < MT; ++a) av[u][a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rYa[a], cy[a] + rw * sya[a], 0, 0));
#pragma unroll
      for (int b = 0; b < NT; ++b) bv[u][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, cx[b] + rw * sx, 0, 0));
    }
  };
  auto mma = [&](const float (&av)[U][MT], const float (&bv)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        asum[a] += av[u][a];
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
      }
    }
  };
  load(r0, av0, bv0);
  for (int r = r0; r < r1; r += 4 * U) {
    load(r + 2 * U, av1, bv1);
    mma(av0, bv0);
    load(r + 4 * U, av0, bv0);
    mma(av1, bv1);
  }
  // ---- tree-sum the eight waves through LDS: waves [h, 2h) park their partial in slot (wave - h), waves [0, h) add it ----
  constexpr int NW = NT * 32;
#pragma unroll
  for (int a = 0; a < MT; ++a) asum[a] += __shfl_xor(asum[a], 32, 64);
#pragma unroll
  for (int h = TN_WAVES / 2; h >= 1; h >>= 1) {
    if (wave >= h && wave < 2 * h) {
      float* slot = red + (wave - h) * SLOT;
#pragma unroll
      for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) slot[(a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * NW + b * 32 + li] = acc[a][b][e];
        if (lh == 0) slot[MT * 32 * NW + a * 32 + li] = asum[a];
      }
    }
    __syncthreads();
    if (wave < h) {
      const float* slot = red + wave * SLOT;
#pragma unroll
      for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][b][e] += slot[(a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * NW + b * 32 + li];
        asum[a] += slot[MT * 32 * NW + a * 32 + li];
      }
    }
    __syncthreads();
  }
  // wave 0 holds the block's sum: write the slab straight from its registers
  if (wave == 0) {
    float* slab = slabs + (long)blk * (M * N + M);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, n = b * 32 + li;
          if (m < M && n < N) slab[m * N + n] = acc[a][b][e];
        }
      if (lh == 0 && a * 32 + li < M) slab[M * N + a * 32 + li] = asum[a];
    }
  }
}

template <int MT, int NT>
__global__ __launch_bounds__(TN_WAVES * 64) void tn_small_kernel(const float* __restrict__ dY, long lddy, const float* __restrict__ X, long ldx,
                                                       int M, int N, int R, int rows_per_blk, float* __restrict__ slabs) {
  __shared__ float red[(TN_WAVES / 2) * (MT * 32 * NT * 32 + MT * 32)];
  tn_small_body<MT, NT>(dY, lddy, dY, lddy, 0x7fffffff, X, ldx, M, N, R, rows_per_blk, slabs, blockIdx.x, red);
}

// Grouped form: up to TN_GROUP independent problems in one launch (blockIdx.y = problem), e.g. the four weight gradients of a
// BiGRU block.  Every problem gets gridDim.x blocks (fewer, longer row ranges and fewer slabs than separate launches).
constexpr int TN_GROUP = 8;
struct TnGroup {
  const float* dY[TN_GROUP];
  const float* dY2[TN_GROUP];
  long lddy2[TN_GROUP];
  int msplit[TN_GROUP];
  const float* X[TN_GROUP];
  float* dW[TN_GROUP];
  float* db[TN_GROUP];
  long lddy[TN_GROUP], ldx[TN_GROUP], lddw[TN_GROUP], slab_off[TN_GROUP];
  int M[TN_GROUP], N[TN_GROUP], R[TN_GROUP], rpb[TN_GROUP], nblk[TN_GROUP];
  int first[TN_GROUP + 1];  // problem p owns work items [first[p], first[p + 1])
  // dispatch index -> (problem << 8 | row block): the problems of a group read the same tensors (a GruBlock's six share dgi, the
  // block's inputs and h_prev between them), and workgroups are dealt to the eight XCDs round-robin, each XCD with an L2 of its
  // own.  The host sorts the work items by the position of their row range and gives every XCD a contiguous run of them, so the
  // items that read the same rows run on ONE XCD at the same time and every row crosses the fabric once instead of once per
  // problem (whole-step FETCH_SIZE table, profiles/r03_step_traffic.md).
  unsigned short wg[TN_MAXBLK];
};

__global__ __launch_bounds__(TN_WAVES * 64) void tn_small_group_kernel(TnGroup g, float* __restrict__ slabs) {
  __shared__ float red[(TN_WAVES / 2) * (3 * 32 * 2 * 32 + 3 * 32)];
  // The grid is exactly the blocks that have work, problem after problem: at most one per CU, and the hardware's round-robin
  // of consecutive blocks over the eight XCDs then gives every XCD at most 32 of them.  (A [nb_max, nprob] grid whose surplus
  // blocks exit at once put up to 39 working blocks on one XCD: two rounds, 89 us instead of 56.)
  const int item = g.wg[blockIdx.x];
  const int p = item >> 8, blk = item & 255;
  // block-uniform dispatch on the problem's tile shape (a per-MFMA "skip this tile" branch inside one shape split the loop into
  // a basic block per MFMA: nothing overlapped)
  const int mt = (g.M[p] + 31) >> 5, nt = (g.N[p] + 31) >> 5;
#define TN_BODY(a, b) tn_small_body<a, b>(g.dY[p], g.lddy[p], g.dY2[p], g.lddy2[p], g.msplit[p], g.X[p], g.ldx[p], g.M[p], g.N[p], g.R[p], g.rpb[p], slabs + g.slab_off[p], blk, red)
  if (nt == 1) {
    if (mt == 1) TN_BODY(1, 1);
    else if (mt == 2) TN_BODY(2, 1);
    else TN_BODY(3, 1);
  } else {
    if (mt == 1) TN_BODY(1, 2);
    else if (mt == 2) TN_BODY(2, 2);
    else TN_BODY(3, 2);
  }
#undef TN_BODY
}

__global__ void tn_small_group_reduce(TnGroup g, const float* __restrict__ slabs) {
  __shared__ float red[256];
  const int p = blockIdx.y, M = g.M[p], N = g.N[p];
  const int per = M * N + M;
  if ((int)blockIdx.x * 16 >= per) return;
  const int i = blockIdx.x * 16 + (threadIdx.x & 15);
  const float s = block16_colsum(slabs + g.slab_off[p], g.nblk[p], (long)per, i, i < per, red);
  if (threadIdx.x >= 16 || i >= per) return;
  if (i < M * N) {
    g.dW[p][(long)(i / N) * g.lddw[p] + (i % N)] += s;
  } else if (g.db[p]) {
    g.db[p][i - M * N] += s;
  }
}

// dW[m*lddw + n] += sum_blk slab[blk][m*N+n] ; db[m] += sum_blk slab[blk][M*N+m]
__global__ void tn_small_reduce(const float* __restrict__ slabs, int nblk, int M, int N, float* __restrict__ dW, long lddw,
                                float* __restrict__ db) {
  // 16 outputs x 16 slab lanes per block (a thread per output would serialise nblk dependent L2 round trips)
  __shared__ float red[256];
  const int per = M * N + M;
  const int i = blockIdx.x * 16 + (threadIdx.x & 15);
  const float s = block16_colsum(slabs, nblk, (long)per, i, i < per, red);
  if (threadIdx.x >= 16 || i >= per) return;
  if (i < M * N) {
    dW[(long)(i / N) * lddw + (i % N)] += s;
  } else if (db) {
    db[i - M * N] += s;
  }
}

}  // namespace

extern "C" {

size_t tatt_tn_small_workspace(int M, int N) { return (size_t)TN_MAXBLK * ((size_t)M * N + M); }

// dW[M,N] (row stride lddw) += dY[R,M]^T X[R,N];  db[M] += colsum(dY) (db may be null).  M <= 96, N <= 64.
int tatt_tn_small(const float* dY, long lddy, const float* X, long ldx, float* dW, long lddw, float* db, int M, int N, int R,
                  float* workspace, hipStream_t st) {
  TATT_CHECK_ARG(M > 0 && N > 0 && R > 0 && M <= 96 && N <= 64 && workspace, "tatt_tn_small: need 0 < M <= 96, 0 < N <= 64 (got %d, %d)", M, N);
  TATT_CHECK_ARG(((long)R + 1) * lddy * 4 < (1L << 31) && ((long)R + 1) * ldx * 4 < (1L << 31) && lddy > 0 && ldx > 0,
                 "tatt_tn_small: operands must span less than 2 GiB (32-bit buffer offsets)");
  int nblk = (R + 255) / 256;
  if (nblk > TN_MAXBLK) nblk = TN_MAXBLK;  // one 8-wave block per CU: enough loads in flight to stream near the HBM rate
  int rpb = ((R + nblk - 1) / nblk + 15) / 16 * 16;
  nblk = (R + rpb - 1) / rpb;
  const int MT = (M + 31) / 32, NT = (N + 31) / 32;
#define TN_LAUNCH(a, b) hipLaunchKernelGGL((tn_small_kernel<a, b>), dim3(nblk), dim3(TN_WAVES * 64), 0, st, dY, lddy, X, ldx, M, N, R, rpb, workspace)
  if (MT == 1 && NT == 1) TN_LAUNCH(1, 1);
  else if (MT == 1 && NT == 2) TN_LAUNCH(1, 2);
  else if (MT == 2 && NT == 1) TN_LAUNCH(2, 1);
  else if (MT == 2 && NT == 2) TN_LAUNCH(2, 2);
  else if (MT == 3 && NT == 1) TN_LAUNCH(3, 1);
  else TN_LAUNCH(3, 2);
#undef TN_LAUNCH
  TATT_LAUNCH_CHECK("tn_small_kernel");
  const int per = M * N + M;
  hipLaunchKernelGGL(tn_small_reduce, dim3(cdiv(per, 16)), dim3(256), 0, st, workspace, nblk, M, N, dW, lddw, db);
  TATT_LAUNCH_CHECK("tn_small_reduce");
  return TATT_OK;
}

// Grouped form of tatt_tn_small: nprob <= 8 independent problems (host arrays of length nprob) in one launch pair.
// workspace: tatt_tn_small_workspace(96, 64) floats cover any group.  dY2 / lddy2 / msplit (all three null, or arrays of length
// nprob; dY2[p] null = none): columns m >= msplit[p] (a multiple of 32) of problem p's dY are read from dY2[p] at column m - msplit[p].
int tatt_tn_small_group(int nprob, const float* const* dY, const long* lddy, const float* const* X, const long* ldx, float* const* dW,
                        const long* lddw, float* const* db, const int* M, const int* N, const int* R, const float* const* dY2,
                        const long* lddy2, const int* msplit, float* workspace, hipStream_t st) {
  TATT_CHECK_ARG((dY2 && lddy2 && msplit) || (!dY2 && !lddy2 && !msplit), "tatt_tn_small_group: dY2, lddy2 and msplit come together");
  TATT_CHECK_ARG(nprob > 0 && nprob <= TN_GROUP && dY && lddy && X && ldx && dW && lddw && db && M && N && R && workspace,
                 "tatt_tn_small_group: need 1..%d problems and non-null arrays", TN_GROUP);
  TnGroup g;
  memset(&g, 0, sizeof(g));
  // one 8-wave block per CU in total, shared out in proportion to each problem's matrix work per row (its count of 32x32
  // output tiles): equal shares left the blocks of a [96,32] problem idle while those of a [96,64] one ran twice as long
  int wsum = 0;
  for (int p = 0; p < nprob; ++p) wsum += ((M[p] + 31) / 32) * ((N[p] + 31) / 32);
  long off = 0;
  int per_max = 0;
  for (int p = 0; p < nprob; ++p) {
    TATT_CHECK_ARG(M[p] > 0 && N[p] > 0 && R[p] > 0 && M[p] <= 96 && N[p] <= 64 && dY[p] && X[p] && dW[p],
                   "tatt_tn_small_group: problem %d: need 0 < M <= 96, 0 < N <= 64, R > 0", p);
    TATT_CHECK_ARG(lddy[p] > 0 && ldx[p] > 0 && ((long)R[p] + 1) * lddy[p] * 4 < (1L << 31) && ((long)R[p] + 1) * ldx[p] * 4 < (1L << 31),
                   "tatt_tn_small_group: problem %d: operands must span less than 2 GiB (32-bit buffer offsets)", p);
    int nb = wsum > 0 ? TN_MAXBLK * (((M[p] + 31) / 32) * ((N[p] + 31) / 32)) / wsum : 1;
    if (nb < 1) nb = 1;
    int nblk = (R[p] + 255) / 256;
    if (nblk > nb) nblk = nb;
    const int rpb = ((R[p] + nblk - 1) / nblk + 15) / 16 * 16;
    nblk = (R[p] + rpb - 1) / rpb;
    g.dY[p] = dY[p]; g.X[p] = X[p]; g.dW[p] = dW[p]; g.db[p] = db[p];
    g.dY2[p] = dY[p]; g.lddy2[p] = lddy[p]; g.msplit[p] = 0x7fffffff;
    if (dY2 && dY2[p]) {
      TATT_CHECK_ARG(msplit[p] > 0 && msplit[p] % 32 == 0 && msplit[p] < M[p] && lddy2[p] > 0 && ((long)R[p] + 1) * lddy2[p] * 4 < (1L << 31),
                     "tatt_tn_small_group: problem %d: msplit must be a multiple of 32 inside (0, M), lddy2 > 0, dY2 under 2 GiB", p);
      g.dY2[p] = dY2[p]; g.lddy2[p] = lddy2[p]; g.msplit[p] = msplit[p];
    }
    g.lddy[p] = lddy[p]; g.ldx[p] = ldx[p]; g.lddw[p] = lddw[p];
    g.M[p] = M[p]; g.N[p] = N[p]; g.R[p] = R[p]; g.rpb[p] = rpb; g.nblk[p] = nblk;
    g.slab_off[p] = off;
    g.first[p + 1] = g.first[p] + nblk;
    const int per = M[p] * N[p] + M[p];
    off += (long)nblk * per;
    if (per > per_max) per_max = per;
  }
  for (int p = nprob; p < TN_GROUP; ++p) g.first[p + 1] = g.first[nprob];
  {
    // work items sorted by where their row range lies (as a fraction of the problem's rows), dealt to the XCDs in contiguous runs:
    // dispatch index i runs on XCD i % 8, so XCD x takes the sorted items [start_x, start_x + count_x) at indices x, x + 8, ...
    const int total = g.first[nprob];
    TATT_CHECK_ARG(total <= TN_MAXBLK, "tatt_tn_small_group: %d work items", total);
    struct Item { double mid; unsigned short code; };
    Item items[TN_MAXBLK];
    int n = 0;
    for (int p = 0; p < nprob; ++p)
      for (int b = 0; b < g.nblk[p]; ++b) {
        const double lo = (double)b * g.rpb[p], hi = lo + g.rpb[p] < g.R[p] ? lo + g.rpb[p] : (double)g.R[p];
        items[n].mid = 0.5 * (lo + hi) / g.R[p];
        items[n].code = (unsigned short)(p << 8 | b);
        ++n;
      }
    std::stable_sort(items, items + n, [](const Item& a, const Item& b) { return a.mid < b.mid; });
    int start = 0;
    for (int x = 0; x < 8; ++x) {
      const int count = total > x ? (total - x + 7) / 8 : 0;
      for (int i = 0; i < count; ++i) g.wg[x + 8 * i] = items[start + i].code;
      start += count;
    }
  }
  hipLaunchKernelGGL(tn_small_group_kernel, dim3(g.first[nprob]), dim3(TN_WAVES * 64), 0, st, g, workspace);
  TATT_LAUNCH_CHECK("tn_small_group_kernel");
  hipLaunchKernelGGL(tn_small_group_reduce, dim3(cdiv(per_max, 16), nprob), dim3(256), 0, st, g, workspace);
  TATT_LAUNCH_CHECK("tn_small_group_reduce");
  return TATT_OK;
}

}  // extern "C"
