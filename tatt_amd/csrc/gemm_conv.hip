// fp32 MFMA GEMM core (v_mfma_f32_32x32x2_f32, exact fp32 fma chains) with three front ends:
//   * tatt_gemm        : C[m,n] = sum_k A(m,k) B(n,k) (+bias, +C, relu), generic strides, batched, split-K
//   * tatt_conv_fwd    : NHWC implicit-GEMM convolution (also used for dgrad with swapped/flipped weight strides)
//   * tatt_conv_wgrad  : dW = dY^T * im2col(X), split-K over pixels into slabs
// Tile: 128(M) x 64(N) x 32(K), 512 threads = 8 waves, each wave one 32x32 accumulator (4 along M x 2 along N): the
// dominant launches have only 2 blocks per CU, so the waves that hide the loader/barrier latency must come from the block.
// LDS images carry no padding (48 KB double-buffered -> three blocks per CU): k-contiguous images ([row][32 k], read
// with ds_read_b128) XOR-swizzle the 16-byte k-group with (row & 7) so that 8 consecutive rows cover all 32 banks;
// row-contiguous images ([k][row], read with ds_read_b32 over 32 consecutive rows) are conflict free as they are.
// The k order inside an 8-wide k-group is permuted identically for A and B (lane half h takes k = 4h+e),
// which lets k-contiguous operands be read as one b128 per 4 MFMAs.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 64, BK = 32, NTHR = 512;
constexpr int LDK = BK;    // k-contiguous LDS image row stride (floats), k-groups swizzled
constexpr int LDA_M = BM;  // row-contiguous A image: [k][m]
constexpr int LDB_N = BN;  // row-contiguous B image: [k][n]
constexpr int A_FLOATS = BM * BK;
constexpr int B_FLOATS = BN * BK;
// Output tile staged in LDS for the stores: [BM rows][WIDE_LD] floats behind the first WIDE_OFF floats (the epilogues' reduction scratch).
// WIDE_LD = 72: rows 4 apart (the two halves of a wave) lie 32 banks apart, and rows stay 16-byte aligned.
constexpr int WIDE_LD = 72, WIDE_OFF = 512, WIDE_FLOATS = WIDE_OFF + BM * WIDE_LD;

enum { OP_STRIDED = 0, OP_ROWCONTIG = 1, OP_IM2COL = 2, OP_CONVW = 3, OP_IM2COL_N = 4, OP_IM2COL_M = 5, OP_CONVW_P = 6 };

struct ConvGeom {
  int H, W, Cin, KH, KW, padH, padW;
  int ldx;  // floats between consecutive pixels of X
};

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;  // per n, or null
  int M, N, K;
  long sam, sak;  // A(m,k) = A[m*sam + k*sak]
  long sbn, sbk;  // B(n,k) = B[n*sbn + k*sbk]
  long ldc;       // C(m,n) = C[m*ldc + n]
  long sAb, sBb, sCb;  // batch strides
  int splitk;     // >1: C is a slab array, slab stride = sCz
  long sCz;
  int kchunk;     // K range per split (multiple of BK)
  int accumulate; // C += result
  int relu;
  const float* A2;  // OP_STRIDED A, FAST only: k >= ksplit is read from A2 (k - ksplit), same row stride: A = [A | A2] without a concat
  int ksplit;       // multiple of BK; INT_MAX = single source
  float* C2;        // columns n >= nsplit are written to C2[m*ldc2 + n - nsplit] (accumulate2); nsplit multiple of BN, INT_MAX = none
  int nsplit, accumulate2;
  long ldc2;
  // EPI kernels (BatchNorm fused into the epilogue): per-row-block column statistics of the output tile go to epi_partial
  // [gridDim.x][2][epi_C]; EPI 1 = (mean, M2) of C itself (forward batch statistics), EPI 2 = (sum g, sum g*xhat) with
  // g = C * act'(bn(x)) and g stored in place of C (backward reduction of the BatchNorm whose output gradient this launch produces)
  float* epi_partial;
  const float* epi_x;  // EPI 2: the BatchNorm's input [M][epi_ldx]
  long epi_ldx;
  const float *epi_stat, *epi_gamma, *epi_beta;  // EPI 2: stat [2][epi_C] (mean, invstd)
  int epi_act, epi_C;
  int a16, b16;   // host only: the A / B operand is STORED as bf16 (bf16 precision only; offsets in elements as ever)
  int wide;       // full output tiles leave through LDS as float4 rows (TATT_WIDE_STORE=0 switches it off)
  int bf16;       // host only, precision of the products: 0 fp32 MFMA, 1 operands rounded to bf16 (TATT_BF16), 2 three bf16 pieces per operand (TATT_F32X3)
  ConvGeom g;     // for OP_IM2COL / OP_CONVW / OP_IM2COL_N
  long w_sn, w_sci, w_stap, w_base;  // OP_CONVW: W[n*w_sn + ci*w_sci + w_base + tap*w_stap]
};

__device__ __forceinline__ float4 ld4_guard(const float* p, bool vec_ok, long stride, int nvalid) {
  // loads up to 4 elements p[0], p[stride], ...; nvalid in [0,4]
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid <= 0) return v;
  if (vec_ok && nvalid == 4) return *reinterpret_cast<const float4*>(p);
  v.x = p[0];
  if (nvalid > 1) v.y = p[stride];
  if (nvalid > 2) v.z = p[2 * stride];
  if (nvalid > 3) v.w = p[3 * stride];
  return v;
}

// Predicated 16-byte loads without branches: a raw buffer descriptor spanning 2 GiB from the operand base; lanes that are
// off use a byte offset beyond num_records, for which buffer_load returns zeros.  The entry points check that every
// valid offset of the im2col operands stays below 2 GiB.
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int BUF_OOB = (int)0x80000000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, BUF_OOB, 0x00020000);
}
// voff: per-lane byte offset (bounds checked), soff: wave-uniform byte offset (added after the check, costs no VALU)
__device__ __forceinline__ float4 bload4s(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return *reinterpret_cast<float4*>(&v);
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, int byte_off, bool ok) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? byte_off : BUF_OOB, 0, 0);
  return *reinterpret_cast<float4*>(&v);
}

// Operands STORED as bf16 (flags TATT_X_BF16 / TATT_DY_BF16 of the convolution entry points: the bf16 step keeps the tensors whose
// every reader is a bf16 matrix-core launch in bf16): four consecutive elements are one 8-byte load, widened to fp32 in registers --
// exact, and the LDS staging below rounds them back to the same bf16 values -- so only the loaders' byte offsets (element size ES)
// and the load instruction differ from the fp32-stored form.
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 widen_bf16x4(i32x2 v) {
  return make_float4(__uint_as_float((unsigned)v.x << 16), __uint_as_float((unsigned)v.x & 0xffff0000u), __uint_as_float((unsigned)v.y << 16),
                     __uint_as_float((unsigned)v.y & 0xffff0000u));
}
template <bool H16>
__device__ __forceinline__ float4 bload4s_t(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  if constexpr (H16) return widen_bf16x4(__builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
  else return bload4s(r, voff, soff);
}
template <bool H16>
__device__ __forceinline__ float4 bload4_t(__amdgpu_buffer_rsrc_t r, int byte_off, bool ok) {
  if constexpr (H16) return widen_bf16x4(__builtin_amdgcn_raw_buffer_load_b64(r, ok ? byte_off : BUF_OOB, 0, 0));
  else return bload4(r, byte_off, ok);
}

// (tap, ci, dy, dx) of k = k0 + off, k = tap*Cin + ci.  When a K tile never straddles a tap (Cin % BK == 0) everything but
// ci is wave-uniform and is computed on the scalar unit.
__device__ __forceinline__ void tap_geom(int k0, int off, int Cin, int KW, int padH, int padW, int& tap, int& ci, int& dy, int& dx) {
  const int k0u = __builtin_amdgcn_readfirstlane(k0);
  if (Cin % BK == 0) {
    const int t = k0u / Cin;
    tap = t;
    ci = k0u - t * Cin + off;
    const int ty = t / KW;
    dy = ty - padH;
    dx = t - ty * KW - padW;
  } else {
    const int k = k0u + off, t = k / Cin, c = k - t * Cin;  // per-lane division: only the Cin < 32 convolutions come here
    tap = t;
    ci = c;
    dy = t / KW - padH;
    dx = t % KW - padW;
  }
}

// pixel index k = k0 + off -> (b, y, x) of an H x W image batch; same uniform-division trick
__device__ __forceinline__ void split_pixel(int k0, int off, int H, int W, int& b, int& y, int& x) {
  const int k0u = __builtin_amdgcn_readfirstlane(k0);
  const int hw = H * W;
  b = k0u / hw;
  const int rem = k0u - b * hw;
  y = rem / W;
  x = rem - y * W + off;
  while (x >= W) {
    x -= W;
    ++y;
  }
  while (y >= H) {
    y -= H;
    ++b;
  }
}

// ---------------------------------------------------------------------------------------------
// FAST: every operand is 16-byte granular and tile aligned (checked by fast_ok() on the host).  The loaders then cost a
// handful of VALU instructions per tile: per-thread byte offsets are computed once, the tile's position is a wave-uniform
// scalar (buffer_load soffset), predication is a v_cndmask to an out-of-range offset.  This matters more than anything
// else in the kernel: on gfx950 VALU instructions do not overlap with MFMA on a SIMD (tools/exp_mfma_lds.hip: 180 extra
// VALU ops per 16-MFMA tile drop 134 TFLOP/s to 82), so loader arithmetic is paid in MFMA issue slots.
//
// BF16 (FAST only; tatt_set_matmul_precision(1)): the operands are rounded to bf16 (round-to-nearest-even) on their way into
// the matrix core and multiplied with v_mfma_f32_32x32x16_bf16, accumulating in fp32; global memory stays fp32.  k-contiguous
// images are stored in LDS as bf16 ([row][32 k] = 64 B per row, the 16-byte k-group c at position c ^ ((row >> 1) & 3): four
// even and four odd consecutive rows cover all 32 banks), one ds_read_b128 per operand per MFMA; row-contiguous images stay
// fp32 in LDS and are packed when read.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return *reinterpret_cast<unsigned*>(&r);
}

//
// TAP9 (FAST, OP_IM2COL x OP_CONVW_P only): the 3x3, Cin = 64 convolutions of the SRBs - the roofline kernel.  K = 576 is
// exactly 18 tiles, two per tap, so the main loop is unrolled over the taps: each thread computes its 9 x 2 im2col byte
// offsets (or the out-of-range offset where the tap falls off the image) once, in registers, and every later load is a
// buffer_load with a register voffset and an immediate offset - no per-tile bounds tests, selects or scalar tile
// bookkeeping, which otherwise cost MFMA issue slots 18 times per launch.
__device__ __forceinline__ float epi_act_grad(float t, int act) {
  if (act == 1) return t > 0.f ? 1.f : 0.f;
  if (act == 2) return mish_gradf_(t);
  return 1.f;
}

// X3 (FAST loaders; flag TATT_F32X3): fp32 products on the bf16 matrix core.  Every operand element is split
// EXACTLY into three bf16 pieces x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1: 3 x 8 significant bits = the
// 24 of an fp32) while it is staged into LDS (three bf16 planes per operand, 72 KB double-buffered), and a.b is accumulated in fp32
// as the six products a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 on v_mfma_f32_32x32x16_bf16; the three dropped ones (a1b2, a2b1,
// a2b2) are <= 2^-24 |a.b| each, the rounding of one fp32 product.  12 bf16 MFMAs (8 passes each) per 32-k tile instead of 16 fp32
// ones (16 passes each): 0.375 of the matrix time, and the bf16 pipe - unlike fp32 MFMA - runs beside the VALU.
__device__ __forceinline__ void split3(const float4& v, uint2& p0, uint2& p1, uint2& p2) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  unsigned o0[2], o1[2], o2[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float a = x[2 * h], b = x[2 * h + 1];
    const unsigned h0 = pack_bf16(a, b);
    const float ra = a - __uint_as_float(h0 << 16), rb = b - __uint_as_float(h0 & 0xffff0000u);
    const unsigned h1 = pack_bf16(ra, rb);
    const float sa = ra - __uint_as_float(h1 << 16), sb = rb - __uint_as_float(h1 & 0xffff0000u);
    o0[h] = h0;
    o1[h] = h1;
    o2[h] = pack_bf16(sa, sb);
  }
  p0 = make_uint2(o0[0], o0[1]);
  p1 = make_uint2(o1[0], o1[1]);
  p2 = make_uint2(o2[0], o2[1]);
}
constexpr int X3_A_PLANE = BM * BK * 2, X3_B_PLANE = BN * BK * 2;    // bytes of one bf16 plane
constexpr int X3_BUF = 3 * (X3_A_PLANE + X3_B_PLANE);                 // bytes of one buffer: A planes 0..2, B planes 0..2

// HALO (TAP9 launches on W = 64 maps in bf16 / split-product precision): the north star's "LDS-staged halo tile".  A workgroup's
// 128 output pixels are two image rows; their 3x3 neighbourhood is four rows x 66 columns.  Instead of fetching (and, for X3,
// splitting) every input element once per tap -- nine times -- the halo of 32 input channels is staged ONCE into LDS as bf16
// planes (rows = halo pixels, the bf16 image format of the BF16 path), and the nine taps read their A fragments from it at
// shifted pixel indices; only the weight tiles stream through the double buffer.  K runs channel-half major (32 channels x 9
// taps, then the other 32): 50.7 KB of halo + 24 KB of weight tiles for X3, two workgroups per CU.  With fp32 MFMA the per-tap
// re-reads are hidden behind the matrix pipe (profiles/r02_conv_knockout.md); with the bf16 pipe they and the split are the time.
constexpr int HALO_Q = 4 * 66;            // halo pixels of a 2 x 64 pixel tile
constexpr int HALO_PLANE = HALO_Q * 64;   // bytes of one bf16 plane (32 channels)

// Workgroups are dealt to the eight XCDs round-robin in dispatch order (x fastest), each XCD with an L2 of its own.  For a split-K
// launch whose `per` tiles of one k-slab read the same operand slabs: dispatch index -> (tile s, slab z) such that the tiles of a
// slab sit on ONE XCD, consecutive in its sequence (eight slabs at a time, one per XCD; the last nz % 8 slabs are dealt tile-major).
__device__ __forceinline__ void xcd_slab_order(int lin, int per, int nz, int& s, int& z) {
  const int full = nz & ~7;
  if (lin < full * per) {
    const int G = lin / (8 * per), r = lin - G * 8 * per;
    z = G * 8 + (r & 7);
    s = r >> 3;
  } else {
    const int r = lin - full * per, t = nz - full;
    s = r / t;
    z = full + r - s * t;
  }
}

template <int AOP, int BOP, bool FAST, bool BF16, bool TAP9, int EPI, bool X3, bool HALO = false, bool A16 = false, bool B16 = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& a) {
  static_assert(!(A16 || B16) || (FAST && BF16), "bf16-stored operands feed the bf16 matrix-core launches (FAST loaders) only");
  static_assert(!A16 || AOP == OP_IM2COL || AOP == OP_IM2COL_M || AOP == OP_ROWCONTIG, "bf16-stored A: activations / gradients, not weights");
  static_assert(!B16 || BOP == OP_ROWCONTIG || BOP == OP_IM2COL_N, "bf16-stored B: activations / gradients, not weights");
  constexpr int ESA = A16 ? 2 : 4, ESB = B16 ? 2 : 4;  // bytes per stored element of the operands
  static_assert(!HALO || (TAP9 && (BF16 || X3)), "HALO is a variant of the TAP9 launches in bf16 / split-product precision");
  static_assert(FAST || !BF16, "the bf16 path exists for the FAST loaders only");
  static_assert(!TAP9 || (FAST && AOP == OP_IM2COL && BOP == OP_CONVW_P), "TAP9 is the 3x3 Cin=64 im2col specialisation");
  static_assert(!X3 || (FAST && !BF16 && BOP != OP_CONVW), "X3 exists for the FAST loaders");
  constexpr int H_NPL = X3 ? 3 : 1, H_BBUF = H_NPL * X3_B_PLANE;  // HALO: planes per operand, bytes of one weight-tile buffer
  // (at least WIDE_FLOATS: the epilogue stages the output tile here for full-line stores, see wide_store)
  constexpr int TILE_LDS = HALO ? (H_NPL * HALO_PLANE + 2 * H_BBUF) / 4 : X3 ? 2 * X3_BUF / 4 : 2 * (A_FLOATS + B_FLOATS);
  __shared__ __attribute__((aligned(16))) float lds[TILE_LDS > WIDE_FLOATS ? TILE_LDS : WIDE_FLOATS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave & 3, wn = wave >> 2;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so consecutive M tiles -
  // which share their im2col halo rows - would land on 8 different L2s and every halo row would be fetched twice from
  // the fabric (measured: FETCH_SIZE = 2x the input).  Give each XCD a contiguous range of M tiles instead.
  // Tiles that read the same operand rows should also run at the same TIME on that XCD (round 3, whole-step FETCH_SIZE table,
  // profiles/r03_step_traffic.md): the N tiles of one M tile used to be a whole grid row apart in dispatch order, so the GruBlock
  // projections (N = 192: three N tiles) fetched their A rows up to three times; they are now consecutive in their XCD's sequence.
  // Split-K launches with a handful of tiles per k-slab (the 9x9 and the up-sampling convolution's weight gradients: 3 and 18
  // tiles that all read the same two pixel slabs) keep each slab's tiles on one XCD (xcd_slab_order).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if ((gridDim.x & 7) == 0) {
    if (gridDim.y == 1) {
      bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    } else {
      const int lin = bx + gridDim.x * by;  // dispatch order inside the z slice; lin & 7 = XCD
      const int q = lin >> 3;
      by = q % gridDim.y;
      bx = (lin & 7) * (gridDim.x >> 3) + q / gridDim.y;
    }
  } else if (a.splitk > 1 && (int)gridDim.z == a.splitk && gridDim.x * gridDim.y <= 32) {
    int s;
    xcd_slab_order(bx + gridDim.x * (by + gridDim.y * bz), gridDim.x * gridDim.y, gridDim.z, s, bz);
    by = s / gridDim.x;
    bx = s - by * gridDim.x;
  }
  const int m0 = bx * BM, n0 = by * BN;
  const int zb = bz / a.splitk, zs = bz % a.splitk;
  const float* __restrict__ Ap = a.A + zb * a.sAb;
  const float* __restrict__ Bp = a.B + zb * a.sBb;
  const int kbeg = zs * a.kchunk;
  const int kend = min(a.K, kbeg + a.kchunk);
  const int ntiles = (kend - kbeg + BK - 1) / BK;

  // ---- per-thread loader state -------------------------------------------------------------
  // k-contiguous images: thread -> (A rows r0 and r0+64, B row r0, k offset kq*4)
  const int kq = tid & 7, r0 = tid >> 3;
  // A row-contiguous: m offset (tid&31)*4, k rows (tid>>5)+16j ; B row-contiguous: n offset (tid&15)*4, k row tid>>4
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(Ap), rsrcB = make_rsrc(Bp);
  const __amdgpu_buffer_rsrc_t rsrcA2 = make_rsrc(a.A2 ? a.A2 : Ap);
  int a_py[2], a_px[2], a_off[2];  // OP_IM2COL: pixel (y, x) of the thread's rows and their byte offset (tap (0,0), ci 0)
  if constexpr (AOP == OP_IM2COL) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int m = m0 + r0 + 64 * j;
      int hw = a.g.H * a.g.W;
      int b = m / hw, rem = m - b * hw;
      int y = rem / a.g.W, x = rem - y * a.g.W;
      a_py[j] = m < a.M ? y : -(1 << 20);  // rows past M fail every bounds test
      a_px[j] = x;
      a_off[j] = m < a.M ? m * a.g.ldx * ESA : 0;
    }
  }
  int bw_off = 0;  // packed OP_CONVW: byte offset of (n, tap 0, ci 0)
  bool bw_ok = false;
  if constexpr (BOP == OP_CONVW_P) {
    bw_ok = n0 + r0 < a.N;
    bw_off = bw_ok ? (int)(((long)(n0 + r0) * a.w_sn + a.w_base) * 4) : 0;
  }
  const bool a_vec = (AOP == OP_STRIDED) && a.sak == 1 && (a.sam % 4 == 0) && ((((uintptr_t)Ap) & 15) == 0);
  const bool b_vec = (BOP == OP_STRIDED) && a.sbk == 1 && (a.sbn % 4 == 0) && ((((uintptr_t)Bp) & 15) == 0);
  const bool a_rvec = (AOP == OP_ROWCONTIG) && (a.sak % 4 == 0) && ((((uintptr_t)Ap) & 15) == 0);
  const bool b_rvec = (BOP == OP_ROWCONTIG) && (a.sbk % 4 == 0) && ((((uintptr_t)Bp) & 15) == 0);
  // packed conv weights ([n][tap][ci], see tatt_conv_pack_w): 4 consecutive ci are one 16-byte load
  // OP_IM2COL_N (wgrad B operand): n = (tap, ci) fixed per thread
  int bn_tap_dy = 0, bn_tap_dx = 0, bn_ci = 0, bn_off = 0;
  bool bn_ok = false;
  if constexpr (BOP == OP_IM2COL_N) {
    int n = n0 + (tid & 15) * 4;
    bn_ok = n < a.N;
    int tap = n / a.g.Cin;
    bn_ci = n - tap * a.g.Cin;
    bn_tap_dy = tap / a.g.KW - a.g.padH;
    bn_tap_dx = tap % a.g.KW - a.g.padW;
    bn_off = ((bn_tap_dy * a.g.W + bn_tap_dx) * a.g.ldx + bn_ci) * ESB;
  }

  // OP_IM2COL_M (wgrad A operand when Cout is tiny): m = (tap, ci) fixed per thread, row(m)-contiguous image
  int am_tap_dy = 0, am_tap_dx = 0, am_ci = 0, am_off = 0;
  bool am_ok = false;
  if constexpr (AOP == OP_IM2COL_M) {
    int m = m0 + (tid & 31) * 4;
    am_ok = m < a.M;
    int tap = m / a.g.Cin;
    am_ci = m - tap * a.g.Cin;
    am_tap_dy = tap / a.g.KW - a.g.padH;
    am_tap_dx = tap % a.g.KW - a.g.padW;
    am_off = ((am_tap_dy * a.g.W + am_tap_dx) * a.g.ldx + am_ci) * ESA;
  }

  // ---- FAST loader state: constant per-thread byte offsets ----------------------------------------
  int fa_vo[2] = {BUF_OOB, BUF_OOB}, fa_k[2] = {0, 0}, fa_x[2] = {0, 0}, fb_vo = BUF_OOB, fb_k = 0, fb_x = 0;
  if constexpr (FAST) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (AOP == OP_STRIDED) {
        const int m = m0 + r0 + 64 * j;
        fa_k[j] = kq * 4;
        if (m < a.M) fa_vo[j] = (int)(((long)m * a.sam + kq * 4) * 4);
      } else if constexpr (AOP == OP_ROWCONTIG) {
        const int m = m0 + (tid & 31) * 4;
        fa_k[j] = X3 ? 2 * (tid >> 5) + j : (tid >> 5) + 16 * j;  // X3: a thread holds (k, k + 1) of its four rows -- the bf16 pairs
        if (m < a.M) fa_vo[j] = (int)(((long)fa_k[j] * a.sak + m) * ESA);
      } else if constexpr (AOP == OP_IM2COL) {
        fa_vo[j] = a_off[j] + kq * 4 * ESA;
      } else {  // OP_IM2COL_M
        fa_k[j] = X3 ? 2 * (tid >> 5) + j : (tid >> 5) + 16 * j;
        fa_x[j] = fa_k[j] + am_tap_dx;
        fa_vo[j] = am_off + fa_k[j] * a.g.ldx * ESA;
      }
    }
    if constexpr (BOP == OP_STRIDED) {
      fb_k = kq * 4;
      if (n0 + r0 < a.N) fb_vo = (int)(((long)(n0 + r0) * a.sbn + kq * 4) * 4);
    } else if constexpr (BOP == OP_ROWCONTIG) {
      const int n = n0 + (tid & 15) * 4;
      fb_k = tid >> 4;
      if (n < a.N) fb_vo = (int)(((long)fb_k * a.sbk + n) * ESB);
    } else if constexpr (BOP == OP_CONVW_P) {
      if (bw_ok) fb_vo = bw_off + kq * 16;
    } else if constexpr (BOP == OP_IM2COL_N) {
      fb_k = tid >> 4;
      fb_x = fb_k + bn_tap_dx;
      fb_vo = bn_off + fb_k * a.g.ldx * ESB;
    }
  }

  // wave-uniform position of the next tile to load (FAST)
  int t_k0 = kbeg, t_tap = 0, t_ci0 = 0, t_ty = 0, t_tx = 0, t_y = 0, t_x0 = 0;
  if constexpr (FAST && (AOP == OP_IM2COL || BOP == OP_CONVW_P)) {
    t_tap = kbeg / a.g.Cin;
    t_ci0 = kbeg - t_tap * a.g.Cin;
    t_ty = t_tap / a.g.KW;
    t_tx = t_tap - t_ty * a.g.KW;
  }
  if constexpr (FAST && (AOP == OP_IM2COL_M || BOP == OP_IM2COL_N)) {
    const int rem = kbeg % (a.g.H * a.g.W);
    t_y = rem / a.g.W;
    t_x0 = rem - t_y * a.g.W;
  }

  // two register sets: the tiles t+1 and t+2 are in flight while tile t is multiplied (global latency under load is
  // longer than one 16-MFMA tile, a prefetch distance of one tile left the MFMA pipe idle half of the time)
  float4 ra0[2], rb0, ra1[2], rb1;

  auto load_tiles = [&](int k0, float4 (&ra)[2], float4& rb) {
    if constexpr (FAST) {
      // Tiles are requested in order (kbeg, kbeg+BK, ...), so the tile's tap / pixel coordinates are wave-uniform state
      // advanced with a few scalar adds and selects; divisions here would cost ~100 SALU instructions per tile, which on
      // this machine is as good as lost MFMA issue time.
      const int k0u = t_k0;
      const int lim = kend - k0u;  // valid k in this tile (<= 0 past the end: everything predicated off)
      // ---- A ----
      if constexpr (AOP == OP_STRIDED) {
        const bool p = fa_k[0] < lim;
        const bool second = k0u >= a.ksplit;  // wave-uniform: the tile comes from the second source ([A | A2] along k)
        const __amdgpu_buffer_rsrc_t rs = second ? rsrcA2 : rsrcA;
        const int so = (second ? k0u - a.ksplit : k0u) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) ra[j] = bload4s(rs, p ? fa_vo[j] : BUF_OOB, so);
      } else if constexpr (AOP == OP_ROWCONTIG) {
        const int so = k0u * (int)a.sak * ESA;
#pragma unroll
        for (int j = 0; j < 2; ++j) ra[j] = bload4s_t<A16>(rsrcA, fa_k[j] < lim ? fa_vo[j] : BUF_OOB, so);
      } else if constexpr (AOP == OP_IM2COL) {  // Cin % BK == 0: the whole tile shares one tap
        const int dy = t_ty - a.g.padH, dx = t_tx - a.g.padW;
        const int delta = ((dy * a.g.W + dx) * a.g.ldx + t_ci0) * ESA;
        const bool live = lim > 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool ok = live && (unsigned)(a_py[j] + dy) < (unsigned)a.g.H && (unsigned)(a_px[j] + dx) < (unsigned)a.g.W;
          ra[j] = bload4s_t<A16>(rsrcA, ok ? fa_vo[j] + delta : BUF_OOB, 0);
        }
      } else {  // OP_IM2COL_M, W % BK == 0: the tile's 32 pixels share one image row
        const int y = t_y, x0 = t_x0;
        const int spix = k0u * a.g.ldx * ESA;
        const bool yok = am_ok && (unsigned)(y + am_tap_dy) < (unsigned)a.g.H;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool ok = yok && fa_k[j] < lim && (unsigned)(x0 + fa_x[j]) < (unsigned)a.g.W;
          ra[j] = bload4s_t<A16>(rsrcA, ok ? fa_vo[j] + spix : BUF_OOB, 0);
        }
      }
      // ---- B ----
      if constexpr (BOP == OP_STRIDED) {
        rb = bload4s(rsrcB, fb_k < lim ? fb_vo : BUF_OOB, k0u * 4);
      } else if constexpr (BOP == OP_ROWCONTIG) {
        rb = bload4s_t<B16>(rsrcB, fb_k < lim ? fb_vo : BUF_OOB, k0u * (int)a.sbk * ESB);
      } else if constexpr (BOP == OP_CONVW_P) {
        rb = bload4s(rsrcB, lim > 0 ? fb_vo : BUF_OOB, (t_tap * (int)a.w_stap + t_ci0) * 4);
      } else {  // OP_IM2COL_N
        const int y = t_y, x0 = t_x0;
        const bool ok = bn_ok && fb_k < lim && (unsigned)(y + bn_tap_dy) < (unsigned)a.g.H && (unsigned)(x0 + fb_x) < (unsigned)a.g.W;
        rb = bload4s_t<B16>(rsrcB, ok ? fb_vo + k0u * a.g.ldx * ESB : BUF_OOB, 0);
      }
      // ---- advance to the next tile ----
      t_k0 += BK;
      if constexpr (AOP == OP_IM2COL || BOP == OP_CONVW_P) {
        t_ci0 += BK;
        if (t_ci0 >= a.g.Cin) {
          t_ci0 = 0;
          ++t_tap;
          if (++t_tx == a.g.KW) {
            t_tx = 0;
            ++t_ty;
          }
        }
      }
      if constexpr (AOP == OP_IM2COL_M || BOP == OP_IM2COL_N) {
        t_x0 += BK;
        if (t_x0 >= a.g.W) {
          t_x0 = 0;
          if (++t_y == a.g.H) t_y = 0;
        }
      }
      return;
    }
    // ---- A ----
    if constexpr (AOP == OP_STRIDED) {
      int k = k0 + kq * 4;
      int nv = min(4, kend - k);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int m = m0 + r0 + 64 * j;
        ra[j] = ld4_guard(Ap + (long)m * a.sam + (long)k * a.sak, a_vec, a.sak, m < a.M ? nv : 0);
      }
    } else if constexpr (AOP == OP_ROWCONTIG) {
      int m = m0 + (tid & 31) * 4;
      int nv = min(4, a.M - m);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int k = k0 + (tid >> 5) + 16 * j;
        ra[j] = ld4_guard(Ap + (long)k * a.sak + m, a_rvec, 1, k < kend ? nv : 0);
      }
    } else if constexpr (AOP == OP_IM2COL_M) {
      int hw = a.g.H * a.g.W;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int k = k0 + (tid >> 5) + 16 * j;
        bool ok = am_ok && k < kend;
        int b, y, x;
        split_pixel(k0, (tid >> 5) + 16 * j, a.g.H, a.g.W, b, y, x);
        ok = ok && (unsigned)(y + am_tap_dy) < (unsigned)a.g.H && (unsigned)(x + am_tap_dx) < (unsigned)a.g.W;
        ra[j] = bload4(rsrcA, ((b * hw + y * a.g.W + x) * a.g.ldx) * 4 + am_off, ok);
      }
    } else {  // OP_IM2COL: k = tap*Cin + ci, 4 consecutive k share a tap (Cin % 4 == 0)
      int k = k0 + kq * 4;
      bool kok = k < kend;
      int tap, ci, dy, dx;
      tap_geom(k0, kq * 4, a.g.Cin, a.g.KW, a.g.padH, a.g.padW, tap, ci, dy, dx);
      const int delta = ((dy * a.g.W + dx) * a.g.ldx + ci) * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bool ok = kok && (unsigned)(a_py[j] + dy) < (unsigned)a.g.H && (unsigned)(a_px[j] + dx) < (unsigned)a.g.W;
        ra[j] = bload4(rsrcA, a_off[j] + delta, ok);
      }
    }
    // ---- B ----
    if constexpr (BOP == OP_STRIDED) {
      int k = k0 + kq * 4;
      int nv = min(4, kend - k);
      {
        int n = n0 + r0;
        rb = ld4_guard(Bp + (long)n * a.sbn + (long)k * a.sbk, b_vec, a.sbk, n < a.N ? nv : 0);
      }
    } else if constexpr (BOP == OP_ROWCONTIG) {
      int n = n0 + (tid & 15) * 4;
      int nv = min(4, a.N - n);
      {
        int k = k0 + (tid >> 4);
        rb = ld4_guard(Bp + (long)k * a.sbk + n, b_rvec, 1, k < kend ? nv : 0);
      }
    } else if constexpr (BOP == OP_CONVW || BOP == OP_CONVW_P) {
      int k = k0 + kq * 4;
      bool kok = k < kend;
      int tap, ci, dy, dx;
      tap_geom(k0, kq * 4, a.g.Cin, a.g.KW, a.g.padH, a.g.padW, tap, ci, dy, dx);
      if constexpr (BOP == OP_CONVW_P) {  // packed [n][tap][ci] (tatt_conv_pack_w): one 16-byte load, no branch
        rb = bload4(rsrcB, bw_off + (tap * (int)a.w_stap + ci) * 4, kok && bw_ok);
      } else {
        int n = n0 + r0;
        const float* p = Bp + (long)n * a.w_sn + (long)ci * a.w_sci + a.w_base + (long)tap * a.w_stap;
        if (kok && n < a.N) {
          rb = make_float4(p[0], p[a.w_sci], p[2 * a.w_sci], p[3 * a.w_sci]);
        } else {
          rb = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else {  // OP_IM2COL_N : B(n=(tap,ci), k=pixel) = X[pixel shifted by tap][ci], row(n)-contiguous image
      int hw = a.g.H * a.g.W;
      {
        int k = k0 + (tid >> 4);
        bool ok = bn_ok && k < kend;
        int b, y, x;
        split_pixel(k0, tid >> 4, a.g.H, a.g.W, b, y, x);
        ok = ok && (unsigned)(y + bn_tap_dy) < (unsigned)a.g.H && (unsigned)(x + bn_tap_dx) < (unsigned)a.g.W;
        rb = bload4(rsrcB, ((b * hw + y * a.g.W + x) * a.g.ldx) * 4 + bn_off, ok);
      }
    }
  };

  const int st_swz = (kq ^ (r0 & 7)) * 4;  // rows r0 and r0 + 64 share r0 & 7
  const int st_swz16 = (((kq >> 1) ^ ((r0 >> 1) & 3)) << 4) + ((kq & 1) << 3);  // bf16 image: byte offset of k = 4 kq inside the row
  auto store_tiles = [&](int buf, const float4 (&ra)[2], const float4& rb) {
    if constexpr (X3) {
      char* Ab = reinterpret_cast<char*>(lds) + buf * X3_BUF;
      char* Bb = Ab + 3 * X3_A_PLANE;
      uint2 p0, p1, p2;
      if constexpr (AOP == OP_ROWCONTIG || AOP == OP_IM2COL_M) {
        // row-contiguous source: the bf16 dwords need (k, k+1) of one row, so under X3 the loader gives a thread k = 2 (tid >> 5)
        // and k + 1 of its four rows (fa_k above).  Plane layout [k pair 16][row] dwords: one 16-byte store per plane here, four
        // ds_read_b32 per fragment in compute().
        // (the thread's two loads are k = 2 (tid >> 5) and k + 1 of the same four rows: no lane exchange)
        uint2 u0, u1, u2;
        split3(make_float4(ra[0].x, ra[1].x, ra[0].y, ra[1].y), p0, p1, p2);
        split3(make_float4(ra[0].z, ra[1].z, ra[0].w, ra[1].w), u0, u1, u2);
        char* d = Ab + (((tid >> 5) * BM + (tid & 31) * 4) << 2);
        *reinterpret_cast<uint4*>(d) = make_uint4(p0.x, p0.y, u0.x, u0.y);
        *reinterpret_cast<uint4*>(d + X3_A_PLANE) = make_uint4(p1.x, p1.y, u1.x, u1.y);
        *reinterpret_cast<uint4*>(d + 2 * X3_A_PLANE) = make_uint4(p2.x, p2.y, u2.x, u2.y);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          split3(ra[j], p0, p1, p2);
          char* d = Ab + (r0 + 64 * j) * 64 + st_swz16;
          *reinterpret_cast<uint2*>(d) = p0;
          *reinterpret_cast<uint2*>(d + X3_A_PLANE) = p1;
          *reinterpret_cast<uint2*>(d + 2 * X3_A_PLANE) = p2;
        }
      }
      if constexpr (BOP == OP_ROWCONTIG || BOP == OP_IM2COL_N) {  // k = tid >> 4: lanes l and l ^ 16 hold k even / odd of the same 4 columns
        const int h = (lane >> 4) & 1;
        const float s0 = h ? rb.x : rb.z, s1 = h ? rb.y : rb.w;
        const float q0 = __shfl_xor(s0, 16, 64), q1 = __shfl_xor(s1, 16, 64);
        const float4 v = h ? make_float4(q0, rb.z, q1, rb.w) : make_float4(rb.x, q0, rb.y, q1);
        split3(v, p0, p1, p2);
        char* d = Bb + (((2 * wave + (lane >> 5)) * BN + (tid & 15) * 4 + 2 * h) << 2);
        *reinterpret_cast<uint2*>(d) = p0;
        *reinterpret_cast<uint2*>(d + X3_B_PLANE) = p1;
        *reinterpret_cast<uint2*>(d + 2 * X3_B_PLANE) = p2;
      } else {
        split3(rb, p0, p1, p2);
        char* d = Bb + r0 * 64 + st_swz16;
        *reinterpret_cast<uint2*>(d) = p0;
        *reinterpret_cast<uint2*>(d + X3_B_PLANE) = p1;
        *reinterpret_cast<uint2*>(d + 2 * X3_B_PLANE) = p2;
      }
      return;
    }
    float* As = lds + buf * (A_FLOATS + B_FLOATS);
    float* Bs = As + A_FLOATS;
    if constexpr (AOP == OP_ROWCONTIG || AOP == OP_IM2COL_M) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<float4*>(As + ((tid >> 5) + 16 * j) * LDA_M + (tid & 31) * 4) = ra[j];
    } else if constexpr (BF16) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(As) + (r0 + 64 * j) * 64 + st_swz16) =
            make_uint2(pack_bf16(ra[j].x, ra[j].y), pack_bf16(ra[j].z, ra[j].w));
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(As + (r0 + 64 * j) * LDK + st_swz) = ra[j];
    }
    if constexpr (BOP == OP_ROWCONTIG || BOP == OP_IM2COL_N) {
      *reinterpret_cast<float4*>(Bs + (tid >> 4) * LDB_N + (tid & 15) * 4) = rb;
    } else if constexpr (BF16) {
      *reinterpret_cast<uint2*>(reinterpret_cast<char*>(Bs) + r0 * 64 + st_swz16) = make_uint2(pack_bf16(rb.x, rb.y), pack_bf16(rb.z, rb.w));
    } else {
      *reinterpret_cast<float4*>(Bs + r0 * LDK + st_swz) = rb;
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  auto compute = [&](int cur) {
    if constexpr (X3) {
      const char* Ab = reinterpret_cast<const char*>(lds) + cur * X3_BUF;
      const char* Bb = Ab + 3 * X3_A_PLANE;
      const int rowa = wm * 32 + li, rowb = wn * 32 + li;
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        uint4 av[3], bv[3];
        if constexpr (AOP == OP_ROWCONTIG || AOP == OP_IM2COL_M) {
          const char* pa = Ab + (((8 * g + 4 * lh) * BM + rowa) << 2);
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const unsigned* q = reinterpret_cast<const unsigned*>(pa + p * X3_A_PLANE);
            av[p] = make_uint4(q[0], q[BM], q[2 * BM], q[3 * BM]);
          }
        } else {
          const char* pa = Ab + rowa * 64 + (((2 * g + lh) ^ ((rowa >> 1) & 3)) << 4);
#pragma unroll
          for (int p = 0; p < 3; ++p) av[p] = *reinterpret_cast<const uint4*>(pa + p * X3_A_PLANE);
        }
        if constexpr (BOP == OP_ROWCONTIG || BOP == OP_IM2COL_N) {
          const char* pb = Bb + (((8 * g + 4 * lh) * BN + rowb) << 2);
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const unsigned* q = reinterpret_cast<const unsigned*>(pb + p * X3_B_PLANE);
            bv[p] = make_uint4(q[0], q[BN], q[2 * BN], q[3 * BN]);
          }
        } else {
          const char* pb = Bb + rowb * 64 + (((2 * g + lh) ^ ((rowb >> 1) & 3)) << 4);
#pragma unroll
          for (int p = 0; p < 3; ++p) bv[p] = *reinterpret_cast<const uint4*>(pb + p * X3_B_PLANE);
        }
#define X3_MFMA(i, j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&av[i]), *reinterpret_cast<bf16x8*>(&bv[j]), acc, 0, 0, 0)
        X3_MFMA(2, 0); X3_MFMA(1, 1); X3_MFMA(0, 2);  // smallest terms first
        X3_MFMA(1, 0); X3_MFMA(0, 1);
        X3_MFMA(0, 0);
#undef X3_MFMA
      }
      return;
    }
    const float* As = lds + cur * (A_FLOATS + B_FLOATS);
    const float* Bs = As + A_FLOATS;
    if constexpr (BF16) {
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {  // lane half lh supplies k = 16 g + 8 lh .. + 7 of its row, for A and B alike
        uint4 au, bu;
        if constexpr (AOP == OP_ROWCONTIG || AOP == OP_IM2COL_M) {
          const float* p = As + (g * 16 + 8 * lh) * LDA_M + wm * 32 + li;
          au = make_uint4(pack_bf16(p[0], p[LDA_M]), pack_bf16(p[2 * LDA_M], p[3 * LDA_M]), pack_bf16(p[4 * LDA_M], p[5 * LDA_M]),
                          pack_bf16(p[6 * LDA_M], p[7 * LDA_M]));
        } else {
          const int row = wm * 32 + li;
          au = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(As) + row * 64 + (((2 * g + lh) ^ ((row >> 1) & 3)) << 4));
        }
        if constexpr (BOP == OP_ROWCONTIG || BOP == OP_IM2COL_N) {
          const float* p = Bs + (g * 16 + 8 * lh) * LDB_N + wn * 32 + li;
          bu = make_uint4(pack_bf16(p[0], p[LDB_N]), pack_bf16(p[2 * LDB_N], p[3 * LDB_N]), pack_bf16(p[4 * LDB_N], p[5 * LDB_N]),
                          pack_bf16(p[6 * LDB_N], p[7 * LDB_N]));
        } else {
          const int row = wn * 32 + li;
          bu = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Bs) + row * 64 + (((2 * g + lh) ^ ((row >> 1) & 3)) << 4));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&au), *reinterpret_cast<bf16x8*>(&bu), acc, 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float av[4], bv[4];
      if constexpr (AOP == OP_ROWCONTIG || AOP == OP_IM2COL_M) {
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = As[(kg * 8 + 4 * lh + e) * LDA_M + wm * 32 + li];
      } else {
        float4 v = *reinterpret_cast<const float4*>(As + (wm * 32 + li) * LDK + (((2 * kg + lh) ^ (li & 7)) << 2));
        av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
      }
      if constexpr (BOP == OP_ROWCONTIG || BOP == OP_IM2COL_N) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = Bs[(kg * 8 + 4 * lh + e) * LDB_N + wn * 32 + li];
      } else {
        float4 v = *reinterpret_cast<const float4*>(Bs + (wn * 32 + li) * LDK + (((2 * kg + lh) ^ (li & 7)) << 2));
        bv[0] = v.x; bv[1] = v.y; bv[2] = v.z; bv[3] = v.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
    }
  };

  // tile t+1 sits in register set (t+1)&1, tile t+2 in set t&1
  // tiles 0 and 1 are requested back to back (two register sets), tile 2 as soon as tile 0 has been parked in LDS: the
  // short-K launches (K = 64..192: two to six tiles) are one memory latency shorter than with "load, store, load, load"
  if constexpr (HALO) {
    char* Hb = reinterpret_cast<char*>(lds);
    char* Bb0 = Hb + H_NPL * HALO_PLANE;
    const int hw = a.g.H * a.g.W;
    const int bimg = m0 / hw, y0 = (m0 - bimg * hw) >> 6;           // W == 64: the tile is rows y0, y0 + 1 of image bimg
    const int mrow = wm * 32 + li;
    const int hq_base = (mrow >> 6) * 66 + (mrow & 63);             // halo pixel of this lane's A row for tap (0, 0)
    const int rowb = wn * 32 + li;
    float4 hreg[5];
    auto halo_load = [&](int half) {
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        const int q = r0 + 64 * it, hy = q / 66, hx = q - hy * 66;
        const int y = y0 - 1 + hy, x = hx - 1;
        const bool ok = q < HALO_Q && (unsigned)y < (unsigned)a.g.H && (unsigned)x < (unsigned)a.g.W;
        hreg[it] = bload4_t<A16>(rsrcA, (((bimg * a.g.H + y) * a.g.W + x) * a.g.ldx + half * 32 + kq * 4) * ESA, ok);
      }
    };
    auto halo_store = [&]() {
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        const int q = r0 + 64 * it;
        if (q < HALO_Q) {
          char* d = Hb + q * 64 + ((((kq >> 1) ^ ((q >> 1) & 3)) << 4) + ((kq & 1) << 3));
          if constexpr (X3) {
            uint2 p0, p1, p2;
            split3(hreg[it], p0, p1, p2);
            *reinterpret_cast<uint2*>(d) = p0;
            *reinterpret_cast<uint2*>(d + HALO_PLANE) = p1;
            *reinterpret_cast<uint2*>(d + 2 * HALO_PLANE) = p2;
          } else {
            *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16(hreg[it].x, hreg[it].y), pack_bf16(hreg[it].z, hreg[it].w));
          }
        }
      }
    };
    auto load_b = [&](int tt, float4& rb) {  // tile tt: channel half tt / 9, tap tt % 9 (constants after unrolling)
      if (tt >= 18) return;
      rb = bload4s(rsrcB, fb_vo, ((tt % 9) * 64 + (tt / 9) * BK) * 4);
    };
    auto store_b = [&](int buf, const float4& rb) {
      char* d = Bb0 + buf * H_BBUF + r0 * 64 + st_swz16;
      if (tid < 512) {
        if constexpr (X3) {
          uint2 p0, p1, p2;
          split3(rb, p0, p1, p2);
          *reinterpret_cast<uint2*>(d) = p0;
          *reinterpret_cast<uint2*>(d + X3_B_PLANE) = p1;
          *reinterpret_cast<uint2*>(d + 2 * X3_B_PLANE) = p2;
        } else {
          *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16(rb.x, rb.y), pack_bf16(rb.z, rb.w));
        }
      }
    };
    auto compute_h = [&](int buf, int tap) {
      const int hq = hq_base + (tap / 3) * 66 + (tap % 3);
      const char* Bb = Bb0 + buf * H_BBUF;
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        const char* pa = Hb + hq * 64 + (((2 * g + lh) ^ ((hq >> 1) & 3)) << 4);
        const char* pb = Bb + rowb * 64 + (((2 * g + lh) ^ ((rowb >> 1) & 3)) << 4);
        if constexpr (X3) {
          uint4 av[3], bv[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            av[p] = *reinterpret_cast<const uint4*>(pa + p * HALO_PLANE);
            bv[p] = *reinterpret_cast<const uint4*>(pb + p * X3_B_PLANE);
          }
#define X3_MFMA(i, j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&av[i]), *reinterpret_cast<bf16x8*>(&bv[j]), acc, 0, 0, 0)
          X3_MFMA(2, 0); X3_MFMA(1, 1); X3_MFMA(0, 2);
          X3_MFMA(1, 0); X3_MFMA(0, 1);
          X3_MFMA(0, 0);
#undef X3_MFMA
        } else {
          uint4 au = *reinterpret_cast<const uint4*>(pa), bu = *reinterpret_cast<const uint4*>(pb);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&au), *reinterpret_cast<bf16x8*>(&bu), acc, 0, 0, 0);
        }
      }
    };
    halo_load(0);
    load_b(0, rb0);
    load_b(1, rb1);
    halo_store();
    store_b(0, rb0);
    load_b(2, rb0);
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 18; tt += 2) {
      store_b(1, rb1);
      load_b(tt + 3, rb1);
      if (tt == 4) halo_load(1);                 // the other 32 channels: requested five tiles before they are needed
      compute_h(0, tt % 9);
      __syncthreads();
      if (tt == 8) {                             // tile 8 was the last one of the first channel half: swap the halo
        halo_store();
      }
      if (tt + 2 < 18) store_b(0, rb0);
      load_b(tt + 4, rb0);
      if (tt == 8) __syncthreads();
      compute_h(1, (tt + 1) % 9);
      if (tt + 2 < 18) __syncthreads();
    }
  } else if constexpr (TAP9) {
    int va[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const int delta = ((dy * a.g.W + dx) * a.g.ldx) * ESA;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool ok = (unsigned)(a_py[j] + dy) < (unsigned)a.g.H && (unsigned)(a_px[j] + dx) < (unsigned)a.g.W;
        va[tap][j] = ok ? fa_vo[j] + delta : BUF_OOB;
      }
    }
    auto load_tap = [&](int tile, float4 (&ra)[2], float4& rb) {  // tile is a constant after unrolling: tap = tile / 2, ci half = tile % 2
      if (tile >= 18) return;
      const int tap = tile >> 1, half = tile & 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) ra[j] = bload4s_t<A16>(rsrcA, va[tap][j], half * (BK * ESA));
      rb = bload4s(rsrcB, fb_vo, (tap * 64 + half * BK) * 4);
    };
    load_tap(0, ra0, rb0);
    load_tap(1, ra1, rb1);
    store_tiles(0, ra0, rb0);
    load_tap(2, ra0, rb0);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 18; t += 2) {
      store_tiles(1, ra1, rb1);
      load_tap(t + 3, ra1, rb1);
      compute(0);
      __syncthreads();
      if (t + 2 < 18) store_tiles(0, ra0, rb0);
      load_tap(t + 4, ra0, rb0);
      compute(1);
      if (t + 2 < 18) __syncthreads();
    }
  } else {
  load_tiles(kbeg, ra0, rb0);
  load_tiles(kbeg + BK, ra1, rb1);
  store_tiles(0, ra0, rb0);
  load_tiles(kbeg + 2 * BK, ra0, rb0);
  __syncthreads();
  // Two tiles per trip, no early exit: an odd tile count multiplies one all-zero tile (loads past kend are predicated off).
  // Straight-line code matters here: with a conditional path around the second half the compiler's s_waitcnt bookkeeping
  // falls back to vmcnt(0) at the loop head and the second tile in flight is lost.
  for (int t = 0; t < ntiles; t += 2) {
    store_tiles(1, ra1, rb1);
    load_tiles(kbeg + (t + 3) * BK, ra1, rb1);
    compute(0);
    __syncthreads();
    store_tiles(0, ra0, rb0);
    load_tiles(kbeg + (t + 4) * BK, ra0, rb0);
    compute(1);
    __syncthreads();
  }
  }

  // ---- epilogue: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ------------------
  float* Cp = a.C + zb * a.sCb + zs * a.sCz;
  const bool plain = a.splitk == 1;
  // Full tiles leave through LDS: in the accumulator layout a lane holds sixteen rows of ONE column, i.e. sixteen dword stores per
  // lane, each wave instruction two 128-byte row segments -- and the CU's store path takes ~20 cycles per wave store instruction
  // whatever its width (tools/experiments/gru_proj.hip: 6.6 us of a 40 us launch, additive to the matrix time).  Staged as
  // [row][64 columns] the tile goes out as four float4 stores per thread, each wave instruction four full 256-byte rows:
  // plain epilogue (EPI 0) 3x3 convolution 47.6 -> 45.4 us, GruBlock projections 42.0 -> 39.2 (K = 128) and 29.4 -> 27.0 us (K = 64).
  const bool wide_rows = a.wide && a.M - m0 >= BM && a.N - n0 >= BN;  // block-uniform
  auto wide_store = [&](const float (&v)[16], float* dst, long ldc) {       // dst = &C[m0][first column of this N tile]
    __syncthreads();  // operand tiles are dead (the reduction scratch in front of the stage may still be read: it is not touched)
    float* stage = lds + WIDE_OFF;
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * WIDE_LD + wn * 32 + li] = v[r];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BM / 32; ++k) {
      const int row = (tid >> 4) + 32 * k, c4 = tid & 15;
      *reinterpret_cast<float4*>(dst + (long)row * ldc + c4 * 4) = *reinterpret_cast<const float4*>(stage + row * WIDE_LD + c4 * 4);
    }
  };
  if constexpr (EPI != 0) {
    // BatchNorm fused into the epilogue (host guarantees splitk == 1, no accumulate / relu on the first destination).  The
    // workgroup's 128 x 64 tile is reduced over its rows: registers -> the two lane halves -> the four waves along M through LDS.
    __syncthreads();  // every wave has finished reading the operand tiles: the LDS is free
    float* red = lds;  // [2][4][64]
    const int nl = wn * 32 + li, n = n0 + nl;
    const bool first = n0 < a.nsplit;  // block-uniform: the BatchNorm sits on the first destination
    const bool nok = n < a.N;
    const bool sok = nok && first && n < a.epi_C;
    const float bv = (a.bias && nok) ? a.bias[n] : 0.f;
    float v[16];
    bool ok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      ok[r] = m < a.M;
      v[r] = acc[r] + bv;
    }
    const int cnt = min(BM, a.M - m0);
    if constexpr (EPI == 1) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += ok[r] ? v[r] : 0.f;
      s += __shfl_xor(s, 32, 64);
      if (lh == 0) red[wm * 64 + nl] = s;
      __syncthreads();
      const float mean = (red[nl] + red[64 + nl] + red[128 + nl] + red[192 + nl]) / (float)cnt;
      float q = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = v[r] - mean;
        q += ok[r] ? d * d : 0.f;
      }
      q += __shfl_xor(q, 32, 64);
      if (lh == 0) red[256 + wm * 64 + nl] = q;
      __syncthreads();
      if (wm == 0 && lh == 0 && sok) {
        a.epi_partial[((long)bx * 2 + 0) * a.epi_C + n] = mean;
        a.epi_partial[((long)bx * 2 + 1) * a.epi_C + n] = red[256 + nl] + red[256 + 64 + nl] + red[256 + 128 + nl] + red[256 + 192 + nl];
      }
    } else {
      float s1 = 0.f, s2 = 0.f;
      if (sok) {
        float xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)  // row clamped, not predicated: sixteen independent loads, no branch around any of them
          xv[r] = a.epi_x[(long)min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, a.M - 1) * a.epi_ldx + n];
        const float mean = a.epi_stat[n], inv = a.epi_stat[a.epi_C + n], ga = a.epi_gamma[n], be = a.epi_beta[n];
        auto pass = [&](auto actc) {  // the activation is block-uniform: chosen once, outside the element loop
          constexpr int ACT = decltype(actc)::value;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float xh = (xv[r] - mean) * inv;
            const float g = ok[r] ? v[r] * epi_act_grad(xh * ga + be, ACT) : 0.f;
            s1 += g;
            s2 += g * xh;
            v[r] = g;
          }
        };
        if (a.epi_act == 2) pass(std::integral_constant<int, 2>{});
        else if (a.epi_act == 1) pass(std::integral_constant<int, 1>{});
        else pass(std::integral_constant<int, 0>{});
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lh == 0) {
        red[wm * 64 + nl] = s1;
        red[256 + wm * 64 + nl] = s2;
      }
      __syncthreads();
      if (wm == 0 && lh == 0 && sok) {
        a.epi_partial[((long)bx * 2 + 0) * a.epi_C + n] = red[nl] + red[64 + nl] + red[128 + nl] + red[192 + nl];
        a.epi_partial[((long)bx * 2 + 1) * a.epi_C + n] = red[256 + nl] + red[256 + 64 + nl] + red[256 + 128 + nl] + red[256 + 192 + nl];
      }
    }
    if (!nok) return;
    long ldc = a.ldc;
    int accum = a.accumulate, nn = n;
    if (!first) {
      Cp = a.C2;
      nn = n - a.nsplit;
      ldc = a.ldc2;
      accum = a.accumulate2;
    }
    if (accum) {  // all sixteen reads of the destination in flight together (a read-add-write per row waited on each in turn)
      float prev[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[r] = Cp[(long)min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, a.M - 1) * ldc + nn];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += prev[r];
    }
    // (sixteen dword stores per lane: behind the reductions' barriers the staged form measured no faster -- EPI 1 47.4 -> 47.9 us)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (!ok[r]) continue;
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      Cp[(long)m * ldc + nn] = v[r];
    }
    return;
  } else {
    int n = n0 + wn * 32 + li;
    if (n >= a.N) return;
    float bv = (plain && a.bias) ? a.bias[n] : 0.f;
    long ldc = a.ldc;
    int accum = a.accumulate;
    if (n0 >= a.nsplit) {  // block-uniform: this column tile belongs to the second destination
      Cp = a.C2;
      n -= a.nsplit;
      ldc = a.ldc2;
      accum = a.accumulate2;
    }
    {
      float* dst = Cp + (long)m0 * ldc + (n - (wn * 32 + li));
      if (wide_rows && !(plain && accum) && (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {  // block-uniform
        const bool relu = plain && a.relu;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = acc[r] + bv;
          if (relu) v[r] = fmaxf(v[r], 0.f);
        }
        wide_store(v, dst, ldc);
        return;
      }
    }
    float prev[16];
    if (plain && accum) {  // all sixteen reads of the destination in flight together (a read-add-write per row waited on each in turn)
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[r] = Cp[(long)min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, a.M - 1) * ldc + n];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[r] = 0.f;
    }
    const bool relu = plain && a.relu;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= a.M) continue;
      float v = acc[r] + bv + prev[r];
      if (relu) v = fmaxf(v, 0.f);
      Cp[(long)m * ldc + n] = v;
    }
  }
}

template <int AOP, int BOP, bool FAST, bool BF16 = false, bool TAP9 = false, int EPI = 0, bool A16 = false, bool B16 = false>
__global__ __launch_bounds__(NTHR, 2) void gemm_kernel(GemmArgs a) {
  gemm_body<AOP, BOP, FAST, BF16, TAP9, EPI, false, false, A16, B16>(a);
}
// the same launches with fp32 products made of three bf16 pieces per operand (flag TATT_F32X3, see gemm_body)
// (second launch-bound argument = waves per SIMD in HIP: four, i.e. two workgroups per CU and at most 128 registers -- with "2" one
//  rarely used instantiation took 133 and would have run one workgroup per CU)
template <int AOP, int BOP, bool TAP9 = false, int EPI = 0>
__global__ __launch_bounds__(NTHR, 4) void gemm_x3_kernel(GemmArgs a) {
  gemm_body<AOP, BOP, true, false, TAP9, EPI, true>(a);
}
// TAP9 launches on 64-wide maps in bf16 (X3 = false) or split-product (X3 = true) precision with the input halo resident in LDS
template <int AOP, int BOP, int EPI, bool X3, bool A16 = false>
__global__ __launch_bounds__(NTHR, 2) void gemm_halo_kernel(GemmArgs a) {
  gemm_body<AOP, BOP, true, !X3, true, EPI, X3, true, A16>(a);
}
const int g_wide_store_host = [] { const char* e = getenv("TATT_WIDE_STORE"); return (e && e[0] == '0') ? 0 : 1; }();
const int g_halo = [] { const char* e = getenv("TATT_CONV_HALO"); return (e && e[0] == '0') ? 0 : 1; }();  // =0: per-tap loads (A/B runs)
inline bool halo_ok(const GemmArgs& a) {
  return g_halo && a.bf16 != 0 && a.g.W == 64 && a.g.H % 2 == 0 && a.M % BM == 0 && a.M == (a.M / (a.g.H * a.g.W)) * a.g.H * a.g.W &&
         (long)a.M * a.g.ldx * 4 < (1L << 31);
}
template <int AOP, int BOP>
constexpr bool x3_ok = BOP != OP_CONVW;
inline int prec_of(int flags) { return (flags & TATT_BF16) ? 1 : (flags & TATT_F32X3) ? 2 : 0; }

// sum slabs: dst[perm(i)] (+)= bias? + sum_z slab[z][i]
// perm mode 0: identity over [M,N] with ldc; mode 1: conv weight (co, tap, ci) -> OIHW (co, ci, tap)
// ZL threads share one output element: a workgroup covers 256 / ZL consecutive elements (a 128-byte run per slab row for ZL = 8),
// thread group zl sums slabs zl, zl + ZL, ... with independent loads and the ZL partial sums are combined through LDS -- with one
// thread per element the 36k-element weight gradients of the 3x3 convolutions filled 144 workgroups that each walked 154 slabs in
// sequence.
template <int ZL>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int nslab, float* __restrict__ dst,
                                     const float* __restrict__ bias, int M, int N, long ldc, int accumulate, int relu,
                                     int perm, int Cin, int T, long dst_batch_stride) {
  constexpr int EPB = 256 / ZL;  // elements per workgroup
  __shared__ float red[256];
  const long i = (long)blockIdx.x * EPB + (threadIdx.x % EPB);
  const int zl = threadIdx.x / EPB;
  const bool on = i < (long)M * N;
  // blockIdx.y = batch index: slabs [batch][nslab][M*N], dst advanced by dst_batch_stride
  slabs += (long)blockIdx.y * nslab * slab_stride;
  dst += (long)blockIdx.y * dst_batch_stride;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // independent chains so that the loads overlap
  if (on) {
    int z = zl;
    for (; z + 3 * ZL < nslab; z += 4 * ZL) {
      s0 += slabs[(long)(z + 0 * ZL) * slab_stride + i];
      s1 += slabs[(long)(z + 1 * ZL) * slab_stride + i];
      s2 += slabs[(long)(z + 2 * ZL) * slab_stride + i];
      s3 += slabs[(long)(z + 3 * ZL) * slab_stride + i];
    }
    for (; z < nslab; z += ZL) s0 += slabs[(long)z * slab_stride + i];
  }
  float s = (s0 + s1) + (s2 + s3);
  if (ZL > 1) {
    red[threadIdx.x] = s;
    __syncthreads();
    if (zl != 0) return;
    s = 0.f;
#pragma unroll
    for (int l = 0; l < ZL; ++l) s += red[l * EPB + threadIdx.x];
  }
  if (!on) return;
  int m = (int)(i / N), n = (int)(i - (long)m * N);
  long o;
  if (perm == 1) {
    int tap = n / Cin, ci = n - tap * Cin;
    o = ((long)m * Cin + ci) * T + tap;
  } else if (perm == 2) {  // wgrad with (tap,ci) on M: m = (tap, ci), n = co
    int tap = m / Cin, ci = m - tap * Cin;
    o = ((long)n * Cin + ci) * T + tap;
  } else if (perm == 3) {  // tiny-Cout wgrad, roles swapped: m = ci_f, n = (flipped tap, co_f); Cin here = Cout_f
    int tapf = n / Cin, co = n - tapf * Cin;
    o = ((long)co * M + m) * T + (T - 1 - tapf);
  } else if (perm == 4) {  // tiny-Cout wgrad with (flipped tap, co_f) on M and ci_f on N; Cin here = Cout_f
    int tapf = m / Cin, co = m - tapf * Cin;
    o = ((long)co * N + n) * T + (T - 1 - tapf);
  } else {
    o = (long)m * ldc + n;
  }
  if (bias) s += bias[n];
  if (accumulate) s += dst[o];
  if (relu) s = fmaxf(s, 0.f);
  dst[o] = s;
}

// few outputs and many slabs (weight gradients): 8 lanes per element; otherwise one thread per element
inline void launch_splitk_reduce(hipStream_t st, int batch, const float* slabs, long slab_stride, int nslab, float* dst, const float* bias,
                                 int M, int N, long ldc, int accumulate, int relu, int perm, int Cin, int T, long dst_batch_stride) {
  const long tot = (long)M * N;
  if (nslab >= 32 && tot <= 262144)
    hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3(cdiv(tot, 32), batch), dim3(256), 0, st, slabs, slab_stride, nslab, dst, bias, M, N, ldc,
                       accumulate, relu, perm, Cin, T, dst_batch_stride);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(cdiv(tot, 256), batch), dim3(256), 0, st, slabs, slab_stride, nslab, dst, bias, M, N, ldc,
                       accumulate, relu, perm, Cin, T, dst_batch_stride);
}

// ---- skinny-Cout convolution (Cout <= 16: the 9x9 64->4 output conv and the 9x9 dgrad of block1) ----------------
// A Cout of 4 would fill 1/16 of the 64-wide MFMA N tile.  Instead: Z[(tap,co)][pixel] = sum_ci Wr[(tap,co)][ci] * X[pixel][ci]
// (a dense GEMM with M = T*Cout, K = Cin, planar output), then Y[p][co] = bias + sum_tap Z[(tap,co)][p + shift(tap)].
// Wf[co][tap][ci] = W[co][ci][tap] ; Wd[ci][tap][co] = W[co][ci][T-1-tap]  (W is OIHW)
__global__ void conv_pack_w_kernel(const float* __restrict__ Wt, float* __restrict__ Wf, float* __restrict__ Wd, int Cout, int Cin, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cout * Cin * T) return;
  {
    const int ci = i % Cin, tap = (i / Cin) % T, co = i / (Cin * T);
    Wf[i] = Wt[((long)co * Cin + ci) * T + tap];
  }
  {
    const int co = i % Cout, tap = (i / Cout) % T, ci = i / (Cout * T);
    Wd[i] = Wt[((long)co * Cin + ci) * T + (T - 1 - tap)];
  }
}

// the same for up to PACK_MAX layers in one launch (blockIdx.y = layer)
constexpr int PACK_MAX = 32;
struct PackGroup {
  const float* Wt[PACK_MAX];
  float* Wf[PACK_MAX];
  float* Wd[PACK_MAX];
  int Cout[PACK_MAX], Cin[PACK_MAX], T[PACK_MAX];
};
__global__ void conv_pack_w_group_kernel(PackGroup g) {
  const int l = blockIdx.y, Cout = g.Cout[l], Cin = g.Cin[l], T = g.T[l];
  const float* __restrict__ Wt = g.Wt[l];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Cout * Cin * T; i += gridDim.x * blockDim.x) {
    {
      const int ci = i % Cin, tap = (i / Cin) % T, co = i / (Cin * T);
      g.Wf[l][i] = Wt[((long)co * Cin + ci) * T + tap];
    }
    {
      const int co = i % Cout, tap = (i / Cout) % T, ci = i / (Cout * T);
      g.Wd[l][i] = Wt[((long)co * Cin + ci) * T + (T - 1 - tap)];
    }
  }
}

__global__ void skinny_repack_w(const float* __restrict__ Wt, float* __restrict__ Wr, int Cin, int Cout, int T, long w_sn, long w_sci,
                                long w_base, long w_stap) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;  // (tap, co, ci)
  if (i >= T * Cout * Cin) return;
  int ci = i % Cin, co = (i / Cin) % Cout, tap = i / (Cin * Cout);
  Wr[i] = Wt[(long)co * w_sn + (long)ci * w_sci + w_base + (long)tap * w_stap];
}

template <int COUT>
__global__ __launch_bounds__(256) void skinny_gather(const float* __restrict__ Z, const float* __restrict__ bias,
                                                     float* __restrict__ Y, int B, int H, int W, int KH, int KW, long ldy,
                                                     int accumulate) {
  const long P = (long)B * H * W;
  long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int hw = H * W;
  const int rem = (int)(p % hw);
  const int y = rem / W, x = rem - y * W;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = bias ? bias[c] : 0.f;
  for (int dy = 0; dy < KH; ++dy) {
    const int yy = y + dy - KH / 2;
    if (yy < 0 || yy >= H) continue;
    for (int dx = 0; dx < KW; ++dx) {
      const int xx = x + dx - KW / 2;
      if (xx < 0 || xx >= W) continue;
      const long q = p + (long)(dy - KH / 2) * W + (dx - KW / 2);
      const float* zp = Z + (long)(dy * KW + dx) * COUT * P + q;
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[c] += zp[(long)c * P];
    }
  }
  float* dst = Y + p * ldy;
#pragma unroll
  for (int c = 0; c < COUT; ++c) dst[c] = accumulate ? dst[c] + acc[c] : acc[c];
}

// ---- direct 9x9 convolution with FOUR output channels (the 64 -> 4 output conv, model/tsrn.py:625, and the input gradient of the
// 9x9 4 -> 64 conv of block1, :598) ------------------------------------------------------------------------------------------
// With Cout = 4 every input value feeds only four multiply-adds per tap: on the matrix core 15/16 of a 64-wide tile would be padding,
// and the planar-GEMM detour above (Z[(tap,co)][pixel], then a shifted gather) moves 324 floats per pixel through memory (340 MB at
// B = 64).  fp32 FMA on the vector ALU has the same peak as fp32 MFMA on this chip, so the convolution runs there directly:
//   * a workgroup owns an 8 x 64 pixel tile of one image; the input channels are walked in chunks of eight, wave (q, hh) takes
//     channel quad q of the chunk for rows 4hh..4hh+3, and the two quads are summed through LDS at the end;
//   * per chunk the tile + halo (16 x 72 pixels x 8 channels) is staged in LDS as [quad][row][col] float4, one pad slot per four
//     columns so that the 16 lanes of a ds_read_b128 pass (four columns apart) hit distinct banks;
//   * a thread computes 4 consecutive pixels x 4 output channels: per tap row it reads its 12-column window once (12 float4) and
//     reuses it for the nine horizontal taps from registers -- 576 FMAs per 12 window reads;
//   * the weights, packed as [quad][tap][ci][co], are wave-uniform: staged per chunk in LDS and read as broadcasts.  (As scalar
//     loads -- SGPR operands of v_pk_fma -- they cost more than the FMAs: the 83 KB overflow the scalar cache, scalar loads return
//     out of order, so every s_load_dwordx16 was waited for with lgkmcnt(0) at L2 latency: 183 us instead of ~70 us of FMA work.)
constexpr int C4_KS = 9, C4_HALO = 4, C4_TH = 8, C4_TW = 64, C4_PX = 4, C4_Q = 2;  // C4_Q channel quads (8 channels) per LDS chunk
constexpr int C4_RW = C4_TW + 2 * C4_HALO, C4_RWP = C4_RW + C4_RW / 4, C4_RH = C4_TH + 2 * C4_HALO;
constexpr int C4_TILE4 = C4_Q * C4_RH * C4_RWP;   // float4 slots of the staged input tile
constexpr int C4_W4 = C4_Q * C4_KS * C4_KS * 4;   // float4 slots of the chunk's weights

__global__ void conv_c4_pack_w(const float* __restrict__ Wt, float* __restrict__ Wp, int Cin, int T, long w_sn, long w_sci, long w_base,
                               long w_stap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // ((quad * T + tap) * 4 + ci4) * 4 + co
  if (i >= Cin * T * 4) return;
  const int co = i & 3, ci4 = (i >> 2) & 3, tap = (i >> 4) % T, q = (i >> 4) / T;
  Wp[i] = Wt[(long)co * w_sn + (long)(q * 4 + ci4) * w_sci + w_base + (long)tap * w_stap];
}

__global__ __launch_bounds__(256, 2) void conv_c4_direct(const float* __restrict__ X, long ldx, const float* __restrict__ Wp,
                                                         const float* __restrict__ bias, float* __restrict__ Y, long ldy, int H, int W,
                                                         int Cin, int tiles_x, int tiles_y, int accumulate) {
  __shared__ float4 smem[C4_TILE4 + C4_W4];  // 46,080 + 10,368 bytes: two workgroups per CU
  float4* tile = smem;
  float4* wl = smem + C4_TILE4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = wave & 1, hh = wave >> 1;  // this wave: channel quad q of the chunk, rows hh*4 .. hh*4+3 of the tile
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, b = t / tiles_y;
  const int y0 = ty * C4_TH, x0 = tx * C4_TW;
  const int gy = hh * 4 + (lane >> 4), gx = (lane & 15) * C4_PX;  // this thread's pixels: row y0 + gy, columns x0 + gx .. + 3
  const float* Xb = X + (long)b * H * W * ldx;
  // output channels in pairs: v_pk_fma_f32 (two FMAs per lane and issue; the input value is the same for both halves -- op_sel
  // picks it from its register, no splat copy).  Plain v_fma_f32 tops out at half the fp32 vector peak.
  f32x2 acc[C4_PX][2];
#pragma unroll
  for (int p = 0; p < C4_PX; ++p) acc[p][0] = acc[p][1] = f32x2{0.f, 0.f};
  const int nchunk = Cin / (4 * C4_Q);
  // Staging: every thread's share of the chunk (nine tile float4 + three weight float4) is fetched into registers with branch-free
  // buffer loads (out-of-image halo and the tail of the index range: an out-of-range offset, which reads as 0) one chunk AHEAD, while
  // the current chunk is multiplied, and parked in LDS between two barriers.  (With `if (in bounds) load; store` in a loop the loads
  // went out one at a time and the staging, not the FMAs, was the kernel: 175 us for 70 us of arithmetic.)
  constexpr int C4_NT = (C4_RH * C4_RW * C4_Q + 255) / 256, C4_NW = (C4_W4 + 255) / 256;
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(Xb), rsW = make_rsrc(Wp);
  int voff[C4_NT], lidx[C4_NT];
#pragma unroll
  for (int k = 0; k < C4_NT; ++k) {
    const int i = tid + 256 * k;
    const bool valid = i < C4_RH * C4_RW * C4_Q;
    const int qq = i % C4_Q, pos = i / C4_Q, r = pos / C4_RW, c = pos - r * C4_RW;
    const int yy = y0 + r - C4_HALO, xx = x0 + c - C4_HALO;
    const bool inb = valid && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    voff[k] = inb ? (int)((((long)yy * W + xx) * ldx + qq * 4) * 4) : BUF_OOB;
    lidx[k] = valid ? (qq * C4_RH + r) * C4_RWP + c + (c >> 2) : -1;
  }
  float4 tv[C4_NT], wv[C4_NW];
  auto fetch = [&](int ch) {
#pragma unroll
    for (int k = 0; k < C4_NT; ++k) tv[k] = bload4s(rsX, voff[k], ch * (C4_Q * 16));
#pragma unroll
    for (int k = 0; k < C4_NW; ++k) wv[k] = bload4s(rsW, tid + 256 * k < C4_W4 ? (tid + 256 * k) * 16 : BUF_OOB, ch * (C4_W4 * 16));
  };
  fetch(0);
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();  // the previous chunk's window / weight reads are done
#pragma unroll
    for (int k = 0; k < C4_NT; ++k)
      if (lidx[k] >= 0) tile[lidx[k]] = tv[k];
    // the chunk's weights [quad][tap][ci][co]: read below with wave-uniform addresses (LDS broadcast).  Scalar loads would be the
    // natural fit, but the 83 KB of weights overflow the scalar cache and every s_load then waits on L2 with lgkmcnt(0)
#pragma unroll
    for (int k = 0; k < C4_NW; ++k)
      if (tid + 256 * k < C4_W4) wl[tid + 256 * k] = wv[k];
    __syncthreads();
    fetch(min(ch + 1, nchunk - 1));  // in flight under the FMAs below (the last trip re-reads its own chunk: no branch)
    const float4* wq = wl + q * C4_KS * C4_KS * 4;
    const float4* trow = tile + (q * C4_RH + gy) * C4_RWP;
#pragma unroll 1
    for (int dy = 0; dy < C4_KS; ++dy) {
      float4 win[C4_PX + C4_KS - 1];
#pragma unroll
      for (int j = 0; j < C4_PX + C4_KS - 1; ++j) win[j] = trow[dy * C4_RWP + (gx + j) + ((gx + j) >> 2)];
      const float4* wr = wq + dy * C4_KS * 4;
#pragma unroll
      for (int dx = 0; dx < C4_KS; ++dx) {
        const float4 w0 = wr[dx * 4], w1 = wr[dx * 4 + 1], w2 = wr[dx * 4 + 2], w3 = wr[dx * 4 + 3];  // rows ci = 0..3 of [ci][co]
        const f32x2 w0a = {w0.x, w0.y}, w0b = {w0.z, w0.w}, w1a = {w1.x, w1.y}, w1b = {w1.z, w1.w};
        const f32x2 w2a = {w2.x, w2.y}, w2b = {w2.z, w2.w}, w3a = {w3.x, w3.y}, w3b = {w3.z, w3.w};
#pragma unroll
        for (int p = 0; p < C4_PX; ++p) {
          const float4 xv = win[p + dx];
          const f32x2 sx = {xv.x, xv.x}, sy = {xv.y, xv.y}, sz = {xv.z, xv.z}, sw = {xv.w, xv.w};
          acc[p][0] = __builtin_elementwise_fma(sw, w3a, __builtin_elementwise_fma(sz, w2a, __builtin_elementwise_fma(sy, w1a, __builtin_elementwise_fma(sx, w0a, acc[p][0]))));
          acc[p][1] = __builtin_elementwise_fma(sw, w3b, __builtin_elementwise_fma(sz, w2b, __builtin_elementwise_fma(sy, w1b, __builtin_elementwise_fma(sx, w0b, acc[p][1]))));
        }
      }
    }
  }
  // sum the channel partitions of the two quad waves: [quad][pixel of the tile][co]
  __syncthreads();
  float4* red = smem;
  const int pix0 = gy * C4_TW + gx;
#pragma unroll
  for (int p = 0; p < C4_PX; ++p) red[q * (C4_TH * C4_TW) + pix0 + p] = make_float4(acc[p][0].x, acc[p][0].y, acc[p][1].x, acc[p][1].y);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < C4_TH * C4_TW / 256; ++k) {
    const int pix = tid + 256 * k, py = pix / C4_TW, px = pix - py * C4_TW;
    const int yy = y0 + py, xx = x0 + px;
    if (yy >= H || xx >= W) continue;
    float4 s = red[pix];
#pragma unroll
    for (int w2 = 1; w2 < C4_Q; ++w2) {
      const float4 v = red[w2 * (C4_TH * C4_TW) + pix];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) { s.x += bias[0]; s.y += bias[1]; s.z += bias[2]; s.w += bias[3]; }
    float* dst = Y + ((long)(b * H + yy) * W + xx) * ldy;
    if (accumulate) { s.x += dst[0]; s.y += dst[1]; s.z += dst[2]; s.w += dst[3]; }
    dst[0] = s.x; dst[1] = s.y; dst[2] = s.z; dst[3] = s.w;
  }
}

inline bool c4_direct_ok(int Cin, int Cout, int KH, int KW, long ldx, const void* X) {
  return Cout == 4 && KH == C4_KS && KW == C4_KS && Cin % (4 * C4_Q) == 0 && ldx % 4 == 0 && (((uintptr_t)X) & 15) == 0;
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
inline void init_args(GemmArgs& a) {
  memset(&a, 0, sizeof(a));
  a.ksplit = a.nsplit = 0x7fffffff;  // single A source, single destination
  a.wide = g_wide_store_host;
}
const long GIB2 = 1L << 31;

// Can both operands use the FAST loaders (see gemm_kernel)?  Offsets are 32-bit byte offsets from the operand base.
template <int AOP, int BOP>
bool fast_ok(const GemmArgs& a) {
  bool ok = a.kchunk % BK == 0 && a.K > 0;
  if (AOP == OP_STRIDED) ok = ok && a.sak == 1 && a.sam % 4 == 0 && a.K % 4 == 0 && al16(a.A) && a.sAb % 4 == 0 && ((long)a.M * a.sam + a.K) * 4 < GIB2;
  if (AOP == OP_ROWCONTIG) ok = ok && a.sam == 1 && a.sak % 4 == 0 && a.M % 4 == 0 && al16(a.A) && a.sAb % 4 == 0 && ((long)a.K * a.sak + a.M) * 4 < GIB2;
  if (AOP == OP_IM2COL) ok = ok && a.g.Cin % BK == 0;
  if (AOP == OP_IM2COL_M) ok = ok && a.g.W % BK == 0;
  if (BOP == OP_STRIDED) ok = ok && a.sbk == 1 && a.sbn % 4 == 0 && a.K % 4 == 0 && al16(a.B) && a.sBb % 4 == 0 && ((long)a.N * a.sbn + a.K) * 4 < GIB2;
  if (BOP == OP_ROWCONTIG) ok = ok && a.sbn == 1 && a.sbk % 4 == 0 && a.N % 4 == 0 && al16(a.B) && a.sBb % 4 == 0 && ((long)a.K * a.sbk + a.N) * 4 < GIB2;
  if (BOP == OP_CONVW) ok = false;
  if (BOP == OP_CONVW_P) ok = ok && a.g.Cin % BK == 0;
  if (BOP == OP_IM2COL_N) ok = ok && a.g.W % BK == 0;
  return ok;
}

const int g_tap9 = [] { const char* e = getenv("TATT_CONV_TAP9"); return (e && e[0] == '0') ? 0 : 1; }();  // =0: generic loop (A/B runs)

// launches whose activation / gradient operands are STORED as bf16 (GemmArgs::a16 / b16): the bf16 step's trunk convolutions
template <int AOP, int BOP, int EPI>
int launch16(const GemmArgs& a, dim3 grid, hipStream_t st) {
  if (a.bf16 != 1 || !fast_ok<AOP, BOP>(a)) {
    tatt_set_error("bf16-stored operands (TATT_X_BF16 / TATT_DY_BF16) need TATT_BF16 precision and 16-byte granular operand layouts");
    return TATT_ERR_ARG;
  }
  if constexpr (AOP == OP_IM2COL && BOP == OP_CONVW_P) {
    if (a.a16 && !a.b16) {
      const bool tap9 = g_tap9 && a.g.Cin == 64 && a.g.KH == 3 && a.g.KW == 3 && a.splitk == 1 && a.K == 576 && a.w_stap == 64;
      if (tap9 && halo_ok(a)) hipLaunchKernelGGL((gemm_halo_kernel<AOP, BOP, EPI, false, true>), grid, dim3(NTHR), 0, st, a);
      else hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, true, false, EPI, true, false>), grid, dim3(NTHR), 0, st, a);
      TATT_LAUNCH_CHECK("gemm kernel <bf16-stored input>");
      return TATT_OK;
    }
  } else if constexpr (AOP == OP_IM2COL_M && BOP == OP_ROWCONTIG && EPI == 0) {
    if (a.b16) {  // weight gradient, (tap, ci) on M: A = X (fp32 or bf16), B = dY (bf16)
      if (a.a16) hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, true, false, 0, true, true>), grid, dim3(NTHR), 0, st, a);
      else hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, true, false, 0, false, true>), grid, dim3(NTHR), 0, st, a);
      TATT_LAUNCH_CHECK("gemm kernel <bf16-stored wgrad operands>");
      return TATT_OK;
    }
  } else if constexpr (AOP == OP_ROWCONTIG && BOP == OP_IM2COL_N && EPI == 0) {
    if (a.a16 && a.b16) {  // weight gradient, Cout on M: A = dY, B = X
      hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, true, false, 0, true, true>), grid, dim3(NTHR), 0, st, a);
      TATT_LAUNCH_CHECK("gemm kernel <bf16-stored wgrad operands>");
      return TATT_OK;
    }
  }
  tatt_set_error("no kernel for this combination of bf16-stored operands (supported: convolution input; weight gradient with dY, or X and dY, "
                 "stored as bf16 and more than 16 output channels)");
  return TATT_ERR_ARG;
}

template <int AOP, int BOP>
int launch(const GemmArgs& a, int batch, hipStream_t st) {
  dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), batch * a.splitk);
  if (a.a16 || a.b16) return launch16<AOP, BOP, 0>(a, grid, st);
  if constexpr (AOP == OP_IM2COL && BOP == OP_CONVW_P) {  // 3x3, Cin = 64, packed weights, whole K in one pass: tap-unrolled loop
    if (g_tap9 && fast_ok<AOP, BOP>(a) && a.g.Cin == 64 && a.g.KH == 3 && a.g.KW == 3 && a.splitk == 1 && a.K == 576 && a.w_stap == 64) {
      if (halo_ok(a)) {
        if (a.bf16 == 1) hipLaunchKernelGGL((gemm_halo_kernel<AOP, BOP, 0, false>), grid, dim3(NTHR), 0, st, a);
        else hipLaunchKernelGGL((gemm_halo_kernel<AOP, BOP, 0, true>), grid, dim3(NTHR), 0, st, a);
      } else if (a.bf16 == 1) hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, true, true>), grid, dim3(NTHR), 0, st, a);
      else if (a.bf16 == 2) hipLaunchKernelGGL((gemm_x3_kernel<AOP, BOP, true>), grid, dim3(NTHR), 0, st, a);
      else hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, false, true>), grid, dim3(NTHR), 0, st, a);
      TATT_LAUNCH_CHECK("gemm_kernel<tap9>");
      return TATT_OK;
    }
  }
  if (fast_ok<AOP, BOP>(a)) {
    bool done = false;
    if constexpr (x3_ok<AOP, BOP>) {
      if (a.bf16 == 2) {
        hipLaunchKernelGGL((gemm_x3_kernel<AOP, BOP>), grid, dim3(NTHR), 0, st, a);
        done = true;
      }
    }
    if (done) {
    } else if (a.bf16 == 1) hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, true>), grid, dim3(NTHR), 0, st, a);
    else hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true>), grid, dim3(NTHR), 0, st, a);  // (TATT_F32X3 on operand layouts without an X3 kernel: exact fp32)
  } else hipLaunchKernelGGL((gemm_kernel<AOP, BOP, false>), grid, dim3(NTHR), 0, st, a);
  TATT_LAUNCH_CHECK("gemm_kernel");
  return TATT_OK;
}

// ---- weight gradient of the SRB convolutions (3x3, 64 -> 64: T*Cin = 576 = 3 x 192, Cout <= 64), fp32 MFMA ---------------------
// dW^T[(tap,ci)][co] = sum_pixels X[pixel + shift(tap)][ci] * dY[pixel][co]: both operands are pixel-major, so both LDS images are
// row-contiguous ([k][m], [k][n]) and the generic tile above pays two ds_read_b32 per MFMA (knock-out: 57.7 us with them, 47.4 us
// without) and pads M = 576 to five 128-row tiles.  Here the tile is 192 x 64 x 32 with twelve waves = 3 (M) x 2 (N) x 2 (halves of
// the 32-pixel k tile): a wave owns 64 rows as two interleaved 32-row sub-tiles (rows 2i and 2i + 1), so one ds_read_b64 of A and
// one ds_read_b32 of B feed two MFMAs, M needs no padding, and the two k halves are added through LDS once, in the epilogue.
// Slab format, split-K and the slab sum are those of the generic launch.
constexpr int WG_BM = 192, WG_THR = 768, WG_BUF = (WG_BM + BN) * BK;
__global__ __launch_bounds__(WG_THR, 6) void wgrad192_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * WG_BUF];  // 64 KB: two workgroups (24 waves) per CU
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave % 3, wn = (wave / 3) & 1, kh = wave / 6;
  // the three M tiles of a k-slab read the same X and dY pixels: one XCD, side by side (reads per launch 94.9 -> 36.7 MiB against
  // 32 algorithmic, profiles/r03_step_traffic.md)
  int mt, zs;
  xcd_slab_order(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x, gridDim.z, mt, zs);
  const int m0 = mt * WG_BM;
  const int kbeg = zs * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
  const int ntiles = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(a.A), rsrcB = make_rsrc(a.B);
  // A loader: thread -> four consecutive m = (tap, ci..ci+3) at pixels a_k and a_k + 16 of the tile
  const int a_m4 = tid % 48, a_k = tid / 48;
  int fa_vo[2], fa_x[2], tdy;
  {
    const int m = m0 + a_m4 * 4;
    const int tap = m / a.g.Cin, ci = m - tap * a.g.Cin;
    tdy = tap / a.g.KW - a.g.padH;
    const int tdx = tap % a.g.KW - a.g.padW;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fa_x[j] = a_k + 16 * j + tdx;
      fa_vo[j] = (((tdy * a.g.W + tdx) + a_k + 16 * j) * a.g.ldx + ci) * 4;
    }
  }
  // B loader (waves 0..7): thread -> four consecutive co of pixel b_k
  const int b_n4 = tid & 15, b_k = tid >> 4;
  const bool b_thr = wave < 8;
  const int fb_vo = (b_thr && b_n4 * 4 < a.N) ? (int)(((long)b_k * a.sbk + b_n4 * 4) * 4) : BUF_OOB;
  int t_k0 = kbeg, t_y, t_x0;
  {
    const int rem = kbeg % (a.g.H * a.g.W);
    t_y = rem / a.g.W;
    t_x0 = rem - t_y * a.g.W;
  }
  float4 ra0[2], rb0, ra1[2], rb1;
  auto load_tiles = [&](float4 (&ra)[2], float4& rb) {  // W % 32 == 0: the tile's 32 pixels share one image row
    const int lim = kend - t_k0;
    const int spix = t_k0 * a.g.ldx * 4;
    const bool yok = (unsigned)(t_y + tdy) < (unsigned)a.g.H;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = yok && a_k + 16 * j < lim && (unsigned)(t_x0 + fa_x[j]) < (unsigned)a.g.W;
      ra[j] = bload4s(rsrcA, ok ? fa_vo[j] + spix : BUF_OOB, 0);
    }
    rb = bload4s(rsrcB, b_k < lim ? fb_vo : BUF_OOB, t_k0 * (int)a.sbk * 4);
    t_k0 += BK;
    t_x0 += BK;
    if (t_x0 >= a.g.W) {
      t_x0 = 0;
      if (++t_y == a.g.H) t_y = 0;
    }
  };
  auto store_tiles = [&](int buf, const float4 (&ra)[2], const float4& rb) {
    float* As = lds + buf * WG_BUF;
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(As + (a_k + 16 * j) * WG_BM + a_m4 * 4) = ra[j];
    if (b_thr) *reinterpret_cast<float4*>(As + WG_BM * BK + b_k * BN + b_n4 * 4) = rb;
  };
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
  auto compute = [&](int cur) {
    const float* As = lds + cur * WG_BUF + (kh * 16 + lh) * WG_BM + wm * 64 + 2 * li;
    const float* Bs = lds + cur * WG_BUF + WG_BM * BK + (kh * 16 + lh) * BN + wn * 32 + li;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const float2 av = *reinterpret_cast<const float2*>(As + 2 * p * WG_BM);
      const float bv = Bs[2 * p * BN];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv, acc1, 0, 0, 0);
    }
  };
  load_tiles(ra0, rb0);
  load_tiles(ra1, rb1);
  store_tiles(0, ra0, rb0);
  load_tiles(ra0, rb0);
  __syncthreads();
  for (int t = 0; t < ntiles; t += 2) {  // two tiles per trip, straight-line (see gemm_body); tiles past the end are all zeros
    store_tiles(1, ra1, rb1);
    load_tiles(ra1, rb1);
    compute(0);
    __syncthreads();
    store_tiles(0, ra0, rb0);
    load_tiles(ra0, rb0);
    compute(1);
    __syncthreads();
  }
  // ---- epilogue: the k-half-1 waves park their accumulators in LDS, the k-half-0 waves add them and write the slab ----
  float* red = lds;  // [6 waves][32 regs][64 lanes] = 48 KB
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      red[((wave - 6) * 32 + r) * 64 + lane] = acc0[r];
      red[((wave - 6) * 32 + 16 + r) * 64 + lane] = acc1[r];
    }
  }
  __syncthreads();
  if (kh == 1) return;
  const int n = wn * 32 + li;
  if (n >= a.N) return;
  float* Cp = a.C + zs * a.sCz;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh);
    Cp[(long)m * a.ldc + n] = acc0[r] + red[(wave * 32 + r) * 64 + lane];
    Cp[(long)(m + 1) * a.ldc + n] = acc1[r] + red[(wave * 32 + 16 + r) * 64 + lane];
  }
}
const int g_wgrad192 = [] { const char* e = getenv("TATT_WGRAD192"); return (e && e[0] == '0') ? 0 : 1; }();  // =0: the generic tile (A/B runs)

// launches with a BatchNorm epilogue (EPI 1 / 2, see GemmArgs): FAST loaders only, one K pass, whole tile rows
template <int AOP, int BOP, int EPI>
int launch_epi(const GemmArgs& a, hipStream_t st) {
  if (!fast_ok<AOP, BOP>(a) || a.splitk != 1 || a.relu || a.accumulate || !a.epi_partial || a.epi_C <= 0) {
    tatt_set_error("BatchNorm-epilogue launch: needs FAST operand layouts (16-byte granular), one K pass, no relu / accumulate");
    return TATT_ERR_ARG;
  }
  dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), 1);
  if (a.a16 || a.b16) return launch16<AOP, BOP, EPI>(a, grid, st);
  if constexpr (AOP == OP_IM2COL && BOP == OP_CONVW_P) {
    if (g_tap9 && a.g.Cin == 64 && a.g.KH == 3 && a.g.KW == 3 && a.K == 576 && a.w_stap == 64) {
      if (halo_ok(a)) {
        if (a.bf16 == 1) hipLaunchKernelGGL((gemm_halo_kernel<AOP, BOP, EPI, false>), grid, dim3(NTHR), 0, st, a);
        else hipLaunchKernelGGL((gemm_halo_kernel<AOP, BOP, EPI, true>), grid, dim3(NTHR), 0, st, a);
      } else if (a.bf16 == 1) hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, true, true, EPI>), grid, dim3(NTHR), 0, st, a);
      else if (a.bf16 == 2) hipLaunchKernelGGL((gemm_x3_kernel<AOP, BOP, true, EPI>), grid, dim3(NTHR), 0, st, a);
      else hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, false, true, EPI>), grid, dim3(NTHR), 0, st, a);
      TATT_LAUNCH_CHECK("gemm_kernel<tap9, bn epilogue>");
      return TATT_OK;
    }
  }
  if constexpr (x3_ok<AOP, BOP>) {
    if (a.bf16 == 2) {
      hipLaunchKernelGGL((gemm_x3_kernel<AOP, BOP, false, EPI>), grid, dim3(NTHR), 0, st, a);
      TATT_LAUNCH_CHECK("gemm_x3_kernel<bn epilogue>");
      return TATT_OK;
    }
  }
  if (a.bf16 == 1) hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, true, false, EPI>), grid, dim3(NTHR), 0, st, a);
  else hipLaunchKernelGGL((gemm_kernel<AOP, BOP, true, false, false, EPI>), grid, dim3(NTHR), 0, st, a);
  TATT_LAUNCH_CHECK("gemm_kernel<bn epilogue>");
  return TATT_OK;
}

}  // namespace

extern "C" {

int tatt_conv_pack_w(const float* Wt, float* Wf, float* Wd, int Cout, int Cin, int KH, int KW, hipStream_t st) {
  TATT_CHECK_ARG(Wt && Wf && Wd && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "tatt_conv_pack_w: bad arguments");
  const int n = Cout * Cin * KH * KW;
  hipLaunchKernelGGL(conv_pack_w_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, Wt, Wf, Wd, Cout, Cin, KH * KW);
  TATT_LAUNCH_CHECK("conv_pack_w_kernel");
  return TATT_OK;
}

// tatt_conv_pack_w for n <= 32 layers in one launch; host arrays of length n (Wf[i], Wd[i]: Cout[i]*Cin[i]*KH[i]*KW[i] floats each).
int tatt_conv_pack_w_group(int n, const float* const* Wt, float* const* Wf, float* const* Wd, const int* Cout, const int* Cin,
                           const int* KHW, hipStream_t st) {
  TATT_CHECK_ARG(n > 0 && n <= PACK_MAX && Wt && Wf && Wd && Cout && Cin && KHW, "tatt_conv_pack_w_group: 1..%d layers, non-null arrays", PACK_MAX);
  PackGroup g;
  memset(&g, 0, sizeof(g));
  long mx = 0;
  for (int i = 0; i < n; ++i) {
    TATT_CHECK_ARG(Wt[i] && Wf[i] && Wd[i] && Cout[i] > 0 && Cin[i] > 0 && KHW[i] > 0, "tatt_conv_pack_w_group: bad layer %d", i);
    g.Wt[i] = Wt[i]; g.Wf[i] = Wf[i]; g.Wd[i] = Wd[i]; g.Cout[i] = Cout[i]; g.Cin[i] = Cin[i]; g.T[i] = KHW[i];
    const long e = (long)Cout[i] * Cin[i] * KHW[i];
    if (e > mx) mx = e;
  }
  int bx = cdiv(mx, 256 * 4);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(conv_pack_w_group_kernel, dim3(bx, n), dim3(256), 0, st, g);
  TATT_LAUNCH_CHECK("conv_pack_w_group_kernel");
  return TATT_OK;
}

// Workspace floats needed by a split-K call (0 when splitk <= 1).
size_t tatt_gemm_workspace(int M, int N, int splitk) { return splitk > 1 ? (size_t)M * N * splitk : 0; }

int tatt_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sam, long sak,
              long sbn, long sbk, long ldc, int batch, long sAb, long sBb, long sCb, int accumulate, int flags, int splitk,
              float* workspace, hipStream_t st) {
  const int relu = flags & TATT_RELU;
  TATT_CHECK_ARG(M > 0 && N > 0 && K >= 0 && batch > 0, "tatt_gemm: bad dims M=%d N=%d K=%d batch=%d", M, N, K, batch);
  TATT_CHECK_ARG(splitk >= 1 && (splitk == 1 || workspace), "tatt_gemm: split-K needs a workspace");
  TATT_CHECK_ARG(splitk == 1 || batch == 1 || bias == nullptr, "tatt_gemm: batched split-K takes no bias");
  GemmArgs a;
  init_args(a);
  a.A = A; a.B = B; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K;
  a.sam = sam; a.sak = sak; a.sbn = sbn; a.sbk = sbk; a.ldc = ldc;
  a.sAb = sAb; a.sBb = sBb; a.sCb = sCb;
  a.splitk = splitk;
  a.accumulate = accumulate; a.relu = relu; a.bf16 = prec_of(flags);
  a.kchunk = cdiv(cdiv(K, splitk), BK) * BK;
  if (a.kchunk == 0) a.kchunk = BK;
  if (splitk > 1) {  // slabs [batch][splitk][M][N]
    a.C = workspace;
    a.ldc = N;
    a.sCz = (long)M * N;
    a.sCb = (long)M * N * splitk;
  }
  const bool arow = (sam == 1 && sak != 1), brow = (sbn == 1 && sbk != 1);
  int rc;
  if (!arow && !brow) rc = launch<OP_STRIDED, OP_STRIDED>(a, batch, st);
  else if (arow && !brow) rc = launch<OP_ROWCONTIG, OP_STRIDED>(a, batch, st);
  else if (!arow && brow) rc = launch<OP_STRIDED, OP_ROWCONTIG>(a, batch, st);
  else rc = launch<OP_ROWCONTIG, OP_ROWCONTIG>(a, batch, st);
  if (rc) return rc;
  if (splitk > 1) {
    launch_splitk_reduce(st, batch, workspace, (long)M * N, splitk, C, bias, M, N, ldc, accumulate, relu, 0, 0, 0, sCb);
    TATT_LAUNCH_CHECK("splitk_reduce");
  }
  return TATT_OK;
}

// tatt_gemm over a k-concatenated A and/or into two destinations (see include/tatt_hip.h): the GRU blocks of the SRBs read
// cat(conv features, text-prior map) and hand gradients back to both, without materialising the concatenation.
int tatt_gemm_cat_bnbwd(const float* A, const float* A2, int ksplit, const float* B, float* C, float* C2, int nsplit, const float* bias,
                        int M, int N, int K, long sam, long sbn, long sbk, long ldc, long ldc2, int accumulate, int accumulate2, int flags,
                        const float* bn_x, long bn_ldx, const float* bn_stat, const float* bn_gamma, const float* bn_beta, int bn_act,
                        int bn_C, float* partial, hipStream_t st);

int tatt_gemm_cat(const float* A, const float* A2, int ksplit, const float* B, float* C, float* C2, int nsplit, const float* bias, int M,
                  int N, int K, long sam, long sbn, long sbk, long ldc, long ldc2, int accumulate, int accumulate2, int flags,
                  hipStream_t st) {
  return tatt_gemm_cat_bnbwd(A, A2, ksplit, B, C, C2, nsplit, bias, M, N, K, sam, sbn, sbk, ldc, ldc2, accumulate, accumulate2, flags,
                             nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, st);
}

// tatt_gemm_cat whose first destination is the output gradient of a BatchNorm (bn_x = that BatchNorm's input, partial != NULL):
// the epilogue stores g = C * act'(bn(x)) and the per-row-block sums (sum g, sum g xhat) for tatt_bn_bwd_from_partial.
int tatt_gemm_cat_bnbwd(const float* A, const float* A2, int ksplit, const float* B, float* C, float* C2, int nsplit, const float* bias,
                        int M, int N, int K, long sam, long sbn, long sbk, long ldc, long ldc2, int accumulate, int accumulate2, int flags,
                        const float* bn_x, long bn_ldx, const float* bn_stat, const float* bn_gamma, const float* bn_beta, int bn_act,
                        int bn_C, float* partial, hipStream_t st) {
  TATT_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "tatt_gemm_cat: bad arguments");
  TATT_CHECK_ARG(!A2 || (ksplit > 0 && ksplit < K && ksplit % BK == 0 && al16(A2)), "tatt_gemm_cat: ksplit must be a multiple of %d inside (0, K), A2 16-byte aligned", BK);
  TATT_CHECK_ARG(!C2 || (nsplit > 0 && nsplit < N && nsplit % BN == 0), "tatt_gemm_cat: nsplit must be a multiple of %d inside (0, N)", BN);
  GemmArgs a;
  init_args(a);
  a.A = A; a.B = B; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K;
  a.sam = sam; a.sak = 1; a.sbn = sbn; a.sbk = sbk; a.ldc = ldc;
  a.splitk = 1;
  a.accumulate = accumulate; a.relu = flags & TATT_RELU; a.bf16 = prec_of(flags);
  a.kchunk = cdiv(K, BK) * BK;
  if (A2) { a.A2 = A2; a.ksplit = ksplit; }
  if (C2) { a.C2 = C2; a.nsplit = nsplit; a.ldc2 = ldc2; a.accumulate2 = accumulate2; }
  const bool brow = (sbn == 1 && sbk != 1);
  // the second source is read through the FAST loader only; its extent is checked like the first one's
  const long ka = A2 ? ksplit : K, kb = A2 ? K - ksplit : 0;
  TATT_CHECK_ARG(((long)M * sam + (ka > kb ? ka : kb)) * 4 < GIB2, "tatt_gemm_cat: A operands must stay below 2 GiB");
  if (partial) {
    TATT_CHECK_ARG(brow && bn_x && bn_stat && bn_gamma && bn_beta && bn_C > 0 && bn_C <= (C2 ? nsplit : N),
                   "tatt_gemm_cat_bnbwd: BatchNorm epilogue needs a row-contiguous B (input-gradient GEMM) and the BatchNorm's x / stat / gamma / beta");
    a.epi_partial = partial; a.epi_x = bn_x; a.epi_ldx = bn_ldx; a.epi_stat = bn_stat; a.epi_gamma = bn_gamma; a.epi_beta = bn_beta;
    a.epi_act = bn_act; a.epi_C = bn_C;
    return launch_epi<OP_STRIDED, OP_ROWCONTIG, 2>(a, st);
  }
  if (brow) {
    TATT_CHECK_ARG((fast_ok<OP_STRIDED, OP_ROWCONTIG>(a)), "tatt_gemm_cat: operands must be 16-byte aligned with K, strides multiples of 4");
    return launch<OP_STRIDED, OP_ROWCONTIG>(a, 1, st);
  }
  TATT_CHECK_ARG((fast_ok<OP_STRIDED, OP_STRIDED>(a)), "tatt_gemm_cat: operands must be 16-byte aligned with K, strides multiples of 4");
  return launch<OP_STRIDED, OP_STRIDED>(a, 1, st);
}

// NHWC convolution, stride 1, zero padding (KH/2, KW/2):  Y[p, co] = bias[co] + sum_{tap,ci} X[p+tap, ci] * Wv(co, ci, tap)
// Wv(co,ci,tap) = Wt[co*w_sn + ci*w_sci + w_base + tap*w_stap]:
//   forward with OIHW weights [Cout,Cin,KH,KW]:  w_sn=Cin*T, w_sci=T, w_base=0,   w_stap=1
//   dgrad (X:=dY, Cin:=Cout_f, Cout:=Cin_f):     w_sn=T,     w_sci=Cin_f*T, w_base=T-1, w_stap=-1
int tatt_conv_fwd_splitk(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout, int KH,
                         int KW, long ldx, long ldy, long w_sn, long w_sci, long w_base, long w_stap, int accumulate, int flags,
                         int splitk, float* workspace, hipStream_t st);

int tatt_conv_fwd(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout, int KH,
                  int KW, long ldx, long ldy, long w_sn, long w_sci, long w_base, long w_stap, int accumulate, int flags,
                  hipStream_t st) {
  return tatt_conv_fwd_splitk(X, Wt, bias, Y, B, H, W, Cin, Cout, KH, KW, ldx, ldy, w_sn, w_sci, w_base, w_stap, accumulate, flags, 1,
                              nullptr, st);
}

// Same convolution with the K = KH*KW*Cin reduction split `splitk` ways into workspace slabs [splitk][B*H*W][Cout] that a
// second kernel sums (+bias, +Y, relu): for the small feature maps of the STN head (a handful of output tiles, K up to
// 2304) one workgroup per tile would walk 72 K-tiles in sequence while 250 CUs idle.
int tatt_conv_fwd_splitk(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout, int KH,
                         int KW, long ldx, long ldy, long w_sn, long w_sci, long w_base, long w_stap, int accumulate, int flags,
                         int splitk, float* workspace, hipStream_t st) {
  const int relu = flags & TATT_RELU;
  TATT_CHECK_ARG(splitk >= 1 && (splitk == 1 || workspace), "tatt_conv_fwd_splitk: split-K needs a workspace");
  TATT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "tatt_conv_fwd: bad dims");
  TATT_CHECK_ARG(Cin % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)X) & 15) == 0, "tatt_conv_fwd: Cin/ldx must be multiples of 4, X 16B aligned");
  TATT_CHECK_ARG((KH & 1) && (KW & 1), "tatt_conv_fwd: odd kernels only");
  TATT_CHECK_ARG((long)B * H * W * ldx * 4 < (1L << 31) && (long)Cout * Cin * KH * KW * 4 < (1L << 31), "tatt_conv_fwd: operands must stay below 2 GiB (32-bit buffer offsets)");
  GemmArgs a;
  init_args(a);
  a.A = X; a.B = Wt; a.C = Y; a.bias = bias;
  a.M = B * H * W; a.N = Cout; a.K = KH * KW * Cin;
  a.ldc = ldy;
  a.splitk = splitk;
  a.kchunk = cdiv(cdiv(a.K, splitk), BK) * BK;
  a.accumulate = accumulate; a.relu = relu; a.bf16 = prec_of(flags);
  a.a16 = (flags & TATT_X_BF16) != 0;
  a.g = ConvGeom{H, W, Cin, KH, KW, KH / 2, KW / 2, (int)ldx};
  a.w_sn = w_sn; a.w_sci = w_sci; a.w_base = w_base; a.w_stap = w_stap;
  if (splitk > 1) {
    a.C = workspace;
    a.ldc = Cout;
    a.sCz = (long)a.M * Cout;
  }
  int rc;
  if (w_sci == 1 && (w_sn | w_stap | w_base) % 4 == 0 && (((uintptr_t)Wt) & 15) == 0) rc = launch<OP_IM2COL, OP_CONVW_P>(a, 1, st);
  else rc = launch<OP_IM2COL, OP_CONVW>(a, 1, st);
  if (rc || splitk == 1) return rc;
  const long tot = (long)a.M * Cout;
  launch_splitk_reduce(st, 1, workspace, tot, splitk, Y, bias, a.M, Cout, ldy, accumulate, relu, 0, 0, 0, 0L);
  TATT_LAUNCH_CHECK("conv_splitk_reduce");
  return TATT_OK;
}

// tatt_conv_fwd (packed weights, see tatt_conv_pack_w) with a BatchNorm fused into the epilogue; partial: cdiv(B*H*W, 128) * 2 * Cout floats.
//   epi 1: the convolution is FOLLOWED by a BatchNorm in training mode: per-row-block (mean, M2) of Y -> tatt_bn_stats_from_partial
//   epi 2: the convolution is the input-gradient pass whose result is the output gradient of act(BatchNorm(bn_x)): Y receives
//          g = dy * act'(bn(bn_x)) and partial the sums (sum g, sum g xhat) -> tatt_bn_bwd_from_partial
int tatt_conv_fwd_bn(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                     long ldx, long ldy, long w_sn, long w_sci, long w_base, long w_stap, int flags, int epi, const float* bn_x,
                     long bn_ldx, const float* bn_stat, const float* bn_gamma, const float* bn_beta, int bn_act, float* partial,
                     hipStream_t st) {
  TATT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && partial && (epi == 1 || epi == 2), "tatt_conv_fwd_bn: bad arguments");
  TATT_CHECK_ARG(epi == 1 || (bn_x && bn_stat && bn_gamma && bn_beta), "tatt_conv_fwd_bn: epi 2 needs the BatchNorm's x / stat / gamma / beta");
  TATT_CHECK_ARG(Cin % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)X) & 15) == 0 && (KH & 1) && (KW & 1), "tatt_conv_fwd_bn: Cin/ldx multiples of 4, odd kernels");
  TATT_CHECK_ARG(w_sci == 1 && (w_sn | w_stap | w_base) % 4 == 0 && (((uintptr_t)Wt) & 15) == 0, "tatt_conv_fwd_bn: packed weights required");
  TATT_CHECK_ARG((long)B * H * W * ldx * 4 < (1L << 31) && (long)Cout * Cin * KH * KW * 4 < (1L << 31), "tatt_conv_fwd_bn: operands must stay below 2 GiB");
  GemmArgs a;
  init_args(a);
  a.A = X; a.B = Wt; a.C = Y; a.bias = bias;
  a.M = B * H * W; a.N = Cout; a.K = KH * KW * Cin;
  a.ldc = ldy;
  a.splitk = 1;
  a.kchunk = cdiv(a.K, BK) * BK;
  a.bf16 = prec_of(flags);
  a.a16 = (flags & TATT_X_BF16) != 0;
  a.g = ConvGeom{H, W, Cin, KH, KW, KH / 2, KW / 2, (int)ldx};
  a.w_sn = w_sn; a.w_sci = w_sci; a.w_base = w_base; a.w_stap = w_stap;
  a.epi_partial = partial; a.epi_C = Cout;
  a.epi_x = bn_x; a.epi_ldx = bn_ldx; a.epi_stat = bn_stat; a.epi_gamma = bn_gamma; a.epi_beta = bn_beta; a.epi_act = bn_act;
  return epi == 1 ? launch_epi<OP_IM2COL, OP_CONVW_P, 1>(a, st) : launch_epi<OP_IM2COL, OP_CONVW_P, 2>(a, st);
}

// Workspace floats of tatt_conv_skinny_fwd: planar Z [T*Cout][B*H*W] + repacked weights [T*Cout][Cin].
size_t tatt_conv_skinny_workspace(int B, int H, int W, int Cin, int Cout, int KH, int KW) {
  if (c4_direct_ok(Cin, Cout, KH, KW, 4, nullptr)) return (size_t)KH * KW * Cout * Cin + 64;  // direct kernel: packed weights only
  return (size_t)KH * KW * Cout * ((size_t)B * H * W + Cin) + 64;
}

// Same contract as tatt_conv_fwd (no relu) for Cout in {4, 8, 16}; see the kernels above.
int tatt_conv_skinny_fwd(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout, int KH,
                         int KW, long ldx, long ldy, long w_sn, long w_sci, long w_base, long w_stap, int flags,
                         float* workspace, hipStream_t st) {
  const int accumulate = flags & TATT_ACCUMULATE;
  TATT_CHECK_ARG(Cout == 4 || Cout == 8 || Cout == 16, "tatt_conv_skinny_fwd: Cout must be 4, 8 or 16 (got %d)", Cout);
  TATT_CHECK_ARG(workspace && (KH & 1) && (KW & 1) && B > 0, "tatt_conv_skinny_fwd: bad arguments");
  const int T = KH * KW;
  const long P = (long)B * H * W;
  TATT_CHECK_ARG(P < (1L << 31), "tatt_conv_skinny_fwd: too many pixels");
  // tatt_conv_skinny_workspace() sizes the workspace for the direct kernel whenever the SHAPE is eligible (it cannot see ldx or X): a
  // call of that shape whose operand layout rules the direct kernel out must not fall through to the planar path, which would write
  // T*Cout*P floats into a workspace of T*Cout*Cin
  TATT_CHECK_ARG(!c4_direct_ok(Cin, Cout, KH, KW, 4, nullptr) || c4_direct_ok(Cin, Cout, KH, KW, ldx, X),
                 "tatt_conv_skinny_fwd: a %dx%d convolution with %d output channels needs ldx %% 4 == 0 and a 16-byte aligned X (ldx %ld)",
                 KH, KW, Cout, ldx);
  if (c4_direct_ok(Cin, Cout, KH, KW, ldx, X)) {  // 9x9, four output channels: direct convolution, no planar intermediate
    TATT_CHECK_ARG((long)H * W * ldx * 4 < GIB2, "tatt_conv_skinny_fwd: one image must span less than 2 GiB (32-bit buffer offsets)");
    hipLaunchKernelGGL(conv_c4_pack_w, dim3(cdiv((long)T * Cin * 4, 256)), dim3(256), 0, st, Wt, workspace, Cin, T, w_sn, w_sci, w_base, w_stap);
    TATT_LAUNCH_CHECK("conv_c4_pack_w");
    const int tiles_x = cdiv(W, C4_TW), tiles_y = cdiv(H, C4_TH);
    hipLaunchKernelGGL(conv_c4_direct, dim3(tiles_x * tiles_y * B), dim3(256), 0, st, X, ldx, workspace, bias, Y, ldy, H, W, Cin, tiles_x,
                       tiles_y, accumulate);
    TATT_LAUNCH_CHECK("conv_c4_direct");
    return TATT_OK;
  }
  float* Wr = workspace;
  float* Z = workspace + (((size_t)T * Cout * Cin + 63) / 64) * 64;
  hipLaunchKernelGGL(skinny_repack_w, dim3(cdiv((long)T * Cout * Cin, 256)), dim3(256), 0, st, Wt, Wr, Cin, Cout, T, w_sn, w_sci, w_base,
                     w_stap);
  TATT_LAUNCH_CHECK("skinny_repack_w");
  GemmArgs a;
  init_args(a);
  a.A = Wr; a.B = X; a.C = Z;
  a.M = T * Cout; a.N = (int)P; a.K = Cin;
  a.sam = Cin; a.sak = 1; a.sbn = ldx; a.sbk = 1; a.ldc = P;
  a.splitk = 1;
  a.bf16 = prec_of(flags);
  a.kchunk = cdiv(Cin, BK) * BK;
  int rc = launch<OP_STRIDED, OP_STRIDED>(a, 1, st);
  if (rc) return rc;
  dim3 grid(cdiv(P, 256));
  if (Cout == 4) hipLaunchKernelGGL((skinny_gather<4>), grid, dim3(256), 0, st, Z, bias, Y, B, H, W, KH, KW, ldy, accumulate);
  else if (Cout == 8) hipLaunchKernelGGL((skinny_gather<8>), grid, dim3(256), 0, st, Z, bias, Y, B, H, W, KH, KW, ldy, accumulate);
  else hipLaunchKernelGGL((skinny_gather<16>), grid, dim3(256), 0, st, Z, bias, Y, B, H, W, KH, KW, ldy, accumulate);
  TATT_LAUNCH_CHECK("skinny_gather");
  return TATT_OK;
}

size_t tatt_conv_wgrad_workspace(int Cin, int Cout, int KH, int KW, int splitk) {
  return (size_t)Cout * KH * KW * Cin * (splitk > 0 ? splitk : 1);
}

// dW[co,ci,kh,kw] (+)= sum_p dY[p,co] * X[p+tap, ci]   (OIHW output), split over pixels.
int tatt_conv_wgrad(const float* X, const float* dY, float* dW, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                    long ldx, long lddy, int splitk, float* workspace, int flags, hipStream_t st) {
  const int accumulate = flags & TATT_ACCUMULATE;
  TATT_CHECK_ARG(Cin % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)X) & 15) == 0, "tatt_conv_wgrad: Cin/ldx multiples of 4");
  TATT_CHECK_ARG(splitk >= 1 && workspace, "tatt_conv_wgrad: workspace required");
  TATT_CHECK_ARG((long)B * H * W * ldx * 4 < (1L << 31) && (long)B * H * W * lddy * 4 < (1L << 31), "tatt_conv_wgrad: operands must stay below 2 GiB (32-bit buffer offsets)");
  GemmArgs a;
  init_args(a);
  const int T = KH * KW;
  a.A = dY; a.B = X; a.C = workspace;
  a.M = Cout; a.N = T * Cin; a.K = B * H * W;
  a.sam = 1; a.sak = lddy;
  a.ldc = a.N;
  a.splitk = splitk;
  a.sCz = (long)a.M * a.N;
  a.kchunk = cdiv(cdiv(a.K, splitk), BK) * BK;
  a.bf16 = prec_of(flags);
  a.g = ConvGeom{H, W, Cin, KH, KW, KH / 2, KW / 2, (int)ldx};
  const int x16 = (flags & TATT_X_BF16) != 0, dy16 = (flags & TATT_DY_BF16) != 0;
  TATT_CHECK_ARG(!(x16 || dy16) || (Cout > 16 && lddy % 4 == 0 && (((uintptr_t)dY) & 15) == 0),
                 "tatt_conv_wgrad: bf16-stored operands need more than 16 output channels and a 16-byte granular dY");
  int perm = 1, perm_c = Cin;
  if (Cout <= 16 && Cout % 4 == 0 && lddy % 4 == 0 && (((uintptr_t)dY) & 15) == 0) {
    // tiny Cout (the 9x9 64->4 output conv): swap the roles of X and dY so that the wide tensor sits on the M side and
    // the im2col runs over the 4-channel one:  D[ci][(tap',co)] = sum_q X[q][ci] * dY[q+shift(tap')][co]
    //                                          = dW[co][ci][T-1-tap']   (undone by perm 3 in the reduce kernel)
    // Which tensor goes on the 128-wide M side: with X there (M = Cin) a 64-channel conv fills half of every tile; with the
    // im2col of dY there (M = T*Cout = 324 for the 9x9 output conv: 3 tiles, 16 % padding) and X on N the tiles are full.
    a.g = ConvGeom{H, W, Cout, KH, KW, KH / 2, KW / 2, (int)lddy};
    perm_c = Cout;
    if (Cin <= 64 && Cin % 4 == 0) {
      a.A = dY; a.B = X;
      a.M = T * Cout; a.N = Cin;
      a.sam = 0; a.sak = 0;
      a.sbn = 1; a.sbk = ldx;
      a.ldc = a.N;
      a.sCz = (long)a.M * a.N;
      perm = 4;
      if (int rc = launch<OP_IM2COL_M, OP_ROWCONTIG>(a, 1, st)) return rc;
    } else {
      a.A = X; a.B = dY;
      a.M = Cin; a.N = T * Cout;
      a.sam = 1; a.sak = ldx;
      a.ldc = a.N;
      a.sCz = (long)a.M * a.N;
      perm = 3;
      if (int rc = launch<OP_ROWCONTIG, OP_IM2COL_N>(a, 1, st)) return rc;
    }
  } else if (Cout <= 64 && lddy % 4 == 0 && (((uintptr_t)dY) & 15) == 0) {
    // Cout <= 64 fits ONE 64-wide N tile: put (tap,ci) on the 128-wide M side so that no MFMA rows are padding
    // (with Cout on M a 64-channel conv would fill only half of every 128-row tile).
    a.A = X; a.B = dY;
    a.M = T * Cin; a.N = Cout;
    a.sam = 0; a.sak = 0;
    a.sbn = 1; a.sbk = lddy;
    a.ldc = a.N;
    a.sCz = (long)a.M * a.N;
    perm = 2;
    a.a16 = x16; a.b16 = dy16;
    if (g_wgrad192 && a.bf16 == 0 && a.M % WG_BM == 0 && fast_ok<OP_IM2COL_M, OP_ROWCONTIG>(a)) {
      hipLaunchKernelGGL(wgrad192_kernel, dim3(a.M / WG_BM, 1, splitk), dim3(WG_THR), 0, st, a);
      TATT_LAUNCH_CHECK("wgrad192_kernel");
    } else if (int rc = launch<OP_IM2COL_M, OP_ROWCONTIG>(a, 1, st)) return rc;
  } else {
    // the kernel treats splitk>1 as "write raw slabs"; with splitk==1 it applies bias/accumulate (none set): same result.
    a.a16 = dy16; a.b16 = x16;
    if (int rc = launch<OP_ROWCONTIG, OP_IM2COL_N>(a, 1, st)) return rc;
  }
  long tot = (long)a.M * a.N;
  launch_splitk_reduce(st, 1, workspace, tot, splitk, dW, (const float*)nullptr, a.M, a.N, (long)a.N, accumulate, 0, perm, perm_c, T, 0L);
  TATT_LAUNCH_CHECK("conv_wgrad_reduce");
  return TATT_OK;
}

}  // extern "C"
