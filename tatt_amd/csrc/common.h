// Shared helpers for the tatt_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define TATT_OK 0
#define TATT_ERR_ARG 1
#define TATT_ERR_LAUNCH 2

// bits of the `flags` argument of the GEMM / convolution entry points (include/tatt_hip.h)
#define TATT_RELU 1        // tatt_gemm, tatt_conv_fwd*: ReLU epilogue
#define TATT_ACCUMULATE 1  // tatt_conv_wgrad, tatt_conv_skinny_fwd: add to the destination
#define TATT_F32X3 4       // fp32 products from three bf16 pieces per operand (six v_mfma_f32_32x32x16_bf16 terms, fp32 accumulate); TATT_BF16 wins if both are set
#define TATT_BF16 2        // operands rounded to bf16 (nearest-even) into v_mfma_f32_32x32x16_bf16, fp32 accumulate
#define TATT_X_BF16 8      // convolution entry points, with TATT_BF16: the input X (tatt_conv_wgrad: X) is STORED as bf16 (pointer to 2-byte elements)
#define TATT_DY_BF16 16    // tatt_conv_wgrad, with TATT_BF16: dY is stored as bf16

void tatt_set_error(const char* fmt, ...);

#define TATT_CHECK_ARG(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      tatt_set_error(__VA_ARGS__);                \
      return TATT_ERR_ARG;                        \
    }                                             \
  } while (0)

#define TATT_LAUNCH_CHECK(name)                                              \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      tatt_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return TATT_ERR_LAUNCH;                                                \
    }                                                                        \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- counter-based dropout (bit-identical to oracle/tatt_ref.py:hash_u32) -------------
__device__ __forceinline__ uint32_t tatt_hash_u32(uint32_t seed_site, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u + seed_site;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
// host+device: fold (seed, site) like the oracle: (seed*0x85EBCA77 + site*0xC2B2AE3D) mod 2^32
static inline __host__ __device__ uint32_t tatt_seed_site(uint32_t seed, uint32_t site) {
  return seed * 0x85EBCA77u + site * 0xC2B2AE3Du;
}
// returns the multiplicative mask (0 or 1/(1-p)); thresh = (int)(p * 2^24); thresh==0 -> 1
__device__ __forceinline__ float tatt_drop_mask(uint32_t seed_site, uint32_t idx, uint32_t thresh, float scale) {
  if (thresh == 0) return 1.0f;
  return ((tatt_hash_u32(seed_site, idx) >> 8) >= thresh) ? scale : 0.0f;
}

struct DropCfg {       // passed by value to kernels
  uint32_t seed_site;  // tatt_seed_site(seed, site)
  uint32_t thresh;     // 0 = dropout off
  float scale;         // 1/(1-p)
  const uint32_t* seed_dev;  // optional device-resident seed increment (lets a captured hipGraph draw new masks per replay)
};
// effective (seed, site) word: seed_site + seed_dev[0]*0x85EBCA77  ==  tatt_seed_site(seed + seed_dev[0], site)
__device__ __forceinline__ uint32_t tatt_eff_seed(const DropCfg& dc) {
  return dc.seed_dev ? dc.seed_site + dc.seed_dev[0] * 0x85EBCA77u : dc.seed_site;
}

// Activations on the hardware exponential (v_exp_f32, ~1 ulp): one exp each instead of libm's tanhf/log1pf chains.
//   tanh(softplus(x)) = (n^2 + 2n) / (n^2 + 2n + 2) with n = e^x   (exact identity; saturates to 1 for large x)
// Divisions are v_rcp_f32 (1 ulp) + a multiply: an IEEE fp32 division is a ten-instruction sequence, and these sit on the GRU's
// per-step latency chain and in the BatchNorm-backward epilogue of the convolutions (16 mish' per lane).  No branches: the
// exponent argument is clamped where e^x would overflow the rational forms (the result has saturated long before).
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * rcpf_(__expf(2.0f * x) + 1.0f); }
__device__ __forceinline__ float tanh_softplus_(float x) {
  const float n = __expf(fminf(x, 20.0f));
  const float t = n * (n + 2.0f);
  return t * rcpf_(t + 2.0f);
}
__device__ __forceinline__ float mishf_(float x) { return x * tanh_softplus_(x); }
__device__ __forceinline__ float mish_gradf_(float x) {  // one exponential for both factors: sigmoid(x) = n / (1 + n)
  const float n = __expf(fminf(x, 20.0f));
  const float u = n * (n + 2.0f);
  const float t = u * rcpf_(u + 2.0f);
  const float s = n * rcpf_(1.0f + n);
  return t + x * (1.0f - t * t) * s;
}

// Second stage of the two-stage column reductions: a 256-thread block sums partial[b*stride_b + c] over b for
// 16 columns (c = blockIdx.x*16 + (tid&15)) with 16 row lanes (tid>>4); the result is valid in the threads with
// tid < 16.  red: 256 floats of LDS.  Keeps the reduction deterministic and ~16x shorter than one thread per column.
__device__ __forceinline__ float block16_colsum(const float* __restrict__ partial, int nblk, long stride_b, int c, bool cvalid,
                                                float* red) {
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  float s = 0.f;
  if (cvalid) {  // four independent chains: the loads of a trip are in flight together (one chain = nblk/16 dependent L2 trips)
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = rl;
    for (; b + 48 < nblk; b += 64) {
      s += partial[(long)b * stride_b + c];
      s1 += partial[(long)(b + 16) * stride_b + c];
      s2 += partial[(long)(b + 32) * stride_b + c];
      s3 += partial[(long)(b + 48) * stride_b + c];
    }
    for (; b < nblk; b += 16) s += partial[(long)b * stride_b + c];
    s = (s + s1) + (s2 + s3);
  }
  red[rl * 16 + cl] = s;
  __syncthreads();
  float t = 0.f;
  if (rl == 0)
    for (int j = 0; j < 16; ++j) t += red[j * 16 + cl];
  __syncthreads();
  return t;
}

// four fp32 values -> four bf16 (round to nearest even), 8 bytes: the store of a tensor that is kept in bf16 (bf16 step, see engine.py)
__device__ __forceinline__ uint2 tatt_pack_bf16x4(float a, float b, float c, float d) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ lo = {a, b}, hi = {c, d};
  const bf16x2_ l = __builtin_convertvector(lo, bf16x2_), h = __builtin_convertvector(hi, bf16x2_);
  return make_uint2(*reinterpret_cast<const unsigned*>(&l), *reinterpret_cast<const unsigned*>(&h));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
