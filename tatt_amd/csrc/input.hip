// Input transform of the reference's collate function on the device:
//   resizeNormalize.__call__ (dataset/dataset.py:282-325) = PIL resize(BICUBIC) -> ToTensor -> gray-mean mask channel,
//   applied per sample by alignCollate.__call__ (:677-693).
// Byte/integer work, bit-exact with Pillow: separable two-pass resampling (horizontal, then vertical) with the support-scaled
// bicubic (a = -0.5), coefficients normalised in double and quantised to 22 fractional bits, 8-bit intermediate image,
// L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16.  One workgroup per image; the intermediate and the result live in LDS.
#include "common.h"

namespace {

constexpr int PREC = 32 - 8 - 2;

__device__ __forceinline__ double bicubic_pil(double x) {
#pragma clang fp contract(off)
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// coefficients of one output position (Resample.c precompute_coeffs + normalize_coeffs_8bpc), evaluated in double exactly
// in the order Pillow evaluates them (no fused multiply-add)
__device__ void coeffs_for(int xx, int in_size, int out_size, int ks, int* __restrict__ kk, int* __restrict__ bnd) {
#pragma clang fp contract(off)
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = (xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > ks) xmax = ks;                                  // cannot happen for ks = ceil(support) * 2 + 1; keeps LDS safe
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += bicubic_pil((x + xmin - center + 0.5) * ss);
  for (int x = 0; x < xmax; ++x) {
    double v = bicubic_pil((x + xmin - center + 0.5) * ss);
    if (ww != 0.0) v /= ww;
    kk[xx * ks + x] = v < 0 ? (int)(-0.5 + v * (double)(1 << PREC)) : (int)(0.5 + v * (double)(1 << PREC));
  }
  bnd[2 * xx] = xmin;
  bnd[2 * xx + 1] = xmax;
}

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PREC;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void resize_normalize_k(const unsigned char* __restrict__ data, const long* __restrict__ offs,
                                                          const int* __restrict__ hs, const int* __restrict__ ws,
                                                          float* __restrict__ out, int OH, int OW, int mask, int max_h, int max_w,
                                                          int ksh, int ksv) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int h = hs[b], w = ws[b];
  const int CO = mask ? 4 : 3;
  float* o = out + (long)b * CO * OH * OW;
  if (h < 1 || w < 1 || h > max_h || w > max_w) {            // sizes the launch was not dimensioned for: poison, do not guess
    for (int i = tid; i < CO * OH * OW; i += nt) o[i] = __builtin_nanf("");
    return;
  }
  const int nco = max(OW * ksh, OH * ksv);
  int* kk = reinterpret_cast<int*>(smem);
  int* bnd = kk + nco;
  unsigned char* res = reinterpret_cast<unsigned char*>(bnd + 2 * max(OW, OH));      // [OH][OW][3]
  unsigned char* hbuf = res + ((OH * OW * 3 + 15) & ~15);                              // [h][OW][3]
  __shared__ int red[4];
  const unsigned char* src = data + offs[b];

  // horizontal pass (skipped, as in Pillow, when the width is unchanged)
  if (w != OW) {
    for (int xx = tid; xx < OW; xx += nt) coeffs_for(xx, w, OW, ksh, kk, bnd);
    __syncthreads();
    for (int i = tid; i < h * OW * 3; i += nt) {
      const int c = i % 3, xx = (i / 3) % OW, y = i / (3 * OW);
      const int xmin = bnd[2 * xx], xmax = bnd[2 * xx + 1];
      const unsigned char* p = src + ((long)y * w + xmin) * 3 + c;
      const int* k = kk + xx * ksh;
      int acc = 1 << (PREC - 1);
      for (int x = 0; x < xmax; ++x) acc += (int)p[3 * x] * k[x];
      hbuf[i] = clip8(acc);
    }
  } else {
    for (int i = tid; i < h * OW * 3; i += nt) hbuf[i] = src[i];
  }
  __syncthreads();
  // vertical pass
  if (h != OH) {
    for (int yy = tid; yy < OH; yy += nt) coeffs_for(yy, h, OH, ksv, kk, bnd);
    __syncthreads();
    for (int i = tid; i < OH * OW * 3; i += nt) {
      const int xc = i % (3 * OW), yy = i / (3 * OW);
      const int ymin = bnd[2 * yy], ymax = bnd[2 * yy + 1];
      const unsigned char* p = hbuf + ymin * OW * 3 + xc;
      const int* k = kk + yy * ksv;
      int acc = 1 << (PREC - 1);
      for (int y = 0; y < ymax; ++y) acc += (int)p[y * OW * 3] * k[y];
      res[i] = clip8(acc);
    }
  } else {
    for (int i = tid; i < OH * OW * 3; i += nt) res[i] = hbuf[i];
  }
  __syncthreads();
  // ToTensor + the mask channel: 1.0 where L <= mean(L) (dataset.py:318-323), compared exactly as L * N > sum(L)
  int lsum = 0;
  for (int i = tid; i < OH * OW; i += nt) {
    const int L = (res[3 * i] * 19595 + res[3 * i + 1] * 38470 + res[3 * i + 2] * 7471 + 0x8000) >> 16;
    lsum += L;
    o[i] = (float)res[3 * i] / 255.0f;
    o[OH * OW + i] = (float)res[3 * i + 1] / 255.0f;
    o[2 * OH * OW + i] = (float)res[3 * i + 2] / 255.0f;
  }
  if (!mask) return;
  for (int ofs = 32; ofs; ofs >>= 1) lsum += __shfl_xor(lsum, ofs);
  if ((tid & 63) == 0) red[tid >> 6] = lsum;
  __syncthreads();
  const long tot = (long)red[0] + red[1] + red[2] + red[3];
  for (int i = tid; i < OH * OW; i += nt) {
    const long L = (res[3 * i] * 19595 + res[3 * i + 1] * 38470 + res[3 * i + 2] * 7471 + 0x8000) >> 16;
    o[3 * OH * OW + i] = (L * (long)(OH * OW) > tot) ? 0.0f : 1.0f;
  }
}

int ksize_for(int in_max, int out) {
  const double scale = (double)in_max / (double)out;
  const double support = 2.0 * (scale < 1.0 ? 1.0 : scale);
  return (int)ceil(support) * 2 + 1;
}

}  // namespace

extern "C" {

size_t tatt_resize_normalize_lds(int OH, int OW, int max_h, int max_w) {
  const int ksh = ksize_for(max_w, OW), ksv = ksize_for(max_h, OH);
  const size_t nco = (size_t)((OW * ksh > OH * ksv) ? OW * ksh : OH * ksv);
  return nco * 4 + 2 * (size_t)(OW > OH ? OW : OH) * 4 + (((size_t)OH * OW * 3 + 15) & ~(size_t)15) + (size_t)max_h * OW * 3;
}

int tatt_resize_normalize(const unsigned char* data, const long* offsets, const int* heights, const int* widths, float* out, int B,
                          int OH, int OW, int mask, int max_h, int max_w, hipStream_t st) {
  TATT_CHECK_ARG(data && offsets && heights && widths && out && B > 0 && OH > 0 && OW > 0 && max_h > 0 && max_w > 0,
                 "tatt_resize_normalize: bad arguments");
  const size_t lds = tatt_resize_normalize_lds(OH, OW, max_h, max_w);
  TATT_CHECK_ARG(lds <= 160 * 1024 - 64, "tatt_resize_normalize: image too large for the LDS-resident transform "
                                          "(max_h * OW * 3 + coefficient tables must fit 160 KB)");
  const int ksh = ksize_for(max_w, OW), ksv = ksize_for(max_h, OH);
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(resize_normalize_k),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess) {
    tatt_set_error("tatt_resize_normalize: cannot raise the dynamic LDS limit");
    return TATT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(resize_normalize_k, dim3(B), dim3(256), lds, st, data, offsets, heights, widths, out, OH, OW, mask, max_h, max_w,
                     ksh, ksv);
  TATT_LAUNCH_CHECK("resize_normalize");
  return TATT_OK;
}

}  // extern "C"
