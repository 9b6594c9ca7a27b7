// Structural-similarity kernels: TRI_SSIM (the TSC loss of the reference, utils/ssim_psnr.py:99-128, :229-254) forward and
// backward, and the two-image SSIM / PSNR evaluation metrics (utils/ssim_psnr.py:9-15, :76-96).
// Layout NCHW fp32, one workgroup per (image, channel) plane; the three planes (and, in the backward pass, the four
// per-pixel derivative fields) live in LDS, every thread walks the 11x11 Gaussian window directly (zero padding, like
// F.conv2d(padding=5, groups=C)).  A plane is 32x128 = 16 KB, so the pass is a few hundred MFLOP of VALU work per step:
// no tiling tricks are needed, the point is to keep the nine blurred maps out of HBM.
#include "common.h"
#include <mutex>

namespace {

constexpr int WIN = 11, HALF = 5, MAXHW = 4096;
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

struct Gauss {
  float g[WIN];
};

// gaussian(11, 1.5) as the reference builds it: float32 values of exp(), float32 sum, float32 division; the 2-D window is
// the float32 outer product g[i]*g[j].
Gauss make_gauss() {
  Gauss G;
  float sum = 0.f;
  for (int i = 0; i < WIN; ++i) {
    G.g[i] = (float)exp(-(double)((i - HALF) * (i - HALF)) / (2.0 * 1.5 * 1.5));
    sum += G.g[i];
  }
  for (int i = 0; i < WIN; ++i) G.g[i] /= sum;
  return G;
}

struct Blur9 {
  float m1, m2, m3, e11, e22, e33, e12, e23, e31;
};

__device__ __forceinline__ Blur9 blur9(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                       const Gauss& G, int py, int px, int H, int W) {
  Blur9 r = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int dy = 0; dy < WIN; ++dy) {
    const int y = py + dy - HALF;
    if (y < 0 || y >= H) continue;
#pragma unroll
    for (int dx = 0; dx < WIN; ++dx) {
      const int x = px + dx - HALF;
      if (x < 0 || x >= W) continue;
      const float w = G.g[dy] * G.g[dx];
      const float va = a[y * W + x], vb = b[y * W + x], vc = c[y * W + x];
      r.m1 += w * va;
      r.m2 += w * vb;
      r.m3 += w * vc;
      r.e11 += w * (va * va);
      r.e22 += w * (vb * vb);
      r.e33 += w * (vc * vc);
      r.e12 += w * (va * vb);
      r.e23 += w * (vb * vc);
      r.e31 += w * (vc * va);
    }
  }
  return r;
}

struct Tri {
  float A, Bq, C, D, S;
};
__device__ __forceinline__ Tri tri_terms(const Blur9& r) {
  Tri t;
  t.A = r.m1 * r.m2 + r.m2 * r.m3 + r.m3 * r.m1 + C1;
  t.Bq = (r.e12 - r.m1 * r.m2) + (r.e23 - r.m2 * r.m3) + (r.e31 - r.m3 * r.m1) + C2;
  t.C = r.m1 * r.m1 + r.m2 * r.m2 + r.m3 * r.m3 + C1;
  t.D = (r.e11 - r.m1 * r.m1) + (r.e22 - r.m2 * r.m2) + (r.e33 - r.m3 * r.m3) + C2;
  t.S = (t.A * t.Bq) / (t.C * t.D);
  return t;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// partial[plane] = sum over the plane of the tri-SSIM map
__global__ __launch_bounds__(256) void tri_ssim_fwd_k(const float* __restrict__ x1, const float* __restrict__ x2,
                                                      const float* __restrict__ x3, float* __restrict__ partial, int H, int W,
                                                      Gauss G) {
  __shared__ float a[MAXHW], b[MAXHW], c[MAXHW];
  __shared__ float red[4];
  const int HW = H * W;
  const long base = (long)blockIdx.x * HW;
  for (int i = threadIdx.x; i < HW; i += 256) {
    a[i] = x1[base + i];
    b[i] = x2[base + i];
    c[i] = x3[base + i];
  }
  __syncthreads();
  float s = 0.f;
  for (int p = threadIdx.x; p < HW; p += 256) s += tri_terms(blur9(a, b, c, G, p / W, p % W, H, W)).S;
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// two-image SSIM map sum (utils/ssim_psnr.py:76-96): ((2 m1 m2 + C1)(2 s12 + C2)) / ((m1^2 + m2^2 + C1)(s1 + s2 + C2))
__global__ __launch_bounds__(256) void ssim_fwd_k(const float* __restrict__ x1, const float* __restrict__ x2,
                                                  float* __restrict__ partial, int Ctot, int Cuse, int H, int W, Gauss G) {
  __shared__ float a[MAXHW], b[MAXHW];
  __shared__ float red[4];
  const int HW = H * W;
  const int img = blockIdx.x / Cuse, ch = blockIdx.x % Cuse;
  const long base = ((long)img * Ctot + ch) * HW;
  for (int i = threadIdx.x; i < HW; i += 256) {
    a[i] = x1[base + i];
    b[i] = x2[base + i];
  }
  __syncthreads();
  float s = 0.f;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const Blur9 r = blur9(a, b, b, G, p / W, p % W, H, W);
    const float s1 = r.e11 - r.m1 * r.m1, s2 = r.e22 - r.m2 * r.m2, s12 = r.e12 - r.m1 * r.m2;
    s += ((2.f * r.m1 * r.m2 + C1) * (2.f * s12 + C2)) / ((r.m1 * r.m1 + r.m2 * r.m2 + C1) * (s1 + s2 + C2));
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// partial[plane] = sum of (255 a - 255 b)^2 over the plane (calculate_psnr, utils/ssim_psnr.py:9-15, weighted_mask = 1)
__global__ __launch_bounds__(256) void sqdiff255_k(const float* __restrict__ x1, const float* __restrict__ x2,
                                                   float* __restrict__ partial, int Ctot, int Cuse, int HW) {
  __shared__ float red[4];
  const int img = blockIdx.x / Cuse, ch = blockIdx.x % Cuse;
  const long base = ((long)img * Ctot + ch) * HW;
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float d = x1[base + i] * 255.f - x2[base + i] * 255.f;
    s += d * d;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// out[0] = f(scale * sum(partial)); mode 0: identity, mode 1: PSNR = 20 log10(255 / sqrt(mse))
__global__ void ssim_final_k(const float* __restrict__ partial, int n, float scale, float* __restrict__ out, int mode) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  s = wave_sum(s);
  if (threadIdx.x != 0) return;
  s *= scale;
  out[0] = mode == 1 ? 20.f * log10f(255.f / sqrtf(s)) : s;
}

// dx1, dx2 (+)= scale * d(sum of the tri-SSIM map)/d(x1, x2).  With A, B, C, D as in tri_terms and S = AB/(CD):
//   dS/dmu1 = (mu2+mu3) (B - A)/(CD) - 2 mu1 S (1/C - 1/D),  dS/dE12 = dS/dE31 = A/(CD),  dS/dE11 = -S/D   (same for image 2)
// and every blurred quantity is a symmetric-window correlation, so
//   dx1 = blur(dS/dmu1) + 2 x1 blur(dS/dE11) + (x2 + x3) blur(dS/dE12).
__global__ __launch_bounds__(256) void tri_ssim_bwd_k(const float* __restrict__ x1, const float* __restrict__ x2,
                                                      const float* __restrict__ x3, float scale, float* __restrict__ dx1,
                                                      float* __restrict__ dx2, int accumulate, int H, int W, Gauss G) {
  __shared__ float a[MAXHW], b[MAXHW], c[MAXHW], f1[MAXHW], f2[MAXHW], fB[MAXHW], fD[MAXHW];
  const int HW = H * W;
  const long base = (long)blockIdx.x * HW;
  for (int i = threadIdx.x; i < HW; i += 256) {
    a[i] = x1[base + i];
    b[i] = x2[base + i];
    c[i] = x3[base + i];
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += 256) {
    const Blur9 r = blur9(a, b, c, G, p / W, p % W, H, W);
    const Tri t = tri_terms(r);
    const float icd = 1.f / (t.C * t.D);
    const float gA = t.Bq * icd, gB = t.A * icd, gC = -t.S / t.C, gD = -t.S / t.D;
    f1[p] = (r.m2 + r.m3) * (gA - gB) + 2.f * r.m1 * (gC - gD);
    f2[p] = (r.m1 + r.m3) * (gA - gB) + 2.f * r.m2 * (gC - gD);
    fB[p] = gB;
    fD[p] = gD;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < HW; q += 256) {
    const int py = q / W, px = q % W;
    float t1 = 0.f, t2 = 0.f, tB = 0.f, tD = 0.f;
    for (int dy = 0; dy < WIN; ++dy) {
      const int y = py + dy - HALF;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int dx = 0; dx < WIN; ++dx) {
        const int x = px + dx - HALF;
        if (x < 0 || x >= W) continue;
        const float w = G.g[dy] * G.g[dx];
        const int p = y * W + x;
        t1 += w * f1[p];
        t2 += w * f2[p];
        tB += w * fB[p];
        tD += w * fD[p];
      }
    }
    const float g1 = scale * (t1 + 2.f * a[q] * tD + (b[q] + c[q]) * tB);
    const float g2 = scale * (t2 + 2.f * b[q] * tD + (a[q] + c[q]) * tB);
    dx1[base + q] = accumulate ? dx1[base + q] + g1 : g1;
    dx2[base + q] = accumulate ? dx2[base + q] + g2 : g2;
  }
}

// ---- TRI_SSIM with the Gaussian window applied as two 1-D passes (end of round 3) -------------------------------------------------
// The 11 x 11 window is the outer product g g^T and the padding is zeros, so blur2d = vertical(horizontal(.)): 22 taps per blurred
// quantity instead of 121.  The direct kernels above walk the window per pixel with one four-wave workgroup per plane (256 planes at
// B = 64: one wave per SIMD) and cost 190 us (forward) + 331 us (backward) of the config-3 step, in series between its forward and
// backward passes.  Here a 1024-thread workgroup owns a plane: the three input planes stay in LDS, the nine products
// (a, b, c, aa, bb, cc, ab, bc, ca) are blurred along x for a strip of rows (+ 5 halo rows each side; two strips for a 32 x 128
// plane: 48 KB of inputs + 95 KB of row-blurred maps) and then along y.  The backward pass keeps the four derivative fields of its
// pixels in registers across the strips, writes them over the dead arena, and blurs them the same way.  Shapes whose strips do not
// fit (or, in the backward pass, need more than two strips) take the direct kernels.  Sums are reassociated (rounding level).
constexpr int SEP_THR = 1024, SEP_ARENA = 37888;  // floats of LDS: 151,552 bytes
constexpr int SEP_NS = 2, SEP_KPS = 2;            // backward: at most two strips of at most SEP_KPS * SEP_THR pixels each

struct SepPlan {
  int ns, SH, rows;  // strips per plane, output rows per strip, most row-blurred rows a strip needs (plane stride of the maps)
};

bool sep_plan(int H, int W, SepPlan& pl) {
  const int HW = H * W;
  if (HW > MAXHW || getenv("TATT_SSIM_DIRECT")) return false;
  const int budget = SEP_ARENA - 3 * HW;
  for (int n = 1; n <= H; ++n) {
    const int SH = (H + n - 1) / n;
    int rows = 0;
    for (int y0 = 0; y0 < H; y0 += SH) {
      const int lo = y0 - HALF > 0 ? y0 - HALF : 0, hi = y0 + SH + HALF < H ? y0 + SH + HALF : H;
      if (hi - lo > rows) rows = hi - lo;
    }
    if (9 * rows * W <= budget) {
      pl = SepPlan{(H + SH - 1) / SH, SH, rows};
      return true;
    }
  }
  return false;
}

// hq[q][r][x] = sum_dx g[dx] Q_q[lo + r][x + dx - 5] for the nr rows from lo on (zero padding: out-of-range taps get weight 0)
__device__ __forceinline__ void sep_hpass9(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                           float* __restrict__ hq, int lo, int nr, int rows, int W, const Gauss& G) {
  const int n = nr * W, ps = rows * W;
  for (int i = threadIdx.x; i < n; i += SEP_THR) {
    const int r = i / W, x = i - r * W;
    const float* ra = a + (lo + r) * W;
    const float* rb = b + (lo + r) * W;
    const float* rc = c + (lo + r) * W;
    float s[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) s[q] = 0.f;
#pragma unroll
    for (int dx = 0; dx < WIN; ++dx) {
      const int xx = x + dx - HALF;
      const bool ok = (unsigned)xx < (unsigned)W;
      const int xc = ok ? xx : x;
      const float w = ok ? G.g[dx] : 0.f;
      const float va = ra[xc], vb = rb[xc], vc = rc[xc];
      const float wa = w * va, wb = w * vb, wc = w * vc;
      s[0] += wa; s[1] += wb; s[2] += wc;
      s[3] += wa * va; s[4] += wb * vb; s[5] += wc * vc;
      s[6] += wa * vb; s[7] += wb * vc; s[8] += wc * va;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) hq[q * ps + i] = s[q];
  }
}

// the nine blurred quantities of pixel (y, x) from the row-blurred maps of the strip that starts at image row lo
__device__ __forceinline__ Blur9 sep_vpass9(const float* __restrict__ hq, int lo, int rows, int y, int x, int H, int W, const Gauss& G) {
  const int ps = rows * W;
  float s[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) s[q] = 0.f;
#pragma unroll
  for (int dy = 0; dy < WIN; ++dy) {
    const int yy = y + dy - HALF;
    const bool ok = (unsigned)yy < (unsigned)H;
    const float w = ok ? G.g[dy] : 0.f;
    const float* src = hq + ((ok ? yy : y) - lo) * W + x;
#pragma unroll
    for (int q = 0; q < 9; ++q) s[q] += w * src[q * ps];
  }
  return Blur9{s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]};
}

__global__ __launch_bounds__(SEP_THR) void tri_ssim_fwd_sep_k(const float* __restrict__ x1, const float* __restrict__ x2,
                                                              const float* __restrict__ x3, float* __restrict__ partial, int H, int W,
                                                              int SH, int rows, Gauss G) {
  __shared__ float arena[SEP_ARENA];
  __shared__ float red[SEP_THR / 64];
  const int HW = H * W;
  const long base = (long)blockIdx.x * HW;
  float* a = arena;
  float* b = a + HW;
  float* c = b + HW;
  float* hq = c + HW;
  for (int i = threadIdx.x; i < HW; i += SEP_THR) {
    a[i] = x1[base + i];
    b[i] = x2[base + i];
    c[i] = x3[base + i];
  }
  __syncthreads();
  float s = 0.f;
  for (int y0 = 0; y0 < H; y0 += SH) {
    const int lo = max(0, y0 - HALF), hi = min(H, y0 + SH + HALF);
    sep_hpass9(a, b, c, hq, lo, hi - lo, rows, W, G);
    __syncthreads();
    const int n = (min(H, y0 + SH) - y0) * W;
    for (int i = threadIdx.x; i < n; i += SEP_THR) {
      const int yr = i / W;
      s += tri_terms(sep_vpass9(hq, lo, rows, y0 + yr, i - yr * W, H, W, G)).S;
    }
    __syncthreads();  // the next strip overwrites the maps
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SEP_THR / 64; ++w) t += red[w];
    partial[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(SEP_THR) void tri_ssim_bwd_sep_k(const float* __restrict__ x1, const float* __restrict__ x2,
                                                              const float* __restrict__ x3, float scale, float* __restrict__ dx1,
                                                              float* __restrict__ dx2, int accumulate, int H, int W, int SH, int rows,
                                                              Gauss G) {
  __shared__ float arena[SEP_ARENA];
  const int HW = H * W;
  const long base = (long)blockIdx.x * HW;
  {
    float* a = arena;
    float* b = a + HW;
    float* c = b + HW;
    for (int i = threadIdx.x; i < HW; i += SEP_THR) {
      a[i] = x1[base + i];
      b[i] = x2[base + i];
      c[i] = x3[base + i];
    }
  }
  __syncthreads();
  // phase 1: the four derivative fields of this thread's pixels (dS/dmu1, dS/dmu2, dS/dE12 = dS/dE31 ..., dS/dE11 ...), strip by strip
  float fv[SEP_NS][SEP_KPS][4];
#pragma unroll
  for (int st = 0; st < SEP_NS; ++st) {
    const int y0 = st * SH;
    if (y0 < H) {  // block-uniform
      const float* a = arena;
      const float* b = a + HW;
      const float* c = b + HW;
      float* hq = arena + 3 * HW;
      const int lo = max(0, y0 - HALF), hi = min(H, y0 + SH + HALF);
      sep_hpass9(a, b, c, hq, lo, hi - lo, rows, W, G);
      __syncthreads();
      const int n = (min(H, y0 + SH) - y0) * W;
#pragma unroll
      for (int k = 0; k < SEP_KPS; ++k) {
        const int i = threadIdx.x + k * SEP_THR;
        if (i < n) {
          const int yr = i / W;
          const Blur9 r = sep_vpass9(hq, lo, rows, y0 + yr, i - yr * W, H, W, G);
          const Tri t = tri_terms(r);
          const float icd = 1.f / (t.C * t.D);
          const float gA = t.Bq * icd, gB = t.A * icd, gC = -t.S / t.C, gD = -t.S / t.D;
          fv[st][k][0] = (r.m2 + r.m3) * (gA - gB) + 2.f * r.m1 * (gC - gD);
          fv[st][k][1] = (r.m1 + r.m3) * (gA - gB) + 2.f * r.m2 * (gC - gD);
          fv[st][k][2] = gB;
          fv[st][k][3] = gD;
        }
      }
      __syncthreads();
    }
  }
  // phase 2: the fields into LDS over the dead arena, blurred along x, then along y and combined with the inputs
  float* f = arena;            // [4][HW]
  float* hf = arena + 4 * HW;  // [4][HW]
#pragma unroll
  for (int st = 0; st < SEP_NS; ++st) {
    const int y0 = st * SH;
    if (y0 < H) {
      const int n = (min(H, y0 + SH) - y0) * W;
#pragma unroll
      for (int k = 0; k < SEP_KPS; ++k) {
        const int i = threadIdx.x + k * SEP_THR;
        if (i < n) {
#pragma unroll
          for (int q = 0; q < 4; ++q) f[q * HW + y0 * W + i] = fv[st][k][q];
        }
      }
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += SEP_THR) {
    const int y = p / W, x = p - y * W;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dx = 0; dx < WIN; ++dx) {
      const int xx = x + dx - HALF;
      const bool ok = (unsigned)xx < (unsigned)W;
      const float w = ok ? G.g[dx] : 0.f;
      const float* src = f + y * W + (ok ? xx : x);
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] += w * src[q * HW];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) hf[q * HW + p] = t[q];
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += SEP_THR) {
    const int y = p / W, x = p - y * W;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < WIN; ++dy) {
      const int yy = y + dy - HALF;
      const bool ok = (unsigned)yy < (unsigned)H;
      const float w = ok ? G.g[dy] : 0.f;
      const float* src = hf + (ok ? yy : y) * W + x;
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] += w * src[q * HW];
    }
    const float va = x1[base + p], vb = x2[base + p], vc = x3[base + p];
    const float g1 = scale * (t[0] + 2.f * va * t[3] + (vb + vc) * t[2]);
    const float g2 = scale * (t[1] + 2.f * vb * t[3] + (va + vc) * t[2]);
    dx1[base + p] = accumulate ? dx1[base + p] + g1 : g1;
    dx2[base + p] = accumulate ? dx2[base + p] + g2 : g2;
  }
}

// per-plane sums of the tri-SSIM map into partial[planes]
void launch_tri_fwd(const float* x1, const float* x2, const float* x3, float* partial, int planes, int H, int W, const Gauss& G,
                    hipStream_t st) {
  SepPlan pl;
  if (sep_plan(H, W, pl))
    hipLaunchKernelGGL(tri_ssim_fwd_sep_k, dim3(planes), dim3(SEP_THR), 0, st, x1, x2, x3, partial, H, W, pl.SH, pl.rows, G);
  else
    hipLaunchKernelGGL(tri_ssim_fwd_k, dim3(planes), dim3(256), 0, st, x1, x2, x3, partial, H, W, G);
}

// ---- scratch of tatt_tri_ssim_fwd_static: the per-plane partial sums that tatt_tri_ssim_fwd takes from its `workspace`
// argument, held in the code object instead so that the entry point allocates nothing.  64 KB per GPU the library runs on, for
// the life of the process.  In host code `scratch` is only a stand-in for the symbol: the device address is resolved once per
// device with hipGetSymbolAddress and kept in static_ptr[]; host code passes THAT to the kernels.
constexpr int STATIC_MAX_PLANES = 16384;
constexpr int STATIC_MAX_DEVICES = 64;
static __device__ __align__(256) float scratch[STATIC_MAX_PLANES];
float* static_ptr[STATIC_MAX_DEVICES] = {nullptr};
std::mutex static_mu;

// device address of this GPU's copy of `scratch` (each GPU holds its own copy at its own address)
int static_scratch(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= STATIC_MAX_DEVICES) {
    tatt_set_error("tatt_tri_ssim_fwd_static: no current device (%s)", hipGetErrorString(e));
    return TATT_ERR_ARG;
  }
  std::lock_guard<std::mutex> lock(static_mu);
  if (!static_ptr[dev]) {
    float* p = nullptr;
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
      tatt_set_error("tatt_tri_ssim_fwd_static: hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
      return TATT_ERR_LAUNCH;
    }
    static_ptr[dev] = p;
  }
  *out = static_ptr[dev];
  return TATT_OK;
}

}  // namespace

extern "C" {

size_t tatt_ssim_workspace(int B, int C) { return (size_t)B * C; }

// out[0] = mean over [B,C,H,W] of the tri-SSIM map of (x1, x2, x3)  (TRI_SSIM.forward, size_average=True)
int tatt_tri_ssim_fwd(const float* x1, const float* x2, const float* x3, float* out, int B, int C, int H, int W, float* workspace,
                      hipStream_t st) {
  TATT_CHECK_ARG(x1 && x2 && x3 && out && workspace && B > 0 && C > 0 && H > 0 && W > 0 && H * W <= MAXHW,
                 "tatt_tri_ssim_fwd: bad arguments (H*W must be <= %d)", MAXHW);
  const Gauss G = make_gauss();
  launch_tri_fwd(x1, x2, x3, workspace, B * C, H, W, G, st);
  TATT_LAUNCH_CHECK("tri_ssim_fwd");
  hipLaunchKernelGGL(ssim_final_k, dim3(1), dim3(64), 0, st, workspace, B * C, 1.f / ((float)B * C * H * W), out, 0);
  TATT_LAUNCH_CHECK("tri_ssim_final");
  return TATT_OK;
}

// tatt_tri_ssim_fwd without a workspace argument and without any allocation: the partial sums go to this GPU's copy of the
// file-scope `scratch` array (at most STATIC_MAX_PLANES = 16384 planes, i.e. B*C <= 16384; larger calls are refused).  Every plane's
// slot is overwritten before it is read, so nothing needs clearing between calls; two calls that overlap in time on one GPU
// (different streams) race on the array -- serialise them.  The first call on a device resolves the array's address
// (hipGetSymbolAddress) and must therefore be made outside stream capture; later calls only enqueue the two kernels.
int tatt_tri_ssim_fwd_static(const float* x1, const float* x2, const float* x3, float* out, int B, int C, int H, int W, hipStream_t st) {
  TATT_CHECK_ARG(x1 && x2 && x3 && out && B > 0 && C > 0 && H > 0 && W > 0 && H * W <= MAXHW,
                 "tatt_tri_ssim_fwd_static: bad arguments (H*W must be <= %d)", MAXHW);
  TATT_CHECK_ARG((long)B * C <= STATIC_MAX_PLANES, "tatt_tri_ssim_fwd_static: B*C = %ld planes exceed the static scratch (%d planes); "
                 "use tatt_tri_ssim_fwd with a caller-provided workspace", (long)B * C, STATIC_MAX_PLANES);
  float* ws = nullptr;
  if (int rc = static_scratch(&ws)) return rc;
  const Gauss G = make_gauss();
  launch_tri_fwd(x1, x2, x3, ws, B * C, H, W, G, st);
  TATT_LAUNCH_CHECK("tri_ssim_fwd_static");
  hipLaunchKernelGGL(ssim_final_k, dim3(1), dim3(64), 0, st, ws, B * C, 1.f / ((float)B * C * H * W), out, 0);
  TATT_LAUNCH_CHECK("tri_ssim_final_static");
  return TATT_OK;
}

// dx1, dx2 (+)= dmean * d(mean tri-SSIM)/d(x1, x2)
int tatt_tri_ssim_bwd(const float* x1, const float* x2, const float* x3, float dmean, float* dx1, float* dx2, int accumulate, int B,
                      int C, int H, int W, hipStream_t st) {
  TATT_CHECK_ARG(x1 && x2 && x3 && dx1 && dx2 && B > 0 && C > 0 && H * W <= MAXHW && H > 0 && W > 0,
                 "tatt_tri_ssim_bwd: bad arguments (H*W must be <= %d)", MAXHW);
  const Gauss G = make_gauss();
  SepPlan pl;
  if (sep_plan(H, W, pl) && pl.ns <= SEP_NS && pl.SH * W <= SEP_KPS * SEP_THR && 8 * H * W <= SEP_ARENA)
    hipLaunchKernelGGL(tri_ssim_bwd_sep_k, dim3(B * C), dim3(SEP_THR), 0, st, x1, x2, x3, dmean / ((float)B * C * H * W), dx1, dx2, accumulate,
                       H, W, pl.SH, pl.rows, G);
  else
    hipLaunchKernelGGL(tri_ssim_bwd_k, dim3(B * C), dim3(256), 0, st, x1, x2, x3, dmean / ((float)B * C * H * W), dx1, dx2, accumulate,
                       H, W, G);
  TATT_LAUNCH_CHECK("tri_ssim_bwd");
  return TATT_OK;
}

// out[0] = SSIM(x1[:, :Cuse], x2[:, :Cuse]) averaged over everything (SSIM.forward, size_average=True); tensors are [B,Ctot,H,W]
int tatt_ssim_fwd(const float* x1, const float* x2, float* out, int B, int Ctot, int Cuse, int H, int W, float* workspace,
                  hipStream_t st) {
  TATT_CHECK_ARG(x1 && x2 && out && workspace && B > 0 && Cuse > 0 && Cuse <= Ctot && H > 0 && W > 0 && H * W <= MAXHW,
                 "tatt_ssim_fwd: bad arguments (H*W must be <= %d)", MAXHW);
  const Gauss G = make_gauss();
  hipLaunchKernelGGL(ssim_fwd_k, dim3(B * Cuse), dim3(256), 0, st, x1, x2, workspace, Ctot, Cuse, H, W, G);
  TATT_LAUNCH_CHECK("ssim_fwd");
  hipLaunchKernelGGL(ssim_final_k, dim3(1), dim3(64), 0, st, workspace, B * Cuse, 1.f / ((float)B * Cuse * H * W), out, 0);
  TATT_LAUNCH_CHECK("ssim_final");
  return TATT_OK;
}

// out[0] = 20 log10(255 / sqrt(mean((255 x1[:, :Cuse] - 255 x2[:, :Cuse])^2)))   (calculate_psnr)
int tatt_psnr(const float* x1, const float* x2, float* out, int B, int Ctot, int Cuse, int H, int W, float* workspace, hipStream_t st) {
  TATT_CHECK_ARG(x1 && x2 && out && workspace && B > 0 && Cuse > 0 && Cuse <= Ctot && H > 0 && W > 0, "tatt_psnr: bad arguments");
  hipLaunchKernelGGL(sqdiff255_k, dim3(B * Cuse), dim3(256), 0, st, x1, x2, workspace, Ctot, Cuse, H * W);
  TATT_LAUNCH_CHECK("psnr_partial");
  hipLaunchKernelGGL(ssim_final_k, dim3(1), dim3(64), 0, st, workspace, B * Cuse, 1.f / ((float)B * Cuse * H * W), out, 1);
  TATT_LAUNCH_CHECK("psnr_final");
  return TATT_OK;
}

}  // extern "C"
