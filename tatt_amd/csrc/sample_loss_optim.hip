// TPS rectification (grid build + bilinear gather), TSC rotation sampler, ImageLoss (MSE + gradient prior)
// and the fused clip-grad-norm + Adam optimiser step.
#include "common.h"

namespace {

// ---- bilinear sampling helpers (grid_sample: bilinear, zeros padding, align_corners=False) -------------------
struct Bilin {
  int x0, y0;
  float fx, fy;  // fractional parts
};
__device__ __forceinline__ Bilin bilin_setup(float gx, float gy, int H, int W) {
  float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
  float x0 = floorf(ix), y0 = floorf(iy);
  return Bilin{(int)x0, (int)y0, ix - x0, iy - y0};
}
__device__ __forceinline__ float tap(const float* __restrict__ img, int y, int x, int H, int W) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? img[y * W + x] : 0.f;
}

// TPS: src[b,p,:] = sum_j T[p,j] * ctrl[b,j,:]  (T = target_coordinate_repr @ inverse_kernel[:, :20], [HW,20]);
// grid = 2*clamp(src,0,1)-1 ; y NHWC [B,H,W,C] = bilinear(x NCHW [B,C,H,W], grid)
__global__ void tps_sample_fwd_k(const float* __restrict__ x, const float* __restrict__ ctrl, const float* __restrict__ T,
                                 float* __restrict__ y, float* __restrict__ src_out, int B, int C, int H, int W) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H * W) return;
  int p = i % (H * W), b = i / (H * W);
  float sx = 0.f, sy = 0.f;
  for (int j = 0; j < 20; ++j) {
    float t = T[p * 20 + j];
    sx += t * ctrl[(b * 20 + j) * 2];
    sy += t * ctrl[(b * 20 + j) * 2 + 1];
  }
  if (src_out) {
    src_out[(long)i * 2] = sx;
    src_out[(long)i * 2 + 1] = sy;
  }
  float gx = 2.f * fminf(fmaxf(sx, 0.f), 1.f) - 1.f, gy = 2.f * fminf(fmaxf(sy, 0.f), 1.f) - 1.f;
  Bilin bl = bilin_setup(gx, gy, H, W);
  for (int c = 0; c < C; ++c) {
    const float* img = x + ((long)b * C + c) * H * W;
    float v00 = tap(img, bl.y0, bl.x0, H, W), v01 = tap(img, bl.y0, bl.x0 + 1, H, W);
    float v10 = tap(img, bl.y0 + 1, bl.x0, H, W), v11 = tap(img, bl.y0 + 1, bl.x0 + 1, H, W);
    y[(long)i * C + c] = v00 * (1.f - bl.fx) * (1.f - bl.fy) + v01 * bl.fx * (1.f - bl.fy) + v10 * (1.f - bl.fx) * bl.fy +
                         v11 * bl.fx * bl.fy;
  }
}

// dctrl[b,j,:] (+)= sum_p T[p,j] * dsrc[b,p,:] ; one block per batch element (H*W <= 1024 assumed by the LDS size)
__global__ __launch_bounds__(256) void tps_sample_bwd_k(const float* __restrict__ x, const float* __restrict__ ctrl,
                                                        const float* __restrict__ T, const float* __restrict__ dy,
                                                        float* __restrict__ dctrl, int C, int H, int W, int accumulate) {
  __shared__ float ds[1024 * 2];
  const int b = blockIdx.x, HW = H * W;
  for (int p = threadIdx.x; p < HW; p += 256) {
    float sx = 0.f, sy = 0.f;
    for (int j = 0; j < 20; ++j) {
      float t = T[p * 20 + j];
      sx += t * ctrl[(b * 20 + j) * 2];
      sy += t * ctrl[(b * 20 + j) * 2 + 1];
    }
    float gx = 2.f * fminf(fmaxf(sx, 0.f), 1.f) - 1.f, gy = 2.f * fminf(fmaxf(sy, 0.f), 1.f) - 1.f;
    Bilin bl = bilin_setup(gx, gy, H, W);
    float dix = 0.f, diy = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* img = x + ((long)b * C + c) * HW;
      float v00 = tap(img, bl.y0, bl.x0, H, W), v01 = tap(img, bl.y0, bl.x0 + 1, H, W);
      float v10 = tap(img, bl.y0 + 1, bl.x0, H, W), v11 = tap(img, bl.y0 + 1, bl.x0 + 1, H, W);
      float g = dy[((long)b * HW + p) * C + c];
      dix += g * ((v01 - v00) * (1.f - bl.fy) + (v11 - v10) * bl.fy);
      diy += g * ((v10 - v00) * (1.f - bl.fx) + (v11 - v01) * bl.fx);
    }
    // ix = s*W - 0.5 inside the clamp; torch.clamp passes gradient on the closed interval
    ds[p * 2] = (sx >= 0.f && sx <= 1.f) ? dix * W : 0.f;
    ds[p * 2 + 1] = (sy >= 0.f && sy <= 1.f) ? diy * H : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < 40) {
    int j = threadIdx.x >> 1, xy = threadIdx.x & 1;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += T[p * 20 + j] * ds[p * 2 + xy];
    float* d = dctrl + (b * 20 + j) * 2 + xy;
    *d = accumulate ? *d + s : s;
  }
}

// rotation of TSC: theta = [[cos, sin*r, 0], [-sin/r, cos, 0]], r = H/W; affine_grid (align_corners=False) + bilinear.
// NCHW in, NCHW out.
__global__ void rotate_fwd_k(const float* __restrict__ x, const float* __restrict__ arc, float* __restrict__ y, int B, int C, int H,
                             int W) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H * W) return;
  int px = i % W, py = (i / W) % H, b = i / (H * W);
  float r = (float)H / (float)W, c = cosf(arc[b]), s = sinf(arc[b]);
  float bx = (2.f * px + 1.f) / W - 1.f, by = (2.f * py + 1.f) / H - 1.f;
  float gx = c * bx + s * r * by, gy = (-s / r) * bx + c * by;
  Bilin bl = bilin_setup(gx, gy, H, W);
  for (int ch = 0; ch < C; ++ch) {
    const float* img = x + ((long)b * C + ch) * H * W;
    float v00 = tap(img, bl.y0, bl.x0, H, W), v01 = tap(img, bl.y0, bl.x0 + 1, H, W);
    float v10 = tap(img, bl.y0 + 1, bl.x0, H, W), v11 = tap(img, bl.y0 + 1, bl.x0 + 1, H, W);
    y[((long)b * C + ch) * H * W + py * W + px] = v00 * (1.f - bl.fx) * (1.f - bl.fy) + v01 * bl.fx * (1.f - bl.fy) +
                                                  v10 * (1.f - bl.fx) * bl.fy + v11 * bl.fx * bl.fy;
  }
}
// dx += the transpose of rotate_fwd applied to dy, as a GATHER: one thread per INPUT pixel q sums, in a fixed order, the
// contributions of the output pixels whose four bilinear taps include q.  (The scatter form -- one thread per output pixel, four
// float atomics -- was the library's one summation whose order changed from run to run.)  In pixel units centred on the image the
// sampling position of output pixel (px, py) is a plain rotation of its centre: (ix', iy') = R(arc) (px + 1/2 - W/2, py + 1/2 - H/2),
// so the output pixels that can touch q have their centre within sqrt(2) of R(-arc) q': a 5 x 5 window of candidates around it, each
// re-evaluated with the forward kernel's own arithmetic (same x0 / y0 / fractions, bit for bit) and tested for a tap on q.
__global__ void rotate_bwd_k(const float* __restrict__ arc, const float* __restrict__ dy, float* __restrict__ dx, int B, int C, int H,
                             int W) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H * W) return;
  const int qx = i % W, qy = (i / W) % H, b = i / (H * W);
  const float r = (float)H / (float)W, c = cosf(arc[b]), s = sinf(arc[b]);
  const float qa = qx + 0.5f - 0.5f * W, qb = qy + 0.5f - 0.5f * H;
  const float u0 = c * qa - s * qb + 0.5f * W - 0.5f, v0 = s * qa + c * qb + 0.5f * H - 0.5f;   // output pixel whose sample point is q
  const int cx = (int)floorf(u0 + 0.5f), cy = (int)floorf(v0 + 0.5f);
  float acc[8];
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) acc[ch] = 0.f;
  for (int py = cy - 2; py <= cy + 2; ++py) {
    if (py < 0 || py >= H) continue;
    for (int px = cx - 2; px <= cx + 2; ++px) {
      if (px < 0 || px >= W) continue;
      const float bx = (2.f * px + 1.f) / W - 1.f, by = (2.f * py + 1.f) / H - 1.f;
      const float gx = c * bx + s * r * by, gy = (-s / r) * bx + c * by;
      const Bilin bl = bilin_setup(gx, gy, H, W);
      const int ox = qx - bl.x0, oy = qy - bl.y0;      // 0 or 1: q is the left / right, upper / lower tap
      if ((unsigned)ox > 1u || (unsigned)oy > 1u) continue;
      const float w = (ox ? bl.fx : 1.f - bl.fx) * (oy ? bl.fy : 1.f - bl.fy);
      const float* g = dy + (long)b * C * H * W + py * W + px;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch)
        if (ch < C) acc[ch] += g[(long)ch * H * W] * w;
    }
  }
  float* o = dx + (long)b * C * H * W + qy * W + qx;
#pragma unroll
  for (int ch = 0; ch < 8; ++ch)
    if (ch < C) o[(long)ch * H * W] += acc[ch];
}

// ---- ImageLoss: per-sample mean MSE over C channels + w1 * mean |g(sr)-g(hr)| over the first 3 -----------------
__device__ __forceinline__ float gmap(const float* __restrict__ img, int y, int x, int H, int W, float* a_out, float* c_out) {
  float r = x + 1 < W ? img[y * W + x + 1] : 0.f, l = x > 0 ? img[y * W + x - 1] : 0.f;
  float t = y > 0 ? img[(y - 1) * W + x] : 0.f, bb = y + 1 < H ? img[(y + 1) * W + x] : 0.f;
  float a = (r - l) * 0.5f, c = (t - bb) * 0.5f;
  if (a_out) { *a_out = a; *c_out = c; }
  return sqrtf(a * a + c * c + 1e-6f);
}

// one 1024-thread workgroup per image (B workgroups of 256 threads left each thread 64 gradient-map evaluations in a row)
__device__ __forceinline__ void image_loss_fwd_body(const float* __restrict__ sr, const float* __restrict__ hr, float* __restrict__ per, int b,
                                                    int C, int H, int W, float w0, float w1) {
  __shared__ float red[2][16];
  const int HW = H * W;
  float mse = 0.f, gl = 0.f;
  for (int i = threadIdx.x; i < C * HW; i += 1024) {
    int c = i / HW, p = i - c * HW;
    float d = sr[(long)b * C * HW + i] - hr[(long)b * C * HW + i];
    mse += d * d;
    if (c < 3) {
      int y = p / W, x = p - y * W;
      gl += fabsf(gmap(sr + ((long)b * C + c) * HW, y, x, H, W, nullptr, nullptr) -
                  gmap(hr + ((long)b * C + c) * HW, y, x, H, W, nullptr, nullptr));
    }
  }
  mse = wave_sum(mse);
  gl = wave_sum(gl);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = mse;
    red[1][threadIdx.x >> 6] = gl;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = 0.f, g = 0.f;
    for (int k = 0; k < 16; ++k) {
      m += red[0][k];
      g += red[1][k];
    }
    per[b] = w0 * m / (float)(C * HW) + w1 * g / (float)(3 * HW);
  }
}

__global__ __launch_bounds__(1024) void image_loss_fwd_k(const float* __restrict__ sr, const float* __restrict__ hr,
                                                         float* __restrict__ per, int C, int H, int W, float w0, float w1) {
  image_loss_fwd_body(sr, hr, per, blockIdx.x, C, H, W, w0, w1);
}

// dsr = dper[b] * d per[b] / d sr   (dper may be null: then uniform weight `uniform`); i = element of [B,C,H,W]
__device__ __forceinline__ void image_loss_bwd_body(const float* __restrict__ sr, const float* __restrict__ hr, const float* __restrict__ dper,
                                                    float uniform, float* __restrict__ dsr, int B, int C, int H, int W, float w0, float w1,
                                                    long i) {
  const int HW = H * W;
  if (i >= (long)B * C * HW) return;
  int p = (int)(i % HW);
  int c = (int)((i / HW) % C);
  int b = (int)(i / ((long)C * HW));
  float wgt = dper ? dper[b] : uniform;
  float g = w0 * 2.f * (sr[i] - hr[i]) / (float)(C * HW);
  if (c < 3) {
    const float* s = sr + ((long)b * C + c) * HW;
    const float* h = hr + ((long)b * C + c) * HW;
    int y = p / W, x = p - y * W;
    float acc = 0.f;
    // this pixel is the 'r' tap of (y,x-1), the 'l' tap of (y,x+1), the 't' tap of (y+1,x), the 'b' tap of (y-1,x)
    const int ny[4] = {y, y, y + 1, y - 1}, nx[4] = {x - 1, x + 1, x, x};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int yy = ny[k], xx = nx[k];
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      float a, cc;
      float gs = gmap(s, yy, xx, H, W, &a, &cc), gh = gmap(h, yy, xx, H, W, nullptr, nullptr);
      float d = gs - gh;
      float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      float num = (k == 0) ? 0.5f * a : (k == 1) ? -0.5f * a : (k == 2) ? 0.5f * cc : -0.5f * cc;
      acc += sg * num / gs;
    }
    g += w1 * acc / (float)(3 * HW);
  }
  dsr[i] = wgt * g;
}
__global__ void image_loss_bwd_k(const float* __restrict__ sr, const float* __restrict__ hr, const float* __restrict__ dper,
                                 float uniform, float* __restrict__ dsr, int B, int C, int H, int W, float w0, float w1) {
  image_loss_bwd_body(sr, hr, dper, uniform, dsr, B, C, H, W, w0, w1, (long)blockIdx.x * blockDim.x + threadIdx.x);
}
// Both in ONE launch when the gradient does not depend on the per-sample losses (uniform weight: no --training_stablize mask): the
// first B workgroups reduce the per-sample loss, the others write the gradient -- the loss values are only reported, and as a launch
// of their own they sat between the forward and the backward pass with nothing beside them (20 us + 17 us, alone on the GPU).
__global__ __launch_bounds__(1024) void image_loss_fwd_bwd_k(const float* __restrict__ sr, const float* __restrict__ hr,
                                                             float* __restrict__ per, float uniform, float* __restrict__ dsr, int B, int C,
                                                             int H, int W, float w0, float w1) {
  if ((int)blockIdx.x < B) image_loss_fwd_body(sr, hr, per, blockIdx.x, C, H, W, w0, w1);
  else image_loss_bwd_body(sr, hr, nullptr, uniform, dsr, B, C, H, W, w0, w1, (long)(blockIdx.x - B) * 1024 + threadIdx.x);
}

// ---- optimiser: global grad norm -> clip coefficient -> Adam ------------------------------------------------
// state floats: [0] total norm, [1] clip coef, [2] step (as float), [3] 1-b1^t, [4] 1-b2^t
__global__ __launch_bounds__(256) void sumsq_partial_k(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void clip_final_k(const float* __restrict__ partial, int n, float max_norm, float b1, float b2, float* __restrict__ state) {
  // one wave, double accumulation: lane l takes partial[l], partial[l + 64], ..., then a butterfly (fixed order)
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x != 0) return;
  float norm = (float)sqrt(s);
  state[0] = norm;
  float coef = max_norm / (norm + 1e-6f);
  state[1] = (max_norm > 0.f) ? fminf(coef, 1.f) : 1.f;
  float step = state[2] + 1.f;
  state[2] = step;
  state[3] = 1.f - powf(b1, step);
  state[4] = 1.f - powf(b2, step);
}
__global__ void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                       float lr, float b1, float b2, float eps, const float* __restrict__ state) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gg = g[i] * state[1];
  float mm = b1 * m[i] + (1.f - b1) * gg;
  float vv = b2 * v[i] + (1.f - b2) * gg * gg;
  m[i] = mm;
  v[i] = vv;
  float denom = sqrtf(vv) / sqrtf(state[4]) + eps;
  p[i] -= (lr / state[3]) * (mm / denom);
}

}  // namespace

extern "C" {

int tatt_tps_sample_fwd(const float* x, const float* ctrl, const float* T, float* y, float* src_out, int B, int C, int H, int W,
                        hipStream_t st) {
  hipLaunchKernelGGL(tps_sample_fwd_k, dim3(cdiv((long)B * H * W, 256)), dim3(256), 0, st, x, ctrl, T, y, src_out, B, C, H, W);
  TATT_LAUNCH_CHECK("tps_sample_fwd");
  return TATT_OK;
}
int tatt_tps_sample_bwd(const float* x, const float* ctrl, const float* T, const float* dy, float* dctrl, int B, int C, int H, int W,
                        int accumulate, hipStream_t st) {
  TATT_CHECK_ARG(H * W <= 1024, "tatt_tps_sample_bwd: H*W must be <= 1024");
  hipLaunchKernelGGL(tps_sample_bwd_k, dim3(B), dim3(256), 0, st, x, ctrl, T, dy, dctrl, C, H, W, accumulate);
  TATT_LAUNCH_CHECK("tps_sample_bwd");
  return TATT_OK;
}
int tatt_rotate_fwd(const float* x, const float* arc, float* y, int B, int C, int H, int W, hipStream_t st) {
  hipLaunchKernelGGL(rotate_fwd_k, dim3(cdiv((long)B * H * W, 256)), dim3(256), 0, st, x, arc, y, B, C, H, W);
  TATT_LAUNCH_CHECK("rotate_fwd");
  return TATT_OK;
}
int tatt_rotate_bwd(const float* arc, const float* dy, float* dx, int B, int C, int H, int W, hipStream_t st) {
  TATT_CHECK_ARG(arc && dy && dx && B > 0 && C >= 1 && C <= 8 && H > 0 && W > 0, "tatt_rotate_bwd: need 1 <= C <= 8 (got %d)", C);
  hipLaunchKernelGGL(rotate_bwd_k, dim3(cdiv((long)B * H * W, 256)), dim3(256), 0, st, arc, dy, dx, B, C, H, W);
  TATT_LAUNCH_CHECK("rotate_bwd");
  return TATT_OK;
}
int tatt_image_loss_fwd(const float* sr, const float* hr, float* per, int B, int C, int H, int W, float w0, float w1, hipStream_t st) {
  hipLaunchKernelGGL(image_loss_fwd_k, dim3(B), dim3(1024), 0, st, sr, hr, per, C, H, W, w0, w1);
  TATT_LAUNCH_CHECK("image_loss_fwd");
  return TATT_OK;
}
int tatt_image_loss_bwd(const float* sr, const float* hr, const float* dper, float uniform, float* dsr, int B, int C, int H, int W,
                        float w0, float w1, hipStream_t st) {
  hipLaunchKernelGGL(image_loss_bwd_k, dim3(cdiv((long)B * C * H * W, 256)), dim3(256), 0, st, sr, hr, dper, uniform, dsr, B, C, H, W,
                     w0, w1);
  TATT_LAUNCH_CHECK("image_loss_bwd");
  return TATT_OK;
}
int tatt_image_loss_fwd_bwd(const float* sr, const float* hr, float* per, float uniform, float* dsr, int B, int C, int H, int W, float w0,
                            float w1, hipStream_t st) {
  TATT_CHECK_ARG(sr && hr && per && dsr && B > 0 && C > 0 && H > 0 && W > 0, "tatt_image_loss_fwd_bwd: bad arguments");
  hipLaunchKernelGGL(image_loss_fwd_bwd_k, dim3(B + cdiv((long)B * C * H * W, 1024)), dim3(1024), 0, st, sr, hr, per, uniform, dsr, B, C, H,
                     W, w0, w1);
  TATT_LAUNCH_CHECK("image_loss_fwd_bwd");
  return TATT_OK;
}
// workspace: 1024 floats; state: 8 floats (persistent across steps, zero-initialised by the caller)
int tatt_clip_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float max_norm,
                   float* state, float* workspace, hipStream_t st) {
  int nblk = cdiv(n, 256 * 16);
  if (nblk > 1024) nblk = 1024;
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(sumsq_partial_k, dim3(nblk), dim3(256), 0, st, g, n, workspace);
  TATT_LAUNCH_CHECK("sumsq_partial");
  hipLaunchKernelGGL(clip_final_k, dim3(1), dim3(64), 0, st, workspace, nblk, max_norm, b1, b2, state);
  TATT_LAUNCH_CHECK("clip_final");
  hipLaunchKernelGGL(adam_k, dim3(cdiv(n, 256)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, state);
  TATT_LAUNCH_CHECK("adam");
  return TATT_OK;
}

}  // extern "C"
