// Achievable HBM read / read+write rates on this GPU for the access shapes the streaming kernels use.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bw/bw tools/bw/bw.hip ; run on the GPU box: ./tools/bw/bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int V>  // V floats per lane per load
__global__ __launch_bounds__(512) void rd(const float* __restrict__ p, long n, float* out, int unroll_dummy) {
  typedef float vec __attribute__((ext_vector_type(V)));
  const vec* q = reinterpret_cast<const vec*>(p);
  long nv = n / V, i = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
  float s = 0.f;
  for (; i + 3 * st < nv; i += 4 * st) {
    vec a = q[i], b = q[i + st], c = q[i + 2 * st], d = q[i + 3 * st];
    for (int k = 0; k < V; ++k) s += a[k] + b[k] + c[k] + d[k];
  }
  if (s == 1.2345f) out[0] = s;
}
template <int V>
__global__ __launch_bounds__(512) void cp(const float* __restrict__ p, float* __restrict__ o, long n) {
  typedef float vec __attribute__((ext_vector_type(V)));
  const vec* q = reinterpret_cast<const vec*>(p);
  vec* w = reinterpret_cast<vec*>(o);
  long nv = n / V, i = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
  for (; i + 3 * st < nv; i += 4 * st) {
    vec a = q[i], b = q[i + st], c = q[i + 2 * st], d = q[i + 3 * st];
    w[i] = a; w[i + st] = b; w[i + 2 * st] = c; w[i + 3 * st] = d;
  }
}
// row-block streaming like tn_small: each block owns a contiguous row range, wave reads 2 rows x 128 B per instruction
__global__ __launch_bounds__(512) void rows(const float* __restrict__ p, int R, int ld, int cols32, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int rpb = R / gridDim.x, rpw = rpb / 8;
  const int r0 = blockIdx.x * rpb + wave * rpw;
  float s = 0.f;
  for (int r = r0; r < r0 + rpw; r += 8) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      for (int c = 0; c < cols32; ++c) s += p[(long)(r + 2 * u + lh) * ld + c * 32 + li];
  }
  if (s == 1.2345f) out[0] = s;
}

template <int V>
__global__ __launch_bounds__(512) void wr(float* __restrict__ o, long n) {
  typedef float vec __attribute__((ext_vector_type(V)));
  vec* w = reinterpret_cast<vec*>(o);
  long nv = n / V, i = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
  vec v;
  for (int k = 0; k < V; ++k) v[k] = (float)threadIdx.x;
  for (; i < nv; i += st) w[i] = v;
}
// 2048 waves (256 blocks x 8), each streaming its own region of len bytes that starts at wave_id * stride: does the spacing of
// concurrent streams matter (HBM channel interleave)?
__global__ __launch_bounds__(512) void streams(const char* __restrict__ p, long stride, int len, float* out) {
  const int gw = blockIdx.x * 8 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const float* q = reinterpret_cast<const float*>(p + gw * stride) + lane;
  float s = 0.f;
  for (int o = 0; o < len / 4; o += 8 * 64) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s += q[o + u * 64];
  }
  if (s == 1.2345f) out[0] = s;
}
// the same bytes, but wave gw reads chunk (it * 2048 + gw): all waves sweep one dense window together
__global__ __launch_bounds__(512) void interleaved(const char* __restrict__ p, int chunk, int nit, float* out) {
  const int gw = blockIdx.x * 8 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  float s = 0.f;
  for (int it = 0; it < nit; ++it) {
    const float* q = reinterpret_cast<const float*>(p + ((long)it * 2048 + gw) * chunk) + lane;
    for (int o = 0; o < chunk / 4; o += 8 * 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s += q[o + u * 64];
    }
  }
  if (s == 1.2345f) out[0] = s;
}

int main() {
  long n = 64L << 20;  // 256 MB
  float *a, *b, *o;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&o, 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto f, const char* name, double bytes) {
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
  };
  for (int g : {256, 512, 1024, 2048, 8192}) {
    char nm[64];
    snprintf(nm, 64, "read b32  grid %d x512", g); time([&] { hipLaunchKernelGGL(rd<1>, dim3(g), dim3(512), 0, 0, a, n, o, 0); }, nm, n * 4.0);
    snprintf(nm, 64, "read b128 grid %d x512", g); time([&] { hipLaunchKernelGGL(rd<4>, dim3(g), dim3(512), 0, 0, a, n, o, 0); }, nm, n * 4.0);
    snprintf(nm, 64, "copy b128 grid %d x512", g); time([&] { hipLaunchKernelGGL(cp<4>, dim3(g), dim3(512), 0, 0, a, b, n); }, nm, n * 8.0);
  }
  // smaller footprints (fit in the 256 MB MALL / partially in L2)
  for (long m : {4L << 20, 12L << 20, 32L << 20}) {
    char nm[64];
    snprintf(nm, 64, "read b128 %ld MB grid 2048", m * 4 >> 20); time([&] { hipLaunchKernelGGL(rd<4>, dim3(2048), dim3(512), 0, 0, a, m, o, 0); }, nm, m * 4.0);
    snprintf(nm, 64, "copy b128 %ld MB grid 2048", m * 4 >> 20); time([&] { hipLaunchKernelGGL(cp<4>, dim3(2048), dim3(512), 0, 0, a, b, m); }, nm, m * 8.0);
  }
  for (int g : {256, 512})
    for (int c : {1, 2, 3, 6}) {
      char nm[64];
      int ld = 192, R = 65536 * 4;
      snprintf(nm, 64, "rows [%d,%d] ld %d grid %d", R, c * 32, ld, g);
      time([&] { hipLaunchKernelGGL(rows, dim3(g), dim3(512), 0, 0, a, R, ld, c, o); }, nm, (double)R * c * 128);
    }
  for (int g : {512, 2048}) {
    char nm[64];
    snprintf(nm, 64, "write b32  grid %d x512", g); time([&] { hipLaunchKernelGGL(wr<1>, dim3(g), dim3(512), 0, 0, b, n); }, nm, n * 4.0);
    snprintf(nm, 64, "write b128 grid %d x512", g); time([&] { hipLaunchKernelGGL(wr<4>, dim3(g), dim3(512), 0, 0, b, n); }, nm, n * 4.0);
    snprintf(nm, 64, "write b128 64MB grid %d", g); time([&] { hipLaunchKernelGGL(wr<4>, dim3(g), dim3(512), 0, 0, b, n / 4); }, nm, n * 1.0);
  }
  for (long extra : {0L, 256L, 512L, 1024L, 2048L, 4096L, 8192L, 16384L, 32768L}) {
    char nm[64];
    long stride = 98304 + extra;
    snprintf(nm, 64, "2048 streams len 96K stride 96K+%ld", extra);
    time([&] { hipLaunchKernelGGL(streams, dim3(256), dim3(512), 0, 0, (const char*)a, stride, 98304, o); }, nm, 2048.0 * 98304);
  }
  for (int chunk : {2048, 6144, 16384}) {
    char nm[64];
    int nit = 98304 / chunk;
    snprintf(nm, 64, "interleaved chunks of %d B", chunk);
    time([&] { hipLaunchKernelGGL(interleaved, dim3(256), dim3(512), 0, 0, (const char*)a, chunk, nit, o); }, nm, 2048.0 * nit * chunk);
  }
  return 0;
}
