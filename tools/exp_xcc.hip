// Which XCD does workgroup i of a 1-D grid land on?  (HW_REG_XCC_ID, hwreg 20, bits 3:0)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/exp_xcc tools/exp_xcc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;
}
int main() {
  const int n = 4096;
  int* d;
  (void)hipMalloc(&d, n * 4);
  for (int threads : {64, 512}) {
    hipLaunchKernelGGL(k, dim3(n), dim3(threads), 0, 0, d);
    int h[n];
    (void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int cnt[16] = {0}, rr = 0;
    for (int i = 0; i < n; ++i) { cnt[h[i]]++; rr += h[i] == (i & 7); }
    printf("threads %d: first 24 ids:", threads);
    for (int i = 0; i < 24; ++i) printf(" %d", h[i]);
    printf("\n  per-XCD counts:");
    for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
    printf("   id == wg %% 8 for %d of %d\n", rr, n);
  }
  return 0;
}
