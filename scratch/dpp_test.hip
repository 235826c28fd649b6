#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  int lane = threadIdx.x;
  int v = lane * 10;
  int up = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false);   // wave_shr:1
  int dn = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xF, 0xF, false);   // wave_shl:1
  out[lane] = up; out[64 + lane] = dn;
}
int main() {
  int* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("wave_shr:1 :"); for (int i = 0; i < 64; ++i) printf(" %d", h[i]); printf("\n");
  printf("wave_shl:1 :"); for (int i = 0; i < 64; ++i) printf(" %d", h[64 + i]); printf("\n");
  return 0;
}
