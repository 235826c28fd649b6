// Does a VALU instruction finish sooner when EXEC leaves whole 16-lane rows idle?  Dependent chains on one wavefront under four
// masks (4, 2, 1 rows enabled, and all rows half enabled):   hipcc --offload-arch=gfx950 -O3 rows.hip -o rows && ./rows
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
#define REP8(s) s s s s s s s s
#define BODY(name, asmtext, decl, outs, ins)                                                                   \
__global__ void name(unsigned long long* out, double* sink, unsigned long long mask, double y, int idx) {      \
  decl                                                                                                          \
  unsigned long long sv; asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=&s"(sv) : "s"(mask));        \
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);                \
  for (int i = 0; i < N / 8; ++i) asm volatile(REP8(asmtext) : outs : ins);                                      \
  __builtin_amdgcn_sched_barrier(0); const unsigned long long t1 = __builtin_amdgcn_s_memtime();                 \
  asm volatile("s_mov_b64 exec, %0" :: "s"(sv));                                                                 \
  if (threadIdx.x == 0) out[idx] = t1 - t0;                                                                      \
  sink[threadIdx.x] = (double)x;                                                                                 \
}
BODY(k_u32, "v_add_u32 %0, %0, %1\n", unsigned x = threadIdx.x; unsigned yy = (unsigned)y;, "+v"(x), "v"(yy))
BODY(k_f64, "v_add_f64 %0, %0, %1\n", double x = threadIdx.x;, "+v"(x), "v"(y))
BODY(k_fma, "v_fma_f64 %0, %0, %1, %1\n", double x = threadIdx.x;, "+v"(x), "v"(y))
BODY(k_mul, "v_mul_lo_u32 %0, %0, %1\n", unsigned x = threadIdx.x; unsigned yy = (unsigned)y;, "+v"(x), "v"(yy))
BODY(k_rcp, "v_rcp_f64 %0, %0\n", double x = threadIdx.x + 2.0;, "+v"(x), "v"(y))
BODY(k_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32 %0, %0, %1\n", unsigned x = threadIdx.x; unsigned yy = (unsigned)y;, "+v"(x), "v"(yy))
int main() {
  unsigned long long* out; double* sink;
  (void)hipMalloc(&out, 8 * 64); (void)hipMalloc(&sink, 8 * 64); (void)hipMemset(out, 0, 8 * 64);
  const unsigned long long masks[4] = {~0ull, 0x00000000FFFFFFFFull, 0x000000000000FFFFull, 0x00FF00FF00FF00FFull};
  for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 4; ++m) {
      k_u32<<<1, 64>>>(out, sink, masks[m], 3.0, m * 6 + 0); k_f64<<<1, 64>>>(out, sink, masks[m], 1.5, m * 6 + 1); k_fma<<<1, 64>>>(out, sink, masks[m], 0.5, m * 6 + 2);
      k_mul<<<1, 64>>>(out, sink, masks[m], 3.0, m * 6 + 3); k_rcp<<<1, 64>>>(out, sink, masks[m], 1.0, m * 6 + 4); k_dpp<<<1, 64>>>(out, sink, masks[m], 3.0, m * 6 + 5);
      (void)hipDeviceSynchronize();
    }
  unsigned long long h[64]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[6] = {"v_add_u32", "v_add_f64", "v_fma_f64", "v_mul_lo_u32", "v_rcp_f64", "v_mov_dpp + v_add_u32"};
  const char* mn[4] = {"4 rows", "2 rows", "1 row", "4 half rows"};
  printf("%-24s", "dependent chain, ticks/op"); for (int m = 0; m < 4; ++m) printf("%12s", mn[m]); printf("\n");
  for (int k = 0; k < 6; ++k) { printf("%-24s", nm[k]); for (int m = 0; m < 4; ++m) printf("%12.2f", h[m * 6 + k] / (double)N); printf("\n"); }
  return 0;
}
