// What it costs one wavefront when a scalar instruction consumes a scalar register a VALU instruction just wrote (lane masks from
// v_cmp, values from v_readlane) -- the compiler's representation of every `bool` and every wave-uniform value taken from a lane:
//   hipcc --offload-arch=gfx950 -O3 cross.hip -o cross && ./cross          (s_memtime ticks per loop step)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
#define T0 const unsigned long long t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#define T1(i) __builtin_amdgcn_sched_barrier(0); const unsigned long long t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[i] = t1 - t0;
#define REP4(s) s s s s
// (0) compare -> select through VCC, all on the VALU
__global__ void k_cmp_sel(unsigned long long* out, int* sink, int y) {
  int x = threadIdx.x;
  T0
  for (int i = 0; i < N / 4; ++i)
    asm volatile(REP4("v_cmp_lt_i32 vcc, %[x], %[y]\n v_cndmask_b32 %[x], 7, %[x], vcc\n v_add_u32 %[x], 3, %[x]\n") : [x] "+v"(x) : [y] "v"(y) : "vcc", "scc");
  T1(0) sink[threadIdx.x] = x;
}
// (1) compare -> scalar AND of the mask -> select with the combined mask
__global__ void k_cmp_sand_sel(unsigned long long* out, int* sink, int y, unsigned long long k) {
  int x = threadIdx.x; unsigned long long m;
  T0
  for (int i = 0; i < N / 4; ++i)
    asm volatile(REP4("v_cmp_lt_i32 vcc, %[x], %[y]\n s_and_b64 %[m], vcc, %[k]\n v_cndmask_b32 %[x], 7, %[x], %[m]\n v_add_u32 %[x], 3, %[x]\n")
                 : [x] "+v"(x), [m] "=&s"(m) : [y] "v"(y), [k] "s"(k) : "vcc", "scc");
  T1(1) sink[threadIdx.x] = x;
}
// (2) the same logic with the mask kept in a VGPR as 0 / -1 (no scalar instruction): cmp, cndmask to 0/-1, v_and, v_cmp_ne, select
__global__ void k_cmp_vand_sel(unsigned long long* out, int* sink, int y, int kv) {
  int x = threadIdx.x; int m;
  T0
  for (int i = 0; i < N / 4; ++i)
    asm volatile(REP4("v_cmp_lt_i32 vcc, %[x], %[y]\n v_cndmask_b32 %[m], 0, -1, vcc\n v_and_b32 %[m], %[m], %[k]\n v_cmp_ne_u32 vcc, 0, %[m]\n v_cndmask_b32 %[x], 7, %[x], vcc\n v_add_u32 %[x], 3, %[x]\n")
                 : [x] "+v"(x), [m] "=&v"(m) : [y] "v"(y), [k] "v"(kv) : "vcc", "scc");
  T1(2) sink[threadIdx.x] = x;
}
// (3) an `if`: compare -> s_and_saveexec -> one VALU instruction -> s_or exec
__global__ void k_if(unsigned long long* out, int* sink, int y) {
  int x = threadIdx.x; unsigned long long sv;
  T0
  for (int i = 0; i < N / 4; ++i)
    asm volatile(REP4("v_cmp_lt_i32 vcc, %[x], %[y]\n s_and_saveexec_b64 %[sv], vcc\n v_add_u32 %[x], 5, %[x]\n s_or_b64 exec, exec, %[sv]\n v_add_u32 %[x], 3, %[x]\n")
                 : [x] "+v"(x), [sv] "=&s"(sv) : [y] "v"(y) : "vcc", "scc");
  T1(3) sink[threadIdx.x] = x;
}
// (4) the same effect predicated: compare -> select of the addend -> add
__global__ void k_if_sel(unsigned long long* out, int* sink, int y) {
  int x = threadIdx.x; int a;
  T0
  for (int i = 0; i < N / 4; ++i)
    asm volatile(REP4("v_cmp_lt_i32 vcc, %[x], %[y]\n v_cndmask_b32 %[a], 0, 5, vcc\n v_add3_u32 %[x], %[x], %[a], 3\n")
                 : [x] "+v"(x), [a] "=&v"(a) : [y] "v"(y) : "vcc", "scc");
  T1(4) sink[threadIdx.x] = x;
}
// (5) v_readfirstlane -> scalar add -> back into the VALU as an operand
__global__ void k_rfl_sadd(unsigned long long* out, int* sink, int y) {
  int x = threadIdx.x; int s;
  T0
  for (int i = 0; i < N / 4; ++i)
    asm volatile(REP4("v_readfirstlane_b32 %[s], %[x]\n s_add_i32 %[s], %[s], 3\n v_add_u32 %[x], %[s], %[x]\n") : [x] "+v"(x), [s] "=&s"(s) : [y] "v"(y) : "scc");
  T1(5) sink[threadIdx.x] = x;
}
// (6) v_readfirstlane -> straight back into the VALU (no scalar instruction in between)
__global__ void k_rfl_vadd(unsigned long long* out, int* sink, int y) {
  int x = threadIdx.x; int s;
  T0
  for (int i = 0; i < N / 4; ++i)
    asm volatile(REP4("v_readfirstlane_b32 %[s], %[x]\n v_add3_u32 %[x], %[s], %[x], 3\n") : [x] "+v"(x), [s] "=&s"(s) : [y] "v"(y) : "scc");
  T1(6) sink[threadIdx.x] = x;
}
// (7) ballot -> s_ff1 (a uniform decision from a vector condition) -> VALU operand
__global__ void k_ballot_ff1(unsigned long long* out, int* sink, int y) {
  int x = threadIdx.x; int s; unsigned long long m;
  T0
  for (int i = 0; i < N / 4; ++i)
    asm volatile(REP4("v_cmp_lt_i32 %[m], %[x], %[y]\n s_ff1_i32_b64 %[s], %[m]\n v_add3_u32 %[x], %[s], %[x], 3\n") : [x] "+v"(x), [s] "=&s"(s), [m] "=&s"(m) : [y] "v"(y) : "scc");
  T1(7) sink[threadIdx.x] = x;
}
// (8) scalar-written mask consumed by the VALU (the cheap direction): s_not of a constant mask -> select
__global__ void k_smask_sel(unsigned long long* out, int* sink, unsigned long long k) {
  int x = threadIdx.x; unsigned long long m = k;
  T0
  for (int i = 0; i < N / 4; ++i)
    asm volatile(REP4("s_not_b64 %[m], %[m]\n v_cndmask_b32 %[x], 7, %[x], %[m]\n v_add_u32 %[x], 3, %[x]\n") : [x] "+v"(x), [m] "+s"(m) : : "scc");
  T1(8) sink[threadIdx.x] = x;
}
int main() {
  unsigned long long* out; int* sink;
  (void)hipMalloc(&out, 8 * 16); (void)hipMalloc(&sink, 4 * 64); (void)hipMemset(out, 0, 8 * 16);
  for (int rep = 0; rep < 2; ++rep) {
    k_cmp_sel<<<1, 64>>>(out, sink, 1 << 30); k_cmp_sand_sel<<<1, 64>>>(out, sink, 1 << 30, 0xFFFF0000FFFF0000ull); k_cmp_vand_sel<<<1, 64>>>(out, sink, 1 << 30, -1);
    k_if<<<1, 64>>>(out, sink, 1 << 30); k_if_sel<<<1, 64>>>(out, sink, 1 << 30); k_rfl_sadd<<<1, 64>>>(out, sink, 1); k_rfl_vadd<<<1, 64>>>(out, sink, 1);
    k_ballot_ff1<<<1, 64>>>(out, sink, 1 << 30); k_smask_sel<<<1, 64>>>(out, sink, 0xFFFF0000FFFF0000ull);
    (void)hipDeviceSynchronize();
  }
  unsigned long long h[16]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[] = {"v_cmp -> v_cndmask (VCC) -> add                [3 instr]", "v_cmp -> s_and_b64 -> v_cndmask -> add         [4 instr]",
                      "v_cmp -> cndmask 0/-1 -> v_and -> v_cmp -> cndmask -> add [6]", "if: v_cmp -> s_and_saveexec -> add -> s_or exec -> add [5]",
                      "predicated if: v_cmp -> cndmask addend -> add3 [3 instr]", "v_readfirstlane -> s_add -> v_add               [3 instr]",
                      "v_readfirstlane -> v_add3                       [2 instr]", "v_cmp (sgpr) -> s_ff1 -> v_add3                 [3 instr]",
                      "s_not mask -> v_cndmask -> add                  [3 instr]"};
  for (int i = 0; i < 9; ++i) printf("%-66s %7.2f ticks/step\n", nm[i], h[i] / (double)N);
  return 0;
}
