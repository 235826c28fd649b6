// What an ordered, lane-masked fp64 sum costs per step on one wavefront (s_memtime ticks), for the forms k_tau_update could use:
//   hipcc --offload-arch=gfx950 -O3 exec.hip -o exec && ./exec
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
#define T0 const unsigned long long t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#define T1(i) __builtin_amdgcn_sched_barrier(0); const unsigned long long t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[i] = t1 - t0;
// (a) EXEC <- scalar mask, add, per step (masks and values already in registers)
__global__ void k_exec_add(unsigned long long* out, double* sink, unsigned long long m0, unsigned long long m1, double d) {
  double t = threadIdx.x; unsigned long long sv;
  T0
  asm volatile("s_mov_b64 %0, exec" : "=s"(sv));
  for (int i = 0; i < N / 8; ++i) {
    asm volatile(
      "s_mov_b64 exec, %[a]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[b]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[a]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[b]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[a]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[b]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[a]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[b]\n v_add_f64 %[t], %[t], %[d]\n"
      : [t] "+v"(t) : [a] "s"(m0), [b] "s"(m1), [d] "v"(d));
  }
  asm volatile("s_mov_b64 exec, %0" :: "s"(sv));
  T1(0) sink[threadIdx.x] = t;
}
// (b) the same with the value as a scalar operand
__global__ void k_exec_add_s(unsigned long long* out, double* sink, unsigned long long m0, unsigned long long m1, double d) {
  double t = threadIdx.x; unsigned long long sv;
  T0
  asm volatile("s_mov_b64 %0, exec" : "=s"(sv));
  for (int i = 0; i < N / 8; ++i) {
    asm volatile(
      "s_mov_b64 exec, %[a]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[b]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[a]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[b]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[a]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[b]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[a]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[b]\n v_add_f64 %[t], %[t], %[d]\n"
      : [t] "+v"(t) : [a] "s"(m0), [b] "s"(m1), [d] "s"(d));
  }
  asm volatile("s_mov_b64 exec, %0" :: "s"(sv));
  T1(1) sink[threadIdx.x] = t;
}
// (c) today's dense form: carry-out mask, EXEC <- mask, add, EXEC <- all
__global__ void k_carry_exec(unsigned long long* out, double* sink, unsigned y0, double d) {
  double t = threadIdx.x; unsigned y = y0 ^ threadIdx.x; unsigned long long sv, m;
  T0
  for (int i = 0; i < N / 4; ++i) {
    asm volatile(
      "s_mov_b64 %[sv], exec\n"
      "v_add_co_u32 %[y], %[m], %[y], %[y]\n s_mov_b64 exec, %[m]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[sv]\n"
      "v_add_co_u32 %[y], %[m], %[y], %[y]\n s_mov_b64 exec, %[m]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[sv]\n"
      "v_add_co_u32 %[y], %[m], %[y], %[y]\n s_mov_b64 exec, %[m]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[sv]\n"
      "v_add_co_u32 %[y], %[m], %[y], %[y]\n s_mov_b64 exec, %[m]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[sv]\n"
      : [t] "+v"(t), [y] "+v"(y), [sv] "=&s"(sv), [m] "=&s"(m) : [d] "v"(d));
    y |= 0x10001u;
  }
  T1(2) sink[threadIdx.x] = t + y;
}
// (d) select form: test, two v_cndmask, add
__global__ void k_select(unsigned long long* out, double* sink, unsigned y0, double d) {
  double t = threadIdx.x; unsigned y = y0 ^ threadIdx.x;
  T0
  for (int i = 0; i < N; ++i) { t += (int)y < 0 ? d : 0.0; y = (y << 1) | 1u; asm volatile("" : "+v"(t), "+v"(y)); }
  T1(3) sink[threadIdx.x] = t + y;
}
// (e) fma with a 0.0 / 1.0 factor converted from the bit (exact: d * 1 = d, d * 0 = 0)
__global__ void k_fma01(unsigned long long* out, double* sink, unsigned y0, double d) {
  double t = threadIdx.x; unsigned y = y0 ^ threadIdx.x;
  T0
  for (int i = 0; i < N; ++i) { const double f = (double)(y >> 31); t = __builtin_fma(d, f, t); y = (y << 1) | 1u; asm volatile("" : "+v"(t), "+v"(y)); }
  T1(4) sink[threadIdx.x] = t + y;
}
// (f) plain dependent adds: the floor
__global__ void k_add(unsigned long long* out, double* sink, double d) {
  double t = threadIdx.x;
  T0
  for (int i = 0; i < N; ++i) { t += d; asm volatile("" : "+v"(t)); }
  T1(5) sink[threadIdx.x] = t;
}
// (g) masks by v_cmp into SGPR pairs first (8 at a time), then EXEC <- mask, add
__global__ void k_exec_add_nosave(unsigned long long* out, double* sink, unsigned long long m0, double d) {
  double t = threadIdx.x;
  T0
  for (int i = 0; i < N / 8; ++i) {
    asm volatile(
      "s_mov_b64 exec, %[a]\n v_add_f64 %[t], %[t], %[d]\n v_add_f64 %[t], %[t], %[d]\n v_add_f64 %[t], %[t], %[d]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[a]\n v_add_f64 %[t], %[t], %[d]\n v_add_f64 %[t], %[t], %[d]\n v_add_f64 %[t], %[t], %[d]\n v_add_f64 %[t], %[t], %[d]\n"
      : [t] "+v"(t) : [a] "s"(m0), [d] "v"(d));
  }
  asm volatile("s_mov_b64 exec, -1");
  T1(6) sink[threadIdx.x] = t;
}
// (h) bit j of x by v_bfe, cvt to 0.0 / 1.0, fma: three instructions per step, no running shift
__global__ void k_bfe_fma(unsigned long long* out, double* sink, unsigned y0, double d) {
  double t = threadIdx.x; unsigned x = y0 ^ threadIdx.x;
  T0
  for (int i = 0; i < N / 32; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) { const double f = (double)((x >> j) & 1u); t = __builtin_fma(d, f, t); }
    asm volatile("" : "+v"(t), "+v"(x));
  }
  T1(7) sink[threadIdx.x] = t + x;
}
// (i) v_cmpx writes EXEC from the VALU, add, EXEC <- all
__global__ void k_cmpx(unsigned long long* out, double* sink, unsigned y0, double d) {
  double t = threadIdx.x; unsigned y = y0 ^ threadIdx.x; unsigned long long sv;
  T0
  asm volatile("s_mov_b64 %0, exec" : "=s"(sv));
  for (int i = 0; i < N / 4; ++i) {
    asm volatile(
      "v_cmpx_gt_i32 vcc, 0, %[y]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[sv]\n v_lshlrev_b32 %[y], 1, %[y]\n"
      "v_cmpx_gt_i32 vcc, 0, %[y]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[sv]\n v_lshlrev_b32 %[y], 1, %[y]\n"
      "v_cmpx_gt_i32 vcc, 0, %[y]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[sv]\n v_lshlrev_b32 %[y], 1, %[y]\n"
      "v_cmpx_gt_i32 vcc, 0, %[y]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[sv]\n v_lshlrev_b32 %[y], 1, %[y]\n"
      : [t] "+v"(t), [y] "+v"(y) : [d] "v"(d), [sv] "s"(sv) : "vcc");
    y |= 0x10001u;
  }
  T1(8) sink[threadIdx.x] = t + y;
}
// (j) masks made eight steps ahead by v_cmp into SGPR pairs, then EXEC <- mask, add
__global__ void k_cmp_ahead(unsigned long long* out, double* sink, unsigned y0, double d) {
  double t = threadIdx.x; unsigned x = y0 ^ threadIdx.x; unsigned long long sv, m0, m1, m2, m3, m4, m5, m6, m7;
  T0
  for (int i = 0; i < N / 8; ++i) {
    asm volatile(
      "s_mov_b64 %[sv], exec\n"
      "v_cmp_lt_i32 %[m0], 4, %[x]\n v_cmp_lt_i32 %[m1], 5, %[x]\n v_cmp_lt_i32 %[m2], 6, %[x]\n v_cmp_lt_i32 %[m3], 7, %[x]\n"
      "v_cmp_lt_i32 %[m4], 8, %[x]\n v_cmp_lt_i32 %[m5], 9, %[x]\n v_cmp_lt_i32 %[m6], 10, %[x]\n v_cmp_lt_i32 %[m7], 11, %[x]\n"
      "s_mov_b64 exec, %[m0]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[m1]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[m2]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[m3]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[m4]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[m5]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[m6]\n v_add_f64 %[t], %[t], %[d]\n s_mov_b64 exec, %[m7]\n v_add_f64 %[t], %[t], %[d]\n"
      "s_mov_b64 exec, %[sv]\n"
      : [t] "+v"(t), [sv] "=&s"(sv), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4), [m5] "=&s"(m5), [m6] "=&s"(m6), [m7] "=&s"(m7)
      : [d] "v"(d), [x] "v"(x));
  }
  T1(9) sink[threadIdx.x] = t + x;
}
int main() {
  unsigned long long* out; double* sink;
  hipMalloc(&out, 8 * 16); hipMalloc(&sink, 8 * 64); hipMemset(out, 0, 8 * 16);
  for (int rep = 0; rep < 2; ++rep) {
    k_exec_add<<<1, 64>>>(out, sink, 0xFFFF0000FFFF0000ull, 0x0F0F0F0F0F0F0F0Full, 1.5);
    k_exec_add_s<<<1, 64>>>(out, sink, 0xFFFF0000FFFF0000ull, 0x0F0F0F0F0F0F0F0Full, 1.5);
    k_carry_exec<<<1, 64>>>(out, sink, 0xA5A5A5A5u, 1.5);
    k_select<<<1, 64>>>(out, sink, 0xA5A5A5A5u, 1.5);
    k_fma01<<<1, 64>>>(out, sink, 0xA5A5A5A5u, 1.5);
    k_add<<<1, 64>>>(out, sink, 1.5);
    k_exec_add_nosave<<<1, 64>>>(out, sink, 0xFFFF0000FFFF0000ull, 1.5);
    k_bfe_fma<<<1, 64>>>(out, sink, 0xA5A5A5A5u, 1.5); k_cmpx<<<1, 64>>>(out, sink, 0xA5A5A5A5u, 1.5); k_cmp_ahead<<<1, 64>>>(out, sink, 0xA5A5A5A5u, 1.5);
    hipDeviceSynchronize();
  }
  unsigned long long h[16]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[] = {"EXEC <- sgpr mask; v_add_f64 (vgpr value)", "EXEC <- sgpr mask; v_add_f64 (sgpr value)", "carry-out mask; EXEC <- mask; add; EXEC <- all",
                      "sign test + 2 cndmask + add", "cvt bit -> 0.0/1.0, fma", "plain dependent v_add_f64", "one EXEC write per 4 adds", "v_bfe bit, cvt -> 0.0/1.0, fma", "v_cmpx (VALU writes EXEC); add; EXEC <- all; shift", "8 v_cmp masks ahead; EXEC <- mask; add"};
  for (int i = 0; i < 10; ++i) printf("%-50s %8.2f ticks/step\n", nm[i], h[i] / (double)N);
  return 0;
}
