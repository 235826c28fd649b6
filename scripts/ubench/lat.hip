// Latency of dependent instruction chains on one wavefront (shader clocks per op): hipcc --offload-arch=gfx950 -O3 lat.hip -o lat && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 2048
#define T0 const unsigned long long t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#define T1(i) __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[i] = t1 - t0;
__global__ void k_f64add(unsigned long long* out, double* sink, double y) { double x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { x = x + y; asm volatile("" : "+v"(x)); } T1(0) sink[threadIdx.x] = x; }
__global__ void k_f64cmp(unsigned long long* out, double* sink, double y) { double x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { x = (x < y) ? y : x; asm volatile("" : "+v"(x)); y = -y; } T1(1) sink[threadIdx.x] = x; }
__global__ void k_u64cmp(unsigned long long* out, double* sink, unsigned long long y) { unsigned long long x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { x = (x < y) ? y : x; asm volatile("" : "+v"(x)); y = ~y; } T1(2) sink[threadIdx.x] = (double)x; }
__global__ void k_u32cmp(unsigned long long* out, double* sink, unsigned y) { unsigned x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { x = (x < y) ? y : x; asm volatile("" : "+v"(x)); y = ~y; } T1(3) sink[threadIdx.x] = (double)x; }
__global__ void k_bperm(unsigned long long* out, double* sink) { int x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { x = __builtin_amdgcn_ds_bpermute(((x + 1) & 63) << 2, x); } T1(4) sink[threadIdx.x] = x; }
__global__ void k_readlane(unsigned long long* out, double* sink) { int x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { int s = __builtin_amdgcn_readlane(x, 5); x = x + s; asm volatile("" : "+v"(x)); } T1(5) sink[threadIdx.x] = x; }
__global__ void k_dpp(unsigned long long* out, double* sink) { int x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { x = __builtin_amdgcn_update_dpp(x, x, 0x111, 0xF, 0xF, false) + 1; asm volatile("" : "+v"(x)); } T1(6) sink[threadIdx.x] = x; }
__global__ void k_f64fma(unsigned long long* out, double* sink, double y) { double x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { x = __builtin_fma(x, y, y); asm volatile("" : "+v"(x)); } T1(7) sink[threadIdx.x] = x; }
__global__ void k_f64sqrt(unsigned long long* out, double* sink, double y) { double x = threadIdx.x + 2.0; T0 for (int i = 0; i < N; ++i) { x = __builtin_sqrt(x) + y; asm volatile("" : "+v"(x)); } T1(8) sink[threadIdx.x] = x; }
__global__ void k_f64div(unsigned long long* out, double* sink, double y) { double x = threadIdx.x + 2.0; T0 for (int i = 0; i < N; ++i) { x = y / x + y; asm volatile("" : "+v"(x)); } T1(9) sink[threadIdx.x] = x; }
__global__ void k_u32add(unsigned long long* out, double* sink, unsigned y) { unsigned x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { x = x + y; asm volatile("" : "+v"(x)); } T1(10) sink[threadIdx.x] = (double)x; }
__global__ void k_ldsrw(unsigned long long* out, double* sink) { __shared__ int s[64]; s[threadIdx.x] = threadIdx.x; int x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { x = s[(x + 1) & 63]; asm volatile("" : "+v"(x)); } T1(11) sink[threadIdx.x] = x; }
__global__ void k_ballot(unsigned long long* out, double* sink) { int x = threadIdx.x; T0 for (int i = 0; i < N; ++i) { unsigned long long b = __ballot(x & 1); x = x + (int)__builtin_popcountll(b); asm volatile("" : "+v"(x)); } T1(12) sink[threadIdx.x] = x; }
__global__ void k_gload(unsigned long long* out, double* sink, const int* chain) { int x = threadIdx.x; T0 for (int i = 0; i < 256; ++i) { x = chain[x]; } T1(13) sink[threadIdx.x] = x; }
__global__ void k_u32indep(unsigned long long* out, double* sink, unsigned y) { unsigned a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3; T0 for (int i = 0; i < N / 4; ++i) { a += y; b += y; c += y; d += y; asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); } T1(14) sink[threadIdx.x] = (double)(a + b + c + d); }
__global__ void k_f64indep(unsigned long long* out, double* sink, double y) { double a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3; T0 for (int i = 0; i < N / 4; ++i) { a += y; b += y; c += y; d += y; asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); } T1(15) sink[threadIdx.x] = a + b + c + d; }
__global__ void k_f64cmpindep(unsigned long long* out, double* sink, double y) { double a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3; int n = 0; T0 for (int i = 0; i < N / 4; ++i) { n += (a < y) + (b < y) + (c < y) + (d < y); y = -y; asm volatile("" : "+v"(n)); } T1(16) sink[threadIdx.x] = n + a + b + c + d; }
int main() {
  unsigned long long* out; double* sink; int* chain;
  hipMalloc(&out, 8 * 32); hipMalloc(&sink, 8 * 64); hipMemset(out, 0, 8 * 32);
  const int CH = 1 << 24; hipMalloc(&chain, 4 * (size_t)CH);
  { int* h = (int*)malloc(4 * (size_t)CH); for (long i = 0; i < CH; ++i) h[i] = (int)((i * 1048583L + 12345) % CH); hipMemcpy(chain, h, 4 * (size_t)CH, hipMemcpyHostToDevice); free(h); }
  for (int rep = 0; rep < 2; ++rep) {
    k_f64add<<<1, 64>>>(out, sink, 1.5); k_f64cmp<<<1, 64>>>(out, sink, 3.0); k_u64cmp<<<1, 64>>>(out, sink, 77ull); k_u32cmp<<<1, 64>>>(out, sink, 77u);
    k_bperm<<<1, 64>>>(out, sink); k_readlane<<<1, 64>>>(out, sink); k_dpp<<<1, 64>>>(out, sink); k_f64fma<<<1, 64>>>(out, sink, 0.5);
    k_f64sqrt<<<1, 64>>>(out, sink, 1.0); k_f64div<<<1, 64>>>(out, sink, 3.0); k_u32add<<<1, 64>>>(out, sink, 3u); k_ldsrw<<<1, 64>>>(out, sink);
    k_ballot<<<1, 64>>>(out, sink); k_gload<<<1, 64>>>(out, sink, chain); k_u32indep<<<1, 64>>>(out, sink, 3u); k_f64indep<<<1, 64>>>(out, sink, 1.5); k_f64cmpindep<<<1, 64>>>(out, sink, 1.5);
    hipDeviceSynchronize();
  }
  unsigned long long h[32]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[] = {"f64 add (dependent)", "f64 cmp + select (dependent)", "u64 cmp + select", "u32 cmp + select", "ds_bpermute (dependent)", "v_readlane + add", "DPP row_shr + add",
                      "f64 fma", "f64 sqrt + add", "f64 div + add", "u32 add (dependent)", "LDS read (dependent)", "ballot + popcount + add", "global load chain (16M ints, x256)", "u32 add x4 independent (per op)", "f64 add x4 independent (per op)", "f64 cmp x4 independent (per cmp)"};
  // s_memtime ticks (the stamps of the pop loop use the same counter)
  for (int i = 0; i < 17; ++i) { const double n = i == 13 ? 256.0 : (double)N; printf("%-40s %8.2f ticks/op\n", nm[i], h[i] / n); }
  return 0;
}
