// pf_device.h -- wave64 device primitives for gfx950 (MI355X): keyed RNG with the
// CPython 3.10 derived distributions, DPP wave reductions, grid helpers.
// One wavefront (= one 64-thread workgroup) serves one agent; "uniform" below
// means the value is identical in all 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PF_DEV __device__ __forceinline__
#define PF_INF (__builtin_huge_val())
#define PF_SQRT2 1.4142135623730951 /* math.sqrt(2), correctly rounded */

namespace pf {

// ---------------------------------------------------------------------------
// per-cell search record, one array of R*C per resident agent slot, in HBM.
// g/parent/flags are valid iff tag == the slot's current solve epoch, so a new
// search never clears the array.  The avoid mark (top 24 bits
// of meta) is valid iff it equals the slot's current eval epoch.
// ---------------------------------------------------------------------------
struct __attribute__((aligned(16))) Rec {
  double g;
  uint32_t tagmm;  // [7:0] static move mask of the cell (helper order, this handle's diagonal policy), [31:8] solve epoch
  uint32_t meta;   // [2:0] parent move, [3] closed, [4] in-open, [31:18] avoid epoch
};
#define PF_M_PARENT 7u
#define PF_M_CLOSED 8u
#define PF_M_INOPEN 16u
#define PF_AVOID_SHIFT 18
#define PF_AVOID_KEEP 0xFFFC0000u
#define PF_POOL_STRIDE ((257 * 1024 + 16384) * 20) /* bytes of open-list HBM scratch per resident agent slot (bucket pool / tier 2) */
#define PF_TAG_SHIFT 8

// helper.py:30-36 / MPA.py:71-77 move order
__device__ __constant__ const int8_t HM_DR[8] = {0, 0, 1, -1, 1, 1, -1, -1};
__device__ __constant__ const int8_t HM_DC[8] = {1, -1, 0, 0, 1, -1, 1, -1};
// MAACO.py:98 move order, and the helper-order bit each one corresponds to
__device__ __constant__ const int8_t AM_DR[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
__device__ __constant__ const int8_t AM_DC[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
__device__ __constant__ const int8_t AM_TO_HM[8] = {7, 3, 6, 1, 0, 5, 2, 4};

struct Grid {
  const uint8_t* occ;     // R*C, 1 = obstacle
  const uint8_t* mm;      // static move mask (helper order) under the call's diagonal policy
  const uint8_t* d2near;  // min(d^2 to nearest obstacle, 255) within radius 7
  const int* comp;        // connected-component label of every free cell under `mm` (obstacles: -1); may be null
  int R, C;
  uint64_t magicC;        // floor(2^40 / C) + 1 : cell / C == (cell * magicC) >> 40 for cell < 2^24
  long long step_cap;     // > 0: lowers the connectors' step cap (tests of the cap path; 0 = the reference's 3RC / 2RC)
};
PF_DEV int row_of(const Grid& G, int cell) { return (int)(((uint64_t)(uint32_t)cell * G.magicC) >> 40); }

PF_DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
PF_DEV int bcast_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
PF_DEV double bcast_d(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// Predicates as wave masks.  The ballot of ONE compare is that compare's lane mask (the v_cmp writes it), but for a compound
// predicate `a & b` the compiler ANDs the two masks, turns the result into a 0/1 vector register and compares that with zero
// again: two vector instructions and a scalar wait per ballot (thirteen per round of k_maaco_walk8, eight per A* trip).  So
// compound predicates are built from the masks of their single compares, B(a) & B(b), in scalar registers, and turned back into
// a lane predicate -- for a select, a store or a branch -- by PL(mask), which costs nothing (the mask IS the condition operand).
// A negated term needs a positive one beside it where lanes may be switched off: ~B(x) holds those lanes too.
typedef unsigned long long pf_u64;
PF_DEV pf_u64 B(bool p) { return __builtin_amdgcn_ballot_w64(p); }
PF_DEV bool PL(pf_u64 m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
PF_DEV int first_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
PF_DEV uint64_t first_u64(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// DPP move of a double (two 32-bit halves); lanes whose row is masked off keep `v`.
template <int CTRL, int ROW_MASK>
PF_DEV double dpp_d(double v) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
PF_DEV double dmin(double a, double b) { return a < b ? a : b; }
// min over the 64 lanes of non-NaN doubles; result uniform.  min is idempotent,
// so rotations inside each 16-lane row then the gfx9 row broadcasts suffice:
// 6 DPP steps, no LDS traffic (ds_bpermute-based __shfl_xor costs an LDS round trip per step).
PF_DEV double wave_min_d(double v) {
  v = dmin(v, dpp_d<0x121, 0xF>(v));  // row_ror:1
  v = dmin(v, dpp_d<0x122, 0xF>(v));  // row_ror:2
  v = dmin(v, dpp_d<0x124, 0xF>(v));  // row_ror:4
  v = dmin(v, dpp_d<0x128, 0xF>(v));  // row_ror:8   -> every lane holds its row's min
  v = dmin(v, dpp_d<0x142, 0xA>(v));  // row_bcast:15 into rows 1,3
  v = dmin(v, dpp_d<0x143, 0xC>(v));  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave min
  return bcast_d(v, 63);
}
// unsigned min with the DPP modifier folded into v_min_u32 (old = identity so the combiner can fold)
template <int CTRL, int ROW_MASK>
PF_DEV unsigned dpp_umin(unsigned v) {
  unsigned m = (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, ROW_MASK, 0xF, false);
  return m < v ? m : v;
}
PF_DEV unsigned wave_min_u32(unsigned v) {
  v = dpp_umin<0x121, 0xF>(v);
  v = dpp_umin<0x122, 0xF>(v);
  v = dpp_umin<0x124, 0xF>(v);
  v = dpp_umin<0x128, 0xF>(v);
  v = dpp_umin<0x142, 0xA>(v);
  v = dpp_umin<0x143, 0xC>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
PF_DEV int wave_sum_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8 -> lane 15 of each row = row sum
  int s = bcast_i(v, 15) + bcast_i(v, 31) + bcast_i(v, 47) + bcast_i(v, 63);
  return s;
}

// sum over the 64 lanes of a double (fixed DPP pattern: deterministic, but NOT the reference's left-to-right order --
// use only where a tolerance is applied); result uniform
PF_DEV double wave_sum_d(double v) {
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xF, 0xF, true));
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x112, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x112, 0xF, 0xF, true));
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x114, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x114, 0xF, 0xF, true));
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x118, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x118, 0xF, 0xF, true));
  return (bcast_d(v, 15) + bcast_d(v, 31)) + (bcast_d(v, 47) + bcast_d(v, 63));   // lane 15 of each row = the row's sum
}

// ---------------------------------------------------------------------------
// keyed counter RNG (twin of pathfit/rng.py AgentRandom) + CPython derivations
// ---------------------------------------------------------------------------
PF_DEV uint64_t mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27; z *= 0x94D049BB133111EBULL;
  z ^= z >> 31; return z;
}
struct Rng {
  uint64_t key, ctr;
  uint64_t kc;      // key + ctr * GOLDEN, kept by addition: word i of the stream is mix64(key + i * GOLDEN), and a 64-bit multiply per draw
                    // (four quarter-rate 32-bit multiplies) is what the counter form costs
  PF_DEV void init(uint64_t seed, uint64_t dom, uint64_t it, uint64_t agent) {
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * (dom + 1));
    k = mix64(k + 0xD1B54A32D192ED03ULL * (it + 1));
    k = mix64(k + 0x8CB92BA72F3D8DD7ULL * (agent + 1));
    key = k; ctr = 0; kc = k;
  }
  PF_DEV void set_ctr(uint64_t c) { ctr = c; kc = key + c * 0x9E3779B97F4A7C15ULL; }
  PF_DEV void advance(uint64_t n) { ctr += n; kc += n * 0x9E3779B97F4A7C15ULL; }   // (n is a literal at every call site)
  PF_DEV uint64_t next64() { ctr += 1; kc += 0x9E3779B97F4A7C15ULL; return mix64(kc); }
  // the word `ahead` draws from now, without consuming it (callers that mix ahead of a decision advance() themselves)
  PF_DEV uint64_t peek64(uint64_t ahead) const { return mix64(kc + ahead * 0x9E3779B97F4A7C15ULL); }
  static PF_DEV double to_unit(uint64_t w) { return (double)(w >> 11) * (1.0 / 9007199254740992.0); }
  PF_DEV double random() { return to_unit(next64()); }
  // random.py _randbelow_with_getrandbits (n >= 1): draws even when n == 1
  PF_DEV uint64_t randbelow(uint64_t n) {
    int k = 64 - __builtin_clzll(n);
    uint64_t r = next64() >> (64 - k);
    while (r >= n) r = next64() >> (64 - k);
    return r;
  }
  // the same, the first word already drawn
  PF_DEV uint64_t randbelow_from(uint64_t first, uint64_t n) {
    int k = 64 - __builtin_clzll(n);
    uint64_t r = first >> (64 - k);
    while (r >= n) r = next64() >> (64 - k);
    return r;
  }
  PF_DEV int64_t randint(int64_t a, int64_t b) { return a + (int64_t)randbelow((uint64_t)(b - a + 1)); }
  PF_DEV double uniform(double a, double b) { return a + (b - a) * random(); }
  // random.py normalvariate (Kinderman-Monahan)
  PF_DEV double normalvariate(double mu, double sigma) {
    double z;
    for (;;) {
      double u1 = random();
      double u2 = 1.0 - random();
      z = 1.7155277699214135 * (u1 - 0.5) / u2;
      double zz = z * z / 4.0;
      if (zz <= -log(u2)) break;
    }
    return mu + z * sigma;
  }
};

// index of the j-th (0-based) set bit of m
PF_DEV int nth_set_bit(uint32_t m, int j) {
  for (int i = 0; i < j; ++i) m &= m - 1;
  return __builtin_ctz(m);
}

}  // namespace pf
