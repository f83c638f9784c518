// Shared host-side helpers for liblatte_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define LT_OK 0
#define LT_ERR_ARG (-1)
#define LT_ERR_HIP (-2)
#define LT_ERR_WORKSPACE (-3)
#define LT_ERR_RANGE (-4)

// rocPRIM's radix sort falls back to a merge sort below 1M items: ~17 tiny launches per sort at the sizes of this path
// (2e4..1e6 keys).  Onesweep needs one launch per 8-bit digit plus two, so it is forced above 4096 items.
#ifdef __HIPCC__
#include <rocprim/device/device_radix_sort.hpp>
using LtSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096>;
#endif

// thread-local message returned by lt_last_error()
void lt_set_error(const char* fmt, ...);

#define LT_CHECK_HIP(expr)                                                          \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      lt_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return LT_ERR_HIP;                                                            \
    }                                                                               \
  } while (0)

#define LT_REQUIRE(cond, code, ...)   \
  do {                                \
    if (!(cond)) {                    \
      lt_set_error(__VA_ARGS__);      \
      return (code);                  \
    }                                 \
  } while (0)

#define LT_LAUNCH_CHECK() LT_CHECK_HIP(hipGetLastError())

static inline int64_t lt_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int lt_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// grid for a memory-bound grid-stride kernel: enough blocks to fill 256 CUs x 8, never 0
static inline int lt_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048 * 4) g = 2048 * 4;
  return (int)g;
}

// bump allocator over a caller-provided workspace (256-B aligned pieces)
struct LtArena {
  char* base;
  int64_t size;
  int64_t off;
  LtArena(void* p, int64_t n) : base((char*)p), size(n), off(0) {}
  void* take(int64_t bytes) {
    int64_t o = lt_align_up(off, 256);
    if (o + bytes > size) return nullptr;
    off = o + bytes;
    return base + o;
  }
};

static inline int64_t lt_pow2_ge(int64_t x) {
  int64_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

// conv plan of a neighbour table [K,V] (built by lt_kmap_plan, consumed by lt_spconv_plan_fwd): one caller-owned buffer
//   order [V] int32 | tnbr [T][K][64] int32 | tmask [T][K] uint64 | torder [T] int32,   T = ceil(V/64), 256-B aligned
// torder = the tiles sorted by descending work (active strip-offsets): workgroups are dispatched heaviest first, which
// removes the long tail a mask-sorted order otherwise has (the densest masks sort last).
struct LtPlanView {
  int* order;
  int* tnbr;
  unsigned long long* tmask;
  int* torder;
  int64_t bytes;
};
static inline LtPlanView lt_plan_view(void* mem, int K, int64_t V) {
  const int64_t T = (V + 63) / 64;
  char* p = (char*)mem;
  LtPlanView v;
  int64_t off = 0;
  v.order = (int*)(p + off);
  off += lt_align_up((V > 0 ? V : 1) * 4, 256);
  v.tnbr = (int*)(p + off);
  off += lt_align_up((T > 0 ? T : 1) * K * 64 * 4, 256);
  v.tmask = (unsigned long long*)(p + off);
  off += lt_align_up((T > 0 ? T : 1) * K * 8, 256);
  v.torder = (int*)(p + off);
  off += lt_align_up((T > 0 ? T : 1) * 4, 256);
  v.bytes = off;
  return v;
}

// Statistics workspace of lt_spconv_plan_fwd_stats / lt_bn_finalize_sums:
// partial[nb][2][C] fp32 column sums per conv workgroup | stage[LT_STAT_SLICES][2][C] fp64 slice sums (nb > 2048)
#define LT_STAT_SLICES 32
struct LtStatsView {
  float* partial;
  double* stage;
  int64_t bytes;
};
static inline LtStatsView lt_stats_view(void* mem, int64_t nb, int C) {
  char* p = (char*)mem;
  LtStatsView v;
  int64_t off = 0;
  v.partial = (float*)(p + off);
  off += lt_align_up((nb > 0 ? nb : 1) * 2 * (int64_t)C * 4, 256);
  v.stage = (double*)(p + off);
  off += lt_align_up((int64_t)LT_STAT_SLICES * 2 * C * 8, 256);
  v.bytes = off;
  return v;
}

// ---- internal (C++ linkage) pieces shared between translation units: not part of the C-ABI
#ifdef __HIPCC__
// lt_spconv_plan_fwd_stats that may pick any conv kernel; *nb = the number of partial rows [nb][2][Cout] it wrote
int lt_spconv_plan_fwd_stats_nb(const float* in, const float* packed, const void* plan, float* out, int64_t Vin, int64_t Vout,
                                int32_t Cin, int32_t Cout, int32_t K, void* stats_ws, int64_t* nb, hipStream_t stream);
// lt_bn_finalize_sums over an explicit number of partial rows
int lt_bn_finalize_sums_nb(void* stats_ws, int64_t nb, int64_t V, int32_t C, float* mean, float* var, hipStream_t stream);
// lt_voxelize_bwd over float4 column groups; accumulate != 0: gfeats += (the gradient of a tensor with a second reader)
int lt_voxelize_bwd_acc(const float* gout, const int32_t* idx, const int32_t* counts, float* gfeats, int64_t N, int32_t C,
                        int64_t V, int accumulate, hipStream_t stream);
#endif

// ---- device helpers -------------------------------------------------------
#define LT_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint64_t lt_fnv1a4(int x, int y, int z, int b) {
  // torchsparse v1.4.0 hash kernel semantics (reference call sites
  // latte/models/torchsparse_utils.py:15,40,45,68,73): 64-bit FNV-1a over the four int32,
  // folded to 60 bits.
  uint64_t h = 14695981039346656037ull;
  h ^= (uint32_t)x; h *= 1099511628211ull;
  h ^= (uint32_t)y; h *= 1099511628211ull;
  h ^= (uint32_t)z; h *= 1099511628211ull;
  h ^= (uint32_t)b; h *= 1099511628211ull;
  return (h >> 60) ^ (h & 0xFFFFFFFFFFFFFFFull);
}

__device__ __forceinline__ uint32_t lt_slot_mix(uint64_t k) {
  // murmur3 finalizer: spreads table slots independently of how the key was formed
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k;
}
