// Coordinate hashing, open-addressing hash table in HBM, unique+sort of voxel keys.
//
// Replaces (semantics, not code) torchsparse v1.4.0 F.sphash / F.sphashquery / F.spcount and the
// `torch.unique` / `np.unique` steps around them -- reference call sites:
//   latte/models/torchsparse_utils.py:15-18 (initial_voxelize), :40-47 (point_to_voxel), :68-74 (voxel_to_point)
//   latte/models/hr_module.py:87-89 (np.unique on ravel keys)
//
// Design (MI355X): one flat open-addressing table (uint64 keys | int32 values, SoA) living in a
// caller-provided workspace; inserts are a 64-bit CAS per point, duplicates resolve to the
// smallest point index with atomicMin; occupied slots are compacted with a wave64 ballot (one
// global atomic per wave); only the V unique keys are radix-sorted (rocPRIM), then ranks are
// written back into the table so `inverse` is one more probe per point.
#include "common.h"
#include "table.h"
#include "../../include/latte_hip.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <limits.h>

// ---------------------------------------------------------------- error text
static thread_local char g_err[512] = "";
void lt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* lt_last_error(void) { return g_err; }
extern "C" int lt_version(void) { return 100; }

// ---------------------------------------------------------------- sphash
__global__ void k_sphash(const int* __restrict__ coords, int64_t n, const int* __restrict__ offsets, int K,
                         int64_t* __restrict__ out) {
  const int64_t total = n * (int64_t)(K > 0 ? K : 1);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    // t = k * n + i : consecutive lanes read consecutive rows (16 B each) and write consecutive hashes
    const int64_t i = t % n;
    const int k = (int)(t / n);
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    int x = c.x, y = c.y, z = c.z;
    if (K > 0) { x += offsets[3 * k]; y += offsets[3 * k + 1]; z += offsets[3 * k + 2]; }
    out[t] = (int64_t)lt_fnv1a4(x, y, z, c.w);
  }
}

extern "C" int lt_sphash(const int32_t* coords, int64_t n, const int32_t* offsets, int32_t K, int64_t* out,
                         hipStream_t stream) {
  LT_REQUIRE(n >= 0 && K >= 0, LT_ERR_ARG, "lt_sphash: bad sizes n=%lld K=%d", (long long)n, K);
  if (n == 0) return LT_OK;
  LT_REQUIRE(coords && out && (K == 0 || offsets), LT_ERR_ARG, "lt_sphash: null pointer");
  const int64_t total = n * (K > 0 ? K : 1);
  hipLaunchKernelGGL(k_sphash, dim3(lt_grid(total, 256)), dim3(256), 0, stream, coords, n, offsets, K, out);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- table build / query
extern "C" int64_t lt_table_bytes(int64_t n) { return lt_table_bytes_for(n); }

__global__ void k_table_insert(const int64_t* __restrict__ keys, int64_t n, LtTable t) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    lt_table_insert_min(t, (uint64_t)keys[i], (int)i);
}

__global__ void k_table_query(const int64_t* __restrict__ q, int64_t n, LtTable t, int64_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int s = lt_table_find(t, (uint64_t)q[i]);
    out[i] = s < 0 ? -1 : t.vals[s];
  }
}

int lt_table_clear(const LtTable& t, hipStream_t stream) {
  const int64_t cap = (int64_t)t.mask + 1;
  LT_CHECK_HIP(hipMemsetAsync(t.keys, 0xFF, cap * 8, stream));
  LT_CHECK_HIP(hipMemsetAsync(t.vals, 0x7F, cap * 4, stream));
  return LT_OK;
}

extern "C" int lt_hash_build(const int64_t* keys, int64_t n, void* table, int64_t table_bytes, hipStream_t stream) {
  LT_REQUIRE(n >= 0 && table && (n == 0 || keys), LT_ERR_ARG, "lt_hash_build: bad args");
  LT_REQUIRE(n < INT_MAX, LT_ERR_RANGE, "lt_hash_build: n=%lld exceeds int32 indices", (long long)n);
  LT_REQUIRE(table_bytes >= lt_table_bytes_for(n), LT_ERR_WORKSPACE, "lt_hash_build: table needs %lld bytes, got %lld",
             (long long)lt_table_bytes_for(n), (long long)table_bytes);
  LtTable t = lt_table_view(table, n);
  int rc = lt_table_clear(t, stream);
  if (rc) return rc;
  if (n > 0) {
    hipLaunchKernelGGL(k_table_insert, dim3(lt_grid(n, 256)), dim3(256), 0, stream, keys, n, t);
    LT_LAUNCH_CHECK();
  }
  return LT_OK;
}

extern "C" int lt_hash_query(const int64_t* queries, int64_t q, const void* table, int64_t n_build, int64_t* out,
                             hipStream_t stream) {
  LT_REQUIRE(q >= 0 && n_build >= 0, LT_ERR_ARG, "lt_hash_query: bad sizes");
  if (q == 0) return LT_OK;
  LT_REQUIRE(queries && table && out, LT_ERR_ARG, "lt_hash_query: null pointer");
  LtTable t = lt_table_view(const_cast<void*>(table), n_build);
  hipLaunchKernelGGL(k_table_query, dim3(lt_grid(q, 256)), dim3(256), 0, stream, queries, q, t, out);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- spcount
__global__ void k_spcount(const int* __restrict__ idx, int64_t n, int* __restrict__ out, int64_t num) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = idx[i];
    if (v >= 0 && v < num) atomicAdd(&out[v], 1);
  }
}

extern "C" int lt_spcount(const int32_t* idx, int64_t n, int32_t* out, int64_t num, hipStream_t stream) {
  LT_REQUIRE(n >= 0 && num >= 0, LT_ERR_ARG, "lt_spcount: bad sizes");
  if (num == 0) return LT_OK;
  LT_REQUIRE(out && (n == 0 || idx), LT_ERR_ARG, "lt_spcount: null pointer");
  LT_CHECK_HIP(hipMemsetAsync(out, 0, num * 4, stream));
  if (n > 0) {
    hipLaunchKernelGGL(k_spcount, dim3(lt_grid(n, 256)), dim3(256), 0, stream, idx, n, out, num);
    LT_LAUNCH_CHECK();
  }
  return LT_OK;
}

// ---------------------------------------------------------------- unique + sort
// wave64 ballot compaction of the occupied slots.  A 256-thread block owns LT_COMPACT_ITEMS x 256 consecutive slots
// (coalesced 8-B reads), counts its occupied slots with ballots, reserves its output range with ONE global atomic and
// hands the waves their sub-ranges through LDS -- one atomic per 2048 slots instead of one per 64 (at one per wave the
// kernel ran at 0.08 TB/s: every wave of the grid serialised on the same counter).  Output order is arbitrary (the
// unique keys are radix-sorted afterwards).
#define LT_COMPACT_ITEMS 8
__global__ __launch_bounds__(256) void k_table_compact(LtTable t, uint64_t* __restrict__ out_keys, int* __restrict__ out_vals,
                                                       unsigned int* __restrict__ counter) {
  __shared__ unsigned int s_wave[4];
  __shared__ unsigned int s_base;
  const int64_t cap = (int64_t)t.mask + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t chunk = (int64_t)LT_COMPACT_ITEMS * 256;
  // cap is a power of two >= 1024: every thread of a block runs the same trip count
  for (int64_t base = blockIdx.x * chunk; base < cap; base += (int64_t)gridDim.x * chunk) {
    uint64_t k[LT_COMPACT_ITEMS];
    unsigned long long m[LT_COMPACT_ITEMS];
    unsigned int wave_total = 0;
#pragma unroll
    for (int it = 0; it < LT_COMPACT_ITEMS; ++it) {
      const int64_t s = base + it * 256 + threadIdx.x;
      k[it] = s < cap ? t.keys[s] : LT_EMPTY_KEY;
      m[it] = __ballot(k[it] != LT_EMPTY_KEY);
      wave_total += (unsigned int)__popcll(m[it]);
    }
    if (lane == 0) s_wave[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
      s_base = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    unsigned int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
#pragma unroll
    for (int it = 0; it < LT_COMPACT_ITEMS; ++it) {
      if (k[it] != LT_EMPTY_KEY) {
        const unsigned int pos = off + (unsigned int)__popcll(m[it] & ((1ull << lane) - 1ull));
        out_keys[pos] = k[it];
        out_vals[pos] = t.vals[base + it * 256 + threadIdx.x];
      }
      off += (unsigned int)__popcll(m[it]);
    }
    __syncthreads();  // s_wave / s_base are rewritten by the next chunk
  }
}

__global__ void k_assign_rank(LtTable t, const uint64_t* __restrict__ sorted_keys, int64_t v) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < v; j += (int64_t)gridDim.x * blockDim.x) {
    const int s = lt_table_find(t, sorted_keys[j]);
    if (s >= 0) t.vals[s] = (int)j;
  }
}

__global__ void k_inverse(LtTable t, const int64_t* __restrict__ keys, int64_t n, int64_t* __restrict__ inv) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int s = lt_table_find(t, (uint64_t)keys[i]);
    inv[i] = s < 0 ? -1 : t.vals[s];
  }
}

__global__ void k_widen_i32(const int* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[i];
}

static int64_t sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  uint64_t* k = nullptr;
  int* v = nullptr;
  (void)rocprim::radix_sort_pairs<LtSortConfig>(nullptr, bytes, k, k, v, v, (size_t)(n > 0 ? n : 1), 0, 64, (hipStream_t)0);
  return (int64_t)bytes;
}

extern "C" int64_t lt_unique_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  return lt_table_bytes_for(n) + lt_align_up(n * 8, 256) * 2 + lt_align_up(n * 4, 256) * 2 + 256 +
         lt_align_up(sort_temp_bytes(n), 256) + 4096;
}

extern "C" int lt_unique_sorted(const int64_t* keys, int64_t n, int32_t key_bits, void* ws, int64_t ws_bytes,
                                int64_t* out_keys, int64_t* out_first, int64_t* out_inverse, int64_t* count,
                                hipStream_t stream) {
  LT_REQUIRE(count, LT_ERR_ARG, "lt_unique_sorted: count pointer is null");
  *count = 0;
  LT_REQUIRE(n >= 0 && n < INT_MAX, LT_ERR_RANGE, "lt_unique_sorted: n=%lld out of range", (long long)n);
  if (n == 0) return LT_OK;
  LT_REQUIRE(keys && ws, LT_ERR_ARG, "lt_unique_sorted: null pointer");
  LT_REQUIRE(ws_bytes >= lt_unique_workspace_bytes(n), LT_ERR_WORKSPACE,
             "lt_unique_sorted: workspace needs %lld bytes, got %lld", (long long)lt_unique_workspace_bytes(n),
             (long long)ws_bytes);
  if (key_bits <= 0 || key_bits > 64) key_bits = 64;
  LtArena a(ws, ws_bytes);
  void* tmem = a.take(lt_table_bytes_for(n));
  uint64_t* ck = (uint64_t*)a.take(n * 8);   // compacted (unsorted) unique keys
  int* cv = (int*)a.take(n * 4);             // their first indices
  uint64_t* sk = (uint64_t*)a.take(n * 8);   // sorted keys
  int* sv = (int*)a.take(n * 4);             // first index, sorted
  unsigned int* counter = (unsigned int*)a.take(256);
  const int64_t tb = sort_temp_bytes(n);
  void* stmp = a.take(tb);
  LT_REQUIRE(tmem && ck && cv && sk && sv && counter && stmp, LT_ERR_WORKSPACE, "lt_unique_sorted: arena exhausted");

  LtTable t = lt_table_view(tmem, n);
  int rc = lt_table_clear(t, stream);
  if (rc) return rc;
  LT_CHECK_HIP(hipMemsetAsync(counter, 0, 4, stream));
  hipLaunchKernelGGL(k_table_insert, dim3(lt_grid(n, 256)), dim3(256), 0, stream, keys, n, t);
  LT_LAUNCH_CHECK();
  const int64_t cap = (int64_t)t.mask + 1;
  hipLaunchKernelGGL(k_table_compact, dim3(lt_grid(cap / LT_COMPACT_ITEMS, 256)), dim3(256), 0, stream, t, ck, cv, counter);
  LT_LAUNCH_CHECK();
  unsigned int v_host = 0;
  LT_CHECK_HIP(hipMemcpyAsync(&v_host, counter, 4, hipMemcpyDeviceToHost, stream));
  LT_CHECK_HIP(hipStreamSynchronize(stream));
  const int64_t V = v_host;
  LT_REQUIRE(V >= 1 && V <= n, LT_ERR_RANGE, "lt_unique_sorted: impossible unique count %lld of %lld", (long long)V,
             (long long)n);
  size_t tbs = (size_t)tb;
  LT_CHECK_HIP(rocprim::radix_sort_pairs<LtSortConfig>(stmp, tbs, ck, sk, cv, sv, (size_t)V, 0, (unsigned)key_bits, stream));
  if (out_keys) LT_CHECK_HIP(hipMemcpyAsync(out_keys, sk, V * 8, hipMemcpyDeviceToDevice, stream));
  if (out_first) {
    hipLaunchKernelGGL(k_widen_i32, dim3(lt_grid(V, 256)), dim3(256), 0, stream, sv, V, out_first);
    LT_LAUNCH_CHECK();
  }
  if (out_inverse) {
    hipLaunchKernelGGL(k_assign_rank, dim3(lt_grid(V, 256)), dim3(256), 0, stream, t, sk, V);
    LT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_inverse, dim3(lt_grid(n, 256)), dim3(256), 0, stream, t, keys, n, out_inverse);
    LT_LAUNCH_CHECK();
  }
  *count = V;
  return LT_OK;
}

// ---------------------------------------------------------------- lexicographic xyz keys
// coords must lie in [0, 2^20): the host wrapper validates min/max before calling (a violation would alias keys).
__global__ void k_pack_xyz(const int* __restrict__ c, int64_t n, int64_t* __restrict__ keys) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = c[3 * i] & 0xFFFFF, y = c[3 * i + 1] & 0xFFFFF, z = c[3 * i + 2] & 0xFFFFF;
    keys[i] = ((int64_t)x << 40) | ((int64_t)y << 20) | (int64_t)z;
  }
}

extern "C" int lt_pack_xyz_keys(const int32_t* coords3, int64_t n, int64_t* keys, hipStream_t stream) {
  LT_REQUIRE(n >= 0, LT_ERR_ARG, "lt_pack_xyz_keys: bad n");
  if (n == 0) return LT_OK;
  LT_REQUIRE(coords3 && keys, LT_ERR_ARG, "lt_pack_xyz_keys: null pointer");
  hipLaunchKernelGGL(k_pack_xyz, dim3(lt_grid(n, 256)), dim3(256), 0, stream, coords3, n, keys);
  LT_LAUNCH_CHECK();
  return LT_OK;
}
