// Kernel-map construction for spnn.Conv3d (torchsparse v1.4.0 conv3d internals; reference call sites
// latte/models/basic_blocks.py:17,34,51,58,68 and latte/models/spvcnn.py:30,32).
//
// The product format is the dense neighbour table results[K,Vout] (row of the input voxel or -1): the
// output-stationary conv kernel consumes it directly, so no (in,out) pair list, no nonzero(), and no
// device->host copy of nbsizes sits on the hot path.  lt_kmap_pairs() materialises torchsparse's
// (nbmaps, nbsizes) from the table on demand (API compatibility + bit-exact parity tests).
#include "common.h"
#include "table.h"
#include "../../include/latte_hip.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <limits.h>

// ---------------------------------------------------------------- spdownsample
// key = (b | x | y | z), 16 bits each, xyz biased by 32768 so that unsigned key order == (b,x,y,z) order
__device__ __forceinline__ int floordiv(int a, int s) {
  int q = a / s;
  if ((a % s != 0) && ((a < 0) != (s < 0))) --q;
  return q;
}

__global__ void k_downsample_keys(const int* __restrict__ coords, int64_t V, int s, int64_t* __restrict__ keys,
                                  int* __restrict__ bad) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const int x = floordiv(c.x, s) * s, y = floordiv(c.y, s) * s, z = floordiv(c.z, s) * s;
    if (x < -32768 || x > 32767 || y < -32768 || y > 32767 || z < -32768 || z > 32767 || c.w < 0 || c.w > 32767)
      *bad = 1;
    keys[i] = ((int64_t)(c.w & 0x7FFF) << 48) | ((int64_t)((x + 32768) & 0xFFFF) << 32) |
              ((int64_t)((y + 32768) & 0xFFFF) << 16) | (int64_t)((z + 32768) & 0xFFFF);
  }
}

__global__ void k_unpack_bxyz(const int64_t* __restrict__ keys, int64_t V, int* __restrict__ coords) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t k = (uint64_t)keys[i];
    int4 c;
    c.w = (int)(k >> 48);
    c.x = (int)((k >> 32) & 0xFFFF) - 32768;
    c.y = (int)((k >> 16) & 0xFFFF) - 32768;
    c.z = (int)(k & 0xFFFF) - 32768;
    reinterpret_cast<int4*>(coords)[i] = c;
  }
}

extern "C" int64_t lt_downsample_workspace_bytes(int64_t V) {
  if (V < 1) V = 1;
  return lt_unique_workspace_bytes(V) + 2 * lt_align_up(V * 8, 256) + 512;
}

extern "C" int lt_downsample_coords(const int32_t* coords, int64_t V, int32_t sample_stride, void* ws,
                                    int64_t ws_bytes, int32_t* out_coords, int64_t* count, hipStream_t stream) {
  LT_REQUIRE(count, LT_ERR_ARG, "lt_downsample_coords: count pointer is null");
  *count = 0;
  LT_REQUIRE(V >= 0 && sample_stride >= 1, LT_ERR_ARG, "lt_downsample_coords: bad sizes");
  if (V == 0) return LT_OK;
  LT_REQUIRE(coords && ws && out_coords, LT_ERR_ARG, "lt_downsample_coords: null pointer");
  LT_REQUIRE(ws_bytes >= lt_downsample_workspace_bytes(V), LT_ERR_WORKSPACE,
             "lt_downsample_coords: workspace needs %lld bytes, got %lld",
             (long long)lt_downsample_workspace_bytes(V), (long long)ws_bytes);
  LtArena a(ws, ws_bytes);
  int64_t* keys = (int64_t*)a.take(V * 8);
  int64_t* ukeys = (int64_t*)a.take(V * 8);
  int* bad = (int*)a.take(256);
  const int64_t uws_bytes = lt_unique_workspace_bytes(V);
  void* uws = a.take(uws_bytes);
  LT_REQUIRE(keys && ukeys && bad && uws, LT_ERR_WORKSPACE, "lt_downsample_coords: arena exhausted");
  LT_CHECK_HIP(hipMemsetAsync(bad, 0, 4, stream));
  hipLaunchKernelGGL(k_downsample_keys, dim3(lt_grid(V, 256)), dim3(256), 0, stream, coords, V, sample_stride, keys,
                     bad);
  LT_LAUNCH_CHECK();
  int64_t n_unique = 0;
  int rc = lt_unique_sorted(keys, V, 63, uws, uws_bytes, ukeys, nullptr, nullptr, &n_unique, stream);
  if (rc) return rc;
  int bad_host = 0;
  LT_CHECK_HIP(hipMemcpyAsync(&bad_host, bad, 4, hipMemcpyDeviceToHost, stream));
  hipLaunchKernelGGL(k_unpack_bxyz, dim3(lt_grid(n_unique, 256)), dim3(256), 0, stream, ukeys, n_unique, out_coords);
  LT_LAUNCH_CHECK();
  LT_CHECK_HIP(hipStreamSynchronize(stream));
  LT_REQUIRE(!bad_host, LT_ERR_RANGE,
             "lt_downsample_coords: coordinate outside [-32768,32767] or batch index outside [0,32767]");
  *count = n_unique;
  return LT_OK;
}

// ---------------------------------------------------------------- neighbour table
__global__ void k_kmap_insert(const int* __restrict__ coords, int64_t V, LtTable t) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    lt_table_insert_min(t, lt_fnv1a4(c.x, c.y, c.z, c.w), (int)i);
  }
}

// one thread per (k, o); o fastest so the coords_out reads and the results writes are coalesced
__global__ void k_kmap_query(const int* __restrict__ coords_out, int64_t Vout, const int* __restrict__ offsets, int K,
                             LtTable t, int* __restrict__ results) {
  const int64_t total = Vout * (int64_t)K;
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(g / Vout);
    const int64_t o = g - (int64_t)k * Vout;
    const int4 c = reinterpret_cast<const int4*>(coords_out)[o];
    const uint64_t h = lt_fnv1a4(c.x + offsets[3 * k], c.y + offsets[3 * k + 1], c.z + offsets[3 * k + 2], c.w);
    const int s = lt_table_find(t, h);
    results[g] = s < 0 ? -1 : t.vals[s];
  }
}

extern "C" int64_t lt_kmap_workspace_bytes(int64_t Vin) { return lt_table_bytes_for(Vin) + 256; }

extern "C" int lt_kmap_build(const int32_t* coords_in, int64_t Vin, const int32_t* coords_out, int64_t Vout,
                             const int32_t* offsets, int32_t K, void* ws, int64_t ws_bytes, int32_t* results,
                             hipStream_t stream) {
  LT_REQUIRE(Vin >= 0 && Vout >= 0 && K >= 1, LT_ERR_ARG, "lt_kmap_build: bad sizes");
  LT_REQUIRE(Vin < INT_MAX && Vout < INT_MAX, LT_ERR_RANGE, "lt_kmap_build: more than 2^31 voxels");
  if (Vout == 0) return LT_OK;
  LT_REQUIRE(coords_out && offsets && ws && results && (Vin == 0 || coords_in), LT_ERR_ARG,
             "lt_kmap_build: null pointer");
  LT_REQUIRE(ws_bytes >= lt_kmap_workspace_bytes(Vin), LT_ERR_WORKSPACE,
             "lt_kmap_build: workspace needs %lld bytes, got %lld", (long long)lt_kmap_workspace_bytes(Vin),
             (long long)ws_bytes);
  LtTable t = lt_table_view(ws, Vin);
  int rc = lt_table_clear(t, stream);
  if (rc) return rc;
  if (Vin > 0) {
    hipLaunchKernelGGL(k_kmap_insert, dim3(lt_grid(Vin, 256)), dim3(256), 0, stream, coords_in, Vin, t);
    LT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_kmap_query, dim3(lt_grid(Vout * K, 256)), dim3(256), 0, stream, coords_out, Vout, offsets, K, t,
                     results);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// Submanifold map (coords_in IS coords_out, unique rows; K odd, offsets[K-1-k] = -offsets[k]): row i is the neighbour of
// row o at offset k exactly when o is the neighbour of i at offset K-1-k, so one probe fills two entries.  One thread
// per (k <= K/2, o): rows k < K/2 are probed and mirrored, the centre row is probed as it is; rows k > K/2 start at -1
// (cleared by the caller) and receive only mirrored hits -- each entry has one writer because the rows are unique.
__global__ void k_kmap_query_sym(const int* __restrict__ coords, int64_t V, const int* __restrict__ offsets, int K, LtTable t,
                                 int* __restrict__ results) {
  const int half = K / 2;
  const int64_t total = V * (int64_t)(half + 1);
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(g / V);
    const int64_t o = g - (int64_t)k * V;
    const int4 c = reinterpret_cast<const int4*>(coords)[o];
    const uint64_t h = lt_fnv1a4(c.x + offsets[3 * k], c.y + offsets[3 * k + 1], c.z + offsets[3 * k + 2], c.w);
    const int s = lt_table_find(t, h);
    const int i = s < 0 ? -1 : t.vals[s];
    results[g] = i;
    if (i >= 0 && k < half) results[(int64_t)(K - 1 - k) * V + i] = (int)o;
  }
}

extern "C" int lt_kmap_build_sub(const int32_t* coords, int64_t V, const int32_t* offsets, int32_t K, void* ws,
                                 int64_t ws_bytes, int32_t* results, hipStream_t stream) {
  LT_REQUIRE(V >= 0 && K >= 1 && K % 2 == 1, LT_ERR_ARG, "lt_kmap_build_sub: bad sizes (K = %d must be odd)", K);
  LT_REQUIRE(V < INT_MAX, LT_ERR_RANGE, "lt_kmap_build_sub: more than 2^31 voxels");
  if (V == 0) return LT_OK;
  LT_REQUIRE(coords && offsets && ws && results, LT_ERR_ARG, "lt_kmap_build_sub: null pointer");
  LT_REQUIRE(ws_bytes >= lt_kmap_workspace_bytes(V), LT_ERR_WORKSPACE, "lt_kmap_build_sub: workspace needs %lld bytes, got %lld",
             (long long)lt_kmap_workspace_bytes(V), (long long)ws_bytes);
  LtTable t = lt_table_view(ws, V);
  int rc = lt_table_clear(t, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_kmap_insert, dim3(lt_grid(V, 256)), dim3(256), 0, stream, coords, V, t);
  LT_LAUNCH_CHECK();
  const int half = K / 2;
  if (half == 0 || V < 32768) {  // small levels: the extra clear costs more than 13 probes per row save (same table)
    hipLaunchKernelGGL(k_kmap_query, dim3(lt_grid(V * K, 256)), dim3(256), 0, stream, coords, V, offsets, K, t, results);
    LT_LAUNCH_CHECK();
    return LT_OK;
  }
  if (half > 0) LT_CHECK_HIP(hipMemsetAsync(results + (int64_t)(half + 1) * V, 0xFF, (size_t)half * V * sizeof(int32_t), stream));
  hipLaunchKernelGGL(k_kmap_query_sym, dim3(lt_grid(V * (half + 1), 256)), dim3(256), 0, stream, coords, V, offsets, K, t,
                     results);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- (nbmaps, nbsizes) in torchsparse order
struct HitFlag {
  __device__ int operator()(int r) const { return r >= 0 ? 1 : 0; }
};

__global__ void k_emit_pairs(const int* __restrict__ results, const int* __restrict__ pos, int K, int64_t Vout,
                             int* __restrict__ nbmaps, int* __restrict__ nbsizes, int* __restrict__ total) {
  const int64_t n = Vout * (int64_t)K;
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
    const int r = results[g];
    const int p = pos[g];
    const int64_t o = g % Vout;
    if (r >= 0) {
      nbmaps[2 * (int64_t)p] = r;
      nbmaps[2 * (int64_t)p + 1] = (int)o;
    }
    if (o == Vout - 1) {  // last column of row k: close the row
      const int k = (int)(g / Vout);
      const int end = p + (r >= 0 ? 1 : 0);
      const int begin = pos[(int64_t)k * Vout];
      nbsizes[k] = end - begin;
      if (k == K - 1) *total = end;
    }
  }
}

static int64_t scan_temp_bytes(int64_t n) {
  size_t bytes = 0;
  int* p = nullptr;
  auto it = rocprim::make_transform_iterator((const int*)nullptr, HitFlag());
  (void)rocprim::exclusive_scan(nullptr, bytes, it, p, 0, (size_t)(n > 0 ? n : 1), rocprim::plus<int>(), (hipStream_t)0);
  return (int64_t)bytes;
}

extern "C" int64_t lt_kmap_pairs_workspace_bytes(int64_t K, int64_t Vout) {
  const int64_t n = (K > 0 ? K : 1) * (Vout > 0 ? Vout : 1);
  return lt_align_up(n * 4, 256) + lt_align_up(scan_temp_bytes(n), 256) + 1024;
}

extern "C" int lt_kmap_pairs(const int32_t* results, int32_t K, int64_t Vout, void* ws, int64_t ws_bytes,
                             int32_t* nbmaps, int32_t* nbsizes, int64_t* M, hipStream_t stream) {
  LT_REQUIRE(M, LT_ERR_ARG, "lt_kmap_pairs: M pointer is null");
  *M = 0;
  LT_REQUIRE(K >= 1 && Vout >= 0, LT_ERR_ARG, "lt_kmap_pairs: bad sizes");
  LT_REQUIRE(nbsizes, LT_ERR_ARG, "lt_kmap_pairs: null pointer");
  if (Vout == 0) {
    LT_CHECK_HIP(hipMemsetAsync(nbsizes, 0, K * 4, stream));
    return LT_OK;
  }
  const int64_t n = (int64_t)K * Vout;
  LT_REQUIRE(n < INT_MAX, LT_ERR_RANGE, "lt_kmap_pairs: K*Vout exceeds int32");
  LT_REQUIRE(results && ws && nbmaps, LT_ERR_ARG, "lt_kmap_pairs: null pointer");
  LT_REQUIRE(ws_bytes >= lt_kmap_pairs_workspace_bytes(K, Vout), LT_ERR_WORKSPACE,
             "lt_kmap_pairs: workspace needs %lld bytes, got %lld", (long long)lt_kmap_pairs_workspace_bytes(K, Vout),
             (long long)ws_bytes);
  LtArena a(ws, ws_bytes);
  int* pos = (int*)a.take(n * 4);
  int* total = (int*)a.take(256);
  const int64_t tb = scan_temp_bytes(n);
  void* tmp = a.take(tb);
  LT_REQUIRE(pos && total && tmp, LT_ERR_WORKSPACE, "lt_kmap_pairs: arena exhausted");
  size_t tbs = (size_t)tb;
  auto it = rocprim::make_transform_iterator(results, HitFlag());
  LT_CHECK_HIP(rocprim::exclusive_scan(tmp, tbs, it, pos, 0, (size_t)n, rocprim::plus<int>(), stream));
  hipLaunchKernelGGL(k_emit_pairs, dim3(lt_grid(n, 256)), dim3(256), 0, stream, results, pos, K, Vout, nbmaps, nbsizes,
                     total);
  LT_LAUNCH_CHECK();
  int total_host = 0;
  LT_CHECK_HIP(hipMemcpyAsync(&total_host, total, 4, hipMemcpyDeviceToHost, stream));
  LT_CHECK_HIP(hipStreamSynchronize(stream));
  *M = total_host;
  return LT_OK;
}

// ---------------------------------------------------------------- reverse table
__global__ void k_kmap_reverse(const int* __restrict__ results, int K, int64_t Vout, int64_t Vin,
                               int* __restrict__ rev) {
  const int64_t n = Vout * (int64_t)K;
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
    const int r = results[g];
    if (r >= 0 && r < Vin) {
      const int k = (int)(g / Vout);
      rev[(int64_t)k * Vin + r] = (int)(g - (int64_t)k * Vout);
    }
  }
}

extern "C" int lt_kmap_reverse(const int32_t* results, int32_t K, int64_t Vout, int64_t Vin, int32_t* rev,
                               hipStream_t stream) {
  LT_REQUIRE(K >= 1 && Vout >= 0 && Vin >= 0, LT_ERR_ARG, "lt_kmap_reverse: bad sizes");
  if (Vin == 0) return LT_OK;
  LT_REQUIRE(rev && (Vout == 0 || results), LT_ERR_ARG, "lt_kmap_reverse: null pointer");
  LT_CHECK_HIP(hipMemsetAsync(rev, 0xFF, (int64_t)K * Vin * 4, stream));
  if (Vout > 0) {
    hipLaunchKernelGGL(k_kmap_reverse, dim3(lt_grid((int64_t)K * Vout, 256)), dim3(256), 0, stream, results, K, Vout,
                       Vin, rev);
    LT_LAUNCH_CHECK();
  }
  return LT_OK;
}

// ---------------------------------------------------------------- processing order of the output rows
// The output-stationary conv kernel issues MFMAs for a 16-row strip and an offset k whenever ANY row of the strip has a
// neighbour at k.  With rows in storage order (ascending hash = spatially random) a LiDAR map keeps only ~20 % of the
// issued row-offsets useful; sorting the rows by their K-bit neighbour mask puts rows with the same offsets into the
// same strip (85-95 % useful).  Stable sort: equal masks keep ascending row order.
__global__ void k_row_masks(const int* __restrict__ table, int K, int64_t V, unsigned int* __restrict__ masks,
                            int* __restrict__ iota, unsigned int* __restrict__ counts) {
  __shared__ unsigned int s_cnt[32];
  if (threadIdx.x < 32) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  for (int64_t o0 = blockIdx.x * (int64_t)blockDim.x; o0 < V; o0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = o0 + threadIdx.x;
    unsigned int m = 0;
    if (o < V) {
      for (int k = 0; k < K; ++k) m |= (table[(int64_t)k * V + o] >= 0 ? 1u : 0u) << k;
      masks[o] = m;
      iota[o] = (int)o;
    }
    // rows per offset (integer atomics: the totals do not depend on the order of the adds)
    for (int k = 0; k < K; ++k) {
      const unsigned long long b = __ballot((m >> k) & 1u);
      if ((threadIdx.x & 63) == 0 && b) atomicAdd(&s_cnt[k], (unsigned int)__popcll(b));
    }
  }
  __syncthreads();
  if (threadIdx.x < K && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
}

// Sort key of a row: its neighbour mask with the bits permuted so that the offsets FEWEST rows have are the most
// significant, then Gray-decoded.  Rows that differ only in a common offset end up next to each other, and walking the
// sorted list changes one high bit at a time -- fewer strips whose union of offsets is larger than each row's own set:
// 0.885 -> 0.914 useful MFMA rows at stride 1 (3 x 131 072-point scans), 0.815 -> 0.846 on 360-degree scans, against
// the plain integer order of the masks.  Only the grouping changes: every output row is still accumulated in the
// same (offset, channel group) order.
__global__ void k_mask_keys(const unsigned int* masks, const unsigned int* __restrict__ counts, int K, int64_t V,
                            unsigned int* keys) {  // keys may be masks (in place)
  __shared__ int s_pos[32];  // s_pos[k] = bit position of offset k in the key
  if (threadIdx.x < 32) {
    const int k = threadIdx.x;
    if (k < K) {
      const unsigned int ck = counts[k];
      int rank = 0;  // number of offsets that come before k in descending-count order (ties: lower k first)
      for (int j = 0; j < K; ++j) {
        const unsigned int cj = counts[j];
        rank += (cj > ck || (cj == ck && j < k)) ? 1 : 0;
      }
      s_pos[k] = rank;  // the most common offset gets bit 0
    }
  }
  __syncthreads();
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < V; o += (int64_t)gridDim.x * blockDim.x) {
    const unsigned int m = masks[o];
    unsigned int g = 0;
    for (int k = 0; k < K; ++k) g |= ((m >> k) & 1u) << s_pos[k];
    g ^= g >> 1; g ^= g >> 2; g ^= g >> 4; g ^= g >> 8; g ^= g >> 16;  // rank of g in the reflected Gray code
    keys[o] = g;
  }
}

static int64_t order_sort_bytes(int64_t n) {
  size_t b = 0;
  unsigned int* k = nullptr;
  int* v = nullptr;
  (void)rocprim::radix_sort_pairs<LtSortConfig>(nullptr, b, k, k, v, v, (size_t)(n > 0 ? n : 1), 0, 32, (hipStream_t)0);
  return (int64_t)b;
}

extern "C" int64_t lt_kmap_order_workspace_bytes(int64_t V) {
  if (V < 1) V = 1;
  return 3 * lt_align_up(V * 4, 256) + lt_align_up(order_sort_bytes(V), 256) + 1024 + 256;
}

extern "C" int lt_kmap_order(const int32_t* table, int32_t K, int64_t V, void* ws, int64_t ws_bytes, int32_t* order,
                             hipStream_t stream) {
  LT_REQUIRE(K >= 1 && K <= 32 && V >= 0 && V < INT_MAX, LT_ERR_ARG, "lt_kmap_order: bad sizes (K=%d)", K);
  if (V == 0) return LT_OK;
  LT_REQUIRE(table && ws && order, LT_ERR_ARG, "lt_kmap_order: null pointer");
  LT_REQUIRE(ws_bytes >= lt_kmap_order_workspace_bytes(V), LT_ERR_WORKSPACE, "lt_kmap_order: workspace too small");
  LtArena a(ws, ws_bytes);
  unsigned int* masks = (unsigned int*)a.take(V * 4);
  unsigned int* smasks = (unsigned int*)a.take(V * 4);
  int* iota = (int*)a.take(V * 4);
  const int64_t sb = order_sort_bytes(V);
  void* tmp = a.take(sb);
  LT_REQUIRE(masks && smasks && iota && tmp, LT_ERR_WORKSPACE, "lt_kmap_order: arena exhausted");
  unsigned int* counts = (unsigned int*)a.take(32 * 4);
  LT_REQUIRE(counts, LT_ERR_WORKSPACE, "lt_kmap_order: arena exhausted");
  LT_CHECK_HIP(hipMemsetAsync(counts, 0, 32 * 4, stream));
  hipLaunchKernelGGL(k_row_masks, dim3(lt_grid(V, 256)), dim3(256), 0, stream, table, K, V, masks, iota, counts);
  LT_LAUNCH_CHECK();
  // the sort key replaces the mask in place (the masks themselves are not needed again)
  hipLaunchKernelGGL(k_mask_keys, dim3(lt_grid(V, 256)), dim3(256), 0, stream, masks, counts, K, V, masks);
  LT_LAUNCH_CHECK();
  size_t sbs = (size_t)sb;
  LT_CHECK_HIP(rocprim::radix_sort_pairs<LtSortConfig>(tmp, sbs, masks, smasks, iota, order, (size_t)V, 0, (unsigned)K, stream));
  return LT_OK;
}

// ---------------------------------------------------------------- conv plan (tile layout of a neighbour table)
// one 256-thread block per 64-row tile: wave w handles offsets k = w, w+4, ...; lane i = row i of the tile
__global__ __launch_bounds__(256) void k_plan_tiles(const int* __restrict__ table, int K, int64_t V,
                                                    const int* __restrict__ order, int* __restrict__ tnbr,
                                                    unsigned long long* __restrict__ tmask,
                                                    unsigned int* __restrict__ twork, int* __restrict__ tiota) {
  __shared__ int s_work[4];
  const int64_t tile = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t slot = tile * 64 + lane;
  const int row = slot < V ? order[slot] : -1;
  int work = 0;  // active (16-row strip, offset) pairs of the tile = MFMA work of its workgroup
  for (int k = wave; k < K; k += 4) {
    const int v = row >= 0 ? table[(int64_t)k * V + row] : -1;
    tnbr[(tile * K + k) * 64 + lane] = v;
    const unsigned long long m = __ballot(v >= 0);
    if (lane == 0) tmask[tile * K + k] = m;
    for (int st = 0; st < 4; ++st) work += ((m >> (16 * st)) & 0xFFFFull) ? 1 : 0;
  }
  if (lane == 0) s_work[wave] = work;
  __syncthreads();
  if (threadIdx.x == 0) {
    twork[tile] = (unsigned int)(s_work[0] + s_work[1] + s_work[2] + s_work[3]);
    tiota[tile] = (int)tile;
  }
}

extern "C" int64_t lt_kmap_plan_bytes(int32_t K, int64_t V) { return lt_plan_view(nullptr, K, V).bytes; }
static int64_t tile_sort_bytes(int64_t t) {
  size_t b = 0;
  unsigned int* k = nullptr;
  int* v = nullptr;
  (void)rocprim::radix_sort_pairs_desc<LtSortConfig>(nullptr, b, k, k, v, v, (size_t)(t > 0 ? t : 1), 0, 32, (hipStream_t)0);
  return (int64_t)b;
}
extern "C" int64_t lt_kmap_plan_workspace_bytes(int64_t V) {
  const int64_t T = (V + 63) / 64 > 0 ? (V + 63) / 64 : 1;
  return lt_kmap_order_workspace_bytes(V) + 3 * lt_align_up(T * 4, 256) + lt_align_up(tile_sort_bytes(T), 256) + 1024;
}

extern "C" int lt_kmap_plan(const int32_t* table, int32_t K, int64_t V, void* ws, int64_t ws_bytes, void* plan,
                            int64_t plan_bytes, hipStream_t stream) {
  LT_REQUIRE(K >= 1 && K <= 32 && V >= 0 && V < INT_MAX, LT_ERR_ARG, "lt_kmap_plan: bad sizes (K=%d)", K);
  if (V == 0) return LT_OK;
  LT_REQUIRE(table && ws && plan, LT_ERR_ARG, "lt_kmap_plan: null pointer");
  LT_REQUIRE(((uintptr_t)plan & 15) == 0, LT_ERR_ARG, "lt_kmap_plan: plan must be 16-byte aligned");
  LtPlanView pv = lt_plan_view(plan, K, V);
  LT_REQUIRE(plan_bytes >= pv.bytes, LT_ERR_WORKSPACE, "lt_kmap_plan: plan needs %lld bytes, got %lld",
             (long long)pv.bytes, (long long)plan_bytes);
  LT_REQUIRE(ws_bytes >= lt_kmap_plan_workspace_bytes(V), LT_ERR_WORKSPACE, "lt_kmap_plan: workspace too small");
  const int64_t ob = lt_kmap_order_workspace_bytes(V);
  int rc = lt_kmap_order(table, K, V, ws, ob, pv.order, stream);
  if (rc) return rc;
  const int64_t T = (V + 63) / 64;
  LtArena a((char*)ws + ob, ws_bytes - ob);
  unsigned int* twork = (unsigned int*)a.take(T * 4);
  unsigned int* twork_s = (unsigned int*)a.take(T * 4);
  int* tiota = (int*)a.take(T * 4);
  const int64_t sb = tile_sort_bytes(T);
  void* tmp = a.take(sb);
  LT_REQUIRE(twork && twork_s && tiota && tmp, LT_ERR_WORKSPACE, "lt_kmap_plan: arena exhausted");
  hipLaunchKernelGGL(k_plan_tiles, dim3((unsigned)T), dim3(256), 0, stream, table, K, V, pv.order, pv.tnbr, pv.tmask,
                     twork, tiota);
  LT_LAUNCH_CHECK();
  size_t sbs = (size_t)sb;
  LT_CHECK_HIP(rocprim::radix_sort_pairs_desc<LtSortConfig>(tmp, sbs, twork, twork_s, tiota, pv.torder, (size_t)T, 0, 8, stream));
  return LT_OK;
}
