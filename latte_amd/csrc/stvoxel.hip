// Spatial-temporal voxel aggregation (HR_Block / refine_ety of the reference) as one batched GPU pipeline.
// Reference: latte/models/hr_module.py:135-331, latte/models/losses.py:10-19,
//            latte/data/utils/refine_pseudo_labels.py:33-53.  See include/latte_hip_st.h for the contract.
//
// Everything here is HBM-bound integer / fp32 streaming work: one lane per point or per (voxel, class), rows
// contiguous across lanes, segmented reductions over radix-sorted voxel keys (atomic-free, summation order =
// ascending point index, i.e. the order of the reference's CPU index_add).
#include "common.h"
#include "../../include/latte_hip.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <limits.h>

#define ST_MASKED 0x7FFFFFFFFFFFFFFFll

// ---------------------------------------------------------------- softmax + entropy
// v[] lives in registers: every loop over the classes is unrolled to the compile-time bound CM (16 / 32 / 64) and guarded by
// c < C -- a runtime-bounded loop over v[] would put the array into scratch memory (272 B per lane).
template <int CM>
__device__ __forceinline__ float st_softmax_entropy(const float* __restrict__ row, int C, float log2c, float (&v)[CM]) {
#pragma clang fp contract(off)
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < CM; ++c)
    if (c < C) { v[c] = row[c]; mx = fmaxf(mx, v[c]); }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CM; ++c)
    if (c < C) { v[c] = expf(v[c] - mx); sum += v[c]; }
  float e = 0.f;
#pragma unroll
  for (int c = 0; c < CM; ++c)
    if (c < C) {
      v[c] = v[c] / sum;
      e += (-(v[c] * log2f(v[c] + 1e-30f))) / log2c;
    }
  return e;
}

template <int CM>
__global__ __launch_bounds__(128) void k_softmax_entropy(const float* __restrict__ logits, int64_t P, int C, float* __restrict__ probs,
                                  float* __restrict__ ety) {
  const float log2c = (float)log2((double)C);
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    float v[CM];
    const float e = st_softmax_entropy<CM>(logits + p * C, C, log2c, v);
    if (probs) {
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) probs[p * C + c] = v[c];
    }
    if (ety) ety[p] = e;
  }
}

#define ST_CM_DISPATCH(C, CALL)            \
  do {                                     \
    if ((C) <= 16) { CALL(16); }           \
    else if ((C) <= 32) { CALL(32); }      \
    else { CALL(64); }                     \
  } while (0)

extern "C" int lt_softmax_entropy(const float* logits, int64_t P, int32_t C, float* probs, float* ety,
                                  hipStream_t stream) {
  LT_REQUIRE(P >= 0 && C >= 1 && C <= 64, LT_ERR_ARG, "lt_softmax_entropy: C=%d must be in [1,64]", C);
  if (P == 0) return LT_OK;
  LT_REQUIRE(logits && (probs || ety), LT_ERR_ARG, "lt_softmax_entropy: null pointer");
#define ST_CALL(CM) hipLaunchKernelGGL(k_softmax_entropy<CM>, dim3(lt_grid(P, 128)), dim3(128), 0, stream, logits, P, C, probs, ety)
  ST_CM_DISPATCH(C, ST_CALL);
#undef ST_CALL
  LT_LAUNCH_CHECK();
  return LT_OK;
}

__global__ void k_softmax_bwd(const float* __restrict__ probs, const float* __restrict__ g, int64_t P, int C,
                              float* __restrict__ gl) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += g[p * C + c] * probs[p * C + c];
    for (int c = 0; c < C; ++c) gl[p * C + c] = probs[p * C + c] * (g[p * C + c] - dot);
  }
}

extern "C" int lt_softmax_bwd(const float* probs, const float* gprobs, int64_t P, int32_t C, float* glogits,
                              hipStream_t stream) {
  LT_REQUIRE(P >= 0 && C >= 1, LT_ERR_ARG, "lt_softmax_bwd: bad sizes");
  if (P == 0) return LT_OK;
  LT_REQUIRE(probs && gprobs && glogits, LT_ERR_ARG, "lt_softmax_bwd: null pointer");
  hipLaunchKernelGGL(k_softmax_bwd, dim3(lt_grid(P, 128)), dim3(128), 0, stream, probs, gprobs, P, C, glogits);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- transform + range + key
__global__ void k_st_keys(const float* __restrict__ pts, const int64_t* __restrict__ seg_out, const int64_t* __restrict__ seg_src,
                          const int* __restrict__ seg_anchor, const double* __restrict__ seg_T, int S, int64_t E,
                          double R, double vs, int min_off, int cb, int64_t* __restrict__ keys, int* __restrict__ src) {
#pragma clang fp contract(off)
  const int64_t cmask = (1ll << cb) - 1;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    // segment of e: largest s with seg_out[s] <= e  (S is small: <= anchors x window)
    int lo = 0, hi = S - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (seg_out[mid] <= e) lo = mid; else hi = mid - 1;
    }
    const int s = lo;
    const int64_t j = seg_src[s] + (e - seg_out[s]);
    const float px = pts[3 * j], py = pts[3 * j + 1], pz = pts[3 * j + 2];
    double x, y, z;
    bool in_range;
    if (seg_T) {
      const double* T = seg_T + 12 * s;
      // (T @ [p,1])[:3] in fp64, k ascending like a dgemm micro-kernel
      x = T[0] * (double)px + T[1] * (double)py + T[2] * (double)pz + T[3];
      y = T[4] * (double)px + T[5] * (double)py + T[6] * (double)pz + T[7];
      z = T[8] * (double)px + T[9] * (double)py + T[10] * (double)pz + T[11];
      in_range = sqrt(x * x + y * y + z * z) <= R;
    } else {
      x = (double)px; y = (double)py; z = (double)pz;
      in_range = sqrtf(px * px + py * py + pz * pz) <= (float)R;  // fp32 norm of the anchor branch
    }
    int64_t key = ST_MASKED;
    if (in_range) {
      const int cx = (int)floor(x / vs) - min_off, cy = (int)floor(y / vs) - min_off, cz = (int)floor(z / vs) - min_off;
      key = ((int64_t)seg_anchor[s] << (3 * cb)) | (((int64_t)cx & cmask) << (2 * cb)) | (((int64_t)cy & cmask) << cb) |
            ((int64_t)cz & cmask);
    }
    keys[e] = key;
    src[e] = (int)j;
  }
}

extern "C" int lt_st_keys(const float* points, const int64_t* seg_out_start, const int64_t* seg_src_start,
                          const int32_t* seg_anchor, const double* seg_T, int32_t S, int64_t E, double max_range,
                          double voxel_size, int32_t min_offset, int32_t coord_bits, int64_t* keys, int32_t* src,
                          hipStream_t stream) {
  LT_REQUIRE(S >= 0 && E >= 0 && E < INT_MAX, LT_ERR_RANGE, "lt_st_keys: bad sizes");
  if (E == 0) return LT_OK;
  LT_REQUIRE(S >= 1 && points && seg_out_start && seg_src_start && seg_anchor && keys && src, LT_ERR_ARG,
             "lt_st_keys: null pointer");
  LT_REQUIRE(voxel_size > 0 && max_range > 0, LT_ERR_ARG, "lt_st_keys: voxel_size and max_range must be positive");
  LT_REQUIRE(coord_bits >= 1 && coord_bits <= 16, LT_ERR_ARG, "lt_st_keys: coord_bits=%d must be in [1,16]", coord_bits);
  LT_REQUIRE(2.0 * max_range / voxel_size + 4.0 < (double)(1 << coord_bits), LT_ERR_RANGE,
             "lt_st_keys: max_range/voxel_size too large for %d-bit voxel coordinates", coord_bits);
  hipLaunchKernelGGL(k_st_keys, dim3(lt_grid(E, 256)), dim3(256), 0, stream, points, seg_out_start, seg_src_start,
                     seg_anchor, seg_T, S, E, max_range, voxel_size, min_offset, coord_bits, keys, src);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- sort + segments
struct HeadFlag {
  const int64_t* k;
  __device__ int operator()(int p) const {
    const int64_t v = k[p];
    return (v != ST_MASKED && (p == 0 || k[p - 1] != v)) ? 1 : 0;
  }
};

// flag (optional): cleared here, raised by k_emit_segments (a later launch on the same stream)
__global__ void k_iota(int* __restrict__ a, int64_t n, int* __restrict__ flag) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    a[i] = (int)i;
    if (i == 0 && flag) *flag = 0;
  }
}

__global__ void k_emit_segments(const int64_t* __restrict__ skeys, const int* __restrict__ selem, const int* __restrict__ vid,
                                int64_t E, int* __restrict__ inverse, int64_t* __restrict__ uniq, int* __restrict__ seg_start,
                                int* __restrict__ counts /* [0]=V, [1]=n_valid, [2]=a key broke the key_bits promise */,
                                unsigned end_bit) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < E; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = skeys[p];
    const int e = selem[p];
    if (k == ST_MASKED) {
      inverse[e] = -1;
      if (p == 0) { counts[0] = 0; counts[1] = 0; seg_start[0] = 0; }
      continue;
    }
    // an unmasked key with a bit at or above the sorted range: the array is NOT sorted, every result below is garbage
    if (k < 0 || (end_bit < 63u && (k >> (end_bit - 1u)) != 0)) counts[2] = 1;
    const int v = vid[p] - 1;  // inclusive scan of head flags
    inverse[e] = v;
    if (p == 0 || skeys[p - 1] != k) { uniq[v] = k; seg_start[v] = (int)p; }
    if (p == E - 1 || skeys[p + 1] == ST_MASKED) {
      counts[0] = v + 1;
      counts[1] = (int)(p + 1);
      seg_start[v + 1] = (int)(p + 1);
    }
  }
}

static int64_t sortseg_sort_bytes(int64_t n) {
  size_t b = 0;
  int64_t* k = nullptr;
  int* v = nullptr;
  (void)rocprim::radix_sort_pairs<LtSortConfig>(nullptr, b, k, k, v, v, (size_t)(n > 0 ? n : 1), 0, 64, (hipStream_t)0);
  return (int64_t)b;
}
static int64_t sortseg_scan_bytes(int64_t n) {
  size_t b = 0;
  int* o = nullptr;
  auto it = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), HeadFlag{nullptr});
  (void)rocprim::inclusive_scan(nullptr, b, it, o, (size_t)(n > 0 ? n : 1), rocprim::plus<int>(), (hipStream_t)0);
  return (int64_t)b;
}

extern "C" int64_t lt_sort_segments_workspace_bytes(int64_t E) {
  if (E < 1) E = 1;
  return lt_align_up(E * 8, 256) + 3 * lt_align_up(E * 4, 256) + lt_align_up(sortseg_sort_bytes(E), 256) +
         lt_align_up(sortseg_scan_bytes(E), 256) + 1024;
}

extern "C" int lt_sort_segments(const int64_t* keys, int64_t E, int32_t key_bits, void* ws, int64_t ws_bytes,
                                int32_t* sorted_elem, int32_t* inverse, int64_t* uniq_keys, int32_t* seg_start,
                                int64_t* V, hipStream_t stream) {
  LT_REQUIRE(V, LT_ERR_ARG, "lt_sort_segments: V pointer is null");
  *V = 0;
  LT_REQUIRE(E >= 0 && E < INT_MAX, LT_ERR_RANGE, "lt_sort_segments: E out of range");
  if (E == 0) return LT_OK;
  LT_REQUIRE(keys && ws && sorted_elem && inverse && uniq_keys && seg_start, LT_ERR_ARG, "lt_sort_segments: null pointer");
  LT_REQUIRE(ws_bytes >= lt_sort_segments_workspace_bytes(E), LT_ERR_WORKSPACE, "lt_sort_segments: workspace too small");
  // valid keys are < 2^key_bits (caller's promise); masked elements carry INT64_MAX, whose bits [0, key_bits] are all ones:
  // a sort over key_bits + 1 bits puts them last, and costs ceil((key_bits + 1) / 8) radix passes instead of 8
  const unsigned end_bit = (key_bits >= 1 && key_bits < 62) ? (unsigned)key_bits + 1u : 63u;
  LtArena a(ws, ws_bytes);
  int64_t* skeys = (int64_t*)a.take(E * 8);
  int* iota = (int*)a.take(E * 4);
  int* vid = (int*)a.take(E * 4);
  int* counts = (int*)a.take(256);
  const int64_t sb = sortseg_sort_bytes(E), cb = sortseg_scan_bytes(E);
  void* stmp = a.take(sb);
  void* ctmp = a.take(cb);
  LT_REQUIRE(skeys && iota && vid && counts && stmp && ctmp, LT_ERR_WORKSPACE, "lt_sort_segments: arena exhausted");
  hipLaunchKernelGGL(k_iota, dim3(lt_grid(E, 256)), dim3(256), 0, stream, iota, E, counts + 2);
  LT_LAUNCH_CHECK();
  size_t sbs = (size_t)sb, cbs = (size_t)cb;
  // LSD radix sort is stable: equal keys keep ascending element order => segment head = first occurrence
  LT_CHECK_HIP(rocprim::radix_sort_pairs<LtSortConfig>(stmp, sbs, keys, skeys, iota, sorted_elem, (size_t)E, 0, end_bit, stream));
  auto it = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), HeadFlag{skeys});
  LT_CHECK_HIP(rocprim::inclusive_scan(ctmp, cbs, it, vid, (size_t)E, rocprim::plus<int>(), stream));
  hipLaunchKernelGGL(k_emit_segments, dim3(lt_grid(E, 256)), dim3(256), 0, stream, skeys, sorted_elem, vid, E, inverse,
                     uniq_keys, seg_start, counts, end_bit);
  LT_LAUNCH_CHECK();
  int host[3] = {0, 0, 0};
  LT_CHECK_HIP(hipMemcpyAsync(host, counts, 12, hipMemcpyDeviceToHost, stream));
  LT_CHECK_HIP(hipStreamSynchronize(stream));
  // the promise is checked on the device (one compare per key in a kernel that reads every key anyway): a stale key_bits or
  // an anchor id beyond its bits would otherwise give a silently mis-sorted array, wrong segments and wrong joins
  LT_REQUIRE(host[2] == 0, LT_ERR_RANGE, "lt_sort_segments: an unmasked key is negative or >= 2^%d (key_bits = %d): outputs invalid",
             (int)end_bit - 1, key_bits);
  *V = host[0];
  return LT_OK;
}

// ---------------------------------------------------------------- segmented reduce
// one thread per (voxel, class): the C lanes of a voxel read its rows coalesced; four points of the segment are in flight
// per thread (independent index -> row chains), accumulated in ascending point order
__global__ void k_segment_reduce(const float* __restrict__ src, int C, const int* __restrict__ src_index,
                                 const int* __restrict__ selem, const int* __restrict__ seg_start, int64_t V, int mode,
                                 float* __restrict__ out, int* __restrict__ argmax) {
  const int64_t total = V * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = t / C;
    const int c = (int)(t - v * C);
    const int p0 = seg_start[v], p1 = seg_start[v + 1];
    float acc = mode == 2 ? -INFINITY : 0.f;
    int arg = -1;
    int p = p0;
    for (; p + 3 < p1; p += 4) {
      int e[4];
      int64_t row[4];
      float x[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = selem[p + i];
#pragma unroll
      for (int i = 0; i < 4; ++i) row[i] = src_index ? src_index[e[i]] : e[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = src[row[i] * C + c];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (mode == 2) { if (x[i] > acc || arg < 0) { acc = x[i]; arg = e[i]; } }
        else acc += x[i];
      }
    }
    for (; p < p1; ++p) {
      const int e = selem[p];
      const int64_t row = src_index ? src_index[e] : e;
      const float x = src[row * C + c];
      if (mode == 2) { if (x > acc || arg < 0) { acc = x; arg = e; } }
      else acc += x;
    }
    if (mode == 1) { const int n = p1 - p0; acc = acc / (float)(n < 1 ? 1 : n); }
    if (mode == 2 && arg < 0) acc = 0.f;
    out[t] = acc;
    if (argmax) argmax[t] = arg;
  }
}

extern "C" int lt_segment_reduce(const float* src, int32_t C, const int32_t* src_index, const int32_t* sorted_elem,
                                 const int32_t* seg_start, int64_t V, int32_t mode, float* out, int32_t* argmax,
                                 hipStream_t stream) {
  LT_REQUIRE(V >= 0 && C >= 1 && mode >= 0 && mode <= 2, LT_ERR_ARG, "lt_segment_reduce: bad args");
  if (V == 0) return LT_OK;
  LT_REQUIRE(src && sorted_elem && seg_start && out, LT_ERR_ARG, "lt_segment_reduce: null pointer");
  hipLaunchKernelGGL(k_segment_reduce, dim3(lt_grid(V * C, 256)), dim3(256), 0, stream, src, C, src_index, sorted_elem,
                     seg_start, V, mode, out, argmax);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

__global__ void k_segment_reduce_bwd(const float* __restrict__ g, int C, const int* __restrict__ inverse,
                                     const int* __restrict__ seg_start, int64_t E, int mean, float* __restrict__ ge) {
  const int64_t total = E * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = t / C;
    const int c = (int)(t - e * C);
    const int v = inverse[e];
    float x = 0.f;
    if (v >= 0) {
      x = g[(int64_t)v * C + c];
      if (mean) { const int n = seg_start[v + 1] - seg_start[v]; x = x / (float)(n < 1 ? 1 : n); }
    }
    ge[t] = x;
  }
}

extern "C" int lt_segment_reduce_bwd(const float* gout, int32_t C, const int32_t* inverse, const int32_t* seg_start,
                                     int64_t E, int32_t mean, float* gelem, hipStream_t stream) {
  LT_REQUIRE(E >= 0 && C >= 1, LT_ERR_ARG, "lt_segment_reduce_bwd: bad args");
  if (E == 0) return LT_OK;
  LT_REQUIRE(gout && inverse && seg_start && gelem, LT_ERR_ARG, "lt_segment_reduce_bwd: null pointer");
  hipLaunchKernelGGL(k_segment_reduce_bwd, dim3(lt_grid(E * C, 256)), dim3(256), 0, stream, gout, C, inverse, seg_start, E,
                     mean, gelem);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- ST entropy back to the points
__global__ void k_st_gather_back(const float* __restrict__ vety, const int* __restrict__ inverse,
                                 const int64_t* __restrict__ frame_start, int F, const int* __restrict__ frame_seg, int Wmax,
                                 const int64_t* __restrict__ seg_out, int64_t P, float* __restrict__ out) {
#pragma clang fp contract(off)
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < P; j += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = F - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (frame_start[mid] <= j) lo = mid; else hi = mid - 1;
    }
    const int f = lo;
    const int64_t local = j - frame_start[f];
    float acc = 0.f;
    int cnt = 0;
    for (int w = 0; w < Wmax; ++w) {  // ascending anchor order == the reference's accumulation order
      const int s = frame_seg[f * Wmax + w];
      if (s < 0) break;
      const int v = inverse[seg_out[s] + local];
      if (v >= 0) { acc = acc + vety[v]; ++cnt; }
    }
    out[j] = cnt > 0 ? acc / (float)cnt : 0.f;
  }
}

extern "C" int lt_st_gather_back(const float* voxel_ety, const int32_t* inverse, const int64_t* frame_start, int32_t F,
                                 const int32_t* frame_seg, int32_t Wmax, const int64_t* seg_out_start, int64_t P,
                                 float* out, hipStream_t stream) {
  LT_REQUIRE(P >= 0 && F >= 1 && Wmax >= 1, LT_ERR_ARG, "lt_st_gather_back: bad sizes");
  if (P == 0) return LT_OK;
  LT_REQUIRE(inverse && frame_start && frame_seg && seg_out_start && out, LT_ERR_ARG, "lt_st_gather_back: null pointer");
  hipLaunchKernelGGL(k_st_gather_back, dim3(lt_grid(P, 256)), dim3(256), 0, stream, voxel_ety, inverse, frame_start, F,
                     frame_seg, Wmax, seg_out_start, P, out);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// The same write-back when a window's frames are NOT contiguous in the batch: the reference then slices the hist rows
// sequentially from the window's first frame on (hr_module.py:203-220, 256-259), so segment s covers the ROW range
// [seg_row[s], seg_row[s] + n_s) whatever frame its points came from.  One thread per row, all S segments tested in
// order (S <= anchors x window, ascending anchor: the reference's accumulation order).
__global__ void k_st_gather_back_rows(const float* __restrict__ vety, const int* __restrict__ inverse,
                                      const int64_t* __restrict__ seg_row, const int64_t* __restrict__ seg_out, int S, int64_t P,
                                      float* __restrict__ out) {
#pragma clang fp contract(off)
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < P; j += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    int cnt = 0;
    for (int s = 0; s < S; ++s) {
      const int64_t d = j - seg_row[s];
      if (d < 0 || d >= seg_out[s + 1] - seg_out[s]) continue;
      const int v = inverse[seg_out[s] + d];
      if (v >= 0) { acc = acc + vety[v]; ++cnt; }
    }
    out[j] = cnt > 0 ? acc / (float)cnt : 0.f;
  }
}

extern "C" int lt_st_gather_back_rows(const float* voxel_ety, const int32_t* inverse, const int64_t* seg_row_start,
                                      const int64_t* seg_out_start, int32_t S, int64_t P, float* out, hipStream_t stream) {
  LT_REQUIRE(P >= 0 && S >= 1, LT_ERR_ARG, "lt_st_gather_back_rows: bad sizes");
  if (P == 0) return LT_OK;
  LT_REQUIRE(inverse && seg_row_start && seg_out_start && out, LT_ERR_ARG, "lt_st_gather_back_rows: null pointer");
  hipLaunchKernelGGL(k_st_gather_back_rows, dim3(lt_grid(P, 256)), dim3(256), 0, stream, voxel_ety, inverse, seg_row_start,
                     seg_out_start, S, P, out);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- sorted-set join
__global__ void k_join_probe(const int64_t* __restrict__ a, int64_t Va, const int64_t* __restrict__ t, int64_t Vt,
                             int* __restrict__ a_hit, unsigned char* __restrict__ a_flag, unsigned char* __restrict__ t_flag) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < Va; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t key = a[i];
    int64_t lo = 0, hi = Vt;  // lower_bound
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (t[mid] < key) lo = mid + 1; else hi = mid;
    }
    const bool hit = lo < Vt && t[lo] == key;
    a_hit[i] = hit ? (int)lo : -1;
    a_flag[i] = hit ? 1 : 0;
    if (hit) t_flag[lo] = 1;
  }
}

static int64_t select_bytes(int64_t n) {
  size_t b = 0;
  unsigned char* f = nullptr;
  int* o = nullptr;
  (void)rocprim::select(nullptr, b, rocprim::counting_iterator<int>(0), f, o, o, (size_t)(n > 0 ? n : 1), (hipStream_t)0);
  return (int64_t)b;
}

extern "C" int64_t lt_st_join_workspace_bytes(int64_t Va, int64_t Vt) {
  const int64_t n = (Va > Vt ? Va : Vt) > 0 ? (Va > Vt ? Va : Vt) : 1;
  return lt_align_up(Va > 0 ? Va * 4 : 4, 256) + lt_align_up(Va > 0 ? Va : 1, 256) + lt_align_up(Vt > 0 ? Vt : 1, 256) +
         lt_align_up(select_bytes(n), 256) + 1024;
}

extern "C" int lt_st_join(const int64_t* a_keys, int64_t Va, const int64_t* t_keys, int64_t Vt, void* ws,
                          int64_t ws_bytes, int32_t* a_sel, int32_t* t_sel, int64_t* M, hipStream_t stream) {
  LT_REQUIRE(M, LT_ERR_ARG, "lt_st_join: M pointer is null");
  *M = 0;
  LT_REQUIRE(Va >= 0 && Vt >= 0 && Va < INT_MAX && Vt < INT_MAX, LT_ERR_RANGE, "lt_st_join: bad sizes");
  if (Va == 0 || Vt == 0) return LT_OK;
  LT_REQUIRE(a_keys && t_keys && ws && a_sel && t_sel, LT_ERR_ARG, "lt_st_join: null pointer");
  LT_REQUIRE(ws_bytes >= lt_st_join_workspace_bytes(Va, Vt), LT_ERR_WORKSPACE, "lt_st_join: workspace too small");
  LtArena a(ws, ws_bytes);
  int* a_hit = (int*)a.take(Va * 4);
  unsigned char* a_flag = (unsigned char*)a.take(Va);
  unsigned char* t_flag = (unsigned char*)a.take(Vt);
  int* cnt = (int*)a.take(256);
  const int64_t sb = select_bytes(Va > Vt ? Va : Vt);
  void* tmp = a.take(sb);
  LT_REQUIRE(a_hit && a_flag && t_flag && cnt && tmp, LT_ERR_WORKSPACE, "lt_st_join: arena exhausted");
  LT_CHECK_HIP(hipMemsetAsync(t_flag, 0, Vt, stream));
  hipLaunchKernelGGL(k_join_probe, dim3(lt_grid(Va, 256)), dim3(256), 0, stream, a_keys, Va, t_keys, Vt, a_hit, a_flag,
                     t_flag);
  LT_LAUNCH_CHECK();
  size_t sbs = (size_t)sb;
  LT_CHECK_HIP(rocprim::select(tmp, sbs, rocprim::counting_iterator<int>(0), a_flag, a_sel, cnt, (size_t)Va, stream));
  sbs = (size_t)sb;
  LT_CHECK_HIP(rocprim::select(tmp, sbs, rocprim::counting_iterator<int>(0), t_flag, t_sel, cnt + 1, (size_t)Vt, stream));
  int host[2] = {0, 0};
  LT_CHECK_HIP(hipMemcpyAsync(host, cnt, 8, hipMemcpyDeviceToHost, stream));
  LT_CHECK_HIP(hipStreamSynchronize(stream));
  LT_REQUIRE(host[0] == host[1], LT_ERR_RANGE, "lt_st_join: duplicate keys in an input (%d vs %d matches)", host[0], host[1]);
  *M = host[0];
  return LT_OK;
}

// ---------------------------------------------------------------- order statistics by radix SELECTION (no sort)
// torch.quantile / torch.median need one or two order statistics per class, not the sorted array: four 8-bit histogram
// passes over order-preserving 32-bit keys, each over the elements whose leading bits equal the bits chosen so far, find
// them exactly -- 10 small dispatches per call where the radix sort took ~25 (12 sort kernels + rocPRIM's own memsets),
// all of them on the main stream between the teacher forward and the losses.  The values cluster (entropies of a
// confident network sit in a handful of bins, and then nearly every element passes every prefix test), so every pass
// counts in a per-workgroup LDS histogram -- a wave whose lanes all hit one counter adds once -- and flushes its
// non-zero bins.  Up to SEL_MAX_CLASSES classes; beyond that the sort path.
#define SEL_MAX_CLASSES 16
#define SEL_NB 256
struct SelState {
  unsigned prefix[2];  // bits chosen so far for the two wanted ranks (mode 0: floor / ceil of q (cnt - 1); mode 1: the median twice)
  unsigned rem[2];     // rank inside the chosen bin
  unsigned cnt;        // elements of the class
  unsigned pad[3];
};

__device__ __forceinline__ unsigned sel_key(float v) {
  const unsigned b = __float_as_uint(v + 0.0f);  // -0 -> +0
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float sel_unkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// one counter bump per lane, or one per wave when every active lane of the wave wants the same counter
__device__ __forceinline__ void sel_bump(unsigned* s_h, int slot) {
  const int first = __builtin_amdgcn_readfirstlane(slot);
  const unsigned long long same = __ballot(slot == first);
  const unsigned long long act = __ballot(1);
  if (same == act) {
    if ((int)__lane_id() == __ffsll((long long)act) - 1) atomicAdd(s_h + first, (unsigned)__popcll(act));
  } else {
    atomicAdd(s_h + slot, 1u);
  }
}

// hist[C][2][256] of byte LEVEL (0 = the most significant) over the elements whose higher bytes equal the query's prefix;
// level 0 has no prefix and fills query 0 only.  label == NULL: one class.
template <int LEVEL>
__global__ __launch_bounds__(256) void k_sel_hist(const float* __restrict__ x, const int* __restrict__ label, int64_t n, int C,
                                                  const SelState* __restrict__ st, unsigned* __restrict__ hist) {
  __shared__ unsigned s_h[SEL_MAX_CLASSES * 2 * SEL_NB];
  __shared__ unsigned s_pre[SEL_MAX_CLASSES * 2];
  __shared__ unsigned s_cnt[SEL_MAX_CLASSES];
  const int nb = C * 2 * SEL_NB;
  for (int e = threadIdx.x; e < nb; e += 256) s_h[e] = 0;
  if (LEVEL > 0 && threadIdx.x < C) {
    s_pre[2 * threadIdx.x] = st[threadIdx.x].prefix[0];
    s_pre[2 * threadIdx.x + 1] = st[threadIdx.x].prefix[1];
    s_cnt[threadIdx.x] = st[threadIdx.x].cnt;
  }
  __syncthreads();
  constexpr int SH = 24 - 8 * LEVEL;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int l = label ? label[i] : 0;
    if (l < 0 || l >= C) continue;
    const unsigned k = sel_key(x[i]);
    const int bin = (int)((k >> SH) & 0xFFu);
    if (LEVEL == 0) {
      sel_bump(s_h, (l * 2) * SEL_NB + bin);
    } else {
      if (s_cnt[l] == 0) continue;
      const unsigned lead = k >> (SH + 8);
      const bool m0 = lead == s_pre[2 * l], m1 = lead == s_pre[2 * l + 1];
      if (m0) sel_bump(s_h, (l * 2) * SEL_NB + bin);
      if (m1) sel_bump(s_h, (l * 2 + 1) * SEL_NB + bin);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nb; e += 256) {
    const unsigned v = s_h[e];
    if (v) atomicAdd(hist + e, v);
  }
}

// 256 threads, thread t owns bin t: the bin that holds rank r and r's rank inside it; *total = all counts.
// r >= total: which / rem are undefined (the caller asks for the total only)
__device__ void sel_locate(const unsigned* __restrict__ hist, unsigned r, unsigned* s_part, unsigned* which, unsigned* rem,
                           unsigned* total) {
  const int t = threadIdx.x;
  const unsigned v = hist[t];
  s_part[t] = v;
  __syncthreads();
  if (t < 64) {  // one wave: lane l scans bins 4l .. 4l+3, then a wave scan of the 64 sums
    const unsigned a0 = s_part[4 * t], a1 = s_part[4 * t + 1], a2 = s_part[4 * t + 2], a3 = s_part[4 * t + 3];
    const unsigned sum = a0 + a1 + a2 + a3;
    unsigned inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(inc, off, 64);
      if (t >= off) inc += o;
    }
    const unsigned ex = inc - sum;
    s_part[4 * t] = ex;
    s_part[4 * t + 1] = ex + a0;
    s_part[4 * t + 2] = ex + a0 + a1;
    s_part[4 * t + 3] = ex + a0 + a1 + a2;
    if (t == 63) s_part[256] = inc;
  }
  __syncthreads();
  const unsigned lo = s_part[t];
  if (v > 0 && r >= lo && r < lo + v) {  // exactly one thread when r < total
    s_part[257] = (unsigned)t;
    s_part[258] = r - lo;
  }
  __syncthreads();
  *total = s_part[256];
  *which = s_part[257];
  *rem = s_part[258];
  __syncthreads();
}

__device__ __forceinline__ void sel_ranks(unsigned cnt, float q, int mode, unsigned r[2]) {
#pragma clang fp contract(off)
  if (mode == 1) {
    r[0] = r[1] = (cnt - 1) / 2;   // torch.median: the lower middle element
  } else {
    const float rank = q * (float)(cnt - 1);   // torch.quantile: fp32 rank, as in ATen
    r[0] = (unsigned)floorf(rank);
    r[1] = (unsigned)ceilf(rank);
  }
}

// one block per class.  LEVEL 0: class count, the two ranks, their first byte.  LEVEL 1, 2: the next byte.
// LEVEL 3: the last byte -> the two values -> thresh[c]: mode 0 torch.quantile(q) with ATen's lerp (empty class: +inf),
// mode 1 min(torch.median, q) (empty: -inf)
template <int LEVEL>
__global__ __launch_bounds__(256) void k_sel_choose(const unsigned* __restrict__ hist, float q, int mode, SelState* __restrict__ st,
                                                    float* __restrict__ thresh) {
#pragma clang fp contract(off)
  __shared__ unsigned s_part[260];
  const int c = blockIdx.x;
  const unsigned* h = hist + (int64_t)c * 2 * SEL_NB;
  unsigned which, rem, total, w2[2], m2[2];
  if (LEVEL == 0) {
    sel_locate(h, 0xFFFFFFFFu, s_part, &which, &rem, &total);
    if (total == 0) {
      if (threadIdx.x == 0) st[c].cnt = 0;
      return;
    }
    unsigned r[2];
    sel_ranks(total, q, mode, r);
    for (int qi = 0; qi < 2; ++qi) sel_locate(h, r[qi], s_part, &w2[qi], &m2[qi], &total);
    if (threadIdx.x == 0) {
      st[c].cnt = total;
      st[c].prefix[0] = w2[0]; st[c].prefix[1] = w2[1];
      st[c].rem[0] = m2[0]; st[c].rem[1] = m2[1];
    }
    return;
  }
  const SelState s = st[c];
  if (s.cnt == 0) {
    if (LEVEL == 3 && threadIdx.x == 0) thresh[c] = mode == 0 ? INFINITY : -INFINITY;
    return;
  }
  for (int qi = 0; qi < 2; ++qi) sel_locate(h + qi * SEL_NB, s.rem[qi], s_part, &w2[qi], &m2[qi], &total);
  if (threadIdx.x != 0) return;
  const unsigned p0 = (s.prefix[0] << 8) | w2[0], p1 = (s.prefix[1] << 8) | w2[1];
  if (LEVEL < 3) {
    st[c].prefix[0] = p0; st[c].prefix[1] = p1;
    st[c].rem[0] = m2[0]; st[c].rem[1] = m2[1];
    return;
  }
  const float v0 = sel_unkey(p0), v1 = sel_unkey(p1);
  if (mode == 1) {
    thresh[c] = v0 < q ? v0 : q;
    return;
  }
  const float rank = q * (float)(s.cnt - 1);
  const float w = rank - floorf(rank);
  const float d = v1 - v0;
  thresh[c] = (w < 0.5f) ? (v0 + w * d) : (v1 - d * (1.0f - w));  // at::lerp
}

static int64_t sel_hist_ints(int C) { return (int64_t)4 * C * 2 * SEL_NB; }
static int64_t sel_bytes(int C) { return sel_hist_ints(C) * 4 + lt_align_up((int64_t)C * sizeof(SelState), 256); }

static bool sel_enabled() {
  static const bool on = [] {
    const char* e = getenv("LT_QUANTILE_SORT");   // A/B and tests: 1 = the radix-sort path
    return !(e && e[0] == '1');
  }();
  return on;
}

// thresholds of C classes (label NULL: one class) into thresh[C]; ws: sel_bytes(C)
static int sel_thresholds(const float* x, const int32_t* label, int64_t n, int C, float q, int mode, void* ws, float* thresh,
                          hipStream_t stream) {
  unsigned* h = (unsigned*)ws;
  const int64_t per = (int64_t)C * 2 * SEL_NB;
  SelState* st = (SelState*)(h + 4 * per);
  LT_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)sel_hist_ints(C) * 4, stream));
  int blocks = (int)((n + 256 * 8 - 1) / (256 * 8));
  blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
  const dim3 g(blocks), b(256), gc(C);
  hipLaunchKernelGGL(k_sel_hist<0>, g, b, 0, stream, x, label, n, C, (const SelState*)st, h);
  hipLaunchKernelGGL(k_sel_choose<0>, gc, b, 0, stream, (const unsigned*)h, q, mode, st, thresh);
  hipLaunchKernelGGL(k_sel_hist<1>, g, b, 0, stream, x, label, n, C, (const SelState*)st, h + per);
  hipLaunchKernelGGL(k_sel_choose<1>, gc, b, 0, stream, (const unsigned*)(h + per), q, mode, st, thresh);
  hipLaunchKernelGGL(k_sel_hist<2>, g, b, 0, stream, x, label, n, C, (const SelState*)st, h + 2 * per);
  hipLaunchKernelGGL(k_sel_choose<2>, gc, b, 0, stream, (const unsigned*)(h + 2 * per), q, mode, st, thresh);
  hipLaunchKernelGGL(k_sel_hist<3>, g, b, 0, stream, x, label, n, C, (const SelState*)st, h + 3 * per);
  hipLaunchKernelGGL(k_sel_choose<3>, gc, b, 0, stream, (const unsigned*)(h + 3 * per), q, mode, st, thresh);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- class-wise quantile mask (refine_ety)
// labels outside [0, num_classes) (an ignore label such as -100, or a class the caller did not count) belong to no class:
// they sort into the segment 0xFFFF, which no threshold is computed for, and k_cq_mask keeps them
__global__ void k_cq_keys(const float* __restrict__ ety, const int* __restrict__ label, int64_t n, int num_classes,
                          int64_t* __restrict__ keys) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned int bits = __float_as_uint(ety[i] + 0.0f);  // ety >= 0: IEEE bit pattern is order-preserving
    const int l = label[i];
    const int64_t cls = (l >= 0 && l < num_classes) ? l : 0xFFFF;
    keys[i] = (cls << 32) | (int64_t)bits;
  }
}

// one thread per class: segment of the class in the sorted keys -> torch.quantile threshold (fp32 arithmetic)
// mode 0: torch.quantile(q) with linear interpolation (refine_ety);  mode 1: torch.median (lower middle element) clamped
// to at most q (refine_pseudo_labels: thresh = min(median, 0.9))
__global__ void k_cq_thresh(const int64_t* __restrict__ skeys, int64_t n, int num_classes, float q, int mode,
                            float* __restrict__ thresh) {
#pragma clang fp contract(off)
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= num_classes) return;
  const int64_t kb = (int64_t)c << 32, ke = (int64_t)(c + 1) << 32;
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (skeys[m] < kb) lo = m + 1; else hi = m; }
  const int64_t b = lo;
  hi = n;
  while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (skeys[m] < ke) lo = m + 1; else hi = m; }
  const int64_t cnt = lo - b;
  if (cnt == 0) { thresh[c] = mode == 0 ? INFINITY : -INFINITY; return; }
  if (mode == 1) {
    const float med = __uint_as_float((unsigned int)(skeys[b + (cnt - 1) / 2] & 0xFFFFFFFFll));
    thresh[c] = med < q ? med : q;
    return;
  }
  const float rank = q * (float)(cnt - 1);
  const float fl = floorf(rank), ce = ceilf(rank);
  const float w = rank - fl;
  const float a = __uint_as_float((unsigned int)(skeys[b + (int64_t)fl] & 0xFFFFFFFFll));
  const float bb = __uint_as_float((unsigned int)(skeys[b + (int64_t)ce] & 0xFFFFFFFFll));
  const float d = bb - a;
  thresh[c] = (w < 0.5f) ? (a + w * d) : (bb - d * (1.0f - w));  // at::lerp
}

__global__ void k_cq_mask(const float* __restrict__ ety, const int* __restrict__ label, int64_t n, const float* __restrict__ thresh,
                          int num_classes, int mode, unsigned char* __restrict__ mask) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = label[i];
    if (l < 0 || l >= num_classes) { mask[i] = 1; continue; }   // no class: never masked
    const float t = thresh[l];
    mask[i] = mode == 0 ? ((ety[i] > t) ? 0 : 1) : ((ety[i] < t) ? 0 : 1);
  }
}

static int64_t cq_sort_bytes(int64_t n) {
  size_t b = 0;
  int64_t* k = nullptr;
  (void)rocprim::radix_sort_keys<LtSortConfig>(nullptr, b, k, k, (size_t)(n > 0 ? n : 1), 0, 48, (hipStream_t)0);
  return (int64_t)b;
}

extern "C" int64_t lt_class_quantile_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  const int64_t sort_path = 2 * lt_align_up(n * 8, 256) + lt_align_up(cq_sort_bytes(n), 256) + 65536 * 4 + 1024;
  const int64_t select_path = sel_bytes(SEL_MAX_CLASSES) + 65536 * 4 + 1024;
  return sort_path > select_path ? sort_path : select_path;
}

static int class_mask_impl(const float* ety, const int32_t* label, int64_t n, int32_t num_classes, float q, int mode,
                           void* ws, int64_t ws_bytes, uint8_t* mask, hipStream_t stream);

extern "C" int lt_class_quantile_mask(const float* ety, const int32_t* label, int64_t n, int32_t num_classes, float q,
                                      void* ws, int64_t ws_bytes, uint8_t* mask, hipStream_t stream) {
  return class_mask_impl(ety, label, n, num_classes, q, 0, ws, ws_bytes, mask, stream);
}

extern "C" int lt_class_median_mask(const float* vals, const int32_t* label, int64_t n, int32_t num_classes, float cap,
                                    void* ws, int64_t ws_bytes, uint8_t* keep, hipStream_t stream) {
  return class_mask_impl(vals, label, n, num_classes, cap, 1, ws, ws_bytes, keep, stream);
}

static int class_mask_impl(const float* ety, const int32_t* label, int64_t n, int32_t num_classes, float q, int mode,
                           void* ws, int64_t ws_bytes, uint8_t* mask, hipStream_t stream) {
  LT_REQUIRE(n >= 0 && num_classes >= 1 && num_classes < 65535, LT_ERR_ARG, "lt_class_quantile_mask: bad sizes");
  LT_REQUIRE(q >= 0.f && q <= 1.f, LT_ERR_ARG, "lt_class_quantile_mask: q must be in [0,1]");
  if (n == 0) return LT_OK;
  LT_REQUIRE(ety && label && ws && mask, LT_ERR_ARG, "lt_class_quantile_mask: null pointer");
  LT_REQUIRE(ws_bytes >= lt_class_quantile_workspace_bytes(n), LT_ERR_WORKSPACE, "lt_class_quantile_mask: workspace too small");
  LtArena a(ws, ws_bytes);
  if (num_classes <= SEL_MAX_CLASSES && n < (1ll << 32) && sel_enabled()) {   // selection: no sort (same thresholds)
    float* th = (float*)a.take((int64_t)num_classes * 4);
    void* sw = a.take(sel_bytes(num_classes));
    LT_REQUIRE(th && sw, LT_ERR_WORKSPACE, "lt_class_quantile_mask: arena exhausted");
    const int rc = sel_thresholds(ety, label, n, num_classes, q, mode, sw, th, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cq_mask, dim3(lt_grid(n, 256)), dim3(256), 0, stream, ety, label, n, th, num_classes, mode, mask);
    LT_LAUNCH_CHECK();
    return LT_OK;
  }
  int64_t* keys = (int64_t*)a.take(n * 8);
  int64_t* skeys = (int64_t*)a.take(n * 8);
  float* thresh = (float*)a.take((int64_t)num_classes * 4);
  const int64_t sb = cq_sort_bytes(n);
  void* tmp = a.take(sb);
  LT_REQUIRE(keys && skeys && thresh && tmp, LT_ERR_WORKSPACE, "lt_class_quantile_mask: arena exhausted");
  hipLaunchKernelGGL(k_cq_keys, dim3(lt_grid(n, 256)), dim3(256), 0, stream, ety, label, n, num_classes, keys);
  LT_LAUNCH_CHECK();
  size_t sbs = (size_t)sb;
  LT_CHECK_HIP(rocprim::radix_sort_keys<LtSortConfig>(tmp, sbs, keys, skeys, (size_t)n, 0, 48, stream));
  hipLaunchKernelGGL(k_cq_thresh, dim3(lt_cdiv(num_classes, 64)), dim3(64), 0, stream, skeys, n, num_classes, q, mode,
                     thresh);
  LT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cq_mask, dim3(lt_grid(n, 256)), dim3(256), 0, stream, ety, label, n, thresh, num_classes, mode, mask);
  LT_LAUNCH_CHECK();
  return LT_OK;
}


// ---------------------------------------------------------------- pseudo-label fusion (latte/tta/tools/latte.py:207-236)
// Teacher logits of both modalities -> entropy-weighted pseudo labels, two kernels instead of ~25 elementwise launches:
//   pc = per-point entropy of softmax(logits) (losses.py:10-19), replaced by the ST-voxel entropy st where st != 0 and
//        st <= thr (thr = the global conf_perc quantile, a device scalar; NULL: no quantile filter)      latte.py:207-222
//   all points: ps_prob = softmax(logits_3d);  in-FOV points: ps_prob = w2*softmax(logits_2d) + w3*softmax(logits_3d)
//        with w2 = rv2/(rv2+rv3), w3 = rv3/(rv2+rv3), rv = exp(-pc)                                      latte.py:224-233
//   ps_conf, ps_label = ps_prob.max(1)                                                                   latte.py:235
// Pass 1 (k_pl_points3d) runs over all N points, pass 2 (k_pl_fov) over the Nf in-FOV points and overwrites their rows.
template <int CM>
__global__ __launch_bounds__(128) void k_pl_points3d(const float* __restrict__ logits3, const float* __restrict__ st3, const float* __restrict__ thr3,
                              int64_t N, int C, float* __restrict__ probs3, float* __restrict__ pc3,
                              float* __restrict__ conf, int* __restrict__ label) {
  const float t = thr3 ? *thr3 : INFINITY;
  const float log2c = (float)log2((double)C);
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
    float v[CM];
    float e = st_softmax_entropy<CM>(logits3 + p * C, C, log2c, v);
    const float s = st3[p];
    if (s != 0.f && s <= t) e = s;
    pc3[p] = e;
    float best = v[0];
    int arg = 0;
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) {
        probs3[p * C + c] = v[c];
        if (v[c] > best) { best = v[c]; arg = c; }
      }
    conf[p] = best;
    label[p] = arg;
  }
}

template <int CM>
__global__ __launch_bounds__(128) void k_pl_fov(const float* __restrict__ logits2, const float* __restrict__ st2, const float* __restrict__ thr2,
                         const int64_t* __restrict__ fov_idx, const float* __restrict__ probs3,
                         const float* __restrict__ pc3, int64_t Nf, int C, float* __restrict__ pc2,
                         float* __restrict__ conf, int* __restrict__ label) {
#pragma clang fp contract(off)
  const float t = thr2 ? *thr2 : INFINITY;
  const float log2c = (float)log2((double)C);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < Nf; i += (int64_t)gridDim.x * blockDim.x) {
    float v[CM];
    float e = st_softmax_entropy<CM>(logits2 + i * C, C, log2c, v);
    const float s = st2[i];
    if (s != 0.f && s <= t) e = s;
    pc2[i] = e;
    const int64_t p = fov_idx[i];
    const float rv2 = expf(-e), rv3 = expf(-pc3[p]);
    const float w2 = rv2 / (rv2 + rv3), w3 = rv3 / (rv2 + rv3);
    float best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) {
        const float m = w2 * v[c] + w3 * probs3[p * C + c];
        if (m > best) { best = m; arg = c; }
      }
    conf[p] = best;
    label[p] = arg;
  }
}

extern "C" int lt_pseudo_label_fuse(const float* logits_2d, const float* logits_3d, const int64_t* fov_idx,
                                    const float* st_ety_2d, const float* st_ety_3d, const float* thr_2d,
                                    const float* thr_3d, int64_t N, int64_t Nf, int32_t C, float* probs_3d, float* pc_2d,
                                    float* pc_3d, float* ps_conf, int32_t* ps_label, hipStream_t stream) {
  LT_REQUIRE(N >= 0 && Nf >= 0 && C >= 1 && C <= 64, LT_ERR_ARG, "lt_pseudo_label_fuse: bad sizes (C=%d must be in [1,64])", C);
  if (N == 0) return LT_OK;
  LT_REQUIRE(logits_3d && st_ety_3d && probs_3d && pc_3d && ps_conf && ps_label, LT_ERR_ARG,
             "lt_pseudo_label_fuse: null pointer");
#define ST_CALL(CM)                                                                                                  \
  hipLaunchKernelGGL(k_pl_points3d<CM>, dim3(lt_grid(N, 128)), dim3(128), 0, stream, logits_3d, st_ety_3d, thr_3d, N, C, \
                     probs_3d, pc_3d, ps_conf, ps_label)
  ST_CM_DISPATCH(C, ST_CALL);
#undef ST_CALL
  LT_LAUNCH_CHECK();
  if (Nf > 0) {
    LT_REQUIRE(logits_2d && fov_idx && st_ety_2d && pc_2d, LT_ERR_ARG, "lt_pseudo_label_fuse: null pointer");
#define ST_CALL(CM)                                                                                                  \
  hipLaunchKernelGGL(k_pl_fov<CM>, dim3(lt_grid(Nf, 128)), dim3(128), 0, stream, logits_2d, st_ety_2d, thr_2d, fov_idx, \
                     probs_3d, pc_3d, Nf, C, pc_2d, ps_conf, ps_label)
    ST_CM_DISPATCH(C, ST_CALL);
#undef ST_CALL
    LT_LAUNCH_CHECK();
  }
  return LT_OK;
}


// ---------------------------------------------------------------- torch.quantile(x, q) of a 1-D float array
// (latte/tta/tools/latte.py:214-215: the conf_perc quantile of the ST-voxel entropies).  torch sorts (key, index) pairs;
// here: order-preserving 32-bit keys, one keys-only radix sort, one thread picks the two order statistics and
// interpolates with torch's own arithmetic (rank = q * (n - 1) in fp32, lerp as in ATen).  No sync: the result stays on the device.
__global__ void k_q_keys(const float* __restrict__ x, int64_t n, unsigned* __restrict__ keys) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned b = __float_as_uint(x[i]);
    keys[i] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending unsigned order = ascending float order
  }
}

__device__ __forceinline__ float q_unkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__global__ void k_q_pick(const unsigned* __restrict__ skeys, int64_t n, float q, float* __restrict__ out) {
#pragma clang fp contract(off)
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const float rank = q * (float)(n - 1);
  const float below = floorf(rank), above = ceilf(rank);
  const float w = rank - below;
  const float a = q_unkey(skeys[(int64_t)below]), b = q_unkey(skeys[(int64_t)above]);
  // at::lerp: a + w (b - a) for w < 0.5, b - (b - a)(1 - w) otherwise
  const float d = b - a;
  *out = w < 0.5f ? a + w * d : b - d * (1.0f - w);
}

static int64_t q_sort_bytes(int64_t n) {
  size_t b = 0;
  unsigned* k = nullptr;
  (void)rocprim::radix_sort_keys<LtSortConfig>(nullptr, b, k, k, (size_t)(n > 0 ? n : 1), 0, 32, (hipStream_t)0);
  return (int64_t)b;
}

extern "C" int64_t lt_quantile_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  const int64_t sort_path = 2 * lt_align_up(n * 4, 256) + lt_align_up(q_sort_bytes(n), 256) + 1024;
  const int64_t select_path = sel_bytes(1) + 1024;
  return sort_path > select_path ? sort_path : select_path;
}

extern "C" int lt_quantile(const float* x, int64_t n, float q, void* ws, int64_t ws_bytes, float* out,
                           hipStream_t stream) {
  LT_REQUIRE(n >= 1 && n < (1ll << 24), LT_ERR_RANGE, "lt_quantile: n=%lld must be in [1, 2^24) (ranks are fp32, as in torch)",
             (long long)n);
  LT_REQUIRE(q >= 0.f && q <= 1.f, LT_ERR_ARG, "lt_quantile: q must be in [0,1]");
  LT_REQUIRE(x && ws && out, LT_ERR_ARG, "lt_quantile: null pointer");
  LT_REQUIRE(ws_bytes >= lt_quantile_workspace_bytes(n), LT_ERR_WORKSPACE, "lt_quantile: workspace too small");
  LtArena a(ws, ws_bytes);
  if (sel_enabled()) {   // two histogram passes instead of the sort: the same two order statistics, the same lerp
    void* sw = a.take(sel_bytes(1));
    LT_REQUIRE(sw, LT_ERR_WORKSPACE, "lt_quantile: arena exhausted");
    return sel_thresholds(x, nullptr, n, 1, q, 0, sw, out, stream);
  }
  unsigned* keys = (unsigned*)a.take(n * 4);
  unsigned* skeys = (unsigned*)a.take(n * 4);
  const int64_t sb = q_sort_bytes(n);
  void* tmp = a.take(sb);
  LT_REQUIRE(keys && skeys && tmp, LT_ERR_WORKSPACE, "lt_quantile: arena exhausted");
  hipLaunchKernelGGL(k_q_keys, dim3(lt_grid(n, 256)), dim3(256), 0, stream, x, n, keys);
  LT_LAUNCH_CHECK();
  size_t sbs = (size_t)sb;
  LT_CHECK_HIP(rocprim::radix_sort_keys<LtSortConfig>(tmp, sbs, keys, skeys, (size_t)n, 0, 32, stream));
  hipLaunchKernelGGL(k_q_pick, dim3(1), dim3(64), 0, stream, skeys, n, q, out);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// CR_Block (latte/models/hr_module.py:344-420): entropy-weighted KL between matched anchor / hist ST-voxel distributions,
// mean over the kept rows of every anchor, mean over the anchors.  The reference loops over the anchors in Python with ~12
// elementwise launches each; here all anchors are rows of one matrix (rows sorted by anchor: anchor_start[B+1]) and the
// loss is three kernels forward (rows, a fixed-order reduction per anchor, the mean over the anchors) and one backward.
//   w[m]    = exp(-t_e) / (exp(-t_e) + exp(-a_e)),  a_e = ac_pc_ety[ac_p2v[m]], t_e = tr_pc_ety[tr_p2v[m]]   (:373-377)
//   row[m]  = w[m] * sum_c tr~ * (log tr~ - log ac~),  x~ = (x == 0 ? 1e-8 : x): the clamp acts on PROBABILITIES (Q6, :405-406)
//   loss    = 1/B * sum_b  sum_{m in b, keep[m]} row[m] / #{m in b : keep[m]}                                 (:407-420)
// Gradient to ac only (tr is detached, :409): d row / d ac[m,c] = -w[m] * tr~[m,c] / ac[m,c], 0 where ac == 0 (the clamp
// replaced the element by a constant).
#define CR_MAXC 64
__global__ void k_cr_rows(const float* __restrict__ ac, const float* __restrict__ tr, int64_t M, int C,
                          const float* __restrict__ ac_ety, const int64_t* __restrict__ ac_p2v,
                          const float* __restrict__ tr_ety, const int64_t* __restrict__ tr_p2v, float* __restrict__ row_w,
                          float* __restrict__ row_loss) {
#pragma clang fp contract(off)
  for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    float w = 1.f;
    if (ac_ety) {
      const float ea = expf(-ac_ety[ac_p2v[m]]), et = expf(-tr_ety[tr_p2v[m]]);
      w = et / (et + ea);
    }
    const float* a = ac + m * C;
    const float* t = tr + m * C;
    float kl = 0.f;
    for (int c = 0; c < C; ++c) {
      float av = a[c], tv = t[c];
      if (av == 0.f) av = 1e-8f;
      if (tv == 0.f) tv = 1e-8f;
      kl += tv * (logf(tv) - logf(av));   // F.kl_div(log ac, tr, 'none'), summed over the classes in class order
    }
    row_w[m] = w;
    row_loss[m] = w * kl;
  }
}

// One workgroup per anchor: a fixed-order block reduction of (sum of kept rows, number of kept rows); stats[b] = (num_b,
// den_b) is kept for the backward.  Then one wave takes the mean over the anchors in anchor order.  (A single workgroup
// walking all anchors cost 75 us a launch on the 64-anchor workload; this is B-way parallel and ~5 us.)
__global__ __launch_bounds__(1024) void k_cr_reduce(const float* __restrict__ row_loss, const uint8_t* __restrict__ keep,
                                                    const int* __restrict__ anchor_start, float* __restrict__ stats) {
  __shared__ float s_num[1024], s_den[1024];
  const int t = threadIdx.x, b = blockIdx.x;
  const int r0 = anchor_start[b], r1 = anchor_start[b + 1];
  float num = 0.f, den = 0.f;
  for (int m = r0 + t; m < r1; m += 1024) {
    const float k = keep ? (keep[m] ? 1.f : 0.f) : 1.f;
    num += k * row_loss[m];
    den += k;
  }
  s_num[t] = num;
  s_den[t] = den;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (t < s) { s_num[t] += s_num[t + s]; s_den[t] += s_den[t + s]; }
    __syncthreads();
  }
  if (t == 0) {
    stats[2 * b] = s_num[0];
    stats[2 * b + 1] = s_den[0];
  }
}

__global__ void k_cr_final(const float* __restrict__ stats, int B, float* __restrict__ loss) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float total = 0.f;
    for (int b = 0; b < B; ++b)
      total += stats[2 * b] / stats[2 * b + 1];   // an anchor without a kept row: 0/0 = NaN, like the reference's mean of nothing
    loss[0] = total / (float)B;
  }
}

__global__ void k_cr_bwd(const float* __restrict__ ac, const float* __restrict__ tr, int64_t M, int C,
                         const float* __restrict__ row_w, const uint8_t* __restrict__ keep,
                         const int64_t* __restrict__ row_anchor, const float* __restrict__ stats, int B,
                         const float* __restrict__ gloss, float* __restrict__ g_ac) {
  const float g = gloss[0] / (float)B;
  const int64_t total = M * (int64_t)C;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = e / C;
    float out = 0.f;
    if (!keep || keep[m]) {
      const float av = ac[e];
      if (av != 0.f) {
        float tv = tr[e];
        if (tv == 0.f) tv = 1e-8f;
        out = -(g * row_w[m] / stats[2 * row_anchor[m] + 1]) * (tv / av);
      }
    }
    g_ac[e] = out;
  }
}

static int cr_check(const char* who, int64_t M, int32_t C, int32_t B) {
  LT_REQUIRE(M >= 0 && M < INT_MAX, LT_ERR_RANGE, "%s: M out of range", who);
  LT_REQUIRE(C >= 1 && C <= CR_MAXC, LT_ERR_ARG, "%s: C=%d must be in [1,%d]", who, C, CR_MAXC);
  LT_REQUIRE(B >= 1, LT_ERR_ARG, "%s: B=%d anchors", who, B);
  return LT_OK;
}

extern "C" int lt_cr_loss_fwd(const float* ac, const float* tr, int64_t M, int32_t C, const float* ac_pc_ety,
                              const int64_t* ac_p2v, const float* tr_pc_ety, const int64_t* tr_p2v, const uint8_t* keep,
                              const int32_t* anchor_start, int32_t B, float* row_w, float* row_loss, float* anchor_stats,
                              float* loss, hipStream_t stream) {
  int rc = cr_check("lt_cr_loss_fwd", M, C, B);
  if (rc) return rc;
  LT_REQUIRE(anchor_start && anchor_stats && loss && (M == 0 || (ac && tr && row_w && row_loss)), LT_ERR_ARG,
             "lt_cr_loss_fwd: null pointer");
  const bool ety = ac_pc_ety || ac_p2v || tr_pc_ety || tr_p2v;
  LT_REQUIRE(!ety || (ac_pc_ety && ac_p2v && tr_pc_ety && tr_p2v), LT_ERR_ARG,
             "lt_cr_loss_fwd: the four entropy-weight arguments come together (all NULL: weight 1)");
  if (M > 0) {
    hipLaunchKernelGGL(k_cr_rows, dim3(lt_grid(M, 256)), dim3(256), 0, stream, ac, tr, M, C, ac_pc_ety, ac_p2v, tr_pc_ety,
                       tr_p2v, row_w, row_loss);
    LT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_cr_reduce, dim3(B), dim3(1024), 0, stream, row_loss, keep, anchor_start, anchor_stats);
  LT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cr_final, dim3(1), dim3(64), 0, stream, anchor_stats, B, loss);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

extern "C" int lt_cr_loss_bwd(const float* ac, const float* tr, int64_t M, int32_t C, const float* row_w, const uint8_t* keep,
                              const int64_t* row_anchor, const float* anchor_stats, int32_t B, const float* gloss,
                              float* g_ac, hipStream_t stream) {
  int rc = cr_check("lt_cr_loss_bwd", M, C, B);
  if (rc) return rc;
  if (M == 0) return LT_OK;
  LT_REQUIRE(ac && tr && row_w && row_anchor && anchor_stats && gloss && g_ac, LT_ERR_ARG, "lt_cr_loss_bwd: null pointer");
  hipLaunchKernelGGL(k_cr_bwd, dim3(lt_grid(M * (int64_t)C, 256)), dim3(256), 0, stream, ac, tr, M, C, row_w, keep, row_anchor,
                     anchor_stats, B, gloss, g_ac);
  LT_LAUNCH_CHECK();
  return LT_OK;
}
