// Point <-> voxel bridges of SPVCNN (torchsparse v1.4.0 F.spvoxelize / F.calc_ti_weights / F.spdevoxelize).
// Reference call sites: latte/models/torchsparse_utils.py:20,23,54 (voxelize), :75 (weights), :81,93 (devoxelize).
// All HBM-bound: one lane per 16-B channel group, rows contiguous across lanes.
#include "common.h"
#include "../../include/latte_hip.h"
#include <rocprim/device/device_radix_sort.hpp>

// ---------------------------------------------------------------- spvoxelize (average pooling of points into voxels)
// out[idx[i], :] += feats[i, :] / counts[idx[i]]   -- same divide-then-add order as the reference kernel.
__global__ void k_voxelize_fwd(const float* __restrict__ f, const int* __restrict__ idx, const int* __restrict__ counts,
                               float* __restrict__ out, int64_t N, int C, int64_t V) {
  const int64_t total = N * (int64_t)C;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / C;
    const int c = (int)(e - i * C);
    const int pos = idx[i];
    if (pos < 0 || pos >= V) continue;
    const int cnt = counts[pos];
    if (cnt == 0) continue;
    atomicAdd(&out[(int64_t)pos * C + c], f[e] / (float)cnt);
  }
}

__global__ void k_voxelize_bwd(const float* __restrict__ g, const int* __restrict__ idx, const int* __restrict__ counts,
                               float* __restrict__ gf, int64_t N, int C, int64_t V) {
  const int64_t total = N * (int64_t)C;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / C;
    const int c = (int)(e - i * C);
    const int pos = idx[i];
    float v = 0.f;
    if (pos >= 0 && pos < V) {
      const int cnt = counts[pos];
      if (cnt > 0) v = g[(int64_t)pos * C + c] / (float)cnt;
    }
    gf[e] = v;
  }
}

// the same gather over float4 column groups, optionally ADDED to what gf already holds (the gradient of a tensor with a
// second reader: lt_net_backward) -- same value per element as the scalar kernel, then one add
__global__ void k_voxelize_bwd4(const float4* __restrict__ g, const int* __restrict__ idx, const int* __restrict__ counts,
                                float4* __restrict__ gf, int64_t N, int C4, int64_t V, int acc) {
  const int64_t total = N * (int64_t)C4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / C4;
    const int c = (int)(e - i * C4);
    const int pos = idx[i];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pos >= 0 && pos < V) {
      const int cnt = counts[pos];
      if (cnt > 0) {
        const float4 t = g[(int64_t)pos * C4 + c];
        const float d = (float)cnt;
        v = make_float4(t.x / d, t.y / d, t.z / d, t.w / d);
      }
    }
    if (acc) {
      const float4 o = gf[e];
      v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
    }
    gf[e] = v;
  }
}

int lt_voxelize_bwd_acc(const float* gout, const int32_t* idx, const int32_t* counts, float* gfeats, int64_t N, int32_t C,
                        int64_t V, int accumulate, hipStream_t stream) {
  LT_REQUIRE(N >= 0 && V >= 0 && C >= 4 && C % 4 == 0, LT_ERR_ARG, "lt_voxelize_bwd_acc: bad sizes (C %% 4 == 0)");
  if (N == 0) return LT_OK;
  LT_REQUIRE(gfeats && idx && counts && (V == 0 || gout), LT_ERR_ARG, "lt_voxelize_bwd_acc: null pointer");
  hipLaunchKernelGGL(k_voxelize_bwd4, dim3(lt_grid(N * (C / 4), 256)), dim3(256), 0, stream, (const float4*)gout, idx, counts,
                     (float4*)gfeats, N, C / 4, V, accumulate);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

extern "C" int lt_voxelize_fwd(const float* feats, const int32_t* idx, const int32_t* counts, float* out, int64_t N,
                               int32_t C, int64_t V, hipStream_t stream) {
  LT_REQUIRE(N >= 0 && V >= 0 && C >= 1, LT_ERR_ARG, "lt_voxelize_fwd: bad sizes");
  if (V == 0) return LT_OK;
  LT_REQUIRE(out && (N == 0 || (feats && idx && counts)), LT_ERR_ARG, "lt_voxelize_fwd: null pointer");
  LT_CHECK_HIP(hipMemsetAsync(out, 0, V * (int64_t)C * 4, stream));
  if (N > 0) {
    hipLaunchKernelGGL(k_voxelize_fwd, dim3(lt_grid(N * C, 256)), dim3(256), 0, stream, feats, idx, counts, out, N, C, V);
    LT_LAUNCH_CHECK();
  }
  return LT_OK;
}

extern "C" int lt_voxelize_bwd(const float* gout, const int32_t* idx, const int32_t* counts, float* gfeats, int64_t N,
                               int32_t C, int64_t V, hipStream_t stream) {
  LT_REQUIRE(N >= 0 && V >= 0 && C >= 1, LT_ERR_ARG, "lt_voxelize_bwd: bad sizes");
  if (N == 0) return LT_OK;
  LT_REQUIRE(gfeats && idx && counts && (V == 0 || gout), LT_ERR_ARG, "lt_voxelize_bwd: null pointer");
  hipLaunchKernelGGL(k_voxelize_bwd, dim3(lt_grid(N * C, 256)), dim3(256), 0, stream, gout, idx, counts, gfeats, N, C, V);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- trilinear weights
// Same fp32 operation order as torchsparse's torch expression: pf = floor(p/scale)*scale (scale != 1) or floor(p);
// pc = pf + scale; w_k = (..)*(..)*(..); w /= scale^3 (scale != 1); w[idx==-1] = 0; w /= (sum_k w + 1e-8).
__global__ void k_ti_weights(const float* __restrict__ pc4, const int64_t* __restrict__ idxq, int64_t N, int scale,
                             float* __restrict__ w) {
#pragma clang fp contract(off)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = reinterpret_cast<const float4*>(pc4)[i];
    const float s = (float)scale;
    float xf, yf, zf;
    if (scale != 1) {
      xf = floorf(p.x / s) * s; yf = floorf(p.y / s) * s; zf = floorf(p.z / s) * s;
    } else {
      xf = floorf(p.x); yf = floorf(p.y); zf = floorf(p.z);
    }
    const float xc = xf + s, yc = yf + s, zc = zf + s;
    const float ax = xc - p.x, ay = yc - p.y, az = zc - p.z;   // weight towards the floor corner
    const float bx = p.x - xf, by = p.y - yf, bz = p.z - zf;   // weight towards the ceil corner
    float wk[8];
    wk[0] = ax * ay * az; wk[1] = ax * ay * bz; wk[2] = ax * by * az; wk[3] = ax * by * bz;
    wk[4] = bx * ay * az; wk[5] = bx * ay * bz; wk[6] = bx * by * az; wk[7] = bx * by * bz;
    const float s3 = (float)(scale * scale * scale);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (scale != 1) wk[k] = wk[k] / s3;
      if (idxq[(int64_t)k * N + i] == -1) wk[k] = 0.f;
      sum += wk[k];
    }
    const float den = sum + 1e-8f;
#pragma unroll
    for (int k = 0; k < 8; ++k) w[(int64_t)k * N + i] = wk[k] / den;
  }
}

extern "C" int lt_calc_ti_weights(const float* pcoords, const int64_t* idx_query, int64_t N, int32_t scale, float* w,
                                  hipStream_t stream) {
  LT_REQUIRE(N >= 0 && scale >= 1, LT_ERR_ARG, "lt_calc_ti_weights: bad sizes");
  if (N == 0) return LT_OK;
  LT_REQUIRE(pcoords && idx_query && w, LT_ERR_ARG, "lt_calc_ti_weights: null pointer");
  hipLaunchKernelGGL(k_ti_weights, dim3(lt_grid(N, 256)), dim3(256), 0, stream, pcoords, idx_query, N, scale, w);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- spdevoxelize
// out[i,:] = sum_{k<8} w[i,k] * feats[idx[i,k],:]  (k ascending, idx<0 skipped).  One lane per float4 of a row.
__global__ void k_devox_fwd(const float* __restrict__ f, const int* __restrict__ idx, const float* __restrict__ w,
                            float* __restrict__ out, int64_t N, int C4, int64_t V) {
#pragma clang fp contract(off)
  const int64_t total = N * (int64_t)C4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / C4;
    const int c4 = (int)(e - i * C4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int v = idx[i * 8 + k];
      const float wk = w[i * 8 + k];
      float4 fv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v >= 0 && v < V) fv = reinterpret_cast<const float4*>(f + (int64_t)v * C4 * 4)[c4];
      acc.x += wk * fv.x; acc.y += wk * fv.y; acc.z += wk * fv.z; acc.w += wk * fv.w;
    }
    reinterpret_cast<float4*>(out)[e] = acc;
  }
}

// gfeats[idx[i,k],:] += w[i,k] * gout[i,:]
__global__ void k_devox_bwd(const float* __restrict__ g, const int* __restrict__ idx, const float* __restrict__ w,
                            float* __restrict__ gf, int64_t N, int C, int64_t V) {
  const int64_t total = N * (int64_t)C;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / C;
    const int c = (int)(e - i * C);
    const float gv = g[e];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int v = idx[i * 8 + k];
      const float wk = w[i * 8 + k];
      if (v >= 0 && v < V && wk != 0.f) atomicAdd(&gf[(int64_t)v * C + c], wk * gv);
    }
  }
}

extern "C" int lt_devoxelize_fwd(const float* feats, const int32_t* idx, const float* w, float* out, int64_t N,
                                 int32_t C, int64_t V, hipStream_t stream) {
  LT_REQUIRE(N >= 0 && V >= 0 && C >= 4 && C % 4 == 0, LT_ERR_ARG, "lt_devoxelize_fwd: bad sizes (C %% 4 == 0 required)");
  if (N == 0) return LT_OK;
  LT_REQUIRE(out && idx && w && (V == 0 || feats), LT_ERR_ARG, "lt_devoxelize_fwd: null pointer");
  hipLaunchKernelGGL(k_devox_fwd, dim3(lt_grid(N * (C / 4), 256)), dim3(256), 0, stream, feats, idx, w, out, N, C / 4, V);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

extern "C" int lt_devoxelize_bwd(const float* gout, const int32_t* idx, const float* w, float* gfeats, int64_t N,
                                 int32_t C, int64_t V, hipStream_t stream) {
  LT_REQUIRE(N >= 0 && V >= 0 && C >= 1, LT_ERR_ARG, "lt_devoxelize_bwd: bad sizes");
  if (V == 0) return LT_OK;
  LT_REQUIRE(gfeats && (N == 0 || (gout && idx && w)), LT_ERR_ARG, "lt_devoxelize_bwd: null pointer");
  LT_CHECK_HIP(hipMemsetAsync(gfeats, 0, V * (int64_t)C * 4, stream));
  if (N > 0) {
    hipLaunchKernelGGL(k_devox_bwd, dim3(lt_grid(N * C, 256)), dim3(256), 0, stream, gout, idx, w, gfeats, N, C, V);
    LT_LAUNCH_CHECK();
  }
  return LT_OK;
}

// ---------------------------------------------------------------- CSR of a scatter index (atomic-free scatter-adds)
// The two scatter-adds above (spvoxelize forward, spdevoxelize backward) as gathers: the entries e with key[e] = v are
// listed per target row v (perm[off[v] .. off[v+1]), ascending e), built once per index map by one stable radix sort.
// Summation order = ascending entry = the order of a sequential index_add: deterministic, no memset, no atomics.
__global__ void k_csr_keys(const int* __restrict__ key, int64_t n, int V, unsigned int* __restrict__ k2, int* __restrict__ iota) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = key[e];
    k2[e] = (k < 0 || k >= V) ? (unsigned int)V : (unsigned int)k;  // dropped entries sort behind every row
    iota[e] = (int)e;
  }
}
// off[v] = first position of a key >= v in the sorted keys, v = 0..V
__global__ void k_csr_offsets(const unsigned int* __restrict__ sk, int64_t n, int V, int* __restrict__ off) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v <= V; v += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sk[mid] < (unsigned int)v) lo = mid + 1; else hi = mid;
    }
    off[v] = (int)lo;
  }
}

static int64_t csr_sort_bytes(int64_t n) {
  size_t b = 0;
  unsigned int* k = nullptr;
  int* v = nullptr;
  (void)rocprim::radix_sort_pairs<LtSortConfig>(nullptr, b, k, k, v, v, (size_t)(n > 0 ? n : 1), 0, 32, (hipStream_t)0);
  return (int64_t)b;
}

extern "C" int64_t lt_csr_workspace_bytes(int64_t n) {
  const int64_t m = n > 0 ? n : 1;
  return lt_align_up(csr_sort_bytes(m), 256) + 3 * lt_align_up(m * 4, 256) + 256;
}

extern "C" int lt_csr_build(const int32_t* key, int64_t n, int64_t V, void* ws, int64_t ws_bytes, int32_t* perm, int32_t* off,
                            hipStream_t stream) {
  LT_REQUIRE(n >= 0 && V >= 0 && n < INT32_MAX && V < INT32_MAX, LT_ERR_RANGE, "lt_csr_build: bad sizes");
  LT_REQUIRE(off && (n == 0 || (key && perm)), LT_ERR_ARG, "lt_csr_build: null pointer");
  LT_REQUIRE(ws && ws_bytes >= lt_csr_workspace_bytes(n), LT_ERR_WORKSPACE, "lt_csr_build: workspace too small");
  if (n == 0) {
    LT_CHECK_HIP(hipMemsetAsync(off, 0, (V + 1) * 4, stream));
    return LT_OK;
  }
  LtArena ar(ws, ws_bytes);
  size_t sbs = (size_t)csr_sort_bytes(n);
  void* tmp = ar.take(sbs);
  unsigned int* k2 = (unsigned int*)ar.take(n * 4);
  unsigned int* sk = (unsigned int*)ar.take(n * 4);
  int* iota = (int*)ar.take(n * 4);
  hipLaunchKernelGGL(k_csr_keys, dim3(lt_grid(n, 256)), dim3(256), 0, stream, key, n, (int)V, k2, iota);
  LT_LAUNCH_CHECK();
  unsigned bits = 1;
  while ((1ll << bits) <= V) ++bits;
  LT_CHECK_HIP(rocprim::radix_sort_pairs<LtSortConfig>(tmp, sbs, k2, sk, iota, perm, (size_t)n, 0, bits, stream));
  hipLaunchKernelGGL(k_csr_offsets, dim3(lt_grid(V + 1, 256)), dim3(256), 0, stream, sk, n, (int)V, off);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// out[v,:] = sum_{j in row v} feats[perm[j],:] / (row length)      (divide, then add, ascending point: as k_voxelize_fwd)
__global__ void k_voxelize_csr(const float* __restrict__ f, const int* __restrict__ perm, const int* __restrict__ off,
                               float* __restrict__ out, int64_t V, int C4) {
  const int64_t total = V * (int64_t)C4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = e / C4;
    const int c4 = (int)(e - v * C4);
    const int j0 = off[v], j1 = off[v + 1];
    const float cnt = (float)(j1 - j0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // 4 gathers in flight (the coarse levels have 100+ points per voxel: one dependent load chain per element was
    // latency-bound); the adds stay in ascending point order
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
      const int p0 = perm[j], p1 = perm[j + 1], p2 = perm[j + 2], p3 = perm[j + 3];
      const float4 x0 = reinterpret_cast<const float4*>(f + (int64_t)p0 * C4 * 4)[c4];
      const float4 x1 = reinterpret_cast<const float4*>(f + (int64_t)p1 * C4 * 4)[c4];
      const float4 x2 = reinterpret_cast<const float4*>(f + (int64_t)p2 * C4 * 4)[c4];
      const float4 x3 = reinterpret_cast<const float4*>(f + (int64_t)p3 * C4 * 4)[c4];
      acc.x += x0.x / cnt; acc.y += x0.y / cnt; acc.z += x0.z / cnt; acc.w += x0.w / cnt;
      acc.x += x1.x / cnt; acc.y += x1.y / cnt; acc.z += x1.z / cnt; acc.w += x1.w / cnt;
      acc.x += x2.x / cnt; acc.y += x2.y / cnt; acc.z += x2.z / cnt; acc.w += x2.w / cnt;
      acc.x += x3.x / cnt; acc.y += x3.y / cnt; acc.z += x3.z / cnt; acc.w += x3.w / cnt;
    }
    for (; j < j1; ++j) {
      const float4 x = reinterpret_cast<const float4*>(f + (int64_t)perm[j] * C4 * 4)[c4];
      acc.x += x.x / cnt; acc.y += x.y / cnt; acc.z += x.z / cnt; acc.w += x.w / cnt;
    }
    reinterpret_cast<float4*>(out)[e] = acc;
  }
}

extern "C" int lt_voxelize_csr_fwd(const float* feats, const int32_t* perm, const int32_t* off, float* out, int64_t V,
                                   int32_t C, hipStream_t stream) {
  LT_REQUIRE(V >= 0 && C >= 4 && C % 4 == 0, LT_ERR_ARG, "lt_voxelize_csr_fwd: bad sizes (C %% 4 == 0 required)");
  if (V == 0) return LT_OK;
  LT_REQUIRE(feats && perm && off && out, LT_ERR_ARG, "lt_voxelize_csr_fwd: null pointer");
  hipLaunchKernelGGL(k_voxelize_csr, dim3(lt_grid(V * (C / 4), 256)), dim3(256), 0, stream, feats, perm, off, out, V, C / 4);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// gfeats[v,:] = sum_{j in row v} w[perm[j]] * gout[perm[j] / 8,:]   (entries e = 8*point + corner, ascending)
__global__ void k_devox_csr_bwd(const float* __restrict__ g, const int* __restrict__ perm, const int* __restrict__ off,
                                const float* __restrict__ w, float* __restrict__ gf, int64_t V, int C4) {
#pragma clang fp contract(off)
  const int64_t total = V * (int64_t)C4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = e / C4;
    const int c4 = (int)(e - v * C4);
    const int j0 = off[v], j1 = off[v + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // 4 gathers in flight, adds in ascending entry order (see k_voxelize_csr)
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
      const int e0 = perm[j], e1 = perm[j + 1], e2 = perm[j + 2], e3 = perm[j + 3];
      const float w0 = w[e0], w1 = w[e1], w2 = w[e2], w3 = w[e3];
      const float4 x0 = reinterpret_cast<const float4*>(g + (int64_t)(e0 >> 3) * C4 * 4)[c4];
      const float4 x1 = reinterpret_cast<const float4*>(g + (int64_t)(e1 >> 3) * C4 * 4)[c4];
      const float4 x2 = reinterpret_cast<const float4*>(g + (int64_t)(e2 >> 3) * C4 * 4)[c4];
      const float4 x3 = reinterpret_cast<const float4*>(g + (int64_t)(e3 >> 3) * C4 * 4)[c4];
      acc.x += w0 * x0.x; acc.y += w0 * x0.y; acc.z += w0 * x0.z; acc.w += w0 * x0.w;
      acc.x += w1 * x1.x; acc.y += w1 * x1.y; acc.z += w1 * x1.z; acc.w += w1 * x1.w;
      acc.x += w2 * x2.x; acc.y += w2 * x2.y; acc.z += w2 * x2.z; acc.w += w2 * x2.w;
      acc.x += w3 * x3.x; acc.y += w3 * x3.y; acc.z += w3 * x3.z; acc.w += w3 * x3.w;
    }
    for (; j < j1; ++j) {
      const int en = perm[j];
      const float wk = w[en];
      const float4 x = reinterpret_cast<const float4*>(g + (int64_t)(en >> 3) * C4 * 4)[c4];
      acc.x += wk * x.x; acc.y += wk * x.y; acc.z += wk * x.z; acc.w += wk * x.w;
    }
    reinterpret_cast<float4*>(gf)[e] = acc;
  }
}

extern "C" int lt_devoxelize_csr_bwd(const float* gout, const int32_t* perm, const int32_t* off, const float* w,
                                     float* gfeats, int64_t V, int32_t C, hipStream_t stream) {
  LT_REQUIRE(V >= 0 && C >= 4 && C % 4 == 0, LT_ERR_ARG, "lt_devoxelize_csr_bwd: bad sizes (C %% 4 == 0 required)");
  if (V == 0) return LT_OK;
  LT_REQUIRE(gout && perm && off && w && gfeats, LT_ERR_ARG, "lt_devoxelize_csr_bwd: null pointer");
  hipLaunchKernelGGL(k_devox_csr_bwd, dim3(lt_grid(V * (C / 4), 256)), dim3(256), 0, stream, gout, perm, off, w, gfeats, V,
                     C / 4);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- devoxelize backward over the point -> voxel CSR
// All points of one cell u (floor(z.C / s): the point -> voxel map of the level) query the same 8 corner hashes, so they
// share their corner voxels: idx[p, c] = corner_c(u).  The points that touch voxel v through corner c are therefore the
// points of ONE cell, cells[v, c], and the backward needs no CSR of its own (lt_csr_build over 8N keys: three radix sorts
// of 6.3 M keys per batch at the a2d2 sizes) -- only this [V, 8] table, filled from the first point of every cell.
__global__ void k_devox_cells(const int* __restrict__ idx, const int* __restrict__ pperm, const int* __restrict__ poff,
                              int2* __restrict__ cells, int64_t Vp, int64_t V) {
  const int64_t total = Vp * 8;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = e >> 3;
    const int c = (int)(e & 7);
    const int j0 = poff[u], j1 = poff[u + 1];
    if (j1 <= j0) continue;
    const int v = idx[(int64_t)pperm[j0] * 8 + c];
    if (v >= 0 && v < V) cells[(int64_t)v * 8 + c] = make_int2(j0, j1);  // one writer: u = the cell at v - offset_c
  }
}

// gfeats[v,:] = sum_c sum_{j in cells[v,c]} w[pperm[j],c] * gout[pperm[j],:]   (corner by corner, points ascending)
// The first point of all 8 rows is fetched up front (most cells of the fine levels hold one or two points: 8 independent
// gathers in flight instead of 8 dependent chains); the sum is still formed in (corner, point) order.
__global__ void k_devox_cell_bwd(const float* __restrict__ g, const int2* __restrict__ cells, const int* __restrict__ pperm,
                                 const float* __restrict__ w, float* __restrict__ gf, int64_t V, int C4) {
#pragma clang fp contract(off)
  const int64_t total = V * (int64_t)C4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = e / C4;
    const int c4 = (int)(e - v * C4);
    int2 seg[8];
    const int4* cv = reinterpret_cast<const int4*>(cells + v * 8);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int4 t = cv[c];
      seg[2 * c] = make_int2(t.x, t.y);
      seg[2 * c + 1] = make_int2(t.z, t.w);
    }
    int p0[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) p0[c] = seg[c].x < seg[c].y ? pperm[seg[c].x] : -1;
    float w0[8];
    float4 x0[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int p = p0[c] < 0 ? 0 : p0[c];
      w0[c] = p0[c] < 0 ? 0.f : w[(int64_t)p * 8 + c];
      x0[c] = p0[c] < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4*>(g + (int64_t)p * C4 * 4)[c4];
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (p0[c] < 0) continue;
      acc.x += w0[c] * x0[c].x; acc.y += w0[c] * x0[c].y; acc.z += w0[c] * x0[c].z; acc.w += w0[c] * x0[c].w;
      int j = seg[c].x + 1;
      const int j1 = seg[c].y;
      for (; j + 4 <= j1; j += 4) {  // 4 gathers in flight, adds in ascending point order
        const int q0 = pperm[j], q1 = pperm[j + 1], q2 = pperm[j + 2], q3 = pperm[j + 3];
        const float a0 = w[(int64_t)q0 * 8 + c], a1 = w[(int64_t)q1 * 8 + c], a2 = w[(int64_t)q2 * 8 + c],
                    a3 = w[(int64_t)q3 * 8 + c];
        const float4 y0 = reinterpret_cast<const float4*>(g + (int64_t)q0 * C4 * 4)[c4];
        const float4 y1 = reinterpret_cast<const float4*>(g + (int64_t)q1 * C4 * 4)[c4];
        const float4 y2 = reinterpret_cast<const float4*>(g + (int64_t)q2 * C4 * 4)[c4];
        const float4 y3 = reinterpret_cast<const float4*>(g + (int64_t)q3 * C4 * 4)[c4];
        acc.x += a0 * y0.x; acc.y += a0 * y0.y; acc.z += a0 * y0.z; acc.w += a0 * y0.w;
        acc.x += a1 * y1.x; acc.y += a1 * y1.y; acc.z += a1 * y1.z; acc.w += a1 * y1.w;
        acc.x += a2 * y2.x; acc.y += a2 * y2.y; acc.z += a2 * y2.z; acc.w += a2 * y2.w;
        acc.x += a3 * y3.x; acc.y += a3 * y3.y; acc.z += a3 * y3.z; acc.w += a3 * y3.w;
      }
      for (; j < j1; ++j) {
        const int q = pperm[j];
        const float a = w[(int64_t)q * 8 + c];
        const float4 y = reinterpret_cast<const float4*>(g + (int64_t)q * C4 * 4)[c4];
        acc.x += a * y.x; acc.y += a * y.y; acc.z += a * y.z; acc.w += a * y.w;
      }
    }
    reinterpret_cast<float4*>(gf)[e] = acc;
  }
}

// points whose cell is not a voxel of the level (pidx < 0: never with SPVCNN, whose coarse levels contain every point's
// cell) are in no CSR row: their contributions are added here, atomically
__global__ void k_devox_unmapped(const float* __restrict__ g, const int* __restrict__ pidx, const int* __restrict__ idx,
                                 const float* __restrict__ w, float* __restrict__ gf, int64_t N, int64_t V, int C) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
    if (pidx[p] >= 0) continue;
    for (int c = 0; c < 8; ++c) {
      const int v = idx[p * 8 + c];
      if (v < 0 || v >= V) continue;
      const float wk = w[p * 8 + c];
      for (int ch = 0; ch < C; ++ch) atomicAdd(gf + (int64_t)v * C + ch, wk * g[p * C + ch]);
    }
  }
}

extern "C" int lt_devoxelize_cells(const int32_t* v2p_idx, const int32_t* p2v_perm, const int32_t* p2v_off, int32_t* cells,
                                   int64_t N, int64_t Vp, int64_t V, hipStream_t stream) {
  LT_REQUIRE(N >= 0 && Vp >= 0 && V >= 0, LT_ERR_ARG, "lt_devoxelize_cells: bad sizes");
  if (V == 0) return LT_OK;
  LT_REQUIRE(cells, LT_ERR_ARG, "lt_devoxelize_cells: null pointer");
  LT_CHECK_HIP(hipMemsetAsync(cells, 0, V * 8 * 2 * sizeof(int32_t), stream));   // (0, 0): an empty row
  if (N == 0 || Vp == 0) return LT_OK;
  LT_REQUIRE(v2p_idx && p2v_perm && p2v_off, LT_ERR_ARG, "lt_devoxelize_cells: null pointer");
  hipLaunchKernelGGL(k_devox_cells, dim3(lt_grid(Vp * 8, 256)), dim3(256), 0, stream, v2p_idx, p2v_perm, p2v_off, (int2*)cells,
                     Vp, V);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

extern "C" int lt_devoxelize_cell_bwd(const float* gout, const int32_t* cells, const int32_t* p2v_perm,
                                      const int32_t* p2v_off, const int32_t* p2v_idx, const int32_t* v2p_idx, const float* w,
                                      float* gfeats, int64_t N, int64_t V, int32_t C, hipStream_t stream) {
  LT_REQUIRE(N >= 0 && V >= 0 && C >= 4 && C % 4 == 0, LT_ERR_ARG, "lt_devoxelize_cell_bwd: bad sizes (C %% 4 == 0 required)");
  if (V == 0) return LT_OK;
  LT_REQUIRE(gout && cells && p2v_perm && p2v_off && p2v_idx && v2p_idx && w && gfeats, LT_ERR_ARG,
             "lt_devoxelize_cell_bwd: null pointer");
  hipLaunchKernelGGL(k_devox_cell_bwd, dim3(lt_grid(V * (C / 4), 256)), dim3(256), 0, stream, gout, (const int2*)cells,
                     p2v_perm, w, gfeats, V, C / 4);
  LT_LAUNCH_CHECK();
  if (N > 0) {
    hipLaunchKernelGGL(k_devox_unmapped, dim3(lt_grid(N, 256)), dim3(256), 0, stream, gout, p2v_idx, v2p_idx, w, gfeats, N, V,
                       C);
    LT_LAUNCH_CHECK();
  }
  return LT_OK;
}
