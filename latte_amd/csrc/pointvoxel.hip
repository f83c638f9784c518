// Point <-> voxel bridges of SPVCNN (torchsparse v1.4.0 F.spvoxelize / F.calc_ti_weights / F.spdevoxelize).
// Reference call sites: latte/models/torchsparse_utils.py:20,23,54 (voxelize), :75 (weights), :81,93 (devoxelize).
// All HBM-bound: one lane per 16-B channel group, rows contiguous across lanes.
#include "common.h"
#include "../../include/latte_hip.h"

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
