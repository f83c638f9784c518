// BatchNorm with batch statistics + ReLU + residual add over sparse-tensor features [V,C] (fp32).
//
// Replaces spnn.BatchNorm / spnn.ReLU / the skip add of ResidualBlock -- reference
// latte/models/basic_blocks.py:22-23,40-41,56-57,64,69,72,75; under TTA every BN uses batch statistics
// (latte/common/utils/ema_util.py:26-29) and gamma/beta are the only trainable 3-D parameters (:16-24).
//
// HBM-bound.  Column reductions: a block owns a slab of rows, a thread owns one float4 column group (16-B
// loads, a wave reads whole rows), partial sums per block go to a workspace, the finalising kernel combines
// them in fp64.  Sums are shifted by row 0 so that var = E[d^2]-E[d]^2 does not cancel when |mean| >> std.
#include "common.h"
#include "../../include/latte_hip.h"

#define BN_BLOCK 256
#define BN_MAX_BLOCKS 1024

static int bn_blocks(int64_t V, int C) {
  const int cg = C / 4;
  const int rpi = BN_BLOCK / cg > 0 ? BN_BLOCK / cg : 1;
  int64_t nb = (V + (int64_t)rpi * 8 - 1) / ((int64_t)rpi * 8);
  if (nb < 1) nb = 1;
  if (nb > BN_MAX_BLOCKS) nb = BN_MAX_BLOCKS;
  return (int)nb;
}

extern "C" int64_t lt_bn_workspace_bytes(int64_t V, int32_t C) {
  (void)V;
  return (int64_t)BN_MAX_BLOCKS * 2 * C * 4 + 256;
}

struct StatsOp {
  const float* x;
  __device__ void operator()(int64_t row, int cg, int C, float4& a, float4& b) const {
    const float4 v = reinterpret_cast<const float4*>(x + row * C)[cg];
    const float4 s = reinterpret_cast<const float4*>(x)[cg];  // shift = row 0
    const float dx = v.x - s.x, dy = v.y - s.y, dz = v.z - s.z, dw = v.w - s.w;
    a.x += dx; a.y += dy; a.z += dz; a.w += dw;
    b.x += dx * dx; b.y += dy * dy; b.z += dz * dz; b.w += dw * dw;
  }
};

// the BatchNorm output of one element, the same expression (and rounding) wherever it is evaluated: the forward apply,
// and the backward kernels that rebuild the ReLU mask from x instead of reading y (relu == 2: no residual was added)
__device__ __forceinline__ float bn_affine(float v, float m, float vr, float g, float b, float eps) {
  return __fmaf_rn((v - m) * (1.0f / sqrtf(vr + eps)), g, b);
}

struct BwdOp {
  const float* dy;
  const float* x;
  const float* y;  // ReLU mask from y; null: no ReLU, or (gamma != null) mask rebuilt from x
  const float* mean;
  const float* var;
  float eps;
  const float* gamma;  // non-null: rebuild the ReLU mask  gamma * xhat + beta > 0
  const float* beta;
  __device__ void operator()(int64_t row, int cg, int C, float4& a, float4& b) const {
    float4 g = reinterpret_cast<const float4*>(dy + row * C)[cg];
    if (y) {
      const float4 yy = reinterpret_cast<const float4*>(y + row * C)[cg];
      g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
      g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
    }
    const float4 v = reinterpret_cast<const float4*>(x + row * C)[cg];
    const float4 m = reinterpret_cast<const float4*>(mean)[cg];
    const float4 vr = reinterpret_cast<const float4*>(var)[cg];
    if (gamma) {
      const float4 gm = reinterpret_cast<const float4*>(gamma)[cg];
      const float4 bt = reinterpret_cast<const float4*>(beta)[cg];
      g.x = bn_affine(v.x, m.x, vr.x, gm.x, bt.x, eps) > 0.f ? g.x : 0.f;
      g.y = bn_affine(v.y, m.y, vr.y, gm.y, bt.y, eps) > 0.f ? g.y : 0.f;
      g.z = bn_affine(v.z, m.z, vr.z, gm.z, bt.z, eps) > 0.f ? g.z : 0.f;
      g.w = bn_affine(v.w, m.w, vr.w, gm.w, bt.w, eps) > 0.f ? g.w : 0.f;
    }
    a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
    b.x += g.x * ((v.x - m.x) * (1.0f / sqrtf(vr.x + eps)));
    b.y += g.y * ((v.y - m.y) * (1.0f / sqrtf(vr.y + eps)));
    b.z += g.z * ((v.z - m.z) * (1.0f / sqrtf(vr.z + eps)));
    b.w += g.w * ((v.w - m.w) * (1.0f / sqrtf(vr.w + eps)));
  }
};

// partial[(blk*2 + which)*C + c]
template <class Op>
__global__ __launch_bounds__(BN_BLOCK) void k_col_partial(Op op, int64_t V, int C, float* __restrict__ partial) {
  extern __shared__ float s_red[];  // [rpi][2][C]
  const int cg_n = C / 4;
  const int rpi = BN_BLOCK / cg_n > 0 ? BN_BLOCK / cg_n : 1;
  const int tid = threadIdx.x;
  const int rr = tid / cg_n, cg = tid % cg_n;
  const bool live = rr < rpi && cg_n <= BN_BLOCK;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (live) {
    const int64_t rows_per_block = (V + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > V) r1 = V;
    for (int64_t row = r0 + rr; row < r1; row += rpi) op(row, cg, C, a, b);
    float* d = s_red + (int64_t)rr * 2 * C;
    reinterpret_cast<float4*>(d)[cg] = a;
    reinterpret_cast<float4*>(d + C)[cg] = b;
  }
  __syncthreads();
  for (int e = tid; e < 2 * C; e += BN_BLOCK) {
    float s = 0.f;
    for (int i = 0; i < rpi; ++i) s += s_red[(int64_t)i * 2 * C + e];
    partial[(int64_t)blockIdx.x * 2 * C + e] = s;
  }
}

// one wave per channel: lane l sums partials l, l+64, ... in fp64, then a wave reduction (fixed order => deterministic)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ __launch_bounds__(64) void k_bn_finalize(const float* __restrict__ partial, int nb, const float* __restrict__ x,
                                                    int64_t V, int C, float* __restrict__ mean, float* __restrict__ var) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int b = lane; b < nb; b += 64) {
    s1 += (double)partial[(int64_t)b * 2 * C + c];
    s2 += (double)partial[(int64_t)b * 2 * C + C + c];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    const double m = s1 / (double)V;
    double v = s2 / (double)V - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)((double)x[c] + m);
    var[c] = (float)v;
  }
}

__global__ __launch_bounds__(64) void k_sum_partials(const float* __restrict__ partial, int nb, int C,
                                                     float* __restrict__ out_a, float* __restrict__ out_b) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int b = lane; b < nb; b += 64) {
    s1 += (double)partial[(int64_t)b * 2 * C + c];
    s2 += (double)partial[(int64_t)b * 2 * C + C + c];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    out_a[c] = (float)s1;
    out_b[c] = (float)s2;
  }
}

// mean / biased variance from the conv epilogue's per-workgroup column sums [nb][2][C].  Block (g, slice): 16 channels,
// 64 row lanes, 64-B coalesced reads, fp64, fixed summation order.  FINAL: one slice, writes mean / var; otherwise the
// slice sums go to stage[slice][2][C] (fp64) and a second pass over the slices finishes (nb > 2048 only).
template <typename T, bool FINAL>
__global__ __launch_bounds__(1024) void k_stats_reduce(const T* __restrict__ in, int64_t nb, int64_t V, int C,
                                                       double* __restrict__ stage, float* __restrict__ mean,
                                                       float* __restrict__ var) {
  __shared__ double red[2][64][16];
  const int t = threadIdx.x, cl = t & 15, rl = t >> 4;
  const int g = blockIdx.x, sl = blockIdx.y, S = gridDim.y;
  const int c = 16 * g + cl;
  const int64_t per = (nb + S - 1) / S;
  const int64_t b0 = sl * per, b1 = (b0 + per < nb) ? b0 + per : nb;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
  for (int64_t b = b0 + rl; b < b1; b += 64) {
    s1 += (double)in[b * 2 * C + c];
    s2 += (double)in[b * 2 * C + C + c];
  }
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  // 2 x 16 sums of 64 values: 4 threads per sum, then a fixed-order add of the four
  if (t < 128) {
    const int which = t >> 6, part = (t >> 4) & 3;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += red[which][16 * part + i][cl];
    red[which][16 * part][cl] = acc;
  }
  __syncthreads();
  if (t < 16) {
    const double a1 = ((red[0][0][cl] + red[0][16][cl]) + red[0][32][cl]) + red[0][48][cl];
    const double a2 = ((red[1][0][cl] + red[1][16][cl]) + red[1][32][cl]) + red[1][48][cl];
    if (FINAL) {
      const double m = a1 / (double)V;
      double v = a2 / (double)V - m * m;
      if (v < 0.0) v = 0.0;
      mean[c] = (float)m;
      var[c] = (float)v;
    } else {
      stage[((int64_t)sl * 2) * C + c] = a1;
      stage[((int64_t)sl * 2 + 1) * C + c] = a2;
    }
  }
}

static int check_vc(const char* who, int64_t V, int32_t C) {
  LT_REQUIRE(V >= 1, LT_ERR_ARG, "%s: V=%lld must be >= 1", who, (long long)V);
  LT_REQUIRE(C >= 4 && C % 4 == 0 && C <= 1024, LT_ERR_ARG, "%s: C=%d must be a multiple of 4 in [4,1024]", who, C);
  return LT_OK;
}

extern "C" int lt_bn_stats(const float* x, int64_t V, int32_t C, void* ws, int64_t ws_bytes, float* mean, float* var,
                           hipStream_t stream) {
  int rc = check_vc("lt_bn_stats", V, C);
  if (rc) return rc;
  LT_REQUIRE(x && ws && mean && var, LT_ERR_ARG, "lt_bn_stats: null pointer");
  LT_REQUIRE(ws_bytes >= lt_bn_workspace_bytes(V, C), LT_ERR_WORKSPACE, "lt_bn_stats: workspace too small");
  const int nb = bn_blocks(V, C);
  const int cg = C / 4, rpi = BN_BLOCK / cg > 0 ? BN_BLOCK / cg : 1;
  StatsOp op{x};
  hipLaunchKernelGGL(k_col_partial<StatsOp>, dim3(nb), dim3(BN_BLOCK), (size_t)rpi * 2 * C * 4, stream, op, V, C,
                     (float*)ws);
  LT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bn_finalize, dim3(C), dim3(64), 0, stream, (const float*)ws, nb, x, V, C, mean, var);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

extern "C" int lt_bn_finalize_sums(void* stats_ws, int64_t V, int32_t C, float* mean, float* var, hipStream_t stream) {
  int rc = check_vc("lt_bn_finalize_sums", V, C);
  if (rc) return rc;
  LT_REQUIRE(C % 16 == 0, LT_ERR_ARG, "lt_bn_finalize_sums: C=%d must be a multiple of 16", C);
  LT_REQUIRE(stats_ws && mean && var, LT_ERR_ARG, "lt_bn_finalize_sums: null pointer");
  const int64_t nb = lt_spconv_stat_blocks(V, C);
  LtStatsView sv = lt_stats_view(stats_ws, nb, C);
  if (nb <= 2048) {
    hipLaunchKernelGGL((k_stats_reduce<float, true>), dim3(C / 16, 1), dim3(1024), 0, stream, sv.partial, nb, V, C, sv.stage,
                       mean, var);
  } else {
    hipLaunchKernelGGL((k_stats_reduce<float, false>), dim3(C / 16, LT_STAT_SLICES), dim3(1024), 0, stream, sv.partial, nb, V,
                       C, sv.stage, mean, var);
    hipLaunchKernelGGL((k_stats_reduce<double, true>), dim3(C / 16, 1), dim3(1024), 0, stream, sv.stage,
                       (int64_t)LT_STAT_SLICES, V, C, sv.stage, mean, var);
  }
  LT_LAUNCH_CHECK();
  return LT_OK;
}

__global__ void k_bn_apply(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ var,
                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                           const float* __restrict__ res, int relu, float* __restrict__ y, int64_t n4, int cg_n) {
  // column group of element e without a 64-bit modulo per element: one modulo per thread, then a wrapped increment
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int cstep = (int)(stride % cg_n);
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int cg = (int)(e % cg_n);
  for (; e < n4; e += stride, cg = cg + cstep >= cg_n ? cg + cstep - cg_n : cg + cstep) {
    const float4 v = reinterpret_cast<const float4*>(x)[e];
    const float4 m = reinterpret_cast<const float4*>(mean)[cg];
    const float4 vr = reinterpret_cast<const float4*>(var)[cg];
    const float4 g = reinterpret_cast<const float4*>(gamma)[cg];
    const float4 b = reinterpret_cast<const float4*>(beta)[cg];
    float4 o;
    o.x = bn_affine(v.x, m.x, vr.x, g.x, b.x, eps);
    o.y = bn_affine(v.y, m.y, vr.y, g.y, b.y, eps);
    o.z = bn_affine(v.z, m.z, vr.z, g.z, b.z, eps);
    o.w = bn_affine(v.w, m.w, vr.w, g.w, b.w, eps);
    if (res) {
      const float4 rr = reinterpret_cast<const float4*>(res)[e];
      o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
    }
    if (relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    reinterpret_cast<float4*>(y)[e] = o;
  }
}

extern "C" int lt_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                           float eps, const float* residual, int32_t relu, float* y, int64_t V, int32_t C,
                           hipStream_t stream) {
  int rc = check_vc("lt_bn_apply", V, C);
  if (rc) return rc;
  LT_REQUIRE(x && mean && var && gamma && beta && y, LT_ERR_ARG, "lt_bn_apply: null pointer");
  const int64_t n4 = V * (C / 4);
  hipLaunchKernelGGL(k_bn_apply, dim3(lt_grid(n4, 256)), dim3(256), 0, stream, x, mean, var, gamma, beta, eps, residual,
                     relu, y, n4, C / 4);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

extern "C" int lt_bn_bwd_reduce(const float* dy, const float* x, const float* y, const float* mean, const float* var,
                                const float* gamma, const float* beta, float eps, int32_t relu, int64_t V, int32_t C,
                                void* ws, int64_t ws_bytes, float* sum_g, float* sum_gx, hipStream_t stream) {
  int rc = check_vc("lt_bn_bwd_reduce", V, C);
  if (rc) return rc;
  LT_REQUIRE(relu >= 0 && relu <= 2, LT_ERR_ARG, "lt_bn_bwd_reduce: relu=%d not in {0,1,2}", relu);
  LT_REQUIRE(dy && x && mean && var && ws && sum_g && sum_gx && (relu != 1 || y) && (relu != 2 || (gamma && beta)),
             LT_ERR_ARG, "lt_bn_bwd_reduce: null pointer");
  LT_REQUIRE(ws_bytes >= lt_bn_workspace_bytes(V, C), LT_ERR_WORKSPACE, "lt_bn_bwd_reduce: workspace too small");
  const int nb = bn_blocks(V, C);
  const int cg = C / 4, rpi = BN_BLOCK / cg > 0 ? BN_BLOCK / cg : 1;
  BwdOp op{dy, x, relu == 1 ? y : nullptr, mean, var, eps, relu == 2 ? gamma : nullptr, relu == 2 ? beta : nullptr};
  hipLaunchKernelGGL(k_col_partial<BwdOp>, dim3(nb), dim3(BN_BLOCK), (size_t)rpi * 2 * C * 4, stream, op, V, C,
                     (float*)ws);
  LT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_partials, dim3(C), dim3(64), 0, stream, (const float*)ws, nb, C, sum_g, sum_gx);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

__global__ void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                               const float* __restrict__ mean, const float* __restrict__ var,
                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                               const float* __restrict__ sum_g, const float* __restrict__ sum_gx,
                               float* __restrict__ dx, float* __restrict__ dres, int64_t n4, int cg_n, float inv_v) {
  // beta != null: the ReLU mask is rebuilt from x (y is null then)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int cstep = (int)(stride % cg_n);
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int cg = (int)(e % cg_n);
  for (; e < n4; e += stride, cg = cg + cstep >= cg_n ? cg + cstep - cg_n : cg + cstep) {
    float4 g = reinterpret_cast<const float4*>(dy)[e];
    if (y) {
      const float4 yy = reinterpret_cast<const float4*>(y)[e];
      g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
      g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
    }
    if (beta) {
      const float4 v = reinterpret_cast<const float4*>(x)[e];
      const float4 m = reinterpret_cast<const float4*>(mean)[cg];
      const float4 vr = reinterpret_cast<const float4*>(var)[cg];
      const float4 gm = reinterpret_cast<const float4*>(gamma)[cg];
      const float4 bt = reinterpret_cast<const float4*>(beta)[cg];
      g.x = bn_affine(v.x, m.x, vr.x, gm.x, bt.x, eps) > 0.f ? g.x : 0.f;
      g.y = bn_affine(v.y, m.y, vr.y, gm.y, bt.y, eps) > 0.f ? g.y : 0.f;
      g.z = bn_affine(v.z, m.z, vr.z, gm.z, bt.z, eps) > 0.f ? g.z : 0.f;
      g.w = bn_affine(v.w, m.w, vr.w, gm.w, bt.w, eps) > 0.f ? g.w : 0.f;
    }
    if (dres) reinterpret_cast<float4*>(dres)[e] = g;
    if (dx) {
      const float4 v = reinterpret_cast<const float4*>(x)[e];
      const float4 m = reinterpret_cast<const float4*>(mean)[cg];
      const float4 vr = reinterpret_cast<const float4*>(var)[cg];
      const float4 gm = reinterpret_cast<const float4*>(gamma)[cg];
      const float4 sg = reinterpret_cast<const float4*>(sum_g)[cg];
      const float4 sx = reinterpret_cast<const float4*>(sum_gx)[cg];
      float4 o;
      float rs;
      rs = 1.0f / sqrtf(vr.x + eps); o.x = gm.x * rs * (g.x - sg.x * inv_v - (v.x - m.x) * rs * sx.x * inv_v);
      rs = 1.0f / sqrtf(vr.y + eps); o.y = gm.y * rs * (g.y - sg.y * inv_v - (v.y - m.y) * rs * sx.y * inv_v);
      rs = 1.0f / sqrtf(vr.z + eps); o.z = gm.z * rs * (g.z - sg.z * inv_v - (v.z - m.z) * rs * sx.z * inv_v);
      rs = 1.0f / sqrtf(vr.w + eps); o.w = gm.w * rs * (g.w - sg.w * inv_v - (v.w - m.w) * rs * sx.w * inv_v);
      reinterpret_cast<float4*>(dx)[e] = o;
    }
  }
}

extern "C" int lt_bn_bwd_apply(const float* dy, const float* x, const float* y, const float* mean, const float* var,
                               const float* gamma, const float* beta, float eps, int32_t relu, const float* sum_g,
                               const float* sum_gx, float* dx, float* dres, int64_t V, int32_t C, hipStream_t stream) {
  int rc = check_vc("lt_bn_bwd_apply", V, C);
  if (rc) return rc;
  LT_REQUIRE(relu >= 0 && relu <= 2, LT_ERR_ARG, "lt_bn_bwd_apply: relu=%d not in {0,1,2}", relu);
  LT_REQUIRE(dy && x && mean && var && gamma && sum_g && sum_gx && (relu != 1 || y) && (relu != 2 || beta), LT_ERR_ARG,
             "lt_bn_bwd_apply: null pointer");
  const int64_t n4 = V * (C / 4);
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(lt_grid(n4, 256)), dim3(256), 0, stream, dy, x, relu == 1 ? y : nullptr, mean,
                     var, gamma, relu == 2 ? beta : nullptr, eps, sum_g, sum_gx, dx, dres, n4, C / 4, 1.0f / (float)V);
  LT_LAUNCH_CHECK();
  return LT_OK;
}
