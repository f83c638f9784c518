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
#include <mutex>
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

// A column-reduction op: cols() = the per-column-group constants of a thread (registers), load() = the raw 16-B loads
// of one row, acc() = the accumulation.  k_col_partial keeps four rows of loads in flight and accumulates in row order.
struct StatsOp {
  const float* x;
  struct Cols { float4 s; };
  struct Ld { float4 v; };
  static constexpr int kLoads = 1;
  __device__ __forceinline__ Cols cols(int cg) const { return Cols{reinterpret_cast<const float4*>(x)[cg]}; }  // shift = row 0
  __device__ __forceinline__ Ld load(int64_t row, int cg, int C) const {
    return Ld{reinterpret_cast<const float4*>(x + row * C)[cg]};
  }
  __device__ __forceinline__ void acc(const Ld& l, const Cols& c, float4& a, float4& b) const {
    const float dx = l.v.x - c.s.x, dy = l.v.y - c.s.y, dz = l.v.z - c.s.z, dw = l.v.w - c.s.w;
    a.x += dx; a.y += dy; a.z += dz; a.w += dw;
    b.x += dx * dx; b.y += dy * dy; b.z += dz * dz; b.w += dw * dw;
  }
};

// the BatchNorm output of one element, the same expression (and rounding) wherever it is evaluated: the forward apply,
// and the backward kernels that rebuild the ReLU mask from x instead of reading y (relu == 2: no residual was added)
__device__ __forceinline__ float bn_affine(float v, float m, float vr, float g, float b, float eps) {
  return __fmaf_rn((v - m) * (1.0f / sqrtf(vr + eps)), g, b);
}

// MODE 0: no ReLU; 1: ReLU mask from y; 2: mask rebuilt from x (gamma * xhat + beta > 0; no residual was added)
template <int MODE>
struct BwdOp {
  const float* dy;
  const float* x;
  const float* y;
  const float* mean;
  const float* var;
  float eps;
  const float* gamma;
  const float* beta;
  struct Cols { float4 m, vr, gm, bt; };
  struct Ld { float4 g, v, yy; };
  static constexpr int kLoads = MODE == 1 ? 3 : 2;
  __device__ __forceinline__ Cols cols(int cg) const {
    Cols c;
    c.m = reinterpret_cast<const float4*>(mean)[cg];
    c.vr = reinterpret_cast<const float4*>(var)[cg];
    c.gm = c.bt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 2) {
      c.gm = reinterpret_cast<const float4*>(gamma)[cg];
      c.bt = reinterpret_cast<const float4*>(beta)[cg];
    }
    return c;
  }
  __device__ __forceinline__ Ld load(int64_t row, int cg, int C) const {
    Ld l;
    l.g = reinterpret_cast<const float4*>(dy + row * C)[cg];
    l.v = reinterpret_cast<const float4*>(x + row * C)[cg];
    l.yy = MODE == 1 ? reinterpret_cast<const float4*>(y + row * C)[cg] : make_float4(0.f, 0.f, 0.f, 0.f);
    return l;
  }
  __device__ __forceinline__ void acc(const Ld& l, const Cols& c, float4& a, float4& b) const {
    float4 g = l.g;
    const float4 v = l.v;
    if (MODE == 1) {
      g.x = l.yy.x > 0.f ? g.x : 0.f; g.y = l.yy.y > 0.f ? g.y : 0.f;
      g.z = l.yy.z > 0.f ? g.z : 0.f; g.w = l.yy.w > 0.f ? g.w : 0.f;
    }
    if (MODE == 2) {
      g.x = bn_affine(v.x, c.m.x, c.vr.x, c.gm.x, c.bt.x, eps) > 0.f ? g.x : 0.f;
      g.y = bn_affine(v.y, c.m.y, c.vr.y, c.gm.y, c.bt.y, eps) > 0.f ? g.y : 0.f;
      g.z = bn_affine(v.z, c.m.z, c.vr.z, c.gm.z, c.bt.z, eps) > 0.f ? g.z : 0.f;
      g.w = bn_affine(v.w, c.m.w, c.vr.w, c.gm.w, c.bt.w, eps) > 0.f ? g.w : 0.f;
    }
    a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
    b.x += g.x * ((v.x - c.m.x) * (1.0f / sqrtf(c.vr.x + eps)));
    b.y += g.y * ((v.y - c.m.y) * (1.0f / sqrtf(c.vr.y + eps)));
    b.z += g.z * ((v.z - c.m.z) * (1.0f / sqrtf(c.vr.z + eps)));
    b.w += g.w * ((v.w - c.m.w) * (1.0f / sqrtf(c.vr.w + eps)));
  }
};

// partial[(blk*2 + which)*C + c]
template <class Op>
__global__ __launch_bounds__(BN_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_col_partial(Op op, int64_t V, int C, float* __restrict__ partial) {
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
    const typename Op::Cols c = op.cols(cg);
    int64_t row = r0 + rr;
    for (; row + 3 * (int64_t)rpi < r1; row += 4 * (int64_t)rpi) {
      typename Op::Ld l[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) l[i] = op.load(row + i * (int64_t)rpi, cg, C);
      __builtin_amdgcn_sched_barrier(0);  // all loads of the four rows issue before the first accumulate
#pragma unroll
      for (int i = 0; i < 4; ++i) op.acc(l[i], c, a, b);
    }
    for (; row < r1; row += rpi) {
      const typename Op::Ld l = op.load(row, cg, C);
      op.acc(l, c, a, b);
    }
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

// ---- the entry point that allocates nothing (task item 7): lt_bn_stats with the per-block partial sums in a file-scope
// device array instead of a caller workspace.  Each GPU holds its own copy of the array at its own address: the address
// is looked up once per device (hipGetSymbolAddress with that device current) and only that pointer is ever handed to a
// kernel.  Every partial the finalising kernel reads is written by the launch before it, so nothing needs clearing.
#define LT_BN_STATIC_MAX_C 256
#define LT_BN_STATIC_MAX_DEVICES 64
static __device__ __align__(256) float scratch[(size_t)BN_MAX_BLOCKS * 2 * LT_BN_STATIC_MAX_C];
static float* g_scratch_ptr[LT_BN_STATIC_MAX_DEVICES];
static std::mutex g_scratch_mu;

// RAII: make `dev` the current device for the scope (lookup + launches), restore the caller's on exit
struct LtDeviceScope {
  int prev = -1, dev = -1;
  bool switched = false;
  int enter(int d) {
    dev = d;
    LT_CHECK_HIP(hipGetDevice(&prev));
    if (prev != d) {
      LT_CHECK_HIP(hipSetDevice(d));
      switched = true;
    }
    return LT_OK;
  }
  ~LtDeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

// device that `stream` belongs to (the null stream: the current device)
static int bn_stream_device(hipStream_t stream, int* dev) {
  LT_CHECK_HIP(hipGetDevice(dev));
  if (stream) LT_CHECK_HIP(hipStreamGetDevice(stream, dev));
  LT_REQUIRE(*dev >= 0 && *dev < LT_BN_STATIC_MAX_DEVICES, LT_ERR_RANGE, "lt_bn_stats_static: device %d out of range", *dev);
  return LT_OK;
}

// scratch pointer of device `dev`, looked up once with THAT device current (each GPU holds its own copy of the array)
static int bn_static_scratch(int dev, float** out) {
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    LT_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    LT_REQUIRE(p != nullptr, LT_ERR_HIP, "lt_bn_stats_static: no device address for the scratch array on device %d", dev);
    g_scratch_ptr[dev] = (float*)p;
  }
  *out = g_scratch_ptr[dev];
  return LT_OK;
}

extern "C" int lt_bn_stats_static(const float* x, int64_t V, int32_t C, float* mean, float* var, hipStream_t stream) {
  int rc = check_vc("lt_bn_stats_static", V, C);
  if (rc) return rc;
  LT_REQUIRE(C <= LT_BN_STATIC_MAX_C, LT_ERR_RANGE,
             "lt_bn_stats_static: C=%d exceeds the built-in scratch (max %d channels); use lt_bn_stats with a workspace", C,
             LT_BN_STATIC_MAX_C);
  LT_REQUIRE(x && mean && var, LT_ERR_ARG, "lt_bn_stats_static: null pointer");
  // the scratch array of the GPU the STREAM lives on, whatever device is current in the calling thread: lookup and
  // launches run with that device current (restored on return)
  int dev = -1;
  rc = bn_stream_device(stream, &dev);
  if (rc) return rc;
  LtDeviceScope scope;
  rc = scope.enter(dev);
  if (rc) return rc;
  float* ws = nullptr;
  rc = bn_static_scratch(dev, &ws);
  if (rc) return rc;
  const int nb = bn_blocks(V, C);
  const int cg = C / 4, rpi = BN_BLOCK / cg > 0 ? BN_BLOCK / cg : 1;
  StatsOp op{x};
  hipLaunchKernelGGL(k_col_partial<StatsOp>, dim3(nb), dim3(BN_BLOCK), (size_t)rpi * 2 * C * 4, stream, op, V, C, ws);
  LT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bn_finalize, dim3(C), dim3(64), 0, stream, (const float*)ws, nb, x, V, C, mean, var);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

extern "C" int lt_bn_finalize_sums(void* stats_ws, int64_t V, int32_t C, float* mean, float* var, hipStream_t stream) {
  return lt_bn_finalize_sums_nb(stats_ws, lt_spconv_stat_blocks(V, C), V, C, mean, var, stream);
}

int lt_bn_finalize_sums_nb(void* stats_ws, int64_t nb, int64_t V, int32_t C, float* mean, float* var, hipStream_t stream) {
  int rc = check_vc("lt_bn_finalize_sums", V, C);
  if (rc) return rc;
  LT_REQUIRE(C % 16 == 0, LT_ERR_ARG, "lt_bn_finalize_sums: C=%d must be a multiple of 16", C);
  LT_REQUIRE(stats_ws && mean && var && nb >= 1, LT_ERR_ARG, "lt_bn_finalize_sums: null pointer");
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

// grid of the two elementwise kernels: the launch's thread count is a multiple of the number of column groups, so a
// thread keeps ONE column group for all its elements (its mean / var / gamma / beta sit in registers), and about four
// float4 elements per thread so that four loads are in flight
static int bn_ew_grid(int64_t n4, int cg_n) {
  int64_t g = (n4 + 1023) / 1024;
  if (g > 2048) g = 2048;
  if (g < 512) {
    g = (n4 + 255) / 256;
    if (g > 512) g = 512;
  }
  if (g < 1) g = 1;
  int a = cg_n, b = 256;
  while (b) { const int t = a % b; a = b; b = t; }
  const int q = cg_n / a;  // smallest block count whose thread count is a multiple of cg_n
  return (int)((g + q - 1) / q * q);
}

__device__ __forceinline__ float4 bn_apply4(float4 v, float4 m, float4 vr, float4 g, float4 b, float eps) {
  float4 o;
  o.x = bn_affine(v.x, m.x, vr.x, g.x, b.x, eps);
  o.y = bn_affine(v.y, m.y, vr.y, g.y, b.y, eps);
  o.z = bn_affine(v.z, m.z, vr.z, g.z, b.z, eps);
  o.w = bn_affine(v.w, m.w, vr.w, g.w, b.w, eps);
  return o;
}

template <bool RES>
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, const float* __restrict__ mean,
                                                  const float* __restrict__ var, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float eps,
                                                  const float* __restrict__ res, int relu, float* __restrict__ y,
                                                  int64_t n4, int cg_n) {
  const int64_t stride = (int64_t)gridDim.x * 256;  // a multiple of cg_n (bn_ew_grid)
  int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x;
  const int cg = (int)(e % cg_n);
  const float4 m = reinterpret_cast<const float4*>(mean)[cg];
  const float4 vr = reinterpret_cast<const float4*>(var)[cg];
  const float4 g = reinterpret_cast<const float4*>(gamma)[cg];
  const float4 b = reinterpret_cast<const float4*>(beta)[cg];
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* r4 = reinterpret_cast<const float4*>(res);
  float4* y4 = reinterpret_cast<float4*>(y);
  const bool post = (relu & 2) != 0;  // the residual is added after the ReLU
  relu &= 1;
  for (; e + 3 * stride < n4; e += 4 * stride) {
    float4 v[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = x4[e + i * stride];
    if (RES) {
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = r4[e + i * stride];
    }
    __builtin_amdgcn_sched_barrier(0);  // all loads issue before the first use
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 o = bn_apply4(v[i], m, vr, g, b, eps);
      if (RES && !post) { o.x += r[i].x; o.y += r[i].y; o.z += r[i].z; o.w += r[i].w; }
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (RES && post) { o.x += r[i].x; o.y += r[i].y; o.z += r[i].z; o.w += r[i].w; }
      y4[e + i * stride] = o;
    }
  }
  for (; e < n4; e += stride) {
    float4 o = bn_apply4(x4[e], m, vr, g, b, eps);
    float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (RES) rr = r4[e];
    if (RES && !post) { o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    if (RES && post) { o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
    y4[e] = o;
  }
}

extern "C" int lt_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                           float eps, const float* residual, int32_t relu, float* y, int64_t V, int32_t C,
                           hipStream_t stream) {
  int rc = check_vc("lt_bn_apply", V, C);
  if (rc) return rc;
  LT_REQUIRE(x && mean && var && gamma && beta && y, LT_ERR_ARG, "lt_bn_apply: null pointer");
  LT_REQUIRE(relu == 0 || relu == 1 || (relu == 3 && residual), LT_ERR_ARG,
             "lt_bn_apply: relu=%d not in {0,1,3} (3 needs a residual)", relu);
  const int64_t n4 = V * (C / 4);
  const dim3 grid(bn_ew_grid(n4, C / 4));
  if (residual)
    hipLaunchKernelGGL(k_bn_apply<true>, grid, dim3(256), 0, stream, x, mean, var, gamma, beta, eps, residual, relu, y, n4,
                       C / 4);
  else
    hipLaunchKernelGGL(k_bn_apply<false>, grid, dim3(256), 0, stream, x, mean, var, gamma, beta, eps, residual, relu, y,
                       n4, C / 4);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// y[p,:] = relu(bn(x[p,:])) + sum_{k<8} w[p,k] * vox[idx[p,k],:]  -- the point branch of SPVCNN with the devoxelized
// features (latte/models/spvcnn.py:141-142,152-153,163-164: z.F = voxel_to_point(...).F + point_transforms(z.F)) summed in
// ON THE FLY: the [points, C] devoxelized tensor is neither written nor read back (2 x 4NC bytes per pass; the voxel rows
// come out of the caches).  Same arithmetic as lt_devoxelize_fwd followed by lt_bn_apply(relu = 3): the trilinear sum in
// corner order with separate multiply and add, then one add -- bit-equal to the two-kernel path.
__global__ __launch_bounds__(256) void k_bn_apply_devox(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ var, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ vox, const int* __restrict__ idx,
                                                        const float* __restrict__ w, float* __restrict__ y, int64_t n4,
                                                        int cg_n, int64_t V) {
#pragma clang fp contract(off)
  const int64_t stride = (int64_t)gridDim.x * 256;  // a multiple of cg_n (bn_ew_grid)
  int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x;
  const int cg = (int)(e % cg_n);
  const float4 m = reinterpret_cast<const float4*>(mean)[cg];
  const float4 vr = reinterpret_cast<const float4*>(var)[cg];
  const float4 g = reinterpret_cast<const float4*>(gamma)[cg];
  const float4 b = reinterpret_cast<const float4*>(beta)[cg];
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4* y4 = reinterpret_cast<float4*>(y);
  for (; e < n4; e += 2 * stride) {
    const int64_t e1 = e + stride;
    const bool two = e1 < n4;
    const int64_t p0 = e / cg_n, p1 = two ? e1 / cg_n : p0;
    const float4 v0 = x4[e];
    const float4 v1 = two ? x4[e1] : v0;
    // corner rows and weights of the two points (every lane of a row reads the same 64 B: one cache line per point)
    const int4 ia0 = reinterpret_cast<const int4*>(idx + p0 * 8)[0], ib0 = reinterpret_cast<const int4*>(idx + p0 * 8)[1];
    const float4 wa0 = reinterpret_cast<const float4*>(w + p0 * 8)[0], wb0 = reinterpret_cast<const float4*>(w + p0 * 8)[1];
    const int4 ia1 = reinterpret_cast<const int4*>(idx + p1 * 8)[0], ib1 = reinterpret_cast<const int4*>(idx + p1 * 8)[1];
    const float4 wa1 = reinterpret_cast<const float4*>(w + p1 * 8)[0], wb1 = reinterpret_cast<const float4*>(w + p1 * 8)[1];
    const int i0[8] = {ia0.x, ia0.y, ia0.z, ia0.w, ib0.x, ib0.y, ib0.z, ib0.w};
    const float w0[8] = {wa0.x, wa0.y, wa0.z, wa0.w, wb0.x, wb0.y, wb0.z, wb0.w};
    const int i1[8] = {ia1.x, ia1.y, ia1.z, ia1.w, ib1.x, ib1.y, ib1.z, ib1.w};
    const float w1[8] = {wa1.x, wa1.y, wa1.z, wa1.w, wb1.x, wb1.y, wb1.z, wb1.w};
    float4 f0[8], f1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f0[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      f1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i0[k] >= 0 && i0[k] < V) f0[k] = reinterpret_cast<const float4*>(vox + (int64_t)i0[k] * cg_n * 4)[cg];
      if (two && i1[k] >= 0 && i1[k] < V) f1[k] = reinterpret_cast<const float4*>(vox + (int64_t)i1[k] * cg_n * 4)[cg];
    }
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a0.x += w0[k] * f0[k].x; a0.y += w0[k] * f0[k].y; a0.z += w0[k] * f0[k].z; a0.w += w0[k] * f0[k].w;
      a1.x += w1[k] * f1[k].x; a1.y += w1[k] * f1[k].y; a1.z += w1[k] * f1[k].z; a1.w += w1[k] * f1[k].w;
    }
    float4 o = bn_apply4(v0, m, vr, g, b, eps);
    o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    o.x += a0.x; o.y += a0.y; o.z += a0.z; o.w += a0.w;
    y4[e] = o;
    if (two) {
      float4 q = bn_apply4(v1, m, vr, g, b, eps);
      q.x = fmaxf(q.x, 0.f); q.y = fmaxf(q.y, 0.f); q.z = fmaxf(q.z, 0.f); q.w = fmaxf(q.w, 0.f);
      q.x += a1.x; q.y += a1.y; q.z += a1.z; q.w += a1.w;
      y4[e1] = q;
    }
  }
}

extern "C" int lt_bn_apply_devox(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                                 float eps, const float* vox, const int32_t* idx, const float* w, float* y, int64_t N,
                                 int32_t C, int64_t V, hipStream_t stream) {
  int rc = check_vc("lt_bn_apply_devox", N, C);
  if (rc) return rc;
  LT_REQUIRE(x && mean && var && gamma && beta && y && idx && w && (V == 0 || vox) && V >= 0, LT_ERR_ARG,
             "lt_bn_apply_devox: null pointer");
  const int64_t n4 = N * (C / 4);
  const dim3 grid(bn_ew_grid(n4, C / 4));
  hipLaunchKernelGGL(k_bn_apply_devox, grid, dim3(256), 0, stream, x, mean, var, gamma, beta, eps, vox, idx, w, y, n4, C / 4,
                     V);
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
  const size_t lds = (size_t)rpi * 2 * C * 4;
  if (relu == 1) {
    BwdOp<1> op{dy, x, y, mean, var, eps, gamma, beta};
    hipLaunchKernelGGL(k_col_partial<BwdOp<1>>, dim3(nb), dim3(BN_BLOCK), lds, stream, op, V, C, (float*)ws);
  } else if (relu == 2) {
    BwdOp<2> op{dy, x, y, mean, var, eps, gamma, beta};
    hipLaunchKernelGGL(k_col_partial<BwdOp<2>>, dim3(nb), dim3(BN_BLOCK), lds, stream, op, V, C, (float*)ws);
  } else {
    BwdOp<0> op{dy, x, y, mean, var, eps, gamma, beta};
    hipLaunchKernelGGL(k_col_partial<BwdOp<0>>, dim3(nb), dim3(BN_BLOCK), lds, stream, op, V, C, (float*)ws);
  }
  LT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_partials, dim3(C), dim3(64), 0, stream, (const float*)ws, nb, C, sum_g, sum_gx);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// one float4 of the BatchNorm backward: g = dy under the ReLU mask (from y, or rebuilt from x when bt is given), then
// dx = gamma * rstd * (g - sum_g/V - xhat * sum_gx/V)
struct BnBwdCols {
  float4 m, vr, gm, bt, sg, sx;
};
template <bool MASK_Y, bool MASK_X>
__device__ __forceinline__ float4 bn_bwd_mask4(float4 g, float4 yy, float4 v, const BnBwdCols& c, float eps) {
  if (MASK_Y) {
    g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
    g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
  }
  if (MASK_X) {
    g.x = bn_affine(v.x, c.m.x, c.vr.x, c.gm.x, c.bt.x, eps) > 0.f ? g.x : 0.f;
    g.y = bn_affine(v.y, c.m.y, c.vr.y, c.gm.y, c.bt.y, eps) > 0.f ? g.y : 0.f;
    g.z = bn_affine(v.z, c.m.z, c.vr.z, c.gm.z, c.bt.z, eps) > 0.f ? g.z : 0.f;
    g.w = bn_affine(v.w, c.m.w, c.vr.w, c.gm.w, c.bt.w, eps) > 0.f ? g.w : 0.f;
  }
  return g;
}
__device__ __forceinline__ float4 bn_bwd_dx4(float4 g, float4 v, const BnBwdCols& c, float eps, float inv_v) {
  float4 o;
  float rs;
  rs = 1.0f / sqrtf(c.vr.x + eps); o.x = c.gm.x * rs * (g.x - c.sg.x * inv_v - (v.x - c.m.x) * rs * c.sx.x * inv_v);
  rs = 1.0f / sqrtf(c.vr.y + eps); o.y = c.gm.y * rs * (g.y - c.sg.y * inv_v - (v.y - c.m.y) * rs * c.sx.y * inv_v);
  rs = 1.0f / sqrtf(c.vr.z + eps); o.z = c.gm.z * rs * (g.z - c.sg.z * inv_v - (v.z - c.m.z) * rs * c.sx.z * inv_v);
  rs = 1.0f / sqrtf(c.vr.w + eps); o.w = c.gm.w * rs * (g.w - c.sg.w * inv_v - (v.w - c.m.w) * rs * c.sx.w * inv_v);
  return o;
}

// MASK_Y: ReLU mask from y; MASK_X: mask rebuilt from x (no residual was added, y is not kept); neither: no ReLU.
// dx and/or dres may be null (a residual branch only needs the masked gradient).
template <bool MASK_Y, bool MASK_X>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ y, const float* __restrict__ mean,
                                                      const float* __restrict__ var, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps,
                                                      const float* __restrict__ sum_g, const float* __restrict__ sum_gx,
                                                      float* __restrict__ dx, float* __restrict__ dres, int64_t n4,
                                                      int cg_n, float inv_v) {
  const int64_t stride = (int64_t)gridDim.x * 256;  // a multiple of cg_n (bn_ew_grid)
  int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x;
  const int cg = (int)(e % cg_n);
  BnBwdCols c;
  c.m = reinterpret_cast<const float4*>(mean)[cg];
  c.vr = reinterpret_cast<const float4*>(var)[cg];
  c.gm = reinterpret_cast<const float4*>(gamma)[cg];
  c.bt = MASK_X ? reinterpret_cast<const float4*>(beta)[cg] : make_float4(0.f, 0.f, 0.f, 0.f);
  c.sg = reinterpret_cast<const float4*>(sum_g)[cg];
  c.sx = reinterpret_cast<const float4*>(sum_gx)[cg];
  const float4* dy4 = reinterpret_cast<const float4*>(dy);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* y4 = reinterpret_cast<const float4*>(y);
  float4* dx4 = reinterpret_cast<float4*>(dx);
  float4* dr4 = reinterpret_cast<float4*>(dres);
  const bool need_x = MASK_X || dx != nullptr;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (; e + 3 * stride < n4; e += 4 * stride) {
    float4 g[4], v[4], yy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = dy4[e + i * stride];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = need_x ? x4[e + i * stride] : z;
#pragma unroll
    for (int i = 0; i < 4; ++i) yy[i] = MASK_Y ? y4[e + i * stride] : z;
    __builtin_amdgcn_sched_barrier(0);  // all loads issue before the first use
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 gi = bn_bwd_mask4<MASK_Y, MASK_X>(g[i], yy[i], v[i], c, eps);
      if (dres) dr4[e + i * stride] = gi;
      if (dx) dx4[e + i * stride] = bn_bwd_dx4(gi, v[i], c, eps, inv_v);
    }
  }
  for (; e < n4; e += stride) {
    const float4 v = need_x ? x4[e] : z;
    const float4 gi = bn_bwd_mask4<MASK_Y, MASK_X>(dy4[e], MASK_Y ? y4[e] : z, v, c, eps);
    if (dres) dr4[e] = gi;
    if (dx) dx4[e] = bn_bwd_dx4(gi, v, c, eps, inv_v);
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
  const dim3 grid(bn_ew_grid(n4, C / 4));
  const float inv_v = 1.0f / (float)V;
  if (relu == 1)
    hipLaunchKernelGGL((k_bn_bwd_apply<true, false>), grid, dim3(256), 0, stream, dy, x, y, mean, var, gamma, beta, eps, sum_g,
                       sum_gx, dx, dres, n4, C / 4, inv_v);
  else if (relu == 2)
    hipLaunchKernelGGL((k_bn_bwd_apply<false, true>), grid, dim3(256), 0, stream, dy, x, y, mean, var, gamma, beta, eps, sum_g,
                       sum_gx, dx, dres, n4, C / 4, inv_v);
  else
    hipLaunchKernelGGL((k_bn_bwd_apply<false, false>), grid, dim3(256), 0, stream, dy, x, y, mean, var, gamma, beta, eps,
                       sum_g, sum_gx, dx, dres, n4, C / 4, inv_v);
  LT_LAUNCH_CHECK();
  return LT_OK;
}
