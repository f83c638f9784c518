// Segmentation heads fused with inverse_to_all: per RAW point logits of both linear heads straight from the per-voxel
// backbone features,  out_h[p,:] = feats[inv[p],:] @ W_h^T + b_h  (h = 1, 2).
//
// Replaces Net3DSeg's two nn.Linear heads (latte/models/xmuda_arch.py:326-335) followed by inverse_to_all
// (latte/data/collate.py:34-55: per scan logit[inverse_map]) -- two GEMMs over [V, 96], then a Python loop of B gathers in
// the reference -- by ONE pass: a wave takes 16 points, gathers their 16 feature rows as MFMA A fragments (lane (r, q) loads
// 16 B of row inv[p0 + r] at channel 16 j + 4 q: the four lanes of a row cover 64 contiguous bytes), multiplies them with the
// packed weight fragments of both heads, which it keeps in registers for all its strips (Cin/16 x NT KiB), and stores the
// two [16, C] blocks with the bias added.  fp32 in, fp32 accumulate on v_mfma_f32_16x16x4_f32 (same fragment layout and
// channel permutation as the sparse conv, csrc/spconv.hip).  HBM-bound: 4*Cin B gathered + 8*C B written per point.
#include "common.h"
#include "../../include/latte_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LT_HEAD_JMAX 8   // Cin <= 128

template <int NT>
__global__ __launch_bounds__(256) void k_head_gather(const float* __restrict__ feats, const int64_t* __restrict__ idx,
                                                     const float* __restrict__ packed, const float* __restrict__ bias,
                                                     int64_t N, int64_t V, int Cin, int C1, int C2,
                                                     float* __restrict__ out1, float* __restrict__ out2) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int J = Cin / 16;
  // weight fragments of this wave: packed[((j*NT + nt)*64 + lane)*4 + t] = E[16j + 4q + t][16nt + n]
  f32x4 b[LT_HEAD_JMAX][NT];
#pragma unroll
  for (int j = 0; j < LT_HEAD_JMAX; ++j)
#pragma unroll
    for (int n = 0; n < NT; ++n)
      b[j][n] = j < J ? reinterpret_cast<const f32x4*>(packed)[(j * NT + n) * 64 + lane] : (f32x4){0.f, 0.f, 0.f, 0.f};
  float bs[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) bs[n] = bias[16 * n + r];

  const int64_t strips = (N + 15) / 16;
  const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t s = wave_id; s < strips; s += n_waves) {
    const int64_t p = s * 16 + r;
    int64_t row = -1;
    if (p < N) row = idx ? idx[p] : p;
    const bool ok = row >= 0 && row < V;
    const float* src = feats + (ok ? row : 0) * Cin + 4 * q;
    f32x4 a[LT_HEAD_JMAX];
#pragma unroll
    for (int j = 0; j < LT_HEAD_JMAX; ++j)
      a[j] = (ok && j < J) ? *reinterpret_cast<const f32x4*>(src + 16 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < LT_HEAD_JMAX; ++j) {
      if (j < J) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][t], b[j][n][t], acc[n], 0, 0, 0);
      }
    }
    // D layout: row = 4q + reg, col = r
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t po = s * 16 + 4 * q + reg;
      if (po < N) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int c = 16 * n + r;
          const float v = acc[n][reg] + bs[n];
          if (c < C1) out1[po * C1 + c] = v;
          else if (c < C1 + C2) out2[po * C2 + (c - C1)] = v;
        }
      }
    }
  }
}

extern "C" int lt_head_gather_fwd(const float* feats, const int64_t* idx, const float* packed, const float* bias, int64_t N,
                                  int64_t V, int32_t Cin, int32_t C1, int32_t C2, float* out1, float* out2,
                                  hipStream_t stream) {
  LT_REQUIRE(N >= 0 && V >= 0, LT_ERR_ARG, "lt_head_gather_fwd: bad sizes");
  LT_REQUIRE(Cin >= 16 && Cin % 16 == 0 && Cin <= 16 * LT_HEAD_JMAX, LT_ERR_ARG,
             "lt_head_gather_fwd: Cin=%d must be a multiple of 16, at most %d", Cin, 16 * LT_HEAD_JMAX);
  LT_REQUIRE(C1 >= 1 && C2 >= 0 && C1 + C2 <= 64, LT_ERR_ARG, "lt_head_gather_fwd: C1=%d, C2=%d (C1 + C2 <= 64)", C1, C2);
  if (N == 0) return LT_OK;
  LT_REQUIRE(feats && packed && bias && out1 && (C2 == 0 || out2), LT_ERR_ARG, "lt_head_gather_fwd: null pointer");
  const int nt = (C1 + C2 + 15) / 16;
  const int grid = lt_grid((N + 15) / 16 * 64, 256);
  switch (nt) {
    case 1: hipLaunchKernelGGL(k_head_gather<1>, dim3(grid), dim3(256), 0, stream, feats, idx, packed, bias, N, V, Cin, C1, C2, out1, out2); break;
    case 2: hipLaunchKernelGGL(k_head_gather<2>, dim3(grid), dim3(256), 0, stream, feats, idx, packed, bias, N, V, Cin, C1, C2, out1, out2); break;
    case 3: hipLaunchKernelGGL(k_head_gather<3>, dim3(grid), dim3(256), 0, stream, feats, idx, packed, bias, N, V, Cin, C1, C2, out1, out2); break;
    default: hipLaunchKernelGGL(k_head_gather<4>, dim3(grid), dim3(256), 0, stream, feats, idx, packed, bias, N, V, Cin, C1, C2, out1, out2); break;
  }
  LT_LAUNCH_CHECK();
  return LT_OK;
}
