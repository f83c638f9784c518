// Backward-weight of the sparse convolution:  dW_k = sum over the pairs (i, o) of offset k of  in[i,:]^T  gout[o,:]
//
// Replaces the `grad_weight` half of torchsparse v1.4.0's conv3d backward (per offset: in[nbmaps[:,0]]^T @ gout[nbmaps[:,1]]);
// reached when the conv kernels train: TRAIN.UPDATE_PARAMS == "all" (latte/common/utils/ema_util.py:44-71,
// latte/tta/tools/latte.py:324-326).  Under the shipped norm-only configuration the kernels are frozen and this never runs.
//
// Mapping to CDNA4 (fp32 in, fp32 accumulate on v_mfma_f32_16x16x4_f32): the pairs are the reduction index.  A wave owns one
// offset k, a (16*MT) x (16*NT) block of dW_k and a contiguous slice of that offset's pair list; per step it takes 4 pairs:
// lane (m, q) loads MT consecutive input channels of pair q's input row and NT consecutive gradient channels of its output
// row (one 4*MT / 4*NT-byte load each; the 16 lanes of a pair cover 64*MT / 64*NT contiguous bytes of the two rows) and
// issues MT*NT MFMAs -- MFMA row m of tile (a, b) is input channel MT*m + a, column n is output channel NT*n + b, a
// permutation that only the final store has to undo.  Slices are summed in a second kernel in ascending slice order:
// bitwise reproducible, no atomics.
#include "common.h"
#include "../../include/latte_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LT_DW_WAVES 4

struct DwArgs {
  const float* in;        // [Vin, Cin]   (the conv's input features; for a transposed conv the roles are swapped by the host)
  const float* gout;      // [Vout, Cout]
  const int* nbmaps;      // [M, 2] (in, out), k-major
  const int* nbsizes;     // [K]
  float* partial;         // [S][K][Cin][Cout]
  int Cin, Cout, K, S, col_in, col_out;
};

template <int T>
struct DwVec {
  float v[T];
};

template <int T>
__device__ __forceinline__ DwVec<T> dw_load(const float* p, bool ok) {
  DwVec<T> r;
#pragma unroll
  for (int t = 0; t < T; ++t) r.v[t] = 0.f;
  if (ok) {
    if (T == 4) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(p);
      r.v[0] = x[0]; r.v[1] = x[1]; r.v[2] = x[2]; r.v[3] = x[3];
    } else if (T == 2) {
      const float2 x = *reinterpret_cast<const float2*>(p);
      r.v[0] = x.x; r.v[1] = x.y;
    } else {
      r.v[0] = *p;
    }
  }
  return r;
}

template <int MT, int NT>
__global__ __launch_bounds__(64 * LT_DW_WAVES) void k_spconv_dw(DwArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int k = blockIdx.x / (a.S / LT_DW_WAVES);
  const int s = (blockIdx.x % (a.S / LT_DW_WAVES)) * LT_DW_WAVES + wave;
  const int ci0 = blockIdx.y * 16 * MT, co0 = blockIdx.z * 16 * NT;
  // pair range of offset k (K <= 32: a serial prefix sum per thread is cheaper than a scan), then this wave's slice
  int64_t begin = 0;
  for (int j = 0; j < k; ++j) begin += a.nbsizes[j];
  const int64_t mk = a.nbsizes[k];
  const int64_t per = ((mk + a.S - 1) / a.S + 3) / 4 * 4;   // multiple of 4 pairs per slice
  int64_t p0 = begin + (int64_t)s * per, p1 = p0 + per;
  const int64_t end = begin + mk;
  if (p1 > end) p1 = end;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int64_t p = p0; p < p1; p += 8) {
    // two groups of 4 pairs in flight
    DwVec<MT> av[2];
    DwVec<NT> bv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t pp = p + 4 * u + q;
      const bool ok = pp < p1;
      int i = 0, o = 0;
      if (ok) {
        const int2 pr = reinterpret_cast<const int2*>(a.nbmaps)[pp];
        i = a.col_in == 0 ? pr.x : pr.y;
        o = a.col_in == 0 ? pr.y : pr.x;
      }
      av[u] = dw_load<MT>(a.in + (int64_t)i * a.Cin + ci0 + MT * m, ok);
      bv[u] = dw_load<NT>(a.gout + (int64_t)o * a.Cout + co0 + NT * m, ok);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].v[x], bv[u].v[y], acc[x][y], 0, 0, 0);
  }
  // D layout: row = 4*(lane>>4) + reg, col = lane&15  ->  dW[ci0 + MT*row + x][co0 + NT*col + y]
  float* out = a.partial + ((int64_t)s * a.K + k) * a.Cin * a.Cout;
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int ci = ci0 + MT * (4 * q + reg) + x;
      float* o = out + (int64_t)ci * a.Cout + co0 + NT * m;
#pragma unroll
      for (int y = 0; y < NT; ++y) o[y] = acc[x][y][reg];
    }
}

__global__ void k_dw_reduce(const float* __restrict__ partial, int S, int64_t n, float* __restrict__ dW) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += partial[(int64_t)s * n + e];   // ascending slice order
    dW[e] = acc;
  }
}

static int dw_tiles(int C) { return C % 64 == 0 ? 4 : C % 32 == 0 ? 2 : 1; }

// slices per offset: enough waves to fill the chip (K * S * blocks >= ~8 k waves), a multiple of the waves per workgroup
static int dw_slices(int64_t M, int K, int Cin, int Cout) {
  const int64_t blocks = (int64_t)(Cin / (16 * dw_tiles(Cin))) * (Cout / (16 * dw_tiles(Cout)));
  int64_t s = (8192 + K * blocks - 1) / (K * blocks);
  const int64_t by_pairs = (M / K + 255) / 256;   // at least ~256 pairs per slice
  if (s > by_pairs) s = by_pairs;
  if (s < LT_DW_WAVES) s = LT_DW_WAVES;
  if (s > 256) s = 256;
  return (int)((s + LT_DW_WAVES - 1) / LT_DW_WAVES * LT_DW_WAVES);
}

extern "C" int64_t lt_spconv_bwd_weight_workspace_bytes(int64_t M, int32_t K, int32_t Cin, int32_t Cout) {
  if (K < 1 || Cin < 16 || Cout < 16) return 256;
  return (int64_t)dw_slices(M, K, Cin, Cout) * K * Cin * Cout * 4 + 256;
}

template <int MT>
static void dw_launch_n(const DwArgs& a, int nt, dim3 grid, hipStream_t stream) {
  if (nt == 4) hipLaunchKernelGGL((k_spconv_dw<MT, 4>), grid, dim3(64 * LT_DW_WAVES), 0, stream, a);
  else if (nt == 2) hipLaunchKernelGGL((k_spconv_dw<MT, 2>), grid, dim3(64 * LT_DW_WAVES), 0, stream, a);
  else hipLaunchKernelGGL((k_spconv_dw<MT, 1>), grid, dim3(64 * LT_DW_WAVES), 0, stream, a);
}

extern "C" int lt_spconv_bwd_weight(const float* in, const float* gout, const int32_t* nbmaps, const int32_t* nbsizes,
                                    int64_t M, int64_t Vin, int64_t Vout, int32_t Cin, int32_t Cout, int32_t K,
                                    int32_t transposed, void* ws, int64_t ws_bytes, float* dW, hipStream_t stream) {
  LT_REQUIRE(M >= 0 && Vin >= 0 && Vout >= 0 && K >= 1 && K <= 32, LT_ERR_ARG, "lt_spconv_bwd_weight: bad sizes (K=%d)", K);
  LT_REQUIRE(Cin >= 16 && Cin % 16 == 0 && Cout >= 16 && Cout % 16 == 0, LT_ERR_ARG,
             "lt_spconv_bwd_weight: Cin=%d and Cout=%d must be multiples of 16", Cin, Cout);
  LT_REQUIRE(dW && nbsizes && ws, LT_ERR_ARG, "lt_spconv_bwd_weight: null pointer");
  const int64_t n = (int64_t)K * Cin * Cout;
  if (M == 0) {
    LT_CHECK_HIP(hipMemsetAsync(dW, 0, n * 4, stream));
    return LT_OK;
  }
  LT_REQUIRE(in && gout && nbmaps, LT_ERR_ARG, "lt_spconv_bwd_weight: null pointer");
  LT_REQUIRE(ws_bytes >= lt_spconv_bwd_weight_workspace_bytes(M, K, Cin, Cout), LT_ERR_WORKSPACE,
             "lt_spconv_bwd_weight: workspace needs %lld bytes, got %lld",
             (long long)lt_spconv_bwd_weight_workspace_bytes(M, K, Cin, Cout), (long long)ws_bytes);
  const int S = dw_slices(M, K, Cin, Cout);
  const int mt = dw_tiles(Cin), nt = dw_tiles(Cout);
  // nbmaps columns are (in, out) of the NON-transposed conv the map was built for; a transposed conv runs it backwards
  DwArgs a{in, gout, nbmaps, nbsizes, (float*)ws, Cin, Cout, K, S, transposed ? 1 : 0, transposed ? 0 : 1};
  dim3 grid((unsigned)(K * (S / LT_DW_WAVES)), (unsigned)(Cin / (16 * mt)), (unsigned)(Cout / (16 * nt)));
  if (mt == 4) dw_launch_n<4>(a, nt, grid, stream);
  else if (mt == 2) dw_launch_n<2>(a, nt, grid, stream);
  else dw_launch_n<1>(a, nt, grid, stream);
  LT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dw_reduce, dim3(lt_grid(n, 256)), dim3(256), 0, stream, (const float*)ws, S, n, dW);
  LT_LAUNCH_CHECK();
  return LT_OK;
}
