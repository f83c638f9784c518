// Sparse 3-D convolution for gfx950: output-stationary gather -> fp32 MFMA -> plain store.
//
// Replaces spnn.Conv3d forward / backward-input of torchsparse v1.4.0 (gather -> cuBLAS GEMM -> scatter-add, one
// round per kernel offset); reference call sites latte/models/basic_blocks.py:17,34,51,58,68 and
// latte/models/spvcnn.py:30,32.
//
//   out[o,:] = sum_k in[nbr[k,o],:] @ Wk
//
// Mapping to CDNA4
//  * a workgroup (4 waves) owns 64 output rows x (16*NT) output channels; each wave owns a 16-row strip and keeps
//    its NT 16x16 accumulators (v_mfma_f32_16x16x4_f32, exact fp32 = fmaf chain) in registers for the whole K loop:
//    every output row is written exactly once -> no scatter, no atomics, bitwise reproducible.
//  * A operand (gathered input rows): lane (r=lane&15, q=lane>>4) loads 16 B of row nbr[k][row r] at channel
//    16j+4q straight into VGPRs (4 lanes cover 64 contiguous bytes of a row); the 4 floats feed 4 consecutive
//    MFMAs, i.e. the reduction index inside a 16-channel group is permuted (4q+t), and the weights are packed
//    with the same permutation, so no LDS transpose of the gathered rows is needed.
//  * B operand (Wk): packed once per call into MFMA-fragment order [k][j][nt][lane][t]; a workgroup stages
//    JB x NT KiB of it per step through LDS with 16-B loads/stores and each lane reads its 4 t-values with one
//    conflict-free ds_read_b128 (lane-linear image).
//  * offsets with no neighbour in the 64-row tile are skipped for the whole workgroup, offsets with no neighbour
//    in a wave's 16-row strip skip that wave's MFMAs (wave-uniform branch).
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "common.h"
#include <cstdlib>
#include "../../include/latte_hip.h"
#include "spconv_args.h"

#define LT_JB 4         // 16-channel groups staged per step (LT_TM, LT_KMAX: spconv_args.h)

// ---------------------------------------------------------------- weight packing
// vec layout  (Cin_e % 16 == 0): packed[(((k*J + j)*NTt + nt)*64 + lane)*4 + t] = E_k[16j + 4q + t][16nt + n]
// scalar layout (otherwise)    : packed[((k*S + s)*NTt + nt)*64 + lane]          = E_k[4s + q][16nt + n]
// with q = lane>>4, n = lane&15 and E_k = Wk (mode 0), Wk^T (mode 1), W(K-1-k)^T (mode 2).
__global__ void k_pack_weights(const float* __restrict__ W, int K, int Cin, int Cout, int mode,
                               float* __restrict__ packed) {
  const int Ce_in = mode == 0 ? Cin : Cout;
  const int Ce_out = mode == 0 ? Cout : Cin;
  const int NTt = Ce_out / 16;
  const bool vec = (Ce_in % 16) == 0;
  const int64_t total = (int64_t)K * Ce_in * Ce_out;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = p;
    int k, ci, co;
    if (vec) {
      const int t = (int)(r & 3); r >>= 2;
      const int lane = (int)(r & 63); r >>= 6;
      const int nt = (int)(r % NTt); r /= NTt;
      const int J = Ce_in / 16;
      const int j = (int)(r % J);
      k = (int)(r / J);
      ci = 16 * j + 4 * (lane >> 4) + t;
      co = 16 * nt + (lane & 15);
    } else {
      const int lane = (int)(r & 63); r >>= 6;
      const int nt = (int)(r % NTt); r /= NTt;
      const int S = Ce_in / 4;
      const int s = (int)(r % S);
      k = (int)(r / S);
      ci = 4 * s + (lane >> 4);
      co = 16 * nt + (lane & 15);
    }
    float v;
    if (mode == 0) v = W[((int64_t)k * Cin + ci) * Cout + co];
    else {
      const int ks = mode == 2 ? (K - 1 - k) : k;
      v = W[((int64_t)ks * Cin + co) * Cout + ci];  // E[ci][co] = W[ks][co][ci]
    }
    packed[p] = v;
  }
}

extern "C" int64_t lt_spconv_packed_floats(int32_t K, int32_t Cin, int32_t Cout) { return (int64_t)K * Cin * Cout; }
__global__ void k_pack_weights_bf16(const float* __restrict__ W, int K, int Cin, int Cout, int mode,
                                    unsigned short* __restrict__ packed);

extern "C" int lt_spconv_pack_weights(const float* W, int32_t K, int32_t Cin, int32_t Cout, int32_t mode,
                                      float* packed, hipStream_t stream) {
  LT_REQUIRE(W && packed, LT_ERR_ARG, "lt_spconv_pack_weights: null pointer");
  LT_REQUIRE(K >= 1 && Cin >= 1 && Cout >= 1 && mode >= 0 && mode <= 6 && mode != 3, LT_ERR_ARG, "lt_spconv_pack_weights: bad args");
  if (mode >= 4) {  // bf16 hi/lo split image of modes 0..2 for the bf16x3 kernel (see k_spconv_bf)
    const int ce_in = mode == 4 ? Cin : Cout, ce_out = mode == 4 ? Cout : Cin;
    LT_REQUIRE(ce_in % 32 == 0 && ce_out % 16 == 0, LT_ERR_ARG,
               "lt_spconv_pack_weights: bf16 image needs effective Cin=%d %% 32 == 0 and Cout=%d %% 16 == 0", ce_in, ce_out);
    const int64_t tot = (int64_t)K * Cin * Cout;
    hipLaunchKernelGGL(k_pack_weights_bf16, dim3(lt_grid(tot, 256)), dim3(256), 0, stream, W, K, Cin, Cout, mode,
                       reinterpret_cast<unsigned short*>(packed));
    LT_LAUNCH_CHECK();
    return LT_OK;
  }
  const int Ce_in = mode == 0 ? Cin : Cout, Ce_out = mode == 0 ? Cout : Cin;
  LT_REQUIRE(Ce_in % 4 == 0 && Ce_out % 16 == 0, LT_ERR_ARG,
             "lt_spconv_pack_weights: effective Cin=%d must be a multiple of 4 and Cout=%d of 16", Ce_in, Ce_out);
  const int64_t total = (int64_t)K * Cin * Cout;
  hipLaunchKernelGGL(k_pack_weights, dim3(lt_grid(total, 256)), dim3(256), 0, stream, W, K, Cin, Cout, mode, packed);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------- forward kernel
struct ConvArgs {
  const float* in;
  const float* packed;
  const int* nbr;
  const int* order;
  float* out;
  int Vin, Vout, Cin, Cout, K;
};

// shared prologue: row ids of the tile + neighbour table of the tile + per-offset row masks
__device__ __forceinline__ void conv_load_tile(const ConvArgs& a, int* s_rows, int* s_nbr, unsigned long long* s_mask) {
  const int tid = threadIdx.x;
  const int64_t tile0 = (int64_t)blockIdx.x * LT_TM;
  if (tid < LT_TM) {
    const int64_t slot = tile0 + tid;
    int row = -1;
    if (slot < a.Vout) row = a.order ? a.order[slot] : (int)slot;
    s_rows[tid] = row;
  }
  __syncthreads();
  for (int e = tid; e < a.K * LT_TM; e += blockDim.x) {
    const int k = e / LT_TM, i = e % LT_TM;
    const int row = s_rows[i];
    int v = -1;
    if (row >= 0) v = a.nbr[(int64_t)k * a.Vout + row];
    if (v >= a.Vin) v = -1;  // defensive: never gather out of bounds
    s_nbr[e] = v;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int k = wave; k < a.K; k += 4) {
    const unsigned long long m = __ballot(s_nbr[k * LT_TM + lane] >= 0);
    if (lane == 0) s_mask[k] = m;
  }
  __syncthreads();
}

template <int NT>
__global__ __launch_bounds__(256) void k_spconv_vec(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float s_w[LT_JB * NT * 64 * 4];
  __shared__ int s_nbr[LT_KMAX * LT_TM];
  __shared__ int s_rows[LT_TM];
  __shared__ unsigned long long s_mask[LT_KMAX];

  conv_load_tile(a, s_rows, s_nbr, s_mask);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int J = a.Cin / 16;
  const int NTt = a.Cout / 16;
  const int nt0 = blockIdx.y * NT;

  f32x4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k = 0; k < a.K; ++k) {
    const unsigned long long m = s_mask[k];
    if (m == 0ull) continue;  // workgroup-uniform
    const bool wave_active = ((m >> (16 * wave)) & 0xFFFFull) != 0ull;
    const int src = s_nbr[k * LT_TM + 16 * wave + r];
    const float* src_row = a.in + (int64_t)(src < 0 ? 0 : src) * a.Cin + 4 * q;
    for (int j0 = 0; j0 < J; j0 += LT_JB) {
      const int jb = (J - j0) < LT_JB ? (J - j0) : LT_JB;
      __syncthreads();  // previous step's readers are done with s_w
      // stage jb slabs of NT KiB: slab (k, j0+jj) starts at float4 index ((k*J + j0+jj)*NTt + nt0)*64
      const int n4 = jb * NT * 64;
      for (int e = tid; e < n4; e += 256) {
        const int jj = e / (NT * 64), rem = e - jj * (NT * 64);
        const f32x4 v = reinterpret_cast<const f32x4*>(a.packed)[((int64_t)(k * J + j0 + jj) * NTt + nt0) * 64 + rem];
        reinterpret_cast<f32x4*>(s_w)[e] = v;
      }
      // gather this step's A fragments (issued before the barrier: overlaps the staging)
      f32x4 av[LT_JB];
#pragma unroll
      for (int jj = 0; jj < LT_JB; ++jj) {
        av[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (wave_active && jj < jb && src >= 0) av[jj] = *reinterpret_cast<const f32x4*>(src_row + 16 * (j0 + jj));
      }
      __syncthreads();
      if (wave_active) {
#pragma unroll
        for (int jj = 0; jj < LT_JB; ++jj) {
          if (jj < jb) {
            // t outermost: consecutive MFMAs hit different accumulators (a dependent 16x16x4 needs 40 cycles, an
            // independent one issues every 32); the sum order per accumulator is unchanged (t ascending)
            f32x4 b[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) b[n] = reinterpret_cast<const f32x4*>(s_w)[(jj * NT + n) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
              for (int n = 0; n < NT; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj][t], b[n][t], acc[n], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  // epilogue: C/D layout col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = s_rows[16 * wave + 4 * q + reg];
    if (row >= 0) {
      float* o = a.out + (int64_t)row * a.Cout + 16 * nt0 + r;
#pragma unroll
      for (int n = 0; n < NT; ++n) o[16 * n] = acc[n][reg];
    }
  }
}

// small-Cin variant (Cin % 16 != 0, e.g. the 4-channel stem): one dword of A per MFMA, all of Wk staged at once
template <int NT>
__global__ __launch_bounds__(256) void k_spconv_scalar(ConvArgs a) {
  __shared__ float s_w[3 * NT * 64];  // S <= 3 (Cin <= 12)
  __shared__ int s_nbr[LT_KMAX * LT_TM];
  __shared__ int s_rows[LT_TM];
  __shared__ unsigned long long s_mask[LT_KMAX];

  conv_load_tile(a, s_rows, s_nbr, s_mask);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int S = a.Cin / 4;
  const int NTt = a.Cout / 16;
  const int nt0 = blockIdx.y * NT;

  f32x4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k = 0; k < a.K; ++k) {
    const unsigned long long m = s_mask[k];
    if (m == 0ull) continue;
    const bool wave_active = ((m >> (16 * wave)) & 0xFFFFull) != 0ull;
    const int src = s_nbr[k * LT_TM + 16 * wave + r];
    __syncthreads();
    for (int e = tid; e < S * NT * 64; e += 256) {
      const int s = e / (NT * 64), rem = e - s * (NT * 64);
      s_w[e] = a.packed[((int64_t)(k * S + s) * NTt + nt0) * 64 + rem];
    }
    float av[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (wave_active && s < S && src >= 0) av[s] = a.in[(int64_t)src * a.Cin + 4 * s + q];
    __syncthreads();
    if (wave_active) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (s < S) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], s_w[(s * NT + n) * 64 + lane], acc[n], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = s_rows[16 * wave + 4 * q + reg];
    if (row >= 0) {
      float* o = a.out + (int64_t)row * a.Cout + 16 * nt0 + r;
#pragma unroll
      for (int n = 0; n < NT; ++n) o[16 * n] = acc[n][reg];
    }
  }
}

// =====================================================================================================
// Planned kernel: the hot path.  Works on a "plan" of the neighbour table (lt_kmap_plan, kmap.hip): rows sorted by
// neighbour mask, the table re-laid out per 64-row tile ([tile][k][64], coalesced) and one 64-bit row mask per
// (tile, k).  Differences to k_spconv_vec:
//  * only the offsets that are active in the tile are visited (compacted list), 16-row strips without a neighbour at
//    k skip that k's MFMAs;
//  * the weights of the next stage are copied global -> LDS by the LDS-DMA engine (global_load_lds_dwordx4, the packed
//    fragment image is lane-linear) while the MFMAs of the current stage run: two LDS buffers, ONE barrier per stage;
//    the gathered A rows of the next stage are prefetched into a second register set at the same point;
//  * a stage is UB "slabs" (= one offset x one 16-channel group of Cin), so narrow layers still get >= 16 KiB stages.
// Every output element is still one k-ascending, channel-ascending fmaf chain -> bitwise equal to k_spconv_vec.
// PlanArgs: spconv_args.h (shared with spconv_ls.hip)

// events of the armed conv launch timer for the launch being issued (null otherwise): the conv kernels are launched
// with hipExtLaunchKernelGGL, which stamps them with the dispatch's own begin / end times (no extra barrier packets)
// thread_local: set by LtConvTimerScope in the thread that issues the launch, read by LT_CONV_LAUNCH in the same thread
static thread_local hipEvent_t g_conv_ev0 = nullptr, g_conv_ev1 = nullptr;
#define LT_CONV_LAUNCH(kernel, grid, block, stream, ...) \
  hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, g_conv_ev0, g_conv_ev1, 0, __VA_ARGS__)
void lt_conv_launch_events(hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = g_conv_ev0;
  *e1 = g_conv_ev1;
}

// diagnostics: per-wave time stamps of k_spconv_rf (variant bit dbg & 4); buffer set by lt_conv_debug_stamps
static long long* g_conv_stamps = nullptr;
static long long g_conv_stamps_cap = 0;
extern "C" int lt_conv_debug_stamps(void* buf, int64_t cap_words) {
  g_conv_stamps = (long long*)buf;
  g_conv_stamps_cap = cap_words;
  return LT_OK;
}

typedef __attribute__((address_space(3))) void* lt_lds_ptr;
typedef __attribute__((address_space(1))) const void* lt_glb_ptr;

template <int NT, int UB, int SR>
__global__ __launch_bounds__(256) void k_spconv_plan(PlanArgs a) {
  constexpr int WB = UB * NT * 256;  // floats per weight buffer (UB*NT fragments of 1 KiB)
  constexpr int TMW = LT_TM * SR;    // output rows per workgroup: SR plan tiles, wave w owns strip w of each
  // one LDS object (a second one beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) float smem[2 * WB + LT_KMAX * TMW + TMW + LT_KMAX * (1 + 2 * SR) + 4];
  float* s_w = smem;
  int* s_nbr = reinterpret_cast<int*>(smem + 2 * WB);        // [nact][TMW] (compacted offsets)
  int* s_rows = s_nbr + LT_KMAX * TMW;                         // [TMW]
  int* s_klist = s_rows + TMW;                                 // [32] active offsets, ascending
  unsigned int* s_km = reinterpret_cast<unsigned int*>(s_klist + LT_KMAX);  // [32][SR][2] their row masks
  int* s_nact = reinterpret_cast<int*>(s_km + 2 * SR * LT_KMAX);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  // first plan tile of this workgroup; single-tile workgroups are dispatched heaviest first
  const int64_t tile0 = SR == 1 ? (int64_t)a.torder[blockIdx.x] : (int64_t)blockIdx.x * SR;
  const int64_t T = ((int64_t)a.Vout + LT_TM - 1) / LT_TM;
  const int J = a.Cin / 16, NTt = a.Cout / 16, nt0 = blockIdx.y * NT;

  if (wave == 0) {
    unsigned long long m[SR];
    unsigned long long any = 0ull;
#pragma unroll
    for (int h = 0; h < SR; ++h) {
      m[h] = 0ull;
      if (lane < a.K && tile0 + h < T) m[h] = a.tmask[(tile0 + h) * a.K + lane];
      any |= m[h];
    }
    const unsigned long long bal = __ballot(any != 0ull);
    if (any != 0ull) {
      const int pos = __popcll(bal & ((1ull << lane) - 1ull));
      s_klist[pos] = lane;
#pragma unroll
      for (int h = 0; h < SR; ++h) {
        s_km[(pos * SR + h) * 2] = (unsigned int)m[h];
        s_km[(pos * SR + h) * 2 + 1] = (unsigned int)(m[h] >> 32);
      }
    }
    if (lane == 0) *s_nact = __popcll(bal);
  } else if (wave <= SR) {
    const int h = wave - 1;
    const int64_t slot = (tile0 + h) * LT_TM + lane;
    s_rows[h * LT_TM + lane] = slot < a.Vout ? a.order[slot] : -1;
  }
  __syncthreads();
  const int nact = *s_nact;
  for (int e = tid; e < nact * TMW; e += 256) {
    const int ki = e / TMW, i = e - ki * TMW;
    const int h = i >> 6;
    int v = -1;
    if (tile0 + h < T) v = a.tnbr[((tile0 + h) * a.K + s_klist[ki]) * LT_TM + (i & 63)];
    if (v >= a.Vin) v = -1;  // defensive: never gather out of bounds
    s_nbr[e] = v;
  }
  __syncthreads();

  f32x4 acc[SR][NT];
#pragma unroll
  for (int h = 0; h < SR; ++h)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[h][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int U = nact * J;                 // slabs of this workgroup, (ki, j) with j fastest
  const int S = (U + UB - 1) / UB;        // stages
  const float* in_q = a.in + 4 * q;

  // 16-bit row mask of this wave's strip of plan tile h at compacted offset ki
  auto strip_mask = [&](int ki, int h) -> unsigned int {
    return (s_km[(ki * SR + h) * 2 + (wave >> 1)] >> (16 * (wave & 1))) & 0xFFFFu;
  };

  // issue stage s: LDS-DMA of its weight fragments into buffer `buf`, gather of its A fragments into `av`
  auto issue = [&](int s, int buf, f32x4 (&av)[UB][SR]) {
    const int u0 = s * UB;
    const int nu = (U - u0) < UB ? (U - u0) : UB;
    // fragment f = uu*NT + n of the stage; wave w copies f = w, w+4, ...
    for (int f = wave; f < nu * NT; f += 4) {
      const int uu = f / NT, n = f - uu * NT;
      const int u = u0 + uu;
      const int ki = u / J, j = u - ki * J;
      const int k = s_klist[ki];
      const float* g = a.packed + ((((int64_t)k * J + j) * NTt + nt0 + n) * 64 + lane) * 4;
      __builtin_amdgcn_global_load_lds((lt_glb_ptr)g, (lt_lds_ptr)(s_w + buf * WB + f * 256), 16, 0, 0);
    }
#pragma unroll
    for (int uu = 0; uu < UB; ++uu) {
#pragma unroll
      for (int h = 0; h < SR; ++h) {
        av[uu][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (uu < nu) {
          const int u = u0 + uu;
          const int ki = u / J, j = u - ki * J;
          if (strip_mask(ki, h) != 0u) {
            const int src = s_nbr[ki * TMW + h * LT_TM + 16 * wave + r];
            if (src >= 0) av[uu][h] = *reinterpret_cast<const f32x4*>(in_q + (int64_t)src * a.Cin + 16 * j);
          }
        }
      }
    }
  };
  auto compute = [&](int s, int buf, const f32x4 (&av)[UB][SR]) {
    const int u0 = s * UB;
    const int nu = (U - u0) < UB ? (U - u0) : UB;
#pragma unroll
    for (int uu = 0; uu < UB; ++uu) {
      if (uu < nu) {
        const int ki = (u0 + uu) / J;
        bool on[SR];
        bool any = false;
#pragma unroll
        for (int h = 0; h < SR; ++h) { on[h] = strip_mask(ki, h) != 0u; any = any || on[h]; }
        if (any) {  // wave-uniform
          f32x4 b[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n)
            b[n] = reinterpret_cast<const f32x4*>(s_w + buf * WB)[(uu * NT + n) * 64 + lane];
#pragma unroll
          for (int h = 0; h < SR; ++h) {
            if (on[h]) {
              // t outermost: consecutive MFMAs hit different accumulators; per accumulator the order stays t ascending
#pragma unroll
              for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
                  acc[h][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[uu][h][t], b[n][t], acc[h][n], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  };

  f32x4 avA[UB][SR], avB[UB][SR];
  if (S > 0) issue(0, 0, avA);
  for (int s = 0; s < S; s += 2) {
    __syncthreads();  // stage s has landed (the compiler drains vmcnt here); nobody reads buffer 1 any more
    if (s + 1 < S) issue(s + 1, 1, avB);
    compute(s, 0, avA);
    if (s + 1 >= S) break;
    __syncthreads();
    if (s + 2 < S) issue(s + 2, 0, avA);
    compute(s + 1, 1, avB);
  }

  // epilogue: C/D layout col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int h = 0; h < SR; ++h) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = s_rows[h * LT_TM + 16 * wave + 4 * q + reg];
      if (row >= 0) {
        float* o = a.out + (int64_t)row * a.Cout + 16 * nt0 + r;
#pragma unroll
        for (int n = 0; n < NT; ++n) o[16 * n] = acc[h][n][reg];
      }
    }
  }
}

template <int NT, int UB>
static int launch_plan(const PlanArgs& a, hipStream_t stream) {
  const int64_t T = ((int64_t)a.Vout + LT_TM - 1) / LT_TM;
  const int64_t cb = a.Cout / (16 * NT);
  // two plan tiles per workgroup (half the weight traffic per flop) once the grid still fills the chip a few times over
  if (T * cb >= 2 * 4 * 256) {
    dim3 grid((unsigned)((T + 1) / 2), (unsigned)cb);
    LT_CONV_LAUNCH((k_spconv_plan<NT, UB, 2>), grid, dim3(256), stream, a);
  } else {
    dim3 grid((unsigned)T, (unsigned)cb);
    LT_CONV_LAUNCH((k_spconv_plan<NT, UB, 1>), grid, dim3(256), stream, a);
  }
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// =====================================================================================================
// Register-fed planned kernel (no LDS operands, no barriers in the main loop).
//
// A workgroup owns SR 16-row strips (SR/4 plan tiles); its W waves own the SAME rows and different 16*NTW-column groups,
// so every wave of a workgroup does the same amount of work for every (offset, strip) -- there is no intra-workgroup
// imbalance to wait for, and a strip without a neighbour at offset k is skipped by all waves alike.
// Operands go global/L2 -> VGPR directly: the packed weight fragment of (k, 16-channel group j, column tile) is one
// coalesced 1-KiB load, the gathered A rows are 16 x 64-B pieces.  The next slab's operands are loaded into a second
// register set while the current slab's MFMAs run; latency beyond that is covered by the other resident waves.
// Summation order per output element is unchanged (k ascending, channel ascending) -> bitwise equal to the others.
//
// SK (split-k) flavour: the W waves own the same rows AND the same 16*NTW columns and share the slab list (wave w takes
// slabs w, w+W, ...); their partial accumulators are added through LDS in wave order at the end.  Same parallelism as
// W column groups, but a workgroup W times taller for the same chain length: 1/W of the weight traffic per flop, and
// (SR+NTW)/(SR*NTW) operand loads per MFMA instead of (SR/W+NTW)/(SR/W*NTW).  Summation order differs from the table
// kernel's single chain (W sub-chains, then a fixed-order sum): deterministic, but not bitwise equal to k_spconv_vec.
template <int SR, int NTW, int W, bool SK = false>
__global__ __launch_bounds__(64 * W) void k_spconv_rf(PlanArgs a) {
  constexpr int TPW = SR >= 4 ? SR / 4 : 1;    // plan tiles per workgroup
  constexpr int PARTS = SR >= 4 ? 1 : 4 / SR;  // workgroups per plan tile (SR < 4: a workgroup owns SR of its 4 strips)
  constexpr int TMW = 16 * SR;                 // rows per workgroup
  __shared__ int s_nbr[LT_KMAX * TMW]; // [nact][TMW] neighbour rows of the active offsets
  __shared__ int s_rows[TMW];
  __shared__ int s_klist[LT_KMAX];
  __shared__ unsigned int s_act[LT_KMAX];  // per active offset: bit h = strip h has a neighbour
  __shared__ int s_nact;
  __shared__ f32x4 s_red[SK ? W * NTW * 64 : 1];  // SK: [W][NTW][64] partial accumulators of one strip

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0, tI = 0, tC = 0;
  if (a.dbg & 4) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  // workgroups that own at most one plan tile are dispatched heaviest tile first
  // (bx, by) = (row workgroup, column group).  Split-k launches a 1-D grid and maps it XCD-aware: hardware hands
  // consecutive workgroups to the 8 XCDs round-robin, so workgroup L runs on XCD L % 8; the column groups of one row
  // tile (same gathered rows) are made consecutive ON ONE XCD, so its L2 serves the gather to all of them.
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if constexpr (SK) {
    const unsigned gy = a.Cout / (16 * NTW);
    if (a.swz) {
      const unsigned L = blockIdx.x, i = L >> 3;
      bx = (i / gy) * 8 + (L & 7);
      by = i % gy;
    } else {
      bx = blockIdx.x % a.gx;
      by = blockIdx.x / a.gx;
    }
    if (bx >= (unsigned)a.gx || by >= gy) return;  // padding of the 1-D grid
  }
  const int64_t tile0 = TPW == 1 ? (int64_t)a.torder[bx / PARTS] : (int64_t)bx * TPW;
  const int s0 = PARTS == 1 ? 0 : (int)(bx % PARTS) * SR;  // first strip inside the plan tile
  const int64_t T = ((int64_t)a.Vout + LT_TM - 1) / LT_TM;
  const int J = a.Cin / 16, NTt = a.Cout / 16;
  const int nt0 = SK ? by * NTW : (by * W + wave) * NTW;

  if (wave == 0) {
    unsigned int act = 0u;
#pragma unroll
    for (int h = 0; h < TPW; ++h) {
      unsigned long long m = 0ull;
      if (lane < a.K && tile0 + h < T) m = a.tmask[(tile0 + h) * a.K + lane];
#pragma unroll
      for (int st = 0; st < (SR >= 4 ? 4 : SR); ++st)
        if ((m >> (16 * (s0 + st))) & 0xFFFFull) act |= 1u << (4 * h + st);
    }
    const unsigned long long bal = __ballot(act != 0u);
    if (act != 0u) {
      const int pos = __popcll(bal & ((1ull << lane) - 1ull));
      s_klist[pos] = lane;
      s_act[pos] = act;
    }
    if (lane == 0) s_nact = __popcll(bal);
  }
  for (int i = tid; i < TMW; i += 64 * W) {
    const int64_t slot = tile0 * LT_TM + 16 * s0 + i;
    s_rows[i] = slot < a.Vout ? a.order[slot] : -1;
  }
  __syncthreads();
  const int nact = s_nact;
  for (int e = tid; e < nact * TMW; e += 64 * W) {
    const int ki = e / TMW, i = e - ki * TMW;
    const int h = i >> 6;
    int v = -1;
    if (tile0 + h < T) v = a.tnbr[((tile0 + h) * a.K + s_klist[ki]) * LT_TM + 16 * s0 + (i & 63)];
    if (v >= a.Vin) v = -1;  // defensive: never gather out of bounds
    s_nbr[e] = v;
  }
  __syncthreads();

  f32x4 acc[SR][NTW];
#pragma unroll
  for (int h = 0; h < SR; ++h)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[h][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int U = nact * J;  // slabs (ki, j), j fastest
  const float* in_q = a.in + 4 * q;
  const f32x4* wfrag = reinterpret_cast<const f32x4*>(a.packed) + (int64_t)nt0 * 64 + lane;

  // Loads are unconditional (a missing neighbour or an inactive strip reads row 0, the value is dropped at compute
  // time), so the main loop has no branch around a load and hipcc can keep the NEXT slab's loads in flight with a
  // counted s_waitcnt while the current slab's MFMAs run.
  auto issue = [&](int ki, int j, f32x4 (&av)[SR], f32x4 (&bv)[NTW], unsigned int& valid) {
    const int k = __builtin_amdgcn_readfirstlane(s_klist[ki]);
    const f32x4* wp = wfrag + ((a.dbg & 1) ? (int64_t)0 : ((int64_t)k * J + j) * NTt * 64);
#pragma unroll
    for (int n = 0; n < NTW; ++n) bv[n] = wp[n * 64];
    valid = 0u;
#pragma unroll
    for (int h = 0; h < SR; ++h) {
      const int src = s_nbr[ki * TMW + 16 * h + r];
      valid |= (src >= 0 ? 1u : 0u) << h;
      const int srow = (a.dbg & 2) ? r : (src < 0 ? 0 : src);
      av[h] = *reinterpret_cast<const f32x4*>(in_q + (int64_t)srow * a.Cin + ((a.dbg & 2) ? 0 : 16 * j));
    }
  };
  auto compute = [&](unsigned int act, const f32x4 (&av)[SR], const f32x4 (&bv)[NTW], unsigned int valid) {
#pragma unroll
    for (int h = 0; h < SR; ++h) {
      if (act & (1u << h)) {  // wave-uniform: the strip has a neighbour at this offset
        const bool ok = (valid >> h) & 1u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float at = ok ? av[h][t] : 0.f;
#pragma unroll
          for (int n = 0; n < NTW; ++n)
            acc[h][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, bv[n][t], acc[h][n], 0, 0, 0);
        }
      }
    }
  };
  auto act_of = [&](int ki) -> unsigned int { return __builtin_amdgcn_readfirstlane(s_act[ki]); };

  if (a.dbg & 4) st1 = __builtin_amdgcn_s_memtime();
  f32x4 avA[SR], avB[SR], bA[NTW], bB[NTW];
  unsigned int vA = 0u, vB = 0u;
  // this wave's slabs: first, first + step, ... (all of them unless SK)
  const int first = SK ? wave : 0, step = SK ? W : 1;
  const int Uw = U > first ? (U - first + step - 1) / step : 0;
  if (Uw > 0) {
    // slab u = (ki, j); slabs past the end are clamped to the last one and computed with an empty strip mask
    int ki = 0, jj = first, pos = first;
    while (jj >= J) { jj -= J; ++ki; }
    auto advance = [&]() {
      if (pos + step < U) {
        pos += step;
        jj += step;
#pragma unroll
        for (int c = 0; c < step; ++c)
          if (jj >= J) { jj -= J; ++ki; }
      }
    };
    issue(ki, jj, avA, bA, vA);
    if (a.dbg & 4) {
      // diagnostic loop: the same work with stamps around the issue and compute parts
      for (int u = 0; u < Uw; u += 2) {
        long long s0 = __builtin_amdgcn_s_memtime();
        const unsigned int act0 = act_of(ki);
        advance();
        issue(ki, jj, avB, bB, vB);
        long long s1 = __builtin_amdgcn_s_memtime();
        compute(act0, avA, bA, vA);
        long long s2 = __builtin_amdgcn_s_memtime();
        const unsigned int act1 = (u + 1 < Uw) ? act_of(ki) : 0u;
        advance();
        issue(ki, jj, avA, bA, vA);
        long long s3 = __builtin_amdgcn_s_memtime();
        compute(act1, avB, bB, vB);
        long long s4 = __builtin_amdgcn_s_memtime();
        tI += (s1 - s0) + (s3 - s2);
        tC += (s2 - s1) + (s4 - s3);
      }
    } else
    for (int u = 0; u < Uw; u += 2) {
      const unsigned int act0 = act_of(ki);
      advance();
      issue(ki, jj, avB, bB, vB);
      compute(act0, avA, bA, vA);
      const unsigned int act1 = (u + 1 < Uw) ? act_of(ki) : 0u;
      advance();
      issue(ki, jj, avA, bA, vA);
      compute(act1, avB, bB, vB);
    }
  }

  if (a.dbg & 4) st2 = __builtin_amdgcn_s_memtime();
  auto stamp_end = [&]() {
    if ((a.dbg & 4) && a.stamps && lane == 0) {
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      long long* o = a.stamps + ((long long)blockIdx.x * W + wave) * 8;
      o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime();
      o[4] = rt0 | (tI << 40); o[5] = __builtin_amdgcn_s_memrealtime() | (tC << 40);
      o[6] = (long long)U | ((long long)nact << 32);
      o[7] = (long long)hwid | ((long long)xcc << 32);
    }
  };
  if constexpr (SK) {
    // strip h is finished by wave h % W: partial accumulators of the others through LDS, added in wave order; the
    // finishing wave stores the rows and keeps their column sums
    float s1[NTW], s2[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) s1[n] = s2[n] = 0.f;
#pragma unroll
    for (int h = 0; h < SR; ++h) {
      const int owner = h % W;
      f32x4* red = s_red;
      if (h > 0) __syncthreads();  // the previous strip's partials have been read
      if (wave != owner) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) red[(wave * NTW + n) * 64 + lane] = acc[h][n];
      }
      __syncthreads();
      if (wave == owner) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          f32x4 sum = owner == 0 ? acc[h][n] : red[n * 64 + lane];
#pragma unroll
          for (int w = 1; w < W; ++w) {
            const f32x4 v = w == owner ? acc[h][n] : red[(w * NTW + n) * 64 + lane];
            sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
          }
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = s_rows[16 * h + 4 * q + reg];
            if (row >= 0) a.out[(int64_t)row * a.Cout + 16 * (nt0 + n) + r] = sum[reg];
            s1[n] += sum[reg];
            s2[n] += sum[reg] * sum[reg];
          }
        }
      }
    }
    if (a.partial) {
      // column sums of the workgroup's rows: per wave over its strips, then over the waves in wave order through LDS
      float* redf = reinterpret_cast<float*>(s_red);  // [W][NTW][2][16]
      __syncthreads();                                // the last strip's partial accumulators have been read
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        float t1 = s1[n], t2 = s2[n];
        t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
        t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
        if (q == 0) {
          redf[((wave * NTW + n) * 2 + 0) * 16 + r] = t1;
          redf[((wave * NTW + n) * 2 + 1) * 16 + r] = t2;
        }
      }
      __syncthreads();
      if (wave == 0 && lane < 32) {
        const int which = lane >> 4;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < W; ++w) t += redf[((w * NTW + n) * 2 + which) * 16 + r];
          a.partial[((int64_t)bx * 2 + which) * a.Cout + 16 * (nt0 + n) + r] = t;
        }
      }
    }
    stamp_end();
    return;
  }

#pragma unroll
  for (int h = 0; h < SR; ++h) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = s_rows[16 * h + 4 * q + reg];
      if (row >= 0) {
        float* o = a.out + (int64_t)row * a.Cout + 16 * nt0 + r;
#pragma unroll
        for (int n = 0; n < NTW; ++n) o[16 * n] = acc[h][n][reg];
      }
    }
  }
  // BatchNorm statistics of the rows this workgroup produced (rows past Vout hold exact zeros): the conv output is
  // not read again for them.  Fixed summation order -> deterministic.
  if (a.partial) {
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int h = 0; h < SR; ++h)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const float v = acc[h][n][reg];
          s1 += v;
          s2 += v * v;
        }
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (q == 0) {
        float* p = a.partial + (int64_t)blockIdx.x * 2 * a.Cout + 16 * (nt0 + n) + r;
        p[0] = s1;
        p[a.Cout] = s2;
      }
    }
  }
}

template <int SR, int NTW, int W>
static void launch_rf_one(const PlanArgs& a, int gy, hipStream_t stream) {
  unsigned gx;
  if (SR >= 4) {
    const int64_t rows = 16 * SR;
    gx = (unsigned)((a.Vout + rows - 1) / rows);
  } else {
    gx = (unsigned)(((a.Vout + LT_TM - 1) / LT_TM) * (4 / SR));
  }
  LT_CONV_LAUNCH((k_spconv_rf<SR, NTW, W>), dim3(gx, (unsigned)gy), dim3(64 * W), stream, a);
}

template <int SR, int NTW, int W>
static void launch_sk_one(const PlanArgs& a, hipStream_t stream) {
  unsigned gx;
  if (SR >= 4) gx = (unsigned)((a.Vout + 16 * SR - 1) / (16 * SR));
  else gx = (unsigned)(((a.Vout + LT_TM - 1) / LT_TM) * (4 / SR));
  PlanArgs b = a;
  b.gx = (int)gx;
  const unsigned gy = (unsigned)(a.Cout / (16 * NTW));
  const unsigned total = ((gx + 7) / 8) * 8 * gy;
  if (b.stamps && g_conv_stamps_cap < (long long)total * W * 8) b.stamps = nullptr;
  LT_CONV_LAUNCH((k_spconv_rf<SR, NTW, W, true>), dim3(total), dim3(64 * W), stream, b);
}

// split-k launch: sr in {1,2,4}, ntw in {1,2,3,4,6,8}, w in {2,4,8}
static int launch_sk(const PlanArgs& a, int sr, int ntw, int w, hipStream_t stream) {
#define LT_SK_W(SRv, NTWv)                                          \
  do {                                                              \
    if (w == 8) launch_sk_one<SRv, NTWv, 8>(a, stream);             \
    else if (w == 4) launch_sk_one<SRv, NTWv, 4>(a, stream);        \
    else launch_sk_one<SRv, NTWv, 2>(a, stream);                    \
  } while (0)
#define LT_SK_NTW(SRv)                                              \
  do {                                                              \
    switch (ntw) {                                                  \
      case 8: LT_SK_W(SRv, 8); break;                               \
      case 6: LT_SK_W(SRv, 6); break;                               \
      case 4: LT_SK_W(SRv, 4); break;                               \
      case 3: LT_SK_W(SRv, 3); break;                               \
      case 2: LT_SK_W(SRv, 2); break;                               \
      default: LT_SK_W(SRv, 1); break;                              \
    }                                                               \
  } while (0)
  if (sr == 4) LT_SK_NTW(4);
  else if (sr == 2) LT_SK_NTW(2);
  else LT_SK_NTW(1);
#undef LT_SK_NTW
#undef LT_SK_W
  LT_LAUNCH_CHECK();
  return LT_OK;
}

template <int NTW>
static int launch_rf(const PlanArgs& a, int sr, hipStream_t stream) {
  const int groups = a.Cout / (16 * NTW);  // column groups of one wave each
  int w = 1;
  for (int c = 4; c >= 1; --c)
    if (groups % c == 0) { w = c; break; }
  const int gy = groups / w;
#define LT_RF(SRv)                                                                  \
  do {                                                                              \
    if (w == 4) launch_rf_one<SRv, NTW, 4>(a, gy, stream);                          \
    else if (w == 3) launch_rf_one<SRv, NTW, 3>(a, gy, stream);                     \
    else if (w == 2) launch_rf_one<SRv, NTW, 2>(a, gy, stream);                     \
    else launch_rf_one<SRv, NTW, 1>(a, gy, stream);                                 \
  } while (0)
  if (sr == 8) LT_RF(8);
  else if (sr == 4) LT_RF(4);
  else if (sr == 2) LT_RF(2);
  else LT_RF(1);
#undef LT_RF
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// =====================================================================================================
// Split-j kernel: the split-k kernel above with the slab list cut along the channel groups instead of flat.
//
// Same decomposition (W waves own the same SR strips x NTW column tiles, partial sums added in wave order), but wave w
// takes the channel groups j = w, w+W, ... of EVERY active offset (needs J % W == 0).  For such shapes the assignment and
// the order of every partial sum are those of k_spconv_rf<.., SK> -> bitwise equal results.  What changes is the cost of
// getting from one slab to the next, which the in-kernel stamps (tools/conv_stamps.py) showed to be ~2000 wave-cycles
// per slab against 1024 MFMA-pipe cycles in the flat loop (three dependent LDS round trips, ~45 scalar and ~25 vector
// instructions of index arithmetic per slab):
//  * everything that depends on the offset only (kernel offset k, strip activity, the SR gathered row pointers) is
//    computed once per offset and kept in registers; inside an offset the next slab is a constant stride away for both
//    operands (A: +16*W floats in the row, B: +W*NTt KiB in the packed image);
//  * the per-offset table (k, activity) lives in one VGPR (lane ki = entry ki) and is read with v_readlane, not LDS;
//  * the neighbour rows of the NEXT offset are read from LDS one offset ahead;
//  * operands come through raw buffer loads (scalar slab offsets); a missing neighbour is an out-of-range offset that
//    the hardware answers with zeros, instead of a v_cndmask at compute time.

template <int SR, int NTW, int W>
__global__ __launch_bounds__(64 * W) void k_spconv_sj(PlanArgs a) {
#include "spconv_sj_body.inc"
}
// The same code under its own name for the K = 1 launches (the network's dense layers over lt_kmap_plan_identity: 1x1x1
// shortcut convs, point-branch Linears).  A kernel trace then tells the HBM-bound dense products from the sparse convs
// (bench.py's roofline family and the rocprofv3 summaries under profiles/ count k_spconv_* only).
template <int SR, int NTW, int W>
__global__ __launch_bounds__(64 * W) void k_dense_sj(PlanArgs a) {
#include "spconv_sj_body.inc"
}

template <int SR, int NTW, int W>
static void launch_sj_one(const PlanArgs& a, hipStream_t stream) {
  const unsigned gx = (unsigned)(((a.Vout + LT_TM - 1) / LT_TM) * (SR >= 4 ? 1 : 4 / SR));
  PlanArgs b = a;
  b.gx = (int)gx;
  const unsigned gy = (unsigned)(a.Cout / (16 * NTW));
  const unsigned total = ((gx + 7) / 8) * 8 * gy;
  if (b.stamps && g_conv_stamps_cap < (long long)total * W * 8) b.stamps = nullptr;
  if (b.K == 1) LT_CONV_LAUNCH((k_dense_sj<SR, NTW, W>), dim3(total), dim3(64 * W), stream, b);
  else LT_CONV_LAUNCH((k_spconv_sj<SR, NTW, W>), dim3(total), dim3(64 * W), stream, b);
}

// split-j launch: sr in {1,2,4}, ntw in {2,3,4}, w in {2,3,4,6,8}; needs (Cin/16) % w == 0
static bool sj_supported(int Cin, int sr, int ntw, int w) {
  if (ntw == 6) return (sr == 1 || sr == 2) && (w == 2 || w == 3) && (Cin / 16) % w == 0;  // 96 columns per wave
  return (sr == 1 || sr == 2 || sr == 4) && (ntw == 2 || ntw == 3 || ntw == 4) &&
         (w == 2 || w == 3 || w == 4 || w == 6 || w == 8) && (Cin / 16) % w == 0;
}
static int launch_sj(const PlanArgs& a, int sr, int ntw, int w, hipStream_t stream) {
#define LT_SJ_W(SRv, NTWv)                                          \
  do {                                                              \
    switch (w) {                                                    \
      case 8: launch_sj_one<SRv, NTWv, 8>(a, stream); break;        \
      case 6: launch_sj_one<SRv, NTWv, 6>(a, stream); break;        \
      case 4: launch_sj_one<SRv, NTWv, 4>(a, stream); break;        \
      case 3: launch_sj_one<SRv, NTWv, 3>(a, stream); break;        \
      default: launch_sj_one<SRv, NTWv, 2>(a, stream); break;       \
    }                                                               \
  } while (0)
#define LT_SJ_NTW(SRv)                                              \
  do {                                                              \
    if (ntw == 4) LT_SJ_W(SRv, 4);                                  \
    else if (ntw == 3) LT_SJ_W(SRv, 3);                             \
    else LT_SJ_W(SRv, 2);                                           \
  } while (0)
  if (ntw == 6) {  // one column group of 96: every gathered row is read once
    if (sr == 2) { if (w == 3) launch_sj_one<2, 6, 3>(a, stream); else launch_sj_one<2, 6, 2>(a, stream); }
    else { if (w == 3) launch_sj_one<1, 6, 3>(a, stream); else launch_sj_one<1, 6, 2>(a, stream); }
  } else if (sr == 4) LT_SJ_NTW(4);
  else if (sr == 2) LT_SJ_NTW(2);
  else LT_SJ_NTW(1);
#undef LT_SJ_NTW
#undef LT_SJ_W
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// =====================================================================================================
// bf16x3 split-precision kernel (opt-in: lt_spconv_plan_fwd_v variant 3000 + 100*SR + 10*NTW + W, pack mode 4).
//
// Same decomposition as k_spconv_sj (output-stationary, W waves split the channel groups of every active offset, partial
// sums added in wave order), but the contraction runs on the bf16 matrix pipe: every fp32 operand is split into
// hi = bf16(x) and lo = bf16(x - hi), and a product is hi*hi + hi*lo + lo*hi accumulated in fp32 by the MFMA (the dropped
// lo*lo term and the rounding of lo are ~2^-17 relative per product: features agree with the fp32 kernel to ~1e-5, two
// orders inside the 1e-3 bar, but NOT bit for bit).  One v_mfma_f32_16x16x32_bf16 (16 cycles) contracts 32 channels where
// the fp32 kernel needs eight v_mfma_f32_16x16x4_f32 (256 cycles): with three products per tile the matrix pipe has 5.3x
// less to do and the kernel becomes bound by operand delivery.
//  * unit of work: a channel PAIR-group jp (32 channels).  A: lane (r, q) loads 32 contiguous bytes of row nbr[k][r] at
//    channel 32 jp + 8 q (MFMA k index 8 q + e <-> channel 32 jp + 8 q + e) and splits them in registers (v_cvt_pk_bf16_f32);
//  * B: split once at pack time (mode 4): per (k, jp, n-tile) 2 KiB = the hi operand image (64 lanes x 16 B) followed by the
//    lo image -- the same bytes per weight as fp32, two 16-B loads per tile, no conversion in the kernel.
// Needs Cin % 32 == 0 and (Cin / 32) % W == 0.
typedef __bf16 lt_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_pack_weights_bf16(const float* __restrict__ W, int K, int Cin, int Cout, int mode,
                                    unsigned short* __restrict__ packed) {
  // mode 4: E = Wk; 5: Wk^T; 6: W(K-1-k)^T  (modes 0, 1, 2 of the fp32 image)
  const int m = mode - 4;
  const int Ce_in = m == 0 ? Cin : Cout;
  const int Ce_out = m == 0 ? Cout : Cin;
  const int NTt = Ce_out / 16, JP = Ce_in / 32;
  const int64_t total = (int64_t)K * Ce_in * Ce_out;  // (hi, lo) pairs
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = p;
    const int e = (int)(r & 7); r >>= 3;
    const int lane = (int)(r & 63); r >>= 6;
    const int nt = (int)(r % NTt); r /= NTt;
    const int jp = (int)(r % JP);
    const int k = (int)(r / JP);
    const int ci = 32 * jp + 8 * (lane >> 4) + e;
    const int co = 16 * nt + (lane & 15);
    float v;
    if (m == 0) v = W[((int64_t)k * Cin + ci) * Cout + co];
    else {
      const int ks = m == 2 ? (K - 1 - k) : k;
      v = W[((int64_t)ks * Cin + co) * Cout + ci];
    }
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    const int64_t base = ((((int64_t)k * JP + jp) * NTt + nt) * 2) * 512 + lane * 8 + e;   // in bf16 units
    packed[base] = __builtin_bit_cast(unsigned short, hi);
    packed[base + 512] = __builtin_bit_cast(unsigned short, lo);
  }
}

// 8 fp32 -> (hi, lo) bf16x8 operands
__device__ __forceinline__ void lt_split_bf16(const f32x4& x0, const f32x4& x1, lt_bf16x8& hi, lt_bf16x8& lo) {
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const f32x8 x = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  hi = __builtin_convertvector(x, lt_bf16x8);
  const u32x4 hb = __builtin_bit_cast(u32x4, hi);
  f32x8 back;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    back[2 * i] = __uint_as_float(hb[i] << 16);
    back[2 * i + 1] = __uint_as_float(hb[i] & 0xFFFF0000u);
  }
  lo = __builtin_convertvector(x - back, lt_bf16x8);
}

template <int SR, int NTW, int W>
__global__ __launch_bounds__(64 * W) void k_spconv_bf(PlanArgs a) {
  constexpr int PARTS = SR >= 4 ? 1 : 4 / SR;
  constexpr int TMW = 16 * SR;
  __shared__ int s_nbr[LT_KMAX * TMW];
  __shared__ int s_rows[TMW];
  __shared__ int s_klist[LT_KMAX];
  __shared__ unsigned int s_act[LT_KMAX];
  __shared__ int s_nact;
  __shared__ f32x4 s_red[W * NTW * 64];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const unsigned gy = a.Cout / (16 * NTW);
  unsigned bx, by;
  {
    const unsigned L = blockIdx.x, i = L >> 3;   // XCD-aware mapping (see k_spconv_rf)
    bx = (i / gy) * 8 + (L & 7);
    by = i % gy;
  }
  if (bx >= (unsigned)a.gx || by >= gy) return;
  const int64_t tile0 = (int64_t)a.torder[bx / PARTS];
  const int s0 = PARTS == 1 ? 0 : (int)(bx % PARTS) * SR;
  const int JP = a.Cin / 32, NTt = a.Cout / 16;
  const int JPW = JP / W;
  const int nt0 = by * NTW;

  if (wave == 0) {
    unsigned int act = 0u;
    unsigned long long m = 0ull;
    if (lane < a.K) m = a.tmask[tile0 * a.K + lane];
#pragma unroll
    for (int st = 0; st < SR; ++st)
      if ((m >> (16 * (s0 + st))) & 0xFFFFull) act |= 1u << st;
    const unsigned long long bal = __ballot(act != 0u);
    if (act != 0u) {
      const int pos = __popcll(bal & ((1ull << lane) - 1ull));
      s_klist[pos] = lane;
      s_act[pos] = act;
    }
    if (lane == 0) s_nact = __popcll(bal);
  }
  for (int i = tid; i < TMW; i += 64 * W) {
    const int64_t slot = tile0 * LT_TM + 16 * s0 + i;
    s_rows[i] = slot < a.Vout ? a.order[slot] : -1;
  }
  __syncthreads();
  const int nact = s_nact;
  for (int e = tid; e < nact * TMW; e += 64 * W) {
    const int ki = e / TMW, i = e - ki * TMW;
    int v = a.tnbr[(tile0 * a.K + s_klist[ki]) * LT_TM + 16 * s0 + i];
    if (v >= a.Vin) v = -1;
    s_nbr[e] = v;
  }
  __syncthreads();

  f32x4 acc[SR][NTW];
#pragma unroll
  for (int h = 0; h < SR; ++h)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[h][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0,
                                                                        (int)((int64_t)a.Vin * a.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0,
                                                                        (int)((int64_t)a.K * a.Cin * a.Cout * 4), 0x00020000);
  const int lane_a = (8 * q + 32 * wave) * 4;       // bytes into a gathered row: this wave's first pair-group
  const int lane_b = lane * 16;
  const int row_bytes = a.Cin * 4;
  const int wstep = W * NTt * 2048;                 // pair-group jp -> jp + W in the packed image
  const int wkstep = JP * NTt * 2048;               // offset k -> k + 1
  const int wbase = (wave * NTt + nt0) * 2048;

  const int U = nact * JPW;                         // units (offset, pair-group) of this wave
  if (U > 0) {
    // unit u = (ki, jj): operands are loaded one unit ahead of the MFMAs
    f32x4 a0[SR], a1[SR];
    u32x4 bh[NTW], bl[NTW];
    unsigned int act_nxt = 0u;
    auto load_unit = [&](int u) {
      const int ki = u / JPW, jj = u - ki * JPW;
      const int k = s_klist[ki];
      act_nxt = s_act[ki];
      const int so_a = jj * 128 * W;
      const int so_b = wbase + k * wkstep + jj * wstep;
#pragma unroll
      for (int h = 0; h < SR; ++h) {
        const int nb = s_nbr[ki * TMW + 16 * h + r];
        const int vo = nb >= 0 ? nb * row_bytes + lane_a : 0x7FFFF000;   // past the end: the hardware returns zeros
        a0[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, so_a, 0));
        a1[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, so_a + 16, 0));
      }
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        bh[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, lane_b + n * 2048, so_b, 0));
        bl[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, lane_b + n * 2048 + 1024, so_b, 0));
      }
    };
    load_unit(0);
    for (int u = 0; u < U; ++u) {
      // operands of unit u into MFMA form, then the next unit's loads go out under this unit's MFMAs
      const unsigned int act = act_nxt;
      lt_bf16x8 ah[SR], al[SR], wh[NTW], wl[NTW];
#pragma unroll
      for (int h = 0; h < SR; ++h) lt_split_bf16(a0[h], a1[h], ah[h], al[h]);
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        wh[n] = __builtin_bit_cast(lt_bf16x8, bh[n]);
        wl[n] = __builtin_bit_cast(lt_bf16x8, bl[n]);
      }
      if (u + 1 < U) load_unit(u + 1);
#pragma unroll
      for (int h = 0; h < SR; ++h) {
        if ((act >> h) & 1u) {   // wave-uniform
#pragma unroll
          for (int n = 0; n < NTW; ++n) acc[h][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[h], wh[n], acc[h][n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < NTW; ++n) acc[h][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[h], wl[n], acc[h][n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < NTW; ++n) acc[h][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[h], wh[n], acc[h][n], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: as k_spconv_sj (strip h finished by wave h % W, partials through LDS in wave order, column sums)
  float s1[NTW], s2[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) s1[n] = s2[n] = 0.f;
#pragma unroll
  for (int h = 0; h < SR; ++h) {
    const int owner = h % W;
    f32x4* red = s_red;
    if (W > 1) {
      if (h > 0) __syncthreads();
      if (wave != owner) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) red[(wave * NTW + n) * 64 + lane] = acc[h][n];
      }
      __syncthreads();
    }
    if (wave == owner) {
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        f32x4 sum = (owner == 0 || W == 1) ? acc[h][n] : red[n * 64 + lane];
#pragma unroll
        for (int w = 1; w < W; ++w) {
          const f32x4 v = w == owner ? acc[h][n] : red[(w * NTW + n) * 64 + lane];
          sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = s_rows[16 * h + 4 * q + reg];
          if (row >= 0) a.out[(int64_t)row * a.Cout + 16 * (nt0 + n) + r] = sum[reg];
          s1[n] += sum[reg];
          s2[n] += sum[reg] * sum[reg];
        }
      }
    }
  }
  if (a.partial) {
    float* redf = reinterpret_cast<float*>(s_red);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      float t1 = s1[n], t2 = s2[n];
      t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
      t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
      if (q == 0) {
        redf[((wave * NTW + n) * 2 + 0) * 16 + r] = t1;
        redf[((wave * NTW + n) * 2 + 1) * 16 + r] = t2;
      }
    }
    __syncthreads();
    if (wave == 0 && lane < 32) {
      const int which = lane >> 4;
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) t += redf[((w * NTW + n) * 2 + which) * 16 + r];
        a.partial[((int64_t)bx * 2 + which) * a.Cout + 16 * (nt0 + n) + r] = t;
      }
    }
  }
}

template <int SR, int NTW, int W>
static void launch_bf_one(const PlanArgs& a, hipStream_t stream) {
  const unsigned gx = (unsigned)(((a.Vout + LT_TM - 1) / LT_TM) * (SR >= 4 ? 1 : 4 / SR));
  PlanArgs b = a;
  b.gx = (int)gx;
  const unsigned gy = (unsigned)(a.Cout / (16 * NTW));
  const unsigned total = ((gx + 7) / 8) * 8 * gy;
  LT_CONV_LAUNCH((k_spconv_bf<SR, NTW, W>), dim3(total), dim3(64 * W), stream, b);
}

static bool bf_supported(int Cin, int Cout, int sr, int ntw, int w) {
  return Cin % 32 == 0 && (sr == 1 || sr == 2 || sr == 4) && (ntw == 2 || ntw == 3 || ntw == 4) && (Cout / 16) % ntw == 0 &&
         (w == 1 || w == 2 || w == 3 || w == 4) && (Cin / 32) % w == 0;
}
static int launch_bf(const PlanArgs& a, int sr, int ntw, int w, hipStream_t stream) {
#define LT_BF_W(SRv, NTWv)                                          \
  do {                                                              \
    switch (w) {                                                    \
      case 4: launch_bf_one<SRv, NTWv, 4>(a, stream); break;        \
      case 3: launch_bf_one<SRv, NTWv, 3>(a, stream); break;        \
      case 2: launch_bf_one<SRv, NTWv, 2>(a, stream); break;        \
      default: launch_bf_one<SRv, NTWv, 1>(a, stream); break;       \
    }                                                               \
  } while (0)
#define LT_BF_NTW(SRv)                                              \
  do {                                                              \
    if (ntw == 4) LT_BF_W(SRv, 4);                                  \
    else if (ntw == 3) LT_BF_W(SRv, 3);                             \
    else LT_BF_W(SRv, 2);                                           \
  } while (0)
  if (sr == 4) LT_BF_NTW(4);
  else if (sr == 2) LT_BF_NTW(2);
  else LT_BF_NTW(1);
#undef LT_BF_NTW
#undef LT_BF_W
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Conv launch timer (bench.py's roofline leg): while armed, every sparse-conv kernel launch of this library is bracketed
// by two hipEvents on the stream the kernel is launched on (attached to the dispatch itself, hipExtLaunchKernelGGL).  Process-global, one caller thread (as everywhere).
struct LtConvTimer {
  hipEvent_t* ev = nullptr;  // 2 * cap
  int32_t* meta = nullptr;   // 4 * cap: Cin, Cout, K, Vout
  int cap = 0, n = 0, dropped = 0;
  std::atomic<bool> armed{false};
  std::mutex mu;             // slot hand-out and begin / end: conv launches may come from several threads
};
static LtConvTimer g_timer;

struct LtConvTimerScope {
  int slot = -1;
  hipStream_t stream;
  LtConvTimerScope(hipStream_t s, int Cin, int Cout, int K, int64_t Vout) : stream(s) {
    if (!g_timer.armed.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lock(g_timer.mu);
    if (!g_timer.armed.load()) return;
    if (g_timer.n >= g_timer.cap) { ++g_timer.dropped; return; }
    slot = g_timer.n++;
    int32_t* m = g_timer.meta + 4 * slot;
    m[0] = Cin; m[1] = Cout; m[2] = K; m[3] = (int32_t)Vout;
    g_conv_ev0 = g_timer.ev[2 * slot];
    g_conv_ev1 = g_timer.ev[2 * slot + 1];
  }
  ~LtConvTimerScope() { g_conv_ev0 = g_conv_ev1 = nullptr; }
};

extern "C" int lt_conv_timer_begin(int32_t capacity) {
  LT_REQUIRE(capacity > 0 && capacity <= (1 << 20), LT_ERR_ARG, "lt_conv_timer_begin: capacity %d not in [1, 2^20]", capacity);
  std::lock_guard<std::mutex> lock(g_timer.mu);
  if (capacity > g_timer.cap) {
    hipEvent_t* ev = (hipEvent_t*)realloc(g_timer.ev, sizeof(hipEvent_t) * 2 * capacity);
    int32_t* meta = (int32_t*)realloc(g_timer.meta, sizeof(int32_t) * 4 * capacity);
    LT_REQUIRE(ev && meta, LT_ERR_ARG, "lt_conv_timer_begin: out of host memory");
    g_timer.ev = ev;
    g_timer.meta = meta;
    for (int i = 2 * g_timer.cap; i < 2 * capacity; ++i) LT_CHECK_HIP(hipEventCreate(&g_timer.ev[i]));
    g_timer.cap = capacity;
  }
  g_timer.n = 0;
  g_timer.dropped = 0;
  g_timer.armed = true;
  return LT_OK;
}

extern "C" int lt_conv_timer_arm(int32_t on) {
  LT_REQUIRE(g_timer.cap > 0, LT_ERR_ARG, "lt_conv_timer_arm: call lt_conv_timer_begin first");
  g_timer.armed = on != 0;
  return LT_OK;
}

extern "C" int64_t lt_conv_timer_end(float* ms, int32_t* meta, int64_t capacity) {
  g_timer.armed = false;
  std::lock_guard<std::mutex> lock(g_timer.mu);
  LT_REQUIRE(g_timer.dropped == 0, LT_ERR_RANGE, "lt_conv_timer_end: %d launches beyond the capacity given to lt_conv_timer_begin",
             g_timer.dropped);
  const int n = g_timer.n;
  LT_REQUIRE(capacity >= n && (n == 0 || (ms && meta)), LT_ERR_ARG, "lt_conv_timer_end: room for %lld launches, %d recorded",
             (long long)capacity, n);
  for (int i = 0; i < n; ++i) {
    LT_CHECK_HIP(hipEventSynchronize(g_timer.ev[2 * i + 1]));
    LT_CHECK_HIP(hipEventElapsedTime(&ms[i], g_timer.ev[2 * i], g_timer.ev[2 * i + 1]));
    for (int j = 0; j < 4; ++j) meta[4 * i + j] = g_timer.meta[4 * i + j];
  }
  g_timer.n = 0;
  return n;
}

// Default kernel shape (measured on every level / width of the U-Net: tools/conv_bench.py + tools/cb_sj.sh):
// 16-row workgroups (SR = 1: the finest work items -> the shortest tail, and no strip of a workgroup is ever inactive),
// 4 column tiles per wave (3 for Cout = 48, 96, ...), and the channel groups of an offset split over W = 2 waves, or 4
// when 2 would leave fewer than 7000 waves or the input is 256 channels or wider.  The split-j kernel runs it whenever W
// divides Cin/16; other widths take the flat split-k loop with the same shape.
// Rows per workgroup: 16 (one strip), or 32 (two strips per wave: every weight fragment a wave loads feeds two strips) on
// levels with enough rows that the coarser work items still fill the chip many times over -- measured per level at the
// a2d2_kitti_128k sizes (tools/cb_a2d2.sh): 310 k rows, C=96: 377 -> 349 us; 141 k rows: 173 -> 164 us; below ~100 k rows
// the shorter tail of 16-row workgroups wins.  A function of Vout only: lt_spconv_stat_blocks must agree with it.
// Process-wide arithmetic of the planned conv kernels: 0 = fp32 MFMA (default; bit-reproducible fmaf chains), 1 = bf16x3 split
// precision (k_spconv_bf) for every conv whose effective Cin is a multiple of 32 and at least 64; the caller packs the
// weights of exactly those convs with mode 4..6 (lt_conv_uses_bf16 is the rule, for both sides).
static std::atomic<int> g_conv_precision{0};
static int64_t conv_sr2_rows() {  // rows from which two strips per wave pay (LATTE_DIAG_CONV_SR2_ROWS: A/B measurements)
  static const int64_t v = [] {
    const char* e = getenv("LATTE_DIAG_CONV_SR2_ROWS");
    return e ? (int64_t)atoll(e) : (int64_t)100000;
  }();
  return v;
}
static int conv_default_sr(int64_t Vout, int Cout) {
  (void)Cout;
  if (g_conv_precision) return 2;
  return Vout >= conv_sr2_rows() ? 2 : 1;
}
// ONE rule for the pack side (lt_conv_uses_bf16 -> functional._packed picks mode 4..6) and the launch side: the conv takes
// the bf16x3 kernel iff this says so.  It includes what the default bf16 shape needs -- a column-tile count per wave
// (4, 3 or 2) that divides Cout/16 -- so that Cout = 80 or 112 (5 or 7 tiles) stay on the fp32 kernels on BOTH sides.
static int bf_default_ntw(int Cout) {
  const int ntt = Cout / 16;
  return ntt % 4 == 0 ? 4 : ntt % 3 == 0 ? 3 : ntt % 2 == 0 ? 2 : 0;
}
static bool conv_uses_bf16(int Cin, int Cout) {
  return g_conv_precision == 1 && Cin >= 64 && Cin % 32 == 0 && Cout >= 32 && Cout % 16 == 0 && bf_default_ntw(Cout) != 0;
}
extern "C" int lt_conv_set_precision(int32_t mode) {
  LT_REQUIRE(mode == 0 || mode == 1, LT_ERR_ARG, "lt_conv_set_precision: mode %d not in {0 (fp32), 1 (bf16x3)}", mode);
  g_conv_precision = mode;
  return LT_OK;
}
extern "C" int lt_conv_uses_bf16(int32_t Cin, int32_t Cout) { return conv_uses_bf16(Cin, Cout) ? 1 : 0; }


struct LtConvShape {
  int sr, ntw, w;
  bool sj;
};
static bool sj_supported(int Cin, int sr, int ntw, int w);
static LtConvShape conv_default_shape(int64_t Vin, int64_t Vout, int Cin, int Cout, int K) {
  const int ntt = Cout / 16, J = Cin / 16;
  LtConvShape s;
  s.ntw = ntt % 4 == 0 ? 4 : ntt % 3 == 0 ? 3 : ntt % 2 == 0 ? 2 : 1;
  s.sr = conv_default_sr(Vout, Cout);
  const int64_t waves2 = ((Vout + 16 * s.sr - 1) / (16 * s.sr)) * (ntt / s.ntw) * 2;
  s.w = (waves2 >= 7000 && J < 16) ? 2 : 4;
  if (J % s.w != 0 && J % 2 == 0) s.w = 2;
  s.sj = s.ntw >= 2 && sj_supported(Cin, s.sr, s.ntw, s.w) && Vin * Cin * 4 < 0x7FFFF000ll &&
         (int64_t)K * Cin * Cout * 4 < 0x7FFFF000ll;
  if (!s.sj && s.w != 2 && s.w != 4 && s.w != 8) s.w = 4;
  return s;
}
static int64_t rf_grid_x(int64_t Vout, int sr) {
  if (sr >= 4) return (Vout + 16 * sr - 1) / (16 * sr);
  return ((Vout + LT_TM - 1) / LT_TM) * (4 / sr);
}

extern "C" int64_t lt_spconv_stat_blocks(int64_t Vout, int32_t Cout) {
  if (Vout <= 0 || Cout < 16) return 0;
  return rf_grid_x(Vout, conv_default_sr(Vout, Cout));
}

extern "C" int64_t lt_spconv_stats_bytes(int64_t Vout, int32_t Cout) {
  if (Cout < 16) return 256;
  return lt_stats_view(nullptr, lt_spconv_stat_blocks(Vout, Cout), Cout).bytes;
}

static int spconv_plan_impl(const float* in, const float* packed, const void* plan, float* out, int64_t Vin,
                            int64_t Vout, int32_t Cin, int32_t Cout, int32_t K, int32_t variant, float* partial,
                            int64_t* nb_out, bool allow_ls, hipStream_t stream);

extern "C" int lt_spconv_plan_fwd_v(const float* in, const float* packed, const void* plan, float* out, int64_t Vin,
                                    int64_t Vout, int32_t Cin, int32_t Cout, int32_t K, int32_t variant,
                                    hipStream_t stream) {
  return spconv_plan_impl(in, packed, plan, out, Vin, Vout, Cin, Cout, K, variant, nullptr, nullptr, true, stream);
}

// The public pair lt_spconv_plan_fwd_stats / lt_bn_finalize_sums agrees on the number of partial rows through
// lt_spconv_stat_blocks(Vout, Cout), a function of the output shape only -- so it stays on the general kernels.  The
// composites (fused.hip) use lt_spconv_plan_fwd_stats_nb, which may pick the LDS-staged kernel and REPORTS how many
// partial rows it wrote.
extern "C" int lt_spconv_plan_fwd_stats(const float* in, const float* packed, const void* plan, float* out, int64_t Vin,
                                        int64_t Vout, int32_t Cin, int32_t Cout, int32_t K, void* stats_ws,
                                        hipStream_t stream) {
  LT_REQUIRE(stats_ws, LT_ERR_ARG, "lt_spconv_plan_fwd_stats: stats_ws is null");
  return spconv_plan_impl(in, packed, plan, out, Vin, Vout, Cin, Cout, K, 0, (float*)stats_ws, nullptr, false, stream);
}

int lt_spconv_plan_fwd_stats_nb(const float* in, const float* packed, const void* plan, float* out, int64_t Vin, int64_t Vout,
                                int32_t Cin, int32_t Cout, int32_t K, void* stats_ws, int64_t* nb, hipStream_t stream) {
  LT_REQUIRE(stats_ws && nb, LT_ERR_ARG, "lt_spconv_plan_fwd_stats_nb: null pointer");
  return spconv_plan_impl(in, packed, plan, out, Vin, Vout, Cin, Cout, K, 0, (float*)stats_ws, nb, true, stream);
}

// Which convs take the LDS-staged kernel (spconv_ls.hip) by default: maps of a 2^3 kernel in the direction where every
// output row has ONE pair (more output rows than input rows: transposed conv forward, strided conv backward-input).
// Measured at the a2d2_kitti_128k sizes (tools/k8_bench.py, profiles/diag/r03_k8_bench.txt): alone it wins on the finest
// levels (96 -> 96 channels at 310 k rows 75 vs 97 us, 32 -> 32 19 vs 25 us) and loses up to 15 % on the coarse ones; inside
// the step, where the general kernel's operand loads compete with everything else for the address path, taking it for
// all of them is the fastest (31.63 -> 31.49 ms same box).  In the other direction (2-3 pairs per row, several offsets
// per tile) it re-stages the weights per (tile, offset) and loses by 2x: never.  LT_CONV_LS=0 switches it off (A/B).
static bool conv_ls_default(int64_t Vin, int64_t Vout, int Cin, int Cout, int K) {
  static const int mode = [] {
    const char* e = getenv("LT_CONV_LS");
    return e ? atoi(e) : 1;
  }();
  if (K == 1 && Vin == Vout && g_conv_precision == 0) {
    // dense layers over an identity plan (lt_kmap_plan_identity): ONE offset for the whole launch -- the weights are staged
    // once per workgroup and stay; LT_CONV_LS_DENSE=0 keeps them on the split-j kernel (A/B)
    static const int dense = [] {
      const char* e = getenv("LT_CONV_LS_DENSE");
      return e ? atoi(e) : 1;
    }();
    return dense != 0 && lt_spconv_ls_supported(Vin, Vout, Cin, Cout, K);
  }
  if (mode == 0 || K > 8 || K < 2 || Vout <= Vin || g_conv_precision != 0) return false;
  return lt_spconv_ls_supported(Vin, Vout, Cin, Cout, K);
}

static int spconv_plan_impl(const float* in, const float* packed, const void* plan, float* out, int64_t Vin,
                            int64_t Vout, int32_t Cin, int32_t Cout, int32_t K, int32_t variant, float* partial,
                            int64_t* nb_out, bool allow_ls, hipStream_t stream) {
  if (nb_out) *nb_out = lt_spconv_stat_blocks(Vout, Cout);
  if (variant == 5001) {   // A/B measurements: the general default kernel where the default would be the LDS-staged one
    variant = 0;
    allow_ls = false;
  }
  LT_REQUIRE(Vin >= 0 && Vout >= 0 && Vin < INT32_MAX && Vout < INT32_MAX, LT_ERR_RANGE, "lt_spconv_plan_fwd: bad row counts");
  LT_REQUIRE(K >= 1 && K <= LT_KMAX, LT_ERR_ARG, "lt_spconv_plan_fwd: kernel volume %d not in [1,%d]", K, LT_KMAX);
  LT_REQUIRE(Cin >= 16 && Cin % 16 == 0, LT_ERR_ARG, "lt_spconv_plan_fwd: Cin=%d must be a multiple of 16", Cin);
  LT_REQUIRE(Cout >= 16 && Cout % 16 == 0, LT_ERR_ARG, "lt_spconv_plan_fwd: Cout=%d must be a multiple of 16", Cout);
  if (Vout == 0) return LT_OK;
  LT_REQUIRE(packed && plan && out && (Vin == 0 || in), LT_ERR_ARG, "lt_spconv_plan_fwd: null pointer");
  LT_REQUIRE(((uintptr_t)plan & 15) == 0, LT_ERR_ARG, "lt_spconv_plan_fwd: plan must be 16-byte aligned");
  LtPlanView pv = lt_plan_view(const_cast<void*>(plan), K, Vout);
  PlanArgs a{in, packed, pv.tnbr, pv.tmask, pv.order, pv.torder, out, (int)Vin, (int)Vout, Cin, Cout, K, partial, 0, 1, 0, nullptr};
  LtConvTimerScope timed(stream, Cin, Cout, K, Vout);
  const int ntt = Cout / 16;
  LT_REQUIRE(!(partial && variant != 0 && variant != 5000), LT_ERR_ARG, "lt_spconv_plan_fwd: statistics need the default variant");
  if (conv_uses_bf16(Cin, Cout) && !(variant >= 3000 && variant < 4000)) {
    // The caller packed this conv's weights as the bf16 hi|lo image (same rule, lt_conv_uses_bf16): only the bf16x3 kernel
    // may read them.  What it cannot run is an error, never a silent fall-through to an fp32 kernel over a bf16 image.
    LT_REQUIRE(variant == 0, LT_ERR_ARG,
               "lt_spconv_plan_fwd_v: variant %d is an fp32 kernel, but under lt_conv_set_precision(1) the weights of a "
               "Cin=%d Cout=%d conv are packed as the bf16 image (use variant 0 or 3xxx, or precision 0)", variant, Cin, Cout);
    LT_REQUIRE(Vin * Cin * 4 < 0x7FFFF000ll && (int64_t)K * Cin * Cout * 4 < 0x7FFFF000ll, LT_ERR_RANGE,
               "lt_spconv_plan_fwd: bf16x3 kernel addresses its operands with 32-bit buffer offsets: Vin*Cin*4 = %lld and "
               "K*Cin*Cout*4 = %lld must stay below 2 GiB (use lt_conv_set_precision(0) for this size)",
               (long long)(Vin * Cin * 4), (long long)((int64_t)K * Cin * Cout * 4));
    // bf16x3: 32-row workgroups on every level, channel pair-groups over two waves when their count is even (measured per
    // level at the a2d2 sizes, tools/cb_bf16.py)
    const int jp = Cin / 32, ntw = bf_default_ntw(Cout), w = jp % 2 == 0 ? 2 : 1;
    LT_REQUIRE(bf_supported(Cin, Cout, 2, ntw, w), LT_ERR_ARG, "lt_spconv_plan_fwd: no bf16x3 shape for Cin=%d Cout=%d", Cin, Cout);
    return launch_bf(a, 2, ntw, w, stream);
  }
  if (variant > 5001 && variant < 5100) {   // 5010 + dbg bits: diagnostics of the LDS-staged kernel (wrong results)
    a.dbg = variant - 5010;
    if (a.dbg & 4) a.stamps = g_conv_stamps;   // the caller sized the buffer for [workgroups][8 waves][8] words
    variant = 5000;
  }
  if (variant == 5000 || (variant == 0 && allow_ls && conv_ls_default(Vin, Vout, Cin, Cout, K))) {
    LT_REQUIRE(lt_spconv_ls_supported(Vin, Vout, Cin, Cout, K), LT_ERR_ARG,
               "lt_spconv_plan_fwd_v: the LDS-staged kernel does not take Cin=%d Cout=%d K=%d at these sizes", Cin, Cout, K);
    if (nb_out) *nb_out = lt_spconv_ls_stat_blocks(Vout, Cin, Cout, K);
    LT_REQUIRE(!partial || nb_out, LT_ERR_ARG, "lt_spconv_plan_fwd: the LDS-staged kernel with statistics needs nb_out");
    return lt_spconv_ls_launch(a, stream);
  }
  if (variant == 0) {
    const LtConvShape sh = conv_default_shape(Vin, Vout, Cin, Cout, K);
    static const bool no_sj = getenv("LT_CONV_NO_SPLITJ") != nullptr;  // A/B measurements: the flat split-k loop
    if (sh.sj && !no_sj) return launch_sj(a, sh.sr, sh.ntw, sh.w, stream);
    return launch_sk(a, sh.sr, sh.ntw, sh.w == 2 || sh.w == 4 || sh.w == 8 ? sh.w : 4, stream);
  }
  int sr = 1, ntw = 1;
  if (variant == 1) {  // the LDS-DMA staged kernel
    if (ntt % 8 == 0) return launch_plan<8, 2>(a, stream);
    if (ntt % 6 == 0) return launch_plan<6, 2>(a, stream);
    if (ntt % 4 == 0) return launch_plan<4, 4>(a, stream);
    if (ntt % 3 == 0) return launch_plan<3, 4>(a, stream);
    if (ntt % 2 == 0) return launch_plan<2, 8>(a, stream);
    return launch_plan<1, 8>(a, stream);
  }
  if (variant >= 40000) {  // 40000/50000/60000 + split-k code: diagnostics with L1-resident weights / rows / both
    a.dbg = variant / 10000 - 3;
    variant %= 10000;
    if (a.dbg & 4) a.stamps = g_conv_stamps;
  }
  if (variant >= 10000) {  // 10000 + split-k code: the same without the XCD-aware workgroup mapping (A/B measurements)
    a.swz = 0;
    variant -= 10000;
    LT_REQUIRE(variant >= 1000, LT_ERR_ARG, "lt_spconv_plan_fwd_v: unknown variant %d", variant + 10000);
  }
  if (variant >= 3000 && variant < 4000) {  // bf16x3 split precision (packed with mode 4..6): 3000 + 100*SR + 10*NTW + W; 3000 = default shape
    int v = variant - 3000;
    if (v == 0) {
      const int jp = Cin / 32;
      const int nw = ntt % 4 == 0 ? 4 : ntt % 3 == 0 ? 3 : 2;
      const int ww = jp % 2 == 0 ? 2 : (jp % 3 == 0 ? 3 : 1);
      v = 100 * (Vout >= 100000 ? 4 : 2) + 10 * nw + ww;
    }
    sr = v / 100;
    ntw = (v / 10) % 10;
    const int w = v % 10;
    LT_REQUIRE(bf_supported(Cin, Cout, sr, ntw, w) && Vin * Cin * 4 < 0x7FFFF000ll && (int64_t)K * Cin * Cout * 4 < 0x7FFFF000ll,
               LT_ERR_ARG, "lt_spconv_plan_fwd_v: bad bf16x3 variant %d for Cin=%d Cout=%d", variant, Cin, Cout);
    return launch_bf(a, sr, ntw, w, stream);
  }
  if (variant >= 2000 && variant < 3000) {  // split-j: 2000 + 100*SR + 10*NTW + W
    const int v = variant - 2000;
    sr = v / 100;
    ntw = (v / 10) % 10;
    const int w = v % 10;
    LT_REQUIRE(sj_supported(Cin, sr, ntw, w) && ntt % ntw == 0 && Vin * Cin * 4 < 0x7FFFF000ll &&
                   (int64_t)K * Cin * Cout * 4 < 0x7FFFF000ll, LT_ERR_ARG,
               "lt_spconv_plan_fwd_v: bad variant %d for Cin=%d Cout=%d", variant, Cin, Cout);
    return launch_sj(a, sr, ntw, w, stream);
  }
  if (variant >= 1000) {  // split-k: 1000 + 100*SR + 10*NTW + W
    const int v = variant - 1000;
    sr = v / 100;
    ntw = (v / 10) % 10;
    const int w = v % 10;
    LT_REQUIRE((sr == 1 || sr == 2 || sr == 4) && (ntw == 1 || ntw == 2 || ntw == 3 || ntw == 4 || ntw == 6 || ntw == 8) &&
                   (w == 2 || w == 4 || w == 8) && ntt % ntw == 0,
               LT_ERR_ARG, "lt_spconv_plan_fwd_v: bad variant %d for Cout=%d", variant, Cout);
    return launch_sk(a, sr, ntw, w, stream);
  }
  LT_REQUIRE(variant >= 100, LT_ERR_ARG, "lt_spconv_plan_fwd_v: unknown variant %d", variant);
  {
    sr = variant / 100;
    ntw = variant % 100;
    LT_REQUIRE((sr == 1 || sr == 2 || sr == 4 || sr == 8) && (ntw == 1 || ntw == 2 || ntw == 4) && ntt % ntw == 0, LT_ERR_ARG,
               "lt_spconv_plan_fwd_v: bad variant %d for Cout=%d", variant, Cout);
  }
  if (ntw == 4) return launch_rf<4>(a, sr, stream);
  if (ntw == 2) return launch_rf<2>(a, sr, stream);
  return launch_rf<1>(a, sr, stream);
}

extern "C" int lt_spconv_plan_fwd(const float* in, const float* packed, const void* plan, float* out, int64_t Vin,
                                  int64_t Vout, int32_t Cin, int32_t Cout, int32_t K, hipStream_t stream) {
  return lt_spconv_plan_fwd_v(in, packed, plan, out, Vin, Vout, Cin, Cout, K, 0, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem kernel: Cin = 4 (the network's input: x, y, z, intensity -> 32 channels, latte/models/spvcnn.py:29-31).  With four
// input channels the conv is a streaming problem (per output row: 27 table entries, ~6.5 gathered 16-B rows, 128 B out),
// not a matrix one -- an MFMA tile would be 4/16 full.  One lane per output row, one wave per 64-row plan tile: the tile's
// neighbour rows come coalesced from the plan (tnbr), offsets without a neighbour in the tile are skipped from the tile's
// row masks (rows are mask-sorted: ~7 of 27 offsets survive), the weights of an offset are wave-uniform and come through
// scalar loads (no LDS, no vector loads), COUT accumulators per lane.
template <int COUT>
__global__ __launch_bounds__(256) void k_spconv_stem(PlanArgs a, const float* __restrict__ W) {
  const int lane = threadIdx.x & 63;
  const int64_t T = ((int64_t)a.Vout + LT_TM - 1) / LT_TM;
  const int64_t tile = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
  if (tile >= T) return;
  const int64_t slot = tile * LT_TM + lane;
  const int row = slot < a.Vout ? a.order[slot] : -1;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
  for (int k = 0; k < a.K; ++k) {
    const unsigned long long m = a.tmask[tile * a.K + k];   // wave-uniform
    if (m == 0ull) continue;
    const int nb = a.tnbr[(tile * a.K + k) * LT_TM + lane];
    f32x4 x = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (nb >= 0 && nb < a.Vin) x = *reinterpret_cast<const f32x4*>(a.in + (int64_t)nb * 4);
    const float* wk = W + (int64_t)k * 4 * COUT;             // [4][COUT], uniform address: scalar loads
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[c] = __builtin_fmaf(x[ci], wk[ci * COUT + c], acc[c]);
  }
  if (row >= 0) {
    f32x4* o = reinterpret_cast<f32x4*>(a.out + (int64_t)row * COUT);
#pragma unroll
    for (int c = 0; c < COUT / 4; ++c) o[c] = (f32x4){acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
  }
}

extern "C" int lt_spconv_stem_fwd(const float* in, const float* weight, const void* plan, float* out, int64_t Vin,
                                  int64_t Vout, int32_t Cin, int32_t Cout, int32_t K, hipStream_t stream) {
  LT_REQUIRE(Vin >= 0 && Vout >= 0 && Vin < INT32_MAX && Vout < INT32_MAX, LT_ERR_RANGE, "lt_spconv_stem_fwd: bad row counts");
  LT_REQUIRE(K >= 1 && K <= LT_KMAX, LT_ERR_ARG, "lt_spconv_stem_fwd: kernel volume %d not in [1,%d]", K, LT_KMAX);
  LT_REQUIRE(Cin == 4 && (Cout == 16 || Cout == 32), LT_ERR_ARG,
             "lt_spconv_stem_fwd: Cin=%d, Cout=%d (the stem kernel takes 4 input channels and 16 or 32 outputs)", Cin, Cout);
  if (Vout == 0) return LT_OK;
  LT_REQUIRE(weight && plan && out && (Vin == 0 || in), LT_ERR_ARG, "lt_spconv_stem_fwd: null pointer");
  LT_REQUIRE(((uintptr_t)plan & 15) == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, LT_ERR_ARG,
             "lt_spconv_stem_fwd: plan, in and out must be 16-byte aligned");
  LtPlanView pv = lt_plan_view(const_cast<void*>(plan), K, Vout);
  PlanArgs a{in, nullptr, pv.tnbr, pv.tmask, pv.order, pv.torder, out, (int)Vin, (int)Vout, Cin, Cout, K, nullptr, 0, 0, 0, nullptr};
  LtConvTimerScope timed(stream, Cin, Cout, K, Vout);
  const unsigned grid = (unsigned)(((Vout + LT_TM - 1) / LT_TM + 3) / 4);
  if (Cout == 32) LT_CONV_LAUNCH(k_spconv_stem<32>, dim3(grid), dim3(256), stream, a, weight);
  else LT_CONV_LAUNCH(k_spconv_stem<16>, dim3(grid), dim3(256), stream, a, weight);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

template <int NT>
static int launch_conv(const ConvArgs& a, hipStream_t stream) {
  dim3 grid(lt_cdiv(a.Vout, LT_TM), a.Cout / (16 * NT));
  if (a.Cin % 16 == 0) LT_CONV_LAUNCH(k_spconv_vec<NT>, grid, dim3(256), stream, a);
  else LT_CONV_LAUNCH(k_spconv_scalar<NT>, grid, dim3(256), stream, a);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

extern "C" int lt_spconv_fwd(const float* in, const float* packed, const int32_t* nbr, const int32_t* order, float* out,
                             int64_t Vin, int64_t Vout, int32_t Cin, int32_t Cout, int32_t K, hipStream_t stream) {
  LT_REQUIRE(Vin >= 0 && Vout >= 0 && Vin < INT32_MAX && Vout < INT32_MAX, LT_ERR_RANGE, "lt_spconv_fwd: bad row counts");
  LT_REQUIRE(K >= 1 && K <= LT_KMAX, LT_ERR_ARG, "lt_spconv_fwd: kernel volume %d not in [1,%d]", K, LT_KMAX);
  LT_REQUIRE(Cin >= 4 && Cin % 4 == 0 && (Cin % 16 == 0 || Cin <= 12), LT_ERR_ARG,
             "lt_spconv_fwd: Cin=%d must be a multiple of 16, or 4/8/12", Cin);
  LT_REQUIRE(Cout >= 16 && Cout % 16 == 0, LT_ERR_ARG, "lt_spconv_fwd: Cout=%d must be a multiple of 16", Cout);
  if (Vout == 0) return LT_OK;
  LT_REQUIRE(packed && nbr && out && (Vin == 0 || in), LT_ERR_ARG, "lt_spconv_fwd: null pointer");
  ConvArgs a{in, packed, nbr, order, out, (int)Vin, (int)Vout, Cin, Cout, K};
  LtConvTimerScope timed(stream, Cin, Cout, K, Vout);
  const int ntt = Cout / 16;
  // widest accumulator strip (<= 8 tiles = 128 channels) that divides the channel-tile count
  if (ntt % 8 == 0) return launch_conv<8>(a, stream);
  if (ntt % 6 == 0) return launch_conv<6>(a, stream);
  if (ntt % 4 == 0) return launch_conv<4>(a, stream);
  if (ntt % 3 == 0) return launch_conv<3>(a, stream);
  if (ntt % 2 == 0) return launch_conv<2>(a, stream);
  return launch_conv<1>(a, stream);
}
