// Sparse conv for maps with FEW offsets per output row: the 2^3 stride-2 convs and, above all, their transposes
// (spnn.Conv3d(ks=2, stride=2[, transposed=True]): latte/models/spvcnn.py:36,42,48,54,60,69,78,87 through
// basic_blocks.py:13-45), forward and backward-input.
//
// In the transposed direction every fine output row has exactly ONE (parent row, offset k) pair: out[i] = in[parent(i)] @ W_k.
// The mask-sorted plan (lt_kmap_plan) therefore groups the rows by k, and the conv is eight dense [m_k, Cin] x [Cin, Cout]
// products with gathered A rows.  The general kernel (k_spconv_sj) pays a prologue and a split-reduction epilogue per 16..32
// rows for ONE offset's worth of MFMAs there (45 TFLOP/s at 96 channels, 20 at 32).  Here
//  * a workgroup owns `tpw` consecutive plan tiles (64 rows each) and 16*NB output columns, and stages the weights of the
//    current offset -- W_k[:, column block] in MFMA-fragment order, J*NB KiB -- in LDS ONCE; it is re-staged only when the
//    offset changes (rows are grouped by k: about once per workgroup).  B fragments are ds_read_b128s, not vector-memory
//    loads: the per-CU address path only carries the gathered rows;
//  * a wave owns one 16-row strip of a tile for ALL of Cin (no split over waves, no partial sums through LDS): every output
//    element is one k-ascending, channel-ascending fmaf chain on v_mfma_f32_16x16x4_f32 -- bitwise equal to the table kernel;
//  * gathered rows come through raw buffer loads (a missing neighbour is an out-of-range offset: zeros, no memory touched),
//    JC channel groups per load burst, one burst ahead of the MFMAs; the neighbour rows of the NEXT (tile, offset) run are
//    fetched one run ahead;
//  * the BatchNorm statistics epilogue (column sums of the workgroup's rows) as in the general kernel.
// The kernel is correct for any plan (it walks every active (tile, offset)); it pays off when a tile has few active offsets.
#include <hip/hip_ext.h>
#include <stdlib.h>
#include "common.h"
#include "spconv_args.h"

#define LS_MAXT 32      // plan tiles per workgroup, at most

template <int NB, int JC, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_spconv_ls(PlanArgs a, int tpw) {
  // dynamic LDS (as little as possible: the number of resident workgroups per CU is what it buys):
  //   weights of the staged offset [J][NB][64 lanes][4] floats | row masks of this workgroup's tiles [tpw][K] u64
  // the statistics epilogue reuses the weight area
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  __shared__ unsigned int s_kb[LS_MAXT];                        // offsets present in each tile
  constexpr int G = WPB / 4;                                    // tiles processed side by side (4 waves = 4 strips each)
  unsigned long long* s_tm = reinterpret_cast<unsigned long long*>(s_w + (a.Cin / 16) * NB * 256);
  float* s_stat = s_w;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int K = a.K, J = a.Cin / 16, NTt = a.Cout / 16, NC = J / JC;
  // diagnostics (a.dbg & 4): per-wave s_memtime stamps -> a.stamps[workgroup][wave][8]
  const bool stamp = (a.dbg & 4) && a.stamps;
  long long st0 = 0, st1 = 0, st2 = 0, t_mma = 0, t_stage = 0, t_epi = 0, t_adv = 0;
  if (stamp) st0 = __builtin_amdgcn_s_memtime();
  // XCD-aware mapping of the 1-D grid: the column blocks of one row block run on ONE XCD (its L2 serves the gather to all)
  const unsigned gy = NTt / NB;
  const unsigned L = blockIdx.x, li = L >> 3;
  const unsigned bx = (li / gy) * 8 + (L & 7), by = li % gy;
  if (bx >= (unsigned)a.gx) return;
  const int64_t T = ((int64_t)a.Vout + LT_TM - 1) / LT_TM;
  const int64_t t0 = (int64_t)bx * tpw;
  const int nt = (int)((T - t0) < tpw ? (T - t0) : tpw);
  const int nt0 = by * NB;

  for (int i = tid; i < nt * K; i += 64 * WPB) s_tm[i] = a.tmask[t0 * K + i];
  __syncthreads();
  if (tid < nt) {
    unsigned int kb = 0u;
    for (int k = 0; k < K; ++k)
      if (s_tm[tid * K + k] != 0ull) kb |= 1u << k;
    s_kb[tid] = kb;
  }
  __syncthreads();

  const int sub = wave >> 2, strip = wave & 3;
  const int ngroups = (nt + G - 1) / G;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0,
                                                                        (int)((int64_t)a.Vin * a.Cin * 4), 0x00020000);
  const int row_bytes = a.Cin * 4;
  const f32x4* s_w4 = reinterpret_cast<const f32x4*>(s_w);

  // this wave's strip of tile tl has a neighbour at offset k
  auto smask = [&](int tl, int k) -> bool {
    return ((s_tm[tl * K + k] >> (16 * strip)) & 0xFFFFull) != 0ull;
  };
  // The runs of this wave -- (tile of this wave, offset) pairs with a neighbour in this wave's strip -- as ONE 64-bit mask in
  // processing order: bit i = (i / K)-th tile of this wave, offset i % K (host: tiles per wave * K <= 64).  "Next run" is a
  // find-first-set on scalars instead of a walk over the LDS mask table (measured: 2 900 cycles per run in that walk).
  unsigned long long runs;
  {
    const int ti = lane / K, kk = lane - ti * K, tl_ = ti * G + sub;
    runs = __ballot(tl_ < nt && smask(tl_ < nt ? tl_ : 0, kk));
  }
  // (tl, k) -> the next run after it; false: none
  auto next_run = [&](int& tl, int& k) -> bool {
    const int i = ((tl - sub) / G) * K + k + 1;          // first slot after (tl, k); (sub - G, K - 1) -> 0
    const unsigned long long rest = i < 64 ? (runs >> i) << i : 0ull;
    if (rest == 0ull) return false;
    const int nx = __builtin_ctzll(rest);
    tl = (nx / K) * G + sub;
    k = nx % K;
    return true;
  };
  // neighbour rows of run (tl, k): lane r reads the row of strip row r
  auto load_nbr = [&](int tl, int k) -> int { return a.tnbr[((t0 + tl) * K + k) * LT_TM + 16 * strip + r]; };
  // a.dbg (diagnostics, wrong results): 1 = no output stores, 2 = gathered rows from a 1024-row window (cache-resident),
  // 8 = no gathered-row loads
  auto voff_of = [&](int nb) -> int {
    if (a.dbg & 2) nb &= 1023;
    return (nb >= 0 && nb < a.Vin) ? nb * row_bytes + 16 * q : 0x7FFFF000;
  };

  f32x4 nxt[JC];   // the load burst in flight: channel groups [c*JC, (c+1)*JC) of the run being entered
#pragma unroll
  for (int jj = 0; jj < JC; ++jj) nxt[jj] = (f32x4){1.f, 1.f, 1.f, 1.f};
  auto issue = [&](int voff, int c) {
    if (a.dbg & 8) return;
#pragma unroll
    for (int jj = 0; jj < JC; ++jj)
      nxt[jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voff, 64 * (c * JC + jj), 0));
  };

  // run state: cur = the run whose bursts are being issued / consumed, nx = the one after it
  int cur_tl = sub - G, cur_k = K - 1;   // "before the first tile of this wave"
  bool have_cur = next_run(cur_tl, cur_k);
  int nx_tl = cur_tl, nx_k = cur_k;
  bool have_nx = have_cur && next_run(nx_tl, nx_k);
  int voff_cur = 0x7FFFF000, nbn = -1;
  if (have_cur) {
    voff_cur = voff_of(load_nbr(cur_tl, cur_k));
    issue(voff_cur, 0);
  }
  if (have_nx) nbn = load_nbr(nx_tl, nx_k);

  float s1[NB], s2[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) s1[n] = s2[n] = 0.f;
  int staged_k = -1;
  if (stamp) st1 = __builtin_amdgcn_s_memtime();

  for (int g = 0; g < ngroups; ++g) {
    const int tl = g * G + sub;              // this wave's tile of the group
    const bool mine = tl < nt;
    int rowv = -1;                           // output row of strip row r (lanes 0..15 matter)
    if (mine) {
      const int64_t slot = (t0 + tl) * LT_TM + 16 * strip + r;
      if (slot < a.Vout) rowv = a.order[slot];
    }
    unsigned int uk = 0u;                    // offsets present in the group's tiles (workgroup-uniform)
#pragma unroll
    for (int s = 0; s < G; ++s)
      if (g * G + s < nt) uk |= s_kb[g * G + s];
    f32x4 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    while (uk) {
      const int k = __builtin_ctz(uk);
      uk &= uk - 1u;
      if (k != staged_k) {                   // workgroup-uniform
        long long ts = 0;
        if (stamp) ts = __builtin_amdgcn_s_memtime();
        __syncthreads();                     // everyone is done with the previous offset's weights
        for (int e = tid; e < J * NB * 64; e += 64 * WPB) {
          const int j = e / (NB * 64), rem = e - j * (NB * 64);
          reinterpret_cast<f32x4*>(s_w)[e] =
              reinterpret_cast<const f32x4*>(a.packed)[((int64_t)(k * J + j) * NTt + nt0) * 64 + rem];
        }
        __syncthreads();
        staged_k = k;
        if (stamp) t_stage += __builtin_amdgcn_s_memtime() - ts;
      }
      if (!(mine && have_cur && cur_tl == tl && cur_k == k)) continue;   // wave-uniform: no row of this strip at k
      long long tm = 0;
      if (stamp) tm = __builtin_amdgcn_s_memtime();
      for (int c = 0; c < NC; ++c) {
        f32x4 av[JC];
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) av[jj] = nxt[jj];
        // the next burst goes out before this one's MFMAs: the rest of this run, else the first burst of the next run
        if (c + 1 < NC) {
          issue(voff_cur, c + 1);
        } else if (have_nx) {
          voff_cur = voff_of(nbn);
          issue(voff_cur, 0);
        }
        // B fragments: ds_read_b128s that the scheduler interleaves with the MFMAs of the previous channel group (an explicit
        // second register set, one group ahead, measured no faster: the LDS round trip is not what the waves wait for)
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) {
          const int j = c * JC + jj;
          f32x4 b[NB];
#pragma unroll
          for (int n = 0; n < NB; ++n) b[n] = s_w4[(j * NB + n) * 64 + lane];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj][t], b[n][t], acc[n], 0, 0, 0);
          }
        }
      }
      long long ta = 0;
      if (stamp) { ta = __builtin_amdgcn_s_memtime(); t_mma += ta - tm; }
      // advance: the next run becomes current; look one further ahead and fetch its neighbour rows
      have_cur = have_nx;
      cur_tl = nx_tl;
      cur_k = nx_k;
      if (have_nx) {
        have_nx = next_run(nx_tl, nx_k);
        if (have_nx) nbn = load_nbr(nx_tl, nx_k);
      }
      if (stamp) t_adv += __builtin_amdgcn_s_memtime() - ta;
    }
    long long te = 0;
    if (stamp) te = __builtin_amdgcn_s_memtime();

    // the group's rows are complete: store this wave's strip, keep its column sums (C/D layout: col = lane & 15, row = 4q + reg)
    if (mine) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = __shfl(rowv, 4 * q + reg, 64);
        if (row >= 0 && !(a.dbg & 1)) {
          float* o = a.out + (int64_t)row * a.Cout + 16 * nt0 + r;
#pragma unroll
          for (int n = 0; n < NB; ++n) o[16 * n] = acc[n][reg];
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          s1[n] += acc[n][reg];
          s2[n] += acc[n][reg] * acc[n][reg];
        }
      }
    }
    if (stamp) t_epi += __builtin_amdgcn_s_memtime() - te;
  }
  if (stamp) st2 = __builtin_amdgcn_s_memtime();

  if (a.partial) {
    // column sums of the workgroup's rows: per wave over its strips (rows past Vout hold exact zeros), then over the waves
    // in wave order through LDS -> deterministic
    __syncthreads();   // the weight area becomes the reduction buffer
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float t1 = s1[n], t2 = s2[n];
      t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
      t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
      if (q == 0) {
        s_stat[((wave * NB + n) * 2 + 0) * 16 + r] = t1;
        s_stat[((wave * NB + n) * 2 + 1) * 16 + r] = t2;
      }
    }
    __syncthreads();
    if (wave == 0 && lane < 32) {
      const int which = lane >> 4;
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) t += s_stat[((w * NB + n) * 2 + which) * 16 + r];
        a.partial[((int64_t)bx * 2 + which) * a.Cout + 16 * (nt0 + n) + r] = t;
      }
    }
  }
  if (stamp && lane == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long* o = a.stamps + ((long long)blockIdx.x * WPB + wave) * 8;
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime();
    o[4] = t_mma; o[5] = t_stage; o[6] = t_epi | (t_adv << 32);
    o[7] = (long long)hwid | ((long long)xcc << 32);
  }
}

static int lt_cu_count() {
  static const int n = [] {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cu <= 0)
      cu = 256;   // MI355X
    return cu;
  }();
  return n;
}

struct LsShape {
  int nb, jc, wpb, tpw;
  int64_t gx;
};

static bool ls_shape(int64_t Vout, int Cin, int Cout, int K, LsShape& s) {
  if (Cin < 16 || Cin % 16 || Cout < 16 || Cout % 16) return false;
  const int J = Cin / 16, ntt = Cout / 16;
  static const int nbs[] = {8, 6, 4, 3, 2, 1};
  s.nb = 0;
  for (int nb : nbs)
    if (ntt % nb == 0 && J * nb <= 64) { s.nb = nb; break; }   // the staged block is J * nb KiB, at most 64 KiB of LDS
  if (!s.nb) return false;
  s.jc = J % 4 == 0 ? 4 : J % 3 == 0 ? 3 : J % 2 == 0 ? 2 : 1;
  s.wpb = (s.nb >= 4 && J * s.nb > 40) ? 8 : 4;                 // big stages: 8 waves share one (two workgroups per CU)
  static const int env_wpb = getenv("LT_LS_WPB") ? atoi(getenv("LT_LS_WPB")) : 0;   // diagnostics (tools/k8_bench.py sweeps)
  static const int env_tpw = getenv("LT_LS_TPW") ? atoi(getenv("LT_LS_TPW")) : 0;
  static const int env_wgs = getenv("LT_LS_WGS") ? atoi(getenv("LT_LS_WGS")) : 0;
  if (env_wpb == 8 && s.nb >= 4) s.wpb = 8;
  if (env_wpb == 4) s.wpb = 4;
  const int64_t T = (Vout + LT_TM - 1) / LT_TM;
  const int G = s.wpb / 4;
  // Tiles per workgroup: every tile of such a map is the same amount of work, so the rows are dealt statically -- ONE
  // round of workgroups that are all resident at once (CUs x workgroups that fit a CU by LDS and registers), each
  // staging its weights once.  More rounds only when a workgroup would exceed LS_MAXT tiles.
  const int64_t lds = (int64_t)J * s.nb * 1024 + 2048;
  int resident = (int)(160 * 1024 / lds);
  const int by_regs = 16 / s.wpb;                               // <= 128 VGPRs: 4 waves per SIMD = 16 per CU
  if (resident > by_regs) resident = by_regs;
  if (resident < 1) resident = 1;
  const int64_t slots = (int64_t)lt_cu_count() * resident;
  const int64_t gy = ntt / s.nb;
  int64_t tpw = (T * gy + slots - 1) / (env_wgs > 0 ? env_wgs : slots);
  if (env_tpw > 0) tpw = env_tpw;
  if (tpw < 1) tpw = 1;
  if (tpw > (64 / K) * G) tpw = (64 / K) * G;                    // the run mask of a wave: tiles per wave * K <= 64 bits
  if (tpw > LS_MAXT) tpw = LS_MAXT;
  tpw = (tpw + G - 1) / G * G;
  if (tpw > LS_MAXT) tpw = LS_MAXT / G * G;
  s.tpw = (int)tpw;
  s.gx = (T + tpw - 1) / tpw;
  return true;
}

bool lt_spconv_ls_supported(int64_t Vin, int64_t Vout, int Cin, int Cout, int K) {
  LsShape s;
  return K >= 1 && K <= LT_KMAX && Vout > 0 && ls_shape(Vout, Cin, Cout, K, s) && Vin * (int64_t)Cin * 4 < 0x7FFFF000ll;
}

int64_t lt_spconv_ls_stat_blocks(int64_t Vout, int Cin, int Cout, int K) {
  LsShape s;
  if (Vout <= 0 || K < 1 || K > LT_KMAX || !ls_shape(Vout, Cin, Cout, K, s)) return 0;
  return s.gx;
}

template <int NB, int JC, int WPB>
static void ls_launch_one(const PlanArgs& a, const LsShape& s, hipStream_t stream) {
  PlanArgs b = a;
  b.gx = (int)s.gx;
  const unsigned gy = (unsigned)(a.Cout / (16 * NB));
  const unsigned total = (unsigned)((s.gx + 7) / 8 * 8) * gy;
  const size_t lds = (size_t)(a.Cin / 16) * NB * 1024 + (size_t)s.tpw * a.K * 8;
  if (lds > 64 * 1024) {   // beyond the default dynamic-LDS limit: allowed once per kernel (and device)
    static thread_local int done_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done_dev != dev) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spconv_ls<NB, JC, WPB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      done_dev = dev;
    }
  }
  hipEvent_t e0, e1;
  lt_conv_launch_events(&e0, &e1);
  hipExtLaunchKernelGGL((k_spconv_ls<NB, JC, WPB>), dim3(total), dim3(64 * WPB), lds, stream, e0, e1, 0, b, s.tpw);
}

template <int NB, int WPB>
static void ls_launch_jc(const PlanArgs& a, const LsShape& s, hipStream_t stream) {
  switch (s.jc) {
    case 4: ls_launch_one<NB, 4, WPB>(a, s, stream); break;
    case 3: ls_launch_one<NB, 3, WPB>(a, s, stream); break;
    case 2: ls_launch_one<NB, 2, WPB>(a, s, stream); break;
    default: ls_launch_one<NB, 1, WPB>(a, s, stream); break;
  }
}

int lt_spconv_ls_launch(const PlanArgs& a, hipStream_t stream) {
  LsShape s;
  LT_REQUIRE(ls_shape(a.Vout, a.Cin, a.Cout, a.K, s), LT_ERR_ARG, "lt_spconv_ls: no shape for Cin=%d Cout=%d", a.Cin, a.Cout);
  if (s.wpb == 8) {
    switch (s.nb) {
      case 8: ls_launch_jc<8, 8>(a, s, stream); break;
      case 6: ls_launch_jc<6, 8>(a, s, stream); break;
      default: ls_launch_jc<4, 8>(a, s, stream); break;
    }
  } else {
    switch (s.nb) {
      case 8: ls_launch_jc<8, 4>(a, s, stream); break;
      case 6: ls_launch_jc<6, 4>(a, s, stream); break;
      case 4: ls_launch_jc<4, 4>(a, s, stream); break;
      case 3: ls_launch_jc<3, 4>(a, s, stream); break;
      case 2: ls_launch_jc<2, 4>(a, s, stream); break;
      default: ls_launch_jc<1, 4>(a, s, stream); break;
    }
  }
  LT_LAUNCH_CHECK();
  return LT_OK;
}
