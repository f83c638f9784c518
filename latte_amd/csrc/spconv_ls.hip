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
#include "spconv_ls_body.inc"
}

// The same code under its own name for the K = 1 launches (the network's dense layers over lt_kmap_plan_identity), so that
// kernel traces and counters tell the HBM-bound dense products from the sparse convs (see k_dense_sj in spconv.hip).
template <int NB, int JC, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_dense_ls(PlanArgs a, int tpw) {
#include "spconv_ls_body.inc"
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
  if (a.K == 1) {
    if (lds > 64 * 1024) {
      static thread_local int done_dev_d = -1;
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (done_dev_d != dev) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_ls<NB, JC, WPB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        done_dev_d = dev;
      }
    }
    hipExtLaunchKernelGGL((k_dense_ls<NB, JC, WPB>), dim3(total), dim3(64 * WPB), lds, stream, e0, e1, 0, b, s.tpw);
    return;
  }
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
