// Shared between the sparse-conv translation units (spconv.hip, spconv_ls.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LT_TM 64        // output rows per plan tile
#define LT_KMAX 32      // max kernel volume (27 used)

// arguments of the planned conv kernels: a conv plan (lt_kmap_plan) + operands
struct PlanArgs {
  const float* in;
  const float* packed;
  const int* tnbr;                    // [T][K][64]
  const unsigned long long* tmask;    // [T][K]
  const int* order;                   // [Vout]
  const int* torder;                  // [T] tiles by descending work
  float* out;
  int Vin, Vout, Cin, Cout, K;
  float* partial;                     // optional [row workgroups][2][Cout]: per-workgroup column sums of out and out^2
  int gx;                             // split-k: row workgroups (the grid is 1-D there, see the XCD mapping)
  int swz;                            // split-k: XCD-aware workgroup mapping on
  int dbg;                            // diagnostics (wrong results): 1 = weights of slab 0 only, 2 = rows 0..15 only, 4 = stamps, 8 / 16 = no A / B loads (split-j)
  long long* stamps;                  // [workgroup][wave][8]
};

// events of the armed conv launch timer for the launch being issued by THIS thread (null otherwise); spconv.hip owns them
void lt_conv_launch_events(hipEvent_t* e0, hipEvent_t* e1);

// spconv_ls.hip: LDS-staged-weights kernel for maps with few offsets per row (the 2^3 stride-2 / transposed convs)
bool lt_spconv_ls_supported(int64_t Vin, int64_t Vout, int Cin, int Cout, int K);
int64_t lt_spconv_ls_stat_blocks(int64_t Vout, int Cin, int Cout, int K);
int lt_spconv_ls_launch(const PlanArgs& a, hipStream_t stream);
