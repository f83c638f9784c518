// Composite entry points of the SPVCNN block pattern  conv -> BatchNorm(batch stats) [+ residual] [+ ReLU]
// (latte/models/basic_blocks.py:13-76): one host call issues the whole kernel sequence, the BN statistics come out of
// the conv epilogue (no extra pass over the conv output).  Pure sequencing of entry points that also exist on their own.
#include "common.h"
#include "../../include/latte_hip.h"

extern "C" int64_t lt_conv_bn_workspace_bytes(int64_t Vout, int32_t Cout) {
  const int64_t a = lt_spconv_stats_bytes(Vout, Cout);
  const int64_t b = lt_bn_workspace_bytes(Vout, Cout);
  return (a > b ? a : b) + 2 * lt_align_up((int64_t)Cout * 4, 256) + 256;
}

extern "C" int lt_conv_bn_fwd(const float* in, const float* packed, const void* plan, int64_t Vin, int64_t Vout,
                              int32_t Cin, int32_t Cout, int32_t K, const float* gamma, const float* beta, float eps,
                              const float* residual, int32_t relu, void* ws, int64_t ws_bytes, float* conv_out, float* y,
                              float* mean, float* var, hipStream_t stream) {
  LT_REQUIRE(Vout >= 2, LT_ERR_ARG, "lt_conv_bn_fwd: batch statistics need more than one row (Vout=%lld)", (long long)Vout);
  LT_REQUIRE(ws && ws_bytes >= lt_conv_bn_workspace_bytes(Vout, Cout), LT_ERR_WORKSPACE, "lt_conv_bn_fwd: workspace too small");
  int rc = lt_spconv_plan_fwd_stats(in, packed, plan, conv_out, Vin, Vout, Cin, Cout, K, ws, stream);
  if (rc) return rc;
  rc = lt_bn_finalize_sums(ws, Vout, Cout, mean, var, stream);
  if (rc) return rc;
  return lt_bn_apply(conv_out, mean, var, gamma, beta, eps, residual, relu, y, Vout, Cout, stream);
}

extern "C" int lt_conv_bn_bwd(const float* dy, const float* conv_out, const float* y, const float* mean, const float* var,
                              const float* gamma, const float* beta, float eps, int32_t relu, int64_t Vout, int32_t Cout,
                              void* ws,
                              int64_t ws_bytes, float* dgamma, float* dbeta, float* dres, float* dconv,
                              const float* packed_bwd, const void* plan_bwd, int64_t Vin, int32_t Cin, int32_t K, float* din,
                              hipStream_t stream) {
  LT_REQUIRE(ws && ws_bytes >= lt_conv_bn_workspace_bytes(Vout, Cout), LT_ERR_WORKSPACE, "lt_conv_bn_bwd: workspace too small");
  LT_REQUIRE(dgamma && dbeta, LT_ERR_ARG, "lt_conv_bn_bwd: null pointer");
  LT_REQUIRE(!din || (dconv && packed_bwd && plan_bwd), LT_ERR_ARG, "lt_conv_bn_bwd: din needs dconv, packed_bwd and plan_bwd");
  // sum_g = dbeta, sum_gx = dgamma
  int rc = lt_bn_bwd_reduce(dy, conv_out, y, mean, var, gamma, beta, eps, relu, Vout, Cout, ws, ws_bytes, dbeta, dgamma,
                            stream);
  if (rc) return rc;
  if (dconv || dres) {
    rc = lt_bn_bwd_apply(dy, conv_out, y, mean, var, gamma, beta, eps, relu, dbeta, dgamma, dconv, dres, Vout, Cout,
                         stream);
    if (rc) return rc;
  }
  if (din) {
    // gradient w.r.t. the conv input: the same product over the backward table (Cout -> Cin channels)
    rc = lt_spconv_plan_fwd(dconv, packed_bwd, plan_bwd, din, Vout, Vin, Cout, Cin, K, stream);
    if (rc) return rc;
  }
  return LT_OK;
}
