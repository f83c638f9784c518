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
  int64_t nb = 0;   // partial rows the conv kernel that ran wrote (the kernel choice depends on the map, not only on the shape)
  int rc = lt_spconv_plan_fwd_stats_nb(in, packed, plan, conv_out, Vin, Vout, Cin, Cout, K, ws, &nb, stream);
  if (rc) return rc;
  rc = lt_bn_finalize_sums_nb(ws, nb, Vout, Cout, mean, var, stream);
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

// ---------------------------------------------------------------------------------------------------------------------
// Parameter update of one TTA iteration over the FLAT buffer of trainable scalars (the 13 248 BatchNorm gamma / beta of
// SPVCNN at cr = 1): torch.optim.Adam's step (latte/tta/tools/latte.py:317-319, optimizer built by
// latte/common/solver/build.py:7-52: Adam, lr 1e-3, default betas / eps, L2 weight decay) followed by the EMA teacher
// update of the same parameters (latte/common/utils/ema_util.py:99-118: ema = alpha*ema + (1-alpha)*param) -- ONE launch
// instead of ~10 multi-tensor launches over 104 tensors.  The arithmetic follows torch's _multi_tensor_adam op by op
// (lerp of exp_avg, mul + addcmul of exp_avg_sq, sqrt / bias_correction2_sqrt + eps, addcdiv), fp32, no contraction.
__global__ void k_adam_ema(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                           float* __restrict__ ema, int64_t n, float grad_scale, float wd, float beta1, float beta2,
                           float eps, float step_size, float bc2_sqrt, float alpha) {
#pragma clang fp contract(off)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i];
    float gi = g[i] * grad_scale;
    if (wd != 0.f) gi = gi + wd * pi;
    float mi = m[i];
    mi = mi + (1.0f - beta1) * (gi - mi);            // exp_avg.lerp_(grad, 1 - beta1)
    float vi = v[i] * beta2;
    vi = vi + (1.0f - beta2) * (gi * gi);            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi + (-step_size) * (mi / denom);           // param.addcdiv_(exp_avg, denom, value=-step_size)
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (ema) ema[i] = ema[i] * alpha + (1.0f - alpha) * pi;
  }
}

extern "C" int lt_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t n,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                float ema_alpha, float grad_scale, hipStream_t stream) {
  LT_REQUIRE(n >= 0 && step >= 1, LT_ERR_ARG, "lt_adam_ema_step: n=%lld, step=%lld (steps count from 1)", (long long)n,
             (long long)step);
  if (n == 0) return LT_OK;
  LT_REQUIRE(param && grad && exp_avg && exp_avg_sq, LT_ERR_ARG, "lt_adam_ema_step: null pointer");
  LT_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && lr >= 0.f, LT_ERR_ARG,
             "lt_adam_ema_step: bad hyper-parameters");
  // scalar factors exactly as torch computes them on the host (Python floats = doubles), rounded to fp32 once
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(k_adam_ema, dim3(lt_grid(n, 256)), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, ema, n,
                     grad_scale, weight_decay, beta1, beta2, eps, step_size, bc2_sqrt, ema_alpha);
  LT_LAUNCH_CHECK();
  return LT_OK;
}
