/* A consumer of liblatte_hip.so written in plain C: no torch, no C++, only include/latte_hip*.h and the HIP runtime's C API.
 * What a cgo / JNI / ctypes binding of the boundary sees.  Checks three entry points against values computed on the host
 * (the 64-bit FNV-1a of torchsparse's sphash, softmax + normalised entropy of latte/models/losses.py:10-19, batch
 * statistics of a BatchNorm input) and the error convention.  Built by latte_amd/csrc/Makefile (target abi_smoke), run
 * by tests/test_gpu_abi_c.py.  Exit code 0 and "abi_smoke: OK" on success. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "latte_hip.h"
#include "latte_hip_st.h"

#define CHECK_HIP(e) do { hipError_t _r = (e); if (_r != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)_r, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_LT(e) do { int _r = (e); if (_r != 0) { printf("lt error %d (%s) at %s:%d\n", _r, lt_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static uint64_t fnv4(int32_t x, int32_t y, int32_t z, int32_t b) {
  uint64_t h = 14695981039346656037ull;
  h ^= (uint32_t)x; h *= 1099511628211ull;
  h ^= (uint32_t)y; h *= 1099511628211ull;
  h ^= (uint32_t)z; h *= 1099511628211ull;
  h ^= (uint32_t)b; h *= 1099511628211ull;
  return (h >> 60) ^ (h & 0xFFFFFFFFFFFFFFFull);
}

int main(void) {
  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1) { printf("no GPU\n"); return 4; }
  CHECK_HIP(hipSetDevice(0));
  printf("liblatte_hip version %d\n", lt_version());

  /* 1. sphash */
  enum { N = 8 };
  int32_t coords[N][4] = {{0, 0, 0, 0}, {1, 2, 3, 0}, {-1, 5, 7, 1}, {1000, 2000, 40, 5}, {3, 2, 1, 0}, {1, 2, 3, 1}, {0, 0, 1, 0}, {65535, 1, 1, 2}};
  int32_t* d_c; int64_t* d_h; int64_t h[N];
  CHECK_HIP(hipMalloc((void**)&d_c, sizeof coords));
  CHECK_HIP(hipMalloc((void**)&d_h, sizeof h));
  CHECK_HIP(hipMemcpy(d_c, coords, sizeof coords, hipMemcpyHostToDevice));
  CHECK_LT(lt_sphash(d_c, N, NULL, 0, d_h, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(h, d_h, sizeof h, hipMemcpyDeviceToHost));
  for (int i = 0; i < N; ++i)
    if ((uint64_t)h[i] != fnv4(coords[i][0], coords[i][1], coords[i][2], coords[i][3])) { printf("sphash mismatch at %d\n", i); return 10; }

  /* 2. softmax + entropy */
  enum { P = 5, C = 3 };
  float logits[P][C] = {{0.f, 0.f, 0.f}, {1.f, 2.f, 3.f}, {-4.f, 0.5f, 0.25f}, {10.f, -10.f, 0.f}, {0.1f, 0.2f, 0.3f}};
  float probs[P][C], ety[P];
  float *d_l, *d_p, *d_e;
  CHECK_HIP(hipMalloc((void**)&d_l, sizeof logits));
  CHECK_HIP(hipMalloc((void**)&d_p, sizeof probs));
  CHECK_HIP(hipMalloc((void**)&d_e, sizeof ety));
  CHECK_HIP(hipMemcpy(d_l, logits, sizeof logits, hipMemcpyHostToDevice));
  CHECK_LT(lt_softmax_entropy(d_l, P, C, d_p, d_e, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(probs, d_p, sizeof probs, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(ety, d_e, sizeof ety, hipMemcpyDeviceToHost));
  for (int p = 0; p < P; ++p) {
    double mx = logits[p][0], s = 0.0, e = 0.0, pr[C];
    for (int c = 1; c < C; ++c) if (logits[p][c] > mx) mx = logits[p][c];
    for (int c = 0; c < C; ++c) { pr[c] = exp((double)logits[p][c] - mx); s += pr[c]; }
    for (int c = 0; c < C; ++c) { pr[c] /= s; e += -pr[c] * log2(pr[c] + 1e-30) / log2((double)C); }
    for (int c = 0; c < C; ++c) if (fabs(probs[p][c] - pr[c]) > 1e-6) { printf("softmax mismatch at %d,%d\n", p, c); return 11; }
    if (fabs(ety[p] - e) > 1e-5) { printf("entropy mismatch at %d: %g vs %g\n", p, ety[p], e); return 12; }
  }

  /* 3. batch statistics through the entry point that allocates nothing */
  enum { V = 1000, CH = 16 };
  float* x = (float*)malloc(sizeof(float) * V * CH);
  double m_ref[CH], v_ref[CH];
  for (int c = 0; c < CH; ++c) m_ref[c] = v_ref[c] = 0.0;
  for (int i = 0; i < V; ++i)
    for (int c = 0; c < CH; ++c) {
      x[i * CH + c] = (float)(((i * 37 + c * 11) % 101) / 50.0 - 1.0 + c);
      m_ref[c] += x[i * CH + c];
    }
  for (int c = 0; c < CH; ++c) m_ref[c] /= V;
  for (int i = 0; i < V; ++i)
    for (int c = 0; c < CH; ++c) { const double d = x[i * CH + c] - m_ref[c]; v_ref[c] += d * d; }
  float *d_x, *d_m, *d_v, mean[CH], var[CH];
  CHECK_HIP(hipMalloc((void**)&d_x, sizeof(float) * V * CH));
  CHECK_HIP(hipMalloc((void**)&d_m, sizeof mean));
  CHECK_HIP(hipMalloc((void**)&d_v, sizeof var));
  CHECK_HIP(hipMemcpy(d_x, x, sizeof(float) * V * CH, hipMemcpyHostToDevice));
  CHECK_LT(lt_bn_stats_static(d_x, V, CH, d_m, d_v, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(mean, d_m, sizeof mean, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(var, d_v, sizeof var, hipMemcpyDeviceToHost));
  for (int c = 0; c < CH; ++c) {
    if (fabs(mean[c] - m_ref[c]) > 1e-5 * (1.0 + fabs(m_ref[c]))) { printf("mean mismatch at %d\n", c); return 13; }
    if (fabs(var[c] - v_ref[c] / V) > 1e-5 * (1.0 + v_ref[c] / V)) { printf("var mismatch at %d\n", c); return 14; }
  }

  /* 4. the error convention: negative code + text, nothing thrown, nothing launched */
  const int rc = lt_bn_stats_static(d_x, V, 512, d_m, d_v, NULL);
  if (rc >= 0 || strlen(lt_last_error()) == 0) { printf("expected an error code and a message, got %d '%s'\n", rc, lt_last_error()); return 15; }

  hipFree(d_c); hipFree(d_h); hipFree(d_l); hipFree(d_p); hipFree(d_e); hipFree(d_x); hipFree(d_m); hipFree(d_v);
  free(x);
  printf("abi_smoke: OK\n");
  return 0;
}
