// A C++ host program that uses ONLY include/kvhip.h and the HIP runtime — no Python, no torch: the
// boundary a TF custom op (tfplus_amd/tf_shim) or any other host language binds.  Exercises create /
// init / GatherOrInsert / GroupAdam V4 apply / GatherOrZeros / size / export and checks the one-step
// Adam closed form the reference's own test asserts (py_ut/tests/test_training_ops.py:437-454).
// Build: hipcc -std=c++17 -I include tests/c_abi/c_abi_smoke.cc -L tfplus_amd/csrc -lkvhip -o smoke
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kvhip.h"

#define CHECK_KV(x)                                                              \
  do {                                                                           \
    int rc_ = (x);                                                               \
    if (rc_ != KV_OK) { std::printf("FAIL %s -> %d: %s\n", #x, rc_, kv_last_error()); return 1; } \
  } while (0)
#define CHECK_HIP(x) do { if ((x) != hipSuccess) { std::printf("FAIL %s\n", #x); return 1; } } while (0)

int main() {
  const int D = 64, N = 10, R = 16;
  kv_handle_t var = nullptr, slot = nullptr;
  CHECK_KV(kv_create(KV_DT_INT64, KV_DT_FLOAT, D, 0, 1000, 0, &var));
  CHECK_KV(kv_create(KV_DT_INT64, KV_DT_FLOAT, 3 * D, 0, 1000, 0, &slot));
  // uninitialised tables refuse lookups with FailedPrecondition (training_ops.cc:7001-7008)
  float* d_out; long long* d_ids; float* d_grad; float* d_tab;
  CHECK_HIP(hipMalloc(&d_out, N * D * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_ids, N * sizeof(long long)));
  CHECK_HIP(hipMalloc(&d_grad, N * D * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_tab, R * 3 * D * sizeof(float)));
  std::vector<long long> ids(N);
  for (int i = 0; i < N; ++i) ids[i] = 1000003ll * i - 7;
  CHECK_HIP(hipMemcpy(d_ids, ids.data(), N * sizeof(long long), hipMemcpyHostToDevice));
  if (kv_gather_or_insert(var, d_ids, nullptr, N, d_out, nullptr) != KV_FAILED_PRECONDITION) {
    std::printf("FAIL: lookup on an uninitialised table must be FailedPrecondition\n");
    return 1;
  }
  std::vector<float> ones(R * D, 1.0f), zeros(R * 3 * D, 0.0f);
  CHECK_HIP(hipMemcpy(d_tab, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
  CHECK_KV(kv_init_table(var, d_tab, R, nullptr));
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(d_tab, zeros.data(), zeros.size() * sizeof(float), hipMemcpyHostToDevice));
  CHECK_KV(kv_init_table(slot, d_tab, R, nullptr));
  CHECK_KV(kv_gather_or_insert(var, d_ids, nullptr, N, d_out, nullptr));
  std::vector<float> out(N * D), grad(N * D);
  CHECK_HIP(hipMemcpy(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (float v : out) if (v != 1.0f) { std::printf("FAIL: ones-initialised rows expected\n"); return 1; }
  unsigned s = 12345;
  for (auto& g : grad) { s = s * 1664525u + 1013904223u; g = (float)((s >> 8) & 0xFFFF) / 65536.0f + 0.01f; }
  CHECK_HIP(hipMemcpy(d_grad, grad.data(), grad.size() * sizeof(float), hipMemcpyHostToDevice));
  const float lr = 0.5f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  CHECK_KV(kv_apply_group_adam(var, slot, d_grad, d_ids, N, lr, b1, b2, b1, b2, eps, 0.f, 0.f, 0.f, 4, nullptr));
  CHECK_KV(kv_gather_or_zeros(var, d_ids, N, d_out, nullptr));
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost));
  const double lr_t = (double)lr * std::sqrt(1.0 - (double)b2) / (1.0 - (double)b1);
  double worst = 0;
  for (int i = 0; i < N * D; ++i) {
    const double g = grad[i];
    const double want = 1.0 - lr_t * ((1.0 - (double)b1) * g) / (std::sqrt((1.0 - (double)b2) * g * g) + (double)eps);
    worst = std::fmax(worst, std::fabs(out[i] - want) / std::fabs(want));
  }
  int64_t size = 0, freq = 0, cnt[3] = {0, 0, 0};
  CHECK_KV(kv_size(var, &size, nullptr));
  CHECK_KV(kv_sum_freq(var, &freq, nullptr));
  CHECK_KV(kv_export_count(var, 6, cnt, nullptr));
  std::printf("size %lld sum_freq %lld export (%lld, %lld, %lld) worst_rel %.3g\n", (long long)size, (long long)freq,
              (long long)cnt[0], (long long)cnt[1], (long long)cnt[2], worst);
  if (size != N || freq != N || cnt[0] != N || worst > 1e-5) { std::printf("FAIL: unexpected state\n"); return 1; }
  if (kv_apply_group_adam(var, slot, d_grad, d_ids, N, -1.f, b1, b2, b1, b2, eps, 0.f, 0.f, 0.f, 4, nullptr) !=
      KV_INVALID_ARGUMENT) {
    std::printf("FAIL: lr <= 0 must be InvalidArgument\n");
    return 1;
  }
  CHECK_KV(kv_destroy(slot));
  CHECK_KV(kv_destroy(var));
  std::printf("C ABI OK\n");
  return 0;
}
