// A consumer of the C ABI with no Python and no torch in the process: plain HIP runtime calls for memory, the entry
// points of include/curla_hip.h for the work, a CPU loop as the check.  Built and run by tests/test_gpu_capi_harness.py
// (hipcc harness.cpp -I include -L curla_amd -lcurla_hip).  Exit code 0 = every check passed; each prints one line.
//
// What it drives: curla_abi_version / curla_set_option (host side), curla_soft_update (utils.py:37-41),
// curla_conv3x3_s1_fwd at 32 filters (the gfx950 row-walk kernel) and at 16 (the generic path), encoder.py:59-63,84-87,
// and curla_f64_pack / curla_f64_unpack (the data-parallel rider).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "curla_hip.h"

#define HIP_OK(x)                                                        \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                          \
    }                                                                    \
  } while (0)

static unsigned g_seed = 12345u;
static float rnd() {  // uniform in [-1, 1)
  g_seed = g_seed * 1664525u + 1013904223u;
  return (float)((g_seed >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

template <class T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

static int check_conv(int B, int H, int W, int C, hipStream_t st) {
  const int Ho = H - 2, Wo = W - 2;
  std::vector<float> x((size_t)B * H * W * C), w((size_t)C * C * 9), b(C), ref((size_t)B * Ho * Wo * C), got(ref.size());
  for (auto& v : x) v = rnd() > 0.f ? rnd() : 0.f;  // (a ReLU output: zeros and positives and, here, negatives too)
  for (auto& v : w) v = 0.1f * rnd();
  for (auto& v : b) v = 0.1f * rnd();
  double scale = 0.0;
  for (int n = 0; n < B; ++n)
    for (int y = 0; y < Ho; ++y)
      for (int xx = 0; xx < Wo; ++xx)
        for (int co = 0; co < C; ++co) {
          double acc = b[co];
          for (int ci = 0; ci < C; ++ci)
            for (int dy = 0; dy < 3; ++dy)
              for (int dx = 0; dx < 3; ++dx)
                acc += (double)x[(((size_t)n * H + y + dy) * W + xx + dx) * C + ci] * w[((size_t)co * C + ci) * 9 + dy * 3 + dx];
          const float r = acc > 0.0 ? (float)acc : 0.f;
          ref[(((size_t)n * Ho + y) * Wo + xx) * C + co] = r;
          if (std::fabs(r) > scale) scale = std::fabs(r);
        }
  float *dx = to_device(x), *dw = to_device(w), *db = to_device(b), *dout = nullptr;
  if (!dx || !dw || !db) return 2;
  HIP_OK(hipMalloc(&dout, got.size() * sizeof(float)));
  const int rc = curla_conv3x3_s1_fwd(dx, dw, db, dout, B, H, W, C, (void*)st);
  if (rc != CURLA_OK) {
    std::printf("curla_conv3x3_s1_fwd(channels %d) returned %d\n", C, rc);
    return 1;
  }
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipMemcpy(got.data(), dout, got.size() * sizeof(float), hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (size_t i = 0; i < got.size(); ++i) worst = std::fmax(worst, std::fabs((double)got[i] - ref[i]));
  std::printf("conv3x3_s1_fwd B=%d %dx%d channels=%d: max|err| / max|ref| = %.3e\n", B, H, W, C, worst / scale);
  (void)hipFree(dx), (void)hipFree(dw), (void)hipFree(db), (void)hipFree(dout);
  return worst / scale <= 1e-4 ? 0 : 1;
}

int main() {
  int bad = 0;
  std::printf("%s, abi %d\n", curla_version(), curla_abi_version());
  if (curla_abi_version() != CURLA_ABI_VERSION) return 1;
  if (curla_set_option("s1_fwd", "nonsense") != CURLA_ERR_ARG || curla_set_option("s1_fwd", "auto") != CURLA_OK) return 1;
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));

  {  // target <- tau p + (1 - tau) target
    const size_t n = 100003;
    std::vector<float> p(n), t(n), want(n), got(n);
    const float tau = 0.05f, omt = 1.0f - tau;
    for (size_t i = 0; i < n; ++i) p[i] = rnd(), t[i] = rnd(), want[i] = tau * p[i] + omt * t[i];
    float *dp = to_device(p), *dt = to_device(t);
    if (!dp || !dt) return 2;
    if (curla_soft_update(dp, dt, n, tau, omt, (void*)st) != CURLA_OK) return 1;
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(got.data(), dt, n * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (size_t i = 0; i < n; ++i) worst = std::fmax(worst, std::fabs((double)got[i] - want[i]));
    std::printf("soft_update n=%zu: max|err| = %.3e\n", n, worst);
    bad += worst <= 1e-7 ? 0 : 1;
    (void)hipFree(dp), (void)hipFree(dt);
  }

  bad += check_conv(3, 13, 16, 32, st);  // the gfx950 row-walk kernels (bf16x3 behind Winograd F(2,3) by default)
  bad += check_conv(2, 9, 11, 16, st);   // another filter count: the generic path behind the same entry point

  {  // a float64 scalar through eight float32 words and back, and the exact mean of two of them
    const double a = 0.123456789012345678, b = -3.3e-5;
    double *dv = nullptr, out = 0.0;
    float* dwords = nullptr;
    HIP_OK(hipMalloc(&dv, sizeof(double)));
    HIP_OK(hipMalloc(&dwords, CURLA_F64_WORDS * sizeof(float)));
    float wa[CURLA_F64_WORDS], wb[CURLA_F64_WORDS], ws[CURLA_F64_WORDS];
    HIP_OK(hipMemcpy(dv, &a, sizeof(double), hipMemcpyHostToDevice));
    if (curla_f64_pack(dv, dwords, (void*)st) != CURLA_OK) return 1;
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(wa, dwords, sizeof(wa), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(dv, &b, sizeof(double), hipMemcpyHostToDevice));
    if (curla_f64_pack(dv, dwords, (void*)st) != CURLA_OK) return 1;
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(wb, dwords, sizeof(wb), hipMemcpyDeviceToHost));
    for (int j = 0; j < CURLA_F64_WORDS; ++j) ws[j] = wa[j] + wb[j];  // (what a SUM all-reduce of two ranks leaves)
    HIP_OK(hipMemcpy(dwords, ws, sizeof(ws), hipMemcpyHostToDevice));
    if (curla_f64_unpack(dwords, 1.0, 2.0, dv, (void*)st) != CURLA_OK) return 1;
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(&out, dv, sizeof(double), hipMemcpyDeviceToHost));
    const bool ok = out == (a + b) / 2.0;
    std::printf("f64 rider: mean of two ranks %s (a + b) / 2 in double (%.17g)\n", ok ? "==" : "!=", out);
    bad += ok ? 0 : 1;
    (void)hipFree(dv), (void)hipFree(dwords);
  }
  HIP_OK(hipStreamDestroy(st));
  std::printf(bad ? "FAILED (%d)\n" : "all checks passed\n", bad);
  return bad ? 1 : 0;
}
