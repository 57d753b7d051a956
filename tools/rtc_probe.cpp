// Developer probe: (1) hiprtc compiles + launches a kernel on this box, (2) device f32/f64
// div and sqrt and fma are IEEE-correctly rounded (compared bitwise against the host).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
static const char *src = R"(
extern "C" __global__ void ops(const double* a, const double* b, double* o, const float* fa, const float* fb, float* fo, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  o[i] = a[i] / b[i];
  o[n + i] = __builtin_sqrt(__builtin_fabs(a[i]));
  o[2*n + i] = __builtin_fma(a[i], b[i], a[i]);
  o[3*n + i] = a[i] * b[i] + a[i];          // must NOT be contracted
  fo[i] = fa[i] / fb[i];
  fo[n + i] = __builtin_sqrtf(__builtin_fabsf(fa[i]));
  fo[2*n + i] = fa[i] * fb[i] + fa[i];
  fo[3*n + i] = (float)__builtin_sqrt((double)__builtin_fabsf(fa[i]));
}
)";
int main() {
  auto t0 = std::chrono::steady_clock::now();
  hiprtcProgram p;
  if (hiprtcCreateProgram(&p, src, "ops.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { puts("create failed"); return 1; }
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off"};
  hiprtcResult r = hiprtcCompileProgram(p, 3, opts);
  size_t ls = 0; hiprtcGetProgramLogSize(p, &ls); std::string log(ls, 0); if (ls) hiprtcGetProgramLog(p, &log[0]);
  if (r != HIPRTC_SUCCESS) { printf("hiprtc compile failed: %s\n", log.c_str()); return 1; }
  size_t cs = 0; hiprtcGetCodeSize(p, &cs); std::vector<char> code(cs); hiprtcGetCode(p, code.data());
  auto t1 = std::chrono::steady_clock::now();
  printf("hiprtc compile ok: %zu bytes in %.1f ms\n", cs, std::chrono::duration<double, std::milli>(t1 - t0).count());
  hipModule_t mod; hipFunction_t fn;
  CK(hipModuleLoadData(&mod, code.data()));
  CK(hipModuleGetFunction(&fn, mod, "ops"));
  const int n = 1 << 20;
  std::mt19937_64 rng(1234);
  std::vector<double> a(n), b(n), o(4 * n); std::vector<float> fa(n), fb(n), fo(4 * n);
  for (int i = 0; i < n; ++i) {
    uint64_t ua = rng(), ub = rng();
    // random bit patterns with moderate exponents + some extremes
    ua = (ua & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 40 + (rng() % 80)) << 52);
    ub = (ub & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 40 + (rng() % 80)) << 52);
    if (i % 1000 == 0) { ua = (ua & ~(0x7FFull << 52)) | ((uint64_t)(rng() % 2046 + 1) << 52); }
    memcpy(&a[i], &ua, 8); memcpy(&b[i], &ub, 8);
    uint32_t fu = (uint32_t)rng(), fv = (uint32_t)rng();
    fu = (fu & 0x807FFFFFu) | ((uint32_t)(127 - 30 + (rng() % 60)) << 23);
    fv = (fv & 0x807FFFFFu) | ((uint32_t)(127 - 30 + (rng() % 60)) << 23);
    if (i % 1000 == 1) { fu = (fu & 0x807FFFFFu) | ((uint32_t)(rng() % 254 + 1) << 23); fv = (fv & 0x807FFFFFu) | ((uint32_t)(rng() % 254 + 1) << 23); }
    memcpy(&fa[i], &fu, 4); memcpy(&fb[i], &fv, 4);
  }
  double *da, *db, *dout; float *dfa, *dfb, *dfo;
  CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dout, 4 * n * 8));
  CK(hipMalloc(&dfa, n * 4)); CK(hipMalloc(&dfb, n * 4)); CK(hipMalloc(&dfo, 4 * n * 4));
  CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dfa, fa.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dfb, fb.data(), n * 4, hipMemcpyHostToDevice));
  int nn = n; void *args[] = {&da, &db, &dout, &dfa, &dfb, &dfo, &nn};
  CK(hipModuleLaunchKernel(fn, n / 256, 1, 1, 256, 1, 1, 0, nullptr, args, nullptr));
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(fo.data(), dfo, 4 * n * 4, hipMemcpyDeviceToHost));
  long bad[8] = {0};
  for (int i = 0; i < n; ++i) {
    volatile double q = a[i] / b[i], s = std::sqrt(std::fabs(a[i])), f = std::fma(a[i], b[i], a[i]);
    volatile double pm = a[i] * b[i]; volatile double pa = pm + a[i];
    volatile float fq = fa[i] / fb[i], fs = std::sqrt(std::fabs(fa[i])); volatile float fm = fa[i] * fb[i]; volatile float fpa = fm + fa[i];
    volatile float fs2 = (float)std::sqrt((double)std::fabs(fa[i]));
    double hq = q, hs = s, hf = f, hpa = pa; float hfq = fq, hfs = fs, hfpa = fpa, hfs2 = fs2;
    bad[0] += memcmp(&hq, &o[i], 8) != 0; bad[1] += memcmp(&hs, &o[n + i], 8) != 0;
    bad[2] += memcmp(&hf, &o[2 * n + i], 8) != 0; bad[3] += memcmp(&hpa, &o[3 * n + i], 8) != 0;
    bad[4] += memcmp(&hfq, &fo[i], 4) != 0; bad[5] += memcmp(&hfs, &fo[n + i], 4) != 0;
    bad[6] += memcmp(&hfpa, &fo[2 * n + i], 4) != 0; bad[7] += memcmp(&hfs2, &fo[3 * n + i], 4) != 0;
  }
  printf("mismatches of %d: f64 div %ld sqrt %ld fma %ld mul+add %ld | f32 div %ld sqrt %ld mul+add %ld sqrt-via-f64 %ld\n",
         n, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], bad[7]);
  return 0;
}
