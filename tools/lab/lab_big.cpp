// Timing + correctness lab for the 192x256 persistent NT GEMM (csrc/gemm_big.hip): includes the product source, checks it against a
// naive kernel on ragged shapes and times the Part-fViT shapes with HIP events.   make -C tools/lab lab_big && gpurun -- tools/lab/lab_big
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../lafs_cvpr2024_amd/csrc/gemm_big.hip"
static int g_big = 2;            // LAFS_OPT_NT_BIG: 2 = one wave per SIMD (192 x 256), 3 = two waves per SIMD (256 x 256)
int lafs_ctx_opt(const lafs_ctx*, int o) { static const int d[LAFS_OPT_COUNT] = {0, 2, 15, 4, 1, 1, 0, 1}; return o == LAFS_OPT_NT_BIG ? g_big : d[o]; }
extern "C" void lafs_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

__global__ void ref_nt(const bf16_t* A, const bf16_t* B, const float* bias, const float* resid, float* C, int M, int N, int K) {
  const int n = blockIdx.x * 16 + threadIdx.x, m = blockIdx.y * 16 + threadIdx.y;
  if (m >= M || n >= N) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc += bf2f(A[(size_t)m * K + k]) * bf2f(B[(size_t)n * K + k]);
  acc += bias ? bias[n] : 0.f;
  if (resid) acc += resid[(size_t)m * N + n];
  C[(size_t)m * N + n] = acc;
}

static std::vector<uint16_t> rnd(size_t n, unsigned seed, float scale) {
  std::vector<uint16_t> h(n);
  unsigned x = seed;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    const float f = ((int)(x >> 8) % 2001 - 1000) * 1e-3f * scale;
    unsigned u; memcpy(&u, &f, 4);
    v = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
  }
  return h;
}

static int check(int M, int N, int K, int epi) {
  std::vector<uint16_t> hA = rnd((size_t)M * K, 1, 1.f), hB = rnd((size_t)N * K, 2, 0.1f);
  bf16_t *A, *B, *C; float *bias, *resid, *Cf, *R;
  hipMalloc(&A, hA.size() * 2); hipMalloc(&B, hB.size() * 2); hipMalloc(&C, (size_t)(M + 8) * N * 2); hipMalloc(&Cf, (size_t)(M + 8) * N * 4);
  hipMalloc(&bias, N * 4); hipMalloc(&resid, (size_t)M * N * 4); hipMalloc(&R, (size_t)M * N * 4);
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  std::vector<float> hb(N), hr((size_t)M * N);
  for (int i = 0; i < N; ++i) hb[i] = 0.01f * (i % 37);
  for (size_t i = 0; i < hr.size(); ++i) hr[i] = 0.001f * (float)(i % 1013);
  hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(resid, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
  hipMemset(C, 0x7f, (size_t)(M + 8) * N * 2); hipMemset(Cf, 0x7f, (size_t)(M + 8) * N * 4);
  lafs_gemm_nt_args g = {};
  g.A = A; g.lda = K; g.B = B; g.ldb = K; g.M = M; g.N = N; g.K = K; g.epilogue = epi; g.bias = bias; g.splits = 1;
  if (epi == LAFS_EPI_RESID_F32) { g.C = Cf; g.ldc = N; g.resid = resid; g.ldr = N; }
  else { g.C = C; g.ldc = N; }
  const int rc = lafs_big_launch(&g, 0);
  hipLaunchKernelGGL(ref_nt, dim3((N + 15) / 16, (M + 15) / 16), dim3(16, 16), 0, 0, A, B, bias, epi == LAFS_EPI_RESID_F32 ? resid : nullptr, R, M, N, K);
  hipDeviceSynchronize();
  std::vector<float> ref((size_t)M * N), got((size_t)M * N);
  hipMemcpy(ref.data(), R, ref.size() * 4, hipMemcpyDeviceToHost);
  std::vector<uint16_t> gb((size_t)(M + 8) * N); std::vector<float> gf((size_t)(M + 8) * N);
  if (epi == LAFS_EPI_RESID_F32) hipMemcpy(gf.data(), Cf, gf.size() * 4, hipMemcpyDeviceToHost);
  else hipMemcpy(gb.data(), C, gb.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0; size_t bad = 0;
  for (size_t i = 0; i < ref.size(); ++i) {
    float v;
    if (epi == LAFS_EPI_RESID_F32) v = gf[i]; else { unsigned u = (unsigned)gb[i] << 16; memcpy(&v, &u, 4); }
    const double e = fabs((double)v - ref[i]);
    if (!(e <= 1e30)) ++bad;
    if (e > maxerr) maxerr = e;
    if (fabs(ref[i]) > maxref) maxref = fabs(ref[i]);
  }
  bool guard_ok = true;                                     // rows beyond M untouched
  for (size_t i = (size_t)M * N; i < (size_t)(M + 8) * N; ++i) {
    if (epi == LAFS_EPI_RESID_F32) { unsigned u; memcpy(&u, &gf[i], 4); guard_ok &= (u == 0x7f7f7f7fu); } else guard_ok &= (gb[i] == 0x7f7f);
  }
  const double tol = (epi == LAFS_EPI_RESID_F32 ? 2e-4 : 1e-2) * maxref;
  printf("check M=%d N=%d K=%d epi=%d rc=%d: max err %.3e (max |ref| %.3e, tol %.1e) nan %zu guard %s -> %s\n", M, N, K, epi, rc, maxerr, maxref, tol, bad,
         guard_ok ? "ok" : "OVERWRITTEN", (rc == 0 && maxerr <= tol && bad == 0 && guard_ok) ? "PASS" : "FAIL");
  hipFree(A); hipFree(B); hipFree(C); hipFree(Cf); hipFree(bias); hipFree(resid); hipFree(R);
  return (rc == 0 && maxerr <= tol && bad == 0 && guard_ok) ? 0 : 1;
}

static void timeit(const char* name, int M, int N, int K, int epi, int iters = 50) {
  bf16_t *A, *B, *C; float* Cf; float* resid;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
  hipMalloc(&Cf, (size_t)M * N * 4); hipMalloc(&resid, (size_t)M * N * 4);
  std::vector<uint16_t> h = rnd(1 << 22, 7, 1.f);
  for (size_t off = 0; off < (size_t)M * K * 2; off += h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min(h.size() * 2, (size_t)M * K * 2 - off), hipMemcpyHostToDevice);
  for (size_t off = 0; off < (size_t)N * K * 2; off += h.size() * 2) hipMemcpy((char*)B + off, h.data() + 99, std::min(h.size() * 2 - 200, (size_t)N * K * 2 - off), hipMemcpyHostToDevice);
  hipMemset(resid, 0, (size_t)M * N * 4);
  lafs_gemm_nt_args g = {};
  g.A = A; g.lda = K; g.B = B; g.ldb = K; g.M = M; g.N = N; g.K = K; g.epilogue = epi; g.splits = 1;
  if (epi == LAFS_EPI_RESID_F32) { g.C = Cf; g.ldc = N; g.resid = resid; g.ldr = N; } else { g.C = C; g.ldc = N; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < iters; ++i) lafs_big_launch(&g, 0);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) lafs_big_launch(&g, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters;
  const int btm = g_big == 3 ? GeoTwo::BTM : GeoOne::BTM, btn = g_big == 3 ? GeoTwo::BTN : GeoOne::BTN;
  const long tiles = (long)((M + btm - 1) / btm) * ((N + btn - 1) / btn);
  printf("[geo %d] %-28s M=%6d N=%5d K=%5d epi %d: %8.1f us  %7.1f TF/s  (%ld tiles = %.2f rounds)\n", g_big, name, M, N, K, epi, us, 2.0 * M * N * K / us / 1e6, tiles,
         tiles / 256.0);
  hipFree(A); hipFree(B); hipFree(C); hipFree(Cf); hipFree(resid);
}

int main() {
  int bad = 0;
  for (g_big = 2; g_big <= 3; ++g_big) {
    printf("---- geometry %d\n", g_big);
    bad += check(192 * 3 + 37, 768, 256, LAFS_EPI_BF16);
    bad += check(192 * 2 + 1, 704, 128, LAFS_EPI_BF16);
    bad += check(192 * 5, 512, 64, LAFS_EPI_BF16);
    bad += check(192 * 3 + 37, 768, 256, LAFS_EPI_RESID_F32);
    bad += check(40000, 768, 64, LAFS_EPI_BF16);             // many tiles per workgroup, one stage each
    bad += check(192 * 300 + 5, 256, 192, LAFS_EPI_BF16);
  }
  printf("%s\n", bad ? "CHECKS FAILED" : "all checks passed");
  for (g_big = 2; g_big <= 3; ++g_big) {
    timeit("Part-fViT fc1 dgrad", 44160, 768, 2048, LAFS_EPI_BF16);
    timeit("Part-fViT fc1 fwd (plain)", 44160, 2048, 768, LAFS_EPI_BF16);
    timeit("Part-fViT qkv fwd", 44160, 2112, 768, LAFS_EPI_BF16);
    timeit("Part-fViT qkv dgrad", 44160, 768, 2112, LAFS_EPI_BF16);
    timeit("Part-fViT proj dgrad", 44160, 704, 768, LAFS_EPI_BF16);
    timeit("Part-fViT fc2 fwd resid", 44160, 768, 2048, LAFS_EPI_RESID_F32);
    timeit("C4 fc1 dgrad", 25216, 768, 2048, LAFS_EPI_BF16);
    timeit("C4 fc1 fwd (plain)", 25216, 2048, 768, LAFS_EPI_BF16);
    timeit("square 8192", 8192, 8192, 8192, LAFS_EPI_BF16, 10);
    timeit("square 4096", 4096, 4096, 4096, LAFS_EPI_BF16, 20);
  }
  return bad;
}
