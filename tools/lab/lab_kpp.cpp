// Timing + identity lab for the ping-pong K-resident GEMM (csrc/gemm_kpp.hip) against the two-workgroup kernel (csrc/gemm_kres.hip):
// both product sources compiled into one binary, same operands, outputs compared byte for byte (the accumulation order per output
// element is the same in both kernels), HIP-event timings interleaved, phase stamps of the ping-pong kernel.
//   make -C tools/lab lab_kpp && gpurun -- tools/lab/lab_kpp
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#define LAFS_KRES_LAB
#define LAFS_KPP_LAB
#include "../../lafs_cvpr2024_amd/csrc/gemm_kres.hip"
#include "gemm_kpp.hip"

#include "../../lafs_cvpr2024_amd/csrc/ctx.hpp"
int lafs_ctx_opt(const lafs_ctx*, int o) { static const int d[LAFS_OPT_COUNT] = {0, 2, 15, 4, 1, 1, 0, 1}; return d[o]; }
extern "C" void lafs_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

struct Bufs { bf16_t *A, *W, *C, *C2, *aux; float *bias, *resid, *Cf; int* row2seq; float* scale; };
static const int T = 44160;
static const size_t OUT_BYTES = (size_t)T * 1536 * 2;

static lafs_gemm_nt_args mk(int epi, int M, int N, const Bufs& b, int act, bool teacher) {
  lafs_gemm_nt_args g = {};
  g.A = b.A; g.lda = 384; g.B = b.W; g.ldb = 384; g.M = M; g.N = N; g.K = 384; g.epilogue = epi;
  g.C = (epi == LAFS_EPI_RESID_F32) ? (void*)b.Cf : (void*)b.C; g.ldc = N; g.C2 = b.C2; g.ldc2 = N; g.bias = b.bias;
  if (teacher) g.C = nullptr;
  g.resid = b.resid; g.ldr = N; g.aux = b.aux; g.ldaux = N; g.splits = 1; g.act = act;
  if (epi == LAFS_EPI_RESID_F32) { g.seq_scale = b.scale; g.row2seq = b.row2seq; }
  return g;
}

static void snapshot(const Bufs& b, std::vector<unsigned char>& h) {
  h.resize(2 * OUT_BYTES);
  hipMemcpy(h.data(), b.C, OUT_BYTES, hipMemcpyDeviceToHost);
  hipMemcpy(h.data() + OUT_BYTES, b.C2, OUT_BYTES, hipMemcpyDeviceToHost);
}
static void snapshot_f(const Bufs& b, std::vector<unsigned char>& h) {
  h.resize((size_t)T * 384 * 4);
  hipMemcpy(h.data(), b.Cf, h.size(), hipMemcpyDeviceToHost);
}

template <typename F> static float time_us(F&& f, int warm, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < warm; ++i) f();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms * 1e3f / iters;
}

static void compare(const char* name, int epi, int M, int N, const Bufs& b, int act = 0, bool teacher = false, int grid = 0) {
  lafs_gemm_nt_args g = mk(epi, M, N, b, act, teacher);
  std::vector<unsigned char> r0, r1;
  const bool f32 = epi == LAFS_EPI_RESID_F32;
  hipMemset(b.C, 0x11, OUT_BYTES); hipMemset(b.C2, 0x11, OUT_BYTES); hipMemset(b.Cf, 0x11, (size_t)T * 384 * 4);
  kres_launch<0>(&g, 0); hipDeviceSynchronize();
  if (f32) snapshot_f(b, r0); else snapshot(b, r0);
  hipMemset(b.C, 0x11, OUT_BYTES); hipMemset(b.C2, 0x11, OUT_BYTES); hipMemset(b.Cf, 0x11, (size_t)T * 384 * 4);
  kpp::kpp_launch<0>(&g, 0, grid); hipDeviceSynchronize();
  hipError_t e = hipGetLastError();
  if (f32) snapshot_f(b, r1); else snapshot(b, r1);
  size_t diff = 0, first = 0;
  for (size_t i = 0; i < r0.size(); ++i) if (r0[i] != r1[i]) { if (!diff) first = i; ++diff; }
  float t_res = 0, t_pp = 0;
  for (int rep = 0; rep < 2; ++rep) {   // interleaved, second round reported (clocks settled)
    t_res = time_us([&] { kres_launch<0>(&g, 0); }, 50, 150);
    t_pp = time_us([&] { kpp::kpp_launch<0>(&g, 0, grid); }, 50, 150);
  }
  printf("%-16s M=%6d N=%5d grid %3d: kres %7.1f us   kpp %7.1f us   ratio %.3f   %s (%zu differing bytes, first at %zu) %s\n", name, M, N, grid, t_res, t_pp,
         t_pp / t_res, diff ? "MISMATCH" : "identical", diff, first, e == hipSuccess ? "" : hipGetErrorString(e));
  fflush(stdout);
}

template <int ABL>
static void abl(const char* name, int epi, int M, int N, const Bufs& b, int act = 0) {
  lafs_gemm_nt_args g = mk(epi, M, N, b, act, false);
  float us = 0;
  for (int rep = 0; rep < 2; ++rep) us = time_us([&] { kpp::kpp_launch<ABL>(&g, 0, 0); }, 50, 150);
  printf("%-22s abl%-3d M=%6d N=%5d: %7.1f us\n", name, ABL, M, N, us);
  fflush(stdout);
}

template <int ABL>
static void phases(const char* name, int epi, int M, int N, const Bufs& b, unsigned long long* dst, int act = 0) {
  lafs_gemm_nt_args g = mk(epi, M, N, b, act, false);
  hipMemset(dst, 0, 4096 * 64);
  for (int i = 0; i < 3; ++i) kpp::kpp_launch<ABL>(&g, 0, 0, dst);
  hipDeviceSynchronize();
  const float us = time_us([&] { kpp::kpp_launch<ABL>(&g, 0, 0, dst); }, 20, 60);
  std::vector<unsigned long long> h(512 * 12);
  hipMemcpy(h.data(), dst, 512 * 12 * 8, hipMemcpyDeviceToHost);
  for (int half = 0; half < 2; ++half) {
    double sum[12] = {0}; int n = 0;
    for (int w = 0; w < 256; ++w) { const unsigned long long* o = &h[(w * 2 + half) * 12]; if (o[7]) { for (int k = 0; k < 12; ++k) sum[k] += (double)o[k]; ++n; } }
    const double it = sum[7];
    printf("%-14s abl%-3d half %c wgs %3d items/wg %.1f: per item [ticks] issue+bias %.0f  mfma %.0f  wait %.0f | epilogue %.0f  vmcnt %.0f  wait %.0f | run %.0f ticks, %.2f segments, prologue %.0f each | %.1f us\n",
           name, ABL, half ? 'B' : 'A', n, it / n, sum[0] / it, sum[1] / it, sum[2] / it, sum[3] / it, sum[4] / it, sum[5] / it, sum[6] / n, sum[9] / n, sum[8] / sum[9], us);
  }
  fflush(stdout);
}

int main() {
  std::vector<uint16_t> h((size_t)8192 * 4096);
  unsigned x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((x >> 9) & 0x3ff) - ((x >> 20) & 1) * 0x8000 + 0x300); }   // ~[-2,2] bf16
  Bufs b;
  const size_t big = OUT_BYTES + (1 << 20);
  hipMalloc(&b.A, big); hipMalloc(&b.W, 4 << 20); hipMalloc(&b.C, big); hipMalloc(&b.C2, big); hipMalloc(&b.aux, big);
  hipMalloc(&b.bias, 8192); hipMalloc(&b.resid, (size_t)T * 384 * 4 + 4096); hipMalloc(&b.Cf, (size_t)T * 384 * 4 + 4096);
  hipMalloc(&b.row2seq, T * 4); hipMalloc(&b.scale, 4096);
  for (size_t off = 0; off < OUT_BYTES; off += h.size() * 2) {
    const size_t n = std::min(h.size() * 2, OUT_BYTES - off);
    hipMemcpy((char*)b.A + off, h.data(), n, hipMemcpyHostToDevice); hipMemcpy((char*)b.aux + off, h.data() + 777, n - 2000, hipMemcpyHostToDevice);
  }
  hipMemcpy(b.W, h.data() + 12345, 4 << 20, hipMemcpyHostToDevice);
  {
    std::vector<float> f(2048); for (int i = 0; i < 2048; ++i) f[i] = 0.01f * (float)((i * 37) % 101 - 50);
    hipMemcpy(b.bias, f.data(), 8192, hipMemcpyHostToDevice);
    std::vector<float> r((size_t)T * 384); for (size_t i = 0; i < r.size(); ++i) r[i] = 0.001f * (float)((i * 7919) % 2003) - 1.0f;
    hipMemcpy(b.resid, r.data(), r.size() * 4, hipMemcpyHostToDevice);
    std::vector<int> s(T); for (int i = 0; i < T; ++i) s[i] = i / 197 % 1000;
    hipMemcpy(b.row2seq, s.data(), T * 4, hipMemcpyHostToDevice);
    std::vector<float> sc(1024); for (int i = 0; i < 1024; ++i) sc[i] = (i % 10 == 3) ? 0.f : 1.0f / 0.9f;
    hipMemcpy(b.scale, sc.data(), 4096, hipMemcpyHostToDevice);
  }
  unsigned long long* stamps; hipMalloc(&stamps, 4096 * 64);

  for (int M : {44160}) {
    compare("qkv", LAFS_EPI_BF16, M, 1152, b);
    compare("fc1 (u + gelu)", LAFS_EPI_BF16_GELU, M, 1536, b);
    compare("fc1 save-grad", LAFS_EPI_BF16_GELU, M, 1536, b, LAFS_GELU_SAVE_GRAD);
    compare("fc1 teacher", LAFS_EPI_BF16_GELU, M, 1536, b, 0, true);
    compare("proj resid", LAFS_EPI_RESID_F32, M, 384, b);
    compare("dgelu (u)", LAFS_EPI_DGELU_BF16, M, 1536, b);
    compare("dgelu saved", LAFS_EPI_DGELU_BF16, M, 1536, b, LAFS_GELU_SAVE_GRAD);
    compare("proj dgrad", LAFS_EPI_BF16, M, 384, b);
  }
  abl<0>("qkv", LAFS_EPI_BF16, T, 1152, b);
  abl<2>("qkv noMFMA", LAFS_EPI_BF16, T, 1152, b);
  abl<16>("qkv noFRAG", LAFS_EPI_BF16, T, 1152, b);
  abl<18>("qkv noMFMA noFRAG", LAFS_EPI_BF16, T, 1152, b);
  abl<1>("qkv noST", LAFS_EPI_BF16, T, 1152, b);
  abl<17>("qkv noST noFRAG", LAFS_EPI_BF16, T, 1152, b);
  abl<19>("qkv noST noFRAG noMFMA", LAFS_EPI_BF16, T, 1152, b);
  abl<0>("fc1", LAFS_EPI_BF16_GELU, T, 1536, b, 1);
  abl<2>("fc1 noMFMA", LAFS_EPI_BF16_GELU, T, 1536, b, 1);
  abl<16>("fc1 noFRAG", LAFS_EPI_BF16_GELU, T, 1536, b, 1);
  abl<18>("fc1 noMFMA noFRAG", LAFS_EPI_BF16_GELU, T, 1536, b, 1);
  abl<8>("fc1 noEPI", LAFS_EPI_BF16_GELU, T, 1536, b, 1);
  abl<1>("fc1 noST", LAFS_EPI_BF16_GELU, T, 1536, b, 1);
  abl<9>("fc1 noST noEPI", LAFS_EPI_BF16_GELU, T, 1536, b, 1);
  abl<27>("fc1 nothing", LAFS_EPI_BF16_GELU, T, 1536, b, 1);
  phases<32>("qkv", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32>("fc1 save-grad", LAFS_EPI_BF16_GELU, T, 1536, b, stamps, LAFS_GELU_SAVE_GRAD);
  phases<32>("dgelu saved", LAFS_EPI_DGELU_BF16, T, 1536, b, stamps, LAFS_GELU_SAVE_GRAD);
  phases<32>("proj resid", LAFS_EPI_RESID_F32, T, 384, b, stamps);
  phases<32 + 256>("qkv FD12", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 512>("qkv FD16", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 512>("fc1 FD16", LAFS_EPI_BF16_GELU, T, 1536, b, stamps, LAFS_GELU_SAVE_GRAD);
  phases<32 + 1>("fc1 noST", LAFS_EPI_BF16_GELU, T, 1536, b, stamps, LAFS_GELU_SAVE_GRAD);
  phases<32 + 2>("fc1 noMFMA", LAFS_EPI_BF16_GELU, T, 1536, b, stamps, LAFS_GELU_SAVE_GRAD);
  phases<32 + 8>("fc1 noEPI", LAFS_EPI_BF16_GELU, T, 1536, b, stamps, LAFS_GELU_SAVE_GRAD);
  phases<32 + 11>("fc1 nothing", LAFS_EPI_BF16_GELU, T, 1536, b, stamps, LAFS_GELU_SAVE_GRAD);
  for (int grid : {128, 192, 224, 248, 256}) compare("fc1 save-grad", LAFS_EPI_BF16_GELU, T, 1536, b, LAFS_GELU_SAVE_GRAD, false, grid);
  return 0;
}
