// Timing lab for the K-resident GEMM (csrc/gemm_kres.hip): includes the product source and times ablated instantiations with
// HIP events (no Python in the loop).  Built here (cross-compiled), run on the GPU box:
//   make -C tools/lab lab_kres && gpurun -- tools/lab/lab_kres
// ABL bits: 1 no stores, 2 no MFMA, 4 no LDS-DMA after the prologue, 8 no epilogue math, 16 no fragment reads
#include <cstdarg>
#include <cstdio>
#include <vector>
#define LAFS_KRES_LAB
#include "../../lafs_cvpr2024_amd/csrc/gemm_kres.hip"

#include "../../lafs_cvpr2024_amd/csrc/ctx.hpp"
int lafs_ctx_opt(const lafs_ctx*, int o) { static const int d[LAFS_OPT_COUNT] = {0, 2, 15, 4, 1, 1, 0, 1}; return d[o]; }
extern "C" void lafs_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

struct Bufs { bf16_t *A, *W, *C, *C2, *aux; float *bias, *resid, *Cf; };

template <int ABL>
void run(const char* name, int epi, int M, int N, const Bufs& b, int grid = 0, int iters = 30) {
  lafs_gemm_nt_args g = {};
  g.A = b.A; g.lda = 384; g.B = b.W; g.ldb = 384; g.M = M; g.N = N; g.K = 384; g.epilogue = epi;
  g.C = (epi == LAFS_EPI_RESID_F32) ? (void*)b.Cf : (void*)b.C; g.ldc = N; g.C2 = b.C2; g.ldc2 = N; g.bias = b.bias;
  g.resid = b.resid; g.ldr = N; g.aux = b.aux; g.ldaux = N; g.splits = 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kres_launch<ABL>(&g, 0, grid);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) kres_launch<ABL>(&g, 0, grid);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const float us = ms * 1e3f / iters;
  printf("%-12s abl%-2d grid %4d M=%6d N=%5d: %8.1f us  %7.1f TF/s\n", name, ABL, grid, M, N, us, 2.0 * M * N * 384 / us / 1e6);
}

template <int ABL>
void phases(const char* name, int epi, int M, int N, const Bufs& b, unsigned long long* dst) {
  lafs_gemm_nt_args g = {};
  g.A = b.A; g.lda = 384; g.B = b.W; g.ldb = 384; g.M = M; g.N = N; g.K = 384; g.epilogue = epi;
  g.C = (epi == LAFS_EPI_RESID_F32) ? (void*)b.Cf : (void*)b.C; g.ldc = N; g.C2 = b.C2; g.ldc2 = N; g.bias = b.bias;
  g.resid = b.resid; g.ldr = N; g.aux = b.aux; g.ldaux = N; g.splits = 1;
  hipMemset(dst, 0, 4096 * 64);
  for (int i = 0; i < 3; ++i) kres_launch<ABL>(&g, 0, 0, dst);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(4096 * 8);
  hipMemcpy(h.data(), dst, 4096 * 64, hipMemcpyDeviceToHost);
  double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int n = 0;
  for (int w = 0; w < 4096; ++w) if (h[w * 8 + 5]) { for (int k = 0; k < 8; ++k) sum[k] += (double)h[w * 8 + k]; ++n; }
  printf("%-12s abl%-2d workgroups %4d, steps/wg %.1f: per step [ticks of s_memtime] wait+barrier %.0f  issue %.0f  mfma %.0f  epilogue %.0f | whole run %.0f ticks, of which %.2f reloads of %.0f ticks each\n",
         name, ABL, n, sum[5] / n, sum[0] / (sum[5] - sum[7]), sum[1] / sum[5], sum[2] / sum[5], sum[3] / sum[5], sum[4] / n, sum[7] / n, sum[6] / sum[7]);
}

int main() {
  const int T = 44160;
  std::vector<uint16_t> h((size_t)8192 * 4096);
  unsigned x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((x >> 9) & 0x3ff) - ((x >> 20) & 1) * 0x8000 + 0x300); }   // ~[-2,2] bf16
  Bufs b;
  const size_t big = (size_t)T * 1536 * 2 + (1 << 20);
  hipMalloc(&b.A, big); hipMalloc(&b.W, 4 << 20); hipMalloc(&b.C, big); hipMalloc(&b.C2, big); hipMalloc(&b.aux, big);
  hipMalloc(&b.bias, 8192); hipMalloc(&b.resid, (size_t)T * 384 * 4 + 4096); hipMalloc(&b.Cf, (size_t)T * 384 * 4 + 4096);
  for (size_t off = 0; off < (size_t)T * 1536 * 2; off += h.size() * 2) {
    const size_t n = std::min(h.size() * 2, (size_t)T * 1536 * 2 - off);
    hipMemcpy((char*)b.A + off, h.data(), n, hipMemcpyHostToDevice); hipMemcpy((char*)b.aux + off, h.data() + 777, n - 2000, hipMemcpyHostToDevice);
  }
  hipMemcpy(b.W, h.data() + 12345, 4 << 20, hipMemcpyHostToDevice);
  hipMemset(b.bias, 0, 8192); hipMemset(b.resid, 0, (size_t)T * 384 * 4);
  unsigned long long* stamps; hipMalloc(&stamps, 4096 * 64);
  phases<32>("qkv", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 31>("qkv nothing", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 2>("qkv noMFMA", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 31 + 64>("qkv nothing nobar", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 30 + 128>("qkv nothing nost", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 64>("qkv nobar", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 128>("qkv nost", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 4>("qkv nodma", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 16>("qkv nofrag", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 4 + 128>("qkv nodma nost", LAFS_EPI_BF16, T, 1152, b, stamps);
  phases<32 + 512>("qkv lane reload", LAFS_EPI_BF16, T, 1152, b, stamps);
  run<512>("qkv lane reload", LAFS_EPI_BF16, T, 1152, b); run<0>("qkv", LAFS_EPI_BF16, T, 1152, b);
  run<512>("fc1 lane reload", LAFS_EPI_BF16_GELU, T, 1536, b); run<0>("fc1", LAFS_EPI_BF16_GELU, T, 1536, b);
  run<512>("dgelu lane reload", LAFS_EPI_DGELU_BF16, T, 1536, b); run<0>("dgelu", LAFS_EPI_DGELU_BF16, T, 1536, b);
  run<512 + 2048>("dgelu lane+ahead", LAFS_EPI_DGELU_BF16, T, 1536, b);
  run<512>("proj lane reload", LAFS_EPI_RESID_F32, T, 384, b); run<0>("proj", LAFS_EPI_RESID_F32, T, 384, b);
  phases<32>("fc1", LAFS_EPI_BF16_GELU, T, 1536, b, stamps);
  phases<32>("dgelu", LAFS_EPI_DGELU_BF16, T, 1536, b, stamps);
  phases<32>("proj", LAFS_EPI_RESID_F32, T, 384, b, stamps);
  if (getenv("LAB_PHASES_ONLY")) return 0;
#define ALL(name, epi, N)                     \
  run<0>(name, epi, T, N, b);                 \
  run<1>(name " noST", epi, T, N, b);         \
  run<2>(name " noMFMA", epi, T, N, b);       \
  run<4>(name " noDMA", epi, T, N, b);        \
  run<8>(name " noEPI", epi, T, N, b);        \
  run<3>(name " noST noMFMA", epi, T, N, b);  \
  run<7>(name " noST/MF/DMA", epi, T, N, b);  \
  run<23>(name " +noFRAG", epi, T, N, b);     \
  run<31>(name " nothing", epi, T, N, b);
  ALL("qkv", LAFS_EPI_BF16, 1152)
  ALL("fc1", LAFS_EPI_BF16_GELU, 1536)
  ALL("dgelu", LAFS_EPI_DGELU_BF16, 1536)
  ALL("proj", LAFS_EPI_RESID_F32, 384)
  for (int grid : {256, 512, 768, 1024, 2048}) { run<0>("qkv", LAFS_EPI_BF16, T, 1152, b, grid); run<0>("fc1", LAFS_EPI_BF16_GELU, T, 1536, b, grid); }
  return 0;
}
