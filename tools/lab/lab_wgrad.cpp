// Timing lab for the wide-tile weight-gradient kernel: includes the product source and times ablated instantiations with
// HIP events.  Built here (cross-compiled), run on the GPU box:  make -C tools/lab && gpurun -- tools/lab/lab_wgrad
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define LAFS_LAB_WGRAD_OCC2 1
#include "../../lafs_cvpr2024_amd/csrc/wgrad.hip"

extern "C" void lafs_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

template <int ABL>
void run(const char* name, int M, std::vector<std::pair<int, int>> dims, const bf16_t* A, const bf16_t* B, float* C, float* ws, int iters = 20) {
  if (getenv("LAB_ITERS")) iters = atoi(getenv("LAB_ITERS"));
  std::vector<lafs_wgrad_item> it(dims.size());
  double fl = 0;
  size_t offA = 0, offB = 0;
  for (size_t g = 0; g < dims.size(); ++g) {
    it[g] = {};
    it[g].A = A + offA; it[g].lda = dims[g].first; it[g].B = B + offB; it[g].ldb = dims[g].second;       // every operand its own tensor
    static const size_t al = getenv("LAB_ALIGN") ? (size_t)atol(getenv("LAB_ALIGN")) / 2 : 8;       // elements
    static const size_t skew = getenv("LAB_SKEW") ? (size_t)atol(getenv("LAB_SKEW")) / 2 : 0;
    offA = (offA + (size_t)M * dims[g].first + al - 1) / al * al + skew * (2 * g + 1);
    offB = (offB + (size_t)M * dims[g].second + al - 1) / al * al + skew * (2 * g + 2);
    if (getenv("LAB_ACC")) { it[g].accumulate = 1; it[g].colsum_a = C + (60 << 18) + g * 4096; }
    it[g].C = C + g * (4 << 20); it[g].ldc = dims[g].second; it[g].N1 = dims[g].first; it[g].N2 = dims[g].second; it[g].accumulate = 0;
    fl += 2.0 * M * dims[g].first * dims[g].second;
  }
  const Plan pl = make_plan(it.data(), (int)it.size(), M, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) group_impl<ABL>(it.data(), (int)it.size(), M, 0, ws, 256 << 20, 0);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) group_impl<ABL>(it.data(), (int)it.size(), M, 0, ws, 256 << 20, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const float us = ms * 1e3f / iters;
  printf("%-34s abl%-4d f%dx%d sl %2d st %3d: %7.1f us %7.1f TF/s\n", name, ABL, pl.fa, pl.fb, pl.slices, pl.mlen / KB, us, fl / us / 1e6);
}

int main() {
  const int T = 44160;
  std::vector<uint16_t> h((size_t)8192 * 4096);
  unsigned x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((x >> 9) & 0x3ff) - ((x >> 20) & 1) * 0x8000 + 0x300); }   // ~[-2,2] bf16
  bf16_t *A, *B; float *ws, *C;
  const size_t OPB = (size_t)T * 3584 * 2 + (64 << 20);
  hipMalloc(&A, OPB + (1 << 20)); hipMalloc(&B, OPB + (1 << 20)); hipMalloc(&ws, 256u << 20); hipMalloc(&C, 64u << 20);
  for (size_t off = 0; off < OPB; off += h.size() * 2) {
    const size_t n = std::min(h.size() * 2, OPB - off);
    hipMemcpy((char*)A + off, h.data(), n, hipMemcpyHostToDevice); hipMemcpy((char*)B + off, h.data() + 777, n - 2000, hipMemcpyHostToDevice);
  }
  typedef std::vector<std::pair<int, int>> D;
#define V(abl, tag) run<abl>("ViT-S block " tag, T, D{{384, 1536}, {1536, 384}, {384, 384}, {1152, 384}}, A, B, C, ws);
  V(0, "product")
  if (!getenv("LAB_SHORT")) {
    V(2048, "lab (same code)") V(1, "noDMA") V(25, "MFMAonly") V(4, "DMAonly") V(36, "DMAonly slice0 rows") V(68, "DMAonly same rows")
    V(64, "same rows") V(256, "nt") V(512, "sc1") V(768, "sc0 sc1") V(1024, "sc0") V(256 + 4, "DMAonly nt") V(512 + 4, "DMAonly sc1")
  }
  return 0;
}
