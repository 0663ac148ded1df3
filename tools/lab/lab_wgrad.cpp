// Timing lab for the wide-tile weight-gradient kernel: includes the product source and times ablated instantiations with
// HIP events.  Built here (cross-compiled), run on the GPU box:  make -C tools/lab && gpurun -- tools/lab/lab_wgrad
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../../lafs_cvpr2024_amd/csrc/wgrad.hip"

extern "C" void lafs_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

template <int ABL>
void run(const char* name, int M, std::vector<std::pair<int, int>> dims, const bf16_t* A, const bf16_t* B, float* C, float* ws, int iters = 20) {
  std::vector<lafs_wgrad_item> it(dims.size());
  double fl = 0;
  for (size_t g = 0; g < dims.size(); ++g) {
    it[g] = {};
    it[g].A = A + (g * 7919) % 4096 * 8; it[g].lda = dims[g].first; it[g].B = B + (g * 104729) % 4096 * 8; it[g].ldb = dims[g].second;
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
  printf("%-34s abl%-2d M=%6d f%dx%d slices %3d tiles %3d stages %4d: %8.1f us  %7.1f TF/s\n", name, ABL, M, pl.fa, pl.fb, pl.slices, pl.tiles,
         pl.mlen / KB, us, fl / us / 1e6);
}

int main() {
  const int T = 44160;
  std::vector<uint16_t> h((size_t)8192 * 4096);
  unsigned x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((x >> 9) & 0x3ff) - ((x >> 20) & 1) * 0x8000 + 0x300); }   // ~[-2,2] bf16
  bf16_t *A, *B; float *ws, *C;
  hipMalloc(&A, (size_t)T * 1536 * 2 + (1 << 20)); hipMalloc(&B, (size_t)T * 1536 * 2 + (1 << 20)); hipMalloc(&ws, 256u << 20); hipMalloc(&C, 64u << 20);
  for (size_t off = 0; off < (size_t)T * 1536 * 2; off += h.size() * 2) {
    const size_t n = std::min(h.size() * 2, (size_t)T * 1536 * 2 - off);
    hipMemcpy((char*)A + off, h.data(), n, hipMemcpyHostToDevice); hipMemcpy((char*)B + off, h.data() + 777, n - 2000, hipMemcpyHostToDevice);
  }
  typedef std::vector<std::pair<int, int>> D;
#define ALL(name, M, ...)                              \
  run<0>(name, M, D{__VA_ARGS__}, A, B, C, ws);          \
  run<1>(name " noDMA", M, D{__VA_ARGS__}, A, B, C, ws); \
  run<25>(name " MFMAonly", M, D{__VA_ARGS__}, A, B, C, ws); \
  run<4>(name " DMAonly", M, D{__VA_ARGS__}, A, B, C, ws);
  ALL("ViT-S block (4 GEMMs)", T, {384, 1536}, {1536, 384}, {384, 384}, {1152, 384})
  ALL("fc1 alone", T, {1536, 384})
  ALL("fc1+fc2", T, {1536, 384}, {384, 1536})
  ALL("ViT-B block", 25216, {768, 2048}, {2048, 768}, {768, 704}, {2112, 768})
  return 0;
}
