// Occupancy of the NT GEMM variants as the runtime sees it (workgroups per CU), next to the resources each one asks for.
#include "gemm.hip"
#include <cstdio>
#include <cstdarg>
extern "C" void lafs_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
template <typename K> void occ(const char* name, K kern, int threads) {
  int n = -1; hipFuncAttributes a{};
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, 0);
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kern));
  printf("%-40s threads %4d  regs %3d  LDS %6zu B  -> %d workgroups/CU (%d waves/SIMD)\n", name, threads, a.numRegs, a.sharedSizeBytes, n,
         n * threads / 256);
}
int main() {
  occ("nt<BF16_GELU, WM4, BK32, WN2> 256x128", gemm_nt_kernel<EPI_BF16_GELU, 4, 32, 2>, 512);
  occ("nt<BF16, WM4, BK32, WN2> 256x128", gemm_nt_kernel<EPI_BF16, 4, 32, 2>, 512);
  occ("nt<BF16, WM2, BK32, WN2> 128x128", gemm_nt_kernel<EPI_BF16, 2, 32, 2>, 256);
  occ("nt<BF16, WM2, BK64, WN2> 128x128", gemm_nt_kernel<EPI_BF16, 2, 64, 2>, 256);
  occ("nt<RESID_F32, WM2, BK64, WN2> 128x128", gemm_nt_kernel<EPI_RESID_F32, 2, 64, 2>, 256);
  occ("nt<DGELU, WM4, BK32, WN2> 256x128", gemm_nt_kernel<EPI_DGELU_BF16, 4, 32, 2>, 512);
  return 0;
}
