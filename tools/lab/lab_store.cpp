// Lab: what a burst of S global_store_dwordx4 instructions costs the issuing wave (s_memtime ticks), with W waves per CU doing the
// same, PAUSE ticks of VALU work between bursts (a GEMM epilogue stores 4-8 times per item, then computes for ~2-4 thousand ticks).
//   usage: lab_store            (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// HALF_LINES: 0 one contiguous KiB per instruction; 1 16 rows x 64 B, rows 256 B apart (the four instructions of a burst fill 4 KiB);
// 2 16 rows x 64 B, rows 3072 B apart (a row-major [M, 1536] bf16 output: what the GEMM epilogue writes).  MFMA: the pause is MFMAs.
template <int S, bool ONE_LANE, int HALF_LINES, bool MFMA = false>
__global__ void burst_kernel(unsigned char* out, unsigned long long* ticks, int bursts, int pause) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  unsigned long long acc = 0;
  float junk = (float)lane;
  for (int b = 0; b < bursts; ++b) {
    // 16 rows x 64 contiguous bytes per instruction (the GEMM epilogue pattern) or one contiguous KiB
    unsigned char* base = HALF_LINES == 2 ? out + (wave * 16) * 3072 * 2 + (size_t)(b % 12) * (S * 64) + (size_t)(b / 12) * (2048 * 16 * 3072 * 2)
                                          : out + ((wave * bursts + b) * S) * 1024;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int s = 0; s < S; ++s) {
      unsigned char* q = HALF_LINES == 2 ? base + (size_t)(lane & 15) * 3072 + s * 64 + (lane >> 4) * 16
                         : (HALF_LINES ? base + (size_t)(lane & 15) * (S * 64) + s * 64 + (lane >> 4) * 16 : base + s * 1024 + lane * 16);
      const u32x4 v = {(unsigned)b, (unsigned)s, (unsigned)lane, 0u};
      if (!ONE_LANE || lane == 0) *reinterpret_cast<u32x4*>(q) = v;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    acc += t1 - t0;
    if constexpr (MFMA) {
      f32x4 c0 = {junk, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
      const bf16x8 a = {(__bf16)1.0f, (__bf16)junk, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < pause / 64; ++i) {            // four independent 16-cycle MFMAs per trip
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c3, 0, 0, 0);
      }
      junk = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
      for (int i = 0; i < pause / 8; ++i) junk = junk * 1.0001f + 0.5f;      // ~8 ticks per dependent FMA at two waves per SIMD
    }
  }
  if (lane == 0) ticks[wave] = acc + (junk == 123.456f ? 1 : 0);
}

template <int S, bool ONE, int HALF, bool MFMA = false>
int run(const char* name, int waves_per_cu, int pause, unsigned char* out, unsigned long long* ticks) {
  const int bursts = 64, blocks = 256 * (waves_per_cu > 4 ? 2 : 1), threads = 64 * (waves_per_cu > 4 ? waves_per_cu / 2 : waves_per_cu);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((burst_kernel<S, ONE, HALF, MFMA>), dim3(blocks), dim3(threads), 0, 0, out, ticks, bursts, pause);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((burst_kernel<S, ONE, HALF, MFMA>), dim3(blocks), dim3(threads), 0, 0, out, ticks, bursts, pause);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = blocks * threads / 64;
  std::vector<unsigned long long> h(nw);
  CK(hipMemcpy(h.data(), ticks, nw * 8, hipMemcpyDeviceToHost));
  double sum = 0; for (auto v : h) sum += (double)v;
  const double bytes = (double)nw * bursts * S * (ONE ? 16 : 1024);
  printf("%-34s S=%d waves/CU=%d pause=%5d: %7.1f ticks per store instruction (burst %7.1f) | kernel %7.1f us, %6.2f TB/s\n", name, S, waves_per_cu,
         pause, sum / nw / bursts / S, sum / nw / bursts, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  unsigned char* out; unsigned long long* ticks;
  CK(hipMalloc(&out, (size_t)6 * 2048 * 16 * 3072 * 2 + (1 << 20))); CK(hipMalloc(&ticks, 4096 * 8));
  for (int pause : {0, 2000}) {
    for (int w : {1, 4, 8}) {
      run<4, false, 1>("16 rows x 64 B, rows 256 B apart", w, pause, out, ticks);
      run<4, false, 2>("16 rows x 64 B, rows 3072 B apart", w, pause, out, ticks);
      run<4, false, 0>("one contiguous KiB per instruction", w, pause, out, ticks);
      run<4, true, 1>("one lane only", w, pause, out, ticks);
      if (pause) {
        run<4, false, 1, true>("rows 256 B apart, MFMA pause", w, pause, out, ticks);
        run<4, false, 2, true>("rows 3072 B apart, MFMA pause", w, pause, out, ticks);
        run<4, true, 1, true>("one lane only, MFMA pause", w, pause, out, ticks);
      }
    }
  }
  return 0;
}
