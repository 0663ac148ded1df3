// MFMA issue-rate probe: how many cycles per v_mfma_f32_32x32x16_bf16 does ONE wave per SIMD sustain from compiled HIP code?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int NA, int NB, int WAVES_PER_SIMD, int VARIANT>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void probe(const uint4* in, float* out, int iters) {
  f32x16_t acc[NA][NB];
  bf16x8_t a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = __builtin_bit_cast(bf16x8_t, in[threadIdx.x + 256 * i]);
  for (int i = 0; i < NB; ++i) b[i] = __builtin_bit_cast(bf16x8_t, in[threadIdx.x + 256 * (NA + i)]);
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (VARIANT == 1) __builtin_amdgcn_sched_barrier(0);
      }
    if (VARIANT == 2) { asm volatile("" : "+v"(a[0])); }
  }
  float s = 0.f;
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NA, int NB, int W, int V>
void run(const char* name, const uint4* in, float* out, int grid) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NA, NB, W, V>), dim3(grid), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<NA, NB, W, V>), dim3(grid), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * NA * NB;            // MFMAs per wave
  const double flops = n * 32768.0 * grid * 4;
  printf("%-40s grid %4d: %8.1f us  %7.1f TF/s  %6.1f ns per MFMA per wave (x%d waves/SIMD resident)\n", name, grid, ms * 1e3, flops / ms / 1e9,
         ms * 1e6 / n, W);
}

int main() {
  uint4* in; float* out;
  hipMalloc(&in, 256 * 16 * 16); hipMalloc(&out, 4096 * 256 * 4);
  unsigned h[256 * 16 * 4]; unsigned x = 1;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x007f007f) | 0x3f003f00 | (x & 0x80008000); }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<4, 4, 1, 0>("4x4 acc, 1 wave/SIMD", in, out, 256);
  run<4, 4, 1, 1>("4x4 acc, 1 wave/SIMD, pinned", in, out, 256);
  run<4, 3, 1, 0>("4x3 acc, 1 wave/SIMD", in, out, 256);
  run<2, 2, 1, 0>("2x2 acc, 1 wave/SIMD", in, out, 256);
  run<2, 2, 2, 0>("2x2 acc, 2 waves/SIMD", in, out, 512);
  run<2, 4, 2, 0>("2x4 acc, 2 waves/SIMD", in, out, 512);
  run<2, 2, 4, 0>("2x2 acc, 4 waves/SIMD", in, out, 1024);
  run<1, 1, 1, 0>("1x1 acc (dependent chain)", in, out, 256);
  hipMemset(in, 0, 256 * 16 * 16);
  run<4, 4, 1, 0>("4x4 acc, 1 wave/SIMD, ZERO data", in, out, 256);
  run<2, 4, 2, 0>("2x4 acc, 2 waves/SIMD, ZERO data", in, out, 512);
  return 0;
}
