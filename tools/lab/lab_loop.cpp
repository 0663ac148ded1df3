// Inner-loop probe for the K-resident GEMMs: cycles per ring stage (32 weight rows x 384 k against 32 resident token rows) of
//   V0  the product pattern: v_mfma_f32_16x16x32_bf16, one ds_read_b128 weight fragment per 2 MFMAs, reads FD ahead (48 MFMAs / stage)
//   V1  v_mfma_f32_32x32x16_bf16: one ds_read_b128 weight fragment (32 rows x 16 k... 8 bf16 per lane) per MFMA (24 MFMAs / stage)
//   V2  V0 without the LDS reads (fragments stay in registers): the MFMA issue rate alone
//   V3  V1 without the LDS reads
// with 1 or 2 waves per SIMD, everything LDS-resident (no DMA, no epilogue): what the MFMA phase costs by itself.
//   make -C tools/lab lab_loop && gpurun -- tools/lab/lab_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int STAGE = 32 * 768;   // 24 KiB: 32 weight rows x 384 k bf16

template <int V, int FD>
__global__ __launch_bounds__(256, 2) void probe(const uint4* in, float* out, int iters, unsigned long long* ticks) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * STAGE];
  for (int i = threadIdx.x; i < 3 * STAGE / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = in[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int t = lane & 15, q = lane >> 4;
  bf16x8_t areg[2][12];
  for (int b = 0; b < 2; ++b) for (int k = 0; k < 12; ++k) areg[b][k] = __builtin_bit_cast(bf16x8_t, in[(threadIdx.x + 64 * (b * 12 + k)) & 4095]);
  float sum = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (V == 0 || V == 2) {
    int foff[4];
    for (int i = 0; i < 4; ++i) { foff[i] = t * 768 + (((4 * i + q) ^ t) << 4); asm volatile("" : "+v"(foff[i])); }
    for (int it = 0; it < iters; ++it) {
      f32x4_t a[2][2];
      for (int gi = 0; gi < 2; ++gi) for (int b = 0; b < 2; ++b) a[gi][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const unsigned char* st = smem + (it % 3) * STAGE;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 12; ++kk)
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          bf16x8_t w = areg[gi][(kk + 1) % 12];
          if constexpr (V == 0) w = *reinterpret_cast<const bf16x8_t*>(st + foff[kk & 3] + (kk >> 2) * 256 + gi * (16 * 768));
          a[gi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, areg[0][kk], a[gi][0], 0, 0, 0);
          a[gi][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, areg[1][kk], a[gi][1], 0, 0, 0);
        }
      if constexpr (V == 0) {
        __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);
#pragma unroll
        for (int i = 0; i < 24 - FD; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * FD, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      for (int gi = 0; gi < 2; ++gi) for (int b = 0; b < 2; ++b) sum += a[gi][b][0] + a[gi][b][3];
    }
  } else {
    // 32x32x16: A operand (weights, from LDS): lane (row = lane & 31, k half = lane >> 5) reads 16 bytes = 8 k of its row per 16-k step;
    // B operand (tokens, resident): areg reinterpreted as 24 k16-steps of 32 tokens
    const int r = lane & 31, hk = lane >> 5;
    const bf16x8_t* ar = &areg[0][0];
    for (int it = 0; it < iters; ++it) {
      f32x16_t acc = {0.f};
      const unsigned char* st = smem + (it % 3) * STAGE;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 24; ++ks) {
        bf16x8_t w = ar[(ks + 1) % 24];
        if constexpr (V == 1) w = *reinterpret_cast<const bf16x8_t*>(st + r * 768 + ((((2 * ks + hk)) ^ (r & 15)) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, ar[ks], acc, 0, 0, 0);
      }
      if constexpr (V == 1) {
        __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);
#pragma unroll
        for (int i = 0; i < 24 - FD; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, FD, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      sum += acc[0] + acc[15];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int V, int FD>
void run(const char* name, const uint4* in, float* out, unsigned long long* ticks, int grid) {
  const int iters = 400;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<V, FD>), dim3(grid), dim3(256), 0, 0, in, out, iters, ticks);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<V, FD>), dim3(grid), dim3(256), 0, 0, in, out, iters, ticks);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1024]; hipMemcpy(h, ticks, grid * 8, hipMemcpyDeviceToHost);
  double tk = 0; for (int i = 0; i < grid; ++i) tk += (double)h[i];
  tk /= grid;
  const double flops = (double)iters * 2.0 * 32 * 32 * 384 * 4 * grid;      // per wave-stage: 32 tokens x 32 weight rows x 384 k
  printf("%-52s grid %4d (%d waves/SIMD): %6.0f memtime ticks, %6.0f cycles at 2.4 GHz per wave-stage   %7.1f us   %7.1f TF/s\n", name, grid,
         grid / 256, tk / iters, ms * 1e-3 * 2.4e9 / iters, ms * 1e3, flops / ms / 1e9);
}

int main() {
  uint4* in; float* out; unsigned long long* ticks;
  hipMalloc(&in, 4096 * 16); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&ticks, 1024 * 8);
  unsigned h[4096 * 4]; unsigned x = 1;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x007f007f) | 0x3f003f00 | (x & 0x80008000); }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int grid : {256, 512}) {
    run<0, 8>("V0 16x16x32, 1 LDS fragment per 2 MFMAs, FD 8", in, out, ticks, grid);
    run<0, 12>("V0 ... FD 12", in, out, ticks, grid);
    run<2, 8>("V2 16x16x32, no LDS reads", in, out, ticks, grid);
    run<1, 8>("V1 32x32x16, 1 LDS fragment per MFMA, FD 8", in, out, ticks, grid);
    run<1, 4>("V1 ... FD 4", in, out, ticks, grid);
    run<3, 8>("V3 32x32x16, no LDS reads", in, out, ticks, grid);
  }
  return 0;
}
