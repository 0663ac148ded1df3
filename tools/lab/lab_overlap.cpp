// Does VALU work issued between the MFMAs of ONE wave hide in their shadow on gfx950?  (round 5: every "work placed beside other work"
// experiment of rounds 2-5 measured level; this asks the hardware directly.)  A wave issues NM independent v_mfma_f32_16x16x32_bf16 per
// loop iteration with KV independent v_fma_f32 (or KX v_exp_f32) between consecutive MFMAs, pinned by sched_barrier; one or two waves per
// SIMD.  Reported: ns per MFMA per wave, and what the same VALU work costs alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int KV, int KX, int W, bool MFMA>
__global__ __launch_bounds__(256, W) void probe(const uint4* in, float* out, int iters) {
  constexpr int NM = 16;
  f32x4_t acc[NM];
  bf16x8_t a = __builtin_bit_cast(bf16x8_t, in[threadIdx.x]), b = __builtin_bit_cast(bf16x8_t, in[threadIdx.x + 256]);
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
  for (int i = 0; i < NM; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (MFMA) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < KV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
#pragma unroll
      for (int k = 0; k < KX; ++k) v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KV, int KX, int W, bool MFMA>
void run(const uint4* in, float* out) {
  const int iters = 4000, grid = 256 * W;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<KV, KX, W, MFMA>), dim3(grid), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<KV, KX, W, MFMA>), dim3(grid), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 16;
  printf("%s  fma %2d exp %d per MFMA slot, %d wave(s)/SIMD: %7.2f ns per slot per wave%s\n", MFMA ? "MFMA + VALU" : "VALU only  ", KV, KX, W, ms * 1e6 / n,
         MFMA ? "" : "");
}

int main() {
  uint4* in; float* out;
  hipMalloc(&in, 512 * 16); hipMalloc(&out, 2048 * 256 * 4);
  unsigned h[512 * 4]; unsigned x = 1;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x007f007f) | 0x3f003f00; }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0, 0, 1, true>(in, out);  run<1, 0, 1, true>(in, out);  run<2, 0, 1, true>(in, out);  run<3, 0, 1, true>(in, out);
  run<4, 0, 1, true>(in, out);  run<6, 0, 1, true>(in, out);  run<8, 0, 1, true>(in, out);  run<0, 1, 1, true>(in, out);  run<2, 1, 1, true>(in, out);
  run<2, 0, 1, false>(in, out); run<4, 0, 1, false>(in, out); run<8, 0, 1, false>(in, out); run<0, 1, 1, false>(in, out);
  run<0, 0, 2, true>(in, out);  run<2, 0, 2, true>(in, out);  run<4, 0, 2, true>(in, out);  run<8, 0, 2, true>(in, out);
  run<4, 0, 2, false>(in, out); run<8, 0, 2, false>(in, out);
  return 0;
}
