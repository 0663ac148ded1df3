// Shared device helpers for the LAFS gfx950 kernels (wave64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                            // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;        // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

#define LAFS_WAVE 64

// ---- status codes of the C ABI (include/lafs_hip.h) ----
#define LAFS_OK 0
#define LAFS_EINVAL (-1)
#define LAFS_ESHAPE (-2)

// (internal: shared by the translation units of the library, not part of the C ABI)
extern "C" __attribute__((visibility("hidden"))) void lafs_set_error(const char* fmt, ...);
#define LAFS_CHECK_ARG(cond, msg)                                                     \
  do {                                                                                \
    if (!(cond)) {                                                                    \
      lafs_set_error("%s:%d: %s (%s)", __FILE__, __LINE__, msg, #cond);               \
      return LAFS_ESHAPE;                                                             \
    }                                                                                 \
  } while (0)
// hipGetLastError is sticky per thread: drop whatever an earlier, unrelated runtime call left behind
#define LAFS_CLEAR_ERROR() (void)hipGetLastError()
#define LAFS_LAUNCH_CHECK()                                                           \
  do {                                                                                \
    hipError_t e_ = hipGetLastError();                                                \
    if (e_ != hipSuccess) {                                                           \
      lafs_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return (int)e_;                                                                 \
    }                                                                                 \
  } while (0)

// ---- bf16 <-> fp32 (round to nearest even) ----
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16 through the native __bf16 type: on gfx950 the compiler lowers this to v_cvt_pk_bf16_f32 (round to nearest
// even, two values per instruction) -- the hand-rolled integer rounding cost ~5 VALU per value and made the attention
// kernels and GEMM epilogues VALU-issue bound.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- erf GELU (nn.GELU() default).  erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, one v_exp + one v_rcp):
// the library erff costs ~5x more VALU work and made the GELU epilogue the longest phase of the fc1 GEMM. ----
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);                       // v_rcp_f32 (1 ulp)
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float y = 1.0f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);  // v_exp_f32
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
// d/dx gelu(x) = Phi(x) + x phi(x).  The A&S 7.1.26 form of erf(x / sqrt 2) already evaluates exp(-x^2 / 2) -- the same
// exponential the density needs -- so one v_exp and one v_rcp serve both terms (the GELU' epilogues are VALU-bound).
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);               // exp(-x^2 / 2)
  const float erf_abs = 1.0f - poly * e;
  const float cdf = 0.5f + 0.5f * copysignf(erf_abs, x);
  return cdf + x * (0.39894228040143268f * e);
}

// gelu(x) and gelu'(x) together (one v_exp, one v_rcp): the fc1 epilogue that saves gelu'(u) for the backward instead of u
__device__ __forceinline__ void gelu_both_f(float x, float& g, float& dg) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);               // exp(-x^2 / 2)
  const float cdf = 0.5f + 0.5f * copysignf(1.0f - poly * e, x);
  g = x * cdf;
  dg = cdf + x * (0.39894228040143268f * e);
}

// ---- element dropout (nn.Dropout of Part-fViT, face_pre_pro/ViT_face.py:131-133,150-153,614): counter-based mask, so
// the backward kernels regenerate exactly the forward's mask from (seed, row, col) instead of storing it.
// keep <=> mix32(row * n_cols + col, seed) >= thresh, thresh = p * 2^32; kept values are scaled by 1/(1-p).
// A launch over rows [row0, row0 + R) of a larger batch numbers its elements from idx0 = row0 * n_cols, and a captured hipGraph
// draws a new mask on every replay: the seed of a step is seed + 7919 * step with `step` read from DEVICE memory (hyper[HP_STEP],
// the counter the DropPath scales use), so the same launch arguments give new masks step after step.  Passing step = nullptr and
// seed' = seed + 7919 * step on the host gives the identical mask (lafs_debug_dropout_mask: tests).
struct DropCfg { unsigned thresh; unsigned seed; float scale; unsigned idx0; const float* step; };        // thresh == 0: disabled
__device__ __forceinline__ unsigned drop_mix32(unsigned idx, unsigned seed) {
  unsigned x = idx * 0x9E3779B1u ^ seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// once per kernel: raw seed (+ 7919 * device step) -> scrambled seed
__device__ __forceinline__ DropCfg drop_resolve(DropCfg d) {
  if (d.thresh) {
    unsigned raw = d.seed;
    if (d.step != nullptr) raw += 7919u * (unsigned)(*d.step);
    d.seed = raw * 0x632BE5ABu + 0x7F4A7C15u;
  }
  return d;
}
__device__ __forceinline__ float drop_mult(const DropCfg& d, unsigned idx) {
  return drop_mix32(idx + d.idx0, d.seed) >= d.thresh ? d.scale : 0.f;
}
inline DropCfg make_drop(float p, unsigned seed, const float* step = nullptr, unsigned idx0 = 0) {
  DropCfg d;
  d.idx0 = 0; d.step = nullptr;
  if (!(p > 0.f)) { d.thresh = 0; d.seed = 0; d.scale = 1.f; return d; }
  const double t = (double)p * 4294967296.0;
  d.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
  if (d.thresh == 0) d.thresh = 1;
  d.seed = seed;                                  // raw: drop_resolve scrambles it on the device
  d.scale = 1.0f / (1.0f - p);
  d.idx0 = idx0; d.step = step;
  return d;
}

// ---- activations of the landmark CNN (MobileNetV3: relu, x*relu6(x+3)/6, relu6(x+3)/6) ----
__device__ __forceinline__ float act_f(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
  if (act == 3) return fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
  return v;
}

// ---- LayerNorm row arithmetic shared by ln_fwd2_kernel (layernorm.hip) and the fused MLP's LayerNorm prologue (mlp_fused.hip), which
// must agree BIT FOR BIT: the multiply-adds are explicit and in a fixed order -- left to -ffp-contract, `a*a + b*b` may become
// fma(a, a, b*b) in one kernel and fma(b, b, a*a) in another (observed: 2 of 1 966 080 bf16 outputs differed between two
// instantiations of the same source).
__device__ __forceinline__ float ln_sum4(float s, const float4& v) { return s + (((v.x + v.y) + v.z) + v.w); }
__device__ __forceinline__ float ln_sq4(float q, const float4& v, float mean) {
  const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
  return q + __builtin_fmaf(d, d, __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a)));
}
__device__ __forceinline__ float ln_out1(float x, float mean, float rstd, float g, float b) { return __builtin_fmaf((x - mean) * rstd, g, b); }

// ---- wave reductions (64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- LDS transpose read: each 16-lane group reads a 4(row) x 16(col) bf16 block; lane p of the group
// supplies the address of 4 contiguous elements (row p>>2, column chunk p&3) and receives column p. ----
__device__ __forceinline__ s16x4_t lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds_addr));
}

__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- fp16 storage (IEEE half, 10-bit mantissa): the trainable landmark CNN keeps its activations, operand images and activation
// gradients in fp16 like the reference's autocast run (train_largescale.py:803-804) -- gfx950 runs f16 MFMA at the bf16 rate, and
// batch-statistics BatchNorm on this network is 8x less sensitive to fp16 roundings than to bf16 ones (tests/test_gpu_finetune.py F18).
// The containers stay bf16_t / bf16x8_t (raw 16-bit lanes); only the interpretation differs.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ float h2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ bf16_t f2h(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
  f16x2_t v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float h_lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
__device__ __forceinline__ float h_hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
__device__ __forceinline__ f32x4_t mfma16_f16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// ---- LDS-DMA (global_load_lds_dwordx4) issued from inline asm.  Each lane supplies its own 16-byte source; the destination
// is lds_base (wave-uniform byte address inside the workgroup's LDS, carried in M0) + 16 * lane.  Why asm and not
// __builtin_amdgcn_global_load_lds: hipcc's wait-count pass knows the builtin writes LDS and drains the whole DMA queue
// (s_waitcnt vmcnt(0)) in front of every LDS read it cannot disambiguate -- ds_read_b64_tr_b16 has no memory operand, so the
// round-1 weight-gradient kernel emptied its 3-stage ring at every stage.  An asm statement is invisible to that pass: the
// kernels count these loads themselves (s_waitcnt vmcnt(N) + s_barrier before the first read of a stage).
// M0 is compiler-reserved and not preserved around asm, so it is saved and restored inside the statement.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
// Same, without saving M0: for kernels whose compiled code never uses M0 itself (no LDS-DMA / GWS / movrel builtins) -- two
// scalar instructions less per 1 KiB piece, which matters when a single wave per SIMD has to issue everything.
__device__ __forceinline__ void lds_dma16_m0(const void* gsrc, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_base) : "memory");
}
// Same with a scalar base address and a 32-bit per-lane byte offset (global_load_lds ... v_off, s[base]): one VGPR of address per
// lane instead of a 64-bit pair, no 64-bit add per piece.
__device__ __forceinline__ void lds_dma16_m0_s(const void* sbase, unsigned voff_bytes, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff_bytes), "s"(sbase), "s"(lds_base) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
