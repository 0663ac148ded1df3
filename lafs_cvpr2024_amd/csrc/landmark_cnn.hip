// Inference kernels of the frozen landmark CNN (MobileNetV3-large trunk, reference face_pre_pro/mobilenet.py:224-313, driven
// by face_landmark_4simmin_glo_loc.forward, face_pre_pro/ViT_face.py:1338-1344).  Activations are NHWC bf16 with the channel
// count padded to a multiple of 32 so that every 1x1 convolution is a plain lafs_gemm_nt call (epilogue BF16_ACT: folded
// BatchNorm bias + residual + activation); what is left -- the 3x3 stem, the depthwise convolutions, the squeeze-excite
// pooling and rescale -- are the bandwidth-bound kernels below.  BatchNorm is folded into weights/biases on the host
// (eval mode: the CNN is frozen on the LAFS path, lafs_train.py:262-269).
#include "common.hpp"
#include "lafs_hip.h"

namespace {

// x f32 NCHW [N,3,S,S] -> y bf16 NHWC [N,S/2,S/2,ldy]: 16 channels of act(conv3x3 stride 2 pad 1 + b), channels 16.. zero
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                   int N, int S, int act, bf16_t* __restrict__ y, int ldy) {
  __shared__ float sw[27 * 16 + 16];
  for (int i = threadIdx.x; i < 27 * 16 + 16; i += 256) sw[i] = i < 27 * 16 ? w[i] : b[i - 27 * 16];
  __syncthreads();
  const int So = S >> 1;
  const long total = (long)N * So * So;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ox = (int)(idx % So), oy = (int)((idx / So) % So), n = (int)(idx / ((long)So * So));
  float acc[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) acc[o] = sw[27 * 16 + o];
  const float* xn = x + (size_t)n * 3 * S * S;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= S) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= S) continue;
        const float v = xn[((size_t)c * S + iy) * S + ix];
        const float* wk = sw + (c * 9 + ky * 3 + kx) * 16;
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = fmaf(v, wk[o], acc[o]);
      }
    }
  bf16_t* yo = y + (size_t)idx * ldy;
  uint4 lo = make_uint4(pack_bf2(act_f(acc[0], act), act_f(acc[1], act)), pack_bf2(act_f(acc[2], act), act_f(acc[3], act)),
                        pack_bf2(act_f(acc[4], act), act_f(acc[5], act)), pack_bf2(act_f(acc[6], act), act_f(acc[7], act)));
  uint4 hi = make_uint4(pack_bf2(act_f(acc[8], act), act_f(acc[9], act)), pack_bf2(act_f(acc[10], act), act_f(acc[11], act)),
                        pack_bf2(act_f(acc[12], act), act_f(acc[13], act)), pack_bf2(act_f(acc[14], act), act_f(acc[15], act)));
  *reinterpret_cast<uint4*>(yo) = lo;
  *reinterpret_cast<uint4*>(yo + 8) = hi;
  for (int c = 16; c < ldy; c += 8) *reinterpret_cast<uint4*>(yo + c) = make_uint4(0, 0, 0, 0);
}

// depthwise k x k convolution, NHWC bf16, 8 channels per thread: y = act(conv(x) + b)   (act < 0: none)
template <int K>
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                     int N, int H, int W, int C, int stride, int act, bf16_t* __restrict__ y) {
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride, C8 = C >> 3;
  const long total = (long)N * Ho * Wo * C8;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C8) * 8;
  const long pix = idx / C8;
  const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), n = (int)(pix / ((long)Wo * Ho));
  constexpr int P = (K - 1) / 2;
  float acc[8];
  {
    const float4 b0 = *reinterpret_cast<const float4*>(b + c), b1 = *reinterpret_cast<const float4*>(b + c + 4);
    acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
  }
  const bf16_t* xn = x + (size_t)n * H * W * C + c;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy * stride - P + ky;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int ix = ox * stride - P + kx;
      if (ix < 0 || ix >= W) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(xn + ((size_t)iy * W + ix) * C);
      const float* wk = w + (size_t)(ky * K + kx) * C + c;
      const float4 w0 = *reinterpret_cast<const float4*>(wk), w1 = *reinterpret_cast<const float4*>(wk + 4);
      acc[0] = fmaf(bf_lo(v.x), w0.x, acc[0]); acc[1] = fmaf(bf_hi(v.x), w0.y, acc[1]);
      acc[2] = fmaf(bf_lo(v.y), w0.z, acc[2]); acc[3] = fmaf(bf_hi(v.y), w0.w, acc[3]);
      acc[4] = fmaf(bf_lo(v.z), w1.x, acc[4]); acc[5] = fmaf(bf_hi(v.z), w1.y, acc[5]);
      acc[6] = fmaf(bf_lo(v.w), w1.z, acc[6]); acc[7] = fmaf(bf_hi(v.w), w1.w, acc[7]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = act_f(acc[e], act);
  *reinterpret_cast<uint4*>(y + (size_t)pix * C + c) =
      make_uint4(pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3]), pack_bf2(acc[4], acc[5]), pack_bf2(acc[6], acc[7]));
}

// out(bf16)[n, c] = mean_p x[n, p, c]     (squeeze of squeeze-excite; final 4x4 average pool)
__global__ __launch_bounds__(256) void pool_kernel(const bf16_t* __restrict__ x, int N, int HW, int C, bf16_t* __restrict__ out, int ldo) {
  const int C8 = C >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * C8) return;
  const int c = (int)(idx % C8) * 8, n = (int)(idx / C8);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16_t* xn = x + (size_t)n * HW * C + c;
  for (int p = 0; p < HW; ++p) {
    const uint4 v = *reinterpret_cast<const uint4*>(xn + (size_t)p * C);
    acc[0] += bf_lo(v.x); acc[1] += bf_hi(v.x); acc[2] += bf_lo(v.y); acc[3] += bf_hi(v.y);
    acc[4] += bf_lo(v.z); acc[5] += bf_hi(v.z); acc[6] += bf_lo(v.w); acc[7] += bf_hi(v.w);
  }
  const float inv = 1.0f / (float)HW;
  *reinterpret_cast<uint4*>(out + (size_t)n * ldo + c) =
      make_uint4(pack_bf2(acc[0] * inv, acc[1] * inv), pack_bf2(acc[2] * inv, acc[3] * inv),
                 pack_bf2(acc[4] * inv, acc[5] * inv), pack_bf2(acc[6] * inv, acc[7] * inv));
}

// The same mean with one WORKGROUP per image: lanes run along the channel axis (16-byte pieces of one pixel: coalesced), the pixels are
// dealt to the 256 / C8 thread rows and the partial sums meet in the LDS.  The one-thread-per-(image, 8 channels) kernel above has
// N C / 8 threads in all -- 7 680 for 640 images x 96 channels = 30 workgroups on 256 CUs, 65-90 us for 24 MB (0.3 TB/s).
__global__ __launch_bounds__(256) void pool_wg_kernel(const bf16_t* __restrict__ x, int HW, int C, bf16_t* __restrict__ out, int ldo) {
  __shared__ float part[256 * 8];
  const int C8 = C >> 3, n = blockIdx.x;
  const int parts = 256 / C8, c8 = threadIdx.x % C8, pr = threadIdx.x / C8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pr < parts) {
    const bf16_t* xn = x + (size_t)n * HW * C + c8 * 8;
    for (int p = pr; p < HW; p += parts) {
      const uint4 v = *reinterpret_cast<const uint4*>(xn + (size_t)p * C);
      acc[0] += bf_lo(v.x); acc[1] += bf_hi(v.x); acc[2] += bf_lo(v.y); acc[3] += bf_hi(v.y);
      acc[4] += bf_lo(v.z); acc[5] += bf_hi(v.z); acc[6] += bf_lo(v.w); acc[7] += bf_hi(v.w);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[(pr * C8 + c8) * 8 + e] = acc[e];
  }
  __syncthreads();
  if (threadIdx.x < C8) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < parts; ++q)                      // fixed order: the result does not depend on the schedule
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += part[(q * C8 + c8) * 8 + e];
    const float inv = 1.0f / (float)HW;
    *reinterpret_cast<uint4*>(out + (size_t)n * ldo + c8 * 8) =
        make_uint4(pack_bf2(s[0] * inv, s[1] * inv), pack_bf2(s[2] * inv, s[3] * inv), pack_bf2(s[4] * inv, s[5] * inv), pack_bf2(s[6] * inv, s[7] * inv));
  }
}

// x[n, p, c] = act(x[n, p, c] * s[n, c])   in place (excite + the block's non-linearity)
__global__ __launch_bounds__(256) void scale_act_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ s, int lds_, int N, int HW, int C,
                                                        int act) {
  const int C8 = C >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * HW * C8) return;
  const int c = (int)(idx % C8) * 8;
  const long pix = idx / C8;
  const int n = (int)(pix / HW);
  bf16_t* xp = x + (size_t)pix * C + c;
  const uint4 v = *reinterpret_cast<const uint4*>(xp), g = *reinterpret_cast<const uint4*>(s + (size_t)n * lds_ + c);
  *reinterpret_cast<uint4*>(xp) =
      make_uint4(pack_bf2(act_f(bf_lo(v.x) * bf_lo(g.x), act), act_f(bf_hi(v.x) * bf_hi(g.x), act)),
                 pack_bf2(act_f(bf_lo(v.y) * bf_lo(g.y), act), act_f(bf_hi(v.y) * bf_hi(g.y), act)),
                 pack_bf2(act_f(bf_lo(v.z) * bf_lo(g.z), act), act_f(bf_hi(v.z) * bf_hi(g.z), act)),
                 pack_bf2(act_f(bf_lo(v.w) * bf_lo(g.w), act), act_f(bf_hi(v.w) * bf_hi(g.w), act)));
}

inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

extern "C" int lafs_cnn_stem(const float* x, const float* w, const float* b, int N, int S, int act, void* y, int ldy, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && w && b && y && N > 0 && S > 0 && S % 2 == 0 && ldy >= 16 && ldy % 8 == 0, "bad operand");
  const long total = (long)N * (S / 2) * (S / 2);
  hipLaunchKernelGGL(stem_kernel, dim3(blocks_for(total)), dim3(256), 0, stream, x, w, b, N, S, act, (bf16_t*)y, ldy);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_dwconv(const void* x, const float* w, const float* b, int N, int H, int W, int C, int k, int stride, int act,
                               void* y, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && w && b && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "C must be a multiple of 8");
  LAFS_CHECK_ARG((k == 3 || k == 5) && (stride == 1 || stride == 2), "k in {3,5}, stride in {1,2}");
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const long total = (long)N * Ho * Wo * (C / 8);
  if (k == 3)
    hipLaunchKernelGGL(dwconv_kernel<3>, dim3(blocks_for(total)), dim3(256), 0, stream, (const bf16_t*)x, w, b, N, H, W, C, stride, act,
                       (bf16_t*)y);
  else
    hipLaunchKernelGGL(dwconv_kernel<5>, dim3(blocks_for(total)), dim3(256), 0, stream, (const bf16_t*)x, w, b, N, H, W, C, stride, act,
                       (bf16_t*)y);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_pool(const void* x, int N, int HW, int C, void* out, int ldo, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && out && N > 0 && HW > 0 && C > 0 && C % 8 == 0 && ldo >= C && ldo % 8 == 0, "bad operand");
  if (C / 8 <= 256 && HW >= 16)                            // (a 4 x 4 map: nothing to split)
    hipLaunchKernelGGL(pool_wg_kernel, dim3(N), dim3(256), 0, stream, (const bf16_t*)x, HW, C, (bf16_t*)out, ldo);
  else
    hipLaunchKernelGGL(pool_kernel, dim3(blocks_for((long)N * (C / 8))), dim3(256), 0, stream, (const bf16_t*)x, N, HW, C, (bf16_t*)out, ldo);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_scale_act(void* x, const void* s, int lds_, int N, int HW, int C, int act, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && s && N > 0 && HW > 0 && C > 0 && C % 8 == 0 && lds_ >= C && lds_ % 8 == 0, "bad operand");
  hipLaunchKernelGGL(scale_act_kernel, dim3(blocks_for((long)N * HW * (C / 8))), dim3(256), 0, stream, (bf16_t*)x, (const bf16_t*)s, lds_,
                     N, HW, C, act);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Trainable landmark branch of the fine-tune step (Part-fViT with_land=True, train_largescale.py:432): the MobileNetV3 trunk
// stays on torch autograd there, but its depthwise convolutions are the kernels below -- on this ROCm image MIOpen has no
// tuned depthwise solver for these shapes (naive / Winograd / im2col fall-backs: ~14 ms of a 50 ms step at batch 128).
// fp32 NCHW, bias-free, pad (k-1)/2, exactly nn.Conv2d(C, C, k, stride, (k-1)//2, groups=C) and its two gradients.
// ------------------------------------------------------------------------------------------------------------------------
namespace {

// Index arithmetic: planes are at most 56x56 and the divisors are run-time values, so quotients are taken with a float
// reciprocal (exact for operands < 2^22 with the +0.5 bias) instead of the ~40-instruction integer division sequence.
__device__ __forceinline__ int fdiv(int a, float inv_b) { return (int)(((float)a + 0.5f) * inv_b); }

template <int K, int S>
__global__ __launch_bounds__(256) void dw_nchw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, long total, int C,
                                                          int H, int W, float* __restrict__ y) {
  const int Ho = (H + S - 1) / S, Wo = (W + S - 1) / S, plane = Ho * Wo;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;            // flat (n, c, oy, ox): small planes still fill the workgroup
  if (idx >= total) return;
  const int nc = (int)(idx / plane);                                 // (the flat index exceeds fdiv's exact range)
  const int o = (int)(idx - (long)nc * plane);
  const int c = nc - fdiv(nc, 1.0f / (float)C) * C;
  const int oy = fdiv(o, 1.0f / (float)Wo), ox = o - oy * Wo;
  constexpr int P = (K - 1) / 2;
  const float* wk = w + (size_t)c * K * K;
  const float* xp = x + (size_t)nc * H * W;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy * S - P + ky;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int ix = ox * S - P + kx;
      if (ix < 0 || ix >= W) continue;
      acc = fmaf(xp[iy * W + ix], wk[ky * K + kx], acc);
    }
  }
  y[idx] = acc;
}

// dx[iy,ix] = sum_{ky,kx} dy[(iy+P-ky)/s, (ix+P-kx)/s] * w[ky,kx]   over the taps where the division is exact and in range
template <int K, int S>
__global__ __launch_bounds__(256) void dw_nchw_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w, long total,
                                                               int C, int H, int W, float* __restrict__ dx) {
  const int Ho = (H + S - 1) / S, Wo = (W + S - 1) / S, plane = H * W;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;            // flat (n, c, iy, ix)
  if (idx >= total) return;
  const int nc = (int)(idx / plane);
  const int i = (int)(idx - (long)nc * plane);
  const int c = nc - fdiv(nc, 1.0f / (float)C) * C;
  const int iy = fdiv(i, 1.0f / (float)W), ix = i - iy * W;
  constexpr int P = (K - 1) / 2;
  const float* wk = w + (size_t)c * K * K;
  const float* dp = dy + (size_t)nc * Ho * Wo;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int ty = iy + P - ky;
    if (ty < 0 || (S == 2 && (ty & 1))) continue;
    const int oy = ty / S;
    if (oy >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int tx = ix + P - kx;
      if (tx < 0 || (S == 2 && (tx & 1))) continue;
      const int ox = tx / S;
      if (ox >= Wo) continue;
      acc = fmaf(dp[oy * Wo + ox], wk[ky * K + kx], acc);
    }
  }
  dx[idx] = acc;
}

// dw[c, ky, kx] += sum_{n in chunk, oy, ox} dy[n,c,oy,ox] * x[n,c,oy*s-P+ky, ox*s-P+kx];  grid (C, n_chunks)
template <int K, int S>
__global__ __launch_bounds__(256) void dw_nchw_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy, int N, int C,
                                                                 int H, int W, int n_per_block, float* __restrict__ dw) {
  __shared__ float red[4][K * K];
  const int Ho = (H + S - 1) / S, Wo = (W + S - 1) / S;
  const int c = blockIdx.x, n0 = blockIdx.y * n_per_block, n1 = min(N, n0 + n_per_block);
  constexpr int P = (K - 1) / 2;
  float acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = 0.f;
  const int per_img = Ho * Wo, total = (n1 - n0) * per_img;
  const float inv_img = 1.0f / (float)per_img, inv_wo = 1.0f / (float)Wo;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int dn = fdiv(e, inv_img), o = e - dn * per_img, n = n0 + dn;
    const int oy = fdiv(o, inv_wo), ox = o - oy * Wo;
    const float g = dy[((size_t)n * C + c) * per_img + o];
    const float* xp = x + ((size_t)n * C + c) * H * W;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * S - P + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int ix = ox * S - P + kx;
        if (ix < 0 || ix >= W) continue;
        acc[ky * K + kx] = fmaf(g, xp[iy * W + ix], acc[ky * K + kx]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    const float s = wave_sum(acc[t]);
    if (lane == 0) red[wave][t] = s;
  }
  __syncthreads();
  if (threadIdx.x < K * K) atomicAdd(dw + (size_t)c * K * K + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

#define DW_DISPATCH(KERNEL, ...)                                                              \
  do {                                                                                        \
    if (k == 3 && stride == 1) hipLaunchKernelGGL((KERNEL<3, 1>), __VA_ARGS__);               \
    else if (k == 3) hipLaunchKernelGGL((KERNEL<3, 2>), __VA_ARGS__);                         \
    else if (stride == 1) hipLaunchKernelGGL((KERNEL<5, 1>), __VA_ARGS__);                    \
    else hipLaunchKernelGGL((KERNEL<5, 2>), __VA_ARGS__);                                     \
  } while (0)

int dw_check(const void* a, const void* b, const void* c, int N, int C, int H, int W, int k, int stride) {
  LAFS_CHECK_ARG(a && b && c && N > 0 && C > 0 && H > 0 && W > 0, "bad operand");
  LAFS_CHECK_ARG((k == 3 || k == 5) && (stride == 1 || stride == 2), "k in {3,5}, stride in {1,2}");
  LAFS_CHECK_ARG(C <= 65535 && N <= 65535, "N and C must fit a grid dimension");
  LAFS_CHECK_ARG((long)N * C * H * W < (1L << 31) && (long)N * C < (1L << 22) && H * W < (1 << 22), "tensor too large for the index arithmetic");
  return LAFS_OK;
}

}  // namespace

extern "C" int lafs_dwconv_nchw_fwd(const float* x, const float* w, int N, int C, int H, int W, int k, int stride, float* y,
                                    hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  if (int rc = dw_check(x, w, y, N, C, H, W, k, stride)) return rc;
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const long total = (long)N * C * Ho * Wo;
  const dim3 grid(blocks_for(total));
  DW_DISPATCH(dw_nchw_fwd_kernel, grid, dim3(256), 0, stream, x, w, total, C, H, W, y);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_dwconv_nchw_bwd_data(const float* dy, const float* w, int N, int C, int H, int W, int k, int stride, float* dx,
                                         hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  if (int rc = dw_check(dy, w, dx, N, C, H, W, k, stride)) return rc;
  const long total = (long)N * C * H * W;
  const dim3 grid(blocks_for(total));
  DW_DISPATCH(dw_nchw_bwd_data_kernel, grid, dim3(256), 0, stream, dy, w, total, C, H, W, dx);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_dwconv_nchw_bwd_weight(const float* x, const float* dy, int N, int C, int H, int W, int k, int stride, float* dw,
                                           hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  if (int rc = dw_check(x, dy, dw, N, C, H, W, k, stride)) return rc;
  const int Ho = (H + stride - 1) / stride;
  int n_per_block = 4096 / (Ho * Ho);                              // ~4k (image, pixel) pairs per workgroup: thousands of workgroups
  if (n_per_block < 1) n_per_block = 1;
  if (n_per_block > N) n_per_block = N;
  const dim3 grid(C, (N + n_per_block - 1) / n_per_block);
  DW_DISPATCH(dw_nchw_bwd_weight_kernel, grid, dim3(256), 0, stream, x, dy, N, C, H, W, n_per_block, dw);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// BatchNorm2d (+ ReLU / h-swish) of the trainable landmark branch, fp32 NCHW, training and eval mode, forward and backward
// (face_pre_pro/mobilenet.py:104-111,177-190: Conv -> BatchNorm2d -> activation).  Replaces MIOpen BatchNorm + separate
// activation kernels (2 + 2 passes per pair and direction) by one statistics pass + one apply pass.
// ------------------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float act_grad_f(float z, int act) {       // d act(z) / dz
  if (act == 1) return z > 0.f ? 1.f : 0.f;
  if (act == 2) return z <= -3.f ? 0.f : (z >= 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
  return 1.f;
}

// sums[c] += sum x, sums[C + c] += sum x^2 over the (n-chunk, HW) slab of channel c.   grid (C, n_chunks)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int N, int C, int HW, int n_per_block,
                                                       float* __restrict__ sums) {
  __shared__ float red[2][4];
  const int c = blockIdx.x, n0 = blockIdx.y * n_per_block, n1 = min(N, n0 + n_per_block);
  float s = 0.f, q = 0.f;
  const float inv_hw = 1.0f / (float)HW;
  const int total = (n1 - n0) * HW;                                  // flat (image, pixel): small planes keep all lanes busy
  for (int e = threadIdx.x; e < total; e += 256) {
    const int dn = fdiv(e, inv_hw), i = e - dn * HW;
    const float v = x[((size_t)(n0 + dn) * C + c) * HW + i];
    s += v; q = fmaf(v, v, q);
  }
  s = wave_sum(s); q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(sums + c, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(sums + C + c, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// stat[c] = mean, stat[C + c] = rstd from the sums (training: also the running-statistics update) or from running stats (eval)
__global__ void bn_finalize_kernel(const float* __restrict__ sums, float count, float eps, float momentum, int C, int training,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ stat) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (training) {
    const float mean = sums[c] / count;
    const float var = fmaxf(sums[C + c] / count - mean * mean, 0.f);
    stat[c] = mean; stat[C + c] = rsqrtf(var + eps);
    if (running_mean != nullptr) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / fmaxf(count - 1.f, 1.f));
    }
  } else {
    stat[c] = running_mean[c]; stat[C + c] = rsqrtf(running_var[c] + eps);
  }
}

// y = act((x - mean) * rstd * gamma + beta)
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ stat,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, long total4, int C,
                                                         int HW4, int act, float* __restrict__ y) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;                  // float4 index (< 2^32: checked by the launcher); 32-bit division
  if (i >= total4) return;
  const int c = (int)((i / (unsigned)HW4) % (unsigned)C);
  const float sc = stat[C + c] * gamma[c], sh = beta[c] - stat[c] * sc;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  reinterpret_cast<float4*>(y)[i] = make_float4(act_f(fmaf(v.x, sc, sh), act), act_f(fmaf(v.y, sc, sh), act), act_f(fmaf(v.z, sc, sh), act),
                                                act_f(fmaf(v.w, sc, sh), act));
}
__global__ __launch_bounds__(256) void bn_act_fwd_scalar_kernel(const float* __restrict__ x, const float* __restrict__ stat,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, long total,
                                                                int C, int HW, int act, float* __restrict__ y) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const int c = (int)((i / (unsigned)HW) % (unsigned)C);
  const float sc = stat[C + c] * gamma[c], sh = beta[c] - stat[c] * sc;
  y[i] = act_f(fmaf(x[i], sc, sh), act);
}

// dsum[c] += sum dz, dsum[C + c] += sum dz * xhat   with z = xhat*gamma + beta, dz = dy * act'(z)
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const float* __restrict__ stat, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int N, int C, int HW, int n_per_block, int act,
                                                                float* __restrict__ dsum) {
  __shared__ float red[2][4];
  const int c = blockIdx.x, n0 = blockIdx.y * n_per_block, n1 = min(N, n0 + n_per_block);
  const float mean = stat[c], rstd = stat[C + c], g = gamma[c], b = beta[c];
  float s = 0.f, q = 0.f;
  const float inv_hw = 1.0f / (float)HW;
  const int total = (n1 - n0) * HW;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int dn = fdiv(e, inv_hw), i = e - dn * HW;
    const size_t at = ((size_t)(n0 + dn) * C + c) * HW + i;
    const float xh = (x[at] - mean) * rstd;
    const float dz = dy[at] * act_grad_f(fmaf(xh, g, b), act);
    s += dz; q = fmaf(dz, xh, q);
  }
  s = wave_sum(s); q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(dsum + c, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(dsum + C + c, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// training: dx = gamma*rstd*(dz - dsum/cnt - xhat*dsumx/cnt);  eval: dx = gamma*rstd*dz
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ stat, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ dsum, float inv_count,
                                                               long total, int C, int HW, int act, int training, float* __restrict__ dx) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const int c = (int)((i / (unsigned)HW) % (unsigned)C);
  const float mean = stat[c], rstd = stat[C + c], g = gamma[c];
  const float xh = (x[i] - mean) * rstd;
  const float dz = dy[i] * act_grad_f(fmaf(xh, g, beta[c]), act);
  float v = dz;
  if (training) v -= (dsum[c] + xh * dsum[C + c]) * inv_count;
  dx[i] = g * rstd * v;
}

inline int bn_chunk(int N, int HW) {                                   // ~8k elements per workgroup of the reduction kernels
  int n = 8192 / (HW > 0 ? HW : 1);
  if (n < 1) n = 1;
  return n > N ? N : n;
}

}  // namespace

extern "C" int lafs_bn_act_fwd_nchw(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    float eps, float momentum, int training, int N, int C, int HW, int act, float* sums_ws, float* stat,
                                    float* y, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && gamma && beta && stat && y && N > 0 && C > 0 && HW > 0, "bad operand");
  LAFS_CHECK_ARG(training ? (sums_ws != nullptr) : (running_mean && running_var), "training needs sums_ws, eval needs running statistics");
  LAFS_CHECK_ARG(C <= 65535 && (long)N * C * HW < (1L << 32), "tensor too large for the 32-bit index arithmetic");
  if (training) {
    (void)hipMemsetAsync(sums_ws, 0, 2 * (size_t)C * sizeof(float), stream);
    const int npb = bn_chunk(N, HW);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, (N + npb - 1) / npb), dim3(256), 0, stream, x, N, C, HW, npb, sums_ws);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sums_ws, (float)N * (float)HW, eps, momentum, C, training,
                     running_mean, running_var, stat);
  const long total = (long)N * C * HW;
  if (HW % 4 == 0)
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(blocks_for(total / 4)), dim3(256), 0, stream, x, stat, gamma, beta, total / 4, C, HW / 4, act, y);
  else
    hipLaunchKernelGGL(bn_act_fwd_scalar_kernel, dim3(blocks_for(total)), dim3(256), 0, stream, x, stat, gamma, beta, total, C, HW, act, y);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_bn_act_bwd_nchw(const float* x, const float* dy, const float* stat, const float* gamma, const float* beta, int training,
                                    int N, int C, int HW, int act, float* dsum, float* dx, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && dy && stat && gamma && beta && dsum && dx && N > 0 && C > 0 && HW > 0 && C <= 65535, "bad operand");
  LAFS_CHECK_ARG((long)N * C * HW < (1L << 32), "tensor too large for the 32-bit index arithmetic");
  (void)hipMemsetAsync(dsum, 0, 2 * (size_t)C * sizeof(float), stream);
  const int npb = bn_chunk(N, HW);
  hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(C, (N + npb - 1) / npb), dim3(256), 0, stream, x, dy, stat, gamma, beta, N, C, HW, npb, act,
                     dsum);
  const long total = (long)N * C * HW;
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(blocks_for(total)), dim3(256), 0, stream, x, dy, stat, gamma, beta, dsum,
                     1.0f / ((float)N * (float)HW), total, C, HW, act, training, dx);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
