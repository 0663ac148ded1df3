// Training kernels of the TRAINABLE landmark CNN (MobileNetV3-large trunk of Part-fViT with_land=True: reference
// face_pre_pro/mobilenet.py:224-313 driven by ViT_face.py:679-711, trained by train_largescale.py:785-891).
//
// Round 2 left this branch of the fine-tune step on torch autograd over MIOpen (~700 launches, 10.8 of 36.3 ms at batch 128: fp32
// NCHW BatchNorm, 1x1 convolutions, elementwise chains).  Here it is a launch plan over NHWC bf16 activations whose channel count
// is padded to a multiple of 32 (pad channels are exactly zero everywhere), so that
//   * every 1x1 convolution, its input gradient and its weight gradient are lafs_gemm_nt / lafs_wgrad calls on [N H W, C] matrices;
//   * BatchNorm in TRAINING mode (batch statistics, biased variance for the normalisation, unbiased for running_var, momentum
//     update: nn.BatchNorm2d) is a statistics pass (column sums) + an apply pass fused with the activation (+ residual add); its
//     backward a reduction pass (sum dz, sum dz xhat, with the activation's derivative recomputed from the raw input) + an apply pass;
//   * the depthwise convolutions (forward, input gradient, weight gradient), squeeze-excite rescale and its backward, pooling
//     backward, the 3x3 stem as im2col rows, and the min-max landmark scaling backward are the bandwidth-bound kernels below;
//   * weights stay fp32 in the parameter arena; padded bf16 operand copies (W and W^T) are refreshed by ONE table-driven launch per
//     optimizer step, padded fp32 weight gradients are folded back into the arena by ONE table-driven launch per micro-step.
// fp32 accumulation everywhere.  Round 4: activations, operand images and activation gradients are IEEE fp16 (round 3: bf16) -- the
// format of the reference's own autocast run (train_largescale.py:803-804); gfx950 runs f16 MFMA at the bf16 rate, and on this
// freshly initialised batch-statistics network fp16 roundings move the regressor by ~1 % where bf16 ones moved it by 9 % (F18).
// Gradients are scaled into fp16's range by a per-call factor derived ON THE DEVICE from max |d loss / d t| (lafs_cnn_grad_scale)
// and un-scaled where they leave the 16-bit domain (BatchNorm affine gradients, the fold of the padded weight gradients).
// BatchNorm sums are accumulated in fp64 (per-block fp32 partials, one fp64 atomic each): mean and E[x^2] - mean^2 are formed in
// double, so neither the cancellation nor the order of the atomics reaches the fp32 statistics.
// (The containers keep the names bf16_t / uint4 of the other files: raw 16-bit lanes.)
#include <algorithm>
#include "common.hpp"
#include "lafs_hip.h"

namespace {

inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  f[0] = h_lo(v.x); f[1] = h_hi(v.x); f[2] = h_lo(v.y); f[3] = h_hi(v.y);
  f[4] = h_lo(v.z); f[5] = h_hi(v.z); f[6] = h_lo(v.w); f[7] = h_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_h2(f[0], f[1]), pack_h2(f[2], f[3]), pack_h2(f[4], f[5]), pack_h2(f[6], f[7]));
}
// derivative of the block non-linearities w.r.t. their INPUT z (MobileNetV3: relu, x relu6(x+3)/6, relu6(x+3)/6)
__device__ __forceinline__ float act_grad_f(float z, int act) {
  if (act == 1) return z > 0.f ? 1.f : 0.f;
  if (act == 2) return z <= -3.f ? 0.f : (z >= 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
  if (act == 3) return (z > -3.f && z < 3.f) ? (1.f / 6.f) : 0.f;
  return 1.f;
}

// ---- stem as im2col: x f32 NCHW [N,3,S,S] -> P bf16 [N So So, 32]: columns (c, ky, kx) of the 3x3 stride-2 pad-1 window, 27..31 zero
__global__ __launch_bounds__(256) void im2col_stem_kernel(const float* __restrict__ x, int N, int S, bf16_t* __restrict__ P) {
  const int So = S >> 1;
  const long total = (long)N * So * So;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ox = (int)(idx % So), oy = (int)((idx / So) % So), n = (int)(idx / ((long)So * So));
  const float* xn = x + (size_t)n * 3 * S * S;
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (iy >= 0 && iy < S && ix >= 0 && ix < S) v[c * 9 + ky * 3 + kx] = xn[((size_t)c * S + iy) * S + ix];
      }
    }
  uint4* o = reinterpret_cast<uint4*>(P + (size_t)idx * 32);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    o[k] = make_uint4(pack_h2(v[8 * k], v[8 * k + 1]), pack_h2(v[8 * k + 2], v[8 * k + 3]), pack_h2(v[8 * k + 4], v[8 * k + 5]),
                      pack_h2(v[8 * k + 6], v[8 * k + 7]));
}

// Column reductions over the rows of bf16 [R, ld] matrices.  Thread layout: GB channel groups (8 channels each) x 256 / GB row lanes,
// GB = 4, 8, 16 or 32 (the smallest that covers ld / 8, or 32 with grid.y tiles for wider matrices): at 32 channels a wave reads
// 16 whole rows = 1 KiB contiguous, and every thread of the workgroup has work (a fixed 32-group layout left 7 of 8 threads idle
// on the early, largest layers of the trunk).  Two sums per channel, LDS tree over the row lanes (a fixed order: the workgroup's partial is deterministic), ONE fp64 atomic per channel and workgroup into
// out[0..C), [C..2C): the order of the atomics moves the sums at the 1e-16 level, far below the fp32 statistics derived from them.
template <typename F>
__device__ __forceinline__ void col_reduce2(int GB, int C, long R, int rows_per_block, double* __restrict__ out, F body) {
  __shared__ float red[256][17];                       // (+1: the tree below walks columns of 16 floats)
  const int RL = 256 / GB;
  const int gl = threadIdx.x % GB, rl = threadIdx.x / GB;
  const int c = (blockIdx.y * GB + gl) * 8;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (c < C) body(c, r0 + rl, r1, RL, s1, s2);
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][8 + e] = s2[e]; }
  __syncthreads();
  for (int half = RL >> 1; half > 0; half >>= 1) {     // tree over the row lanes (thread = rl * GB + gl)
    if (rl < half) {
#pragma unroll
      for (int e = 0; e < 16; ++e) red[threadIdx.x][e] += red[threadIdx.x + half * GB][e];
    }
    __syncthreads();
  }
  // one atomic per thread, consecutive addresses per wave instruction (16 serial atomics from GB threads each made the kernel 2-5x
  // slower: the atomic round trips, not the reads, were its critical path)
  for (int i = threadIdx.x; i < GB * 16; i += 256) {
    const int half = i / (GB * 8), j = i % (GB * 8);     // j = channel offset inside this block's GB * 8 channels
    const int cc = blockIdx.y * GB * 8 + j;
    if (cc < C) atomicAdd(out + half * C + cc, (double)red[j >> 3][half * 8 + (j & 7)]);      // global_atomic_add_f64
  }
}
inline int group_block(int ld) { const int g = ld / 8; return g <= 4 ? 4 : (g <= 8 ? 8 : (g <= 16 ? 16 : 32)); }

// BatchNorm statistics: sums[c] += sum_r x[r,c], sums[C + c] += sum_r x[r,c]^2
__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16_t* __restrict__ x, int ldx, long R, int C, int rows_per_block, int GB,
                                                       double* __restrict__ sums) {
  col_reduce2(GB, C, R, rows_per_block, sums, [&](int c, long ra, long rb, int step, float (&s1)[8], float (&s2)[8]) {
    long r = ra;
    for (; r + 3 * (long)step < rb; r += 4 * (long)step) {                 // four rows in flight per thread
      uint4 q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = *reinterpret_cast<const uint4*>(x + (size_t)(r + k * (long)step) * ldx + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v[8];
        unpack8(q[k], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
      }
    }
    for (; r < rb; r += step) {
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(x + (size_t)r * ldx + c), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
    }
  });
}

// BatchNorm apply (training): y = act(x scale + shift) (+ resid) with scale = rstd gamma, shift = beta - mean scale staged in LDS
// once per workgroup; block 0 also writes stat = {mean[C], rstd[C]} and updates the running statistics.  Pad channels (>= C) -> 0.
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ x, int ldx, long R, int C, const double* __restrict__ sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       int act, const bf16_t* __restrict__ resid, int ldr, bf16_t* __restrict__ y, int ldy,
                                                       float* __restrict__ stat) {
  extern __shared__ float sc[];                          // [2 ldy]: scale, shift
  const int C8 = ldy >> 3;
  const long total = R * C8;
  const double invR = 1.0 / (double)R;
  for (int c = threadIdx.x; c < ldy; c += 256) {
    float scale = 0.f, shift = 0.f;
    if (c < C) {
      const double md = sums[c] * invR, vd = fmax(sums[C + c] * invR - md * md, 0.0);       // fp64: no cancellation at |mean| >> std
      const float m = (float)md, var = (float)vd, rs = (float)(1.0 / sqrt(vd + (double)eps));
      scale = rs * gamma[c]; shift = beta[c] - m * scale;
      if (blockIdx.x == 0) {
        stat[c] = m; stat[C + c] = rs;
        if (running_mean != nullptr) {
          running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
          running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * ((float)R / (float)(R > 1 ? R - 1 : 1));
        }
      }
    }
    sc[c] = scale; sc[ldy + c] = shift;
  }
  __syncthreads();
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C8) * 8;
    const long r = idx / C8;
    float v[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(x + (size_t)r * ldx + c), v);
    float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (resid != nullptr) unpack8(*reinterpret_cast<const uint4*>(resid + (size_t)r * ldr + c), rs);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (c + e < C) ? act_f(fmaf(v[e], sc[c + e], sc[ldy + c + e]), act) + rs[e] : 0.f;
    *reinterpret_cast<uint4*>(y + (size_t)r * ldy + c) = pack8(o);
  }
}

// BatchNorm backward, pass 1: dz = (dy + add[n, c] / HW) act'(z), z recomputed from the raw input;  dsums[c] += sum dz,
// dsums[C + c] += sum dz xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx,
                                                            long R, int C, const float* __restrict__ stat, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int act, const bf16_t* __restrict__ add,
                                                            int ldadd, int HW, int rows_per_block, int GB, double* __restrict__ dsums) {
  col_reduce2(GB, C, R, rows_per_block, dsums, [&](int c, long ra, long rb, int step, float (&s1)[8], float (&s2)[8]) {
    float m[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cc = min(c + e, C - 1);
      m[e] = stat[cc]; rs[e] = stat[C + cc]; ga[e] = gamma[cc]; be[e] = beta[cc];
    }
    const float invHW = 1.0f / (float)HW;
    uint4 nv = make_uint4(0, 0, 0, 0), nd = nv;          // the next row's operands are requested before this row's arithmetic
    if (ra < rb) { nv = *reinterpret_cast<const uint4*>(x + (size_t)ra * ldx + c); nd = *reinterpret_cast<const uint4*>(dy + (size_t)ra * lddy + c); }
    for (long r = ra; r < rb; r += step) {
      float v[8], d[8];
      unpack8(nv, v);
      unpack8(nd, d);
      if (r + step < rb) {
        nv = *reinterpret_cast<const uint4*>(x + (size_t)(r + step) * ldx + c);
        nd = *reinterpret_cast<const uint4*>(dy + (size_t)(r + step) * lddy + c);
      }
      if (add != nullptr) {
        float a[8];
        unpack8(*reinterpret_cast<const uint4*>(add + (size_t)(r / HW) * ldadd + c), a);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = fmaf(a[e], invHW, d[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[e] - m[e]) * rs[e];
        const float dz = (c + e < C) ? d[e] * act_grad_f(fmaf(xh, ga[e], be[e]), act) : 0.f;
        s1[e] += dz; s2[e] = fmaf(dz, xh, s2[e]);
      }
    }
  });
}

// BatchNorm backward, pass 2: dx = gamma rstd (dz - sum dz / R - xhat sum(dz xhat) / R) with the per-channel constants staged in
// LDS once per workgroup; block 0 adds dgamma, dbeta into the arena
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx,
                                                           long R, int C, const float* __restrict__ stat, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int act, const bf16_t* __restrict__ add,
                                                           int ldadd, int HW, const double* __restrict__ dsums, bf16_t* __restrict__ dx,
                                                           int lddx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           const float* __restrict__ gscale, int frozen) {
  extern __shared__ float sc[];                          // [6 lddx]: mean, rstd, gamma, beta, sum dz / R, sum dz xhat / R
  const int C8 = lddx >> 3;
  const long total = R * C8;
  const double invR = 1.0 / (double)R;
  const float invHW = 1.0f / (float)HW;
  for (int c = threadIdx.x; c < lddx; c += 256) {
    const bool ok = c < C;
    sc[c] = ok ? stat[c] : 0.f; sc[lddx + c] = ok ? stat[C + c] : 0.f; sc[2 * lddx + c] = ok ? gamma[c] : 0.f;
    sc[3 * lddx + c] = ok ? beta[c] : 0.f;
    // (frozen statistics -- eval mode -- are constants of the backward: dx = gamma rstd dz, no batch-mean terms)
    sc[4 * lddx + c] = (ok && !frozen) ? (float)(dsums[c] * invR) : 0.f; sc[5 * lddx + c] = (ok && !frozen) ? (float)(dsums[C + c] * invR) : 0.f;
    if (ok && blockIdx.x == 0 && dgamma != nullptr) {      // the affine gradients leave the scaled 16-bit domain here: x 1 / scale
      const double inv = gscale != nullptr ? (double)gscale[1] : 1.0;
      dgamma[c] += (float)(dsums[C + c] * inv); dbeta[c] += (float)(dsums[c] * inv);
    }
  }
  __syncthreads();
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C8) * 8;
    const long r = idx / C8;
    float v[8], d[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(x + (size_t)r * ldx + c), v);
    unpack8(*reinterpret_cast<const uint4*>(dy + (size_t)r * lddy + c), d);
    if (add != nullptr) {
      float a[8];
      unpack8(*reinterpret_cast<const uint4*>(add + (size_t)(r / HW) * ldadd + c), a);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = fmaf(a[e], invHW, d[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cc = c + e;
      const float rs = sc[lddx + cc], ga = sc[2 * lddx + cc];
      const float xh = (v[e] - sc[cc]) * rs;
      const float dz = d[e] * act_grad_f(fmaf(xh, ga, sc[3 * lddx + cc]), act);
      o[e] = ga * rs * (dz - sc[4 * lddx + cc] - xh * sc[5 * lddx + cc]);        // pad channels: gamma = 0 -> 0
    }
    *reinterpret_cast<uint4*>(dx + (size_t)r * lddx + c) = pack8(o);
  }
}

// ---- depthwise k x k convolution on NHWC bf16 (no bias, no activation).  Weights: tap-major fp32 image wt [k*k][ld] (pad channels
// zero) built from the module's [C][1][k][k] tensor by lafs_cnn_dw_layout_table -- a thread's 8 channels of one tap are 32 contiguous
// bytes (read straight from the [C][k*k] tensor they were 8 loads 100 bytes apart per tap: 95 us for a 6 MB layer)
template <int K>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, int N, int H, int W, int ld,
                                                     int C, int stride, bf16_t* __restrict__ y) {
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride, C8 = ld >> 3;
  const long total = (long)N * Ho * Wo * C8;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C8) * 8;
  const long pix = idx / C8;
  const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), n = (int)(pix / ((long)Wo * Ho));
  constexpr int P = (K - 1) / 2;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16_t* xn = x + (size_t)n * H * W * ld + c;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy * stride - P + ky;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int ix = ox * stride - P + kx;
      if (ix < 0 || ix >= W) continue;
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(xn + ((size_t)iy * W + ix) * ld), v);
      const float* wk = w + (size_t)(ky * K + kx) * ld + c;
      const float4 w0 = *reinterpret_cast<const float4*>(wk), w1 = *reinterpret_cast<const float4*>(wk + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(v[e], wv[e], acc[e]);
    }
  }
  *reinterpret_cast<uint4*>(y + (size_t)pix * ld + c) = pack8(acc);
}

// dx[n,iy,ix,c] = sum over the taps with (iy + P - ky) divisible by the stride and in range of dy[n,oy,ox,c] w[c][ky,kx]
template <int K>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ w, int N, int H, int W,
                                                          int ld, int C, int stride, bf16_t* __restrict__ dx) {
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride, C8 = ld >> 3;
  const long total = (long)N * H * W * C8;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C8) * 8;
  const long pix = idx / C8;
  const int ix = (int)(pix % W), iy = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
  constexpr int P = (K - 1) / 2;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16_t* dn = dy + (size_t)n * Ho * Wo * ld + c;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int ty = iy + P - ky;
    if (ty < 0 || ty % stride != 0) continue;
    const int oy = ty / stride;
    if (oy >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int tx = ix + P - kx;
      if (tx < 0 || tx % stride != 0) continue;
      const int ox = tx / stride;
      if (ox >= Wo) continue;
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(dn + ((size_t)oy * Wo + ox) * ld), v);
      const float* wk = w + (size_t)(ky * K + kx) * ld + c;
      const float4 w0 = *reinterpret_cast<const float4*>(wk), w1 = *reinterpret_cast<const float4*>(wk + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(v[e], wv[e], acc[e]);
    }
  }
  *reinterpret_cast<uint4*>(dx + (size_t)pix * ld + c) = pack8(acc);
}

// dwt[ky,kx][c] += sum_{n,oy,ox} dy[n,oy,ox,c] x[n,iy,ix,c] (tap-major image, folded into the arena's [C][k*k] gradient by
// lafs_cnn_unpad_add_table).  Grid (pixel slabs, channel groups of 8): a thread walks a strided set of output pixels of its slab with
// K*K x 8 accumulators; wave reduction, then fp32 atomics.
template <int K>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, int N, int H, int W,
                                                            int ld, int C, int stride, int pix_per_block, float* __restrict__ dw) {
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const int c = blockIdx.y * 8;
  const long npix = (long)N * Ho * Wo;
  const long p0 = (long)blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
  constexpr int P = (K - 1) / 2;
  float acc[K * K][8];
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  for (long pix = p0 + threadIdx.x; pix < p1; pix += 256) {
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), n = (int)(pix / ((long)Wo * Ho));
    float g[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + (size_t)pix * ld + c), g);
    const bf16_t* xn = x + (size_t)n * H * W * ld + c;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * stride - P + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int ix = ox * stride - P + kx;
        if (ix < 0 || ix >= W) continue;
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(xn + ((size_t)iy * W + ix) * ld), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[ky * K + kx][e] = fmaf(g[e], v[e], acc[ky * K + kx][e]);
      }
    }
  }
  __shared__ float red[4][K * K * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float s = wave_sum(acc[t][e]);
      if (lane == 0) red[wave][t * 8 + e] = s;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K * 8; i += 256) {
    const int t = i >> 3, e = i & 7;
    if (c + e < C) atomicAdd(dw + (size_t)t * ld + c + e, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

// ---- squeeze-excite: out = act(z gate[n, c]) out of place (z is kept for the backward)
__global__ __launch_bounds__(256) void scale_act_out_kernel(const bf16_t* __restrict__ z, const bf16_t* __restrict__ s, int lds_, int N, int HW,
                                                            int ld, int act, bf16_t* __restrict__ out) {
  const int C8 = ld >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * HW * C8) return;
  const int c = (int)(idx % C8) * 8;
  const long pix = idx / C8;
  const int n = (int)(pix / HW);
  float v[8], g[8], o[8];
  unpack8(*reinterpret_cast<const uint4*>(z + (size_t)pix * ld + c), v);
  unpack8(*reinterpret_cast<const uint4*>(s + (size_t)n * lds_ + c), g);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = act_f(v[e] * g[e], act);
  *reinterpret_cast<uint4*>(out + (size_t)pix * ld + c) = pack8(o);
}
// backward: ds = dout act'(z gate); dz = ds gate (bf16, written); dgate[n, c] = sum_p ds z (fp32).  One workgroup per (image,
// tile of GB channel groups): the pixels are spread over 256 / GB lanes and reduced through the LDS (one thread per (image, group)
// walking all pixels left the early 14x14 layers with 2048 threads on the whole chip: 85 us each).
__global__ __launch_bounds__(256) void se_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ z, const bf16_t* __restrict__ gate,
                                                     int ldg, int N, int HW, int ld, int act, int GB, bf16_t* __restrict__ dz,
                                                     float* __restrict__ dgate, int lddg) {
  __shared__ float red[256][9];
  const int PL = 256 / GB;
  const int gl = threadIdx.x % GB, pl = threadIdx.x / GB;
  const int c = (blockIdx.y * GB + gl) * 8, n = blockIdx.x;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < ld) {
    float g[8];
    unpack8(*reinterpret_cast<const uint4*>(gate + (size_t)n * ldg + c), g);
    for (int p = pl; p < HW; p += PL) {
      const size_t off = ((size_t)n * HW + p) * ld + c;
      float d[8], v[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(dout + off), d);
      unpack8(*reinterpret_cast<const uint4*>(z + off), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ds = d[e] * act_grad_f(v[e] * g[e], act);
        o[e] = ds * g[e];
        acc[e] = fmaf(ds, v[e], acc[e]);
      }
      *reinterpret_cast<uint4*>(dz + off) = pack8(o);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  for (int half = PL >> 1; half > 0; half >>= 1) {
    if (pl < half) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[threadIdx.x + half * GB][e];
    }
    __syncthreads();
  }
  if (threadIdx.x < GB && c < ld) {
    float* dg = dgate + (size_t)n * lddg + c;
    *reinterpret_cast<float4*>(dg) = make_float4(red[threadIdx.x][0], red[threadIdx.x][1], red[threadIdx.x][2], red[threadIdx.x][3]);
    *reinterpret_cast<float4*>(dg + 4) = make_float4(red[threadIdx.x][4], red[threadIdx.x][5], red[threadIdx.x][6], red[threadIdx.x][7]);
  }
}
// out(bf16)[n, c] = mean_p x[n, p, c] with the same workgroup shape (the squeeze of squeeze-excite and the final average pool of the
// TRAINING plan; the inference plan keeps lafs_cnn_pool)
__global__ __launch_bounds__(256) void pool_wg_kernel(const bf16_t* __restrict__ x, int N, int HW, int ld, int GB, bf16_t* __restrict__ out, int ldo) {
  __shared__ float red[256][9];
  const int PL = 256 / GB;
  const int gl = threadIdx.x % GB, pl = threadIdx.x / GB;
  const int c = (blockIdx.y * GB + gl) * 8, n = blockIdx.x;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < ld) {
    for (int p = pl; p < HW; p += PL) {
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(x + ((size_t)n * HW + p) * ld + c), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  for (int half = PL >> 1; half > 0; half >>= 1) {
    if (pl < half) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[threadIdx.x + half * GB][e];
    }
    __syncthreads();
  }
  if (threadIdx.x < GB && c < ld) {
    const float inv = 1.0f / (float)HW;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = red[threadIdx.x][e] * inv;
    *reinterpret_cast<uint4*>(out + (size_t)n * ldo + c) = pack8(o);
  }
}
// out(bf16)[i] = dy[i] act'(.) with the derivative taken from the POST-activation value y (valid for relu and h-sigmoid: the
// squeeze-excite FCs); dy fp32 (dy_f32 != null) or bf16
__global__ __launch_bounds__(256) void act_bwd_post_kernel(const float* __restrict__ dy_f32, const bf16_t* __restrict__ dy_bf, const bf16_t* __restrict__ y,
                                                           long n, int act, bf16_t* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float yv = h2f(y[i]);
  const float d = dy_f32 != nullptr ? dy_f32[i] : h2f(dy_bf[i]);
  float g = 1.f;
  if (act == 1) g = yv > 0.f ? 1.f : 0.f;
  else if (act == 3) g = (yv > 0.f && yv < 1.f) ? (1.f / 6.f) : 0.f;
  out[i] = f2h(d * g);
}
// dx[n, p, c] = dfeat[n, c] / HW   (backward of the final average pool)
__global__ __launch_bounds__(256) void pool_bwd_kernel(const bf16_t* __restrict__ dfeat, int ldf, int N, int HW, int ld, bf16_t* __restrict__ dx) {
  const int C8 = ld >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * HW * C8) return;
  const int c = (int)(idx % C8) * 8;
  const long pix = idx / C8;
  float v[8];
  unpack8(*reinterpret_cast<const uint4*>(dfeat + (size_t)(pix / HW) * ldf + c), v);
  const float inv = 1.0f / (float)HW;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= inv;
  *reinterpret_cast<uint4*>(dx + (size_t)pix * ld + c) = pack8(v);
}

// ---- table-driven operand refresh: entry e = {src offset (floats into master), rows, cols, dst offset (bf16 elements), dst ld,
// transpose}: dst[r][c] = bf16(src[r][c]) for r < rows, c < cols, zero elsewhere inside the padded [prow x ld] image
// (transpose: dst[c][r] = src[r][c]).  One workgroup per 1024 destination elements; starts[] = prefix sums of workgroups.
__global__ __launch_bounds__(256) void pad_cast_table_kernel(const float* __restrict__ master, bf16_t* __restrict__ dst, const long* __restrict__ table,
                                                             const int* __restrict__ starts, int n_ent) {
  int lo = 0, hi = n_ent;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (starts[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const long* e = table + 8 * lo;
  const long src = e[0], rows = e[1], cols = e[2], doff = e[3], ld = e[4], tr = e[5], prow = e[6];
  const long local = (long)(blockIdx.x - starts[lo]) * 1024 + threadIdx.x * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long i = local + k;
    if (i >= prow * ld) return;
    const long dr = i / ld, dc = i % ld;
    const long sr = tr ? dc : dr, scol = tr ? dr : dc;
    dst[doff + i] = (sr < rows && scol < cols) ? f2h(master[src + sr * cols + scol]) : (bf16_t)0;
  }
}
// depthwise weight images: entry = {src offset (floats into master), C, k*k, dst offset (floats), ld}: dst[t][c] = src[c][t], pad 0
__global__ __launch_bounds__(256) void dw_layout_table_kernel(const float* __restrict__ master, float* __restrict__ dst, const long* __restrict__ table,
                                                              const int* __restrict__ starts, int n_ent) {
  int lo = 0, hi = n_ent;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (starts[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const long* e = table + 8 * lo;
  const long src = e[0], C = e[1], kk = e[2], doff = e[3], ld = e[4];
  const long i = (long)(blockIdx.x - starts[lo]) * 256 + threadIdx.x;
  if (i >= kk * ld) return;
  const long t = i / ld, c = i % ld;
  dst[doff + i] = c < C ? master[src + c * kk + t] : 0.f;
}
// gradient fold: entry = {padded src offset (floats), rows, cols, src ld, arena grad offset, transpose}: grad[r][c] += src[r][c]
// (transpose: the source image is [cols][ld] tap-major and grad[r][c] += src[c][r]: depthwise weight gradients)
__global__ __launch_bounds__(256) void unpad_add_table_kernel(const float* __restrict__ padded, float* __restrict__ grad, const long* __restrict__ table,
                                                              const int* __restrict__ starts, int n_ent, const float* __restrict__ gscale) {
  int lo = 0, hi = n_ent;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (starts[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const long* e = table + 8 * lo;
  const long src = e[0], rows = e[1], cols = e[2], ld = e[3], goff = e[4], tr = e[5];
  const long i = (long)(blockIdx.x - starts[lo]) * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols, c = i % cols;
  const float inv = gscale != nullptr ? gscale[1] : 1.f;               // un-scale: see lafs_cnn_grad_scale
  grad[goff + i] += inv * (tr ? padded[src + c * ld + r] : padded[src + r * ld + c]);
}

// ---- backward of the per-image min-max scaling theta = (t - min) / (max - min) * 111 (ViT_face.py:698-706): the gradient also
// reaches the arg-min and arg-max entries (first occurrence, as torch.max / torch.min select)
__global__ __launch_bounds__(256) void theta_bwd_kernel(const float* __restrict__ t, const float* __restrict__ dth, int n, float* __restrict__ dt) {
  __shared__ float smn[256], smx[256], sa[256], sb[256];
  __shared__ int imn[256], imx[256];
  const float* tb = t + (size_t)blockIdx.x * n;
  const float* gb = dth + (size_t)blockIdx.x * n;
  float mn = 3.4e38f, mx = -3.4e38f; int in_ = 0, ix_ = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = tb[i];
    if (v < mn) { mn = v; in_ = i; }
    if (v > mx) { mx = v; ix_ = i; }
  }
  smn[threadIdx.x] = mn; smx[threadIdx.x] = mx; imn[threadIdx.x] = in_; imx[threadIdx.x] = ix_;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const int o = threadIdx.x + s;
      if (smn[o] < smn[threadIdx.x] || (smn[o] == smn[threadIdx.x] && imn[o] < imn[threadIdx.x])) { smn[threadIdx.x] = smn[o]; imn[threadIdx.x] = imn[o]; }
      if (smx[o] > smx[threadIdx.x] || (smx[o] == smx[threadIdx.x] && imx[o] < imx[threadIdx.x])) { smx[threadIdx.x] = smx[o]; imx[threadIdx.x] = imx[o]; }
    }
    __syncthreads();
  }
  mn = smn[0]; mx = smx[0];
  const float r = mx - mn, k = 111.0f / r, k2 = 111.0f / (r * r);
  float a = 0.f, b = 0.f;                              // a = d/dmin, b = d/dmax
  for (int i = threadIdx.x; i < n; i += 256) {
    const float g = gb[i], v = tb[i];
    dt[(size_t)blockIdx.x * n + i] = g * k;
    a += g * k2 * (v - mx);
    b -= g * k2 * (v - mn);
  }
  sa[threadIdx.x] = a; sb[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dt[(size_t)blockIdx.x * n + imn[0]] += sa[0];
    dt[(size_t)blockIdx.x * n + imx[0]] += sb[0];
  }
}

// ---- gradient scale + overflow guard (the reference's torch.cuda.amp.GradScaler, train_largescale.py:739, 867-880, on the device).
// state(f32)[8] = {scale, 1 / scale, target, found_inf, skipped backwards, clean backwards in a row, -, -}:
//   grad_scale  (start of a backward)  scale = the power of two that brings max |g| to ~target (target = state[2], or the argument
//               while state[2] <= 0), clamped to [2^-24, 2^24]; found_inf = 1 when g itself is not finite (fmaxf would skip a NaN);
//   grad_check  (end of a backward)    found_inf |= any non-finite entry of the CNN's gradient range of the arena;
//   grad_guard                         found_inf: the range is zeroed (this window's CNN update is dropped: the moments and weights stay
//               finite), target /= 2 (>= 1), skipped += 1; else after 2000 clean backwards in a row target *= 2 (<= target_max).
// one workgroup (the gradient of the raw regressor output is a few thousand floats)
__global__ __launch_bounds__(256) void grad_scale_kernel(const float* __restrict__ g, long n, float target, float* __restrict__ scale,
                                                        int has_state) {
  __shared__ float red[4];
  __shared__ int bad[4];
  float m = 0.f;
  int b = 0;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float v = fabsf(g[i]);
    if (!(v < 3.0e38f)) b = 1;                                         // inf or NaN
    else m = fmaxf(m, v);
  }
  m = wave_max(m);
  b = __any(b) ? 1 : 0;
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; bad[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (has_state && scale[2] > 0.f) target = scale[2];
    float s = (m > 0.f) ? target / m : 1.f;
    s = fminf(fmaxf(s, 5.9604645e-8f), 16777216.0f);                   // [2^-24, 2^24]: finite, and 1 / s is finite
    s = exp2f(floorf(log2f(s)));                                       // a power of two: scaling and un-scaling are exact
    scale[0] = s; scale[1] = 1.0f / s;
    if (has_state) scale[3] = (bad[0] | bad[1] | bad[2] | bad[3]) ? 1.f : 0.f;
  }
}
__global__ __launch_bounds__(256) void grad_check_kernel(const float* __restrict__ g, long n, float* __restrict__ state) {
  int b = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b |= !(fabsf(g[i]) < 3.0e38f);
  if (__any(b) && (threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(state + 3), 0x3f800000u);   // 1.0f
}
__global__ __launch_bounds__(256) void grad_guard_kernel(float* __restrict__ g, long n, float* __restrict__ state, float target_max) {
  const bool found = state[3] != 0.f;
  if (found)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) g[i] = 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {                          // (state[3] itself is reset by the next grad_scale)
    float t = state[2] > 0.f ? state[2] : target_max;
    if (found) { t = fmaxf(t * 0.5f, 1.f); state[4] += 1.f; state[5] = 0.f; }
    else if ((state[5] += 1.f) >= 2000.f) { t = fminf(t * 2.f, target_max); state[5] = 0.f; }
    state[2] = t;
  }
}
// dst(fp16)[r, c] = src(f32)[r, c] * scale[0] for c < cols, 0 for cols <= c < ld   (scale == nullptr: 1)
__global__ __launch_bounds__(256) void cast_pad_f16_kernel(const float* __restrict__ src, int rows, int cols, bf16_t* __restrict__ dst, int ld,
                                                           const float* __restrict__ scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * ld) return;
  const int r = (int)(i / ld), c = (int)(i % ld);
  const float s = scale != nullptr ? scale[0] : 1.f;
  dst[i] = c < cols ? f2h(src[(size_t)r * cols + c] * s) : (bf16_t)0;
}
__global__ __launch_bounds__(256) void cast_f16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = h2f(src[i]);
}

#define K_DISPATCH(KERNEL, ...)                                    \
  do {                                                             \
    if (k == 3) hipLaunchKernelGGL((KERNEL<3>), __VA_ARGS__);      \
    else hipLaunchKernelGGL((KERNEL<5>), __VA_ARGS__);             \
  } while (0)

int rows_per_block_for(long R) {                       // ~1024 row slabs at most (4 workgroups per CU), at least 256 rows each
  long rpb = (R + 1023) / 1024;
  if (rpb < 256) rpb = 256;
  return (int)((rpb + 63) / 64 * 64);
}

}  // namespace

extern "C" int lafs_cnn_im2col_stem(const float* x, int N, int S, void* P, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && P && N > 0 && S > 0 && S % 2 == 0, "bad operand");
  hipLaunchKernelGGL(im2col_stem_kernel, dim3(blocks_for((long)N * (S / 2) * (S / 2))), dim3(256), 0, stream, x, N, S, (bf16_t*)P);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_bn_stats(const void* x, int ldx, int64_t R, int C, double* sums, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && sums && R > 0 && C > 0 && ldx >= C && ldx % 8 == 0, "bad operand");
  const int rpb = rows_per_block_for(R), GB = group_block(ldx);
  hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)((R + rpb - 1) / rpb), (unsigned)((C + 8 * GB - 1) / (8 * GB))), dim3(256), 0, stream,
                     (const bf16_t*)x, ldx, (long)R, C, rpb, GB, sums);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_bn_apply(const void* x, int ldx, int64_t R, int C, const double* sums, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, int act, const void* resid, int ldr, void* y,
                                 int ldy, float* stat, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && sums && gamma && beta && y && stat && R > 0 && C > 0, "null operand");
  LAFS_CHECK_ARG(ldx >= ldy && ldy >= C && ldy % 8 == 0 && ldx % 8 == 0 && (resid == nullptr || (ldr >= ldy && ldr % 8 == 0)), "bad strides");
  LAFS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running statistics come in pairs");
  long blocks = (R * (ldy / 8) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 2 * ldy * sizeof(float), stream, (const bf16_t*)x, ldx, (long)R, C, sums, gamma, beta, eps,
                     momentum, running_mean, running_var, act, (const bf16_t*)resid, ldr, (bf16_t*)y, ldy, stat);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

static int bn_bwd_impl(const void* dy, int lddy, const void* x, int ldx, int64_t R, int C, const float* stat, const float* gamma,
                       const float* beta, int act, const void* add_nc, int ldadd, int HW, double* dsums, void* dx, int lddx,
                       float* dgamma, float* dbeta, const float* grad_scale, int frozen, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(dy && x && stat && gamma && beta && dsums && dx && R > 0 && C > 0 && HW > 0, "null operand");
  LAFS_CHECK_ARG(lddy % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 && lddx >= C && lddy >= lddx && ldx >= lddx, "bad strides");
  LAFS_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr) && (add_nc == nullptr || (ldadd >= lddx && ldadd % 8 == 0)), "bad operand");
  const int rpb = rows_per_block_for(R), GB = group_block(lddx);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)((R + rpb - 1) / rpb), (unsigned)((C + 8 * GB - 1) / (8 * GB))), dim3(256), 0, stream,
                     (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, (long)R, C, stat, gamma, beta, act, (const bf16_t*)add_nc, ldadd, HW,
                     rpb, GB, dsums);
  LAFS_LAUNCH_CHECK();
  long blocks = (R * (lddx / 8) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 6 * lddx * sizeof(float), stream, (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, (long)R,
                     C, stat, gamma, beta, act, (const bf16_t*)add_nc, ldadd, HW, dsums, (bf16_t*)dx, lddx, dgamma, dbeta, grad_scale, frozen);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_bn_bwd(const void* dy, int lddy, const void* x, int ldx, int64_t R, int C, const float* stat, const float* gamma,
                               const float* beta, int act, const void* add_nc, int ldadd, int HW, double* dsums, void* dx, int lddx,
                               float* dgamma, float* dbeta, const float* grad_scale, hipStream_t stream) {
  return bn_bwd_impl(dy, lddy, x, ldx, R, C, stat, gamma, beta, act, add_nc, ldadd, HW, dsums, dx, lddx, dgamma, dbeta, grad_scale, 0, stream);
}

extern "C" int lafs_cnn_bn_bwd_eval(const void* dy, int lddy, const void* x, int ldx, int64_t R, int C, const float* stat, const float* gamma,
                                    const float* beta, int act, const void* add_nc, int ldadd, int HW, double* dsums, void* dx, int lddx,
                                    float* dgamma, float* dbeta, const float* grad_scale, hipStream_t stream) {
  return bn_bwd_impl(dy, lddy, x, ldx, R, C, stat, gamma, beta, act, add_nc, ldadd, HW, dsums, dx, lddx, dgamma, dbeta, grad_scale, 1, stream);
}

// eval-mode BatchNorm through the same apply / backward kernels: the sums a batch with exactly the running statistics would have
__global__ void bn_eval_sums_kernel(const float* __restrict__ rm, const float* __restrict__ rv, double R, int C, double* __restrict__ sums) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    const double m = (double)rm[c], v = (double)rv[c];
    sums[c] = R * m; sums[C + c] = R * (v + m * m);
  }
}

extern "C" int lafs_cnn_bn_eval_sums(const float* running_mean, const float* running_var, int64_t R, int C, double* sums, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(running_mean && running_var && sums && R > 0 && C > 0, "bad operand");
  hipLaunchKernelGGL(bn_eval_sums_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, running_mean, running_var, (double)R, C, sums);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

static int dwt_check(const void* a, const void* b, const void* c, int N, int H, int W, int ld, int C, int k, int stride) {
  LAFS_CHECK_ARG(a && b && c && N > 0 && H > 0 && W > 0 && C > 0 && ld >= C && ld % 8 == 0, "bad operand");
  LAFS_CHECK_ARG((k == 3 || k == 5) && (stride == 1 || stride == 2), "k in {3,5}, stride in {1,2}");
  return LAFS_OK;
}

extern "C" int lafs_cnn_dwconv_train_fwd(const void* x, const float* w, int N, int H, int W, int ld, int C, int k, int stride, void* y,
                                         hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  if (int rc = dwt_check(x, w, y, N, H, W, ld, C, k, stride)) return rc;
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  K_DISPATCH(dw_fwd_kernel, dim3(blocks_for((long)N * Ho * Wo * (ld / 8))), dim3(256), 0, stream, (const bf16_t*)x, w, N, H, W, ld, C, stride,
             (bf16_t*)y);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_dwconv_train_bwd(const void* x, const void* dy, const float* w, int N, int H, int W, int ld, int C, int k, int stride,
                                         void* dx, float* dw, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  if (int rc = dwt_check(x, dy, w, N, H, W, ld, C, k, stride)) return rc;
  LAFS_CHECK_ARG(dx && dw, "null output");
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  K_DISPATCH(dw_bwd_data_kernel, dim3(blocks_for((long)N * H * W * (ld / 8))), dim3(256), 0, stream, (const bf16_t*)dy, w, N, H, W, ld, C, stride,
             (bf16_t*)dx);
  LAFS_LAUNCH_CHECK();
  const long npix = (long)N * Ho * Wo;
  long ppb = (npix + 255) / 256;                       // <= 256 pixel slabs per channel group
  if (ppb < 1024) ppb = 1024;
  K_DISPATCH(dw_bwd_weight_kernel, dim3((unsigned)((npix + ppb - 1) / ppb), (unsigned)((C + 7) / 8)), dim3(256), 0, stream, (const bf16_t*)x,
             (const bf16_t*)dy, N, H, W, ld, C, stride, (int)ppb, dw);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_scale_act_out(const void* z, const void* s, int lds_, int N, int HW, int ld, int act, void* out, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(z && s && out && N > 0 && HW > 0 && ld > 0 && ld % 8 == 0 && lds_ >= ld && lds_ % 8 == 0, "bad operand");
  hipLaunchKernelGGL(scale_act_out_kernel, dim3(blocks_for((long)N * HW * (ld / 8))), dim3(256), 0, stream, (const bf16_t*)z, (const bf16_t*)s,
                     lds_, N, HW, ld, act, (bf16_t*)out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_se_bwd(const void* dout, const void* z, const void* gate, int ldg, int N, int HW, int ld, int act, void* dz, float* dgate,
                               int lddg, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(dout && z && gate && dz && dgate && N > 0 && HW > 0 && ld > 0 && ld % 8 == 0 && ldg >= ld && ldg % 8 == 0 && lddg >= ld && lddg % 4 == 0,
                 "bad operand");
  const int GB = group_block(ld);
  hipLaunchKernelGGL(se_bwd_kernel, dim3((unsigned)N, (unsigned)((ld / 8 + GB - 1) / GB)), dim3(256), 0, stream, (const bf16_t*)dout,
                     (const bf16_t*)z, (const bf16_t*)gate, ldg, N, HW, ld, act, GB, (bf16_t*)dz, dgate, lddg);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_pool_train(const void* x, int N, int HW, int ld, void* out, int ldo, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && out && N > 0 && HW > 0 && ld > 0 && ld % 8 == 0 && ldo >= ld && ldo % 8 == 0, "bad operand");
  const int GB = group_block(ld);
  hipLaunchKernelGGL(pool_wg_kernel, dim3((unsigned)N, (unsigned)((ld / 8 + GB - 1) / GB)), dim3(256), 0, stream, (const bf16_t*)x, N, HW, ld, GB,
                     (bf16_t*)out, ldo);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_dw_layout_table(const float* master, float* dst, const int64_t* table, const int32_t* starts, int n_entries, int n_blocks,
                                        hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(master && dst && table && starts && n_entries > 0 && n_blocks > 0, "bad operand");
  hipLaunchKernelGGL(dw_layout_table_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, master, dst, (const long*)table, starts, n_entries);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_act_bwd_post(const float* dy_f32, const void* dy_bf16, const void* y, int64_t n, int act, void* out, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG((dy_f32 != nullptr) != (dy_bf16 != nullptr) && y && out && n > 0 && (act == 0 || act == 1 || act == 3), "bad operand");
  hipLaunchKernelGGL(act_bwd_post_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, dy_f32, (const bf16_t*)dy_bf16, (const bf16_t*)y, (long)n, act,
                     (bf16_t*)out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_pool_bwd(const void* dfeat, int ldf, int N, int HW, int ld, void* dx, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(dfeat && dx && N > 0 && HW > 0 && ld > 0 && ld % 8 == 0 && ldf >= ld && ldf % 8 == 0, "bad operand");
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(blocks_for((long)N * HW * (ld / 8))), dim3(256), 0, stream, (const bf16_t*)dfeat, ldf, N, HW, ld,
                     (bf16_t*)dx);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_pad_cast_table(const float* master, void* dst, const int64_t* table, const int32_t* starts, int n_entries, int n_blocks,
                                       hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(master && dst && table && starts && n_entries > 0 && n_blocks > 0, "bad operand");
  hipLaunchKernelGGL(pad_cast_table_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, master, (bf16_t*)dst, (const long*)table, starts, n_entries);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_unpad_add_table(const float* padded, float* grad, const int64_t* table, const int32_t* starts, int n_entries, int n_blocks,
                                        const float* grad_scale, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(padded && grad && table && starts && n_entries > 0 && n_blocks > 0, "bad operand");
  hipLaunchKernelGGL(unpad_add_table_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, padded, grad, (const long*)table, starts, n_entries,
                     grad_scale);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_grad_scale(const float* g, int64_t n, float target, float* scale, int has_state, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(g && scale && n > 0 && target > 0.f, "bad operand");
  hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(256), 0, stream, g, (long)n, target, scale, has_state);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
extern "C" int lafs_cnn_grad_guard(float* grad, int64_t n, float* state, float target_max, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(grad && state && n > 0 && target_max >= 1.f, "bad operand");
  const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(grad_check_kernel, dim3(blocks), dim3(256), 0, stream, grad, (long)n, state);
  hipLaunchKernelGGL(grad_guard_kernel, dim3(blocks), dim3(256), 0, stream, grad, (long)n, state, target_max);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
extern "C" int lafs_cnn_cast_pad_f16(const float* src, int rows, int cols, void* dst, int ld, const float* scale, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld >= cols, "bad operand");
  hipLaunchKernelGGL(cast_pad_f16_kernel, dim3(blocks_for((long)rows * ld)), dim3(256), 0, stream, src, rows, cols, (bf16_t*)dst, ld, scale);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cnn_cast_f16_f32(const void* src, float* dst, int64_t n, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(src && dst && n > 0, "bad operand");
  hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(blocks_for((long)n)), dim3(256), 0, stream, (const bf16_t*)src, dst, (long)n);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_landmark_theta_bwd(const float* t, const float* dtheta, int B, int n, float* dt, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(t && dtheta && dt && B > 0 && n > 1, "bad operand");
  hipLaunchKernelGGL(theta_bwd_kernel, dim3((unsigned)B), dim3(256), 0, stream, t, dtheta, n, dt);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
