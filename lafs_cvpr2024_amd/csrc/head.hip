// DINO head glue: row L2-normalisation and the weight-normalised last layer
// (vision_transformer.py:284-287 nn.utils.weight_norm, :299-300 F.normalize + last_layer).
// The three Linear+GELU layers and the [n,256]x[256,K] product run on the MFMA GEMM (gemm.hip).
#include "common.hpp"
#include "lafs_hip.h"

namespace {

// one wave per row, D <= 1024 (lane owns float4 at lane*4 + 256*i)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                        float* __restrict__ inv_norm, int rows, int D) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    v[i] = (c < D) ? *reinterpret_cast<const float4*>(x + (size_t)row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
  }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  if (lane == 0) inv_norm[row] = inv;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < D)
      *reinterpret_cast<uint2*>(y + (size_t)row * ldy + c) =
          make_uint2(pack_bf2(v[i].x * inv, v[i].y * inv), pack_bf2(v[i].z * inv, v[i].w * inv));
  }
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                        const float* __restrict__ inv_norm, float* __restrict__ dx, int lddx,
                                                        int rows, int D) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float inv = inv_norm[row];
  float4 y[4], d[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < D) {
      const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + c);
      d[i] = *reinterpret_cast<const float4*>(dy + (size_t)row * lddy + c);
      y[i] = make_float4(xv.x * inv, xv.y * inv, xv.z * inv, xv.w * inv);
      s += y[i].x * d[i].x + y[i].y * d[i].y + y[i].z * d[i].z + y[i].w * d[i].w;
    }
  }
  s = wave_sum(s);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < D)
      *reinterpret_cast<float4*>(dx + (size_t)row * lddx + c) =
          make_float4(inv * (d[i].x - y[i].x * s), inv * (d[i].y - y[i].y * s), inv * (d[i].z - y[i].z * s), inv * (d[i].w - y[i].w * s));
  }
}

// Workgroup = 64 weight rows.  Phase 1: one wave per row computes ||v||, writes w (bf16) and parks the scaled row in
// LDS; phase 2: thread d writes 64 consecutive k of w_t[d][:] (128 contiguous bytes).  D == 256 only when w_t != NULL.
__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ gsc, int K,
                                                            int Kpad, int D, bf16_t* __restrict__ w, bf16_t* __restrict__ wt,
                                                            int ldwt, float* __restrict__ inv_norm) {
  __shared__ bf16_t tile[64][258];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.x * 64;
  // four rows per wave and trip: their loads and their three reductions are independent (one row at a time left a single
  // 1 KiB request per wave in flight: 2.5 TB/s)
  for (int it = 0; it < 4; ++it) {
    float4 x[4][4];
    float s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wave + 4 * (4 * it + j);
      s[j] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane * 4 + 256 * i;
        x[j][i] = (k < K && c < D) ? *reinterpret_cast<const float4*>(v + (size_t)k * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s[j] += x[j][i].x * x[j][i].x + x[j][i].y * x[j][i].y + x[j][i].z * x[j][i].z + x[j][i].w * x[j][i].w;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = wave_sum(s[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rr = wave + 4 * (4 * it + j);
      const int k = k0 + rr;
      if (k >= Kpad) continue;
      float inv = 0.f, sc = 0.f;
      if (k < K) {
        inv = rsqrtf(s[j]);
        sc = gsc[k] * inv;
        if (lane == 0) inv_norm[k] = inv;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
          const uint2 pk = make_uint2(pack_bf2(x[j][i].x * sc, x[j][i].y * sc), pack_bf2(x[j][i].z * sc, x[j][i].w * sc));
          *reinterpret_cast<uint2*>(w + (size_t)k * D + c) = pk;
          if (wt != nullptr) {
            tile[rr][c] = (bf16_t)(pk.x & 0xffff); tile[rr][c + 1] = (bf16_t)(pk.x >> 16);
            tile[rr][c + 2] = (bf16_t)(pk.y & 0xffff); tile[rr][c + 3] = (bf16_t)(pk.y >> 16);
          }
        }
      }
    }
  }
  if (wt == nullptr) return;
  __syncthreads();
  const int d = threadIdx.x;
  if (d < D) {
    bf16_t* dst = wt + (size_t)d * ldwt + k0;
    const int n = min(64, Kpad - k0);
    for (int r = 0; r < n; r += 8) {
      uint32_t p[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) p[e] = (uint32_t)tile[r + 2 * e][d] | ((uint32_t)tile[r + 2 * e + 1][d] << 16);
      *reinterpret_cast<uint4*>(dst + r) = make_uint4(p[0], p[1], p[2], p[3]);
    }
  }
}

__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                            const float* __restrict__ gsc, const float* __restrict__ inv_norm,
                                                            int K, int D, float* __restrict__ dv, float* __restrict__ dg,
                                                            int accumulate) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const float inv = inv_norm[k], gk = gsc[k];
  float4 vh[4], d[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < D) {
      const float4 x = *reinterpret_cast<const float4*>(v + (size_t)k * D + c);
      d[i] = *reinterpret_cast<const float4*>(dw + (size_t)k * D + c);
      vh[i] = make_float4(x.x * inv, x.y * inv, x.z * inv, x.w * inv);
      s += vh[i].x * d[i].x + vh[i].y * d[i].y + vh[i].z * d[i].z + vh[i].w * d[i].w;
    }
  }
  s = wave_sum(s);
  if (dg != nullptr && lane == 0) dg[k] = (accumulate ? dg[k] : 0.f) + s;
  const float f = gk * inv;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < D) {
      float4 o = make_float4(f * (d[i].x - vh[i].x * s), f * (d[i].y - vh[i].y * s), f * (d[i].z - vh[i].z * s), f * (d[i].w - vh[i].w * s));
      float* p = dv + (size_t)k * D + c;
      if (accumulate) { const float4 old = *reinterpret_cast<const float4*>(p); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
      *reinterpret_cast<float4*>(p) = o;
    }
  }
}

}  // namespace

extern "C" int lafs_l2norm_fwd(const float* x, int ldx, void* y_bf16, int ldy, float* inv_norm, int rows, int D, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && y_bf16 && inv_norm && rows > 0 && D > 0 && D % 4 == 0 && D <= 1024, "D must be a multiple of 4 and <= 1024");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, x, ldx, (bf16_t*)y_bf16, ldy, inv_norm, rows, D);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_l2norm_bwd(const float* x, int ldx, const float* dy, int lddy, const float* inv_norm, float* dx, int lddx,
                               int rows, int D, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && dy && inv_norm && dx && rows > 0 && D > 0 && D % 4 == 0 && D <= 1024, "D must be a multiple of 4 and <= 1024");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, x, ldx, dy, lddy, inv_norm, dx, lddx, rows, D);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_weightnorm_fwd(const float* v, const float* g, int K, int Kpad, int D, void* w, void* w_t, int ldwt,
                                   float* inv_norm, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(v && g && w && inv_norm && K > 0 && Kpad >= K && D > 0 && D % 4 == 0 && D <= 1024, "bad operand");
  LAFS_CHECK_ARG(w_t == nullptr || (D <= 256 && Kpad % 8 == 0 && ldwt % 8 == 0 && ldwt >= Kpad), "w_t needs D <= 256 and 8-aligned Kpad/ldwt");
  hipLaunchKernelGGL(weightnorm_fwd_kernel, dim3(ceil_div(Kpad, 64)), dim3(256), 0, stream, v, g, K, Kpad, D, (bf16_t*)w, (bf16_t*)w_t,
                     ldwt, inv_norm);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_weightnorm_bwd(const float* dw, const float* v, const float* g, const float* inv_norm, int K, int D,
                                   float* dv, float* dg, int accumulate, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(dw && v && g && inv_norm && dv && K > 0 && D > 0 && D % 4 == 0 && D <= 1024, "bad operand");
  hipLaunchKernelGGL(weightnorm_bwd_kernel, dim3(ceil_div(K, 4)), dim3(256), 0, stream, dw, v, g, inv_norm, K, D, dv, dg, accumulate);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
