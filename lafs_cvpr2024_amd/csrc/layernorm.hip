// LayerNorm forward/backward over the fp32 residual stream, one wave64 per token row (HBM-bound).
// Replaces nn.LayerNorm at vision_transformer.py:99,103,156 (eps 1e-6) and face_pre_pro/ViT_face.py:117
// (eps 1e-5).  Forward emits the bf16 GEMM operand; backward fuses the residual-gradient accumulation, the
// DropPath-scaled bf16 cast that feeds the previous branch's GEMMs, and the gamma/beta reductions.
#include <algorithm>
#include <stdlib.h>
#include "common.hpp"
#ifndef LAFS_LN_NT
#define LAFS_LN_NT 1
#endif
#include "lafs_hip.h"

namespace {

// Loads of the LayerNorm backward's operands -- x, dy and the old value of the gradient stream are each read ONCE by this pass: with
// the non-temporal hint (global_load ... nt) they do not displace what the kernels running beside this one re-read from the L2 (the
// fused MLP's weight slices).  Same-box A/B of the step, 9 interleaved pairs: 14.45 against 14.55 ms with plain loads; alone, on
// operands a timing loop keeps cache-warm, the kernel is 10 % slower with it (tools/lab/NOTES.md).  LAFS_LN_NT=0: plain loads (lab).
__device__ __forceinline__ float4 ld_stream4(const float* p) {
#if LAFS_LN_NT
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ uint2 ld_stream2(const bf16_t* p) {
#if LAFS_LN_NT
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const u2v v = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(p));
  return make_uint2(v[0], v[1]);
#else
  return *reinterpret_cast<const uint2*>(p);
#endif
}

constexpr int MAXI = 8;   // D <= 2048: lane owns float4 at columns lane*4 + 256*i

template <int NI>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, bf16_t* __restrict__ y,
                                                    int ldy, float* __restrict__ yf, int ldyf,
                                                    float* __restrict__ stats, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  float4 v[NI];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane * 4 + 256 * i;
    v[i] = (c < D) ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < D) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + cc * cc + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < D) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(beta + c);
      const float o0 = (v[i].x - mean) * rstd * g.x + b.x, o1 = (v[i].y - mean) * rstd * g.y + b.y;
      const float o2 = (v[i].z - mean) * rstd * g.z + b.z, o3 = (v[i].w - mean) * rstd * g.w + b.w;
      if (y != nullptr)
        *reinterpret_cast<uint2*>(y + (size_t)row * ldy + c) = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
      if (yf != nullptr) *reinterpret_cast<float4*>(yf + (size_t)row * ldyf + c) = make_float4(o0, o1, o2, o3);
    }
  }
}

// D = 128 * NI, NI <= 4 (ViT-S: 384): TWO rows per wave (32 lanes per row, lane owns float4 at columns l*4 + 128*i -- no ragged
// last chunk as with 64 lanes on 384 columns), 1024 workgroups walk the rows with the next pair's operand requested a trip ahead.
// tools/lab/lab_ln.cpp, 25 216 x 384: 15.2 -> 12.7 us (3.8 -> 4.6 TB/s), 18 944 rows 11.3 -> 9.0 us.
template <int NI>
__global__ __launch_bounds__(256) void ln_fwd2_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, bf16_t* __restrict__ y,
                                                     int ldy, float* __restrict__ yf, int ldyf,
                                                     float* __restrict__ stats, int rows) {
  constexpr int D = NI * 128;
  const int lane = threadIdx.x & 63, l = lane & 31;
  float4 g4[NI], b4[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    g4[i] = *reinterpret_cast<const float4*>(gamma + l * 4 + 128 * i);
    b4[i] = *reinterpret_cast<const float4*>(beta + l * 4 + 128 * i);
  }
  auto half_sum = [](float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  const int stride = gridDim.x * 8;
  int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  float4 cur[NI], nxt[NI];
  auto load_row = [&](int r, float4 (&v)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + l * 4 + 128 * i);
  };
  if (row < rows) load_row(row, cur);
  for (; row < rows; row += stride) {
    const bool more = row + stride < rows;
    if (more) load_row(row + stride, nxt);
    float s = 0.f;                                     // (row arithmetic: common.hpp ln_sum4 / ln_sq4 / ln_out1, shared with mlp_fused.hip)
#pragma unroll
    for (int i = 0; i < NI; ++i) s = ln_sum4(s, cur[i]);
    const float mean = half_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) q = ln_sq4(q, cur[i], mean);
    const float rstd = rsqrtf(half_sum(q) / (float)D + eps);
    if (l == 0) *reinterpret_cast<float2*>(stats + 2 * (size_t)row) = make_float2(mean, rstd);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = l * 4 + 128 * i;
      const float o0 = ln_out1(cur[i].x, mean, rstd, g4[i].x, b4[i].x), o1 = ln_out1(cur[i].y, mean, rstd, g4[i].y, b4[i].y);
      const float o2 = ln_out1(cur[i].z, mean, rstd, g4[i].z, b4[i].z), o3 = ln_out1(cur[i].w, mean, rstd, g4[i].w, b4[i].w);
      if (y != nullptr) *reinterpret_cast<uint2*>(y + (size_t)row * ldy + c) = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
      if (yf != nullptr) *reinterpret_cast<float4*>(yf + (size_t)row * ldyf + c) = make_float4(o0, o1, o2, o3);
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < NI; ++i) cur[i] = nxt[i];
    }
  }
}

// Each wave walks rows (grid-stride) and keeps the gamma/beta partial sums for its columns in registers;
// they are combined across the 4 waves through LDS and leave the workgroup as one atomicAdd per column.
// NW waves per workgroup: the gamma/beta sums cost one atomic per column and WORKGROUP, and 1024 workgroups x 768 same-address
// atomics were ~15 us of the 75 us call at the ViT-S shape; 16-wave workgroups keep the 4096 waves in flight with a quarter of
// the atomics.
template <int NI, int NW, bool DYF>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const float* __restrict__ dyf,
                                                    int lddyf, const float* __restrict__ x, int ldx,
                                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                                    float* __restrict__ g_io, int ldg, int accumulate,
                                                    bf16_t* __restrict__ gb, int ldgb, const float* __restrict__ seq_scale,
                                                    const int* __restrict__ row2seq, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, float* __restrict__ part_out, int rows, int D,
                                                    DropCfg drop_in) {
  const DropCfg drop = drop_resolve(drop_in);
  __shared__ float red[NW][NI * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 gam[NI], ag[NI], ab[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane * 4 + 256 * i;
    gam[i] = (c < D) ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // One row per wave and trip; the NEXT row's operands (x, dy, the gradient stream it accumulates into, statistics) are requested
  // before this row's reductions, so a wave keeps two rows of loads in flight instead of one (the kernel is latency-bound on
  // its 16 waves per CU otherwise).
  struct RowIn { float4 xv[NI], df[DYF ? NI : 1], old[NI]; uint2 dw[DYF ? 1 : NI]; float mean, rstd, sc; };  // (dy stays packed until used)
  auto load_row = [&](int row, RowIn& r) {
    r.mean = stats[2 * row]; r.rstd = stats[2 * row + 1];
    r.sc = (seq_scale != nullptr) ? seq_scale[row2seq[row]] : 1.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < D) {
        r.xv[i] = ld_stream4(x + (size_t)row * ldx + c);      // (single-use streams: see ld_stream4)
        if constexpr (DYF) r.df[i] = ld_stream4(dyf + (size_t)row * lddyf + c);
        else r.dw[i] = ld_stream2(dy + (size_t)row * lddy + c);
        r.old[i] = accumulate ? ld_stream4(g_io + (size_t)row * ldg + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  const int stride = gridDim.x * NW;
  int row = blockIdx.x * NW + wave;
  RowIn cur;
  if (row < rows) load_row(row, cur);
  for (; row < rows; row += stride) {
    RowIn nxt;
    const bool more = row + stride < rows;              // (wave-uniform)
    if (more) load_row(row + stride, nxt);
    const float mean = cur.mean, rstd = cur.rstd;
    float4 xh[NI], d[NI];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < D) {
        const float4 xv = cur.xv[i];
        if constexpr (DYF) d[i] = cur.df[i];
        else d[i] = make_float4(bf_lo(cur.dw[i].x), bf_hi(cur.dw[i].x), bf_lo(cur.dw[i].y), bf_hi(cur.dw[i].y));
        xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
        ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
        d[i].x *= gam[i].x; d[i].y *= gam[i].y; d[i].z *= gam[i].z; d[i].w *= gam[i].w;
        s1 += d[i].x + d[i].y + d[i].z + d[i].w;
        s2 += d[i].x * xh[i].x + d[i].y * xh[i].y + d[i].z * xh[i].z + d[i].w * xh[i].w;
      } else {
        xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); d[i] = xh[i];
      }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
    const float sc = cur.sc;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < D) {
        float4 o = make_float4(rstd * (d[i].x - m1 - xh[i].x * m2), rstd * (d[i].y - m1 - xh[i].y * m2),
                               rstd * (d[i].z - m1 - xh[i].z * m2), rstd * (d[i].w - m1 - xh[i].w * m2));
        float* gp = g_io + (size_t)row * ldg + c;
        o.x += cur.old[i].x; o.y += cur.old[i].y; o.z += cur.old[i].z; o.w += cur.old[i].w;
        *reinterpret_cast<float4*>(gp) = o;
        if (gb != nullptr) {
          float4 q = make_float4(sc * o.x, sc * o.y, sc * o.z, sc * o.w);
          if (drop.thresh) {                              // gradient entering the dropped-out branch output
            const unsigned idx = (unsigned)row * (unsigned)D + (unsigned)c;
            q.x *= drop_mult(drop, idx); q.y *= drop_mult(drop, idx + 1); q.z *= drop_mult(drop, idx + 2); q.w *= drop_mult(drop, idx + 3);
          }
          *reinterpret_cast<uint2*>(gb + (size_t)row * ldgb + c) = make_uint2(pack_bf2(q.x, q.y), pack_bf2(q.z, q.w));
        }
      }
    }
    if (more) cur = nxt;
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {               // gamma sums, then beta sums, through the same LDS image
    if (pass) __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<float4*>(&red[wave][lane * 4 + 256 * i]) = pass ? ab[i] : ag[i];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += NW * 64) {
      float sg = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sg += red[w][c];
      // part_out: this workgroup's sums go to a slot of their own ([workgroup][gamma | beta][D]) and lafs_layernorm_bwd_fold adds
      // the slots in a fixed order -- run-to-run deterministic parameter gradients (one fp32 atomic per column and workgroup otherwise)
      if (part_out != nullptr) part_out[((size_t)blockIdx.x * 2 + pass) * D + c] = sg;
      else atomicAdd((pass ? dbeta : dgamma) + c, sg);
    }
  }
}

// The same pass with TWO rows per wave (D = 128 NI <= 512: 32 lanes x NI float4 cover a row exactly, where the kernel above leaves half
// of its last 256-column chunk idle at D = 384 -- a quarter of its load / store instructions carry 32 lanes) -- the backward
// counterpart of ln_fwd2_kernel.  Lane l of a half-wave holds columns 4 l + 128 i of its row; the two halves keep separate gamma /
// beta sums for the same columns, added once at the end.  Row reductions are 32-lane butterflies.
template <int NI, int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd2_kernel(const bf16_t* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     float* __restrict__ g_io, int ldg, int accumulate,
                                                     bf16_t* __restrict__ gb, int ldgb, const float* __restrict__ seq_scale,
                                                     const int* __restrict__ row2seq, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ part_out, int rows, DropCfg drop_in) {
  constexpr int D = NI * 128;
  const DropCfg drop = drop_resolve(drop_in);
  __shared__ float red[NW][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 31;
  float4 gam[NI], ag[NI], ab[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    gam[i] = *reinterpret_cast<const float4*>(gamma + l * 4 + 128 * i);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto half_sum = [](float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  struct RowIn { float4 xv[NI], old[NI]; uint2 dw[NI]; float mean, rstd, sc; };
  auto load_row = [&](int row, RowIn& r) {
    const float2 st = *reinterpret_cast<const float2*>(stats + 2 * (size_t)row);
    r.mean = st.x; r.rstd = st.y;
    r.sc = (seq_scale != nullptr) ? seq_scale[row2seq[row]] : 1.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = l * 4 + 128 * i;
r.xv[i] = ld_stream4(x + (size_t)row * ldx + c);        // (single-use streams: see ld_stream4)
      r.dw[i] = ld_stream2(dy + (size_t)row * lddy + c);
      r.old[i] = accumulate ? ld_stream4(g_io + (size_t)row * ldg + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  const int stride = gridDim.x * NW * 2;
  int row = (blockIdx.x * NW + wave) * 2 + (lane >> 5);
  RowIn cur;
  if (row < rows) load_row(row, cur);
  for (; row < rows; row += stride) {
    RowIn nxt;
    const bool more = row + stride < rows;              // (uniform per half-wave)
    if (more) load_row(row + stride, nxt);
    const float mean = cur.mean, rstd = cur.rstd;
    float4 xh[NI], d[NI];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float4 xv = cur.xv[i];
      d[i] = make_float4(bf_lo(cur.dw[i].x), bf_hi(cur.dw[i].x), bf_lo(cur.dw[i].y), bf_hi(cur.dw[i].y));
      xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
      ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
      ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
      d[i].x *= gam[i].x; d[i].y *= gam[i].y; d[i].z *= gam[i].z; d[i].w *= gam[i].w;
      s1 += d[i].x + d[i].y + d[i].z + d[i].w;
      s2 += d[i].x * xh[i].x + d[i].y * xh[i].y + d[i].z * xh[i].z + d[i].w * xh[i].w;
    }
    const float m1 = half_sum(s1) / (float)D, m2 = half_sum(s2) / (float)D;
    const float sc = cur.sc;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = l * 4 + 128 * i;
      float4 o = make_float4(rstd * (d[i].x - m1 - xh[i].x * m2), rstd * (d[i].y - m1 - xh[i].y * m2),
                             rstd * (d[i].z - m1 - xh[i].z * m2), rstd * (d[i].w - m1 - xh[i].w * m2));
      o.x += cur.old[i].x; o.y += cur.old[i].y; o.z += cur.old[i].z; o.w += cur.old[i].w;
      *reinterpret_cast<float4*>(g_io + (size_t)row * ldg + c) = o;
      if (gb != nullptr) {
        float4 q = make_float4(sc * o.x, sc * o.y, sc * o.z, sc * o.w);
        if (drop.thresh) {                              // gradient entering the dropped-out branch output
          const unsigned idx = (unsigned)row * (unsigned)D + (unsigned)c;
          q.x *= drop_mult(drop, idx); q.y *= drop_mult(drop, idx + 1); q.z *= drop_mult(drop, idx + 2); q.w *= drop_mult(drop, idx + 3);
        }
        *reinterpret_cast<uint2*>(gb + (size_t)row * ldgb + c) = make_uint2(pack_bf2(q.x, q.y), pack_bf2(q.z, q.w));
      }
    }
    if (more) cur = nxt;
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {               // gamma sums, then beta sums, through the same LDS image
    if (pass) __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float4 v = pass ? ab[i] : ag[i];                  // the wave's two rows-halves first
      v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64); v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
      if (lane < 32) *reinterpret_cast<float4*>(&red[wave][l * 4 + 128 * i]) = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += NW * 64) {
      float sg = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sg += red[w][c];
      if (part_out != nullptr) part_out[((size_t)blockIdx.x * 2 + pass) * D + c] = sg;     // (slots: see ln_bwd_kernel)
      else atomicAdd((pass ? dbeta : dgamma) + c, sg);
    }
  }
}

__global__ __launch_bounds__(256) void scale_cast_kernel(const float* __restrict__ g, int ldg, bf16_t* __restrict__ gb, int ldgb,
                                                        const float* __restrict__ seq_scale, const int* __restrict__ row2seq,
                                                        int rows, int D, DropCfg drop_in) {
  const DropCfg drop = drop_resolve(drop_in);
  const int per_row = D >> 2;
  const size_t total = (size_t)rows * per_row;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / per_row), c = (int)(i % per_row) * 4;
    const float sc = (seq_scale != nullptr) ? seq_scale[row2seq[row]] : 1.0f;
    float4 v = *reinterpret_cast<const float4*>(g + (size_t)row * ldg + c);
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    if (drop.thresh) {
      const unsigned idx = (unsigned)row * (unsigned)D + (unsigned)c;
      v.x *= drop_mult(drop, idx); v.y *= drop_mult(drop, idx + 1); v.z *= drop_mult(drop, idx + 2); v.w *= drop_mult(drop, idx + 3);
    }
    *reinterpret_cast<uint2*>(gb + (size_t)row * ldgb + c) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
  }
}

// x(f32)[r, c] *= mask(r, c)/(1-p) in place (embedding dropout and its backward); out_mask (optional) receives the factors
__global__ __launch_bounds__(256) void dropout_f32_kernel(float* __restrict__ x, int ldx, int rows, int D, DropCfg drop_in,
                                                         float* __restrict__ out_mask) {
  const DropCfg drop = drop_resolve(drop_in);
  const size_t total = (size_t)rows * D;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / D), c = (int)(i % D);
    const float f = drop.thresh ? drop_mult(drop, (unsigned)i) : 1.0f;
    if (x != nullptr) x[(size_t)row * ldx + c] *= f;
    if (out_mask != nullptr) out_mask[i] = f;
  }
}

// out[n] += sum_m X[m, n]; each workgroup owns a 64-column x 256-row slab (4 waves x 64 rows, 64 lanes = columns)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ X, int ldx, int M, int N,
                                                         float* __restrict__ out) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (n < N) {
    const int m0 = blockIdx.y * 256 + wave * 64;
    const int m1 = min(M, m0 + 64);
    for (int m = m0; m < m1; ++m) s += bf2f(X[(size_t)m * ldx + n]);
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && n < N) atomicAdd(out + n, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}


// dgamma[c] += sum over the partial slots of every listed launch (row chains), slot by slot in ascending order; dbeta likewise.
// One workgroup of 1024 threads per (item, 64-column group of the 2 D sums): wave w adds slots w, w + 16, ... of its 64 columns,
// the 16 wave sums are added in wave order through LDS -- the same order on every run.
struct LnFoldArgs { lafs_ln_fold_item it[LAFS_LN_FOLD_MAX]; int n_items, D, groups; };
__global__ __launch_bounds__(1024) void ln_fold_kernel(LnFoldArgs p) {
  __shared__ float red[16][64];
  const int item = blockIdx.x / p.groups, grp = blockIdx.x % p.groups;
  const lafs_ln_fold_item& it = p.it[item];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = grp * 64 + lane;                      // column of the [gamma | beta] image: 0 .. 2 D - 1
  float acc = 0.f;
  if (col < 2 * p.D) {
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const float* part = it.part[ch];
      const int n = it.n_parts[ch];
      if (part == nullptr) continue;
      for (int b = wave; b < n; b += 16) acc += part[(size_t)b * 2 * p.D + col];
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col < 2 * p.D) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][lane];
    float* dst = (col < p.D) ? it.dgamma + col : it.dbeta + (col - p.D);
    *dst += s;
  }
}

}  // namespace

#define LN_DISPATCH(NI_, KERNEL, ...)                                                             \
  switch (NI_) {                                                                                  \
    case 1: hipLaunchKernelGGL(KERNEL<1>, grid, dim3(256), 0, stream, __VA_ARGS__); break;        \
    case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(256), 0, stream, __VA_ARGS__); break;        \
    case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(256), 0, stream, __VA_ARGS__); break;        \
    case 4: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), 0, stream, __VA_ARGS__); break;        \
    default: hipLaunchKernelGGL(KERNEL<8>, grid, dim3(256), 0, stream, __VA_ARGS__); break;       \
  }

#define LN_BWD_DISPATCH(NI_, DYF_, ...)                                                                       \
  switch (NI_) {                                                                                              \
    case 1: hipLaunchKernelGGL((ln_bwd_kernel<1, 16, DYF_>), grid, dim3(1024), 0, stream, __VA_ARGS__); break; \
    case 2: hipLaunchKernelGGL((ln_bwd_kernel<2, 16, DYF_>), grid, dim3(1024), 0, stream, __VA_ARGS__); break; \
    case 3: hipLaunchKernelGGL((ln_bwd_kernel<3, 8, DYF_>), grid, dim3(512), 0, stream, __VA_ARGS__); break;  \
    case 4: hipLaunchKernelGGL((ln_bwd_kernel<4, 8, DYF_>), grid, dim3(512), 0, stream, __VA_ARGS__); break;  \
    default: hipLaunchKernelGGL((ln_bwd_kernel<8, 4, DYF_>), grid, dim3(256), 0, stream, __VA_ARGS__); break; \
  }

extern "C" int lafs_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                                  void* y_bf16, int ldy, float* y_f32, int ldyf, float* stats, int rows, int D,
                                  hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && gamma && beta && stats && (y_bf16 || y_f32), "null operand");
  LAFS_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && D <= 256 * MAXI, "D must be a multiple of 4 and <= 2048");
  LAFS_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldyf % 4 == 0, "row strides must be multiples of 4");
  if (D % 128 == 0 && D <= 512 && rows >= 4096) {      // two rows per wave, grid-stride (see ln_fwd2_kernel)
    const dim3 grid2(std::min(1024, ceil_div(rows, 8)));
    switch (D / 128) {
      case 1: hipLaunchKernelGGL(ln_fwd2_kernel<1>, grid2, dim3(256), 0, stream, x, ldx, gamma, beta, eps, (bf16_t*)y_bf16, ldy, y_f32, ldyf, stats, rows); break;
      case 2: hipLaunchKernelGGL(ln_fwd2_kernel<2>, grid2, dim3(256), 0, stream, x, ldx, gamma, beta, eps, (bf16_t*)y_bf16, ldy, y_f32, ldyf, stats, rows); break;
      case 3: hipLaunchKernelGGL(ln_fwd2_kernel<3>, grid2, dim3(256), 0, stream, x, ldx, gamma, beta, eps, (bf16_t*)y_bf16, ldy, y_f32, ldyf, stats, rows); break;
      default: hipLaunchKernelGGL(ln_fwd2_kernel<4>, grid2, dim3(256), 0, stream, x, ldx, gamma, beta, eps, (bf16_t*)y_bf16, ldy, y_f32, ldyf, stats, rows); break;
    }
    LAFS_LAUNCH_CHECK();
    return LAFS_OK;
  }
  const dim3 grid(ceil_div(rows, 4));
  const int ni = ceil_div(D, 256);
  LN_DISPATCH(ni, ln_fwd_kernel, x, ldx, gamma, beta, eps, (bf16_t*)y_bf16, ldy, y_f32, ldyf, stats, rows, D);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_layernorm_bwd_parts(int rows, int D) {
  if (rows <= 0 || D <= 0) return 0;
  const int ni = ceil_div(D, 256);
  const int nw = ni <= 2 ? 16 : (ni <= 4 ? 8 : 4);          // 32 KB of LDS per workgroup in every case
  int blocks = ceil_div(rows, nw);
  if (blocks > 4096 / nw) blocks = 4096 / nw;
  return blocks;
}

extern "C" int lafs_layernorm_bwd_fold(const lafs_ln_fold_item* items, int n_items, int D, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(items != nullptr && n_items > 0 && D > 0, "bad operand");
  for (int i0 = 0; i0 < n_items; i0 += LAFS_LN_FOLD_MAX) {
    LnFoldArgs a;
    a.n_items = std::min(LAFS_LN_FOLD_MAX, n_items - i0); a.D = D; a.groups = ceil_div(2 * D, 64);
    for (int i = 0; i < a.n_items; ++i) {
      a.it[i] = items[i0 + i];
      LAFS_CHECK_ARG(a.it[i].dgamma && a.it[i].dbeta, "null gradient");
    }
    hipLaunchKernelGGL(ln_fold_kernel, dim3(a.n_items * a.groups), dim3(1024), 0, stream, a);
  }
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_layernorm_bwd(const void* dy_bf16, int lddy, const float* dy_f32, int lddyf, const float* x, int ldx,
                                  const float* stats, const float* gamma, float* g_io, int ldg, int accumulate,
                                  void* gb_out, int ldgb, const float* seq_scale, const int32_t* row2seq,
                                  float* dgamma, float* dbeta, int rows, int D, float drop_p, uint32_t drop_seed,
                                  const float* drop_step, int drop_row0, float* part_out, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG((dy_bf16 || dy_f32) && x && stats && gamma && g_io && ((dgamma && dbeta) || part_out), "null operand");
  LAFS_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && D <= 256 * MAXI, "D must be a multiple of 4 and <= 2048");
  LAFS_CHECK_ARG(seq_scale == nullptr || row2seq != nullptr, "seq_scale needs row2seq");
  const int ni = ceil_div(D, 256);
  const dim3 grid(lafs_layernorm_bwd_parts(rows, D));
#ifndef LAFS_LAB_LN_BWD1
  if (dy_f32 == nullptr && D % 128 == 0 && D <= 512 && rows >= 4096) {      // two rows per wave (see ln_bwd2_kernel); same grid, same slots
    const DropCfg dc = make_drop(drop_p, drop_seed, drop_step, (unsigned)drop_row0 * (unsigned)D);
#define LN_BWD2(NI_, NW_)                                                                                                                  \
    hipLaunchKernelGGL((ln_bwd2_kernel<NI_, NW_>), grid, dim3(NW_ * 64), 0, stream, (const bf16_t*)dy_bf16, lddy, x, ldx, stats, gamma, g_io, \
                       ldg, accumulate, (bf16_t*)gb_out, ldgb, seq_scale, row2seq, dgamma, dbeta, part_out, rows, dc)
    switch (D / 128) {
      case 1: LN_BWD2(1, 8); break;
      case 2: LN_BWD2(2, 8); break;
      case 3: LN_BWD2(3, 8); break;
      default: LN_BWD2(4, 8); break;
    }
#undef LN_BWD2
    LAFS_LAUNCH_CHECK();
    return LAFS_OK;
  }
#endif
  if (dy_f32 != nullptr) {
    LN_BWD_DISPATCH(ni, true, (const bf16_t*)dy_bf16, lddy, dy_f32, lddyf, x, ldx, stats, gamma, g_io, ldg, accumulate,
                    (bf16_t*)gb_out, ldgb, seq_scale, row2seq, dgamma, dbeta, part_out, rows, D,
                    make_drop(drop_p, drop_seed, drop_step, (unsigned)drop_row0 * (unsigned)D));
  } else {
    LN_BWD_DISPATCH(ni, false, (const bf16_t*)dy_bf16, lddy, dy_f32, lddyf, x, ldx, stats, gamma, g_io, ldg, accumulate,
                    (bf16_t*)gb_out, ldgb, seq_scale, row2seq, dgamma, dbeta, part_out, rows, D,
                    make_drop(drop_p, drop_seed, drop_step, (unsigned)drop_row0 * (unsigned)D));
  }
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_scale_cast_bf16(const float* g, int ldg, void* gb, int ldgb, const float* seq_scale,
                                    const int32_t* row2seq, int rows, int D, float drop_p, uint32_t drop_seed,
                                    const float* drop_step, int drop_row0, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(g && gb && rows > 0 && D > 0 && D % 4 == 0, "bad operand");
  LAFS_CHECK_ARG(seq_scale == nullptr || row2seq != nullptr, "seq_scale needs row2seq");
  const size_t total = (size_t)rows * (D / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_cast_kernel, dim3(blocks), dim3(256), 0, stream, g, ldg, (bf16_t*)gb, ldgb, seq_scale, row2seq, rows, D,
                     make_drop(drop_p, drop_seed, drop_step, (unsigned)drop_row0 * (unsigned)D));
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_dropout_f32(float* x, int ldx, int rows, int D, float drop_p, uint32_t drop_seed, const float* drop_step,
                                hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && rows > 0 && D > 0 && drop_p >= 0.f && drop_p < 1.f && (long)rows * D < 4294967296L, "bad operand");
  if (!(drop_p > 0.f)) return LAFS_OK;
  const size_t total = (size_t)rows * D;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dropout_f32_kernel, dim3(blocks), dim3(256), 0, stream, x, ldx, rows, D, make_drop(drop_p, drop_seed, drop_step), nullptr);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_debug_dropout_mask(int rows, int cols, float drop_p, uint32_t drop_seed, float* out, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(out && rows > 0 && cols > 0 && drop_p >= 0.f && drop_p < 1.f && (long)rows * cols < 4294967296L, "bad operand");
  const size_t total = (size_t)rows * cols;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dropout_f32_kernel, dim3(blocks), dim3(256), 0, stream, nullptr, 0, rows, cols, make_drop(drop_p, drop_seed), out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_colsum_bf16_acc(const void* X, int ldx, int M, int N, float* out, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(X && out && M > 0 && N > 0, "bad operand");
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(ceil_div(N, 64), ceil_div(M, 256)), dim3(256), 0, stream,
                     (const bf16_t*)X, ldx, M, N, out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
