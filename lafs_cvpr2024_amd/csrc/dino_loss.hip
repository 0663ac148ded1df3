// Fused DINO student/teacher cross-entropy: sharpen + center + softmax + CE forward AND backward in two
// streaming passes over the logits (HBM-bound).  Replaces the 18 log_softmax + 18 mul-sum passes of the
// reference's double loop (lafs_train.py:643-667) and its center update (:669-679).
//
//   pass 1 (row_stats):  per logits row, online max / log-sum-exp of s/tau_s resp. (t-c)/tau_t
//   pass 2 (loss_grad):  per (sample b, class chunk): q0,q1 once, then for the ncrops student rows of that sample
//                        p = exp(s/tau_s - lse), grad = coef*(n_v p - sum_{i!=v} q_i), partial <sum q_i, s/tau_s>
//   pass 3 (finalize):   loss = 1/(n_terms B) * sum_{v,b} (n_v lse_vb - dot_vb)
#include "common.hpp"
#include "lafs_hip.h"

namespace {

constexpr int CHUNK = 1024;      // classes per workgroup in pass 2 (256 threads x float4)
constexpr int MAXC = 16;

__device__ __forceinline__ void online(float& m, float& s, float v) {
  if (v > m) { s = s * __expf(m - v) + 1.0f; m = v; } else { s += __expf(v - m); }
}

// stats[row] = {max, lse} of (x[row,:] - center) * inv_temp   (center may be NULL)
// Four float4 per thread and trip (independent loads), one running (max, sum) update per 16 values in the log2 domain: one
// v_exp per value plus one per trip (the per-value online update it replaces took a compare and up to two exponentials).
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, int ld, const float* __restrict__ center,
                                                       float inv_temp, const float* __restrict__ dtemp, int K,
                                                       float* __restrict__ stats) {
  __shared__ float sm[4], ss[4];
  if (dtemp != nullptr) inv_temp = 1.0f / dtemp[0];
  const float sc = inv_temp * 1.4426950408889634f;    // natural -> log2 domain
  const int row = blockIdx.x;
  const float* xr = x + (size_t)row * ld;
  float m = -INFINITY, s = 0.f;
  const int K4 = K >> 2;
  for (int i = threadIdx.x; i < K4; i += 1024) {
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ii = i + 256 * j;
      if (ii < K4) {
        v[j] = reinterpret_cast<const float4*>(xr)[ii];
        if (center != nullptr) {
          const float4 c = reinterpret_cast<const float4*>(center)[ii];
          v[j].x -= c.x; v[j].y -= c.y; v[j].z -= c.z; v[j].w -= c.w;
        }
        v[j].x *= sc; v[j].y *= sc; v[j].z *= sc; v[j].w *= sc;
      } else {
        v[j] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
    }
    float bm = fmaxf(fmaxf(v[0].x, v[0].y), fmaxf(v[0].z, v[0].w));        // (j = 0 is always in range: bm is finite)
#pragma unroll
    for (int j = 1; j < 4; ++j) bm = fmaxf(bm, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
    const float mn = fmaxf(m, bm);
    float add = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      add += __builtin_amdgcn_exp2f(v[j].x - mn) + __builtin_amdgcn_exp2f(v[j].y - mn) + __builtin_amdgcn_exp2f(v[j].z - mn) +
             __builtin_amdgcn_exp2f(v[j].w - mn);
    s = s * __builtin_amdgcn_exp2f(m - mn) + add;
    m = mn;
  }
  for (int k = (K4 << 2) + threadIdx.x; k < K; k += 256) {
    float v = xr[k];
    if (center != nullptr) v -= center[k];
    v *= sc;
    const float mn = fmaxf(m, v);
    s = s * __builtin_amdgcn_exp2f(m - mn) + __builtin_amdgcn_exp2f(v - mn);
    m = mn;
  }
  // combine (m, s) pairs: wave, then block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float mo = __shfl_xor(m, o, 64), so = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, mo);
    s = (mn == -INFINITY) ? 0.f : s * __builtin_amdgcn_exp2f(m - mn) + so * __builtin_amdgcn_exp2f(mo - mn);
    m = mn;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sm[wave] = m; ss[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float S = 0.f;
    for (int w = 0; w < 4; ++w) if (sm[w] != -INFINITY) S += ss[w] * __builtin_amdgcn_exp2f(sm[w] - M);
    stats[2 * row] = M * 0.6931471805599453f;
    stats[2 * row + 1] = (M + __log2f(S)) * 0.6931471805599453f;
  }
}

template <bool GRAD_BF16>
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ student, const float* __restrict__ teacher, int ld,
                                                       const float* __restrict__ center, int ncrops, int B, int K, float its,
                                                       float itt, const float* __restrict__ s_stats, const float* __restrict__ t_stats,
                                                       void* __restrict__ grad, int ldg, float coef, float* __restrict__ dots,
                                                       const float* __restrict__ dtemps) {
  __shared__ float part[MAXC][4];
  if (dtemps != nullptr) { coef = coef / (its * dtemps[0]); its = 1.0f / dtemps[0]; itt = 1.0f / dtemps[1]; }
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int k = chunk * CHUNK + threadIdx.x * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool full = (k + 4 <= K);
  float q0[4] = {0.f, 0.f, 0.f, 0.f}, q1[4] = {0.f, 0.f, 0.f, 0.f};
  if (k < K) {
    const float l0 = t_stats[2 * b + 1], l1 = t_stats[2 * (B + b) + 1];
    const float* t0 = teacher + (size_t)b * ld + k;
    const float* t1 = teacher + (size_t)(B + b) * ld + k;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (full || k + e < K) {
        const float c = center[k + e];
        q0[e] = __expf((t0[e] - c) * itt - l0);
        q1[e] = __expf((t1[e] - c) * itt - l1);
      }
  }
  // the student pieces of this (sample, class chunk) are requested four crop rows at a time (unconditional loads, row index
  // clamped) before the exponentials of the first: the per-crop load -> exp -> store chain ran at 3.6 TB/s
  for (int v0 = 0; v0 < ncrops; v0 += 4) {
  float4 sv[4];
  float lses[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t row = (size_t)min(v0 + j, ncrops - 1) * B + b;
    lses[j] = s_stats[2 * row + 1];
    const float* sp = student + row * ld + min(k, max(K - 4, 0));  // (inside the row: ld >= 4; a ragged tail piece is re-read below)
    sv[j] = *reinterpret_cast<const float4*>(sp);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = v0 + j;
    if (v >= ncrops) break;
    float dot = 0.f;
    if (k < K) {
      const size_t row = (size_t)v * B + b;
      const float lse = lses[j];
      float spv[4] = {sv[j].x, sv[j].y, sv[j].z, sv[j].w};
      if (!full) {
        const float* sp = student + row * ld + k;
#pragma unroll
        for (int e = 0; e < 4; ++e) spv[e] = (k + e < K) ? sp[e] : 0.f;
      }
      const float nv = (v < 2) ? 1.f : 2.f;
      float gq[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gq[e] = 0.f;
        if (full || k + e < K) {
          const float s = spv[e] * its;
          const float qs = (v == 0) ? q1[e] : (v == 1) ? q0[e] : q0[e] + q1[e];
          dot += qs * s;
          gq[e] = coef * (nv * __expf(s - lse) - qs);
        }
      }
      if (GRAD_BF16) {
        bf16_t* gp = reinterpret_cast<bf16_t*>(grad) + row * ldg + k;
        if (full) *reinterpret_cast<uint2*>(gp) = make_uint2(pack_bf2(gq[0], gq[1]), pack_bf2(gq[2], gq[3]));
        else for (int e = 0; e < 4; ++e) if (k + e < K) gp[e] = f2bf(gq[e]);
      } else {
        float* gp = reinterpret_cast<float*>(grad) + row * ldg + k;
        if (full) *reinterpret_cast<float4*>(gp) = make_float4(gq[0], gq[1], gq[2], gq[3]);
        else for (int e = 0; e < 4; ++e) if (k + e < K) gp[e] = gq[e];
      }
    }
    dot = wave_sum(dot);
    if (lane == 0) part[v][wave] = dot;
  }
  }
  __syncthreads();
  if (threadIdx.x < ncrops)
    dots[((size_t)chunk * B + b) * ncrops + threadIdx.x] =
        part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
}

// One workgroup of 1024 threads; thread j owns one (sample b, crop v) in the order the partial dot products are stored
// ([chunk][b][v]: coalesced), fourteen chunks in flight per trip (the former one-(v, b)-per-thread loop over strided
// addresses was a chain of ~100 dependent-latency loads: 52 us on the step's critical path).
__global__ __launch_bounds__(1024) void loss_final_kernel(const float* __restrict__ s_stats, const float* __restrict__ dots, int ncrops,
                                                         int B, int nchunks, float* __restrict__ loss) {
  __shared__ float red[16];
  float acc = 0.f;
  const int n = ncrops * B;
  for (int j = threadIdx.x; j < n; j += 1024) {
    const int b = j / ncrops, v = j - b * ncrops;
    float d = 0.f;
    int c = 0;
    for (; c + 14 <= nchunks; c += 14) {
      float t[14];
#pragma unroll
      for (int u = 0; u < 14; ++u) t[u] = dots[(size_t)(c + u) * n + j];
#pragma unroll
      for (int u = 0; u < 14; ++u) d += t[u];
    }
    for (; c < nchunks; ++c) d += dots[(size_t)c * n + j];
    acc += ((v < 2) ? 1.f : 2.f) * s_stats[2 * (v * B + b) + 1] - d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    loss[0] = t / (float)((2 * ncrops - 2) * B);
  }
}

// out[k] = sum_r x[r, k]; thread owns one class (a wave reads 256 contiguous bytes of a row), eight rows in flight
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, int ld, int rows, int K, float* __restrict__ out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const float* xc = x + k;
  float a = 0.f;
  int r = 0;
  for (; r + 8 <= rows; r += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = xc[(size_t)(r + j) * ld];
#pragma unroll
    for (int j = 0; j < 8; ++j) a += v[j];
  }
  for (; r < rows; ++r) a += xc[(size_t)r * ld];
  out[k] = a;
}

__global__ __launch_bounds__(256) void center_ema_kernel(float* __restrict__ center, const float* __restrict__ colsum, int K,
                                                        float inv_rows, float mom) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < K) center[k] = center[k] * mom + colsum[k] * inv_rows * (1.0f - mom);
}

}  // namespace

extern "C" int64_t lafs_dino_loss_workspace(int ncrops, int B, int K) {
  const int64_t nchunks = (K + CHUNK - 1) / CHUNK;
  return 2 * (int64_t)(ncrops * B + 2 * B) + nchunks * B * ncrops;
}

extern "C" int lafs_dino_loss_fwd_bwd(const float* student, const float* teacher, int ld, const float* center, int ncrops,
                                      int B, int K, float student_temp, float teacher_temp, float* loss_out,
                                      void* grad, int ldg, int grad_is_bf16, float grad_scale, float* workspace,
                                      const float* dev_temps, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(student && teacher && center && loss_out && grad && workspace, "null operand");
  LAFS_CHECK_ARG(ncrops >= 2 && ncrops <= MAXC && B > 0 && K > 0, "ncrops must be in 2..16");
  LAFS_CHECK_ARG(ld % 4 == 0 && ldg % 4 == 0 && ld >= K && ldg >= K, "row strides must be multiples of 4 and >= K");
  const int nchunks = ceil_div(K, CHUNK);
  float* s_stats = workspace;
  float* t_stats = s_stats + 2 * (size_t)ncrops * B;
  float* dots = t_stats + 2 * (size_t)2 * B;
  const float its = 1.0f / student_temp, itt = 1.0f / teacher_temp;
  hipLaunchKernelGGL(row_stats_kernel, dim3(ncrops * B), dim3(256), 0, stream, student, ld, (const float*)nullptr, its,
                     dev_temps, K, s_stats);
  hipLaunchKernelGGL(row_stats_kernel, dim3(2 * B), dim3(256), 0, stream, teacher, ld, center, itt,
                     dev_temps ? dev_temps + 1 : nullptr, K, t_stats);
  const float coef = grad_scale / ((float)(2 * ncrops - 2) * (float)B * student_temp);
  if (grad_is_bf16)
    hipLaunchKernelGGL(loss_grad_kernel<true>, dim3(nchunks, B), dim3(256), 0, stream, student, teacher, ld, center, ncrops, B, K, its,
                       itt, s_stats, t_stats, grad, ldg, coef, dots, dev_temps);
  else
    hipLaunchKernelGGL(loss_grad_kernel<false>, dim3(nchunks, B), dim3(256), 0, stream, student, teacher, ld, center, ncrops, B, K, its,
                       itt, s_stats, t_stats, grad, ldg, coef, dots, dev_temps);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1024), 0, stream, s_stats, dots, ncrops, B, nchunks, loss_out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_colsum_f32(const float* x, int ld, int rows, int K, float* out, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && out && rows > 0 && K > 0 && ld % 4 == 0, "bad operand");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3(ceil_div(K, 256)), dim3(256), 0, stream, x, ld, rows, K, out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_center_ema(float* center, const float* colsum, int K, float inv_rows_total, float momentum, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(center && colsum && K > 0, "bad operand");
  hipLaunchKernelGGL(center_ema_kernel, dim3(ceil_div(K, 256)), dim3(256), 0, stream, center, colsum, K, inv_rows_total, momentum);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
