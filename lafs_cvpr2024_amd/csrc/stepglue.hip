// Small kernels that keep ATen / rocBLAS out of the captured LAFS step (round-1 profile: uniform_ + floor + div for the DropPath
// masks, Tensile GEMMs + cat + add for the position-table resampling, a 211 MB FillFunctor for the gradient memset).
//
//  lafs_droppath_scales : stochastic-depth scales of every (block, branch, sequence), counter-based (reference
//                         vision_transformer.py:27-35 / face_pre_pro/ViT_face.py:106-112: torch.rand -> floor(keep + u) / keep)
//  lafs_pos_interp_fwd/bwd : position table through the fixed bicubic matrix of interpolate_pos_encoding
//                         (vision_transformer.py:174-194), cls row copied; backward accumulates into the table's gradient
//  lafs_zero_chunks     : zero the gradient arena EXCEPT the tensors whose first writer overwrites them
#include "common.hpp"
#include "lafs_hip.h"

namespace {

// scales[(l * 2 + br) * n_seq + s] = keep_l > u ? 1 / keep_l : 0 with u = hash(seed, step, l, br, s) in [0, 1).
// `step` is read from device memory (hyper[LAFS_HP_STEP]) so that a replayed hipGraph draws fresh masks every step.
__global__ __launch_bounds__(256) void droppath_kernel(const float* __restrict__ keep, int depth, int n_seq, unsigned seed,
                                                      const float* __restrict__ step, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= depth * 2 * n_seq) return;
  const int l = i / (2 * n_seq);
  const unsigned st = step ? (unsigned)step[0] : 0u;
  const unsigned h = drop_mix32((unsigned)i, seed ^ (st * 0x9E3779B9u + 0x85EBCA6Bu));
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  const float k = keep[l];
  out[i] = (u < k) ? 1.0f / k : 0.f;                     // floor(keep + u') / keep with u' = 1 - u uniform: P(nonzero) = keep
}

// out[0, :] = pe[0, :]; out[1 + r, d] = sum_g M[r, g] * pe[1 + g, d]        (M: [R, G] row-major)
// One workgroup per (output row, 64-column slab): the row of M sits in LDS, the 4 waves split the G source rows and are folded
// through LDS -- 197 x 6 workgroups of short, independent load streams instead of 784 dependent iterations per thread.
__global__ __launch_bounds__(256) void pos_fwd_kernel(const float* __restrict__ pe, const float* __restrict__ M, float* __restrict__ out,
                                                     int R, int G, int D) {
  __shared__ float mrow[1024];
  __shared__ float part[4][64];
  const int r = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + lane;
  if (r == 0) {
    if (w == 0 && d < D) out[d] = pe[d];
    return;
  }
  const float* m = M + (size_t)(r - 1) * G;
  for (int g = threadIdx.x; g < G; g += 256) mrow[g] = m[g];
  __syncthreads();
  float acc = 0.f;
  if (d < D) {
    const int per = (G + 3) / 4, g0 = w * per, g1 = min(G, g0 + per);
#pragma unroll 8
    for (int g = g0; g < g1; ++g) acc += mrow[g] * pe[(size_t)(1 + g) * D + d];
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0 && d < D) out[(size_t)r * D + d] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}

// gpe[0, :] += dp[0, :]; gpe[1 + g, d] += sum_r M[r, g] * dp[1 + r, d]
__global__ __launch_bounds__(256) void pos_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ M, float* __restrict__ gpe,
                                                     int R, int G, int D) {
  const int d = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (d >= D) return;
  if (g == 0) { gpe[d] += dp[d]; return; }
  float acc = 0.f;
#pragma unroll 8
  for (int r = 0; r < R; ++r) acc += M[(size_t)r * G + (g - 1)] * dp[(size_t)(1 + r) * D + d];
  gpe[(size_t)g * D + d] += acc;
}

__global__ __launch_bounds__(256) void zero_chunks_kernel(float* __restrict__ buf, const int* __restrict__ chunk_seg,
                                                         const int* __restrict__ seg_flags, int skip_mask) {
  if (seg_flags[chunk_seg[blockIdx.x]] & skip_mask) return;
  reinterpret_cast<float4*>(buf + (size_t)blockIdx.x * LAFS_CHUNK)[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace

extern "C" int lafs_droppath_scales(const float* keep_prob, int depth, int n_seq, uint32_t seed, const float* step_dev, float* scales,
                                    hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(keep_prob && scales && depth > 0 && n_seq > 0, "bad argument");
  const int n = depth * 2 * n_seq;
  hipLaunchKernelGGL(droppath_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, keep_prob, depth, n_seq, seed, step_dev, scales);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_pos_interp_fwd(const float* pos_table, const float* interp, float* out, int R, int G, int D, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(pos_table && interp && out && R > 0 && G > 0 && D > 0, "bad argument");
  LAFS_CHECK_ARG(G <= 1024, "source grid larger than 32 x 32");
  hipLaunchKernelGGL(pos_fwd_kernel, dim3(ceil_div(D, 64), R + 1), dim3(256), 0, stream, pos_table, interp, out, R, G, D);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_pos_interp_bwd(const float* dpos, const float* interp, float* grad_table, int R, int G, int D, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(dpos && interp && grad_table && R > 0 && G > 0 && D > 0, "bad argument");
  hipLaunchKernelGGL(pos_bwd_kernel, dim3(ceil_div(D, 256), G + 1), dim3(256), 0, stream, dpos, interp, grad_table, R, G, D);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_zero_chunks(float* buf, const int32_t* chunk_seg, const int32_t* seg_flags, int64_t n_chunks, int skip_mask,
                                hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(buf && chunk_seg && seg_flags && n_chunks > 0, "bad argument");
  hipLaunchKernelGGL(zero_chunks_kernel, dim3((unsigned)n_chunks), dim3(256), 0, stream, buf, chunk_seg, seg_flags, skip_mask);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

// Zeroing of scratch buffers inside the captured step, as a plain kernel node (no at::native fill, no memset node).
namespace {
__global__ __launch_bounds__(256) void fill_zero_kernel(unsigned char* __restrict__ p, size_t bytes) {
  const size_t n16 = bytes / 16;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    reinterpret_cast<uint4*>(p)[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) p[n16 * 16 + threadIdx.x] = 0;
}
}  // namespace

extern "C" int lafs_fill_zero(void* buf, int64_t bytes, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(buf && bytes > 0 && ((uintptr_t)buf & 15) == 0, "buffer must be 16-byte aligned");
  const size_t n16 = (size_t)bytes / 16;
  size_t blocks = (n16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned char*)buf, (size_t)bytes);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
