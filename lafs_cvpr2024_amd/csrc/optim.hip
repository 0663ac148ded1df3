// Multi-tensor step glue over a flat fp32 parameter arena (HBM-bound, one launch each instead of the
// reference's ~150-iteration Python loops with a host sync per tensor):
//   per-tensor gradient L2 norms            utils.py:132-141 (clip_gradients: PER TENSOR, not global)
//   clip + cancel-last-layer + AdamW        utils.py:144-149, lafs_train.py:400,606 (torch.optim.AdamW defaults)
//   teacher EMA                             lafs_train.py:610-613
//   bf16 shadow refresh (GEMM operands)     -- build-specific
// The arena is cut into LAFS_CHUNK-element chunks; every tensor starts on a chunk boundary so a chunk belongs to
// exactly one tensor (`chunk_seg`).  Padding elements are zero in every buffer and stay zero.
#include "common.hpp"
#include "lafs_hip.h"

namespace {

// Per-tensor squared norms in two passes, no atomics and no pre-zeroed accumulator: one wave per chunk writes the chunk's sum
// of squares, then one workgroup per tensor adds its chunks in a fixed order (chunk_seg is ascending, so a tensor's chunks are
// the run [lower_bound(seg), lower_bound(seg+1))).  The norms -- and with them the clip coefficients and the whole update --
// are bitwise reproducible from the gradients.
__global__ __launch_bounds__(256) void chunk_sumsq_kernel(const float* __restrict__ grad, long n_chunks, float* __restrict__ chunk_sumsq) {
  const long c = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const float* g = grad + c * LAFS_CHUNK + (threadIdx.x & 63) * 4;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LAFS_CHUNK / 256; ++i) {
    const float4 v = *reinterpret_cast<const float4*>(g + i * 256);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) chunk_sumsq[c] = s;
}

__device__ __forceinline__ long first_chunk_of(const int* __restrict__ chunk_seg, long n_chunks, int seg) {
  long lo = 0, hi = n_chunks;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if (chunk_seg[mid] < seg) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void seg_sumsq_kernel(const float* __restrict__ chunk_sumsq, const int* __restrict__ chunk_seg,
                                                       long n_chunks, const float* __restrict__ hyper, float* __restrict__ seg_sumsq,
                                                       int seg0) {
  __shared__ float red[4];
  const int seg = blockIdx.x + seg0;
  const long c0 = first_chunk_of(chunk_seg, n_chunks, seg), c1 = first_chunk_of(chunk_seg, n_chunks, seg + 1);
  float s = 0.f;
  // (the head's last layer is one segment of ~25 k chunks: eight independent loads per trip instead of a chain of 100)
  long c = c0 + threadIdx.x;
  for (; c + 7 * 256 < c1; c += 8 * 256) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = chunk_sumsq[c + j * 256];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; c < c1; c += 256) s += chunk_sumsq[c];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float gs = hyper[LAFS_HP_GRAD_SCALE];
    seg_sumsq[seg] = (red[0] + red[1] + red[2] + red[3]) * gs * gs;
  }
}

__global__ __launch_bounds__(256) void seg_step_kernel(const int* __restrict__ seg_flags, int* __restrict__ seg_step, int seg0, int n_seg,
                                                      const float* __restrict__ hyper) {
  const int s = seg0 + blockIdx.x * 256 + threadIdx.x;
  if (s >= n_seg) return;
  const int f = seg_flags[s];
  const bool frozen = (f & LAFS_SEG_LAST_LAYER) && hyper[LAFS_HP_FREEZE_LAST] != 0.f;
  if ((f & LAFS_SEG_TRAINABLE) && !frozen) seg_step[s] += 1;
}

// The optimizer's state streams -- gradient, both moments, the teacher's fp32 copy and the master weights are each touched once per
// step, by this kernel, which runs on its own stream BESIDE the backward pass: non-temporal loads / stores (the moments and the
// teacher copy; the master weights are re-read by the transposed-shadow refresh, the bf16 shadows by the next forward: plain
// stores) keep 2 GB per step from displacing what the backward's kernels re-read from the L2.  LAFS_OPT_NT=0: plain accesses (lab).
#ifndef LAFS_OPT_NT
#define LAFS_OPT_NT 1
#endif
__device__ __forceinline__ float4 ldst4(const float* p) {
#if LAFS_OPT_NT
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void stst4(float* p, float4 x) {
#if LAFS_OPT_NT
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v v = {x.x, x.y, x.z, x.w};
  __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(p));
#else
  *reinterpret_cast<float4*>(p) = x;
#endif
}

__global__ __launch_bounds__(256) void clip_adamw_ema_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                            float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                            float* __restrict__ teacher, bf16_t* __restrict__ param_bf,
                                                            bf16_t* __restrict__ teacher_bf, const int* __restrict__ chunk_seg,
                                                            const int* __restrict__ seg_flags, const int* __restrict__ seg_step,
                                                            const float* __restrict__ seg_sumsq, const float* __restrict__ hyper,
                                                            long chunk0) {
  const size_t i = ((size_t)blockIdx.x + chunk0) * LAFS_CHUNK + threadIdx.x * 4;
  const int seg = chunk_seg[blockIdx.x + chunk0];
  const int flags = seg_flags[seg];
  const bool frozen = (flags & LAFS_SEG_LAST_LAYER) && hyper[LAFS_HP_FREEZE_LAST] != 0.f;
  const bool update = (flags & LAFS_SEG_TRAINABLE) && !frozen;
  float4 p = ldst4(param + i);
  if (update) {
    const float lr = hyper[LAFS_HP_LR];
    const float wd = (flags & LAFS_SEG_LOW_DECAY) ? hyper[LAFS_HP_WD_LOW] : ((flags & LAFS_SEG_DECAY) ? hyper[LAFS_HP_WD] : 0.f);
    const float b1 = hyper[LAFS_HP_BETA1], b2 = hyper[LAFS_HP_BETA2], eps = hyper[LAFS_HP_EPS], clip = hyper[LAFS_HP_CLIP];
    float gsc = hyper[LAFS_HP_GRAD_SCALE];
    if (clip > 0.f) {
      const float coef = clip / (sqrtf(seg_sumsq[seg]) + 1e-6f);
      if (coef < 1.f) gsc *= coef;
    }
    const float t = (float)seg_step[seg];
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    float4 g = ldst4(grad + i);
    float4 m = ldst4(exp_avg + i);
    float4 v = ldst4(exp_avg_sq + i);
    const float decay = 1.f - lr * wd;
#define LAFS_ADAM(c)                                              \
    {                                                             \
      const float gg = g.c * gsc;                                 \
      p.c *= decay;                                               \
      m.c = m.c * b1 + gg * (1.f - b1);                           \
      v.c = v.c * b2 + gg * gg * (1.f - b2);                      \
      p.c -= step_size * m.c / (sqrtf(v.c) * inv_sqrt_bc2 + eps); \
    }
    LAFS_ADAM(x) LAFS_ADAM(y) LAFS_ADAM(z) LAFS_ADAM(w)
#undef LAFS_ADAM
    *reinterpret_cast<float4*>(param + i) = p;
    stst4(exp_avg + i, m);
    stst4(exp_avg_sq + i, v);
    if (param_bf != nullptr) *reinterpret_cast<uint2*>(param_bf + i) = make_uint2(pack_bf2(p.x, p.y), pack_bf2(p.z, p.w));
  }
  if (teacher != nullptr) {                                 // EMA covers every parameter, trainable or not
    const float em = hyper[LAFS_HP_EMA_M];
    float4 tp = ldst4(teacher + i);
    tp.x = tp.x * em + (1.f - em) * p.x; tp.y = tp.y * em + (1.f - em) * p.y;
    tp.z = tp.z * em + (1.f - em) * p.z; tp.w = tp.w * em + (1.f - em) * p.w;
    stst4(teacher + i, tp);
    if (teacher_bf != nullptr) *reinterpret_cast<uint2*>(teacher_bf + i) = make_uint2(pack_bf2(tp.x, tp.y), pack_bf2(tp.z, tp.w));
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(src + i);
      *reinterpret_cast<uint2*>(dst + i) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    } else {
      for (size_t e = i; e < n; ++e) dst[e] = f2bf(src[e]);
    }
  }
}

__global__ __launch_bounds__(256) void cast_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const uint2 v = *reinterpret_cast<const uint2*>(src + i);
      *reinterpret_cast<float4*>(dst + i) = make_float4(bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y));
    } else {
      for (size_t e = i; e < n; ++e) dst[e] = bf2f(src[e]);
    }
  }
}

// dst[c, r] = bf16(src[r, c]); 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, int rows, int cols, bf16_t* __restrict__ dst,
                                                            int ld) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < cols && r < rows) dst[(size_t)c * ld + r] = f2bf(tile[tx][j]);
  }
}

// dst[c, r] = src[r, c], bf16 both sides; 64x64 tiles through LDS, 16-byte pieces on both sides when the shapes allow
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, int rows, int cols, int lds_, bf16_t* __restrict__ dst,
                                                            int ldd) {
  __shared__ bf16_t tile[64][66];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int j = i >> 6, k = i & 63, r = r0 + j, c = c0 + k;
    tile[j][k] = (r < rows && c < cols) ? src[(size_t)r * lds_ + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int j = i >> 6, k = i & 63, c = c0 + j, r = r0 + k;
    if (c < cols && r < rows) dst[(size_t)c * ldd + r] = tile[k][j];
  }
}

}  // namespace

extern "C" int lafs_transpose_bf16(const void* src, int rows, int cols, int ld_src, void* dst, int ld_dst, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "bad operand");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(ceil_div(cols, 64), ceil_div(rows, 64)), dim3(256), 0, stream, (const bf16_t*)src, rows, cols,
                     ld_src, (bf16_t*)dst, ld_dst);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

// Range forms: the tensors [seg_lo, seg_hi) = the chunks [chunk_lo, chunk_hi) of the arena (a tensor is a whole number of chunks).
// The training engine updates a range as soon as its gradients are final -- the DINO head while the trunk backward still runs,
// each run of blocks while the next one is computed -- so that these HBM-bound passes overlap GEMM-heavy work instead of
// forming a serial tail of the step.  Base pointers are those of the WHOLE arena.
extern "C" int lafs_grad_sumsq_range(const float* grad, const int32_t* chunk_seg, int64_t n_chunks, int n_seg, int64_t chunk_lo,
                                     int64_t chunk_hi, int seg_lo, int seg_hi, const float* hyper, float* chunk_sumsq, float* seg_sumsq,
                                     hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(grad && chunk_seg && hyper && chunk_sumsq && seg_sumsq && n_chunks > 0 && n_chunks < (1ll << 31) && n_seg > 0, "bad operand");
  LAFS_CHECK_ARG(0 <= chunk_lo && chunk_lo < chunk_hi && chunk_hi <= n_chunks && 0 <= seg_lo && seg_lo < seg_hi && seg_hi <= n_seg, "range outside the arena");
  const long nc = (long)(chunk_hi - chunk_lo);
  hipLaunchKernelGGL(chunk_sumsq_kernel, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, stream, grad + chunk_lo * LAFS_CHUNK, nc,
                     chunk_sumsq + chunk_lo);
  LAFS_LAUNCH_CHECK();
  hipLaunchKernelGGL(seg_sumsq_kernel, dim3((unsigned)(seg_hi - seg_lo)), dim3(256), 0, stream, chunk_sumsq, chunk_seg, (long)n_chunks, hyper,
                     seg_sumsq, seg_lo);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_grad_sumsq(const float* grad, const int32_t* chunk_seg, int64_t n_chunks, int n_seg, const float* hyper,
                               float* chunk_sumsq, float* seg_sumsq, hipStream_t stream) {
  return lafs_grad_sumsq_range(grad, chunk_seg, n_chunks, n_seg, 0, n_chunks, 0, n_seg, hyper, chunk_sumsq, seg_sumsq, stream);
}

extern "C" int lafs_clip_adamw_ema_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* teacher,
                                         void* param_bf16, void* teacher_bf16, const int32_t* chunk_seg, int64_t n_chunks, int64_t chunk_lo,
                                         int64_t chunk_hi, const int32_t* seg_flags, int32_t* seg_step, int n_seg, int seg_lo, int seg_hi,
                                         const float* seg_sumsq, const float* hyper, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && chunk_seg && seg_flags && seg_step && seg_sumsq && hyper, "null operand");
  LAFS_CHECK_ARG(n_chunks > 0 && n_chunks < (1ll << 31) && n_seg > 0, "bad sizes");
  LAFS_CHECK_ARG(0 <= chunk_lo && chunk_lo < chunk_hi && chunk_hi <= n_chunks && 0 <= seg_lo && seg_lo < seg_hi && seg_hi <= n_seg, "range outside the arena");
  hipLaunchKernelGGL(seg_step_kernel, dim3(ceil_div(seg_hi - seg_lo, 256)), dim3(256), 0, stream, seg_flags, seg_step, seg_lo, seg_hi, hyper);
  LAFS_LAUNCH_CHECK();
  hipLaunchKernelGGL(clip_adamw_ema_kernel, dim3((unsigned)(chunk_hi - chunk_lo)), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq,
                     teacher, (bf16_t*)param_bf16, (bf16_t*)teacher_bf16, chunk_seg, seg_flags, seg_step, seg_sumsq, hyper, (long)chunk_lo);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_clip_adamw_ema(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* teacher,
                                   void* param_bf16, void* teacher_bf16, const int32_t* chunk_seg, int64_t n_chunks,
                                   const int32_t* seg_flags, int32_t* seg_step, int n_seg, const float* seg_sumsq,
                                   const float* hyper, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(n_chunks > 0 && n_seg > 0, "bad sizes");
  return lafs_clip_adamw_ema_range(param, grad, exp_avg, exp_avg_sq, teacher, param_bf16, teacher_bf16, chunk_seg, n_chunks, 0, n_chunks,
                                   seg_flags, seg_step, n_seg, 0, n_seg, seg_sumsq, hyper, stream);
}

extern "C" int lafs_cast_bf16(const float* src, void* dst, int64_t n, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(src && dst && n > 0, "bad operand");
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, (bf16_t*)dst, (size_t)n);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cast_f32(const void* src, float* dst, int64_t n, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(src && dst && n > 0, "bad operand");
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cast_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)src, dst, (size_t)n);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

// table[4*i + {0,1,2,3}] = {src offset (elements into master), rows, cols, dst offset (elements into shadow_t)};
// tile_start[i] = first 32x32-tile index of matrix i (prefix sum), tile_start[n] = total tiles.
// 64x64 tiles: a wave reads 256 contiguous bytes of four source rows and writes whole 128-byte lines of the transposed bf16
// image (thread = one destination row piece of 8 consecutive source rows); the 32x32 form wrote 64-byte pieces (2 TB/s).
__global__ __launch_bounds__(256) void transpose_cast_table_kernel(const float* __restrict__ master, bf16_t* __restrict__ shadow_t,
                                                                  const long* __restrict__ table, const int* __restrict__ tile_start,
                                                                  int n_mat) {
  __shared__ float tile[64][65];
  int lo = 0, hi = n_mat;                                  // binary search: which matrix owns this tile
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const long* e = table + 4 * lo;
  const int rows = (int)e[1], cols = (int)e[2];
  const float* src = master + e[0];
  bf16_t* dst = shadow_t + e[3];
  const int local = blockIdx.x - tile_start[lo], tcols = (cols + 63) >> 6;
  const int c0 = (local % tcols) * 64, r0 = (local / tcols) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (int j = ty; j < 64; j += 4) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  const int r8 = (threadIdx.x & 7) * 8;
  const bool vec = ((rows & 7) == 0) && ((e[3] & 7) == 0);       // 16-byte destination pieces
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cl = (threadIdx.x >> 3) + 32 * pass, c = c0 + cl, r = r0 + r8;
    if (c >= cols || r >= rows) continue;
    if (vec) {
      *reinterpret_cast<uint4*>(dst + (size_t)c * rows + r) =
          make_uint4(pack_bf2(tile[r8][cl], tile[r8 + 1][cl]), pack_bf2(tile[r8 + 2][cl], tile[r8 + 3][cl]),
                     pack_bf2(tile[r8 + 4][cl], tile[r8 + 5][cl]), pack_bf2(tile[r8 + 6][cl], tile[r8 + 7][cl]));
    } else {
      for (int u = 0; u < 8 && r + u < rows; ++u) dst[(size_t)c * rows + r + u] = f2bf(tile[r8 + u][cl]);
    }
  }
}

extern "C" int lafs_transpose_cast_table(const float* master, void* shadow_t, const int64_t* table, const int32_t* tile_start,
                                         int n_mat, int n_tiles, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(master && shadow_t && table && tile_start && n_mat > 0 && n_tiles > 0, "bad operand");
  hipLaunchKernelGGL(transpose_cast_table_kernel, dim3(n_tiles), dim3(256), 0, stream, master, (bf16_t*)shadow_t, (const long*)table,
                     tile_start, n_mat);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_transpose_cast_bf16(const float* src, int rows, int cols, void* dst, int ld_dst, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld_dst >= rows, "bad operand");
  hipLaunchKernelGGL(transpose_cast_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32)), dim3(256), 0, stream, src, rows, cols,
                     (bf16_t*)dst, ld_dst);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
