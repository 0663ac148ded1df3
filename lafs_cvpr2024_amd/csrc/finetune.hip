// Fine-tune step kernels (train_largescale.py): fused margin-softmax + soft-target cross-entropy over the class
// axis, batch mixup with the u8 -> [-1,1] normalisation folded in, and the landmark patch gather.
//   CosFace                         face_pre_pro/ViT_face.py:49-89   s*(cos - m*y), y one-hot OR dense soft label
//   SoftTargetCrossEntropy (timm)   train_largescale.py:602,820
//   Mixup batch mode                util/mixup_my.py:189-200, 18-24  (target has <= 2 non-zeros per row, so it is
//                                   passed as (y1, y2, lam) and the dense [B,C] matrix is never materialised)
//   extract_patches_pytorch_gridsample   face_pre_pro/ViT_face.py:1615-1656 (n sequential grid_sample launches -> 1)
#include "common.hpp"
#include "lafs_hip.h"

namespace {

__device__ __forceinline__ float margin_logit(float c, float y, float s, float m, int type) {
  if (type == 0) return s * (c - m * y);
  if (y > 0.f) {                                      // ArcFace, hard label
    const float cc = fminf(fmaxf(c, -1.f), 1.f);
    return s * __cosf(acosf(cc) + m);
  }
  return s * c;
}
__device__ __forceinline__ float margin_dlogit(float c, float y, float s, float m, int type) {
  if (type == 0 || y <= 0.f) return s;
  const float cc = fminf(fmaxf(c, -1.f + 1e-7f), 1.f - 1e-7f);
  const float th = acosf(cc);
  return s * __sinf(th + m) / __sinf(th);
}

// one workgroup per sample row
__global__ __launch_bounds__(256) void margin_ce_kernel(float* __restrict__ cosv, int ld, int C, const int* __restrict__ y1,
                                                       const int* __restrict__ y2, float lam, float s, float m, int type,
                                                       float gscale, float* __restrict__ row_loss) {
  __shared__ float sm[4], ss[4];
  __shared__ float bc[2];
  const int b = blockIdx.x;
  float* row = cosv + (size_t)b * ld;
  const int a1 = y1[b], a2 = y2[b];
  auto label = [&](int k) { return ((k == a1) ? lam : 0.f) + ((k == a2) ? (1.f - lam) : 0.f); };
  float mx = -INFINITY, sum = 0.f;
  for (int k = threadIdx.x; k < C; k += 256) {
    const float z = margin_logit(row[k], label(k), s, m, type);
    if (z > mx) { sum = sum * __expf(mx - z) + 1.f; mx = z; } else { sum += __expf(z - mx); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float mo = __shfl_xor(mx, o, 64), so = __shfl_xor(sum, o, 64);
    const float mn = fmaxf(mx, mo);
    sum = (mn == -INFINITY) ? 0.f : sum * __expf(mx - mn) + so * __expf(mo - mn);
    mx = mn;
  }
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = mx; ss[threadIdx.x >> 6] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float S = 0.f;
    for (int w = 0; w < 4; ++w) if (sm[w] != -INFINITY) S += ss[w] * __expf(sm[w] - M);
    const float lse = M + __logf(S);
    bc[0] = lse;
    float dot = label(a1) * margin_logit(row[a1], label(a1), s, m, type);
    if (a2 != a1) dot += label(a2) * margin_logit(row[a2], label(a2), s, m, type);
    row_loss[b] = lse - dot;                               // sum_k y_k = 1
  }
  __syncthreads();
  const float lse = bc[0];
  for (int k = threadIdx.x; k < C; k += 256) {
    const float c = row[k], y = label(k);
    const float z = margin_logit(c, y, s, m, type);
    row[k] = gscale * (__expf(z - lse) - y) * margin_dlogit(c, y, s, m, type);
  }
}

// The same loss over (row, chunk) workgroups: one workgroup per row streams 824 KB twice at C = 205 990 and 128 rows leave half
// the chip idle (556 us, latency-bound); NCH chunks per row fill it.  Pass 1: per-chunk (max, sum exp) of the margin logits; pass 2:
// every workgroup folds its row's NCH partials, writes d loss / d cos of its chunk as bf16 (the operand format of the two
// gradient GEMMs: no fp32 round trip, no separate cast) and chunk 0 the row loss.  cos is read-only.  y2 == nullptr: the mixup
// partner of row b is row B-1-b (Mixup batch mode, util/mixup_my.py:189-200).  lam_dev != nullptr: lambda read from device memory.
constexpr int MCE_NCH = 16;
__device__ __forceinline__ void mce_labels(const int* y1, const int* y2, int b, int B, int& a1, int& a2) {
  a1 = y1[b];
  a2 = (y2 != nullptr) ? y2[b] : y1[B - 1 - b];
}
__global__ __launch_bounds__(256) void margin_ce_part_kernel(const float* __restrict__ cosv, int ld, int B, int C, const int* __restrict__ y1,
                                                            const int* __restrict__ y2, float lam, const float* __restrict__ lam_dev, float s,
                                                            float m, int type, float* __restrict__ part) {
  __shared__ float sm[4], ss[4];
  if (lam_dev != nullptr) lam = *lam_dev;
  const int b = blockIdx.y, ch = blockIdx.x;
  const float* row = cosv + (size_t)b * ld;
  int a1, a2;
  mce_labels(y1, y2, b, B, a1, a2);
  auto label = [&](int k) { return ((k == a1) ? lam : 0.f) + ((k == a2) ? (1.f - lam) : 0.f); };
  const int per = ((C + MCE_NCH - 1) / MCE_NCH + 3) & ~3, k0 = ch * per, k1 = min(C, k0 + per);
  float mx = -INFINITY, sum = 0.f;
  for (int k = k0 + threadIdx.x; k < k1; k += 256) {
    const float z = margin_logit(row[k], label(k), s, m, type);
    if (z > mx) { sum = sum * __expf(mx - z) + 1.f; mx = z; } else { sum += __expf(z - mx); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float mo = __shfl_xor(mx, o, 64), so = __shfl_xor(sum, o, 64);
    const float mn = fmaxf(mx, mo);
    sum = (mn == -INFINITY) ? 0.f : sum * __expf(mx - mn) + so * __expf(mo - mn);
    mx = mn;
  }
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = mx; ss[threadIdx.x >> 6] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float S = 0.f;
    for (int w = 0; w < 4; ++w) if (sm[w] != -INFINITY) S += ss[w] * __expf(sm[w] - M);
    part[((size_t)b * MCE_NCH + ch) * 2] = M;
    part[((size_t)b * MCE_NCH + ch) * 2 + 1] = S;
  }
}
__global__ __launch_bounds__(256) void margin_ce_grad_kernel(const float* __restrict__ cosv, int ld, int B, int C, const int* __restrict__ y1,
                                                            const int* __restrict__ y2, float lam, const float* __restrict__ lam_dev, float s,
                                                            float m, int type, float gscale, const float* __restrict__ part,
                                                            bf16_t* __restrict__ dcos, int lddc, float* __restrict__ row_loss) {
  if (lam_dev != nullptr) lam = *lam_dev;
  const int b = blockIdx.y, ch = blockIdx.x;
  const float* row = cosv + (size_t)b * ld;
  int a1, a2;
  mce_labels(y1, y2, b, B, a1, a2);
  auto label = [&](int k) { return ((k == a1) ? lam : 0.f) + ((k == a2) ? (1.f - lam) : 0.f); };
  float M = -INFINITY;
#pragma unroll
  for (int c = 0; c < MCE_NCH; ++c) M = fmaxf(M, part[((size_t)b * MCE_NCH + c) * 2]);
  float S = 0.f;
#pragma unroll
  for (int c = 0; c < MCE_NCH; ++c) {
    const float mc = part[((size_t)b * MCE_NCH + c) * 2];
    if (mc != -INFINITY) S += part[((size_t)b * MCE_NCH + c) * 2 + 1] * __expf(mc - M);
  }
  const float lse = M + __logf(S);
  if (ch == 0 && threadIdx.x == 0) {
    float dot = label(a1) * margin_logit(row[a1], label(a1), s, m, type);
    if (a2 != a1) dot += label(a2) * margin_logit(row[a2], label(a2), s, m, type);
    row_loss[b] = lse - dot;
  }
  const int per = ((C + MCE_NCH - 1) / MCE_NCH + 3) & ~3, k0 = ch * per, k1 = min(C, k0 + per);
  bf16_t* drow = dcos + (size_t)b * lddc;
  for (int k = k0 + 4 * threadIdx.x; k < k1; k += 1024) {           // 4 consecutive classes per thread: 16-byte loads, 8-byte stores
    if (k + 3 < k1) {
      const float4 c4 = *reinterpret_cast<const float4*>(row + k);
      const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
      float g[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = label(k + e);
        g[e] = gscale * (__expf(margin_logit(cv[e], y, s, m, type) - lse) - y) * margin_dlogit(cv[e], y, s, m, type);
      }
      *reinterpret_cast<uint2*>(drow + k) = make_uint2(pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]));
    } else {
      for (int e = 0; e < 4 && k + e < k1; ++e) {
        const float c = row[k + e], y = label(k + e);
        drow[k + e] = f2bf(gscale * (__expf(margin_logit(c, y, s, m, type) - lse) - y) * margin_dlogit(c, y, s, m, type));
      }
    }
  }
  // pad columns [C, lddc) of the gradient operand are zero (chunk NCH-1 owns them)
  if (ch == MCE_NCH - 1)
    for (int k = C + threadIdx.x; k < lddc; k += 256) drow[k] = 0;
}

__global__ __launch_bounds__(256) void cast_i64_i32_kernel(const long long* __restrict__ src, int* __restrict__ dst, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (int)src[i];
}

// patch-vector gradient [B, r*r, 192] -> image gradient [B, 3, 8r, 8r]: the inverse re-indexing of lafs_patchify
// (order 0: '(c p1 p2)' vectors of the conv patch embedding, 1: '(p1 p2 c)' of Part-fViT's Rearrange)
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ dp, int B, int r, int order, float* __restrict__ dimg) {
  const int S = 8 * r;
  const size_t total = (size_t)B * 3 * S * S;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % S), y = (int)((i / S) % S), c = (int)((i / ((size_t)S * S)) % 3), b = (int)(i / ((size_t)3 * S * S));
    const int pi = y >> 3, pj = x >> 3, u = y & 7, v = x & 7;
    const int e = order == 1 ? (u * 8 + v) * 3 + c : c * 64 + u * 8 + v;
    dimg[i] = dp[((size_t)b * r * r + pi * r + pj) * 192 + e];
  }
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int n, float scale, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += v[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}

__global__ __launch_bounds__(256) void mixup_norm_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int B, size_t per,
                                                        float lam, const float* __restrict__ lam_dev) {
  if (lam_dev != nullptr) lam = *lam_dev;                // a captured step reads this micro-step's lambda from device memory
  const size_t total = (size_t)B * per;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / per, r = i % per;
    const float x = (float)src[i] * (2.f / 255.f) - 1.f;
    const float xf = (float)src[(B - 1 - b) * per + r] * (2.f / 255.f) - 1.f;
    dst[i] = (lam == 1.f) ? x : x * lam + xf * (1.f - lam);
  }
}

// ---- landmark patch gather: one wave (8x8 lanes = one patch) per (image, landmark) ----
__device__ __forceinline__ float tap(const float* im, int S, int x, int y) {
  return (x >= 0 && x < S && y >= 0 && y < S) ? im[y * S + x] : 0.f;
}

__global__ __launch_bounds__(64) void gather_fwd_kernel(const float* __restrict__ img, const float* __restrict__ theta, int S, int n,
                                                       int r, float* __restrict__ out) {
  const int b = blockIdx.y, k = blockIdx.x;
  const int i = threadIdx.x >> 3, j = threadIdx.x & 7;          // output row i walks along x (transposed patch)
  const float px = theta[((size_t)b * n + k) * 2 + 0] + (float)(i - 4) - 0.5f;
  const float py = theta[((size_t)b * n + k) * 2 + 1] + (float)(j - 4) - 0.5f;
  const float fx0 = floorf(px), fy0 = floorf(py);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float fx = px - fx0, fy = py - fy0;
  const int R = r * 8;
  const int Y = (k / r) * 8 + i, X = (k % r) * 8 + j;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* im = img + ((size_t)b * 3 + c) * S * S;
    const float v = (1.f - fy) * ((1.f - fx) * tap(im, S, x0, y0) + fx * tap(im, S, x0 + 1, y0)) +
                    fy * ((1.f - fx) * tap(im, S, x0, y0 + 1) + fx * tap(im, S, x0 + 1, y0 + 1));
    out[(((size_t)b * 3 + c) * R + Y) * R + X] = v;
  }
}

__global__ __launch_bounds__(64) void gather_bwd_kernel(const float* __restrict__ img, const float* __restrict__ theta,
                                                       const float* __restrict__ dmos, int S, int n, int r,
                                                       float* __restrict__ dtheta, float* __restrict__ dimg) {
  const int b = blockIdx.y, k = blockIdx.x;
  const int i = threadIdx.x >> 3, j = threadIdx.x & 7;
  const float px = theta[((size_t)b * n + k) * 2 + 0] + (float)(i - 4) - 0.5f;
  const float py = theta[((size_t)b * n + k) * 2 + 1] + (float)(j - 4) - 0.5f;
  const float fx0 = floorf(px), fy0 = floorf(py);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float fx = px - fx0, fy = py - fy0;
  const int R = r * 8;
  const int Y = (k / r) * 8 + i, X = (k % r) * 8 + j;
  float gx = 0.f, gy = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* im = img + ((size_t)b * 3 + c) * S * S;
    const float d = dmos[(((size_t)b * 3 + c) * R + Y) * R + X];
    const float v00 = tap(im, S, x0, y0), v01 = tap(im, S, x0 + 1, y0), v10 = tap(im, S, x0, y0 + 1), v11 = tap(im, S, x0 + 1, y0 + 1);
    gx += d * ((1.f - fy) * (v01 - v00) + fy * (v11 - v10));
    gy += d * ((1.f - fx) * (v10 - v00) + fx * (v11 - v01));
    if (dimg != nullptr) {
      float* di = dimg + ((size_t)b * 3 + c) * S * S;
      auto add = [&](int x, int y, float w) { if (x >= 0 && x < S && y >= 0 && y < S) atomicAdd(di + y * S + x, d * w); };
      add(x0, y0, (1.f - fx) * (1.f - fy)); add(x0 + 1, y0, fx * (1.f - fy));
      add(x0, y0 + 1, (1.f - fx) * fy); add(x0 + 1, y0 + 1, fx * fy);
    }
  }
  gx = wave_sum(gx); gy = wave_sum(gy);
  if (threadIdx.x == 0) {
    dtheta[((size_t)b * n + k) * 2 + 0] = gx;
    dtheta[((size_t)b * n + k) * 2 + 1] = gy;
  }
}

// ---- class-sharded margin softmax (PartialFC): three row passes with the cross-rank statistics exchanged in between ----
// y[b] = local index of the row's target class inside this shard, or -1 when another rank owns it.  Soft (mixup) targets as the
// reference feeds them to its margin head (dense lam e_y1 + (1 - lam) e_y2 entering the margin itself, ViT_face.py:69-73): y2[b] =
// local index of the partner's class (or -1), lam[b] = the row's lambda; y2 == nullptr: hard labels.
struct ShardLabel {
  int t1, t2; float w1, w2;
  __device__ __forceinline__ float operator()(int k) const { return (k == t1 ? w1 : 0.f) + (k == t2 ? w2 : 0.f); }
};
__device__ __forceinline__ ShardLabel shard_label(const int* y, const int* y2, const float* lam, int b) {
  ShardLabel L;
  L.t1 = y[b]; L.t2 = -1; L.w1 = 1.f; L.w2 = 0.f;
  if (y2 != nullptr) {
    const float l = lam != nullptr ? lam[b] : 1.f;
    L.w1 = l; L.w2 = 1.f - l; L.t2 = y2[b];
    if (L.t2 == L.t1) { L.w1 = (L.t1 >= 0) ? 1.f : 0.f; L.t2 = -1; L.w2 = 0.f; }      // both labels on the same class
  }
  return L;
}
__global__ __launch_bounds__(256) void shard_rowmax_kernel(const float* __restrict__ cosv, int ld, int S, const int* __restrict__ y,
                                                          const int* __restrict__ y2, const float* __restrict__ lam,
                                                          float s, float m, int type, float* __restrict__ rowmax) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* row = cosv + (size_t)b * ld;
  const ShardLabel L = shard_label(y, y2, lam, b);
  float mx = -INFINITY;
  for (int k = threadIdx.x; k < S; k += 256) mx = fmaxf(mx, margin_logit(row[k], L(k), s, m, type));
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) rowmax[b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// rowsum[b] = sum_k exp(z_k - gmax[b]);  tgt[b] = sum_k y_k z_k over the targets this rank owns (0 when they live elsewhere)
__global__ __launch_bounds__(256) void shard_rowsum_kernel(const float* __restrict__ cosv, int ld, int S, const int* __restrict__ y,
                                                          const int* __restrict__ y2, const float* __restrict__ lam,
                                                          float s, float m, int type, const float* __restrict__ gmax,
                                                          float* __restrict__ rowsum, float* __restrict__ tgt) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* row = cosv + (size_t)b * ld;
  const ShardLabel L = shard_label(y, y2, lam, b);
  const float g = gmax[b];
  float acc = 0.f;
  for (int k = threadIdx.x; k < S; k += 256) acc += __expf(margin_logit(row[k], L(k), s, m, type) - g);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    rowsum[b] = red[0] + red[1] + red[2] + red[3];
    float t = 0.f;
    if (L.t1 >= 0) t += L.w1 * margin_logit(row[L.t1], L(L.t1), s, m, type);
    if (L.t2 >= 0) t += L.w2 * margin_logit(row[L.t2], L(L.t2), s, m, type);
    tgt[b] = t;
  }
}

// in place: cos -> dL/dcos = gscale * (exp(z - gmax)/Z - y) * dz/dcos
__global__ __launch_bounds__(256) void shard_grad_kernel(float* __restrict__ cosv, int ld, int S, const int* __restrict__ y,
                                                        const int* __restrict__ y2, const float* __restrict__ lam, float s,
                                                        float m, int type, const float* __restrict__ gmax,
                                                        const float* __restrict__ Z, float gscale) {
  const int b = blockIdx.x;
  float* row = cosv + (size_t)b * ld;
  const ShardLabel L = shard_label(y, y2, lam, b);
  const float g = gmax[b], iz = 1.0f / Z[b];
  for (int k = threadIdx.x; k < S; k += 256) {
    const float c = row[k], yk = L(k);
    const float z = margin_logit(c, yk, s, m, type);
    row[k] = gscale * (__expf(z - g) * iz - yk) * margin_dlogit(c, yk, s, m, type);
  }
}

// ---- landmark post-processing: raw regressor output -> pixel landmarks (face_pre_pro/ViT_face.py:1347-1378) ----
// one workgroup per image: per-sample min-max to [0,111], + noise_scale * noise, optional selection (with replacement)
__global__ __launch_bounds__(256) void landmark_theta_kernel(const float* __restrict__ t, int n_full, const float* __restrict__ noise,
                                                             float noise_scale, const int* __restrict__ sel, int n_out,
                                                             float* __restrict__ theta) {
  __shared__ float rmin[4], rmax[4];
  const int b = blockIdx.x, L = 2 * n_full;
  const float* row = t + (size_t)b * L;
  float mn = INFINITY, mx = -INFINITY;
  for (int k = threadIdx.x; k < L; k += 256) { const float v = row[k]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
  mx = wave_max(mx); mn = -wave_max(-mn);
  if ((threadIdx.x & 63) == 0) { rmin[threadIdx.x >> 6] = mn; rmax[threadIdx.x >> 6] = mx; }
  __syncthreads();
  mn = fminf(fminf(rmin[0], rmin[1]), fminf(rmin[2], rmin[3]));
  mx = fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
  const float range = mx - mn;
  for (int k = threadIdx.x; k < 2 * n_out; k += 256) {
    const int lm = sel ? sel[(size_t)b * n_out + (k >> 1)] : (k >> 1);
    const int src = 2 * lm + (k & 1);
    float v = (row[src] - mn) / range * 111.0f;
    if (noise) v += noise_scale * noise[(size_t)b * L + src];
    theta[(size_t)b * 2 * n_out + k] = v;
  }
}

int isqrt_exact(int n) {
  int r = 0;
  while ((r + 1) * (r + 1) <= n) ++r;
  return (r * r == n) ? r : -1;
}

}  // namespace

extern "C" int lafs_margin_softmax_ce(float* cos, int ld, int B, int C, const int32_t* y1, const int32_t* y2, float lam,
                                      float s, float m, int margin_type, float loss_scale, float* loss_out, float* row_ws,
                                      hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(cos && y1 && y2 && loss_out && row_ws && B > 0 && C > 0 && ld >= C, "bad operand");
  LAFS_CHECK_ARG(margin_type == 0 || margin_type == 1, "margin_type must be 0 (CosFace) or 1 (ArcFace)");
  hipLaunchKernelGGL(margin_ce_kernel, dim3(B), dim3(256), 0, stream, cos, ld, C, y1, y2, lam, s, m, margin_type, loss_scale / (float)B,
                     row_ws);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, row_ws, B, 1.0f / (float)B, loss_out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_margin_softmax_ce_bf16(const float* cos, int ld, int B, int C, const int32_t* y1, const int32_t* y2, float lam,
                                           const float* lam_dev, float s, float m, int margin_type, float loss_scale, void* dcos, int lddc,
                                           float* loss_out, float* row_ws, float* part_ws, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(cos && y1 && dcos && loss_out && row_ws && part_ws && B > 0 && C > 0 && ld >= C && lddc >= C, "bad operand");
  LAFS_CHECK_ARG(ld % 4 == 0 && lddc % 4 == 0, "row strides must be multiples of 4 elements");
  LAFS_CHECK_ARG(margin_type == 0 || margin_type == 1, "margin_type must be 0 (CosFace) or 1 (ArcFace)");
  hipLaunchKernelGGL(margin_ce_part_kernel, dim3(MCE_NCH, B), dim3(256), 0, stream, cos, ld, B, C, y1, y2, lam, lam_dev, s, m, margin_type, part_ws);
  LAFS_LAUNCH_CHECK();
  hipLaunchKernelGGL(margin_ce_grad_kernel, dim3(MCE_NCH, B), dim3(256), 0, stream, cos, ld, B, C, y1, y2, lam, lam_dev, s, m, margin_type,
                     loss_scale / (float)B, part_ws, (bf16_t*)dcos, lddc, row_ws);
  LAFS_LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, row_ws, B, 1.0f / (float)B, loss_out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_cast_i64_i32(const int64_t* src, int32_t* dst, int n, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(src && dst && n > 0, "bad operand");
  hipLaunchKernelGGL(cast_i64_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const long long*)src, dst, n);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_unpatchify_f32(const float* dpatch, int B, int S, int order, float* dimg, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(dpatch && dimg && B > 0 && S > 0 && S % 8 == 0 && (order == 0 || order == 1), "bad operand");
  size_t blocks = ((size_t)B * 3 * S * S + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dpatch, B, S / 8, order, dimg);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_mixup_normalize(const uint8_t* src_u8, float* dst, int B, int S, float lam, const float* lam_dev, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(src_u8 && dst && B > 0 && S > 0, "bad operand");
  const size_t per = (size_t)3 * S * S;
  size_t blocks = ((size_t)B * per + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mixup_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src_u8, dst, B, per, lam, lam_dev);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_patch_gather_fwd(const float* img, const float* theta, int B, int S, int n, float* mosaic, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  const int r = isqrt_exact(n);
  LAFS_CHECK_ARG(img && theta && mosaic && B > 0 && S > 0 && r > 0, "n must be a perfect square");
  hipLaunchKernelGGL(gather_fwd_kernel, dim3(n, B), dim3(64), 0, stream, img, theta, S, n, r, mosaic);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_patch_gather_bwd(const float* img, const float* theta, const float* dmosaic, int B, int S, int n,
                                     float* dtheta, float* dimg, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  const int r = isqrt_exact(n);
  LAFS_CHECK_ARG(img && theta && dmosaic && dtheta && B > 0 && S > 0 && r > 0, "n must be a perfect square");
  hipLaunchKernelGGL(gather_bwd_kernel, dim3(n, B), dim3(64), 0, stream, img, theta, dmosaic, S, n, r, dtheta, dimg);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_shard_margin_rowmax(const float* cos, int ld, int B, int S, const int32_t* y_local, const int32_t* y2_local,
                                        const float* lam, float s, float m, int margin_type, float* rowmax, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(cos && y_local && rowmax && B > 0 && S > 0 && ld >= S, "bad operand");
  LAFS_CHECK_ARG(y2_local == nullptr || margin_type == 0, "soft (mixup) targets: CosFace margin only (ArcFace takes hard labels)");
  hipLaunchKernelGGL(shard_rowmax_kernel, dim3(B), dim3(256), 0, stream, cos, ld, S, y_local, y2_local, lam, s, m, margin_type, rowmax);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_shard_margin_rowsum(const float* cos, int ld, int B, int S, const int32_t* y_local, const int32_t* y2_local,
                                        const float* lam, float s, float m, int margin_type, const float* gmax, float* rowsum,
                                        float* target_logit, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(cos && y_local && gmax && rowsum && target_logit && B > 0 && S > 0 && ld >= S, "bad operand");
  LAFS_CHECK_ARG(y2_local == nullptr || margin_type == 0, "soft (mixup) targets: CosFace margin only (ArcFace takes hard labels)");
  hipLaunchKernelGGL(shard_rowsum_kernel, dim3(B), dim3(256), 0, stream, cos, ld, S, y_local, y2_local, lam, s, m, margin_type, gmax,
                     rowsum, target_logit);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_shard_margin_grad(float* cos, int ld, int B, int S, const int32_t* y_local, const int32_t* y2_local,
                                      const float* lam, float s, float m, int margin_type, const float* gmax, const float* Z,
                                      float grad_scale, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(cos && y_local && gmax && Z && B > 0 && S > 0 && ld >= S, "bad operand");
  LAFS_CHECK_ARG(y2_local == nullptr || margin_type == 0, "soft (mixup) targets: CosFace margin only (ArcFace takes hard labels)");
  hipLaunchKernelGGL(shard_grad_kernel, dim3(B), dim3(256), 0, stream, cos, ld, S, y_local, y2_local, lam, s, m, margin_type, gmax, Z,
                     grad_scale);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_landmark_theta(const float* t, int B, int n_full, const float* noise, float noise_scale, const int32_t* sel,
                                   int n_out, float* theta, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(t && theta && B > 0 && n_full > 0 && n_out > 0, "bad operand");
  LAFS_CHECK_ARG(sel != nullptr || n_out <= n_full, "n_out > n_full needs a selection");
  hipLaunchKernelGGL(landmark_theta_kernel, dim3(B), dim3(256), 0, stream, t, n_full, noise, noise_scale, sel, n_out, theta);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
