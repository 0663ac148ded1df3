// Patch-embedding front end and token assembly (HBM-bound glue around the patch-embed GEMM).
//   vision_transformer.py:126-131 (Conv2d k=s=8 as a GEMM over patch vectors), :196-207 (cls cat + pos add)
//   face_pre_pro/ViT_face.py:760-766 (rearrange '(p1 p2 c)' + Linear + cls + pos)
// NCHW crops are read with coalesced 32-byte segments along the image row; patch vectors leave as bf16 rows
// that the MFMA GEMM consumes directly.
#include "common.hpp"
#include "lafs_hip.h"

namespace {

// one thread per (image b, patch row pr, in-patch row p1, patch col pc): 3 channels x 8 pixels
template <int ORDER>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int B, int S, bf16_t* __restrict__ out) {
  const int r = S >> 3;
  const int total = B * S * r;                           // B * (r*8) * r
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int pc = i % r;
  const int y = (i / r) % S;                             // pixel row
  const int b = i / (r * S);
  const int pr = y >> 3, p1 = y & 7;
  float v[3][8];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* src = img + (((size_t)b * 3 + c) * S + y) * S + pc * 8;
    const float4 a = reinterpret_cast<const float4*>(src)[0], d = reinterpret_cast<const float4*>(src)[1];
    v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w; v[c][4] = d.x; v[c][5] = d.y; v[c][6] = d.z; v[c][7] = d.w;
  }
  bf16_t* dst = out + ((size_t)b * r * r + pr * r + pc) * 192;
  if (ORDER == LAFS_PATCH_ORDER_CHW) {                    // k = c*64 + p1*8 + j
#pragma unroll
    for (int c = 0; c < 3; ++c)
      *reinterpret_cast<uint4*>(dst + c * 64 + p1 * 8) =
          make_uint4(pack_bf2(v[c][0], v[c][1]), pack_bf2(v[c][2], v[c][3]), pack_bf2(v[c][4], v[c][5]), pack_bf2(v[c][6], v[c][7]));
  } else {                                                // k = (p1*8 + j)*3 + c
    float f[24];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) f[j * 3 + c] = v[c][j];
    uint4* d4 = reinterpret_cast<uint4*>(dst + p1 * 24);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      d4[q] = make_uint4(pack_bf2(f[8 * q], f[8 * q + 1]), pack_bf2(f[8 * q + 2], f[8 * q + 3]),
                         pack_bf2(f[8 * q + 4], f[8 * q + 5]), pack_bf2(f[8 * q + 6], f[8 * q + 7]));
  }
}

__global__ __launch_bounds__(256) void embed_cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                       float* __restrict__ tok, int ldt, int n_seq, int npatch, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_seq * D) return;
  const int s = i / D, d = i % D;
  tok[(size_t)s * (npatch + 1) * ldt + d] = cls[d] + pos[d];
}

// grid (column-quad chunks, sequence chunks of EB_SEQ): thread = (token t, 4 channels); each workgroup sums its sequence
// chunk in registers and adds it to dpos with one atomic per element (n_seq/EB_SEQ adds per address).
constexpr int EB_SEQ = 16;
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ g, int ldg, int n_seq, int npatch, int D,
                                                       bf16_t* __restrict__ gp, float* __restrict__ dpos, float* __restrict__ dcls,
                                                       float* __restrict__ part) {
  const int per = D >> 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (npatch + 1) * per) return;
  const int t = i / per, c = (i % per) * 4;
  const int s0 = blockIdx.y * EB_SEQ, s1 = min(n_seq, s0 + EB_SEQ);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = s0; s < s1; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(g + ((size_t)s * (npatch + 1) + t) * ldg + c);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    if (t > 0 && gp != nullptr)
      *reinterpret_cast<uint2*>(gp + ((size_t)s * npatch + (t - 1)) * D + c) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
  }
  if (part != nullptr) {                               // deterministic form: this sequence chunk's sums go to a slot of their own
    *reinterpret_cast<float4*>(part + ((size_t)blockIdx.y * (npatch + 1) + t) * D + c) = acc;
    return;
  }
  float* dp = dpos + (size_t)t * D + c;
  atomicAdd(dp, acc.x); atomicAdd(dp + 1, acc.y); atomicAdd(dp + 2, acc.z); atomicAdd(dp + 3, acc.w);
  if (t == 0 && dcls != nullptr) {
    atomicAdd(dcls + c, acc.x); atomicAdd(dcls + c + 1, acc.y); atomicAdd(dcls + c + 2, acc.z); atomicAdd(dcls + c + 3, acc.w);
  }
}

// dpos[t, c] += sum over the chunk slots in ascending order (dcls likewise from the t = 0 rows): one writer per element, a fixed order
__global__ __launch_bounds__(256) void embed_bwd_fold_kernel(const float* __restrict__ part, int n_chunks, int npatch, int D,
                                                            float* __restrict__ dpos, float* __restrict__ dcls) {
  const int per = D >> 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (npatch + 1) * per) return;
  const int t = i / per, c = (i % per) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < n_chunks; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)k * (npatch + 1) + t) * D + c);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  float4* dp = reinterpret_cast<float4*>(dpos + (size_t)t * D + c);
  float4 o = *dp;
  o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
  *dp = o;
  if (t == 0 && dcls != nullptr) {
    float4* dc = reinterpret_cast<float4*>(dcls + c);
    float4 q = *dc;
    q.x += acc.x; q.y += acc.y; q.z += acc.z; q.w += acc.w;
    *dc = q;
  }
}

__global__ __launch_bounds__(256) void gather_cls_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ cu,
                                                        int n_seq, int D, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_seq * D) return;
  const int s = i / D, d = i % D;
  out[i] = x[(size_t)cu[s] * ldx + d];
}
__global__ __launch_bounds__(256) void scatter_cls_kernel(const float* __restrict__ src, const int* __restrict__ cu, int n_seq,
                                                         int D, float* __restrict__ g, int ldg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_seq * D) return;
  const int s = i / D, d = i % D;
  g[(size_t)cu[s] * ldg + d] = src[i];
}

}  // namespace

extern "C" int lafs_patchify(const float* img, int B, int S, int order, void* patches, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(img && patches && B > 0 && S > 0 && S % 8 == 0, "image side must be a multiple of the 8-pixel patch");
  const int total = B * S * (S / 8);
  if (order == LAFS_PATCH_ORDER_CHW)
    hipLaunchKernelGGL(patchify_kernel<LAFS_PATCH_ORDER_CHW>, dim3(ceil_div(total, 256)), dim3(256), 0, stream, img, B, S, (bf16_t*)patches);
  else
    hipLaunchKernelGGL(patchify_kernel<LAFS_PATCH_ORDER_HWC>, dim3(ceil_div(total, 256)), dim3(256), 0, stream, img, B, S, (bf16_t*)patches);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_embed_cls(const float* cls, const float* pos, float* tokens, int ldt, int n_seq, int npatch, int D,
                              hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(cls && pos && tokens && n_seq > 0 && npatch > 0 && D > 0, "bad operand");
  hipLaunchKernelGGL(embed_cls_kernel, dim3(ceil_div(n_seq * D, 256)), dim3(256), 0, stream, cls, pos, tokens, ldt, n_seq, npatch, D);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int64_t lafs_embed_bwd_workspace_bytes(int n_seq, int npatch, int D) {
  if (n_seq <= 0 || npatch <= 0 || D <= 0) return -1;
  return (int64_t)ceil_div(n_seq, EB_SEQ) * (npatch + 1) * D * 4;
}

extern "C" int lafs_embed_bwd(const float* g, int ldg, int n_seq, int npatch, int D, void* gp, float* dpos, float* dcls,
                              float* workspace, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(g && dpos && n_seq > 0 && npatch > 0 && D > 0 && D % 4 == 0 && ldg % 4 == 0, "bad operand");
  const int chunks = ceil_div(n_seq, EB_SEQ), cols = ceil_div((npatch + 1) * (D / 4), 256);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(cols, chunks), dim3(256), 0, stream, g, ldg, n_seq, npatch, D, (bf16_t*)gp, dpos, dcls, workspace);
  if (workspace != nullptr)
    hipLaunchKernelGGL(embed_bwd_fold_kernel, dim3(cols), dim3(256), 0, stream, workspace, chunks, npatch, D, dpos, dcls);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_gather_cls(const float* x, int ldx, const int32_t* cu_seqlens, int n_seq, int D, float* out_f32,
                               hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(x && cu_seqlens && out_f32 && n_seq > 0 && D > 0, "bad operand");
  hipLaunchKernelGGL(gather_cls_kernel, dim3(ceil_div(n_seq * D, 256)), dim3(256), 0, stream, x, ldx, cu_seqlens, n_seq, D, out_f32);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_scatter_cls(const float* src, const int32_t* cu_seqlens, int n_seq, int D, float* g, int ldg,
                                hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(src && cu_seqlens && g && n_seq > 0 && D > 0, "bad operand");
  hipLaunchKernelGGL(scatter_cls_kernel, dim3(ceil_div(n_seq * D, 256)), dim3(256), 0, stream, src, cu_seqlens, n_seq, D, g, ldg);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
