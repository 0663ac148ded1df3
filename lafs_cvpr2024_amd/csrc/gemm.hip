// bf16 MFMA GEMMs for the LAFS hot path (gfx950, wave64, v_mfma_f32_16x16x32_bf16).
//
//  gemm_nt : C[M,N] = A[M,K] * B[N,K]^T  (+ fused epilogue)      forward linears and dgrad (with W^T copies)
//  gemm_tn : C[N1,N2] += A[M,N1]^T * B[M,N2]  (split over M, fp32 atomics)   weight gradients
//
// Replaces the cuBLAS calls behind nn.Linear / nn.Conv2d(k=s=8) on the reference's path
// (vision_transformer.py:59-65, 75-90, 126-131, 295-301; face_pre_pro/ViT_face.py:126-137, 147-149, 761).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 each = 4x4 MFMA tiles),
// 64-deep k-steps, double-buffered LDS (64 KiB -> 2 workgroups per CU), register-staged global loads
// issued one tile ahead.  NT operands sit K-contiguous in LDS with a 16-byte-chunk XOR swizzle
// (chunk ^= row & 7) so that ds_read_b128 fragment reads are bank-conflict free; TN operands sit
// M-major and are read with ds_read_b64_tr_b16 (hardware transpose) through an XOR swizzle on 8-byte units.
// The NT kernel computes C^T tiles (MFMA A-operand = weight rows) and permutes which weight row feeds which
// MFMA row so that every lane ends up with 16 CONTIGUOUS output columns of one row: epilogues read/write
// 32-64 B per lane (bias, residual, GELU, fp32/bf16 stores) with no LDS transpose.
#include "common.hpp"
#include "lafs_hip.h"

namespace {

enum {
  EPI_BF16 = LAFS_EPI_BF16,
  EPI_BF16_GELU = LAFS_EPI_BF16_GELU,
  EPI_RESID_F32 = LAFS_EPI_RESID_F32,
  EPI_F32 = LAFS_EPI_F32,
  EPI_DGELU_BF16 = LAFS_EPI_DGELU_BF16,
  EPI_ATOMIC_F32 = LAFS_EPI_ATOMIC_F32,
  EPI_EMBED_F32 = LAFS_EPI_EMBED_F32,
};

struct NTArgs {
  const bf16_t* A; const bf16_t* B;
  int M, N, K, lda, ldb, klen;
  void* C; int ldc; void* C2; int ldc2;
  const float* bias; const float* resid; int ldr;
  const float* seq_scale; const int* row2seq;
  const bf16_t* aux; int ldaux;
  const float* pos; int npatch;
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile

__device__ __forceinline__ int nt_perm(int r) {   // LDS row -> weight row inside the 128-row tile
  return (r & 64) + ((r >> 2) & 3) * 16 + ((r >> 4) & 3) * 4 + (r & 3);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NTArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.z * p.klen;
  const int kend = min(p.K, kbeg + p.klen);
  const int nk = (kend - kbeg) / BK;

  // ---- per-thread staging coordinates: 4 x 16-byte chunks of A and of B per k-step ----
  const bf16_t* ga[4]; const bf16_t* gb[4]; int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
    ga[i] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + kbeg + ch * 8;
    gb[i] = p.B + (size_t)min(n0 + nt_perm(row), p.N - 1) * p.ldb + kbeg + ch * 8;
    soff[i] = row * 128 + ((ch ^ (row & 7)) << 4);
  }
  uint4 ra[4], rb[4];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const uint4*>(ga[i] + t * BK);
      rb[i] = *reinterpret_cast<const uint4*>(gb[i] + t * BK);
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* sa = smem + buf * 2 * TILE_BYTES;
    unsigned char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(sa + soff[i]) = ra[i];
      *reinterpret_cast<uint4*>(sb + soff[i]) = rb[i];
    }
  };

  f32x4_t acc[4][4];                                 // [j: column group][i: row tile]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  if (nk > 0) { gload(0); sstore(0); }
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) gload(t + 1);
    const unsigned char* sa = smem + cur * 2 * TILE_BYTES;
    const unsigned char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra_ = wr * 64 + i * 16 + frow;
        fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + ra_ * 128 + (((kk * 4 + fq) ^ (ra_ & 7)) << 4));
        const int rb_ = wc * 64 + i * 16 + frow;
        fb[i] = *reinterpret_cast<const bf16x8_t*>(sb + rb_ * 128 + (((kk * 4 + fq) ^ (rb_ & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mfma16(fb[j], fa[i], acc[j][i]);
    }
    if (t + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane owns rows m = m0 + wr*64 + i*16 + (lane&15), columns nb .. nb+15 ----
  const int nb = n0 + wc * 64 + fq * 16;
  if (nb >= p.N) return;
  const bool full = (nb + 16 <= p.N);
  float bias[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bias[e] = 0.f;
  if (p.bias != nullptr && EPI != EPI_ATOMIC_F32 && EPI != EPI_DGELU_BF16) {
#pragma unroll
    for (int e = 0; e < 16; ++e) if (full || nb + e < p.N) bias[e] = p.bias[nb + e];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + frow;
    if (m >= p.M) continue;
    float v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[j][i][r] + bias[j * 4 + r];

    if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_DGELU_BF16) {
      if (EPI == EPI_DGELU_BF16) {
        const bf16_t* ax = p.aux + (size_t)m * p.ldaux + nb;
#pragma unroll
        for (int e = 0; e < 16; ++e) if (full || nb + e < p.N) v[e] *= gelu_grad_f(bf2f(ax[e]));
      }
      bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + nb;
      bf16_t* c2 = (EPI == EPI_BF16_GELU) ? reinterpret_cast<bf16_t*>(p.C2) + (size_t)m * p.ldc2 + nb : nullptr;
      if (full) {
        uint32_t w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        reinterpret_cast<uint4*>(c)[0] = make_uint4(w[0], w[1], w[2], w[3]);
        reinterpret_cast<uint4*>(c)[1] = make_uint4(w[4], w[5], w[6], w[7]);
        if (EPI == EPI_BF16_GELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = pack_bf2(gelu_f(v[2 * e]), gelu_f(v[2 * e + 1]));
          reinterpret_cast<uint4*>(c2)[0] = make_uint4(w[0], w[1], w[2], w[3]);
          reinterpret_cast<uint4*>(c2)[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (nb + e < p.N) {
            c[e] = f2bf(v[e]);
            if (EPI == EPI_BF16_GELU) c2[e] = f2bf(gelu_f(v[e]));
          }
      }
    } else if (EPI == EPI_ATOMIC_F32) {
      float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + nb;
#pragma unroll
      for (int e = 0; e < 16; ++e) if (full || nb + e < p.N) atomicAdd(c + e, v[e]);
    } else {
      size_t orow = (size_t)m;
      if (EPI == EPI_RESID_F32) {
        const float s = (p.seq_scale != nullptr) ? p.seq_scale[p.row2seq[m]] : 1.0f;
        const float* rs = p.resid + (size_t)m * p.ldr + nb;
#pragma unroll
        for (int e = 0; e < 16; ++e) if (full || nb + e < p.N) v[e] = rs[e] + s * v[e];
      } else if (EPI == EPI_EMBED_F32) {
        const int b = m / p.npatch, t = m - b * p.npatch;
        orow = (size_t)m + b + 1;                       // one cls row in front of every sequence
        const float* ps = p.pos + (size_t)(t + 1) * p.N + nb;
#pragma unroll
        for (int e = 0; e < 16; ++e) if (full || nb + e < p.N) v[e] += ps[e];
      }
      float* c = reinterpret_cast<float*>(p.C) + orow * p.ldc + nb;
      if (full) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          reinterpret_cast<float4*>(c)[e] = make_float4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) if (nb + e < p.N) c[e] = v[e];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ TN
struct TNArgs {
  const bf16_t* A; const bf16_t* B; float* C;
  int M, N1, N2, lda, ldb, ldc, mlen;
};

__device__ __forceinline__ int tn_f(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 2; }

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TNArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n1_0 = blockIdx.y * 128, n2_0 = blockIdx.x * 128;
  const int mbeg = blockIdx.z * p.mlen;
  const int mend = min(p.M, mbeg + p.mlen);
  const int nk = (mend - mbeg + 63) / 64;

  int srow[4], sch[4], soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    srow[i] = c >> 4; sch[i] = c & 15;
    soff[i] = srow[i] * 256 + ((sch[i] ^ (tn_f(srow[i]) >> 1)) << 4);
  }
  uint4 ra[4], rb[4];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mbeg + t * 64 + srow[i];
      const int ca = n1_0 + sch[i] * 8, cb = n2_0 + sch[i] * 8;
      ra[i] = (m < mend && ca < p.N1) ? *reinterpret_cast<const uint4*>(p.A + (size_t)m * p.lda + ca) : make_uint4(0, 0, 0, 0);
      rb[i] = (m < mend && cb < p.N2) ? *reinterpret_cast<const uint4*>(p.B + (size_t)m * p.ldb + cb) : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* sa = smem + buf * 2 * TILE_BYTES;
    unsigned char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(sa + soff[i]) = ra[i];
      *reinterpret_cast<uint4*>(sb + soff[i]) = rb[i];
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, pl = lane & 15;
  if (nk > 0) { gload(0); sstore(0); }
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) gload(t + 1);
    const unsigned char* sa = smem + cur * 2 * TILE_BYTES;
    const unsigned char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        s16x4_t lo_a, hi_a, lo_b, hi_b;
        {
          const int r0 = kk * 32 + g * 8 + (pl >> 2), r1 = r0 + 4;
          const int ua = wm * 16 + x * 4 + (pl & 3), ub = wn * 16 + x * 4 + (pl & 3);
          lo_a = lds_read_tr16(sa + r0 * 256 + ((ua ^ tn_f(r0)) << 3));
          hi_a = lds_read_tr16(sa + r1 * 256 + ((ua ^ tn_f(r1)) << 3));
          lo_b = lds_read_tr16(sb + r0 * 256 + ((ub ^ tn_f(r0)) << 3));
          hi_b = lds_read_tr16(sb + r1 * 256 + ((ub ^ tn_f(r1)) << 3));
        }
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        s16x8_t va = __builtin_shufflevector(lo_a, hi_a, 0, 1, 2, 3, 4, 5, 6, 7);
        s16x8_t vb = __builtin_shufflevector(lo_b, hi_b, 0, 1, 2, 3, 4, 5, 6, 7);
        fa[x] = __builtin_bit_cast(bf16x8_t, va);
        fb[x] = __builtin_bit_cast(bf16x8_t, vb);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
    }
    if (t + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n1 = n1_0 + wm * 64 + a * 16 + g * 4 + r;
      if (n1 >= p.N1) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int n2 = n2_0 + wn * 64 + b * 16 + pl;
        if (n2 < p.N2) atomicAdd(p.C + (size_t)n1 * p.ldc + n2, acc[a][b][r]);
      }
    }
}

template <int EPI>
int launch_nt(const NTArgs& a, int splits, hipStream_t s) {
  const int tiles = ceil_div(a.M, BM) * ceil_div(a.N, BN);
  hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(tiles, 1, splits), dim3(256), 0, s, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

}  // namespace

extern "C" int lafs_gemm_nt(const lafs_gemm_nt_args* g, hipStream_t stream) {
  LAFS_CHECK_ARG(g != nullptr && g->A && g->B && g->C, "null operand");
  LAFS_CHECK_ARG(g->M > 0 && g->N > 0 && g->K > 0, "empty problem");
  LAFS_CHECK_ARG(g->K % 64 == 0, "K must be a multiple of 64");
  LAFS_CHECK_ARG(g->lda % 8 == 0 && g->ldb % 8 == 0, "lda/ldb must be multiples of 8 elements (16-byte rows)");
  NTArgs a;
  a.A = (const bf16_t*)g->A; a.B = (const bf16_t*)g->B;
  a.M = g->M; a.N = g->N; a.K = g->K; a.lda = g->lda; a.ldb = g->ldb;
  a.C = g->C; a.ldc = g->ldc; a.C2 = g->C2; a.ldc2 = g->ldc2;
  a.bias = g->bias; a.resid = g->resid; a.ldr = g->ldr;
  a.seq_scale = g->seq_scale; a.row2seq = g->row2seq;
  a.aux = (const bf16_t*)g->aux; a.ldaux = g->ldaux; a.pos = g->pos; a.npatch = g->npatch;
  int splits = 1;
  a.klen = g->K;
  if (g->epilogue == LAFS_EPI_ATOMIC_F32) {
    splits = g->splits > 0 ? g->splits : 1;
    const int ksteps = g->K / 64;
    splits = splits > ksteps ? ksteps : splits;
    a.klen = ceil_div(ksteps, splits) * 64;
    splits = ceil_div(g->K, a.klen);
  }
  const bool vec_ok = (g->ldc % 8 == 0);
  LAFS_CHECK_ARG(vec_ok, "ldc must be a multiple of 8 elements");
  switch (g->epilogue) {
    case LAFS_EPI_BF16: return launch_nt<EPI_BF16>(a, 1, stream);
    case LAFS_EPI_BF16_GELU:
      LAFS_CHECK_ARG(g->C2 != nullptr && g->ldc2 % 8 == 0, "GELU epilogue needs C2");
      return launch_nt<EPI_BF16_GELU>(a, 1, stream);
    case LAFS_EPI_RESID_F32:
      LAFS_CHECK_ARG(g->resid != nullptr && g->ldr % 4 == 0, "residual epilogue needs resid");
      LAFS_CHECK_ARG(g->seq_scale == nullptr || g->row2seq != nullptr, "seq_scale needs row2seq");
      return launch_nt<EPI_RESID_F32>(a, 1, stream);
    case LAFS_EPI_F32: return launch_nt<EPI_F32>(a, 1, stream);
    case LAFS_EPI_DGELU_BF16:
      LAFS_CHECK_ARG(g->aux != nullptr, "dGELU epilogue needs aux (pre-activation)");
      return launch_nt<EPI_DGELU_BF16>(a, 1, stream);
    case LAFS_EPI_ATOMIC_F32: return launch_nt<EPI_ATOMIC_F32>(a, splits, stream);
    case LAFS_EPI_EMBED_F32:
      LAFS_CHECK_ARG(g->pos != nullptr && g->npatch > 0 && g->M % g->npatch == 0, "embed epilogue needs pos/npatch");
      return launch_nt<EPI_EMBED_F32>(a, 1, stream);
    default:
      lafs_set_error("lafs_gemm_nt: unknown epilogue %d", g->epilogue);
      return LAFS_EINVAL;
  }
}

extern "C" int lafs_gemm_tn_acc(const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                                int M, int N1, int N2, int splits, hipStream_t stream) {
  LAFS_CHECK_ARG(A && B && C, "null operand");
  LAFS_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0, "empty problem");
  LAFS_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && N1 % 8 == 0 && N2 % 8 == 0, "N1/N2/lda/ldb must be multiples of 8");
  TNArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C;
  a.M = M; a.N1 = N1; a.N2 = N2; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  const int msteps = ceil_div(M, 64);
  if (splits <= 0) {                       // aim for ~4 workgroups per CU
    const int tiles = ceil_div(N1, 128) * ceil_div(N2, 128);
    splits = ceil_div(1024, tiles);
  }
  if (splits > msteps) splits = msteps;
  a.mlen = ceil_div(msteps, splits) * 64;
  splits = ceil_div(M, a.mlen);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(ceil_div(N2, 128), ceil_div(N1, 128), splits), dim3(256), 0, stream, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
