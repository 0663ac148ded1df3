// Fused multi-head self-attention, forward and backward, for short variable-length sequences
// (LAFS: 197 tokens for 112x112 crops, 37 for 48x48 landmark crops; head_dim 64) on gfx950.
//
// Replaces the three unfused kernels that materialise [B,h,N,N] at vision_transformer.py:85-89 and
// face_pre_pro/ViT_face.py:155-179.  A whole K/V head (<= 256 x 64 bf16 = 32 KiB) is staged once in LDS; the
// score matrix never leaves registers.
//
// MFMA orientation.  Everything is computed TRANSPOSED so that a lane's accumulator registers line up with
// what the next MFMA needs and with wide global stores:
//   S^T = K Q^T  (A-operand = K rows from LDS, B-operand = Q rows from HBM)  -> lane holds, for ONE query
//   (lane&15), 4 consecutive keys per 16-key tile.  Two tiles = the 8 k-slots of the next MFMA's B operand,
//   so P feeds  O^T = V^T P^T  straight from registers; V^T (A-operand) comes from a row-major V tile via the
//   gfx950 LDS transpose read (ds_read_b64_tr_b16).  The d-columns are permuted in that read so each lane
//   ends with 16 contiguous output channels (two 16-byte stores).
// The backward uses the same trick in two kernels: dQ (one wave per 16-query tile, streams key tiles) and
// dK/dV (one wave per 16-key tile, streams query tiles); P is recomputed from the saved log-sum-exp.
//
// LDS layouts for a [rows][64] bf16 tile (128-byte rows):
//   R (row fragments, ds_read_b128): 16-byte chunk c of row r stored at chunk c ^ (r & 7)
//   T (transpose reads):             8-byte unit u of row r stored at unit  u ^ ((r >> 1) & 3)
//   F (both, used by the backward):  8-byte unit u of row r stored at unit  u ^ F(r), F = x0 | x1<<1 | x1<<2 | x0<<3 with
//                                    x = (r >> 1) & 3; a row fragment is then ONE ds_read_b128 at chunk c ^ (F >> 1) whose
//                                    two 8-byte halves arrive swapped when F & 1 (undone with 4 v_cndmask).
// All three are conflict-free for their access patterns (tools/lds_bank_sim.py).
#include <mutex>
#include <set>
#include "common.hpp"
#include "lafs_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ bf16x8_t to_frag(s16x4_t lo, s16x4_t hi) {
  s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t pack_frag(const float* v) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  u32x4_t w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
  return __builtin_bit_cast(bf16x8_t, w);
}

__device__ __forceinline__ int lds_f(int row) {
  const int x = (row >> 1) & 3;
  return (x & 1) | ((x >> 1) << 1) | ((x >> 1) << 2) | ((x & 1) << 3);
}

// Stage ROWS rows x 64 columns (bf16) starting at src (row stride ld elements) into LDS; rows >= len are zero.
// LAYOUT 0 = R, 1 = T, 2 = F.  NTHR cooperating threads, this thread's index `lt`.  All global loads are issued before the
// first LDS store (a load->store loop serialises ~7 HBM round trips per tile and dominated the kernel).
template <int LAYOUT>
__device__ __forceinline__ void lds_put(unsigned char* dst, int row, int ch, uint4 v) {
  if (LAYOUT == 0) {
    *reinterpret_cast<uint4*>(dst + row * 128 + ((ch ^ (row & 7)) << 4)) = v;
  } else if (LAYOUT == 2) {
    const int f = lds_f(row);
    if (f & 1) v = make_uint4(v.z, v.w, v.x, v.y);
    *reinterpret_cast<uint4*>(dst + row * 128 + ((ch ^ (f >> 1)) << 4)) = v;
  } else {
    const int x = (row >> 1) & 3;
    if (x & 1) v = make_uint4(v.z, v.w, v.x, v.y);          // units 2ch,2ch+1 swap places under ^1
    *reinterpret_cast<uint4*>(dst + row * 128 + ((ch ^ (x >> 1)) << 4)) = v;
  }
}
template <int LA, int LB, int ROWS, int NTHR>
__device__ __forceinline__ void stage_pair(unsigned char* dA, const bf16_t* sA, int ldA, unsigned char* dB, const bf16_t* sB, int ldB,
                                           int len, int lt) {
  constexpr int IT = (ROWS * 8 + NTHR - 1) / NTHR;
  uint4 va[IT], vb[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int idx = lt + i * NTHR, row = idx >> 3, ch = idx & 7;
    va[i] = make_uint4(0, 0, 0, 0); vb[i] = make_uint4(0, 0, 0, 0);
    if (idx < ROWS * 8 && row < len) {
      va[i] = *reinterpret_cast<const uint4*>(sA + (size_t)row * ldA + ch * 8);
      vb[i] = *reinterpret_cast<const uint4*>(sB + (size_t)row * ldB + ch * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int idx = lt + i * NTHR, row = idx >> 3, ch = idx & 7;
    if (idx < ROWS * 8) { lds_put<LA>(dA, row, ch, va[i]); lds_put<LB>(dB, row, ch, vb[i]); }
  }
}

// row fragment (8 consecutive channels of one row) from an R tile
__device__ __forceinline__ bf16x8_t rfrag(const unsigned char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}
// transposed fragment from a T tile: k-slots = rows {ta*16 + g*4 + 0..3, tb*16 + g*4 + 0..3}; MFMA row i <-> channel
// (i>>2)*16 + dt*4 + (i&3)
__device__ __forceinline__ bf16x8_t tfrag(const unsigned char* tile, int ta, int tb, int dt, int lane) {
  const int g = lane >> 4, p = lane & 15;
  const int unit = (p & 3) * 4 + dt;
  const int ra = ta * 16 + g * 4 + (p >> 2), rb = tb * 16 + g * 4 + (p >> 2);
  s16x4_t lo = lds_read_tr16(tile + ra * 128 + ((unit ^ ((ra >> 1) & 3)) << 3));
  s16x4_t hi = lds_read_tr16(tile + rb * 128 + ((unit ^ ((rb >> 1) & 3)) << 3));
  return to_frag(lo, hi);
}

// row fragment from an F tile (swap = F(row) & 1 is the same for every 16-row tile: it only depends on lane & 15)
__device__ __forceinline__ bf16x8_t rfrag_f(const unsigned char* tile, int row, int chunk, bool swap) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  u32x4_t v = *reinterpret_cast<const u32x4_t*>(tile + row * 128 + ((chunk ^ (lds_f(row) >> 1)) << 4));
  u32x4_t w = {swap ? v[2] : v[0], swap ? v[3] : v[1], swap ? v[0] : v[2], swap ? v[1] : v[3]};
  return __builtin_bit_cast(bf16x8_t, w);
}
// transposed fragment from an F tile (same contract as tfrag)
__device__ __forceinline__ bf16x8_t tfrag_f(const unsigned char* tile, int ta, int tb, int dt, int lane) {
  const int g = lane >> 4, p = lane & 15;
  const int unit = (p & 3) * 4 + dt;
  const int ra = ta * 16 + g * 4 + (p >> 2), rb = tb * 16 + g * 4 + (p >> 2);
  s16x4_t lo = lds_read_tr16(tile + ra * 128 + ((unit ^ lds_f(ra)) << 3));
  s16x4_t hi = lds_read_tr16(tile + rb * 128 + ((unit ^ lds_f(rb)) << 3));
  return to_frag(lo, hi);
}

struct AttnArgs {
  const bf16_t* qkv; int ldqkv;
  const int* cu; int n_seq, heads; float scale;
  bf16_t* out; int ldo; float* lse;
  const bf16_t* dout; int lddo; const float* delta; bf16_t* dqkv; int lddqkv;
};

// ------------------------------------------------------------------------------------------------ forward
template <int NT, int PPB, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TILE = NT * 16 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int inner = a.heads * 64;
  const int n_pairs = a.n_seq * a.heads;
  {
    constexpr int TPP = (NW * 64) / PPB;
    const int pl = tid / TPP, lt = tid % TPP;
    const int pair = blockIdx.x * PPB + pl;
    if (pair < n_pairs) {
      const int seq = pair / a.heads, h = pair % a.heads;
      const int tok0 = a.cu[seq], len = a.cu[seq + 1] - tok0;
      const bf16_t* base = a.qkv + (size_t)tok0 * a.ldqkv + h * 64;
      stage_pair<0, 1, NT * 16, TPP>(smem + pl * 2 * TILE, base + inner, a.ldqkv, smem + pl * 2 * TILE + TILE, base + 2 * inner,
                                     a.ldqkv, len, lt);
    }
  }
  __syncthreads();
  for (int item = wave; item < PPB * NT; item += NW) {
    const int pl = item % PPB, qt = item / PPB;
    const int pair = blockIdx.x * PPB + pl;
    if (pair >= n_pairs) continue;
    const int seq = pair / a.heads, h = pair % a.heads;
    const int tok0 = a.cu[seq], len = a.cu[seq + 1] - tok0;
    if (qt * 16 >= len) continue;
    const unsigned char* Ks = smem + pl * 2 * TILE;
    const unsigned char* Vs = Ks + TILE;
    const int q = qt * 16 + c16;
    const bf16_t* qp = a.qkv + (size_t)(tok0 + min(q, len - 1)) * a.ldqkv + h * 64 + g * 8;
    const bf16x8_t qf0 = *reinterpret_cast<const bf16x8_t*>(qp);
    const bf16x8_t qf1 = *reinterpret_cast<const bf16x8_t*>(qp + 32);

    f32x4_t st[NT];
    float mx = -INFINITY;
    const float c2 = a.scale * 1.4426950408889634f;          // scores kept in the log2 domain: p = exp2(s*c2 - max)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4_t s = {0.f, 0.f, 0.f, 0.f};
      const int krow = t * 16 + c16;
      s = mfma16(rfrag(Ks, krow, g), qf0, s);
      s = mfma16(rfrag(Ks, krow, 4 + g), qf1, s);
      if (t * 16 + 16 <= len) {                               // wave-uniform: only a ragged last tile pays for the mask
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r] *= c2; mx = fmaxf(mx, s[r]); }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[r] = (t * 16 + g * 4 + r < len) ? s[r] * c2 : -INFINITY;
          mx = fmaxf(mx, s[r]);
        }
      }
      st[t] = s;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(st[t][r] - mx);
        st[t][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);

    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < (NT + 1) / 2; ++u) {
      const int t0 = 2 * u, t1 = (2 * u + 1 < NT) ? 2 * u + 1 : 2 * u;
      float pv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[r] = st[t0][r];
        pv[4 + r] = (2 * u + 1 < NT) ? st[t1][r] : 0.f;
      }
      const bf16x8_t pf = pack_frag(pv);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(tfrag(Vs, t0, t1, dt, lane), pf, o[dt]);
    }
    if (q < len) {
      const float inv = 1.0f / sum;
      uint32_t w[8];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        w[2 * dt] = pack_bf2(o[dt][0] * inv, o[dt][1] * inv);
        w[2 * dt + 1] = pack_bf2(o[dt][2] * inv, o[dt][3] * inv);
      }
      bf16_t* op = a.out + (size_t)(tok0 + q) * a.ldo + h * 64 + g * 16;
      reinterpret_cast<uint4*>(op)[0] = make_uint4(w[0], w[1], w[2], w[3]);
      reinterpret_cast<uint4*>(op)[1] = make_uint4(w[4], w[5], w[6], w[7]);
      if (g == 0) a.lse[(size_t)(tok0 + q) * a.heads + h] = (mx + __log2f(sum)) * 0.6931471805599453f;   // natural log-sum-exp
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: delta
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, int ldo, const bf16_t* __restrict__ d_o,
                                                        int lddo, float* __restrict__ delta, int T, int heads) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= T * heads) return;
  const int t = i / heads, h = i % heads;
  const uint4* po = reinterpret_cast<const uint4*>(o + (size_t)t * ldo + h * 64);
  const uint4* pd = reinterpret_cast<const uint4*>(d_o + (size_t)t * lddo + h * 64);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const uint4 x = po[c], y = pd[c];
    s += bf_lo(x.x) * bf_lo(y.x) + bf_hi(x.x) * bf_hi(y.x) + bf_lo(x.y) * bf_lo(y.y) + bf_hi(x.y) * bf_hi(y.y) +
         bf_lo(x.z) * bf_lo(y.z) + bf_hi(x.z) * bf_hi(y.z) + bf_lo(x.w) * bf_lo(y.w) + bf_hi(x.w) * bf_hi(y.w);
  }
  delta[i] = s;
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// LDS per pair: K (F: row fragments and transpose reads), V (R)
template <int NT, int PPB, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TILE = NT * 16 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int inner = a.heads * 64;
  const int n_pairs = a.n_seq * a.heads;
  {
    constexpr int TPP = (NW * 64) / PPB;
    const int pl = tid / TPP, lt = tid % TPP;
    const int pair = blockIdx.x * PPB + pl;
    if (pair < n_pairs) {
      const int seq = pair / a.heads, h = pair % a.heads;
      const int tok0 = a.cu[seq], len = a.cu[seq + 1] - tok0;
      const bf16_t* base = a.qkv + (size_t)tok0 * a.ldqkv + h * 64;
      unsigned char* s0 = smem + pl * 2 * TILE;
      stage_pair<2, 0, NT * 16, TPP>(s0, base + inner, a.ldqkv, s0 + TILE, base + 2 * inner, a.ldqkv, len, lt);
    }
  }
  __syncthreads();
  for (int item = wave; item < PPB * NT; item += NW) {
    const int pl = item % PPB, qt = item / PPB;
    const int pair = blockIdx.x * PPB + pl;
    if (pair >= n_pairs) continue;
    const int seq = pair / a.heads, h = pair % a.heads;
    const int tok0 = a.cu[seq], len = a.cu[seq + 1] - tok0;
    if (qt * 16 >= len) continue;
    const unsigned char* KsF = smem + pl * 2 * TILE;
    const unsigned char* VsR = KsF + TILE;
    const bool swp = (lds_f(c16) & 1) != 0;
    const int q = qt * 16 + c16;
    const int qc = min(q, len - 1);
    const bf16_t* qp = a.qkv + (size_t)(tok0 + qc) * a.ldqkv + h * 64 + g * 8;
    const bf16_t* dp_ = a.dout + (size_t)(tok0 + qc) * a.lddo + h * 64 + g * 8;
    const bf16x8_t qf0 = *reinterpret_cast<const bf16x8_t*>(qp), qf1 = *reinterpret_cast<const bf16x8_t*>(qp + 32);
    const bf16x8_t df0 = *reinterpret_cast<const bf16x8_t*>(dp_), df1 = *reinterpret_cast<const bf16x8_t*>(dp_ + 32);
    const float lse_q = a.lse[(size_t)(tok0 + qc) * a.heads + h];
    const float del_q = a.delta[(size_t)(tok0 + qc) * a.heads + h];

    f32x4_t dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < (NT + 1) / 2; ++u) {
      float ds[8];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * u + tt;
        if (t < NT) {
          const int krow = t * 16 + c16;
          f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = mfma16(rfrag_f(KsF, krow, g, swp), qf0, s);
          s = mfma16(rfrag_f(KsF, krow, 4 + g, swp), qf1, s);
          dp = mfma16(rfrag(VsR, krow, g), df0, dp);
          dp = mfma16(rfrag(VsR, krow, 4 + g), df1, dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = (t * 16 + g * 4 + r < len) ? __expf(s[r] * a.scale - lse_q) : 0.f;
            ds[tt * 4 + r] = p * (dp[r] - del_q) * a.scale;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[tt * 4 + r] = 0.f;
        }
      }
      const bf16x8_t dsf = pack_frag(ds);
      const int t0 = 2 * u, t1 = (2 * u + 1 < NT) ? 2 * u + 1 : 2 * u;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16(tfrag_f(KsF, t0, t1, dt, lane), dsf, dq[dt]);
    }
    if (q < len) {
      uint32_t w[8];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        w[2 * dt] = pack_bf2(dq[dt][0], dq[dt][1]);
        w[2 * dt + 1] = pack_bf2(dq[dt][2], dq[dt][3]);
      }
      bf16_t* op = a.dqkv + (size_t)(tok0 + q) * a.lddqkv + h * 64 + g * 16;
      reinterpret_cast<uint4*>(op)[0] = make_uint4(w[0], w[1], w[2], w[3]);
      reinterpret_cast<uint4*>(op)[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// LDS per pair: Q (F), dO (F), then lse[NT*16] and delta[NT*16] (f32)
template <int NT, int PPB, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TILE = NT * 16 * 128;
  constexpr int PAIR_BYTES = 2 * TILE + 2 * NT * 16 * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int inner = a.heads * 64;
  const int n_pairs = a.n_seq * a.heads;
  {
    constexpr int TPP = (NW * 64) / PPB;
    const int pl = tid / TPP, lt = tid % TPP;
    const int pair = blockIdx.x * PPB + pl;
    if (pair < n_pairs) {
      const int seq = pair / a.heads, h = pair % a.heads;
      const int tok0 = a.cu[seq], len = a.cu[seq + 1] - tok0;
      const bf16_t* qb = a.qkv + (size_t)tok0 * a.ldqkv + h * 64;
      const bf16_t* db = a.dout + (size_t)tok0 * a.lddo + h * 64;
      unsigned char* s0 = smem + pl * PAIR_BYTES;
      stage_pair<2, 2, NT * 16, TPP>(s0, qb, a.ldqkv, s0 + TILE, db, a.lddo, len, lt);
      float* lsd = reinterpret_cast<float*>(s0 + 2 * TILE);
      for (int r = lt; r < NT * 16; r += TPP) {
        const bool ok = r < len;
        lsd[r] = ok ? a.lse[(size_t)(tok0 + r) * a.heads + h] : 0.f;
        lsd[NT * 16 + r] = ok ? a.delta[(size_t)(tok0 + r) * a.heads + h] : 0.f;
      }
    }
  }
  __syncthreads();
  for (int item = wave; item < PPB * NT; item += NW) {
    const int pl = item % PPB, kt = item / PPB;
    const int pair = blockIdx.x * PPB + pl;
    if (pair >= n_pairs) continue;
    const int seq = pair / a.heads, h = pair % a.heads;
    const int tok0 = a.cu[seq], len = a.cu[seq + 1] - tok0;
    if (kt * 16 >= len) continue;
    const unsigned char* QsF = smem + pl * PAIR_BYTES;
    const unsigned char* DsF = QsF + TILE;
    const float* lsd = reinterpret_cast<const float*>(QsF + 2 * TILE);
    const bool swp = (lds_f(c16) & 1) != 0;
    const int key = kt * 16 + c16;
    const int kc = min(key, len - 1);
    const bf16_t* kp = a.qkv + (size_t)(tok0 + kc) * a.ldqkv + inner + h * 64 + g * 8;
    const bf16_t* vp = kp + inner;
    const bf16x8_t kf0 = *reinterpret_cast<const bf16x8_t*>(kp), kf1 = *reinterpret_cast<const bf16x8_t*>(kp + 32);
    const bf16x8_t vf0 = *reinterpret_cast<const bf16x8_t*>(vp), vf1 = *reinterpret_cast<const bf16x8_t*>(vp + 32);
    const bool key_ok = key < len;

    f32x4_t dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[dt] = dk[dt]; }
#pragma unroll
    for (int u = 0; u < (NT + 1) / 2; ++u) {
      float pv[8], ds[8];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int qt = 2 * u + tt;
        if (qt < NT) {
          const int qrow = qt * 16 + c16;
          f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = mfma16(rfrag_f(QsF, qrow, g, swp), kf0, s);
          s = mfma16(rfrag_f(QsF, qrow, 4 + g, swp), kf1, s);
          dp = mfma16(rfrag_f(DsF, qrow, g, swp), vf0, dp);
          dp = mfma16(rfrag_f(DsF, qrow, 4 + g, swp), vf1, dp);
          const float4 l4 = *reinterpret_cast<const float4*>(lsd + qt * 16 + g * 4);
          const float4 d4 = *reinterpret_cast<const float4*>(lsd + NT * 16 + qt * 16 + g * 4);
          const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + g * 4 + r;
            const float del_q = dq4[r];
            const float p = (q < len && key_ok) ? __expf(s[r] * a.scale - lq[r]) : 0.f;
            pv[tt * 4 + r] = p;
            ds[tt * 4 + r] = p * (dp[r] - del_q) * a.scale;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { pv[tt * 4 + r] = 0.f; ds[tt * 4 + r] = 0.f; }
        }
      }
      const bf16x8_t pf = pack_frag(pv), dsf = pack_frag(ds);
      const int t0 = 2 * u, t1 = (2 * u + 1 < NT) ? 2 * u + 1 : 2 * u;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = mfma16(tfrag_f(DsF, t0, t1, dt, lane), pf, dv[dt]);
        dk[dt] = mfma16(tfrag_f(QsF, t0, t1, dt, lane), dsf, dk[dt]);
      }
    }
    if (key_ok) {
      uint32_t w[8];
      bf16_t* op = a.dqkv + (size_t)(tok0 + key) * a.lddqkv + inner + h * 64 + g * 16;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        w[2 * dt] = pack_bf2(dk[dt][0], dk[dt][1]);
        w[2 * dt + 1] = pack_bf2(dk[dt][2], dk[dt][3]);
      }
      reinterpret_cast<uint4*>(op)[0] = make_uint4(w[0], w[1], w[2], w[3]);
      reinterpret_cast<uint4*>(op)[1] = make_uint4(w[4], w[5], w[6], w[7]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        w[2 * dt] = pack_bf2(dv[dt][0], dv[dt][1]);
        w[2 * dt + 1] = pack_bf2(dv[dt][2], dv[dt][3]);
      }
      reinterpret_cast<uint4*>(op + inner)[0] = make_uint4(w[0], w[1], w[2], w[3]);
      reinterpret_cast<uint4*>(op + inner)[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

template <typename K>
int launch_attn(K kernel, int n_pairs, int ppb, int threads, size_t lds, const AttnArgs& a, hipStream_t s) {
  // raise the dynamic-LDS limit once per kernel instantiation (not a stream operation; kept out of graph capture)
  static std::mutex mu;
  static std::set<const void*> done;
  {
    std::lock_guard<std::mutex> lock(mu);
    const void* key = reinterpret_cast<const void*>(kernel);
    if (done.find(key) == done.end()) {
      hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        lafs_set_error("attention: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
        return (int)e;
      }
      done.insert(key);
    }
  }
  hipLaunchKernelGGL(kernel, dim3(ceil_div(n_pairs, ppb)), dim3(threads), lds, s, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

// which: 0 = fwd (2 tiles/pair), 1 = dq (3), 2 = dkv (4)
template <int NT, int PPB>
int dispatch(int which, const AttnArgs& a, hipStream_t s) {
  // 8 waves per workgroup for long sequences: 13 query/key tiles spread over 8 waves (2 rounds) instead of 4 (4 rounds)
  constexpr int NW = 4;          // measured: 8 waves/workgroup is slower (36.7 vs 31.0 us fwd at 128 x 197)
  const int n_pairs = a.n_seq * a.heads;
  const size_t tile = (size_t)NT * 16 * 128;
  if (which == 0) return launch_attn(attn_fwd_kernel<NT, PPB, NW>, n_pairs, PPB, NW * 64, PPB * 2 * tile, a, s);
  if (which == 1) return launch_attn(attn_bwd_dq_kernel<NT, PPB, NW>, n_pairs, PPB, NW * 64, PPB * 2 * tile, a, s);
  return launch_attn(attn_bwd_dkv_kernel<NT, PPB, NW>, n_pairs, PPB, NW * 64, PPB * (2 * tile + 2 * NT * 16 * 4), a, s);
}

int dispatch_len(int which, int max_len, const AttnArgs& a, hipStream_t s) {
  const int nt = ceil_div(max_len, 16);
  if (nt <= 1) return dispatch<1, 4>(which, a, s);
  if (nt <= 2) return dispatch<2, 4>(which, a, s);
  if (nt <= 3) return dispatch<3, 4>(which, a, s);
  if (nt <= 4) return dispatch<4, 4>(which, a, s);
  if (nt <= 7) return dispatch<7, 2>(which, a, s);
  if (nt <= 10) return dispatch<10, 1>(which, a, s);
  if (nt <= 13) return dispatch<13, 1>(which, a, s);
  return dispatch<16, 1>(which, a, s);
}

}  // namespace

extern "C" int lafs_attention_fwd(const void* qkv, int ldqkv, const int32_t* cu_seqlens, int n_seq, int max_len, int heads,
                                  float scale, void* out_bf16, int ldo, float* lse, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(qkv && cu_seqlens && out_bf16 && lse, "null operand");
  LAFS_CHECK_ARG(n_seq > 0 && heads > 0 && max_len > 0 && max_len <= 256, "sequence length must be in 1..256");
  LAFS_CHECK_ARG(ldqkv % 8 == 0 && ldo % 8 == 0, "row strides must be multiples of 8 elements");
  AttnArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ldqkv = ldqkv; a.cu = cu_seqlens; a.n_seq = n_seq; a.heads = heads; a.scale = scale;
  a.out = (bf16_t*)out_bf16; a.ldo = ldo; a.lse = lse;
  return dispatch_len(0, max_len, a, stream);
}

extern "C" int lafs_attention_bwd(const void* qkv, int ldqkv, const void* out_bf16, int ldo, const void* dout_bf16, int lddo,
                                  const float* lse, float* delta, const int32_t* cu_seqlens, int n_seq, int n_tok, int max_len,
                                  int heads, float scale, void* dqkv, int lddqkv, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(qkv && out_bf16 && dout_bf16 && lse && delta && cu_seqlens && dqkv, "null operand");
  LAFS_CHECK_ARG(n_seq > 0 && n_tok >= 0 && heads > 0 && max_len > 0 && max_len <= 256, "sequence length must be in 1..256");
  LAFS_CHECK_ARG(ldqkv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddqkv % 8 == 0, "row strides must be multiples of 8");
  AttnArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ldqkv = ldqkv; a.cu = cu_seqlens; a.n_seq = n_seq; a.heads = heads; a.scale = scale;
  a.out = (bf16_t*)out_bf16; a.ldo = ldo; a.lse = const_cast<float*>(lse);
  a.dout = (const bf16_t*)dout_bf16; a.lddo = lddo; a.delta = delta; a.dqkv = (bf16_t*)dqkv; a.lddqkv = lddqkv;
  if (n_tok > 0) {
    hipLaunchKernelGGL(attn_delta_kernel, dim3(ceil_div(n_tok * heads, 256)), dim3(256), 0, stream, a.out, ldo, a.dout, lddo,
                       delta, n_tok, heads);
    LAFS_LAUNCH_CHECK();
  }
  const int rc = dispatch_len(1, max_len, a, stream);
  if (rc != LAFS_OK) return rc;
  return dispatch_len(2, max_len, a, stream);
}
