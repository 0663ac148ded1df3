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
// The backward uses the same trick in two phases of ONE kernel over the LDS-resident Q, K, V, dO of a (sequence, head):
// dK/dV (a wave per run of key tiles, streams the query tiles) and dQ (a wave per run of query tiles, streams the key
// tiles); P is recomputed from the saved log-sum-exp, delta = rowsum(O * dO) is formed while staging.
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

// 16 bytes of an operand the BACKWARD kernels read exactly once (q, k, v, dO, O of a pair): non-temporal, so that these streams do not
// displace what the kernels running beside the attention backward re-read from the L2.  LAFS_ATTN_NT=0: plain loads (lab).
#ifndef LAFS_ATTN_NT
#define LAFS_ATTN_NT 1
#endif
__device__ __forceinline__ uint4 ld_once16(const bf16_t* p) {
#if LAFS_ATTN_NT
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const uint4*>(p);
#endif
}

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
  const int* __restrict__ cu; int n_seq, heads; float scale;
  bf16_t* out; int ldo; float* lse;
  const bf16_t* dout; int lddo; bf16_t* dqkv; int lddqkv;
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

// ------------------------------------------------------------------------------------------------ backward, one launch
// Q (F), dO (F), K (F) and V (R) of a (sequence, head) pair are staged ONCE; delta = rowsum(O * dO) is formed while staging
// (the 8 threads that carry a row's eight 16-byte chunks reduce their partial dot products with three shuffles).  Phase A is
// the dK/dV loop (one wave per 16-key tile, streams the query tiles), phase B the dQ loop (one wave per 16-query tile,
// streams the key tiles); both only read LDS.  Against the three-kernel form (delta, dQ, dK/dV) the pair's operands cross HBM
// once instead of 2-3 times: 4.4 -> 3.1 bytes per token-channel.
//
// No masks in either phase: rows beyond the sequence are ZERO in all four tiles, so an out-of-range query contributes
// dO = 0 / Q = 0 to dV / dK, an out-of-range key contributes K = 0 to dQ, and out-of-range outputs are never stored; the
// probabilities of padded positions only have to stay finite (their lse entry is 0).
// Each wave item covers TPI consecutive 16-row tiles, so every LDS fragment (the MFMA A operand) feeds TPI MFMAs: with one
// tile per item the loops ask the LDS for 256 B/clk/CU at MFMA rate -- twice what it delivers -- and sit in LDS issue stalls.

// phase A item `it`: dK, dV of key tiles it*TPI .. +TPI-1 (their K / V row fragments in kf / vf: a tile index past NT only occurs
// in the last item, its results are dropped); streams the query tiles of Q (F) / dO (F); lsd = -log2(e) lse | -scale delta
// the thread id as a value the compiler cannot trace: per-thread offsets derived from it are recomputed where they are used instead of
// being hoisted out of the pair loop, kept live across the MFMA loops and spilled (their scratch reloads wait on vmcnt, i.e. on every
// request in flight)
__device__ __forceinline__ int opaque_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
// a wave-uniform read of read-only memory through the scalar cache (a plain load of `cu` becomes a vector load + s_waitcnt vmcnt(0):
// in the pair loop that drains every request and store in flight)
__device__ __forceinline__ int load_const(const int* p) {
  typedef const __attribute__((address_space(4))) int* cptr_t;
  return *(cptr_t)(unsigned long long)p;
}
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
// (`hook(u)` runs in front of step u of the unrolled loop: the streaming kernel hangs its staging there)
template <int NT, int TPI, typename H = NoHook>
__device__ __forceinline__ void bwd_phase_keys(const unsigned char* QsF, const unsigned char* DsF, const float* lsd,
                                               const bf16x8_t (&kf)[TPI][2], const bf16x8_t (&vf)[TPI][2], int it, int len, bf16_t* drow,
                                               int ld, int inner, int lane, float c2, float scale, const H& hook = H()) {
  const int g = lane >> 4, c16 = lane & 15;
  const bool swp = (lds_f(c16) & 1) != 0;
  f32x4_t dk[TPI][4], dv[TPI][4];
#pragma unroll
  for (int x = 0; x < TPI; ++x)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[x][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[x][dt] = dk[x][dt]; }
#pragma unroll
  for (int u = 0; u < (NT + 1) / 2; ++u) {
    hook(u);
    float pv[TPI][8], ds[TPI][8];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int qt = 2 * u + tt;
      if (qt < NT) {
        const int qrow = qt * 16 + c16;
        const bf16x8_t qa0 = rfrag_f(QsF, qrow, g, swp), qa1 = rfrag_f(QsF, qrow, 4 + g, swp);
        const bf16x8_t da0 = rfrag_f(DsF, qrow, g, swp), da1 = rfrag_f(DsF, qrow, 4 + g, swp);
        const float4 l4 = *reinterpret_cast<const float4*>(lsd + qt * 16 + g * 4);
        const float4 d4 = *reinterpret_cast<const float4*>(lsd + NT * 16 + qt * 16 + g * 4);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int x = 0; x < TPI; ++x) {
          f32x4_t sx = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sx = mfma16(qa0, kf[x][0], sx);
          sx = mfma16(qa1, kf[x][1], sx);
          dp = mfma16(da0, vf[x][0], dp);
          dp = mfma16(da1, vf[x][1], dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sx[r], c2, lq[r]));
            pv[x][tt * 4 + r] = p;
            ds[x][tt * 4 + r] = p * fmaf(dp[r], scale, dq4[r]);
          }
        }
      } else {
#pragma unroll
        for (int x = 0; x < TPI; ++x)
#pragma unroll
          for (int r = 0; r < 4; ++r) { pv[x][tt * 4 + r] = 0.f; ds[x][tt * 4 + r] = 0.f; }
      }
    }
    bf16x8_t pf[TPI], dsf[TPI];
#pragma unroll
    for (int x = 0; x < TPI; ++x) { pf[x] = pack_frag(pv[x]); dsf[x] = pack_frag(ds[x]); }
    const int t0 = 2 * u, t1 = (2 * u + 1 < NT) ? 2 * u + 1 : 2 * u;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16x8_t td = tfrag_f(DsF, t0, t1, dt, lane), tq = tfrag_f(QsF, t0, t1, dt, lane);
#pragma unroll
      for (int x = 0; x < TPI; ++x) {
        dv[x][dt] = mfma16(td, pf[x], dv[x][dt]);
        dk[x][dt] = mfma16(tq, dsf[x], dk[x][dt]);
      }
    }
  }
#pragma unroll
  for (int x = 0; x < TPI; ++x) {
    const int key = (it * TPI + x) * 16 + c16;
    if (it * TPI + x < NT && key < len) {
      uint32_t w[8];
      bf16_t* op = drow + (size_t)key * ld + inner + g * 16;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        w[2 * dt] = pack_bf2(dk[x][dt][0], dk[x][dt][1]);
        w[2 * dt + 1] = pack_bf2(dk[x][dt][2], dk[x][dt][3]);
      }
      reinterpret_cast<uint4*>(op)[0] = make_uint4(w[0], w[1], w[2], w[3]);
      reinterpret_cast<uint4*>(op)[1] = make_uint4(w[4], w[5], w[6], w[7]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        w[2 * dt] = pack_bf2(dv[x][dt][0], dv[x][dt][1]);
        w[2 * dt + 1] = pack_bf2(dv[x][dt][2], dv[x][dt][3]);
      }
      reinterpret_cast<uint4*>(op + inner)[0] = make_uint4(w[0], w[1], w[2], w[3]);
      reinterpret_cast<uint4*>(op + inner)[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

// phase B item `it`: dQ of query tiles it*TPI .. +TPI-1 (their own Q / dO fragments from the F tiles); streams K (F) / V (R)
template <int NT, int TPI, typename H = NoHook>
__device__ __forceinline__ void bwd_phase_queries(const unsigned char* QsF, const unsigned char* DsF, const unsigned char* KsF,
                                                  const unsigned char* VsR, const float* lsd, int it, int len, bf16_t* drow, int ld,
                                                  int lane, float c2, float scale, const H& hook = H()) {
  const int g = lane >> 4, c16 = lane & 15;
  const bool swp = (lds_f(c16) & 1) != 0;
  bf16x8_t qf[TPI][2], df[TPI][2];
  float nlse_q[TPI], ndel_q[TPI];
  f32x4_t dq[TPI][4];
#pragma unroll
  for (int x = 0; x < TPI; ++x) {
    const int q = min(it * TPI + x, NT - 1) * 16 + c16;
    qf[x][0] = rfrag_f(QsF, q, g, swp); qf[x][1] = rfrag_f(QsF, q, 4 + g, swp);
    df[x][0] = rfrag_f(DsF, q, g, swp); df[x][1] = rfrag_f(DsF, q, 4 + g, swp);
    nlse_q[x] = lsd[q]; ndel_q[x] = lsd[NT * 16 + q];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[x][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < (NT + 1) / 2; ++u) {
    hook(u);
    float ds[TPI][8];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = 2 * u + tt;
      if (t < NT) {
        const int krow = t * 16 + c16;
        const bf16x8_t ka0 = rfrag_f(KsF, krow, g, swp), ka1 = rfrag_f(KsF, krow, 4 + g, swp);
        const bf16x8_t va0 = rfrag(VsR, krow, g), va1 = rfrag(VsR, krow, 4 + g);
#pragma unroll
        for (int x = 0; x < TPI; ++x) {
          f32x4_t sx = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sx = mfma16(ka0, qf[x][0], sx);
          sx = mfma16(ka1, qf[x][1], sx);
          dp = mfma16(va0, df[x][0], dp);
          dp = mfma16(va1, df[x][1], dp);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            ds[x][tt * 4 + r] = __builtin_amdgcn_exp2f(fmaf(sx[r], c2, nlse_q[x])) * fmaf(dp[r], scale, ndel_q[x]);
        }
      } else {
#pragma unroll
        for (int x = 0; x < TPI; ++x)
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[x][tt * 4 + r] = 0.f;
      }
    }
    bf16x8_t dsf[TPI];
#pragma unroll
    for (int x = 0; x < TPI; ++x) dsf[x] = pack_frag(ds[x]);
    const int t0 = 2 * u, t1 = (2 * u + 1 < NT) ? 2 * u + 1 : 2 * u;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16x8_t tk = tfrag_f(KsF, t0, t1, dt, lane);
#pragma unroll
      for (int x = 0; x < TPI; ++x) dq[x][dt] = mfma16(tk, dsf[x], dq[x][dt]);
    }
  }
#pragma unroll
  for (int x = 0; x < TPI; ++x) {
    const int q = (it * TPI + x) * 16 + c16;
    if (it * TPI + x < NT && q < len) {
      uint32_t w[8];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        w[2 * dt] = pack_bf2(dq[x][dt][0], dq[x][dt][1]);
        w[2 * dt + 1] = pack_bf2(dq[x][dt][2], dq[x][dt][3]);
      }
      bf16_t* op = drow + (size_t)q * ld + g * 16;
      reinterpret_cast<uint4*>(op)[0] = make_uint4(w[0], w[1], w[2], w[3]);
      reinterpret_cast<uint4*>(op)[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

// LDS per pair: Q | dO | K | V tiles, then lse[NT*16] and delta[NT*16] (f32).  PPB pairs per workgroup, NW waves, no barrier between
// the phases (both only read LDS).
template <int NT, int PPB, int NW, int TPI>
__global__ __launch_bounds__(NW * 64) void attn_bwd_fused_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TILE = NT * 16 * 128;
  constexpr int PAIR_BYTES = 4 * TILE + 2 * NT * 16 * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int inner = a.heads * 64;
  const int n_pairs = a.n_seq * a.heads;
  {
    constexpr int TPP = (NW * 64) / PPB;
    constexpr int ROWS = NT * 16;
    constexpr int IT = (ROWS * 8 + TPP - 1) / TPP;
    const int pl = tid / TPP, lt = tid % TPP;
    const int pair = blockIdx.x * PPB + pl;
    if (pair < n_pairs) {
      const int seq = pair / a.heads, h = pair % a.heads;
      const int tok0 = a.cu[seq], len = a.cu[seq + 1] - tok0;
      const bf16_t* qb = a.qkv + (size_t)tok0 * a.ldqkv + h * 64;
      const bf16_t* db = a.dout + (size_t)tok0 * a.lddo + h * 64;
      const bf16_t* ob = a.out + (size_t)tok0 * a.ldo + h * 64;
      unsigned char* s0 = smem + pl * PAIR_BYTES;
      float* lsd = reinterpret_cast<float*>(s0 + 4 * TILE);
      uint4 vq[IT], vd[IT], vk[IT], vv[IT], vo[IT];
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int idx = lt + i * TPP, row = idx >> 3, ch = idx & 7;
        vq[i] = make_uint4(0, 0, 0, 0); vd[i] = vq[i]; vk[i] = vq[i]; vv[i] = vq[i]; vo[i] = vq[i];
        if (idx < ROWS * 8 && row < len) {
          const bf16_t* qr = qb + (size_t)row * a.ldqkv + ch * 8;
          vq[i] = ld_once16(qr);
          vk[i] = ld_once16(qr + inner);
          vv[i] = ld_once16(qr + 2 * inner);
          vd[i] = ld_once16(db + (size_t)row * a.lddo + ch * 8);
          vo[i] = ld_once16(ob + (size_t)row * a.ldo + ch * 8);
        }
      }
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int idx = lt + i * TPP, row = idx >> 3, ch = idx & 7;
        const uint4 x = vo[i], y = vd[i];
        float d = bf_lo(x.x) * bf_lo(y.x) + bf_hi(x.x) * bf_hi(y.x) + bf_lo(x.y) * bf_lo(y.y) + bf_hi(x.y) * bf_hi(y.y) +
                  bf_lo(x.z) * bf_lo(y.z) + bf_hi(x.z) * bf_hi(y.z) + bf_lo(x.w) * bf_lo(y.w) + bf_hi(x.w) * bf_hi(y.w);
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
        if (idx < ROWS * 8) {
          lds_put<2>(s0, row, ch, vq[i]); lds_put<2>(s0 + TILE, row, ch, vd[i]);
          lds_put<2>(s0 + 2 * TILE, row, ch, vk[i]); lds_put<0>(s0 + 3 * TILE, row, ch, vv[i]);
          if (ch == 0) {                                   // stored negated and pre-scaled: the loops use them as FMA addends
            lsd[row] = (row < len) ? -1.4426950408889634f * a.lse[(size_t)(tok0 + row) * a.heads + h] : 0.f;
            lsd[ROWS + row] = -a.scale * d;
          }
        }
      }
    }
  }
  __syncthreads();
  const bool swp = (lds_f(c16) & 1) != 0;
  const float c2 = a.scale * 1.4426950408889634f;      // softmax scale folded with log2(e): p = exp2(s * c2 - lse * log2(e))
  constexpr int NI = (NT + TPI - 1) / TPI;              // items per pair and phase
  // ---- phase A: dK, dV
  for (int item = wave; item < PPB * NI; item += NW) {
    const int pl = item % PPB, it = item / PPB;
    const int pair = blockIdx.x * PPB + pl;
    if (pair >= n_pairs) continue;
    const int seq = pair / a.heads, h = pair % a.heads;
    const int tok0 = a.cu[seq], len = a.cu[seq + 1] - tok0;
    if (it * TPI * 16 >= len) continue;
    const unsigned char* QsF = smem + pl * PAIR_BYTES;
    const unsigned char* KsF = QsF + 2 * TILE;
    const unsigned char* VsR = QsF + 3 * TILE;
    bf16x8_t kf[TPI][2], vf[TPI][2];
#pragma unroll
    for (int x = 0; x < TPI; ++x) {
      const int key = min(it * TPI + x, NT - 1) * 16 + c16;
      kf[x][0] = rfrag_f(KsF, key, g, swp); kf[x][1] = rfrag_f(KsF, key, 4 + g, swp);
      vf[x][0] = rfrag(VsR, key, g); vf[x][1] = rfrag(VsR, key, 4 + g);
    }
    bwd_phase_keys<NT, TPI>(QsF, QsF + TILE, reinterpret_cast<const float*>(QsF + 4 * TILE), kf, vf, it, len,
                            a.dqkv + (size_t)tok0 * a.lddqkv + h * 64, a.lddqkv, inner, lane, c2, a.scale);
  }
  // ---- phase B: dQ (waves take the items in the opposite order, which evens out the two phases' remainders)
  for (int item = NW - 1 - wave; item < PPB * NI; item += NW) {
    const int pl = item % PPB, it = item / PPB;
    const int pair = blockIdx.x * PPB + pl;
    if (pair >= n_pairs) continue;
    const int seq = pair / a.heads, h = pair % a.heads;
    const int tok0 = a.cu[seq], len = a.cu[seq + 1] - tok0;
    if (it * TPI * 16 >= len) continue;
    const unsigned char* QsF = smem + pl * PAIR_BYTES;
    bwd_phase_queries<NT, TPI>(QsF, QsF + TILE, QsF + 2 * TILE, QsF + 3 * TILE, reinterpret_cast<const float*>(QsF + 4 * TILE), it, len,
                               a.dqkv + (size_t)tok0 * a.lddqkv + h * 64, a.lddqkv, lane, c2, a.scale);
  }
}

// ------------------------------------------------------------------------------------------------ backward, long sequences
// Sequences of 10-13 tiles (the 197-token crops): 108 KB of LDS per pair leave ONE 8-wave workgroup per CU, so the time a pair's
// operands take to arrive (a third of the kernel: the same loops on resident operands run 37 us, the staging alone 19, the
// one-pair-per-workgroup kernel 73) was exposed three times per CU.  The dQ phase only streams K / V and the dK/dV phase only
// Q / dO; the other pair of matrices is read once, as each wave's own row fragments, when the phase starts.  So the LDS holds
// Q | dO twice and K | V once (6 tiles + 2 x lse / delta = 159.3 of 160 KiB at 13 tiles) and ONE workgroup per CU walks over the
// pairs b, b + grid, ..., dQ phase first:
//   loop top : barrier (Q | dO and K | V of the pair are in place)
//   dQ       : streams K | V; in front of its steps 0 / H2: request | deposit the NEXT pair's Q and lse into the other Q | dO buffer
//   between  : every wave reads its own K / V fragments of the dK/dV phase; barrier (nobody reads K | V any more)
//   dK, dV   : streams Q | dO; stops 0 / H2 / H3: request the next pair's K, V | deposit them (F / R), request dO, O | deposit dO and
//              delta = rowsum(O * dO)
// Every matrix crosses HBM once, a tile or two at a time on the phases' unrolled loops (16-32 registers per thread in flight), and
// only the first pair's staging and two barriers per pair are outside the MFMA loops.
template <int NT>
__global__ __launch_bounds__(512) void attn_bwd_stream_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TILE = NT * 16 * 128, ROWS = NT * 16, TPI = 2;
  constexpr int IT = (ROWS * 8 + 511) / 512;
  constexpr int NI = (NT + TPI - 1) / TPI, U = (NT + 1) / 2, H2 = U / 3, H3 = 2 * U / 3 + 1;
  static_assert(NI <= 8 && 6 * TILE + 4 * ROWS * 4 <= 160 * 1024 && H2 >= 1 && H3 > H2 && H3 < U, "one item per wave and phase; LDS; stops");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const bool swp = (lds_f(c16) & 1) != 0;
  const int inner = a.heads * 64;
  const int n_pairs = a.n_seq * a.heads;
  const float c2 = a.scale * 1.4426950408889634f;
  unsigned char* const KV = smem + 4 * TILE;               // K (F) | V (R)
  float* const lsd_base = reinterpret_cast<float*>(smem + 6 * TILE);
  const int itB = wave, itA = 7 - wave;                     // this wave's item of the dQ / the dK,dV phase (idle when >= NI)
  uint4 r0[IT], r1[IT]; float rl[IT];                      // a tile (or two) on its way to the LDS (thread: row idx >> 3, chunk idx & 7)
  // staging context: the pair whose operands are being fetched, and where its Q | dO go
  const bf16_t* s_q = nullptr; const bf16_t* s_d = nullptr; const bf16_t* s_o = nullptr; const float* s_l = nullptr;
  int s_len = 0; unsigned char* s_x = nullptr; float* s_lsd = nullptr;
  auto stage_pair = [&](int pair, int buf) __attribute__((always_inline)) {
    const int seq = pair / a.heads, h = pair % a.heads;
    const int tok0 = load_const(a.cu + seq);
    s_len = load_const(a.cu + seq + 1) - tok0;
    s_q = a.qkv + (size_t)tok0 * a.ldqkv + h * 64;
    s_d = a.dout + (size_t)tok0 * a.lddo + h * 64;
    s_o = a.out + (size_t)tok0 * a.ldo + h * 64;
    s_l = a.lse + (size_t)tok0 * a.heads + h;
    s_x = smem + buf * 2 * TILE; s_lsd = lsd_base + buf * 2 * ROWS;
  };
  // one matrix of the staged pair -> r: rows of `src` (row stride ld), zero past the sequence
  auto request = [&](uint4 (&r)[IT], const bf16_t* src, int ld) __attribute__((always_inline)) {
    const int t_ = opaque_tid();
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = t_ + i * 512, row = idx >> 3, ch = idx & 7;
      r[i] = make_uint4(0, 0, 0, 0);
      if (idx < ROWS * 8 && row < s_len) r[i] = ld_once16(src + (size_t)row * ld + ch * 8);
    }
  };
  auto request_lse = [&]() __attribute__((always_inline)) {
    const int t_ = opaque_tid();
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = t_ + i * 512, row = idx >> 3, ch = idx & 7;
      rl[i] = 0.f;
      if (ch == 0 && idx < ROWS * 8 && row < s_len) rl[i] = s_l[(size_t)row * a.heads];
    }
  };
  auto deposit_f = [&](unsigned char* dst, const uint4 (&r)[IT]) __attribute__((always_inline)) {
    const int t_ = opaque_tid();
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = t_ + i * 512, row = idx >> 3, ch = idx & 7;
      if (idx < ROWS * 8) lds_put<2>(dst, row, ch, r[i]);
    }
  };
  auto deposit_r = [&](unsigned char* dst, const uint4 (&r)[IT]) __attribute__((always_inline)) {
    const int t_ = opaque_tid();
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = t_ + i * 512, row = idx >> 3, ch = idx & 7;
      if (idx < ROWS * 8) lds_put<0>(dst, row, ch, r[i]);
    }
  };
  auto deposit_lse = [&]() __attribute__((always_inline)) {
    const int t_ = opaque_tid();
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = t_ + i * 512, row = idx >> 3, ch = idx & 7;
      if (ch == 0 && idx < ROWS * 8) s_lsd[row] = -1.4426950408889634f * rl[i];       // negated and pre-scaled: FMA addends
    }
  };
  // dO in r0, O in r1: dO -> LDS (F), delta = rowsum(O * dO) -> lsd
  auto deposit_do_delta = [&]() __attribute__((always_inline)) {
    const int t_ = opaque_tid();
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = t_ + i * 512, row = idx >> 3, ch = idx & 7;
      const uint4 x = r1[i], y = r0[i];
      float d = bf_lo(x.x) * bf_lo(y.x) + bf_hi(x.x) * bf_hi(y.x) + bf_lo(x.y) * bf_lo(y.y) + bf_hi(x.y) * bf_hi(y.y) +
                bf_lo(x.z) * bf_lo(y.z) + bf_hi(x.z) * bf_hi(y.z) + bf_lo(x.w) * bf_lo(y.w) + bf_hi(x.w) * bf_hi(y.w);
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
      if (idx < ROWS * 8) {
        lds_put<2>(s_x + TILE, row, ch, r0[i]);
        if (ch == 0) s_lsd[ROWS + row] = -a.scale * d;
      }
    }
  };
#ifndef LAFS_LAB_ATTN_ABL
#define LAFS_LAB_ATTN_ABL 0
#endif
  bool more = true;
  // the stops of the dQ phase (the next pair's Q, lse) and of the dK/dV phase (its K, V, dO, delta)
  auto stop_q = [&](int u) __attribute__((always_inline)) {
    if (!more || (LAFS_LAB_ATTN_ABL & 2)) return;
    if (u == 0) { request(r0, s_q, a.ldqkv); request_lse(); }
    else if (u == H2) { deposit_f(s_x, r0); deposit_lse(); }
  };
  auto stop_k = [&](int u) __attribute__((always_inline)) {
    if (!more || (LAFS_LAB_ATTN_ABL & 1)) return;
    if (u == 0) { request(r0, s_q + inner, a.ldqkv); request(r1, s_q + 2 * inner, a.ldqkv); }
    else if (u == H2) { deposit_f(KV, r0); deposit_r(KV + TILE, r1); request(r0, s_d, a.lddo); request(r1, s_o, a.ldo); }
    else if (u == H3) deposit_do_delta();
  };

  int pair = blockIdx.x, cur = 0;
  stage_pair(pair, 0);                                      // the first pair: nothing to hide behind
  stop_q(0); stop_q(H2); stop_k(0); stop_k(H2); stop_k(H3);
  for (;;) {
    const int seq = pair / a.heads, h = pair % a.heads;
    const int tok0 = load_const(a.cu + seq), len = load_const(a.cu + seq + 1) - tok0;
    bf16_t* drow = a.dqkv + (size_t)tok0 * a.lddqkv + h * 64;
    const int next = pair + gridDim.x;
    more = next < n_pairs;
    if (more) stage_pair(next, cur ^ 1);
    __syncthreads();
    const unsigned char* QsF = smem + cur * 2 * TILE;
    const float* lsd = lsd_base + cur * 2 * ROWS;
    if (itB < NI && itB * TPI * 16 < len && !(LAFS_LAB_ATTN_ABL & 8))
      bwd_phase_queries<NT, TPI>(QsF, QsF + TILE, KV, KV + TILE, lsd, itB, len, drow, a.lddqkv, lane, c2, a.scale, stop_q);
    else { stop_q(0); stop_q(H2); }
    bf16x8_t kf[TPI][2], vf[TPI][2];
#pragma unroll
    for (int x = 0; x < TPI; ++x) {
      const int key = min(itA * TPI + x, NT - 1) * 16 + c16;
      kf[x][0] = rfrag_f(KV, key, g, swp); kf[x][1] = rfrag_f(KV, key, 4 + g, swp);
      vf[x][0] = rfrag(KV + TILE, key, g); vf[x][1] = rfrag(KV + TILE, key, 4 + g);
    }
    __syncthreads();
    if (itA < NI && itA * TPI * 16 < len && !(LAFS_LAB_ATTN_ABL & 4))
      bwd_phase_keys<NT, TPI>(QsF, QsF + TILE, lsd, kf, vf, itA, len, drow, a.lddqkv, inner, lane, c2, a.scale, stop_k);
    else { stop_k(0); stop_k(H2); stop_k(H3); }
    if (!more) break;
    pair = next; cur ^= 1;
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

// which: 0 = forward (2 tiles per pair), 1 = backward (4 tiles + lse/delta per pair)
template <int NT, int PPB>
int dispatch(int which, const AttnArgs& a, hipStream_t s) {
  constexpr int NW = 4;          // measured: 8 waves/workgroup is slower (36.7 vs 31.0 us fwd at 128 x 197)
  const int n_pairs = a.n_seq * a.heads;
  const size_t tile = (size_t)NT * 16 * 128;
  if (which == 0) return launch_attn(attn_fwd_kernel<NT, PPB, NW>, n_pairs, PPB, NW * 64, PPB * 2 * tile, a, s);
  // Backward (tools/bench_kernels.py attn, 6 heads): long sequences take one pair per 8-wave workgroup (108 KB of LDS at 197
  // tokens: one workgroup per CU) with two tiles per wave item: 128 x 197 in 74 us against 97 (one tile per item) and 99 for
  // the former delta + dQ + dK/dV kernels; short ones two pairs per 4-wave workgroup, one tile per item: 512 x 37 in 32 us
  // against 42.
  const size_t pair_bytes = 4 * tile + 2 * NT * 16 * 4;
#ifndef LAFS_LAB_ATTN_BWD_OLD
  if constexpr (NT >= 10 && NT <= 13) {
    // (per call, for the device that is current NOW: no process-global state -- another device of the same process has its own count)
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    return launch_attn(attn_bwd_stream_kernel<NT>, min(n_pairs, n_cu), 1, 512, 6 * tile + 4 * NT * 16 * 4, a, s);
  }
#endif
  if constexpr (NT >= 7) return launch_attn(attn_bwd_fused_kernel<NT, 1, 8, 2>, n_pairs, 1, 512, pair_bytes, a, s);
  else return launch_attn(attn_bwd_fused_kernel<NT, 2, 4, 1>, n_pairs, 2, 256, 2 * pair_bytes, a, s);
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
                                  const float* lse, const int32_t* cu_seqlens, int n_seq, int max_len, int heads, float scale,
                                  void* dqkv, int lddqkv, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(qkv && out_bf16 && dout_bf16 && lse && cu_seqlens && dqkv, "null operand");
  LAFS_CHECK_ARG(n_seq > 0 && heads > 0 && max_len > 0 && max_len <= 256, "sequence length must be in 1..256");
  LAFS_CHECK_ARG(ldqkv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddqkv % 8 == 0, "row strides must be multiples of 8");
  AttnArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ldqkv = ldqkv; a.cu = cu_seqlens; a.n_seq = n_seq; a.heads = heads; a.scale = scale;
  a.out = (bf16_t*)out_bf16; a.ldo = ldo; a.lse = const_cast<float*>(lse);
  a.dout = (const bf16_t*)dout_bf16; a.lddo = lddo; a.dqkv = (bf16_t*)dqkv; a.lddqkv = lddqkv;
  return dispatch_len(1, max_len, a, stream);
}
