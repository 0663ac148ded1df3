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
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "lafs_hip.h"
#include "gemm_kres.hpp"
#include "gemm_big.hpp"
#include "ctx.hpp"

// Timing ablations that change RESULTS (no stores / no MFMA / ...) exist only in the -DLAFS_ABLATE build (make ablate ->
// liblafs_hip_ablate.so, used by tools/bench_kernels.py): in the product library the branches below are compiled out, so no
// environment variable or lafs_debug_set call can make a kernel skip work.
#ifdef LAFS_ABLATE
#define DBG(p, bits) ((p).dbg & (bits))
#else
#define DBG(p, bits) (0)
#endif

namespace {

#ifdef LAFS_ABLATE
unsigned long long* g_stamps = nullptr;                // lab: per-workgroup phase time stamps of the next NT launches
#define STAMP(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)(stamp_id + n_vb * blockIdx.z) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

enum {
  EPI_BF16 = LAFS_EPI_BF16,
  EPI_BF16_GELU = LAFS_EPI_BF16_GELU,
  EPI_RESID_F32 = LAFS_EPI_RESID_F32,
  EPI_F32 = LAFS_EPI_F32,
  EPI_DGELU_BF16 = LAFS_EPI_DGELU_BF16,
  EPI_ATOMIC_F32 = LAFS_EPI_ATOMIC_F32,
  EPI_EMBED_F32 = LAFS_EPI_EMBED_F32,
  EPI_BF16_ACT = LAFS_EPI_BF16_ACT,
};

struct NTArgs {
  const bf16_t* A; const bf16_t* B;
  int M, N, K, lda, ldb, klen;
  int n_tiles;                      // persistent variants: output tiles to walk (the grid is one residency wave)
  void* C; int ldc; void* C2; int ldc2;
  const float* bias; const float* resid; int ldr;
  const float* seq_scale; const int* row2seq;
  const bf16_t* aux; int ldaux;
  const float* pos; int npatch;
  int f16;                          // operands / 16-bit outputs are fp16 (lafs_gemm_nt_args::operand_f16)
  int dbg;                          // timing ablations (lafs_debug_set): 16 = no epilogue stores, 32 = no MFMA/ds_read
  DropCfg drop;                     // element dropout on the linear's output (RESID_F32) / on GELU(u) (BF16_GELU, DGELU_BF16)
  int act;                          // BF16_ACT: LAFS_ACT_*
#ifdef LAFS_ABLATE
  unsigned long long* stamps;
#endif
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));
// 16-byte epilogue stores; `nt` selects the non-temporal (streaming) form -- activations written here are consumed by a
// later kernel, never by this one
__device__ __forceinline__ void st16(void* p, unsigned a, unsigned b, unsigned c, unsigned d, bool nt) {
  const u32x4_t v = {a, b, c, d};
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
  else *reinterpret_cast<u32x4_t*>(p) = v;
}
__device__ __forceinline__ void st16f(void* p, float a, float b, float c, float d, bool nt) {
  const f32x4v_t v = {a, b, c, d};
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4v_t*>(p));
  else *reinterpret_cast<f32x4v_t*>(p) = v;
}


// LDS row -> weight row inside the 128-row tile.  MFMA row slot s = (rho & 15) of column-group j = (rho >> 4) & 3 ends
// up in lane group g = s >> 2, register r = s & 3.  VPL = how many CONSECUTIVE output columns one lane should own so
// that the 4 lane groups of a row write one contiguous 64-byte segment per store instruction:
//   VPL 4 (fp32 outputs): column = j*16 + g*4 + r          (identity)
//   VPL 8 (bf16 outputs): column = (j>>1)*32 + g*8 + (j&1)*4 + r
template <int VPL>
__device__ __forceinline__ int nt_perm(int rho) {
  const int j = (rho >> 4) & 3, g = (rho >> 2) & 3, r = rho & 3;
  if (VPL == 4) return rho;
  return (rho & ~63) + (j >> 1) * 32 + g * 8 + (j & 1) * 4 + r;       // bits >= 6: which 64-column wave slice (WN up to 4)
}
template <int EPI> struct EpiTraits {
  static constexpr bool f32_out = (EPI == LAFS_EPI_RESID_F32 || EPI == LAFS_EPI_F32 || EPI == LAFS_EPI_ATOMIC_F32 || EPI == LAFS_EPI_EMBED_F32);
  static constexpr int VPL = f32_out ? 4 : 8;
};
// 64-byte LDS rows (BK = 32): logical 16-byte chunk c of row r lives at chunk c ^ ((-(r >> 2)) & 3), which makes the
// ds_read_b128 fragment reads (row = lane & 15, chunk = lane >> 4) conflict-free in every 16-lane service group.
__device__ __forceinline__ int nt_swz(int row) { return (-(row >> 2)) & 3; }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware, bijective block -> tile map: blocks b, b+8, b+16, ... share an XCD (and its L2); give each XCD a
// contiguous run of tiles so that the n-tiles of one m-tile (same A rows) hit the same L2.
__device__ __forceinline__ int xcd_tile(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// WM = wave rows: 2 -> 128x128 tile, 256 threads; 4 -> 256x128 tile, 512 threads (one third less L2->LDS traffic per
// flop).  BK = k-depth of a pipeline stage: 32 (3-stage ring, 64-byte row pieces) or 64 (2-stage ring, full 128-byte
// cache lines per row piece).
template <int BK> __device__ __forceinline__ int nt_swzk(int row) { return BK == 32 ? nt_swz(row) : (row & 7); }

// WN = wave columns: 2 -> 128-wide tiles (default), 4 -> 256-wide (256x256 with WM = 4: 16 waves, one workgroup per CU, half the
// L2->LDS traffic of 128x128 per flop).
// PERSIST (lab build only, see launch_nt): the grid is one residency wave of workgroups and each walks virtual blocks b,
// b + grid, b + 2 grid, ...; while a workgroup stores tile j it already has the first ring stages of tile j+1 in flight.
// F16: operands (and a 16-bit output / residual) are IEEE fp16 instead of bf16 -- the trainable landmark CNN's plan
// (landmark_train.py); instantiated for the plain, activation and fp32 epilogues on 128x128 tiles only.
template <int EPI, int WM, int BK, int WN = 2, bool PERSIST = false, bool F16 = false, int MB = 4>
__global__ __launch_bounds__(WM * WN * 64, (WN == 6) ? 3 : ((WN == 4) ? 4 : ((WM == 2) ? (BK == 32 ? 3 : 2) : (BK == 32 ? 4 : 2)))) void gemm_nt_kernel(NTArgs p) {
  static_assert(!F16 || EPI == EPI_BF16 || EPI == EPI_BF16_ACT || EPI == EPI_F32, "fp16 operands: plain / activation / fp32 epilogues only");
  auto PK2 = [](float lo, float hi) { return F16 ? pack_h2(lo, hi) : pack_bf2(lo, hi); };
  auto CV1 = [](float x) { return F16 ? f2h(x) : f2bf(x); };
  auto LD1 = [](bf16_t h) { return F16 ? h2f(h) : bf2f(h); };
  const DropCfg drop = drop_resolve(p.drop);
  static_assert(MB == 4 || (WM == 2 && WN == 2 && BK == 64 && !PERSIST), "tall tiles: 2 x 2 waves on 64-deep stages only");
  constexpr int RW = MB * 16;                         // rows per wave: MB 16-row MFMA blocks (4: 128-row tiles at WM = 2; 5: 160-row tiles)
  constexpr int THREADS = WM * WN * 64, BMT = WM * RW, BN = WN * 64;
  constexpr bool PIN32 = (WM == 2);                   // (the 8-wave variants run at the 128-register cap: pinning spills there)
  constexpr int CPR = BK / 8;                         // 16-byte chunks per LDS row
  constexpr int ROWB = BK * 2;                        // bytes per LDS row
  constexpr int STAGE = (BMT + BN) * ROWB;
  constexpr int NSTG = (BK == 32) ? 3 : 2;
  constexpr int NA = (BMT * CPR + THREADS - 1) / THREADS;   // 16-byte A chunks per thread per stage (the last round may be partial:
  constexpr bool RAGGED_A = (BMT * CPR) % THREADS != 0;     // 128x384 tiles, 12 waves: 1024 chunks on 768 threads)
  constexpr int NB = (BN * CPR) / THREADS;            // 16-byte B chunks per thread per stage
  static_assert((BN * CPR) % THREADS == 0, "B chunks must divide evenly");
  static_assert(!RAGGED_A || NSTG == 2, "a partial A round is only counted right by the draining 2-stage ring");
  constexpr int NMAX = NA > NB ? NA : NB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTG * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int n_vb = PERSIST ? p.n_tiles : (int)gridDim.x;     // virtual blocks = output tiles (per k-split)
  int m0, n0;
  const int kbeg = blockIdx.z * p.klen;
  const int kend = min(p.K, kbeg + p.klen);
  const int nk = (kend - kbeg) / BK;
#ifdef LAFS_ABLATE
  // phase-lock experiment: hold back every second round-robin slot of a CU by (dbg >> 20) & 15 sleeps of ~4 us
  if ((((blockIdx.x >> 3) >> 5) & 1) && ((p.dbg >> 20) & 15)) {
    for (int z = 0; z < ((p.dbg >> 20) & 15); ++z) __builtin_amdgcn_s_sleep(127);
  }
#endif

  // ---- async global -> LDS staging (LDS-DMA).  The LDS image is lane-linear (wave base + lane*16), so the XOR
  // swizzle is applied to the SOURCE column instead. ----
  struct Src { const bf16_t* a[NA]; const bf16_t* b[NB]; int m0, n0; };
  int ldsoff[NMAX];
  auto locate = [&](int vb, Src& o) {               // operand pointers of virtual block vb
    const int tile = xcd_tile(vb, n_vb);
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    o.m0 = tm * BMT; o.n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      const int q = i * THREADS + tid, row = q / CPR, ch = (q % CPR) ^ nt_swzk<BK>(row);
      if (i < NA) o.a[i] = p.A + (size_t)min(o.m0 + min(row, BMT - 1), p.M - 1) * p.lda + kbeg + ch * 8;
      if (i < NB) o.b[i] = p.B + (size_t)min(o.n0 + nt_perm<EpiTraits<EPI>::VPL>(row), p.N - 1) * p.ldb + kbeg + ch * 8;
    }
  };
  auto issue = [&](const Src& o, int t) {
    if DBG(p, 524288) return;                        // ablation: no operand loads (the epilogue works on whatever LDS holds)
    unsigned char* st = smem + (t % NSTG) * STAGE;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      if (i < NA && (!RAGGED_A || i * THREADS + (int)(threadIdx.x & ~63u) < BMT * CPR)) glds16(o.a[i] + t * BK, st + ldsoff[i]);   // (wave-uniform)
      if (i < NB) glds16(o.b[i] + t * BK, st + BMT * ROWB + ldsoff[i]);
    }
  };
  Src src;

  f32x4_t acc[4][MB];                                // [j: column group][i: row tile]

  const int frow = lane & 15, fq = lane >> 4;
  int arow[MB], brow[4];

  auto compute = [&](int t) {
    if DBG(p, 32) return;
    const unsigned char* st = smem + (t % NSTG) * STAGE;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8_t fa[MB], fb[4];
#pragma unroll
      for (int i = 0; i < (MB > 4 ? MB : 4); ++i) {
        if (i < MB) fa[i] = *reinterpret_cast<const bf16x8_t*>(st + arow[i] * ROWB + (((kk * 4 + fq) ^ nt_swzk<BK>(arow[i])) << 4));
        if (i < 4) fb[i] = *reinterpret_cast<const bf16x8_t*>(st + BMT * ROWB + brow[i] * ROWB + (((kk * 4 + fq) ^ nt_swzk<BK>(brow[i])) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[j][i] = F16 ? mfma16_f16(fb[j], fa[i], acc[j][i]) : mfma16(fb[j], fa[i], acc[j][i]);
    }
    // 64-deep stages at two waves per SIMD (the long-K GEMMs of the trunk): pin the 16 fragment reads ahead of the 32 MFMAs that
    // consume them -- left alone, hipcc issues them in batches of 2-6 behind s_waitcnt lgkmcnt(0) and exposes the LDS latency four or
    // five times per stage (the same finding as in gemm_kres.hip)
    if constexpr (BK == 64 && WM == 2 && WN == 2) {
      if (!DBG(p, 32)) {                                 // 2 (MB + 4) reads, 8 MB MFMAs per stage
        __builtin_amdgcn_sched_group_barrier(0x100, MB + 4, 0);
#pragma unroll
        for (int x = 0; x < MB + 4; ++x) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * MB - 2 * (MB + 4), 0);
      }
    } else if constexpr (BK == 32 && WN == 2 && PIN32) {
      if (!DBG(p, 32)) {                               // 32-deep stages: 8 reads, 16 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
    }
  };

  bool primed = false;                               // the first ring stages of this tile were issued before the previous epilogue
  for (int vb = blockIdx.x; vb < n_vb; vb += (int)gridDim.x) {
#ifdef LAFS_ABLATE
  const int stamp_id = vb;
  STAMP(0);
  if (p.stamps && threadIdx.x == 0) {
    p.stamps[(size_t)(stamp_id + n_vb * blockIdx.z) * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
    p.stamps[(size_t)(stamp_id + n_vb * blockIdx.z) * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
  }
#endif
  {
    // k-loop addressing is re-derived per tile from an opaque copy of the thread id: hoisted out of the loop it would stay live
    // across the epilogue, which has no registers to spare at 4 waves per SIMD
    int t_ = threadIdx.x;
    if (PERSIST) asm volatile("" : "+v"(t_));
    const int w_ = t_ >> 6, l_ = t_ & 63;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) ldsoff[i] = (i * THREADS + w_ * 64) * 16;       // wave-uniform
#pragma unroll
    for (int i = 0; i < (MB > 4 ? MB : 4); ++i) {
      if (i < MB) arow[i] = (w_ / WN) * RW + i * 16 + (l_ & 15);
      if (i < 4) brow[i] = (w_ % WN) * 64 + i * 16 + (l_ & 15);
    }
  }
  if (!PERSIST || !primed) locate(vb, src);
  m0 = src.m0; n0 = src.n0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (NSTG == 3) {
    if (!primed) {
      if (nk > 0) issue(src, 0);
      if (nk > 1) issue(src, 1);
    }
    for (int t = 0; t < nk; ++t) {
      // stage t has landed once at most the NA+NB loads of stage t+1 are still in flight (loads retire in order).  The first
      // step of a primed tile drains everything instead: the previous tile's stores are younger than its two stages.
      if (t + 1 < nk && !(PERSIST && primed && t == 0)) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                  // everyone's part of stage t landed; stage (t-1)%3 is free again
      if (t + 2 < nk) issue(src, t + 2);
      compute(t);
    }
  } else {
    if (!primed && nk > 0) issue(src, 0);
    for (int t = 0; t < nk; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // stage t landed everywhere; the other buffer is free again
      if (t + 1 < nk) issue(src, t + 1);
      compute(t);
    }
  }
  if (PERSIST) {
    primed = vb + (int)gridDim.x < n_vb;
    if (primed) {
      locate(vb + gridDim.x, src);                   // (m0, n0 keep this tile's origin for the epilogue)
      __builtin_amdgcn_s_barrier();                  // every wave is out of the ring before the next tile's stages land in it
      if (nk > 0) issue(src, 0);
      if (NSTG == 3 && nk > 1) issue(src, 1);
    }
  }

  STAMP(1);
  // (WHOLE: the tile lies inside the matrix.  Without guards there are no branches around the epilogue's loads and stores; with
  // them every guarded piece is its own basic block that hipcc opens with s_waitcnt vmcnt(0) -- the bias / prefetched operand
  // registers came from loads -- so the one-row-ahead prefetch below was waited for right where it was issued, and every store sat
  // out the store before it)
  auto epilogue = [&](auto whole_c) __attribute__((always_inline)) {
  constexpr bool WHOLE = decltype(whole_c)::value;
  // ---- epilogue: lane owns rows m = m0 + wr*64 + i*16 + (lane&15) and, per row, NG groups of VPL consecutive columns:
  // group q starts at n0 + wc*64 + q*4*VPL + fq*VPL; register e = j*4 + r of the row is element e % VPL of group e / VPL.
  constexpr int VPL = EpiTraits<EPI>::VPL, NG = 16 / VPL;
  const int ncol0 = n0 + wc * 64 + fq * VPL;
  if DBG(p, 16) {                                    // ablation: keep the accumulators live, store nothing
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MB; ++i) asm volatile("" :: "v"(acc[j][i]));
    return;
  }
  float bias[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bias[e] = 0.f;
  if (p.bias != nullptr && EPI != EPI_ATOMIC_F32 && EPI != EPI_DGELU_BF16) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = ncol0 + (e / VPL) * 4 * VPL + (e % VPL);
      if (WHOLE || n < p.N) bias[e] = p.bias[n];
    }
  }
  // Operands the epilogue reads per output piece (the fp32 residual of RESID_F32, the pre-activation of DGELU_BF16) are
  // fetched one ROW AHEAD of their use: resid may alias C, so the compiler cannot hoist these loads above the previous row's
  // stores by itself, and a load -> use -> store chain per piece exposes a full memory round trip 8-16 times per tile.
  constexpr bool PRE = (EPI == EPI_RESID_F32 || EPI == EPI_DGELU_BF16);
  uint4 pre[PRE ? NG : 1], cur[PRE ? NG : 1];
  auto prefetch = [&](int i) {
    if (!PRE || i >= MB) return;
    const int mi = m0 + wr * RW + i * 16 + frow;
    if (!WHOLE && mi >= p.M) return;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int n = ncol0 + q * 4 * VPL;
      if (!WHOLE && n + VPL > p.N) continue;
      if (EPI == EPI_RESID_F32) pre[q] = *reinterpret_cast<const uint4*>(p.resid + (size_t)mi * p.ldr + n);
      else pre[q] = *reinterpret_cast<const uint4*>(p.aux + (size_t)mi * p.ldaux + n);
    }
  };
  prefetch(0);
  float scv[MB];                                        // DropPath scales of the MB rows: two dependent loads each, issued together
#pragma unroll
  for (int i = 0; i < MB; ++i) scv[i] = 1.0f;
  if (EPI == EPI_RESID_F32 && p.seq_scale != nullptr) {
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int mi = m0 + wr * RW + i * 16 + frow;
      if (WHOLE || mi < p.M) scv[i] = p.seq_scale[p.row2seq[mi]];
    }
  }
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int m = m0 + wr * RW + i * 16 + frow;
    if (PRE) {
#pragma unroll
      for (int q = 0; q < NG; ++q) cur[q] = pre[q];
      prefetch(i + 1);
    }
    if (!WHOLE && m >= p.M) continue;
    float v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[j][i][r] + bias[j * 4 + r];
    float sc = 1.0f;
    const bool ntst = DBG(p, 256) != 0;              // streaming stores: A/B experiment only (no gain in a real layer chain)
    size_t orow = (size_t)m;
    int tpos = 0;
    if (EPI == EPI_RESID_F32) sc = scv[i];
    if (EPI == EPI_EMBED_F32) {
      const int b = m / p.npatch;
      tpos = m - b * p.npatch + 1;
      orow = (size_t)m + b + 1;                         // one cls row in front of every sequence
    }
    // BF16_GELU writes two tensors: all pieces of u first, then all pieces of GELU(u), so that the 64-byte halves of a
    // 128-byte line leave the CU back to back and merge into full-line writes (fc1 at C2: 150 -> ~125 us; debug flag 32768
    // restores the piece-by-piece order for A/B runs)
    constexpr int NPASS = (EPI == EPI_BF16_GELU) ? 2 : 1;
    uint4 gpk[NG];                                        // (BF16_GELU saving gelu'(u): gelu(u) of the row's pieces, packed)
#pragma unroll
    for (int q = 0; q < NG; ++q) gpk[q] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int n = ncol0 + q * 4 * VPL;
      if (!WHOLE && n >= p.N) continue;
      const bool full = WHOLE || (n + VPL <= p.N);
      float* w = v + q * VPL;
#ifdef LAFS_ABLATE
      // pacing experiment: (dbg >> 24) & 31 sleeps of 128 clocks before every store group, so that a workgroup's stores enter
      // the CU's in-order memory pipeline spread out instead of as one burst in front of the neighbour workgroup's loads
      for (int z = 0; z < ((p.dbg >> 24) & 31); ++z) __builtin_amdgcn_s_sleep(2);
#endif
      const bool legacy = DBG(p, 32768) != 0;
      const bool do_first = legacy ? (pass == 0) : (pass == 0), do_second = legacy ? (pass == 0) : (pass == NPASS - 1);
      if (legacy && pass > 0) continue;
      if (EPI == EPI_BF16_ACT) {                          // 1x1 convolution of the landmark CNN: + residual (bf16), activation
        if (p.aux != nullptr) {
          const bf16_t* ax = p.aux + (size_t)m * p.ldaux + n;
#pragma unroll
          for (int e = 0; e < VPL; ++e) if (full || n + e < p.N) w[e] += LD1(ax[e]);
        }
#pragma unroll
        for (int e = 0; e < VPL; ++e) w[e] = act_f(w[e], p.act);
      }
      if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_DGELU_BF16 || EPI == EPI_BF16_ACT) {   // VPL == 8: one 16-byte store
        if (EPI == EPI_DGELU_BF16) {
          const bf16_t* ax = p.aux + (size_t)m * p.ldaux + n;
          if (full) {
            const uint4 a4 = cur[q];
            if (p.act == LAFS_GELU_SAVE_GRAD) {                  // aux already holds gelu'(u)
              w[0] *= bf_lo(a4.x); w[1] *= bf_hi(a4.x); w[2] *= bf_lo(a4.y); w[3] *= bf_hi(a4.y);
              w[4] *= bf_lo(a4.z); w[5] *= bf_hi(a4.z); w[6] *= bf_lo(a4.w); w[7] *= bf_hi(a4.w);
            } else {
              w[0] *= gelu_grad_f(bf_lo(a4.x)); w[1] *= gelu_grad_f(bf_hi(a4.x)); w[2] *= gelu_grad_f(bf_lo(a4.y)); w[3] *= gelu_grad_f(bf_hi(a4.y));
              w[4] *= gelu_grad_f(bf_lo(a4.z)); w[5] *= gelu_grad_f(bf_hi(a4.z)); w[6] *= gelu_grad_f(bf_lo(a4.w)); w[7] *= gelu_grad_f(bf_hi(a4.w));
            }
          } else {
#pragma unroll
            for (int e = 0; e < VPL; ++e) if (n + e < p.N) w[e] *= (p.act == LAFS_GELU_SAVE_GRAD) ? bf2f(ax[e]) : gelu_grad_f(bf2f(ax[e]));
          }
          if (drop.thresh) {                                     // d(dropout(gelu(u))): the forward's mask, regenerated
#pragma unroll
            for (int e = 0; e < VPL; ++e) w[e] *= drop_mult(drop, (unsigned)m * (unsigned)p.N + (unsigned)(n + e));
          }
        }
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n;
#ifdef LAFS_ABLATE
        // ablation (timing only, values land in the wrong places): every store instruction covers 8 full 128-byte lines
        // (8 rows x 64 columns of the wave's 64x64 tile) instead of 16 half lines
        const bool fullline = DBG(p, 2097152) && VPL == 8;
        const size_t flrow = (size_t)(m0 + wr * RW + (i * NG + q) * 8 + (lane >> 3));
        if (fullline) c = reinterpret_cast<bf16_t*>(p.C) + min(flrow, (size_t)p.M - 1) * p.ldc + n0 + wc * 64 + (lane & 7) * 8;
#endif
        // experiment (debug flag 16384): u and GELU(u) interleaved in 64-byte pieces of ONE [M, 2N] buffer, so the two stores
        // of a lane group complete a 128-byte line
        if (EPI == EPI_BF16_GELU && DBG(p, 16384)) c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc * 2 + (n >> 5) * 64 + (n & 31);
        if (EPI == EPI_BF16_GELU && (p.C == nullptr || !do_first)) {
          // forward-only pass (teacher): the pre-activation is not needed, only GELU(u) is written; second pass: already stored
        } else if (EPI == EPI_BF16_GELU && p.act == LAFS_GELU_SAVE_GRAD) {        // the first tensor is gelu'(u), not u
          if (full) {
            // gelu'(u) and gelu(u) from ONE v_rcp + v_exp (gelu_both_f: bit-identical to gelu_grad_f / gelu_f evaluated apart, half the
            // quarter-rate instructions of this VALU-bound epilogue); gelu(u), with its dropout mask, waits packed for the second pass
            float d[8], g[8];                               // (VPL == 8 whenever this branch is live; fixed size keeps the VPL == 4 instantiations in bounds)
#pragma unroll
            for (int e = 0; e < VPL; ++e) {
              gelu_both_f(w[e], g[e], d[e]);
              if (drop.thresh) g[e] *= drop_mult(drop, (unsigned)m * (unsigned)p.N + (unsigned)(n + e));
            }
            st16(c, pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3]), pack_bf2(d[4], d[5]), pack_bf2(d[6], d[7]), ntst);
            gpk[q] = make_uint4(pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]), pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7]));
          } else {
#pragma unroll
            for (int e = 0; e < VPL; ++e) if (n + e < p.N) c[e] = f2bf(gelu_grad_f(w[e]));
          }
        } else if (full) {
          st16(c, PK2(w[0], w[1]), PK2(w[2], w[3]), PK2(w[4], w[5]), PK2(w[6], w[7]), ntst);
        } else {
#pragma unroll
          for (int e = 0; e < VPL; ++e) if (n + e < p.N) c[e] = CV1(w[e]);
        }
        if (EPI == EPI_BF16_GELU && !DBG(p, 64) && do_second) {
          bf16_t* c2 = reinterpret_cast<bf16_t*>(p.C2) + (size_t)m * p.ldc2 + n;
#ifdef LAFS_ABLATE
          if (fullline) c2 = reinterpret_cast<bf16_t*>(p.C2) + min(flrow, (size_t)p.M - 1) * p.ldc2 + n0 + wc * 64 + (lane & 7) * 8;
#endif
          if DBG(p, 16384) c2 = c + 32;
          if (full && DBG(p, 128)) {
            *reinterpret_cast<uint4*>(c2) = make_uint4(pack_bf2(w[0], w[1]), pack_bf2(w[2], w[3]), pack_bf2(w[4], w[5]), pack_bf2(w[6], w[7]));
          } else if (full && p.C != nullptr && p.act == LAFS_GELU_SAVE_GRAD) {          // evaluated with gelu'(u) in the first pass
            st16(c2, gpk[q].x, gpk[q].y, gpk[q].z, gpk[q].w, ntst);
          } else if (drop.thresh && full) {              // dropout(gelu(u)): one 16-byte store like the plain form
            float g[8];
#pragma unroll
            for (int e = 0; e < VPL; ++e) g[e] = gelu_f(w[e]) * drop_mult(drop, (unsigned)m * (unsigned)p.N + (unsigned)(n + e));
            st16(c2, pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]), pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7]), ntst);
          } else if (drop.thresh) {
#pragma unroll
            for (int e = 0; e < VPL; ++e)
              if (n + e < p.N) c2[e] = f2bf(gelu_f(w[e]) * drop_mult(drop, (unsigned)m * (unsigned)p.N + (unsigned)(n + e)));
          } else if (full) {
            st16(c2, pack_bf2(gelu_f(w[0]), gelu_f(w[1])), pack_bf2(gelu_f(w[2]), gelu_f(w[3])),
                 pack_bf2(gelu_f(w[4]), gelu_f(w[5])), pack_bf2(gelu_f(w[6]), gelu_f(w[7])), ntst);
          } else {
#pragma unroll
            for (int e = 0; e < VPL; ++e) if (n + e < p.N) c2[e] = f2bf(gelu_f(w[e]));
          }
        }
      } else if (EPI == EPI_ATOMIC_F32) {
        float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
#pragma unroll
        for (int e = 0; e < VPL; ++e) if (full || n + e < p.N) atomicAdd(c + e, w[e]);
      } else {                                                                        // VPL == 4: one float4
        if (EPI == EPI_RESID_F32) {
          const float* rs = p.resid + (size_t)m * p.ldr + n;
          if (drop.thresh) {
#pragma unroll
            for (int e = 0; e < VPL; ++e) w[e] *= drop_mult(drop, (unsigned)m * (unsigned)p.N + (unsigned)(n + e));
          }
          if (full) {
            const float4 r4 = make_float4(__uint_as_float(cur[q].x), __uint_as_float(cur[q].y), __uint_as_float(cur[q].z),
                                          __uint_as_float(cur[q].w));
            w[0] = r4.x + sc * w[0]; w[1] = r4.y + sc * w[1]; w[2] = r4.z + sc * w[2]; w[3] = r4.w + sc * w[3];
          } else {
#pragma unroll
            for (int e = 0; e < VPL; ++e) if (n + e < p.N) w[e] = rs[e] + sc * w[e];
          }
        } else if (EPI == EPI_EMBED_F32) {
          const float* ps = p.pos + (size_t)tpos * p.N + n;
#pragma unroll
          for (int e = 0; e < VPL; ++e) if (full || n + e < p.N) w[e] += ps[e];
        }
        // F32 with the K axis split over blockIdx.z: every slice stores its own [M][ldc] image (lafs_sum_slices folds them)
        float* c = reinterpret_cast<float*>(p.C) + (EPI == EPI_F32 ? (size_t)blockIdx.z * p.M * p.ldc : 0) + orow * p.ldc + n;
        if (full) {
          st16f(c, w[0], w[1], w[2], w[3], ntst);
        } else {
#pragma unroll
          for (int e = 0; e < VPL; ++e) if (n + e < p.N) c[e] = w[e];
        }
      }
    }
  }
  };
  // (plain / residual / GELU' epilogues only: the others are rare or already store back to back, and two copies of them cost registers)
  if ((EPI == EPI_BF16 || EPI == EPI_RESID_F32 || EPI == EPI_DGELU_BF16) && m0 + BMT <= p.M && n0 + BN <= p.N) epilogue(std::true_type());
  else epilogue(std::false_type());
#ifdef LAFS_ABLATE
  STAMP(2);
  if (p.stamps && !PERSIST) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  STAMP(3);
#endif
  if (!PERSIST) break;
  if (primed) locate(vb + gridDim.x, src);          // recomputed rather than kept live across the epilogue
  }                                                  // virtual blocks
}

// ------------------------------------------------------------------------------------------------ TN
struct TNArgs {
  const bf16_t* A; const bf16_t* B; float* C; float* colsum;
  int M, N1, N2, lda, ldb, ldc, mlen, mode, splits, tiles;
  long part_stride;                                // mode 2: C is [n_xcd][...] partial sums, one image per XCD
};

// Reduction rows per pipeline stage: KB = 32 (3-stage ring) or 64 (2-stage ring, half as many barriers per row).
// One 128-column operand panel of a stage is KB rows x 256 B.

// XOR applied to the 8-byte unit index (0..31) of a 256-byte row so that ds_read_b64_tr_b16 of 4 consecutive rows x
// 4 units is conflict-free for both 16-lane groups sharing an LDS cycle (rows r..r+3 and r+8..r+11).
__device__ __forceinline__ int tn_f(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 2; }

// Output tile (64*WM) x (64*WN), one 64x64 block of 4x4 MFMA tiles per wave.  Operands are staged as 128-column panels
// (WM/2 panels of A, WN/2 of B per stage).  The product path uses 2x2 (see launch_tn for the measured comparison).
// CS = false compiles the bias-gradient (column-sum) accumulators out: 16 registers the 16-wave 256x256 variant needs back.
template <int WM, int WN, int KB, bool CS = true>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? (KB == 64 ? 2 : 3) : (WM * WN == 8 ? 4 : 1)) void gemm_tn_kernel(TNArgs p) {
  constexpr int NTH = 64 * WM * WN;
  constexpr int PA = WM / 2, PB = WN / 2;
  constexpr int TN_BM = KB, TN_PANEL = KB * 256, NS = (KB == 32) ? 3 : 2;
  constexpr int STAGE = (PA + PB) * TN_PANEL;
  constexpr int NCH = (PA + PB) * TN_BM * 16;        // 16-byte chunks per stage
  constexpr int NI = NCH / NTH;                      // LDS-DMA loads per thread and stage
  static_assert(WM % 2 == 0 && WN % 2 == 0 && NCH % NTH == 0, "tile shape");
  constexpr int TN1 = 64 * WM, TN2 = 64 * WN;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // 1-D grid.  All output tiles of one M-slice run on the SAME XCD (blocks b, b+8, ... share an L2), back to back, so
  // the slice's rows are fetched from HBM once and re-read by the other tiles out of that L2.
  int zslice, tile;
  if (p.splits % 8 == 0) {
    const int j = blockIdx.x >> 3;
    zslice = (blockIdx.x & 7) + 8 * (j / p.tiles);
    tile = j % p.tiles;
  } else {
    zslice = blockIdx.x / p.tiles;
    tile = blockIdx.x % p.tiles;
  }
  const int tiles_n2 = (p.N2 + TN2 - 1) / TN2;
  const int n1_0 = (tile / tiles_n2) * TN1, n2_0 = (tile % tiles_n2) * TN2;
  const int mbeg = zslice * p.mlen;
  const int mend = min(p.M, mbeg + p.mlen);
  const int nk = (mend - mbeg + TN_BM - 1) / TN_BM;
  const bool do_colsum = CS && (p.colsum != nullptr) && n2_0 == 0 && wn == 0;

  // LDS-DMA staging: the image is lane-linear (chunk q of the stage lands at byte 16*q), so the unit swizzle goes on the
  // source column.  Rows past the end of the slice are clamped to a valid row here and zeroed in LDS before use.
  // Nothing about a chunk is kept in registers: its panel / row / source column are recomputed from the thread index at issue
  // time (a dozen integer operations per 1 KiB DMA), which frees ~20 VGPRs for occupancy.
  constexpr int CHP = TN_BM * 16;                    // chunks per panel
  auto issue = [&](int t) {
    unsigned char* st = smem + (t % NS) * STAGE;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = i * NTH + tid, panel = q / CHP, within = q % CHP;
      const int row = within >> 4, ch = (within & 15) ^ (tn_f(row) >> 1);
      const int m = min(mbeg + t * TN_BM + row, mend - 1);
      const bf16_t* src;
      if (panel < PA) {
        const int c = n1_0 + panel * 128 + ch * 8;
        src = p.A + (size_t)m * p.lda + (c < p.N1 ? c : 0);
      } else {
        const int c = n2_0 + (panel - PA) * 128 + ch * 8;
        src = p.B + (size_t)m * p.ldb + (c < p.N2 ? c : 0);
      }
      glds16(src, st + (i * NTH + wave * 64) * 16);
    }
  };

  f32x4_t acc[4][4];
  float cs[4] = {0.f, 0.f, 0.f, 0.f};                // bias gradient: this lane's share of the column sums of A (fragment a)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;

  const int g = lane >> 4, pl = lane & 15;
  const int r0 = g * 8 + (pl >> 2), r1 = r0 + 4;
  // Fragment x (0..3) of a wave reads unit (base | x*4) ^ tn_f(row): the x field occupies its own two bits, so the offset of
  // fragment x is the offset of fragment 0 XOR (x << 5) -- four base offsets instead of sixteen live registers.
  const int ua = (wm & 1) * 16 + (pl & 3), ub = (wn & 1) * 16 + (pl & 3);
  const int pa = (wm >> 1) * TN_PANEL, pb = (PA + (wn >> 1)) * TN_PANEL;
  const int oa0 = r0 * 256 + ((ua ^ tn_f(r0)) << 3), oa1 = r1 * 256 + ((ua ^ tn_f(r1)) << 3);
  const int ob0 = r0 * 256 + ((ub ^ tn_f(r0)) << 3), ob1 = r1 * 256 + ((ub ^ tn_f(r1)) << 3);
  if (nk > 0) issue(0);
  if (NS == 3 && nk > 1) issue(1);
  for (int t = 0; t < nk; ++t) {
    // NS == 3: stage t+1 may still be in flight; NS == 2: only stage t is outstanding here
    if (NS == 3 && t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + NS - 1 < nk) issue(t + NS - 1);
    unsigned char* st = smem + (t % NS) * STAGE;
    const int valid = mend - (mbeg + t * TN_BM);
    if (valid < TN_BM) {                              // ragged tail: clear the rows that were clamped
      for (int q = tid; q < NCH; q += NTH) {
        const int row = (q >> 4) % TN_BM;
        if (row >= valid) *reinterpret_cast<uint4*>(st + q * 16) = make_uint4(0, 0, 0, 0);
      }
      __syncthreads();
    }
#pragma unroll
    for (int kk = 0; kk < KB / 32; ++kk) {            // one k = 32 MFMA step per 32 staged rows
      const unsigned char* sk = st + kk * 32 * 256;
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        s16x8_t va = __builtin_shufflevector(lds_read_tr16(sk + pa + (oa0 ^ (x << 5))), lds_read_tr16(sk + pa + (oa1 ^ (x << 5))),
                                             0, 1, 2, 3, 4, 5, 6, 7);
        s16x8_t vb = __builtin_shufflevector(lds_read_tr16(sk + pb + (ob0 ^ (x << 5))), lds_read_tr16(sk + pb + (ob1 ^ (x << 5))),
                                             0, 1, 2, 3, 4, 5, 6, 7);
        fa[x] = __builtin_bit_cast(bf16x8_t, va);
        fb[x] = __builtin_bit_cast(bf16x8_t, vb);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
      if (CS && do_colsum) {                          // column sums of A (bias gradient): a lane holds 8 k-values of one column
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const uint4 w = __builtin_bit_cast(uint4, fa[a]);
          cs[a] += (bf_lo(w.x) + bf_hi(w.x)) + (bf_lo(w.y) + bf_hi(w.y)) + (bf_lo(w.z) + bf_hi(w.z)) + (bf_lo(w.w) + bf_hi(w.w));
        }
      }
    }
  }
  // mode 2: every XCD accumulates into its OWN image of C, so all adds to one address come from CUs behind the same L2
  // and can be executed there (workgroup-scope atomics: no write-through to the fabric); lafs_reduce_partials folds the
  // images afterwards.  XCC_ID is read from the hardware register, so this does not depend on the dispatch order.
  float* Cx = p.C;
  if (p.mode == 2) Cx += (size_t)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15) * p.part_stride;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n1 = n1_0 + wm * 64 + a * 16 + g * 4 + r;
      if (n1 >= p.N1) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int n2 = n2_0 + wn * 64 + b * 16 + pl;
        if (n2 < p.N2) {
          float* dst = Cx + (size_t)n1 * p.ldc + n2;
          if (p.mode == 0) atomicAdd(dst, acc[a][b][r]);
          else if (p.mode == 2) __hip_atomic_fetch_add(dst, acc[a][b][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else *dst = acc[a][b][r];                                   // timing experiment only (lafs_debug_set)
        }
      }
    }
  if (CS && do_colsum) {                              // lanes l, l^16, l^32, l^48 hold the four k-slices of column (l & 15)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float v = cs[a];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n1 = n1_0 + wm * 64 + a * 16 + pl;
      if (g == 0 && n1 < p.N1) atomicAdd(p.colsum + n1, v);
    }
  }
}

// out[i] += sum_x part[x][i]; part[x][i] = 0   (the images are left zeroed for the next accumulation)
__global__ __launch_bounds__(256) void reduce_partials_kernel(float* __restrict__ part, long stride, int n_part, long n4,
                                                             float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 acc = reinterpret_cast<const float4*>(out)[i];
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int x = 0; x < n_part; ++x) {
    float4* src = reinterpret_cast<float4*>(part + x * stride) + i;
    const float4 v = *src;
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    *src = z;
  }
  reinterpret_cast<float4*>(out)[i] = acc;
}

// out[i] = sum_x part[x][i]   (slice images of a K-split F32 GEMM)
__global__ __launch_bounds__(256) void sum_slices_kernel(const float* __restrict__ part, long stride, int n_part, long n4,
                                                        float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int x = 0; x < n_part; ++x) {
    const float4 v = reinterpret_cast<const float4*>(part + x * stride)[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  reinterpret_cast<float4*>(out)[i] = acc;
}

int g_debug_flags = 0;

// 128x384 / 12-wave tiles (the whole N per workgroup) when the tiles fit one round of one workgroup per CU: see launch_nt
// 160-row tiles (MB = 5) instead of 128-row ones where they save a round of workgroup slots.  The tiled kernel keeps two workgroups per
// CU, so a launch costs whole rounds of 512 tiles: tools/lab/t_quant.py measures the staircase (fc1 input gradient of ViT-S, N = 384,
// K = 1536: 1023 tiles 50.8 us, 1035 tiles 63.4 us; Part-fViT N = 768, K = 2048: 2046 tiles 122 us, 2052 tiles 135 us).  A 160-row
// tile costs ~1.2x a 128-row one (1.25x the MFMAs, 1.125x the staging): taken when rounds(160) x 1.2 < rounds(128).
// LAFS_OPT_NT_TALL = 0 switches it off (A/B).
bool tall_tile_shape(const lafs_ctx* cx, int M, int N, int splits) {
  if (!lafs_ctx_opt(cx, LAFS_OPT_NT_TALL) || splits != 1) return false;
  const long tn = ceil_div(N, 128);
  const long r128 = ceil_div((long)ceil_div(M, 128) * tn, 512L), r160 = ceil_div((long)ceil_div(M, 160) * tn, 512L);
  return 6 * r160 < 5 * r128;
}

bool wide_tile_shape(const lafs_ctx* cx, int M, int N, int splits) {
  const int mt = ceil_div(M, 128);
  return lafs_ctx_opt(cx, LAFS_OPT_NT_WIDE) && splits == 1 && N % 384 == 0 && mt * (N / 384) >= 160 && mt * (N / 384) <= 256;
}

template <int EPI>
int launch_nt(const NTArgs& a, int splits, const lafs_ctx* cx, hipStream_t s) {
  // Shape heuristics from tools/bench_kernels.py on MI355X (ViT-S/B shapes):
  //  * wide outputs (N >= 1024) on many rows: 256x128 tiles (less L2->LDS traffic per flop, 16 resident waves/CU);
  //  * long reductions (K >= 640): 64-deep stages (full 128-byte lines per row piece) in a 2-stage ring, 128x128 tiles.
  const int tn = ceil_div(a.N, 128);
  const long t2 = (long)ceil_div(a.M, 128) * tn * splits, t4 = (long)ceil_div(a.M, 256) * tn * splits;
  //  * 256x256 tiles (16 waves, one workgroup per CU, half the operand re-reads of 128x128) win 6-9 % on the isolated ViT-S
  //    fc1 forward / GELU' dgrad (141 -> 128 us, 138 -> 129 us) but LOSE in the real step (21.3 -> 21.5 ms: a 16-wave
  //    workgroup owns the CU while the weight-gradient stream wants to share it); kept behind debug flag 65536.
  const bool wide256 = (g_debug_flags & 65536) != 0;
  if (wide256 && splits == 1 && a.N >= 1024 && a.N % 256 == 0 && a.M >= 4096 &&
      (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_DGELU_BF16)) {
    const unsigned t44 = (unsigned)(ceil_div(a.M, 256) * ceil_div(a.N, 256));
    if ((g_debug_flags & 8) && a.klen % 64 == 0) hipLaunchKernelGGL((gemm_nt_kernel<EPI, 4, 64, 4>), dim3(t44, 1, 1), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<EPI, 4, 32, 4>), dim3(t44, 1, 1), dim3(1024), 0, s, a);
    LAFS_LAUNCH_CHECK();
    return LAFS_OK;
  }
  if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_ACT || EPI == EPI_F32) {
    if (a.f16) {                                     // fp16 operands (landmark CNN training plan): 128x128 tiles, no K split
      const bool bk = (a.klen % 64 == 0) && a.klen >= 640;
      if (bk) hipLaunchKernelGGL((gemm_nt_kernel<EPI, 2, 64, 2, false, true>), dim3((unsigned)t2, 1, 1), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_nt_kernel<EPI, 2, 32, 2, false, true>), dim3((unsigned)t2, 1, 1), dim3(256), 0, s, a);
      LAFS_LAUNCH_CHECK();
      return LAFS_OK;
    }
  }
  bool bk64 = (a.klen % 64 == 0) && a.klen >= 640 && (splits == 1 || a.K % 64 == 0);    // K = 704 / 768 (ViT-B) included: 5-15 % over 32-deep stages
  int wm = (!bk64 && a.N >= 1024 && a.M >= 4096) ? 4 : 2;
  if (g_debug_flags & 2) wm = 2;
  if (g_debug_flags & 4) wm = 4;
  if (g_debug_flags & 8) bk64 = (a.klen % 64 == 0);
  if (g_debug_flags & 2) bk64 = bk64 && (g_debug_flags & 8);
#ifdef LAFS_ABLATE
  // Lab build only (debug flag 8388608): problems that need more than one residency wave of workgroups run persistently -- one
  // wave of workgroups walking the tiles, the next tile's first ring stages in flight during the epilogue.  Measured on the C2
  // shapes: within +-2 % of the one-tile-per-workgroup launch in isolation (fc1 135 vs 133 us) and 0.2 ms SLOWER per step
  // (18.45 vs 18.23 ms): resident workgroups keep the side streams' kernels off the CUs.  Not built into the product library.
  const long tiles = (wm == 4 ? t4 : t2);
  const int resident = 256 * ((wm == 4 || bk64) ? 2 : 3);
  if (splits == 1 && tiles > resident && (g_debug_flags & 8388608)) {
    NTArgs b = a;
    b.n_tiles = (int)tiles;
    const dim3 grid((unsigned)resident, 1, 1);
    if (wm == 4) {
      if (bk64) hipLaunchKernelGGL((gemm_nt_kernel<EPI, 4, 64, 2, true>), grid, dim3(512), 0, s, b);
      else hipLaunchKernelGGL((gemm_nt_kernel<EPI, 4, 32, 2, true>), grid, dim3(512), 0, s, b);
    } else {
      if (bk64) hipLaunchKernelGGL((gemm_nt_kernel<EPI, 2, 64, 2, true>), grid, dim3(256), 0, s, b);
      else hipLaunchKernelGGL((gemm_nt_kernel<EPI, 2, 32, 2, true>), grid, dim3(256), 0, s, b);
    }
    LAFS_LAUNCH_CHECK();
    return LAFS_OK;
  }
#endif
  // Long reductions onto N = 384 (fc2 forward, fc1 / qkv input gradients of ViT-S): 128x384 tiles, 12 waves -- the whole N per
  // workgroup, so the A rows are staged once instead of three times and a wave issues 5.3 instead of 8 LDS-DMA instructions per
  // 32 MFMAs (the staging cost is issue time in the wave)
  // ... when its tiles fit ONE round of one workgroup per CU: 197 tiles (teacher, M = 25216) run 12-17 % faster than on the
  // 128x128 kernel; 345 tiles (student: a second round of 89) are slower, and so is a split into whole rounds here + the rest on
  // the 128x128 kernel (fc2 forward 92-98 against 83-86 us; tools/lab/nt_variants.py).  LAFS_OPT_NT_WIDE = 0 switches it off (A/B).
  if constexpr (EPI == EPI_BF16 || EPI == EPI_RESID_F32) {
    const int mt = ceil_div(a.M, 128);
    if (bk64 && wide_tile_shape(cx, a.M, a.N, splits)) {
      hipLaunchKernelGGL((gemm_nt_kernel<EPI, 2, 64, 6>), dim3((unsigned)(mt * (a.N / 384)), 1, 1), dim3(768), 0, s, a);
      LAFS_LAUNCH_CHECK();
      return LAFS_OK;
    }
  }
  if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_RESID_F32 || EPI == EPI_DGELU_BF16) {
    if (bk64 && wm == 2 && tall_tile_shape(cx, a.M, a.N, splits)) {
      hipLaunchKernelGGL((gemm_nt_kernel<EPI, 2, 64, 2, false, false, 5>), dim3((unsigned)(ceil_div(a.M, 160) * tn), 1, 1), dim3(256), 0, s, a);
      LAFS_LAUNCH_CHECK();
      return LAFS_OK;
    }
  }
  if (wm == 4) {
    if (bk64) hipLaunchKernelGGL((gemm_nt_kernel<EPI, 4, 64>), dim3((unsigned)t4 / splits, 1, splits), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<EPI, 4, 32>), dim3((unsigned)t4 / splits, 1, splits), dim3(512),
                            (g_debug_flags & 4194304) ? 24576 : 0 /* lab: extra LDS -> one workgroup per CU */, s, a);
  } else {
    if (bk64) hipLaunchKernelGGL((gemm_nt_kernel<EPI, 2, 64>), dim3((unsigned)t2 / splits, 1, splits), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<EPI, 2, 32>), dim3((unsigned)t2 / splits, 1, splits), dim3(256), 0, s, a);
  }
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

// length of one K slice when K is split `splits` ways (a multiple of the stage depth; the last slice takes the remainder)
int ksplit_len(int K, int splits) {
  const int ksteps = K / 32;
  splits = splits < 1 ? 1 : (splits > ksteps ? ksteps : splits);
  int klen = ceil_div(ksteps, splits) * 32;
  if (K % 64 == 0 && klen >= 640) klen = (klen + 63) / 64 * 64;     // long slices: 64-deep stages (full-line DMA requests)
  return klen;
}

}  // namespace

extern "C" int lafs_gemm_nt_slices(int K, int splits) { return K >= 32 ? ceil_div(K, ksplit_len(K, splits)) : 1; }

extern "C" int lafs_gemm_nt_route(const lafs_gemm_nt_args* g) {
  if (g == nullptr) return 0;
  if (lafs_kres_eligible(g)) return 1;
  if (lafs_big_eligible(g)) return 5;               // wide long-K shapes: 192x256 tiles, one persistent workgroup per CU (gemm_big.hip)
  if (g->operand_f16) return 0;                     // fp16 operands (landmark CNN plan): launch_nt takes the 128x128 fp16 kernel first
  const int splits = (g->epilogue == LAFS_EPI_ATOMIC_F32 || g->epilogue == LAFS_EPI_F32) && g->splits > 1 ? g->splits : 1;
  // the tiled kernel's 128x384 form (launch_nt): plain / residual epilogue, 64-deep stages (K % 64 == 0, K >= 640), no K split
  const bool bk64 = g->K % 64 == 0 && g->K >= 640 && !(g_debug_flags & 2);
  if ((g->epilogue == LAFS_EPI_BF16 || g->epilogue == LAFS_EPI_RESID_F32) && bk64 && wide_tile_shape(g->ctx, g->M, g->N, splits)) return 3;
  // ... and its 160-row form where that saves a round of workgroup slots (plain, GELU, residual and GELU' epilogues)
  if ((g->epilogue == LAFS_EPI_BF16 || g->epilogue == LAFS_EPI_BF16_GELU || g->epilogue == LAFS_EPI_RESID_F32 || g->epilogue == LAFS_EPI_DGELU_BF16) &&
      bk64 && tall_tile_shape(g->ctx, g->M, g->N, splits))
    return 4;
  return 0;
}

extern "C" int lafs_gemm_nt(const lafs_gemm_nt_args* g, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(g != nullptr && g->A && g->B && (g->C || (g->epilogue == LAFS_EPI_BF16_GELU && g->C2)), "null operand");
  LAFS_CHECK_ARG(g->M > 0 && g->N > 0 && g->K > 0, "empty problem");
  LAFS_CHECK_ARG(g->K % 32 == 0, "K must be a multiple of 32");
  LAFS_CHECK_ARG(g->lda % 8 == 0 && g->ldb % 8 == 0, "lda/ldb must be multiples of 8 elements (16-byte rows)");
  NTArgs a;
  a.A = (const bf16_t*)g->A; a.B = (const bf16_t*)g->B;
  a.M = g->M; a.N = g->N; a.K = g->K; a.lda = g->lda; a.ldb = g->ldb;
  a.C = g->C; a.ldc = g->ldc; a.C2 = g->C2; a.ldc2 = g->ldc2;
  a.bias = g->bias; a.resid = g->resid; a.ldr = g->ldr;
  a.seq_scale = g->seq_scale; a.row2seq = g->row2seq;
  a.aux = (const bf16_t*)g->aux; a.ldaux = g->ldaux; a.pos = g->pos; a.npatch = g->npatch; a.dbg = g_debug_flags; a.n_tiles = 0;
#ifdef LAFS_ABLATE
  a.stamps = g_stamps;
#endif
  a.drop = make_drop(g->drop_p, g->drop_seed, g->drop_step, (unsigned)g->drop_row0 * (unsigned)g->N);
  a.act = g->act;
  a.f16 = g->operand_f16 ? 1 : 0;
  LAFS_CHECK_ARG(!a.f16 || ((g->epilogue == LAFS_EPI_BF16 || g->epilogue == LAFS_EPI_BF16_ACT || g->epilogue == LAFS_EPI_F32) && g->splits <= 1),
                 "fp16 operands: plain / activation / fp32 epilogue, no K split");
  LAFS_CHECK_ARG(g->drop_p >= 0.f && g->drop_p < 1.f, "drop_p must be in [0, 1)");
  LAFS_CHECK_ARG(!(g->drop_p > 0.f) || ((long)g->M + g->drop_row0) * g->N < 4294967296L, "dropout needs (row0 + M) * N < 2^32");
  LAFS_CHECK_ARG(g->drop_row0 >= 0, "drop_row0 must be >= 0");
  if (lafs_kres_eligible(g)) return lafs_kres_launch(g, stream);   // K = 384 streaming shapes of the ViT-S trunk (gemm_kres.hip)
  if (lafs_big_eligible(g)) {                                      // wide long-K shapes of the Part-fViT trunk (gemm_big.hip)
    LAFS_CHECK_ARG(g->epilogue != LAFS_EPI_RESID_F32 || g->seq_scale == nullptr || g->row2seq != nullptr, "seq_scale needs row2seq");
    return lafs_big_launch(g, stream);
  }
  int splits = 1;
  a.klen = g->K;
  if (g->epilogue == LAFS_EPI_ATOMIC_F32 || (g->epilogue == LAFS_EPI_F32 && g->splits > 1)) {
    LAFS_CHECK_ARG(g->epilogue == LAFS_EPI_ATOMIC_F32 || g->bias == nullptr, "a K-split F32 GEMM takes no bias (it would be added per slice)");
    a.klen = ksplit_len(g->K, g->splits);
    splits = ceil_div(g->K, a.klen);
  }
  const bool vec_ok = (g->ldc % 8 == 0) || g->C == nullptr;
  LAFS_CHECK_ARG(vec_ok, "ldc must be a multiple of 8 elements");
  switch (g->epilogue) {
    case LAFS_EPI_BF16: return launch_nt<EPI_BF16>(a, 1, g->ctx, stream);
    case LAFS_EPI_BF16_GELU:
      LAFS_CHECK_ARG(g->C2 != nullptr && g->ldc2 % 8 == 0, "GELU epilogue needs C2");
      return launch_nt<EPI_BF16_GELU>(a, 1, g->ctx, stream);
    case LAFS_EPI_RESID_F32:
      LAFS_CHECK_ARG(g->resid != nullptr && g->ldr % 4 == 0, "residual epilogue needs resid");
      LAFS_CHECK_ARG(g->seq_scale == nullptr || g->row2seq != nullptr, "seq_scale needs row2seq");
      return launch_nt<EPI_RESID_F32>(a, 1, g->ctx, stream);
    case LAFS_EPI_F32: return launch_nt<EPI_F32>(a, splits, g->ctx, stream);
    case LAFS_EPI_DGELU_BF16:
      LAFS_CHECK_ARG(g->aux != nullptr, "dGELU epilogue needs aux (pre-activation)");
      return launch_nt<EPI_DGELU_BF16>(a, 1, g->ctx, stream);
    case LAFS_EPI_ATOMIC_F32: return launch_nt<EPI_ATOMIC_F32>(a, splits, g->ctx, stream);
    case LAFS_EPI_BF16_ACT:
      LAFS_CHECK_ARG(g->act >= 0 && g->act <= LAFS_ACT_HSIGMOID, "unknown activation");
      return launch_nt<EPI_BF16_ACT>(a, 1, g->ctx, stream);
    case LAFS_EPI_EMBED_F32:
      LAFS_CHECK_ARG(g->pos != nullptr && g->npatch > 0 && g->M % g->npatch == 0, "embed epilogue needs pos/npatch");
      return launch_nt<EPI_EMBED_F32>(a, 1, g->ctx, stream);
    default:
      lafs_set_error("lafs_gemm_nt: unknown epilogue %d", g->epilogue);
      return LAFS_EINVAL;
  }
}

static int launch_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2, int splits,
                     float* colsum_a, long part_stride, hipStream_t stream) {
  LAFS_CHECK_ARG(A && B && C, "null operand");
  LAFS_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0, "empty problem");
  LAFS_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && N1 % 8 == 0 && N2 % 8 == 0, "N1/N2/lda/ldb must be multiples of 8");
  TNArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.colsum = colsum_a;
  a.M = M; a.N1 = N1; a.N2 = N2; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.mode = 0;
#ifdef LAFS_ABLATE
  a.mode = g_debug_flags & 1;
#endif
  a.part_stride = part_stride;
  if (part_stride > 0) a.mode = 2;
  // 32-row stages in a 3-stage ring.  (64-row stages in a 2-stage ring -- half the barriers -- measured 1.8x slower: the
  // next stage's DMA cannot be issued before the barrier and only 2 workgroups fit a CU; debug flag 8192 selects them.)
  const int kb = (g_debug_flags & 8192) ? 64 : 32;
  const int msteps = ceil_div(M, kb);
  // Tile shape.  128x128 (4 waves, 3 workgroups per CU) is the fastest on nearly every LAFS shape: the wider variants re-read the
  // operands from L2 fewer times but 256x128 is on par at best (105 vs 97 us on the ViT-S fc1 wgrad, once it fits 128 VGPRs and
  // two workgroups share a CU) and 256x256 slower (tools/bench_kernels.py tn); debug flags 2048 = 256x256, 4096 = 256x128.
  static const int cand[4][2] = {{2, 2}, {4, 2}, {2, 4}, {4, 4}};
  int best = 0;
  {
    // ... except when the 128x128 grid would spill into a second round of workgroups (3 fit a CU: 768 slots): then 256x128
    // tiles (8 waves, 128 VGPRs, 2 workgroups per CU) win (ViT-B qkv wgrad: 123 us against 191 us; 256x256 gives 139 us)
    const int t22 = ceil_div(N1, 128) * ceil_div(N2, 128);
    int s22 = ceil_div(512, t22);
    if (s22 > 4) s22 = (s22 + 7) & ~7;
    if (splits <= 0 && t22 * s22 > 768 && M >= 8192) best = (N1 >= N2) ? 1 : 2;
  }
  if (g_debug_flags & 1024) best = 0;
  if (g_debug_flags & 2048) best = 3;
  if (g_debug_flags & 4096) best = (N1 >= N2) ? 1 : 2;
  const int wm = cand[best][0], wn = cand[best][1];
  const int tiles = ceil_div(N1, 64 * wm) * ceil_div(N2, 64 * wn);
  if (splits <= 0) {
    if (best == 0) {                       // 3 workgroups of 4 waves per CU: ~2 per CU, slices in multiples of 8 (one run per XCD)
      splits = ceil_div(512, tiles);
      if (splits > 4) splits = (splits + 7) & ~7;
    } else if (best == 3) {                // 1 workgroup of 16 waves per CU: a single wave of workgroups
      splits = 256 / tiles;
      if (splits < 1) splits = 1;
    } else {                               // 2 workgroups of 8 waves per CU: a bit more than one per CU
      splits = ceil_div(288, tiles);
      if (splits > 4) splits = (splits + 7) & ~7;
    }
  }
  if (splits > msteps) splits = msteps;
  a.mlen = ceil_div(msteps, splits) * kb;
  if (splits % 8 != 0) splits = ceil_div(M, a.mlen);       // (empty trailing slices are harmless for the x8 layout)
  a.splits = splits; a.tiles = tiles;
  const dim3 grid(tiles * splits);
  if (best == 0 && kb == 64) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 64>), grid, dim3(256), 0, stream, a);   // experiment
  else if (best == 0) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 32>), grid, dim3(256), 0, stream, a);
  else if (best == 1) hipLaunchKernelGGL((gemm_tn_kernel<4, 2, 32>), grid, dim3(512), 0, stream, a);
  else if (best == 2) hipLaunchKernelGGL((gemm_tn_kernel<2, 4, 32>), grid, dim3(512), 0, stream, a);
  else if (colsum_a != nullptr) hipLaunchKernelGGL((gemm_tn_kernel<4, 4, 32, true>), grid, dim3(1024), 0, stream, a);
  else hipLaunchKernelGGL((gemm_tn_kernel<4, 4, 32, false>), grid, dim3(1024), 0, stream, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_gemm_tn_acc(const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                                int M, int N1, int N2, int splits, float* colsum_a, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  return launch_tn(A, lda, B, ldb, C, ldc, M, N1, N2, splits, colsum_a, 0, stream);
}

extern "C" int lafs_gemm_tn_part(const void* A, int lda, const void* B, int ldb, float* part, int ldc, int64_t part_stride,
                                 int M, int N1, int N2, int splits, float* colsum_a, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(part_stride > 0, "part_stride must be positive");
  return launch_tn(A, lda, B, ldb, part, ldc, M, N1, N2, splits, colsum_a, part_stride, stream);
}

extern "C" int lafs_reduce_partials(float* part, int64_t part_stride, int n_part, int64_t n, float* out, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(part && out && n_part > 0 && n > 0 && n % 4 == 0 && part_stride % 4 == 0, "n and part_stride must be multiples of 4");
  const long n4 = n / 4;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, part, (long)part_stride,
                     n_part, n4, out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_sum_slices(const float* part, int64_t part_stride, int n_part, int64_t n, float* out, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(part && out && n_part > 0 && n > 0 && n % 4 == 0 && part_stride % 4 == 0, "n and part_stride must be multiples of 4");
  const long n4 = n / 4;
  hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, part, (long)part_stride, n_part, n4, out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

// Diagnostic knob for timing experiments (bit 0: TN GEMM stores instead of atomics -> WRONG results).  Never set by
// the product path.
extern "C" int lafs_debug_set(int flags) { g_debug_flags = flags; return LAFS_OK; }
extern "C" int lafs_debug_get(void) { return g_debug_flags; }
#ifdef LAFS_ABLATE
extern "C" void lafs_lab_set_stamps(void* buf) { g_stamps = (unsigned long long*)buf; }
#endif
extern "C" int lafs_ablation_build(void) {
#ifdef LAFS_ABLATE
  return 1;
#else
  return 0;
#endif
}
