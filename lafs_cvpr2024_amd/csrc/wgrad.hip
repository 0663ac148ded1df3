// Weight-gradient GEMM for gfx950, second generation:  C[N1,N2] (+)= A[M,N1]^T * B[M,N2],  reduction over the token axis M.
//
// Replaces autograd's dW = dY^T X behind every nn.Linear of the reference's backward pass
// (vision_transformer.py:59-65, 75-90; face_pre_pro/ViT_face.py:126-137, 147-149).
//
// Why a second kernel (profiles/round1_*): the 128x128 / 4x(64x64) kernel of gemm.hip re-read every operand panel from L2 once
// per 128 output columns (814 MB of L2->LDS traffic per ViT-S fc GEMM), paid one workgroup barrier per 16 MFMAs and 25x write
// amplification on its device-wide fp32 atomics.  Here:
//   * ONE workgroup of 4 waves per CU, one wave per SIMD with the whole 512-entry register file: a wave owns a
//     (32*FA) x (32*FB) block of v_mfma_f32_32x32x16_bf16 tiles (up to 128x128: 256 accumulator registers), the workgroup a
//     (64*FA) x (64*FB) output tile (256x192 for the ViT-S fc layers) -> 1.7x less L2->LDS traffic, 24-32 MFMAs per barrier;
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4) into a ring of NS 32-row stages, NS-1 stages in flight across the
//     raw s_barrier (counted s_waitcnt vmcnt);
//   * LDS image = 64-column panels of [32 rows][128 B]: every DMA instruction fetches 8 full 128-byte lines; the two 64-byte
//     halves of rows with bit 1 set are swapped (on the SOURCE column, the image is lane-linear), which makes the
//     ds_read_b64_tr_b16 fragment reads (4 rows x 64 B per half-wave) hit four distinct 64-byte bank slots: conflict-free;
//   * the token axis is cut into slices; every workgroup STORES its fp32 tile into a slice-major scratch image and a second
//     kernel folds the slices into the gradient (overwrite or accumulate): no atomics, each partial written once, read once.
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "lafs_hip.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

constexpr int MAXG = 8;       // GEMMs per grouped launch
struct WgItem {
  const bf16_t* A; const bf16_t* B;
  float* out;                 // slices > 1: scratch [slices][N1][N2]; slices == 1: C itself
  float* colsum;
  float* cs_part;             // bias-gradient partials [slices][tiles_n2][2 wave columns][N1] (folded in a fixed order by wgrad_fold_kernel)
  long ldo, slice_stride;     // row stride of `out`, distance between slice images
  int N1, N2, lda, ldb;
  int tile0, tiles_n2;        // first tile of this GEMM in the launch's tile list; tiles along N2
  int accumulate, pad_;       // slices == 1 only
};
struct WgArgs {
  WgItem it[MAXG];
  int n_items, M, mlen, slices, tiles, nblk;          // tiles: all GEMMs together
};

template <bool F16> __device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
constexpr int WG_F16 = 65536;          // ABL bit: operands are fp16 (lafs_wgrad_f16: the trainable landmark CNN); everything else as bf16
// LDS-DMA with a cache policy chosen at compile time (lab: 0 = default, 1 = nt, 2 = sc1, 3 = sc0 sc1, 4 = sc0)
template <int POL> __device__ __forceinline__ void lds_dma16_pol(const void* gsrc, unsigned lds_base) {
  if constexpr (POL == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(gsrc), "s"(lds_base) : "memory");
  else if constexpr (POL == 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" : : "v"(gsrc), "s"(lds_base) : "memory");
  else if constexpr (POL == 3) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc0 sc1" : : "v"(gsrc), "s"(lds_base) : "memory");
  else if constexpr (POL == 4) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc0" : : "v"(gsrc), "s"(lds_base) : "memory");
  else lds_dma16_m0(gsrc, lds_base);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int KB = 32;                 // reduction rows per stage
constexpr int PANEL = KB * 128;        // one 64-column panel of a stage

// Wave grid WM x WN, FA x FB 32x32 accumulator blocks per wave: output tile (32 FA WM) x (32 FB WN), 64 WM WN threads.
// ABL: timing ablations for tools/lab (0 on the product path): 1 = no DMA after the prologue, 2 = no MFMA, 4 = no fragment reads,
// 8 = fragments read once (MFMA-only loop), 16 = no wait / barrier in the steady loop, 32 = every slice streams the rows of slice 0
// (a fifth of the bytes: Infinity-Cache resident), 64 = the row pointers never advance (L2-resident stream)
template <int WM, int WN, int FA, int FB, int NS, int OCC, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN, OCC * (WM * WN) / 4) void wgrad_kernel(WgArgs p) {
  constexpr int NTH = 64 * WM * WN;
  constexpr int T1 = 32 * FA * WM, T2 = 32 * FB * WN;
  static_assert(T1 % 64 == 0 && T2 % 64 == 0, "tile sides must be whole 64-column panels");
  constexpr int PA = T1 / 64, PB = T2 / 64, NP = PA + PB;          // panels per stage
  constexpr int STAGE = NP * PANEL;
  constexpr int NCH = NP * 256;                                      // 16-byte pieces per stage
  constexpr int NR = (NCH + NTH - 1) / NTH;                          // LDS-DMA instructions per thread and stage
  static_assert(NS >= 3 && OCC * NS * STAGE <= 160 * 1024, "ring does not fit the LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // blocks b, b+8, ... share an XCD (and its L2): give every XCD a contiguous run of (slice, tile) pairs, so that the tiles of
  // one token slice -- which read the same rows of A and B -- run behind the same L2.  (Packing every (GEMM, slice) group whole
  // onto ONE XCD was measured in round 3: fabric reads 847 -> 794 MB per ViT-S block, step +0.13 ms; tools/lab/NOTES.md.)
  const int per = p.nblk >> 3;
  const int id = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (id >= p.slices * p.tiles) return;
  const int slice = id / p.tiles, gtile = id % p.tiles;
  int gi = 0;                                          // which GEMM of the group this tile belongs to (wave-uniform)
#pragma unroll
  for (int k = 1; k < MAXG; ++k)
    if (k < p.n_items && gtile >= p.it[k].tile0) gi = k;
  const WgItem& g_ = p.it[gi];
  const bf16_t* gA = g_.A; const bf16_t* gB = g_.B;
  const int N1 = g_.N1, N2 = g_.N2, lda = g_.lda, ldb = g_.ldb;
  const int tile = gtile - g_.tile0;
  const int n1_0 = (tile / g_.tiles_n2) * T1, n2_0 = (tile % g_.tiles_n2) * T2;
  const int mbeg = slice * p.mlen;
  const int mend = min(p.M, mbeg + p.mlen);
  const int nk = (mend - mbeg + KB - 1) / KB;

  // ---- LDS-DMA: piece q = round * NTH + tid of a stage is (panel q >> 8, row (q >> 3) & 31, 16-byte piece q & 7) and lands at
  // LDS byte 16 q; piece j of a row with bit 1 set comes from source piece j ^ 4 (64-byte halves swapped).  A wave that would
  // fall off the end of the stage in the last round re-loads piece q - NCH instead (same bytes, same place): every wave
  // issues NR loads per stage, so one vmcnt immediate is right for all of them.
  const int drow = (tid >> 3) & 31;
  const int dcol = ((tid & 7) ^ (((drow >> 1) & 1) << 2)) * 8;
  const bf16_t* gp[NR];              // this thread's piece i at the row it reads in the next stage to be issued (running pointer)
  int dld[NR]; unsigned ddst[NR];
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    int q = i * NTH + tid;
    if (q >= NCH) q -= NCH;
    const int panel = __builtin_amdgcn_readfirstlane(q >> 8);
    const bf16_t* col;               // column base of the piece (row 0)
    if (panel < PA) { const int c = n1_0 + panel * 64 + dcol; col = gA + (c < N1 ? c : 0); dld[i] = lda; }
    else { const int c = n2_0 + (panel - PA) * 64 + dcol; col = gB + (c < N2 ? c : 0); dld[i] = ldb; }
    ddst[i] = smem_base + __builtin_amdgcn_readfirstlane((q & ~63) * 16);
    gp[i] = col + (long)(((ABL & 32) ? 0 : mbeg) + drow) * dld[i];
  }
  // Stages are issued in order; `ti` = next stage, `islot` = byte offset of its ring slot.  Only a stage that reaches past the
  // last row of the whole matrix needs its row indices clamped (rows past the SLICE but inside the matrix are real memory and
  // are multiplied by zeroed A rows); that can only be the final stage, so the running pointers stay valid until then.
  int ti = 0; unsigned islot = 0;
  const bool last_clamped = (mbeg + nk * KB > p.M);
  auto issue_piece = [&](int i) {
    lds_dma16_pol<(ABL >> 8) & 7>(gp[i], ddst[i] + islot);
    if (!(ABL & 64)) gp[i] += KB * dld[i];
  };
  auto issue_done = [&]() { ++ti; islot += STAGE; if (islot == NS * STAGE) islot = 0; };
  auto issue = [&]() {
    if (last_clamped && ti == nk - 1) {
      const long back = max(mbeg + ti * KB + drow - (p.M - 1), 0);     // rows past the end of the matrix re-read its last row
#pragma unroll
      for (int i = 0; i < NR; ++i) lds_dma16_pol<(ABL >> 8) & 7>(gp[i] - back * dld[i], ddst[i] + islot);
    } else {
#pragma unroll
      for (int i = 0; i < NR; ++i) issue_piece(i);
    }
    issue_done();
  };

  f32x16_t acc[FA][FB];
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float cs[FA];
#pragma unroll
  for (int a = 0; a < FA; ++a) cs[a] = 0.f;
  // Bias gradient = column sums of A.  The tiles_n2 tiles of a row share A's columns and both wave columns of a workgroup hold
  // the same A fragments: the half stage (t, half) is summed by the wave column wn == half of tile t % tiles_n2 only, so every
  // wave of every workgroup carries the same small share.  (One tile per row doing all of it -- 50 VALU instructions per half
  // stage in two of its waves -- made those workgroups fall behind the ones that share their operand panels: L2 hit rate
  // 0.62 -> 0.49, fabric reads 620 -> 850 MB per ViT-S block, +20 us; tools/lab/NOTES.md.)
  static_assert(WN == 2, "the bias-gradient share assumes two wave columns");
  float* const colsum = g_.colsum;
  const bool do_colsum = (colsum != nullptr);
  int cs_wait = tile % g_.tiles_n2;                    // stages until this tile's turn
  bool cs_mine = false;
  auto cs_next = [&]() { cs_mine = do_colsum && cs_wait == 0; cs_wait = cs_wait == 0 ? g_.tiles_n2 - 1 : cs_wait - 1; };

  // ---- fragment addressing.  32x32x16 operand: lane (n = lane & 31, half h = lane >> 5) holds k = 8h .. 8h+7 of column n.
  // Two transpose reads give it: 16-lane group g = lane >> 4 covers columns (g & 1)*16 .. +15, rows 8h + 4j + 0..3 (j = 0, 1);
  // lane pl of the group supplies the address of row (pl >> 2), 8-byte unit (pl & 3).
  const int g = lane >> 4, pl = lane & 15, h = lane >> 5;
  const int lb0 = (h * 8 + (pl >> 2)) * 128 + (((pl >> 3) & 1) * 64) + (g & 1) * 32 + (pl & 3) * 8;
  int offA[FA], offB[FB];
#pragma unroll
  for (int a = 0; a < FA; ++a) { const int f = wm * FA + a; offA[a] = (f >> 1) * PANEL + (lb0 ^ ((f & 1) * 64)); }
#pragma unroll
  for (int b = 0; b < FB; ++b) { const int f = wn * FB + b; offB[b] = (PA + (f >> 1)) * PANEL + (lb0 ^ ((f & 1) * 64)); }

  auto load_one = [&](bf16x8_t (&fa)[FA], bf16x8_t (&fb)[FB], const unsigned char* sk, int f) {   // fragment f: A blocks first
    if (ABL & 4) return;
    const unsigned char* q = sk + (f < FA ? offA[f < FA ? f : 0] : offB[f < FA ? 0 : f - FA]);
    const s16x8_t v = __builtin_shufflevector(lds_read_tr16(q), lds_read_tr16(q + 512), 0, 1, 2, 3, 4, 5, 6, 7);
    if (f < FA) fa[f < FA ? f : 0] = __builtin_bit_cast(bf16x8_t, v);
    else fb[f < FA ? 0 : f - FA] = __builtin_bit_cast(bf16x8_t, v);
  };
  auto load_frags = [&](bf16x8_t (&fa)[FA], bf16x8_t (&fb)[FB], const unsigned char* sk) {
#pragma unroll
    for (int f = 0; f < FA + FB; ++f) load_one(fa, fb, sk, f);
  };
  // One half stage of MFMAs on (fa, fb).  Everything else the wave has to issue meanwhile is spread through the MFMAs' 32-cycle
  // shadows, in program order pinned by sched_barrier: the transpose reads of the NEXT half stage into (na, nb) (READS), and
  // the NR LDS-DMA pieces of the ring stage that is due (DMA).  With one wave per SIMD, whatever is issued outside those
  // shadows is lost matrix time (PMC: profiles/round2_wgrad_pmc.txt).
  auto group = [&](const bf16x8_t (&fa)[FA], const bf16x8_t (&fb)[FB], bf16x8_t (&na)[FA], bf16x8_t (&nb)[FB],
                   const unsigned char* nsk, auto READS, auto DMA, int half) {
    constexpr int NM = FA * FB, NF = FA + FB;
    if (ABL & (2 | 4)) {
      if (decltype(READS)::value) load_frags(na, nb, nsk);
      if (decltype(DMA)::value && !(ABL & 1)) {
#pragma unroll
        for (int i = 0; i < NR; ++i) issue_piece(i);
      }
      if (!(ABL & 4)) {
#pragma unroll
        for (int a = 0; a < FA; ++a) asm volatile("" ::"v"(fa[a]));
#pragma unroll
        for (int b = 0; b < FB; ++b) asm volatile("" ::"v"(fb[b]));
      }
      return;
    }
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
      for (int b = 0; b < FB; ++b) {
        const int k = a * FB + b;
        acc[a][b] = mfma32<(ABL & WG_F16) != 0>(fa[a], fb[b], acc[a][b]);
        if (decltype(READS)::value && !(ABL & 8)) {
#pragma unroll
          for (int f = 0; f < NF; ++f)
            if (f * NM / NF == k) load_one(na, nb, nsk, f);
        }
        if (decltype(DMA)::value && !(ABL & 1)) {
#pragma unroll
          for (int i = 0; i < NR; ++i)
            if (i * NM / NR == k) issue_piece(i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    if (cs_mine && half == wn) {                      // bias gradient: this lane's 8 token rows of column n, summed
#pragma unroll
      for (int a = 0; a < FA; ++a) {
        typedef __bf16 bf16x2v_t __attribute__((ext_vector_type(2)));
        const uint4 w = __builtin_bit_cast(uint4, fa[a]);
        if constexpr ((ABL & WG_F16) != 0) {
          cs[a] += (h_lo(w.x) + h_hi(w.x)) + (h_lo(w.y) + h_hi(w.y)) + (h_lo(w.z) + h_hi(w.z)) + (h_lo(w.w) + h_hi(w.w));
        } else {
        const bf16x2v_t one = __builtin_bit_cast(bf16x2v_t, 0x3f803f80u);        // v_dot2c_f32_bf16: cs += lo * 1 + hi * 1
        cs[a] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, w.x), one, cs[a], false);
        cs[a] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, w.y), one, cs[a], false);
        cs[a] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, w.z), one, cs[a], false);
        cs[a] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, w.w), one, cs[a], false);
        }
      }
    }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  // stage t is complete in LDS once at most `ahead` younger stages are still in flight (loads retire in order)
  auto wait_stage = [&](int ahead) {
    if (ahead >= 3) wait_vm<3 * NR>();
    else if (ahead == 2) wait_vm<2 * NR>();
    else if (ahead == 1) wait_vm<NR>();
    else wait_vm<0>();
  };
  auto fix_tail = [&](int t) {                        // ragged tail: rows past the slice were clamped to its last row; clear A's
    const int valid = mend - (mbeg + t * KB);
    if (valid >= KB) return;
    unsigned char* st = smem + (t % NS) * STAGE;
    for (int q = tid; q < PA * 256; q += NTH) {
      const int row = (q >> 3) & (KB - 1);
      if (row >= valid) *reinterpret_cast<uint4*>(st + q * 16) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
  };

#pragma unroll
  for (int t = 0; t < NS - 2; ++t)
    if (ti < nk) issue();
  // Software pipeline over 16-row half stages with two fragment register sets: each group of MFMAs carries the reads of the next
  // half stage; the wait + barrier that opens stage t+1 sits between the two groups of stage t; the DMA of stage t+NS-1 rides
  // in the second group.
  bf16x8_t a0[FA], b0[FB], a1[FA], b1[FB];
  wait_stage(min(nk - 1, NS - 3));
  __builtin_amdgcn_s_barrier();
  if (ti < nk) issue();
  fix_tail(0);
  load_frags(a0, b0, smem);
  if (ABL & 8) load_frags(a1, b1, smem + 16 * 128);
  unsigned rslot = 0;                                 // ring slot (byte offset) of the stage being multiplied
  int t = 0;
  for (; t < nk - NS; ++t) {                          // steady state: full ring, never the clamped stage
    cs_next();
    group(a0, b0, a1, b1, smem + rslot + 16 * 128, Yes{}, No{}, 0);
    if (!(ABL & 16)) {
      wait_vm<(NS - 3) * NR>();
      __builtin_amdgcn_s_barrier();                  // everyone's pieces of stage t+1 landed; the slot of stage t-1 is free
    }
    __builtin_amdgcn_sched_barrier(0);
    rslot += STAGE; if (rslot == NS * STAGE) rslot = 0;
    group(a1, b1, a0, b0, smem + rslot, Yes{}, Yes{}, 1);
    issue_done();
  }
  for (; t < nk; ++t) {                               // ring drains
    cs_next();
    group(a0, b0, a1, b1, smem + rslot + 16 * 128, Yes{}, No{}, 0);
    if (t + 1 < nk) {
      wait_stage(min(nk - 2 - t, NS - 3));
      __builtin_amdgcn_s_barrier();
      if (ti < nk && !(ABL & 1)) issue();
      fix_tail(t + 1);
      rslot += STAGE; if (rslot == NS * STAGE) rslot = 0;
      group(a1, b1, a0, b0, smem + rslot, Yes{}, No{}, 1);
    } else {
      group(a1, b1, a0, b0, smem, No{}, No{}, 1);
    }
  }

  // ---- store the tile: accumulator register r of a 32x32 block is row (r & 3) + 8 (r >> 2) + 4h, column lane & 31
  float* out = g_.out + (size_t)slice * g_.slice_stride;
  const long ldo = g_.ldo;
  const bool direct_acc = (p.slices == 1) && g_.accumulate;
  const int c2 = n2_0 + wn * FB * 32 + (lane & 31);
  // (WHOLE: the tile lies inside the matrix.  A guard around every store makes every store a basic block of its own, which hipcc
  // opens with s_waitcnt vmcnt(0): FA x FB x 16 = 144 stores per wave each sat out the store before it -- found in round 5 with
  // tools/lab/scan_waits.py; ViT-S and Part-fViT weights are multiples of the 192 x 192 tile except 704 and 2048)
  auto store_tile = [&](auto whole_c) __attribute__((always_inline)) {
    constexpr bool WHOLE = decltype(whole_c)::value;
    if (!direct_acc) {
#pragma unroll
      for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n1 = n1_0 + (wm * FA + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          float* dst = out + (size_t)n1 * ldo + c2;
#pragma unroll
          for (int b = 0; b < FB; ++b)
            if (WHOLE || (n1 < N1 && c2 + b * 32 < N2)) dst[b * 32] = acc[a][b][r];
        }
    } else {                                          // a single slice accumulating into C: one writer per element
#pragma unroll
      for (int a = 0; a < FA; ++a) {                   // (all 16 x FB old values of a block row requested before the first store: C may
        float old[16][FB];                             // alias nothing the compiler can prove, so it keeps each load behind the stores before it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n1 = n1_0 + (wm * FA + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float* dst = out + (size_t)n1 * ldo + c2;
#pragma unroll
          for (int b = 0; b < FB; ++b) old[r][b] = (WHOLE || (n1 < N1 && c2 + b * 32 < N2)) ? dst[b * 32] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n1 = n1_0 + (wm * FA + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          float* dst = out + (size_t)n1 * ldo + c2;
#pragma unroll
          for (int b = 0; b < FB; ++b)
            if (WHOLE || (n1 < N1 && c2 + b * 32 < N2)) dst[b * 32] = old[r][b] + acc[a][b][r];
        }
      }
    }
  };
  if (n1_0 + WM * FA * 32 <= N1 && n2_0 + WN * FB * 32 <= N2) store_tile(std::true_type());
  else store_tile(std::false_type());
  if (do_colsum) {                                    // lanes l and l + 32 hold the two k-halves of column l & 31
    // every (slice, tile of the row, wave column) stores its share -- possibly 0 -- into a slot of its own; wgrad_fold_kernel adds
    // the slots in a fixed order: the bias gradient is run-to-run deterministic (fp32 atomics before round 5)
    float* part = g_.cs_part + ((size_t)(slice * g_.tiles_n2 + tile % g_.tiles_n2) * 2 + wn) * N1;
#pragma unroll
    for (int a = 0; a < FA; ++a) {
      const float v = cs[a] + __shfl_xor(cs[a], 32, 64);
      const int n1 = n1_0 + (wm * FA + a) * 32 + (lane & 31);
      if (h == 0 && n1 < N1) part[n1] = v;
    }
  }
}

// C_g[r, c] = (accumulate_g ? C_g : 0) + sum_s part_g[s][r][c] for every GEMM g of the group (part rows are dense: N2 floats);
// behind the matrix elements: colsum_g[n] += sum over the (slice, tile, wave column) slots of the bias-gradient partials, in slot order
struct FoldItem { const float* part; float* C; long slice_stride, n4_begin, ldc; int n2_4, accumulate;
                  const float* cs_part; float* colsum; long cs_begin; int cs_slots, N1; };
struct FoldArgs { FoldItem it[MAXG]; int n_items, slices; long n4_total, cs_total; };
__global__ __launch_bounds__(256) void wgrad_fold_kernel(FoldArgs p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n4_total) {
    const long c = i - p.n4_total;
    if (c >= p.cs_total) return;
    int gi = 0;
#pragma unroll
    for (int k = 1; k < MAXG; ++k)
      if (k < p.n_items && c >= p.it[k].cs_begin) gi = k;
    const FoldItem g = p.it[gi];
    const long n = c - g.cs_begin;
    if (g.colsum == nullptr || n >= g.N1) return;
    float acc = 0.f;
    for (int s = 0; s < g.cs_slots; ++s) acc += g.cs_part[(size_t)s * g.N1 + n];
    g.colsum[n] += acc;
    return;
  }
  int gi = 0;
#pragma unroll
  for (int k = 1; k < MAXG; ++k)
    if (k < p.n_items && i >= p.it[k].n4_begin) gi = k;
  const FoldItem g = p.it[gi];
  const long j = i - g.n4_begin;
  const long row = j / g.n2_4, col = (j - row * g.n2_4) * 4;
  typedef float f32x4v_t __attribute__((ext_vector_type(4)));
  f32x4v_t* dst = reinterpret_cast<f32x4v_t*>(g.C + row * g.ldc + col);
  f32x4v_t acc = {0.f, 0.f, 0.f, 0.f};
  if (g.accumulate) acc = *dst;
  const f32x4v_t* src = reinterpret_cast<const f32x4v_t*>(g.part) + j;
#pragma unroll 4
  for (int s = 0; s < p.slices; ++s) acc += __builtin_nontemporal_load(src + s * (g.slice_stride / 4));   // read once, never again
  *dst = acc;
}

struct Plan { int fa, fb, tiles, slices, mlen; int tile0[MAXG], tiles_n2[MAXG]; };

// One tile shape for the whole group: the one with the least matrix work per workgroup (padding included) when the tiles of
// all GEMMs times the token slices fill the 256 CUs once.
// max_wg: workgroups (= CUs, one 512-register workgroup each) the launch may occupy; a two-stream backward keeps part of the
// chip free for the HBM-bound kernels of its other stream this way (0 = the whole chip)
// (Two workgroups per CU on three-stage rings -- OCC = 2, 256 registers per wave, twice the token slices -- were measured in
// round 3: 233 against 194 us per ViT-S block; the lab instantiation is tools/lab/lab_wgrad.cpp's LAB_OCC2.)
#ifdef LAFS_LAB_WGRAD_OCC2
int wgrad_occ() { static const int occ = getenv("LAB_OCC2") ? 2 : 1; return occ; }
#else
constexpr int wgrad_occ() { return 1; }
#endif
Plan make_plan(const lafs_wgrad_item* items, int n, int M, int max_wg) {
  const int occ = wgrad_occ();
  const int g_wgrad_cus = occ * ((max_wg >= 8 && max_wg <= 256) ? max_wg : 256);
  // (4x4 blocks per wave = 256 accumulators leave hipcc no room: it spills the accumulators around the loop nest)
  static const int cand[4][2] = {{3, 3}, {4, 3}, {3, 4}, {2, 2}};
  Plan best = {};
  long best_cost = -1;
  const int msteps = ceil_div(M, KB);
  for (int c = 0; c < 4; ++c) {
    Plan pl = {};
    pl.fa = cand[c][0]; pl.fb = cand[c][1];
    if (occ == 2 && pl.fa * pl.fb > 9) continue;
    for (int g = 0; g < n; ++g) {
      pl.tile0[g] = pl.tiles;
      pl.tiles_n2[g] = ceil_div(items[g].N2, 64 * pl.fb);
      pl.tiles += ceil_div(items[g].N1, 64 * pl.fa) * pl.tiles_n2[g];
    }
    int slices = g_wgrad_cus / pl.tiles;
    if (slices < 1) slices = 1;
    if (slices > msteps / 4) slices = msteps / 4 > 0 ? msteps / 4 : 1;     // at least 4 stages per workgroup
    pl.mlen = ceil_div(msteps, slices) * KB;
    pl.slices = ceil_div(M, pl.mlen);
    const long rounds = ceil_div(pl.tiles * pl.slices, g_wgrad_cus);
    long cost = rounds * (2L * (pl.mlen / KB) + 10) * pl.fa * pl.fb;
    if (pl.fa * pl.fb < 9) cost += cost / 2;          // 128x128 tiles: 2.25x the L2->LDS traffic and LDS reads per MFMA of 192x192
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = pl; }
  }
  return best;
}

// bias-gradient partials of GEMM g: one slot of N1 floats per (slice, tile along N2, wave column)
inline int64_t cs_slots(const Plan& pl, int g) { return (int64_t)pl.slices * pl.tiles_n2[g] * 2; }
int64_t plan_bytes(const Plan& pl, const lafs_wgrad_item* items, int n) {
  int64_t b = 0;
  for (int g = 0; g < n; ++g) {
    if (pl.slices > 1) b += (int64_t)pl.slices * items[g].N1 * items[g].N2 * 4;
    b += cs_slots(pl, g) * items[g].N1 * 4;            // (sized whether or not the caller passes colsum_a: the size query has no pointers)
  }
  return b;
}

template <int FA, int FB, int ABL>
int launch(const WgArgs& a, hipStream_t s) {
  constexpr int NS = 5;                              // 2-3 stages (28-32 KiB each) in flight per CU
#ifdef LAFS_LAB_WGRAD_OCC2
  if constexpr (FA * FB <= 9) {
    if (wgrad_occ() == 2) {
      hipLaunchKernelGGL((wgrad_kernel<2, 2, FA, FB, 3, 2, ABL>), dim3(a.nblk), dim3(256), 0, s, a);
      LAFS_LAUNCH_CHECK();
      return LAFS_OK;
    }
  }
#endif
#ifdef LAFS_LAB_WGRAD_OCC2
  if constexpr (FA * FB <= 9 && (ABL & ~WG_F16) != 0) {          // lab: a sixth ring stage (3 x 3 blocks: 6 x 24 KiB)
    static const bool ns6 = getenv("LAFS_WGRAD_NS6") != nullptr;
    if (ns6) {
      hipLaunchKernelGGL((wgrad_kernel<2, 2, FA, FB, 6, 1, ABL>), dim3(a.nblk), dim3(256), 0, s, a);
      LAFS_LAUNCH_CHECK();
      return LAFS_OK;
    }
  }
#endif
  hipLaunchKernelGGL((wgrad_kernel<2, 2, FA, FB, NS, 1, ABL>), dim3(a.nblk), dim3(256), 0, s, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

template <int ABL>
int group_impl(const lafs_wgrad_item* items, int n_items, int M, int max_workgroups, void* workspace, int64_t workspace_bytes,
               hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(items != nullptr && n_items > 0 && n_items <= MAXG, "1..8 GEMMs per group");
  LAFS_CHECK_ARG(M > 0, "empty problem");
  for (int g = 0; g < n_items; ++g) {
    const lafs_wgrad_item& it = items[g];
    LAFS_CHECK_ARG(it.A && it.B && it.C, "null operand");
    LAFS_CHECK_ARG(it.N1 > 0 && it.N2 > 0, "empty problem");
    LAFS_CHECK_ARG(it.lda % 8 == 0 && it.ldb % 8 == 0 && it.N1 % 8 == 0 && it.N2 % 8 == 0 && it.ldc % 4 == 0,
                   "N1/N2/lda/ldb must be multiples of 8, ldc of 4");
  }
  const Plan pl = make_plan(items, n_items, M, max_workgroups);
  const int64_t need = plan_bytes(pl, items, n_items);
  LAFS_CHECK_ARG(workspace != nullptr && workspace_bytes >= need, "workspace too small (lafs_wgrad_group_workspace_bytes)");
  WgArgs a = {};
  FoldArgs f = {};
  float* ws = (float*)workspace;
  long n4 = 0, ncs = 0;
  bool any_cs = false;
  for (int g = 0; g < n_items; ++g) {
    const lafs_wgrad_item& it = items[g];
    WgItem& w = a.it[g];
    w.A = (const bf16_t*)it.A; w.B = (const bf16_t*)it.B; w.colsum = it.colsum_a;
    w.N1 = it.N1; w.N2 = it.N2; w.lda = it.lda; w.ldb = it.ldb;
    w.tile0 = pl.tile0[g]; w.tiles_n2 = pl.tiles_n2[g]; w.accumulate = it.accumulate;
    if (pl.slices > 1) { w.out = ws; w.ldo = it.N2; w.slice_stride = (long)it.N1 * it.N2; }
    else { w.out = it.C; w.ldo = it.ldc; w.slice_stride = 0; }
    FoldItem& fi = f.it[g];
    fi.part = ws; fi.C = it.C; fi.slice_stride = (long)it.N1 * it.N2; fi.n4_begin = n4; fi.ldc = it.ldc;
    fi.n2_4 = it.N2 / 4; fi.accumulate = it.accumulate;
    if (pl.slices > 1) { n4 += (long)it.N1 * it.N2 / 4; ws += (size_t)pl.slices * it.N1 * it.N2; }
    w.cs_part = ws; fi.cs_part = ws; fi.colsum = it.colsum_a; fi.cs_begin = ncs; fi.cs_slots = (int)cs_slots(pl, g); fi.N1 = it.N1;
    ws += (size_t)cs_slots(pl, g) * it.N1;
    if (it.colsum_a != nullptr) any_cs = true;
    ncs += it.N1;
  }
  a.n_items = n_items; a.M = M; a.mlen = pl.mlen; a.slices = pl.slices; a.tiles = pl.tiles;
  a.nblk = (pl.slices * pl.tiles + 7) & ~7;
  int rc;
  if (pl.fa == 4 && pl.fb == 3) rc = launch<4, 3, ABL>(a, stream);
  else if (pl.fa == 3 && pl.fb == 4) rc = launch<3, 4, ABL>(a, stream);
  else if (pl.fa == 3 && pl.fb == 3) rc = launch<3, 3, ABL>(a, stream);
  else rc = launch<2, 2, ABL>(a, stream);
  if (rc != LAFS_OK) return rc;
  if (pl.slices > 1 || any_cs) {
    f.n_items = n_items; f.slices = pl.slices; f.n4_total = n4; f.cs_total = any_cs ? ncs : 0;
    hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)((n4 + f.cs_total + 255) / 256)), dim3(256), 0, stream, f);
    LAFS_LAUNCH_CHECK();
  }
  return LAFS_OK;
}

}  // namespace

extern "C" int64_t lafs_wgrad_group_workspace_bytes(const lafs_wgrad_item* items, int n_items, int M, int max_workgroups) {
  if (items == nullptr || n_items <= 0 || n_items > MAXG || M <= 0) return -1;
  for (int g = 0; g < n_items; ++g)
    if (items[g].N1 <= 0 || items[g].N2 <= 0) return -1;
  return plan_bytes(make_plan(items, n_items, M, max_workgroups), items, n_items);
}

extern "C" int lafs_wgrad_group(const lafs_wgrad_item* items, int n_items, int M, int max_workgroups, void* workspace,
                                int64_t workspace_bytes, hipStream_t stream) {
  return group_impl<0>(items, n_items, M, max_workgroups, workspace, workspace_bytes, stream);
}

// fp16 operands (the trainable landmark CNN's activations / activation gradients): same kernel, v_mfma_f32_32x32x16_f16
extern "C" int lafs_wgrad_f16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                              int accumulate, float* colsum_a, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  lafs_wgrad_item it = {};
  it.A = A; it.lda = lda; it.B = B; it.ldb = ldb; it.C = C; it.ldc = ldc; it.N1 = N1; it.N2 = N2;
  it.accumulate = accumulate; it.colsum_a = colsum_a;
  return group_impl<WG_F16>(&it, 1, M, 0, workspace, workspace_bytes, stream);
}

extern "C" int64_t lafs_wgrad_workspace_bytes(int M, int N1, int N2) {
  lafs_wgrad_item it = {};
  it.N1 = N1; it.N2 = N2;
  return lafs_wgrad_group_workspace_bytes(&it, 1, M, 0);
}

extern "C" int lafs_wgrad(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                          int accumulate, float* colsum_a, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  lafs_wgrad_item it = {};
  it.A = A; it.lda = lda; it.B = B; it.ldb = ldb; it.C = C; it.ldc = ldc; it.N1 = N1; it.N2 = N2;
  it.accumulate = accumulate; it.colsum_a = colsum_a;
  return lafs_wgrad_group(&it, 1, M, 0, workspace, workspace_bytes, stream);
}
