// 192x256 NT GEMM for gfx950, one persistent workgroup per CU:  C[M,N] = epilogue(A[M,K] * B[N,K]^T)
//
// Replaces the cuBLAS GEMMs behind the wide long-K linears of the reference's Part-fViT trunk -- FeedForward 768 -> 2048 -> 768,
// to_qkv 768 -> 2112, to_out 704 -> 768 and their input gradients (face_pre_pro/ViT_face.py:126-137, 147-149; the backbone of
// train_largescale.py's fine-tune step and of lafs_train.py's real pre-training pair) -- wherever lafs_gemm_nt (gemm.hip) routes
// them here (lafs_big_eligible); everything else stays on the 128x128 tiled kernel / the K-resident kernel.
//
// Why another kernel (round 5): the tiled kernel's two 128x128 workgroups per CU stage 32 KiB of operands
// per 32 MFMAs of a wave -- 64 B per MFMA-rate clock and CU, twice what the CU's global -> LDS path sustains (~32 B/clk: the
// LDS-DMA instructions' issue time is what its waves wait for), which caps it near 1.0-1.1 PFLOP/s whatever the stage depth; and
// a 64x64 wave tile reads 512 B of LDS per MFMA, the LDS's whole 128 B/clk at MFMA rate.  Here:
//   * ONE workgroup of 4 waves per CU, one wave per SIMD with the whole 512-entry register file: a wave owns 96x128 outputs
//     = 6x8 v_mfma_f32_16x16x32_bf16 blocks (192 accumulator registers; 8x8 = 256 leave hipcc no room: it spills them around
//     the loop), the workgroup a 192x256 tile: 56 KiB of operands per 96 MFMAs of a wave = 37 B per MFMA-rate clock and CU, and
//     ~300 B of LDS reads per MFMA.  192 rows also divide the LAFS batches' 44 160 token rows (230 row tiles);
//   * operands arrive by LDS-DMA as whole 128-byte lines (64-deep stages, rows of 128 B, chunk c of row r at chunk position
//     c ^ (r & 7): conflict-free ds_read_b128 fragment reads; the image is lane-linear, so the swizzle is on the SOURCE column)
//     into two 64 KiB buffers.  A stage is consumed as two 32-deep halves from two fragment register sets; the ONE barrier per
//     stage sits between the halves: behind it every wave has read all of the stage (the second half's fragments are read
//     under the first half's MFMAs), so the next-but-one stage is requested into the same buffer under the second half's MFMAs;
//   * everything a wave issues besides its MFMAs -- 32 fragment reads and 16 LDS-DMA pieces per stage -- sits between two
//     MFMAs in an order pinned by sched_barrier (with one wave per SIMD nothing else hides it);
//   * the workgroup is PERSISTENT: it walks tiles b, b + grid, ... (an XCD gets a contiguous run of tiles per round: the
//     column tiles of a row tile share its L2) with the operand ring running across tile boundaries -- while a tile's epilogue
//     converts and stores, the next tile's first two stages are already in flight and its first fragments in registers, and the
//     stores drain under the next tile's MFMAs (counted s_waitcnt vmcnt: the epilogue's stores are a compile-time count);
//   * C^T blocks (MFMA A-operand = 16 weight rows) with the weight rows of a 64-column group permuted as in gemm.hip, so a lane
//     ends with 8 (bf16) / 4 (fp32) consecutive output columns and the four lane groups of a row store 64 contiguous bytes.
#include <type_traits>
#include "common.hpp"
#include "ctx.hpp"
#include "gemm_big.hpp"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

constexpr int BK = 64;                      // k-depth of a stage
constexpr int ROWB = BK * 2;                // bytes per LDS row
// Geometry: NWM x NWN waves, each MBW x NBW blocks of 16 x 16 outputs.
//   Geo<6, 8, 2, 2>: 192 x 256 tile, 4 waves, ONE per SIMD (192 accumulators + two fragment sets of 56 registers)
//   Geo<8, 4, 2, 4>: 256 x 256 tile, 8 waves, TWO per SIMD (128 accumulators + two fragment sets of 48: 256 registers per wave) --
//                    what one wave issues besides its MFMAs (LDS-DMA pieces, fragment reads, the epilogue) hides behind the
//                    MFMAs of the other wave of its SIMD
template <int MBW_, int NBW_, int NWM_, int NWN_> struct Geo {
  static constexpr int MBW = MBW_, NBW = NBW_, NWM = NWM_, NWN = NWN_;
  static constexpr int NTH = 64 * NWM * NWN;
  static constexpr int BTM = 16 * MBW * NWM, BTN = 16 * NBW * NWN;       // tile rows / columns
  static constexpr int RPP = NTH / 8;                                     // LDS rows one LDS-DMA instruction of the workgroup covers
  // LDS-DMA instructions per thread and stage.  The stage's LDS rows are the tile's A rows, then its B rows, then (BTM + BTN not
  // a multiple of RPP: the 176-row tile) rows nobody reads: the last piece's idle waves fetch the last B row again, so that every
  // wave issues the same count of pieces (s_waitcnt vmcnt is a count).  A piece may straddle the A / B boundary: a wave covers 8
  // rows and BTM % 8 == 0, so which operand a wave's lanes address is wave-uniform.
  static constexpr int NPIECE = (BTM + BTN + RPP - 1) / RPP;
  static constexpr int STAGE = NPIECE * RPP * ROWB;
  static constexpr int NMF = MBW * NBW;                                   // MFMAs of a half stage per wave
  static constexpr int NFR = MBW + NBW;                                   // fragment reads of a half stage per wave
  static constexpr int DSTEP = NMF / NPIECE;                              // one LDS-DMA piece per DSTEP MFMAs of a half stage
  static_assert(BTM % 16 == 0 && BTN % 64 == 0 && NBW % 4 == 0 && DSTEP >= 1 && 2 * STAGE <= 160 * 1024, "geometry");
};

struct BArgs {
  const bf16_t* A; const bf16_t* B;
  int M, N, K, lda, ldb;
  void* C; int ldc; void* C2; int ldc2;
  const float* bias; const float* resid; int ldr;
  const float* seq_scale; const int* row2seq;
  const bf16_t* aux; int ldaux;
  DropCfg drop; int act;
  int tiles_m, tiles_n, grid;
  int swizzle;                            // 1: blocks b, b + 8, ... (one XCD) take a contiguous run of tile numbers per round
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void st16(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
  const u32x4_t v = {a, b, c, d};
  *reinterpret_cast<u32x4_t*>(p) = v;
}
__device__ __forceinline__ void st16f(void* p, float a, float b, float c, float d) {
  const f32x4v_t v = {a, b, c, d};
  *reinterpret_cast<f32x4v_t*>(p) = v;
}
// LDS row rho (0..255 of the B half) <- weight row of the tile: inside every 64-row group the MFMA row slot (column group j, lane
// group g, register r) takes the weight row that makes a lane's registers consecutive output columns (gemm.hip: nt_perm)
template <int VPL> __device__ __forceinline__ int big_perm(int rho) {
  if (VPL == 4) return rho;
  const int j = (rho >> 4) & 3, g = (rho >> 2) & 3, r = rho & 3;
  return (rho & ~63) + (j >> 1) * 32 + g * 8 + (j & 1) * 4 + r;
}

// TWO (BF16_GELU only): both tensors are written (C = u or gelu'(u), C2 = gelu(u)); otherwise only C2
template <int EPI, bool TWO, typename G>
__global__ __launch_bounds__(G::NTH, 1) void gemm_big_kernel(BArgs p) {
  constexpr int MBW = G::MBW, NBW = G::NBW, BTM = G::BTM, BT = G::BTN, STAGE = G::STAGE, NPIECE = G::NPIECE, RPP = G::RPP;
  constexpr bool F32 = (EPI == LAFS_EPI_RESID_F32);
  constexpr int VPL = F32 ? 4 : 8;
  constexpr int ESTORES = (F32 ? 4 : (TWO ? 4 : 2)) * MBW * (NBW / 4);          // stores of one tile's epilogue per wave
  // The cross-tile wait below uses the largest of the immediates 16 / 32 / 48 that does not exceed ESTORES (a LOWER bound of the
  // stores younger than the awaited stage): a geometry / epilogue with fewer than 16 stores per wave would need a smaller immediate.
  // That hipcc emits exactly one store instruction per counted store and skips none is what the bit-exact comparison with the tiled
  // kernel holds for every Geo x EPI (tests/test_gpu_kernels.py::test_gemm_nt_big_tiles_equal_the_tiled_kernel_bit_for_bit).
  static_assert(ESTORES >= 16 && ESTORES < 64, "cross-tile s_waitcnt vmcnt immediates assume 16 <= ESTORES < 64");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const DropCfg drop = drop_resolve(p.drop);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / G::NWN, wc = wave % G::NWN;
  const int t16 = lane & 15, fq = lane >> 4;
  // blocks b, b+8, ... share an XCD: give each XCD a contiguous run of tile numbers in every round
  const int wid = p.swizzle ? (blockIdx.x & 7) * (p.grid >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nk = p.K / BK;
  const int my_tiles = wid < ntiles ? (ntiles - wid + p.grid - 1) / p.grid : 0;
  const int total = my_tiles * nk;                          // stages this workgroup consumes
  if (total == 0) return;

  // ---- LDS-DMA producer.  Piece i of a stage covers LDS rows RPP i + (tid >> 3) (rows < BTM: A rows, then the B rows);
  // this thread's 16 bytes are chunk position tid & 7 of its row = source chunk (tid & 7) ^ (row & 7) = (tid & 7) ^ ((tid >> 3) & 7).
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem)) + wave * 1024;
  unsigned voff[NPIECE];                                   // byte offset of this thread's piece i from the tile's A / B origin
  const bf16_t* gA = p.A; const bf16_t* gB = p.B;          // origin of the producer's tile (+ its k position), wave-uniform
  int d_tile = 0, d_k = 0, d_issued = 0;
  auto dma_tile = [&]() __attribute__((always_inline)) {
    const int t = wid + d_tile * p.grid;
    const int m0 = (t / p.tiles_n) * BTM, n0 = (t % p.tiles_n) * BT;
    gA = p.A + (size_t)m0 * p.lda; gB = p.B + (size_t)n0 * p.ldb;
    // (re-derived per tile from an opaque copy of the thread id: hoisted out of the tile loop the 14 row numbers stay live across
    // the MFMA loops, spill, and their reload's compiler-inserted vmcnt(0) drains the LDS-DMA queue at every tile change)
    int t_ = threadIdx.x;
    asm volatile("" : "+v"(t_));
    const int drow = t_ >> 3;
    const unsigned dk2 = (unsigned)(((t_ & 7) ^ (drow & 7)) * 16);               // source byte offset inside the 128-byte k slice
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int lrow = RPP * i + drow;                       // LDS row of the stage
      if (RPP * (i + 1) <= BTM || (RPP * i < BTM && lrow < BTM)) voff[i] = (unsigned)min(lrow, p.M - 1 - m0) * (unsigned)p.lda * 2u + dk2;
      else voff[i] = (unsigned)min(big_perm<VPL>(min(lrow - BTM, BT - 1)), p.N - 1 - n0) * (unsigned)p.ldb * 2u + dk2;
    }
  };
  // piece i of the producer's current stage into ring buffer `buf`
  auto dma_piece = [&](int i, int buf) __attribute__((always_inline)) {
    const bool from_a = RPP * (i + 1) <= BTM || (RPP * i < BTM && RPP * i + 8 * wave < BTM);          // wave-uniform
    lds_dma16_m0_s(from_a ? (const void*)gA : (const void*)gB, voff[i], lds0 + buf * STAGE + i * (G::NTH * 16));
  };
  // (behind the last stage the cursor stays where it is: the steady-state loop keeps issuing -- the last stage again, into a
  // buffer nobody reads any more -- instead of carrying a branch between its MFMAs)
  auto dma_advance = [&]() __attribute__((always_inline)) {
    if (++d_issued < total) {
      gA += BK; gB += BK;
      if (++d_k == nk) { d_k = 0; ++d_tile; dma_tile(); }
    }
  };
  dma_tile();

  // ---- fragment addressing: row 16 i + t16 of the wave's A rows / B rows, chunk (4 kk + fq) ^ (t16 & 7)
  const int fl0 = t16 * ROWB + ((fq ^ (t16 & 7)) << 4), fl1 = t16 * ROWB + (((4 + fq) ^ (t16 & 7)) << 4);
  const int offA0 = wr * (16 * MBW) * ROWB + fl0, offA1 = wr * (16 * MBW) * ROWB + fl1;
  const int offB0 = BTM * ROWB + wc * (16 * NBW) * ROWB + fl0, offB1 = BTM * ROWB + wc * (16 * NBW) * ROWB + fl1;
  auto load_frag = [&](bf16x8_t (&fa)[MBW], bf16x8_t (&fb)[NBW], const unsigned char* st, int kk, int f) __attribute__((always_inline)) {
    if (f < MBW) fa[f] = *reinterpret_cast<const bf16x8_t*>(st + (kk ? offA1 : offA0) + f * 16 * ROWB);
    else fb[f - MBW] = *reinterpret_cast<const bf16x8_t*>(st + (kk ? offB1 : offB0) + (f - MBW) * 16 * ROWB);
  };

  f32x4_t acc[NBW][MBW];                                   // [j: 16-column group][i: 16-row block]
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NBW; ++j)
#pragma unroll
      for (int i = 0; i < MBW; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  // One half stage: MBW x NBW MFMAs on (fa, fb); in their shadows the fragment reads of the next half stage (from `nst`, half `nkk`)
  // -- the A fragments ROLLING: row block i's next fragment replaces fa[i] right behind its last MFMA (one A set instead of two:
  // 32 registers the two-waves-per-SIMD geometry does not have), the B fragments into the other B set `nb` -- and, second half
  // only, the LDS-DMA pieces of the producer's stage into buffer `dbuf`, evenly spread.  sched_barrier pins the order: whatever a
  // wave issues outside the MFMAs' shadows is lost matrix time.
  auto group = [&](bf16x8_t (&fa)[MBW], const bf16x8_t (&fb)[NBW], bf16x8_t (&nb)[NBW], const unsigned char* nst, int nkk, bool dma,
                   int dbuf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int k = i * NBW + j;
        acc[j][i] = mfma16(fb[j], fa[i], acc[j][i]);
        if (j == NBW - 1) fa[i] = *reinterpret_cast<const bf16x8_t*>(nst + (nkk ? offA1 : offA0) + i * 16 * ROWB);
        if (k % MBW == MBW / 2 && k / MBW < NBW)
          nb[k / MBW] = *reinterpret_cast<const bf16x8_t*>(nst + (nkk ? offB1 : offB0) + (k / MBW) * 16 * ROWB);
        if (k % G::DSTEP == 0 && k / G::DSTEP < NPIECE && dma) dma_piece(k / G::DSTEP, dbuf);   // (`dma` is a literal at both call sites)
        __builtin_amdgcn_sched_barrier(0);
      }
  };

  // ---- epilogue of the tile at (m0, n0): lane owns rows m0 + wr*16*MBW + 16 i + t16 and, per row and 64-column group jg, 16 / VPL
  // pieces of VPL consecutive columns: piece qq of group jg starts at n0 + wc*128 + 64 jg + qq*4*VPL + fq*VPL; register
  // e = (j & 3) * 4 + r of the group is element e % VPL of piece e / VPL
  // (WHOLE: the tile lies inside the matrix -- no guard, hence no branch, around any load or store.  With guards every store sits in
  // its own basic block, and hipcc opens each block with s_waitcnt vmcnt(0) -- the bias registers came from loads -- which also
  // sits out the store before it: 22 serialised write round trips per tile in the plain epilogue)
  auto epilogue_body = [&](int m0, int n0, auto whole_c) __attribute__((always_inline)) {
    constexpr bool WHOLE = decltype(whole_c)::value;
    constexpr int NG = 16 / VPL;
#pragma unroll
    for (int jg = 0; jg < NBW / 4; ++jg) {
      const int ncol0 = n0 + wc * (16 * NBW) + jg * 64 + fq * VPL;
      float bias[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) bias[e] = 0.f;
      if (p.bias != nullptr && EPI != LAFS_EPI_DGELU_BF16) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int n = ncol0 + (e / VPL) * 4 * VPL + (e % VPL);
          if (WHOLE || n < p.N) bias[e] = p.bias[n];
        }
      }
      // Operand loads of the epilogue (the residual rows, the saved gelu'(u) / u rows, the per-sequence DropPath scale) are requested
      // for RB rows at once, in front of the arithmetic and the stores of those rows: left inside the per-piece code each load was
      // followed by its own s_waitcnt vmcnt(0) -- which also sits out every store issued before it -- i.e. 4 x MBW serialised HBM round
      // trips per tile (~30 us of a 69 us tile of the fc2 forward)
      constexpr bool PRE = (EPI == LAFS_EPI_RESID_F32 || EPI == LAFS_EPI_DGELU_BF16);
      constexpr int RB = PRE ? (G::NTH > 256 ? 2 : 4) : 1;      // (two waves per SIMD: half the registers)
#pragma unroll
      for (int i0 = 0; i0 < MBW; i0 += RB) {
        uint4 pre[RB][NG];
        float scs[RB];
        if (PRE) {
          int sidx[RB];                                       // (index loads first: the scale loads behind the row loads wait on these only)
#pragma unroll
          for (int ii = 0; ii < RB; ++ii) {
            const int m = m0 + wr * (16 * MBW) + min(i0 + ii, MBW - 1) * 16 + t16;
            sidx[ii] = (EPI == LAFS_EPI_RESID_F32 && p.seq_scale != nullptr) ? p.row2seq[(WHOLE || m < p.M) ? m : p.M - 1] : 0;
          }
#pragma unroll
          for (int ii = 0; ii < RB; ++ii) {
            const int i = i0 + ii;
            if (i < MBW) {
              const int m = m0 + wr * (16 * MBW) + i * 16 + t16;
              const int mc = (WHOLE || m < p.M) ? m : p.M - 1;
#pragma unroll
              for (int qq = 0; qq < NG; ++qq) {
                const int n = ncol0 + qq * 4 * VPL;
                const int nc = (WHOLE || n + VPL <= p.N) ? n : 0;
                if (EPI == LAFS_EPI_RESID_F32) pre[ii][qq] = *reinterpret_cast<const uint4*>(p.resid + (size_t)mc * p.ldr + nc);
                else pre[ii][qq] = *reinterpret_cast<const uint4*>(p.aux + (size_t)mc * p.ldaux + nc);
              }
            }
          }
#pragma unroll
          for (int ii = 0; ii < RB; ++ii) scs[ii] = (EPI == LAFS_EPI_RESID_F32 && p.seq_scale != nullptr) ? p.seq_scale[sidx[ii]] : 1.0f;
        }
#pragma unroll
      for (int ii = 0; ii < RB; ++ii) {
        const int i = i0 + ii;
        if (i >= MBW) continue;
        const int m = m0 + wr * (16 * MBW) + i * 16 + t16;
        const bool rowok = WHOLE || m < p.M;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[jg * 4 + j][i][r] + bias[j * 4 + r];
        const float sc = PRE ? scs[ii] : 1.0f;
#pragma unroll
        for (int qq = 0; qq < NG; ++qq) {
          const int n = ncol0 + qq * 4 * VPL;
          const bool ok = WHOLE || (rowok && (n + VPL <= p.N));        // (N % 8 == 0: a piece is inside the matrix or outside it)
          float* w = v + qq * VPL;
          if (EPI == LAFS_EPI_RESID_F32) {
            const uint4 r4 = pre[ii][qq];
            if (drop.thresh) {
#pragma unroll
              for (int e = 0; e < VPL; ++e) w[e] *= drop_mult(drop, (unsigned)m * (unsigned)p.N + (unsigned)(n + e));
            }
            w[0] = __uint_as_float(r4.x) + sc * w[0]; w[1] = __uint_as_float(r4.y) + sc * w[1];
            w[2] = __uint_as_float(r4.z) + sc * w[2]; w[3] = __uint_as_float(r4.w) + sc * w[3];
            if (ok) st16f(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, w[0], w[1], w[2], w[3]);
          } else {
            if (EPI == LAFS_EPI_DGELU_BF16) {
              const uint4 a4 = pre[ii][qq];
              if (p.act == LAFS_GELU_SAVE_GRAD) {           // aux already holds gelu'(u)
                w[0] *= bf_lo(a4.x); w[1] *= bf_hi(a4.x); w[2] *= bf_lo(a4.y); w[3] *= bf_hi(a4.y);
                w[4] *= bf_lo(a4.z); w[5] *= bf_hi(a4.z); w[6] *= bf_lo(a4.w); w[7] *= bf_hi(a4.w);
              } else {
                w[0] *= gelu_grad_f(bf_lo(a4.x)); w[1] *= gelu_grad_f(bf_hi(a4.x)); w[2] *= gelu_grad_f(bf_lo(a4.y)); w[3] *= gelu_grad_f(bf_hi(a4.y));
                w[4] *= gelu_grad_f(bf_lo(a4.z)); w[5] *= gelu_grad_f(bf_hi(a4.z)); w[6] *= gelu_grad_f(bf_lo(a4.w)); w[7] *= gelu_grad_f(bf_hi(a4.w));
              }
              if (drop.thresh) {                             // d(dropout(gelu(u))): the forward's mask, regenerated
#pragma unroll
                for (int e = 0; e < VPL; ++e) w[e] *= drop_mult(drop, (unsigned)m * (unsigned)p.N + (unsigned)(n + e));
              }
            }
            if (EPI == LAFS_EPI_BF16_GELU) {
              float g[8];
              if (TWO) {
                float d[8];
                if (p.act == LAFS_GELU_SAVE_GRAD) {         // (as the tiled kernel evaluates them: bit-identical routes)
#pragma unroll
                  for (int e = 0; e < 8; ++e) gelu_both_f(w[e], g[e], d[e]);
                } else {
#pragma unroll
                  for (int e = 0; e < 8; ++e) { d[e] = w[e]; g[e] = gelu_f(w[e]); }
                }
                if (ok) st16(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n, pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3]),
                             pack_bf2(d[4], d[5]), pack_bf2(d[6], d[7]));
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = gelu_f(w[e]);
              }
              if (drop.thresh) {
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] *= drop_mult(drop, (unsigned)m * (unsigned)p.N + (unsigned)(n + e));
              }
              if (ok) st16(reinterpret_cast<bf16_t*>(p.C2) + (size_t)m * p.ldc2 + n, pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]),
                           pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7]));
            } else {
              if (ok) st16(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n, pack_bf2(w[0], w[1]), pack_bf2(w[2], w[3]),
                           pack_bf2(w[4], w[5]), pack_bf2(w[6], w[7]));
            }
          }
        }
      }
      }
    }
  };
  auto epilogue = [&](int m0, int n0) __attribute__((always_inline)) {
    // (the GELU epilogues keep the one guarded form: their stores already issue back to back, and two copies of that much
    // arithmetic cost the 176x256 geometry 13 spilled registers -- 230 -> 277 us on the fc1 forward)
    if (EPI != LAFS_EPI_BF16_GELU && m0 + BTM <= p.M && n0 + BT <= p.N) epilogue_body(m0, n0, std::true_type());
    else epilogue_body(m0, n0, std::false_type());
  };

  // ---- prologue: stages 0 and 1 in flight, stage 0's first fragments in registers
  bf16x8_t a0[MBW], b0[NBW], b1[NBW];
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) dma_piece(i, 0);
  dma_advance();
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) dma_piece(i, 1);          // (total == 1: the same stage again)
  dma_advance();
  wait_vm<NPIECE>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int f = 0; f < G::NFR; ++f) load_frag(a0, b0, smem, 0, f);
  zero_acc();

  // Stores the previous tile's epilogue issued behind the pieces of the stage the next barrier waits for (vmcnt counts loads and
  // stores in issue order).  Only a LOWER bound is safe here -- a larger immediate would let pieces of the awaited stage stay in
  // flight -- so a tile that is not whole (rows beyond M / columns beyond N skip stores) counts as 0.
  int younger = 0;
  unsigned par = 0;                                         // ring buffer of the stage being multiplied
  for (int c_tile = 0; c_tile < my_tiles; ++c_tile) {
    for (int ks = 0; ks < nk; ++ks) {
      const unsigned char* cur = smem + par * STAGE;
      const unsigned char* nxt = smem + (par ^ 1) * STAGE;
      // first half: MFMAs on k 0..31, fragment reads of k 32..63 of the same stage
      group(a0, b0, b1, cur, 1, false, 0);
      // barrier: the next stage has landed everywhere (its pieces were issued half a stage ago or earlier), and every wave has read
      // all of this one -- its buffer is free for the stage after the next
      if (younger == 0) wait_vm<0>();
      else if (younger < 32) wait_vm<16>();
      else if (younger < 48) wait_vm<32>();
      else wait_vm<48>();
      __builtin_amdgcn_s_barrier();
      younger = 0;
      __builtin_amdgcn_sched_barrier(0);
      // second half: MFMAs on k 32..63, fragment reads of the next stage's k 0..31, LDS-DMA of the stage after the next into this
      // stage's buffer (behind the last stages: reads of a buffer that holds nothing new, pieces of the last stage once more)
      group(a0, b1, b0, nxt, 0, true, (int)par);
      dma_advance();
      par ^= 1;
    }
    const int t = wid + c_tile * p.grid;
    const int m0 = (t / p.tiles_n) * BTM, n0 = (t % p.tiles_n) * BT;
    epilogue(m0, n0);
    zero_acc();
    younger = (m0 + BTM <= p.M && n0 + BT <= p.N) ? ESTORES : 0;
  }
  wait_vm<0>();                                             // no LDS-DMA piece may land after the workgroup has given up its LDS
}

using GeoOne = Geo<6, 8, 2, 2>;            // 192 x 256, one wave per SIMD
using GeoTwo = Geo<8, 4, 2, 4>;            // 256 x 256, two waves per SIMD
using GeoFive = Geo<5, 8, 2, 2>;           // 160 x 256, one wave per SIMD: 158 row tiles of the fine-tune step's 25 216 rows (474 tiles = 1.85 rounds)
using GeoSlim = Geo<11, 4, 1, 4>;          // 176 x 256, one wave per SIMD, a wave owns 176 x 64: 251 row tiles of the 44 160-row batches

template <int EPI, bool TWO, typename G>
int launch(BArgs a, hipStream_t s) {
  a.tiles_m = ceil_div(a.M, G::BTM); a.tiles_n = ceil_div(a.N, G::BTN);
  const int tiles = a.tiles_m * a.tiles_n;
  a.grid = tiles >= 256 ? 256 : tiles;
  a.swizzle = (a.grid % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((gemm_big_kernel<EPI, TWO, G>), dim3(a.grid), dim3(G::NTH), 0, s, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

// useful outputs of the launch / outputs its rounds of one tile per CU could hold, for a tile shape (partial edge tiles count as
// what they are: 640 columns fill 2.5 of 3 column tiles)
inline long fill_permille(int M, int N, int btm, int btn) {
  const long tiles = (long)ceil_div(M, btm) * ceil_div(N, btn);
  const long slots = ceil_div(tiles, 256L) * 256;
  return (long)((double)M * N * 1000.0 / ((double)slots * btm * btn));
}

}  // namespace

// LAFS_OPT_NT_BIG: 0 off, 1 = where it wins (default), 2 / 3 / 4 / 5 = force the 192 x 256 / 256 x 256 / 176 x 256 / 160 x 256 geometry on every shape and
// epilogue the kernel covers (tests, A/B runs).  Measured on one MI355X against the 128x128 tiled kernel, interleaved
// (tools/lab/t_big_ab.py, M = 44 160): PLAIN epilogue, one wave per SIMD: qkv forward 163.0 -> 149.0 us, fc1 input gradient 131.5 ->
// 122.1 (1138 TFLOP/s), qkv input gradient 139.0 -> 130.1, projection input gradient 56.5 -> 55.4; the GELU pair / residual / GELU'
// epilogues LOSE 5-13 % (one workgroup per CU: nothing covers an epilogue that reads operands and writes 2-4x the bytes), the
// two-waves-per-SIMD geometry loses everywhere on these shapes (519 tiles = 2.03 rounds of 256), and so does a launch whose last
// round is emptier than ~15 % (M = 25 216, N = 768: 1.55 rounds).  Only the winning combination is routed here by default.
// geometry of a request: 1 = 192 x 256, 2 = 256 x 256 (two waves per SIMD), 3 = 176 x 256, 4 = 160 x 256.  Unforced, the one-wave-per-SIMD tile
// whose rounds of one tile per CU are fuller (44 160 x 768: 690 tiles of 192 rows = 2.70 rounds, 753 of 176 rows = 2.94)
static long geo_fill(const lafs_gemm_nt_args* g, int geo) {
  return geo == 4 ? fill_permille(g->M, g->N, GeoFive::BTM, GeoFive::BTN) : geo == 3 ? fill_permille(g->M, g->N, GeoSlim::BTM, GeoSlim::BTN)
       : geo == 2 ? fill_permille(g->M, g->N, GeoTwo::BTM, GeoTwo::BTN) : fill_permille(g->M, g->N, GeoOne::BTM, GeoOne::BTN);
}
static long geo_tiles(const lafs_gemm_nt_args* g, int geo) {
  const int btm = geo == 4 ? GeoFive::BTM : geo == 3 ? GeoSlim::BTM : geo == 2 ? GeoTwo::BTM : GeoOne::BTM;
  return (long)ceil_div(g->M, btm) * ceil_div(g->N, 256);
}
static int big_geometry(const lafs_gemm_nt_args* g) {
  const int o = lafs_ctx_opt(g->ctx, LAFS_OPT_NT_BIG);
  if (o >= 2) return o <= 5 ? o - 1 : 1;
  int best = 1;                                             // (a taller tile wins a tie: fewer operand bytes per MFMA)
  if (geo_fill(g, 3) > geo_fill(g, best) + 30) best = 3;
  if (geo_fill(g, 4) > geo_fill(g, best) + 30) best = 4;
  return best;
}

bool lafs_big_eligible(const lafs_gemm_nt_args* g) {
  if (!lafs_ctx_opt(g->ctx, LAFS_OPT_NT_BIG)) return false;
  const int e = g->epilogue;
  if (!(e == LAFS_EPI_BF16 || e == LAFS_EPI_BF16_GELU || e == LAFS_EPI_RESID_F32 || e == LAFS_EPI_DGELU_BF16)) return false;
  if (g->splits > 1 || g->operand_f16) return false;
  if (g->K % BK != 0 || g->K < 512 || g->N < 512 || g->N % 8 != 0 || g->M < 8192) return false;
  if (g->lda % 8 != 0 || g->ldb % 8 != 0 || (g->C != nullptr && g->ldc % 8 != 0)) return false;
  if (e == LAFS_EPI_BF16_GELU && (g->C2 == nullptr || g->ldc2 % 8 != 0)) return false;
  if (e == LAFS_EPI_RESID_F32 && (g->resid == nullptr || g->ldr % 4 != 0 || g->C == nullptr)) return false;
  if (e == LAFS_EPI_DGELU_BF16 && (g->aux == nullptr || g->ldaux % 8 != 0 || g->C == nullptr)) return false;
  if (e == LAFS_EPI_BF16 && g->C == nullptr) return false;
  if ((long)g->M * g->lda * 2 >= (1L << 32) || (long)g->N * g->ldb * 2 >= (1L << 32)) return false;     // 32-bit operand offsets
  if (lafs_ctx_opt(g->ctx, LAFS_OPT_NT_BIG) >= 2) return true;
  // Which epilogues (same box, tools/lab/t_big_ab.py, us tiled -> here): the plain one; the GELU pair that saves gelu'(u) (VALU-bound:
  // 283 -> 230 at 44 160 x 2048 x 768, 129 -> 119 at 25 216 rows); GELU' (224 -> 206-212, 111 -> 104); residual + DropPath scale only
  // from three rounds on (44 160 rows: fc2 forward 209 -> 173, projection 122 -> 96; a tie at 25 216 rows = 2 rounds).  The GELU pair
  // that writes u, and the forward-only GELU, stay tiled (not measured).  How full: >= 84 % of the rounds' outputs useful; 12-stage
  // tiles (K = 768) expose the ring's fill and the epilogue at every tile change and need >= 4 rounds, or 3 that are >= 88 % full
  // (25 216 x 704 x 768 in 2 rounds: 33.5-33.9 us against 31.9 tiled; 25 216 x 2112 x 768 in 5 rounds: 84.7 against 94.0).
  const int geo = big_geometry(g);
  const long fill = geo_fill(g, geo);
  const long rounds = (geo_tiles(g, geo) + 255) / 256;
#ifdef LAFS_LAB_BIG_OLDRULE
  const bool epi_ok = e == LAFS_EPI_BF16 || (e == LAFS_EPI_BF16_GELU && g->C != nullptr && g->act == LAFS_GELU_SAVE_GRAD);
#else
  const bool epi_ok = e == LAFS_EPI_BF16 || e == LAFS_EPI_DGELU_BF16 || (e == LAFS_EPI_RESID_F32 && rounds >= 3) ||
                      (e == LAFS_EPI_BF16_GELU && g->C != nullptr && g->act == LAFS_GELU_SAVE_GRAD);
#endif
  return epi_ok && fill >= 840 && (g->K >= 1024 || rounds >= 4 || (rounds >= 3 && fill >= 880));
}

template <typename G>
static int big_launch_geo(const lafs_gemm_nt_args* g, const BArgs& a, hipStream_t stream) {
  switch (g->epilogue) {
    case LAFS_EPI_BF16: return launch<LAFS_EPI_BF16, false, G>(a, stream);
    case LAFS_EPI_BF16_GELU:
      return g->C != nullptr ? launch<LAFS_EPI_BF16_GELU, true, G>(a, stream) : launch<LAFS_EPI_BF16_GELU, false, G>(a, stream);
    case LAFS_EPI_RESID_F32: return launch<LAFS_EPI_RESID_F32, false, G>(a, stream);
    default: return launch<LAFS_EPI_DGELU_BF16, false, G>(a, stream);
  }
}

int lafs_big_launch(const lafs_gemm_nt_args* g, hipStream_t stream) {
  BArgs a = {};
  a.A = (const bf16_t*)g->A; a.B = (const bf16_t*)g->B; a.M = g->M; a.N = g->N; a.K = g->K; a.lda = g->lda; a.ldb = g->ldb;
  a.C = g->C; a.ldc = g->ldc; a.C2 = g->C2; a.ldc2 = g->ldc2;
  a.bias = g->bias; a.resid = g->resid; a.ldr = g->ldr; a.seq_scale = g->seq_scale; a.row2seq = g->row2seq;
  a.aux = (const bf16_t*)g->aux; a.ldaux = g->ldaux;
  a.drop = make_drop(g->drop_p, g->drop_seed, g->drop_step, (unsigned)g->drop_row0 * (unsigned)g->N);
  a.act = g->act;
  const int geo = big_geometry(g);
  return geo == 2 ? big_launch_geo<GeoTwo>(g, a, stream) : geo == 3 ? big_launch_geo<GeoSlim>(g, a, stream)
       : geo == 4 ? big_launch_geo<GeoFive>(g, a, stream) : big_launch_geo<GeoOne>(g, a, stream);
}
