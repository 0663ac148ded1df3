// ============================================================================================================================
// EXPERIMENT (round 2), NOT part of liblafs_hip.so -- kept for the record, see DESIGN.md section 6.
// A persistent one-workgroup-per-CU NT GEMM in the style of csrc/wgrad.hip (32x32x16 MFMA, asm LDS-DMA ring across tile
// boundaries, LDS-transposed full-line epilogue through buffer loads / stores).  Measured on MI355X against the 128x128
// 3-workgroups-per-CU kernel of csrc/gemm.hip (tools/bench_kernels.py history, ViT-S student shapes, us):
//   qkv fwd 66 vs 61 | proj fwd 38 vs 42 | fc1 fwd (GELU) 132-155 vs 131 | fc2 fwd 86 vs 84 | fc2 dgrad (GELU') 146-172 vs 122 |
//   fc1 dgrad 69 vs 66 | 4096^3 133-175 vs 141
// i.e. parity at best.  Why it does not pay here although the same structure doubled the weight-gradient kernel: these GEMMs
// have 12-stage main loops (K = 384) and VALU- / load-heavy epilogues (GELU: ~25 VALU per output; residual / GELU' operands
// straight from HBM).  With ONE wave per SIMD nothing overlaps an epilogue: its VALU work and its operand latency are serial
// with the next tile's MFMAs, whereas three 4-wave workgroups per CU overlap them for free.  Two persistent workgroups per CU
// (shapes 4/5) lose to the LDS budget (3-stage ring).  The 256x192 / DGELU instantiation also miscompares (not chased).
// ============================================================================================================================
// Persistent streaming GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * B[N,K]^T)   (every nn.Linear forward / dgrad of the trunk)
//
// Replaces the cuBLAS GEMMs behind Mlp / Attention.qkv / Attention.proj of the reference (vision_transformer.py:59-65, 75-90;
// face_pre_pro/ViT_face.py:126-137, 147-149) for the large-M shapes of the training step; lafs_gemm_nt (gemm.hip) dispatches
// here and keeps its own 128x128 kernel for small problems and the rare epilogues.
//
// The trunk's GEMMs are STREAMING problems (M = 44160 tokens, K or N = 384): the weights fit L2, the activations are read once
// and the outputs written once, so a tile's main loop (12 stages at K = 384) is no longer than its prologue + epilogue.  Design:
//   * one PERSISTENT 4-wave workgroup per CU (one wave per SIMD, 512 registers) walks the output tiles b, b+G, b+2G, ...;
//     a wave owns (32 FM) x (32 FN) of v_mfma_f32_32x32x16_bf16 blocks, the workgroup a (64 FM) x (64 FN) tile (256 x 256);
//   * both operands arrive by LDS-DMA into ONE ring of 32-deep stages that runs across tile boundaries: while a tile's
//     epilogue is busy, the first stages of the next tile are already in flight (counted s_waitcnt vmcnt, raw s_barrier);
//   * main loop software-pipelined over 16-deep half stages with two fragment register sets; the fragment reads of the next
//     half stage and the DMA of the next ring stage are issued from inside the MFMA groups (wgrad.hip has the PMC story);
//   * epilogue through LDS: a wave transposes its accumulators in 32x64 fp32 pieces so that every lane ends up with 8
//     CONSECUTIVE output columns of one row -- bias / GELU / GELU' / residual / DropPath / dropout then read and write global
//     memory in full 128-byte lines per row (16-32 B per lane), whatever the MFMA register layout.
#include <type_traits>
#include "common.hpp"
#include "lafs_hip.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef float f32x4v_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v_t __attribute__((ext_vector_type(4)));

struct SArgs {
  const bf16_t* A; const bf16_t* B;
  int M, N, K, lda, ldb;
  void* C; int ldc; void* C2; int ldc2;
  const float* bias; const float* resid; int ldr;
  const float* seq_scale; const int* row2seq;
  const bf16_t* aux; int ldaux;
  DropCfg drop;
  int tiles_m, tiles_n, grid;
};

__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int SBK = 32;                  // k-depth of a ring stage

// ABL (tools/lab only): 1 = no DMA after the prologue, 2 = no epilogue stores, 4 = no MFMA
// PF: 32-column accumulator blocks per epilogue transpose piece (1 or 2; the piece is 32 rows x 32 PF columns of fp32 per wave)
// WPC: persistent workgroups per CU (2: the epilogue of one overlaps the main loop of the other; needs <= 80 KiB LDS, 256 registers)
template <int EPI, int FM, int FN, int NS, int PF, int WPC = 1, int ABL = 0>
__global__ __launch_bounds__(256, WPC) void gemm_stream_kernel(SArgs p) {
  constexpr int BMT = 64 * FM, BNT = 64 * FN;
  constexpr int NR = FM + FN;                                  // LDS-DMA instructions per thread and stage (4 KiB each)
  constexpr int STAGE = (BMT + BNT) * 64;
  constexpr int SROW = 32 * PF + 4;                            // floats per piece row (pad: conflict-free 16-byte writes)
  constexpr int STG = 32 * SROW * 4;                           // one wave's transpose piece, bytes
  static_assert(FN % PF == 0 && (PF == 1 || PF == 2), "piece width");
  static_assert(NS >= 3 && NS * STAGE + 4 * STG <= 160 * 1024 / WPC, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE + 4 * STG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5;
  // blocks b, b+8, ... share an XCD: give each XCD a contiguous run of tile numbers in every round (the n-tiles of one m-tile
  // read the same activation rows)
  const int wid = (blockIdx.x & 7) * (p.grid >> 3) + (blockIdx.x >> 3);
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nk = p.K / SBK;
  const int my_tiles = wid < ntiles ? (ntiles - wid + p.grid - 1) / p.grid : 0;
  const int total = my_tiles * nk;                              // ring stages this workgroup consumes
  if (total == 0) return;

  // ---- LDS-DMA producer: its own (tile, k) cursor, NS-2 .. NS-1 stages ahead of the MFMAs, across tile boundaries.
  // Stage image: rows 0..BMT-1 = activation rows, BMT.. = weight rows, 64 B each; 16-byte piece c of row r sits at piece
  // c ^ ((r >> 2) & 3) (ds_read_b128 fragment reads conflict-free); the image is lane-linear, so the swizzle is on the source.
  const int drow = tid >> 2;                                     // row inside a 64-row instruction block
  const int dk = ((tid & 3) ^ ((tid >> 4) & 3)) * 8;            // source k offset (elements) of this thread's piece
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem)) + wave * 1024;
  const bf16_t* gp[NR];
  int d_tile = 0, d_k = 0; unsigned d_slot = 0; int d_issued = 0;
  auto dma_tile = [&]() {                                       // (re)compute the source rows for tile number d_tile of this block
    const int t = wid + d_tile * p.grid;
    const int m0 = (t / p.tiles_n) * BMT, n0 = (t % p.tiles_n) * BNT;
#pragma unroll
    for (int i = 0; i < FM; ++i) gp[i] = p.A + (long)min(m0 + i * 64 + drow, p.M - 1) * p.lda + dk;
#pragma unroll
    for (int i = 0; i < FN; ++i) gp[FM + i] = p.B + (long)min(n0 + i * 64 + drow, p.N - 1) * p.ldb + dk;
  };
  auto dma_piece = [&](int i) {
    lds_dma16_m0(gp[i], smem_base + d_slot + i * 4096);
    gp[i] += SBK;
  };
  auto dma_advance = [&]() {
    ++d_issued;
    d_slot += STAGE; if (d_slot == NS * STAGE) d_slot = 0;
    if (++d_k == nk) { d_k = 0; ++d_tile; if (d_issued < total) dma_tile(); }
  };
  auto dma_stage = [&]() {
#pragma unroll
    for (int i = 0; i < NR; ++i) dma_piece(i);
    dma_advance();
  };
  dma_tile();

  // ---- fragment addressing: 32x32x16 operand = lane (row = lane & 31, k half = lane >> 5) holds 8 consecutive k
  const int sw = (lane >> 2) & 3;
  const int fb0 = (lane & 31) * 64 + ((h ^ sw) << 4);           // k-step 0; k-step 1 is fb0 ^ 32
  const int offX = wm * FM * 2048 + fb0, offW = BMT * 64 + wn * FN * 2048 + fb0;
  auto load_one = [&](bf16x8_t (&fx)[FM], bf16x8_t (&fw)[FN], const unsigned char* st, int ks, int f) {
    if (f < FM) fx[f < FM ? f : 0] = *reinterpret_cast<const bf16x8_t*>(st + ((offX + f * 2048) ^ (ks * 32)));
    else fw[f < FM ? 0 : f - FM] = *reinterpret_cast<const bf16x8_t*>(st + ((offW + (f - FM) * 2048) ^ (ks * 32)));
  };
  auto load_frags = [&](bf16x8_t (&fx)[FM], bf16x8_t (&fw)[FN], const unsigned char* st, int ks) {
#pragma unroll
    for (int f = 0; f < FM + FN; ++f) load_one(fx, fw, st, ks, f);
  };

  f32x16_t acc[FM][FN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int b = 0; b < FN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  };
  // MFMAs of one half stage; the reads of the next half stage (READS) and the DMA pieces of the next ring stage (DMA) are spread
  // through the MFMAs' shadows, order pinned
  auto group = [&](const bf16x8_t (&fx)[FM], const bf16x8_t (&fw)[FN], bf16x8_t (&nx)[FM], bf16x8_t (&nw)[FN],
                   const unsigned char* nst, int nks, auto READS, auto DMA) {
    constexpr int NM = FM * FN, NF = FM + FN;
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int b = 0; b < FN; ++b) {
        const int k = a * FN + b;
        if (!(ABL & 4)) acc[a][b] = mfma32(fw[b], fx[a], acc[a][b]);          // D[n][m]: weights are the A operand
        if (decltype(READS)::value) {
#pragma unroll
          for (int f = 0; f < NF; ++f)
            if (f * NM / NF == k) load_one(nx, nw, nst, nks, f);
        }
        if (decltype(DMA)::value && !(ABL & 1)) {
#pragma unroll
          for (int i = 0; i < NR; ++i)
            if (i * NM / NR == k) dma_piece(i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  // stage g is complete in LDS once at most `ahead` younger stages are in flight (loads retire in order; younger STORES of an
  // epilogue only make the wait stricter)
  auto wait_stage = [&](int ahead) {
    if (ahead >= 2 && NS >= 5) wait_vm<2 * NR>();
    else if (ahead >= 1 && NS >= 4) wait_vm<NR>();
    else wait_vm<0>();
  };

  // ---- epilogue of one tile
  float* const stg = reinterpret_cast<float*>(smem + NS * STAGE + wave * STG);
    // All global accesses of the epilogue are BUFFER loads / stores: an out-of-range row or column gets a byte offset past the
  // end of the buffer (loads return 0, stores are dropped by the hardware), so the code is straight-line -- no exec-masked
  // branches for hipcc to serialise with s_waitcnt vmcnt(0), and every lane executes the same number of memory instructions.
  const unsigned esz = (EPI == LAFS_EPI_RESID_F32 || EPI == LAFS_EPI_F32) ? 4u : 2u;
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, p.C ? (int)((unsigned)p.M * (unsigned)p.ldc * esz) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rC2 = __builtin_amdgcn_make_buffer_rsrc(p.C2, 0, p.C2 ? (int)((unsigned)p.M * (unsigned)p.ldc2 * 2u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid), 0, p.resid ? (int)((unsigned)p.M * (unsigned)p.ldr * 4u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.aux), 0, p.aux ? (int)((unsigned)p.M * (unsigned)p.ldaux * 2u) : 0, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  auto epilogue = [&](int t) {
    const int m0 = (t / p.tiles_n) * BMT + wm * FM * 32, n0 = (t % p.tiles_n) * BNT + wn * FN * 32;
    constexpr int LPR = 4 * PF, RPP = 64 / LPR, NPASS = 32 / RPP;   // lanes per piece row, rows per pass, passes per piece
    const int er = lane / LPR, ec = (lane % LPR) * 8;             // read-back: 8 consecutive columns per lane
#pragma unroll
    for (int j = 0; j < FN / PF; ++j) {
      const int n = n0 + j * 32 * PF + ec;
      const bool ncol = n < p.N;                                   // N % 8 == 0: the lane's 8 columns are all in or all out
      float bias[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bias[e] = 0.f;
      if (p.bias != nullptr && EPI != LAFS_EPI_DGELU_BF16) {
        const float* bp = p.bias + (ncol ? n : 0);
        const f32x4v_t b0 = *reinterpret_cast<const f32x4v_t*>(bp), b1 = *reinterpret_cast<const f32x4v_t*>(bp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias[e] = b0[e]; bias[4 + e] = b1[e]; }
      }
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        // operands the epilogue reads, requested before the LDS round trip
        u32x4v_t r0[NPASS], r1[NPASS], ax[NPASS]; float sc[NPASS]; bool ok[NPASS]; unsigned em[NPASS];
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
          const int m = m0 + a * 32 + q * RPP + er;
          ok[q] = ncol && m < p.M;
          em[q] = (unsigned)m;
          sc[q] = 1.f;
          if (EPI == LAFS_EPI_RESID_F32) {
            const unsigned off = ok[q] ? ((unsigned)m * (unsigned)p.ldr + (unsigned)n) * 4u : OOB;
            r0[q] = __builtin_amdgcn_raw_buffer_load_b128(rR, off, 0, 0);
            r1[q] = __builtin_amdgcn_raw_buffer_load_b128(rR, off + 16, 0, 0);
            if (p.seq_scale != nullptr) sc[q] = p.seq_scale[p.row2seq[min(m, p.M - 1)]];
          }
          if (EPI == LAFS_EPI_DGELU_BF16)
            ax[q] = __builtin_amdgcn_raw_buffer_load_b128(rX, ok[q] ? ((unsigned)m * (unsigned)p.ldaux + (unsigned)n) * 2u : OOB, 0, 0);
        }
        // transpose: accumulator register r of a block is column (lane & 31) = token row, row 8 (r >> 2) + 4h + (r & 3) = output column
#pragma unroll
        for (int bb = 0; bb < PF; ++bb)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const f32x16_t& c = acc[a][PF * j + bb];
            *reinterpret_cast<f32x4v_t*>(stg + (lane & 31) * SROW + bb * 32 + 8 * rr + 4 * h) =
                f32x4v_t{c[4 * rr], c[4 * rr + 1], c[4 * rr + 2], c[4 * rr + 3]};
          }
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
          const unsigned m = em[q];
          const f32x4v_t v0 = *reinterpret_cast<const f32x4v_t*>(stg + (q * RPP + er) * SROW + ec);
          const f32x4v_t v1 = *reinterpret_cast<const f32x4v_t*>(stg + (q * RPP + er) * SROW + ec + 4);
          float w[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { w[e] = v0[e] + bias[e]; w[4 + e] = v1[e] + bias[4 + e]; }
          const bool st = ok[q] && !(ABL & 2);
          const unsigned idx0 = m * (unsigned)p.N + (unsigned)n;                    // dropout counter of element (m, n)
          if (EPI == LAFS_EPI_BF16 || EPI == LAFS_EPI_BF16_GELU) {
            if (EPI == LAFS_EPI_BF16 || p.C != nullptr)
              __builtin_amdgcn_raw_buffer_store_b128(u32x4v_t{pack_bf2(w[0], w[1]), pack_bf2(w[2], w[3]), pack_bf2(w[4], w[5]), pack_bf2(w[6], w[7])},
                                                     rC, st ? (m * (unsigned)p.ldc + (unsigned)n) * 2u : OOB, 0, 0);
            if (EPI == LAFS_EPI_BF16_GELU) {
              float g[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) g[e] = gelu_f(w[e]);
              if (p.drop.thresh) {
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] *= drop_mult(p.drop, idx0 + e);
              }
              __builtin_amdgcn_raw_buffer_store_b128(u32x4v_t{pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]), pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7])},
                                                     rC2, st ? (m * (unsigned)p.ldc2 + (unsigned)n) * 2u : OOB, 0, 0);
            }
          } else if (EPI == LAFS_EPI_DGELU_BF16) {
            const u32x4v_t a4 = ax[q];
            w[0] *= gelu_grad_f(bf_lo(a4[0])); w[1] *= gelu_grad_f(bf_hi(a4[0])); w[2] *= gelu_grad_f(bf_lo(a4[1])); w[3] *= gelu_grad_f(bf_hi(a4[1]));
            w[4] *= gelu_grad_f(bf_lo(a4[2])); w[5] *= gelu_grad_f(bf_hi(a4[2])); w[6] *= gelu_grad_f(bf_lo(a4[3])); w[7] *= gelu_grad_f(bf_hi(a4[3]));
            if (p.drop.thresh) {
#pragma unroll
              for (int e = 0; e < 8; ++e) w[e] *= drop_mult(p.drop, idx0 + e);
            }
            __builtin_amdgcn_raw_buffer_store_b128(u32x4v_t{pack_bf2(w[0], w[1]), pack_bf2(w[2], w[3]), pack_bf2(w[4], w[5]), pack_bf2(w[6], w[7])},
                                                   rC, st ? (m * (unsigned)p.ldc + (unsigned)n) * 2u : OOB, 0, 0);
          } else {                                                  // fp32 outputs
            if (EPI == LAFS_EPI_RESID_F32) {
              if (p.drop.thresh) {
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] *= drop_mult(p.drop, idx0 + e);
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                w[e] = __uint_as_float(r0[q][e]) + sc[q] * w[e];
                w[4 + e] = __uint_as_float(r1[q][e]) + sc[q] * w[4 + e];
              }
            }
            const unsigned off = st ? (m * (unsigned)p.ldc + (unsigned)n) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4v_t{__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3])}, rC, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4v_t{__float_as_uint(w[4]), __float_as_uint(w[5]), __float_as_uint(w[6]), __float_as_uint(w[7])}, rC, off + 16, 0, 0);
          }
        }
      }
    }
  };

  // ---- main pipeline over the flat sequence of ring stages g = 0 .. total-1 (tile = g / nk)
#pragma unroll
  for (int i = 0; i < NS - 2; ++i)
    if (d_issued < total) dma_stage();
  bf16x8_t x0[FM], w0[FN], x1[FM], w1[FN];
  wait_stage(min(total - 1, NS - 3));
  __builtin_amdgcn_s_barrier();
  if (d_issued < total) dma_stage();
  load_frags(x0, w0, smem, 0);
  unsigned rslot = 0;
  int g = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    zero_acc();
    for (int k = 0; k < nk; ++k, ++g) {
      const unsigned char* st = smem + rslot;
      group(x0, w0, x1, w1, st, 1, Yes{}, No{});
      const bool more = g + 1 < total;
      if (more) {
        wait_stage(min(total - 2 - g, NS - 3));
        __builtin_amdgcn_s_barrier();                    // everyone's pieces of stage g+1 landed; the slot of stage g-1 is free
        __builtin_amdgcn_sched_barrier(0);
        rslot += STAGE; if (rslot == NS * STAGE) rslot = 0;
      }
      const unsigned char* nst = smem + rslot;
      if (d_issued < total && !(ABL & 1)) {              // the ring stage that is due rides in this group
        group(x1, w1, x0, w0, nst, 0, Yes{}, Yes{});
        dma_advance();
      } else if (more) {
        group(x1, w1, x0, w0, nst, 0, Yes{}, No{});
      } else {
        group(x1, w1, x0, w0, nst, 0, No{}, No{});
      }
    }
    epilogue(wid + ti * p.grid);
  }
}

template <int EPI, int FM, int FN, int NS, int PF, int WPC = 1>
int launch(SArgs a, hipStream_t s) {
  a.tiles_m = ceil_div(a.M, 64 * FM); a.tiles_n = ceil_div(a.N, 64 * FN);
  const int nt = a.tiles_m * a.tiles_n;
  a.grid = nt >= 256 * WPC ? 256 * WPC : ((nt + 7) & ~7);
  hipLaunchKernelGGL((gemm_stream_kernel<EPI, FM, FN, NS, PF, WPC>), dim3(a.grid), dim3(256), 0, s, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

// Tile shape: fewest rounds of 256 workgroups weighted by the tile's MFMA count (the persistent grid walks tiles round by round;
// a last round that is nearly empty costs as much as a full one).  shape < 0: automatic.
template <int EPI>
int dispatch(const SArgs& a, int shape, hipStream_t s) {
  static const int cand[4][2] = {{4, 3}, {4, 2}, {2, 3}, {2, 2}};
  if (shape < 0) {
    long best = -1;
    for (int c = 0; c < 4; ++c) {
      const long tiles = (long)ceil_div(a.M, 64 * cand[c][0]) * ceil_div(a.N, 64 * cand[c][1]);
      const long rounds = (tiles + 255) / 256;
      long cost = rounds * cand[c][0] * cand[c][1] * 100;
      if (cand[c][0] * cand[c][1] <= 4) cost += cost / 4;         // 128x128: most operand traffic per flop
      if (best < 0 || cost < best) { best = cost; shape = c; }
    }
  }
  switch (shape) {
    case 0: return launch<EPI, 4, 3, 5, 1>(a, s);                   // 256 x 192, 5-stage ring
    case 1: return launch<EPI, 4, 2, 5, 2>(a, s);                   // 256 x 128
    case 2: return launch<EPI, 2, 3, 6, 1>(a, s);                   // 128 x 192
    case 3: return launch<EPI, 2, 2, 6, 2>(a, s);                   // 128 x 128
    case 4: return launch<EPI, 2, 3, 3, 1, 2>(a, s);                // 128 x 192, two workgroups per CU
    default: return launch<EPI, 2, 2, 3, 1, 2>(a, s);               // 128 x 128, two workgroups per CU
  }
}

}  // namespace

// Returns LAFS_OK when the problem was launched here, 1 when it is not eligible (lafs_gemm_nt then uses its own kernel).
// force: accept small problems too (tests); shape: -1 automatic, 0..3 = 256x192 / 256x128 / 128x192 / 128x128 tiles.
int lafs_gemm_stream_try(const lafs_gemm_nt_args* g, const DropCfg& drop, int force, int shape, hipStream_t stream) {
  const int e = g->epilogue;
  if (!(e == LAFS_EPI_BF16 || e == LAFS_EPI_BF16_GELU || e == LAFS_EPI_RESID_F32 || e == LAFS_EPI_F32 || e == LAFS_EPI_DGELU_BF16)) return 1;
  if (g->N % 8 != 0 || g->K % 32 != 0 || g->K < 32) return 1;
  if (!force && (g->M < 4096 || g->N < 128 || (long)g->M * g->N < (1L << 22))) return 1;
  if (g->ldc % 8 != 0 || (g->C2 && g->ldc2 % 8 != 0) || (g->resid && g->ldr % 4 != 0) || (g->aux && g->ldaux % 8 != 0)) return 1;
  const long lim = 0x7FFFFF00L;                                    // the epilogue addresses its operands through 32-bit buffer offsets
  if ((long)g->M * g->ldc * 4 > lim || (long)g->M * g->ldc2 * 2 > lim || (long)g->M * g->ldr * 4 > lim || (long)g->M * g->ldaux * 2 > lim) return 1;
  if (e == LAFS_EPI_BF16_GELU && g->C2 == nullptr) return 1;
  SArgs a = {};
  a.A = (const bf16_t*)g->A; a.B = (const bf16_t*)g->B; a.M = g->M; a.N = g->N; a.K = g->K; a.lda = g->lda; a.ldb = g->ldb;
  a.C = g->C; a.ldc = g->ldc; a.C2 = g->C2; a.ldc2 = g->ldc2; a.bias = g->bias; a.resid = g->resid; a.ldr = g->ldr;
  a.seq_scale = g->seq_scale; a.row2seq = g->row2seq; a.aux = (const bf16_t*)g->aux; a.ldaux = g->ldaux; a.drop = drop;
  switch (e) {
    case LAFS_EPI_BF16: return dispatch<LAFS_EPI_BF16>(a, shape, stream);
    case LAFS_EPI_BF16_GELU: return dispatch<LAFS_EPI_BF16_GELU>(a, shape, stream);
    case LAFS_EPI_RESID_F32: return dispatch<LAFS_EPI_RESID_F32>(a, shape, stream);
    case LAFS_EPI_F32: return dispatch<LAFS_EPI_F32>(a, shape, stream);
    default: return dispatch<LAFS_EPI_DGELU_BF16>(a, shape, stream);
  }
}
