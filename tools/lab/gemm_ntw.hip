// Wide-tile NT GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * W[N,K]^T)  for the long-K, wide-N linears of Part-fViT / ViT-B
// (reference face_pre_pro/ViT_face.py:122-182: qkv 768 -> 2112, proj 704 -> 768, FeedForward 768 -> 2048 -> 768, and their input
// gradients) -- the NT twin of the weight-gradient kernel (wgrad.hip): ONE workgroup of 4 waves per CU, one wave per SIMD with
// the whole register file, a 192 x 192 output tile (3 x 3 blocks of v_mfma_f32_32x32x16_bf16 per wave), both operands through a
// six-stage LDS-DMA ring with counted s_waitcnt vmcnt, the k-loop software-pipelined by hand (fragment reads and DMA pieces of
// the next half stage sit in the MFMAs' shadows).  Against the 128 x 128 / two-workgroups-per-CU kernel of gemm.hip an operand
// byte fetched from L2 feeds 1.5x the MFMA work and a barrier covers 18 instead of 16 (half as long) MFMAs.
//
// LDS image of a stage: [192 weight rows | 192 token rows] x 64 bytes (32 k); 16-byte chunk c of row r sits at chunk position
// c ^ ((r >> 2) & 3) (swizzle on the SOURCE column, the image is lane-linear for the DMA): the ds_read_b128 fragment reads
// (32 rows x one chunk per half-wave) are conflict-free.  Weight rows are permuted inside each 32-row block so that a lane's 16
// accumulator registers of a block are 16 CONSECUTIVE output columns of its token row.
// Epilogue: the fp32 tile is staged through the (then free) ring as [token][column] and leaves in 16-byte pieces, consecutive
// lanes on consecutive pieces of a row -- full-line writes whatever the epilogue; bias, GELU / GELU', residual + DropPath scale
// and the counter-based dropout masks are applied per piece exactly as gemm.hip's epilogues do (same indices, same arithmetic).
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "gemm_ntw.hpp"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int KS = 32;                    // k per ring stage (64-byte rows)
constexpr int NS = 6;                     // ring depth: 6 x 24 KiB = the fp32 192 x 192 tile the epilogue stages

struct WArgs {
  const bf16_t* A; const bf16_t* B;
  int M, N, K, lda, ldb;
  void* C; int ldc; void* C2; int ldc2;
  const float* bias; const float* resid; int ldr;
  const float* seq_scale; const int* row2seq;
  const bf16_t* aux; int ldaux;
  DropCfg drop;
  int act;
  int tiles_n, n_tiles, nblk;
};

template <int EPI> struct WEpi {
  static constexpr bool f32_out = (EPI == LAFS_EPI_RESID_F32);
  static constexpr int VPL = f32_out ? 4 : 8;          // output columns per 16-byte piece
};

template <int EPI, int FA, int FB>
__global__ __launch_bounds__(256, 1) void gemm_ntw_kernel(WArgs p) {
  constexpr int NTH = 256, WM = 2, WN = 2;
  constexpr int T1 = 32 * FA * WM, T2 = 32 * FB * WN;                // columns (weight rows) x token rows of the tile
  constexpr int ROWS = T1 + T2;
  constexpr int STAGE = ROWS * 64;
  constexpr int NCH = ROWS * 4;                                      // 16-byte pieces per stage
  static_assert(NCH % NTH == 0, "pieces must divide evenly over the threads");
  constexpr int NR = NCH / NTH;                                      // LDS-DMA instructions per thread and stage
  constexpr int CP = T1 + 4;                                         // pitch (floats) of the staged output tile
  constexpr int SMEM = (NS * STAGE > T2 * CP * 4) ? NS * STAGE : T2 * CP * 4;
  static_assert(SMEM <= 160 * 1024, "ring / staged tile does not fit the LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  const DropCfg drop = drop_resolve(p.drop);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // blocks b, b+8, ... share an XCD: contiguous runs of tiles per XCD, the column tiles of one row tile (same token rows) adjacent
  const int per = p.nblk >> 3;
  const int id = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (id >= p.n_tiles) return;
  const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
  const int n0 = tn * T1, m0 = tm * T2;
  const int nk = p.K / KS;

  // ---- LDS-DMA: piece q = i * NTH + tid of a stage is (row q >> 2, chunk position q & 3) and lands at LDS byte 16 q
  const bf16_t* gp[NR];
  unsigned ddst[NR];
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int q = i * NTH + tid, rho = q >> 2, c = (q & 3) ^ ((rho >> 2) & 3);
    if (rho < T1) {                                                  // weight row: permuted inside its 32-row block
      const int s = rho & 31;
      const int n = n0 + (rho & ~31) + 16 * ((s >> 2) & 1) + 4 * (s >> 3) + (s & 3);
      gp[i] = p.B + (size_t)min(n, p.N - 1) * p.ldb + c * 8;
    } else {
      gp[i] = p.A + (size_t)min(m0 + rho - T1, p.M - 1) * p.lda + c * 8;
    }
    ddst[i] = smem_base + __builtin_amdgcn_readfirstlane((q & ~63) * 16);
  }
  int ti = 0; unsigned islot = 0;
  auto issue_piece = [&](int i) {
    lds_dma16_m0(gp[i], ddst[i] + islot);
    gp[i] += KS;
  };
  auto issue_done = [&]() { ++ti; islot += STAGE; if (islot == NS * STAGE) islot = 0; };
  auto issue = [&]() {
#pragma unroll
    for (int i = 0; i < NR; ++i) issue_piece(i);
    issue_done();
  };

  f32x16_t acc[FA][FB];
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- fragment addressing: lane (r = lane & 31, h = lane >> 5) holds k = 16 half + 8 h .. + 7 of row r: chunk 2 half + h
  const int r32 = lane & 31, h = lane >> 5, sw = (r32 >> 2) & 3;
  int offW[FA], offA[FB];
#pragma unroll
  for (int a = 0; a < FA; ++a) offW[a] = ((wm * FA + a) * 32 + r32) * 64;
#pragma unroll
  for (int b = 0; b < FB; ++b) offA[b] = (T1 + (wn * FB + b) * 32 + r32) * 64;
  const int coff[2] = {((0 + h) ^ sw) * 16, ((2 + h) ^ sw) * 16};

  auto load_one = [&](bf16x8_t (&fw)[FA], bf16x8_t (&fa)[FB], const unsigned char* sk, int half, int f) {   // weight blocks first
    const unsigned char* q = sk + (f < FA ? offW[f < FA ? f : 0] : offA[f < FA ? 0 : f - FA]) + coff[half];
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(q);
    if (f < FA) fw[f < FA ? f : 0] = v;
    else fa[f < FA ? 0 : f - FA] = v;
  };
  auto load_frags = [&](bf16x8_t (&fw)[FA], bf16x8_t (&fa)[FB], const unsigned char* sk, int half) {
#pragma unroll
    for (int f = 0; f < FA + FB; ++f) load_one(fw, fa, sk, half, f);
  };
  // One half stage of MFMAs on (fw, fa); the fragment reads of the NEXT half stage into (nw, na) and the NR LDS-DMA pieces of the
  // ring stage that is due are spread through the MFMAs' shadows, the order pinned by sched_barrier (see wgrad.hip)
  auto group = [&](const bf16x8_t (&fw)[FA], const bf16x8_t (&fa)[FB], bf16x8_t (&nw)[FA], bf16x8_t (&na)[FB],
                   const unsigned char* nsk, int nhalf, auto READS, auto DMA) {
    constexpr int NM = FA * FB, NF = FA + FB;
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
      for (int b = 0; b < FB; ++b) {
        const int k = a * FB + b;
        acc[a][b] = mfma32(fw[a], fa[b], acc[a][b]);
        if (decltype(READS)::value) {
#pragma unroll
          for (int f = 0; f < NF; ++f)
            if (f * NM / NF == k) load_one(nw, na, nsk, nhalf, f);
        }
        if (decltype(DMA)::value) {
#pragma unroll
          for (int i = 0; i < NR; ++i)
            if (i * NM / NR == k) issue_piece(i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  auto wait_stage = [&](int ahead) {                  // a stage is complete once at most `ahead` younger stages are in flight
    if (ahead >= 3) wait_vm<3 * NR>();
    else if (ahead == 2) wait_vm<2 * NR>();
    else if (ahead == 1) wait_vm<NR>();
    else wait_vm<0>();
  };

#pragma unroll
  for (int t = 0; t < NS - 2; ++t)
    if (ti < nk) issue();
  bf16x8_t w0[FA], a0[FB], w1[FA], a1[FB];
  wait_stage(min(nk - 1, NS - 3));
  __builtin_amdgcn_s_barrier();
  if (ti < nk) issue();
  load_frags(w0, a0, smem, 0);
  unsigned rslot = 0;
  int t = 0;
  for (; t < nk - NS; ++t) {                          // steady state: full ring
    group(w0, a0, w1, a1, smem + rslot, 1, Yes{}, No{});
    wait_vm<(NS - 3) * NR>();
    __builtin_amdgcn_s_barrier();                    // everyone's pieces of stage t+1 landed; the slot of stage t-1 is free
    __builtin_amdgcn_sched_barrier(0);
    rslot += STAGE; if (rslot == NS * STAGE) rslot = 0;
    group(w1, a1, w0, a0, smem + rslot, 0, Yes{}, Yes{});
    issue_done();
  }
  for (; t < nk; ++t) {                               // ring drains
    group(w0, a0, w1, a1, smem + rslot, 1, Yes{}, No{});
    if (t + 1 < nk) {
      wait_stage(min(nk - 2 - t, NS - 3));
      __builtin_amdgcn_s_barrier();
      if (ti < nk) issue();
      rslot += STAGE; if (rslot == NS * STAGE) rslot = 0;
      group(w1, a1, w0, a0, smem + rslot, 0, Yes{}, No{});
    } else {
      group(w1, a1, w0, a0, smem, 0, No{}, No{});
    }
  }

  // ---- stage the fp32 tile through the ring: lane (c = lane & 31, h) of block (a, b) holds token row c, columns 16 h .. 16 h + 15
  __builtin_amdgcn_s_barrier();                      // every wave is out of the ring
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      float* dst = ct + ((wn * FB + b) * 32 + r32) * CP + (wm * FA + a) * 32 + 16 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4v_t*>(dst + 4 * q) = f32x4v_t{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
    }
  __syncthreads();

  // ---- epilogue, piece by piece: consecutive threads on consecutive 16-byte pieces of a row
  constexpr int VPL = WEpi<EPI>::VPL, PR = T1 / VPL;
  constexpr int NPIECE = T2 * PR;
  const bool save_grad = (p.act == LAFS_GELU_SAVE_GRAD);
#pragma unroll 2
  for (int idx = tid; idx < NPIECE; idx += NTH) {
    const int row = idx / PR, j = idx - row * PR;
    const int m = m0 + row, n = n0 + j * VPL;
    if (m >= p.M || n >= p.N) continue;                             // (N is a multiple of 8: a piece is inside or outside)
    float v[VPL];
    const float* src = ct + row * CP + j * VPL;
#pragma unroll
    for (int e = 0; e < VPL; e += 4) {
      const f32x4v_t x = *reinterpret_cast<const f32x4v_t*>(src + e);
      v[e] = x[0]; v[e + 1] = x[1]; v[e + 2] = x[2]; v[e + 3] = x[3];
    }
    if (p.bias != nullptr && EPI != LAFS_EPI_DGELU_BF16) {
#pragma unroll
      for (int e = 0; e < VPL; ++e) v[e] += p.bias[n + e];
    }
    const unsigned didx = (unsigned)m * (unsigned)p.N + (unsigned)n;
    if constexpr (EPI == LAFS_EPI_RESID_F32) {
      const f32x4v_t rs = *reinterpret_cast<const f32x4v_t*>(p.resid + (size_t)m * p.ldr + n);
      const float sc = p.seq_scale != nullptr ? p.seq_scale[p.row2seq[m]] : 1.0f;
      if (drop.thresh) {
#pragma unroll
        for (int e = 0; e < VPL; ++e) v[e] *= drop_mult(drop, didx + e);
      }
      const f32x4v_t o = {rs[0] + sc * v[0], rs[1] + sc * v[1], rs[2] + sc * v[2], rs[3] + sc * v[3]};
      *reinterpret_cast<f32x4v_t*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = o;
    } else if constexpr (EPI == LAFS_EPI_BF16) {
      const u32x4_t o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
      *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) = o;
    } else if constexpr (EPI == LAFS_EPI_BF16_GELU) {
      if (p.C != nullptr) {                                          // the pre-activation u, or gelu'(u) (LAFS_GELU_SAVE_GRAD)
        float w[VPL];
#pragma unroll
        for (int e = 0; e < VPL; ++e) w[e] = save_grad ? gelu_grad_f(v[e]) : v[e];
        const u32x4_t o = {pack_bf2(w[0], w[1]), pack_bf2(w[2], w[3]), pack_bf2(w[4], w[5]), pack_bf2(w[6], w[7])};
        *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) = o;
      }
      float g[VPL];
#pragma unroll
      for (int e = 0; e < VPL; ++e) g[e] = gelu_f(v[e]);
      if (drop.thresh) {
#pragma unroll
        for (int e = 0; e < VPL; ++e) g[e] *= drop_mult(drop, didx + e);
      }
      const u32x4_t o2 = {pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]), pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7])};
      *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(p.C2) + (size_t)m * p.ldc2 + n) = o2;
    } else {                                                         // LAFS_EPI_DGELU_BF16
      const uint4 a4 = *reinterpret_cast<const uint4*>(p.aux + (size_t)m * p.ldaux + n);
      const float ax[8] = {bf_lo(a4.x), bf_hi(a4.x), bf_lo(a4.y), bf_hi(a4.y), bf_lo(a4.z), bf_hi(a4.z), bf_lo(a4.w), bf_hi(a4.w)};
#pragma unroll
      for (int e = 0; e < VPL; ++e) v[e] *= save_grad ? ax[e] : gelu_grad_f(ax[e]);
      if (drop.thresh) {
#pragma unroll
        for (int e = 0; e < VPL; ++e) v[e] *= drop_mult(drop, didx + e);
      }
      const u32x4_t o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
      *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) = o;
    }
  }
}

template <int EPI>
int launch(const lafs_gemm_nt_args* g, hipStream_t stream) {
  WArgs a = {};
  a.A = (const bf16_t*)g->A; a.B = (const bf16_t*)g->B;
  a.M = g->M; a.N = g->N; a.K = g->K; a.lda = g->lda; a.ldb = g->ldb;
  a.C = g->C; a.ldc = g->ldc; a.C2 = g->C2; a.ldc2 = g->ldc2;
  a.bias = g->bias; a.resid = g->resid; a.ldr = g->ldr;
  a.seq_scale = g->seq_scale; a.row2seq = g->row2seq;
  a.aux = (const bf16_t*)g->aux; a.ldaux = g->ldaux;
  a.drop = make_drop(g->drop_p, g->drop_seed, g->drop_step, (unsigned)g->drop_row0 * (unsigned)g->N);
  a.act = g->act;
  a.tiles_n = ceil_div(g->N, 192);
  a.n_tiles = a.tiles_n * ceil_div(g->M, 192);
  a.nblk = (a.n_tiles + 7) & ~7;
  hipLaunchKernelGGL((gemm_ntw_kernel<EPI, 3, 3>), dim3(a.nblk), dim3(256), 0, stream, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

}  // namespace

// Shapes this kernel takes over from the tiled kernel: long reductions onto wide outputs with enough tiles to fill the chip in
// whole rounds (the Part-fViT / ViT-B linears at training batch sizes).  LAFS_NTW=0 switches it off (A/B).
bool lafs_ntw_eligible(const lafs_gemm_nt_args* g) {
  static const int mode = [] { const char* v = getenv("LAFS_NTW"); return v == nullptr ? 1 : atoi(v); }();
  if (mode == 0) return false;
  const int e = g->epilogue;
  if (e != LAFS_EPI_BF16 && e != LAFS_EPI_BF16_GELU && e != LAFS_EPI_RESID_F32 && e != LAFS_EPI_DGELU_BF16) return false;
  if (g->K % 32 != 0 || g->K < 512 || g->N % 8 != 0 || g->N < 576) return false;
  if (g->ldc % 8 != 0 || (g->C2 != nullptr && g->ldc2 % 8 != 0) || (g->resid != nullptr && g->ldr % 4 != 0)) return false;
  if (g->aux != nullptr && g->ldaux % 8 != 0) return false;
  if (e == LAFS_EPI_BF16_GELU && g->C2 == nullptr) return false;
  if (e == LAFS_EPI_RESID_F32 && (g->resid == nullptr || (g->seq_scale != nullptr && g->row2seq == nullptr))) return false;
  if (e == LAFS_EPI_DGELU_BF16 && g->aux == nullptr) return false;
  if (g->C == nullptr && e != LAFS_EPI_BF16_GELU) return false;
  const long tiles = (long)ceil_div(g->M, 192) * ceil_div(g->N, 192);
  if (mode == 2) return true;                                                      // lab: every shape the kernel can compute
  const long rounds = (tiles + 255) / 256;
  return tiles >= 512 && tiles * 100 >= rounds * 256 * 85;                         // >= 2 rounds, >= 85 % of the CU slots used
}

int lafs_ntw_launch(const lafs_gemm_nt_args* g, hipStream_t stream) {
  switch (g->epilogue) {
    case LAFS_EPI_BF16: return launch<LAFS_EPI_BF16>(g, stream);
    case LAFS_EPI_BF16_GELU: return launch<LAFS_EPI_BF16_GELU>(g, stream);
    case LAFS_EPI_RESID_F32: return launch<LAFS_EPI_RESID_F32>(g, stream);
    default: return launch<LAFS_EPI_DGELU_BF16>(g, stream);
  }
}
