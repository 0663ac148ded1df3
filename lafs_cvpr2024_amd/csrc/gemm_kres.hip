// K-resident streaming GEMM for gfx950:  C[M,N] = epilogue(A[M,384] * W[N,384]^T)
//
// Replaces the cuBLAS GEMMs behind Attention.qkv / Attention.proj / Mlp.fc1 (forward) and the input gradients of proj / fc2
// of the reference's ViT-S trunk (vision_transformer.py:59-65, 75-90): the five GEMM shapes of a block whose reduction axis is
// the embedding width.  lafs_gemm_nt (gemm.hip) dispatches here; everything else stays on the tiled kernel.
//
// Why another kernel (profiles/round2_*): at K = 384 the tiled kernel's 12-step main loop is bound by the
// L2 -> LDS request rate (612 MB of half-line LDS-DMA requests per fc1 GEMM: both operands re-staged per 256x128 tile), its
// epilogue by the HBM write rate, and the two phases add up (fc1: 52 + 58 us alone, 131-138 together) because a tile's stores
// leave as one burst per workgroup.  Here the token operand never touches the LDS and the stores never burst:
//   * a wave keeps its 32 token rows x all 384 k RESIDENT IN REGISTERS (two 16-row MFMA blocks, 96 VGPRs, loaded once per
//     128-row unit) and walks along N; only the weights stream through the LDS -- as whole contiguous rows (768 B: full cache
//     lines), 32 rows = 24 KiB per ring stage, 3 stages;  L2 -> LDS traffic per fc1 GEMM 407 MB, none for A;
//   * v_mfma_f32_16x16x32_bf16 computes C^T blocks (first operand = 16 weight rows, second = 16 tokens): 48 MFMAs per stage and
//     barrier.  A lane (token t, quarter q) ends up with 4 consecutive output columns per MFMA; the weight rows of a stage are
//     interleaved so that two MFMAs give it 8 consecutive bf16 columns: every store instruction writes 16 rows x 64 contiguous
//     bytes (the pattern the HBM write path sustains at full rate; 16-byte pieces scattered over 32 rows -- what a 32x32 MFMA
//     layout produces -- measured 1.8-2.6 TB/s, tools/lab/lab_kres.cpp);
//   * work = (128-row unit, 32-column block) steps in unit-major order, cut into equal contiguous runs, one per workgroup
//     (4 waves, 2 workgroups per CU): no tail round; every stage a wave emits the 2-4 stores of its 32x32 block, so the memory
//     pipeline sees a steady trickle of stores between the LDS-DMA requests instead of per-tile bursts;
//   * the LDS-DMA ring runs across steps with counted s_waitcnt vmcnt (loads and stores retire in issue order on gfx950, the
//     epilogue's memory operations are a compile-time count per step).
#include <stdlib.h>
#include "common.hpp"
#include "gemm_kres.hpp"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

constexpr int KK = 384;                    // reduction length (compile-time: it sizes the register-resident operand)
constexpr int CPR = KK / 8;                // 16-byte chunks per weight row
constexpr int ROWB = KK * 2;               // bytes per weight row
constexpr int GROWS = 32;                  // weight rows per stage = output columns per step
constexpr int STAGE = GROWS * ROWB;        // 24 KiB
constexpr int NSTG = 3;
constexpr int NTH = 256;
constexpr int NDMA = STAGE / 16 / NTH;     // LDS-DMA instructions per thread and stage (6)
constexpr int NKK = KK / 32;               // k steps of a 16x16x32 MFMA (12)
constexpr int MAXN = 1536;                 // bias vector staged in LDS
constexpr int FD_MAX = 8;                  // fragment reads in flight ahead of their MFMAs (6 where the epilogue operands need the registers)
static_assert(CPR % 16 == 0 && STAGE % (16 * NTH) == 0, "stage layout");

struct KArgs {
  const bf16_t* A; const bf16_t* B;
  int M, N, lda, ldb;
  void* C; int ldc; void* C2; int ldc2;
  const float* bias; const float* resid; int ldr;
  const float* seq_scale; const int* row2seq;
  const bf16_t* aux; int ldaux;
  int cbn, steps;                          // 32-column blocks per row unit; row units x column blocks
  unsigned long long* stamps;              // lab (ABL & 32): per workgroup {wait + barrier, issue, MFMA loop, epilogue, whole run, steps} cycles of wave 0
};
// ABL (template argument, 0 in the library; tools/lab/lab_kres.cpp instantiates others): timing ablations
//   1 stores only from lane 0 (dead-code-proof "no stores"), 2 no MFMA, 4 no LDS-DMA after the prologue, 8 no epilogue math,
//   16 no fragment reads, 32 phase time stamps of wave 0 (s_memtime) into KArgs::stamps,
//   64 no workgroup barrier (racy), 128 no store instructions at all (accumulators kept alive by an empty asm)
#define KABL(bit) ((ABL & (bit)) != 0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void st16(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
  const u32x4_t v = {a, b, c, d};
  *reinterpret_cast<u32x4_t*>(p) = v;
}
__device__ __forceinline__ void st16f(void* p, float a, float b, float c, float d) {
  const f32x4v_t v = {a, b, c, d};
  *reinterpret_cast<f32x4v_t*>(p) = v;
}

// memory operations of one step besides its LDS-DMA: S stores (active waves only) + P epilogue-operand loads
template <int EPI, bool HAS_U> struct EpiOps {
  static constexpr bool F32 = (EPI == LAFS_EPI_RESID_F32);
  static constexpr int S = F32 ? 4 : ((EPI == LAFS_EPI_BF16_GELU && HAS_U) ? 4 : 2);
  static constexpr int P = F32 ? 4 : ((EPI == LAFS_EPI_DGELU_BF16) ? 2 : 0);
};

template <int EPI, bool HAS_U, int ABL>
__global__ __launch_bounds__(NTH, 2) void gemm_kres_kernel(KArgs p) {
  constexpr int S = EpiOps<EPI, HAS_U>::S, P = EpiOps<EPI, HAS_U>::P;
  constexpr bool F32 = EpiOps<EPI, HAS_U>::F32;
  // epilogue operand fetched one step ahead (GELU': its load latency no longer sits in front of the epilogue math); the
  // residual epilogue fetches within the step -- a second 16-register buffer does not fit beside its other state
  constexpr bool AHEAD = (EPI == LAFS_EPI_DGELU_BF16);
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTG * STAGE];
  __shared__ __attribute__((aligned(16))) float sbias[MAXN];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15, q = lane >> 4;
  // blocks b, b+8, ... share an XCD: give each XCD a contiguous run of step ranges (the ranges of one row unit read the same A rows)
  const int G = (int)gridDim.x, per = G >> 3;
  const int id = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int sb = (int)((long)p.steps * id / G), se = (int)((long)p.steps * (id + 1) / G);
  if (se <= sb) return;

  for (int i = tid; i < p.N; i += NTH) sbias[i] = p.bias ? p.bias[i] : 0.f;
  __syncthreads();                                    // (also keeps the bias loads out of the counted waits below)

  // LDS image of a stage: row rho (0..31) = 48 chunks; logical chunk c sits at chunk position c ^ (rho & 15) (conflict-free
  // ds_read_b128 of 16 rows x one chunk).  The image is lane-linear for the DMA, so the swizzle goes on the source column.
  // MFMA row s of 16-row group gi lands in lane quarter s >> 2, accumulator register s & 3.  bf16 outputs: the stage's weight
  // rows are interleaved (row = 8 (s >> 2) + 4 gi + (s & 3)) so that the two groups give a lane 8 consecutive columns; fp32
  // outputs: row = 16 gi + s (4 consecutive columns = 16 bytes per MFMA already).
  int doff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int x = i * NTH + tid, rho = x / CPR, cp = x % CPR, c = cp ^ (rho & 15);
    const int s16 = rho & 15, gi = rho >> 4;
    const int rowrel = F32 ? rho : (8 * (s16 >> 2) + 4 * gi + (s16 & 3));
    doff[i] = rowrel * p.ldb + c * 8;
  }
  const int cbn = p.cbn;
  // LDS-DMA from inline asm (common.hpp): with the builtin, hipcc's wait-count pass drains the whole DMA queue (vmcnt(0)) in
  // front of the first fragment read of every stage; the ring is counted by hand instead (wait_vm below)
  const unsigned lds0 = lds_addr_of(smem);
  auto issue = [&](int step) {                        // stage of step `step` (absolute step index)
    const int cb = step % cbn;
    const bf16_t* base = p.B + (size_t)(cb * 32) * p.ldb;
    const unsigned st = lds0 + ((step - sb) % NSTG) * STAGE + wave * 1024;
    fence();
    if (!(KABL(4) && step >= sb + 2)) {
#pragma unroll
      for (int i = 0; i < NDMA; ++i) lds_dma16_m0(base + doff[i], st + i * (NTH * 16));
    }
    fence();
  };

  // fragment read offsets inside a stage: row 16 gi + t, chunk (4 kk + q) ^ t: four register offsets (kk & 3) + immediates
  int foff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    foff[i] = t * ROWB + (((4 * i + q) ^ t) << 4);
    asm volatile("" : "+v"(foff[i]));
  }
  bf16x8_t areg[2][NKK];                              // two 16-token blocks x 12 k steps: lane (t, q) holds k = 32 kk + 8 q .. + 7
  f32x4_t acc[2][2];                                  // [weight row group][token block]
  uint4 pre[P > 0 ? P : 1], nxt[P > 0 ? P : 1];       // epilogue operand of this step / of the next one (fetched a step ahead)
  float sc[2] = {1.0f, 1.0f};
  int cur_mu = -1;
  bool active = false;
  int m0 = 0;

  unsigned long long tacc[4] = {0, 0, 0, 0}, t_begin = 0, t_last = 0;
  auto lap = [&](int slot) {
    if constexpr (KABL(32)) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tacc[slot] += now - t_last;
      t_last = now;
    }
  };
  if constexpr (KABL(32)) t_begin = t_last = __builtin_amdgcn_s_memtime();
  auto fetch = [&](int step, uint4 (&dst)[P > 0 ? P : 1]) {        // exactly P loads: the epilogue operand of `step`
    if (P == 0) return;
    const int st = min(step, p.steps - 1);
    const int mu = st / cbn, n0 = (st - mu * cbn) * 32;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int b = i & 1, gi = i >> 1;
      const int mr = min(mu * 128 + wave * 32 + t + 16 * b, p.M - 1);
      if (F32) dst[i] = *reinterpret_cast<const uint4*>(p.resid + (size_t)mr * p.ldr + n0 + 16 * gi + 4 * q);
      else dst[i] = *reinterpret_cast<const uint4*>(p.aux + (size_t)mr * p.ldaux + n0 + 8 * q);
    }
    fence();
  };
  issue(sb);
  if (sb + 1 < se) issue(sb + 1);
  int stage = 0;
  for (int s = sb; s < se; ++s) {
    const int mu = s / cbn, cb = s - mu * cbn;
    if (mu != cur_mu) {                               // new row unit: (re)load the resident operand, then drain everything
      cur_mu = mu;
      m0 = mu * 128 + wave * 32 + t;
      active = (mu * 128 + wave * 32) < p.M;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int mr = min(m0 + 16 * b, p.M - 1);
        const bf16_t* arow = p.A + (size_t)mr * p.lda;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) areg[b][kk] = *reinterpret_cast<const bf16x8_t*>(arow + (4 * kk + q) * 8);
        if (EPI == LAFS_EPI_RESID_F32 && p.seq_scale != nullptr) sc[b] = p.seq_scale[p.row2seq[mr]];
      }
      if (AHEAD && s == sb) fetch(s, pre);            // first step of the run: nobody fetched its epilogue operand ahead
      wait_vm<0>();
      __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0) again, in a form hipcc's wait-count pass sees: no waits on areg inside the MFMA loop
    } else if (s + 1 < se) {                          // younger than this stage's DMA: two steps of epilogue operations + one stage
      if (active) wait_vm<NDMA + 2 * (S + P)>(); else wait_vm<NDMA + 2 * P>();
    } else {                                          // last step of the run: no younger stage
      if (active) wait_vm<2 * (S + P)>(); else wait_vm<2 * P>();
    }
    if constexpr (!KABL(64)) __builtin_amdgcn_s_barrier();
    lap(0);
    if (s + 2 < se) issue(s + 2);
    const int n0 = cb * 32;
    if (AHEAD) fetch(s + 1, nxt);                     // P loads, consumed by the NEXT step's epilogue
    else fetch(s, pre);
    lap(1);
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[gi][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* st = smem + stage * STAGE;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          bf16x8_t w = areg[gi][(kk + 1) % NKK];
          if constexpr (!KABL(16)) w = *reinterpret_cast<const bf16x8_t*>(st + foff[kk & 3] + (kk >> 2) * 256 + gi * (16 * ROWB));
          if constexpr (!KABL(2)) {
            acc[gi][0] = mfma16(w, areg[0][kk], acc[gi][0]);
            acc[gi][1] = mfma16(w, areg[1][kk], acc[gi][1]);
          } else {
            asm volatile("" :: "v"(w));
          }
        }
      // fragment reads run FD ahead of the MFMA pairs that consume them (hipcc on its own keeps one read in flight and exposes
      // the LDS latency 24 times per stage: 1440 instead of ~800 cycles, tools/lab/lab_kres.cpp)
      constexpr int FD = FD_MAX;
      if constexpr (!KABL(16) && !KABL(2)) {
        __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);
#pragma unroll
        for (int i = 0; i < 2 * NKK - FD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * FD, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stage = (stage + 1 == NSTG) ? 0 : stage + 1;
    lap(2);

    // ---------------- epilogue: lane (t, q) owns rows m0 and m0 + 16 and, per row, 8 consecutive columns (bf16 outputs: both
    // row groups) or 2 x 4 consecutive columns (fp32 outputs: one piece per row group)
    fence();
    if (active) {
      const bool lab_lane0 = !KABL(1) || lane == 0;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int m = m0 + 16 * b;
        const bool rowok = (m < p.M) && lab_lane0 && !KABL(128);
        if constexpr (KABL(128)) asm volatile("" :: "v"(acc[0][b]), "v"(acc[1][b]));
        if (F32) {
#pragma unroll
          for (int gi = 0; gi < 2; ++gi) {
            const int n = n0 + 16 * gi + 4 * q;
            const float4 b4 = *reinterpret_cast<const float4*>(sbias + n);
            const uint4 r4 = pre[gi * 2 + b];
            float v0 = acc[gi][b][0] + b4.x, v1 = acc[gi][b][1] + b4.y, v2 = acc[gi][b][2] + b4.z, v3 = acc[gi][b][3] + b4.w;
            if (!KABL(8)) {
              v0 = __uint_as_float(r4.x) + sc[b] * v0; v1 = __uint_as_float(r4.y) + sc[b] * v1;
              v2 = __uint_as_float(r4.z) + sc[b] * v2; v3 = __uint_as_float(r4.w) + sc[b] * v3;
            }
            if (rowok) st16f(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, v0, v1, v2, v3);
          }
        } else {
          const int n = n0 + 8 * q;
          const float4 b0 = *reinterpret_cast<const float4*>(sbias + n), b1 = *reinterpret_cast<const float4*>(sbias + n + 4);
          float v[8] = {acc[0][b][0] + b0.x, acc[0][b][1] + b0.y, acc[0][b][2] + b0.z, acc[0][b][3] + b0.w,
                        acc[1][b][0] + b1.x, acc[1][b][1] + b1.y, acc[1][b][2] + b1.z, acc[1][b][3] + b1.w};
          if (EPI == LAFS_EPI_DGELU_BF16 && !KABL(8)) {
            const uint4 a4 = pre[b];
            v[0] *= gelu_grad_f(bf_lo(a4.x)); v[1] *= gelu_grad_f(bf_hi(a4.x)); v[2] *= gelu_grad_f(bf_lo(a4.y)); v[3] *= gelu_grad_f(bf_hi(a4.y));
            v[4] *= gelu_grad_f(bf_lo(a4.z)); v[5] *= gelu_grad_f(bf_hi(a4.z)); v[6] *= gelu_grad_f(bf_lo(a4.w)); v[7] *= gelu_grad_f(bf_hi(a4.w));
          }
          if (EPI != LAFS_EPI_BF16_GELU || HAS_U) {
            if (rowok) st16(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n, pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]),
                            pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
          }
          if (EPI == LAFS_EPI_BF16_GELU) {
            if (!KABL(8)) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
            }
            if (rowok) st16(reinterpret_cast<bf16_t*>(p.C2) + (size_t)m * p.ldc2 + n, pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]),
                            pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
          }
        }
      }
    }
    fence();
    if (AHEAD) {
#pragma unroll
      for (int i = 0; i < P; ++i) pre[i] = nxt[i];
    }
    lap(3);
  }
  if constexpr (KABL(32)) {
    if (tid == 0 && p.stamps != nullptr) {
      unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
      o[0] = tacc[0]; o[1] = tacc[1]; o[2] = tacc[2]; o[3] = tacc[3];
      o[4] = __builtin_amdgcn_s_memtime() - t_begin; o[5] = (unsigned long long)(se - sb);
    }
  }
}

template <int EPI, bool HAS_U, int ABL>
int launch(const KArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((gemm_kres_kernel<EPI, HAS_U, ABL>), dim3(grid), dim3(NTH), 0, s, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

}  // namespace

bool lafs_kres_eligible(const lafs_gemm_nt_args* g) {
  // LAFS_KRES = bit mask of the epilogues routed here (1 plain, 2 GELU, 4 residual, 8 GELU'); 0 = tiled kernel everywhere.
  // Default 7: the GELU' input gradient stays on the tiled kernel (134 vs 120 us in the step: its epilogue is VALU-bound)
  static const int mask = [] { const char* v = getenv("LAFS_KRES"); return v != nullptr ? atoi(v) : 7; }();
  const int e = g->epilogue;
  const int bit = e == LAFS_EPI_BF16 ? 1 : (e == LAFS_EPI_BF16_GELU ? 2 : (e == LAFS_EPI_RESID_F32 ? 4 : (e == LAFS_EPI_DGELU_BF16 ? 8 : 0)));
  if (!(mask & bit)) return false;
  if (g->splits > 1) return false;
  if (g->K != KK || g->N % 32 != 0 || g->N > MAXN || g->N < 32 || g->M < 2048) return false;
  if (!(e == LAFS_EPI_BF16 || e == LAFS_EPI_BF16_GELU || e == LAFS_EPI_RESID_F32 || e == LAFS_EPI_DGELU_BF16)) return false;
  if (g->drop_p > 0.f) return false;
  if (g->lda % 8 != 0 || g->ldb % 8 != 0 || g->ldc % 8 != 0) return false;
  if (e == LAFS_EPI_BF16_GELU && (g->C2 == nullptr || g->ldc2 % 8 != 0)) return false;
  if (e == LAFS_EPI_RESID_F32 && (g->resid == nullptr || g->ldr % 4 != 0)) return false;
  if (e == LAFS_EPI_DGELU_BF16 && (g->aux == nullptr || g->ldaux % 8 != 0)) return false;
  if (e != LAFS_EPI_BF16_GELU && g->C == nullptr) return false;
  return true;
}

namespace {
template <int ABL>
int kres_launch(const lafs_gemm_nt_args* g, hipStream_t stream, int grid_override = 0, unsigned long long* stamps = nullptr) {
  const int e = g->epilogue;
  KArgs a;
  a.A = (const bf16_t*)g->A; a.B = (const bf16_t*)g->B; a.M = g->M; a.N = g->N; a.lda = g->lda; a.ldb = g->ldb;
  a.C = g->C; a.ldc = g->ldc; a.C2 = g->C2; a.ldc2 = g->ldc2;
  a.bias = (e == LAFS_EPI_DGELU_BF16) ? nullptr : g->bias;
  a.resid = g->resid; a.ldr = g->ldr; a.seq_scale = g->seq_scale; a.row2seq = g->row2seq;
  a.aux = (const bf16_t*)g->aux; a.ldaux = g->ldaux;
  a.cbn = g->N / 32;
  a.stamps = stamps;

  const int mus = (g->M + 127) / 128;
  a.steps = mus * a.cbn;
  // two 4-wave workgroups per CU: one residency wave of equal step runs (at least ~8 steps each, or the reload of the
  // resident operand per run stops being amortised)
  int grid = 512;
  while (grid > 8 && a.steps / grid < 8) grid >>= 1;
  if (grid_override > 0) grid = grid_override;
  switch (e) {
    case LAFS_EPI_BF16: return launch<LAFS_EPI_BF16, true, ABL>(a, grid, stream);
    case LAFS_EPI_BF16_GELU:
      return g->C != nullptr ? launch<LAFS_EPI_BF16_GELU, true, ABL>(a, grid, stream) : launch<LAFS_EPI_BF16_GELU, false, ABL>(a, grid, stream);
    case LAFS_EPI_RESID_F32: return launch<LAFS_EPI_RESID_F32, true, ABL>(a, grid, stream);
    default: return launch<LAFS_EPI_DGELU_BF16, true, ABL>(a, grid, stream);
  }
}
}  // namespace

#ifndef LAFS_KRES_LAB
int lafs_kres_launch(const lafs_gemm_nt_args* g, hipStream_t stream) { return kres_launch<0>(g, stream); }
#endif
