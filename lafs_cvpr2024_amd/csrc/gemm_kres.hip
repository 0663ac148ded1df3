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
//     128-row unit -- through the ring buffers, as four 32-row stages, while no weight stage is in flight) and walks along N;
//     only the weights stream through the LDS -- as whole contiguous rows (768 B: full cache lines), 32 rows = 24 KiB per ring
//     stage, 3 stages;  L2 -> LDS traffic per fc1 GEMM 407 MB for the weights, 34-55 MB for the rows;
//   * v_mfma_f32_16x16x32_bf16 computes C^T blocks (first operand = 16 weight rows, second = 16 tokens): 48 MFMAs per stage and
//     barrier.  A lane (token t, quarter q) ends up with 4 consecutive output columns per MFMA; the weight rows of a stage are
//     interleaved so that two MFMAs give it 8 consecutive bf16 columns: every store instruction writes 16 rows x 64 contiguous
//     bytes (the pattern the HBM write path sustains at full rate; 16-byte pieces scattered over 32 rows -- what a 32x32 MFMA
//     layout produces -- measured 1.8-2.6 TB/s, tools/lab/lab_kres.cpp);
//   * work = (128-row unit, 64-column block) items in unit-major order, cut into equal contiguous runs, one per workgroup
//     (4 waves, 2 workgroups per CU, all resident at once).  An item is two ring stages (32 columns each) and one epilogue: the two
//     stages' 64-byte pieces of a row are stored back to back and complete 128-byte lines; a wave emits 4-8 stores per item, so
//     the memory pipeline sees a steady trickle of stores between the LDS-DMA requests instead of per-tile bursts;
//   * the accumulators of a stage start from the bias of their columns (LDS reads issued a phase ahead, straight into the
//     accumulator registers): the epilogue neither zero-fills nor adds a bias;
//   * the LDS-DMA ring runs across items with counted s_waitcnt vmcnt (loads and stores retire in issue order on gfx950, the
//     epilogue's memory operations are a compile-time count per item).
#include <stdlib.h>
#include "common.hpp"
#include "gemm_kres.hpp"
#include "ctx.hpp"

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
constexpr int NDMA4 = STAGE / 16 / NTH;    // LDS-DMA instructions per thread and stage (6)
constexpr int NKK = KK / 32;               // k steps of a 16x16x32 MFMA (12)
constexpr int MAXN = 1536;                 // bias vector staged in LDS
constexpr int FD = 8;                      // fragment reads in flight ahead of their MFMAs
static_assert(CPR % 16 == 0 && STAGE % (16 * NTH) == 0, "stage layout");

struct KArgs {
  const bf16_t* A; const bf16_t* B;
  int M, N, lda, ldb;
  void* C; int ldc; void* C2; int ldc2;
  const float* bias; const float* resid; int ldr;
  const float* seq_scale; const int* row2seq;
  const bf16_t* aux; int ldaux;
  int cbn, items;                          // 64-column blocks per row unit; row units x column blocks
  int save_grad;                           // LAFS_GELU_SAVE_GRAD: GELU epilogue stores gelu'(u) for u / GELU' epilogue multiplies aux in as it is
  unsigned long long* stamps;              // lab (ABL & 32): per workgroup {wait + barrier, issue, MFMA loop, epilogue, whole run, stages, reload wait, reloads} cycles of wave 0
};
// ABL (template argument, 0 in the library; tools/lab/lab_kres.cpp instantiates others): timing ablations and variants
//   1 stores only from lane 0 (dead-code-proof "no stores"), 2 no MFMA, 4 no weight stages after the first two, 8 no epilogue math,
//   16 no fragment reads, 32 phase time stamps of wave 0 (s_memtime) into KArgs::stamps, 64 no workgroup barrier (racy),
//   128 no store instructions at all (accumulators kept alive by an empty asm), 256 every store writes one contiguous KiB,
//   512 resident rows (re)loaded by per-lane global loads instead of through the ring buffers, 2048 GELU' operand fetched an item
//   ahead (needs 512: both together spill)
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

// memory operations of one item besides its LDS-DMA: S stores (active waves only) + P epilogue-operand loads
template <int EPI, bool HAS_U> struct EpiOps {
  static constexpr bool F32 = (EPI == LAFS_EPI_RESID_F32);
  static constexpr int S = F32 ? 8 : ((EPI == LAFS_EPI_BF16_GELU && HAS_U) ? 8 : 4);      // (HAS_U of the GELU' variants: see AUX_IS_GRAD)
  static constexpr int P = F32 ? 8 : ((EPI == LAFS_EPI_DGELU_BF16) ? 4 : 0);
};

template <int EPI, bool HAS_U, int ABL>
__global__ __launch_bounds__(NTH, 2) void gemm_kres_kernel(KArgs p) {
  constexpr int UROWS = 128, NDMA = NDMA4;
  constexpr int S = EpiOps<EPI, HAS_U>::S, P = EpiOps<EPI, HAS_U>::P;
  constexpr bool F32 = EpiOps<EPI, HAS_U>::F32;
  // epilogue operand fetched one item ahead (GELU': its load latency no longer sits in front of the epilogue math); the
  // residual epilogue fetches within the item -- a second 32-register buffer does not fit beside its other state
  // GELU' variants: HAS_U = false means that aux already holds gelu'(u) (LAFS_GELU_SAVE_GRAD): no derivative math in the epilogue
  constexpr bool AUX_IS_GRAD = (EPI == LAFS_EPI_DGELU_BF16) && !HAS_U;
  constexpr bool AHEAD = (EPI == LAFS_EPI_DGELU_BF16) && KABL(2048);   // (lab only: beside the staged reload the second operand buffer spills)
  // resident rows (re)loaded through the ring buffers (the GELU' variant has no registers to spare for that code path: it keeps
  // the per-lane loads)
  constexpr bool STAGED = !KABL(512) && !AHEAD;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTG * STAGE];
  __shared__ __attribute__((aligned(16))) float sbias[MAXN];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15, q = lane >> 4;
  // blocks b, b+8, ... share an XCD: give each XCD a contiguous run of item ranges (the ranges of one row unit read the same A rows)
  const int G = (int)gridDim.x, per = G >> 3;
  const int id = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int ib = (int)((long)p.items * id / G), ie = (int)((long)p.items * (id + 1) / G);
  if (ie <= ib) return;
  const int kb = 2 * ib, ke = 2 * ie;                 // stages of this run (two per item)

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
    doff[i] = (rowrel * p.ldb + c * 8) * 2;           // bytes
  }
  const int cbn = p.cbn;
  // LDS-DMA from inline asm (common.hpp): with the builtin, hipcc's wait-count pass drains the whole DMA queue (vmcnt(0)) in
  // front of the first fragment read of every stage; the ring is counted by hand instead (wait_vm below)
  const unsigned lds0 = lds_addr_of(smem);
  int pstage = 0;                                     // ring buffer the next issued stage goes to
  auto issue = [&](int k) {                           // stage k (absolute: item k >> 1, half k & 1)
    const int cb = (k >> 1) % cbn;
    const bf16_t* base = p.B + (size_t)(cb * 64 + 32 * (k & 1)) * p.ldb;
    const unsigned st = lds0 + pstage * STAGE + wave * 1024;
    pstage = (pstage + 1 == NSTG) ? 0 : pstage + 1;
    fence();
    if (!(KABL(4) && k >= kb + 2)) {
#pragma unroll
      for (int i = 0; i < NDMA; ++i) lds_dma16_m0_s(base, (unsigned)doff[i], st + i * (NTH * 16));
    }
    fence();
  };
  // The 32 token rows of wave `w` of unit `mu` as ONE ring stage (same 768-byte rows, same chunk swizzle as a weight stage):
  // LDS-DMA fetches whole rows (the per-lane gather of 16 rows x 64 bytes per instruction it replaces took ~7 us per reload,
  // bound by the CU's outstanding misses); the owning wave then reads its fragments with the weight-fragment addressing.
  auto issue_rows = [&](int mu, int w, int buf) {
    const unsigned st = lds0 + buf * STAGE + wave * 1024;
    fence();
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int x = i * NTH + tid, rho = x / CPR, cp = x % CPR, c = cp ^ (rho & 15);
      const int row = min(mu * UROWS + w * 32 + rho, p.M - 1);
      lds_dma16_m0(p.A + (size_t)row * p.lda + c * 8, st + i * (NTH * 16));
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
  f32x4_t acc[2][2][2];                               // [stage of the item][weight row group][token block]
  uint4 pre[P > 0 ? P : 1], nxt[P > 0 ? P : 1];       // epilogue operand of this item / of the next one (fetched an item ahead)
  float sc[2] = {1.0f, 1.0f};
  int cur_mu = -1;
  bool active = false;
  int m0 = 0;

  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, t_begin = 0, t_last = 0;   // [4]: wait + barrier of the reload stages, [5]: how many
  auto lap = [&](int slot) {
    if constexpr (KABL(32)) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tacc[slot] += now - t_last;
      t_last = now;
    }
  };
  if constexpr (KABL(32)) t_begin = t_last = __builtin_amdgcn_s_memtime();
  auto fetch = [&](int item, uint4 (&dst)[P > 0 ? P : 1]) {        // exactly P loads: the epilogue operand of `item`
    if (P == 0) return;
    const int it = min(item, p.items - 1);
    const int mu = it / cbn, n0 = (it - mu * cbn) * 64;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int b = i & 1, x = i >> 1;                // bf16: x = stage; fp32: x = stage * 2 + row group
      const int mr = min(mu * UROWS + wave * 32 + t + 16 * b, p.M - 1);
      if (F32) dst[i] = *reinterpret_cast<const uint4*>(p.resid + (size_t)mr * p.ldr + n0 + 16 * x + 4 * q);
      else dst[i] = *reinterpret_cast<const uint4*>(p.aux + (size_t)mr * p.ldaux + n0 + 32 * x + 8 * q);
    }
    fence();
  };
  // The accumulators of a stage start from the bias of their four columns (LDS reads issued a phase ahead -- before the wait and
  // barrier in front of the item's first stage, before the first stage's MFMAs for the second -- straight into the registers the
  // MFMAs accumulate in): no zero fill, no bias add, and no LDS latency exposed at the head of the epilogue.
  auto bias_init = [&](f32x4_t (&a)[2][2], int n0, int g) {
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int n = F32 ? n0 + 16 * (2 * g + gi) + 4 * q : n0 + 32 * g + 8 * q + 4 * gi;
#pragma unroll
      for (int b = 0; b < 2; ++b) a[gi][b] = *reinterpret_cast<const f32x4_t*>(sbias + n);
    }
  };
  auto mfma_stage = [&](int stage, f32x4_t (&a)[2][2]) {            // 48 MFMAs on one ring stage
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* st = smem + stage * STAGE;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        bf16x8_t w = areg[gi][(kk + 1) % NKK];
        if constexpr (!KABL(16)) w = *reinterpret_cast<const bf16x8_t*>(st + foff[kk & 3] + (kk >> 2) * 256 + gi * (16 * ROWB));
        if constexpr (!KABL(2)) {
          a[gi][0] = mfma16(w, areg[0][kk], a[gi][0]);
          a[gi][1] = mfma16(w, areg[1][kk], a[gi][1]);
        } else {
          asm volatile("" :: "v"(w));
        }
      }
    // fragment reads run FD ahead of the MFMA pairs that consume them (hipcc on its own keeps one read in flight and exposes
    // the LDS latency 24 times per stage: 1440 instead of ~1000 cycles, tools/lab/lab_kres.cpp)
    if constexpr (!KABL(16) && !KABL(2)) {
      constexpr int FDV = FD;                            // (12 in flight measured no better: the phase is not latency-bound any more)
      __builtin_amdgcn_sched_group_barrier(0x100, FDV, 0);
#pragma unroll
      for (int i = 0; i < 2 * NKK - FDV; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * FDV, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  auto read_rows = [&](int buf) {                     // this wave's 32 rows out of ring buffer `buf` into areg
    const unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk)
        areg[b][kk] = *reinterpret_cast<const bf16x8_t*>(st + foff[kk & 3] + (kk >> 2) * 256 + b * (16 * ROWB));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads have returned before the buffer is handed back at the barrier
  };
  int stage = 0;                                      // ring buffer the next consumed stage sits in
  bool fresh = false;                                 // first item after a (re)load: no epilogue operations in flight yet
  for (int it = ib; it < ie; ++it) {
    const int mu = it / cbn, cb = it - mu * cbn;
    const int k0 = 2 * it;
    const bool reloaded_lab = (mu != cur_mu);
    // the stages of the NEXT item are issued from this one unless it starts a new row unit (its rows go through the ring first)
    const bool feed_next = (it + 1 < ie) && ((it + 1) / cbn == mu);
    bias_init(acc[0], cb * 64, 0);
    // ---------------- first stage of the item
    if (mu != cur_mu) {                               // new row unit: (re)load the resident operand
      cur_mu = mu;
      m0 = mu * UROWS + wave * 32 + t;
      active = (mu * UROWS + wave * 32) < p.M;
      if (EPI == LAFS_EPI_RESID_F32 && p.seq_scale != nullptr) {
#pragma unroll
        for (int b = 0; b < 2; ++b) sc[b] = p.seq_scale[p.row2seq[min(m0 + 16 * b, p.M - 1)]];
      }
      if (AHEAD && it == ib) fetch(it, pre);          // first item of the run: nobody fetched its epilogue operand ahead
      if constexpr (STAGED) {
        // No stage of this item is in flight (feed_next above), every ring buffer is free once all waves are here.
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        issue_rows(mu, 0, 0); issue_rows(mu, 1, 1); issue_rows(mu, 2, 2);
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (wave < 3) read_rows(wave);
        __builtin_amdgcn_s_barrier();                 // buffers free again
        issue_rows(mu, 3, 0);
        pstage = 1;
        issue(k0);                                    // -> buffer 1
        issue(k0 + 1);                                // -> buffer 2; the stage after them goes to buffer 0
        wait_vm<2 * NDMA>();                          // wave 3's rows have landed (the two weight stages are younger)
        __builtin_amdgcn_s_barrier();
        if (wave == 3) read_rows(0);
        stage = 1;
        wait_vm<NDMA>();                              // stage k0 has landed; the barrier below also hands buffer 0 back
      } else {                                        // lab: per-lane global loads, then drain
        if (it == ib) { pstage = 0; issue(k0); issue(k0 + 1); stage = 0; }
        else { pstage = stage; issue(k0); issue(k0 + 1); }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const bf16_t* arow = p.A + (size_t)min(m0 + 16 * b, p.M - 1) * p.lda;
#pragma unroll
          for (int kk = 0; kk < NKK; ++kk) areg[b][kk] = *reinterpret_cast<const bf16x8_t*>(arow + (4 * kk + q) * 8);
        }
        wait_vm<0>();
        __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0) again, in a form hipcc's wait-count pass sees
      }
      fresh = true;
    } else {                                          // younger than this stage's DMA: the other stage of the previous item + its epilogue operations
      if (active) wait_vm<NDMA + S + P>(); else wait_vm<NDMA + P>();
    }
    if constexpr (!KABL(64)) __builtin_amdgcn_s_barrier();
    if constexpr (KABL(32)) {
      if (reloaded_lab) { lap(4); tacc[5] += 1; } else lap(0);
    }
    if (feed_next) issue(k0 + 2);
    if (AHEAD) fetch(it + 1, nxt);                    // P loads, consumed by the NEXT item's epilogue
    bias_init(acc[1], cb * 64, 1);
    lap(1);
    mfma_stage(stage, acc[0]);
    stage = (stage + 1 == NSTG) ? 0 : stage + 1;
    lap(2);
    // ---------------- second stage: younger than its DMA are the previous item's epilogue operations (none right after a
    // reload), the stage issued above (if any) and, with AHEAD, the fetch above
    if (fresh) {
      if (feed_next) wait_vm<NDMA + (AHEAD ? P : 0)>(); else wait_vm<(AHEAD ? P : 0)>();
    } else if (feed_next) {
      if (active) wait_vm<NDMA + S + P>(); else wait_vm<NDMA + P>();
    } else {
      if (active) wait_vm<S + P>(); else wait_vm<P>();
    }
    fresh = false;
    if constexpr (!KABL(64)) __builtin_amdgcn_s_barrier();
    lap(0);
    if (feed_next) issue(k0 + 3);
    if (!AHEAD) fetch(it, pre);                       // P loads, consumed a stage of MFMAs later
    lap(1);
    mfma_stage(stage, acc[1]);
    stage = (stage + 1 == NSTG) ? 0 : stage + 1;
    lap(2);

    // ---------------- epilogue: lane (t, q) owns rows m0 and m0 + 16 and, per row and stage, 8 consecutive columns (bf16 outputs:
    // both row groups) or 2 x 4 consecutive columns (fp32 outputs: one piece per row group).  The two stages' pieces of a row
    // are stored back to back: each pair completes 128-byte lines (one step apart they reached the HBM as separate half-line
    // writes: 1.28x write traffic, profiles/round2_kernel_pmc.json history).
    fence();
    const int n0 = cb * 64;
    // lab (ABL & 256): every store instruction writes one contiguous KiB (values land in the wrong places)
    auto lin = [&](int k) { return (size_t)((((it * 4 + wave) * 8 + k) & 32767) * 1024 + lane * 16); };
    if (active) {
      const bool lab_lane0 = !KABL(1) || lane == 0;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int m = m0 + 16 * b;
        const bool rowok = (m < p.M) && lab_lane0 && !KABL(128);
        if constexpr (KABL(128)) asm volatile("" :: "v"(acc[0][0][b]), "v"(acc[0][1][b]), "v"(acc[1][0][b]), "v"(acc[1][1][b]));
        if (F32) {
#pragma unroll
          for (int x = 0; x < 4; ++x) {                // x = stage * 2 + row group: 16 columns each
            const int g = x >> 1, gi = x & 1;
            const int n = n0 + 16 * x + 4 * q;
            const uint4 r4 = pre[x * 2 + b];
            float v0 = acc[g][gi][b][0], v1 = acc[g][gi][b][1], v2 = acc[g][gi][b][2], v3 = acc[g][gi][b][3];
            if (!KABL(8)) {
              v0 = __uint_as_float(r4.x) + sc[b] * v0; v1 = __uint_as_float(r4.y) + sc[b] * v1;
              v2 = __uint_as_float(r4.z) + sc[b] * v2; v3 = __uint_as_float(r4.w) + sc[b] * v3;
            }
            if (KABL(256)) st16f(reinterpret_cast<unsigned char*>(p.C) + lin(b * 4 + x), v0, v1, v2, v3);
            else if (rowok) st16f(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, v0, v1, v2, v3);
          }
        } else {
          float v[2][8];
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            v[g][0] = acc[g][0][b][0]; v[g][1] = acc[g][0][b][1]; v[g][2] = acc[g][0][b][2]; v[g][3] = acc[g][0][b][3];
            v[g][4] = acc[g][1][b][0]; v[g][5] = acc[g][1][b][1]; v[g][6] = acc[g][1][b][2]; v[g][7] = acc[g][1][b][3];
            if (EPI == LAFS_EPI_DGELU_BF16 && !KABL(8)) {
              const uint4 a4 = pre[g * 2 + b];
              if constexpr (AUX_IS_GRAD) {               // aux already holds gelu'(u)
                v[g][0] *= bf_lo(a4.x); v[g][1] *= bf_hi(a4.x); v[g][2] *= bf_lo(a4.y); v[g][3] *= bf_hi(a4.y);
                v[g][4] *= bf_lo(a4.z); v[g][5] *= bf_hi(a4.z); v[g][6] *= bf_lo(a4.w); v[g][7] *= bf_hi(a4.w);
              } else {
                v[g][0] *= gelu_grad_f(bf_lo(a4.x)); v[g][1] *= gelu_grad_f(bf_hi(a4.x)); v[g][2] *= gelu_grad_f(bf_lo(a4.y)); v[g][3] *= gelu_grad_f(bf_hi(a4.y));
                v[g][4] *= gelu_grad_f(bf_lo(a4.z)); v[g][5] *= gelu_grad_f(bf_hi(a4.z)); v[g][6] *= gelu_grad_f(bf_lo(a4.w)); v[g][7] *= gelu_grad_f(bf_hi(a4.w));
              }
            }
          }
          float dv[2][8];                                // GELU epilogue saving gelu'(u): value and derivative from one exp / rcp
          const bool both = (EPI == LAFS_EPI_BF16_GELU) && HAS_U && p.save_grad && !KABL(8);
          if (both) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
              for (int e = 0; e < 8; ++e) { float gv; gelu_both_f(v[g][e], gv, dv[g][e]); v[g][e] = gv; }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int n = n0 + 32 * g + 8 * q;
              if (rowok) st16(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n, pack_bf2(dv[g][0], dv[g][1]), pack_bf2(dv[g][2], dv[g][3]),
                              pack_bf2(dv[g][4], dv[g][5]), pack_bf2(dv[g][6], dv[g][7]));
            }
          }
          if ((EPI != LAFS_EPI_BF16_GELU || HAS_U) && !both) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int n = n0 + 32 * g + 8 * q;
              if (KABL(256)) st16(reinterpret_cast<unsigned char*>(p.C) + lin(b * 2 + g), pack_bf2(v[g][0], v[g][1]), pack_bf2(v[g][2], v[g][3]),
                                  pack_bf2(v[g][4], v[g][5]), pack_bf2(v[g][6], v[g][7]));
              else if (rowok) st16(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n, pack_bf2(v[g][0], v[g][1]), pack_bf2(v[g][2], v[g][3]),
                                   pack_bf2(v[g][4], v[g][5]), pack_bf2(v[g][6], v[g][7]));
            }
          }
          if (EPI == LAFS_EPI_BF16_GELU) {
            if (!KABL(8) && !both) {
#pragma unroll
              for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[g][e] = gelu_f(v[g][e]);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int n = n0 + 32 * g + 8 * q;
              if (KABL(256)) st16(reinterpret_cast<unsigned char*>(p.C2) + lin(b * 2 + g), pack_bf2(v[g][0], v[g][1]), pack_bf2(v[g][2], v[g][3]),
                                  pack_bf2(v[g][4], v[g][5]), pack_bf2(v[g][6], v[g][7]));
              else if (rowok) st16(reinterpret_cast<bf16_t*>(p.C2) + (size_t)m * p.ldc2 + n, pack_bf2(v[g][0], v[g][1]), pack_bf2(v[g][2], v[g][3]),
                                   pack_bf2(v[g][4], v[g][5]), pack_bf2(v[g][6], v[g][7]));
            }
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
      o[4] = __builtin_amdgcn_s_memtime() - t_begin; o[5] = (unsigned long long)(ke - kb); o[6] = tacc[4]; o[7] = tacc[5];
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
  // LAFS_OPT_KRES_MASK = bit mask of the epilogues routed here (1 plain, 2 GELU, 4 residual, 8 GELU'); 0 = tiled kernel everywhere.
  // Default 15 (whole step, one box, tools/lab/ab_env.sh: 16.95 ms against 17.22 with mask 7 and 17.57 with 0).
  const int mask = lafs_ctx_opt(g->ctx, LAFS_OPT_KRES_MASK);
  const int e = g->epilogue;
  const int bit = e == LAFS_EPI_BF16 ? 1 : (e == LAFS_EPI_BF16_GELU ? 2 : (e == LAFS_EPI_RESID_F32 ? 4 : (e == LAFS_EPI_DGELU_BF16 ? 8 : 0)));
  if (!(mask & bit)) return false;
  if (g->splits > 1 || g->operand_f16) return false;
  if (g->K != KK || g->N % 64 != 0 || g->N > MAXN || g->N < 64 || g->M < 2048) return false;
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
  a.cbn = g->N / 64;
  a.save_grad = (g->act == LAFS_GELU_SAVE_GRAD && (e == LAFS_EPI_BF16_GELU || e == LAFS_EPI_DGELU_BF16)) ? 1 : 0;
  a.stamps = stamps;

  const int mus = (g->M + 127) / 128;
  a.items = mus * a.cbn;
  // two 4-wave workgroups per CU: one residency wave of equal item runs (at least ~4 items each, or the reload of the
  // resident operand per run stops being amortised)
  const int min_items = lafs_ctx_opt(g->ctx, LAFS_OPT_KRES_MIN_ITEMS);      // lab knob
  const int g_comm_cus = lafs_ctx_opt(g->ctx, LAFS_OPT_COMM_CUS);
  int grid = 512;
  while (grid > 8 && a.items / grid < min_items) grid >>= 1;
  if (g_comm_cus > 0 && grid > 2 * (256 - g_comm_cus)) grid = (2 * (256 - g_comm_cus)) & ~7;     // CUs left to RCCL (LAFS_OPT_COMM_CUS)
  if (grid_override > 0) grid = grid_override;
  switch (e) {
    case LAFS_EPI_BF16: return launch<LAFS_EPI_BF16, true, ABL>(a, grid, stream);
    case LAFS_EPI_BF16_GELU:
      return g->C != nullptr ? launch<LAFS_EPI_BF16_GELU, true, ABL>(a, grid, stream) : launch<LAFS_EPI_BF16_GELU, false, ABL>(a, grid, stream);
    case LAFS_EPI_RESID_F32: return launch<LAFS_EPI_RESID_F32, true, ABL>(a, grid, stream);
    default:
      return a.save_grad ? launch<LAFS_EPI_DGELU_BF16, false, ABL>(a, grid, stream) : launch<LAFS_EPI_DGELU_BF16, true, ABL>(a, grid, stream);
  }
}
}  // namespace

#ifndef LAFS_KRES_LAB
int lafs_kres_launch(const lafs_gemm_nt_args* g, hipStream_t stream) { return kres_launch<0>(g, stream); }
#endif
