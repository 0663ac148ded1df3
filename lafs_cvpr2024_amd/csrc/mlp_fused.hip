// Fused MLP of a ViT-S block for gfx950: two chained GEMMs with the 1536-wide intermediate kept on chip.
//
//   forward  (vision_transformer.py:59-65 Mlp.forward + the residual / DropPath of Block.forward :112):
//       out(f32)[M,384] = resid + seq_scale[row2seq[m]] * ( gelu(X W1^T + b1) W2^T + b2 )
//       (saving pass: gelu'(u) and gelu(u) are also written, bf16 [M,H] -- what the backward and the fc2 weight gradient read)
//   backward (the input-gradient chain of the same lines): du = (dY W2) * gelu'(u)  (written, bf16: the fc1 weight gradient's
//       operand) ;  dX(bf16)[M,384] = du W1
//
// It replaces two launches of lafs_gemm_nt (fc1 + GELU on gemm_kres.hip, fc2 + residual on gemm.hip; in the backward the GELU'
// input gradient and the fc1 input gradient) whose 1536-wide intermediate made a round trip through HBM: 25 216 teacher rows write
// and re-read 77 MB of gelu(u) per layer, 44 160 student rows read 135 MB of gelu(u) in fc2 and 135 MB of du in the fc1 input gradient.
// Both GEMMs of a direction have the SAME shape of dataflow -- [rows, 384] x Wa[H, 384]^T -> [rows, H] -> x Wb[384, H]^T ->
// [rows, 384] (backward: Wa = fc2.weight^T shadow, Wb = fc1.weight^T shadow) -- so one kernel template serves both.
//
// Structure (why: DESIGN.md section 4 "Fused MLP"; what round 2's tools/lab/mlp_fused.hip lacked):
//   * one 8-wave workgroup per CU = TWO waves per SIMD, each wave owning 16 token rows of a 128-row unit: its 16 x 384 operand
//     is resident in registers (48), its 16 x 384 result tile of GEMM 2 in accumulators (96), a 16 x 64 tile of GEMM 1 in 16 more.
//     With one wave per SIMD (32 rows each, round 2) every LDS-DMA issue stall, fragment-read latency and GELU evaluation was
//     exposed; two waves cover each other's stalls (MFMA and VALU time still add on a SIMD: profiles/lab_mfma_valu_overlap.txt);
//   * the hidden axis is walked in items of 64 columns: stage A = 64 rows of Wa (64 x 768 B = 48 KiB, whole rows), stage B = the
//     item's 64-column slice of Wb (384 rows x 128 B = 48 KiB, whole cache lines); a ring of three 48-KiB LDS buffers, filled by
//     LDS-DMA (global_load_lds_dwordx4) with counted s_waitcnt vmcnt and one s_barrier per stage (48 MFMAs per wave and stage);
//   * GEMM 1 computes C^T blocks (first MFMA operand = 16 weight rows, second = 16 tokens) with the weight rows of a stage
//     interleaved so that a lane ends with 8 CONSECUTIVE hidden columns of its token per 32-column slice: after GELU (or the
//     multiplication by gelu'(u)) and the bf16 rounding those 8 values ARE the lane's B-operand fragment of GEMM 2 -- the
//     intermediate never leaves the registers, not even for the LDS;
//   * same MFMA (v_mfma_f32_16x16x32_bf16), same operand roles, same ascending k order, same bias / GELU / residual arithmetic as
//     the two kernels it replaces: results are BIT-IDENTICAL to the two-launch path (tests/test_gpu_kernels.py).
#include "common.hpp"
#include "lafs_hip.h"
#include "ctx.hpp"

// lab builds only (tools/lab/Makefile: mlp_abl): timing ablations -- 1 no GELU / gelu' arithmetic, 2 no fragment reads, 4 no LDS-DMA
// after the prologue's stages, 8 no MFMA, 16 no workgroup barriers (racy), 32 / 64 every second fragment read of stage B / A only.
// 128 phase time stamps of wave 0 (s_memtime; lafs_mlp_args::ctx carries the output buffer).  0 in the library.
#ifndef LAFS_MLP_ABL
#define LAFS_MLP_ABL 0
#endif
#define MABL(bit) ((LAFS_MLP_ABL & (bit)) != 0)

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

constexpr int D = 384;                       // embedding width: reduction of GEMM 1, output width of GEMM 2 (compile time: sizes the registers)
constexpr int ROWB = D * 2;                  // bytes of a 384-wide bf16 row
constexpr int CPR = D / 8;                   // 16-byte chunks per such row (48)
constexpr int HC = 64;                       // hidden columns per item
constexpr int STAGE = HC * ROWB;             // 48 KiB: 64 rows x 768 B (Wa rows / token rows) or 384 rows x 128 B (Wb slice)
constexpr int NSTG = 3;
constexpr int NTH = 512, NWV = NTH / 64;
constexpr int NDMA = STAGE / 16 / NTH;       // LDS-DMA instructions per thread and stage (6)
constexpr int NKK = D / 32;                  // k steps of GEMM 1 (12)
constexpr int NOB = D / 16;                  // 16-column output blocks of GEMM 2 (24)
constexpr int UROWS = 16 * NWV;              // token rows per workgroup (128)
constexpr int MAXH = 1536;
#ifndef LAFS_MLP_FD
#define LAFS_MLP_FD 6
#endif
constexpr int FD = LAFS_MLP_FD;              // fragment reads in flight ahead of their MFMAs
static_assert(STAGE % (16 * NTH) == 0 && STAGE == D * 128 && NWV == 8, "stage layout");

struct MArgs {
  const bf16_t* X; int ldx;
  const bf16_t* Wa; int ldwa;
  const bf16_t* Wb; int ldwb;
  int M, H;
  const float* bias_a; const float* bias_b;
  const float* resid; int ldr;
  const float* seq_scale; const int* row2seq;
  void* out; int ldo;
  bf16_t* g; int ldg;                        // gelu'(u): written by the saving forward, read by the backward
  bf16_t* a; int lda;                        // saving forward: gelu(u) written;  backward: du written
  // LNB (backward): LayerNorm-2 backward in the final epilogue -- x (resid), its statistics (ln_stats), gamma (ln_g); the residual
  // gradient stream g_io (read-modify-write), its DropPath-scaled bf16 copy gb (the projection input gradient's operand), and this
  // workgroup's gamma / beta sums part[blockIdx][2][384]
  float* g_io; int ldgio; bf16_t* gb; int ldgb; float* part;
  // forward modes: the NEXT block's LayerNorm 1 on the finished rows, in the final epilogue (they are in registers there: no loads)
  const float* nln_g; const float* nln_b; float nln_eps; float* nln_stats; bf16_t* nln_out; int ldnln;
  // PRJ (forward modes with LNP): the attention branch's output projection in front -- resid(x1) = resid0 + scale_p[row2seq] * (X Wp^T + bias_p)
  // is COMPUTED here (X = the attention output rows), written to `resid`, normalised from registers and used as the residual at the end
  const bf16_t* Wp; int ldwp; const float* bias_p; const float* resid0; int ldr0; const float* scale_p;
  const float* ln_g; const float* ln_b; float ln_eps;   // LNP: X = LayerNorm(resid) computed in the prologue
  float* ln_stats; bf16_t* ln_out; int ldln;            //      (mean, rstd) per row and the bf16 operand as by-products (optional)
  int unit_waves;                            // waves of a workgroup that own rows (8: 128-row units; 4: 64-row units, one computing wave per SIMD)
  int row0;                                  // first row of this launch's first unit
  unsigned long long* stamps;                // lab (ablation 128): per workgroup, wave 0: cycles in {wait + barrier A, DMA issue A, MFMA A, mid-epilogue,
};                                           //   wait + barrier B, DMA issue B, MFMA B, prologue, final epilogue, whole kernel}

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

// sum over the 16 lanes of a DPP row (every lane ends with it): rotations by 1, 2, 4, 8 inside the row (row_ror)
template <int CTRL> __device__ __forceinline__ float dpp_rot(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_rot<0x121>(v); v += dpp_rot<0x122>(v); v += dpp_rot<0x124>(v); v += dpp_rot<0x128>(v);
  return v;
}

// MODE: LAFS_MLP_FWD (0) forward-only, LAFS_MLP_FWD_SAVE (1) forward saving gelu'(u) and gelu(u), LAFS_MLP_BWD (2) input gradients
// LNP, backward mode: the LayerNorm backward of norm2 runs in the final epilogue on the finished rows (dX rounded to bf16 as the
// separate path stores it, then dx = LN'(dX), g += dx, gb = bf16(scale g), per-workgroup gamma / beta sums) -- no dX round trip
// through HBM and no lafs_layernorm_bwd launch.
// LNP (forward modes): the GEMM-1 operand is LayerNorm(resid) (vision_transformer.py:112 norm2), computed by each wave for its own 16
// rows with the row arithmetic of ln_fwd2_kernel (layernorm.hip: 32 lanes per row, float4 pieces at columns 4 l + 128 i, the same
// summation order -> the same bits), written into the ring buffers as the stage-A image the fragments are read from.
// PRJ (forward modes, with LNP): the projection GEMM of the attention branch (vision_transformer.py:88-90 `self.proj`, :111 the residual and
// DropPath of Block.forward) runs in front as six more stages of the GEMM-2 kind on the wave's 16 attention-output rows: x1 = x0 +
// scale * (o Wp^T + b) leaves for HBM once (the backward and this kernel's own final epilogue read it), LayerNorm 2 is formed from the
// registers, and the separate projection launch with its read of o / x0 and the prologue's read of x1 are gone.  Same MFMA, operand
// roles, ascending k order and epilogue arithmetic as gemm_kres_kernel<RESID_F32> (accumulators start from the bias): the same bits.
template <int MODE, bool LNP, bool PRJ = false>
__global__ __launch_bounds__(NTH, 1) void mlp_fused_kernel(MArgs p) {
  constexpr bool FWD = (MODE != LAFS_MLP_BWD);
  static_assert(!PRJ || (FWD && LNP), "the projection prologue belongs to the forward modes with LayerNorm 2 inside");
  constexpr int NS = (MODE == LAFS_MLP_FWD_SAVE) ? 4 : (MODE == LAFS_MLP_BWD ? 2 : 0);   // stores of an item's mid-epilogue (active waves)
  constexpr int NL = (MODE == LAFS_MLP_BWD) ? 2 : 0;                                     // its operand loads (every wave)
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTG * STAGE];
  __shared__ __attribute__((aligned(16))) float sba[FWD ? MAXH : 4];
  __shared__ __attribute__((aligned(16))) float sbb[(FWD || LNP) ? D : 4];        // forward: fc2 bias; backward + LNP: gamma
  __shared__ __attribute__((aligned(16))) float snl[FWD ? 2 * D : 4];             // forward: the next block's LayerNorm 1 gamma | beta
  __shared__ __attribute__((aligned(16))) float spj[PRJ ? 3 * D : 4];             // PRJ: projection bias | LayerNorm 2 gamma | beta
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15, q = lane >> 4;
  unsigned long long t_begin = 0;
  if constexpr (MABL(128)) t_begin = __builtin_amdgcn_s_memtime();
  const int u0 = p.row0 + blockIdx.x * (16 * p.unit_waves);   // first row of this workgroup's unit
  const int NI = p.H / HC;                             // items (>= 2)
  const int row = u0 + wave * 16 + t;                  // this lane's token row
  const bool active = wave < p.unit_waves && (u0 + wave * 16) < p.M;   // wave-uniform: the wave owns at least one row of the matrix
  const bool rowok = active && row < p.M;              // (idle waves take part in the LDS-DMA and the barriers only)
  const int rowc = min(row, p.M - 1);

  float sc = 1.0f;                                     // DropPath scale of this lane's row (forward)
  if constexpr (FWD) {
    if (p.seq_scale != nullptr) sc = p.seq_scale[p.row2seq[rowc]];
    for (int i = tid; i < p.H; i += NTH) sba[i] = p.bias_a ? p.bias_a[i] : 0.f;
    for (int i = tid; i < D; i += NTH) sbb[i] = p.bias_b ? p.bias_b[i] : 0.f;
    if (p.nln_g != nullptr)
      for (int i = tid; i < D; i += NTH) { snl[i] = p.nln_g[i]; snl[D + i] = p.nln_b[i]; }
    if constexpr (PRJ)
      for (int i = tid; i < D; i += NTH) { spj[i] = p.bias_p ? p.bias_p[i] : 0.f; spj[D + i] = p.ln_g[i]; spj[2 * D + i] = p.ln_b[i]; }
    __syncthreads();                                   // (also keeps these loads out of the counted waits below)
  } else if constexpr (LNP) {
    for (int i = tid; i < D; i += NTH) sbb[i] = p.ln_g[i];
    __syncthreads();
  }

  // ---- LDS images.  Stage A (64 rows of 48 chunks): logical chunk c of row rho at position c ^ (rho & 15); MFMA row s of 16-row
  // group gi (rho = 16 gi + s) lands in lane quarter s >> 2, register s & 3, and carries hidden column 32 (gi >> 1) + 8 (s >> 2) +
  // 4 (gi & 1) + (s & 3) of the item: groups (2 ks, 2 ks + 1) give a lane 8 consecutive columns of slice ks.  Stage B (384 rows of
  // 8 chunks): chunk c of row rho at position c ^ (rho & 7); row rho carries output column rho (fp32 results: 4 consecutive columns
  // = 16 bytes per MFMA) or, for the bf16 result of the backward, column 32 (rho >> 5) + 8 ((rho & 15) >> 2) + 4 ((rho >> 4) & 1) +
  // (rho & 3): blocks (2 P, 2 P + 1) give a lane 8 consecutive columns.  The images are lane-linear for the DMA, so swizzle and
  // permutation go on the source address.
  int doffa[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int x = i * NTH + tid, rho = x / CPR, cp = x % CPR, c = cp ^ (rho & 15);
    const int s16 = rho & 15, gi = rho >> 4;
    const int hrel = 32 * (gi >> 1) + 8 * (s16 >> 2) + 4 * (gi & 1) + (s16 & 3);
    doffa[i] = (hrel * p.ldwa + c * 8) * 2;            // bytes from the item's first Wa row
  }
  int doffb;                                           // piece i adds i * 64 rows: a scalar offset (the row permutation keeps multiples of 32)
  {
    const int rho = tid >> 3, cp = tid & 7, c = cp ^ (rho & 7);
    const int rsrc = FWD ? rho : (32 * (rho >> 5) + 8 * ((rho & 15) >> 2) + 4 * ((rho >> 4) & 1) + (rho & 3));
    doffb = (rsrc * p.ldwb + c * 8) * 2;
  }
  const unsigned lds0 = lds_addr_of(smem);
  auto issue_a = [&](int item, int buf) {              // Wa rows [64 item, 64 item + 64) -> ring buffer buf
    const bf16_t* base = p.Wa + (size_t)item * HC * p.ldwa;
    const unsigned st = lds0 + buf * STAGE + wave * 1024;
    fence();
    if (MABL(4) && item >= 2) return;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) lds_dma16_m0_s(base, (unsigned)doffa[i], st + i * (NTH * 16));
    fence();
  };
  auto issue_b = [&](int item, int buf) {              // Wb[:, 64 item .. + 64) -> ring buffer buf
    const bf16_t* base = p.Wb + (size_t)item * HC;
    const unsigned st = lds0 + buf * STAGE + wave * 1024;
    fence();
    if (MABL(4) && item >= 1) return;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) lds_dma16_m0_s(base + (size_t)i * 64 * p.ldwb, (unsigned)doffb, st + i * (NTH * 16));
    fence();
  };
  int doffp = 0;                                       // PRJ: the projection weight's slices, stage-B images (output column = row, unpermuted)
  if constexpr (PRJ) { const int rho = tid >> 3, cp = tid & 7, c = cp ^ (rho & 7); doffp = (rho * p.ldwp + c * 8) * 2; }
  auto issue_p = [&](int slice, int buf) {             // Wp[:, 64 slice .. + 64) -> ring buffer buf
    const bf16_t* base = p.Wp + (size_t)slice * HC;
    const unsigned st = lds0 + buf * STAGE + wave * 1024;
    fence();
#pragma unroll
    for (int i = 0; i < NDMA; ++i) lds_dma16_m0_s(base + (size_t)i * 64 * p.ldwp, (unsigned)doffp, st + i * (NTH * 16));
    fence();
  };
  auto issue_rows = [&](int half, int buf) {           // token rows [u0 + 64 half, + 64) as a stage-A image (unpermuted)
    const unsigned st = lds0 + buf * STAGE + wave * 1024;
    fence();
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int x = i * NTH + tid, rho = x / CPR, cp = x % CPR, c = cp ^ (rho & 15);
      const int r = min(u0 + 64 * half + rho, p.M - 1);
      lds_dma16_m0(p.X + (size_t)r * p.ldx + c * 8, st + i * (NTH * 16));
    }
    fence();
  };

  // fragment offsets.  Stage A: row 16 gi + t, chunk (4 kk + q) ^ t (four registers by kk & 3, the rest immediates);
  // stage B: row 16 ob + t, chunk (4 ks + q) ^ (t & 7)
  int foff[4], goff[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    foff[i] = t * ROWB + (((4 * i + q) ^ t) << 4);
    asm volatile("" : "+v"(foff[i]));
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    goff[ks] = t * 128 + (((4 * ks + q) ^ (t & 7)) << 4);
    asm volatile("" : "+v"(goff[ks]));
  }

  // ---- prologue: the unit's 128 token rows through ring buffers 0 and 1 while stage 0 flies into buffer 2
  // PRJ: x1 and the packed LayerNorm output of this lane's row, kept until the three stages of the main loop's start are in flight:
  // their stores are issued BEHIND those stages, so no wait of the main loop has to drain them (58 MB per launch leave at once)
  f32x4_t accp[PRJ ? NOB : 1];
  uint2 hpk[PRJ ? NOB : 1];
  float pmean = 0.f, prstd = 0.f;
  if constexpr (PRJ) {
    issue_rows(0, 0);                                  // the attention output rows (p.X) through buffers 0 and 1
    issue_rows(1, 1);
    issue_p(0, 2);
    wait_vm<NDMA>();                                   // this thread's row pieces have landed (slice 0 is younger)
    __builtin_amdgcn_s_barrier();
    bf16x8_t oreg[NKK];
    {
      const unsigned char* st = smem + (wave >> 2) * STAGE + (wave & 3) * (16 * ROWB);
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) oreg[kk] = *reinterpret_cast<const bf16x8_t*>(st + foff[kk & 3] + (kk >> 2) * 256);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                      // buffers 0 and 1 are free again
    issue_p(1, 0);
    issue_p(2, 1);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) accp[ob] = *reinterpret_cast<const f32x4_t*>(spj + 16 * ob + 4 * q);
    // six stages of 64 k each, ring discipline of the main loop: wait for the slice, barrier, refill the buffer read one step ago
    // (slices 3, 4, 5, then stage 0 of the MLP into buffer 2, where the main loop expects it), 48 MFMAs
#pragma unroll
    for (int sl = 0; sl < D / HC; ++sl) {
      if (sl == 0) wait_vm<2 * NDMA>(); else wait_vm<NDMA>();    // younger than slice sl: slices 1, 2 | the one stage issued a step ago
      __builtin_amdgcn_s_barrier();
      if (sl >= 1 && sl <= 3) issue_p(sl + 2, (sl + 1) % 3);
      else if (sl == 4) issue_a(0, 2);
      if (active) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* st = smem + ((2 + sl) % 3) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ob = 0; ob < NOB; ++ob) {
            const bf16x8_t w = *reinterpret_cast<const bf16x8_t*>(st + goff[ks] + ob * 2048);
            accp[ob] = mfma16(w, oreg[2 * sl + ks], accp[ob]);
          }
        __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);
#pragma unroll
        for (int i = 0; i < 2 * NOB - FD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, FD, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_barrier();                      // every wave is done with the last slice: buffers 0 and 1 take the row images
    if (active) {
      float sca = 1.0f;                                // DropPath scale of the attention branch
      if (p.scale_p != nullptr) sca = p.scale_p[p.row2seq[rowc]];
      const float* r0 = p.resid0 + (size_t)rowc * p.ldr0;
#pragma unroll
      for (int o0 = 0; o0 < NOB; o0 += 6) {
        uint4 r4[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) r4[j] = *reinterpret_cast<const uint4*>(r0 + 16 * (o0 + j) + 4 * q);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const f32x4_t a4 = accp[o0 + j];
          const float v0 = __uint_as_float(r4[j].x) + sca * a4[0], v1 = __uint_as_float(r4[j].y) + sca * a4[1];
          const float v2 = __uint_as_float(r4[j].z) + sca * a4[2], v3 = __uint_as_float(r4[j].w) + sca * a4[3];
          accp[o0 + j] = f32x4_t{v0, v1, v2, v3};
        }
      }
      // LayerNorm 2 on the finished rows, in the order of ln_fwd2_kernel (see the next block's LayerNorm in the final epilogue)
      auto tree = [&](auto&& part) {
        float pm[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) pm[m] = part(part(part(0.f, m), m + 8), m + 16);
        float v = ((pm[0] + pm[4]) + (pm[2] + pm[6])) + ((pm[1] + pm[5]) + (pm[3] + pm[7]));
        v += __shfl_xor(v, 32, 64); v += __shfl_xor(v, 16, 64);
        return v;
      };
      const float sm = tree([&](float s_, int ob) { return ln_sum4(s_, make_float4(accp[ob][0], accp[ob][1], accp[ob][2], accp[ob][3])); });
      const float mean = sm / (float)D;
      const float qs = tree([&](float s_, int ob) { return ln_sq4(s_, make_float4(accp[ob][0], accp[ob][1], accp[ob][2], accp[ob][3]), mean); });
      const float rstd = rsqrtf(qs / (float)D + p.ln_eps);
      pmean = mean; prstd = rstd;
      unsigned char* img = smem + (wave >> 2) * STAGE + ((wave & 3) * 16 + t) * ROWB;      // this lane's row of the stage-A image
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        const int n = 16 * ob + 4 * q;
        const float4 g4 = *reinterpret_cast<const float4*>(spj + D + n), b4 = *reinterpret_cast<const float4*>(spj + 2 * D + n);
        const float h0 = ln_out1(accp[ob][0], mean, rstd, g4.x, b4.x), h1 = ln_out1(accp[ob][1], mean, rstd, g4.y, b4.y);
        const float h2 = ln_out1(accp[ob][2], mean, rstd, g4.z, b4.z), h3 = ln_out1(accp[ob][3], mean, rstd, g4.w, b4.w);
        const uint2 pk = make_uint2(pack_bf2(h0, h1), pack_bf2(h2, h3));
        *reinterpret_cast<uint2*>(img + (((2 * ob + (q >> 1)) ^ t) << 4) + (q & 1) * 8) = pk;     // chunk n / 8 at position chunk ^ (row & 15)
        hpk[ob] = pk;
      }
    }
    // (a wave reads back only its own rows, and the LDS serves a wave's operations in order: no barrier in between)
  } else if constexpr (!(LNP && FWD)) {
    issue_rows(0, 0);
    issue_rows(1, 1);
    issue_a(0, 2);
    wait_vm<NDMA>();                                   // this thread's row pieces have landed (stage 0 is younger)
    __builtin_amdgcn_s_barrier();
  } else {
    issue_a(0, 2);
    if (active) {
      const int l = lane & 31, rsel = lane >> 5;
      float4 xr[8][3], g4[3], b4[3];                   // the wave's 16 rows as 8 pairs: all loads in flight before the first reduction
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = min(u0 + wave * 16 + 2 * j + rsel, p.M - 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) xr[j][i] = *reinterpret_cast<const float4*>(p.resid + (size_t)r * p.ldr + l * 4 + 128 * i);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        g4[i] = *reinterpret_cast<const float4*>(p.ln_g + l * 4 + 128 * i);
        b4[i] = *reinterpret_cast<const float4*>(p.ln_b + l * 4 + 128 * i);
      }
      unsigned char* img = smem + (wave >> 2) * STAGE;
      // the 8 row pairs' reductions side by side (8 independent butterflies in flight instead of 8 x 2 dependent chains of 5 cross-lane
      // exchanges); per row the arithmetic and its order are ln_fwd2_kernel's
      float mean[8], rstd[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) s = ln_sum4(s, xr[j][i]);
        mean[j] = s;
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) mean[j] += __shfl_xor(mean[j], o, 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        mean[j] = mean[j] / (float)D;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) qq = ln_sq4(qq, xr[j][i], mean[j]);
        rstd[j] = qq;
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) rstd[j] += __shfl_xor(rstd[j], o, 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 (&cur)[3] = xr[j];
        const int r = u0 + wave * 16 + 2 * j + rsel;
        const float mn = mean[j], rs = rsqrtf(rstd[j] / (float)D + p.ln_eps);
        if (p.ln_stats != nullptr && l == 0 && r < p.M) *reinterpret_cast<float2*>(p.ln_stats + 2 * (size_t)r) = make_float2(mn, rs);
        const int rho = (wave & 3) * 16 + 2 * j + rsel;                 // row of the stage-A image
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float o0 = ln_out1(cur[i].x, mn, rs, g4[i].x, b4[i].x), o1 = ln_out1(cur[i].y, mn, rs, g4[i].y, b4[i].y);
          const float o2 = ln_out1(cur[i].z, mn, rs, g4[i].z, b4[i].z), o3 = ln_out1(cur[i].w, mn, rs, g4[i].w, b4[i].w);
          const uint2 pk = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
          const int c = (l >> 1) + 16 * i;                               // 16-byte chunk of columns 4 l + 128 i .. + 3, half l & 1
          *reinterpret_cast<uint2*>(img + rho * ROWB + ((c ^ (rho & 15)) << 4) + (l & 1) * 8) = pk;
          if (p.ln_out != nullptr && r < p.M) *reinterpret_cast<uint2*>(p.ln_out + (size_t)r * p.ldln + l * 4 + 128 * i) = pk;
        }
      }
    }
    // (a wave reads back only its own rows, and the LDS serves a wave's operations in order: no barrier in between)
  }
  bf16x8_t areg[NKK];                                  // 16 tokens x 384 k: lane (t, q) holds k = 32 kk + 8 q .. + 7 of token t
  {
    const unsigned char* st = smem + (wave >> 2) * STAGE + (wave & 3) * (16 * ROWB);
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) areg[kk] = *reinterpret_cast<const bf16x8_t*>(st + foff[kk & 3] + (kk >> 2) * 256);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();                        // buffers 0 and 1 are free again
  uint4 gp[NL > 0 ? NL : 1];                           // backward: gelu'(u) of the item's two slices (8 bf16 each)
  auto fetch_g = [&](int item) {                       // exactly NL loads
    if constexpr (NL > 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) gp[ks] = *reinterpret_cast<const uint4*>(p.g + (size_t)rowc * p.ldg + item * HC + 32 * ks + 8 * q);
      fence();
    }
  };
  fetch_g(0);
  issue_b(0, 0);                                       // stage 1
  issue_a(1, 1);                                       // stage 2
  const bool prj_h = PRJ && p.ln_out != nullptr;       // (uniform) the LayerNorm output also leaves for HBM: 24 more stores
  if constexpr (PRJ) {
    if (rowok) {                                       // (active waves issue every one of these stores: at least one of their rows exists)
      float* x1 = const_cast<float*>(p.resid) + (size_t)row * p.ldr;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) st16f(x1 + 16 * ob + 4 * q, accp[ob][0], accp[ob][1], accp[ob][2], accp[ob][3]);
      if (prj_h) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) *reinterpret_cast<uint2*>(p.ln_out + (size_t)row * p.ldln + 16 * ob + 4 * q) = hpk[ob];
      }
      if (p.ln_stats != nullptr && q == 0) *reinterpret_cast<float2*>(p.ln_stats + 2 * (size_t)row) = make_float2(pmean, prstd);
    }
    fence();
  }
  // in flight, oldest first: stage 0, [gelu' of item 0], stage 1, stage 2

  f32x4_t acc2[NOB];
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) acc2[ob] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t acc1[4];
  bf16x8_t gfrag[2];
  // NOTE: the waits below count THIS wave's vector-memory instructions in issue order (loads, LDS-DMA and stores retire in order on
  // gfx950).  In flight here, oldest first: [whatever the prologue stored: LayerNorm by-products], stage 0, [gelu' of item 0], stage 1,
  // stage 2 -- a wait for "at most the 12 youngest" therefore covers stage 0 and everything older.

  bf16x8_t wprev = areg[0];                            // (lab: ablations 32 / 64)
  unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
  auto lap = [&](int slot) {
    if constexpr (MABL(128)) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tacc[slot] += now - t_last;
      t_last = now;
    }
  };
  if constexpr (MABL(128)) { t_last = __builtin_amdgcn_s_memtime(); tacc[7] = t_last - t_begin; }
  int buf = 2;                                         // ring buffer of the stage consumed next
  auto next_buf = [&](int b) { return (b + 1 == NSTG) ? 0 : b + 1; };
  for (int it = 0; it < NI; ++it) {
    const bool has_next = (it + 1 < NI);
    // ================= stage A of the item: acc1 = Wa rows x tokens (48 MFMAs)
    if constexpr (FWD) {
#pragma unroll
      for (int gi = 0; gi < 4; ++gi)
        acc1[gi] = *reinterpret_cast<const f32x4_t*>(sba + it * HC + 32 * (gi >> 1) + 8 * q + 4 * (gi & 1));
    } else {
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) acc1[gi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    // younger than this stage's DMA: the previous item's mid-epilogue stores, the stage B issued behind it, this item's gelu' loads
    if (PRJ && it <= 1 && active) {                    // PRJ: the x1 (24) [+ LayerNorm output (24)] stores sit behind stages 1 and 2 (the statistics store is not counted: a stronger wait)
      if (it == 0) { if (prj_h) wait_vm<2 * NDMA + 2 * NOB>(); else wait_vm<2 * NDMA + NOB>(); }
      else { if (prj_h) wait_vm<NS + NDMA + 2 * NOB>(); else wait_vm<NS + NDMA + NOB>(); }
    }
    else if (it == 0) wait_vm<2 * NDMA>();             // younger than stage 0 and the gelu' loads of item 0: stages 1 and 2
    else if (active) wait_vm<NS + NDMA + NL>();
    else wait_vm<NDMA + NL>();
    if constexpr (!MABL(16)) __builtin_amdgcn_s_barrier();
    lap(0);
    if (it > 0 && has_next) issue_a(it + 1, next_buf(next_buf(buf)));           // stage 2 it + 2 (item 0: issued by the prologue)
    lap(1);
    if (active) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* st = smem + buf * STAGE;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
          bf16x8_t w = areg[(kk + 1) % NKK];
          if constexpr (MABL(64)) { if (gi % 2 == 0) wprev = *reinterpret_cast<const bf16x8_t*>(st + foff[kk & 3] + (kk >> 2) * 256 + gi * (16 * ROWB)); w = wprev; }
          else if constexpr (!MABL(2)) w = *reinterpret_cast<const bf16x8_t*>(st + foff[kk & 3] + (kk >> 2) * 256 + gi * (16 * ROWB));
          if constexpr (!MABL(8)) acc1[gi] = mfma16(w, areg[kk], acc1[gi]);
          else asm volatile("" :: "v"(w));
        }
      if constexpr ((LAFS_MLP_ABL & (2 | 8 | 32 | 64)) == 0) {
      __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);
#pragma unroll
      for (int i = 0; i < 4 * NKK - FD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, FD, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    buf = next_buf(buf);
    lap(2);
    // ----------------- mid-epilogue: GELU (forward) / x gelu'(u) (backward), bf16 rounding, hand-over to GEMM 2 in registers
    fence();
    if constexpr (MODE == LAFS_MLP_BWD) {              // younger than the gelu' loads: the stage A issued above (item 0: stages 1 and 2)
      if (it == 0) wait_vm<2 * NDMA>();
      else if (has_next) wait_vm<NDMA>();
      else wait_vm<0>();
    }
    if (active) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc1[2 * ks][r]; v[4 + r] = acc1[2 * ks + 1][r]; }
      const int n = it * HC + 32 * ks + 8 * q;
      if constexpr (MABL(1)) {
      } else if constexpr (MODE == LAFS_MLP_FWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
      } else if constexpr (MODE == LAFS_MLP_FWD_SAVE) {
        float dv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { float gv; gelu_both_f(v[e], gv, dv[e]); v[e] = gv; }
        if (rowok) st16(p.g + (size_t)row * p.ldg + n, pack_bf2(dv[0], dv[1]), pack_bf2(dv[2], dv[3]), pack_bf2(dv[4], dv[5]), pack_bf2(dv[6], dv[7]));
      } else {
        const uint4 a4 = gp[ks];
        v[0] *= bf_lo(a4.x); v[1] *= bf_hi(a4.x); v[2] *= bf_lo(a4.y); v[3] *= bf_hi(a4.y);
        v[4] *= bf_lo(a4.z); v[5] *= bf_hi(a4.z); v[6] *= bf_lo(a4.w); v[7] *= bf_hi(a4.w);
      }
      const u32x4_t pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
      gfrag[ks] = __builtin_bit_cast(bf16x8_t, pk);
      if constexpr (MODE != LAFS_MLP_FWD) {
        if (rowok) st16(p.a + (size_t)row * p.lda + n, pk[0], pk[1], pk[2], pk[3]);
      }
    }
    }
    fence();
    lap(3);
    // ================= stage B of the item: acc2 += Wb slice rows x intermediate (48 MFMAs)
    // younger than this stage's DMA: the stage A issued above (if any) and the mid-epilogue's stores
    if (PRJ && it == 0 && active) { if (prj_h) wait_vm<NDMA + NS + 2 * NOB>(); else wait_vm<NDMA + NS + NOB>(); }   // (NI >= 2: has_next)
    else if (has_next) { if (active) wait_vm<NDMA + NS>(); else wait_vm<NDMA>(); }
    else { if (active) wait_vm<NS>(); else wait_vm<0>(); }
    if constexpr (!MABL(16)) __builtin_amdgcn_s_barrier();
    lap(4);
    if (has_next) {
      issue_b(it + 1, next_buf(next_buf(buf)));        // stage 2 it + 3
      fetch_g(it + 1);
    }
    lap(5);
    if (active) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* st = smem + buf * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
          bf16x8_t w = areg[ob % NKK];
          if constexpr (MABL(32)) { if (ob % 2 == 0) wprev = *reinterpret_cast<const bf16x8_t*>(st + goff[ks] + ob * 2048); w = wprev; }
          else if constexpr (!MABL(2)) w = *reinterpret_cast<const bf16x8_t*>(st + goff[ks] + ob * 2048);
          if constexpr (!MABL(8)) acc2[ob] = mfma16(w, gfrag[ks], acc2[ob]);
          else asm volatile("" :: "v"(w));
        }
      if constexpr ((LAFS_MLP_ABL & (2 | 8 | 32 | 64)) == 0) {
      __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);
#pragma unroll
      for (int i = 0; i < 2 * NOB - FD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, FD, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    buf = next_buf(buf);
    lap(6);
  }
  auto stamp_out = [&]() {
    if constexpr (MABL(128)) {
      if (tid == 0 && p.stamps != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        tacc[8] = now - t_last; tacc[9] = now - t_begin;
        for (int i = 0; i < 10; ++i) p.stamps[(size_t)blockIdx.x * 10 + i] = tacc[i];
      }
    }
  };

  // ---- final epilogue of the unit: lane (t, q) owns row `row` and, per output block ob, 4 consecutive fp32 columns (forward)
  // or, per block pair, 8 consecutive bf16 columns (backward)
  if constexpr (!FWD && LNP) {
    // ------ LayerNorm backward on the wave's 16 finished rows.  Lane (t, q): row t, columns 32 P + 8 q + e (P < 12, e < 8).
    float xh[NOB][4];                                   // x-hat, same register <-> column map as acc2
    float s1 = 0.f, s2 = 0.f, mean = 0.f, rstd = 0.f, scl = 1.0f;
    if (active) {
      const float2 ms = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (size_t)rowc);
      mean = ms.x; rstd = ms.y;
      if (p.seq_scale != nullptr) scl = p.seq_scale[p.row2seq[rowc]];
      const float* xr = p.resid + (size_t)rowc * p.ldr;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {                // all 24 pieces of the row in flight before the first use
        const float4 v = *reinterpret_cast<const float4*>(xr + 32 * (ob >> 1) + 8 * q + 4 * (ob & 1));
        xh[ob][0] = v.x; xh[ob][1] = v.y; xh[ob][2] = v.z; xh[ob][3] = v.w;
      }
    }
    float* red = reinterpret_cast<float*>(smem + 2 * STAGE);      // [wave][gamma 384 | beta 384]: ring buffer 2 holds stage 45, read by all
    if (active) {
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        const f32x4_t gm = *reinterpret_cast<const f32x4_t*>(sbb + 32 * (ob >> 1) + 8 * q + 4 * (ob & 1));
        float cg[4], cb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // dX as the separate path hands it over: rounded to bf16; rows past M contribute nothing
          const float dy = rowok ? bf2f(f2bf(acc2[ob][r])) : 0.f;
          const float xv = (xh[ob][r] - mean) * rstd;
          xh[ob][r] = xv;
          cg[r] = dy * xv; cb[r] = dy;
          const float d = dy * gm[r];
          acc2[ob][r] = d;
          s1 += d; s2 += d * xv;
        }
        // column sums over the wave's 16 rows: rotations inside the 16-lane row (DPP row_ror 1, 2, 4, 8)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cg[r] = row16_sum(cg[r]);
          cb[r] = row16_sum(cb[r]);
        }
        if (t == 0) {
          const int c = 32 * (ob >> 1) + 8 * q + 4 * (ob & 1);
          st16f(red + wave * (2 * D) + c, cg[0], cg[1], cg[2], cg[3]);
          st16f(red + wave * (2 * D) + D + c, cb[0], cb[1], cb[2], cb[3]);
        }
      }
      // row sums over the four quarter lanes of the row
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      const float m1 = s1 / (float)D, m2 = s2 / (float)D;
      if (rowok) {
        float* gr = p.g_io + (size_t)row * p.ldgio;
        bf16_t* gbr = p.gb + (size_t)row * p.ldgb;
#pragma unroll
        for (int P0 = 0; P0 < NOB / 2; P0 += 3) {
          float4 old[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) old[j] = *reinterpret_cast<const float4*>(gr + 32 * (P0 + (j >> 1)) + 8 * q + 4 * (j & 1));
#pragma unroll
          for (int jp = 0; jp < 3; ++jp) {
            float o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int ob = 2 * (P0 + jp) + h;
              const float4 og = old[2 * jp + h];
              o[4 * h + 0] = rstd * (acc2[ob][0] - m1 - xh[ob][0] * m2) + og.x;
              o[4 * h + 1] = rstd * (acc2[ob][1] - m1 - xh[ob][1] * m2) + og.y;
              o[4 * h + 2] = rstd * (acc2[ob][2] - m1 - xh[ob][2] * m2) + og.z;
              o[4 * h + 3] = rstd * (acc2[ob][3] - m1 - xh[ob][3] * m2) + og.w;
              st16f(gr + 32 * (P0 + jp) + 8 * q + 4 * h, o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3]);
            }
            st16(gbr + 32 * (P0 + jp) + 8 * q, pack_bf2(scl * o[0], scl * o[1]), pack_bf2(scl * o[2], scl * o[3]),
                 pack_bf2(scl * o[4], scl * o[5]), pack_bf2(scl * o[6], scl * o[7]));
          }
        }
      }
    }
    __syncthreads();                                    // every computing wave's column sums are in the LDS
    {
      const int n_act = min(p.unit_waves, (p.M - u0 + 15) / 16);      // waves that own rows (>= 1)
      for (int c = tid; c < 2 * D; c += NTH) {
        float sg = 0.f;
        for (int w = 0; w < n_act; ++w) sg += red[w * (2 * D) + c];
        p.part[(size_t)blockIdx.x * (2 * D) + c] = sg;               // slot layout of lafs_layernorm_bwd: [workgroup][gamma | beta][D]
      }
    }
    stamp_out();
    return;
  }
  if (!rowok) { stamp_out(); return; }
  if constexpr (FWD) {
    const float* rs = p.resid + (size_t)row * p.ldr;
    float* o = reinterpret_cast<float*>(p.out) + (size_t)row * p.ldo;
#pragma unroll
    for (int o0 = 0; o0 < NOB; o0 += 6) {
      uint4 r4[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) r4[j] = *reinterpret_cast<const uint4*>(rs + 16 * (o0 + j) + 4 * q);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int n = 16 * (o0 + j) + 4 * q;
        const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(sbb + n);
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = acc2[o0 + j][r] + b4[r];
        w[0] = __uint_as_float(r4[j].x) + sc * w[0]; w[1] = __uint_as_float(r4[j].y) + sc * w[1];
        w[2] = __uint_as_float(r4[j].z) + sc * w[2]; w[3] = __uint_as_float(r4[j].w) + sc * w[3];
        st16f(o + n, w[0], w[1], w[2], w[3]);
        acc2[o0 + j] = f32x4_t{w[0], w[1], w[2], w[3]};          // (kept: the next block's LayerNorm below)
      }
    }
    if (p.nln_g != nullptr) {
      // LayerNorm 1 of the NEXT block (vision_transformer.py:110 `self.norm1` of block l + 1) on the rows just finished: lane (t, q) holds
      // the float4 groups f = 4 ob + q of its row, the row's other three lanes (q') the rest.  The sums are formed in the ORDER of
      // ln_fwd2_kernel (csrc/layernorm.hip: lane l of 32 holds f = l, l + 32, l + 64 and the 32 partial sums meet in an xor
      // butterfly 16, 8, 4, 2, 1): here l = 4 (ob % 8) + q, so the butterfly's first three levels are register adds and the last
      // two the exchanges with q' -- the operand and the statistics are bit-identical to the LayerNorm launch this replaces.  No
      // global load at all (gamma / beta sit in LDS since the kernel's start): the launch and its read of the residual stream are gone.
      static_assert(NOB == 24, "the reduction tree below is ln_fwd2_kernel<3>'s");
      auto tree = [&](auto&& part) {                      // part(ob): this lane's contribution of group ob
        float pm[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) pm[m] = part(part(part(0.f, m), m + 8), m + 16);
        float v = ((pm[0] + pm[4]) + (pm[2] + pm[6])) + ((pm[1] + pm[5]) + (pm[3] + pm[7]));
        v += __shfl_xor(v, 32, 64); v += __shfl_xor(v, 16, 64);
        return v;
      };
      const float sm = tree([&](float s, int ob) { return ln_sum4(s, make_float4(acc2[ob][0], acc2[ob][1], acc2[ob][2], acc2[ob][3])); });
      const float mean = sm / (float)D;
      const float qs = tree([&](float s, int ob) { return ln_sq4(s, make_float4(acc2[ob][0], acc2[ob][1], acc2[ob][2], acc2[ob][3]), mean); });
      const float rstd = rsqrtf(qs / (float)D + p.nln_eps);
      if (p.nln_stats != nullptr && q == 0) *reinterpret_cast<float2*>(p.nln_stats + 2 * (size_t)row) = make_float2(mean, rstd);
      bf16_t* ho = p.nln_out + (size_t)row * p.ldnln;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        const int n = 16 * ob + 4 * q;
        const float4 g4 = *reinterpret_cast<const float4*>(snl + n), b4 = *reinterpret_cast<const float4*>(snl + D + n);
        const float h0 = ln_out1(acc2[ob][0], mean, rstd, g4.x, b4.x), h1 = ln_out1(acc2[ob][1], mean, rstd, g4.y, b4.y);
        const float h2 = ln_out1(acc2[ob][2], mean, rstd, g4.z, b4.z), h3 = ln_out1(acc2[ob][3], mean, rstd, g4.w, b4.w);
        *reinterpret_cast<uint2*>(ho + n) = make_uint2(pack_bf2(h0, h1), pack_bf2(h2, h3));
      }
    }
  } else {
    bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + (size_t)row * p.ldo;
#pragma unroll
    for (int P = 0; P < NOB / 2; ++P) {
      const f32x4_t x = acc2[2 * P], y = acc2[2 * P + 1];
      st16(o + 32 * P + 8 * q, pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]));
    }
  }
  stamp_out();
}

#ifndef LAFS_MLP_UW
#define LAFS_MLP_UW 0
#endif
// One workgroup per CU and unit: a launch costs whole rounds of the chip.  Rows beyond the last full round of 128-row units go out as
// a second launch of 64-row units (one computing wave per SIMD: about 0.6 of a full unit's time, tools/lab/NOTES.md) when they fit one
// round that way -- 44 160 rows: 256 x 128 + 178 x 64 instead of two rounds of 128.
template <int MODE, bool LNP, bool PRJ = false>
int launch(MArgs a, int n_cu, hipStream_t s) {
  const int units = (a.M + UROWS - 1) / UROWS;
  int full = units, half = 0;
  constexpr int LAB_UW = LAFS_MLP_UW > 0 ? LAFS_MLP_UW : NWV;
  if (LAFS_MLP_UW > 0) { a.unit_waves = LAB_UW; full = (a.M + 16 * LAB_UW - 1) / (16 * LAB_UW); }      // lab
  else if (n_cu > 0 && units > n_cu) {
    const int rounds = units / n_cu, rest = units - rounds * n_cu;          // rest: units of the last, partial round
    const int rest_rows = a.M - rounds * n_cu * UROWS;
    const int h = (rest_rows + 63) / 64;
    if (rest > 0 && h <= n_cu) { full = rounds * n_cu; half = h; }
  }
  if (full > 0) {
    MArgs b = a;
    if (half > 0) b.M = full * UROWS;                                          // (whole units: nothing ragged in the first launch)
    hipLaunchKernelGGL((mlp_fused_kernel<MODE, LNP, PRJ>), dim3(full), dim3(NTH), 0, s, b);
  }
  if (half > 0) {
    a.unit_waves = 4; a.row0 = full * UROWS;
    hipLaunchKernelGGL((mlp_fused_kernel<MODE, LNP, PRJ>), dim3(half), dim3(NTH), 0, s, a);
  }
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

}  // namespace

extern "C" int lafs_mlp_fused_supported(int dim, int hidden, int rows) {
  return (dim == D && hidden % HC == 0 && hidden >= 2 * HC && hidden <= MAXH && rows > 0) ? 1 : 0;
}

extern "C" int lafs_mlp_fused(const lafs_mlp_args* g, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(g != nullptr, "null arguments");
  LAFS_CHECK_ARG(g->mode == LAFS_MLP_FWD || g->mode == LAFS_MLP_FWD_SAVE || g->mode == LAFS_MLP_BWD, "bad mode");
  LAFS_CHECK_ARG(lafs_mlp_fused_supported(D, g->H, g->M), "hidden width must be a multiple of 64 in [128, 1536], M > 0 (the embedding width is 384)");
  LAFS_CHECK_ARG(g->mode == LAFS_MLP_BWD || g->next_ln_gamma == nullptr ||
                 (g->next_ln_beta != nullptr && g->next_ln_out != nullptr && g->ldnln_next >= D && g->ldnln_next % 4 == 0),
                 "next block's LayerNorm: beta, output and its stride");
  const bool has_ln = g->mode != LAFS_MLP_BWD && g->ln_gamma != nullptr;
  if (g->mode == LAFS_MLP_BWD && g->ln_gamma != nullptr) {
    LAFS_CHECK_ARG(g->resid != nullptr && g->ldr >= D && g->ldr % 4 == 0 && g->ln_stats != nullptr, "LayerNorm backward: x and its statistics");
    LAFS_CHECK_ARG(g->ln_g_io != nullptr && g->ldgio >= D && g->ldgio % 4 == 0 && g->ln_gb_out != nullptr && g->ldgb >= D && g->ldgb % 8 == 0 &&
                   g->ln_part_out != nullptr, "LayerNorm backward: gradient stream, its bf16 copy and the gamma / beta slots");
    LAFS_CHECK_ARG(g->seq_scale == nullptr || g->row2seq != nullptr, "seq_scale needs row2seq");
  }
  LAFS_CHECK_ARG((has_ln || g->X) && g->Wa && g->Wb, "null operand");
  LAFS_CHECK_ARG(!has_ln || (g->ln_beta != nullptr && (g->ln_out == nullptr || (g->ldln >= D && g->ldln % 4 == 0))), "LayerNorm prologue: beta / output stride");
  LAFS_CHECK_ARG(has_ln || (g->ldx >= D && g->ldx % 8 == 0), "operand strides: multiples of 8 elements");
  LAFS_CHECK_ARG(g->ldwa >= D && g->ldwa % 8 == 0 && g->ldwb >= g->H && g->ldwb % 8 == 0, "operand strides: multiples of 8 elements");
  LAFS_CHECK_ARG((g->mode == LAFS_MLP_BWD && g->ln_gamma != nullptr) || (g->out != nullptr && g->ldo >= D && g->ldo % 8 == 0),
                 "output stride: a multiple of 8 elements");
  const bool fwd = g->mode != LAFS_MLP_BWD;
  if (fwd) {
    LAFS_CHECK_ARG(g->resid != nullptr && g->ldr >= D && g->ldr % 4 == 0, "the forward needs the fp32 residual");
    LAFS_CHECK_ARG(g->seq_scale == nullptr || g->row2seq != nullptr, "seq_scale needs row2seq");
  }
  if (g->mode != LAFS_MLP_FWD) {
    LAFS_CHECK_ARG(g->save_grad != nullptr && g->ldsg >= g->H && g->ldsg % 8 == 0, "gelu'(u) buffer");
    LAFS_CHECK_ARG(g->save_act != nullptr && g->ldsa >= g->H && g->ldsa % 8 == 0, "gelu(u) / du buffer");
  }
  MArgs a = {};
  a.X = (const bf16_t*)g->X; a.ldx = g->ldx; a.Wa = (const bf16_t*)g->Wa; a.ldwa = g->ldwa; a.Wb = (const bf16_t*)g->Wb; a.ldwb = g->ldwb;
  a.M = g->M; a.H = g->H; a.bias_a = fwd ? g->bias_a : nullptr; a.bias_b = fwd ? g->bias_b : nullptr;
  a.resid = g->resid; a.ldr = g->ldr; a.seq_scale = g->seq_scale; a.row2seq = g->row2seq;
  a.out = g->out; a.ldo = g->ldo; a.g = (bf16_t*)g->save_grad; a.ldg = g->ldsg; a.a = (bf16_t*)g->save_act; a.lda = g->ldsa;
  a.unit_waves = NWV; a.row0 = 0;
  a.stamps = MABL(128) ? (unsigned long long*)g->ctx : nullptr;
  const int n_cu = (g->ctx != nullptr && !MABL(128)) ? g->ctx->n_cu : 0;                      // (no context: one launch of 128-row units)
  a.nln_g = fwd ? g->next_ln_gamma : nullptr; a.nln_b = g->next_ln_beta; a.nln_eps = g->next_ln_eps; a.nln_stats = g->next_ln_stats;
  a.nln_out = (bf16_t*)g->next_ln_out; a.ldnln = g->ldnln_next;
  const bool lnp = fwd && g->ln_gamma != nullptr;
  a.ln_g = g->ln_gamma; a.ln_b = g->ln_beta; a.ln_eps = g->ln_eps; a.ln_stats = g->ln_stats; a.ln_out = (bf16_t*)g->ln_out; a.ldln = g->ldln;
  const bool prj = lnp && g->proj_x != nullptr;
  if (prj) {
    LAFS_CHECK_ARG(g->proj_w != nullptr && g->proj_resid != nullptr && g->ldpx >= D && g->ldpx % 8 == 0 && g->ldpw >= D && g->ldpw % 8 == 0 &&
                   g->ldpr >= D && g->ldpr % 4 == 0, "projection prologue: weight, residual and strides");
    LAFS_CHECK_ARG(g->proj_scale == nullptr || g->row2seq != nullptr, "proj_scale needs row2seq");
    a.X = (const bf16_t*)g->proj_x; a.ldx = g->ldpx; a.Wp = (const bf16_t*)g->proj_w; a.ldwp = g->ldpw; a.bias_p = g->proj_bias;
    a.resid0 = g->proj_resid; a.ldr0 = g->ldpr; a.scale_p = g->proj_scale;
  }
  switch (g->mode) {
    case LAFS_MLP_FWD: return prj ? launch<LAFS_MLP_FWD, true, true>(a, n_cu, stream) : lnp ? launch<LAFS_MLP_FWD, true>(a, n_cu, stream) : launch<LAFS_MLP_FWD, false>(a, n_cu, stream);
    case LAFS_MLP_FWD_SAVE: return prj ? launch<LAFS_MLP_FWD_SAVE, true, true>(a, n_cu, stream) : lnp ? launch<LAFS_MLP_FWD_SAVE, true>(a, n_cu, stream) : launch<LAFS_MLP_FWD_SAVE, false>(a, n_cu, stream);
    default:
      if (g->ln_gamma != nullptr) {                      // LayerNorm backward in the epilogue: the slots are numbered by workgroup of ONE launch
        a.g_io = g->ln_g_io; a.ldgio = g->ldgio; a.gb = (bf16_t*)g->ln_gb_out; a.ldgb = g->ldgb; a.part = g->ln_part_out;
        return launch<LAFS_MLP_BWD, true>(a, 0, stream);
      }
      return launch<LAFS_MLP_BWD, false>(a, n_cu, stream);
  }
}

extern "C" int lafs_mlp_fused_ln_parts(int rows) { return rows > 0 ? (rows + UROWS - 1) / UROWS : 0; }
