// LAB KERNEL (round 3, NOT part of liblafs_hip.so): measured 1.4-1.8x SLOWER than the two launches it replaces -- student forward
// 257.6 us against 101.9 + 84.3, backward 283.9 against 89.4 + 68.5, whole step 16.78 against 15.77 ms (tools/lab/NOTES.md: at one
// wave per SIMD the MFMA work, the GELU epilogue, the LDS-DMA issue and the store stalls of an item serialise).  Kept as the record
// of the experiment; `make -C tools/lab mlp_fused.o` compiles it (it was parity-checked against fp32 torch and bit-for-bit against
// the two-launch path for the saved tensors while it was wired into the engine).
//
// Fused two-GEMM MLP kernels for gfx950 (embedding width 384): the hidden activation never round-trips through HBM between the GEMMs.
//
//   forward   x_out = x1 + s * (gelu(h2 W1^T + b1) W2^T + b2)          Mlp.forward + DropPath + residual
//             (reference vision_transformer.py:49-65, 107-113); optionally stores gelu'(u) and gelu(u) for the backward
//   backward  du = (g W2) .* gelu'(u);   dh = du W1                      input gradients of fc2 / GELU / fc1 (loss.backward(),
//             lafs_train.py:593-600); du is stored (the fc1 weight gradient reads it)
//
// Why (profiles/round2_*: every GEMM of the block is bound by bytes through the CU's memory pipeline, 2-2.4x above its own HBM
// floor): as separate launches fc1 writes gelu(u) (136 MB at C2) and fc2 reads it back through the LDS, 99 + 110 us; the backward
// pair writes and re-reads du, 91 + 70 us.  Here ONE wave per SIMD owns 32 token rows for a whole pass:
//   * the rows' 384-wide operand stays in registers (96 VGPRs, as in gemm_kres.hip);
//   * per 64 hidden columns ("item"): GEMM 1 = 96 v_mfma_f32_16x16x32_bf16 onto a 32 x 64 tile, epilogue in registers (bias + GELU,
//     or .* gelu'(u)), the bf16 result IS the second operand of GEMM 2's next two k steps (the column interleave that makes the
//     stores 64-byte runs is also the MFMA operand layout: lane (token t, quarter q) holds 8 consecutive hidden columns);
//   * GEMM 2 = 96 MFMAs accumulating the 32 x 384 output tile in 192 accumulator registers over all items;
//   * both weight matrices stream through a ring of three 48 KiB LDS stages by LDS-DMA (W1: 64 rows x 768 B; W2: 384 rows x 128 B,
//     full cache lines both), two stages in flight, counted s_waitcnt vmcnt, one barrier per 96 MFMAs;
//   * stores trickle out per item (8 x 1 KiB per wave: 16 rows x 64 contiguous bytes each, pairs completing 128-byte lines).
// One workgroup (4 waves, 128 rows) per CU, one workgroup per 128-row unit; the hardware fills freed CUs from the launches of the
// other streams (row chains, teacher pass, weight gradients).
#include <stdlib.h>
#include "common.hpp"
#include "lafs_hip.h"

// (the C ABI this kernel had while it was in the library)
typedef struct lafs_mlp_args {
  const void* x; int ldx;
  const void* w1; int ldw1; const float* b1;
  const void* w2; int ldw2; const float* b2;
  void* save_dgelu; void* save_act; int lds;
  const float* resid; int ldr;
  const float* seq_scale; const int32_t* row2seq;
  void* out; int ldo;
  int M, D, H;
} lafs_mlp_args;

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

constexpr int DD = 384;                    // embedding width: reduction of GEMM 1, output width of GEMM 2 (compile-time: register tiles)
constexpr int NTH = 256;
constexpr int ROWB = DD * 2;               // bytes per row of the resident operand / of W1
constexpr int CPR = DD / 8;                // 16-byte chunks per such row (48)
constexpr int SUB = 32 * ROWB;             // 24 KiB: 32 rows of W1 = one k-step-pair of hidden columns
constexpr int STAGE = 2 * SUB;             // 48 KiB: 64 rows of W1  ==  384 rows x 128 B of W2
constexpr int NBUF = 3;
constexpr int NDMA = STAGE / 16 / NTH;     // LDS-DMA instructions per thread and stage (12)
constexpr int NKK = DD / 32;               // k steps of GEMM 1 (12)
constexpr int NJ = DD / 16;                // 16-column output groups of GEMM 2 (24)
constexpr int MAXH = 2048;
constexpr int UROWS = 128;
constexpr int FD = 8;                      // fragment reads in flight ahead of their MFMAs
static_assert(NDMA * NTH * 16 == STAGE && (SUB / 16) % NTH == 0, "stage layout");

struct MArgs {
  const bf16_t* A; int lda;                // [M, 384] resident operand: LN2 output (forward) / upstream gradient (backward)
  const bf16_t* W1; int ldw1;              // [H, 384]: rows = hidden units          (forward: fc1.weight; backward: fc2.weight^T)
  const bf16_t* W2; int ldw2;              // [384, H]: rows = output columns        (forward: fc2.weight; backward: fc1.weight^T)
  const float* b1; const float* b2;        // forward biases ([H], [384]); may be null
  const bf16_t* aux; int ldaux;            // backward: gelu'(u) bf16 [M, H]
  bf16_t* S1; int lds1;                    // forward: gelu'(u) out (or null);  backward: du out
  bf16_t* S2; int lds2;                    // forward: gelu(u) out (or null)
  const float* resid; int ldr;             // forward: fp32 [M, 384]
  const float* seq_scale; const int* row2seq;
  void* C; int ldc;                        // forward: fp32 [M, 384];  backward: bf16 [M, 384]
  int M, H;
};

__device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void st16(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
  const u32x4_t v = {a, b, c, d};
  *reinterpret_cast<u32x4_t*>(p) = v;
}
__device__ __forceinline__ void st16f(void* p, float a, float b, float c, float d) {
  const f32x4v_t v = {a, b, c, d};
  *reinterpret_cast<f32x4v_t*>(p) = v;
}
// 16-byte global load the compiler's wait-count pass cannot see (it would drain the LDS-DMA ring with a vmcnt(0) in front of the
// first use): the kernel counts it itself; `landed` orders the uses behind the explicit wait
__device__ __forceinline__ void ld16_untracked(u32x4_t& dst, const void* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void landed(u32x4_t& v) { asm volatile("" : "+v"(v)); }
// s_waitcnt vmcnt(n) for a wave-uniform runtime n (multiples of 2 up to 40; anything else waits for everything)
__device__ __forceinline__ void wait_vm_rt(int n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    W_(2) W_(4) W_(6) W_(8) W_(10) W_(12) W_(14) W_(16) W_(18) W_(20) W_(22) W_(24) W_(26) W_(28) W_(30) W_(32) W_(34) W_(36) W_(38) W_(40)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef W_
}

// MODE 0 = forward (SAVE: also store gelu'(u), gelu(u)), MODE 1 = backward (du always stored)
template <int MODE, bool SAVE>
__global__ __launch_bounds__(NTH, 1) void mlp_fused_kernel(MArgs p) {
  constexpr bool FWD = (MODE == 0);
  constexpr int SPB = FWD ? (SAVE ? 4 : 0) : 2;        // store instructions per item and valid 16-row token block
  constexpr int P = FWD ? 0 : 4;                       // epilogue-operand loads per item (gelu'(u): 2 token blocks x 2 k steps)
  __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * STAGE];
  __shared__ __attribute__((aligned(16))) float sb1[MAXH];
  __shared__ __attribute__((aligned(16))) float sb2[DD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15, q = lane >> 4;
  const int unit = blockIdx.x;
  const int NI = p.H >> 6;                             // items: 64 hidden columns each
  const int NS = 2 * NI;                               // ring stages: W1(0), W2(0), W1(1), W2(1), ...

  // everything loaded the ordinary way comes first: memory operations complete in order, so a plain load issued behind an LDS-DMA
  // request could only be consumed after draining the ring
  float sc[2] = {1.0f, 1.0f};
  if (FWD) {
    for (int i = tid; i < p.H; i += NTH) sb1[i] = p.b1 ? p.b1[i] : 0.f;
    for (int i = tid; i < DD; i += NTH) sb2[i] = p.b2 ? p.b2[i] : 0.f;
    if (p.seq_scale != nullptr) {
#pragma unroll
      for (int b = 0; b < 2; ++b) sc[b] = p.seq_scale[p.row2seq[min((int)blockIdx.x * UROWS + (tid >> 6) * 32 + (tid & 15) + 16 * b, p.M - 1)]];
    }
  }
  __syncthreads();                                     // (also keeps these loads out of the counted waits below)

  // ---- LDS-DMA source offsets (bytes).  Piece x = r * 256 + tid of a stage lands at LDS byte 16 x (lane-linear), so every layout
  // choice is made on the SOURCE address.
  // W1 stage: two 32-row halves g (one per k step of GEMM 2); half row rho = 48 chunks, logical chunk c at position c ^ (rho & 15)
  // (conflict-free ds_read_b128 of 16 rows x one chunk); MFMA row s of 16-row group gi is W1 row 8 (s >> 2) + 4 gi + (s & 3) of the
  // half, so that a lane ends with 8 consecutive hidden columns (= one 16-byte store, = one MFMA operand of GEMM 2).
  int doff1[NDMA];
#pragma unroll
  for (int r = 0; r < NDMA; ++r) {
    const int x = r * NTH + tid, g = x / (SUB / 16), y = x % (SUB / 16);
    const int rho = y / CPR, cp = y % CPR, c = cp ^ (rho & 15);
    const int s16 = rho & 15, gi = rho >> 4;
    const int rowrel = 32 * g + 8 * (s16 >> 2) + 4 * gi + (s16 & 3);
    doff1[r] = (rowrel * p.ldw1 + c * 8) * 2;
  }
  // W2 stage: LDS row r2 (0..383) = 128 B = the item's 64 hidden columns of one output column; chunk c at position c ^ (r2 & 7).
  // Forward (fp32 output, 4 consecutive columns per MFMA = 16 bytes): LDS row = output column.  Backward (bf16 output): rows of a
  // pair of 16-row groups interleaved as above, so that two MFMAs give a lane 8 consecutive columns.
  int doff2[NDMA];
#pragma unroll
  for (int r = 0; r < NDMA; ++r) {
    const int x = r * NTH + tid, r2 = x >> 3, pos = x & 7, c = pos ^ (r2 & 7);
    const int j = r2 >> 4, s16 = r2 & 15;
    const int col = FWD ? r2 : (32 * (j >> 1) + 8 * (s16 >> 2) + 4 * (j & 1) + (s16 & 3));
    doff2[r] = (col * p.ldw2 + c * 8) * 2;
  }
  const unsigned lds0 = lds_addr_of(smem);
  // stage s lives in ring buffer (s + 2) % 3: stage 0 can be requested while the resident rows still pass through buffers 0 and 1
  auto issue = [&](int s) {
    const int i = s >> 1;
    const unsigned st = lds0 + ((s + 2) % NBUF) * STAGE + wave * 1024;
    fence();
    if ((s & 1) == 0) {
      const bf16_t* base = p.W1 + (size_t)(64 * i) * p.ldw1;
#pragma unroll
      for (int r = 0; r < NDMA; ++r) lds_dma16_m0_s(base, (unsigned)doff1[r], st + r * (NTH * 16));
    } else {
      const bf16_t* base = p.W2 + 64 * i;
#pragma unroll
      for (int r = 0; r < NDMA; ++r) lds_dma16_m0_s(base, (unsigned)doff2[r], st + r * (NTH * 16));
    }
    fence();
  };

  // fragment read offsets.  GEMM 1 (as gemm_kres.hip): row 16 gi + t, chunk (4 kk + q) ^ t;  GEMM 2: row 16 j + t, chunk (4 g + q) ^ (t & 7)
  int foff[4], f2off[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    foff[i] = t * ROWB + (((4 * i + q) ^ t) << 4);
    asm volatile("" : "+v"(foff[i]));
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    f2off[g] = t * 128 + (((4 * g + q) ^ (t & 7)) << 4);
    asm volatile("" : "+v"(f2off[g]));
  }

  // ---- the resident rows: 4 x 32 rows as four 24 KiB sub-stages through buffers 0 and 1 (whole 768-byte rows by LDS-DMA), then
  // each wave reads its 32 rows into registers with the GEMM-1 fragment addressing
  const int row0 = unit * UROWS + wave * 32;           // first token row of this wave
  const int m0 = row0 + t;
  const bool blk_ok[2] = {row0 < p.M, row0 + 16 < p.M};
  const int nst = SPB * ((blk_ok[0] ? 1 : 0) + (blk_ok[1] ? 1 : 0));
  fence();
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int r = 0; r < SUB / 16 / NTH; ++r) {
      const int y = r * NTH + tid, rho = y / CPR, cp = y % CPR, c = cp ^ (rho & 15);
      const int row = min(unit * UROWS + w * 32 + rho, p.M - 1);
      lds_dma16_m0(p.A + (size_t)row * p.lda + c * 8, lds0 + w * SUB + r * (NTH * 16) + wave * 1024);
    }
  fence();
  issue(0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");           // the rows have landed (stage 0 is younger)
  __builtin_amdgcn_s_barrier();
  bf16x8_t areg[2][NKK];                               // two 16-token blocks x 12 k steps: lane (t, q) holds k = 32 kk + 8 q .. + 7
  {
    const unsigned char* st = smem + wave * SUB;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk)
        areg[b][kk] = *reinterpret_cast<const bf16x8_t*>(st + foff[kk & 3] + (kk >> 2) * 256 + b * (16 * ROWB));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();                        // buffers 0 and 1 are free again

  u32x4_t aux[P > 0 ? P : 1];
  auto fetch_aux = [&](int i) {                        // exactly P loads: gelu'(u) of item i, this lane's 2 x 2 x 8 columns
    if (P == 0) return;
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const int b = k & 1, g = k >> 1;
      const int mr = min(m0 + 16 * b, p.M - 1);
      ld16_untracked(aux[k], p.aux + (size_t)mr * p.ldaux + 64 * i + 32 * g + 8 * q);
    }
  };
  fetch_aux(0);
  issue(1);

  f32x4_t acc2[NJ][2];                                 // [output column group][token block]: the 32 x 384 output tile
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (FWD) acc2[j][b] = *reinterpret_cast<const f32x4_t*>(sb2 + 16 * j + 4 * q);
      else acc2[j][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

  for (int i = 0; i < NI; ++i) {
    const bool more = (i + 1 < NI);
    const int n0 = 64 * i;
    // ================= GEMM 1 on stage 2 i: younger than its DMA are the previous item's stores, this item's epilogue operand
    // and the request of stage 2 i + 1
    wait_vm_rt((i > 0 ? nst : 0) + P + NDMA);
    __builtin_amdgcn_s_barrier();
    if (more) issue(2 * i + 2);
    f32x4_t acc1[2][2][2];                             // [k step of GEMM 2 = half stage][weight row group][token block]
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        const int n = n0 + 32 * g + 8 * q + 4 * gi;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (FWD) acc1[g][gi][b] = *reinterpret_cast<const f32x4_t*>(sb1 + n);
          else acc1[g][gi][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
      }
    {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* st = smem + ((2 * i + 2) % NBUF) * STAGE;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
          for (int gi = 0; gi < 2; ++gi) {
            const bf16x8_t w = *reinterpret_cast<const bf16x8_t*>(st + g * SUB + foff[kk & 3] + (kk >> 2) * 256 + gi * (16 * ROWB));
            acc1[g][gi][0] = mfma16(w, areg[0][kk], acc1[g][gi][0]);
            acc1[g][gi][1] = mfma16(w, areg[1][kk], acc1[g][gi][1]);
          }
      // fragment reads FD ahead of the MFMA pairs that consume them (left alone, hipcc keeps one read in flight)
      __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);
#pragma unroll
      for (int k = 0; k < 4 * NKK - FD; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * FD, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ================= epilogue of GEMM 1: lane (t, q) owns rows m0, m0 + 16 and, per row and half stage, 8 consecutive columns
    if (P > 0) {                                        // gelu'(u) of this item has arrived (younger: the two stage requests)
      wait_vm_rt(NDMA + (more ? NDMA : 0));
#pragma unroll
      for (int k = 0; k < P; ++k) landed(aux[k]);
    }
    bf16x8_t afrag[2][2];                               // [token block][k step]: second operand of GEMM 2
    fence();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = m0 + 16 * b;
      const bool rowok = m < p.M;
      unsigned pk[2][4], pd[2][4];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float v[8];
        v[0] = acc1[g][0][b][0]; v[1] = acc1[g][0][b][1]; v[2] = acc1[g][0][b][2]; v[3] = acc1[g][0][b][3];
        v[4] = acc1[g][1][b][0]; v[5] = acc1[g][1][b][1]; v[6] = acc1[g][1][b][2]; v[7] = acc1[g][1][b][3];
        if (FWD) {
          float dv[8];
          if (SAVE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float gv; gelu_both_f(v[e], gv, dv[e]); v[e] = gv; }
#pragma unroll
            for (int e = 0; e < 4; ++e) pd[g][e] = pack_bf2(dv[2 * e], dv[2 * e + 1]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
          }
        } else {
          const u32x4_t a4 = aux[g * 2 + b];
          v[0] *= bf_lo(a4[0]); v[1] *= bf_hi(a4[0]); v[2] *= bf_lo(a4[1]); v[3] *= bf_hi(a4[1]);
          v[4] *= bf_lo(a4[2]); v[5] *= bf_hi(a4[2]); v[6] *= bf_lo(a4[3]); v[7] *= bf_hi(a4[3]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[g][e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        const u32x4_t pv = {pk[g][0], pk[g][1], pk[g][2], pk[g][3]};
        afrag[b][g] = __builtin_bit_cast(bf16x8_t, pv);
      }
      if (SPB > 0 && blk_ok[b]) {                       // wave-uniform: exactly SPB store instructions per valid block
        if (FWD && SAVE) {
#pragma unroll
          for (int g = 0; g < 2; ++g)
            if (rowok) st16(p.S1 + (size_t)m * p.lds1 + n0 + 32 * g + 8 * q, pd[g][0], pd[g][1], pd[g][2], pd[g][3]);
#pragma unroll
          for (int g = 0; g < 2; ++g)
            if (rowok) st16(p.S2 + (size_t)m * p.lds2 + n0 + 32 * g + 8 * q, pk[g][0], pk[g][1], pk[g][2], pk[g][3]);
        } else if (!FWD) {
#pragma unroll
          for (int g = 0; g < 2; ++g)
            if (rowok) st16(p.S1 + (size_t)m * p.lds1 + n0 + 32 * g + 8 * q, pk[g][0], pk[g][1], pk[g][2], pk[g][3]);
        }
      }
    }
    fence();
    // ================= GEMM 2 on stage 2 i + 1: younger than its DMA are the request of stage 2 i + 2 and this item's stores
    wait_vm_rt((more ? NDMA : 0) + nst);
    __builtin_amdgcn_s_barrier();
    if (more) { fetch_aux(i + 1); issue(2 * i + 3); }
    {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* st = smem + ((2 * i + 3) % NBUF) * STAGE;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const bf16x8_t w = *reinterpret_cast<const bf16x8_t*>(st + f2off[g] + j * 2048);
          acc2[j][0] = mfma16(w, afrag[0][g], acc2[j][0]);
          acc2[j][1] = mfma16(w, afrag[1][g], acc2[j][1]);
        }
      __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);
#pragma unroll
      for (int k = 0; k < 2 * NJ - FD; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * FD, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ================= final epilogue: the 32 x 384 output tile.  Lane (t, q): rows m0, m0 + 16; per column group j 4 consecutive
  // columns (forward, fp32: one 16-byte store; a store instruction writes 16 rows x 64 contiguous bytes) or, with the row interleave
  // of the backward, 8 consecutive bf16 columns per PAIR of groups.
  fence();
  if (FWD) {
    constexpr int GJ = 8;                              // column groups per batch of residual loads (2 x 8 x 16 B in flight per lane)
    uint4 rbuf[2][GJ][2];
    auto load_batch = [&](int k, uint4 (&dst)[GJ][2]) {
#pragma unroll
      for (int jj = 0; jj < GJ; ++jj)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int mr = min(m0 + 16 * b, p.M - 1);
          dst[jj][b] = *reinterpret_cast<const uint4*>(p.resid + (size_t)mr * p.ldr + 16 * (k * GJ + jj) + 4 * q);
        }
    };
    auto store_batch = [&](int k, const uint4 (&src)[GJ][2]) {
#pragma unroll
      for (int jj = 0; jj < GJ; ++jj)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int j = k * GJ + jj, m = m0 + 16 * b;
          const uint4 r4 = src[jj][b];
          const f32x4_t a = acc2[j][b];
          if (m < p.M)
            st16f(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + 16 * j + 4 * q, __uint_as_float(r4.x) + sc[b] * a[0],
                  __uint_as_float(r4.y) + sc[b] * a[1], __uint_as_float(r4.z) + sc[b] * a[2], __uint_as_float(r4.w) + sc[b] * a[3]);
        }
    };
    load_batch(0, rbuf[0]);
    load_batch(1, rbuf[1]);
    store_batch(0, rbuf[0]);
    load_batch(2, rbuf[0]);
    store_batch(1, rbuf[1]);
    store_batch(2, rbuf[0]);
  } else {
#pragma unroll
    for (int jj = 0; jj < NJ / 2; ++jj)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int m = m0 + 16 * b;
        const f32x4_t a0 = acc2[2 * jj][b], a1 = acc2[2 * jj + 1][b];
        if (m < p.M)
          st16(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + 32 * jj + 8 * q, pack_bf2(a0[0], a0[1]), pack_bf2(a0[2], a0[3]),
               pack_bf2(a1[0], a1[1]), pack_bf2(a1[2], a1[3]));
      }
  }
}

template <int MODE, bool SAVE>
int launch(const MArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((mlp_fused_kernel<MODE, SAVE>), dim3((a.M + UROWS - 1) / UROWS), dim3(NTH), 0, s, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

}  // namespace

extern "C" int lafs_mlp_fused_eligible(int M, int D, int H) {
  static const bool on = [] { const char* v = getenv("LAFS_MLP_FUSED"); return v == nullptr || v[0] != '0'; }();   // A/B knob
  return (on && D == DD && H >= 64 && H % 64 == 0 && H <= MAXH && M >= 2048) ? 1 : 0;
}

extern "C" int lafs_mlp_fwd(const lafs_mlp_args* g, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(g != nullptr && g->x && g->w1 && g->w2 && g->resid && g->out, "null operand");
  LAFS_CHECK_ARG(g->D == DD && g->H >= 64 && g->H % 64 == 0 && g->H <= MAXH && g->M > 0, "fused MLP: D = 384, H a multiple of 64 up to 2048");
  LAFS_CHECK_ARG((g->save_dgelu == nullptr) == (g->save_act == nullptr), "gelu'(u) and gelu(u) are saved together or not at all");
  LAFS_CHECK_ARG(g->ldx % 8 == 0 && g->ldw1 % 8 == 0 && g->ldw2 % 8 == 0 && g->ldr % 4 == 0 && g->ldo % 4 == 0 && g->lds % 8 == 0,
                 "row strides must keep 16-byte alignment");
  LAFS_CHECK_ARG(g->seq_scale == nullptr || g->row2seq != nullptr, "seq_scale needs row2seq");
  MArgs a = {};
  a.A = (const bf16_t*)g->x; a.lda = g->ldx; a.W1 = (const bf16_t*)g->w1; a.ldw1 = g->ldw1; a.W2 = (const bf16_t*)g->w2; a.ldw2 = g->ldw2;
  a.b1 = g->b1; a.b2 = g->b2; a.S1 = (bf16_t*)g->save_dgelu; a.lds1 = g->lds; a.S2 = (bf16_t*)g->save_act; a.lds2 = g->lds;
  a.resid = g->resid; a.ldr = g->ldr; a.seq_scale = g->seq_scale; a.row2seq = g->row2seq; a.C = g->out; a.ldc = g->ldo;
  a.M = g->M; a.H = g->H;
  return g->save_act != nullptr ? launch<0, true>(a, stream) : launch<0, false>(a, stream);
}

extern "C" int lafs_mlp_bwd(const lafs_mlp_args* g, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(g != nullptr && g->x && g->w1 && g->w2 && g->save_dgelu && g->save_act && g->out, "null operand");
  LAFS_CHECK_ARG(g->D == DD && g->H >= 64 && g->H % 64 == 0 && g->H <= MAXH && g->M > 0, "fused MLP: D = 384, H a multiple of 64 up to 2048");
  LAFS_CHECK_ARG(g->ldx % 8 == 0 && g->ldw1 % 8 == 0 && g->ldw2 % 8 == 0 && g->ldo % 8 == 0 && g->lds % 8 == 0,
                 "row strides must keep 16-byte alignment");
  MArgs a = {};
  a.A = (const bf16_t*)g->x; a.lda = g->ldx; a.W1 = (const bf16_t*)g->w1; a.ldw1 = g->ldw1; a.W2 = (const bf16_t*)g->w2; a.ldw2 = g->ldw2;
  a.aux = (const bf16_t*)g->save_dgelu; a.ldaux = g->lds; a.S1 = (bf16_t*)g->save_act; a.lds1 = g->lds;
  a.C = g->out; a.ldc = g->ldo; a.M = g->M; a.H = g->H;
  return launch<1, true>(a, stream);
}
