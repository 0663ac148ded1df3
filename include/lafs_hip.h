/* lafs_hip.h -- C ABI of liblafs_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the LAFS
 * data-parallel hot path.
 *
 * The reference (szlbiubiubiu/LAFS_CVPR2024) has no FFI of its own: the hot path sits behind
 * torch.nn.Module objects that dispatch to ATen/cuBLAS/cuDNN.  Each entry point below replaces the
 * implicit device kernels behind one reference call site (cited as file:line, relative to the reference
 * root).  Conventions:
 *   - plain pointers and sizes only (device pointers unless stated), no torch types;
 *   - the CALLER allocates every buffer including workspaces; the library keeps no process-wide device state and reads no
 *     environment variable: the few device objects it needs itself (side streams and events of the trunk passes) and its
 *     kernel-selection options live in a caller-owned per-device context, lafs_ctx (below); calls that share no context and no
 *     buffer are independent (re-entrant per stream);
 *   - every function enqueues work on `stream` and never synchronises the device;
 *   - return value: 0 = success, < 0 = bad argument (see lafs_last_error()), > 0 = hipError_t;
 *   - "bf16" = raw 16-bit bfloat16, "f32" = IEEE float; row-major everywhere, ld* in ELEMENTS.
 */
#ifndef LAFS_HIP_H
#define LAFS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define LAFS_ABI_VERSION 1

int lafs_version(void);
/* Diagnostic: lane l of one wave issues ds_read_b64_tr_b16 at LDS byte 8*l over in(i16)[512]; out(i16)[256] gets
 * the 4 values each lane received.  Pins the LDS-transpose-read model used by the attention / wgrad kernels. */
int lafs_debug_tr16(const void* in, void* out, hipStream_t stream);
/* Kernel-selection flags for the timing experiments of tools/bench_kernels.py; never set on the product
 * path, and bench.py refuses to run unless lafs_debug_get() == 0.  In liblafs_hip.so every flag only selects an alternative,
 * equally correct code path.  The flags marked (W) make results WRONG (they skip work) and exist only in the -DLAFS_ABLATE
 * build (`make -C lafs_cvpr2024_amd/csrc ablate` -> liblafs_hip_ablate.so, lafs_ablation_build() == 1); the product library
 * compiles them out.
 *   1 (W) TN GEMM: plain stores instead of atomics      2 / 4  NT GEMM: force 128x128 / 256x128 tiles     8  NT: force 64-deep stages
 *  16 (W) NT: no epilogue stores   32 (W) NT: no MFMA   64 (W) NT GELU: no second store   128 (W) NT GELU: store u twice
 * 256     NT: non-temporal epilogue stores       1024  TN: force 128x128     2048 / 4096  TN: 256x256 / 256x128 tiles
 * 8192    TN: 64-row stages in a 2-stage ring   16384 (W) NT GELU: u / GELU(u) interleaved in one buffer
 * 32768   NT GELU: piece-by-piece store order (the pre-optimisation order)      65536  NT: 256x256 tiles for bf16 epilogues */
int lafs_debug_set(int flags);
int lafs_debug_get(void);
int lafs_ablation_build(void);
/* Thread-local text of the last error returned on this thread ("" if none). */
const char* lafs_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Per-device context.  The reference has no counterpart: its modules issue every op on torch's current stream from one Python
 * thread per rank (lafs_train.py:513-613).  This library forks a trunk pass over side streams of its own (row chains, weight
 * gradients), so it needs streams / events that live across calls; they belong to a handle the CALLER creates and destroys.
 *   lafs_ctx_create(device)  device < 0 = the current one.  Creates three non-blocking side streams, their fork / join events and a
 *                            pool of 64 events now (never inside a hipGraph capture); NULL on failure (lafs_last_error()).
 *   lafs_ctx_destroy         after every graph that captured work of this context has been destroyed.
 *   lafs_ctx_set / _get      kernel-selection options (defaults = the measured-best settings; the others exist for A/B runs).
 * A NULL context is legal wherever one is accepted: compiled-in default options, no side streams (everything on `stream`).
 * Two contexts share nothing: engines built on different contexts may be driven from different host threads and streams. */
typedef struct lafs_ctx lafs_ctx;
enum {
  LAFS_OPT_SIDE_STREAMS = 0,   /* 1: row chains / attention groups / weight gradients may use the context's side streams; 0: caller's stream only */
  LAFS_OPT_ROW_CHAINS = 1,     /* 1, 2 (default) or 4 chains of launches over the crop-resolution groups' rows (csrc/engine.hip) */
  LAFS_OPT_KRES_MASK = 2,      /* epilogues routed to the K-resident GEMM: 1 plain, 2 GELU, 4 residual, 8 GELU' (default 15) */
  LAFS_OPT_KRES_MIN_ITEMS = 3, /* least items per workgroup run of the K-resident GEMM (default 4) */
  LAFS_OPT_NT_WIDE = 4,        /* tiled GEMM: 128x384 tiles where they fit one round of the chip (default 1) */
  LAFS_OPT_NT_TALL = 5,        /* tiled GEMM: 160-row tiles where they save a round of workgroup slots (default 1) */
  LAFS_OPT_COMM_CUS = 6,       /* data-parallel runs: CUs left to the collective library's kernels by the K-resident GEMM (default 0) */
  LAFS_OPT_NT_BIG = 7,         /* tiled GEMM: one-workgroup-per-CU 192x256 / 176x256 tiles for the wide long-K shapes (default 1; 2-4 force a geometry) */
  LAFS_OPT_MLP_FUSED = 8,      /* trunk passes: the block's MLP as ONE launch (lafs_mlp_fused) where it applies (dim 384, no element dropout): bit mask
                                  1 forward-only pass, 2 saving forward, 4 backward input gradients, 8 LayerNorm 2 inside the fused forward, 16 its backward inside the fused backward,
                                  32 (lab) one MLP launch over all row chains, 64 the NEXT block's LayerNorm 1 in the fused forward's epilogue (bit-identical
                                  to its launch), 128 / 256 the attention branch's output projection + residual in FRONT of the fused forward of the
                                  forward-only / the saving pass.  Default 79 = 1 + 2 + 4 + 8 + 64: step A/B in DESIGN.md section 6 */
  LAFS_OPT_COUNT = 9
};
lafs_ctx* lafs_ctx_create(int device);
void lafs_ctx_destroy(lafs_ctx* ctx);
int lafs_ctx_set(lafs_ctx* ctx, int opt, int value);
int lafs_ctx_get(const lafs_ctx* ctx, int opt);

/* ------------------------------------------------------------------------------------------------
 * GEMMs (nn.Linear / nn.Conv2d(k=s=p) forward, dgrad, wgrad)
 *   vision_transformer.py:59-65 (Mlp), :75-90 (qkv/proj), :126-131 (PatchEmbed conv), :295-301 (DINOHead)
 *   face_pre_pro/ViT_face.py:126-137 (FeedForward), :147-149 (to_qkv/to_out), :761 (patch_to_embedding)
 * ------------------------------------------------------------------------------------------------ */
enum {
  LAFS_EPI_BF16 = 0,       /* C(bf16) = acc + bias                                                   */
  LAFS_EPI_BF16_GELU = 1,  /* C(bf16) = u = acc + bias (skipped when C is NULL) ; C2(bf16) = GELU_erf(u) */
  LAFS_EPI_RESID_F32 = 2,  /* C(f32) = resid + seq_scale[row2seq[m]] * (acc + bias)   (DropPath)     */
  LAFS_EPI_F32 = 3,        /* C(f32) = acc + bias                                                    */
  LAFS_EPI_DGELU_BF16 = 4, /* C(bf16) = acc * GELU'(aux[m][n])                                       */
  LAFS_EPI_ATOMIC_F32 = 5, /* C(f32) += acc, K split into `splits` slices (C must be pre-zeroed)     */
                           /* (LAFS_EPI_F32 with splits > 1: C is [splits][M][ldc], one plain-store image per K slice, no bias;
                            *  lafs_sum_slices adds them -- no atomics, deterministic) */
  LAFS_EPI_EMBED_F32 = 6,  /* C(f32)[m + m/npatch + 1] = acc + bias + pos[1 + m%npatch]  (tokens)    */
  LAFS_EPI_BF16_ACT = 7    /* C(bf16) = act(acc + bias + aux[m][n])   (aux bf16 residual or NULL)    */
};

enum { LAFS_ACT_NONE = 0, LAFS_ACT_RELU = 1, LAFS_ACT_HSWISH = 2, LAFS_ACT_HSIGMOID = 3 };

typedef struct lafs_gemm_nt_args {
  const void* A; int lda;          /* bf16 [M, K]                                  */
  const void* B; int ldb;          /* bf16 [N, K]  (nn.Linear weight layout)       */
  int M, N, K;                     /* K % 32 == 0                                  */
  int epilogue;                    /* LAFS_EPI_*                                   */
  void* C; int ldc;                /* bf16 or f32 [M, N]                           */
  void* C2; int ldc2;              /* second output (BF16_GELU)                    */
  const float* bias;               /* f32 [N] or NULL                              */
  const float* resid; int ldr;     /* f32 [M, N] (RESID_F32; may alias C)          */
  const float* seq_scale;          /* f32 [n_seq] or NULL (RESID_F32)              */
  const int32_t* row2seq;          /* i32 [M]  row -> sequence index               */
  const void* aux; int ldaux;      /* bf16 [M, N] pre-activation (DGELU_BF16)      */
  const float* pos; int npatch;    /* f32 [npatch+1, N] (EMBED_F32)                */
  int splits;                      /* ATOMIC_F32 / F32: number of K slices (>=1)   */
  float drop_p; uint32_t drop_seed; /* element dropout (0 = off): on the linear's output before the residual add
                                      (RESID_F32), on GELU(u) (BF16_GELU: C2 only), and its backward (DGELU_BF16).
                                      Counter-based mask of (drop_seed, row, col): see lafs_debug_dropout_mask.       */
  const float* drop_step;          /* NULL, or DEVICE pointer to the step counter: the mask's seed is drop_seed + 7919 * step, read
                                      when the kernel runs -- a captured hipGraph then draws a new mask on every replay            */
  int drop_row0;                   /* this launch's row 0 is row drop_row0 of the mask (a launch over a row sub-range of a batch) */
  int act;                         /* BF16_ACT: LAFS_ACT_*.  BF16_GELU / DGELU_BF16: LAFS_GELU_SAVE_GRAD (see below) */
  int operand_f16;                 /* 1: A, B, a 16-bit C and the BF16_ACT residual (aux) are IEEE fp16 instead of bf16 (BF16 / BF16_ACT /
                                      F32 epilogues, no K split): the trainable landmark CNN runs in the reference's autocast
                                      format (train_largescale.py:803-804; v_mfma_f32_16x16x32_f16, same rate) */
  const lafs_ctx* ctx;             /* kernel-selection options (NULL = defaults) */
} lafs_gemm_nt_args;

/* act = LAFS_GELU_SAVE_GRAD with LAFS_EPI_BF16_GELU: C receives gelu'(u) (bf16) instead of the pre-activation u; with
 * LAFS_EPI_DGELU_BF16: aux holds that gelu'(u) and is multiplied in as it is.  The backward of Mlp (vision_transformer.py:59-65)
 * needs u only through gelu'(u): saved this way, the derivative's exponential / reciprocal are the forward's (shared with
 * gelu(u)) and the GELU' epilogue of the input gradient is one multiply per value. */
#define LAFS_GELU_SAVE_GRAD 1
/* C[M,N] = A[M,K] * B[N,K]^T with a fused epilogue. */
/* Number of K slices a request for `splits` actually produces (slices are whole pipeline stages): the image count of a
 * K-split LAFS_EPI_F32 GEMM. */
int lafs_gemm_nt_slices(int K, int splits);
int lafs_gemm_nt(const lafs_gemm_nt_args* args, hipStream_t stream);
/* Which kernel lafs_gemm_nt runs for this request: 0 = the tiled LDS-DMA kernel (gemm.hip), 1 = the K-resident streaming kernel
 * (gemm_kres.hip: K == 384, N % 64 == 0, N <= 1536, M >= 2048, plain / GELU / GELU' / residual epilogue, no dropout, bf16 operands),
 * 3 = the tiled kernel in its 128x384 / 12-wave form (whole N per workgroup: long reductions onto N = 384 whose tiles fit one
 * round of the chip), 4 = the tiled kernel with 160-row tiles (long reductions whose 128-row tiles would spill into one more
 * round of the 512 workgroup slots than 160-row ones need), 5 = the 256x256 one-workgroup-per-CU kernel (gemm_big.hip).
 * Mirrors lafs_gemm_nt's own decisions, including its operand-format and validation order. */
int lafs_gemm_nt_route(const lafs_gemm_nt_args* args);

/* ------------------------------------------------------------------------------------------------
 * Fused MLP of a ViT-S block (csrc/mlp_fused.hip): two chained GEMMs, the hidden-wide intermediate never leaves the chip.
 *   vision_transformer.py:59-65 (Mlp.forward: fc1 -> GELU -> fc2) with the residual + DropPath of Block.forward (:112), and the
 *   input-gradient chain of the same lines in the backward.  Embedding width 384 (compile time), hidden width H % 64 == 0,
 *   128 <= H <= 1536.  Results are bit-identical to the two lafs_gemm_nt launches each mode replaces.
 *   LAFS_MLP_FWD       out(f32)[M,384] = resid + seq_scale[row2seq[m]] * (gelu(X Wa^T + bias_a) Wb^T + bias_b)
 *                      X = LayerNorm-2 output (bf16), Wa = fc1.weight [H,384], Wb = fc2.weight [384,H] (bf16 shadows);
 *                      replaces lafs_gemm_nt(LAFS_EPI_BF16_GELU, C == NULL) + lafs_gemm_nt(LAFS_EPI_RESID_F32)
 *   LAFS_MLP_FWD_SAVE  the same, and save_grad(bf16)[M,H] = gelu'(u), save_act(bf16)[M,H] = gelu(u) are written (what the backward
 *                      and the fc2 weight gradient read): replaces (LAFS_EPI_BF16_GELU, act = LAFS_GELU_SAVE_GRAD) + RESID_F32
 *   LAFS_MLP_BWD       save_act(bf16)[M,H] = du = (X Wa^T) * save_grad  (written: the fc1 weight gradient's operand),
 *                      out(bf16)[M,384] = du Wb^T;  X = upstream gradient (bf16), Wa = fc2.weight^T shadow [H,384],
 *                      Wb = fc1.weight^T shadow [384,H]: replaces (LAFS_EPI_DGELU_BF16, act = LAFS_GELU_SAVE_GRAD) + LAFS_EPI_BF16
 * One 8-wave workgroup per 128 rows (rows past the last full round of the chip: 64-row units); caller-allocated buffers only;
 * never synchronises. */
enum { LAFS_MLP_FWD = 0, LAFS_MLP_FWD_SAVE = 1, LAFS_MLP_BWD = 2 };
typedef struct lafs_mlp_args {
  const void* X; int ldx;          /* bf16 [M, 384]                                       */
  const void* Wa; int ldwa;        /* bf16 [H, 384]                                       */
  const void* Wb; int ldwb;        /* bf16 [384, H]                                       */
  int M, H;
  int mode;                        /* LAFS_MLP_*                                          */
  const float* bias_a;             /* f32 [H]   or NULL (forward modes)                   */
  const float* bias_b;             /* f32 [384] or NULL (forward modes)                   */
  const float* resid; int ldr;     /* f32 [M, 384] (forward modes; may alias out)         */
  const float* seq_scale;          /* f32 [n_seq] or NULL (forward modes: DropPath)       */
  const int32_t* row2seq;          /* i32 [M]                                             */
  void* out; int ldo;              /* f32 [M, 384] (forward) / bf16 [M, 384] (backward)   */
  void* save_grad; int ldsg;       /* bf16 [M, H]: gelu'(u), written by FWD_SAVE, read by BWD */
  void* save_act; int ldsa;        /* bf16 [M, H]: FWD_SAVE writes gelu(u); BWD writes du */
  const lafs_ctx* ctx;             /* rounds of the chip are sized by the context's CU count (NULL: one launch of 128-row units) */
  /* LayerNorm prologue (forward modes; vision_transformer.py:112 `self.norm2`): with ln_gamma != NULL, X is ignored and the GEMM-1
   * operand is LayerNorm(resid; ln_gamma, ln_beta, ln_eps) computed inside the kernel, bit-identical to lafs_layernorm_fwd on
   * >= 4096 rows; ln_stats (f32 [M, 2]: mean, rstd) and ln_out (bf16 [M, 384]: the operand itself -- what the LayerNorm backward
   * and the fc1 weight gradient read) are written when not NULL. */
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  float* ln_stats; void* ln_out; int ldln;
  /* LayerNorm backward in the epilogue (LAFS_MLP_BWD with ln_gamma != NULL): out is not written; instead, with x = resid (f32
   * [M, 384]) and its statistics ln_stats (read), dx = LN'(bf16(dX)) is added into ln_g_io (f32 [M, 384], the residual gradient
   * stream), ln_gb_out (bf16 [M, 384]) = bf16(seq_scale[row2seq[m]] * ln_g_io) and the launch's gamma / beta sums go to
   * ln_part_out (f32 [lafs_mlp_fused_ln_parts(M)][2][384], the slot layout of lafs_layernorm_bwd: fold with
   * lafs_layernorm_bwd_fold) -- what lafs_layernorm_bwd(accumulate = 1) computes from the stored dX, without the launch. */
  float* ln_g_io; int ldgio; void* ln_gb_out; int ldgb; float* ln_part_out;
  /* The NEXT block's LayerNorm 1 in the epilogue (forward modes; vision_transformer.py:110 `self.norm1` of block l + 1): with
   * next_ln_gamma != NULL the finished rows of `out` are normalised where they sit in registers -- next_ln_out (bf16 [M, 384]) =
   * LayerNorm(out; next_ln_gamma, next_ln_beta, next_ln_eps), next_ln_stats (f32 [M, 2], may be NULL) = (mean, rstd) -- what
   * lafs_layernorm_fwd on `out` computes, bit for bit at >= 4096 rows (the summation order of its two-rows-per-wave kernel is repeated). */
  const float* next_ln_gamma; const float* next_ln_beta; float next_ln_eps;
  float* next_ln_stats; void* next_ln_out; int ldnln_next;
  /* The attention branch's output projection in FRONT (forward modes with ln_gamma != NULL; vision_transformer.py:88-90 `self.proj`
   * and the residual / DropPath of Block.forward :111): with proj_x != NULL (bf16 [M, 384], the attention output) the kernel first
   * computes  resid = proj_resid + proj_scale[row2seq[m]] * (proj_x proj_w^T + proj_bias)  -- `resid` (fp32 [M, 384]) is then an
   * OUTPUT, written once, normalised from registers (LayerNorm 2) and read back by the final epilogue as the residual; proj_w bf16
   * [384, 384] row-major (out, in), proj_bias fp32 [384] or NULL, proj_resid fp32 [M, 384], proj_scale per sequence or NULL.
   * Replaces lafs_gemm_nt(LAFS_EPI_RESID_F32) of the projection, bit for bit where that call runs on the K-resident kernel. */
  const void* proj_x; int ldpx; const void* proj_w; int ldpw; const float* proj_bias;
  const float* proj_resid; int ldpr; const float* proj_scale;
} lafs_mlp_args;
int lafs_mlp_fused(const lafs_mlp_args* args, hipStream_t stream);
/* Slots a LAFS_MLP_BWD launch with the LayerNorm epilogue writes for M rows (one per 128-row workgroup). */
int lafs_mlp_fused_ln_parts(int rows);
/* 1 when lafs_mlp_fused takes this geometry (dim == 384, hidden % 64 == 0 in [128, 1536]), else 0. */
int lafs_mlp_fused_supported(int dim, int hidden, int rows);

/* C[N1,N2] (f32) += A[M,N1]^T * B[M,N2]   (weight gradient dW = dY^T X; reduction over the token axis,
 * split over `splits` workgroups with fp32 atomics; splits <= 0 picks a default).  N1,N2,lda,ldb % 8 == 0.
 * colsum_a (optional f32 [N1]) += column sums of A: the bias gradient db = sum_m dY[m,:] rides along for free. */
int lafs_gemm_tn_acc(const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                     int M, int N1, int N2, int splits, float* colsum_a, hipStream_t stream);
/* Same contraction, accumulated into per-XCD images: part(f32) [LAFS_N_XCD][...], image x at part + x*part_stride.  Every
 * XCD adds into its own image with L2-local atomics (no cross-XCD traffic); fold with lafs_reduce_partials.  The images must
 * be zero before the first accumulation (lafs_reduce_partials leaves them zeroed). */
int lafs_gemm_tn_part(const void* A, int lda, const void* B, int ldb, float* part, int ldc, int64_t part_stride,
                      int M, int N1, int N2, int splits, float* colsum_a, hipStream_t stream);
/* out(f32)[i] += sum_x part[x*part_stride + i], then part[...] = 0;  n, part_stride multiples of 4. */
int lafs_reduce_partials(float* part, int64_t part_stride, int n_part, int64_t n, float* out, hipStream_t stream);
/* out(f32)[n] = sum over x < n_part of part[x * part_stride + i]: folds the slice images of a K-split LAFS_EPI_F32 GEMM. */
int lafs_sum_slices(const float* part, int64_t part_stride, int n_part, int64_t n, float* out, hipStream_t stream);

/* Second-generation weight gradient (csrc/wgrad.hip): C[N1,N2] = (accumulate ? C : 0) + A[M,N1]^T * B[M,N2].
 * One 4-wave workgroup per CU (one wave per SIMD, 512 registers) owns a (64 FA) x (64 FB) tile of v_mfma_f32_32x32x16_bf16
 * blocks (192x192 .. 256x256) over one slice of the token axis and STORES its fp32 tile into `workspace`
 * ([slices][N1][N2] f32 per GEMM); a fold kernel then writes C: no atomics, and with accumulate = 0 the gradient buffer needs
 * no memset.  colsum_a (f32 [N1]) += column sums of A: every (slice, tile, wave column) stores its share into the workspace and
 * the fold kernel adds the shares in a fixed order (deterministic; fp32 atomics before round 5).
 * lafs_wgrad_group runs up to 8 such GEMMs that share the token count M (the four weight gradients of one transformer block)
 * as ONE launch + ONE fold: the tiles of all GEMMs fill the chip together, so the token axis is cut into 4x fewer slices --
 * 4x less partial-sum traffic, 4x longer main loops per workgroup.
 * Calls that share a workspace must be ordered on one stream.  N1, N2, lda, ldb % 8 == 0, ldc % 4 == 0. */
typedef struct lafs_wgrad_item {
  const void* A; int lda;          /* bf16 [M, N1]  (dY)                            */
  const void* B; int ldb;          /* bf16 [M, N2]  (X)                             */
  float* C; int ldc;               /* f32 [N1, N2]  (dW)                            */
  int N1, N2;
  int accumulate;                  /* 0: C = A^T B ; 1: C += A^T B                  */
  float* colsum_a;                 /* f32 [N1] += column sums of A (bias gradient), or NULL */
} lafs_wgrad_item;
/* max_workgroups: how many workgroups (= CUs) the launch may occupy, 8..256; 0 = the whole chip.  A backward pass that runs
 * its weight gradients on a side stream keeps part of the chip free for the HBM-bound kernels of the main stream this way
 * (the LAFS step: 160 of 256, 18.8 -> 18.5 ms).  The workspace size depends on it. */
int64_t lafs_wgrad_group_workspace_bytes(const lafs_wgrad_item* items, int n_items, int M, int max_workgroups);
int lafs_wgrad_group(const lafs_wgrad_item* items, int n_items, int M, int max_workgroups, void* workspace,
                     int64_t workspace_bytes, hipStream_t stream);
int64_t lafs_wgrad_workspace_bytes(int M, int N1, int N2);
/* lafs_wgrad with fp16 operands (the landmark CNN's activations and activation gradients, see lafs_gemm_nt_args::operand_f16). */
int lafs_wgrad_f16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                   int accumulate, float* colsum_a, void* workspace, int64_t workspace_bytes, hipStream_t stream);
int lafs_wgrad(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2,
               int accumulate, float* colsum_a, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* out(f32)[n] += sum_m X(bf16)[m, n]   (bias gradients). */
int lafs_colsum_bf16_acc(const void* X, int ldx, int M, int N, float* out, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm  (vision_transformer.py:99,103,156 eps 1e-6; face_pre_pro/ViT_face.py:117 eps 1e-5)
 * ------------------------------------------------------------------------------------------------ */
/* y(bf16)[r,:] = (x(f32)[r,:] - mean) * rstd * gamma + beta ; stats(f32)[r] = {mean, rstd}.
 * y_f32 (optional, may be NULL) receives the same result in fp32.  D % 4 == 0, D <= 2048. */
int lafs_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                       void* y_bf16, int ldy, float* y_f32, int ldyf, float* stats, int rows, int D,
                       hipStream_t stream);
/* Backward.  dy is bf16 [rows, D] (dy_f32 != NULL: use that fp32 gradient instead).
 *   dx = LN'(dy);  g_io(f32)[r,:] = (accumulate ? g_io : 0) + dx
 *   dgamma(f32)[D] += sum_r dy*xhat ; dbeta(f32)[D] += sum_r dy
 *   gb_out(bf16, optional)[r,:] = bf16(seq_scale[row2seq[r]] * g_io[r,:])  -- the DropPath-scaled gradient fed to
 *   the previous residual branch's GEMMs (seq_scale NULL -> scale 1). */
int lafs_layernorm_bwd(const void* dy_bf16, int lddy, const float* dy_f32, int lddyf, const float* x, int ldx,
                       const float* stats, const float* gamma, float* g_io, int ldg, int accumulate,
                       void* gb_out, int ldgb, const float* seq_scale, const int32_t* row2seq,
                       float* dgamma, float* dbeta, int rows, int D, float drop_p, uint32_t drop_seed,
                       const float* drop_step, int drop_row0, float* part_out, hipStream_t stream);
/* Deterministic parameter gradients: with part_out != NULL (f32 [lafs_layernorm_bwd_parts(rows, D)][2][D]) the launch stores every
 * workgroup's gamma / beta sums into a slot of its own and leaves dgamma / dbeta alone (they may be NULL);
 * lafs_layernorm_bwd_fold then adds the slots of up to 4 launches per LayerNorm (the row chains of a trunk pass) in a fixed order:
 * dgamma[c] += sum, dbeta[c] += sum.  Without part_out the sums leave as one fp32 atomic per column and workgroup. */
#define LAFS_LN_FOLD_MAX 24
typedef struct lafs_ln_fold_item {
  const float* part[4]; int n_parts[4];      /* unused entries: NULL / 0 */
  float* dgamma; float* dbeta;
} lafs_ln_fold_item;
int lafs_layernorm_bwd_parts(int rows, int D);
int lafs_layernorm_bwd_fold(const lafs_ln_fold_item* items, int n_items, int D, hipStream_t stream);

/* gb(bf16)[r,:] = bf16(seq_scale[row2seq[r]] * g(f32)[r,:])  (seq_scale NULL -> plain cast). */
int lafs_scale_cast_bf16(const float* g, int ldg, void* gb, int ldgb, const float* seq_scale,
                         const int32_t* row2seq, int rows, int D,
                         float drop_p, uint32_t drop_seed, const float* drop_step, int drop_row0, hipStream_t stream);
/* Element dropout (nn.Dropout of Part-fViT, face_pre_pro/ViT_face.py:131-133,150-153,614).  The mask is a pure function
 * of (drop_seed, row, col, n_cols): factor(r,c) = mix32(r*n_cols + c, seed) >= drop_p*2^32 ? 1/(1-drop_p) : 0, so a
 * backward kernel regenerates the forward's mask from the same seed; drop_p = 0 disables it.  In lafs_layernorm_bwd and
 * lafs_scale_cast_bf16 the factor multiplies gb_out (the gradient entering a dropped-out branch output).
 * lafs_dropout_f32: x(f32)[rows, D] *= factor in place (embedding dropout, forward and backward).
 * lafs_debug_dropout_mask: out(f32)[rows, cols] = factor (tests feed it to the oracle).
 * drop_step (NULL or a DEVICE pointer to a float step counter): the seed used is drop_seed + 7919 * (uint32)step, read when the
 * kernel runs, so a hipGraph-captured training step draws new masks on every replay (reference: nn.Dropout draws per call);
 * drop_row0: the launch covers rows [drop_row0, drop_row0 + rows) of the mask (row chains of the trunk passes).  A host-side
 * seed' = drop_seed + 7919 * step with drop_step = NULL produces the identical mask. */
int lafs_dropout_f32(float* x, int ldx, int rows, int D, float drop_p, uint32_t drop_seed, const float* drop_step, hipStream_t stream);
int lafs_debug_dropout_mask(int rows, int cols, float drop_p, uint32_t drop_seed, float* out, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-head attention over variable-length sequences, head_dim = 64
 *   vision_transformer.py:80-89 (scale = head_dim^-0.5); face_pre_pro/ViT_face.py:155-179 (scale = dim^-0.5)
 * qkv(bf16) [T, 3*H*64]: columns [q | k | v], each H*64 wide, head-major.  cu_seqlens(i32, device) [n_seq+1].
 * All sequences of one call must have length <= max_len <= 256.  cu_seqlens is read through the scalar (constant) cache: it must not
 * be written while a launch that reads it is in flight (the engines build it once per geometry).
 * ------------------------------------------------------------------------------------------------ */
int lafs_attention_fwd(const void* qkv, int ldqkv, const int32_t* cu_seqlens, int n_seq, int max_len, int heads,
                       float scale, void* out_bf16, int ldo, float* lse, hipStream_t stream);
/* dqkv(bf16) [T, 3*H*64] from dout(bf16) [T, H*64], out(bf16) and lse(f32) [T, H] of the forward: one launch per call (Q, K, V,
 * dO of a (sequence, head) staged once; delta = rowsum(dO * O) formed on the way in). */
int lafs_attention_bwd(const void* qkv, int ldqkv, const void* out_bf16, int ldo, const void* dout_bf16, int lddo,
                       const float* lse, const int32_t* cu_seqlens, int n_seq, int max_len, int heads, float scale,
                       void* dqkv, int lddqkv, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Patch embedding front end  (vision_transformer.py:126-131, 196-207; face_pre_pro/ViT_face.py:760-766)
 * ------------------------------------------------------------------------------------------------ */
enum { LAFS_PATCH_ORDER_CHW = 0 /* Conv2d weight [D,c,p1,p2] */, LAFS_PATCH_ORDER_HWC = 1 /* '(p1 p2 c)' */ };
/* img(f32) [B,3,S,S] -> patches(bf16) [B*(S/p)^2, 3*p*p], p = 8. */
int lafs_patchify(const float* img, int B, int S, int order, void* patches, hipStream_t stream);
/* tokens(f32)[b*(np+1), :] = cls[:] + pos[0,:]  for every sequence (the patch rows are written by the
 * LAFS_EPI_EMBED_F32 GEMM epilogue). */
int lafs_embed_cls(const float* cls, const float* pos, float* tokens, int ldt, int n_seq, int npatch, int D,
                   hipStream_t stream);
/* Backward of the token assembly: g(f32) [n_seq*(np+1), D] ->
 *   gp(bf16) [n_seq*np, D] (patch rows only), dpos(f32)[np+1, D] += sum over sequences, dcls(f32)[D] += sum g[cls rows].
 * workspace (f32, lafs_embed_bwd_workspace_bytes; may be NULL): with it every chunk of 16 sequences stores its sums into a slot of its
 * own and a second launch adds the slots in ascending order -- run-to-run deterministic gradients of the position table and the cls
 * token; without it the sums leave as one fp32 atomic per element and chunk (round 1-5 behaviour). */
int64_t lafs_embed_bwd_workspace_bytes(int n_seq, int npatch, int D);
int lafs_embed_bwd(const float* g, int ldg, int n_seq, int npatch, int D, void* gp, float* dpos, float* dcls,
                   float* workspace, hipStream_t stream);
/* feat(bf16 and/or f32) [n_seq, D] = x[row of cls token of each sequence]; and its scatter-back. */
int lafs_gather_cls(const float* x, int ldx, const int32_t* cu_seqlens, int n_seq, int D, float* out_f32,
                    hipStream_t stream);
/* g(f32)[first row of each sequence, :] = src[n_seq, D]  (g is expected to be zero elsewhere). */
int lafs_scatter_cls(const float* src, const int32_t* cu_seqlens, int n_seq, int D, float* g, int ldg,
                     hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DINO head pieces  (vision_transformer.py:284-287, 299-300)
 * ------------------------------------------------------------------------------------------------ */
/* y = x / max(||x||_2, 1e-12) row-wise: x f32 [rows, D] -> y bf16 (+ inv_norm f32 [rows]). */
int lafs_l2norm_fwd(const float* x, int ldx, void* y_bf16, int ldy, float* inv_norm, int rows, int D,
                    hipStream_t stream);
/* dx(f32) = inv_norm * (dy - y * <y, dy>)   with y recomputed from x. */
int lafs_l2norm_bwd(const float* x, int ldx, const float* dy, int lddy, const float* inv_norm, float* dx, int lddx,
                    int rows, int D, hipStream_t stream);
/* weight_norm (dim=0): w(bf16)[k,:] = g[k] * v[k,:] / ||v[k,:]|| ; also w_t(bf16) [D, ldwt] transposed copy
 * (NULL to skip); rows k in [K, Kpad) of w and columns of w_t are zero-filled.  inv_norm f32 [K]. */
int lafs_weightnorm_fwd(const float* v, const float* g, int K, int Kpad, int D, void* w, void* w_t, int ldwt,
                        float* inv_norm, hipStream_t stream);
/* dv(f32)[k,:] (+)= g/||v|| * (dw - vhat * <dw, vhat>) ; dg(f32)[k] (+)= <dw, vhat>  (dg may be NULL). */
int lafs_weightnorm_bwd(const float* dw, const float* v, const float* g, const float* inv_norm, int K, int D,
                        float* dv, float* dg, int accumulate, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DINO loss (lafs_train.py:643-679), closed form:
 *   q_i = softmax((t_i - c)/tau_t),  p_v = softmax(s_v / tau_s)
 *   loss = 1/n_terms * sum_{i in {0,1}, v != i} mean_b [ lse(s_v/tau_s) - <q_i, s_v/tau_s> ]
 *   dL/ds_v = 1/(n_terms * B * tau_s) * sum_{i != v} (p_v - q_i)
 * student f32 [ncrops*B, ld], teacher f32 [2*B, ld], center f32 [K]; grad bf16 or f32 [ncrops*B, ldg].
 * workspace f32: lafs_dino_loss_workspace(ncrops, B, K) floats.  dev_temps (optional, device f32[2] =
 * {student_temp, teacher_temp}) overrides the two host values so that a captured hipGraph follows the
 * teacher-temperature schedule.
 * ------------------------------------------------------------------------------------------------ */
int64_t lafs_dino_loss_workspace(int ncrops, int B, int K);
int lafs_dino_loss_fwd_bwd(const float* student, const float* teacher, int ld, const float* center, int ncrops,
                           int B, int K, float student_temp, float teacher_temp, float* loss_out,
                           void* grad, int ldg, int grad_is_bf16, float grad_scale, float* workspace,
                           const float* dev_temps, hipStream_t stream);
/* DINOHead's last layer of BOTH networks + DINOLoss.forward + the row sums of update_center in one piece (vision_transformer.py:
 * 295-301, lafs_train.py:643-679): the [ncrops B + 2 B, K] logits are formed twice on the MFMA (a K = 256 contraction per class) and
 * never stored.  zn_*: bf16 [rows, 256] L2-normalised bottleneck rows (student: ncrops B rows in crop-major order, teacher: 2 B),
 * wn_*: bf16 [Kpad, 256] weight-normalised last-layer rows (pad rows zero), center f32 [K].  Outputs: loss_out f32 [1];
 * grad_bf16 [ncrops B, ldg] = grad_scale * dL/d(student logits) (pad columns K..Kpad zero); colsum_out f32 [K] (optional) =
 * sum over the teacher rows of the raw teacher logits.  Same closed form as lafs_dino_loss_fwd_bwd; no atomics: deterministic.
 * dim must be 256, B <= 64, Kpad % 8 == 0.  workspace f32: lafs_dino_head_loss_workspace(ncrops, B, K) floats. */
int64_t lafs_dino_head_loss_workspace(int ncrops, int B, int K);
int lafs_dino_head_loss(const void* zn_student, const void* zn_teacher, const void* wn_student, const void* wn_teacher, int dim,
                        const float* center, int ncrops, int B, int K, int Kpad, float student_temp, float teacher_temp,
                        const float* dev_temps, float* loss_out, void* grad_bf16, int ldg, float grad_scale, float* colsum_out,
                        float* workspace, hipStream_t stream);
/* colsum(f32)[k] = sum_rows teacher[r, k]   (lafs_train.py:674; all-reduced by the caller) */
int lafs_colsum_f32(const float* x, int ld, int rows, int K, float* out, hipStream_t stream);
/* center = center*m + colsum/(rows_total) * (1-m)   (lafs_train.py:676-679) */
int lafs_center_ema(float* center, const float* colsum, int K, float inv_rows_total, float momentum,
                    hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-tensor step glue over a flat parameter arena split into 1024-element chunks
 *   utils.py:132-141 (per-tensor clip), lafs_train.py:400 + torch AdamW, lafs_train.py:610-613 (EMA)
 * chunk_seg(i32)[n_chunks] maps each chunk to its tensor (segment).  seg_flags bit0 = weight decay on,
 * bit1 = "last_layer" (skipped while frozen), bit2 = trainable.  Hyper-parameters live in a device f32[16]:
 * {lr, wd, beta1, beta2, eps, clip, ema_m, freeze_last_layer(0/1), grad_scale, ...}.
 * ------------------------------------------------------------------------------------------------ */
#define LAFS_N_XCD 8          /* XCDs (L2 domains) of an MI355X */
#define LAFS_CHUNK 1024
enum { LAFS_SEG_DECAY = 1, LAFS_SEG_LAST_LAYER = 2, LAFS_SEG_TRAINABLE = 4,
       LAFS_SEG_OVERWRITTEN = 16 /* the step's first gradient writer of this tensor OVERWRITES it (wgrad fold with accumulate = 0,
                                     weight-norm backward): lafs_zero_chunks(skip_mask = 16) leaves it alone */,
       LAFS_SEG_LOW_DECAY = 8 /* decays at hyper[LAFS_HP_WD_LOW] instead of hyper[LAFS_HP_WD]: the `stn*` matrices of the
                                  fine-tune step, train_largescale.py:143-145 (5e-2 against 1e-1) */ };
enum { LAFS_HP_LR = 0, LAFS_HP_WD, LAFS_HP_BETA1, LAFS_HP_BETA2, LAFS_HP_EPS, LAFS_HP_CLIP, LAFS_HP_EMA_M,
       LAFS_HP_FREEZE_LAST, LAFS_HP_GRAD_SCALE, LAFS_HP_WD_LOW,
       LAFS_HP_STEP /* optimisation step count (as a float): seeds the per-step DropPath masks of a replayed graph */,
       LAFS_HP_MIX_LAM /* mixup lambda of this fine-tune micro-step (lam_dev of lafs_mixup_normalize / lafs_margin_softmax_ce_bf16) */,
       LAFS_HP_COUNT = 16 };
/* seg_sumsq(f32)[n_seg] = sum of (grad_scale*g)^2 per segment -- the per-tensor norms of utils.clip_gradients
 * (utils.py:132-141).  Two passes through chunk_sumsq(f32)[n_chunks] scratch, fixed summation order: no atomics, nothing
 * to pre-zero, bitwise reproducible. */
int lafs_grad_sumsq(const float* grad, const int32_t* chunk_seg, int64_t n_chunks, int n_seg, const float* hyper,
                    float* chunk_sumsq, float* seg_sumsq, hipStream_t stream);
/* Fused per-tensor clip + AdamW + teacher EMA + bf16 shadow refresh.  seg_step(i32)[n_seg] counts the
 * optimizer steps each tensor has actually taken.  teacher/shadow pointers may be NULL. */
int lafs_clip_adamw_ema(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* teacher,
                        void* param_bf16, void* teacher_bf16, const int32_t* chunk_seg, int64_t n_chunks,
                        const int32_t* seg_flags, int32_t* seg_step, int n_seg, const float* seg_sumsq,
                        const float* hyper, hipStream_t stream);
/* The same two passes restricted to the tensors [seg_lo, seg_hi) = the chunks [chunk_lo, chunk_hi) (base pointers are those of the
 * whole arena): the engine updates a range of the arena as soon as its gradients are final, beside the rest of the backward. */
int lafs_grad_sumsq_range(const float* grad, const int32_t* chunk_seg, int64_t n_chunks, int n_seg, int64_t chunk_lo, int64_t chunk_hi,
                          int seg_lo, int seg_hi, const float* hyper, float* chunk_sumsq, float* seg_sumsq, hipStream_t stream);
/* n_chunks / n_seg are the sizes of the WHOLE arena: a range that leaves them is refused (LAFS_ESHAPE), not clipped. */
int lafs_clip_adamw_ema_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* teacher,
                              void* param_bf16, void* teacher_bf16, const int32_t* chunk_seg, int64_t n_chunks, int64_t chunk_lo,
                              int64_t chunk_hi, const int32_t* seg_flags, int32_t* seg_step, int n_seg, int seg_lo, int seg_hi,
                              const float* seg_sumsq, const float* hyper, hipStream_t stream);
/* dst(bf16)[i] = src(f32)[i] */
int lafs_cast_bf16(const float* src, void* dst, int64_t n, hipStream_t stream);
/* dst(f32)[i] = src(bf16)[i]   (gradients that travelled over the wire as bf16: LAFS_GRAD_WIRE=bf16, distributed.py) */
int lafs_cast_f32(const void* src, float* dst, int64_t n, hipStream_t stream);
/* dst(bf16)[c, r] = src(bf16)[r, c]: operand transposes of the fine-tune margin head (dL/dcos [B, C] -> [C, B] feeds both class-
 * gradient GEMMs, train_largescale.py:820-867's autograd through ViT_face.py:49-89; was a torch .t().contiguous()). */
int lafs_transpose_bf16(const void* src, int rows, int cols, int ld_src, void* dst, int ld_dst, hipStream_t stream);
/* dst(bf16)[c, r] = src(f32)[r, c]   (W^T shadows used by the dgrad GEMMs) */
int lafs_transpose_cast_bf16(const float* src, int rows, int cols, void* dst, int ld_dst, hipStream_t stream);
/* All W^T shadows of an arena in ONE launch.  table(i64, device)[4*i..] = {src offset into master, rows, cols, dst offset
 * into shadow_t}; tile_start(i32, device)[n_mat+1] = prefix sum of each matrix's 64x64-tile count; n_tiles = tile_start[n_mat]. */
int lafs_transpose_cast_table(const float* master, void* shadow_t, const int64_t* table, const int32_t* tile_start,
                              int n_mat, int n_tiles, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Transformer engine: whole pre-LN ViT trunk forward/backward over a packed token batch
 *   vision_transformer.py:107-113, 209-215; face_pre_pro/ViT_face.py:106-120, 184-213
 * Offsets are in ELEMENTS into the caller's arenas: `master` (f32 parameters), `shadow` (bf16 copy, same
 * offsets), `shadow_t` (bf16 transposed weights, offsets *_t), `grad` (f32, same offsets as master).
 * ------------------------------------------------------------------------------------------------ */
typedef struct lafs_block_offsets {
  int64_t ln1_g, ln1_b, w_qkv, b_qkv /* -1 if none */, w_proj, b_proj, ln2_g, ln2_b, w_fc1, b_fc1, w_fc2, b_fc2;
  int64_t w_qkv_t, w_proj_t, w_fc1_t, w_fc2_t;     /* into shadow_t */
} lafs_block_offsets;

typedef struct lafs_trunk_desc {
  int dim, inner /* heads*64 */, heads, mlp, depth;
  float ln_eps, attn_scale;
  int n_tok, n_seq, max_len;
  const int32_t* cu_seqlens;      /* device i32 [n_seq+1] */
  const int32_t* row2seq;         /* device i32 [n_tok]   */
  const float* drop_scales;       /* device f32 [depth, 2, n_seq] or NULL */
  float dropout_p;                /* element dropout after to_out / after GELU / after fc2 (Part-fViT); 0 = off.  Site s of
                                     layer l uses seed dropout_seed + 3*l + s (s: 0 to_out, 1 GELU, 2 fc2)                 */
  uint32_t dropout_seed;
  const float* master; const void* shadow; const void* shadow_t; float* grad;
  const lafs_block_offsets* blocks;   /* HOST array [depth] */
  /* Sequence groups of equal length (the crop resolutions of a packed multi-crop batch, in packing order): attention is
   * launched once per group with the tile configuration of THAT length (197-token global crops: one (sequence, head) pair
   * per workgroup, 53 KB of LDS; 37-token local crops: four pairs per workgroup) instead of once with the longest one.
   * n_groups = 0: a single launch over all sequences with max_len. */
  int n_groups; int group_n_seq[4]; int group_max_len[4];
  int wgrad_workgroups;               /* CUs the grouped weight-gradient launch may occupy (0 = all 256), see lafs_wgrad_group */
  const float* dropout_step;          /* NULL or DEVICE pointer to the step counter added (x 7919) to dropout_seed inside the kernels */
  int wgrad_overwrite;                /* != 0: the block weight gradients are WRITTEN (not accumulated): the caller zeroes only the
                                         other tensors (lafs_zero_chunks, LAFS_SEG_OVERWRITTEN) and runs one backward per step */
  int wgrad_defer;                    /* != 0: lafs_trunk_backward launches NO weight-gradient GEMMs and keeps every layer's operands
                                         (dY of fc2 / fc1 / proj / qkv) in slots of their own -- the workspace grows by depth - 2 sets --
                                         until lafs_trunk_wgrad launches them, e.g. beside work that leaves the chip idle (the
                                         fine-tune step's landmark-CNN backward, train_largescale.py:785-891) */
  lafs_ctx* ctx;                      /* side streams / events / options of these passes (NULL: caller's stream only, default options) */
} lafs_trunk_desc;

/* Bytes of activation workspace for a forward with (1) / without (0) saving activations for backward. */
int64_t lafs_trunk_workspace_bytes(const lafs_trunk_desc* d, int save_for_backward);
/* Number of independent chains of launches (row ranges of the token batch, one stream each) the trunk passes of this descriptor
 * run as: 2 when there are two crop-resolution groups of >= 4096 full-length rows each (csrc/engine.hip:
 * row_ranges), else 1.  Tests assert the route they mean to cover. */
int lafs_trunk_row_ranges(const lafs_trunk_desc* d);
/* x_in(f32) [n_tok, dim] -> x_out(f32) [n_tok, dim]: residual stream after the last block.  With
 * save_for_backward != 0 x_in must stay untouched until lafs_trunk_backward has run (it is layer 0's saved input)
 * and must not alias x_out. */
int lafs_trunk_forward(const lafs_trunk_desc* d, const float* x_in, float* x_out, void* workspace,
                       int save_for_backward, hipStream_t stream);
/* g(f32) [n_tok, dim]: dL/dx_out on entry, dL/dx_in on exit.  Parameter gradients are ACCUMULATED into d->grad.
 * Blocks layer_hi-1 .. layer_lo run; pass (depth, 0) for the whole trunk, or walk it in slices to start the
 * gradient all-reduce of finished blocks early.  wgrad_stream (optional, may be NULL or == stream): the weight-gradient
 * GEMMs are enqueued there and overlap the dgrad / attention / LayerNorm chain on `stream` (forked and joined with
 * events, so the call is still hipGraph-capturable and complete on `stream` when it returns to stream order). */
int lafs_trunk_backward(const lafs_trunk_desc* d, const float* x_in, float* g, void* workspace, int layer_hi,
                        int layer_lo, hipStream_t wgrad_stream, hipStream_t stream);
/* The weight gradients (and bias gradients) of blocks layer_hi-1 .. layer_lo that a lafs_trunk_backward with d->wgrad_defer set
 * left out: one grouped launch per block on `stream` (at most d->wgrad_workgroups workgroups each), reading the per-layer operand
 * slots of `workspace`.  Same kernels, same operands, same accumulate / overwrite rule as the immediate form: identical results.
 * Replaces autograd's dW = dY^T X of the block linears (vision_transformer.py:59-65, 75-90; face_pre_pro/ViT_face.py:126-149). */
int lafs_trunk_wgrad(const lafs_trunk_desc* d, void* workspace, int layer_hi, int layer_lo, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Step glue that used to run on ATen / rocBLAS inside the captured step (csrc/stepglue.hip)
 * ------------------------------------------------------------------------------------------------ */
/* Stochastic depth (vision_transformer.py:27-35; ViT_face.py:106-112): scales(f32)[depth, 2, n_seq] = 1/keep_l with probability
 * keep_l, else 0; keep_prob f32 [depth] on the device.  Counter-based: hash(seed, step, index), `step_dev` (device f32, may be
 * NULL) = &hyper[LAFS_HP_STEP], so a replayed hipGraph draws new masks every step.  Same distribution as torch.rand-based
 * drop_path, different random stream. */
int lafs_droppath_scales(const float* keep_prob, int depth, int n_seq, uint32_t seed, const float* step_dev, float* scales,
                         hipStream_t stream);
/* interpolate_pos_encoding (vision_transformer.py:174-194) as its fixed linear map: out(f32)[1+R, D]: row 0 = cls row of the
 * table, rows 1.. = interp(f32 [R, G]) @ table[1:, :]  (table f32 [1+G, D]).  bwd: grad_table += the transposed map of dpos. */
int lafs_pos_interp_fwd(const float* pos_table, const float* interp, float* out, int R, int G, int D, hipStream_t stream);
int lafs_pos_interp_bwd(const float* dpos, const float* interp, float* grad_table, int R, int G, int D, hipStream_t stream);
/* Zero the 1024-float chunks of a gradient arena whose segment flags do not intersect skip_mask. */
int lafs_zero_chunks(float* buf, const int32_t* chunk_seg, const int32_t* seg_flags, int64_t n_chunks, int skip_mask,
                     hipStream_t stream);
/* buf[0 .. bytes) = 0 as a kernel launch on `stream` (16-byte aligned buffer). */
int lafs_fill_zero(void* buf, int64_t bytes, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fine-tune path (train_largescale.py): margin-softmax head, mixup, landmark patch gather
 * ------------------------------------------------------------------------------------------------ */
/* CosFace logits + soft-target CE, fused over the class axis (face_pre_pro/ViT_face.py:49-89;
 * timm SoftTargetCrossEntropy, train_largescale.py:820).  cos(f32) [B, ld] holds x_hat . w_hat^T on entry and
 * dL/dcos on exit.  The (mixup) target is given sparsely: y1,y2 (i32 [B]) with weights lam,(1-lam).
 * loss_out(f32)[1] = mean_b CE.  margin_type 0 = CosFace s*(cos - m*y), 1 = ArcFace (hard labels only). */
int lafs_margin_softmax_ce(float* cos, int ld, int B, int C, const int32_t* y1, const int32_t* y2, float lam,
                           float s, float m, int margin_type, float loss_scale, float* loss_out, float* row_ws,
                           hipStream_t stream);
/* Class-sharded margin softmax (PartialFC; PARITY UNPINNED: absent from the reference, ViT_face.py:645-649 is a commented
 * import; semantics follow InsightFace partial_fc_v2).  cos(f32) [B, ld] over this rank's S (sampled) class centres;
 * y_local(i32)[B] = local index of the target or -1.  Soft (mixup) targets as the reference always feeds its margin head
 * (train_largescale.py:802; dense lam e_y1 + (1 - lam) e_y2 entering the margin itself, ViT_face.py:69-73): y2_local(i32)[B] = local
 * index of the mixup partner's class or -1, lam(f32)[B] = the row's lambda (NULL: 1); y2_local == NULL: hard labels.  CosFace only.
 * Three passes with the cross-rank statistics exchanged in between (all-reduce MAX of rowmax, all-reduce SUM of rowsum and
 * target_logit):  z = margin logits s (cos - m y);
 *   rowmax[b] = max_k z ;  rowsum[b] = sum_k exp(z - gmax[b]), target_logit[b] = sum over the targets this rank owns of y_k z_k ;
 *   grad (in place) cos <- grad_scale * (exp(z - gmax)/Z - y) * dz/dcos.      loss_b = log Z + gmax - target_logit. */
int lafs_shard_margin_rowmax(const float* cos, int ld, int B, int S, const int32_t* y_local, const int32_t* y2_local, const float* lam,
                             float s, float m, int margin_type, float* rowmax, hipStream_t stream);
int lafs_shard_margin_rowsum(const float* cos, int ld, int B, int S, const int32_t* y_local, const int32_t* y2_local, const float* lam,
                             float s, float m, int margin_type, const float* gmax, float* rowsum, float* target_logit, hipStream_t stream);
int lafs_shard_margin_grad(float* cos, int ld, int B, int S, const int32_t* y_local, const int32_t* y2_local, const float* lam, float s,
                           float m, int margin_type, const float* gmax, const float* Z, float grad_scale, hipStream_t stream);
/* x(f32) [B,3,S,S]: x = lam*x + (1-lam)*flip_batch(x), from a u8 source with (x/255*2-1) folded in (util/mixup_my.py:189-200 on
 * the loader's tensors, train_largescale.py:842-846).  lam_dev != NULL: lambda is read from DEVICE memory when the kernel runs (a
 * captured fine-tune step replays with a new lambda per micro-step); otherwise `lam`. */
int lafs_mixup_normalize(const uint8_t* src_u8, float* dst, int B, int S, float lam, const float* lam_dev, hipStream_t stream);
/* The fused margin-softmax + soft-target CE of lafs_margin_softmax_ce over (row, chunk) workgroups, for the captured fine-tune
 * step: cos(f32) [B, ld] is READ-ONLY, dL/dcos leaves as bf16 in dcos [B, lddc] (pad columns [C, lddc) zeroed) -- the operand of
 * the two class-gradient GEMMs; y2 == NULL: the mixup partner of row b is row B-1-b; lam_dev != NULL: lambda from device memory.
 * row_ws f32 [B], part_ws f32 [B * 32].  Same references as lafs_margin_softmax_ce. */
int lafs_margin_softmax_ce_bf16(const float* cos, int ld, int B, int C, const int32_t* y1, const int32_t* y2, float lam,
                                const float* lam_dev, float s, float m, int margin_type, float loss_scale, void* dcos, int lddc,
                                float* loss_out, float* row_ws, float* part_ws, hipStream_t stream);
/* dst(i32)[n] = src(i64)[n]: the loader's int64 labels (train_largescale.py:842) into the kernels' index type without an ATen cast */
int lafs_cast_i64_i32(const int64_t* src, int32_t* dst, int n, hipStream_t stream);
/* Patch-vector gradient f32 [B, (S/8)^2, 192] -> image gradient f32 [B, 3, S, S]: the inverse re-indexing of lafs_patchify (the
 * landmark branch differentiates through the patches, face_pre_pro/ViT_face.py:679-711; was a torch permute + copy). */
int lafs_unpatchify_f32(const float* dpatch, int B, int S, int order, float* dimg, hipStream_t stream);
/* ------------------------------------------------------------------------------------------------
 * Frozen landmark CNN, inference only (MobileNetV3-large trunk, face_pre_pro/mobilenet.py:224-313, called by
 * face_landmark_4simmin_glo_loc.forward, face_pre_pro/ViT_face.py:1338-1344).  NHWC bf16 activations, channel counts padded
 * to multiples of 32, BatchNorm folded into (w, b) by the host; the 1x1 convolutions are lafs_gemm_nt(LAFS_EPI_BF16_ACT).
 * act: LAFS_ACT_* (negative = none).
 * ------------------------------------------------------------------------------------------------ */
/* stem: x f32 NCHW [N,3,S,S], w f32 [27][16] (index (c*9+ky*3+kx)*16+o), b f32 [16] -> y bf16 NHWC [N,S/2,S/2,ldy]
 * = act(conv3x3 stride 2 pad 1 + b) in channels 0..15, zeros in 16..ldy-1. */
int lafs_cnn_stem(const float* x, const float* w, const float* b, int N, int S, int act, void* y, int ldy, hipStream_t stream);
/* depthwise k x k (k in {3,5}), stride in {1,2}, pad (k-1)/2: x bf16 [N,H,W,C], w f32 [k*k][C], b f32 [C] -> y bf16
 * [N,ceil(H/stride),ceil(W/stride),C] = act(conv + b).  C % 8 == 0. */
int lafs_cnn_dwconv(const void* x, const float* w, const float* b, int N, int H, int W, int C, int k, int stride, int act,
                    void* y, hipStream_t stream);
/* out(bf16)[n, c] = mean over the HW positions of x bf16 [N,HW,C]  (squeeze; final average pool). */
int lafs_cnn_pool(const void* x, int N, int HW, int C, void* out, int ldo, hipStream_t stream);
/* x(bf16)[n,p,c] = act(x[n,p,c] * s(bf16)[n,c]) in place  (excite + the block's non-linearity). */
int lafs_cnn_scale_act(void* x, const void* s, int lds, int N, int HW, int C, int act, hipStream_t stream);

/* Depthwise convolution of the TRAINABLE landmark branch (fine-tune step, nn.Conv2d(C, C, k, stride, (k-1)//2, groups=C,
 * bias=False) of face_pre_pro/mobilenet.py:177-186 and its two gradients), fp32 NCHW.  x [N,C,H,W], w [C,1,k,k],
 * y / dy [N,C,ceil(H/stride),ceil(W/stride)]; k in {3,5}, stride in {1,2}.  bwd_weight ACCUMULATES into dw (zero it first). */
int lafs_dwconv_nchw_fwd(const float* x, const float* w, int N, int C, int H, int W, int k, int stride, float* y, hipStream_t stream);
int lafs_dwconv_nchw_bwd_data(const float* dy, const float* w, int N, int C, int H, int W, int k, int stride, float* dx,
                              hipStream_t stream);
int lafs_dwconv_nchw_bwd_weight(const float* x, const float* dy, int N, int C, int H, int W, int k, int stride, float* dw,
                                hipStream_t stream);

/* BatchNorm2d (+ activation LAFS_ACT_NONE / RELU / HSWISH) of the trainable landmark branch, fp32 NCHW
 * (face_pre_pro/mobilenet.py:104-111,177-190).  x, y, dy, dx [N,C,HW]; stat f32 [2C] receives {mean, rstd} (saved for the
 * backward); sums_ws / dsum f32 [2C] are scratch (dsum returns {dbeta, dgamma} = {sum dz, sum dz*xhat}).
 * training = 1: batch statistics (biased variance) + running-statistics update with `momentum` (unbiased variance), exactly
 * nn.BatchNorm2d.train(); training = 0: running statistics, the backward treats them as constants. */
int lafs_bn_act_fwd_nchw(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float eps,
                         float momentum, int training, int N, int C, int HW, int act, float* sums_ws, float* stat, float* y,
                         hipStream_t stream);
int lafs_bn_act_bwd_nchw(const float* x, const float* dy, const float* stat, const float* gamma, const float* beta, int training,
                         int N, int C, int HW, int act, float* dsum, float* dx, hipStream_t stream);

/* View augmentation of the loader (DataAugmentation_LAFS, lafs_train.py:790-886; Pillow arithmetic, bit-exact):
 * images u8 NCHW [B,3,112,112]; params i32 [B,K,20] records (lafs_cvpr2024_amd/augment.py pack_params: crop box, flags
 * {flip, jitter, gray, solarize}, ColorJitter order + factors, hue shift, box-blur radius/weights); table i32 [113,112,8] =
 * Pillow's bicubic resampling coefficients for every source extent (augment.py coeff_table).
 * views f32 [2K,B,3,112,112]: view 2k = normalised resized crop (+flip) of crop k, view 2k+1 = its colour-augmented twin. */
int lafs_augment_views(const uint8_t* images, const int32_t* params, const int32_t* table, int B, int K, float* views,
                       hipStream_t stream);

/* RandAugment of the fine-tune loader (reference util/rand_aa_face.py:606-672 as FaceDataset builds it, face_pre_pro/
 * dataloader_web.py:240-243 -- 'rand-m1-mstd0.5-inc1', train_largescale.py:506 -- applied per decoded sample at
 * dataloader_web.py:342-346 / image_iter.py:324-329; Pillow arithmetic, bit-exact).  One record per image and layer, sampled on
 * the host in the reference's random order (lafs_cvpr2024_amd/randaug.py):
 *   op        -1 = layer not applied; 0..12 = AutoContrast, Equalize, Invert, Rotate, PosterizeIncreasing, ColorIncreasing,
 *             ContrastIncreasing, BrightnessIncreasing, SharpnessIncreasing, ShearX, ShearY, TranslateXRel, TranslateYRel
 *             (the order of _RAND_INCREASING_TRANSFORMS :560-576); 13 / 14 / 15 = the transposes Image.rotate substitutes for 180 / 90 /
 *             270 degrees
 *   resample  2 = PIL BILINEAR, 3 = BICUBIC (geometric ops)       iarg   Posterize: the byte mask ~(2^(8-bits) - 1)
 *   farg      enhancement factor of Color / Contrast / Brightness / Sharpness (Image.blend's float alpha)
 *   m[6]      the AFFINE matrix handed to Image.transform (output pixel centre -> source position), double precision
 * images / out: u8 [B, H, W, 3] (chw = 0) or [B, 3, H, W] (chw = 1), 3 <= H, W and H * W <= 112 * 112; out may alias images. */
typedef struct lafs_randaug_op {
  int32_t op, resample, iarg;
  float farg;
  double m[6];
} lafs_randaug_op;
int lafs_randaug_apply(const uint8_t* images, uint8_t* out, const lafs_randaug_op* records, int B, int H, int W, int layers, int chw,
                       hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------------
 * TRAINABLE landmark CNN of the fine-tune step (csrc/landmark_train.hip; reference face_pre_pro/mobilenet.py:224-313 trained through
 * ViT_face.py:679-711 by train_largescale.py:785-891).  Activations are NHWC bf16 [N H W, ld] matrices, ld = channel count padded
 * to a multiple of 32 with the pad channels exactly zero; 1x1 convolutions and their two gradients are lafs_gemm_nt / lafs_wgrad
 * calls on those matrices; the rest is below.  `C` is the true channel count, `ld*` the padded row strides.
 * ------------------------------------------------------------------------------------------------------------------------ */
/* 3x3 stride-2 pad-1 stem as im2col rows: x f32 NCHW [N,3,S,S] -> P bf16 [N (S/2)^2, 32], column (c, ky, kx), columns 27..31 zero. */
int lafs_cnn_im2col_stem(const float* x, int N, int S, void* P, hipStream_t stream);
/* Round 4: every 16-bit tensor of this plan (activations, operand images, activation gradients -- "bf16" in the comments below is
 * historical) is IEEE fp16, the format of the reference's autocast run (train_largescale.py:803-804); its GEMMs are lafs_gemm_nt
 * with operand_f16 = 1 and lafs_wgrad_f16.  Activation gradients are scaled by a per-call power of two derived on the device
 * (lafs_cnn_grad_scale) and un-scaled where they leave the 16-bit domain (grad_scale arguments below; NULL = no scaling).
 * nn.BatchNorm2d in TRAINING mode over the rows of x fp16 [R, ldx]:
 *   lafs_cnn_bn_stats : sums(f64)[0..C) += column sums, sums[C..2C) += column sums of squares (caller zeroes `sums`); fp64: mean and
 *                       E[x^2] - mean^2 are formed in double, the order of the atomics stays below fp32 resolution;
 *   lafs_cnn_bn_apply : y = act((x - mean) rstd gamma + beta) (+ resid), biased batch variance; stat(f32)[2C] = {mean, rstd} for the
 *                       backward; running_mean / running_var (NULL or both) get the momentum update with the unbiased variance;
 *   lafs_cnn_bn_bwd   : dz = (dy + add_nc[n, c] / HW) act'(z) with z recomputed from x; dx(fp16) = gamma rstd (dz - mean(dz) -
 *                       xhat mean(dz xhat)); dgamma += sum dz xhat / scale, dbeta += sum dz / scale (arena gradients, may be NULL);
 *                       dsums(f64)[2C] = caller-zeroed scratch.  add_nc fp16 [N, ldadd] (the squeeze-excite pooling gradient) or NULL. */
int lafs_cnn_bn_stats(const void* x, int ldx, int64_t R, int C, double* sums, hipStream_t stream);
int lafs_cnn_bn_apply(const void* x, int ldx, int64_t R, int C, const double* sums, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var, int act, const void* resid, int ldr, void* y, int ldy,
                      float* stat, hipStream_t stream);
int lafs_cnn_bn_bwd(const void* dy, int lddy, const void* x, int ldx, int64_t R, int C, const float* stat, const float* gamma,
                    const float* beta, int act, const void* add_nc, int ldadd, int HW, double* dsums, void* dx, int lddx,
                    float* dgamma, float* dbeta, const float* grad_scale, hipStream_t stream);
/* nn.BatchNorm2d in EVAL mode (a model.eval() forward that is still differentiated, face_pre_pro/ViT_face.py:679-711 under
 * train_largescale.py's test-time paths): lafs_cnn_bn_eval_sums writes the sums a batch with exactly the running statistics would have
 * (sums[c] = R mean, sums[C + c] = R (var + mean^2), fp64), so that lafs_cnn_bn_apply (running_mean = NULL: no update) normalises with
 * them; lafs_cnn_bn_bwd_eval is lafs_cnn_bn_bwd with the statistics as constants: dx = gamma rstd dz, same dgamma / dbeta. */
int lafs_cnn_bn_eval_sums(const float* running_mean, const float* running_var, int64_t R, int C, double* sums, hipStream_t stream);
int lafs_cnn_bn_bwd_eval(const void* dy, int lddy, const void* x, int ldx, int64_t R, int C, const float* stat, const float* gamma,
                    const float* beta, int act, const void* add_nc, int ldadd, int HW, double* dsums, void* dx, int lddx,
                    float* dgamma, float* dbeta, const float* grad_scale, hipStream_t stream);
/* Loss scaling and overflow guard of the fp16 backward -- the reference's torch.cuda.amp.GradScaler (train_largescale.py:739,
 * 803-804, 867-880: scale, skip the step on inf / NaN, back the scale off, grow it again) decided on the device, so that a captured
 * step needs no host round trip.
 *   lafs_cnn_grad_scale: scale(f32, device)[0..1] = {s, 1/s}, s = the power of two <= target / max |g|, clamped to [2^-24, 2^24]
 *     (1 for an all-zero gradient).  has_state != 0: scale is the 8-float state {s, 1/s, target, found_inf, skipped, clean, -, -}:
 *     the target is state[2] when that is > 0 (else the argument), and found_inf is RESET to "g itself holds inf / NaN".
 *   lafs_cnn_grad_guard (after the backward, over the CNN's gradient range of the arena): found_inf |= any non-finite entry;
 *     found_inf: the range is zeroed (this accumulation window's CNN update is dropped; AdamW's moments stay finite), target
 *     halves (>= 1), skipped += 1; otherwise 2000 clean backwards in a row double the target again (<= target_max).
 *     DEVIATION from GradScaler, by design of this guard: it runs once per MICRO-step and covers the landmark CNN's range only --
 *     (a) with acc_step > 1 an overflow zeroes what earlier micro-steps of the window accumulated there and later ones still add
 *     to it, so the optimizer then applies a partial CNN gradient instead of skipping; (b) the optimizer step itself is never
 *     skipped: AdamW still decays the CNN's weights and moments, and the ViT / margin-head ranges (computed in bf16 with fp32
 *     accumulation, which cannot overflow where fp16 does) update normally, whereas the reference's scaler.step skips the whole
 *     step for every parameter (train_largescale.py:878-880); (c) in data-parallel runs only the overflowing rank zeroes its share
 *     before the all-reduce.  An overflow has not been observed in any test or benchmark run (state[4] == 0 throughout).
 * lafs_cnn_cast_pad_f16: dst(fp16)[r, c] = src(f32)[r, c] * scale[0], pad columns [cols, ld) zero.
 * lafs_cnn_cast_f16_f32: dst(f32)[i] = src(fp16)[i]. */
int lafs_cnn_grad_scale(const float* g, int64_t n, float target, float* scale, int has_state, hipStream_t stream);
int lafs_cnn_grad_guard(float* grad, int64_t n, float* state, float target_max, hipStream_t stream);
int lafs_cnn_cast_pad_f16(const float* src, int rows, int cols, void* dst, int ld, const float* scale, hipStream_t stream);
int lafs_cnn_cast_f16_f32(const void* src, float* dst, int64_t n, hipStream_t stream);
/* Depthwise k x k convolution (k in {3,5}, stride in {1,2}, pad (k-1)/2, no bias) on NHWC bf16.  w = tap-major fp32 image [k*k][ld] of
 * the module's [C][1][k][k] tensor (lafs_cnn_dw_layout_table: table[8 e ..] = {src offset, C, k*k, dst offset, ld}, 256 elements per
 * workgroup); forward; backward = dx(bf16) and dw(f32, += into a tap-major image [k*k][ld] that lafs_cnn_unpad_add_table folds,
 * transposed, into the arena). */
int lafs_cnn_dw_layout_table(const float* master, float* dst, const int64_t* table, const int32_t* starts, int n_entries, int n_blocks,
                             hipStream_t stream);
int lafs_cnn_dwconv_train_fwd(const void* x, const float* w, int N, int H, int W, int ld, int C, int k, int stride, void* y,
                              hipStream_t stream);
int lafs_cnn_dwconv_train_bwd(const void* x, const void* dy, const float* w, int N, int H, int W, int ld, int C, int k, int stride,
                              void* dx, float* dw, hipStream_t stream);
/* out(bf16)[n, c] = mean over the HW rows of image n (squeeze / final average pool), one workgroup per image. */
int lafs_cnn_pool_train(const void* x, int N, int HW, int ld, void* out, int ldo, hipStream_t stream);
/* Squeeze-excite: out = act(z gate[n, c]) (z kept); backward: ds = dout act'(z gate), dz = ds gate, dgate(f32)[n, c] = sum_p ds z. */
int lafs_cnn_scale_act_out(const void* z, const void* s, int lds_, int N, int HW, int ld, int act, void* out, hipStream_t stream);
int lafs_cnn_se_bwd(const void* dout, const void* z, const void* gate, int ldg, int N, int HW, int ld, int act, void* dz, float* dgate,
                    int lddg, hipStream_t stream);
/* out(bf16)[i] = dy[i] act'(.) from the POST-activation value y (relu, h-sigmoid: the squeeze-excite FCs); dy f32 or bf16. */
int lafs_cnn_act_bwd_post(const float* dy_f32, const void* dy_bf16, const void* y, int64_t n, int act, void* out, hipStream_t stream);
/* dx[n, p, c] = dfeat[n, c] / HW: backward of the final average pool. */
int lafs_cnn_pool_bwd(const void* dfeat, int ldf, int N, int HW, int ld, void* dx, hipStream_t stream);
/* Padded bf16 operand copies of fp32 arena tensors, ONE launch: table(i64, device)[8 e ..] = {src offset, rows, cols, dst offset,
 * dst ld, transpose, padded rows, 0}; starts(i32)[e] = first workgroup of entry e (1024 destination elements per workgroup). */
int lafs_cnn_pad_cast_table(const float* master, void* dst, const int64_t* table, const int32_t* starts, int n_entries, int n_blocks,
                            hipStream_t stream);
/* Padded fp32 weight gradients folded into the arena, ONE launch: table[8 e ..] = {src offset, rows, cols, src ld, grad offset,
 * transpose, ...}; 256 elements per workgroup.  transpose: the source image is [cols][ld] (tap-major depthwise gradients). */
int lafs_cnn_unpad_add_table(const float* padded, float* grad, const int64_t* table, const int32_t* starts, int n_entries, int n_blocks,
                             const float* grad_scale, hipStream_t stream);
/* Backward of theta = (t - min) / (max - min) * 111 per image (ViT_face.py:698-706), incl. the paths through min and max. */
int lafs_landmark_theta_bwd(const float* t, const float* dtheta, int B, int n, float* dt, hipStream_t stream);

/* Landmark post-processing (face_pre_pro/ViT_face.py:1347-1378, 698-706): t f32 [B, 2*n_full] raw regressor output ->
 * theta f32 [B, n_out, 2] pixels:  theta = (t - min_b)/(max_b - min_b)*111  (+ noise_scale * noise[B, n_full, 2], the
 * N(0,1)*5 px jitter), landmark k of the output = landmark sel[b,k] of the input (random choice with replacement) or k
 * when sel is NULL. */
int lafs_landmark_theta(const float* t, int B, int n_full, const float* noise, float noise_scale, const int32_t* sel, int n_out,
                        float* theta, hipStream_t stream);
/* Landmark patch gather (face_pre_pro/ViT_face.py:1615-1656): img f32 [B,3,S,S], theta f32 [B,n,2] (x,y pixels) ->
 * mosaic f32 [B,3,8r,8r], r = sqrt(n). */
int lafs_patch_gather_fwd(const float* img, const float* theta, int B, int S, int n, float* mosaic, hipStream_t stream);
/* dtheta(f32) [B,n,2] from dmosaic; dimg (optional, pre-zeroed) accumulates the image gradient. */
int lafs_patch_gather_bwd(const float* img, const float* theta, const float* dmosaic, int B, int S, int n,
                          float* dtheta, float* dimg, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LAFS_HIP_H */
