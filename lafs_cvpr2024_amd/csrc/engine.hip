// Host-side transformer engine: sequences the HIP kernels of one pre-LN ViT trunk (all blocks) over a packed
// token batch, forward and backward, with every activation in a caller-provided workspace.  One C call per
// trunk pass -> no per-op Python/dispatcher overhead, and the whole pass is hipGraph-capturable (no allocation,
// no synchronisation, only kernel launches on `stream`).
//
// Mirrors Block.forward (vision_transformer.py:107-113) / Residual_droppath(PreNorm(.)) (face_pre_pro/ViT_face.py:106-120):
//   x1 = x0 + s_a * proj(attn(LN1(x0)));   x0' = x1 + s_m * fc2(gelu(fc1(LN2(x1))))
// The residual stream is fp32; GEMM operands are bf16 (fp32 accumulate).
#include <algorithm>
#include <vector>
#include <stdlib.h>
#include "common.hpp"
#include "lafs_hip.h"
#include "ctx.hpp"

namespace {

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct LayerBuf {
  float* x0; float* st1; bf16_t* h1; bf16_t* qkv; float* lse; bf16_t* o; float* x1; float* st2; bf16_t* h2; bf16_t* u; bf16_t* a;
};
struct Scratch {             // operands of the weight gradients: two slots used by layer parity (the wgrad stream may lag one layer behind),
  std::vector<bf16_t*> gbm, gba, du, dqkv;   // or one slot per layer when the weight gradients are deferred (lafs_trunk_desc::wgrad_defer)
  bf16_t* dh; bf16_t* d_o;
};
inline int wg_slot(const lafs_trunk_desc* d, int l) { return d->wgrad_defer ? l : (l & 1); }
struct Carve {
  std::vector<LayerBuf> layers;   // depth entries when saving, 1 otherwise (reused)
  std::vector<float*> ln_part;    // [layer][norm 1 | 2][row chain 0..3]: per-workgroup gamma / beta sums of the LayerNorm backward
  size_t ln_part_floats;          // (lafs_layernorm_bwd part_out; lafs_layernorm_bwd_fold adds them in a fixed order)
  float* xalt;                    // ping-pong residual buffer for the no-save path
  Scratch s;
  void* wg_ws; size_t wg_bytes;   // slice partials of the grouped weight-gradient launch (lafs_wgrad_group)
  size_t bytes;
};

// the four weight gradients of one block as lafs_wgrad_group items (pointers filled in by the caller)
void block_wgrad_shapes(const lafs_trunk_desc* d, lafs_wgrad_item (&it)[4]) {
  const int D = d->dim, I = d->inner, M = d->mlp;
  for (auto& x : it) x = lafs_wgrad_item{};
  it[0].N1 = D; it[0].N2 = M; it[0].lda = D; it[0].ldb = M; it[0].ldc = M;              // fc2:  dW = gbm^T a
  it[1].N1 = M; it[1].N2 = D; it[1].lda = M; it[1].ldb = D; it[1].ldc = D;              // fc1:  dW = du^T h2
  it[2].N1 = D; it[2].N2 = I; it[2].lda = D; it[2].ldb = I; it[2].ldc = I;              // proj: dW = gba^T o
  it[3].N1 = 3 * I; it[3].N2 = D; it[3].lda = 3 * I; it[3].ldb = D; it[3].ldc = D;      // qkv:  dW = dqkv^T h1
}

Carve carve(const lafs_trunk_desc* d, void* ws, int save) {
  Carve c;
  unsigned char* base = reinterpret_cast<unsigned char*>(ws);
  size_t off = 0;
  const size_t T = (size_t)d->n_tok, D = d->dim, I = d->inner, M = d->mlp, H = d->heads;
  auto take = [&](size_t bytes) { unsigned char* p = base ? base + off : nullptr; off += al(bytes); return p; };
  const int nl = save ? d->depth : 1;
  c.layers.resize(nl);
  for (int l = 0; l < nl; ++l) {
    LayerBuf& b = c.layers[l];
    b.x0 = (l == 0) ? nullptr : (float*)take(T * D * 4);     // layer 0 reads the caller's x_in
    b.st1 = (float*)take(T * 2 * 4);
    b.h1 = (bf16_t*)take(T * D * 2);
    b.qkv = (bf16_t*)take(T * 3 * I * 2);
    b.lse = (float*)take(T * H * 4);
    b.o = (bf16_t*)take(T * I * 2);
    b.x1 = (float*)take(T * D * 4);
    b.st2 = (float*)take(T * 2 * 4);
    b.h2 = (bf16_t*)take(T * D * 2);
    b.u = (bf16_t*)take(T * M * 2);
    b.a = (bf16_t*)take(T * M * 2);
  }
  c.xalt = save ? nullptr : (float*)take(T * D * 4);
  if (save) {
    const int nslot = d->wgrad_defer ? d->depth : 2;
    c.s.gbm.resize(nslot); c.s.gba.resize(nslot); c.s.du.resize(nslot); c.s.dqkv.resize(nslot);
    for (int q = 0; q < nslot; ++q) {
      c.s.gbm[q] = (bf16_t*)take(T * D * 2);
      c.s.gba[q] = (bf16_t*)take(T * D * 2);
      c.s.du[q] = (bf16_t*)take(T * M * 2);
      c.s.dqkv[q] = (bf16_t*)take(T * 3 * I * 2);
    }
    c.s.dh = (bf16_t*)take(T * D * 2);
    c.s.d_o = (bf16_t*)take(T * I * 2);
    c.ln_part_floats = (size_t)std::max(lafs_layernorm_bwd_parts(d->n_tok, d->dim), lafs_mlp_fused_ln_parts(d->n_tok)) * 2 * D;
    // (four full-size slot buffers per norm whatever the number of row chains in use -- 75 MB of 7.3 GB at C2: the chain count is an
    // option of the context, which may change on a live engine after its workspace was sized; sizing by the chains in use, as the
    // round-5 advisor suggested, would make that a silent overflow)
    c.ln_part.resize((size_t)d->depth * 2 * 4);
    for (auto& q : c.ln_part) q = (float*)take(c.ln_part_floats * 4);
    lafs_wgrad_item it[4];
    block_wgrad_shapes(d, it);
    // (the single-stream backward uses the whole chip: size for whichever plan needs more)
    const int64_t wb = std::max(lafs_wgrad_group_workspace_bytes(it, 4, d->n_tok, d->wgrad_workgroups),
                                lafs_wgrad_group_workspace_bytes(it, 4, d->n_tok, 0));
    c.wg_bytes = wb > 0 ? (size_t)wb : 0;
    c.wg_ws = take(c.wg_bytes > 0 ? c.wg_bytes : 256);
  } else {
    c.s = Scratch{};
    c.ln_part_floats = 0;
    c.wg_ws = nullptr; c.wg_bytes = 0;
  }
  c.bytes = off;
  return c;
}

int check_desc(const lafs_trunk_desc* d) {
  LAFS_CHECK_ARG(d != nullptr, "null descriptor");
  LAFS_CHECK_ARG(d->dim > 0 && d->dim % 64 == 0 && d->mlp % 64 == 0 && d->inner == d->heads * 64, "dims must be multiples of 64");
  LAFS_CHECK_ARG(d->depth > 0 && d->n_tok > 0 && d->n_seq > 0 && d->max_len > 0 && d->max_len <= 256, "bad geometry");
  LAFS_CHECK_ARG(d->cu_seqlens && d->row2seq && d->master && d->shadow && d->blocks, "null pointer in descriptor");
  LAFS_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "dropout_p must be in [0, 1)");
  LAFS_CHECK_ARG(d->n_groups >= 0 && d->n_groups <= 4, "at most 4 sequence groups");
  if (d->n_groups > 0) {
    int ns = 0;
    for (int gi = 0; gi < d->n_groups; ++gi) {
      LAFS_CHECK_ARG(d->group_n_seq[gi] > 0 && d->group_max_len[gi] > 0 && d->group_max_len[gi] <= d->max_len, "bad sequence group");
      ns += d->group_n_seq[gi];
    }
    LAFS_CHECK_ARG(ns == d->n_seq, "sequence groups must cover n_seq");
  }
  return LAFS_OK;
}

#define RUN(call)                        \
  do {                                   \
    const int rc_ = (call);              \
    if (rc_ != LAFS_OK) return rc_;      \
  } while (0)

int gemm(const lafs_ctx* cx, const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi, void* C, int ldc, const float* bias,
         hipStream_t s, void* C2 = nullptr, int ldc2 = 0, const float* resid = nullptr, int ldr = 0,
         const float* seq_scale = nullptr, const int32_t* row2seq = nullptr, const void* aux = nullptr, int ldaux = 0,
         float drop_p = 0.f, uint32_t drop_seed = 0, int act = 0, const float* drop_step = nullptr, int drop_row0 = 0) {
  lafs_gemm_nt_args g = {};
  g.ctx = cx;
  g.drop_p = drop_p; g.drop_seed = drop_seed; g.act = act; g.drop_step = drop_step; g.drop_row0 = drop_row0;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.epilogue = epi;
  g.C = C; g.ldc = ldc; g.C2 = C2; g.ldc2 = ldc2; g.bias = bias; g.resid = resid; g.ldr = ldr;
  g.seq_scale = seq_scale; g.row2seq = row2seq; g.aux = aux; g.ldaux = ldaux; g.splits = 1;
  return lafs_gemm_nt(&g, s);
}

// The block's MLP as one launch (csrc/mlp_fused.hip) where the context asks for it and the geometry allows: bit 1 forward-only pass,
// 2 saving forward, 4 backward input gradients
bool mlp_fused_on(const lafs_trunk_desc* d, int bit, int rows) {
  return (lafs_ctx_opt(d->ctx, LAFS_OPT_MLP_FUSED) & bit) != 0 && d->dropout_p == 0.f && lafs_mlp_fused_supported(d->dim, d->mlp, rows) != 0;
}

}  // namespace

// Side streams for the attention launches of the second and later crop-resolution groups and for the row chains: the 197-token and
// the 37-token launch of a layer are independent (both read the qkv GEMM's output, both feed the projection) and latency-bound on
// their own (profiles/round2_attention_pmc.txt: waves waiting 53-67 % of the time), so they run beside each other.  The streams and
// events belong to the descriptor's lafs_ctx (created with it, before anything is captured); without a context -- or with
// LAFS_OPT_SIDE_STREAMS = 0 -- everything stays on `stream`.
static inline lafs_ctx* side_ctx(const lafs_trunk_desc* d) {
  lafs_ctx* c = d->ctx;
  return (c != nullptr && c->streams_ok && c->opt[LAFS_OPT_SIDE_STREAMS] != 0) ? c : nullptr;
}
// stream of group gi's attention launch; call attn_fork before the first launch and attn_join after the last one
static hipStream_t attn_stream_of(const lafs_trunk_desc* d, int gi, hipStream_t stream) {
  lafs_ctx* c = side_ctx(d);
  return (gi > 0 && c != nullptr) ? c->side[0] : stream;
}
#define HIP_TRY(call)                                                                         \
  do {                                                                                        \
    const hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                                   \
      lafs_set_error("%s:%d: %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));      \
      return (int)e_;                                                                         \
    }                                                                                         \
  } while (0)
static int attn_fork(const lafs_trunk_desc* d, hipStream_t stream, int n = 2) {        // n streams in all: `stream`, side[0..2]
  lafs_ctx* a = side_ctx(d);
  if (a == nullptr) return LAFS_OK;
  HIP_TRY(hipEventRecord(a->fork, stream));
  for (int i = 0; i + 1 < n; ++i) HIP_TRY(hipStreamWaitEvent(a->side[i], a->fork, 0));
  return LAFS_OK;
}
// (always reached once a fork has been issued -- also on the error path of the forked work: side streams left forked inside a
// hipGraph capture would make the capture fail later with an unrelated error)
static int attn_join(const lafs_trunk_desc* d, hipStream_t stream, int n = 2) {
  lafs_ctx* a = side_ctx(d);
  if (a == nullptr) return LAFS_OK;
  for (int i = 0; i + 1 < n; ++i) {
    HIP_TRY(hipEventRecord(a->join[i], a->side[i]));
    HIP_TRY(hipStreamWaitEvent(stream, a->join[i], 0));
  }
  return LAFS_OK;
}
// forked region: run `body`, join in any case, report the first failure
#define FORKED(stream, n, body)                 \
  do {                                          \
    RUN(attn_fork(d, stream, n));               \
    int rc_f = LAFS_OK;                         \
    do { body } while (0);                      \
    const int rc_j = attn_join(d, stream, n);   \
    if (rc_f != LAFS_OK) return rc_f;           \
    if (rc_j != LAFS_OK) return rc_j;           \
  } while (0)
#define TRY_F(call)                             \
  {                                             \
    rc_f = (call);                              \
    if (rc_f != LAFS_OK) break;                 \
  }
// Row ranges of a trunk pass: one per crop-resolution group (2) or per half group (4, cut at a sequence boundary) when there are
// two groups of full-length sequences (element-dropout masks are indexed by absolute rows: drop_row0); else one range.
struct RowRange { int r0, R, gi, seq_lo, nseq; hipStream_t st; };
static int row_ranges(const lafs_trunk_desc* d, hipStream_t stream, RowRange (&rr)[4]) {
  lafs_ctx* a = side_ctx(d);
  rr[0] = {0, d->n_tok, 0, 0, d->n_seq, stream};
  const int chains = a != nullptr ? a->opt[LAFS_OPT_ROW_CHAINS] : 1;
  if (a == nullptr || chains < 2 || d->n_groups != 2) return 1;
  const int T0 = d->group_n_seq[0] * d->group_max_len[0], T1 = d->group_n_seq[1] * d->group_max_len[1];
  if (T0 + T1 != d->n_tok || T0 < 4096 || T1 < 4096) return 1;
  hipStream_t st[4] = {stream, a->side[0], a->side[1], a->side[2]};
  int n = 0;
  const int parts = (chains == 4 && d->group_n_seq[0] >= 2 && d->group_n_seq[1] >= 2) ? 2 : 1;
  int row = 0, seq = 0;
  for (int gi = 0; gi < 2; ++gi) {
    const int ns = d->group_n_seq[gi], len = d->group_max_len[gi];
    for (int h = 0; h < parts; ++h) {
      const int q0 = ns * h / parts, q1 = ns * (h + 1) / parts;
      rr[n] = {row + q0 * len, (q1 - q0) * len, gi, seq + q0, q1 - q0, st[n]};
      ++n;
    }
    row += ns * len; seq += ns;
  }
  return n;
}

extern "C" int64_t lafs_trunk_workspace_bytes(const lafs_trunk_desc* d, int save_for_backward) {
  if (check_desc(d) != LAFS_OK) return -1;
  return (int64_t)carve(d, nullptr, save_for_backward).bytes;
}

extern "C" int lafs_trunk_row_ranges(const lafs_trunk_desc* d) {
  if (check_desc(d) != LAFS_OK) return -1;
  RowRange rr[4];
  return row_ranges(d, nullptr, rr);
}

extern "C" int lafs_trunk_forward(const lafs_trunk_desc* d, const float* x_in, float* x_out, void* workspace,
                                  int save_for_backward, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  RUN(check_desc(d));
  LAFS_CHECK_ARG(x_in && x_out && workspace, "null buffer");
  const Carve c = carve(d, workspace, save_for_backward);
  const lafs_ctx* cx = d->ctx;
  const int T = d->n_tok, D = d->dim, I = d->inner, M = d->mlp;
  const bf16_t* sh = reinterpret_cast<const bf16_t*>(d->shadow);
  // Nothing in the forward mixes token rows of different sequences: with two crop-resolution groups of full-length sequences
  // (element-dropout masks are indexed by absolute rows) the groups' rows run as two independent chains of
  // launches over row sub-ranges of the same buffers, the second on the attention side stream -- every kernel of the chain is
  // latency-bound to some degree, and two chains side by side fill each other's gaps.  LAFS_ROW_CHAINS=0: one chain, 4: half groups.
  RowRange rr[4];
  const int n_rr = row_ranges(d, stream, rr);
  // rows [r0, r0 + R) = the sequences of groups [g_lo, g_hi) (seq_lo = index of their first sequence) on stream st
  // (nseq >= 0: ONE attention launch over nseq sequences of group g_lo -- a row range of a split pass)
  // residual stream of every layer: input / output buffer (the same for every row range)
  std::vector<const float*> lay_in(d->depth);
  std::vector<float*> lay_out(d->depth);
  {
    const float* cur = x_in;
    for (int l = 0; l < d->depth; ++l) {
      float* nxt;
      if (l == d->depth - 1) nxt = x_out;
      else if (save_for_backward) nxt = c.layers[l + 1].x0;
      else nxt = (cur == c.xalt) ? x_out : c.xalt;            // ping-pong; never aliases x_in
      lay_in[l] = cur; lay_out[l] = nxt; cur = nxt;
    }
  }
  // LAFS_OPT_MLP_FUSED bit 64: the fused MLP of block l also writes LayerNorm 1 of block l + 1 (not with the merged launch of bit 32, whose
  // row count differs from the attention branch's)
  const bool next_ln_opt = (lafs_ctx_opt(d->ctx, LAFS_OPT_MLP_FUSED) & (64 | 32)) == 64;
  // layers [l_lo, l_hi); parts: 1 = the attention branch (LayerNorm 1 .. projection + residual), 2 = the MLP branch
  auto chain = [&](int r0, int R, int g_lo, int g_hi, int seq_lo, int nseq, hipStream_t st, bool attn_two_streams, int l_lo, int l_hi,
                   int parts) -> int {
    for (int l = l_lo; l < l_hi; ++l) {
      const lafs_block_offsets& o = d->blocks[l];
      const LayerBuf& b = c.layers[save_for_backward ? l : 0];
      const float* sa = d->drop_scales ? d->drop_scales + ((size_t)l * 2 + 0) * d->n_seq : nullptr;
      const float* sm = d->drop_scales ? d->drop_scales + ((size_t)l * 2 + 1) * d->n_seq : nullptr;
      const int32_t* r2s = d->row2seq ? d->row2seq + r0 : nullptr;
      const float* cur = lay_in[l];
      float* nxt = lay_out[l];
      const size_t rD = (size_t)r0 * D, rI = (size_t)r0 * I, rM = (size_t)r0 * M;
      const float dp = d->dropout_p;
      const uint32_t ds = d->dropout_seed + 3u * (uint32_t)l;                 // sites: +0 to_out, +1 GELU, +2 fc2
      // LayerNorm 1 of this layer was produced by the previous layer's fused MLP (its epilogue holds the finished rows in registers)
      const bool ln1_done = l > 0 && next_ln_opt && mlp_fused_on(d, save_for_backward ? 2 : 1, R);
      const bool mlp_one = mlp_fused_on(d, save_for_backward ? 2 : 1, R);
      const bool ln_inside = mlp_one && mlp_fused_on(d, 8, R);          // LayerNorm 2 as the fused kernel's prologue (no launch, no h2 round trip)
      // bits 128 (forward-only pass) / 256 (saving pass): the attention branch's projection + residual as the fused kernel's prologue
      const bool prj = parts == 3 && ln_inside && I == D && mlp_fused_on(d, save_for_backward ? 256 : 128, R);
      if (parts & 1) {
      if (!ln1_done)
      RUN(lafs_layernorm_fwd(cur + rD, D, d->master + o.ln1_g, d->master + o.ln1_b, d->ln_eps, b.h1 + rD, D, nullptr, 0, b.st1 + 2 * (size_t)r0,
                             R, D, st));
      RUN(gemm(cx, b.h1 + rD, D, sh + o.w_qkv, D, R, 3 * I, D, LAFS_EPI_BF16, b.qkv + 3 * rI, 3 * I, o.b_qkv >= 0 ? d->master + o.b_qkv : nullptr, st));
      if (g_hi - g_lo > 1 || d->n_groups > 1) {               // one launch per crop resolution, each with its own tile shape
        int s0 = seq_lo;                                      // (the attention kernels address tokens through cu_seqlens: base pointers)
        if (attn_two_streams) {
          FORKED(st, 2, for (int gi = g_lo; gi < g_hi; ++gi) {
            TRY_F(lafs_attention_fwd(b.qkv, 3 * I, d->cu_seqlens + s0, d->group_n_seq[gi], d->group_max_len[gi], d->heads, d->attn_scale, b.o, I,
                                     b.lse, attn_stream_of(d, gi - g_lo, st)));
            s0 += d->group_n_seq[gi];
          });
        } else {
          for (int gi = g_lo; gi < g_hi; ++gi) {
            RUN(lafs_attention_fwd(b.qkv, 3 * I, d->cu_seqlens + s0, nseq >= 0 ? nseq : d->group_n_seq[gi], d->group_max_len[gi], d->heads,
                                   d->attn_scale, b.o, I, b.lse, st));
            s0 += d->group_n_seq[gi];
          }
        }
      } else {
        RUN(lafs_attention_fwd(b.qkv, 3 * I, d->cu_seqlens, d->n_seq, d->max_len, d->heads, d->attn_scale, b.o, I, b.lse, st));
      }
      if (!prj)
      RUN(gemm(cx, b.o + rI, I, sh + o.w_proj, I, R, D, I, LAFS_EPI_RESID_F32, b.x1 + rD, D, d->master + o.b_proj, st, nullptr, 0, cur + rD, D, sa,
               r2s, nullptr, 0, dp, ds + 0, 0, d->dropout_step, r0));
      }
      if (!(parts & 2)) continue;
      if (!ln_inside)
      RUN(lafs_layernorm_fwd(b.x1 + rD, D, d->master + o.ln2_g, d->master + o.ln2_b, d->ln_eps, b.h2 + rD, D, nullptr, 0, b.st2 + 2 * (size_t)r0,
                             R, D, st));
      // a forward-only pass (teacher) never reads the pre-activation u: skip its store (77 MB per layer at C2).  A saving pass
      // stores gelu'(u) in its place (LAFS_GELU_SAVE_GRAD): that is all the backward needs of u, and the GELU' input gradient
      // becomes one multiply per value
      if (mlp_one) {                                           // fc1 -> GELU -> fc2 -> residual in one launch, the hidden tile on chip
        lafs_mlp_args m = {};
        m.X = b.h2 + rD; m.ldx = D; m.Wa = sh + o.w_fc1; m.ldwa = D; m.Wb = sh + o.w_fc2; m.ldwb = M; m.M = R; m.H = M;
        m.mode = save_for_backward ? LAFS_MLP_FWD_SAVE : LAFS_MLP_FWD;
        m.bias_a = d->master + o.b_fc1; m.bias_b = d->master + o.b_fc2; m.resid = b.x1 + rD; m.ldr = D; m.seq_scale = sm; m.row2seq = r2s;
        m.out = nxt + rD; m.ldo = D;
        if (save_for_backward) { m.save_grad = b.u + rM; m.ldsg = M; m.save_act = b.a + rM; m.ldsa = M; }
        if (ln_inside) {
          m.X = nullptr; m.ln_gamma = d->master + o.ln2_g; m.ln_beta = d->master + o.ln2_b; m.ln_eps = d->ln_eps;
          if (save_for_backward) { m.ln_stats = b.st2 + 2 * (size_t)r0; m.ln_out = b.h2 + rD; m.ldln = D; }
        }
        if (prj) {                                             // x1 = cur + sa * (o Wproj^T + b) is computed (and stored to b.x1) by this launch
          m.proj_x = b.o + rI; m.ldpx = I; m.proj_w = sh + o.w_proj; m.ldpw = I; m.proj_bias = d->master + o.b_proj;
          m.proj_resid = cur + rD; m.ldpr = D; m.proj_scale = sa;
        }
        if (l + 1 < d->depth && next_ln_opt) {        // the next block's LayerNorm 1, from the rows in this launch's registers
          const lafs_block_offsets& on = d->blocks[l + 1];
          const LayerBuf& bn = c.layers[save_for_backward ? l + 1 : 0];
          m.next_ln_gamma = d->master + on.ln1_g; m.next_ln_beta = d->master + on.ln1_b; m.next_ln_eps = d->ln_eps;
          m.next_ln_out = bn.h1 + rD; m.ldnln_next = D;
          m.next_ln_stats = save_for_backward ? bn.st1 + 2 * (size_t)r0 : nullptr;
        }
        m.ctx = cx;
        RUN(lafs_mlp_fused(&m, st));
      } else {
      RUN(gemm(cx, b.h2 + rD, D, sh + o.w_fc1, D, R, M, D, LAFS_EPI_BF16_GELU, save_for_backward ? b.u + rM : nullptr, M, d->master + o.b_fc1, st,
               b.a + rM, M, nullptr, 0, nullptr, nullptr, nullptr, 0, dp, ds + 1, LAFS_GELU_SAVE_GRAD, d->dropout_step, r0));
      RUN(gemm(cx, b.a + rM, M, sh + o.w_fc2, M, R, D, M, LAFS_EPI_RESID_F32, nxt + rD, D, d->master + o.b_fc2, st, nullptr, 0, b.x1 + rD, D, sm,
               r2s, nullptr, 0, dp, ds + 2, 0, d->dropout_step, r0));
      }
    }
    return LAFS_OK;
  };
  // LAFS_OPT_MLP_FUSED bit 32 (lab): the row chains meet in front of every MLP, which then runs as ONE launch over all rows -- whole
  // rounds of the chip plus a round of 64-row units instead of a round per chain (csrc/mlp_fused.hip)
  const bool merge_mlp = n_rr > 1 && (lafs_ctx_opt(d->ctx, LAFS_OPT_MLP_FUSED) & 32) != 0 && mlp_fused_on(d, save_for_backward ? 2 : 1, T);
  if (merge_mlp) {
    for (int l = 0; l < d->depth; ++l) {
      FORKED(stream, n_rr, for (int i = 0; i < n_rr; ++i)
        TRY_F(chain(rr[i].r0, rr[i].R, rr[i].gi, rr[i].gi + 1, rr[i].seq_lo, rr[i].nseq, rr[i].st, false, l, l + 1, 1)););
      RUN(chain(0, T, 0, d->n_groups, 0, -1, stream, false, l, l + 1, 2));
    }
  } else if (n_rr > 1) {                                    // the side streams join behind everything `stream` has enqueued so far
    FORKED(stream, n_rr, for (int i = 0; i < n_rr; ++i)
      TRY_F(chain(rr[i].r0, rr[i].R, rr[i].gi, rr[i].gi + 1, rr[i].seq_lo, rr[i].nseq, rr[i].st, false, 0, d->depth, 3)););
  } else {
    RUN(chain(0, T, 0, d->n_groups, 0, -1, stream, d->n_groups > 1, 0, d->depth, 3));
  }
  return LAFS_OK;
}

// The block's four weight gradients: ONE grouped launch, once all their operands exist.  Its 48 (ViT-S) output tiles x 5 token
// slices fill the chip together: 4x fewer slices -> 4x less partial-sum traffic than four separate launches (csrc/wgrad.hip)
static int block_wgrad(const lafs_trunk_desc* d, const Carve& c, int l, int max_wg, hipStream_t st) {
  const lafs_block_offsets& o = d->blocks[l];
  const LayerBuf& b = c.layers[l];
  const Scratch& s = c.s;
  const int p = wg_slot(d, l);
  float* gr = d->grad;
  lafs_wgrad_item it[4];
  block_wgrad_shapes(d, it);
  it[0].A = s.gbm[p]; it[0].B = b.a; it[0].C = gr + o.w_fc2; it[0].colsum_a = gr + o.b_fc2;
  it[1].A = s.du[p]; it[1].B = b.h2; it[1].C = gr + o.w_fc1; it[1].colsum_a = gr + o.b_fc1;
  it[2].A = s.gba[p]; it[2].B = b.o; it[2].C = gr + o.w_proj; it[2].colsum_a = gr + o.b_proj;
  it[3].A = s.dqkv[p]; it[3].B = b.h1; it[3].C = gr + o.w_qkv; it[3].colsum_a = o.b_qkv >= 0 ? gr + o.b_qkv : nullptr;
  for (auto& x : it) x.accumulate = d->wgrad_overwrite ? 0 : 1;
  return lafs_wgrad_group(it, 4, d->n_tok, max_wg, c.wg_ws, (int64_t)c.wg_bytes, st);
}

extern "C" int lafs_trunk_backward(const lafs_trunk_desc* d, const float* x_in, float* g, void* workspace, int layer_hi,
                                   int layer_lo, hipStream_t wgrad_stream, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  RUN(check_desc(d));
  LAFS_CHECK_ARG(x_in && g && workspace && d->shadow_t && d->grad, "null buffer");
  LAFS_CHECK_ARG(0 <= layer_lo && layer_lo < layer_hi && layer_hi <= d->depth, "bad layer range");
  const Carve c = carve(d, workspace, 1);
  const lafs_ctx* cx = d->ctx;
  const int T = d->n_tok, D = d->dim, I = d->inner, M = d->mlp;
  const bf16_t* sht = reinterpret_cast<const bf16_t*>(d->shadow_t);
  float* gr = d->grad;
  const Scratch& s = c.s;
  const bool defer = d->wgrad_defer != 0;     // no weight-gradient launches here: lafs_trunk_wgrad issues them later from the per-layer slots
  // (the two-stream protocol takes its events from the context's pool: without a context the weight gradients stay on `stream`)
  const bool two = !defer && (wgrad_stream != nullptr) && (wgrad_stream != stream) && d->ctx != nullptr && d->ctx->streams_ok;
  hipStream_t s2 = two ? wgrad_stream : stream;
  const int nl = layer_hi - layer_lo;
  RowRange rr[4];
  const int n_rr = row_ranges(d, stream, rr);
  static std::vector<hipEvent_t> no_events;
  std::vector<hipEvent_t>& ev = two ? d->ctx->pool : no_events;
  LAFS_CHECK_ARG(!two || ev.size() >= (size_t)2 * nl + 1, "the context's event pool is too small for this layer range");
  int evi = 0;
  bool ev_failed = false;                  // a failed event call of the two-stream protocol (reported at the end of the call)
  auto fork = [&]() {                      // work enqueued on s2 after this sees everything enqueued on `stream` so far
    if (!two) return;
    hipEvent_t e = ev[evi++];
    if (hipEventRecord(e, stream) != hipSuccess || hipStreamWaitEvent(s2, e, 0) != hipSuccess) ev_failed = true;
  };
  std::vector<hipEvent_t> done(d->depth, nullptr);
  auto scale = [&](int l, int br) { return d->drop_scales ? d->drop_scales + ((size_t)l * 2 + br) * d->n_seq : nullptr; };
  const float dp = d->dropout_p;
  auto dseed = [&](int l, int site) { return d->dropout_seed + 3u * (uint32_t)l + (uint32_t)site; };
  RUN(lafs_scale_cast_bf16(g, D, s.gbm[wg_slot(d, layer_hi - 1)], D, scale(layer_hi - 1, 1), d->row2seq, T, D, dp, dseed(layer_hi - 1, 2),
                           d->dropout_step, 0, stream));
  // rows [r0, r0 + R) of layer l from the GELU' input gradient to the attention backward (groups [g_lo, g_hi), first sequence
  // seq_lo; nseq >= 0: one attention launch over nseq sequences of group g_lo) on stream st
  auto part1 = [&](int l, int ci, int r0, int R, int g_lo, int g_hi, int seq_lo, int nseq, hipStream_t st, bool attn_two_streams) -> int {
    const lafs_block_offsets& o = d->blocks[l];
    const LayerBuf& b = c.layers[l];
    const int p = wg_slot(d, l);
    const size_t rD = (size_t)r0 * D, rI = (size_t)r0 * I, rM = (size_t)r0 * M;
    const int32_t* r2s = d->row2seq ? d->row2seq + r0 : nullptr;
    // ---- MLP branch ----
    // (LayerNorm 2's backward inside the same launch: its gamma / beta slots are numbered by workgroup, one launch of <= capacity units)
    const bool ln_inside = mlp_fused_on(d, 4, R) && mlp_fused_on(d, 16, R) &&
                           (size_t)lafs_mlp_fused_ln_parts(R) * 2 * D <= c.ln_part_floats;
    if (mlp_fused_on(d, 4, R)) {                               // GELU' input gradient -> fc1 input gradient in one launch (du written once)
      lafs_mlp_args m = {};
      m.X = s.gbm[p] + rD; m.ldx = D; m.Wa = sht + o.w_fc2_t; m.ldwa = D; m.Wb = sht + o.w_fc1_t; m.ldwb = M; m.M = R; m.H = M;
      m.mode = LAFS_MLP_BWD; m.out = s.dh + rD; m.ldo = D; m.save_grad = b.u + rM; m.ldsg = M; m.save_act = s.du[p] + rM; m.ldsa = M;
      m.ctx = cx;
      if (ln_inside) {
        m.resid = b.x1 + rD; m.ldr = D; m.ln_stats = b.st2 + 2 * (size_t)r0; m.ln_gamma = d->master + o.ln2_g;
        m.ln_g_io = g + rD; m.ldgio = D; m.ln_gb_out = s.gba[p] + rD; m.ldgb = D; m.seq_scale = scale(l, 0); m.row2seq = r2s;
        m.ln_part_out = c.ln_part[((size_t)l * 2 + 1) * 4 + ci];
      }
      RUN(lafs_mlp_fused(&m, st));
    } else {
    RUN(gemm(cx, s.gbm[p] + rD, D, sht + o.w_fc2_t, D, R, M, D, LAFS_EPI_DGELU_BF16, s.du[p] + rM, M, nullptr, st, nullptr, 0, nullptr, 0, nullptr,
             nullptr, b.u + rM, M, dp, dseed(l, 1), LAFS_GELU_SAVE_GRAD, d->dropout_step, r0));
    RUN(gemm(cx, s.du[p] + rM, M, sht + o.w_fc1_t, M, R, D, M, LAFS_EPI_BF16, s.dh + rD, D, nullptr, st));
    }
    if (!ln_inside)
    RUN(lafs_layernorm_bwd(s.dh + rD, D, nullptr, 0, b.x1 + rD, D, b.st2 + 2 * (size_t)r0, d->master + o.ln2_g, g + rD, D, 1, s.gba[p] + rD, D,
                           scale(l, 0), r2s, gr + o.ln2_g, gr + o.ln2_b, R, D, dp, dseed(l, 0), d->dropout_step, r0,
                           c.ln_part[((size_t)l * 2 + 1) * 4 + ci], st));
    // ---- attention branch ----
    RUN(gemm(cx, s.gba[p] + rD, D, sht + o.w_proj_t, D, R, I, D, LAFS_EPI_BF16, s.d_o + rI, I, nullptr, st));
    if (d->n_groups > 1) {
      int s0 = seq_lo;
      if (attn_two_streams) {
        FORKED(st, 2, for (int gi = g_lo; gi < g_hi; ++gi) {
          TRY_F(lafs_attention_bwd(b.qkv, 3 * I, b.o, I, s.d_o, I, b.lse, d->cu_seqlens + s0, d->group_n_seq[gi], d->group_max_len[gi], d->heads,
                                   d->attn_scale, s.dqkv[p], 3 * I, attn_stream_of(d, gi - g_lo, st)));
          s0 += d->group_n_seq[gi];
        });
      } else {
        for (int gi = g_lo; gi < g_hi; ++gi) {
          RUN(lafs_attention_bwd(b.qkv, 3 * I, b.o, I, s.d_o, I, b.lse, d->cu_seqlens + s0, nseq >= 0 ? nseq : d->group_n_seq[gi],
                                 d->group_max_len[gi], d->heads, d->attn_scale, s.dqkv[p], 3 * I, st));
          s0 += d->group_n_seq[gi];
        }
      }
    } else {
      RUN(lafs_attention_bwd(b.qkv, 3 * I, b.o, I, s.d_o, I, b.lse, d->cu_seqlens, d->n_seq, d->max_len, d->heads, d->attn_scale,
                             s.dqkv[p], 3 * I, st));
    }
    return LAFS_OK;
  };
  // ... and from the qkv input gradient to the LayerNorm backward that produces layer l-1's upstream gradient gbm[(l-1)&1]
  auto part2 = [&](int l, int ci, int r0, int R, hipStream_t st) -> int {
    const lafs_block_offsets& o = d->blocks[l];
    const LayerBuf& b = c.layers[l];
    const float* x0 = (l == 0) ? x_in : b.x0;
    const int p = wg_slot(d, l);
    const bool more = l > 0;
    const size_t rD = (size_t)r0 * D, rI = (size_t)r0 * I;
    const int32_t* r2s = d->row2seq ? d->row2seq + r0 : nullptr;
    RUN(gemm(cx, s.dqkv[p] + 3 * rI, 3 * I, sht + o.w_qkv_t, 3 * I, R, D, 3 * I, LAFS_EPI_BF16, s.dh + rD, D, nullptr, st));
    RUN(lafs_layernorm_bwd(s.dh + rD, D, nullptr, 0, x0 + rD, D, b.st1 + 2 * (size_t)r0, d->master + o.ln1_g, g + rD, D, 1,
                           more ? s.gbm[wg_slot(d, l - 1)] + rD : nullptr, D, more ? scale(l - 1, 1) : nullptr, r2s, gr + o.ln1_g, gr + o.ln1_b, R, D,
                           more ? dp : 0.f, more ? dseed(l - 1, 2) : 0u, d->dropout_step, r0, c.ln_part[((size_t)l * 2 + 0) * 4 + ci], st));
    return LAFS_OK;
  };
  // One forked section = the tail of layer l2 (part2) and the head of layer l1 = l2 - 1 (part1) for every row range: with two
  // crop-resolution groups of full-length sequences the row ranges run beside each other (second on the side stream), forked and
  // joined once per layer -- the pattern hipGraph captures; a chain that stays forked across layers and meets the weight-gradient
  // stream's events does not.  -1 = no such part.
  auto section = [&](int l2, int l1) -> int {
    if (n_rr > 1) {
      FORKED(stream, n_rr, for (int i = 0; i < n_rr; ++i) {
        if (l2 >= 0) TRY_F(part2(l2, i, rr[i].r0, rr[i].R, rr[i].st));
        if (l1 >= 0) TRY_F(part1(l1, i, rr[i].r0, rr[i].R, rr[i].gi, rr[i].gi + 1, rr[i].seq_lo, rr[i].nseq, rr[i].st, false));
      });
    } else {
      if (l2 >= 0) RUN(part2(l2, 0, 0, T, stream));
      if (l1 >= 0) RUN(part1(l1, 0, 0, T, 0, d->n_groups, 0, -1, stream, d->n_groups > 1));
    }
    return LAFS_OK;
  };
  // the block's four weight gradients: ONE grouped launch on the side stream, once all their operands exist.  Its 48 (ViT-S)
  // output tiles x 5 token slices fill the chip together: 4x fewer slices -> 4x less partial-sum traffic than four separate
  // launches (csrc/wgrad.hip)
  auto wgrad = [&](int l) -> int {
    if (defer) return LAFS_OK;
    fork();
    RUN(block_wgrad(d, c, l, two ? d->wgrad_workgroups : 0, s2));
    if (two) { done[l] = ev[evi++]; if (hipEventRecord(done[l], s2) != hipSuccess) ev_failed = true; }
    return LAFS_OK;
  };
  RUN(section(-1, layer_hi - 1));
  RUN(wgrad(layer_hi - 1));
  for (int l = layer_hi - 1; l >= layer_lo; --l) {
    // The section below rewrites what layer l+1's weight gradient reads on s2 -- gbm[(l-1)&1] (by layer l's LayerNorm backward:
    // gbm is produced one layer EARLY) and du / gba / dqkv of parity (l-1)&1 (by layer l-1's first part) -- so that launch has to
    // have retired (the two parity buffers cover a lag of one layer, not two)
    if (two && l + 1 < layer_hi && hipStreamWaitEvent(stream, done[l + 1], 0) != hipSuccess) ev_failed = true;
    const int l1 = (l - 1 >= layer_lo) ? l - 1 : -1;
    RUN(section(l, l1));
    if (l1 >= 0) RUN(wgrad(l1));
  }
  // LayerNorm parameter gradients of the layers just walked: the row chains' per-workgroup sums, added in a fixed order (one
  // launch for the whole range; every chain has joined `stream` by now)
  {
    std::vector<lafs_ln_fold_item> items;
    for (int l = layer_hi - 1; l >= layer_lo; --l)
      for (int k = 0; k < 2; ++k) {
        const lafs_block_offsets& o = d->blocks[l];
        lafs_ln_fold_item it = {};
        for (int i = 0; i < n_rr; ++i) {
          it.part[i] = c.ln_part[((size_t)l * 2 + k) * 4 + i];
          const bool fused = k == 1 && mlp_fused_on(d, 4, rr[i].R) && mlp_fused_on(d, 16, rr[i].R) &&
                             (size_t)lafs_mlp_fused_ln_parts(rr[i].R) * 2 * D <= c.ln_part_floats;      // (the same test as part1's)
          it.n_parts[i] = fused ? lafs_mlp_fused_ln_parts(rr[i].R) : lafs_layernorm_bwd_parts(rr[i].R, D);
        }
        it.dgamma = gr + (k == 0 ? o.ln1_g : o.ln2_g); it.dbeta = gr + (k == 0 ? o.ln1_b : o.ln2_b);
        items.push_back(it);
      }
    RUN(lafs_layernorm_bwd_fold(items.data(), (int)items.size(), D, stream));
  }
  if (two && hipStreamWaitEvent(stream, done[layer_lo], 0) != hipSuccess) ev_failed = true;      // join (s2 is in-order)
  LAFS_CHECK_ARG(!ev_failed, "a HIP event call of the weight-gradient stream protocol failed");
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}

extern "C" int lafs_trunk_wgrad(const lafs_trunk_desc* d, void* workspace, int layer_hi, int layer_lo, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  RUN(check_desc(d));
  LAFS_CHECK_ARG(workspace && d->grad, "null buffer");
  LAFS_CHECK_ARG(d->wgrad_defer != 0, "lafs_trunk_wgrad needs a descriptor with wgrad_defer set (per-layer operand slots)");
  LAFS_CHECK_ARG(0 <= layer_lo && layer_lo < layer_hi && layer_hi <= d->depth, "bad layer range");
  const Carve c = carve(d, workspace, 1);
  for (int l = layer_hi - 1; l >= layer_lo; --l) RUN(block_wgrad(d, c, l, d->wgrad_workgroups, stream));
  return LAFS_OK;
}
