"""Class-sharded margin-softmax classifier ("PartialFC") for the WebFace42M fine-tune.

PARITY UNPINNED: the reference never defines PartialFC (only a commented import, face_pre_pro/ViT_face.py:645-649); the
semantics follow InsightFace `partial_fc_v2` (class centres sharded by rank, positives always kept, negatives sampled to
`sample_rate`, distributed softmax).  The self-check (tests/test_gpu_finetune.py) is: sharded result at sample_rate=1 ==
the unsharded CosFace + CE of the oracle.

MI355X layout: each rank owns C/world contiguous class centres in its own flat arena (fp32 master + AdamW moments, bf16
normalised operand built per step).  Per step, per rank:
    all-gather embeddings [W*B, D] + labels  ->  cos = E_n @ W_n[sampled]^T (MFMA)  ->  row max  -> all-reduce MAX
    -> row sum-exp + target logit -> all-reduce SUM  ->  d cos (in place)  ->  dE_n (TN MFMA), dW_n (TN MFMA)
    -> reduce-scatter(SUM) of dE back to the rank that owns the rows.
The three statistics exchanged are [W*B] floats each; the only bulk collectives are the embedding all-gather and its
mirror reduce-scatter (W*B*D*4 bytes: 1.5 MB at B=64, W=8, D=768), far below the xGMI per-link budget.
"""
import torch
import torch.distributed as dist

from . import _lib, ops
from .arena import ParamArena
from .ops import _p, call

f32, bf16 = torch.float32, torch.bfloat16


def _gather_rows(t, world):
    """[n, ...] on every rank -> [world*n, ...] (rank-major).  One collective on RCCL; list form on backends without it."""
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, t)
    else:
        dist.all_gather(list(out.chunk(world)), t)
    return out


def _reduce_scatter_rows(t, world, rank):
    """[world*n, ...] per rank -> sum over ranks of this rank's [n, ...] slice.  reduce-scatter on RCCL; all-reduce + slice on
    backends without it (gloo: the CPU / single-GPU tests)."""
    n = t.shape[0] // world
    if dist.get_backend() == "nccl":
        out = torch.empty((n,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        dist.reduce_scatter_tensor(out, t)
        return out
    dist.all_reduce(t)
    return t[rank * n:(rank + 1) * n].contiguous()


def shard_range(num_classes, rank, world):
    """[start, start+n) of the class ids rank owns (partial_fc_v2: ceil split, the last rank takes the remainder)."""
    n = num_classes // world + int(rank < num_classes % world)
    start = num_classes // world * rank + min(rank, num_classes % world)
    return start, n


def sample_classes(labels, class_start, num_local, num_sample, generator=None, labels2=None):
    """Pick the class centres this rank scores the batch against.

    Returns (index [S] sorted local class ids, y_local [N] int32 position of each row's target inside `index`, -1 if the
    target belongs to another rank) -- and, with `labels2` (the mixup partners' classes), y2_local likewise.  All positives present
    in `labels` (and `labels2`) are kept; negatives are drawn uniformly without replacement until `num_sample` (partial_fc_v2.sample)."""
    labels = labels.long()
    pos_src = labels if labels2 is None else torch.cat([labels, labels2.long()])
    own_p = (pos_src >= class_start) & (pos_src < class_start + num_local)
    if num_sample >= num_local:
        index = torch.arange(num_local, device=labels.device)
    else:
        positive = torch.unique((pos_src - class_start)[own_p], sorted=True)
        if num_sample > positive.numel():
            perm = torch.rand(num_local, device=labels.device, generator=generator)
            perm[positive] = 2.0
            index = torch.topk(perm, k=num_sample)[1].sort()[0]
        else:
            index = positive

    def local_pos(lab):
        own = (lab >= class_start) & (lab < class_start + num_local)
        y = torch.full_like(lab, -1)
        y[own] = torch.searchsorted(index, (lab - class_start)[own])
        return y.to(torch.int32)
    if labels2 is None:
        return index, local_pos(labels)
    return index, local_pos(labels), local_pos(labels2.long())


def normalize_targets(world, margin_type, labels, labels2, lam):
    """(labels2, lam, soft) as forward_backward uses them.  On more than one rank the CosFace head ALWAYS takes the soft form -- a row
    without a mixup partner is its own partner with weight 1 - lam = 0, which the kernels evaluate exactly as a hard label -- because
    the soft form issues two more all-gathers (partner classes, per-row lambdas) and every rank draws its OWN lambda (90 % of the
    draws are 1 at mixup_prob 0.1): a per-rank choice would let the ranks' collective sequences diverge (hang, or mismatched
    buffers)."""
    if labels2 is None and world > 1 and margin_type == 0:
        return labels, 1.0, True
    return labels2, lam, labels2 is not None


class _Centres(torch.nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.normal(0, 0.01, (n, d)))


class PartialFC:
    def __init__(self, embedding_size, num_classes, batch_size, sample_rate=1.0, s=64.0, m=0.4, margin_type=0, device=None,
                 seed=0):
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.D, self.C, self.B = embedding_size, num_classes, batch_size
        if (batch_size * self.world) % 8:
            raise _lib.LafsHipError("PartialFC needs a global batch that is a multiple of 8 (16-byte rows in the class-gradient GEMM)")
        self.s, self.m, self.margin_type = float(s), float(m), margin_type
        self.class_start, self.num_local = shard_range(num_classes, self.rank, self.world)
        self.num_sample = max(1, int(sample_rate * self.num_local)) if sample_rate < 1.0 else self.num_local
        g = torch.Generator().manual_seed(seed + self.rank)
        self.centres = _Centres(self.num_local, embedding_size)
        with torch.no_grad():
            self.centres.weight.copy_(torch.normal(0, 0.01, (self.num_local, embedding_size), generator=g))
        self.arena = ParamArena(self.centres, self.device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed + 1000 + self.rank)
        dev, N = self.device, batch_size * self.world
        self.N = N
        # a rank scores the batch against max(num_sample, #positive classes it owns) centres: every positive is kept even when
        # there are more of them than num_sample (small shards, low sample rates, large global batches); at most N distinct
        # labels exist, so this bound holds on every rank and no buffer can overflow on one rank while the others wait in a
        # collective
        # (mixup: up to 2 N distinct classes -- the rows' own and their partners')
        self.Spad = (max(self.num_sample, min(self.num_local, 2 * N)) + 127) // 128 * 128
        self.ones = torch.ones(self.Spad, device=dev, dtype=f32)
        self.cos = torch.empty(N, self.Spad, device=dev, dtype=f32)
        self.dcos = torch.zeros(N, self.Spad, device=dev, dtype=bf16)
        self.wn = torch.empty(self.Spad, self.D, device=dev, dtype=bf16)
        self.inv_w = torch.empty(self.Spad, device=dev, dtype=f32)
        self.dwn = torch.zeros(self.Spad, self.D, device=dev, dtype=f32)
        self.stats = torch.empty(3, N, device=dev, dtype=f32)          # rowmax | rowsum | target logit
        self.hyper = torch.zeros(_lib.HP_COUNT, device=dev, dtype=f32)

    @property
    def weight(self):
        return self.centres.weight

    def forward_backward(self, emb, labels, grad_scale=1.0, labels2=None, lam=1.0):
        """emb f32 [B, D] (this rank's embeddings), labels [B] global class ids.  Soft (mixup) targets as the reference's margin head
        takes them (train_largescale.py:802, ViT_face.py:69-73): labels2 [B] = the mixup partners' classes, lam = this rank's lambda
        (target lam e_labels + (1 - lam) e_labels2, entering the CosFace margin itself); labels2 None = hard labels.
        Returns (loss = mean CE over the GLOBAL batch, d loss/d emb [B, D] * grad_scale); the class-centre gradient
        accumulates in this rank's arena."""
        dev, D, B, N, W = self.device, self.D, self.B, self.N, self.world
        a = self.arena
        emb = emb.contiguous()
        labels = labels.to(dev, torch.int64).contiguous()
        labels2, lam, soft = normalize_targets(W, self.margin_type, labels, labels2, lam)
        if soft:
            if self.margin_type != 0:
                raise _lib.LafsHipError("soft (mixup) targets need the CosFace margin (ArcFace takes hard labels)")
            labels2 = labels2.to(dev, torch.int64).contiguous()
            lam_rows = torch.full((B,), float(lam), device=dev, dtype=f32)
        if W > 1:
            E, L = _gather_rows(emb, W), _gather_rows(labels, W)
            if soft:
                L2, lam_rows = _gather_rows(labels2, W), _gather_rows(lam_rows, W)      # every rank draws its own lambda
        else:
            E, L = emb, labels
            L2 = labels2 if soft else None
        if soft:
            index, y, y2 = sample_classes(L, self.class_start, self.num_local, self.num_sample, self.gen, labels2=L2)
        else:
            index, y = sample_classes(L, self.class_start, self.num_local, self.num_sample, self.gen)
            y2, lam_rows = None, None
        S = index.numel()
        full = S == self.num_local
        v = a.view(a.master, "weight", (self.num_local, D))
        v_s = v if full else v.index_select(0, index)                   # row gather of the sampled centres (data movement)
        en = torch.empty(N, D, device=dev, dtype=bf16); inv_e = torch.empty(N, device=dev, dtype=f32)
        call("lafs_l2norm_fwd", _p(E), D, _p(en), D, _p(inv_e), N, D)
        call("lafs_weightnorm_fwd", _p(v_s), _p(self.ones), S, self.Spad, D, _p(self.wn), None, self.Spad, _p(self.inv_w))
        ops.gemm_nt(en, self.wn, _lib.EPI_F32, out=self.cos, n_cols=self.Spad)
        rowmax, rowsum, tgt = self.stats[0], self.stats[1], self.stats[2]
        call("lafs_shard_margin_rowmax", _p(self.cos), self.Spad, N, S, _p(y), _p(y2), _p(lam_rows), self.s, self.m, self.margin_type, _p(rowmax))
        if W > 1:
            dist.all_reduce(rowmax, op=dist.ReduceOp.MAX)
        call("lafs_shard_margin_rowsum", _p(self.cos), self.Spad, N, S, _p(y), _p(y2), _p(lam_rows), self.s, self.m, self.margin_type, _p(rowmax),
             _p(rowsum), _p(tgt))
        if W > 1:
            dist.all_reduce(self.stats[1:3])
        call("lafs_shard_margin_grad", _p(self.cos), self.Spad, N, S, _p(y), _p(y2), _p(lam_rows), self.s, self.m, self.margin_type, _p(rowmax),
             _p(rowsum), grad_scale / N)
        loss = (torch.log(rowsum) + rowmax - tgt).mean()
        ops.scale_cast_bf16(self.cos, out=self.dcos)
        den = torch.zeros(N, D, device=dev, dtype=f32)
        ops.gemm_tn_acc(self.dcos.t().contiguous(), self.wn, den)       # dE_n = dcos @ W_n
        self.dwn.zero_()
        ops.gemm_tn_acc(self.dcos, en, self.dwn, splits=1)              # dW_n = dcos^T @ E_n
        gw = a.view(a.grad, "weight", (self.num_local, D))
        if full:
            call("lafs_weightnorm_bwd", _p(self.dwn), _p(v_s), _p(self.ones), _p(self.inv_w), S, D, _p(gw), None, 1)
        else:
            dv = torch.empty(S, D, device=dev, dtype=f32)
            call("lafs_weightnorm_bwd", _p(self.dwn), _p(v_s), _p(self.ones), _p(self.inv_w), S, D, _p(dv), None, 0)
            gw.index_add_(0, index, dv)                                 # row scatter back to the shard (indices are unique)
        dE = torch.empty(N, D, device=dev, dtype=f32)
        call("lafs_l2norm_bwd", _p(E), D, _p(den), D, _p(inv_e), _p(dE), D, N, D)
        demb = _reduce_scatter_rows(dE, W, self.rank) if W > 1 else dE
        return loss, demb

    def optimizer_step(self, lr, weight_decay=0.1, beta1=0.9, beta2=0.999, eps=1e-8):
        """Dense AdamW over the shard (centres that were not sampled see a zero gradient this step)."""
        a = self.arena
        h = torch.zeros(_lib.HP_COUNT, dtype=f32)
        h[_lib.HP_LR], h[_lib.HP_WD], h[_lib.HP_BETA1], h[_lib.HP_BETA2], h[_lib.HP_EPS] = lr, weight_decay, beta1, beta2, eps
        h[_lib.HP_CLIP], h[_lib.HP_EMA_M], h[_lib.HP_FREEZE_LAST], h[_lib.HP_GRAD_SCALE] = 0.0, 0.0, 0.0, 1.0
        self.hyper.copy_(h)
        call("lafs_clip_adamw_ema", _p(a.master), _p(a.grad), _p(a.exp_avg), _p(a.exp_avg_sq), None, _p(a.shadow), None,
             _p(a.chunk_seg), a.n_chunks, _p(a.seg_flags), _p(a.seg_step), a.n_seg, _p(a.seg_sumsq), _p(self.hyper))
        a.zero_grad()
