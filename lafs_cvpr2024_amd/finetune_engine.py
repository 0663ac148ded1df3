"""Fused Part-fViT + CosFace fine-tune micro-step (reference train_largescale.py:785-891) on the HIP kernels.

    u8 batch -> (x/255*2-1) + batch mixup (one kernel) -> [trainable landmark CNN -> theta -> patch gather] -> Part-fViT trunk ->
    L2-normalised embedding x L2-normalised class centres (MFMA GEMM) -> fused margin + softmax + soft-target CE (the dense [B, C]
    mixup target is never built: it has <= 2 non-zeros per row) -> backward -> every `acc_step` micro-steps: AdamW over the flat arena.

Round 4: every per-step tensor is allocated once, no ATen kernel is left in a micro-step (labels, DropPath masks, operand
transposes, the image gradient re-indexing and the gradient zeroing are lafs_* launches), the mixup lambda / step counter are read
from device memory, and the whole micro-step is ONE hipGraph (two captured variants: the first micro-step of an accumulation
window WRITES the block and class-table weight gradients instead of accumulating into a zeroed 1 GB arena).

Round 6, data-parallel runs (reference: DistributedDataParallel(broadcast_buffers=True), train_largescale.py:676-677): parameters and
buffers are broadcast from rank 0 at construction; the micro-steps of an accumulation window that reduce nothing are the same single
graph as on one rank (deferred weight gradients included), and the window's LAST micro-step -- whose gradient slices go out over RCCL
as the backward retires them -- is captured as one hipGraph per segment with the FlatReducer launches between the replays, exactly as
LafsPretrainEngine does.  Only the class-sharded head keeps the eager form (its collectives sit inside the head itself).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, functional as Fn, ops
from .face_pre_pro.ViT_face import ViT_face_landmark_patch8
from .distributed import FlatReducer
from .ops import _p, call
from .utils import PinnedRing
from .vision_transformer import attach_arena

f32, bf16 = torch.float32, torch.bfloat16


LOW_WEIGHT_DECAY = 5e-2


def finetune_decay_group(name, param):
    """Weight-decay class of a fine-tune parameter, as param_groups_lrd assigns it (reference train_largescale.py:139-149):
    1-D tensors do not decay, the landmark CNN's matrices (`stn*`) decay at 5e-2, everything else at the configured rate
    (1e-1).  The `lr_scale` the reference also stores per group is never applied by torch's AdamW."""
    if param.dim() == 1:
        return "none"
    return "low" if name.startswith("stn") else "decay"


class FinetuneEngine:
    def __init__(self, backbone: ViT_face_landmark_patch8, batch_size, acc_step=3, mixup_alpha=0.2, mixup_prob=0.1,
                 s=64.0, m=0.4, margin_type=0, image_size=112, device=None, sharded_head=None, use_graph=None):
        """margin_type 0 = CosFace (the reference), 1 = ArcFace (parity unpinned), on the dense `backbone.loss.weight` head;
        `sharded_head` (a partial_fc.PartialFC) replaces it by the class-sharded head (CosFace: mixup targets as on the dense head).
        use_graph: None = capture the micro-step whenever it is capturable (single rank, dense head; LAFS_FT_GRAPH=0 disables)."""
        if not isinstance(backbone, ViT_face_landmark_patch8) or (sharded_head is None and not hasattr(backbone, "loss")):
            raise _lib.LafsHipError("FinetuneEngine drives ViT_face_landmark_patch8(loss_type='CosFace') or a sharded head")
        if batch_size % 8:
            raise _lib.LafsHipError("FinetuneEngine needs a batch size that is a multiple of 8 (16-byte rows in the class-gradient GEMM)")
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        self.model, self.B, self.acc_step = backbone, batch_size, acc_step
        self.mixup_alpha, self.mixup_prob = mixup_alpha, mixup_prob
        self.s, self.m, self.margin_type = float(s), float(m), margin_type
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.arena = attach_arena(backbone, self.device)
        self.arena.set_decay_groups(finetune_decay_group)
        self.ctx = _lib.Ctx(self.device)                 # this engine's side streams / events / kernel options (lafs_ctx)
        self.arena.ctx = self.ctx
        self.cnn_flush = None
        if self.world > 1:
            # DDP's construction-time broadcast of parameters AND buffers from rank 0 (train_largescale.py:676-677; before the landmark
            # plan below takes its references to the BatchNorm statistics).  DDP also re-broadcasts the buffers before every forward:
            # in training mode the BatchNorm layers normalise with batch statistics, the running statistics are only written, and
            # only rank 0 saves checkpoints (train_largescale.py:899-910) -- rank 0's buffers evolve as under DDP and the other ranks'
            # copies are never read; sync_buffers() makes every rank hold rank 0's values (before an evaluation on all ranks).
            dist.broadcast(self.arena.master, 0)
            self.arena.refresh_shadows()
            self.sync_buffers()
        self.head = sharded_head
        self.C = backbone.loss.out_features if sharded_head is None else 8
        self.Cpad = (self.C + 127) // 128 * 128
        self.D = backbone.dim
        self.S = image_size
        self.geom = Fn.geometry([(batch_size, image_size)], self.device)
        dev, a, m, B, D = self.device, self.arena, backbone, batch_size, self.D
        self.hyper = torch.zeros(_lib.HP_COUNT, device=dev, dtype=f32)
        self.hyper_ring = PinnedRing((_lib.HP_COUNT,), f32)     # asynchronous upload; a pageable copy would block the host
        self._hp = {}                                            # host copy of the hyper-parameters (every upload rewrites the whole vector)
        # data parallelism (reference: DDP bucketed all-reduce overlapped with backward, train_largescale.py:676-677, 867): on the
        # last micro-step of an accumulation window the flat gradient goes out over RCCL in slices AS THE BACKWARD RETIRES
        # THEM -- the 0.6 GB margin head first (it is final before the trunk backward starts), then runs of blocks from the top,
        # last the embedding / landmark-CNN range -- and optimizer_step only waits.  Reducing once per window instead of once
        # per micro-step is the same sum (no_sync semantics).
        self.reducer = FlatReducer()
        self.grad_slices = 4
        names = m._spec.trunk.block_names
        self.block0 = a.offsets[names[0]["ln1_g"]]
        self.block_off = [a.offsets[n["ln1_g"]] for n in names] + [a.offsets[m._spec.prefix + m._spec.final_g]]
        self.wname = m._spec.prefix + "loss.weight"
        self.head_off = a.offsets[self.wname] if (sharded_head is None) else a.size
        self._reduced = False
        # tensors whose gradient is WRITTEN by its producer on the first micro-step of an accumulation window (block weights: wgrad
        # fold with accumulate = 0; class table: weight-norm backward): that micro-step zeroes only the others (lafs_zero_chunks)
        flags = a.seg_flags.cpu().tolist()
        over = {nm[k] for nm in names for k in ("w_qkv", "w_proj", "w_fc1", "w_fc2")}
        if sharded_head is None:
            over.add(self.wname)
        for i, name in enumerate(a.names):
            if name in over:
                flags[i] |= _lib.SEG_OVERWRITTEN
        a.seg_flags.copy_(torch.tensor(flags, dtype=torch.int32))
        # ---- static buffers of a micro-step
        S = image_size
        self.in_u8 = torch.zeros(B, 3, S, S, device=dev, dtype=torch.uint8)
        self.y1 = torch.zeros(B, device=dev, dtype=torch.int32)
        self.x = torch.empty(B, 3, S, S, device=dev, dtype=f32)
        self.img_in = torch.empty(B, 3, S, S, device=dev, dtype=f32)
        self.loss = torch.zeros(1, device=dev, dtype=f32)
        self.g_buf = torch.empty(self.geom.n_tok, D, device=dev, dtype=f32)
        self.x_in = torch.empty(self.geom.n_tok, D, device=dev, dtype=f32)
        self.x_out = torch.empty(self.geom.n_tok, D, device=dev, dtype=f32)
        self.demb = torch.empty(B, D, device=dev, dtype=f32)
        self.drop = (torch.empty(m.depth, 2, self.geom.n_seq, device=dev, dtype=f32) if m.drop_path_rate else None)
        self.keep = (torch.full((m.depth,), 1.0 - m.drop_path_rate, device=dev, dtype=f32) if m.drop_path_rate else None)
        self.drop_seed = 0x0F17E
        self.pos_rows = a.view(a.master, m._spec.prefix + "pos_embedding").view(-1, D)[: self.geom.npatch(0) + 1]
        self.dpos_rows = a.view(a.grad, m._spec.prefix + "pos_embedding").view(-1, D)[: self.geom.npatch(0) + 1]
        if sharded_head is None:
            self.ones = torch.ones(self.C, device=dev, dtype=f32)
            self.cos = torch.empty(B, self.Cpad, device=dev, dtype=f32)
            self.dcos = torch.zeros(B, self.Cpad, device=dev, dtype=bf16)
            self.dcos_t = torch.zeros(self.Cpad, B, device=dev, dtype=bf16)
            self.wn = torch.empty(self.Cpad, D, device=dev, dtype=bf16)
            self.inv_w = torch.empty(self.C, device=dev, dtype=f32)
            self.dwn = torch.empty(self.Cpad, D, device=dev, dtype=f32)
            self.row_ws = torch.empty(B, device=dev, dtype=f32)
            self.part_ws = torch.empty(B * 32, device=dev, dtype=f32)
            self.xn = torch.empty(B, D, device=dev, dtype=bf16)
            self.xn_t = torch.empty(D, B, device=dev, dtype=bf16)
            self.inv_x = torch.empty(B, device=dev, dtype=f32)
            self.dxn = torch.empty(B, D, device=dev, dtype=f32)
            self.dxn_ws = ops.wgrad_workspace(self.Cpad, B, D, dev)                 # d(emb_n): reduction over the classes
            self.dwn_ws = None if B % 32 == 0 else ops.wgrad_workspace(B, self.Cpad, D, dev)
        self.dmosaic = torch.empty(B, 3, S, S, device=dev, dtype=f32) if m.with_land else None
        self.dth = None
        self.micro = 0                                   # micro-steps taken (seeds the per-step masks)
        self._since_opt = 0                              # ... since the last optimizer step
        # optimizer_step leaves the (dead) gradients in place: the next micro-step zeroes / overwrites them itself.  True restores
        # "gradients are zero after a step" for callers that inspect arena.grad / p.grad afterwards (a 1 GB memset per step at C4)
        self.zero_after_step = False
        # trainable landmark branch: the HIP plan (landmark_train.HipLandmarkTrainer) in training mode (BatchNorm batch statistics,
        # Dropout(0.5)) and in eval mode (running statistics as constants of the backward, no dropout); LAFS_FT_CNN=torch, for A/B
        # runs, takes the nn.Module on torch autograd
        self.cnn = None
        if m.with_land and os.environ.get("LAFS_FT_CNN", "hip") == "hip":
            from .landmark_train import HipLandmarkTrainer
            self.cnn = HipLandmarkTrainer(m, a, batch_size, image_size, device=dev)
            self.cnn.step_dev = self.hyper[_lib.HP_STEP:]
            m.register_state_dict_pre_hook(lambda *a_, **k_: self.cnn.flush_batches_tracked())
            self.cnn_flush = self.cnn.flush_batches_tracked
        # streams / graphs
        # (The blocks' weight gradients on a second stream INSIDE the chain were measured neutral at C4 -- 27.24 ms with, 27.03 without --
        # and removed in round 5: tools/lab/NOTES.md.  What pays is deferring them beside the landmark CNN's backward, below.)
        self.wgrad_workgroups = 0
        # Deferred weight gradients (single GPU, HIP landmark plan): the trunk backward launches only its input-gradient chain and keeps
        # every block's dY operands; the twelve grouped weight-gradient launches then run on a second stream BESIDE the landmark
        # CNN's backward -- ~300 small launch-latency-sized kernels that leave most of the chip idle -- capped to
        # LAFS_FT_DEFER_WG workgroups so that the CNN's kernels find free CUs.  Measured at C4 (tools/lab/NOTES.md): the same launches
        # cost 4.5 ms in front of the CNN backward and 2.75 ms beside it.  LAFS_FT_WGRAD_DEFER=0 restores the immediate form.
        # (data-parallel runs: on every micro-step of a window but the last, whose gradient slices leave as the backward retires them)
        self.defer = (self.cnn is not None and sharded_head is None
                      and os.environ.get("LAFS_SINGLE_STREAM") != "1" and os.environ.get("LAFS_FT_WGRAD_DEFER", "1") != "0")
        self.defer_stream = torch.cuda.Stream(device=dev) if self.defer else None
        if self.defer:
            self.wgrad_workgroups = int(os.environ.get("LAFS_FT_DEFER_WG", "128"))
        if use_graph is None:
            use_graph = os.environ.get("LAFS_FT_GRAPH", "1") != "0"
        self.use_graph = bool(use_graph) and sharded_head is None
        self._graphs = {}
        self._pool = None
        self._ws = None

    def sync_buffers(self):
        """Every rank takes rank 0's buffers (BatchNorm running statistics and counters of the landmark CNN): what DDP's
        broadcast_buffers does in front of every forward (train_largescale.py:676-677)."""
        if self.world > 1:
            if self.cnn_flush is not None:
                self.cnn_flush()
            for b in self.model.buffers():
                dist.broadcast(b, 0)

    def _defer_now(self):
        """Deferred weight gradients on this micro-step?  Not on the one that completes a data-parallel window: its gradient
        ranges are handed to the collective run by run, so every block's weight gradients have to exist when its run ends."""
        return self.defer and not self._reduce_now()

    # ------------------------------------------------------------------ host side of a micro-step
    def draw_lambda(self):
        if np.random.rand() < self.mixup_prob:
            return float(np.random.beta(self.mixup_alpha, self.mixup_alpha))
        return 1.0

    def _upload_hyper(self, kw):
        self._hp.update(kw)

        def fill(h):
            h.zero_()
            for k, v in self._hp.items():
                h[k] = v
        self.hyper_ring.upload(self.hyper, fill)

    def _stage(self, inputs_u8, labels):
        """Copy the batch into the static buffers the captured step reads (no ATen math: one copy, one lafs_cast_i64_i32)."""
        if inputs_u8.shape != self.in_u8.shape or inputs_u8.dtype != torch.uint8:
            raise _lib.LafsHipError(f"expected a uint8 batch of shape {tuple(self.in_u8.shape)}")
        self.in_u8.copy_(inputs_u8, non_blocking=True)
        lab = labels if labels.is_cuda else labels.to(self.device, non_blocking=True)
        if lab.dtype == torch.int64:
            call("lafs_cast_i64_i32", _p(lab.contiguous()), _p(self.y1), self.B)
        else:
            self.y1.copy_(lab.to(torch.int32))

    def _capturable(self):
        m = self.model
        return self.use_graph and (not m.with_land or self.cnn is not None)

    def micro_step(self, inputs_u8, labels, lam=None):
        """One forward/backward on a uint8 NCHW batch.  Gradients accumulate in the arena (loss pre-divided by acc_step).
        The FIRST micro-step after an optimizer step overwrites / zeroes the gradients itself: between optimizer_step and the next
        micro_step, arena.grad and p.grad hold the previous window's (stale) gradients unless `zero_after_step` is set."""
        a, m = self.arena, self.model
        a.ensure_fresh()
        lam = self.draw_lambda() if lam is None else float(lam)
        if self.head is not None and self.head.margin_type != 0:
            lam = 1.0                                    # ArcFace on the sharded head takes hard labels (parity unpinned either way)
        self._lam = lam
        self._stage(inputs_u8, labels)
        self._labels = labels
        # (the counter seeds the DropPath / dropout / CNN-dropout masks; fp32 holds integers exactly below 2^24: wrap like the LAFS engine)
        self._upload_hyper({_lib.HP_MIX_LAM: lam, _lib.HP_STEP: float((self.micro + 1) % (1 << 24))})
        first = (self._since_opt == 0)                   # first micro-step since the last optimizer step: gradients are written, not accumulated
        if self._capturable():
            key = (first, bool(m.training), self._reduce_now())
            segs = self._graphs.get(key)
            if segs is None:
                segs = self._capture(first)
                self._graphs[key] = segs
            for g, out in segs:                          # one graph per segment; the gradient range a segment completes goes out behind it
                g.replay()
                if out is not None:
                    self.reducer.launch(out)
            if key[2]:
                self._reduced = True
            if self._cnn_hip and m.training:             # nn.BatchNorm2d.num_batches_tracked: one training forward of the landmark CNN per replay
                self.cnn.n_forward += 1
        else:
            self._body(first)
        self.micro += 1
        self._since_opt += 1
        return self.loss

    def _warmup(self, first):
        """One eager micro-step body on a side stream BEFORE the first capture (as LafsPretrainEngine._capture does): code objects
        are loaded, kernel attributes set and the per-step workspaces (`_ws`, `dth`) allocated by the ordinary allocator instead of
        from inside a stream capture / the graph's private pool.  The state the body mutates -- gradients, the loss, the landmark
        CNN's BatchNorm running statistics and loss-scale state -- is snapshotted and put back."""
        a, cnn = self.arena, self.cnn
        bn = []
        if cnn is not None:
            bn = [t for s in [cnn.stem["bn"]] + [L[k] for L in cnn.blocks for k in ("bn1", "bn2", "bn3")] for t in (s["rm"], s["rv"]) if t is not None]
            bn.append(cnn.gscale)
        saved = [t.clone() for t in [a.grad, self.loss] + bn]
        n_fwd = cnn.n_forward if cnn is not None else 0
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            self._body(first)                            # (a window's last micro-step: its all-reduces really run, on every rank alike)
            self.reducer.wait_all()
        self._reduced = False
        cur.wait_stream(s)
        torch.cuda.synchronize()
        for dst, src in zip([a.grad, self.loss] + bn, saved):
            dst.copy_(src)
        if cnn is not None:
            cnn.n_forward = n_fwd
            cnn.step = max(cnn.step - 1, 0)
        self._warm = True

    def _capture(self, first):
        if not getattr(self, "_warm", False):
            self._warmup(first)
        if self.cnn is not None:
            self.cnn.mark_stale()                        # the operand refresh becomes part of the graph: every replay sees fresh weights
        # The body is a generator that yields the gradient range to all-reduce wherever a collective has to sit between kernels (only on
        # the last micro-step of a data-parallel window): every stretch between two yields becomes one hipGraph.  With a process group
        # alive its watchdog thread polls events while this thread captures -- thread-local capture mode keeps its (legal) runtime
        # calls from invalidating the capture (as in LafsPretrainEngine._capture).
        segs, gen, done = [], self._body_gen(first), False
        mode = dict(capture_error_mode="thread_local") if self.world > 1 else {}
        while not done:
            g, out = torch.cuda.CUDAGraph(), None
            with torch.cuda.graph(g, pool=self._pool, **mode):
                try:
                    out = next(gen)
                except StopIteration:
                    done = True
            if self._pool is None:
                self._pool = g.pool()
            segs.append((g, out))
        return segs

    # ------------------------------------------------------------------ the device side (capturable: lafs_* launches only)
    def _body(self, first):
        """Eager form: the segments back to back, each completed gradient range handed to the reducer as it appears."""
        for out in self._body_gen(first):
            self.reducer.launch(out)
        if self._reduce_now():
            self._reduced = True

    def _body_gen(self, first):
        a, m, B, D, dev = self.arena, self.model, self.B, self.D, self.device
        defer = self._defer_now()
        hp = self.hyper
        if first:                                        # gradient of everything no kernel overwrites <- 0
            call("lafs_zero_chunks", _p(a.grad), _p(a.chunk_seg), _p(a.seg_flags), a.n_chunks, _lib.SEG_OVERWRITTEN)
        call("lafs_mixup_normalize", _p(self.in_u8), _p(self.x), B, self.S, 1.0, _p(hp[_lib.HP_MIX_LAM:]))
        drop = None
        if m.training and self.drop is not None:         # Residual_droppath: the same rate on both branches of every layer (:106-112)
            call("lafs_droppath_scales", _p(self.keep), m.depth, self.geom.n_seq, self.drop_seed, _p(hp[_lib.HP_STEP:]), _p(self.drop))
            drop = self.drop
        dropout = None
        if m.training and (m.dropout_rate > 0.0 or m.emb_dropout_rate > 0.0):
            dropout = (m.dropout_rate, m.emb_dropout_rate, m._drop_seed0, hp[_lib.HP_STEP:])
        img_in, theta, th = self.x, None, None
        self._cnn_hip = m.with_land and self.cnn is not None
        if m.with_land:
            # trainable landmark regressor -> one-launch patch gather (ViT_face.py:679-711)
            theta = self.cnn.forward(self.x) if self._cnn_hip else m.landmarks(self.x)
            m.theta = theta
            th = theta.detach().contiguous()
            img_in = self.img_in
            call("lafs_patch_gather_fwd", _p(self.x), _p(th), B, self.S, th.shape[1], _p(img_in))
        if self._ws is None:
            # (sized for the deferred form -- one operand slot per layer -- which also holds the immediate form's two)
            probe = Fn.make_trunk_desc(a, m._spec.trunk, self.geom, drop, with_grad=True, wgrad_defer=self.defer,
                                       wgrad_workgroups=self.wgrad_workgroups if self.defer else 0)
            self._ws = Fn.trunk_workspace(probe, True, dev)
        emb, st, _ = Fn.vit_forward(a, m._spec, self.geom, [img_in], [self.pos_rows], drop, save=True, dropout=dropout, ws=self._ws,
                                    x_in=self.x_in, x_out=self.x_out, wgrad_overwrite=first,
                                    wgrad_workgroups=self.wgrad_workgroups if defer else 0,
                                    wgrad_defer=defer)
        if self.head is not None:
            # class-sharded head: all-gather embeddings, local logits, exchanged softmax statistics, reduce-scatter of dE.
            # demb is the gradient of the GLOBAL-batch mean loss, so the later all-reduce of the backbone gradients is a SUM.
            # soft (mixup) targets as the reference's margin head always gets them (train_largescale.py:802): the partner of row b is
            # row B-1-b of this rank's batch (util/mixup_my.py:189-200), its weight 1 - lambda
            soft = self._lam != 1.0                      # (more than one rank: PartialFC takes the soft form on every rank, normalize_targets)
            loss, demb = self.head.forward_backward(emb, self._labels, grad_scale=1.0 / self.acc_step,
                                                    labels2=self._labels.flip(0) if soft else None, lam=self._lam)
            self.loss.copy_(loss.detach().view(1))
            yield from self._backward_trunk(st, demb, th, theta)
            return
        # ---- margin head (face_pre_pro/ViT_face.py:49-89 + timm SoftTargetCrossEntropy, train_largescale.py:820)
        call("lafs_l2norm_fwd", _p(emb), D, _p(self.xn), D, _p(self.inv_x), B, D)
        call("lafs_weightnorm_fwd", _p(a.view(a.master, self.wname)), _p(self.ones), self.C, self.Cpad, D, _p(self.wn), None, self.Cpad,
             _p(self.inv_w))
        ops.gemm_nt(self.xn, self.wn, _lib.EPI_F32, out=self.cos, n_cols=self.Cpad)
        # margin + softmax + soft-target CE; d/dcos leaves as bf16 (the operand of the two class-gradient GEMMs); the mixup partner of
        # row b is row B-1-b, lambda comes from device memory
        call("lafs_margin_softmax_ce_bf16", _p(self.cos), self.Cpad, B, self.C, _p(self.y1), None, 1.0, _p(hp[_lib.HP_MIX_LAM:]), self.s,
             self.m, self.margin_type, 1.0 / self.acc_step, _p(self.dcos), self.Cpad, _p(self.loss), _p(self.row_ws), _p(self.part_ws))
        call("lafs_transpose_bf16", _p(self.dcos), B, self.Cpad, self.Cpad, _p(self.dcos_t), B)
        # d(emb_n) [B, D] = dcos @ Wn: reduction over the classes = the token axis of the wide-tile weight-gradient kernel (slices
        # + fold: no atomics, no zero fill)
        ops.wgrad(self.dcos_t, self.wn, self.dxn, accumulate=False, workspace=self.dxn_ws)
        # d(Wn) and the weight-norm backward of the class table (~0.55 ms of HBM-bound work at C4) gate nothing in the trunk backward:
        # with deferred weight gradients they follow those on the second stream, beside the landmark CNN's backward
        self._head_first = first
        if not (defer and self._cnn_hip and os.environ.get("LAFS_FT_DEFER_HEAD", "1") != "0"):
            self._head_param_grads()
            self._head_first = None
        call("lafs_l2norm_bwd", _p(emb), D, _p(self.dxn), D, _p(self.inv_x), _p(self.demb), D, B, D)
        if self._reduce_now():
            yield a.grad[self.head_off:]                         # margin head (+ anything behind it): final from here on
        yield from self._backward_trunk(st, self.demb, th, theta)

    def _head_param_grads(self):
        """d(Wn) [C, D] = dcos^T @ emb_n (reduction over the batch: an NT GEMM that writes the 633 MB matrix once) and the weight-norm
        backward into the class table's gradient (written on the first micro-step of a window, accumulated after)."""
        a, B, D = self.arena, self.B, self.D
        if self.dwn_ws is None:
            call("lafs_transpose_bf16", _p(self.xn), B, D, D, _p(self.xn_t), B)
            ops.gemm_nt(self.dcos_t, self.xn_t, _lib.EPI_F32, out=self.dwn)
        else:
            ops.wgrad(self.dcos, self.xn, self.dwn, accumulate=False, workspace=self.dwn_ws)
        call("lafs_weightnorm_bwd", _p(self.dwn), _p(a.view(a.master, self.wname)), _p(self.ones), _p(self.inv_w), self.C, D,
             _p(a.view(a.grad, self.wname)), None, 0 if self._head_first else 1)

    def _reduce_now(self):
        """True on the micro-step that completes an accumulation window of a data-parallel run."""
        return self.world > 1 and (self.micro + 1) % self.acc_step == 0

    def _trunk_layers_backward(self, st, demb):
        """Final norm + all blocks; with DP on the window's last micro-step in `grad_slices` runs, each run's gradient range
        handed to RCCL as soon as it has been enqueued (a generator: it yields those ranges)."""
        a, m = self.arena, self.model
        g = Fn.vit_backward_begin(a, m._spec, st, demb, g_buf=self.g_buf)
        self._g = g
        depth = m.depth
        if not self._reduce_now():
            Fn.vit_backward_layers(st, g, depth, 0)
            return
        ns = max(1, min(self.grad_slices, depth))
        cuts = [depth - (depth * k) // ns for k in range(ns + 1)]
        hi = self.head_off
        for k in range(ns):
            Fn.vit_backward_layers(st, g, cuts[k], cuts[k + 1])
            lo = self.block_off[cuts[k + 1]]
            if cuts[k + 1] > 0:                                  # the range below block 0 still waits for the embedding / CNN gradients
                yield a.grad[lo:hi]
                hi = lo
        self._hi_left = hi

    def _backward_trunk(self, st, demb, th, theta):
        a, m, B, D = self.arena, self.model, self.B, self.D
        reduce_now = self._reduce_now()
        yield from self._trunk_layers_backward(st, demb)
        g = self._g
        deferred = bool(st.desc.wgrad_defer)
        cur = torch.cuda.current_stream()
        if deferred and self._cnn_hip:                   # the blocks' weight gradients: beside everything from here to the end of the CNN backward
            self.defer_stream.wait_stream(cur)
            with torch.cuda.stream(self.defer_stream):
                Fn.vit_wgrad_layers(st, m.depth, 0)
                if getattr(self, "_head_first", None) is not None:
                    self._head_param_grads()
                    self._head_first = None
        elif deferred:                                   # (eval-mode model / torch CNN: nothing to hide them behind)
            Fn.vit_wgrad_layers(st, m.depth, 0)
        if m.with_land:
            _, dx = Fn.vit_backward_end(a, m._spec, st, g, want_dx=True, dpos_out=[self.dpos_rows])
            call("lafs_unpatchify_f32", _p(dx[0]), B, self.S, m._spec.patch_order, _p(self.dmosaic))
            if self.dth is None:
                self.dth = torch.empty_like(th)
            call("lafs_patch_gather_bwd", _p(self.x), _p(th), _p(self.dmosaic), B, self.S, th.shape[1], _p(self.dth), None)
            if self._cnn_hip:
                self.cnn.backward(self.dth)              # into the arena's stn.* / output_layer.* gradients
            else:
                theta.backward(self.dth)                 # torch autograd: p.grad are views of the arena
        else:
            Fn.vit_backward_end(a, m._spec, st, g, dpos_out=[self.dpos_rows])
        if deferred and self._cnn_hip:
            cur.wait_stream(self.defer_stream)
        if reduce_now:                                           # what is left: block 0's run + embedding + landmark CNN
            yield a.grad[: self._hi_left]

    # ------------------------------------------------------------------ optimizer
    def optimizer_step(self, lr, weight_decay=0.1, beta1=0.9, beta2=0.999, eps=1e-8):
        """AdamW over every tensor (decay only on >= 2-D tensors, train_largescale.py:122-173); all-reduces the flat gradient
        first when running data-parallel (the reference's DDP does it on every micro-step, :676-677)."""
        a = self.arena
        if self.world > 1:
            if self._reduced:
                self.reducer.wait_all()                          # launched slice by slice during the last backward
            else:
                dist.all_reduce(a.grad)                          # optimizer_step called outside the acc_step cadence
            self._reduced = False
        # dense head: every rank's loss is its local mean -> average; sharded head: gradients of the global mean -> sum
        gscale = 1.0 if self.head is not None else 1.0 / self.world
        self._upload_hyper({_lib.HP_LR: lr, _lib.HP_WD: weight_decay, _lib.HP_BETA1: beta1, _lib.HP_BETA2: beta2, _lib.HP_EPS: eps,
                            _lib.HP_GRAD_SCALE: gscale, _lib.HP_WD_LOW: LOW_WEIGHT_DECAY})
        if self.head is not None:
            self.head.optimizer_step(lr, weight_decay, beta1, beta2, eps)
        call("lafs_clip_adamw_ema", _p(a.master), _p(a.grad), _p(a.exp_avg), _p(a.exp_avg_sq), None, _p(a.shadow), None,
             _p(a.chunk_seg), a.n_chunks, _p(a.seg_flags), _p(a.seg_step), a.n_seg, _p(a.seg_sumsq), _p(self.hyper))
        a.refresh_transposed()
        if self.zero_after_step:
            ops.zero_(a.grad)
        if self.cnn is not None:
            self.cnn.mark_stale()                        # the master weights changed in place: new operand images next forward
        self._since_opt = 0

    def zero_grad(self):
        ops.zero_(self.arena.grad)
        self._since_opt = 0

    def step(self, inputs_u8, labels, lr, weight_decay=0.1):
        """micro_step + optimizer step every acc_step micro-steps (train_largescale.py:842-891)."""
        loss = self.micro_step(inputs_u8, labels)
        if self.micro % self.acc_step == 0:
            self.optimizer_step(lr, weight_decay)
        return loss
