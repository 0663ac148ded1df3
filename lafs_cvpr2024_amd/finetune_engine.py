"""Fused Part-fViT + CosFace fine-tune micro-step (reference train_largescale.py:785-891) on the HIP kernels.

    u8 batch -> (x/255*2-1) + batch mixup (one kernel) -> Part-fViT trunk -> L2-normalised embedding x L2-normalised class
    centres (MFMA GEMM) -> fused margin + softmax + soft-target CE (the dense [B, C] mixup target is never built: it has
    <= 2 non-zeros per row) -> backward -> every `acc_step` micro-steps: AdamW over the flat arena.
"""
import math
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
                 s=64.0, m=0.4, margin_type=0, image_size=112, device=None, sharded_head=None):
        """margin_type 0 = CosFace (the reference), 1 = ArcFace (parity unpinned), on the dense `backbone.loss.weight` head;
        `sharded_head` (a partial_fc.PartialFC) replaces it by the class-sharded head (hard labels: mixup is off)."""
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
        _lib.lib().lafs_trunk_streams_init()
        self.head = sharded_head
        self.C = backbone.loss.out_features if sharded_head is None else 8
        self.Cpad = (self.C + 127) // 128 * 128
        self.D = backbone.dim
        self.geom = Fn.geometry([(batch_size, image_size)], self.device)
        dev = self.device
        self.hyper = torch.zeros(_lib.HP_COUNT, device=dev, dtype=f32)
        self.hyper_ring = PinnedRing((_lib.HP_COUNT,), f32)     # asynchronous upload; a pageable copy would block the host
        # data parallelism (reference: DDP bucketed all-reduce overlapped with backward, train_largescale.py:676-677, 867): on the
        # last micro-step of an accumulation window the flat gradient goes out over RCCL in slices AS THE BACKWARD RETIRES
        # THEM -- the 0.6 GB margin head first (it is final before the trunk backward starts), then runs of blocks from the top,
        # last the embedding / landmark-CNN range -- and optimizer_step only waits.  Reducing once per window instead of once
        # per micro-step is the same sum (no_sync semantics).
        self.reducer = FlatReducer()
        self.grad_slices = 4
        names = backbone._spec.trunk.block_names
        self.block0 = self.arena.offsets[names[0]["ln1_g"]]
        self.block_off = [self.arena.offsets[n["ln1_g"]] for n in names] + [self.arena.offsets[backbone._spec.prefix + backbone._spec.final_g]]
        hn = backbone._spec.prefix + "loss.weight"
        self.head_off = self.arena.offsets[hn] if (sharded_head is None) else self.arena.size
        self._reduced = False
        self.ones = torch.ones(self.C, device=dev, dtype=f32)
        self.x = torch.empty(batch_size, 3, image_size, image_size, device=dev, dtype=f32)
        self.cos = torch.empty(batch_size, self.Cpad, device=dev, dtype=f32)
        self.dcos = torch.zeros(batch_size, self.Cpad, device=dev, dtype=bf16)
        self.wn = torch.empty(self.Cpad, self.D, device=dev, dtype=bf16)
        self.inv_w = torch.empty(self.C, device=dev, dtype=f32)
        self.dwn = torch.zeros(self.Cpad, self.D, device=dev, dtype=f32)
        self.loss = torch.zeros(1, device=dev, dtype=f32)
        self.row_ws = torch.empty(batch_size, device=dev, dtype=f32)
        self.micro = 0
        # trainable landmark branch: the HIP training plan (landmark_train.HipLandmarkTrainer) whenever the model is in training mode
        # (BatchNorm batch statistics, Dropout(0.5)); an eval-mode model -- and LAFS_FT_CNN=torch, for A/B runs -- takes the nn.Module
        # on torch autograd
        self.cnn = None
        if backbone.with_land and os.environ.get("LAFS_FT_CNN", "hip") == "hip":
            from .landmark_train import HipLandmarkTrainer
            self.cnn = HipLandmarkTrainer(backbone, self.arena, batch_size, image_size, device=dev)
            backbone.register_state_dict_pre_hook(lambda *a, **k: self.cnn.flush_batches_tracked())

    def draw_lambda(self):
        if np.random.rand() < self.mixup_prob:
            return float(np.random.beta(self.mixup_alpha, self.mixup_alpha))
        return 1.0

    def micro_step(self, inputs_u8, labels, lam=None):
        """One forward/backward on a uint8 NCHW batch.  Gradients accumulate in the arena (loss pre-divided by acc_step)."""
        a, m, B, D, dev = self.arena, self.model, self.B, self.D, self.device
        a.ensure_fresh()
        lam = self.draw_lambda() if lam is None else float(lam)
        if self.head is not None:
            lam = 1.0                                    # the sharded head takes hard labels
        call("lafs_mixup_normalize", _p(inputs_u8.contiguous()), _p(self.x), B, self.x.shape[-1], lam)
        y1 = labels.to(dev, torch.int32).contiguous()
        y2 = y1.flip(0).contiguous()
        pos = a.view(a.master, m._spec.prefix + "pos_embedding").view(-1, D)[: self.geom.npatch(0) + 1]
        drop = m._sample_drop_scales(self.geom) if m.training else None
        img_in, theta = self.x, None
        self._cnn_hip = m.with_land and self.cnn is not None and m.training
        if m.with_land:
            # trainable landmark regressor -> one-launch patch gather (ViT_face.py:679-711)
            theta = self.cnn.forward(self.x) if self._cnn_hip else m.landmarks(self.x)
            m.theta = theta
            th = theta.detach().contiguous()
            img_in = torch.empty_like(self.x)
            call("lafs_patch_gather_fwd", _p(self.x), _p(th), B, self.x.shape[-1], th.shape[1], _p(img_in))
        emb, st, _ = Fn.vit_forward(a, m._spec, self.geom, [img_in], [pos], drop, save=True, dropout=m._next_dropout())
        if self.head is not None:
            # class-sharded head: all-gather embeddings, local logits, exchanged softmax statistics, reduce-scatter of dE.
            # demb is the gradient of the GLOBAL-batch mean loss, so the later all-reduce of the backbone gradients is a SUM.
            loss, demb = self.head.forward_backward(emb, labels, grad_scale=1.0 / self.acc_step)
            self.loss = loss.detach().view(1)
            self._backward_trunk(st, demb, th if m.with_land else None, theta)
            self.micro += 1
            return self.loss
        # cosine logits
        xn = torch.empty(B, D, device=dev, dtype=bf16); inv_x = torch.empty(B, device=dev, dtype=f32)
        call("lafs_l2norm_fwd", _p(emb), D, _p(xn), D, _p(inv_x), B, D)
        wname = m._spec.prefix + "loss.weight"
        call("lafs_weightnorm_fwd", _p(a.view(a.master, wname)), _p(self.ones), self.C, self.Cpad, D, _p(self.wn), None, self.Cpad,
             _p(self.inv_w))
        ops.gemm_nt(xn, self.wn, _lib.EPI_F32, out=self.cos, n_cols=self.Cpad)
        # margin + softmax + soft-target CE, forward and d/dcos in place
        call("lafs_margin_softmax_ce", _p(self.cos), self.Cpad, B, self.C, _p(y1), _p(y2), lam, self.s, self.m, self.margin_type,
             1.0 / self.acc_step, _p(self.loss), _p(self.row_ws))
        ops.scale_cast_bf16(self.cos, out=self.dcos)
        # d(emb_n) = dcos @ Wn (reduction over the classes), d(Wn) = dcos^T @ emb_n (reduction over the batch)
        dxn = torch.zeros(B, D, device=dev, dtype=f32)
        ops.gemm_tn_acc(self.dcos.t().contiguous(), self.wn, dxn)
        self.dwn.zero_()
        ops.gemm_tn_acc(self.dcos, xn, self.dwn, splits=1)
        call("lafs_weightnorm_bwd", _p(self.dwn), _p(a.view(a.master, wname)), _p(self.ones), _p(self.inv_w), self.C, D,
             _p(a.view(a.grad, wname)), None, 1)
        demb = torch.empty(B, D, device=dev, dtype=f32)
        call("lafs_l2norm_bwd", _p(emb), D, _p(dxn), D, _p(inv_x), _p(demb), D, B, D)
        if self._reduce_now():
            self.reducer.launch(a.grad[self.head_off:])          # margin head (+ anything behind it): final from here on
        self._backward_trunk(st, demb, th if m.with_land else None, theta)
        self.micro += 1
        return self.loss

    def _reduce_now(self):
        """True on the micro-step that completes an accumulation window of a data-parallel run."""
        return self.world > 1 and (self.micro + 1) % self.acc_step == 0

    def _trunk_layers_backward(self, st, demb):
        """Final norm + all blocks; with DP on the window's last micro-step in `grad_slices` runs, each run's gradient range
        handed to RCCL as soon as it has been enqueued."""
        a, m = self.arena, self.model
        g = Fn.vit_backward_begin(a, m._spec, st, demb)
        depth = m.depth
        if not self._reduce_now():
            Fn.vit_backward_layers(st, g, depth, 0)
            return g
        ns = max(1, min(self.grad_slices, depth))
        cuts = [depth - (depth * k) // ns for k in range(ns + 1)]
        hi = self.head_off
        for k in range(ns):
            Fn.vit_backward_layers(st, g, cuts[k], cuts[k + 1])
            lo = self.block_off[cuts[k + 1]]
            if cuts[k + 1] > 0:                                  # the range below block 0 still waits for the embedding / CNN gradients
                self.reducer.launch(a.grad[lo:hi])
                hi = lo
        self._hi_left = hi
        return g

    def _backward_trunk(self, st, demb, th, theta):
        a, m, B, D = self.arena, self.model, self.B, self.D
        reduce_now = self._reduce_now()
        if m.with_land:
            g = self._trunk_layers_backward(st, demb)
            dpos, dx = Fn.vit_backward_end(a, m._spec, st, g, want_dx=True)
            dmosaic = Fn.unpatchify_grad(dx[0], m._spec.patch_order).contiguous()
            dth = torch.empty_like(th)
            call("lafs_patch_gather_bwd", _p(self.x), _p(th), _p(dmosaic), B, self.x.shape[-1], th.shape[1], _p(dth), None)
            if self._cnn_hip:
                self.cnn.backward(dth)                   # into the arena's stn.* / output_layer.* gradients
            else:
                theta.backward(dth)                      # torch autograd: p.grad are views of the arena
        else:
            g = self._trunk_layers_backward(st, demb)
            dpos = Fn.vit_backward_end(a, m._spec, st, g)
        a.view(a.grad, m._spec.prefix + "pos_embedding").view(-1, D)[: dpos[0].shape[0]] += dpos[0]
        if reduce_now:                                           # what is left: block 0's run + embedding + landmark CNN
            self.reducer.launch(a.grad[: self._hi_left])
            self._reduced = True

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

        def fill(h):
            h.zero_()
            h[_lib.HP_LR], h[_lib.HP_WD], h[_lib.HP_BETA1], h[_lib.HP_BETA2], h[_lib.HP_EPS] = lr, weight_decay, beta1, beta2, eps
            h[_lib.HP_GRAD_SCALE], h[_lib.HP_WD_LOW] = gscale, LOW_WEIGHT_DECAY
        self.hyper_ring.upload(self.hyper, fill)
        if self.head is not None:
            self.head.optimizer_step(lr, weight_decay, beta1, beta2, eps)
        call("lafs_clip_adamw_ema", _p(a.master), _p(a.grad), _p(a.exp_avg), _p(a.exp_avg_sq), None, _p(a.shadow), None,
             _p(a.chunk_seg), a.n_chunks, _p(a.seg_flags), _p(a.seg_step), a.n_seg, _p(a.seg_sumsq), _p(self.hyper))
        a.refresh_transposed()
        a.zero_grad()
        if self.cnn is not None:
            self.cnn.mark_stale()                        # the master weights changed in place: new operand images next forward

    def step(self, inputs_u8, labels, lr, weight_decay=0.1):
        """micro_step + optimizer step every acc_step micro-steps (train_largescale.py:842-891)."""
        loss = self.micro_step(inputs_u8, labels)
        if self.micro % self.acc_step == 0:
            self.optimizer_step(lr, weight_decay)
        return loss
