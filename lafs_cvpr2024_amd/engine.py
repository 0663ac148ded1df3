"""Fused LAFS pre-training step (reference lafs_train.py:513-613) for one MI355X rank.

    teacher fwd (global crops) -> student fwd (all crops, one packed pass) -> fused DINO loss fwd+bwd ->
    head bwd -> [RCCL all-reduce of head grads + center sum, overlapped with] trunk bwd -> [all-reduce trunk grads] ->
    per-tensor clip + AdamW + teacher EMA + bf16 shadow refresh (one launch) -> center EMA

Everything between the brackets is captured once into hipGraphs (static buffers, hyper-parameters in a device
buffer that is refreshed by one async H2D copy per step), so a step costs three graph launches instead of ~1500
kernel launches from Python.  Collectives stay outside the graphs and run through torch.distributed ("nccl" = RCCL).
"""
import math

import numpy as np
import os

import torch
import torch.distributed as dist

from . import _lib, functional as Fn, ops
from .arena import ParamArena
from .distributed import FlatReducer
from .ops import _p, call
from .utils import PinnedRing
from .vision_transformer import VisionTransformer, attach_arena, _is_matrix_for_dgrad


def _is_partfvit(m):
    from .face_pre_pro.ViT_face import ViT_face_landmark_patch8
    return isinstance(m, ViT_face_landmark_patch8)

f32, bf16 = torch.float32, torch.bfloat16


def _interp_matrix(grid_src, grid_dst, device):
    """Dense matrix of torch's bicubic F.interpolate from a g x g grid to an r x r grid with the reference's
    scale_factor = (r + 0.1) / g (vision_transformer.py:184-191): a fixed linear map, built once by pushing the
    identity through the same torch op, so pos tokens (and their gradient) become one small matmul."""
    n = grid_src * grid_src
    eye = torch.eye(n, device=device).view(1, n, grid_src, grid_src)
    sf = (grid_dst + 0.1) / grid_src
    out = torch.nn.functional.interpolate(eye, scale_factor=(sf, sf), mode="bicubic")
    assert out.shape[-1] == grid_dst and out.shape[-2] == grid_dst
    return out.view(n, grid_dst * grid_dst).t().contiguous()            # [r*r, g*g]


class LafsPretrainEngine:
    def __init__(self, student, teacher, dino_loss, batch_size, n_local=8, global_size=112, local_size=48,
                 clip_grad=3.0, freeze_last_layer=1, use_graph=True, device=None, grad_slices=4):
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        self.student, self.teacher, self.dino_loss = student, teacher, dino_loss
        self.B, self.n_local, self.ncrops = batch_size, n_local, 2 + n_local
        self.clip_grad, self.freeze_last_layer = float(clip_grad or 0.0), freeze_last_layer
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        vit_s, vit_t = student.backbone, teacher.backbone
        # Two backbones: the DINO VisionTransformer (NCHW crops, bicubic-resampled position table) and Part-fViT,
        # ViT_face_landmark_patch8 -- the pair the reference actually pre-trains (lafs_train.py:300-335, arch 'mynet'): landmark
        # mosaics as [B, n, 192] patch tokens, position table sliced to n+1 rows, DropPath 0.1 + dropout 0.1 live in the student
        # AND the teacher (the reference never calls teacher.eval()).
        self.partfvit = _is_partfvit(vit_s)
        if not (isinstance(vit_s, VisionTransformer) or self.partfvit) or type(vit_s) is not type(vit_t):
            raise _lib.LafsHipError("LafsPretrainEngine drives a VisionTransformer or a ViT_face_landmark_patch8 pair")
        if self.partfvit and (vit_s.with_land or vit_t.with_land):
            raise _lib.LafsHipError("LAFS pre-training uses with_land=False backbones (the landmark CNN is the frozen front-end)")
        self.sa = attach_arena(student, self.device)
        # the library's side streams / events / kernel options of THIS engine (created now: nothing is created inside a capture;
        # two engines share no stream or event).  The environment's A/B switches are read by _lib.Ctx, not by the library.
        self.ctx = _lib.Ctx(self.device)
        # The row chains (long / short sequences on two streams) pay on the ViT-S trunk, whose K = 384 kernels leave CUs idle; the
        # Part-fViT trunk's wide GEMMs run on one-workgroup-per-CU persistent tiles (gemm_big.hip) that want the whole chip and the
        # merged 44 160 rows (753 tiles of 176x256 = 2.94 rounds): `mynet` pair 36.1 ms with two chains, 35.4-35.5 with one.
        if self.partfvit and "LAFS_ROW_CHAINS" not in os.environ:
            self.ctx.set(_lib.OPT_ROW_CHAINS, 1)
        self.ta = getattr(teacher, "_lafs_arena", None)
        if self.ta is None:
            for p in teacher.parameters():
                p.requires_grad = False
            self.ta = ParamArena(teacher, self.device, with_grad=False, transposed=None)
            object.__setattr__(teacher, "_lafs_arena", self.ta)
            for name, m in teacher.named_modules():
                if hasattr(m, "_bind_arena"):
                    m._bind_arena(self.ta, name + "." if name else "")
        if self.sa.names != self.ta.names or self.sa.size != self.ta.size:
            raise _lib.LafsHipError("student and teacher must have identical parameter layouts")
        self.sa.ctx = self.ta.ctx = self.ctx
        dino_loss.to(self.device)
        if self.world > 1:                       # DDP's initial parameter broadcast (reference lafs_train.py:375)
            for t in (self.sa.master, self.ta.master, dino_loss.center):
                dist.broadcast(t, 0)
            self.sa.refresh_shadows(); self.ta.refresh_shadows()
        self.K = student.head.out_dim
        self.Kpad = (self.K + 127) // 128 * 128
        B, D = batch_size, (vit_s.dim if self.partfvit else vit_s.embed_dim)
        self.geom_s = Fn.geometry([(2 * B, global_size), (n_local * B, local_size)] if n_local else [(2 * B, global_size)], self.device)
        self.geom_t = Fn.geometry([(2 * B, global_size)], self.device)
        self.spec_s, self.spec_t = vit_s._spec, vit_t._spec
        self.head_prefix_s, self.head_prefix_t = student.head._prefix, teacher.head._prefix
        # bicubic resampling matrices for the stored position table (Part-fViT: plain slices, no resampling)
        self.grids = [global_size // 8] + ([local_size // 8] if n_local else [])
        if self.partfvit:
            if vit_s.num_patches < self.grids[0] ** 2:
                raise _lib.LafsHipError("pos_embedding is shorter than the global crops' token count")
            self.interp = [None for _ in self.grids]
        else:
            g = int(math.isqrt(vit_s.pos_embed.shape[1] - 1))
            self.interp = [None if r == g else _interp_matrix(g, r, self.device) for r in self.grids]
        # element dropout (Part-fViT: 0.1 live in student AND teacher, the reference never calls teacher.eval()): counter-based masks
        # whose seed is (network seed + 7919 * hyper[HP_STEP]), the step read on the DEVICE inside the kernels exactly as the
        # DropPath draw reads it -- a captured graph draws new masks on every replay, and the masks are indexed by absolute token
        # rows, so the row chains stay legal
        self.has_dropout = self.partfvit and (vit_s.dropout_rate > 0 or vit_s.emb_dropout_rate > 0 or
                                              vit_t.dropout_rate > 0 or vit_t.emb_dropout_rate > 0)
        self.dropout_seed_s, self.dropout_seed_t = 0x5EED, 0x7EAC4E5       # independent masks in the two networks
        # static buffers
        dev = self.device
        self.hyper = torch.zeros(_lib.HP_COUNT, device=dev, dtype=f32)
        self.hyper_ring = PinnedRing((_lib.HP_COUNT,), f32)   # the host runs steps ahead of the GPU: never reuse a pinned
        self.temps = torch.zeros(2, device=dev, dtype=f32)    # staging buffer whose copy may still be pending
        self.temps_ring = PinnedRing((2,), f32)
        # The last layer of both heads + the DINO loss + the centre sums as ONE fused group of launches (csrc/dino_head_loss.hip): the
        # logits are formed twice on the MFMA instead of written once and read twice (0.9 GB and ~0.2 ms per step at C2).  Needs the
        # DINOHead's default bottleneck (256) and at most 64 images per rank; LAFS_FUSED_HEAD=0 keeps the unfused kernels (A/B runs,
        # and what the tests that look at the logits themselves use: `logits_s` / `logits_t` only exist there).
        bott = student.head.last_layer.weight_v.shape[1]
        self.fused_head = (bott == 256 and 2 * B <= 128 and os.environ.get("LAFS_FUSED_HEAD", "1") != "0")
        self._logits_s = None if self.fused_head else torch.zeros(self.ncrops * B, self.Kpad, device=dev, dtype=f32)
        self._logits_t = None if self.fused_head else torch.zeros(2 * B, self.Kpad, device=dev, dtype=f32)
        self._head_states = None
        self.head_ws = (torch.empty(_lib.lib().lafs_dino_head_loss_workspace(self.ncrops, B, self.K), device=dev, dtype=f32)
                        if self.fused_head else None)
        self.dlogits = torch.zeros(self.ncrops * B, self.Kpad, device=dev, dtype=bf16)
        self.loss = torch.zeros(1, device=dev, dtype=f32)
        self.loss_ws = torch.empty(_lib.lib().lafs_dino_loss_workspace(self.ncrops, B, self.K), device=dev, dtype=f32)
        self.colsum = torch.zeros(self.K, device=dev, dtype=f32)
        self.in_global_all = torch.zeros(2 * B, 3, global_size, global_size, device=dev)
        self.in_local_all = torch.zeros(max(n_local, 1) * B, 3, local_size, local_size, device=dev)
        self.in_global = list(self.in_global_all.split(B))
        self.in_local = list(self.in_local_all.split(B))[:n_local]
        # DropPath scales are drawn by lafs_droppath_scales from (seed, hyper[HP_STEP]): new masks on every graph replay without
        # an ATen RNG kernel; keep probabilities per block on the device, None when every rate is 0
        def keep_probs(v):
            rates = [v.drop_path_rate] * v.depth if self.partfvit else list(v.drop_path_rates)
            return torch.tensor([1.0 - r for r in rates], device=dev, dtype=f32) if any(rates) else None
        self.keep_s, self.keep_t = keep_probs(vit_s), keep_probs(vit_t)
        self.drop_s = torch.empty(vit_s.depth, 2, self.geom_s.n_seq, device=dev, dtype=f32)
        self.drop_t = torch.empty(vit_t.depth, 2, self.geom_t.n_seq, device=dev, dtype=f32)
        self.drop_seed = 0x0D20FA7
        # resampled position tables (static: [1 + r*r, D] per crop size and network)
        self.pos_s = [torch.empty(r * r + 1, D, device=dev, dtype=f32) for r in self.grids]
        self.pos_t = [torch.empty(self.grids[0] ** 2 + 1, D, device=dev, dtype=f32)]
        # tensors whose gradient is WRITTEN by its first producer (block weights: wgrad fold with accumulate = 0; last layer:
        # weight-norm backward): the per-step zeroing skips them (lafs_zero_chunks)
        flags = self.sa.seg_flags.cpu().tolist()
        over = {nm[k] for nm in self.spec_s.trunk.block_names for k in ("w_qkv", "w_proj", "w_fc1", "w_fc2")}
        over.add(self.head_prefix_s + "last_layer.weight_v")
        for i, name in enumerate(self.sa.names):
            # (a frozen tensor's gradient is never written nor read: nothing to zero either)
            if name in over or not self.sa.params[i].requires_grad:
                flags[i] |= _lib.SEG_OVERWRITTEN
        self.sa.seg_flags.copy_(torch.tensor(flags, dtype=torch.int32))
        # gradient ranges for the two all-reduces: [trunk | head]
        self.head_start = min(o for n, o in self.sa.offsets.items() if n.startswith(self.head_prefix_s))
        self.depth = vit_s.depth
        self.pos_name = self.spec_s.pos
        # the trunk backward is cut in two graph segments so that the upper blocks' gradients are already on the wire
        # (RCCL) while the lower blocks are still being computed
        # the trunk backward is cut into `grad_slices` graph segments (blocks depth-1 .. 0 in equal runs) so that each run's
        # gradients are already on the wire (RCCL) while the next run is computed; only the last run's all-reduce is exposed
        ns = max(1, min(int(grad_slices), self.depth))
        self.cuts = [self.depth - (self.depth * k) // ns for k in range(ns + 1)]            # e.g. depth 12, 4 slices: 12 9 6 3 0
        off = lambda blk: self.sa.offsets[self.spec_s.trunk.block_names[blk]["ln1_g"]] if blk > 0 else 0
        self.cut_offsets = [off(c) for c in self.cuts[1:]]                                   # arena offset where each run starts
        self.reducer = FlatReducer()
        if self.world > 1:
            # LAFS_COMM_CUS=n (opt-in, default 0) shrinks the K-resident GEMM's grid by 2 n workgroups while gradients are on the
            # wire.  It does not reserve CUs (the dispatcher still spreads the remaining workgroups over the chip) and has never been
            # A/B-measured beside RCCL kernels, so it stays off until a multi-GPU box shows a gain.
            self.ctx.set(_lib.OPT_COMM_CUS, int(os.environ.get("LAFS_COMM_CUS", "0")))
        # Update pieces: ranges of the arena in the order their gradients become final -- the DINO head (31 of 53 M parameters at C2)
        # after the head backward, then each run of blocks after its segment of the trunk backward.  A piece's per-tensor norms,
        # clip + AdamW + teacher EMA + shadow refresh run on a stream of their own at the START of a later segment, beside that
        # segment's GEMM-heavy work, instead of as a serial HBM-bound tail of the step (the reference's optimizer.step() /
        # EMA loop, lafs_train.py:606-613, touches nothing the backward still reads: a block's weights are dead once its
        # gradients exist).  Data-parallel runs give every piece one segment of slack for its all-reduce.
        bounds = [self.sa.size, self.head_start] + list(self.cut_offsets)
        self.pieces = [(bounds[i + 1], bounds[i]) for i in range(len(bounds) - 1) if bounds[i] > bounds[i + 1]]
        seg_at = {self.sa.offsets[n]: i for i, n in enumerate(self.sa.names)}
        seg_at[self.sa.size] = self.sa.n_seg
        self.piece_segs = [(seg_at[lo], seg_at[hi]) for lo, hi in self.pieces]
        chunk_seg = self.sa.chunk_seg.cpu()
        for (lo, hi), (s_lo, s_hi) in zip(self.pieces, self.piece_segs):        # a piece is whole tensors: its first / last chunk carry its first / last segment
            assert lo % _lib.CHUNK == 0 and hi % _lib.CHUNK == 0 and 0 <= s_lo < s_hi <= self.sa.n_seg
            assert int(chunk_seg[lo // _lib.CHUNK]) == s_lo and int(chunk_seg[hi // _lib.CHUNK - 1]) == s_hi - 1, (lo, hi, s_lo, s_hi)
        n_segments = len(self.cuts) + 1                     # forward, the runs of the trunk backward, the final update
        slack = 1 if self.world > 1 else 0
        serial = os.environ.get("LAFS_OPT_OVERLAP", "1") == "0"          # A/B knob: every piece in the final segment
        self.piece_runs_at = [n_segments - 1 if serial else min(p + 1 + slack, n_segments - 1) for p in range(len(self.pieces))]
        self._piece_tokens = [[] for _ in self.pieces]
        self.opt_stream = None if os.environ.get("LAFS_SINGLE_STREAM") == "1" else torch.cuda.Stream(device=self.device)
        # large frozen tensors (Part-fViT's 30000 x 768 CosFace table: 92 MB of zeros per step on the wire otherwise) are cut out
        # of the all-reduced ranges; small frozen ones (weight_g) ride along rather than splitting a collective
        self._frozen_runs = [(self.sa.offsets[n], self.sa.offsets[n] + (p.numel() + _lib.CHUNK - 1) // _lib.CHUNK * _lib.CHUNK)
                             for n, p in zip(self.sa.names, self.sa.params) if not p.requires_grad and p.numel() >= (1 << 20)]
        # teacher forward / weight-gradient GEMMs run on a second stream; LAFS_SINGLE_STREAM=1 serialises everything (profiling)
        self.side_stream = None if os.environ.get("LAFS_SINGLE_STREAM") == "1" else torch.cuda.Stream(device=self.device)
        self.use_graph = use_graph
        self._graphs = None
        self._st = {}
        self.step_count = 0

    def _logits(self, which):
        """The last step's logits [rows, Kpad] (f32).  With the fused head they are never stored: formed here on demand from the
        last layer's saved operands (tests / diagnostics only)."""
        if not self.fused_head:
            return self._logits_s if which == 0 else self._logits_t
        if self._head_states is None:
            raise _lib.LafsHipError("no step has run yet")
        st = self._head_states[which]
        return ops.gemm_nt(st.zn, st.wn, _lib.EPI_F32, n_cols=st.Kpad)

    logits_s = property(lambda self: self._logits(0))
    logits_t = property(lambda self: self._logits(1))

    # ------------------------------------------------------------------ pieces (all capturable)
    def _pos_tokens(self, arena, spec, bufs):
        pe = arena.view(arena.master, spec.prefix + spec.pos).view(-1, spec.trunk.dim)
        out = []
        for M, r, buf in zip(self.interp, self.grids, bufs):
            if self.partfvit:
                out.append(pe[:r * r + 1])                 # pos_embedding[:, :n+1] (reference ViT_face.py:766)
            elif M is None:
                out.append(pe)
            else:                                          # bicubic resampling as its fixed linear map, one small launch
                call("lafs_pos_interp_fwd", _p(pe), _p(M), _p(buf), M.shape[0], M.shape[1], spec.trunk.dim)
                out.append(buf)
        return out

    def set_droppath_scales(self, student=None, teacher=None):
        """Parity hook: use the given stochastic-depth scales ([depth, 2, n_seq] of 0 or 1/keep, sequences in packed order: global
        crops first) instead of drawing them on the device -- the tests hand the engine the masks the reference drew (F17).  None
        restores the device draw.  Must be set before the step is captured."""
        if self._graphs is not None:
            raise _lib.LafsHipError("set_droppath_scales after the step has been captured")
        dev = lambda t, like: None if t is None else t.to(self.device, f32).reshape(like.shape).contiguous()
        self._drop_fixed = {0: dev(student, self.drop_s), 1: dev(teacher, self.drop_t)}

    def _drop_scales(self, keep, out, salt):
        fixed = getattr(self, "_drop_fixed", {}).get(salt)
        if fixed is not None:
            out.copy_(fixed)
            return out
        if keep is None:
            return None
        call("lafs_droppath_scales", _p(keep), out.shape[0], out.shape[2], (self.drop_seed + salt) & 0xFFFFFFFF,
             _p(self.hyper[_lib.HP_STEP:]), _p(out))
        return out

    def _dropout_cfg(self, vit, seed):
        """(p_trunk, p_embedding, seed, device step counter) of a Part-fViT network in training mode, else None."""
        if not self.partfvit or not vit.training or (vit.dropout_rate == 0.0 and vit.emb_dropout_rate == 0.0):
            return None
        return (vit.dropout_rate, vit.emb_dropout_rate, seed, self.hyper[_lib.HP_STEP:])

    def _seg_forward(self):
        sa, ta, B = self.sa, self.ta, self.B
        call("lafs_zero_chunks", _p(sa.grad), _p(sa.chunk_seg), _p(sa.seg_flags), sa.n_chunks, _lib.SEG_OVERWRITTEN)
        # teacher (two global views, no activations kept) runs on the side stream, concurrently with the student
        cur = torch.cuda.current_stream()
        side = self.side_stream if self.side_stream is not None else cur
        if os.environ.get("LAFS_TEACHER_SERIAL") == "1":       # lab knob: teacher forward in front of the student on the main stream
            side = cur
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            pos_t = self._pos_tokens(ta, self.spec_t, self.pos_t)[:1]
            vit_t = self.teacher.backbone
            drop_t = self._drop_scales(self.keep_t, self.drop_t, 1) if vit_t.training else None   # rate 0 for the DINO ViT teacher
            dd_t = self._dropout_cfg(vit_t, self.dropout_seed_t)
            feat_t, _, _ = Fn.vit_forward(ta, self.spec_t, self.geom_t, [self.in_global_all], pos_t, drop_t, save=False, dropout=dd_t)
            _, st_ht = Fn.head_forward(ta, self.head_prefix_t, feat_t, self.K, save=False, logits=self._logits_t, skip_logits=self.fused_head)
        # student: all views in one packed pass
        vit = self.student.backbone
        drop = self._drop_scales(self.keep_s, self.drop_s, 0) if vit.training else None
        imgs = [self.in_global_all] + ([self.in_local_all] if self.n_local else [])
        dd_s = self._dropout_cfg(vit, self.dropout_seed_s)
        feat_s, st_v, _ = Fn.vit_forward(sa, self.spec_s, self.geom_s, imgs, self._pos_tokens(sa, self.spec_s, self.pos_s), drop,
                                         save=True, dropout=dd_s, wgrad_overwrite=True,
                                         wgrad_workgroups=int(os.environ.get("LAFS_WGRAD_WG", 200)) if self.side_stream is not None else 0)
        _, st_h = Fn.head_forward(sa, self.head_prefix_s, feat_s, self.K, save=True, logits=self._logits_s, skip_logits=self.fused_head)
        cur.wait_stream(side)
        if self.fused_head:
            # last layer of both heads + loss forward + dL/dlogits + center column sums, the logits never stored
            ops.dino_head_loss(st_h.zn, st_ht.zn, st_h.wn, st_ht.wn, self.dino_loss.center.view(-1), self.ncrops, self.K,
                               float(self.dino_loss.student_temp), 0.04, grad=self.dlogits, loss=self.loss, colsum=self.colsum,
                               ws=self.head_ws, dev_temps=self.temps)
        else:
            # loss forward + dL/dlogits in the same two passes; center column sums of the raw teacher logits
            ops.dino_loss_fwd_bwd(self._logits_s, self._logits_t, self.dino_loss.center.view(-1), self.ncrops,
                                  float(self.dino_loss.student_temp), 0.04, K=self.K, grad=self.dlogits, ws=self.loss_ws,
                                  loss=self.loss, dev_temps=self.temps)
            call("lafs_colsum_f32", _p(self._logits_t), self.Kpad, 2 * B, self.K, _p(self.colsum))
        train_g = self.student.head.last_layer.weight_g.requires_grad
        dfeat = Fn.head_backward(sa, self.head_prefix_s, st_h, self.dlogits, train_g=train_g, overwrite_last=True)
        self._st = dict(vit=st_v, dfeat=dfeat)
        self._head_states = (st_h, st_ht)

    def _seg_trunk_backward(self, k):
        """Run k of the trunk backward: blocks cuts[k]-1 .. cuts[k+1]; run 0 starts with the final norm, the last run ends with
        the patch embedding and the position table."""
        sa = self.sa
        if k == 0:
            self._st["g"] = Fn.vit_backward_begin(sa, self.spec_s, self._st["vit"], self._st["dfeat"])
        Fn.vit_backward_layers(self._st["vit"], self._st["g"], self.cuts[k], self.cuts[k + 1], wgrad_stream=self.side_stream)
        if k == len(self.cuts) - 2:
            gpe = sa.view(sa.grad, self.spec_s.prefix + self.pos_name).view(-1, self.spec_s.trunk.dim)
            # a table that is used as stored (Part-fViT slices, or a crop size equal to the table's grid) takes its gradient rows
            # directly; a resampled one goes through the transposed interpolation map
            direct = [gpe[:r * r + 1] if (self.partfvit or M is None) else None for M, r in zip(self.interp, self.grids)]
            dpos = Fn.vit_backward_end(sa, self.spec_s, self._st["vit"], self._st["g"], dpos_out=direct)
            for M, dp, dr in zip(self.interp, dpos, direct):
                if dr is None:
                    call("lafs_pos_interp_bwd", _p(dp), _p(M), _p(gpe), M.shape[0], M.shape[1], self.spec_s.trunk.dim)

    def _update_piece(self, p):
        """Per-tensor gradient norms, clip + AdamW + teacher EMA + bf16 shadows for the tensors of update piece p."""
        sa, ta = self.sa, self.ta
        (lo, hi), (s_lo, s_hi) = self.pieces[p], self.piece_segs[p]
        c_lo, c_hi = lo // _lib.CHUNK, hi // _lib.CHUNK
        call("lafs_grad_sumsq_range", _p(sa.grad), _p(sa.chunk_seg), sa.n_chunks, sa.n_seg, c_lo, c_hi, s_lo, s_hi, _p(self.hyper),
             _p(sa.chunk_sumsq), _p(sa.seg_sumsq))
        call("lafs_clip_adamw_ema_range", _p(sa.master), _p(sa.grad), _p(sa.exp_avg), _p(sa.exp_avg_sq), _p(ta.master),
             _p(sa.shadow), _p(ta.shadow), _p(sa.chunk_seg), sa.n_chunks, c_lo, c_hi, _p(sa.seg_flags), _p(sa.seg_step), sa.n_seg,
             s_lo, s_hi, _p(sa.seg_sumsq), _p(self.hyper))

    def _seg_update(self):
        call("lafs_center_ema", _p(self.dino_loss.center), _p(self.colsum), self.K, 1.0 / (2 * self.B * self.world),
             float(self.dino_loss.center_momentum))

    def _with_pieces(self, k, body):
        """Segment k = `body` with the update pieces scheduled at k: on their own stream beside the body (forked at the start, joined
        at the end: the pattern hipGraph captures), or -- in the final segment -- in line, followed by the W^T shadow refresh."""
        here = [p for p, at in enumerate(self.piece_runs_at) if at == k]
        last = k == len(self.cuts)
        cur = torch.cuda.current_stream()
        if here and not last and self.opt_stream is not None:
            self.opt_stream.wait_stream(cur)
            with torch.cuda.stream(self.opt_stream):
                for p in here:
                    self._update_piece(p)
            body()
            cur.wait_stream(self.opt_stream)
            return
        body()
        for p in here:
            self._update_piece(p)
        if last:
            self.sa.refresh_transposed()

    # ------------------------------------------------------------------ step
    def _segments(self):
        bodies = [self._seg_forward] + [(lambda k=k: self._seg_trunk_backward(k)) for k in range(len(self.cuts) - 1)] + [self._seg_update]
        return [(lambda k=k, b=b: self._with_pieces(k, b)) for k, b in enumerate(bodies)]

    def _capture(self):
        segs = self._segments()
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                       # warm-up outside capture (lazy inits, caches)
            for f in segs:
                f()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graphs, pool = [], None
        # a single rank has no collective to place between the segments: the whole step is ONE graph (five replay boundaries less)
        self._one_graph = (self.world == 1) and os.environ.get("LAFS_ONE_GRAPH", "1") != "0"
        if self._one_graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for f in segs:
                    f()
            self._graphs = [g]
            return
        # world > 1: the process group's watchdog thread polls its events while this thread captures -- thread-local capture mode
        # keeps another thread's (legal) runtime calls from invalidating the capture
        for f in segs:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                f()
            pool = g.pool()
            graphs.append(g)
        self._graphs = graphs

    def set_inputs(self, crops):
        """crops: list of 2 global + n_local local crops (any device), copied into the static NCHW input buffers.  Each is an
        NCHW image or, as the reference feeds its Part-fViT pair (lafs_train.py:538-569), a [B, n, 192] tensor of '(p1 p2 c)'
        patch vectors of the landmark mosaic -- the same pixels in another order, re-indexed here."""
        for dst, src in zip(self.in_global + self.in_local, crops):
            if src.dim() == 3:
                src = Fn.unpatchify_grad(src, _lib.PATCH_ORDER_HWC)
            dst.copy_(src, non_blocking=True)

    def step(self, crops=None, *, lr, wd, momentum, teacher_temp, epoch, beta1=0.9, beta2=0.999, eps=1e-8):
        """One optimisation step.  Returns the device scalar loss (no host sync)."""
        if crops is not None:
            self.set_inputs(crops)
        def fill_hyper(h):
            h[_lib.HP_LR], h[_lib.HP_WD], h[_lib.HP_BETA1], h[_lib.HP_BETA2], h[_lib.HP_EPS] = lr, wd, beta1, beta2, eps
            h[_lib.HP_CLIP], h[_lib.HP_EMA_M] = self.clip_grad, momentum
            h[_lib.HP_FREEZE_LAST] = 1.0 if epoch < self.freeze_last_layer else 0.0
            h[_lib.HP_GRAD_SCALE] = 1.0 / self.world        # DDP mean of the summed gradients
            h[_lib.HP_STEP] = float(self.step_count % (1 << 24))

        def fill_temps(t):
            t[0], t[1] = float(self.dino_loss.student_temp), teacher_temp

        self.hyper_ring.upload(self.hyper, fill_hyper)
        self.temps_ring.upload(self.temps, fill_temps)
        if self.use_graph and self._graphs is None:
            saved = self._snapshot()
            self._capture()
            self._restore(saved)
            self.hyper_ring.upload(self.hyper, fill_hyper)
        if self.use_graph and getattr(self, "_one_graph", False):
            self._graphs[0].replay()
            self.step_count += 1
            return self.loss
        segs = self._segments()
        replay = (lambda i: self._graphs[i].replay()) if self.use_graph else (lambda i: segs[i]())

        def run(i):                              # a segment's update pieces need their gradients reduced: wait (stream-side) first
            for p, at in enumerate(self.piece_runs_at):
                if at == i:
                    self.reducer.wait(self._piece_tokens[p])
            replay(i)
        run(0)
        # head gradients + center sums go out over RCCL while the trunk backward runs; each run of blocks follows as soon as
        # its graph segment has been enqueued (arena order: [embed | blocks 0..depth-1 | norm | head])
        self._piece_tokens[0] = self._reduce_range(*self.pieces[0])
        center_tok = self.reducer.launch(self.colsum)
        for k in range(len(self.cuts) - 1):
            run(1 + k)
            if 1 + k < len(self.pieces):
                self._piece_tokens[1 + k] = self._reduce_range(*self.pieces[1 + k])
        self.reducer.wait([center_tok])
        run(len(segs) - 1)
        self.reducer.wait_all()
        self.step_count += 1
        return self.loss

    def _reduce_range(self, lo, hi):
        """Asynchronous SUM all-reduce of grad[lo:hi) minus the large frozen tensors inside it; returns the reducer's tokens."""
        return [self.reducer.launch(self.sa.grad[a:b]) for a, b in self._trainable_runs(lo, hi)]

    def _trainable_runs(self, lo, hi):
        runs, cur = [], lo
        for a, b in sorted(self._frozen_runs):
            if b <= cur or a >= hi:
                continue
            if a > cur:
                runs.append((cur, min(a, hi)))
            cur = max(cur, b)
        if cur < hi:
            runs.append((cur, hi))
        return runs

    # warm-up/capture executes real optimisation steps on whatever is in the buffers: snapshot and restore the state
    def _snapshot(self):
        sa, ta = self.sa, self.ta
        return [t.clone() for t in (sa.master, sa.exp_avg, sa.exp_avg_sq, sa.seg_step, ta.master, self.dino_loss.center)], \
            torch.cuda.get_rng_state(self.device)

    def _restore(self, saved):
        tensors, rng = saved
        sa, ta = self.sa, self.ta
        for dst, src in zip((sa.master, sa.exp_avg, sa.exp_avg_sq, sa.seg_step, ta.master, self.dino_loss.center), tensors):
            dst.copy_(src)
        sa.refresh_shadows()
        ta.refresh_shadows()
        torch.cuda.set_rng_state(rng, self.device)

    # ------------------------------------------------------------------ checkpoint helpers (reference layout)
    def _adamw_order(self):
        """Parameter names in torch.optim.AdamW(utils.get_params_groups(student)) index order: the regularised group first
        (>= 2-D non-bias tensors), then the rest, each in named_parameters order (reference lafs_train.py:385-392,
        utils.py:662-673).  The reference registers every tensor with requires_grad=True -- including Part-fViT's CosFace
        `loss.weight`, which the SSL step never touches (gradient None: AdamW keeps no state for it, but it holds an index in the
        regularised group).  Here such a tensor is frozen in the arena and marked `_lafs_optimizer_registered` by
        build_backbones: it keeps its index and never gets a state entry."""
        reg, noreg = [], []
        for name, p in self.student.named_parameters():
            if p.requires_grad or getattr(p, "_lafs_optimizer_registered", False):
                (noreg if (name.endswith(".bias") or p.dim() == 1) else reg).append(name)
        return reg, noreg

    def optimizer_state_dict(self):
        """The state_dict torch.optim.AdamW(get_params_groups(student)) would have at this point -- the format the reference
        stores under checkpoint['optimizer'] and restores with optimizer.load_state_dict (lafs_train.py:428-463,
        utils.py:152-184): per-parameter {step, exp_avg, exp_avg_sq} (only for tensors that have been stepped, like torch:
        the last layer has none while it is frozen) and the two param_groups."""
        sa = self.sa
        reg, noreg = self._adamw_order()
        steps = sa.seg_step.cpu().tolist()
        h = self.hyper.cpu().tolist()
        state = {}
        for idx, name in enumerate(reg + noreg):
            t = steps[sa.names.index(name)]
            if t > 0:
                shape = sa.params[sa.names.index(name)].shape
                state[idx] = {"step": torch.tensor(float(t)), "exp_avg": sa.view(sa.exp_avg, name, shape).detach().cpu().clone(),
                              "exp_avg_sq": sa.view(sa.exp_avg_sq, name, shape).detach().cpu().clone()}
        common = {"lr": h[_lib.HP_LR], "betas": (h[_lib.HP_BETA1] or 0.9, h[_lib.HP_BETA2] or 0.999), "eps": h[_lib.HP_EPS] or 1e-8,
                  "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None}
        groups = [dict(common, weight_decay=h[_lib.HP_WD], params=list(range(len(reg)))),
                  dict(common, weight_decay=0.0, params=list(range(len(reg), len(reg) + len(noreg))))]
        return {"state": state, "param_groups": groups}

    def load_optimizer_state_dict(self, sd):
        """Inverse of optimizer_state_dict; also accepts a checkpoint written by the reference itself (same format)."""
        if "state" not in sd or "param_groups" not in sd:
            raise _lib.LafsHipError("checkpoint['optimizer'] is not a torch.optim.AdamW state_dict (keys: %s)" % sorted(sd))
        sa = self.sa
        reg, noreg = self._adamw_order()
        names = reg + noreg
        sizes = [len(g["params"]) for g in sd["param_groups"]]
        if sizes != [len(reg), len(noreg)]:
            raise _lib.LafsHipError(f"optimizer param_groups of sizes {sizes} do not fit this model ({len(reg)} regularised + "
                                    f"{len(noreg)} other tensors)")
        sa.exp_avg.zero_(); sa.exp_avg_sq.zero_(); sa.seg_step.zero_()
        steps = sa.seg_step.cpu()
        flat = [i for g in sd["param_groups"] for i in g["params"]]
        for pos, idx in enumerate(flat):
            st = sd["state"].get(idx)
            if st is None:
                continue
            name = names[pos]
            shape = sa.params[sa.names.index(name)].shape
            sa.view(sa.exp_avg, name, shape).copy_(st["exp_avg"]); sa.view(sa.exp_avg_sq, name, shape).copy_(st["exp_avg_sq"])
            steps[sa.names.index(name)] = int(float(st["step"]))
        sa.seg_step.copy_(steps)
        # the per-step DropPath masks are seeded with the step count: continue the sequence of the interrupted run
        self.step_count = int(steps.max()) if steps.numel() else 0
