"""Oracle: one full LAFS pre-training step on CPU in fp32 (lafs_train.py:513-613).

teacher fwd (2 global crops) -> student fwd (all crops) -> DINO loss -> backward ->
per-tensor clip -> cancel last-layer grads while frozen -> AdamW (2 groups) -> teacher EMA ->
center update.  Test infrastructure only (also the `cpu_baseline` leg of bench.py).
"""
import torch

from . import dino, optim, partfvit, vit


class LafsState:
    """Student/teacher parameters ('backbone.*' / 'head.*' keys as in the checkpoint layout,
    lafs_train.py:451-460), DINO center and AdamW moments."""

    def __init__(self, cfg, out_dim: int, seed: int = 0, norm_last_layer=True,
                 hidden_dim=2048, bottleneck_dim=256):
        """cfg: vit.ViTConfig (DINO ViT) or partfvit.PartFViTConfig (the reference's actual `mynet` pair,
        lafs_train.py:300-335: ViT_face_landmark_patch8 fed [B, n, 192] patch tokens)."""
        g = torch.Generator().manual_seed(seed)
        self.cfg, self.out_dim, self.norm_last_layer = cfg, out_dim, norm_last_layer
        if isinstance(cfg, partfvit.PartFViTConfig):
            sb = partfvit.init_params(cfg, g)
            sh = vit.init_head_params(cfg.dim, out_dim, g, hidden_dim, bottleneck_dim)
        else:
            sb = vit.init_vit_params(cfg, g)
            sh = vit.init_head_params(cfg.embed_dim, out_dim, g, hidden_dim, bottleneck_dim)
        self.student = {**{"backbone." + k: v for k, v in sb.items()},
                        **{"head." + k: v for k, v in sh.items()}}
        # teacher.load_state_dict(student.state_dict())  (lafs_train.py:377)
        self.teacher = {k: v.clone() for k, v in self.student.items()}
        self.center = torch.zeros(1, out_dim)
        self.exp_avg = {k: torch.zeros_like(v) for k, v in self.student.items()}
        self.exp_avg_sq = {k: torch.zeros_like(v) for k, v in self.student.items()}
        self.steps = {k: 0 for k in self.student}

    def trainable(self, k):
        return not (self.norm_last_layer and k.endswith("last_layer.weight_g"))


def _split(P):
    return ({k[len("backbone."):]: v for k, v in P.items() if k.startswith("backbone.")},
            {k[len("head."):]: v for k, v in P.items() if k.startswith("head.")})


def multicrop_forward(Pb, Ph, crops, cfg, drop_scales=None):
    """MultiCropWrapper.forward (utils.py:610-659) over either backbone: one pass per run of equal-resolution crops
    (4-D: last dim; 3-D patch tokens: token count), one head pass."""
    if not isinstance(cfg, partfvit.PartFViTConfig):
        return vit.multicrop_forward(Pb, Ph, crops, cfg, drop_scales)
    feats, start = [], 0
    for gi, end in enumerate(vit.crop_groups(crops)):
        ds = None if drop_scales is None else drop_scales[gi]
        feats.append(partfvit.forward_embedding(Pb, torch.cat(crops[start:end]), cfg, ds))
        start = end
    return vit.dino_head_forward(Ph, torch.cat(feats))


def lafs_step(st: LafsState, crops, *, epoch, lr, wd, momentum, teacher_temp, clip_grad=3.0,
              freeze_last_layer=1, student_temp=0.1, center_momentum=0.9, drop_scales=None,
              world_size=1, all_reduce=None, teacher_drop_scales=None):
    """Runs one step in place on ``st``.  ``crops``: list of NCHW tensors, the first two are the
    global views.  ``drop_scales``: None or list (per resolution group) of [depth,2,B_group]
    stochastic-depth scales for the STUDENT (the DINO ViT teacher is built with rate 0).
    Returns dict(loss, grads (post-clip), norms (pre-clip))."""
    ncrops = len(crops)
    with torch.no_grad():
        tb, th = _split(st.teacher)
        # the reference never calls teacher.eval(): a Part-fViT teacher keeps its DropPath 0.1 live (teacher_drop_scales)
        t_out = multicrop_forward(tb, th, crops[:2], st.cfg, teacher_drop_scales)
    leaves = {k: v.detach().clone().requires_grad_(st.trainable(k)) for k, v in st.student.items()}
    sb, sh = _split(leaves)
    s_out = multicrop_forward(sb, sh, crops, st.cfg, drop_scales)
    loss = dino.dino_loss(s_out, t_out, st.center, ncrops, teacher_temp, student_temp)
    loss.backward()

    names = [k for k in leaves if leaves[k].grad is not None]
    grads = {k: leaves[k].grad for k in names}
    if all_reduce is not None:                       # DDP gradient mean (lafs_train.py:375)
        for g in grads.values():
            all_reduce(g)
            g.div_(world_size)
    norms = dict(zip(names, optim.clip_gradients_([grads[k] for k in names], clip_grad))) if clip_grad else {}
    if epoch < freeze_last_layer:                    # utils.py:144-149
        for k in list(grads):
            if "last_layer" in k:
                del grads[k]
    with torch.no_grad():
        for k, g in grads.items():
            st.steps[k] += 1
            this_wd = wd if optim.is_regularized(k, st.student[k].shape) else 0.0
            optim.adamw_step_(st.student[k], g, st.exp_avg[k], st.exp_avg_sq[k], st.steps[k], lr, this_wd)
        for k in st.student:                         # every parameter incl. frozen weight_g
            optim.ema_(st.teacher[k], st.student[k], momentum)
        st.center = dino.update_center(st.center, t_out, center_momentum, world_size, all_reduce)
    return {"loss": loss.detach(), "grads": grads, "norms": norms, "teacher_out": t_out,
            "student_out": s_out.detach()}
