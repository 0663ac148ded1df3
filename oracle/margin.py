"""Oracle: margin-softmax heads, soft-target CE and batch mixup (fp32 CPU).  Test infrastructure only."""
import numpy as np
import torch
import torch.nn.functional as F


def cosface_logits(x, weight, label, s=64.0, m=0.4):
    """CosFace.forward (face_pre_pro/ViT_face.py:49-89): s * (cos(x, W) - m * y).

    ``label`` is [B] integer class ids or a dense [B, C] soft target (the mixup branch
    :69-73 multiplies the margin by the soft label itself)."""
    cos = F.linear(F.normalize(x), F.normalize(weight))
    if label.dim() > 1:
        y = label.to(cos.dtype)
    else:
        y = torch.zeros_like(cos).scatter_(1, label.view(-1, 1).long(), 1.0)
    return s * (y * (cos - m) + (1.0 - y) * cos)


def arcface_logits(x, weight, label, s=64.0, m=0.5):
    """ArcFace s*cos(theta + m) on the target class.  PARITY UNPINNED: the reference only
    names the class (ViT_face.py:416-417, 654-655) and never defines it; this follows
    Deng et al. 2019 with the InsightFace defaults, hard labels only."""
    cos = F.linear(F.normalize(x), F.normalize(weight)).clamp(-1, 1)
    theta = torch.acos(cos)
    y = torch.zeros_like(cos).scatter_(1, label.view(-1, 1).long(), 1.0)
    return s * torch.where(y > 0, torch.cos(theta + m), cos)


def soft_target_cross_entropy(logits, target):
    """timm SoftTargetCrossEntropy (external package; call site train_largescale.py:602, 820):
    mean_b sum_k -y_k log_softmax(x)_k."""
    return torch.sum(-target * F.log_softmax(logits, dim=-1), dim=-1).mean()


def mixup_batch(x, target, num_classes, lam, smoothing=0.0):
    """Batch-mode mixup given lambda (util/mixup_my.py:189-200, 18-24, 202-211).

    x is mixed in place with its batch-flip when lam != 1; the dense target is
    lam*onehot(y) + (1-lam)*onehot(flip(y)) with label smoothing folded in."""
    if lam != 1.0:
        xf = x.flip(0).mul_(1.0 - lam)
        x.mul_(lam).add_(xf)
    off = smoothing / num_classes
    on = 1.0 - smoothing + off
    oh = lambda t: torch.full((t.numel(), num_classes), off).scatter_(1, t.long().view(-1, 1), on)
    return x, oh(target) * lam + oh(target.flip(0)) * (1.0 - lam)


def draw_mixup_lambda(rng: np.random.RandomState, mixup_alpha=0.2, prob=0.1):
    """lambda draw of _params_per_batch for mixup-only (util/mixup_my.py:134-150):
    one uniform for the apply decision, then Beta(alpha, alpha)."""
    if rng.rand() < prob:
        return float(rng.beta(mixup_alpha, mixup_alpha))
    return 1.0


def partial_fc_reference(emb, weight, labels, s=64.0, m=0.4):
    """Unsharded full-FC CosFace + hard-label CE, the self-check target for the sharded
    PartialFC path at sample_rate=1.  PARITY UNPINNED (PartialFC is absent from the
    reference: only a commented import, ViT_face.py:645-649)."""
    logits = cosface_logits(emb, weight, labels, s, m)
    return F.cross_entropy(logits, labels.long())


def partial_fc_sharded(emb_local, labels_local, weight_shard, class_start, s=64.0, m=0.4,
                       all_gather=None, all_reduce_max=None, all_reduce_sum=None, labels2_local=None, lam_local=None):
    """Class-sharded CosFace + CE with the distributed softmax of InsightFace partial_fc_v2
    (DistCrossEntropy), sample_rate = 1.  PARITY UNPINNED (see partial_fc_reference).
    The three callables perform the cross-rank exchange (identity when None).  Returns the
    mean loss over the GLOBAL batch and d loss / d emb restricted to this shard's classes for
    ALL rows (the caller reduce-scatters it).  labels2_local / lam_local: the mixup partners'
    classes and this rank's lambda -- the dense soft target lam e_y1 + (1-lam) e_y2 of
    util/mixup_my.py:18-24 entering the margin as CosFace.forward's soft branch does
    (ViT_face.py:69-73), loss = soft-target CE."""
    ident = lambda t: t
    all_gather = all_gather or ident
    all_reduce_max = all_reduce_max or ident
    all_reduce_sum = all_reduce_sum or ident
    E = all_gather(emb_local).detach().requires_grad_(True)
    L = all_gather(labels_local).long()
    n_local = weight_shard.shape[0]
    cos = F.linear(F.normalize(E), F.normalize(weight_shard))

    def onehot(lab):
        own = (lab >= class_start) & (lab < class_start + n_local)
        y = torch.zeros_like(cos)
        y[own, (lab - class_start)[own]] = 1.0
        return y
    y = onehot(L)
    if labels2_local is not None:
        L2 = all_gather(labels2_local).long()
        lam = all_gather(torch.full((emb_local.shape[0],), float(lam_local)))
        y = lam[:, None] * y + (1.0 - lam)[:, None] * onehot(L2)
    z = s * (cos - m * y)
    gmax = all_reduce_max(z.detach().max(dim=1).values)
    ez = torch.exp(z - gmax[:, None])
    Z = all_reduce_sum(ez.detach().sum(dim=1))
    tgt = all_reduce_sum((z.detach() * y).sum(dim=1))
    loss = (torch.log(Z) + gmax - tgt).mean()
    dz = (ez.detach() / Z[:, None] - y) / E.shape[0]
    z.backward(dz)
    return loss, E.grad
