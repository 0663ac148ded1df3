"""Oracle: step glue -- schedules, parameter groups, per-tensor clip, AdamW, EMA (fp32 CPU).

Test infrastructure only.
"""
import math

import numpy as np
import torch


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """Linear warm-up then half-cosine, one value per iteration (utils.py:187-198)."""
    warm_iters = warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warm_iters) if warmup_epochs > 0 else np.array([])
    n = epochs * niter_per_ep - warm_iters
    it = np.arange(n)
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * it / n))
    out = np.concatenate((warm, sched))
    assert len(out) == epochs * niter_per_ep
    return out


def is_regularized(name, shape):
    """Weight-decay membership rule of get_params_groups (utils.py:662-673):
    biases and 1-D tensors are not decayed."""
    return not (name.endswith(".bias") or len(shape) == 1)


def clip_gradients_(grads, clip):
    """Per-TENSOR L2 clip (not global): g *= clip/(||g||+1e-6) when that factor < 1 (utils.py:132-141).
    Returns the list of pre-clip norms."""
    norms = []
    for g in grads:
        if g is None:
            continue
        n = g.norm(2)
        norms.append(float(n))
        c = clip / (n + 1e-6)
        if c < 1:
            g.mul_(c)
    return norms


def adamw_step_(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8):
    """One torch.optim.AdamW update (decoupled decay; lafs_train.py:400 defaults)."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def ema_(teacher_p, student_p, m):
    """param_k <- m*param_k + (1-m)*param_q (lafs_train.py:610-613)."""
    teacher_p.mul_(m).add_((1 - m) * student_p)


# -------------------------------------------------------------- fine-tune (train_largescale.py)
def finetune_weight_decay(name, shape, weight_decay=0.1, stn_decay=0.05):
    """Decay assigned by param_groups_lrd (train_largescale.py:122-173): 0 for 1-D tensors,
    0.05 for the landmark CNN ('stn*'), weight_decay otherwise.  The lr_scale entries the
    reference also stores are never consumed by torch AdamW, so they are not restated."""
    if len(shape) == 1:
        return 0.0
    if name.startswith("stn"):
        return stn_decay
    return weight_decay


def warmup_cosine_lr(base_lr, it_epoch, warmup_epochs, total_epochs, eta_min=1e-6):
    """LR trajectory of GradualWarmupScheduler(multiplier=1) wrapping CosineAnnealingLR
    (train_largescale.py:728-733).  PARITY UNPINNED: `warmup_scheduler` is an un-vendored,
    un-versioned PyPI package; this restates its documented behaviour (linear 0 -> base over
    the warm-up epochs, then cosine to eta_min)."""
    if it_epoch < warmup_epochs:
        return base_lr * it_epoch / warmup_epochs
    t = it_epoch - warmup_epochs
    T = total_epochs - warmup_epochs
    return eta_min + 0.5 * (base_lr - eta_min) * (1 + math.cos(math.pi * t / T))
