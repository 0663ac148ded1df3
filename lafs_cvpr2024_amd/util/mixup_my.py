"""Batch-mode Mixup with the reference's interface (util/mixup_my.py:84-211).  Only mode='batch' without cutmix is on
the fine-tune path (train_largescale.py:383-393 defaults); the other modes raise."""
import numpy as np
import torch


def one_hot(x, num_classes, on_value=1., off_value=0., device='cuda'):
    x = x.long().view(-1, 1)
    return torch.full((x.size()[0], num_classes), off_value, device=device).scatter_(1, x, on_value)


def mixup_target(target, num_classes, lam=1., smoothing=0.0, device='cuda'):
    """Dense [B, C] soft target lam*onehot(y) + (1-lam)*onehot(flip(y)) (reference :18-24).  The fused training engine
    never builds this matrix: it passes (y, flip(y), lam) to lafs_margin_softmax_ce."""
    off = smoothing / num_classes
    on = 1. - smoothing + off
    return one_hot(target, num_classes, on, off, device) * lam + one_hot(target.flip(0), num_classes, on, off, device) * (1. - lam)


class Mixup:
    def __init__(self, mixup_alpha=1., cutmix_alpha=0., cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode='batch',
                 correct_lam=True, label_smoothing=0.1, num_classes=1000):
        if cutmix_alpha > 0. or cutmix_minmax is not None or mode != 'batch':
            raise NotImplementedError("only batch-mode mixup (no cutmix) is on the reference's fine-tune path")
        self.mixup_alpha, self.mix_prob, self.label_smoothing, self.num_classes = mixup_alpha, prob, label_smoothing, num_classes
        self.mixup_enabled = True

    def draw_lambda(self):
        """_params_per_batch (reference :134-150): one uniform for the apply decision, then Beta(alpha, alpha)."""
        if self.mixup_enabled and np.random.rand() < self.mix_prob:
            return float(np.random.beta(self.mixup_alpha, self.mixup_alpha))
        return 1.

    def __call__(self, x, target, device='cuda'):
        assert len(x) % 2 == 0, 'Batch size should be even when using this'
        lam = self.draw_lambda()
        if lam != 1.:
            xf = x.flip(0).mul_(1. - lam)
            x.mul_(lam).add_(xf)
        return x, mixup_target(target, self.num_classes, lam, self.label_smoothing, device=x.device)
