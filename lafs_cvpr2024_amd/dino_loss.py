"""DINOLoss with the reference's constructor / forward / buffer layout (lafs_train.py:626-679) on the fused HIP kernel."""
import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .ops import _p, call


class _DinoLossFunction(torch.autograd.Function):
    """Loss and dL/dstudent come out of the same two streaming passes; backward only rescales."""

    @staticmethod
    def forward(ctx, student, teacher, center, ncrops, student_temp, teacher_temp):
        K = student.shape[1]
        s = student if (student.stride(-1) == 1 and student.stride(0) % 4 == 0) else student.contiguous()
        t = teacher if (teacher.stride(-1) == 1 and teacher.stride(0) == s.stride(0)) else None
        if t is None:                                   # the kernel wants one row stride for both inputs
            s, t = student.contiguous(), teacher.contiguous()
        loss, grad = ops.dino_loss_fwd_bwd(s.float(), t.float(), center.reshape(-1), ncrops, student_temp, teacher_temp, K=K,
                                           grad_bf16=False)
        ctx.save_for_backward(grad)
        ctx.K = K
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return grad[:, :ctx.K] * gout, None, None, None, None, None


class DINOLoss(nn.Module):
    def __init__(self, out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs,
                 student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp = student_temp
        self.center_momentum = center_momentum
        self.ncrops = ncrops
        self.register_buffer("center", torch.zeros(1, out_dim))
        # warm-up of the teacher temperature, then constant (reference lafs_train.py:636-641)
        self.teacher_temp_schedule = np.concatenate((
            np.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs),
            np.ones(nepochs - warmup_teacher_temp_epochs) * teacher_temp))

    def forward(self, student_output, teacher_output, epoch):
        temp = float(self.teacher_temp_schedule[epoch])
        loss = _DinoLossFunction.apply(student_output, teacher_output.detach(), self.center, self.ncrops,
                                       float(self.student_temp), temp)
        self.update_center(teacher_output)
        return loss

    @torch.no_grad()
    def update_center(self, teacher_output):
        """center <- m*center + (1-m) * mean over all ranks of the RAW teacher logits (reference :669-679)."""
        t = teacher_output.detach()
        if t.stride(-1) != 1 or t.stride(0) % 4:
            t = t.contiguous()
        rows, K = t.shape
        colsum = torch.empty(K, device=t.device, dtype=torch.float32)
        call("lafs_colsum_f32", _p(t), t.stride(0), rows, K, _p(colsum))
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(colsum)
            world = dist.get_world_size()
        call("lafs_center_ema", _p(self.center), _p(colsum), K, 1.0 / (rows * world), float(self.center_momentum))
