"""Oracle: DINO student/teacher cross-entropy with centering (fp32 CPU).  Test infrastructure only."""
import numpy as np
import torch
import torch.nn.functional as F


def teacher_temp_schedule(warmup_teacher_temp, teacher_temp, warmup_epochs, nepochs):
    """linspace(warmup, final, warmup_epochs) followed by a constant (lafs_train.py:636-641)."""
    return np.concatenate((np.linspace(warmup_teacher_temp, teacher_temp, warmup_epochs),
                           np.ones(nepochs - warmup_epochs) * teacher_temp))


def dino_loss(student_output, teacher_output, center, ncrops, teacher_temp, student_temp=0.1):
    """Definition form of DINOLoss.forward (lafs_train.py:643-667).

    student_output [ncrops*B, K], teacher_output [2*B, K], center [1, K].
    Mean over the 2*ncrops-2 (teacher view, student view) pairs with different
    view index of the per-sample cross-entropy  -sum_k q_k log softmax(s/tau_s)_k,
    q = softmax((t - c)/tau_t) with no gradient.
    """
    s = (student_output / student_temp).chunk(ncrops)
    q = F.softmax((teacher_output - center) / teacher_temp, dim=-1).detach().chunk(2)
    total, n = 0.0, 0
    for iq in range(2):
        for v in range(ncrops):
            if v == iq:
                continue
            total = total + torch.sum(-q[iq] * F.log_softmax(s[v], dim=-1), dim=-1).mean()
            n += 1
    return total / n


def dino_loss_closed_form(student_output, teacher_output, center, ncrops, teacher_temp, student_temp=0.1):
    """Closed form used by the fused HIP kernel:

        loss = 1/n_terms * sum_{iq, v != iq} mean_b [ lse(s_v) - <q_iq, s_v> ]
        dL/dstudent_v = 1/(n_terms * B * tau_s) * sum_{iq != v} (softmax(s_v) - q_iq)

    Returns (loss, grad wrt student_output).
    """
    B = teacher_output.shape[0] // 2
    s = (student_output / student_temp).view(ncrops, B, -1)
    q = F.softmax((teacher_output - center) / teacher_temp, dim=-1).view(2, B, -1)
    lse = torch.logsumexp(s, dim=-1)                                   # [ncrops, B]
    n_terms = 2 * ncrops - 2
    loss = student_output.new_zeros(())
    grad = torch.zeros_like(s)
    p = torch.softmax(s, dim=-1)
    for iq in range(2):
        for v in range(ncrops):
            if v == iq:
                continue
            loss = loss + (lse[v] - (q[iq] * s[v]).sum(-1)).mean()
            grad[v] += (p[v] - q[iq])
    grad = grad / (n_terms * B * student_temp)
    return loss / n_terms, grad.view_as(student_output)


def update_center(center, teacher_output, momentum=0.9, world_size=1, all_reduce=None):
    """center <- m*center + (1-m) * mean_rows(teacher_output) over all ranks (lafs_train.py:669-679).

    Uses the raw (un-centred) teacher logits; the sum is all-reduced, then divided by
    rows * world_size.
    """
    bc = teacher_output.sum(dim=0, keepdim=True)
    if all_reduce is not None:
        all_reduce(bc)
    bc = bc / (teacher_output.shape[0] * world_size)
    return center * momentum + bc * (1 - momentum)
