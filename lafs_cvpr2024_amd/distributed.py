"""Data-parallel plumbing of the LAFS step: flat-arena gradient all-reduce (sum; the 1/world mean is folded into the fused
AdamW kernel's grad_scale) and the DINO center column-sum all-reduce (reference: DDP at lafs_train.py:375 and
dist.all_reduce at :675).  Backend-agnostic: "nccl" is RCCL over xGMI on MI355X, "gloo" is used by the CPU tests.

The arena orders parameters [trunk | head]; the head range (the 25.6 M-parameter last layer dominates) is reduced while the
trunk backward is still running, the trunk range right after it."""
import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class FlatReducer:
    """Launches asynchronous SUM all-reduces on slices of flat buffers and waits for all of them at once."""

    def __init__(self, group=None):
        self.group = group
        self.pending = []

    @property
    def active(self):
        return world_size() > 1

    def launch(self, tensor):
        if self.active:
            self.pending.append(dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait_all(self):
        for w in self.pending:
            w.wait()
        self.pending = []


def center_from_colsum(center, colsum, rows_per_rank, momentum):
    """center <- m*center + (1-m) * (sum over ranks of column sums) / (rows * world)   (reference lafs_train.py:674-679).
    `colsum` must already be all-reduced.  Pure torch: used by the CPU tests as the statement of what the HIP
    lafs_center_ema kernel computes."""
    return center * momentum + colsum / (rows_per_rank * world_size()) * (1 - momentum)
