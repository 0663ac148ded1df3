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
    """Launches asynchronous SUM all-reduces on slices of flat buffers and waits for all of them at once.

    wire="bf16" (LAFS_GRAD_WIRE=bf16, opt-in): fp32 slices travel as bf16 -- cast into a staging buffer, reduced there, cast back
    on wait_all -- which halves the bytes on the xGMI links (211 -> 105 MB per LAFS step) at the price of a bf16 rounding of every
    rank's contribution and of the partial sums inside the collective; the default "f32" is what the reference's DDP sends."""

    def __init__(self, group=None, wire=None):
        import os
        self.group = group
        self.pending = []
        self.wire = wire or os.environ.get("LAFS_GRAD_WIRE", "f32")
        if self.wire not in ("f32", "bf16"):
            raise ValueError("LAFS_GRAD_WIRE must be f32 or bf16")

    @property
    def active(self):
        # LAFS_REDUCE_SINGLE_RANK=1 (tests): issue the collectives even in a one-rank process group, so that a single GPU runs the
        # whole multi-rank launch structure (graph segments, asynchronous RCCL all-reduces between them, stream-side waits)
        if world_size() > 1:
            return True
        import os
        return os.environ.get("LAFS_REDUCE_SINGLE_RANK") == "1" and dist.is_available() and dist.is_initialized()

    def launch(self, tensor):
        """Starts the all-reduce; returns a token for wait() (None when there is nothing to wait for)."""
        if not self.active:
            return None
        if self.wire == "bf16" and tensor.dtype == torch.float32 and tensor.is_cuda and tensor.numel() >= (1 << 16):
            from .ops import _p, call
            stage = torch.empty(tensor.numel(), device=tensor.device, dtype=torch.bfloat16)
            call("lafs_cast_bf16", _p(tensor), _p(stage), tensor.numel())
            tok = (dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group, async_op=True), stage, tensor)
        else:
            tok = (dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None)
        self.pending.append(tok)
        return tok

    def wait(self, tokens):
        """Makes the current stream wait for these launches (and casts bf16-wire slices back to fp32)."""
        for tok in tokens:
            if tok is None or tok not in self.pending:
                continue
            work, stage, tensor = tok
            work.wait()
            if stage is not None:
                from .ops import _p, call
                call("lafs_cast_f32", _p(stage), _p(tensor), tensor.numel())
            self.pending.remove(tok)

    def wait_all(self):
        self.wait(list(self.pending))


def center_from_colsum(center, colsum, rows_per_rank, momentum):
    """center <- m*center + (1-m) * (sum over ranks of column sums) / (rows * world)   (reference lafs_train.py:674-679).
    `colsum` must already be all-reduced.  Pure torch: used by the CPU tests as the statement of what the HIP
    lafs_center_ema kernel computes."""
    return center * momentum + colsum / (rows_per_rank * world_size()) * (1 - momentum)
