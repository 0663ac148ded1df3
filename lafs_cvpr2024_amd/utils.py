"""Run-time utilities with the reference's names and semantics (reference ``utils.py``), minus the unused
image-retrieval / hub-download leftovers.  MultiCropWrapper packs all resolutions into ONE kernel pass when the
backbone is the HIP VisionTransformer."""
import argparse
import datetime
import os
import sys
import time
from collections import defaultdict, deque

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn


# --------------------------------------------------------------------------------------------- multi-crop wrapper
class MultiCropWrapper(nn.Module):
    """One backbone pass per run of equal-resolution crops, one head pass on the concatenated features
    (reference utils.py:594-659).  Output rows are crop-major in the order the crops were given."""

    def __init__(self, backbone, head):
        super().__init__()
        backbone.fc, backbone.head = nn.Identity(), nn.Identity()
        self.backbone = backbone
        self.head = head

    @staticmethod
    def group_ends(x):
        """Cumulative counts of consecutive crops with equal size: last dim for NCHW crops, token count for
        3-D patch tensors (reference utils.py:618-629)."""
        key = (lambda t: t.shape[-1]) if x[0].dim() >= 4 else (lambda t: t.shape[-2])
        ends, prev = [], None
        for i, inp in enumerate(x):
            k = key(inp)
            if prev is not None and k != prev:
                ends.append(i)
            prev = k
        ends.append(len(x))
        return ends

    def forward(self, x, return_land=False, idx_crops=None, image_noaug=None):
        if return_land:
            raise NotImplementedError("return_land drives an experiment variant that the LAFS entry point never enables")
        if not isinstance(x, list):
            x = [x]
        ends = [int(e) for e in idx_crops] if idx_crops is not None else self.group_ends(x)
        groups, start = [], 0
        for end in ends:
            groups.append(torch.cat(x[start:end]))
            start = end
        if hasattr(self.backbone, "forward_groups"):
            feats = self.backbone.forward_groups(groups)          # single packed kernel pass
        else:
            outs = []
            for g in groups:
                o = self.backbone(g)
                outs.append(o[0] if isinstance(o, tuple) else o)
            feats = torch.cat(outs)
        return self.head(feats)


# --------------------------------------------------------------------------------------------- schedules / groups
def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """Per-iteration linear warm-up followed by a half cosine (reference utils.py:187-198)."""
    warm_iters = warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warm_iters) if warmup_epochs > 0 else np.array([])
    n = epochs * niter_per_ep - warm_iters
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * np.arange(n) / n))
    sched = np.concatenate((warm, sched))
    assert len(sched) == epochs * niter_per_ep
    return sched


def get_params_groups(model):
    """[decayed, not decayed]: biases and 1-D tensors are not regularised (reference utils.py:662-673)."""
    reg, noreg = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (noreg if (name.endswith(".bias") or len(p.shape) == 1) else reg).append(p)
    return [{"params": reg}, {"params": noreg, "weight_decay": 0.}]


def clip_gradients(model, clip):
    """Per-tensor L2 clipping through torch ops (API parity with reference utils.py:132-141; the training engine
    uses the fused lafs_clip_adamw_ema kernel instead and never calls this)."""
    norms = []
    with torch.no_grad():
        for p in model.parameters():
            if p.grad is None:
                continue
            gnorm = float(p.grad.norm(2))
            norms.append(gnorm)
            if clip < gnorm + 1e-6:                         # per-tensor coefficient min(1, clip / (norm + 1e-6))
                p.grad.mul_(clip / (gnorm + 1e-6))
    return norms


def cancel_gradients_last_layer(epoch, model, freeze_last_layer):
    """During the first `freeze_last_layer` epochs the DINO head's last layer takes no gradient (reference utils.py:144-149); the
    fused engine does the same through LAFS_SEG_LAST_LAYER / hyper[LAFS_HP_FREEZE_LAST]."""
    if epoch < freeze_last_layer:
        for name, p in model.named_parameters():
            if "last_layer" in name:
                p.grad = None


def has_batchnorms(model):
    bn = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)
    return any(isinstance(m, bn) for m in model.modules())


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


_FLAG_WORDS = {"on": True, "true": True, "1": True, "off": False, "false": False, "0": False}


def bool_flag(s):
    """argparse type of the reference's on/off switches (utils.py:186-198)."""
    try:
        return _FLAG_WORDS[s.lower()]
    except KeyError:
        raise argparse.ArgumentTypeError("invalid value for a boolean flag") from None


def fix_random_seeds(seed=31):
    """Seeds torch (all devices) and numpy, as the reference's entry points do before building models (utils.py:144-149)."""
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


# --------------------------------------------------------------------------------------------- checkpoints
def restart_from_checkpoint(ckp_path, run_variables=None, **kwargs):
    """Resume hook with the reference's contract (utils.py:152-184): every keyword names a checkpoint entry and the object that
    takes it (`load_state_dict`, non-strict where the object accepts the flag); `run_variables` is updated in place with the
    scalar entries it lists (e.g. `epoch`).  A missing file is not an error: the run starts fresh."""
    if not os.path.isfile(ckp_path):
        return
    ckpt = torch.load(ckp_path, map_location="cpu", weights_only=False)
    print(f"Found checkpoint at {ckp_path}")
    for name, target in kwargs.items():
        if target is None or name not in ckpt:
            print(f"=> key '{name}' not found in checkpoint: '{ckp_path}'")
            continue
        state = ckpt[name]
        try:
            report = target.load_state_dict(state, strict=False)
        except TypeError:                                   # optimizers / loss objects without a `strict` argument
            report = target.load_state_dict(state)
        print(f"=> loaded '{name}' from checkpoint '{ckp_path}' with msg {report}")
    for name in (run_variables or {}):
        if name in ckpt:
            run_variables[name] = ckpt[name]


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


# --------------------------------------------------------------------------------------------- distributed (RCCL)
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def setup_for_distributed(is_master):
    """Only rank 0 prints unless force=True is passed."""
    import builtins
    builtin_print = builtins.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)
    builtins.print = print


def init_distributed_mode(args):
    """One process per GPU; backend "nccl" is RCCL on ROCm (reference utils.py:467-499).  Launched by torchrun /
    torch.distributed.launch (RANK, WORLD_SIZE, LOCAL_RANK) or stand-alone on one GPU."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    elif torch.cuda.is_available():
        args.rank, args.gpu, args.world_size = 0, 0, 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    else:
        print("Does not support training without GPU.")
        sys.exit(1)
    torch.cuda.set_device(args.gpu)
    dist.init_process_group(backend="nccl", init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank,
                            device_id=torch.device("cuda", args.gpu))
    print("| distributed init (rank {}): {}".format(args.rank, getattr(args, "dist_url", "env://")), flush=True)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


# --------------------------------------------------------------------------------------------- logging
class SmoothedValue:
    """Window median / global average tracker (reference utils.py:224-300)."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{median:.6f} ({global_avg:.6f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device="cuda")
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    median = property(lambda self: float(np.median(list(self.deque))))
    avg = property(lambda self: float(np.mean(list(self.deque))))
    global_avg = property(lambda self: self.total / max(self.count, 1))
    max = property(lambda self: max(self.deque))
    value = property(lambda self: self.deque[-1])

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(float(v))

    def __str__(self):
        return self.delimiter.join("{}: {}".format(n, str(m)) for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=""):
        start = end = time.time()
        iter_time, data_time = SmoothedValue(fmt="{avg:.6f}"), SmoothedValue(fmt="{avg:.6f}")
        n = len(iterable)
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                mem = torch.cuda.max_memory_allocated() / 2 ** 20 if torch.cuda.is_available() else 0
                print(f"{header} [{i}/{n}] eta: {eta} {self} time: {iter_time} data: {data_time} max mem: {mem:.0f}")
            end = time.time()
        total = time.time() - start
        print("{} Total time: {} ({:.6f} s / it)".format(header, str(datetime.timedelta(seconds=int(total))), total / max(n, 1)))


# --------------------------------------------------------------------------------------------- host -> device staging
class PinnedRing:
    """Pinned staging buffers for small per-step host -> device uploads (hyper-parameters, augmentation records).

    An asynchronous copy from ONE reused pinned buffer races with the host: the training loop runs many steps ahead of the GPU,
    so the host would overwrite the buffer before an earlier step's copy has executed (that step would then see a later step's
    learning rate / freeze flag / crop boxes).  Each upload takes the next of `depth` buffers and blocks only if the copy issued
    `depth` uploads ago has still not run -- which also bounds how far the host can run ahead."""

    def __init__(self, shape, dtype, depth=8):
        self.bufs = [torch.zeros(shape, dtype=dtype).pin_memory() for _ in range(depth)]
        self.events = [None] * depth
        self.i = 0

    def upload(self, dst, fill):
        """fill(host_buffer) writes the values; they are then copied to the device tensor `dst` on the current stream."""
        k = self.i
        self.i = (k + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        fill(self.bufs[k])
        dst.copy_(self.bufs[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return self.bufs[k]
