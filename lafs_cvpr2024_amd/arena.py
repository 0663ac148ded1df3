"""Flat device arenas for parameters / gradients / optimizer state.

MI355X-first memory layout: every tensor of a model lives in ONE contiguous fp32 buffer (its nn.Parameter is a
view into it), starts on a 1024-element chunk boundary, and has a bf16 shadow at the same offset (MFMA operand) plus,
for weight matrices, a transposed bf16 shadow (dgrad operand).  Per-tensor work (clip norms, AdamW, EMA, casts,
all-reduce) then runs as single launches / single collectives over the flat buffers instead of per-parameter loops
(reference: utils.py:132-141, lafs_train.py:610-613).
"""
import torch

from . import _lib
from .ops import _p, call

CHUNK = _lib.CHUNK


def _round_up(n, m):
    return (n + m - 1) // m * m


class ParamArena:
    """Owns the flat buffers of one nn.Module.  `with_grad` adds grad + AdamW moments."""

    def __init__(self, module: torch.nn.Module, device, with_grad=True, transposed=None):
        self.module = module
        self.device = torch.device(device)
        self.names, self.params, self.offsets, self.numels = [], [], {}, {}
        off = 0
        for name, p in module.named_parameters():
            self.names.append(name)
            self.params.append(p)
            self.offsets[name] = off
            self.numels[name] = p.numel()
            off += _round_up(p.numel(), CHUNK)
        self.size = off
        self.n_chunks = off // CHUNK
        self.n_seg = len(self.names)
        dev = self.device
        self.master = torch.zeros(off, device=dev, dtype=torch.float32)
        self.shadow = torch.zeros(off, device=dev, dtype=torch.bfloat16)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32) if with_grad else None
        self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32) if with_grad else None
        self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32) if with_grad else None
        chunk_seg, flags = [], []
        for i, (name, p) in enumerate(zip(self.names, self.params)):
            chunk_seg += [i] * (_round_up(p.numel(), CHUNK) // CHUNK)
            f = 0
            if p.requires_grad:
                f |= _lib.SEG_TRAINABLE
            # utils.get_params_groups (utils.py:662-673): no decay for biases and 1-D tensors
            if not (name.endswith(".bias") or p.dim() == 1):
                f |= _lib.SEG_DECAY
            if "last_layer" in name:                          # utils.cancel_gradients_last_layer (utils.py:144-149)
                f |= _lib.SEG_LAST_LAYER
            flags.append(f)
        self.chunk_seg = torch.tensor(chunk_seg, dtype=torch.int32, device=dev)
        self.seg_flags = torch.tensor(flags, dtype=torch.int32, device=dev)
        self.seg_step = torch.zeros(self.n_seg, dtype=torch.int32, device=dev)
        self.seg_sumsq = torch.zeros(self.n_seg, dtype=torch.float32, device=dev)
        self.chunk_sumsq = torch.zeros(self.n_chunks, dtype=torch.float32, device=dev) if with_grad else None
        # transposed bf16 shadows for the 2-D weights named in `transposed` (name -> True)
        self.t_offsets = {}
        toff = 0
        for name, p in zip(self.names, self.params):
            if transposed is not None and transposed(name, p):
                self.t_offsets[name] = toff
                toff += _round_up(p.numel(), 64)
        self.shadow_t = torch.zeros(max(toff, 64), device=dev, dtype=torch.bfloat16)
        # move the parameters into the arena (their storage becomes a view of `master`)
        with torch.no_grad():
            for name, p in zip(self.names, self.params):
                v = self.view(self.master, name, p.shape)
                v.copy_(p.data.to(dev, torch.float32))
                p.data = v
                if with_grad and p.requires_grad:
                    p.grad = self.view(self.grad, name, p.shape)
        self._versions = None
        self.refresh_shadows()

    def set_decay_groups(self, rule):
        """Re-assign the weight-decay class of every tensor: rule(name, param) -> 'decay' | 'low' | 'none'.  The default (set in
        __init__) is utils.get_params_groups of the LAFS step; the fine-tune step uses param_groups_lrd (finetune_decay_group)."""
        flags = self.seg_flags.cpu().tolist()
        for i, (name, p) in enumerate(zip(self.names, self.params)):
            f = flags[i] & ~(_lib.SEG_DECAY | _lib.SEG_LOW_DECAY)
            g = rule(name, p)
            if g == "decay":
                f |= _lib.SEG_DECAY
            elif g == "low":
                f |= _lib.SEG_LOW_DECAY
            elif g != "none":
                raise ValueError(f"unknown decay group {g!r} for {name}")
            flags[i] = f
        self.seg_flags.copy_(torch.tensor(flags, dtype=torch.int32))

    # ------------------------------------------------------------------ views / pointers
    def view(self, buf, name, shape=None):
        o, n = self.offsets[name], self.numels[name]
        v = buf[o:o + n]
        return v.view(shape) if shape is not None else v

    def tview(self, name):
        p = self.params[self.names.index(name)]
        o = self.t_offsets[name]
        return self.shadow_t[o:o + p.numel()].view(p.shape[1], p.shape[0])

    def bf(self, name):
        p = self.params[self.names.index(name)]
        return self.view(self.shadow, name, p.shape)

    # ------------------------------------------------------------------ shadow maintenance
    def _version_key(self):
        return tuple(p._version for p in self.params)

    def refresh_shadows(self):
        """bf16 copy of everything + transposed copies of the registered weights."""
        call("lafs_cast_bf16", _p(self.master), _p(self.shadow), self.size)
        self.refresh_transposed()
        self._versions = self._version_key()

    def refresh_transposed(self):
        """All W^T shadows in one launch (table-driven 64x64 tile transpose)."""
        if not self.t_offsets:
            return
        if getattr(self, "_t_table", None) is None:
            rows_, starts, n = [], [0], 0
            for name, toff in self.t_offsets.items():
                p = self.params[self.names.index(name)]
                r, c = p.shape[0], p.numel() // p.shape[0]
                rows_ += [self.offsets[name], r, c, toff]
                n += ((r + 63) // 64) * ((c + 63) // 64)            # 64x64 tiles (lafs_transpose_cast_table)
                starts.append(n)
            self._t_table = torch.tensor(rows_, dtype=torch.int64, device=self.device)
            self._t_starts = torch.tensor(starts, dtype=torch.int32, device=self.device)
            self._t_n = (len(self.t_offsets), n)
        call("lafs_transpose_cast_table", _p(self.master), _p(self.shadow_t), _p(self._t_table), _p(self._t_starts),
             self._t_n[0], self._t_n[1])

    def ensure_fresh(self):
        """Re-derive the shadows if someone modified a parameter through torch (load_state_dict, .copy_ ...)."""
        if self._versions != self._version_key():
            self.refresh_shadows()

    def scratch(self, key, n_floats):
        """A float32 device buffer of at least n_floats, allocated once per key and kept (workspaces of kernels that fold their own
        partial sums: static across the steps of a captured graph)."""
        cache = self.__dict__.setdefault("_scratch", {})
        buf = cache.get(key)
        if buf is None or buf.numel() < n_floats:
            buf = cache[key] = torch.empty(max(int(n_floats), 4), device=self.device, dtype=torch.float32)
        return buf

    def zero_grad(self):
        self.grad.zero_()
