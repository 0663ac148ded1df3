"""Guard bands around every device tensor the package allocates (debug aid, GPU box only).

`install()` wraps torch.empty / zeros / ones / full / empty_like / zeros_like so that each CUDA allocation sits between two
PAD-byte bands of a fixed byte pattern; `check()` reports the allocation sites (python stack) whose bands were written.  An
out-of-bounds write of a kernel then shows up every run, wherever the caching allocator happened to put the neighbours.

    python tools/lab/canary.py            # F5 LAFS step (graph + eager) under guard bands
"""
import math
import os
import sys
import traceback

import torch

PAD = 4096
PATTERN = 0xA5
_registry = []
_orig = {}


def _is_cuda(dev):
    if dev is None:
        return False
    return torch.device(dev).type == "cuda"


def _guarded(shape, dtype, device, fill):
    dtype = dtype or torch.get_default_dtype()
    n = int(math.prod(shape))
    es = torch.empty((), dtype=dtype).element_size()
    padn = PAD // es
    base = _orig["empty"](n + 2 * padn, dtype=dtype, device=device)
    raw = base.view(torch.uint8)
    raw[:PAD] = PATTERN
    raw[PAD + n * es:] = PATTERN
    mid = base[padn:padn + n]
    if fill is not None:
        mid.fill_(fill)
    _registry.append((base, n * es, "".join(traceback.format_stack(limit=7)[:-2])))
    return mid.view(tuple(shape))


def _shape_of(args, kwargs):
    if "size" in kwargs:
        return tuple(kwargs["size"])
    if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
        return tuple(args[0])
    return tuple(int(a) for a in args)


def _wrap_new(name, fill):
    orig = getattr(torch, name)
    _orig[name] = orig

    def f(*args, **kwargs):
        if _is_cuda(kwargs.get("device")) and "out" not in kwargs and not kwargs.get("pin_memory"):
            return _guarded(_shape_of(args, kwargs), kwargs.get("dtype"), kwargs["device"], fill)
        return orig(*args, **kwargs)
    setattr(torch, name, f)


def _wrap_like(name, fill):
    orig = getattr(torch, name)
    _orig[name] = orig

    def f(t, **kwargs):
        dev = kwargs.get("device", t.device)
        if _is_cuda(dev):
            return _guarded(t.shape, kwargs.get("dtype", t.dtype), dev, fill)
        return orig(t, **kwargs)
    setattr(torch, name, f)


def install():
    _wrap_new("empty", None)
    _wrap_new("zeros", 0)
    _wrap_new("ones", 1)
    _wrap_like("empty_like", None)
    _wrap_like("zeros_like", 0)
    orig_full = torch.full
    _orig["full"] = orig_full

    def full(size, fill_value, **kwargs):
        if _is_cuda(kwargs.get("device")):
            return _guarded(tuple(size), kwargs.get("dtype", torch.float32 if isinstance(fill_value, float) else torch.int64),
                            kwargs["device"], fill_value)
        return orig_full(size, fill_value, **kwargs)
    torch.full = full


def check(tag=""):
    torch.cuda.synchronize()
    bad = 0
    for base, nbytes, where in _registry:
        raw = base.view(torch.uint8)
        lo = (raw[:PAD] != PATTERN).nonzero().flatten()
        hi = (raw[PAD + nbytes:] != PATTERN).nonzero().flatten()
        if lo.numel() or hi.numel():
            bad += 1
            print(f"[canary {tag}] {nbytes}-byte {base.dtype} buffer: {lo.numel()} bytes written below "
                  f"(first at -{PAD - int(lo[0]) if lo.numel() else 0}), {hi.numel()} bytes written above "
                  f"(first at +{int(hi[0]) if hi.numel() else 0}, last at +{int(hi[-1]) if hi.numel() else 0}); allocated at\n{where}")
    print(f"[canary {tag}] {len(_registry)} buffers checked, {bad} overrun")
    return bad


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    install()
    from conftest import load_golden
    import test_gpu_step as T
    total = 0
    for graph in (False, True):
        fx = load_golden("f5_lafs_step")
        lrs, wds, moms = fx["hyper"].tolist()
        student, teacher, crit, eng = T._build(fx, graph)
        tt = crit.teacher_temp_schedule
        for s in range(2):
            crops = [fx[f"s{s}.crop{i}"] for i in range(5)]
            eng.step(crops, lr=lrs[s], wd=wds[s], momentum=moms[s], teacher_temp=float(tt[s]), epoch=s)
            total += check(f"f5 graph={graph} step {s}")
    sys.exit(1 if total else 0)
