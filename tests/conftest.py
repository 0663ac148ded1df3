import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """LAFS_TEST_SHUFFLE=<seed>: run the collected tests in a seeded random order (no plugin needed) -- the suite must not depend on
    the alphabetical order the driver happens to use (round 4's order-dependent hipGraphLaunch crash: DESIGN.md section 7)."""
    seed = os.environ.get("LAFS_TEST_SHUFFLE")
    if seed:
        import random
        random.Random(int(seed)).shuffle(items)


@pytest.fixture(autouse=True)
def _release_gpu_memory_between_tests(request):
    """GPU tests build whole engines (hipGraph pools of several GB each): drop what a test leaves behind before the next one starts, so
    that the suite's footprint is one test's, whatever order the tests run in.  LAFS_TEST_MEM=1 prints the allocator's state per test."""
    yield
    if torch.cuda.is_available() and request.node.get_closest_marker("gpu") is not None:
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if os.environ.get("LAFS_TEST_MEM") == "1":
            free, total = torch.cuda.mem_get_info()
            print(f"\n[mem] after {request.node.name}: allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB, reserved "
                  f"{torch.cuda.memory_reserved() / 2**30:.1f} GiB, device free {free / 2**30:.1f} of {total / 2**30:.1f} GiB", flush=True)


def load_golden(name):
    """Return {key: torch tensor} for tests/golden/<name>.npz (numeric arrays only -> tensors)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" else a
    return out


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()


def det_fill(module):
    """Fill every tensor of a module's state_dict with a closed-form pattern that depends only on its key and shape
    (used by tools/make_golden.py on the reference model and by the tests on this build's model: same keys -> same weights)."""
    import zlib
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            n = v.numel()
            ph = (zlib.crc32(k.encode()) % 1000) / 1000.0
            t = torch.sin(torch.arange(n, dtype=torch.float64) * 0.618 + ph * 6.283)
            fan = max(v[0].numel() if v.dim() > 1 else 1, 1)
            if k.endswith("running_var"):
                val = 1.0 + 0.2 * t
            elif k.endswith("running_mean"):
                val = 0.05 * t
            elif v.dim() == 1 and k.endswith("weight"):
                val = 1.0 + 0.1 * t
            elif v.dim() == 1:
                val = 0.05 * t
            else:
                val = t * (1.5 / fan ** 0.5)
            v.copy_(val.view(v.shape).to(v.dtype))



def det_fill_random(module):
    """Like det_fill, but pseudo-random: every floating tensor of the state_dict is drawn from a CPU generator seeded with the
    crc32 of its key -- identical on the reference model (tools/make_golden.py) and on this build's model without storing 2.8 M
    weights.  Convolution / linear weights ~ N(0, 2 / fan_out) (the trunk's own kaiming fan_out initialisation), BatchNorm weights
    1 + 0.1 n, biases 0.1 n, running statistics untouched.  Unlike det_fill's sinusoids (whose convolutions cancel almost exactly:
    a BatchNorm in TRAINING mode then renormalises rounding noise) these weights give the well-conditioned network bf16 kernels can
    be held against."""
    import zlib
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if not v.dtype.is_floating_point or k.endswith(("running_mean", "running_var")):
                continue
            g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
            t = torch.randn(v.shape, generator=g, dtype=torch.float64)
            if v.dim() >= 2:
                fan_out = v.shape[0] * (v[0][0].numel() if v.dim() > 2 else 1)
                val = t * (2.0 / fan_out) ** 0.5
            elif k.endswith("weight"):
                val = 1.0 + 0.1 * t
            else:
                val = 0.1 * t
            v.copy_(val.to(v.dtype))


def gate_errors(name, errs, gate):
    """Assert every per-tensor relative-L2 error in `errs` (dict key -> error) is below `gate`, and always PRINT the worst
    observed value (pytest -s / the captured output of a failure): the gates are 2x the worst value seen on MI355X and the
    observed table lives in DESIGN.md section 2, so a drift is visible before it becomes a failure."""
    if not errs:
        return
    k = max(errs, key=errs.get)
    print(f"[grad-gate] {name}: worst rel-L2 {errs[k]:.3e} at {k} (gate {gate:.1e}, {len(errs)} tensors)")
    bad = {k: v for k, v in errs.items() if not v <= gate}
    assert not bad, (name, bad)
