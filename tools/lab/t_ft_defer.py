"""C4 fine-tune step with the trunk's weight gradients deferred beside the landmark CNN's backward: workgroup cap sweep against the
immediate form.   gpurun -- python tools/lab/t_ft_defer.py"""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
bench.EXTRA_ROOFLINE = False
for defer, cap, head in (("0", 0, "0"), ("1", 96, "1"), ("1", 128, "0"), ("1", 128, "1"), ("1", 160, "1"), ("1", 192, "1"), ("1", 224, "1"), ("1", 256, "1"), ("0", 0, "0")):
    os.environ["LAFS_FT_WGRAD_DEFER"] = defer
    os.environ["LAFS_FT_DEFER_WG"] = str(cap)
    os.environ["LAFS_FT_DEFER_HEAD"] = head
    r = bench.extra_finetune(dev, steps=16, warmup=4)
    print(f"defer={defer} cap={cap:4d} class-table gradients deferred={head}: {r['ms_per_step']:.2f} ms/step  loss {r['loss']}", flush=True)
    gc.collect(); torch.cuda.empty_cache()
