"""CPU oracle for the LAFS hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain fp32 torch-CPU restatement of the arithmetic on the
LAFS data-parallel hot path.  It is the *checker* for the
HIP kernels in ``lafs_cvpr2024_amd/csrc``: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product package never imports it and fails loudly when the
HIP library is missing.

Parity status: PINNED.  Every function here is checked in
``tests/test_oracle_golden.py`` against fixtures under ``tests/golden/`` that
were produced by importing the reference itself (``tools/make_golden.py``,
run in the build container where ``/root/reference`` is mounted).  Pieces whose
specification does not exist in the reference (ArcFace, PartialFC, the
``warmup_scheduler`` package) say "parity unpinned" in their docstring.

All ``file:line`` citations are relative to the reference repository root.
"""
from . import vit, dino, optim, partfvit, margin, gather, step  # noqa: F401
