"""lafs_cvpr2024_amd -- MI355X-native (gfx950) implementation of the LAFS data-parallel hot path.

Python is only the host: device memory, streams and torch.distributed (RCCL).  All arithmetic on the path runs in
hand-written HIP kernels from ``liblafs_hip.so`` (C ABI: ``include/lafs_hip.h``).  There is no CPU fallback.
"""
__version__ = "0.1.0"
