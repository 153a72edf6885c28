"""genomicsbench_amd — MI355X-native (gfx950) kernels for GenomicsBench's DP hot path.

The product is ``libgbx.so`` (hand-written HIP behind the C-ABI of ``include/gbx.h``)
plus the C++ drivers under ``csrc/drivers``.  This Python package is the thin host
mirror used by ``tests/`` and ``bench.py``: ctypes bindings, the reference file
formats, synthetic dataset generators and the multi-GPU sharding helper.  Nothing
here computes a DP on the CPU; if ``libgbx.so`` is missing, imports of the compute
wrappers raise.
"""
from .version import __version__  # noqa: F401
