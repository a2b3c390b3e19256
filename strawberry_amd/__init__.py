"""strawberry_amd -- MI355X (gfx950) implementation of Strawberry's per-locus EM hot path.

Only the path BASELINE.json's north_star names lives here:
  csrc/      hand-written HIP kernels + the extern "C" boundary (include/sbgpu.h)
  _lib.py    ctypes binding of libsbgpu.so
  em.py      host-side mirror of the reference's EmSolver / abundance call surface
  synth.py   deterministic synthetic locus batches (configs C2 / C2-U / C3)
  dist.py    one-process-per-GPU sharding + the single TPM all-reduce (RCCL)
"""
__version__ = "0.1.0"
