"""metaeuk_amd -- MI355X-native prefilter+align hot path of `metaeuk predictexons`.

The product is metaeuk_amd/lib/libmetaeuk_amd.so (HIP kernels behind the C ABI of
include/metaeuk_amd.h) and the `metaeuk-amd` command-line front end; `api` is a thin ctypes
binding used by the tests and bench.py, `synth` the deterministic workload generator.
"""
__all__ = ["api", "build", "synth"]
