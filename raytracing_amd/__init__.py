"""MI355X-native path-solve hot path of the MaximeLeu/RayTracing urban radio ray tracer.

Host API mirrors the reference's Python surface (``geometry``, ``ray_tracing.RayTracingProblem``,
``electromagnetism.my_field_computation``, ``multithread_solve.multithread_solve_place``); the arithmetic runs in
hand-written HIP kernels for gfx950 behind a C-ABI shared library (``include/rt_pathsolve.h``).
"""
__version__ = "0.1.0"
