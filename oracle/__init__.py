"""CPU oracle of the path-solve path — test infrastructure only.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this package; the product
(``raytracing_amd``) never does.
"""
