"""Timing probe of the channel statistics kernel on a coverage-map-sized table (not a test)."""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from raytracing_amd import _lib  # noqa: E402

rng = np.random.default_rng(1)
n_rx = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
mean_paths = int(sys.argv[2]) if len(sys.argv) > 2 else 17
counts = rng.poisson(mean_paths, size=n_rx)
off = np.zeros(n_rx + 1, dtype=np.int64)
np.cumsum(counts, out=off[1:])
n = int(off[-1])
power = 10.0 ** rng.uniform(-12, -7, size=n)
el, az = rng.uniform(0, 180, n), rng.uniform(-180, 180, n)
xyz = np.column_stack((np.sin(el) * np.cos(az), np.sin(el) * np.sin(az), np.cos(el)))
t = rng.uniform(1e-8, 2e-6, n)
los = (rng.random(n) < 0.05).astype(np.uint8)
lib = _lib.require_gpu()
dev = torch.device("cuda", 0)
d = {k: torch.tensor(v, device=dev) for k, v in dict(off=off, power=power, time=t, el=el, az=az, xyz=xyz, los=los).items()}
out = torch.empty((n_rx, 8), dtype=torch.float64, device=dev)


def run():
    rc = lib.rt_channel_stats(0, n_rx, d["off"].data_ptr(), d["power"].data_ptr(), d["time"].data_ptr(), d["el"].data_ptr(),
                              d["az"].data_ptr(), d["xyz"].data_ptr(), d["los"].data_ptr(), 3, 0.9, out.data_ptr(), 1)
    assert rc == 0


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
host = _lib.channel_stats(off, power, t, el, az, xyz, los, 3, 0.9)
assert np.allclose(host, out.cpu().numpy(), equal_nan=True)
print(f"{n_rx} receivers, {n} paths: {dt * 1e3:.3f} ms per call device-resident "
      f"({n / dt:.3e} paths/s, {(n * 57 + n_rx * 72) / dt / 1e9:.1f} GB/s of table read once)")
