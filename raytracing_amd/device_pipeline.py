"""The per-receiver part of the path-solve path kept on the GPU from receivers to per-receiver results.

``DevicePipeline.run`` chains, for one emitter and a block of receivers resident in HBM, the three device-pointer entry
points of the library — ``rt_solve_paths`` (lines of sight, every (receiver, candidate) solve with its checks;
ray_tracing.py:232-237, 281-390 of the reference), ``rt_describe_paths`` and ``rt_fields`` (my_field_computation,
electromagnetism.py:648-802) — and leaves everything in device tensors.  ``receiver_records`` condenses the result into
one fixed-size record per receiver (power in dB per frequency, path counts per path type, status), which is what the
ranks of a multi-GPU run exchange (``all_gather_records``): the device-side replacement of the reference's
``merge_solved_problems`` (multithread_solve.py:45-60) for a coverage map.

torch is used for device memory and the collective only; every number comes from the library's kernels.
"""
import numpy as np

from . import _lib

# path-type columns of a receiver record: LOS, R, RR, RRR, RRRR, D, RD, RRD, RRRD
N_TYPES = 1 + _lib.RT_MAX_ORDER + _lib.RT_MAX_ORDER
TYPE_NAMES = ["LOS"] + ["R" * k for k in range(1, _lib.RT_MAX_ORDER + 1)] + \
             ["R" * k + "D" for k in range(0, _lib.RT_MAX_ORDER)]
# int32 columns of a record after the dB values: N_TYPES counts, then paths in all, paths with a non-zero field
# status, largest status code
N_INT_COLUMNS = N_TYPES + 3


def record_bytes(n_freq):
    """Size of one receiver record: n_freq float64 dB values, then N_INT_COLUMNS int32, rounded up to a multiple of 8
    (56 bytes at one frequency)."""
    return 8 * n_freq + 4 * N_INT_COLUMNS + (4 * N_INT_COLUMNS) % 8


class PathBlock:
    """Result of one ``DevicePipeline.run``: views of the pipeline's device tensors (valid until its next run)."""

    def __init__(self, buffers=None, **kw):
        self.__dict__.update(kw)
        self._buffers = buffers or {}

    def __getattr__(self, name):
        # the per-path arrays (rx, group, cand, points, kind, chain, ...): the first n_paths rows of the pipeline's
        # buffers, cut when first asked for (a dozen slices per step were a tenth of the host's share of a small step)
        buffers = self.__dict__.get("_buffers")
        if buffers is not None and name in buffers:
            view = buffers[name][:self.n_paths]
            self.__dict__[name] = view
            return view
        raise AttributeError(name)

    def to_host(self):
        """The per-path arrays as NumPy arrays, sorted by (receiver, group, candidate): the reference's row order
        (line of sight, reflections by order in visiting order, diffractions by order)."""
        import torch

        n = self.n_paths
        # group and candidate index out of the library's two words (a candidate index of 2^31 or more is split over them)
        word_g, word_c = self.group.to(torch.int64), self.cand.to(torch.int64)
        group = torch.where(word_g < 0, word_g, word_g & 255)
        index = torch.where(word_g < 0, word_c, ((word_g >> 8) << 31) | word_c)
        if n and int(index.max()) >= (1 << 31):
            order = torch.argsort(index, stable=True)
            order = order[torch.argsort(group[order], stable=True)]
            order = order[torch.argsort(self.rx[order], stable=True)]
        else:
            order = torch.argsort(self.rx.to(torch.int64) * (1 << 40) + (group + 1) * (1 << 32) + index)
            index = index.to(torch.int32)
        decoded = {"group": group.to(torch.int32), "cand": index}
        out = {}
        for name in ("rx", "group", "cand", "points", "kind", "chain", "edge_polys", "edge_points", "field", "power",
                     "angles", "time", "status"):
            t = decoded.get(name, getattr(self, name))
            out[name] = t[order].cpu().numpy() if n else t[:0].cpu().numpy()
        out["power_db"] = self.power_db.cpu().numpy()
        return out


class DevicePipeline:
    """Solve + description + fields of one emitter on one GPU, buffers kept between runs."""

    def __init__(self, scene, candidates, tx, capacity=1 << 16, field_scene=None, n_late_polygons=0):
        """``field_scene`` / ``n_late_polygons``: the scene the field step looks at when polygons were added after the
        paths were found (the tree crowns of my_field_computation, electromagnetism.py:745): ``scene`` plus those
        polygons at the end of its list."""
        import torch

        self.torch = torch
        self.scene = scene
        self.field_scene = field_scene if field_scene is not None else scene
        self.n_late_polygons = int(n_late_polygons)
        self.cand = candidates
        self.tx = np.asarray(tx, dtype=np.float64).reshape(3).copy()
        self.device = torch.device("cuda", scene.device)
        self.capacity = 0
        self.n_freq = 0
        self._grow(int(capacity), 1)

    def _grow(self, capacity, n_freq):
        torch, dev, K = self.torch, self.device, _lib.RT_MAX_ORDER
        self.capacity, self.n_freq = capacity, n_freq
        i32, f64 = torch.int32, torch.float64
        self.b = dict(
            rx=torch.empty(capacity, dtype=i32, device=dev), group=torch.empty(capacity, dtype=i32, device=dev),
            cand=torch.empty(capacity, dtype=i32, device=dev), points=torch.empty((capacity, K, 3), dtype=f64, device=dev),
            kind=torch.empty(capacity, dtype=i32, device=dev), chain=torch.empty((capacity, K), dtype=i32, device=dev),
            edge_polys=torch.empty((capacity, 2), dtype=i32, device=dev),
            edge_points=torch.empty((capacity, 6), dtype=f64, device=dev),
            field=torch.empty((capacity, n_freq, 6), dtype=f64, device=dev),
            power=torch.empty((capacity, n_freq), dtype=f64, device=dev),
            angles=torch.empty((capacity, 4), dtype=f64, device=dev), time=torch.empty(capacity, dtype=f64, device=dev),
            status=torch.empty(capacity, dtype=i32, device=dev))

    def run(self, d_rx, frequencies, fields=True, records=False, **antenna):
        """d_rx: float64 CUDA tensor [n, 3] on this pipeline's GPU.  Returns a PathBlock.  ``records``: also make the
        per-receiver records (``receiver_records(block)`` then returns them).  Description, field step and records
        are queued one behind the other on the library's stream and waited for once (three waits, with the GPU idle
        while the host came back for the next call, were a tenth of a small block's step)."""
        torch = self.torch
        assert d_rx.is_cuda and d_rx.dtype == torch.float64 and d_rx.is_contiguous() and d_rx.shape[1] == 3
        freqs = np.atleast_1d(np.asarray(frequencies, dtype=np.float64))
        n_rx = int(d_rx.shape[0])
        if len(freqs) != self.n_freq:
            self._grow(self.capacity, len(freqs))
        # the library works on a stream of its own: it waits, on the device, for what torch has queued for d_rx
        handle = torch.cuda.current_stream(self.device).cuda_stream
        self.scene.wait_for_stream(handle)
        if self.field_scene is not self.scene:
            self.field_scene.wait_for_stream(handle)
        # (everything the host can do ahead of the solve's result is done ahead of it: between the solve's last kernel
        # and the description's first the GPU has nothing queued)
        power_db = torch.empty((n_rx, len(freqs)), dtype=torch.float64, device=self.device)
        rec = None
        if records and fields and n_rx:
            rec = torch.empty((n_rx, record_bytes(len(freqs))), dtype=torch.uint8, device=self.device)
        while True:
            b = self.b
            found, attempted = self.scene.solve_paths_device(self.cand, self.tx, d_rx.data_ptr(), n_rx, b["rx"].data_ptr(),
                                                             b["group"].data_ptr(), b["cand"].data_ptr(),
                                                             b["points"].data_ptr(), self.capacity)
            if found <= self.capacity:
                break
            self._grow(found + found // 8 + 1024, len(freqs))
        n = int(found)
        fields_timing = None
        if fields and n_rx:
            # one stream for all three steps: queued without waiting in between (two scenes have two streams, and the
            # description then has to be waited for before the other scene's field step)
            one_stream = self.field_scene is self.scene
            if one_stream:
                self.scene.set_deferred_sync(True)
            try:
                self.cand.describe_device(b["group"].data_ptr(), b["cand"].data_ptr(), n, b["kind"].data_ptr(),
                                          b["chain"].data_ptr(), b["edge_polys"].data_ptr(), b["edge_points"].data_ptr())
                self.field_scene.fields_device(self.tx, d_rx.data_ptr(), n_rx, b["rx"].data_ptr(), b["kind"].data_ptr(),
                                         b["chain"].data_ptr(), b["edge_polys"].data_ptr(), b["edge_points"].data_ptr(),
                                         b["points"].data_ptr(), n, freqs, b["field"].data_ptr(), b["power"].data_ptr(),
                                         b["angles"].data_ptr(), b["time"].data_ptr(), b["status"].data_ptr(),
                                         power_db.data_ptr(), n_late_polygons=self.n_late_polygons, **antenna)
                if records:
                    self.field_scene.receiver_records_device(n_rx, len(freqs), power_db.data_ptr(), n, b["rx"].data_ptr(),
                                                             b["kind"].data_ptr(), b["status"].data_ptr(), rec.data_ptr())
            finally:
                if one_stream:
                    self.scene.set_deferred_sync(False)
                    self.scene.synchronize()
            fields_timing = self.field_scene.fields_timing()
        solve_timing = self.scene.last_timing()  # (kept by the library until the next solve)
        solve_timing["rejudge"] = self.scene.rejudge_timing()  # (zeros unless the scene is in reference-compatible mode)
        return PathBlock(buffers=b, scene=self.field_scene, n_rx=n_rx, n_paths=n, n_attempted=int(attempted),
                         power_db=power_db, frequencies=freqs, records=rec, solve_timing=solve_timing,
                         fields_timing=fields_timing)


def receiver_records(block):
    """One fixed-size record per receiver of a PathBlock, as a uint8 CUDA tensor [n_rx, record_bytes(n_freq)]:
    float64 dB per frequency (-inf without a path), then int32 path counts per type (TYPE_NAMES), the number of paths,
    the number of paths whose field status is non-zero and the largest status code (``rt_receiver_records``: two
    launches on the library's stream)."""
    import torch

    n_rx, nf, n = block.n_rx, len(block.frequencies), block.n_paths
    assert _lib.load().rt_receiver_record_bytes(nf) == record_bytes(nf)
    if getattr(block, "records", None) is not None:
        return block.records  # made by the run itself (DevicePipeline.run(records=True))
    rec = torch.empty((n_rx, record_bytes(nf)), dtype=torch.uint8, device=block.power_db.device)
    if n_rx:
        scene = block.scene
        scene.wait_for_stream(torch.cuda.current_stream(block.power_db.device).cuda_stream)
        scene.receiver_records_device(n_rx, nf, block.power_db.data_ptr(), n, block.rx.data_ptr(), block.kind.data_ptr(),
                                      block.status.data_ptr(), rec.data_ptr())
    return rec


def unpack_records(records, n_freq):
    """NumPy view of gathered records: dict(power_db [R, n_freq], counts [R, N_TYPES], n_paths, n_flagged, worst_status)."""
    raw = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1, record_bytes(n_freq))
    db = raw[:, :8 * n_freq].copy().view(np.float64).reshape(-1, n_freq)
    ints = raw[:, 8 * n_freq:8 * n_freq + 4 * N_INT_COLUMNS].copy().view(np.int32).reshape(-1, N_INT_COLUMNS)
    return dict(power_db=db, counts=ints[:, :N_TYPES], n_paths=ints[:, N_TYPES], n_flagged=ints[:, N_TYPES + 1],
                worst_status=ints[:, N_TYPES + 2])


def all_gather_records(local, n_total, dist=None):
    """The records of all receivers on every rank.  ``local``: this rank's uint8 tensor [n_local, width] for its
    contiguous block of the n_total receivers (blocks as ``shard_bounds`` deals them).  One all-gather of the padded
    blocks: on RCCL the device tensors go as they are; ``gloo`` gets host copies (made with a synchronous copy), as it
    must.  Returns a uint8 tensor [n_total, width] (on the device for RCCL, on the host for gloo or a single rank)."""
    import torch

    from .multithread_solve import shard_bounds

    if dist is None:
        assert local.shape[0] == n_total
        return local
    world = dist.get_world_size()
    width = int(local.shape[1])
    pad = (int(n_total) + world - 1) // world
    on_device = dist.get_backend() == "nccl"
    mine = local if on_device else local.cpu()
    if int(n_total) % world == 0:
        # equal blocks: the gathered tensor is the answer as it stands (no padding to add, nothing to cut out)
        assert mine.shape[0] == pad
        out = torch.empty((world * pad, width), dtype=torch.uint8, device=mine.device)
        if on_device:
            dist.all_gather_into_tensor(out, mine.contiguous())
        else:
            dist.all_gather(list(out.view(world, pad, width).unbind(0)), mine.contiguous())
        return out
    buf = torch.zeros((pad, width), dtype=torch.uint8, device=mine.device)
    buf[:mine.shape[0]] = mine
    out = torch.empty((world * pad, width), dtype=torch.uint8, device=mine.device)
    if on_device:
        dist.all_gather_into_tensor(out, buf)
    else:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        out = torch.cat(parts, dim=0)
    keep = []
    for r in range(world):
        b, e = shard_bounds(n_total, world, r)
        keep.append(out[r * pad:r * pad + (e - b)])
    return torch.cat(keep, dim=0)
