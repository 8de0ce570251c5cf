"""The entry point that allocates nothing (rt_obstruct_segments_noalloc) against rt_obstruct_segments."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _segments(place, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = place.get_domain()
    a = np.column_stack([rng.uniform(lo[0], hi[0], n), rng.uniform(lo[1], hi[1], n), rng.uniform(0.5, 40.0, n)])
    b = np.column_stack([rng.uniform(lo[0], hi[0], n), rng.uniform(lo[1], hi[1], n), rng.uniform(0.5, 40.0, n)])
    short = rng.random(n) < 0.3  # a share of short segments, which mostly stay clear
    b[short] = a[short] + rng.normal(0.0, 3.0, (int(short.sum()), 3))
    return np.concatenate([a, b], axis=1)


def test_no_alloc_entry_point_gives_the_flags_of_the_allocating_one():
    import torch

    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom

    place = geom.load_place("data/place_saint_jean/place_saint_jean.geojson")
    scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
    dev = torch.device("cuda", scene.device)
    rng = np.random.default_rng(5)
    mask = (rng.random(scene.n) < 0.5).astype(np.uint8)
    d_mask = torch.tensor(mask, device=dev)
    # two calls in a row on the same GPU, the second smaller than the first: the library's fixed array still holds the
    # first call's hand-over list when the second starts
    for n, seed, masked in ((300_000, 1, False), (4_097, 2, False), (70_000, 3, True), (1, 4, True)):
        segs = _segments(place, n, seed)
        d_segs = torch.tensor(segs, dtype=torch.float64, device=dev)
        want = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        got = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        m_ptr = d_mask.data_ptr() if masked else None
        scene.obstructed_device(d_segs.data_ptr(), n, m_ptr, want.data_ptr())
        scene.obstructed_noalloc(d_segs.data_ptr(), n, m_ptr, got.data_ptr())
        torch.cuda.synchronize()  # the entry point does not wait for its kernels
        want_h, got_h = want.cpu().numpy(), got.cpu().numpy()
        assert set(np.unique(want_h)) <= {0, 1}
        np.testing.assert_array_equal(got_h, want_h)
        if n > 1000:
            assert 0 < want_h.sum() < n
        # and both are the host-array route's
        np.testing.assert_array_equal(scene.obstructed(segs, mask=mask if masked else None), want_h.astype(bool))
    with pytest.raises(_lib.RtError, match="at most"):
        scene.obstructed_noalloc(d_segs.data_ptr(), (1 << 22) + 1, None, got.data_ptr())
    scene.close()


def test_no_alloc_entry_point_on_a_side_stream_and_replayed_from_a_graph():
    """The two callers the entry point is for: a stream of the caller's own (no allocation during the call: PyTorch's
    and the device's free memory stay where they were), and a captured HIP graph replayed several times with eager work
    and synchronisations in between, every replay giving the eager flags."""
    import torch

    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom

    place = geom.load_place("data/place_saint_jean/place_saint_jean.geojson")
    scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
    dev = torch.device("cuda", scene.device)
    n = 50_000
    segs = _segments(place, n, 11)
    d_segs = torch.tensor(segs, dtype=torch.float64, device=dev)
    want = torch.zeros(n, dtype=torch.uint8, device=dev)
    got = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    scene.obstructed_device(d_segs.data_ptr(), n, None, want.data_ptr())
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)  # default priority
    with torch.cuda.stream(side):
        got.fill_(9)  # (the runtime sets a stream's queue up at its first launch, 2 MiB: not the entry point's doing)
    side.synchronize()
    allocated, free = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
    scene.obstructed_noalloc(d_segs.data_ptr(), n, None, got.data_ptr(), stream=side.cuda_stream)
    side.synchronize()
    assert torch.cuda.memory_allocated(dev) == allocated and torch.cuda.mem_get_info(dev)[0] == free
    assert torch.equal(got, want) and 0 < int(want.sum()) < n

    graph = torch.cuda.CUDAGraph()
    got.fill_(9)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):  # (capture refuses any allocation made while it records)
        scene.obstructed_noalloc(d_segs.data_ptr(), n, None, got.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
    for replay in range(3):
        got.fill_(9)
        # eager work of the program's own between the replays: the allocating entry point on other segments
        other = torch.tensor(_segments(place, 3_000 + replay, 20 + replay), dtype=torch.float64, device=dev)
        flags = torch.zeros(len(other), dtype=torch.uint8, device=dev)
        scene.obstructed_device(other.data_ptr(), len(other), None, flags.data_ptr())
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"replay {replay}"
    scene.close()
