"""The collectives of the sharded path over RCCL, with a process group of ONE rank on the one GPU of the test box: the
device-tensor branches (rows of the visibility matrix gathered and symmetrised on the GPU, one fixed record per
receiver gathered as bytes) that the two-rank CPU tests only reach through gloo's host tensors.  Several ranks cannot
share a GPU under RCCL, so the multi-rank arithmetic (interleaved rows, padded blocks) stays with the gloo tests."""
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_rows_and_records_gathered_over_rccl_are_those_of_the_single_process(places, rccl_group):
    import torch

    from raytracing_amd import _lib, scenarios
    from raytracing_amd import device_pipeline as dp
    from raytracing_amd import geometry as geom
    from raytracing_amd import multithread_solve as mts

    dist = rccl_group
    place = places("place_levant")
    polygons = place.get_polygons_list()
    lo, hi = place.get_domain()
    tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 20.0)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    n = len(polygons)

    # rows painted by "this rank" (all of them), gathered on the device, symmetrised there
    rows = mts._gather_dealt_rows_tensor(lambda ids: scene.visibility_rows(ids), n, n, dist)
    assert rows.is_cuda and rows.dtype == torch.uint8
    matrix = (rows | rows.T).contiguous()
    torch.cuda.synchronize()
    want_matrix, want_screens = scene.visibility_with_faces(screens)
    np.testing.assert_array_equal(matrix.cpu().numpy().astype(bool), want_matrix)
    np.testing.assert_array_equal(mts.gather_visibility_rows(lambda ids: scene.visibility_rows(ids), n, dist), want_matrix)

    # the sharded call itself: rows and screens painted into the buffer the all-gather sends, matrix left on the device
    d_matrix, got_screens = mts.sharded_visibility(scene, screens, dist=dist, on_device=True)
    assert d_matrix.is_cuda
    np.testing.assert_array_equal(d_matrix.cpu().numpy().astype(bool), want_matrix)
    np.testing.assert_array_equal(got_screens, want_screens)
    h_matrix, got_screens = mts.sharded_visibility(scene, screens, dist=dist)
    np.testing.assert_array_equal(h_matrix, want_matrix)
    np.testing.assert_array_equal(got_screens, want_screens)

    # the device matrix goes into the enumeration as it is; one record per receiver through the collective
    ev = np.logical_or.reduce(want_screens)
    pairs, edges = scene.sharp_edges()
    cand = scene.enumerate(matrix, ev, pairs, edges, 2)
    rx = scenarios.lattice_receivers(place, 12, 1.5)
    d_rx = torch.tensor(rx, dtype=torch.float64, device="cuda:0").reshape(-1, 3)
    pipe = dp.DevicePipeline(scene, cand, tx, capacity=1 << 16)
    block = pipe.run(d_rx, [12.5e9], records=True)
    local = dp.receiver_records(block)
    gathered = dp.all_gather_records(local, len(rx), dist)
    assert gathered.is_cuda and gathered.shape == local.shape
    torch.cuda.synchronize()
    assert torch.equal(gathered, local)
    got = dp.unpack_records(gathered.cpu().numpy(), 1)
    assert got["n_paths"].sum() == block.n_paths > 0
    assert np.isfinite(got["power_db"]).sum() > 0
    cand.close()
    scene.close()
