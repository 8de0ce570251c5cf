"""Would a test "the ray leaves the edge into the solid behind both of the edge's faces" remove only paths that are
obstructed anyway?  On the lln map: the paths the solve keeps are checked against the test (none may satisfy it), and
the share of (edge, receiver) pairs it would remove is estimated on random pairs.  Not a test."""
import sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, ".")
from raytracing_amd import scenarios
from raytracing_amd.ray_tracing import RayTracingProblem

place, tx, rx = scenarios.lln_grid(100)
place.add_set_of_points(rx)
problem = RayTracingProblem(tx.reshape(1, 3), place)
problem.solve(max_order=2)
batch = problem._pending[0]
polygons = place.get_polygons_list()
planes = np.array([p.get_parametric() for p in polygons])
sel = np.flatnonzero(batch["has_edge"].astype(bool))
n_refl = batch["n_refl"][sel]
pts = batch["points"][sel]
D = pts[np.arange(len(sel)), n_refl]                  # the diffraction point follows the reflection points
R = rx[batch["rx"][sel]]
prev = np.where((n_refl > 0)[:, None], pts[np.arange(len(sel)), np.maximum(n_refl - 1, 0)], tx[None, :])
key = batch["edge_key"][sel]
n1, n2 = planes[key[:, 0], :3], planes[key[:, 1], :3]
def behind_both(P):
    v = P - D
    v /= np.linalg.norm(v, axis=1)[:, None]
    return (np.einsum("ij,ij->i", v, n1) < -1e-3) & (np.einsum("ij,ij->i", v, n2) < -1e-3)
print("paths with an edge:", len(sel), "| receiver side behind both faces:", int(behind_both(R).sum()),
      "| emitter side behind both faces:", int(behind_both(prev).sum()), flush=True)
# share of random (edge, receiver) pairs of the emitter-visible edges
rng = np.random.default_rng(0)
keys_ = list(problem.sharp_edges.keys())
edges = problem._edge_array
pairs = np.array([list(k) for k in keys_])
print(edges.shape, pairs.shape, flush=True)
m = 200000
ei = rng.integers(0, len(edges), m)
ri = rng.integers(0, len(rx), m)
mid = edges[ei].mean(axis=1)
bid = np.array([(-1 if p.building_id is None else hash(p.building_id) % (1 << 30)) for p in polygons])
same_k = bid[key[:, 0]] == bid[key[:, 1]]
print('kept paths behind both faces with both faces of one building:', int((behind_both(R) & same_k).sum()), 'of', int(same_k.sum()), 'edge paths of one building', flush=True)
print('which kept paths:', np.unique(key[behind_both(R)], axis=0)[:10].tolist(), [polygons[i].part if hasattr(polygons[i], 'part') else None for i in np.unique(key[behind_both(R)])[:10]], flush=True)
v = rx[ri] - mid
v /= np.linalg.norm(v, axis=1)[:, None]
a = np.einsum("ij,ij->i", v, planes[pairs[ei, 0], :3]) < -1e-3
b = np.einsum("ij,ij->i", v, planes[pairs[ei, 1], :3]) < -1e-3
same = bid[pairs[ei, 0]] == bid[pairs[ei, 1]]
print("share of random pairs whose edge joins faces of one building:", float(same.mean()), "| of all pairs, behind both faces of one building:", float((a & b & same).mean()))
print("random (edge, receiver) pairs behind both faces:", float((a & b).mean()), "behind one at least:", float((a | b).mean()))
