import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import parity_configs as pc
from raytracing_amd.ray_tracing import RayTracingProblem
w = pc.workload("c3_lln")
rx = w["rx_all"][w["rx_index"]]
place = w["place"]; place.add_set_of_points(rx)
problem = RayTracingProblem(np.asarray(w["tx"]).reshape(1, 3), place)
print("popcount", int(problem.visibility_matrix.sum()), "emitter", int(problem.emitter_visibility.sum()), "edges", len(problem.sharp_edges))
np.save("gpurun_out/lln_matrix_gpu.npy", np.packbits(problem.visibility_matrix.astype(bool)))
