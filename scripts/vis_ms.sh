# visibility kernel times of the bench's precompute (not a test): bash scripts/vis_ms.sh
cd /root/repo
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-config2 --steps 5 --warmup 2 > gpurun_out/vis_ms.json 2> gpurun_out/vis_ms.err || exit 1
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/vis_ms.json").read().strip().splitlines()[-1])
print("coverage_map_wall_s", d["coverage_map_wall_s"], d["coverage_map"]["precompute_breakdown_rank0"])
print(d["coverage_map"]["visibility_kernels_ms_rank0"])
print("ms_per_step", d["ms_per_step"], d["kernel_ms_per_step"])
PY
