"""The measurement line that bench.py prints, checked on the line committed under profiles/ (no GPU needed): the
fields the driver reads are there, and the derived figures follow from the line's own inputs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def line():
    with open(os.path.join(ROOT, "profiles", "r3_bench_line.json")) as f:
        return json.load(f)


def test_the_committed_line_has_the_fields_of_the_contract(line):
    for key, kind in dict(metric=str, value=float, unit=str, n_gpus=int, steps=int, warmup=int, ms_per_step=float,
                          higher_is_better=bool, scaling=str, dtype=str, data=str, config=dict, roofline=dict,
                          cpu_baseline=dict).items():
        assert isinstance(line[key], kind), key
    assert line["vs_baseline"] is None  # the reference publishes no number for this metric
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "strong"
    assert "lln.geojson" in line["config"]["workload"] and "model" not in line["config"]
    assert line["dtype"] == "f64" and line["data"] == "synthetic"
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert line["cpu_baseline"]["kind"] in ("port", "reference")


def test_value_and_roofline_follow_from_the_lines_own_inputs(line):
    per_step = line["config"]["path_solves_per_step"]
    assert per_step == line["config"]["receivers"] * line["config"]["candidates_per_receiver"]
    # (every receiver has the line of sight as one more candidate beside the groups' chains and edges)
    assert sum(line["config"]["candidates_by_group"]) + 1 == line["config"]["candidates_per_receiver"]
    assert line["value"] == pytest.approx(per_step / (line["ms_per_step"] * 1e-3), rel=1e-9)
    for name, r in [("headline", line["roofline"])] + list(line["roofline_by_kernel"].items()):
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9), name
        assert 0.0 <= r["frac"] <= 1.0, name  # a fraction of a peak: executed work, not the brute-force count
        ms = r.get("ms_per_step", r.get("ms"))
        work = r.get("work_per_step", r.get("work"))
        scale = 1e12 if r["unit"] == "TFLOP/s" else 1e9
        assert r["achieved"] == pytest.approx(work / (ms * 1e-3) / scale, rel=1e-6), name
    # the headline roofline is the kernel family with the most device time in a step
    fam = line["kernel_ms_per_step"]
    busiest = max(("reflect", "diffract", "obstruct", "rejudge", "fields"), key=lambda k: fam[k])
    assert line["roofline"]["ms_per_step"] == pytest.approx(fam[busiest]) and line["roofline"]["family"] == busiest
    # kernels of a step (events around every launch) add up to the step within the host's share
    assert sum(fam[k] for k in ("reflect", "diffract", "obstruct", "finalize", "rejudge", "fields_call")) < 1.1 * line["ms_per_step"]
    # the headline is measured in the mode that gives the reference's path set, the other mode is beside it
    assert line["config"]["solver_mode"] == "reference" and line["other_solver_mode"]["solver_mode"] == "exact"
    assert line["config"]["paths_before_and_after_rejudge_rank0"] == [173760, 172630]
    assert line["coverage_map_wall_s"] > line["coverage_map"]["precompute_s"]


def test_a_rank_count_that_differs_from_the_launchers_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert run.returncode != 0
    assert "WORLD_SIZE=3" in run.stderr and run.stdout.strip() == ""
