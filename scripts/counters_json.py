"""Per-kernel figures from the rocprofv3 --pmc passes of bench.py (scripts/profile_round3.sh), written as the JSON that
bench.py quotes in its roofline entries:

    python scripts/counters_json.py fetch.csv write.csv sq.csv [f64.csv] out.json

hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
read requests at 64 bytes, so it is doubled as the hardware guide prescribes; the doubling is calibrated for wide
streaming reads and is an upper bound for narrower ones); the wave_cycles_* shares say where the cycles of the kernel's
waves go (waiting for memory or a barrier, stalled at issue, issuing; SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY
and SQ_ACTIVE_INST_VALU over SQ_WAVE_CYCLES, summed over the kernel's dispatches).

fp64_flop_per_launch = the double-precision operations the kernel's lanes actually executed: (SQ_INSTS_VALU_ADD_F64 +
SQ_INSTS_VALU_MUL_F64 + SQ_INSTS_VALU_TRANS_F64 + 2 x SQ_INSTS_VALU_FMA_F64) wave instructions x 64 lanes x the share of
lanes that were switched on while vector instructions ran (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)), per
dispatch.  launches_per_step = dispatches of the kernel / steps in the profiled run (steps = dispatches of the final
filter, one per step on this workload).  Kernels are keyed by name without template arguments."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name].add(row["Dispatch_Id"])
    return acc, {k: len(v) for k, v in calls.items()}


def main(argv):
    out_json = argv[-1]
    fetch, fcalls = per_kernel(argv[0])
    write, wcalls = per_kernel(argv[1])
    sq, scalls = per_kernel(argv[2])
    f64, f64_calls = per_kernel(argv[3]) if len(argv) > 4 else ({}, {})
    f64_steps = float(max(f64_calls.get("k_finalize", 1), 1))
    out = {}
    for name in sorted(set(fetch) | set(sq)):
        e = {"source": "rocprofv3 --pmc passes over bench.py --no-cpu-baseline --no-config2 --no-other-mode --no-extra-configs "
                       "--steps 2 --warmup 1 (one pass per counter group; scripts/profile_round3.sh)"}
        if name in fetch and fcalls.get(name):
            rd = 1024.0 * fetch[name]["FETCH_SIZE"] / fcalls[name]
            wr = 1024.0 * write.get(name, {}).get("WRITE_SIZE", 0.0) / max(wcalls.get(name, 1), 1)
            e.update(read_bytes_per_launch_as_counted=rd, read_bytes_per_launch_doubled=2 * rd, write_bytes_per_launch=wr,
                     hbm_bytes_per_launch=2 * rd + wr, launches_profiled=fcalls[name])
        if name in sq:
            c = sq[name]
            if c.get("SQ_BUSY_CYCLES"):
                # as the two counters come: they are tallied over different units of the chip, so this is a figure to
                # compare between versions of one kernel, not a fraction of a peak
                e["active_inst_valu_over_busy_cycles_as_counted"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_BUSY_CYCLES"]
            if c.get("SQ_WAVE_CYCLES"):
                # WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (disjoint): where a wave's cycles go
                e["wave_cycles_waiting_share"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
                e["wave_cycles_issue_stalled_share"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
                e["wave_cycles_issuing_share"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
                e["wave_cycles_issuing_valu_share"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
            if c.get("SQ_WAVES"):
                e["valu_instructions_per_wave"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_WAVES"]
            e["sq_counters"] = {k: v for k, v in sorted(c.items())}
        if name in f64:
            c = f64[name]
            n64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                               "SQ_INSTS_VALU_TRANS_F64"))
            if c.get("SQ_INSTS_VALU"):
                e["fp64_share_of_valu_instructions"] = n64 / c["SQ_INSTS_VALU"]
            ops = n64 + c.get("SQ_INSTS_VALU_FMA_F64", 0.0)  # a fused multiply-add is two operations
            lanes = 1.0
            if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
                lanes = min(1.0, c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]))
                e["lanes_on_share_of_valu_cycles"] = lanes
            n_disp = max(f64_calls.get(name, 1), 1)
            e["fp64_flop_per_launch_all_lanes"] = 64.0 * ops / n_disp
            e["fp64_flop_per_launch"] = 64.0 * ops * lanes / n_disp
            e["launches_per_step"] = n_disp / f64_steps
            e["steps_profiled"] = f64_steps
            e["f64_counters"] = {k: v for k, v in sorted(c.items())}
        out[name] = e
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for name, e in out.items():
        print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if not k.endswith("counters") and k != "source"})


if __name__ == "__main__":
    main(sys.argv[1:])
