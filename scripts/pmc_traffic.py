"""HBM bytes per launch of the solve's kernel families from two rocprofv3 --pmc passes over bench.py
(FETCH_SIZE in one, WRITE_SIZE in the other: together they need more TCC slots than one pass has).

    python scripts/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so wide
streaming reads show at half their size; the read side is reported both as counted and doubled (upper bound)."""
import csv
import json
import sys
from collections import defaultdict

FAMILIES = {"reflect": ("k_reflect",), "obstruct": ("k_obstruct",), "diffract": ("k_diffract",),
            "finalize": ("k_finalize", "k_los_finalize", "k_los_segments")}


def per_kernel(path, counter):
    total, calls = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            total[name] += float(row["Counter_Value"])
            calls[name].add(row["Dispatch_Id"])
    return total, calls


def main(fetch_csv, write_csv, out_json):
    fetch, fcalls = per_kernel(fetch_csv, "FETCH_SIZE")
    write, wcalls = per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for fam, prefixes in FAMILIES.items():
        names = [n for n in fetch if any(p in n.split("(")[0] for p in prefixes)]
        n_launch = sum(len(fcalls[n]) for n in names)
        if n_launch == 0:
            continue
        rd = 1024.0 * sum(fetch[n] for n in names) / n_launch
        wr_launch = sum(len(wcalls[n]) for n in names if n in wcalls) or 1
        wr = 1024.0 * sum(write.get(n, 0.0) for n in names) / wr_launch
        out[fam] = {"read_bytes_per_launch_as_counted": rd, "read_bytes_per_launch_doubled": 2 * rd,
                    "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr, "launches_profiled": n_launch,
                    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over bench.py (separate passes)"}
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
