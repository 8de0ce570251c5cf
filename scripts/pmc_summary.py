"""Sums rocprofv3 --pmc counter_collection CSVs per kernel name: python scripts/pmc_summary.py file.csv [file.csv ...]"""
import csv
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for path in paths:
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0]
                acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[name].add(row["Dispatch_Id"])
    for name, counters in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_WAIT_INST_ANY", 0))):
        print(f"{name[:60]:60s} dispatches={len(calls[name])}")
        for k, v in sorted(counters.items()):
            print(f"    {k:28s} {v:18.0f}")


if __name__ == "__main__":
    main(sys.argv[1:])
