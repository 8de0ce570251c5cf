"""Prints, for the last visibility call of scripts/vis_host_probe.py traced with rocprofv3 --kernel-trace --hip-trace
--output-format csv, the host API calls longer than 0.2 ms and the kernels, in time order (ms from the call's start)."""
import csv
import glob
import sys

d = sys.argv[1]
api = list(csv.DictReader(open(glob.glob(d + "/*hip_api_trace.csv")[0])))
ker = list(csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])))
k_end = max(int(k["End_Timestamp"]) for k in ker if k["Kernel_Name"].startswith("k_symmetrise"))
# the last call: from 232 ms before the end of its last kernel to 3 ms after it
t0 = k_end - 232_000_000
k_end += 3_000_000
api = [a for a in api if int(a["Start_Timestamp"]) < k_end]
ker = [k for k in ker if int(k["Start_Timestamp"]) < k_end]
rows = []
for a in api:
    s, e = int(a["Start_Timestamp"]), int(a["End_Timestamp"])
    if e > t0 and e - s > 100_000:
        rows.append((s, f"API  {a['Function']:28s} {(e - s) / 1e6:8.2f} ms"))
for k in ker:
    s, e = int(k["Start_Timestamp"]), int(k["End_Timestamp"])
    if e > t0:
        rows.append((s, f"KERN {k['Kernel_Name'][:60]:60s} {(e - s) / 1e6:8.2f} ms  (ends {(e - t0) / 1e6:7.2f})"))
for s, line in sorted(rows):
    print(f"{(s - t0) / 1e6:8.2f}  {line}")
