"""Kernels of the last timed step of `bench.py --no-cpu-baseline --no-config2 --steps 2 --warmup 1` traced with
rocprofv3 --kernel-trace --output-format csv, in time order with start, duration and the gap to the previous kernel's
end (us).  The steps with per-launch events come after the timed ones; the last k_describe marks the last step."""
import csv
import glob
import sys

ker = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0])))
ker.sort(key=lambda k: int(k["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
starts = [i for i, k in enumerate(ker) if k["Kernel_Name"].startswith("k_rx_boxes")]
first = starts[which]
last = starts[which + 1] if which + 1 < 0 and which + 1 != 0 else len(ker)
if which != -1:
    last = starts[which + 1]
t0 = int(ker[first]["Start_Timestamp"])
prev_end = t0
for k in ker[first:last]:
    s, e = int(k["Start_Timestamp"]), int(k["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  q{k.get('Queue_Id', '?'):>2s}  {k['Kernel_Name'][:90]}")
    prev_end = max(prev_end, e)
