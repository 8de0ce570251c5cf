"""Prints the kernel timeline of a few steady-state solve steps from a rocprofv3 --kernel-trace CSV (not a test)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_los_emit"
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(first) or (" " + first) in r["Kernel_Name"][:60]]
print(len(rows), "dispatches,", len(idx), "steps")
for j in idx[10:13]:
    t0 = int(rows[j]["Start_Timestamp"])
    k = j
    while True:
        r = rows[k]
        print(f'{r["Kernel_Name"][:48]:48s} start {int(r["Start_Timestamp"]) - t0:8d} ns  dur '
              f'{int(r["End_Timestamp"]) - int(r["Start_Timestamp"]):7d} ns  grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}')
        k += 1
        if k >= len(rows) or k in idx:
            if k < len(rows):
                print(f'next step starts at {int(rows[k]["Start_Timestamp"]) - t0} ns')
            break
    print()
