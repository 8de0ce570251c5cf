"""Per-kernel summary (calls, total, average, min, max, share, registers) of a rocprofv3 rocpd database, as CSV."""
import sqlite3
import sys


def main(path, out=None):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), avg(grid_x*grid_y*grid_z), "
        "avg(workgroup_x), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,AvgGridThreads,Workgroup,VGPR,AGPR,SGPR,LDS,Scratch"]
    for r in rows:
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f},{r[6]:.0f},{r[7]:.0f},"
                     f"{r[8]},{r[9]},{r[10]},{r[11]},{r[12]}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
