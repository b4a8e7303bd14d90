#!/usr/bin/env python3
"""Per-dispatch timeline of one step from a rocprofv3 --kernel-trace run (rocpd .db or *_kernel_trace.csv):
    python tools/timeline.py <file> <marker substring> [step index]
prints start offset, duration, kernel, grid for the dispatches between two launches of the marker kernel."""
import csv, sqlite3, sys


def load(path):
    if path.endswith(".db"):
        cur = sqlite3.connect(path).cursor()
        return [(n, s, e, f"{gx}x{gy}x{gz}") for n, s, e, gx, gy, gz in
                cur.execute("select name,start,end,grid_x,grid_y,grid_z from kernels order by start")]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}") for r in rows]


def short(n):
    return n.replace("stgcn::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]


rows = load(sys.argv[1])
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) - 2
a, b = idx[k], idx[k + 1]
t0 = rows[a][1]
print(f"step {k}: {(rows[b][1] - t0) / 1e3:.1f} us, {b - a} launches, busy {sum(r[2] - r[1] for r in rows[a:b]) / 1e3:.1f} us")
for r in rows[a:b]:
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  {short(r[0])}  grid={r[3]}")
