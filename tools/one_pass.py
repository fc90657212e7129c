"""All kernels of ONE steady-state contig pass of a rocprofv3 kernel trace, in start order: name, queue, start offset, duration, grid.
usage: python tools/one_pass.py <dir with *_kernel_trace.csv> [which pass, counted from the end: default 20]"""
import csv, glob, os, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
scans = [i for i, r in enumerate(rows) if "k_wire_expand" in r["Kernel_Name"]]
a, b = scans[-back], scans[-back + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-34s q%-3s +%9.1f us  %8.1f us  grid %s wg %s" % (nm, r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
