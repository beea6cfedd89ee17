"""Anatomy of the multigrid V-cycle from a rocprofv3 kernel trace.  The SpMV kernel serves every operator of every level, so the
dispatches are told apart by their POSITION in the cycle: the cycle's launch sequence is fixed (mg.hip: mg_cycle / mg_smooth), the
coarsest level's dense GEMV marks its centre, and position k relative to that marker is the same operator in every cycle.

    python3 tools/mg_anatomy.py <dir or *_kernel_trace.csv> [launches before the marker=32] [launches after it=33]

Prints the average duration per position (down leg, coarsest solve, up leg) and the sum - the V-cycle's kernel time."""
import csv, glob, os, sys

src = sys.argv[1]
a = int(sys.argv[2]) if len(sys.argv) > 2 else 32
b = int(sys.argv[3]) if len(sys.argv) > 3 else 33
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("npg::", "") for r in rows]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
wgs = [int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) for r in rows]
marks = [i for i, n in enumerate(name) if "k_dense_gemv_part" in n]       # (the fp16 instance's name comes back mangled)
marks = marks[len(marks) // 4:]                      # skip the first cycles (set-up, warm-up)
if not marks:
    print("no dense GEMV launch found; kernels with 'gemv' in their names:", sorted({n for n in name if "gemv" in n}))
    sys.exit(1)
acc = {}
for g in marks:
    if g - a < 0 or g + b >= len(rows):
        continue
    for k in range(-a, b + 1):
        key = (k, name[g + k], wgs[g + k])
        s = acc.setdefault(key, [0, 0])
        s[0] += 1
        s[1] += dur[g + k]
print(f"# {len(rows)} dispatches, {len(marks)} V-cycles averaged; position 0 = the coarsest level's dense GEMV")
tot, gaps = 0.0, 0.0
for k in range(-a, b + 1):
    cands = sorted(((v[0], key, v[1]) for key, v in acc.items() if key[0] == k), reverse=True)
    if not cands:
        continue
    n, key, t = cands[0]
    flag = "" if n == len(marks) else f"   (!) {n} of {len(marks)} cycles"
    print(f"{k:4d}  {key[1][:44]:44s} wgs {key[2]:5d}  {t / n / 1e3:8.1f} us{flag}")
    tot += t / n / 1e3
wall = [int(rows[g + b]["End_Timestamp"]) - int(rows[g - a]["Start_Timestamp"]) for g in marks if g - a >= 0 and g + b < len(rows)]
print(f"kernel time per V-cycle {tot:.1f} us; first start to last end {sum(wall) / len(wall) / 1e3:.1f} us (eager launches under the profiler)")
