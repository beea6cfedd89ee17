C="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"
PROF_TIMEOUT=200 tools/prof.sh pmcs r04_k1win "$C" python3 tools/pmc_probe.py bowl3D_h0.02 2 > gpurun_out/r04_k1win.log 2>&1
NPG_GMRES_WINDOW=0 PROF_TIMEOUT=200 tools/prof.sh pmcs r04_k1ord "$C" python3 tools/pmc_probe.py bowl3D_h0.02 2 > gpurun_out/r04_k1ord.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in ("r04_k1win", "r04_k1ord"):
    print("==", tag)
    for f in sorted(glob.glob(f"gpurun_out/{tag}_*.csv")):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "arnoldi" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for c, (s, n) in acc.items():
            print(f"  {c:36s} launches {n:4d}  avg over traced launches {s / n:16.1f}")
PY
