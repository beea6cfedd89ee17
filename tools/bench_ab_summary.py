import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), d["config"]["gmres_iterations_per_step"], round(d["roofline"]["avg_launch_us"],2), d["roofline"]["launches"], round(d["roofline"]["frac"],4))
    except Exception as e:
        print(f, "ERR", e)
