import json,sys
for f in (sys.argv[1:] or ["-"]):
    d=json.load(sys.stdin if f == "-" else open(f)); r=d["roofline"]
    print(f, round(d["value"]), round(d["ms_per_step"],3), round(r["avg_launch_ms"],4), round(r["frac"],3), round(r["whole_forward_frac"],3), r.get("clock_ghz"))
