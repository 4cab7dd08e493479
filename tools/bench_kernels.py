import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    print("ms_per_step", d["ms_per_step"])
    for e in d.get("kernels",[]):
        if e["kernel"] in ("k_ode_nn","k_ode_sing","k_int1<field>","k_int1<linear>","k_fftz"): print("  ",e["kernel"],e["calls"],round(e["avg_ms"],4))
