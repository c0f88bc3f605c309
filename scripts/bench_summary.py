import json, sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"])
print("roofline", json.dumps({k:v for k,v in d["roofline"].items() if k not in ("note","avg_launch_us_source")})[:1800])
print("h2h", d.get("value_host_to_host"), d.get("frac_host_to_host"), d.get("ms_per_step_host_to_host"))
c2=d["config2_dag"]
if "error" not in c2:
    print("config2", c2["value"], c2["ms_per_step"], c2.get("run_path"), json.dumps({k:v for k,v in c2["roofline"].items() if k not in ("note","avg_launch_us_source")})[:1500])
    print("config2 tile", c2.get("tile_kernels"), "h2h", c2["value_host_to_host"], "batch", json.dumps(c2.get("batch"))[:400])
lw=d["config5_lw"]
if "error" not in lw: print("lw", lw["value"], lw["ms_per_step"], json.dumps(lw["roofline"])[:700])
print("alarm", d["config1_alarm"].get("value"), "mid", d["mid_mixed300"].get("us_per_query"), "grid2048", d["grid2048"].get("value"))
for k in ("batch","config1_alarm","mid_mixed300","config2_dag","config5_lw","grid2048"):
    if "error" in d.get(k,{}): print("ERROR", k, d[k]["error"])
