import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f=j["opus_file_decode"]
for k in (None,"long_streams","surround_7_1"):
    g=f if k is None else f[k]
    print(k, "files/s", round(g["files_per_sec"],1), "vs_cpu", round(g.get("vs_cpu_baseline") or 0,3), "busy", round(g["gpu_busy_fraction"],3), "entropy_s", g["breakdown_cpu_entropy_stage_s"], "wall", round(g["wall_seconds"],4))
print("value", j["value"], "roofline", j["roofline"]["frac"], j["roofline"].get("frac_of_measured_copy"))
