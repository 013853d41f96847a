import json,sys
d=json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"])
r=d["roofline"]
print("frac (counter traffic / HBM peak)", r["frac"], "layout x peak", r["layout_bytes"]["x_hbm_peak"], "contract x peak", r["survey_8d_contract"]["x_hbm_peak"], "avg_us", r["avg_launch_us"], r["latency_budget"]["fixed_us_per_round"], r["latency_budget"]["converged_round_us"])
print("traffic", r.get("traffic"), r.get("traffic_error"))
print(json.dumps(d["stress_k64_b8"], indent=1)[:3500])
print(d["pipeline_end_to_end"])
print(d["cpu_baseline"])
print(d["resident_loop"], d["single_registration_latency_ms"])
