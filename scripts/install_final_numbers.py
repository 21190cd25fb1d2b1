#!/usr/bin/env python3
"""Copies what scripts/gpu_final_numbers.sh wrote under gpurun_out/r03_final/ into profiles/r03_* and prints the headline figures."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = "gpurun_out/%s_final/" % tag
m = {"bench_line": "bench_line", "bench_configs": "bench_configs", "first_call": "first_call", "host_boundary": "host_boundary",
     "bench_fanout_1": "bench_fanout_1gpu", "bench_2ranks_gloo_one_gpu": "bench_2ranks_gloo_one_gpu"}
for a, b in m.items():
    open("profiles/%s_%s.json" % (tag, b), "w").write(open(src + a + ".json").read())
last = open(src + "scaling_projection.json").read().strip().split("\n")[-1]
open("profiles/%s_scaling_projection.json" % tag, "w").write(last + "\n")
d = json.loads(open("profiles/%s_bench_line.json" % tag).read().strip().split("\n")[-1])
r = d["roofline"]
print("bench %.1f Msamples/s, %.2f ms, frac %.3f, traffic %s (%s)" % (d["value"], d["ms_per_step"], r["frac"], r["traffic"], r["traffic_from"]))
print("verify", {k: v for k, v in d["verify"].items() if not isinstance(v, dict)}, d["verify"]["work_counts_equal_host_walk"]["equal"], d["verify"]["oracle_window"]["bit_identical"])
print("projection", [(x["G"], round(x["rank_ms_max"], 2), round(x["projected_efficiency"], 3)) for x in json.loads(last)["rows"]])
for k, v in json.load(open("profiles/%s_bench_configs.json" % tag)).items():
    print(k, round(v["ms"], 2), round(v["Msamples_per_s"], 1))
