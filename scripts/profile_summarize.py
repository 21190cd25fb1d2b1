"""Turns gpurun_out/<tag> (scripts/profile_round.sh) into the committed profiles/<tag>_* files."""
import csv, glob as _glob, json, os, sys, collections, shutil

class glob:                      # gpurun merges a call's files into what earlier calls left: always take the newest match
    @staticmethod
    def glob(pattern):
        return sorted(_glob.glob(pattern), key=os.path.getmtime, reverse=True)[:1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join("profiles", "%s_kernel_stats.csv" % tag))
for ks1 in glob.glob(os.path.join(src, "trace_single", "*", "*_kernel_stats.csv")):
    shutil.copy(ks1, os.path.join("profiles", "%s_kernel_stats_single_pipeline.csv" % tag))
def kname(n):
    if "k_trace<false, false" in n: return "k_trace_first"      # first launch of a trace step (every ray, budgeted)
    if "k_trace<false, true" in n: return "k_trace_resume"      # second launch (the rays set aside)
    if "k_trace<true" in n: return "k_trace_count"               # the counting render's plain traversal
    for k in ("k_trace", "k_shade", "k_generate", "k_resolve", "k_finalize", "k_untile", "k_extend", "k_connect"):
        if k in n: return k
    return None
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC", "SQ", "SQ2", "TA1", "TA2"):
    fs = glob.glob(os.path.join(src, "pmc_" + c, "*", "*_counter_collection.csv"))
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = kname(r["Kernel_Name"])
        if not k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in agg:
        for cn, v in agg[k].items(): out[k][cn] = v
        out[k]["dispatches"] = len(disp[k])
import bench
res = {"tag": tag, "kernel_source_sha": bench.kernel_source_sha(),
       "workload": "bench.py --steps 1 --warmup 0 --spp 128 --no-cpu-baseline --single-pipeline --no-count-step (cfg3 scene, 1024x1024; exactly ONE 128 Mi-slot pass "
                   "is rendered = the launch sizes of the 256-spp bench), one PMC counter set per run",
       "note": "FETCH_SIZE/WRITE_SIZE are in KiB, summed over the kernel's dispatches; on gfx950 FETCH_SIZE tallies 128-B "
               "requests as 64 B (MI355X_MICROARCH.md, HBM section): read bytes = 2 x FETCH_SIZE x 1024; counts fabric-side "
               "requests including Infinity-Cache hits", "kernels": {}}
for k, a in out.items():
    d = {"dispatches": a.get("dispatches")}
    if "FETCH_SIZE" in a: d["FETCH_SIZE_KiB"] = a["FETCH_SIZE"]; d["read_bytes_corrected"] = 2 * 1024 * a["FETCH_SIZE"]
    if "WRITE_SIZE" in a: d["WRITE_SIZE_KiB"] = a["WRITE_SIZE"]; d["write_bytes"] = 1024 * a["WRITE_SIZE"]
    if "TCC_HIT_sum" in a: d["l2_hit_rate"] = a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a["TCC_MISS_sum"], 1); d["TCC_REQ_sum"] = a["TCC_REQ_sum"]
    if "read_bytes_corrected" in d and "write_bytes" in d and d["dispatches"]:
        d["hbm_bytes_per_launch"] = (d["read_bytes_corrected"] + d["write_bytes"]) / d["dispatches"]
    if "SQ_INSTS_VALU" in a and a.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a wave64 VALU instruction occupies its SIMD for 4 cycles,
        # 256 CUs x 4 SIMDs: peak = 256 wave-instructions per cycle
        cycles = a["GRBM_GUI_ACTIVE"] / 8.0
        d["valu_wave_insts"] = a["SQ_INSTS_VALU"]; d["salu_insts"] = a.get("SQ_INSTS_SALU"); d["gpu_cycles"] = cycles
        d["valu_utilization"] = a["SQ_INSTS_VALU"] / (cycles * 256.0)
        if a.get("SQ_THREAD_CYCLES_VALU"): d["valu_active_lanes_avg"] = a["SQ_THREAD_CYCLES_VALU"] / a["SQ_INSTS_VALU"]
    extra = {cn: v for cn, v in a.items() if cn.startswith(("TA_", "SQ_WAVE", "SQ_BUSY", "SQ_ACTIVE", "SQ_WAIT", "SQ_WAVES", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"))}
    if extra: d["other_counters"] = extra
    res["kernels"][k] = d
kf, kr = res["kernels"].get("k_trace_first"), res["kernels"].get("k_trace_resume")
if kf and kf.get("hbm_bytes_per_launch") is not None:
    # one trace step = first launch + resume launch (same number of dispatches)
    res["hbm_bytes_per_launch"] = kf["hbm_bytes_per_launch"] + ((kr or {}).get("hbm_bytes_per_launch") or 0.0)
    res["hbm_bytes_per_launch_note"] = "k_trace_first + k_trace_resume, per trace step"
if kf and kf.get("valu_utilization") is not None:
    res["trace_valu_utilization"] = {"first": kf["valu_utilization"], "resume": (kr or {}).get("valu_utilization")}
    res["trace_active_lanes"] = {"first": kf.get("valu_active_lanes_avg"), "resume": (kr or {}).get("valu_active_lanes_avg")}
# path iterations of one 64-spp pass, from the counting render of the kernel-trace run's bench line
try:
    line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith("{")][-1]
    b = json.loads(line)
    iters = b["work"]["path_iterations_per_sample"] * 1024 * 1024 * 128
    res["path_iterations_per_pass"] = iters
    ksh = res["kernels"].get("k_shade")
    if ksh and "read_bytes_corrected" in ksh and "write_bytes" in ksh:
        res["shade_read_bytes_per_path_iteration"] = ksh["read_bytes_corrected"] / iters
        res["shade_write_bytes_per_path_iteration"] = ksh["write_bytes"] / iters
        res["shade_bytes_per_path_iteration"] = (ksh["read_bytes_corrected"] + ksh["write_bytes"]) / iters
        res["shade_bytes_note"] = "k_shade fabric bytes of the pass / (path, bounce) shading steps of the pass; algorithmic figure (SURVEY 8d): 156 B"
    if ksh and ksh.get("valu_utilization") is not None: res["shade_valu_utilization"] = ksh["valu_utilization"]
    res["bench_line_of_the_kernel_trace_run"] = {k: b[k] for k in ("value", "ms_per_step") if k in b}
except Exception as e:
    res["path_iterations_per_pass"] = None; res["note_iters"] = "no bench line found: %s" % e
# per-dispatch timeline of the single pass
tl = glob.glob(os.path.join(src, "pass_timeline", "*", "*_kernel_trace.csv"))
if tl:
    rows = sorted(csv.DictReader(open(tl[0])), key=lambda r: int(r["Start_Timestamp"]))
    seq = []
    for r in rows:
        k = kname(r["Kernel_Name"])
        if k: seq.append([k, round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1)])
    res["pass_timeline_us"] = seq
json.dump(res, open(os.path.join("profiles", "%s_pmc_traffic.json" % tag), "w"), indent=1)
print(json.dumps(res, indent=1))
print(open(os.path.join("profiles", "%s_kernel_stats.csv" % tag)).read()[:1500])
