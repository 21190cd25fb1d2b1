"""gpurun_out/<tag> (scripts/profile_bdpt.sh) -> profiles/<tag>_kernel_stats.csv + profiles/<tag>_pmc.json"""
import csv, glob as _glob, json, os, re, shutil, sys, collections

class glob:                      # gpurun merges a call's files into what earlier calls left: always take the newest match
    @staticmethod
    def glob(pattern):
        return sorted(_glob.glob(pattern), key=os.path.getmtime, reverse=True)[:1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r02_bdpt"
src = os.path.join("gpurun_out", tag)
shutil.copy(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0], os.path.join("profiles", "%s_kernel_stats.csv" % tag))
def kname(n):
    m = re.search(r"(k_bdpt_\w+|k_resolve|k_finalize|k_untile|k_tri_frames)", n)
    return m.group(1) if m else None
res = {"tag": tag, "kernel_source_sha": bench.kernel_source_sha(),
       "workload": "scripts/bench_configs.py restricted to the BDPT configurations (config 1: input.txt 256^2 x 4 spp x spl 8, 4 renders; "
                   "input.txt 1024^2 x 8 spp, 3 renders; config 4: mis_test.txt 1024^2 x 64 spp, 3 renders), summed over all renders; one counter set per run",
       "kernels": {}}
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for c in ("SQ", "FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(os.path.join(src, "pmc_" + c, "*", "*_counter_collection.csv"))
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = kname(r["Kernel_Name"])
        if not k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, a in agg.items():
    d = {"dispatches": len(disp[k])}
    if a.get("GRBM_GUI_ACTIVE"):
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        d.update(valu_wave_insts=a["SQ_INSTS_VALU"], salu_insts=a["SQ_INSTS_SALU"], gpu_cycles=cyc, ms_at_2p4GHz=cyc / 2.4e6,
                 valu_utilization=a["SQ_INSTS_VALU"] / (cyc * 256.0), valu_active_lanes_avg=a["SQ_THREAD_CYCLES_VALU"] / max(a["SQ_INSTS_VALU"], 1))
    if "FETCH_SIZE" in a: d["read_bytes_corrected"] = 2 * 1024 * a["FETCH_SIZE"]
    if "WRITE_SIZE" in a: d["write_bytes"] = 1024 * a["WRITE_SIZE"]
    res["kernels"][k] = d
try:
    txt = open(os.path.join(src, "trace.log")).read()
    res["bench_configs_output_of_the_kernel_trace_run"] = json.loads(txt[txt.index("{"):])
except Exception as e:
    res["bench_configs_output_of_the_kernel_trace_run"] = "unavailable: %s" % e
json.dump(res, open(os.path.join("profiles", "%s_pmc.json" % tag), "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
