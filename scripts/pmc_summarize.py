"""Summarises the rocprofv3 PMC passes collected by scripts/pmc_profile.sh per kernel."""
import csv, glob, json, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
def kname(n):
    for k in ("k_trace", "k_shade", "k_generate", "k_resolve", "k_finalize", "k_untile", "k_extend", "k_connect"):
        if k in n: return k
    return None
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float); ndur = collections.defaultdict(int)
for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
    p = f.split(os.sep)[-3]
    seen = set()
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        if not k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Counter_Name"], r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); calls[k][r["Counter_Name"]] += 1
    if p == "sq1":
        for r in csv.DictReader(open(f.replace("counter_collection", "kernel_trace"))):
            k = kname(r["Kernel_Name"])
            if k: dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; ndur[k] += 1
out = {}
for k in agg:
    out[k] = {c: v for c, v in sorted(agg[k].items())}
    out[k]["_dispatches"] = max(calls[k].values())
    out[k]["_us_total_profiled(sq1 pass)"] = dur.get(k, 0.0)
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
for k in ("k_trace", "k_shade", "k_generate", "k_resolve"):
    if k not in out: continue
    a = out[k]; g = lambda n: a.get(n, 0.0)
    print("== %s: %d dispatches, %.1f ms profiled" % (k, a["_dispatches"], a["_us_total_profiled(sq1 pass)"] / 1e3))
    wc = g("SQ_WAVE_CYCLES"); busy = g("SQ_BUSY_CYCLES")
    if wc:
        print("   waves %.3g  wave_cycles(quad) %.3g  active_any %.1f%%  wait_any %.1f%%  wait_inst_any %.1f%%  active_valu %.1f%%" % (
            g("SQ_WAVES"), wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_ACTIVE_INST_VALU") / wc))
        print("   insts: VALU %.3g  VMEM_RD %.3g  VMEM_WR %.3g  SALU %.3g  SMEM %.3g  LDS %.3g  BRANCH %.3g ; VALU per wave %.0f" % (
            g("SQ_INSTS_VALU"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), g("SQ_INSTS_SALU"), g("SQ_INSTS_SMEM"), g("SQ_INSTS_LDS"), g("SQ_INSTS_BRANCH"), g("SQ_INSTS_VALU") / max(g("SQ_WAVES"), 1)))
        if g("SQ_THREAD_CYCLES_VALU"): print("   VALU lane utilisation %.1f%% (thread_cycles_valu / (64 * active_inst_valu*4?)) raw %.3g" % (100 * g("SQ_THREAD_CYCLES_VALU") / max(64 * g("SQ_ACTIVE_INST_VALU") * 4, 1), g("SQ_THREAD_CYCLES_VALU")))
        print("   LDS bank conflict cycles %.3g of idx_active %.3g" % (g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_IDX_ACTIVE")))
    if g("TCC_REQ_sum"):
        print("   L2: req %.3g hit %.3g miss %.3g -> hit rate %.1f%% ; TCP accesses %.3g, TCP->TCC reads %.3g (L1 hit %.1f%%)" % (
            g("TCC_REQ_sum"), g("TCC_HIT_sum"), g("TCC_MISS_sum"), 100 * g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1),
            g("TCP_TOTAL_CACHE_ACCESSES_sum"), g("TCP_TCC_READ_REQ_sum"), 100 * (1 - g("TCP_TCC_READ_REQ_sum") / max(g("TCP_TOTAL_CACHE_ACCESSES_sum"), 1))))
    print("   FETCH_SIZE %.4g KB  WRITE_SIZE %.4g KB (raw counter units, per all dispatches)" % (g("FETCH_SIZE"), g("WRITE_SIZE")))
