"""Assemble profiles/r02_* from the output of tools/prof_run.sh (gpurun_out/prof_r02): newest file of every rocprofv3 run,
per-kernel means of the PMC passes, and profiles/traffic_latest.json stamped with the hash of the sweep sources."""
import csv, glob, hashlib, json, os, re, shutil, collections, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r02")
DST = os.path.join(ROOT, "profiles")

def newest(pattern):
    fs = glob.glob(pattern)
    return max(fs, key=os.path.getmtime) if fs else None

shutil.copy(newest(SRC + "/kt/runc/*_kernel_stats.csv"), DST + "/r02_bench_kernel_stats.csv")
shutil.copy(SRC + "/grid.md", DST + "/r02_grid_table.md")
shutil.copy(SRC + "/bench.json", DST + "/r02_bench_line.json")
shutil.rmtree(DST + "/r02_pmc", ignore_errors=True); os.makedirs(DST + "/r02_pmc")
agg = {}
for d in sorted(glob.glob(SRC + "/pmc_*")):
    if not os.path.isdir(d): continue
    f = newest(d + "/runc/*_counter_collection.csv")
    name = os.path.basename(d)
    shutil.copy(f, f"{DST}/r02_pmc/{name}.csv")
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "persist_sweep" in kn:
            k = "fwd" if "<0, 0, false>" in kn else ("grad" if "<0, 1, true>" in kn else "other")
            a[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in a.items(): agg[(name,) + k] = (len(v), sum(v) / len(v))
g = lambda n, k, c: agg[(n, k, c)][1]
out = {"fwd_fetch_kb": g("pmc_fwd_FETCH_SIZE", "fwd", "FETCH_SIZE"), "fwd_write_kb": g("pmc_fwd_WRITE_SIZE", "fwd", "WRITE_SIZE"),
       "grad_fetch_kb": g("pmc_bwd_FETCH_SIZE", "grad", "FETCH_SIZE"), "grad_write_kb": g("pmc_bwd_WRITE_SIZE", "grad", "WRITE_SIZE")}
n = "pmc_bwd_TCC_HIT_sum_TCC_MISS_sum"
for k in ("grad", "fwd"):
    out[k + "_l2_hit"] = g(n, k, "TCC_HIT_sum") / (g(n, k, "TCC_HIT_sum") + g(n, k, "TCC_MISS_sum"))
n = "pmc_bwd_TCP_PENDING_STALL_CYCLES_sum_TCP_TCC_READ_REQ_sum"
for k in ("grad", "fwd"): out[k + "_tcp_pending_stall_per_cu_Mcycles"] = g(n, k, "TCP_PENDING_STALL_CYCLES_sum") / 256 / 1e6
n = "pmc_bwd_SQ_WAVE_CYCLES_SQ_WAIT_ANY_SQ_WAIT_INST_ANY_SQ_ACTIVE_INST_ANY"
for k in ("grad", "fwd"):
    out[k + "_wait_any"] = g(n, k, "SQ_WAIT_ANY") / g(n, k, "SQ_WAVE_CYCLES")
    out[k + "_active_inst"] = g(n, k, "SQ_ACTIVE_INST_ANY") / g(n, k, "SQ_WAVE_CYCLES")
traffic = int(out["fwd_fetch_kb"] * 1024 * 2 + out["fwd_write_kb"] * 1024)
h = hashlib.sha256()
for f in ("persist.hip", "common.h"): h.update(open(os.path.join(ROOT, "transkun_amd", "csrc", f), "rb").read())
json.dump({"kernel_source_sha16": h.hexdigest()[:16], "logz_fwd_T1024_B352_bytes": traffic,
           "how": "rocprofv3 --pmc FETCH_SIZE (KB, x2: gfx950 counts 128-byte requests of wide reads as 64) + --pmc WRITE_SIZE (KB), separate "
                  "passes, mean over the dispatches of tools/bench_sweep.py --ops fwd --n 5; profiles/r02_pmc/"},
          open(DST + "/traffic_latest.json", "w"), indent=1)
out["fwd_traffic_bytes"] = traffic; out["fwd_traffic_over_algorithmic"] = traffic / 740358784
rows = [r for r in csv.DictReader(open(DST + "/r02_bench_kernel_stats.csv")) if "semicrf::" in r["Name"] or "rocclr" in r["Name"]]
out["kernels"] = [(r["Name"][:90], int(r["Calls"]), round(float(r["AverageNs"]) / 1e3, 1), round(float(r["MinNs"]) / 1e3, 1), round(float(r["MaxNs"]) / 1e3, 1)) for r in rows]
json.dump(out, sys.stdout, indent=1)
