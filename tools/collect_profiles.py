"""Assemble profiles/r03_* from the output of tools/prof_run.sh (gpurun_out/prof_r03): the kernel-stats tables, per-kernel means of
every PMC pass (with the kernel durations of the same pass), the derived figures the bench line and DESIGN.md quote, and
profiles/traffic_latest.json stamped with the hash of the sweep sources.  Run here after gpurun merged the files back."""
import collections, csv, glob, hashlib, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
DST = os.path.join(ROOT, "profiles")


def newest(pattern):
    fs = glob.glob(pattern)
    return max(fs, key=os.path.getmtime) if fs else None


def short(kn):
    kn = kn.replace("void ", "").replace("semicrf::", "")
    return kn[:kn.index("(")] if "(" in kn else kn[:70]


for sub, name in (("kt", "bench_kernel_stats"), ("kt_all", "bench_all_kernels_stats"), ("kt_scorer", "scorer_kernel_stats"),
                  ("kt_train", "train_step_kernel_stats"), ("kt_train_bf16x3", "train_step_bf16x3_train_kernel_stats")):
    f = newest(f"{SRC}/{sub}/runc/*_kernel_stats.csv")
    if f:
        shutil.copy(f, f"{DST}/{TAG}_{name}.csv")
for f, name in (("bwd3.txt", "bwd_bf16x3_timing.txt"), ("train_modes.txt", "train_step_modes.txt"), ("grid.md", "grid_table.md"), ("shapes.md", "model_shapes.md"), ("bench.json", "bench_line.json"), ("scorer.txt", "scorer_timing.txt")):
    if os.path.exists(f"{SRC}/{f}"):
        shutil.copy(f"{SRC}/{f}", f"{DST}/{TAG}_{name}")

# ---- PMC passes: one compact table (kernel, counter, launches, mean value, mean duration of those launches in the same pass) ------
rows = []
agg = {}
for d in sorted(glob.glob(SRC + "/pmc_*")):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)
    cc = newest(d + "/runc/*_counter_collection.csv")
    if not cc:
        continue
    dur = collections.defaultdict(list)
    kt = newest(d + "/runc/*_kernel_trace.csv")
    if kt:
        for r in csv.DictReader(open(kt)):
            dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(cc)):
        kn = short(r["Kernel_Name"])
        if any(s in kn for s in ("persist_sweep", "score", "interval")):
            a[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, c), v in sorted(a.items()):
        us = sum(dur[kn]) / len(dur[kn]) / 1e3 if dur.get(kn) else float("nan")
        rows.append((name, kn, c, len(v), sum(v) / len(v), us))
        agg[(name, kn, c)] = (sum(v) / len(v), us)
with open(f"{DST}/{TAG}_pmc_summary.csv", "w") as fo:
    w = csv.writer(fo)
    w.writerow(["pass", "kernel", "counter", "launches", "mean_value", "mean_kernel_us_in_this_pass"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], r[3], f"{r[4]:.1f}", f"{r[5]:.1f}"])


def g(p, k, c):
    for (pp, kk, cc), v in agg.items():
        if pp == p and cc == c and k in kk:
            return v
    return (float("nan"), float("nan"))


out = {}
FWD, GRAD = "persist_sweep_kernel<0, 0, false>", "persist_sweep_kernel<0, 1, true>"
out["fwd_fetch_kb"] = g("pmc_fwd_FETCH_SIZE", FWD, "FETCH_SIZE")[0]
out["fwd_write_kb"] = g("pmc_fwd_WRITE_SIZE", FWD, "WRITE_SIZE")[0]
out["grad_fetch_kb"] = g("pmc_bwd_FETCH_SIZE", GRAD, "FETCH_SIZE")[0]
out["grad_write_kb"] = g("pmc_bwd_WRITE_SIZE", GRAD, "WRITE_SIZE")[0]
n = "pmc_bwd_TCC_HIT_sum_TCC_MISS_sum"
for k, kn in (("grad", GRAD), ("fwd", FWD)):
    h, m = g(n, kn, "TCC_HIT_sum")[0], g(n, kn, "TCC_MISS_sum")[0]
    out[k + "_l2_hit"] = h / (h + m)
n = "pmc_bwd_SQ_WAVE_CYCLES_SQ_WAIT_ANY_SQ_WAIT_INST_ANY_SQ_ACTIVE_INST_ANY"
for k, kn in (("grad", GRAD), ("fwd", FWD)):
    out[k + "_wait_any"] = g(n, kn, "SQ_WAIT_ANY")[0] / g(n, kn, "SQ_WAVE_CYCLES")[0]
# gfx950: FETCH_SIZE counts the 128-byte requests of wide reads as 64 bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section): x2
traffic = int(out["fwd_fetch_kb"] * 1024 * 2 + out["fwd_write_kb"] * 1024)
out["fwd_traffic_bytes"] = traffic
out["fwd_traffic_over_algorithmic"] = traffic / 740358784
out["grad_traffic_bytes"] = int(out["grad_fetch_kb"] * 1024 * 2 + out["grad_write_kb"] * 1024)
out["grad_traffic_over_algorithmic"] = out["grad_traffic_bytes"] / (740358784 * 2 + 4 * 352 * 1024 * 1023 // 2)

# ---- scorer kernels: matrix-pipe utilisation and the clock under load ------------------------------------------------------------
sc = {}
for kn, flop_per_inst in (("interval_score_tiled_kernel<4>", 4096), ("interval_score_tile_kernel<128>", 4096), ("interval_score_tile3_kernel<128>", 32768),
                          ("score_bwd_gemm_kernel<false, 4>", 4096), ("score_bwd_gemm_kernel<true, 4>", 4096),
                          ("score_bwd_gemm3_kernel<false, 4>", 32768), ("score_bwd_gemm3_kernel<true, 4>", 32768)):
    gui, us = g("pmc_scorer_SQ_BUSY_CYCLES_GRBM_GUI_ACTIVE", kn, "GRBM_GUI_ACTIVE")
    insts = g("pmc_scorer_SQ_INSTS_MFMA_SQ_VALU_MFMA_BUSY_CYCLES", kn, "SQ_INSTS_MFMA")[0]
    busy = g("pmc_scorer_SQ_INSTS_MFMA_SQ_VALU_MFMA_BUSY_CYCLES", kn, "SQ_VALU_MFMA_BUSY_CYCLES")[0]
    if gui != gui:
        continue
    cyc_per_xcd = gui / 8                                     # the counter is summed over the 8 XCDs
    W = "pmc_scorer_SQ_WAVE_CYCLES_SQ_WAIT_ANY_SQ_WAIT_INST_ANY_SQ_ACTIVE_INST_ANY"
    sc[kn] = {"kernel_us_in_pmc_pass": round(us, 1), "gfx_cycles_per_xcd": round(cyc_per_xcd),
              "clock_ghz_under_load": round(cyc_per_xcd / us / 1e3, 3),
              "mfma_wave_instructions": round(insts), "executed_gflop": round(insts * flop_per_inst / 1e9, 2),
              "mfma_busy_cycles_per_simd": round(busy / 1024), "mfma_pipe_busy_fraction": round(busy / 1024 / cyc_per_xcd, 3),
              "wait_any_fraction": round(g(W, kn, "SQ_WAIT_ANY")[0] / g(W, kn, "SQ_WAVE_CYCLES")[0], 3),
              "hbm_read_mb": round(g("pmc_scorer_FETCH_SIZE", kn, "FETCH_SIZE")[0] * 2 / 1024, 1),
              "hbm_write_mb": round(g("pmc_scorer_WRITE_SIZE", kn, "WRITE_SIZE")[0] / 1024, 1)}
out["scorer"] = sc

h = hashlib.sha256()
for f in ("persist.hip", "common.h"):
    h.update(open(os.path.join(ROOT, "transkun_amd", "csrc", f), "rb").read())
json.dump({"kernel_source_sha16": h.hexdigest()[:16], "logz_fwd_T1024_B352_bytes": traffic,
           "how": "rocprofv3 --pmc FETCH_SIZE (KB, x2: gfx950 counts 128-byte requests of wide reads as 64) + --pmc WRITE_SIZE (KB), separate "
                  f"passes, mean over the dispatches of tools/bench_sweep.py --ops fwd --n 5; profiles/{TAG}_pmc_summary.csv"},
          open(DST + "/traffic_latest.json", "w"), indent=1)
f = f"{DST}/{TAG}_bench_kernel_stats.csv"
if os.path.exists(f):
    rws = [r for r in csv.DictReader(open(f)) if "semicrf::" in r["Name"]]
    out["headline_kernels"] = [(short(r["Name"]), int(r["Calls"]), round(float(r["AverageNs"]) / 1e3, 1), round(float(r["MinNs"]) / 1e3, 1),
                                round(float(r["MaxNs"]) / 1e3, 1)) for r in rws]
json.dump(out, open(f"{DST}/{TAG}_derived.json", "w"), indent=1)
json.dump(out, sys.stdout, indent=1)
