"""Fill README.md's number table from profiles/r06_bench_line.json (one source per number).  python tools/fill_readme.py"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_line.json")))
e, r, dec, cpu = d["extra"], d["roofline"], d["decode"], d["cpu_baseline"]
c2 = e.get("config2_T1024_B88") or {}
sr = e["scorer_roofline"]
v = {
    "HEADLINE": f"{d['value']:.0f}", "MS": f"{d['ms_per_step']:.3f}", "REPS": " / ".join(f"{1e3 / x:.0f}" for x in e["headline_ms_per_step_repeats"]),
    "FWDUS": f"{r['us_per_launch']:.1f}", "ACH": f"{r['achieved']:.0f}", "FRAC": f"{r['frac']:.3f}",
    "C2FWD": f"{c2.get('logz_fwd_us', float('nan')):.1f}", "C2FRAC": f"{c2.get('logz_fwd_frac_of_8TBs', float('nan')):.2f}", "C2MS": f"{c2.get('logprob_fwd_bwd_ms', float('nan')):.3f}",
    "DECDEV": f"{dec['segments_per_s_device']:.0f}", "DECMS": f"{dec['ms_device']:.2f}", "DECPK": f"{dec['ms_api_packed_arrays']:.2f}",
    "DECAPI": f"{dec['segments_per_s_api']:.0f}", "DECLIST": f"{dec['ms_api_python_lists']:.1f}",
    "SF": f"{e['interval_score_fwd_ms']:.2f}", "SFR": f"{sr['frac']:.2f}", "SB": f"{e['interval_score_bwd_ms']:.2f}",
    "SF3": f"{e['interval_score_fwd_bf16x3_ms']:.2f}", "SB3": f"{e['interval_score_bwd_bf16x3_ms']:.2f}",
    "S3F": f"{sr['bf16x3']['forward_frac']:.2f}", "S3B": f"{sr['bf16x3']['backward_frac']:.2f}",
    "SEG1": f"{e['segment_T691_P90_N1_scorer_crf_logprob_fwd_bwd_ms_fused']:.2f}", "SEG1U": f"{e['segment_T691_P90_N1_scorer_crf_logprob_fwd_bwd_ms_unfused']:.2f}",
    "SEG4": f"{e['segment_T691_P90_N4_scorer_crf_logprob_fwd_bwd_ms_fused']:.2f}",
    "TR": f"{e['train_step_ms']:.2f}", "TRT": f"{e['train_step_ms_bf16x3_train']:.2f}", "TRA": f"{e['train_step_ms_bf16x3_all']:.2f}",
    "TL1": f"{e['transcribe_loop_T691_P90_F1_segments_per_s_end_to_end']:.0f}", "TL4": f"{e['transcribe_loop_T691_P90_F4_segments_per_s_end_to_end']:.0f}",
    "CPUC": str(cpu["cores"]), "CPU": f"{cpu['value']:.2f}", "CPUH": f"{cpu['product_host_kernels']['value']:.1f}",
}
p = os.path.join(ROOT, "README.md")
s = open(p).read()
tpl = os.path.join(ROOT, "tools", "README.table.tpl")
if "__HEADLINE__" in s:
    a, b = s.index("| what | number |"), s.index("At the model's own shapes")
    open(tpl, "w").write(s[a:b])
t = open(tpl).read()
for k, x in v.items():
    t = t.replace("__" + k + "__", x)
a, b = s.index("| what | number |"), s.index("At the model's own shapes")
open(p, "w").write(s[:a] + t + s[b:])
print("README table filled from", d.get("value"))
