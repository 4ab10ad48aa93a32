"""Print the figures profiles/README.md and DESIGN.md section 9 quote, from the files banked under profiles/ (after tools/bank_profiles.sh):
python tools/round_figures.py r5"""
import csv, json, sys
T = sys.argv[1] if len(sys.argv) > 1 else "r5"
d = json.load(open(f"profiles/{T}_bench_line_default_run.json"))
print("default  value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 3), "isolated", d["roofline"].get("frac_isolated"))
l = d.get("fp32_mfma_kernels")
if l: print("fp32mfma value", round(l["value"]), "ms/step", round(l["ms_per_step"], 2), "headline is x", round(l["headline_over_this"], 3), "frac", round(l["roofline"]["frac"], 3), "iso", l["roofline"]["frac_isolated"])
print("f64 recurrence", d.get("float64_recurrence_error"))
oc = d.get("other_configs", {})
print("others  ", {k: (v.get("ms_per_step") or v.get("us_per_hop") or v.get("ms_per_call")) if isinstance(v, dict) else v for k, v in oc.items()})
def stats(f, names):
    rows = {r["Name"]: r for r in csv.DictReader(open(f))}
    print(f, [(n, r["Calls"], round(float(r["AverageNs"]))) for n in names for k, r in rows.items() if n in k])
for t in ("pipelined", "serial", "fp32mfma_pipelined", "fp32mfma_serial"):
    j = json.load(open(f"profiles/{T}_bench_line_under_trace_{t}.json"))
    print(t, "ms/step", round(j["ms_per_step"], 2), "avg_launch_ms", round(j["roofline"]["avg_launch_ms"], 4), "frac", round(j["roofline"]["frac"], 3))
G = ["gru64_epi_kernel<2>", "gru64_epi_kernel<1>", "gru64_scan_kernel", "gru256_clusterx"]
L = ["gru64_l3_kernel<2>", "gru64_l3_kernel<0>", "gru64_l3_kernel<1>"]
stats(f"profiles/{T}_pipelined_kernel_stats.csv", L + G[3:]); stats(f"profiles/{T}_serial_kernel_stats.csv", L + G[3:])
stats(f"profiles/{T}_fp32mfma_pipelined_kernel_stats.csv", G); stats(f"profiles/{T}_fp32mfma_serial_kernel_stats.csv", G)
stats(f"profiles/{T}_offline_48k_nb2_256x10s_kernel_stats.csv", ["dec_seg2_kernel<3", "dec_seg2_kernel<2"])
