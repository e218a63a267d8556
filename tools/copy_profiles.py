"""Dev tool: copy the evidence tools/collect_profiles.sh left under gpurun_out/r02_prof/ into profiles/ and print the key numbers."""
import csv
import json
import os
import shutil

O = "gpurun_out/r02_prof"
for f in ["r02_bench.json", "r02_bench_profiled.json", "r02_bench_profiled_legs.json", "r02_kernel_stats.csv", "r02_kernel_stats_legs.csv"]:
    shutil.copy(os.path.join(O, f), "profiles/" + f)
summ = {p: json.load(open(f"{O}/{p}.json")) for p in ["pmc1", "pmc2", "pmc3", "cf1", "cf2", "cf3"]}
json.dump(summ, open("profiles/r02_pmc_summary.json", "w"), indent=1)
b = json.load(open("profiles/r02_bench.json"))
print("headline", b["value"], b["roofline"]["frac"], b["roofline"]["kernel_ms"], "cpu", b["cpu_baseline"]["value"], b["cpu_baseline_python"]["value"])
for k, v in b["roofline_closed_form"].items():
    if isinstance(v, dict):
        print(k, round(v["kernel_ms"], 2), v.get("frac"), v["kernel"], v.get("trajectories_per_s"))
print("grad", b["gradient_config5"]["forward_with_checkpoints_s"], b["gradient_config5"]["backward_s"])
print("regress", b["regression_step"]["ms_per_iteration"], b["regression_step"]["frac"])
print("order", {k: (round(v["ms"], 1), round(v["frac_of_fp32_peak"], 4)) for k, v in b["launch_order_16384"].items() if isinstance(v, dict)})
for name in ("r02_kernel_stats.csv", "r02_kernel_stats_legs.csv"):
    print("--", name)
    for r in csv.DictReader(open("profiles/" + name)):
        if "ionode" in r["Name"]:
            print(r["Name"][:95], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3), round(float(r["MinNs"]) / 1e6, 3), round(float(r["MaxNs"]) / 1e6, 3))
for p, v in summ.items():
    print(p, json.dumps(v)[:420])
