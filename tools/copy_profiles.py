"""Dev tool: copy the evidence tools/collect_profiles.sh left under gpurun_out/<R>_prof/ into profiles/ and print the key numbers.
python tools/copy_profiles.py [r05]"""
import csv
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r05"
O = f"gpurun_out/{R}_prof"
for f in [f"{R}_bench.json", f"{R}_bench_profiled.json", f"{R}_bench_profiled_legs.json", f"{R}_kernel_stats.csv", f"{R}_kernel_stats_legs.csv",
          f"{R}_kernel_stats_regression.csv", f"{R}_regression.json", f"{R}_kernel_stats_gradient.csv", f"{R}_gradient.json"]:
    if os.path.exists(os.path.join(O, f)):
        shutil.copy(os.path.join(O, f), "profiles/" + f)
names = ["pmc1", "pmc2", "pmc3", "pmc4", "reg1", "reg2", "reg3", "grad1", "grad2"] + [f"cf{i}{c}" for i in (1, 2, 3, 4, 5, 6, 7, 8) for c in "abc"]
summ = {p: json.load(open(f"{O}/{p}.json")) for p in names if os.path.exists(f"{O}/{p}.json")}
summ["libionode_sha256"] = open(f"{O}/libionode.sha256").read().strip()   # the build the counters belong to (bench.py checks it)
summ["legend"] = {"pmc1-4": "headline s00 kernel (bench.py --steps 1)", "cf1": "HH 2-state 393216 x 20001 fp64, index order (launch_order=None)", "cf2": "6-state 65536 x 20001",
                  "cf3": "NN-f 5x10, 65536 (64 per wavefront, per-lane net, one wavefront per SIMD)", "cf4": "NN-f 5x10, 262144 (64 per wavefront, per-lane net)",
                  "cf5": "HH 2-state 393216 x 20001 fp64, default launch order (auto = protocol-major, in-kernel)", "cf6": "6-state 262144 x 20001 (lean variant, two wavefronts per SIMD)",
                  "cf7": "HH 2-state 524288 x 20001 fp64 (two full residency rounds)", "cf8": "NN-f 5x10, 262144, fp32 state",
                  "grad1-2": "gradient share (tools/bench_grad.py --reps 1): forward with checkpoints on the 4-trajectory tile, recompute / walk / reduce kernels of the two-phase sweep, summed over their dispatches",
                  "reg1-3": "MLP regression step (tools/bench_regression.py --iters 2: 5 dispatches of each kernel): tile kernel, reduce, Adam",
                  "units": "FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x2 on gfx950 for wide streaming reads); SQ_* summed over the chip"}
json.dump(summ, open(f"profiles/{R}_pmc_summary.json", "w"), indent=1)
b = json.load(open(f"profiles/{R}_bench.json"))
print("headline", b["value"], b["roofline"]["frac"], b["roofline"]["kernel_ms"], "traffic", b["roofline"].get("traffic"), "cpu", b["cpu_baseline"]["value"])
for k, v in b["roofline_closed_form"].items():
    if isinstance(v, dict):
        print(k, round(v["kernel_ms"], 2), v.get("frac"), v["kernel"], v.get("trajectories_per_s"))
print("grad", b["gradient_config5"]["forward_with_checkpoints_s"], b["gradient_config5"]["backward_s"])
print("regress", b["regression_step"]["ms_per_iteration"], b["regression_step"]["frac"])
print("order", {k: (round(v["ms"], 1), round(v["frac_of_fp32_peak"], 4)) for k, v in b["launch_order_16384"].items() if isinstance(v, dict)})
print("architectures", {k: (round(v["kernel_ms"], 2), round(v["frac"], 3)) for k, v in b.get("other_architectures_4096", {}).items() if isinstance(v, dict)})
print("config3", {k: v for k, v in b["config3_nnd_staircase_16384"].items() if not isinstance(v, (dict, list))})
for name in (f"{R}_kernel_stats.csv", f"{R}_kernel_stats_legs.csv"):
    print("--", name)
    for r in csv.DictReader(open("profiles/" + name)):
        if "ionode" in r["Name"]:
            print(r["Name"][:95], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3), round(float(r["MinNs"]) / 1e6, 3), round(float(r["MaxNs"]) / 1e6, 3))
for p, v in summ.items():
    print(p, json.dumps(v)[:420])
