"""Dev/bench tool (GPU box): effect of the launch order (schedule.lpt_order) on the s00 kernel at several tiles per CU.

python tools/bench_order.py [--batch 16384] [--nt 100001]
Prints one JSON object: kernel ms and fp32-MFMA roofline fraction for the arbitrary order, the pilot (closed-form HH)
order and the order from the previous solve's own counters; and the rank correlation of the pilot with the real counts.
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import kat_cases as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16384)
ap.add_argument("--nt", type=int, default=100001)
ap.add_argument("--no-states", action="store_true", help="fused objective only (no traces written)")
a = ap.parse_args()

ion = importlib.import_module("neural-ode-ion-channels_amd")
P, S = ion.protocols, ion.schedule
dev = torch.device("cuda:0")
B, Nt = a.batch, a.nt
weights, _ = bench.load_weights()
pv = P.sinewave(P.sinewave_scales(0, B), n_samples=Nt, dt=0.1, xp=torch, device=dev)
params = torch.from_numpy(np.tile(K.P_HH, (B, 1))).to(dev)
y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64)
te = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1
FLOP = 401350.0


def run(order):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sol = ion.solve(ion.capi.MODEL_NNF, params, pv, y0, te, weights=weights, mlp_layers=5, mlp_width=200, weights_key="s1",
                    prot_t0=0.0, prot_dt=0.1, t_eval_hint=(0.0, 0.1), launch_order=order)
    e1.record()
    torch.cuda.synchronize()
    nfe = sol.to_original(sol.stats[:, 2]).double()
    ms = e0.elapsed_time(e1)
    ok = int((sol.status == 0).sum())
    del sol
    return ms, nfe, ok


res = {"B": B, "Nt": Nt}
run(None)  # warm-up (weights image, allocator)
ms0, nfe, ok = run(None)
res["arbitrary"] = {"ms": ms0, "frac": float(nfe.sum()) * FLOP / (ms0 * 1e-3) / 157.3e12, "ok": ok}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
pc = S.pilot_cost(params, pv, 0.0, float(te[-1]), prot_t0=0.0, prot_dt=0.1)
e1.record()
torch.cuda.synchronize()
res["pilot_ms"] = e0.elapsed_time(e1)
ra, rb = torch.argsort(torch.argsort(pc)).double(), torch.argsort(torch.argsort(nfe)).double()
res["pilot_rank_corr"] = float(torch.corrcoef(torch.stack([ra, rb]))[0, 1])
ms1, nfe1, ok = run(S.lpt_order(pc))
res["pilot_order"] = {"ms": ms1, "frac": float(nfe1.sum()) * FLOP / (ms1 * 1e-3) / 157.3e12, "ok": ok, "same_nfe": bool(torch.equal(nfe, nfe1))}
ms2, nfe2, ok = run(S.lpt_order(nfe))
res["previous_nfe_order"] = {"ms": ms2, "frac": float(nfe2.sum()) * FLOP / (ms2 * 1e-3) / 157.3e12, "ok": ok, "same_nfe": bool(torch.equal(nfe, nfe2))}
res["mean_nfe"], res["max_nfe"] = float(nfe.mean()), float(nfe.max())
print(json.dumps(res))
