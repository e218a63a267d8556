"""Dev/bench tool (GPU box): BASELINE.json configs[4] -- gradient through the solve, fp32 state, one GPU's share of the
8192-trajectory batch (1024 trajectories = 64 tiles), NN-f s00 with the reference's s1 weights, synthetic sine-wave
protocols (the reference's cell-5 recordings are absent from its tree), loss = mean |i - i_ref| (train-s1.py:329).

python tools/bench_grad.py [--batch 1024] [--nt 100001] [--f64] [--budget-gb 24] [--reps 2]
Prints one JSON line: forward (with checkpoints), backward sweep + record reduction, trajectories/s of fwd+bwd.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--nt", type=int, default=100001)
ap.add_argument("--f64", action="store_true")
ap.add_argument("--budget-gb", type=float, default=24.0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--model", default="s1", choices=["s1", "d2"])
ap.add_argument("--no-weight-grad", action="store_true", help="dL/dp and dL/dy0 only: no record stream, no reduce (timing experiment)")
ap.add_argument("--max-step", type=float, default=0.0, help="ms; 0 = off (the reference's dopri5)")
a = ap.parse_args()

ion = importlib.import_module("neural-ode-ion-channels_amd")
import kat_cases as K  # noqa: E402

dev = torch.device("cuda:0")
B, Nt = a.batch, a.nt
P = ion.protocols
pv = P.sinewave(P.sinewave_scales(0, B), n_samples=Nt, xp=torch, device=dev)
model = ion.capi.MODEL_NNF if a.model == "s1" else ion.capi.MODEL_NND
p0 = K.MODELS[a.model][4]
sdt = torch.float64 if a.f64 else torch.float32
te = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1
i_ref = torch.zeros((B, Nt), dtype=torch.float64, device=dev)
vobs = pv + 86.0
res = []
for rep in range(a.reps + 1):
    w = torch.from_numpy(K.load_weights(a.model).copy()).to(dev).requires_grad_(not a.no_weight_grad)
    params = torch.from_numpy(np.tile(p0, (B, 1))).to(dev).requires_grad_(True)
    y0 = torch.tensor([[0.0, 1.0]], dtype=sdt, device=dev).repeat(B, 1).requires_grad_(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y, status = ion.grad.solve(model, w, params, pv, y0, te, mlp_layers=5, mlp_width=200, prot_t0=0.0, prot_dt=0.1,
                               t_eval_hint=(0.0, 0.1), record_budget_bytes=int(a.budget_gb * (1 << 30)), max_step=a.max_step)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss = ((y[..., 0] * y[..., 1]).double() * vobs - i_ref).abs().mean()
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if rep:
        res.append((t1 - t0, t2 - t1))
    gnorm = float(w.grad.double().norm()) if w.grad is not None else 0.0
    del y, loss
fwd, bwd = float(np.mean([r[0] for r in res])), float(np.mean([r[1] for r in res]))
print(json.dumps({"workload": f"configs[4]: gradient through odeint, NN-f s00, {B} trajectories (1/8 of 8192), "
                              f"{'fp64' if a.f64 else 'fp32'} state, N_t = N_p = {Nt}", "batch": B,
                  "forward_with_checkpoints_s": fwd, "backward_s": bwd, "traj_per_s_fwd_bwd": B / (fwd + bwd),
                  "ok": int((status == 0).sum()), "grad_w_norm": gnorm, "grad_p_norm": float(params.grad.norm()),
                  "record_budget_gb": a.budget_gb, "max_step_ms": a.max_step}))
