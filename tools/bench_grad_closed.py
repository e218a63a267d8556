"""Dev/bench tool (GPU box): gradient through the solve for the closed-form models (dL/dp, dL/dy0), B trajectories.
python tools/bench_grad_closed.py [--batch 65536] [--nt 10001] [--model hh|m6]"""
import argparse, importlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kat_cases as K  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--nt", type=int, default=10001)
ap.add_argument("--model", default="hh")
a = ap.parse_args()
ion = importlib.import_module("neural-ode-ion-channels_amd")
dev = torch.device("cuda:0")
B, Nt = a.batch, a.nt
m6 = a.model == "m6"
model = K.MODEL_MARKOV6 if m6 else K.MODEL_HH2
p0 = K.P_M6 if m6 else K.P_HH
rng = np.random.default_rng(0)
pv = ion.protocols.sinewave(ion.protocols.sinewave_scales(0, 64), n_samples=Nt, dt=0.1, xp=torch, device=dev)
pot = (torch.arange(B, device=dev) % 64).to(torch.int32)
te = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1
out = {}
for rep in range(2):
    p = torch.from_numpy(np.tile(p0, (B, 1)) * rng.uniform(0.9, 1.1, (B, p0.size))).to(dev).requires_grad_(True)
    y0 = torch.tensor([[0.0, 1.0] + [0.0] * (4 if m6 else 0)], dtype=torch.float64, device=dev).repeat(B, 1).requires_grad_(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y, st = ion.grad.solve(model, None, p, pv, y0, te, prot_t0=0.0, prot_dt=0.1, prot_of_traj=pot, t_eval_hint=(0.0, 0.1), max_step=10.0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (y[..., -1] ** 2).sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out = {"model": a.model, "B": B, "Nt": Nt, "forward_s": t1 - t0, "backward_s": t2 - t1, "ok": int((st == 0).sum()),
           "gp_norm": float(p.grad.norm())}
print(json.dumps(out))
