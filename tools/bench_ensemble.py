"""Dev/bench tool (GPU box): the s00 kernel with one weight set per 16-trajectory tile (traj_per_image = 16) against shared weights.
python tools/bench_ensemble.py [--sets 256] [--nt 20001]"""
import argparse, importlib, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kat_cases as K  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--sets", type=int, default=256)
ap.add_argument("--nt", type=int, default=20001)
a = ap.parse_args()
ion = importlib.import_module("neural-ode-ion-channels_amd")
dev = torch.device("cuda:0")
B, Nt = 16 * a.sets, a.nt
w0 = K.load_weights("s1")
rng = np.random.default_rng(0)
ws = np.stack([w0 * (1.0 + 0.01 * rng.normal(size=w0.size)).astype(np.float32) for _ in range(a.sets)])   # perturbed copies
pv = ion.protocols.sinewave(ion.protocols.sinewave_scales(0, 16), n_samples=Nt, dt=0.1, xp=torch, device=dev)
params = np.tile(K.P_HH, (B, 1))
te = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1
y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64)
res = {"sets": a.sets, "B": B, "Nt": Nt}
for name, kw in (("shared", dict(weights=w0, weights_key="ens-shared")), ("per_tile", dict(weights=ws, weights_key="ens-sets", traj_per_image=16))):
    ms = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sol = ion.solve(K.MODEL_NNF, params, pv, y0, te, mlp_layers=5, mlp_width=200, prot_t0=0.0, prot_dt=0.1, t_eval_hint=(0.0, 0.1), **kw)
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    res[name] = {"ms": min(ms), "ok": int((sol.status == 0).sum()), "mean_nfe": float(sol.stats[:, 2].double().mean())}
print(json.dumps(res))
