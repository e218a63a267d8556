"""Dev/bench tool (GPU box): one iteration of the reference's MLP regression loop (train-s1.py:891-909) at its real size
(132 410 rows, net 2 -> 200 x 5 -> 1): fused forward+backward tile kernel, record reduction, Adam + image refresh.

python tools/bench_regression.py [--rows 132410] [--iters 50]   -> one JSON line (ms per iteration, algorithmic TFLOP/s)
Algorithmic FLOPs per iteration = 3 x 2 x rows x (5 x 200 x 200 + 3 x 200)  (forward, input gradient, weight gradient).
"""
import argparse, importlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=132410)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
reg = importlib.import_module("neural-ode-ion-channels_amd.regression")
import kat_cases as K  # noqa: E402
rng = np.random.default_rng(0)
x = np.stack([rng.uniform(-1.3, 0.7, a.rows), rng.uniform(0.01, 0.99, a.rows)], 1).astype(np.float32)
y = rng.normal(0, 1e-3, a.rows).astype(np.float32)
r = reg.MlpRegression(K.load_weights("s1"), 5, 200, x, y, device="cuda:0")
for _ in range(3):
    r.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    loss = r.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
flops = 3 * 2 * a.rows * (5 * 200 * 200 + 3 * 200)
print(json.dumps({"workload": f"MLP regression step, {a.rows} rows, net 2->200x5->1, fp32", "ms_per_iteration": dt * 1e3,
                  "algorithmic_TFLOPs": flops / dt / 1e12, "frac_of_fp32_mfma_peak": flops / dt / 157.3e12,
                  "loss": float(loss.item())}))
