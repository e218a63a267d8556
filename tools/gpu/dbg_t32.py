import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kat_cases as K
from oracle import oracle
ion = importlib.import_module("neural-ode-ion-channels_amd")
L, N, B = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 200, int(sys.argv[2]) if len(sys.argv) > 2 else 53
rng = np.random.default_rng(100 + L)
n = 2 * N + N + L * (N * N + N) + N + 1
w = (rng.normal(0, 0.08, n)).astype(np.float32)
pv = np.stack([K.activation(v)[1][:1501] for v in (-20, 40)])
te = np.arange(0, 1500, 3.0)
params = np.tile(K.P_HH, (B, 1)) * rng.uniform(0.9, 1.1, (B, 8))
pot = (np.arange(B) % 2).astype(np.int32)
g = ion.solve(K.MODEL_NNF, params, pv, torch.tensor([[0.0, 1.0]], dtype=torch.float64), te, weights=w, mlp_layers=L, mlp_width=N,
              prot_t0=0.0, prot_dt=1.0, prot_of_traj=pot, current=True, tile_waves=8)
o = oracle.solve(K.MODEL_NNF, params, pv, [0.0, 1.0], te, weights=w, mlp_layers=L, mlp_width=N, prot_t0=0.0, prot_dt=1.0, prot_of_traj=pot, nthreads=8)
gs, gy = g.stats.cpu().numpy(), g.y.cpu().numpy()
bad_s = [b for b in range(B) if not np.array_equal(gs[b], o["stats"][b])]
bad_y = [b for b in range(B) if not np.array_equal(gy[b], o["y"][b], equal_nan=True)]
print("kernel", g.kernel)
print("stats differ:", bad_s)
print("y differ:", bad_y)
for b in bad_y[:6]:
    d = np.nonzero((gy[b] != o["y"][b]).any(1))[0]
    print(b, "first differing sample", d[:5], "n", d.size, gy[b][d[0]], o["y"][b][d[0]], "stats", gs[b], o["stats"][b])
