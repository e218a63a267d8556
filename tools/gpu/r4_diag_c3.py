"""Diagnostic: configs[2] at full size, where do big-batch and 48-trajectory results differ?"""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
ion = importlib.import_module("neural-ode-ion-channels_amd")
import kat_cases as K
P = importlib.import_module("neural-ode-ion-channels_amd.protocols")
pv = P.staircase()
B, Nt = 16384, 150001
te = np.arange(Nt) * 0.1
half = np.tile(K.P_NN_D, (B // 2, 1)) * np.random.default_rng(7).uniform(0.9, 1.1, (B // 2, 8))
params = np.concatenate([half, half])
w = K.load_weights("d2")
y0 = torch.tensor([K.NN_Y0], dtype=torch.float64)
kw = dict(weights=w, mlp_layers=5, mlp_width=200, prot_t0=0.0, prot_dt=0.1)
for rep in range(2):
    big = ion.solve(K.MODEL_NND, params, pv, y0, te, **kw)
    torch.cuda.synchronize()
    print("big kernel", big.kernel, "status ok", bool((big.status == 0).all()), flush=True)
    for tw in (0, 4):
        small = ion.solve(K.MODEL_NND, params[:48], pv, y0, te, tile_waves=tw, **kw)
        torch.cuda.synchronize()
        ne = (big.y[:48] != small.y).any(dim=2)
        print(" small kernel", small.kernel, "equal", bool(torch.equal(big.y[:48], small.y)), "stats equal", bool(torch.equal(big.stats[:48], small.stats)))
        for tr in torch.nonzero(ne.any(dim=1)).flatten().tolist():
            idx = torch.nonzero(ne[tr]).flatten()
            print("  traj", tr, "mismatches", idx.numel(), "first", int(idx[0]), "last", int(idx[-1]),
                  "big", big.y[tr, int(idx[0])].tolist(), "small", small.y[tr, int(idx[0])].tolist(),
                  "stats big", big.stats[tr].tolist(), "small", small.stats[tr].tolist())
        del small
    z = (big.y == 0).all(dim=2)
    zc = z.sum(dim=1)
    bad = torch.nonzero(zc > 1).flatten()
    print(" trajectories with all-zero samples:", bad.numel(), bad[:20].tolist(), zc[bad[:20]].tolist(), flush=True)
    print(" halves equal", bool(torch.equal(big.y[: B // 2], big.y[B // 2:])))
    del big, z, zc
    torch.cuda.empty_cache()
