"""Dev tool (GPU box): the N = 200 kernel at 16 against 32 trajectories per tile (tile_waves 4 / 8), s00 sine-wave batches."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kat_cases as K
ion = importlib.import_module("neural-ode-ion-channels_amd")
capi, P = ion.capi, ion.protocols
dev = torch.device("cuda:0")
Nt = int(sys.argv[2]) if len(sys.argv) > 2 else 100001
w = K.load_weights("s1")
packed = torch.from_numpy(capi.mlp_pack(w, 5, 200)).to(dev)
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8192,16384").split(",")]:
    pv = P.sinewave(P.sinewave_scales(0, B), n_samples=Nt, dt=0.1, xp=torch, device=dev)
    params = torch.from_numpy(np.tile(K.P_HH, (B, 1))).to(dev)
    y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    te = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1
    out = {}
    res = {}
    for tw in (4, 8, 4, 8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = capi.dopri5(capi.MODEL_NNF, params, pv, y0, te, mlp_packed=packed, mlp_layers=5, mlp_width=200, prot_t0=0.0, prot_dt=0.1,
                        current=True, tile_waves=tw, t_eval_hint=(0.0, 0.1), t_eval_exact=True, out=out)
        e1.record(); torch.cuda.synchronize()
        out.update({k: r[k] for k in ("y", "i", "status", "stats")})
        nfe = r["stats"][:, 2].double()
        fl = float(nfe.sum()) * 401350
        ms = e0.elapsed_time(e1)
        res.setdefault(tw, []).append(ms)
        chk = (float(r["y"][:, -1].sum()), int((r["status"] == 0).sum()))
        print(B, "tile_waves", tw, r["kernel"][-22:], round(ms, 1), "ms", round(fl / ms / 1e9 / 157.3, 4), "of fp32 peak", chk)
    del pv, out, r
    torch.cuda.empty_cache()
