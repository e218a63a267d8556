"""Dev/bench tool (GPU box): the N = 200 small tiles -- ONE trajectory per tile (tile_waves = 16: a lane owns a row) against FOUR
(tile_waves = 2: v_mfma_f32_4x4x1) -- on the reference's own call shape: NN-f s00, 10 s sine wave, 2001 outputs, fp32 state.
python tools/bench_small_tiles.py [--batches 1,4,64,256] [--tiles 16,2] [--f64]"""
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
ap.add_argument("--batches", default="1,4,64,256")
ap.add_argument("--tiles", default="16,2")
ap.add_argument("--f64", action="store_true")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--ref-grid", action="store_true", help="the reference's own output grid: linspace(0, 10000, 100001) in fp32 (not exactly uniform: the GENERAL variants)")
ap.add_argument("--stamps", action="store_true", help="library built with -DIONODE_STAMPS: cycles per phase and evaluation of wavefront 0")
a = ap.parse_args()
ion = importlib.import_module("neural-ode-ion-channels_amd")
import kat_cases as K  # noqa: E402
capi, P = ion.capi, ion.protocols
dev = torch.device("cuda:0")
w = K.load_weights("s1")
packed = torch.from_numpy(capi.mlp_pack(w, 5, 200)).to(dev)
Np = 100001
te = torch.arange(2001, dtype=torch.float64, device=dev) * 5.0
hint, exact = (0.0, 5.0), True
if a.ref_grid:
    te = torch.linspace(0.0, 10000.0, 100001, dtype=torch.float32).to(torch.float64).to(dev)   # train-s1.py:48
    hint, exact = "auto", None
out = {}
for B in [int(x) for x in a.batches.split(",")]:
    pv = P.sinewave(P.sinewave_scales(0, B), n_samples=Np, dt=0.1, xp=torch, device=dev)
    params = torch.from_numpy(np.tile(K.P_HH, (B, 1))).to(dev)
    y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64 if a.f64 else torch.float32, device=dev).repeat(B, 1).contiguous()
    ref = None
    for tw in [int(x) for x in a.tiles.split(",")]:
        ms = []
        for rep in range(a.reps + 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            slog = torch.zeros((16, 4), dtype=torch.float64, device=dev) if a.stamps else None
            r = capi.dopri5(capi.MODEL_NNF, params, pv, y0, te, mlp_packed=packed, mlp_layers=5, mlp_width=200, prot_t0=0.0, prot_dt=0.1,
                            t_eval_hint=hint, t_eval_exact=exact, tile_waves=tw, step_log=slog)
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
        nfe = float(r["stats"][:, 2].max())
        if a.stamps:
            c = slog.cpu().numpy().reshape(-1)[:16]
            n0 = float(r["stats"][0, 2])
            names = {0: "outside-mlp", 1: "layer0+barrier", 2: "layer-prologue", 3: "walk", 4: "store+barrier", 5: "output-layer", 6: "rk-stage/err", 7: "interp+emit", 8: "interp-fit", 9: "cursor", 10: "emit-gather"}
            print("STAMPS cycles per evaluation (wavefront 0 of tile 0):", {names.get(i, i): int(c[i] / n0) for i in range(16) if c[i] > 0}, "total", int(c.sum() / n0), flush=True)
        same = None if ref is None else bool(torch.equal(ref, r["y"]))
        ref = r["y"] if ref is None else ref
        out[f"B{B}_tile{tw}"] = {"ms": min(ms[1:]), "kernel": r["kernel"], "max_nfe": nfe, "us_per_eval": min(ms[1:]) * 1e3 / nfe, "same_bits_as_first": same}
        print(f"B={B} tile_waves={tw}: {min(ms[1:]):.1f} ms, {min(ms[1:]) * 1e3 / nfe:.2f} us per evaluation of the slowest trajectory, {r['kernel']}, same={same}", flush=True)
print(json.dumps(out))
