"""Dev tool (GPU box, diagnostic build -DIONODE_STAMPS via IONODE_LIB): phase stamps of one tile of an NN-f solve for any architecture.
    IONODE_LIB=.../variants/stamps/libionode.so python tools/stamp_arch.py --layers 5 --width 100 [--batch 16] [--f32] [--tile-waves 0]"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
ion = importlib.import_module("neural-ode-ion-channels_amd")
capi, protocols = ion.capi, ion.protocols

P_HH = np.array([2.26e-4, 6.99e-2, 3.45e-5, 5.46e-2, 8.73e-2, 8.91e-3, 5.15e-3, 3.16e-2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--width", type=int, default=100)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--nt", type=int, default=20001)
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--tile-waves", type=int, default=0)
    ap.add_argument("--out-stride", type=int, default=1, help="keep every k-th output time (same protocol, same steps)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L, N, B, Nt = a.layers, a.width, a.batch, a.nt
    w = np.random.default_rng(1).normal(0, 0.1, 2 * N + N + L * (N * N + N) + N + 1).astype(np.float32)
    packed = torch.from_numpy(capi.mlp_pack(w, L, N)).to(dev)
    pv = protocols.sinewave(protocols.sinewave_scales(0, B), n_samples=Nt, dt=0.1, xp=torch, device=dev)
    params = torch.from_numpy(np.tile(P_HH, (B, 1))).to(dev)
    y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float32 if a.f32 else torch.float64, device=dev).repeat(B, 1).contiguous()
    te = torch.arange(0, Nt, a.out_stride, dtype=torch.float64, device=dev) * 0.1
    slog = torch.zeros((16, 4), dtype=torch.float64, device=dev)
    r = capi.dopri5(capi.MODEL_NNF, params, pv, y0, te, mlp_packed=packed, mlp_layers=L, mlp_width=N, prot_t0=0.0, prot_dt=0.1,
                    tile_waves=a.tile_waves, t_eval_hint=(0.0, 0.1 * a.out_stride), t_eval_exact=True, step_log=slog)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    r = capi.dopri5(capi.MODEL_NNF, params, pv, y0, te, mlp_packed=packed, mlp_layers=L, mlp_width=N, prot_t0=0.0, prot_dt=0.1,
                    tile_waves=a.tile_waves, t_eval_hint=(0.0, 0.1 * a.out_stride), t_eval_exact=True, step_log=slog)
    ev1.record()
    torch.cuda.synchronize()
    slog = slog / 2   # two launches accumulated
    print("launch %.3f ms, %.3f us per evaluation of the slowest trajectory" % (ev0.elapsed_time(ev1), ev0.elapsed_time(ev1) * 1e3 / int(r["stats"][:, 2].max().item())))
    t = slog.cpu().numpy().reshape(-1)[:16]
    nfe0 = int(r["stats"][:16, 2].max().item())
    names = ["outside-mlp", "layer0", "prologue/barriers", "hidden-mfma", "lrelu+store", "last-layer", "rk-stage/err", "interp+emit",
             "interp-fit", "cursor", "emit-gather"]
    tot = t[:11].sum()
    print(r["kernel"], "tile 0: %d evaluations, %.0f cycles each" % (nfe0, tot / max(nfe0, 1)))
    print({n: (int(v / max(nfe0, 1)), round(v / max(tot, 1), 3)) for n, v in zip(names, t[:11])})


if __name__ == "__main__":
    main()
