"""Helpers shared by the -m gpu parity tests: run the same inputs through libionode (HIP) and the oracle."""
import numpy as np
import torch


def run_gpu(ion, dev, model, params, prot_v, y0, t_eval, *, weights=None, L=0, N=0, f32=False, **kw):
    capi = ion.capi
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    B = params.shape[0]
    prot_v = np.atleast_2d(np.asarray(prot_v, dtype=np.float64))
    y0 = np.broadcast_to(np.atleast_2d(np.asarray(y0, dtype=np.float64)), (B, np.atleast_2d(y0).shape[-1]))
    sdt = torch.float32 if f32 else torch.float64
    packed = None
    if weights is not None:
        packed = torch.from_numpy(capi.mlp_pack(weights, L, N)).to(dev)
    kw.setdefault("t_eval_hint", "auto")
    pt = kw.pop("prot_t", None)
    pot = kw.pop("prot_of_traj", None)
    r = capi.dopri5(
        model,
        torch.from_numpy(np.ascontiguousarray(params)).to(dev),
        torch.from_numpy(np.ascontiguousarray(prot_v)).to(dev),
        torch.from_numpy(np.ascontiguousarray(y0)).to(dev).to(sdt).contiguous(),
        torch.from_numpy(np.ascontiguousarray(np.asarray(t_eval, dtype=np.float64))).to(dev),
        mlp_packed=packed, mlp_layers=L, mlp_width=N,
        prot_t=None if pt is None else torch.from_numpy(np.ascontiguousarray(np.asarray(pt, dtype=np.float64))).to(dev),
        prot_of_traj=None if pot is None else torch.from_numpy(np.ascontiguousarray(np.asarray(pot, dtype=np.int32))).to(dev),
        **kw)
    torch.cuda.synchronize()
    return {"y": r["y"].double().cpu().numpy(), "i": None if r["i"] is None else r["i"].cpu().numpy(),
            "status": r["status"].cpu().numpy(), "stats": r["stats"].cpu().numpy()}


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
