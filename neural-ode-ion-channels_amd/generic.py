"""Generic dopri5 in torch for RHS modules the fused kernel does not cover (opt-in: options={'allow_generic': True}).

Same algorithm as the HIP kernels (torchdiffeq 0.2.1's dopri5), with `func.forward(t, y)`
called from Python six times per step -- i.e. the reference's own cost structure.  It exists so that arbitrary
callables (e.g. figure-1.py's GroundTruth_a) still integrate; it is never used for the reference's four RHS
families and is not part of any parity or performance claim.
"""
import torch

_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
_CERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
         -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0]
_CMID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
         187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def _rms(x):
    return x.pow(2).mean().sqrt()


def _prev(t):
    return torch.nextafter(t, t - 1)


@torch.no_grad()
def generic_dopri5(func, y0, t, *, rtol=1e-7, atol=1e-9, max_steps=2**31 - 1):
    dt_ = y0.dtype
    dev = y0.device
    t = t.to(torch.float64)
    beta = [torch.tensor(b, dtype=dt_, device=dev) for b in _BETA]
    cerr = torch.tensor(_CERR, dtype=dt_, device=dev)
    cmid = torch.tensor(_CMID, dtype=dt_, device=dev)
    rtol_t = torch.as_tensor(rtol, dtype=torch.float64, device=dev)
    atol_t = torch.as_tensor(atol, dtype=torch.float64, device=dev)

    def f(tt, yy):
        return func(tt.to(dt_), yy)

    sol = torch.empty((t.numel(),) + tuple(y0.shape), dtype=dt_, device=dev)
    sol[0] = y0
    f0 = f(t[0], y0)
    # _select_initial_step
    t0s = t[0].to(dt_)
    scale = atol_t + y0.abs() * rtol_t
    d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
    h0 = torch.tensor(1e-6, dtype=dt_, device=dev) if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = func(t0s + h0, y0 + h0 * f0)
    d2 = _rms((f1 - f0) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = torch.max(torch.tensor(1e-6, dtype=dt_, device=dev), h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1.0 / 5.0)
    dt = torch.min(100 * h0, h1).to(torch.float64)

    y, fy, t0, t1 = y0, f0, t[0], t[0]
    coeff = [y0] * 5
    for i in range(1, t.numel()):
        n_steps = 0  # torchdiffeq's _advance restarts the max_num_steps counter for every output time
        while t[i] > t1:
            assert n_steps < max_steps, "max_num_steps exceeded"
            ts, tn = t1, t1 + dt
            assert ts + dt > ts, "underflow in dt {}".format(dt.item())
            assert torch.isfinite(y).all(), "non-finite values in state `y`: {}".format(y)
            t0s, dts, t1s = ts.to(dt_), dt.to(dt_), tn.to(dt_)
            k = torch.empty(tuple(fy.shape) + (7,), dtype=dt_, device=dev)
            k[..., 0] = fy
            yi = y
            for s in range(6):
                ti = _prev(t1s) if _ALPHA[s] == 1.0 else t0s + _ALPHA[s] * dts
                yi = y + k[..., : s + 1].matmul(beta[s] * dts).view_as(fy)
                k[..., s + 1] = func(ti, yi)
            y1 = yi
            err = k.matmul(dts * cerr)
            tol = atol_t + rtol_t * torch.max(y.abs(), y1.abs())
            ratio = _rms(err / tol).abs()
            accept = bool(ratio <= 1)
            if ratio == 0:
                dt_next = dt * 10.0
            else:
                dfactor = 1.0 if ratio < 1 else 0.2
                fac = min(10.0, max(0.9 / float(ratio.to(torch.float64)) ** 0.2, dfactor))
                dt_next = dt * fac
            if accept:
                ymid = y + k.matmul(dts * cmid).view_as(y)
                F0, F1 = k[..., 0], k[..., -1]
                a = 2 * dts * (F1 - F0) - 8 * (y1 + y) + 16 * ymid
                b = dts * (5 * F0 - 3 * F1) + 18 * y + 14 * y1 - 32 * ymid
                c = dts * (F1 - 4 * F0) - 11 * y - 5 * y1 + 16 * ymid
                coeff = [y, dts * F0, c, b, a]
                y, fy, t0, t1 = y1, k[..., -1], ts, tn
            else:
                t0, t1 = ts, ts
            dt = dt_next
            n_steps += 1
        x = ((t[i] - t0) / (t1 - t0)).to(dt_)
        total = coeff[0] + x * coeff[1]
        xp = x
        for cf in coeff[2:]:
            xp = xp * x
            total = total + xp * cf
        sol[i] = total
    return sol
