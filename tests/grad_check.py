"""Checker for gradients through the solve (BASELINE config 5).  TEST INFRASTRUCTURE ONLY.

The reference never differentiates through `odeint` (SURVEY.md finding 3: `--adjoint` only switches an import,
train-s1.py:29-32, and every call site runs under torch.no_grad()), so **gradient parity is unpinned**: there is no
reference number to match.  The checker is therefore autograd itself: a torch restatement of the same discretisation
(tableau, FSAL, 4th-order dense output) replaying a GIVEN sequence of accepted steps (t0, dt) --
the oracle's step log, which the HIP forward reproduces bit for bit -- in fp64, with the RHS of train-s1.py:231-247 /
train-d2.py:247-272 written with torch ops.  Step sizes are constants of the replay (discretise-then-optimise with the
controller frozen), exactly what the HIP backward sweep differentiates.

`replay()` returns y at the requested times with an autograd graph to (flat weights, rate parameters, y0).
`manual_adjoint()` is the backward sweep the HIP kernel implements, written out by hand in numpy-style torch (only the
MLP's vector-Jacobian product uses autograd); tests check it against autograd-through-replay, so the kernel's algebra is
validated on the CPU before any GPU run.
"""
import numpy as np
import torch

ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
CMID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
        187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]

MODEL_HH2, MODEL_NNF, MODEL_NND = 0, 2, 3


def split_flat(flat, L, N):
    """flat state dict (reference order) -> [(W, b)] for Linear(2,N), L x Linear(N,N), Linear(N,1); views, differentiable."""
    out, off = [], 0
    for (o, i) in [(N, 2)] + [(N, N)] * L + [(1, N)]:
        W = flat[off:off + o * i].reshape(o, i); off += o * i
        b = flat[off:off + o]; off += o
        out.append((W, b))
    assert off == flat.numel()
    return out


def net_eval(layers, x, net_dtype=torch.float32):
    """nn.Sequential(Linear, LeakyReLU(0.01), ..., Linear) on x [..., 2]; evaluated in `net_dtype` (.float() in the reference)."""
    h = x.to(net_dtype)
    for n, (W, b) in enumerate(layers):
        h = h @ W.to(net_dtype).T + b.to(net_dtype)
        if n + 1 < len(layers):
            h = torch.nn.functional.leaky_relu(h, 0.01)
    return h[..., 0]


def protocol_v(t, prot_t, prot_v, v_oob=-80.0):
    """interp1d + the out-of-range rule (train-s1.py:218-237); a constant of the differentiation."""
    t = float(t)
    if t < prot_t[0] or t > prot_t[-1] or t != t:
        return float(v_oob)
    return float(np.interp(t, prot_t, prot_v))


MODEL_MARKOV6 = 1


def rhs(model, layers, p, v, y, net_dtype):
    """func.forward(t, y) with V(t) already looked up.  y [2] fp64, p [8] fp64 tensors (6-state: y [6], p [12])."""
    if model == MODEL_MARKOV6:  # train-d1.py:165-187
        a1, b1 = p[0] * torch.exp(p[1] * v), p[2] * torch.exp(-p[3] * v)
        bh, ah = p[4] * torch.exp(p[5] * v), p[6] * torch.exp(-p[7] * v)
        a2, b2 = p[8] * torch.exp(p[9] * v), p[10] * torch.exp(-p[11] * v)
        c1, c2, i_, ic1, ic2, o = y[0], y[1], y[2], y[3], y[4], y[5]
        return torch.stack([a1 * c2 + ah * ic1 + b2 * o - (b1 + bh + a2) * c1,
                            b1 * c1 + ah * ic2 - (a1 + bh) * c2,
                            a2 * ic1 + bh * o - (b2 + ah) * i_,
                            a1 * ic2 + bh * c1 + b2 * i_ - (b1 + ah + a2) * ic1,
                            b1 * ic1 + bh * c2 - (ah + a1) * ic2,
                            a2 * c1 + ah * i_ - (b2 + bh) * o])
    a, r = y[0], y[1]
    k3 = p[4] * torch.exp(p[5] * v)
    k4 = p[6] * torch.exp(-p[7] * v)
    drdt = -k3 * r + k4 * (1.0 - r)
    dadt = torch.zeros((), dtype=torch.float64)
    if model in (MODEL_HH2, MODEL_NND):
        dadt = p[0] * torch.exp(p[1] * v) * (1.0 - a) - p[2] * torch.exp(-p[3] * v) * a
    if model in (MODEL_NNF, MODEL_NND):
        x = torch.stack([torch.as_tensor(v / 100.0, dtype=torch.float64), a])
        dadt = dadt + net_eval(layers, x, net_dtype).to(torch.float64) / 1000.0
    return torch.stack([dadt, drdt])


def stage_times(t0, dt, f32):
    """The six stage times as the solver forms them in the state dtype (Perturb.PREV at the alpha = 1 stages)."""
    S = np.float32 if f32 else np.float64
    t0s, dts, t1s = S(t0), S(dt), S(t0 + dt)
    out = []
    for a in ALPHA:
        out.append(float(np.nextafter(t1s, S(t1s - S(1)))) if a == 1.0 else float(S(t0s + S(a) * dts)))
    return out


def accepted_steps(step_log):
    sl = np.asarray(step_log)
    return sl[sl[:, 3] == 1.0][:, :2]


def replay(model, flat, L, N, p, y0, prot_t, prot_v, t_eval, steps, *, f32_times=False, net_dtype=torch.float32,
           anchors=None):
    """Differentiable re-run of the accepted steps.  flat [n] fp32/fp64, p [8] fp64, y0 [2] fp64 (leaf tensors or not).
    Returns y [Nt, 2] fp64 (outputs beyond the last step's end are not produced: the caller passes complete logs).

    anchors [n_steps, 2] (optional): the state the ACTUAL forward solve had at the end of each accepted step.  The
    replay's value is moved onto it after every step (`y + (anchor - y).detach()`: value replaced, derivative kept), so
    the Jacobians are evaluated along the actual trajectory -- which is what the HIP sweep does, since it reads y and
    k1..k7 from the forward launch's checkpoints.  Needed in fp32 state only: there the forward's own rounding noise
    (rtol 1e-7 ~ fp32 epsilon, SURVEY.md finding 4) can move a trajectory by 5e-5 from the fp64 re-run of the same steps,
    and d/dp of a 200 ms hold at -120 mV (exp(-p8 V) ~ 28) turns that into a visibly different gradient."""
    layers = split_flat(flat, L, N) if model in (MODEL_NNF, MODEL_NND) else None
    te = np.asarray(t_eval, dtype=np.float64)
    S = np.float32 if f32_times else np.float64
    y = y0
    f = rhs(model, layers, p, protocol_v(float(S(te[0])), prot_t, prot_v), y, net_dtype)
    outs = [y0]
    oi = 1
    for n, (t0, dt) in enumerate(steps):
        if oi >= te.size:
            break
        t1 = t0 + dt
        ts = stage_times(t0, dt, f32_times)
        dts = float(S(dt))
        k = [f]
        Yi = y
        for i in range(6):
            Yi = y + sum(k[j] * (BETA[i][j] * dts) for j in range(i + 1))
            if i == 5 and anchors is not None:
                Yi = Yi + (torch.as_tensor(anchors[n], dtype=torch.float64) - Yi).detach()
            k.append(rhs(model, layers, p, protocol_v(ts[i], prot_t, prot_v), Yi, net_dtype))
        y1 = Yi
        ymid = y + sum(k[j] * (CMID[j] * dts) for j in range(7))
        F0, F1 = k[0], k[6]
        ca = 2 * dts * (F1 - F0) - 8 * (y1 + y) + 16 * ymid
        cb = dts * (5 * F0 - 3 * F1) + 18 * y + 14 * y1 - 32 * ymid
        cc = dts * (F1 - 4 * F0) - 11 * y - 5 * y1 + 16 * ymid
        cd = dts * F0
        while oi < te.size and te[oi] <= t1:
            x = float(S((te[oi] - t0) / (t1 - t0)))
            outs.append(y + x * cd + x * x * cc + x ** 3 * cb + x ** 4 * ca)
            oi += 1
        y, f = y1, k[6]
    assert oi == te.size, "step log does not cover the output grid"
    return torch.stack(outs)


def manual_adjoint(model, flat, L, N, p, y0, prot_t, prot_v, t_eval, steps, gy, *, f32_times=False,
                   net_dtype=torch.float32):
    """The backward sweep of the HIP kernel (ionode_grad.hpp), by hand: returns (dL/dflat, dL/dp, dL/dy0) for
    L = sum(gy * y_out).  Forward values (y, k1..k7 per accepted step) are recomputed first -- the kernel reads them from
    the forward launch's checkpoints.  Only the RHS vector-Jacobian product uses autograd; everything the kernel does in
    scalar code (interpolant adjoint, stage recursion, FSAL carry) is written out."""
    te = np.asarray(t_eval, dtype=np.float64)
    S = np.float32 if f32_times else np.float64
    flat_d = flat.detach()
    p_d = p.detach()

    def F(v, Y):
        with torch.no_grad():
            layers = split_flat(flat_d, L, N) if model in (MODEL_NNF, MODEL_NND) else None
            return rhs(model, layers, p_d, v, Y, net_dtype)

    # ---- forward: what the checkpoints hold ----
    recs = []
    y = y0.detach().clone()
    f = F(protocol_v(float(S(te[0])), prot_t, prot_v), y)
    oi = 1
    for (t0, dt) in steps:
        if oi >= te.size:
            break
        t1 = t0 + dt
        ts = stage_times(t0, dt, f32_times)
        dts = float(S(dt))
        k = [f]
        Yi = y
        for i in range(6):
            Yi = y + sum(k[j] * (BETA[i][j] * dts) for j in range(i + 1))
            k.append(F(protocol_v(ts[i], prot_t, prot_v), Yi))
        n = 0
        while oi + n < te.size and te[oi + n] <= t1:
            n += 1
        recs.append(dict(t0=t0, dt=dt, oi=oi, n=n, y=y, k=k))
        oi += n
        y, f = Yi, k[6]

    # ---- backward sweep ----
    g_flat = torch.zeros_like(flat_d, dtype=torch.float64)
    g_p = torch.zeros(8, dtype=torch.float64)

    def vjp(v, Y, seed):
        """J_F(Y)^T seed and the parameter gradients of seed . F(v, Y)."""
        nonlocal g_flat, g_p
        Yr = Y.detach().clone().requires_grad_(True)
        fl = flat_d.clone().requires_grad_(True)
        pr = p_d.clone().requires_grad_(True)
        layers = split_flat(fl, L, N) if model in (MODEL_NNF, MODEL_NND) else None
        out = (rhs(model, layers, pr, v, Yr, net_dtype) * seed).sum()
        gY, gf, gp = torch.autograd.grad(out, [Yr, fl, pr], allow_unused=True)
        if gf is not None:
            g_flat += gf.double()
        if gp is not None:
            g_p += gp
        return gY

    lam = torch.zeros(2, dtype=torch.float64)   # adjoint of y at the END of the step being processed
    mu = torch.zeros(2, dtype=torch.float64)    # adjoint of the FSAL derivative f = k7 carried into the next step
    gy = gy.double()
    for rec in reversed(recs):
        t0, dt, y, k = rec["t0"], rec["dt"], rec["y"], rec["k"]
        t1 = t0 + dt
        dts = float(S(dt))
        ts = stage_times(t0, dt, f32_times)
        G = [torch.zeros(2, dtype=torch.float64) for _ in range(5)]
        for q in range(rec["n"]):
            kk = rec["oi"] + q
            x = float(S((te[kk] - t0) / (t1 - t0)))
            xp = 1.0
            for c in range(5):
                G[c] = G[c] + gy[kk] * xp
                xp *= x
        aY0 = G[0] - 8 * G[4] + 18 * G[3] - 11 * G[2]
        aY1 = -8 * G[4] + 14 * G[3] - 5 * G[2] + lam
        aYM = 16 * G[4] - 32 * G[3] + 16 * G[2]
        ak = [CMID[j] * dts * aYM for j in range(7)]
        aY0 = aY0 + aYM
        ak[0] = ak[0] + dts * (-2 * G[4] + 5 * G[3] - 4 * G[2] + G[1])
        ak[6] = ak[6] + dts * (2 * G[4] - 3 * G[3] + G[2]) + mu
        for i in range(5, -1, -1):
            Yi = y + sum(k[j] * (BETA[i][j] * dts) for j in range(i + 1))
            w = vjp(protocol_v(ts[i], prot_t, prot_v), Yi, ak[i + 1])
            if i == 5:
                w = w + aY1
            aY0 = aY0 + w
            for j in range(i + 1):
                ak[j] = ak[j] + (BETA[i][j] * dts) * w
        lam, mu = aY0, ak[0]
    # k1 of the first step is F(t[0], y0)
    lam = lam + vjp(protocol_v(float(S(te[0])), prot_t, prot_v), y0.detach(), mu) + gy[0]
    return g_flat, g_p, lam
