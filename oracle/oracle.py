"""ctypes front-end of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package never imports this module.  numpy in, numpy out; no torch.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IONODE_ORACLE_LIB: the sanitizer build of tests/test_oracle_sanitizers.py (oracle/Makefile target `asan`)
_LIB = os.environ.get("IONODE_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")

MODEL_HH2, MODEL_MARKOV6, MODEL_NNF, MODEL_NND = 0, 1, 2, 3
STATUS_OK, STATUS_DT_UNDERFLOW, STATUS_NONFINITE, STATUS_MAX_STEPS = 0, 1, 2, 3


class OracleDesc(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("state_f32", C.c_int32), ("n_state", C.c_int32), ("n_out", C.c_int32),
        ("n_traj", C.c_int32), ("n_prot", C.c_int32), ("prot_n", C.c_int32), ("mlp_layers", C.c_int32),
        ("mlp_width", C.c_int32), ("n_params", C.c_int32), ("max_steps", C.c_int64),
        ("prot_t0", C.c_double), ("prot_dt", C.c_double), ("v_oob", C.c_double),
        ("rtol", C.c_double), ("atol", C.c_double), ("max_total_steps", C.c_int64),
        ("max_step", C.c_double),
    ]


def build(force=False):
    """Compile liboracle.so with gcc (building the checker is not using it)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("dopri5_oracle.c", "dopri5_impl.h", "Makefile"))
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.oracle_dopri5_batch.restype = C.c_int
        _lib.oracle_mlp_eval.restype = C.c_float
        _lib.oracle_mlp_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def solve(model, params, prot_v, y0, t_eval, *, weights=None, mlp_layers=0, mlp_width=0, prot_t=None,
          prot_t0=0.0, prot_dt=1.0, prot_of_traj=None, state_f32=False, rtol=1e-7, atol=1e-9,
          v_oob=-80.0, max_steps=0, max_total_steps=0, max_step=0.0, nthreads=0, step_log_cap=0):
    """Solve B trajectories.  Returns dict(y[B,Nt,D] f64, status[B], stats[B,4], step_log).
    max_steps: attempts per output interval (torchdiffeq max_num_steps; 0 = 2**31 - 1); max_total_steps: attempts over
    the whole solve (0 = 1 000 000, the product's default runaway bound; < 0 = unbounded)."""
    max_steps = max_steps if max_steps > 0 else 2**31 - 1
    max_total_steps = max_total_steps if max_total_steps > 0 else (1_000_000 if max_total_steps == 0 else 1 << 62)
    params = np.ascontiguousarray(np.atleast_2d(np.asarray(params, dtype=np.float64)))
    B = params.shape[0]
    prot_v = np.ascontiguousarray(np.atleast_2d(np.asarray(prot_v, dtype=np.float64)))
    P, Np = prot_v.shape
    y0 = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(y0, dtype=np.float64)), (B, np.atleast_2d(y0).shape[-1])))
    D = y0.shape[1]
    t_eval = np.ascontiguousarray(np.asarray(t_eval, dtype=np.float64))
    Nt = t_eval.shape[0]
    if prot_t is not None:
        prot_t = np.ascontiguousarray(np.asarray(prot_t, dtype=np.float64))
        assert prot_t.shape == (Np,)
    if prot_of_traj is not None:
        prot_of_traj = np.ascontiguousarray(np.asarray(prot_of_traj, dtype=np.int32))
        assert prot_of_traj.shape == (B,) and prot_of_traj.min() >= 0 and prot_of_traj.max() < P
    if weights is not None:
        weights = np.ascontiguousarray(np.asarray(weights, dtype=np.float32))
        N, L = mlp_width, mlp_layers
        assert weights.size == 2 * N + N + L * (N * N + N) + N + 1, "weights do not match (L, N)"
    d = OracleDesc(model=model, state_f32=int(state_f32), n_state=D, n_out=Nt, n_traj=B, n_prot=P,
                   prot_n=Np, mlp_layers=mlp_layers, mlp_width=mlp_width, n_params=params.shape[1],
                   max_steps=max_steps, prot_t0=prot_t0, prot_dt=prot_dt, v_oob=v_oob, rtol=rtol, atol=atol,
                   max_total_steps=max_total_steps, max_step=max_step)
    y = np.empty((B, Nt, D), dtype=np.float64)
    status = np.zeros(B, dtype=np.int32)
    stats = np.zeros((B, 4), dtype=np.int64)
    slog = np.zeros((step_log_cap, 4), dtype=np.float64) if step_log_cap else None
    rc = lib().oracle_dopri5_batch(C.byref(d), _p(weights), _p(params), _p(prot_v), _p(prot_t), _p(prot_of_traj),
                                   _p(y0), _p(t_eval), _p(y), _p(status), _p(stats), _p(slog),
                                   C.c_int64(step_log_cap), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError(f"oracle_dopri5_batch failed: {rc}")
    if slog is not None:
        slog = slog[: int(stats[0, 0] + stats[0, 1])]
    return {"y": y, "status": status, "stats": stats, "step_log": slog}


def protocol_v(prot_v, t, *, prot_t=None, prot_t0=0.0, prot_dt=1.0, v_oob=-80.0):
    """V(t) as the solver's RHS sees it (interp1d linear + out-of-range -> v_oob)."""
    prot_v = np.ascontiguousarray(np.asarray(prot_v, dtype=np.float64))
    t = np.ascontiguousarray(np.asarray(t, dtype=np.float64))
    if prot_t is not None:
        prot_t = np.ascontiguousarray(np.asarray(prot_t, dtype=np.float64))
    d = OracleDesc(prot_n=prot_v.shape[0], prot_t0=prot_t0, prot_dt=prot_dt, v_oob=v_oob)
    out = np.empty_like(t)
    inr = np.empty(t.shape, dtype=np.int32)
    lib().oracle_protocol_v(C.byref(d), _p(prot_v), _p(prot_t), _p(t), C.c_int(t.size), _p(out), _p(inr))
    return out, inr.astype(bool)


def mlp_eval(weights, L, N, x0, x1):
    weights = np.ascontiguousarray(np.asarray(weights, dtype=np.float32))
    return float(lib().oracle_mlp_eval(_p(weights), L, N, C.c_float(x0), C.c_float(x1)))


def selfcheck():
    out = (C.c_double * 4)()
    lib().oracle_selfcheck(out)
    return list(out)


def current(y, v, *, g=1.0, e_rev=-86.0, open_state_only=False, state_f32=False):
    """Observation model i = g * a * r * (V - E)  (train-s1.py:328; 6-state: O * (V - E), train-d1.py:299).
    The gate product is formed in the state dtype before meeting the fp64 voltage, as torch does."""
    dt = np.float32 if state_f32 else np.float64
    y = np.asarray(y)
    if open_state_only:
        gate = y[..., -1].astype(dt)
    else:
        gate = y[..., 0].astype(dt) * y[..., 1].astype(dt)
    gate = (dt(g) * gate) if g != 1.0 else gate
    return gate.astype(np.float64) * (np.asarray(v, dtype=np.float64) - e_rev)
