"""ctypes binding of libionode.so (include/ionode.h).  torch is used for device memory and streams only.

There is no CPU fallback: if the HIP library is missing or no GPU is visible, the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# IONODE_LIB: dev override to A/B kernel builds (tools/ab_build.sh); the default is the in-tree library
LIB_PATH = os.environ.get("IONODE_LIB") or os.path.join(_HERE, "libionode.so")

MODEL_HH2, MODEL_MARKOV6, MODEL_NNF, MODEL_NND = 0, 1, 2, 3
STATUS_OK, STATUS_DT_UNDERFLOW, STATUS_NONFINITE, STATUS_MAX_STEPS = 0, 1, 2, 3
STATUS_TEXT = {
    0: "ok",
    1: "underflow in dt",                 # torchdiffeq's assertion messages
    2: "non-finite values in state `y`",
    3: "max_num_steps exceeded",
}
ABI_VERSION = 9

EXPORTS = (
    "ionode_abi_version", "ionode_last_error", "ionode_mlp_packed_floats", "ionode_mlp_pack",
    "ionode_launch_geometry", "ionode_kernel_name", "ionode_last_kernel_name", "ionode_lane_wise_from", "ionode_dopri5", "ionode_protocol_at_outputs",
    "ionode_grad_image_floats", "ionode_grad_pack", "ionode_grad_record_floats", "ionode_dopri5_backward",
    "ionode_grad_packet_doubles", "ionode_dopri5_backward_recompute", "ionode_dopri5_backward_sweep",
    "ionode_grad_partial_floats", "ionode_grad_reduce", "ionode_grad_reduce_unit", "ionode_grad_reduce_slabs", "ionode_grad_last_error",
    "ionode_regress_step", "ionode_adam_step", "ionode_image_refresh",
)


class IonodeDesc(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("state_f32", C.c_int32), ("n_state", C.c_int32), ("n_out", C.c_int32),
        ("n_traj", C.c_int32), ("n_prot", C.c_int32), ("prot_n", C.c_int32), ("mlp_layers", C.c_int32),
        ("mlp_width", C.c_int32), ("n_params", C.c_int32), ("max_steps", C.c_int64),
        ("prot_t0", C.c_double), ("prot_dt", C.c_double), ("v_oob", C.c_double),
        ("rtol", C.c_double), ("atol", C.c_double), ("obs_g", C.c_double), ("obs_e", C.c_double),
        ("obs_open_state_only", C.c_int32), ("tile_waves", C.c_int32),
        ("step_log", C.c_void_p), ("step_log_cap", C.c_int64),
        ("t_eval_t0_hint", C.c_double), ("t_eval_dt_hint", C.c_double),
        ("max_total_steps", C.c_int64), ("ckpt", C.c_void_p), ("ckpt_cap", C.c_int32), ("t_eval_exact", C.c_int32),
        ("sse_ref", C.c_void_p), ("sse_out", C.c_void_p), ("max_step", C.c_double), ("v_at_outputs", C.c_void_p),
        ("mlp_image_stride", C.c_int64), ("traj_per_image", C.c_int32), ("launch_order", C.c_void_p),
    ]


class IonodeError(RuntimeError):
    pass


def build(force=False, jobs=8):
    """Compile libionode.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "-s", "clean"])
    subprocess.check_call(["make", "-C", csrc, "-s", f"-j{jobs}"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IonodeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the integrator)")
        L = C.CDLL(LIB_PATH)
        L.ionode_abi_version.restype = C.c_int32
        L.ionode_last_error.restype = C.c_char_p
        L.ionode_kernel_name.restype = C.c_char_p
        L.ionode_kernel_name.argtypes = [C.POINTER(IonodeDesc)]
        L.ionode_mlp_packed_floats.restype = C.c_size_t
        L.ionode_mlp_packed_floats.argtypes = [C.c_int32, C.c_int32]
        L.ionode_mlp_pack.restype = C.c_int
        L.ionode_mlp_pack.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.ionode_launch_geometry.restype = C.c_int
        L.ionode_launch_geometry.argtypes = [C.POINTER(IonodeDesc), C.POINTER(C.c_int32 * 4)]
        L.ionode_dopri5.restype = C.c_int
        L.ionode_dopri5.argtypes = [C.POINTER(IonodeDesc)] + [C.c_void_p] * 12
        L.ionode_grad_last_error.restype = C.c_char_p
        for fn in (L.ionode_grad_image_floats, L.ionode_grad_record_floats, L.ionode_grad_partial_floats):
            fn.restype = C.c_size_t
            fn.argtypes = [C.c_int32, C.c_int32]
        L.ionode_grad_pack.restype = C.c_int
        L.ionode_grad_pack.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.ionode_dopri5_backward.restype = C.c_int
        L.ionode_dopri5_backward.argtypes = [C.POINTER(IonodeDesc), C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 13
        L.ionode_dopri5_backward_recompute.restype = C.c_int
        L.ionode_grad_packet_doubles.restype = C.c_size_t
        L.ionode_grad_packet_doubles.argtypes = []
        L.ionode_dopri5_backward_recompute.argtypes = [C.POINTER(IonodeDesc), C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 11
        L.ionode_dopri5_backward_sweep.restype = C.c_int
        L.ionode_dopri5_backward_sweep.argtypes = [C.POINTER(IonodeDesc), C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 14
        L.ionode_grad_reduce.restype = C.c_int
        L.ionode_grad_reduce.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        L.ionode_grad_reduce_slabs.restype = C.c_int32
        L.ionode_grad_reduce_slabs.argtypes = [C.c_int32, C.c_int32, C.c_int64]
        L.ionode_grad_reduce_unit.restype = C.c_int
        L.ionode_grad_reduce_unit.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        L.ionode_regress_step.restype = C.c_int
        L.ionode_regress_step.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.ionode_adam_step.restype = C.c_int
        L.ionode_adam_step.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                       C.c_void_p, C.c_int32, C.c_void_p]
        L.ionode_image_refresh.restype = C.c_int
        L.ionode_image_refresh.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ionode_last_kernel_name.restype = C.c_char_p
        L.ionode_last_kernel_name.argtypes = []
        L.ionode_lane_wise_from.restype = C.c_int32
        L.ionode_lane_wise_from.argtypes = [C.c_int32, C.c_int32]
        L.ionode_protocol_at_outputs.restype = C.c_int
        L.ionode_protocol_at_outputs.argtypes = [C.POINTER(IonodeDesc)] + [C.c_void_p] * 5
        if L.ionode_abi_version() != ABI_VERSION:
            raise IonodeError("libionode.so ABI version mismatch; rebuild")
        _lib = L
    return _lib


def last_error():
    return lib().ionode_last_error().decode()


def mlp_pack(state_dict_flat, mlp_layers, mlp_width):
    """Flat fp32 state dict (numpy, reference order) -> packed fp32 image (numpy) for upload."""
    w = np.ascontiguousarray(np.asarray(state_dict_flat, dtype=np.float32).reshape(-1))
    N, L = int(mlp_width), int(mlp_layers)
    expect = 2 * N + N + L * (N * N + N) + N + 1
    if w.size != expect:
        raise IonodeError(f"state dict has {w.size} floats, (L={L}, N={N}) needs {expect}")
    out = np.empty(lib().ionode_mlp_packed_floats(L, N), dtype=np.float32)
    rc = lib().ionode_mlp_pack(w.ctypes.data, L, N, out.ctypes.data)
    if rc != 0:
        raise IonodeError(last_error())
    return out


def library_digest():
    """sha256 of the libionode.so this process uses (bench.py / tools/pmc_summary.py: ties counter evidence to a build)."""
    import hashlib
    h = hashlib.sha256()
    with open(LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def make_desc(**kw):
    d = IonodeDesc()
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def launch_geometry(desc):
    out = (C.c_int32 * 4)()
    rc = lib().ionode_launch_geometry(C.byref(desc), C.byref(out))
    if rc != 0:
        raise IonodeError(last_error())
    return {"grid": out[0], "block": out[1], "lds_bytes": out[2], "tile_waves": out[3]}


def kernel_name(desc):
    return lib().ionode_kernel_name(C.byref(desc)).decode()


def _dev_ptr(t, dtype, name, shape=None):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise IonodeError(f"{name}: expected a CUDA/HIP tensor (the integrator has no CPU path)")
    if t.dtype != dtype or not t.is_contiguous():
        raise IonodeError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise IonodeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return C.c_void_p(t.data_ptr())


_order_cache = {}


def _protocol_major(prot_of_traj):
    """Stable argsort of the protocol indices (int32, device), cached per tensor version: the launch order in which the
    trajectories of one protocol are adjacent.  Already-sorted inputs return None (index order)."""
    key = (prot_of_traj.data_ptr(), prot_of_traj._version, int(prot_of_traj.shape[0]), str(prot_of_traj.device))
    hit = _order_cache.get(key)
    if hit is None:
        p = prot_of_traj.to(torch.int64)
        if bool((p[1:] >= p[:-1]).all()):
            hit = (None,)
        else:
            order = torch.argsort(p, stable=True)
            # XCD-aware: workgroups go round-robin over the 8 XCDs (each with its own 4 MB L2), so consecutive wavefronts of the
            # protocol-sorted list would spread every protocol over all eight L2s.  Deal the sorted list out in eight contiguous
            # parts instead -- wavefront i takes wavefront i // 8 of part i % 8 -- and an XCD sees one eighth of the protocols.
            n = int(order.shape[0])
            if n % (64 * 8) == 0:
                order = order.view(8, n // (64 * 8), 64).transpose(0, 1).reshape(-1)
            hit = (order.to(torch.int32).contiguous(),)
        if len(_order_cache) > 8:
            _order_cache.clear()
        _order_cache[key] = hit
    return hit[0]


def dopri5(model, params, prot_v, y0, t_eval, *, mlp_packed=None, mlp_layers=0, mlp_width=0, prot_t=None,
           prot_t0=0.0, prot_dt=1.0, prot_of_traj=None, rtol=1e-7, atol=1e-9, v_oob=-80.0, max_steps=0,
           max_total_steps=0, max_step=0.0, ckpt=None, current=False, obs_g=1.0, obs_e=-86.0, obs_open_state_only=False, tile_waves=0, stats=True,
           step_log=None, t_eval_hint="auto", t_eval_exact=None, sse_ref=None, states=True, out=None, stream=None,
           v_at_outputs="auto", traj_per_image=0, launch_order="auto"):
    """Launch one batched solve.  Every tensor lives on the current HIP device.

    launch_order: None (index order), an int32 device permutation [B] (ionode_desc.launch_order: slot s integrates trajectory
    order[s]; results stay at the trajectories' own indices, bit-identical to index order), or "auto": protocol-major order for
    the one-trajectory-per-lane kernels when several protocols are interleaved (the 64 lanes of a wavefront then interpolate
    one protocol instead of 64: -5 % on 393 216 HH trajectories over 64 protocols).

    params [B, n_params] f64, prot_v [P, Np] f64, y0 [B, D] f32|f64 (selects the state dtype),
    t_eval [Nt] f64.  Returns dict(y [B, Nt, D], i [B, Nt] | None, status [B] i32, stats [B, 4] i64 | None);
    asynchronous on the current stream.
    """
    if not torch.cuda.is_available():
        raise IonodeError("no HIP device visible: ionode has no CPU fallback")
    B, D = y0.shape
    Nt = t_eval.shape[0]
    P, Np = prot_v.shape
    sdt = y0.dtype
    if sdt not in (torch.float32, torch.float64):
        raise IonodeError("y0 must be float32 or float64")
    desc = make_desc(model=model, state_f32=int(sdt == torch.float32), n_state=D, n_out=Nt, n_traj=B, n_prot=P,
                     prot_n=Np, mlp_layers=mlp_layers, mlp_width=mlp_width, n_params=params.shape[1],
                     max_steps=max_steps, max_total_steps=max_total_steps, max_step=max_step, prot_t0=prot_t0, prot_dt=prot_dt, v_oob=v_oob, rtol=rtol, atol=atol,
                     obs_g=obs_g, obs_e=obs_e, obs_open_state_only=int(obs_open_state_only), tile_waves=tile_waves)
    if traj_per_image:  # several weight images: mlp_packed [n_images, floats], trajectory b uses image b // traj_per_image
        if mlp_packed is None or mlp_packed.dim() != 2 or mlp_packed.shape[0] * traj_per_image < B or not mlp_packed.is_contiguous():
            raise IonodeError("traj_per_image needs a contiguous mlp_packed [n_images, floats] with n_images * traj_per_image >= B")
        desc.mlp_image_stride, desc.traj_per_image = int(mlp_packed.shape[1]), int(traj_per_image)
    # output-grid hint (t0, dt): a guess the kernel verifies against t_eval; "auto" derives it from the end points
    # (one tiny device->host read), None disables it (cooperative scan)
    if isinstance(t_eval_hint, str) and t_eval_hint == "auto":
        t_eval_hint = None
        if Nt > 1:
            t0h = t_eval[0]
            dth = (t_eval[Nt - 1] - t0h) / (Nt - 1)
            dev_ = (t_eval - (t0h + torch.arange(Nt, dtype=torch.float64, device=t_eval.device) * dth)).abs().max()
            vals = torch.stack([t0h, dth, dev_]).cpu()  # one small device->host read; pass t_eval_hint=(t0, dt) to avoid it
            if float(vals[1]) > 0 and float(vals[2]) <= 0.5 * float(vals[1]):
                t_eval_hint = (float(vals[0]), float(vals[1]))
                if t_eval_exact is None:
                    t_eval_exact = float(vals[2]) == 0.0
    if t_eval_hint is not None and t_eval_hint[1] > 0:
        desc.t_eval_t0_hint, desc.t_eval_dt_hint = float(t_eval_hint[0]), float(t_eval_hint[1])
        if t_eval_exact is None:
            # exactness is a CLAIM the kernel relies on: verify it here (one tiny device->host read) unless the caller did
            k = torch.arange(Nt, dtype=torch.float64, device=t_eval.device)
            t_eval_exact = bool(torch.equal(t_eval, float(t_eval_hint[0]) + k * float(t_eval_hint[1])))
        desc.t_eval_exact = int(bool(t_eval_exact))
    if isinstance(launch_order, str):
        if launch_order != "auto":
            raise IonodeError("launch_order: None, 'auto' or an int32 device tensor [B]")
        launch_order = None
        # one trajectory per lane (64 per wavefront) from these batch sizes on -- the dispatcher's crossovers (ionode_capi.hip make_plan):
        # there the 64 lanes of a wavefront should read ONE protocol
        lane_from = int(lib().ionode_lane_wise_from(int(model), int(mlp_width or 0)))
        if lane_from > 0 and prot_of_traj is not None and P > 1 and B >= lane_from and not traj_per_image:
            launch_order = _protocol_major(prot_of_traj)
    if launch_order is not None:
        _dev_ptr(launch_order, torch.int32, "launch_order", (B,))
        desc.launch_order = launch_order.data_ptr()
    if ckpt is not None:  # [B, cap, 4 + 8*D] f64 device tensor: accepted-step records for the backward sweep
        _dev_ptr(ckpt, torch.float64, "ckpt", (B, ckpt.shape[1], 4 + 8 * D))
        desc.ckpt = ckpt.data_ptr()
        desc.ckpt_cap = ckpt.shape[1]
    if step_log is not None:  # [cap, 4] f64 device tensor: (t0, dt, ratio, accepted) per attempt of trajectory 0
        _dev_ptr(step_log, torch.float64, "step_log")
        desc.step_log = step_log.data_ptr()
        desc.step_log_cap = step_log.shape[0]
    dev = y0.device
    if out is None:
        out = {}
    y = out.get("y")
    if y is None and states:
        y = torch.empty((B, Nt, D), dtype=sdt, device=dev)
    sse = None
    if sse_ref is not None:  # fused objective: [P, Nt] reference currents -> per-trajectory sum of squared residuals
        _dev_ptr(sse_ref, torch.float64, "sse_ref", (P, Nt))
        sse = out.get("sse")
        if sse is None:
            sse = torch.empty((B,), dtype=torch.float64, device=dev)
        desc.sse_ref, desc.sse_out = sse_ref.data_ptr(), sse.data_ptr()
    elif not states:
        raise IonodeError("states=False needs sse_ref (something must be computed)")
    i_out = out.get("i")
    if current and i_out is None:
        i_out = torch.empty((B, Nt), dtype=torch.float64, device=dev)
    status = out.get("status")
    if status is None:
        status = torch.empty((B,), dtype=torch.int32, device=dev)
    st = out.get("stats")
    if stats and st is None:
        st = torch.empty((B, 4), dtype=torch.int64, device=dev)
    s = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
    # closed-form current / objective epilogue: V(t_k) once per protocol instead of once per trajectory per sample ("auto":
    # when every protocol serves at least four trajectories; a [P, Nt] f64 tensor from an earlier call may be passed back in)
    vtab = None
    if (current or sse_ref is not None) and model in (MODEL_HH2, MODEL_MARKOV6) and v_at_outputs is not None:
        if isinstance(v_at_outputs, str):
            if 4 * P <= B:
                vtab = protocol_at_outputs(desc, prot_v, prot_t, t_eval, stream=s)
        else:
            vtab = v_at_outputs
            _dev_ptr(vtab, torch.float64, "v_at_outputs", (P, Nt))
        if vtab is not None:
            desc.v_at_outputs = vtab.data_ptr()
    rc = lib().ionode_dopri5(
        C.byref(desc),
        _dev_ptr(mlp_packed, torch.float32, "mlp_packed"),
        _dev_ptr(params, torch.float64, "params", (B, params.shape[1])),
        _dev_ptr(prot_v, torch.float64, "prot_v"),
        _dev_ptr(prot_t, torch.float64, "prot_t", (Np,)) if prot_t is not None else None,
        _dev_ptr(prot_of_traj, torch.int32, "prot_of_traj", (B,)) if prot_of_traj is not None else None,
        _dev_ptr(y0, sdt, "y0"),
        _dev_ptr(t_eval, torch.float64, "t_eval"),
        _dev_ptr(y, sdt, "y_out", (B, Nt, D)) if y is not None else None,
        _dev_ptr(i_out, torch.float64, "i_out", (B, Nt)) if i_out is not None else None,
        _dev_ptr(status, torch.int32, "status", (B,)),
        _dev_ptr(st, torch.int64, "stats", (B, 4)) if st is not None else None,
        C.c_void_p(s),
    )
    if rc != 0:
        raise IonodeError(f"ionode_dopri5 failed ({rc}): {last_error()}")
    return {"y": y, "i": i_out, "status": status, "stats": st, "sse": sse, "desc": desc, "v_at_outputs": vtab,
            "kernel": lib().ionode_last_kernel_name().decode()}


def protocol_at_outputs(desc, prot_v, prot_t, t_eval, stream=None):
    """[P, Nt] f64: every protocol's voltage at the output times (ionode_protocol_at_outputs), on prot_v's device."""
    out = torch.empty((desc.n_prot, desc.n_out), dtype=torch.float64, device=prot_v.device)
    s = stream if stream is not None else torch.cuda.current_stream(prot_v.device).cuda_stream
    rc = lib().ionode_protocol_at_outputs(C.byref(desc), _dev_ptr(prot_v, torch.float64, "prot_v"),
                                          _dev_ptr(prot_t, torch.float64, "prot_t") if prot_t is not None else None,
                                          _dev_ptr(t_eval, torch.float64, "t_eval"), out.data_ptr(), C.c_void_p(s))
    if rc != 0:
        raise IonodeError(f"ionode_protocol_at_outputs failed ({rc}): {last_error()}")
    return out
