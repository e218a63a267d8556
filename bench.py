#!/usr/bin/env python3
"""Headline benchmark: trajectories/s of the batched dopri5 + NN-f (arch s00) solve on MI355X.

Workload (BASELINE.json configs[1]): NN-f, architecture s00 (5 x 200 hidden, the reference's trained s1
weights from tests/golden), B = 4096 trajectories per GPU, each on its own synthetic sine-wave protocol
(0.1 ms sampling, 10 s => N_p = N_t = 100 001, "100k steps"), fp64 solver state, rtol 1e-7 / atol 1e-9,
y0 = [0, 1].  A "step" = one solve of the whole batch (one kernel launch) + the fused current-trace
epilogue + the loss reduction; with N > 1 GPUs every rank solves its own B trajectories (weak scaling,
no data-path collective) and the scalar loss is all-reduced over RCCL.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--nt NT] [--no-cpu-baseline]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline`.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X peaks, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP32_TFLOPS = 157.3  # fp32 MFMA (v_mfma_f32_16x16x4_f32) == fp32 vector peak
PEAK_HBM_GBS = 8000.0

P_HH = np.array([1.12592345582957387e-01, 8.26751134920666146e+01, 3.38768033864048357e-02,
                 4.67106147665183542e+01, 8.47769667061995875e+01, 2.04001345352499328e+01,
                 1.02860743916105211e+01, 2.78201179336874098e+01]) * 1e-3  # train-s1.py:139-146 / :211-214
MLP_L, MLP_N = 5, 200  # architectures/s00.py
F_MLP = 2 * (MLP_L * MLP_N * MLP_N + 3 * MLP_N)  # 401 200 FLOP per RHS evaluation
F_RHS_OTHER = 150  # HH rate terms + RK stage combination per evaluation


def load_weights():
    p = os.path.join(ROOT, "tests", "golden", "weights_s1.f32")
    if os.path.exists(p):
        return np.fromfile(p, dtype="<f4"), "reference s1 state dict (tests/golden)"
    rng = np.random.default_rng(0)  # nn.init.normal_(std=0.1), zero bias: train-s1.py:202-205
    n = 2 * MLP_N + MLP_N + MLP_L * (MLP_N * MLP_N + MLP_N) + MLP_N + 1
    w = rng.normal(0, 0.1, n).astype(np.float32)
    return w, "random init N(0, 0.1^2)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU")
    ap.add_argument("--nt", type=int, default=100001, help="protocol samples = output samples")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="trajectories timed on the host (0 = auto)")
    ap.add_argument("--no-python-baseline", action="store_true", help="skip the reference-structured Python leg")
    ap.add_argument("--tile-waves", type=int, default=0)
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the closed-form / gradient / regression measurements appended to the JSON line")
    ap.add_argument("--stamps", action="store_true", help="diagnostic build (-DIONODE_STAMPS): print phase shares")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group even at world size 1 (under torch.distributed.run): runs the RCCL collectives "
                         "of the N > 1 path with one rank on a 1-GPU box")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal of the N > 1 code path on a 1-GPU box: every rank uses cuda:0 (use with gloo)")
    ap.add_argument("--python-prefix-worker", nargs=3, type=int, metavar=("I", "N", "NT"), help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.python_prefix_worker:   # a child of the cpu_baseline_python leg: CPU only, never initialises the GPU
        return _python_prefix_worker(*args.python_prefix_worker)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It has not touched the
        # GPU (no HIP call so far) and never will; it starts N fresh children, one per GPU, and relays rank 0's line.
        raise SystemExit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size is used", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the integrator has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or (args.force_dist and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend)

    ion = importlib.import_module("neural-ode-ion-channels_amd")
    capi, protocols = ion.capi, importlib.import_module("neural-ode-ion-channels_amd.protocols")

    B, Nt = args.batch, args.nt
    weights, wsrc = load_weights()
    packed = torch.from_numpy(capi.mlp_pack(weights, MLP_L, MLP_N)).to(dev)
    first = rank * B  # global trajectory index of this rank's shard
    prot_v = protocols.sinewave(protocols.sinewave_scales(first, B), n_samples=Nt, dt=0.1, xp=torch, device=dev)
    params = torch.from_numpy(np.tile(P_HH, (B, 1))).to(dev)
    y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    t_eval = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1  # == 0.0 + k * 0.1 bit for bit (t_eval_exact)
    i_ref = torch.zeros((B, Nt), dtype=torch.float64, device=dev)  # synthetic "data" current
    out = {}
    if args.stamps:
        slog = torch.zeros((16, 4), dtype=torch.float64, device=dev)
        rs = capi.dopri5(capi.MODEL_NNF, params, prot_v, y0, t_eval, mlp_packed=packed, mlp_layers=MLP_L, mlp_width=MLP_N,
                         prot_t0=0.0, prot_dt=0.1, current=True, tile_waves=args.tile_waves, t_eval_hint=(0.0, 0.1), step_log=slog)
        torch.cuda.synchronize()
        t = slog.cpu().numpy().reshape(-1)[:16]
        nfe0 = int(rs["stats"][:16, 2].max().item())  # RHS evaluations of tile 0 = of its slowest trajectory
        ts = slog.cpu().numpy().reshape(-1)[16:32]
        if ts.sum() > 0:
            print("ASM STAMPS (cycles per layer pass; 0..12 = k-tile steps, 13 = first half of step 0, 14 = barrier, 15 = outside):",
                  [int(x / (nfe0 * MLP_L)) for x in ts], file=sys.stderr)
        print("STAMPS tile 0: %d evaluations, %.0f cycles each" % (nfe0, t[:11].sum() / nfe0), file=sys.stderr)
        names = ["outside-mlp", "layer0", "barriers", "hidden-mfma", "lrelu+store", "last-layer", "rk-stage/err", "interp+emit"]
        names += ["interp-fit", "cursor", "emit-gather"]
        tot = t[:11].sum()
        print("STAMPS (wave 0 of block 0, cycles per evaluation, share):",
              {n: (int(v / nfe0), round(v / tot, 3)) for n, v in zip(names, t[:11])}, file=sys.stderr)

    def allreduce(x, op=None):
        if dist is None:
            return x
        if args.dist_backend == "nccl":
            dist.all_reduce(x) if op is None else dist.all_reduce(x, op=op)
            return x
        y = x.cpu()
        dist.all_reduce(y) if op is None else dist.all_reduce(y, op=op)
        return y.to(x.device)

    def step():
        r = capi.dopri5(capi.MODEL_NNF, params, prot_v, y0, t_eval, mlp_packed=packed, mlp_layers=MLP_L,
                        mlp_width=MLP_N, prot_t0=0.0, prot_dt=0.1, current=True, obs_g=1.0, obs_e=-86.0,
                        tile_waves=args.tile_waves, t_eval_hint=(0.0, 0.1), t_eval_exact=True, out=out)
        out.update({k: r[k] for k in ("y", "i", "status", "stats")})
        part = torch.stack([torch.nn.functional.l1_loss(r["i"], i_ref, reduction="sum"), torch.tensor(float(B * Nt), dtype=torch.float64, device=dev)])
        part = allreduce(part)  # the path's only collective: 16 bytes
        return r, part[0] / part[1]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()  # same stream as the kernel launch (torch's current stream)
        r = capi.dopri5(capi.MODEL_NNF, params, prot_v, y0, t_eval, mlp_packed=packed, mlp_layers=MLP_L,
                        mlp_width=MLP_N, prot_t0=0.0, prot_dt=0.1, current=True, obs_g=1.0, obs_e=-86.0,
                        tile_waves=args.tile_waves, t_eval_hint=(0.0, 0.1), t_eval_exact=True, out=out)
        ev[k][1].record()
        part = torch.stack([torch.nn.functional.l1_loss(r["i"], i_ref, reduction="sum"), torch.tensor(float(B * Nt), dtype=torch.float64, device=dev)])
        part = allreduce(part)
        loss = part[0] / part[1]
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if dist is not None:
        elapsed = allreduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed_own = t1 - t0
    elapsed = float(elapsed.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # what a SCALE record needs to explain itself: every rank's own wall and kernel time (imbalance = device-to-device spread and step
    # counts, not communication) and the cost of the path's one collective, measured by itself after the timed region
    per_rank = None
    if dist is not None:
        mine = torch.tensor([elapsed_own / args.steps * 1e3, kern_ms], dtype=torch.float64, device=dev)
        both = [torch.zeros_like(mine) for _ in range(world)]
        if args.dist_backend == "nccl":
            dist.all_gather(both, mine)
        else:
            both_c = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(both_c, mine.cpu())
            both = both_c
        barrier()
        ta = time.perf_counter()
        for _ in range(20):
            allreduce(torch.zeros(2, dtype=torch.float64, device=dev))
        torch.cuda.synchronize()
        per_rank = {"per_rank_ms": [float(b[0]) for b in both], "per_rank_kernel_ms": [float(b[1]) for b in both],
                    "allreduce_16B_us": (time.perf_counter() - ta) / 20 * 1e6}

    stats = r["stats"].cpu().numpy()
    status = r["status"].cpu().numpy()
    nfe = stats[:, 2].astype(np.float64)
    n_ok = int((status == 0).sum())
    value = world * B * args.steps / elapsed

    # algorithmic work of ONE launch: FLOPs = sum_traj NFE * (F_mlp + ~150);
    # bytes = N_p*8 (protocol) + N_t*D*8 (states) + N_t*8 (current trace) per trajectory + weights once
    flops = float(nfe.sum()) * (F_MLP + F_RHS_OTHER)
    bytes_traj = Nt * 8 + Nt * 2 * 8 + Nt * 8 + 2 * 8 + 64
    bytes_launch = B * bytes_traj + weights.size * 4
    tflops = flops / (kern_ms * 1e-3) / 1e12
    gbs = bytes_launch / (kern_ms * 1e-3) / 1e9

    res = {
        "metric": "trajectories/sec (sinewave protocol, 100k steps)",
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": f"synthetic sinewave protocols; {wsrc}",
        "config": {"workload": "configs[1]: NN-f (arch s00) batched synthetic sinewave trajectories, fp64 state, "
                               "N_p = N_t = %d, rtol 1e-7 atol 1e-9" % Nt,
                   "trajectories_per_gpu": B, "global_batch": world * B, "parallelism": f"traj-shard x{world}",
                   "kernel": r["kernel"], "geometry": capi.launch_geometry(r["desc"]),
                   "mean_nfe": float(nfe.mean()), "max_nfe": float(nfe.max()),
                   "us_per_rhs_eval_slowest_tile": kern_ms * 1e3 / float(nfe.max()),
                   "mean_accepted": float(stats[:, 0].mean()),
                   "mean_rejected": float(stats[:, 1].mean()), "trajectories_ok": n_ok, "loss": float(loss.item())},
        "roofline": {"bound": "mfma", "achieved": tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": tflops / PEAK_FP32_TFLOPS, "traffic": None,
                     "kernel_ms": kern_ms, "flop_per_launch": flops,
                     "note": "s00 is fp32-FMA bound (SURVEY.md finding 5): 401 200 FLOP per RHS evaluation on the "
                             "fp32 MFMA (dense peak = fp32 vector peak 157.3 TFLOP/s)"},
        "roofline_hbm": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": gbs / PEAK_HBM_GBS, "bytes_per_trajectory": bytes_traj},
    }

    # which collective library and how many ranks it reports (a SCALE record can be checked from the line itself)
    res["config"]["dist_backend"] = dist.get_backend() if dist is not None else None
    res["config"]["dist_world_size"] = dist.get_world_size() if dist is not None else 1
    if per_rank is not None:
        res["config"].update(per_rank)
        try:   # BASELINE configs[3] over the ranks: equal-count against cost-balanced candidate shards (never part of `value`)
            res["config"]["objective_sharded"] = sharded_objective_leg(ion, dev, dist, args.dist_backend, rank, world)
        except Exception as e:  # informational leg only
            res["config"]["objective_sharded"] = {"error": repr(e)}
    lib_sha = capi.library_digest()
    res["config"]["libionode_sha256"] = lib_sha[:16]
    pmc, pj_all = pmc_summary(lib_sha)
    if pmc is not None and B == 4096 and Nt == 100001:
        # HBM bytes per launch from the rocprofv3 --pmc passes of this same command (profiles/README.md): FETCH_SIZE is
        # doubled (gfx950 counts 128-B requests as 64 B), WRITE_SIZE is exact; both are reported in KiB.  The counters belong
        # to ONE build of the library: the summary records its digest, and a different library in this process means the
        # byte count is stale -- reported as such, never silently carried over.
        try:
            pj = pj_all
            kname = "void ionode::" + r["kernel"] + "(ionode::KArgs)"
            traffic = (2 * pj["pmc2"][kname]["FETCH_SIZE"] + pj["pmc3"][kname]["WRITE_SIZE"]) * 1024
            stale = pj.get("libionode_sha256") != lib_sha
            for key in ("roofline", "roofline_hbm"):
                res[key]["traffic"] = None if stale else traffic
                res[key]["traffic_stale"] = stale
                res[key]["traffic_source"] = (f"HBM bytes per launch, profiles/{os.path.basename(pmc)} (separate rocprofv3 --pmc "
                                              "passes of this command; FETCH_SIZE x2 per the gfx950 note, WRITE_SIZE exact)"
                                              + ("; NOT reported: the counters were collected with another build of libionode.so "
                                                 f"({str(pj.get('libionode_sha256'))[:16]})" if stale else ""))
            # where the fraction goes (same stale guard): share of chip time the MFMA pipe is busy (ONE denominator: 1024 SIMDs x launch
            # cycles, launch cycles = GRBM_GUI_ACTIVE / 8 XCDs), and useful / executed MFMA FLOP (padding 200 -> 208, lock-step tiles)
            cyc = pj["pmc2"][kname]["GRBM_GUI_ACTIVE"] / 8.0 / max(1, pj["pmc2"][kname].get("dispatches", 1))
            nd1 = max(1, pj["pmc1"][kname].get("dispatches", 1))
            res["roofline"]["mfma_busy"] = None if stale else pj["pmc1"][kname]["SQ_VALU_MFMA_BUSY_CYCLES"] / nd1 / (1024.0 * cyc)
            res["roofline"]["useful_over_executed"] = None if stale else flops / (pj["pmc1"][kname]["SQ_INSTS_MFMA"] / nd1 * 2048.0)
            res["roofline"]["valu_insts_per_mfma"] = None if stale else pj["pmc1"][kname]["SQ_INSTS_VALU"] / pj["pmc1"][kname]["SQ_INSTS_MFMA"] - 1.0
        except (KeyError, ValueError, ZeroDivisionError):
            pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(prot_v, weights, Nt, args.cpu_sample, out)
        if not args.no_python_baseline:
            try:
                res["cpu_baseline_python"] = python_reference_structured_baseline(ion, prot_v, weights, Nt, out)
            except Exception as e:  # informational leg only
                res["cpu_baseline_python"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_extra_legs and B == 4096 and Nt == 100001:
        # the other kernels of the path, measured in the same driver-run record: never part of `value`
        del r, prot_v, i_ref
        out.clear()
        torch.cuda.empty_cache()
        for key, fn in (("gradient_config5", gradient_leg),
                        ("regression_step", regression_leg), ("launch_order_16384", launch_order_leg),
                        ("objective_config4_share", objective_leg), ("config3_nnd_staircase_16384", config3_leg),
                        ("other_architectures_4096", architectures_leg), ("config1_latency", config1_latency_leg),
                        ("roofline_closed_form", closed_form_legs)):   # last: the driver's record keeps the tail of this line
            try:
                res[key] = fn(ion, dev, weights)
            except Exception as e:  # informational legs only
                res[key] = {"error": repr(e)}
            torch.cuda.empty_cache()
        # BASELINE's metric names the HBM roofline; it applies to the lane-wise kernels (SURVEY.md finding 5).  Their figures go INSIDE
        # `roofline` (the part of the line the driver's record keeps whole), each with the roof that actually binds it
        res["roofline"]["hbm_side"] = hbm_side_block(res.get("roofline_closed_form"), pj_all, lib_sha, pmc)
        res["roofline"]["other_legs"] = compact_legs(res)
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


def pmc_summary(lib_sha):
    """The newest profiles/rNN_pmc_summary.json, preferring one collected with the library this process loaded."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")), reverse=True)
    loaded = []
    for f in files:
        try:
            loaded.append((f, json.load(open(f))))
        except ValueError:
            continue
    for f, pj in loaded:
        if pj.get("libionode_sha256") == lib_sha:
            return f, pj
    return loaded[0] if loaded else (None, None)


# bench leg -> (entry of the PMC summary: tools/collect_profiles.sh PART=2 runs the same launch through tools/bench_closed_form.py)
HBM_SIDE = (("hh2_524288", "cf7"), ("hh2_two_full_rounds", "cf5"), ("markov6_262144", "cf6"), ("markov6", "cf2"),
            ("nnf_s03_5x10_262144", "cf4"), ("nnf_s03_5x10", "cf3"), ("nnf_s03_5x10_262144_f32", "cf8"))


def hbm_side_block(legs, pj, lib_sha, pmc_path):
    """Compact per-kernel block for the driver-kept `roofline` dict: HBM fraction measured in THIS run (events on the launch stream) and,
    from the matching rocprofv3 --pmc entry, the vector-issue fraction: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x launch cycles), and
    VALU-busy = SQ_ACTIVE_INST_VALU x 4 / (same denominator) -- null (+ stale flag) when the counters belong to another build."""
    if not isinstance(legs, dict) or "error" in legs:
        return {"error": "closed-form legs did not run"}
    stale = pj is None or pj.get("libionode_sha256") != lib_sha
    out = {"pmc": os.path.basename(pmc_path) if pmc_path else None, "pmc_stale": stale,
           "valu_issue_frac": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)", "peak_GBs": PEAK_HBM_GBS}
    for leg, cf in HBM_SIDE:
        v = legs.get(leg)
        if not isinstance(v, dict):
            continue
        e = {"kernel": v["kernel"].replace("ionode_dopri5_kernel", "k"), "trajectories": v["trajectories"], "kernel_ms": round(v["kernel_ms"], 3),
             "frac": round(v["frac"], 4), "valu_issue_frac": None, "valu_busy": None, "binds": None}
        try:
            if not stale:
                a = next(iter(pj[cf + "a"].values())); b = next(iter(pj[cf + "b"].values()))
                cyc = b["GRBM_GUI_ACTIVE"] / 8.0 / max(1, b.get("dispatches", 1))
                nd = max(1, a.get("dispatches", 1))
                e["valu_issue_frac"] = round(a["SQ_INSTS_VALU"] / nd * 4.0 / (1024.0 * cyc), 4)
                e["valu_busy"] = round(a["SQ_ACTIVE_INST_VALU"] / nd * 4.0 / (1024.0 * cyc), 4)
                e["launch_cycles"] = int(cyc)
                # shader clock the launch averaged (counter pass's cycles over THIS run's duration: same launch, different runs): the
                # fp64-heavy 2-state kernel writing 4 TB/s runs power-limited well below the 2.4 GHz the light launches hold
                e["mean_clock_GHz"] = round(cyc / (v["kernel_ms"] * 1e6), 2)
                # the binding roof: vector issue when the VALU is busy most of the launch, else the store pattern / latency
                e["binds"] = "valu-issue" if e["valu_busy"] >= 0.7 else ("store-pattern+valu" if e["valu_busy"] >= 0.55 else "latency")
        except (KeyError, StopIteration, ZeroDivisionError):
            pass
        out[leg] = e
    return out


def compact_legs(res):
    """One number per extra leg, inside `roofline` (see hbm_side_block): the review's targets can be read from the kept record."""
    g = lambda d, *ks: (None if not isinstance(d, dict) else (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])))
    r3 = lambda x: None if x is None else round(float(x), 4)
    return {"config1_latency_ms_per_call": r3(g(res, "config1_latency", "ms_per_call")),
            "gradient_config5": {k: r3(g(res, "gradient_config5", k)) for k in ("forward_with_checkpoints_s", "backward_s", "frac")},
            "regression_step": {k: r3(g(res, "regression_step", k)) for k in ("ms_per_iteration", "frac")},
            "architectures_frac": {k: r3(g(res, "other_architectures_4096", k, "frac")) for k in ("s09_5x100", "s11_10x100", "s06_5x500")},
            "launch_order_16384_frac": {k: r3(g(res, "launch_order_16384", k, "frac_of_fp32_peak")) for k in ("arbitrary_order", "previous_nfe_order")},
            "config3_nnd_16384_frac": {k: r3(g(res, "config3_nnd_staircase_16384", k, "frac_of_fp32_peak")) for k in ("index_order", "previous_nfe_order")}}


def _timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def closed_form_legs(ion, dev, weights):
    """The HBM-side kernels (SURVEY.md finding 5: the >= 50 %-of-HBM target applies to these, not to s00): 2 s sine-wave
    protocols (64 distinct, L2-resident), 0.1 ms output grid, fp64 state, states only.  Algorithmic bytes = N_t * D * 8 per
    trajectory written + the distinct protocols read once."""
    capi, P = ion.capi, ion.protocols
    Nt, n_prot = 20001, 64
    pv = P.sinewave(P.sinewave_scales(0, n_prot), n_samples=Nt, xp=torch, device=dev)
    te = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1
    rng = np.random.default_rng(0)
    p_m6 = np.array([5.94625498751561316e-02, 1.21417701632850410e+02, 4.76436985414236425e+00, 3.49383233960778904e-03,
                     9.62243079990877703e+01, 2.26404683824047979e+01, 8.00924780462999131e+00, 2.43749808069009823e+01,
                     2.06822607368134157e+02, 3.30791433507312362e+01, 1.26069071928587784e+00, 2.24844970727316245e+01]) * 1e-3
    N10, L10 = 10, 5
    w10 = np.random.default_rng(1).normal(0, 0.1, 2 * N10 + N10 + L10 * (N10 * N10 + N10) + N10 + 1).astype(np.float32)
    legs = {}
    for name, model, p0, y0, B, mlp, f32 in (("hh2", capi.MODEL_HH2, P_HH, [0.0, 1.0], 262144, None, False),
                                        # the library's default for the one-trajectory-per-lane kernels is a protocol-major LAUNCH ORDER
                                        # (ionode_desc.launch_order, in-kernel: inputs and outputs stay where they are): the 64 lanes of a
                                        # wavefront read ONE protocol instead of 64 (the 10 MB of protocols exceed an XCD's 4 MB L2).
                                        # This leg switches it off: trajectory b in launch slot b
                                        ("hh2_two_full_rounds_index_order", capi.MODEL_HH2, P_HH, [0.0, 1.0], 393216, None, False),
                                        # the 6-state model beyond two residency rounds: its two-wavefronts-per-SIMD build
                                        ("markov6_262144", capi.MODEL_MARKOV6, p_m6, [0.0, 1.0, 0, 0, 0, 0], 262144, None, False),
                                        # rounds 2-3: 12 resident wavefronts per CU x 256 CUs x 64 trajectories = 196 608 per "round", 393 216
                                        # two full rounds (the key keeps its name for the reviews' sake).  Since round 4 the lean 2-state kernel
                                        # holds 16 per CU: 262 144 per round -- 393 216 is 1.5 rounds, 524 288 two
                                        ("hh2_two_full_rounds", capi.MODEL_HH2, P_HH, [0.0, 1.0], 393216, None, False),
                                        ("hh2_524288", capi.MODEL_HH2, P_HH, [0.0, 1.0], 524288, None, False),
                                        ("markov6", capi.MODEL_MARKOV6, p_m6, [0.0, 1.0, 0, 0, 0, 0], 65536, None, False),
                                        ("nnf_s03_5x10", capi.MODEL_NNF, P_HH, [0.0, 1.0], 65536, (w10, L10, N10), False),
                                        # from 49 152 trajectories the N <= 16 nets run one trajectory per lane (64 per wavefront; N = 10: the per-lane vector-ALU net)
                                        ("nnf_s03_5x10_262144", capi.MODEL_NNF, P_HH, [0.0, 1.0], 262144, (w10, L10, N10), False),
                                        # the reference's own state dtype (fp32, SURVEY.md finding 4): half the bytes per sample, the
                                        # same instruction stream -- the byte-based fraction halves, the trajectories per second do not
                                        ("hh2_two_full_rounds_f32", capi.MODEL_HH2, P_HH, [0.0, 1.0], 393216, None, True),
                                        ("markov6_f32", capi.MODEL_MARKOV6, p_m6, [0.0, 1.0, 0, 0, 0, 0], 65536, None, True),
                                        ("nnf_s03_5x10_262144_f32", capi.MODEL_NNF, P_HH, [0.0, 1.0], 262144, (w10, L10, N10), True)):
        D = len(y0)
        params = torch.from_numpy(p0[None, :] * rng.uniform(0.8, 1.25, (B, p0.size))).to(dev)
        y0t = torch.tensor([y0], dtype=torch.float32 if f32 else torch.float64, device=dev).repeat(B, 1).contiguous()
        pot = (torch.arange(B, dtype=torch.int32, device=dev) % n_prot).contiguous()
        kw = dict(prot_t0=0.0, prot_dt=0.1, prot_of_traj=pot, t_eval_hint=(0.0, 0.1), t_eval_exact=True)
        if name.endswith("index_order"):
            kw["launch_order"] = None
        if mlp:
            kw.update(mlp_packed=torch.from_numpy(capi.mlp_pack(mlp[0], mlp[1], mlp[2])).to(dev), mlp_layers=mlp[1], mlp_width=mlp[2])
        o = {}
        hold = {}

        def run():
            hold["r"] = capi.dopri5(model, params, pv, y0t, te, out=o, **kw)
            o.update({k: hold["r"][k] for k in ("y", "status", "stats")})
        ms = _timed(run, 2)
        nbytes = B * Nt * D * (4 if f32 else 8) + n_prot * Nt * 8
        st = hold["r"]["stats"].cpu().numpy()
        legs[name] = {"kernel": hold["r"]["kernel"], "trajectories": B, "n_out": Nt, "state": "f32" if f32 else "f64", "kernel_ms": ms,
                      "trajectories_per_s": B / ms * 1e3, "bound": "hbm", "achieved": nbytes / ms / 1e6, "peak": PEAK_HBM_GBS,
                      "unit": "GB/s", "frac": nbytes / ms / 1e6 / PEAK_HBM_GBS, "mean_nfe": float(st[:, 2].mean()),
                      "launch_order": "index" if name.endswith("index_order") else "auto (protocol-major, in-kernel)",
                      "ok": int((hold["r"]["status"] == 0).sum().item())}
        del params, y0t, o, hold
        torch.cuda.empty_cache()
    # BASELINE configs[3]'s inner operation: the fused sum-of-squares objective of HH candidates (fp32 state as train-d0.py:405),
    # nothing but one double per solve written; V(t_k) from the protocol-at-outputs table (ionode_desc.v_at_outputs)
    B = 196608
    params = torch.from_numpy(P_HH[None, :] * rng.uniform(0.8, 1.25, (B, 8))).to(dev)
    y0t = torch.tensor([[0.0, 1.0]], dtype=torch.float32, device=dev).repeat(B, 1).contiguous()
    pot = (torch.arange(B, dtype=torch.int32, device=dev) % n_prot).contiguous()
    ref = torch.zeros((n_prot, Nt), dtype=torch.float64, device=dev)
    hold = {}

    def run_obj():
        hold["r"] = capi.dopri5(capi.MODEL_HH2, params, pv, y0t, te, prot_t0=0.0, prot_dt=0.1, prot_of_traj=pot,
                                t_eval_hint=(0.0, 0.1), t_eval_exact=True, sse_ref=ref, states=False, stats=False)
    ms = _timed(run_obj, 2)
    legs["hh2_fused_objective_f32"] = {"kernel": hold["r"]["kernel"], "trajectories": B, "n_out": Nt, "kernel_ms": ms,
                                       "trajectories_per_s": B / ms * 1e3, "samples_per_s": B * Nt / ms * 1e3,
                                       "ok": int((hold["r"]["status"] == 0).sum().item()),
                                       "note": "ms includes the protocol-at-outputs pre-pass (64 x 20001 lookups)"}
    del params, y0t, hold
    torch.cuda.empty_cache()
    legs["note"] = "binding roof per kernel: roofline.hbm_side; store-pattern ceiling: profiles/r04_hbm_write_probe.md"
    return legs


def architectures_leg(ion, dev, weights):
    """The other MLP widths of architectures/s00-s11.py through their own tile kernels (table-s1.py:132-153 trains them all):
    random-init nets, 4096 trajectories (one tile per compute unit), 2 s sine-wave protocols, fp64 state.  FLOPs = sum over
    trajectories of RHS evaluations x 2 (L N^2 + 3 N): algorithmic, the launch lasts as long as its slowest tile."""
    capi, P = ion.capi, ion.protocols
    Nt, n_prot, B = 20001, 64, 4096
    pv = P.sinewave(P.sinewave_scales(0, n_prot), n_samples=Nt, xp=torch, device=dev)
    te = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1
    params = torch.from_numpy(P_HH[None, :] * np.random.default_rng(0).uniform(0.8, 1.25, (B, 8))).to(dev)
    y0t = torch.tensor([[0.0, 1.0]], dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    pot = (torch.arange(B, dtype=torch.int32, device=dev) % n_prot).contiguous()
    legs = {}
    for name, L, N in (("s09_5x100", 5, 100), ("s11_10x100", 10, 100), ("s06_5x500", 5, 500), ("s00_5x200_random_init", 5, 200)):
        w = np.random.default_rng(1).normal(0, 0.1, 2 * N + N + L * (N * N + N) + N + 1).astype(np.float32)
        packed = torch.from_numpy(capi.mlp_pack(w, L, N)).to(dev)
        o, hold = {}, {}

        def run():
            hold["r"] = capi.dopri5(capi.MODEL_NNF, params, pv, y0t, te, out=o, prot_t0=0.0, prot_dt=0.1, prot_of_traj=pot,
                                    t_eval_hint=(0.0, 0.1), t_eval_exact=True, mlp_packed=packed, mlp_layers=L, mlp_width=N)
            o.update({k: hold["r"][k] for k in ("y", "status", "stats")})
        ms = _timed(run, 2)
        st = hold["r"]["stats"].cpu().numpy()
        flop = float(st[:, 2].sum()) * 2.0 * (L * N * N + 3 * N)
        legs[name] = {"kernel": hold["r"]["kernel"], "kernel_ms": ms, "us_per_rhs_evaluation_of_the_slowest_tile": ms * 1e3 / float(st[:, 2].max()),
                      "mean_over_max_nfe": float(st[:, 2].mean() / st[:, 2].max()), "achieved": flop / ms / 1e9, "unit": "TFLOP/s",
                      "peak": PEAK_FP32_TFLOPS, "frac": flop / ms / 1e9 / PEAK_FP32_TFLOPS, "ok": int((hold["r"]["status"] == 0).sum().item())}
        del packed, o, hold
        torch.cuda.empty_cache()
    return legs


def gradient_leg(ion, dev, weights):
    """BASELINE.json configs[4]: gradient through the solve, one GPU's share (1024 of 8192 trajectories), fp32 state."""
    capi, P = ion.capi, ion.protocols
    B, Nt = 1024, 100001
    pv = P.sinewave(P.sinewave_scales(0, B), n_samples=Nt, xp=torch, device=dev)
    te = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1
    vobs = pv + 86.0
    times = []
    for rep in range(2):
        w = torch.from_numpy(weights.copy()).to(dev).requires_grad_(True)
        params = torch.from_numpy(np.tile(P_HH, (B, 1))).to(dev)
        y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float32, device=dev).repeat(B, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y, status = ion.grad.solve(capi.MODEL_NNF, w, params, pv, y0, te, mlp_layers=MLP_L, mlp_width=MLP_N, prot_t0=0.0,
                                   prot_dt=0.1, t_eval_hint=(0.0, 0.1))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        fwd_kernel = capi.lib().ionode_last_kernel_name().decode()   # the forward that was timed (with checkpoints)
        ((y[..., 0] * y[..., 1]).double() * vobs).abs().mean().backward()   # mean |i - 0| (train-s1.py:329)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        times.append((t1 - t0, t2 - t1))
        gnorm = float(w.grad.double().norm())
        del y
    fwd, bwd = times[-1]
    # algorithmic FLOPs: forward = RHS evaluations x F_MLP; backward = accepted steps x 6 stages x (forward recompute + input-gradient
    # products + weight-gradient contraction = 3 F_MLP); fp32 MFMA peak as for the headline
    st = capi.dopri5(capi.MODEL_NNF, torch.from_numpy(np.tile(P_HH, (B, 1))).to(dev), pv, torch.tensor([[0.0, 1.0]], dtype=torch.float32, device=dev).repeat(B, 1).contiguous(),
                     te, mlp_packed=torch.from_numpy(capi.mlp_pack(weights, MLP_L, MLP_N)).to(dev), mlp_layers=MLP_L, mlp_width=MLP_N,
                     prot_t0=0.0, prot_dt=0.1, t_eval_hint=(0.0, 0.1), t_eval_exact=True)
    stats = st["stats"].cpu().numpy()
    f_fwd = float(stats[:, 2].sum()) * F_MLP
    f_bwd = float(stats[:, 0].sum()) * 6 * 3 * F_MLP
    return {"workload": "configs[4]: dL/dW through odeint, NN-f s00, 1024 trajectories (1/8 of the 8192-trajectory batch), "
                        "fp32 state, sine-wave protocols, N_t = N_p = 100001", "forward_with_checkpoints_s": fwd, "backward_s": bwd,
            "forward_kernel": fwd_kernel, "trajectories_per_s_fwd_bwd": B / (fwd + bwd), "grad_w_norm": gnorm, "ok": int((status == 0).sum().item()),
            "flop_forward": f_fwd, "flop_backward": f_bwd, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac_forward": f_fwd / fwd / 1e12 / PEAK_FP32_TFLOPS, "frac_backward": f_bwd / bwd / 1e12 / PEAK_FP32_TFLOPS,
            "frac": (f_fwd + f_bwd) / (fwd + bwd) / 1e12 / PEAK_FP32_TFLOPS}


def config3_leg(ion, dev, weights):
    """BASELINE.json configs[2]: NN-d (the train-d2 discrepancy model, d2 weights when the fixture is there) on the 15 s
    staircase protocol (150 001 samples at 0.1 ms), 16 384 trajectories with rate parameters x U(0.9, 1.1), fp64 state:
    index order, then the previous solve's counters as the launch order."""
    capi, P, S = ion.capi, ion.protocols, ion.schedule
    B, Nt = 16384, 150001
    wp = os.path.join(ROOT, "tests", "golden", "weights_d2.f32")
    w = np.fromfile(wp, dtype="<f4") if os.path.exists(wp) else weights
    p_nnd = np.concatenate([P_HH[:4], np.array([9.62243079990877703e+01, 2.26404683824047979e+01, 8.00924780462999131e+00,
                                                  2.43749808069009823e+01]) * 1e-3])     # train-d2.py:221-232
    params = torch.from_numpy(np.tile(p_nnd, (B, 1)) * np.random.default_rng(7).uniform(0.9, 1.1, (B, 8))).to(dev)
    pv = torch.from_numpy(P.staircase()).to(dev)
    y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64)
    te = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1

    def run(order):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sol = ion.solve(capi.MODEL_NND, params, pv, y0, te, weights=w, mlp_layers=MLP_L, mlp_width=MLP_N, weights_key="bench-d2",
                        prot_t0=0.0, prot_dt=0.1, t_eval_hint=(0.0, 0.1), launch_order=order)
        e1.record()
        torch.cuda.synchronize()
        nfe = sol.to_original(sol.stats[:, 2]).double()
        return e0.elapsed_time(e1), nfe, int((sol.status == 0).sum())

    def line(ms, nfe, ok):
        return {"ms": ms, "trajectories_per_s": B / (ms * 1e-3), "mean_nfe": float(nfe.mean()), "max_nfe": float(nfe.max()),
                "frac_of_fp32_peak": float(nfe.sum()) * (F_MLP + F_RHS_OTHER) / (ms * 1e-3) / (PEAK_FP32_TFLOPS * 1e12), "ok": ok}

    run(None)
    ms0, nfe0, ok0 = run(None)
    ms1, nfe1, ok1 = run(S.lpt_order(nfe0))
    return {"workload": "NN-d, 16384 staircase trajectories x 150001 samples, fp64 state", "index_order": line(ms0, nfe0, ok0),
            "previous_nfe_order": line(ms1, nfe1, ok1)}


def objective_leg(ion, dev, weights):
    """BASELINE.json configs[3], one GPU's share: 8192 of the 65 536 candidate parameter sets on every Pr3 / Pr4 / Pr5 sweep
    (7 + 16 + 9 protocols at the 0.1 ms data grid = 262 144 solves), fp32 state as train-d0.py:405, sum of squares against a
    (synthetic) recorded current per sweep, fused in the kernel: what one CMA-ES generation costs per GPU.  Two populations:
    the optimiser's own first generation (log-normal, sigma0 = 0.1 around p0: train-d0.py:527-531) and samples of the whole
    prior box p0 x LogUniform(0.1, 10), where about half of the candidates are too stiff for an explicit solver
    and end at the step budget (the reference bounds those with a 600 s alarm, train-d0.py:309-318; here 20 000 attempts)."""
    P = ion.protocols
    obj = importlib.import_module("neural-ode-ion-channels_amd.objective")
    rng = np.random.default_rng(7)
    C = 8192
    p0 = np.array([1.13e-4, 7.45e-2, 3.60e-5, 4.49e-2])                                             # train-d0.py:325-328
    pops = {"cmaes_first_generation": p0 * np.exp(rng.normal(0.0, 0.1, (C, 4))),
            "prior_box": p0 * 10.0 ** rng.uniform(-1, 1, (C, 4))}
    fams = {"pr3": np.stack([P.activation_pr3(v) for v in P.PR3_STEPS]),
            "pr4": np.stack([P.pr4_synthetic(k) for k in range(16)]),
            "pr5": np.stack([P.deactivation_pr5(v) for v in P.PR5_STEPS])}
    out = {"candidates": C, "step_budget": 20000,
           "note": "host wall time per protocol family incl. parameter upload; nothing but one double per solve leaves the kernel"}
    for pname, cand in pops.items():
        res, total_ms, total_solves, total_samples = {}, 0.0, 0, 0
        for name, pv in fams.items():
            S, Np = pv.shape
            te = np.arange(Np) * 0.1
            data = np.zeros((S, Np))
            args = dict(base_params=P_HH, prot_t0=0.0, prot_dt=0.1, max_total_steps=20000, device=dev)
            obj.population_sum_of_squares(cand[:64], pv, data, te, **args)        # warm-up (uploads, allocator)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sse = obj.population_sum_of_squares(cand, pv, data, te, **args)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
            res[name] = {"sweeps": S, "samples_per_sweep": Np, "solves": C * S, "ms": ms, "finite": int(torch.isfinite(sse).sum())}
            total_ms += ms; total_solves += C * S; total_samples += C * S * Np
        res.update({"ms_per_generation_share": total_ms, "candidates_per_s": C / total_ms * 1e3,
                    "solves_per_s": total_solves / total_ms * 1e3, "samples_per_s": total_samples / total_ms * 1e3})
        out[pname] = res
    # the same objective with the NN-f model (configs[3] reads "NN-f param-fit sweep"): s00 weights shared, p5..p8 per candidate;
    # compute-bound by the MLP kernel, so a small population on the Pr5 sweeps
    Cn, pv = 1024, fams["pr5"]
    S, Np = pv.shape
    cand = P_HH[4:8] * np.exp(rng.normal(0.0, 0.1, (Cn, 4)))
    args = dict(base_params=P_HH, free=(4, 5, 6, 7), prot_t0=0.0, prot_dt=0.1, max_total_steps=200000, device=dev,
                model=ion.capi.MODEL_NNF, weights=weights, mlp_layers=MLP_L, mlp_width=MLP_N, weights_key="bench-s00")
    te, data = np.arange(Np) * 0.1, np.zeros((S, Np))
    obj.population_sum_of_squares(cand[:16], pv, data, te, **args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sse = obj.population_sum_of_squares(cand, pv, data, te, **args)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    out["nnf_s00_candidates_pr5"] = {"candidates": Cn, "sweeps": S, "samples_per_sweep": Np, "solves": Cn * S, "ms": ms,
                                     "candidates_per_s": Cn / ms * 1e3, "solves_per_s": Cn * S / ms * 1e3,
                                     "finite": int(torch.isfinite(sse).sum())}
    return out


def sharded_objective_leg(ion, dev, dist, backend, rank, world):
    """BASELINE.json configs[3] under `--gpus N` (every rank calls this): one CMA-ES generation of 1024 x N HH candidates from the
    prior box (step counts differ several-fold between candidates) on the nine Pr5 sweeps, fused sum-of-squares objective,
    candidates sharded over the ranks and the scores all-gathered (objective.population_sum_of_squares).  Twice: contiguous shards of
    equal COUNT, and shards of equal predicted COST (distributed.shard_bounds_by_cost; cost = every candidate's step count on the
    first sweep from a pilot solve, gathered over the ranks: what the previous generation gives an optimiser for free).
    Reports every rank's solve time: max / mean is the imbalance, the gap to the wall time is gather + host overhead."""
    P = ion.protocols
    obj = importlib.import_module("neural-ode-ion-channels_amd.objective")
    batched = importlib.import_module("neural-ode-ion-channels_amd.batched")
    rng = np.random.default_rng(11)
    C = 1024 * world
    p0 = np.array([1.13e-4, 7.45e-2, 3.60e-5, 4.49e-2])
    cand = p0 * 10.0 ** rng.uniform(-0.7, 0.7, (C, 4))
    pv = np.stack([P.deactivation_pr5(v) for v in P.PR5_STEPS])
    S, Np = pv.shape
    te, data = np.arange(Np) * 0.1, np.zeros((S, Np))
    kw = dict(base_params=P_HH, prot_t0=0.0, prot_dt=0.1, max_total_steps=20000, device=dev)

    def gather(x):
        parts = [torch.zeros_like(x) for _ in range(world)]
        if backend == "nccl":
            dist.all_gather(parts, x)
            return torch.cat(parts)
        parts = [torch.zeros_like(x.cpu()) for _ in range(world)]
        dist.all_gather(parts, x.cpu())
        return torch.cat(parts)

    # pilot: this rank's equal-count share of the candidates on sweep 0, two output times -> RHS evaluations per candidate
    lo, hi = rank * (C // world), (rank + 1) * (C // world)
    params = np.tile(P_HH, (hi - lo, 1)); params[:, :4] = cand[lo:hi]
    pilot = batched.solve(ion.capi.MODEL_HH2, params, pv[:1], torch.tensor([[0.0, 1.0]], dtype=torch.float32), np.array([0.0, te[-1]]),
                          prot_t0=0.0, prot_dt=0.1, max_total_steps=20000, device=dev, t_eval_hint=None)
    cost = gather(pilot.stats[:, 2].to(torch.float64)).cpu().numpy()
    out = {"candidates": C, "sweeps": S, "samples_per_sweep": Np}
    for name, c in (("equal_count", None), ("equal_cost", cost)):
        obj.population_sum_of_squares(cand[: 16 * world], pv, data, te, **kw)      # warm-up
        dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
        sse = obj.population_sum_of_squares(cand, pv, data, te, cost=c, **kw)
        torch.cuda.synchronize(); own = (time.perf_counter() - t0) * 1e3
        dist.barrier(); wall = (time.perf_counter() - t0) * 1e3
        ranks = gather(torch.tensor([own], dtype=torch.float64, device=dev)).cpu().numpy()
        out[name] = {"per_rank_ms": [float(x) for x in ranks], "wall_ms": wall, "max_over_mean": float(ranks.max() / ranks.mean()),
                     "finite": int(torch.isfinite(sse).sum())}
    return out


def config1_latency_leg(ion, dev, weights):
    """BASELINE.json configs[0] / the reference's own call shape: ONE trajectory, `odeint(func, y0, t)` through the torchdiffeq shim
    (train-s1.py:319-330), NN-f s00 on the 10 s synthetic sine wave, fp32 state as in the scripts.  Steady-state call (module, weights
    and protocol already device-resident): the serial chain of ~21 k RHS evaluations on the 4-trajectory tile."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_style_modules as M
    from torchdiffeq import odeint
    P = ion.protocols
    func = M.NNf(P_HH)
    M.load_flat_weights(func.net, weights)
    func.eval()
    Nt = 100001
    pv = P.sinewave(P.sinewave_scales(0, 1), n_samples=Nt)[0]
    func.set_fixed_form_voltage_protocol(np.arange(Nt) * 0.1, pv)
    t = torch.linspace(0.0, 10000.0, 2001)
    y0 = torch.tensor([[0.0, 1.0]])
    ms = []
    with torch.no_grad():
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            y = odeint(func, y0, t)
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    return {"workload": "one odeint(func, y0, t) call, NN-f s00, 10 s sine wave, 2001 outputs, fp32 state", "ms_per_call": min(ms[1:]),
            "first_call_ms": ms[0], "kernel": ion.capi.lib().ionode_last_kernel_name().decode(), "finite": bool(torch.isfinite(y).all())}


def launch_order_leg(ion, dev, weights):
    """The s00 kernel at four tiles per compute unit (the batch of BASELINE configs[2]) with the configs[1] protocols: the
    arbitrary trajectory order against schedule.lpt_order() of (a) a closed-form pilot solve and (b) the previous solve's own
    RHS-evaluation counters (what a training loop has).  Same trajectories, bit-identical results, different tiling."""
    capi, P, S = ion.capi, ion.protocols, ion.schedule
    B, Nt = 16384, 100001
    pv = P.sinewave(P.sinewave_scales(0, B), n_samples=Nt, dt=0.1, xp=torch, device=dev)
    params = torch.from_numpy(np.tile(P_HH, (B, 1))).to(dev)
    y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64)
    te = torch.arange(Nt, dtype=torch.float64, device=dev) * 0.1

    def run(order):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sol = ion.solve(capi.MODEL_NNF, params, pv, y0, te, weights=weights, mlp_layers=MLP_L, mlp_width=MLP_N,
                        weights_key="bench-s00", prot_t0=0.0, prot_dt=0.1, t_eval_hint=(0.0, 0.1), launch_order=order)
        e1.record()
        torch.cuda.synchronize()
        nfe = sol.to_original(sol.stats[:, 2]).double()
        ms, ok, y = e0.elapsed_time(e1), int((sol.status == 0).sum()), sol.to_original(sol.y[:, ::5000].contiguous())
        return ms, nfe, ok, y

    def line(ms, nfe, ok):
        return {"ms": ms, "trajectories_per_s": B / (ms * 1e-3), "frac_of_fp32_peak": float(nfe.sum()) * (F_MLP + F_RHS_OTHER) / (ms * 1e-3) / (PEAK_FP32_TFLOPS * 1e12), "ok": ok}

    run(None)
    ms0, nfe0, ok0, ya = run(None)
    pc = S.pilot_cost(params, pv, 0.0, float(te[-1]), prot_t0=0.0, prot_dt=0.1)
    ms1, nfe1, ok1, yb = run(S.lpt_order(pc))
    ms2, nfe2, ok2, yc = run(S.lpt_order(nfe0))
    return {"workload": "NN-f s00, 16384 sine-wave trajectories x 100001 samples, fp64 state (4 tiles per compute unit)",
            "arbitrary_order": line(ms0, nfe0, ok0), "pilot_order": line(ms1, nfe1, ok1), "previous_nfe_order": line(ms2, nfe2, ok2),
            "results_identical": bool(torch.equal(ya, yb) and torch.equal(ya, yc) and torch.equal(nfe0, nfe1) and torch.equal(nfe0, nfe2))}


def regression_leg(ion, dev, weights):
    """one iteration of the reference's MLP regression loop (train-s1.py:891-909) at its real size."""
    import importlib as _il
    reg = _il.import_module("neural-ode-ion-channels_amd.regression")
    rows, iters = 132410, 30
    rng = np.random.default_rng(0)
    x = np.stack([rng.uniform(-1.3, 0.7, rows), rng.uniform(0.01, 0.99, rows)], 1).astype(np.float32)
    y = rng.normal(0, 1e-3, rows).astype(np.float32)
    r = reg.MlpRegression(weights, MLP_L, MLP_N, x, y, device=dev)
    for _ in range(3):
        r.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        r.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    flops = 3 * 2 * rows * (MLP_L * MLP_N * MLP_N + 3 * MLP_N)
    return {"workload": f"MLP regression step, {rows} rows, net 2->200x5->1, fp32 (synthetic rows)", "ms_per_iteration": dt * 1e3,
            "bound": "mfma", "achieved": flops / dt / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac": flops / dt / 1e12 / PEAK_FP32_TFLOPS}


def self_launch(n):
    """Start n ranks of this script (one per GPU, RCCL rendezvous on 127.0.0.1) as child processes and wait for them.
    Rank 0 inherits stdout, so its single JSON line is this command's output; the exit code is non-zero if any rank
    failed.  The parent never initialises the GPU and is never replaced by another program."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = dict(enumerate(procs))
    while live:
        time.sleep(0.2)
        for r, p in list(live.items()):
            c = p.poll()
            if c is None:
                continue
            del live[r]
            if c != 0:
                print(f"bench.py: rank {r} exited with code {c}", file=sys.stderr)
                rc = rc or (c if c > 0 else 1)
                for q in live.values():  # a dead rank would leave the others waiting at the rendezvous / barrier
                    q.terminate()
    return rc


def usable_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(prot_v, weights, Nt, n_sample, gpu_out):
    """The CPU oracle (a port of the reference path, oracle/) on the first trajectories of the same workload,
    all host cores (OpenMP over trajectories).  Also used as a live parity check of the benchmark run."""
    from oracle import oracle
    oracle.build()
    cores = usable_cores()
    n = n_sample or min(prot_v.shape[0], 32 * cores)  # ~10-20 s of host work
    pv = prot_v[:n].cpu().numpy()
    te = np.arange(Nt, dtype=np.float64) * 0.1
    t0 = time.perf_counter()
    o = oracle.solve(oracle.MODEL_NNF, np.tile(P_HH, (n, 1)), pv, [0.0, 1.0], te, weights=weights,
                     mlp_layers=MLP_L, mlp_width=MLP_N, prot_t0=0.0, prot_dt=0.1, nthreads=cores)
    dt = time.perf_counter() - t0
    g = gpu_out["y"][:n].cpu().numpy()
    err = float(np.linalg.norm(g - o["y"]) / np.linalg.norm(o["y"]))
    # BASELINE.md section 2 asks for the port on ONE core as well (the reference process is serial, train-s1.py has no pool)
    n1 = min(n, 12)
    t0 = time.perf_counter()
    oracle.solve(oracle.MODEL_NNF, np.tile(P_HH, (n1, 1)), pv[:n1], [0.0, 1.0], te, weights=weights,
                 mlp_layers=MLP_L, mlp_width=MLP_N, prot_t0=0.0, prot_dt=0.1, nthreads=1)
    dt1 = time.perf_counter() - t0
    return {"value": n / dt, "unit": "trajectories/s", "cores": cores, "kind": "port",
            "sample": f"first {n} trajectories of the same batch, oracle/liboracle.so (C, AVX2 fp32 fmaf chain), "
                      f"{dt:.1f} s wall", "rel_l2_gpu_vs_oracle": err,
            "one_core": {"value": n1 / dt1, "unit": "trajectories/s", "cores": 1,
                         "sample": f"first {n1} trajectories, one thread, {dt1:.1f} s wall"}}


def _ref_style_nnf(weights):
    """nn.Module shaped like the reference's ODEFunc (train-s1.py:186-247): fp32 nn.Sequential + SciPy interp1d on the host per RHS call."""
    import torch.nn as nn
    from scipy.interpolate import interp1d

    class RefStyleNNf(nn.Module):
        def __init__(self):
            super().__init__()
            layers = [nn.Linear(2, MLP_N), nn.LeakyReLU()]
            for _ in range(MLP_L):
                layers += [nn.Linear(MLP_N, MLP_N), nn.LeakyReLU()]
            self.net = nn.Sequential(*layers, nn.Linear(MLP_N, 1))
            off = 0
            with torch.no_grad():
                for m in self.net:
                    if isinstance(m, nn.Linear):
                        n = m.weight.numel()
                        m.weight.copy_(torch.from_numpy(weights[off:off + n].reshape(m.weight.shape))); off += n
                        m.bias.copy_(torch.from_numpy(weights[off:off + m.bias.numel()])); off += m.bias.numel()
            self.p5, self.p6, self.p7, self.p8 = (float(x) for x in P_HH[4:8])

        def set_protocol(self, t, v):
            self._interp = interp1d(t, v)

        def forward(self, t, y):
            a, r = torch.unbind(y, dim=1)
            try:
                v = torch.from_numpy(self._interp([t.detach().numpy()]))
            except ValueError:
                v = torch.tensor([-80])
            k3 = self.p5 * torch.exp(self.p6 * v)
            k4 = self.p7 * torch.exp(-self.p8 * v)
            drdt = -k3 * r + k4 * (1 - r)
            dadt = self.net(torch.stack([(v / 100.0)[0], a[0]]).float()) / 1000.0
            return torch.stack([dadt[0], drdt[0]]).reshape(1, -1)

    return RefStyleNNf().eval()


def _python_prefix_worker(index, n, Nt):
    """One process of the all-cores Python leg (`bench.py --python-prefix-worker i n nt`, started as a plain child process before
    anything touches the GPU): integrate the first n output samples of synthetic trajectory i with the Python stepper, print the time."""
    torch.set_num_threads(1)
    generic = importlib.import_module("neural-ode-ion-channels_amd.generic")
    protocols = importlib.import_module("neural-ode-ion-channels_amd.protocols")
    weights, _ = load_weights()
    pv = protocols.sinewave(protocols.sinewave_scales(index, 1), n_samples=Nt, dt=0.1, xp=torch, device="cpu")[0].numpy()
    f = _ref_style_nnf(weights)
    tp = np.arange(Nt, dtype=np.float64) * 0.1
    f.set_protocol(tp, pv)
    t0 = time.perf_counter()
    generic.generic_dopri5(f, torch.tensor([[0.0, 1.0]], dtype=torch.float64), torch.from_numpy(tp[:n]))
    print(json.dumps({"seconds": time.perf_counter() - t0}))


def python_reference_structured_baseline(ion, prot_v, weights, Nt, gpu_out, budget_s=12.0):
    """BASELINE.md 'Baseline A': the reference's cost structure -- a Python dopri5 (the package's opt-in generic
    stepper) calling an nn.Module.forward shaped like train-s1.py:231-247 (SciPy interp1d on the host per RHS call +
    fp32 nn.Sequential).  torchdiffeq itself is not available here.  Bounded: integrates a prefix of the output grid that takes
    ~budget_s and extrapolates linearly in samples.  Two legs: one process / one thread (the reference is serial), and one process
    per host core over different trajectories (what PINTS' set_parallel(True) does, train-d0.py:538)."""
    generic = importlib.import_module("neural-ode-ion-channels_amd.generic")
    torch.set_num_threads(1)
    f = _ref_style_nnf(weights)
    tp = np.arange(Nt, dtype=np.float64) * 0.1
    f.set_protocol(tp, prot_v[0].cpu().numpy())
    y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64)
    n_probe = min(Nt, 2001)  # 200 ms of protocol to calibrate, then as much as the budget allows
    t0 = time.perf_counter()
    generic.generic_dopri5(f, y0, torch.from_numpy(tp[:n_probe]))
    dt_probe = time.perf_counter() - t0
    n = int(min(Nt, max(n_probe, n_probe * budget_s / max(dt_probe, 1e-3))))
    t0 = time.perf_counter()
    y = generic.generic_dopri5(f, y0, torch.from_numpy(tp[:n]))
    dt = time.perf_counter() - t0
    g = gpu_out["y"][0, :n].cpu().numpy()
    err = float(np.linalg.norm(y[:, 0, :].numpy() - g) / np.linalg.norm(g))
    res = {"value": 1.0 / (dt * Nt / n), "unit": "trajectories/s", "cores": 1, "kind": "port",
           "sample": f"trajectory 0, first {n} of {Nt} output samples in {dt:.1f} s, extrapolated linearly in samples; "
                     "Python dopri5 + nn.Module.forward with host interp1d per RHS call (the reference's cost structure)",
           "rel_l2_vs_gpu": err}
    try:
        import subprocess
        cores = usable_cores()
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--python-prefix-worker", str(i), str(n), str(Nt)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(cores)]
        times = []
        for pr in procs:   # (children of a GPU process are allowed; none of them initialises the GPU, and nothing is exec'ed in place)
            out, _ = pr.communicate(timeout=20 * budget_s + 120)
            times.append(json.loads(out.strip().splitlines()[-1])["seconds"])
        wall = time.perf_counter() - t0
        res["all_cores"] = {"value": cores / (max(times) * Nt / n), "unit": "trajectories/s", "cores": cores,
                            "sample": f"{cores} processes (one per usable core), trajectory i in process i, first {n} of {Nt} output samples: slowest "
                                      f"{max(times):.1f} s, fastest {min(times):.1f} s ({wall:.1f} s wall with interpreter start-up), extrapolated linearly"}
    except Exception as e:  # informational leg only
        res["all_cores"] = {"error": repr(e)}
    return res


if __name__ == "__main__":
    main()
