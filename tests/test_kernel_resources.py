"""Code-object hygiene (no GPU): register spills to scratch are performance bugs the profiler only shows indirectly, so the
build is checked for them.  Round-2 review: 44-72 B of scratch in the 2-state three-per-SIMD builds, 72 B in the N = 500 sweep and
regression kernels.  Round 3: no kernel of libionode.so uses scratch."""
import importlib
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") or shutil.which("c++filt") is None, reason="llvm tools")
def test_no_kernel_uses_scratch():
    ion = importlib.import_module("neural-ode-ion-channels_amd")
    from kernel_resources import kernel_resources
    rows = kernel_resources(ion.capi.LIB_PATH)
    names = [r["kernel"] for r in rows]
    assert len(rows) >= 90 and any("ionode_grad_walk_kernel<3, double>" in n for n in names)
    # round 5: the N = 200 recompute kernels (phase A of the gradient sweep) are built for TWO workgroups per compute unit (256 registers):
    # the weight ring's 172 registers stay live through an iteration's fp64 prologue, which parks up to 26 values (<= 128 bytes per lane) in
    # scratch there -- once per ~140 us iteration, never inside a product (test_recompute_kernel_spills_stay_out_of_the_products)
    allowed = {"ionode_grad_recompute_kernel<2, double, 13>": 128, "ionode_grad_recompute_kernel<2, float, 13>": 128,
               "ionode_grad_recompute_kernel<3, double, 13>": 128, "ionode_grad_recompute_kernel<3, float, 13>": 128}
    bad = [(r["kernel"], r["scratch_bytes"], r["vgpr_spill"]) for r in rows
           if r["scratch_bytes"] > max([v for a, v in allowed.items() if a in r["kernel"]], default=0)]   # (spills into free AGPRs cost no scratch)
    assert not bad, bad
    # the hand-scheduled N = 200 kernels: the asm stream owns a[0:91]; the compiler's own spills must fit the other AGPRs
    for r in rows:
        tail = int(r["kernel"].split(",")[-1].split(">")[0]) if "ionode_dopri5_kernel<" in r["kernel"] else 0
        if "ionode_dopri5_kernel<" in r["kernel"] and ", 4, 4, 13, 13, " in r["kernel"] and not (tail & 16):   # (TAIL & 16: the 4-trajectory tile, no asm stream)
            assert r["vgpr"] <= 512 and 92 <= r["agpr"] <= 256 and r["scratch_bytes"] == 0, r   # vgpr = unified VGPR + AGPR count


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc")
def test_recompute_kernel_spills_stay_out_of_the_products():
    """The 256-register recompute kernels may spill in an iteration's prologue (checkpoint, output gradients, stage inputs) -- every
    scratch instruction must sit in front of the kernel's first MFMA, i.e. outside the six vector-Jacobian products of an iteration."""
    import re
    import asm_stats
    asm = asm_stats.compile_asm("ionode_grad_capi")
    n = 0
    for sym, body in re.findall(r"^(_ZN6ionode28ionode_grad_recompute_kernel\w+):[^\n]*\n(.*?)\n\s+s_endpgm", asm, re.S | re.M):
        lines = body.split("\n")
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        sc = [i for i, l in enumerate(lines) if re.match(r"\s+scratch_", l)]
        assert mf, sym
        assert not sc or max(sc) < mf[0], (sym, [i for i in sc if i >= mf[0]][:5], mf[0])
        n += 1
    assert n == 12, n   # {NN-f, NN-d} x {fp32, fp64} x {N <= 16, 100, 200}


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") or shutil.which("c++filt") is None, reason="llvm tools")
def test_two_state_kernels_fit_three_wavefronts_per_simd():
    """The 2-state closed-form kernels are vector-issue bound and need three resident wavefronts per SIMD: 512 / 3 -> 168 registers
    (allocation granule 8); the lean variant (states only, verified uniform grids: the headline closed-form workload) four: 128.
    They are compiled with a two-per-SIMD launch bound (hipcc then needs 118-158); this guards the budgets."""
    ion = importlib.import_module("neural-ode-ion-channels_amd")
    from kernel_resources import kernel_resources
    rows = [r for r in kernel_resources(ion.capi.LIB_PATH) if "ionode_dopri5_kernel<0, " in r["kernel"]]
    assert len(rows) == 12, [r["kernel"] for r in rows]   # {fp64, fp32} x {64, 16 per wavefront} x {general, lean, table}
    for r in rows:
        lean = r["kernel"].endswith(", 1>(ionode::KArgs)")
        assert r["vgpr"] <= (128 if lean else 168) and r["vgpr_spill"] == 0 and r["scratch_bytes"] == 0, r
    assert sum(r["kernel"].endswith(", 1>(ionode::KArgs)") for r in rows) == 4


# Static instruction budgets of the attempt loops (tools/asm_stats.py: the unit compiled to assembly with the Makefile's flags).
# SGPR-spill lane traffic (v_readlane / v_writelane) and constant materialisation (v_mov of a literal) are VALU instructions on the pipe
# that bounds these kernels; round 3's review found 132 + 44 lane operations and 498 constant moves in the 2-state attempt loop, 626
# v_readlane in the s00 kernel's.  The budgets are the round-4 figures plus ~10 %: a change that lets the traffic creep back fails here.
_BUDGETS = {
    # unit, kernel substring: (max v_readlane + v_writelane, max literal v_mov_b32 + v_mov_b64, max canonicalising v_max x, x, x, max AGPR copies)
    ("inst_closed", "<0, double, 1, 0, 0, 0, 1>"): (20, 175, 0, 0),      # 2-state, lean variant (round 3: 176 lane operations, 498 constant moves)
    ("inst_closed", "<0, float, 1, 0, 0, 0, 2>"): (150, 225, 0, 0),      # 2-state, table variant (fused objective)
    ("inst_closed", "<1, double, 1, 0, 0, 0, 1>"): (20, 220, 0, 0),      # 6-state, lean variant (round 3: 592 lane operations, 353 AGPR copies)
}


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="hipcc")
@pytest.mark.parametrize("unit", sorted({u for u, _ in _BUDGETS}))
def test_attempt_loop_instruction_budgets(unit):
    from asm_stats import compile_asm, kernel_stats
    st = kernel_stats(compile_asm(unit))
    for (u, key), (lane_ops, const_movs, max_self, agpr) in _BUDGETS.items():
        if u != unit:
            continue
        k = [v for n, v in st.items() if key in n]
        assert len(k) == 1, (key, list(st))
        a = k[0]["attempt_loop"]
        assert a["valu"] > 1000, a   # the attempt loop was found
        assert a["readlane"] + a["writelane"] <= lane_ops, (key, a)
        assert a["mov_const32"] + a["mov_const64"] <= const_movs, (key, a)
        assert a["max_self"] <= max_self, (key, a)
        assert a["accvgpr"] <= agpr, (key, a)


_ASM_CACHE = {}


def _nnf_f64_asm():
    if "nnf64" not in _ASM_CACHE:
        from asm_stats import compile_asm
        _ASM_CACHE["nnf64"] = compile_asm("inst_nnf_f64")
    return _ASM_CACHE["nnf64"]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="hipcc")
def test_tile_kernels_keep_their_dense_output_stores():
    """Every forward kernel writes its dense output itself.  (Round 4: an edit of the shared emission code dropped the MLP tile
    kernels' block; the CPU suite stayed green and only the GPU parity suite noticed.)  The N = 200 / N = 100 / N = 500 tile kernels
    and the 4-trajectory tile each keep their state stores (16-byte stores of a 2 x fp64 sample) in the code object."""
    asm = _nnf_f64_asm()
    # (minimum = ~80 % of the round-4 count: 4 owned trajectories x {first chunk, later chunks} per wavefront, + the initial sample)
    for sym, least in (("ILi2EdLi4ELi4ELi13ELi13ELi8EE", 10), ("ILi2EdLi4ELi4ELi13ELi13ELi12EE", 16), ("ILi2EdLi4ELi4ELi13ELi13ELi24EE", 5),
                       ("ILi2EdLi4ELi4ELi7ELi7ELi8EE", 10), ("ILi2EdLi4ELi8ELi32ELi4ELi8EE", 10), ("ILi2EdLi1ELi64ELi1ELi10ELi1EE", 6),
                       ("ILi2EdLi1ELi1ELi1ELi1ELi0EE", 14)):
        start = asm.index("_ZN6ionode20ionode_dopri5_kernel%sEvNS_5KArgsE:" % sym)
        body = asm[start:asm.index(".end_amdhsa_kernel", start)]
        assert body.count("global_store_dwordx4") >= least, (sym, body.count("global_store_dwordx4"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="hipcc")
def test_per_lane_net_issues_packed_fma_with_scalar_weights():
    """The N = 10 kernel at one trajectory per lane evaluates two rows of a layer per v_pk_fma_f32, weights as an SGPR pair and the
    activation broadcast by op_sel (round 4: the vector fp32 peak is the packed rate).  A change that silently falls back to one fmaf per
    instruction, or that reintroduces canonicalising v_max x, x, x around LeakyReLU, fails here."""
    import re
    asm = _nnf_f64_asm()
    start = asm.index("_ZN6ionode20ionode_dopri5_kernelILi2EdLi1ELi64ELi1ELi10ELi1EEEvNS_5KArgsE:")
    body = asm[start:asm.index(".end_amdhsa_kernel", start)]
    pk = re.findall(r"v_pk_fma_f32 v\[\d+:\d+\], s\[\d+:\d+\], v\[\d+:\d+\], v\[\d+:\d+\]", body)
    assert len(pk) >= 400, len(pk)                       # 8 inlined evaluations x (10 layer-0 + 50 per hidden-layer body) = 480
    assert sum("op_sel:[0,1,0]" in l for l in re.findall(r"v_pk_fma_f32[^\n]*", body)) >= 150   # both halves of the activation pairs are used
    assert not re.search(r"v_max_f32(?:_e32|_e64)? (v\d+), \1, \1\b", body)
    assert "scratch_" not in body
    # round 5: Linear(2, N) / Linear(N, 1) are scalar loads of the evaluation that uses them and half a layer's weights are in flight at a
    # time -- the kernel's uniform state stays in scalar registers (round 4: 684 v_readlane in the kernel, 539 of them per attempt)
    assert len(re.findall(r"v_readlane_b32", body)) <= 110, len(re.findall(r"v_readlane_b32", body))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc")
def test_no_dpp_source_is_read_too_early():
    """Round 5: the one-trajectory tile's chain links are inline-asm `v_fmac_f32_dpp`; hipcc's hazard recognizer does not look inside
    inline asm, and gfx9 requires two wait states between a VALU write of a VGPR and a DPP read of it.  The DPP sources come straight
    from LDS reads (no hazard) -- unless a future compiler puts a vector move in front of one.  Every MLP unit, every kernel, statically."""
    from concurrent.futures import ThreadPoolExecutor
    import asm_stats
    units = ("inst_nnf_f32", "inst_nnf_f64", "inst_nnd_f32", "inst_nnd_f64")
    with ThreadPoolExecutor(4) as ex:
        texts = list(ex.map(asm_stats.compile_asm, units))
    for unit, txt in zip(units, texts):
        assert txt.count("v_fmac_f32_dpp") >= 8 * 208 * 2, unit          # the links are there (general + lean variant, eight inlined evaluations)
        bad = asm_stats.dpp_hazards(txt)
        assert not bad, (unit, bad[:3])
    probe = "_Zk:\n\tv_mov_b32_e32 v5, v7\n\tv_fmac_f32_dpp v1, v5, v9 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\ts_endpgm\n"
    assert len(asm_stats.dpp_hazards(probe)) == 1                          # the checker sees what it is looking for
