"""Dev tool: static instruction statistics of the gfx950 code hipcc emits for one translation unit of csrc/ (no GPU needed).

python tools/asm_stats.py inst_closed3 [--kernel 'ILi0EdLi1ELi0ELi0ELi0ELi1E'] [--extra='-DX=1'] [--json]

Compiles the unit to device assembly with the Makefile's flags for that unit (--offload-device-only -S), finds every kernel's
ATTEMPT LOOP (the outermost loop that contains other loops and the most instructions) and counts, inside it and in the whole kernel:
vector instructions, MFMAs, fp64 instructions, literal / constant materialisation (v_mov_b32 / v_mov_b64 of a constant), SGPR-spill lane
traffic (v_readlane / v_writelane), canonicalising v_max_f32 x, x, x, AGPR copies, scalar and memory instructions.
tests/test_kernel_resources.py asserts budgets on these figures, so that avoidable VALU work cannot creep back into the hot loops.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neural-ode-ion-channels_amd", "csrc")
BASE = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function".split()


def unit_flags(unit):
    """The per-unit flags of csrc/Makefile (read from it, so the two cannot drift)."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    var = lambda n: re.search(r"^%s\s*=\s*(.*)$" % n, mk, re.M).group(1).split()
    if unit.startswith("inst_nn"):
        return var("MLPFLAGS")
    if unit == "inst_grad32":
        return var("GRADFLAGS")
    if unit == "ionode_grad_capi":
        return var("GRADFLAGS")
    m = re.search(r"^%s\.o:.*\n\t\$\(HIPCC\) \$\(FLAGS\) (.*?) \$\(EXTRA\)" % re.escape(unit), mk, re.M)
    return m.group(1).split() if m else []


def compile_asm(unit, extra=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc"] + BASE + unit_flags(unit) + list(extra) + ["--offload-device-only", "-S", os.path.join(CSRC, unit + ".hip"), "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL, cwd=CSRC)
    txt = open(out).read()
    os.unlink(out)
    return txt


CONST = r"(?:-?(?:0x[0-9a-f]+|\d+(?:\.\d+)?(?:e[-+]?\d+)?)|0\.5|-0\.5|1\.0|-1\.0|2\.0|-2\.0|4\.0|-4\.0)"
RE_MOVC32 = re.compile(r"^v_mov_b32(?:_e32)? v\d+, %s\b" % CONST)
RE_MOVC64 = re.compile(r"^v_mov_b64(?:_e32)? v\[\d+:\d+\], (?:%s|s\[)" % CONST)
RE_MAXSELF = re.compile(r"^v_max_f(?:32|64)(?:_e32|_e64)? (v\d+|v\[\d+:\d+\]), \1, \1\b")
FP64 = re.compile(r"^v_\w+_f64")


def classify(ins, c):
    op = ins.split()[0]
    c["total"] += 1
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        c["mfma"] += 1
    elif op.startswith("v_"):
        c["valu"] += 1
        if FP64.match(op):
            c["fp64"] += 1
        if RE_MOVC32.match(ins):
            c["mov_const32"] += 1
        elif RE_MOVC64.match(ins):
            c["mov_const64"] += 1
        elif op.startswith("v_mov_b32") or op.startswith("v_mov_b64"):
            c["mov_reg"] += 1
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            c["readlane"] += 1
        if op.startswith("v_writelane"):
            c["writelane"] += 1
        if RE_MAXSELF.match(ins):
            c["max_self"] += 1
        if op.startswith("v_accvgpr"):
            c["accvgpr"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
        if op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
    elif op.startswith(("ds_",)):
        c["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c["vmem"] += 1
        if op.startswith("scratch_"):
            c["scratch"] += 1


KEYS = ["total", "valu", "mfma", "fp64", "mov_const32", "mov_const64", "mov_reg", "readlane", "writelane", "max_self", "accvgpr",
        "salu", "waitcnt", "lds", "vmem", "scratch"]


def kernel_stats(asm):
    """{demangled kernel name: {"kernel": counts, "attempt_loop": counts, "loops": [(label, depth, n_instr)]}}"""
    res = {}
    lines = asm.split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):\s+; @", l)] if m]
    for si, (i0, sym) in enumerate(starts):
        i1 = next((i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines))
        body = lines[i0 + 1:i1]
        # instruction list with positions; labels; loop headers
        instrs, labels, headers, block_notes = [], {}, {}, []
        for k, l in enumerate(body):
            s = l.strip()
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                labels[m.group(1)] = len(instrs)
                hm = re.search(r"(?:Header=|Parent Loop )(BB\d+_\d+)", s)
                if hm:
                    block_notes.append((hm.group(1), len(instrs)))
                dm = re.search(r"Depth=(\d+)", s)
                if "Loop Header" in s and dm:
                    headers[m.group(1)] = int(dm.group(1))
                elif k + 1 < len(body) and "Loop Header" in body[k + 1]:
                    pass
                continue
            if not s or s.startswith((";", ".", "//")):
                # a header comment can sit on a continuation line:  "; =>  This Inner Loop Header: Depth=2"
                hm = re.search(r"Parent Loop (BB\d+_\d+)", s)
                if hm:
                    block_notes.append((hm.group(1), len(instrs)))
                dm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", s)
                if dm and labels:
                    last = max(labels, key=lambda q: labels[q])
                    if labels[last] == len(instrs):
                        headers[last] = int(dm.group(1))
                continue
            instrs.append(s.split(";")[0].strip())
        # loop extent: header position .. end of the last block hipcc annotates with "in Loop: Header=<label>" (a back edge can be a
        # long-branch trampoline, so the branch target alone is not reliable); nested loops' blocks name their own header and their
        # parent chain, so the outermost loop also claims every "Parent Loop <label>" block
        loops = []
        for lab, depth in headers.items():
            a = labels[lab]
            bare = lab.lstrip(".L")
            last = a
            for lab2, pos in block_notes:
                if pos >= a and bare == lab2:
                    last = max(last, pos)
            # extend to the end of that block: the next label position after `last`
            nxt = sorted(v for v in labels.values() if v > last)
            b = (nxt[0] - 1) if nxt else len(instrs) - 1
            bb = max((k for k, ins in enumerate(instrs) if k >= a and re.search(r"(?<![\w.])%s(?!\d)" % re.escape(lab), ins) and ins.startswith(("s_cbranch", "s_branch"))), default=a)
            loops.append((lab, depth, a, max(b, bb)))
        tot = {k: 0 for k in KEYS}
        for ins in instrs:
            classify(ins, tot)
        att = {k: 0 for k in KEYS}
        top = [l for l in loops if l[1] == 1]
        main = max(top, key=lambda l: l[3] - l[2]) if top else None
        if main:
            for ins in instrs[main[2]:main[3] + 1]:
                classify(ins, att)
        name = subprocess.check_output(["c++filt", sym], text=True).strip()
        name = re.sub(r"^void ionode::", "", name).replace("(ionode::KArgs)", "")
        res[name] = {"symbol": sym, "kernel": tot, "attempt_loop": att,
                     "loops": [(l[0], l[1], l[3] - l[2] + 1) for l in sorted(loops, key=lambda l: l[2])]}
    return res


def dpp_hazards(asm_text, wait_states=2):
    """Static check of the gfx9 hazard 'VALU writes a VGPR -> a DPP instruction reads it as its DPP source within `wait_states` wait states'.
    hipcc's hazard recognizer does not look inside inline-asm statements (the one-trajectory tile's v_fmac_f32_dpp links are inline asm; their
    DPP sources come straight from LDS reads, which are not VALU writes) -- this walks every kernel of the unit and returns
    [(kernel symbol, dpp instruction, offending instruction)] for DPP sources written by a VALU instruction fewer than wait_states + 1
    instructions earlier (s_nop N counts N + 1; branches / labels end the window conservatively: a hazard across them is not assumed)."""
    bad = []
    for sym, body in re.findall(r"^(_Z\w+):[^\n]*\n(.*?)\n\s+s_endpgm", asm_text, re.S | re.M):
        window = []   # (distance already accumulated is implicit: list of (n_wait_states, written vgprs)) for the last instructions
        for raw in body.split("\n"):
            ins = raw.split(";")[0].strip()
            if not ins or ins.startswith(".") or ins.startswith("//"):
                continue
            if ins.endswith(":"):
                window = []
                continue
            op = ins.split()[0]
            if "_dpp" in op:
                ops = [o.strip() for o in ins[len(op):].split(",")]
                # VOP2 dpp: vdst, src0 (the DPP source), src1;  VOP1 dpp (v_mov_b32_dpp): vdst, src0
                src = ops[1].split()[0] if len(ops) > 1 else ""
                dist = 0
                for ws, written, text in reversed(window):
                    if dist >= wait_states:
                        break
                    if src in written:
                        bad.append((sym, ins, text))
                        break
                    dist += ws
            written = set()
            ws = 1
            if op == "s_nop":
                ws = int(ins.split()[1]) + 1
            elif op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                d = ins[len(op):].split(",")[0].strip()
                m = re.match(r"v\[(\d+):(\d+)\]", d)
                if m:
                    written = {"v%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1)}
                elif re.match(r"v\d+$", d):
                    written = {d}
            if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
                window = []
            else:
                window.append((ws, written, ins))
                window = window[-4:]
    return bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = {a.split("=", 1)[0]: (a.split("=", 1)[1] if "=" in a else True) for a in sys.argv[1:] if a.startswith("--")}
    unit = args[0]
    extra = opt.get("--extra", "").split() if isinstance(opt.get("--extra"), str) else []
    st = kernel_stats(compile_asm(unit, extra))
    sel = opt.get("--kernel")
    if "--json" in opt:
        json.dump({k: {"kernel": v["kernel"], "attempt_loop": v["attempt_loop"]} for k, v in st.items() if not sel or sel in k or sel in v["symbol"]}, sys.stdout, indent=1)
        return
    cols = ["total", "valu", "mfma", "fp64", "mov_const32", "mov_const64", "mov_reg", "readlane", "writelane", "max_self", "accvgpr", "salu", "lds", "vmem"]
    print(f"{'kernel / scope':78s} " + " ".join(f"{c[:9]:>9s}" for c in cols))
    for k, v in st.items():
        if sel and sel not in k and sel not in v["symbol"]:
            continue
        for scope in ("kernel", "attempt_loop"):
            print(f"{(k + ' / ' + scope)[:78]:78s} " + " ".join(f"{v[scope][c]:9d}" for c in cols))


if __name__ == "__main__":
    main()
