"""Dev tool: per-kernel register / spill / scratch figures of a built libionode.so, read from the code objects' metadata notes.
python tools/kernel_resources.py [path/to/libionode.so] [--json]
(no GPU needed: llvm-objcopy + clang-offload-bundler + llvm-readelf from /opt/rocm/lib/llvm/bin)"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_resources(lib):
    out = []
    with tempfile.TemporaryDirectory() as d:
        fb = os.path.join(d, "fatbin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", lib, os.path.join(d, "copy.so")])
        blob = open(fb, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):  # one bundle per translation unit
            part = os.path.join(d, f"b{i}.bin")
            open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, f"b{i}.co")
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
            notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
            for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                f = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
                name = f("name").group(1)
                out.append({"kernel": subprocess.check_output(["c++filt", name], text=True).strip(),
                            "agpr": int(re.match(r"\s*(\d+)", blk).group(1)), "vgpr": int(f("vgpr_count").group(1)),
                            "sgpr": int(f("sgpr_count").group(1)), "vgpr_spill": int(f("vgpr_spill_count").group(1)),
                            "sgpr_spill": int(f("sgpr_spill_count").group(1)), "scratch_bytes": int(f("private_segment_fixed_size").group(1)),
                            "lds_static": int(f("group_segment_fixed_size").group(1))})
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(__file__), "..", "neural-ode-ion-channels_amd", "libionode.so")
    rows = kernel_resources(lib)
    if "--json" in sys.argv:
        json.dump(rows, sys.stdout, indent=1)
    else:
        print(f"{'kernel':100s} vgpr(unified) agpr sgpr vspill sspill scratch")
        for r in sorted(rows, key=lambda r: r["kernel"]):
            print(f"{r['kernel'][:100]:100s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['vgpr_spill']:6d} {r['sgpr_spill']:6d} {r['scratch_bytes']:7d}")
