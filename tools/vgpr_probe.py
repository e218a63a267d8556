"""Dev tool: compile ONE instantiation unit with extra -D flags and print the register / spill figures of the kernels whose name
contains a substring (reads the .amdhsa_ / metadata lines of the assembly; no GPU).
    python tools/vgpr_probe.py inst_nnf_f64 "1, 64, 1, 10, 1>" -DIONODE_T64_WAVES=4 ..."""
import re
import subprocess
import sys

from asm_stats import compile_asm


def main():
    unit, key, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    txt = compile_asm(unit, extra)
    # metadata block: one entry per kernel with .name, .vgpr_count, .vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size
    for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if key not in dem:
            continue
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        print("%-70s vgpr %3d  vgpr_spill %3d  sgpr_spill %3d  scratch %4d" % (dem[:70], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
