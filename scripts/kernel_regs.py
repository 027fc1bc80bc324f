"""VGPRs / scratch / spills per kernel of assembly listings (hipcc -S --cuda-device-only): python scripts/kernel_regs.py a.s b.s [filter]"""
import re, sys
files = [a for a in sys.argv[1:] if a.endswith(".s")]
flt = [a for a in sys.argv[1:] if not a.endswith(".s")]
for f in files:
    s = open(f).read()
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", s):
        n = m.group(1)
        if flt and not any(x in n for x in flt):
            continue
        short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:40]
        print(f.split("/")[-1], short, "scratch", m.group(2), "vgpr", m.group(3), "spill", m.group(4))
