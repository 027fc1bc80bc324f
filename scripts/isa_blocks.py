#!/usr/bin/env python3
"""VALU instructions per basic block of one kernel of the shipped HIP library, for the live VALU model of bench.py.

    python scripts/isa_blocks.py [--kernel k_extend_lds7] [--src extend_launch.hip] [--extra flags]

Compiles the translation unit to gfx950 assembly with the flags of csrc/Makefile (`hipcc -S --cuda-device-only`), cuts
the kernel at its labels and prints, per block: loop depth (LLVM's loop comments), VALU / SALU / LDS / VMEM instruction
counts and the anchors that identify what the block is (7 x ds_read_b128 = a BVH4 node staged in LDS, ds_read_b96 = a
triangle, global_load = the refill, v_div_fixup = an IEEE divide, ds_write_b32 = a stack push, global_store = the hit
record).  By these anchors the blocks are classified (by hand: profiles/isa_valu_model.json names the labels it summed)
into the seven block kinds the instrumented kernel counts at wave level (pt_stats.wave_*, node_steps, tri_steps);
bench.py multiplies that table with the live counts.  The classification is checked against a PMC run
(SQ_INSTS_VALU) in profiles/r02_valu_model_check.json.
"""
import argparse
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "single-file-vulkan-pathtracing_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split()


def assembly(src, extra):
    out = f"/tmp/isa_{os.path.basename(src)}.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out],
                          stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def kernel_lines(lines, name):
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*" + re.escape(name) + r"\w*:", l):
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    if start is None:
        sys.exit(f"kernel {name} not found")
    return lines[start + 1:end]


def blocks(klines):
    """-> [{label, depth, valu, salu, lds, vmem, anchors, text}] in program order"""
    out = [dict(label="entry", depth=0, ins=[])]
    for l in klines:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            d = 0
            c = m.group(2) or ""
            dm = re.search(r"Depth=(\d+)", c)
            if dm:
                d = int(dm.group(1))
            out.append(dict(label=m.group(1), depth=d, ins=[], comment=c))
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            # loop comments of fall-through blocks ("; %bb.N: ; in Loop: Header=BB0_4 Depth=1") open a new block too
            m2 = re.match(r"^; %bb\.(\d+):\s*(;.*)?$", t)
            if m2:
                c = m2.group(2) or ""
                dm = re.search(r"Depth=(\d+)", c)
                out.append(dict(label="%bb." + m2.group(1), depth=int(dm.group(1)) if dm else 0, ins=[], comment=c))
            continue
        out[-1]["ins"].append(t.split(";")[0].strip())
    res = []
    for b in out:
        ins = b["ins"]
        op = [i.split()[0] for i in ins]
        anchors = []
        for key in ("ds_read_b128", "ds_read_b96", "ds_read_b64", "ds_read_b32", "ds_write_b32", "ds_write_b64", "ds_write_b128",
                    "global_load_dwordx4", "global_load_dwordx2", "global_store_dwordx4", "v_div_fixup_f32", "v_rcp_f32", "global_atomic"):
            n = sum(o.startswith(key) for o in op)
            if n:
                anchors.append(f"{n}x{key}")
        res.append(dict(label=b["label"], depth=b["depth"], valu=sum(o.startswith("v_") for o in op),
                        salu=sum(o.startswith("s_") for o in op), lds=sum(o.startswith("ds_") for o in op),
                        vmem=sum(o.startswith(("global_", "buffer_", "flat_")) for o in op), anchors=anchors, n=len(ins)))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_extend_lds7")
    ap.add_argument("--src", default="extend_launch.hip")
    ap.add_argument("--extra", default="", help="extra compiler flags of that translation unit (extend_hbm.hip: -mllvm -amdgpu-sched-strategy=max-ilp)")
    args = ap.parse_args()
    bl = blocks(kernel_lines(assembly(args.src, args.extra.split()), args.kernel))
    print(f"{'block':<12}{'depth':>5}{'VALU':>6}{'SALU':>6}{'LDS':>5}{'VMEM':>5}  anchors")
    for b in bl:
        if b["n"]:
            print(f"{b['label']:<12}{b['depth']:>5}{b['valu']:>6}{b['salu']:>6}{b['lds']:>5}{b['vmem']:>5}  {' '.join(b['anchors'])}")
    print("total VALU", sum(b["valu"] for b in bl))


if __name__ == "__main__":
    main()
