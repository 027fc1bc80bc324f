#!/usr/bin/env python3
"""VALU instructions of the single-level fused kernel per BLOCK of its loop (include/pt_api.h pt_fused_block), read from the shipped ISA.

    python scripts/isa_regions.py [--kernel k_fusedILi0ELb1E] [--json profiles/isa_valu_model.json] [--extra "-DX ..."]

The kernel's instrumented twin (PT_FLAG_COUNT_VISITS on PT_PIPELINE_FUSED) counts, per block, how often a wave executed it and how many lanes
were inside (pt_get_block_counts); this script supplies the other factor: how many VALU instructions the block is in the ISA of the PRODUCT
kernel.  fused.hip is compiled to gfx950 assembly with the flags of csrc/Makefile plus -gline-tables-only (the opcode sequence is the one without
it: checked here against a plain listing) and every instruction is given the block of the source line its `.loc` names:

  * a line of fused_kernel.h belongs to the innermost `PT_FB(FB_X)` marker whose enclosing braces contain it (the marker of the outer loop,
    FB_ITER, takes what no inner marker claims: the ballots at the loop's head and the votes between the phases);
  * extend_kernel.h (compact_node_step and what it is made of) is FB_NODE; pair_leaf.h is FB_LEAF except its divide block, the range check
    and the closest-hit rule, which are FB_DIV;
  * an instruction inlined from anywhere else (pt_math.h, the HIP headers) takes the block that the majority of the located instructions of
    its BASIC BLOCK have -- scheduling mixes instructions within a basic block only, and the blocks of the loop are separated by branches;
    a basic block without any located instruction inherits from the one before it.

Output: per block VALU / SALU / LDS / VMEM instruction counts; with --json the table is written into the model file bench.py multiplies with
the live counts (key "k_fused").  Re-run after any change to the kernel headers.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "single-file-vulkan-pathtracing_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split()
BLOCKS = ["ITER", "SHADE", "HIT", "MISS", "SURFACE", "ADD", "BOUNCE", "NEXT", "DONE", "HANDOUT", "DRAW", "TAKE", "CULLED", "PRIMARY", "SETUP",
          "NODE", "POP", "LEAF", "DIV", "FINISH", "TRACE", "SPAWN", "PTARGET", "PDIR", "POPTOP"]


def compile_s(extra, debug):
    out = f"/tmp/isa_regions_{'g' if debug else 'p'}.s"
    cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *extra, *(["-gline-tables-only"] if debug else []), "-S", "--cuda-device-only",
           os.path.join(CSRC, "fused.hip"), "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def kernel_body(lines, name):
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


def opcodes(body):
    return [t.split()[0] for t in (l.strip() for l in body) if t and not t.startswith((";", ".")) and not re.match(r"^\.?\w+:", t)]


def marker_regions(path):
    """line -> block name for fused_kernel.h, from the PT_FB markers and the braces around them"""
    src = open(path).read().splitlines()
    # depth before each line and the line where each open block closes
    depth, stack, close_of = 0, [], {}
    opens_at = {}
    for i, l in enumerate(src, 1):
        code = re.sub(r"//.*", "", l)
        code = re.sub(r'"[^"]*"', '""', code)
        for ch in code:
            if ch == "{":
                stack.append(i)
            elif ch == "}":
                if stack:
                    close_of[stack.pop()] = i
    # enclosing block (open line, close line) of a line: the innermost block opened on an earlier (or the same, if it stays open) line
    blocks = sorted(((o, c) for o, c in close_of.items() if c > o), key=lambda b: (b[0], -b[1]))

    def enclosing(line):
        best = None
        for o, c in blocks:
            if o <= line <= c and not (o == line and c == line):
                if best is None or (o >= best[0] and c <= best[1]):
                    best = (o, c)
        return best

    marks = []
    for i, l in enumerate(src, 1):
        if l.lstrip().startswith("#define"):
            continue
        for m in re.finditer(r"PT_FB\(FB_(\w+)\)", l):
            if m.group(1) == "TRACE":   # (a count without code of its own)
                continue
            e = enclosing(i)
            # a marker in a one-line block `if (c) { PT_FB(X) }` or in the body of a one-line lambda speaks for the block around that line
            marks.append((m.group(1), e))
    region = {}
    # widest first, so inner markers overwrite outer ones
    for name, (o, c) in sorted(marks, key=lambda m: -(m[1][1] - m[1][0])):
        for ln in range(o, c + 1):
            region[ln] = name
    return region


def pair_leaf_regions(path):
    src = open(path).read().splitlines()
    region = {i: "LEAF" for i in range(1, len(src) + 1)}

    def span(start_re, end_re):
        a = next(i for i, l in enumerate(src, 1) if re.search(start_re, l))
        b = next(i for i, l in enumerate(src, 1) if i > a and re.search(end_re, l))
        return a, b
    for a, b in (span(r"auto finish = \[&\]", r"^\s*\};"), span(r"if \(inA \|\| inB\) \{", r"if \(inA && inB\) finish"),
                 span(r"closer_single_level\(", r"^\}")):
        for ln in range(a, b + 1):
            region[ln] = "DIV"
    return region


def slow_divide_lines(path):
    """pt_math.h lines of the IEEE-divide fallbacks behind the guards of the three-FMA quotients (div2_dominant, div3_dominant, div3_by_pdf,
    ray_setup): operands outside 2^-100 .. 2^120 -- on the scenes measured they never run"""
    src = open(path).read().splitlines()
    out = set()
    for i, l in enumerate(src, 1):
        if "fdiv(" in l and any("} else {" in src[j - 1] for j in range(max(1, i - 3), i)):
            out.add(i)
    return out


def classify(body, files, reg_fused, reg_pair, slow_lines=frozenset()):
    """-> {block: Counter(valu, salu, lds, vmem)}, list of basic blocks [(label, majority, n_valu)]"""
    bbs = [["entry", [], False]]
    cur_loc = None
    for l in body:
        t = l.strip()
        m = re.match(r"^\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur_loc = (int(m.group(1)), int(m.group(2)))
            continue
        if re.match(r"^(\.LBB\d+_\d+):", t) or re.match(r"^; %bb\.\d+:", t):
            bbs.append([t.split(":")[0].lstrip("; "), [], "Depth=" in t])   # (LLVM's loop comments: a block without one is outside the loop)
            continue
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        blk = None
        if cur_loc:
            f = files.get(cur_loc[0], "")
            if f == "fused_kernel.h":
                blk = reg_fused.get(cur_loc[1])
            elif f == "extend_kernel.h":
                blk = "NODE" if cur_loc[1] < 132 or cur_loc[1] > 139 else None   # (slab_setup is called from the ray set-up)
            elif f == "pair_leaf.h":
                blk = reg_pair.get(cur_loc[1])
        slow = bool(cur_loc) and files.get(cur_loc[0], "") == "pt_math.h" and cur_loc[1] in slow_lines
        bbs[-1][1].append((op, blk, slow))
    totals = collections.defaultdict(collections.Counter)
    out_bbs, bb_ins = [], []
    prev = "PROLOGUE"
    for label, ins, in_loop in bbs:
        if not in_loop:   # set-up before the loop (LDS staging, hoisted invariants) and the two atomics behind it: once per wave
            ins = [(op, "PROLOGUE", False) for op, _, _ in ins]
        tally = collections.Counter(b for _, b, _ in ins if b)
        maj = tally.most_common(1)[0][0] if tally else prev
        prev = maj
        n_valu = sum(op.startswith("v_") for op, _, _ in ins)
        # a block that is nothing but two or three IEEE divide expansions (11 VALU each) is the fallback behind the guard of a three-FMA quotient
        # (pt_math.h div2_dominant / div3_dominant / div3_by_pdf / ray_setup: operands outside 2^-100 .. 2^120): kept apart, priced at zero executions
        n_fix = sum(op.startswith("v_div_fixup_f32") for op, _, _ in ins)
        if n_fix >= 2 and n_valu <= 12 * n_fix:
            maj = maj + "_SLOW"
            ins = [(op, maj, sl) for op, _, sl in ins]
        nv = 0
        for op, b, _ in ins:
            b = b or maj
            kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
            totals[b][kind] += 1
            nv += kind == "valu"
        out_bbs.append((label, maj, nv, len(ins)))
        bb_ins.append((label, ins, in_loop))
    classify.last_bbs = bb_ins
    return totals, out_bbs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_fusedILi0ELb1E")
    ap.add_argument("--extra", default="")
    ap.add_argument("--json", default=None)
    ap.add_argument("--bbs", action="store_true", help="list the basic blocks with their block and VALU count")
    a = ap.parse_args()
    extra = a.extra.split()
    plain = kernel_body(compile_s(extra, False), a.kernel)
    dbg_lines = compile_s(extra, True)
    dbg = kernel_body(dbg_lines, a.kernel)
    if opcodes(plain) != opcodes(dbg):
        sys.exit("the -gline-tables-only listing's opcode sequence differs from the plain one: cannot attribute")
    files = {}
    for l in dbg_lines:
        m = re.match(r'^\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
    reg_fused = marker_regions(os.path.join(CSRC, "fused_kernel.h"))
    reg_pair = pair_leaf_regions(os.path.join(CSRC, "pair_leaf.h"))
    totals, bbs = classify(dbg, files, reg_fused, reg_pair, slow_divide_lines(os.path.join(CSRC, "pt_math.h")))
    # a block inlined at several places (the pop loop: behind a node step that found nothing and behind a leaf step; the divide block: a pair's two
    # halves) is one of its copies per counted execution: the copies are told by an anchor instruction each of them has exactly once
    copies = {"POP": max(1, sum(1 for _ in ())), "DIV": 1}
    anchors = {"POP": "ds_read_b32", "DIV": "v_div_fixup_f32"}
    per_block_ops = collections.defaultdict(collections.Counter)
    cur = None
    for (label, maj, nv, n), (_, ins, _) in zip(bbs, classify.last_bbs):
        for op, b, _ in ins:
            per_block_ops[b or maj][op.split("_e32")[0].split("_e64")[0]] += 1
    for b, anchor in anchors.items():
        copies[b] = max(1, per_block_ops[b][anchor])
    if a.bbs:
        for label, maj, nv, n in bbs:
            print(f"{label:12s} {maj:10s} valu {nv:4d} of {n:4d}")
    print(f"{'block':14s} {'VALU':>5s} {'SALU':>5s} {'LDS':>4s} {'VMEM':>5s}")
    allv = 0
    for b in ["PROLOGUE", *BLOCKS, *sorted(k for k in totals if str(k).endswith("_SLOW")), None]:
        if b in totals:
            c = totals[b]
            allv += c["valu"]
            print(f"{str(b):14s} {c['valu']:5d} {c['salu']:5d} {c['lds']:4d} {c['vmem']:5d}" + (f"   ({copies[b]} inlined copies: {c['valu'] / copies[b]:.1f} per execution)" if copies.get(b, 1) > 1 else ""))
    print(f"VALU instructions of the kernel: {allv} (listing: {sum(o.startswith('v_') for o in opcodes(plain))})")
    if a.json:
        model = json.load(open(a.json)) if os.path.exists(a.json) else {}
        rev = subprocess.check_output(["git", "-C", REPO, "rev-parse", "--short", "HEAD"]).decode().strip()
        model["k_fused"] = {"revision": rev + " (+ working tree)", "kernel": a.kernel,
                            "source": "scripts/isa_regions.py: .loc-attributed instructions of the product kernel's listing",
                            "blocks": {b: {"valu": round(totals[b]["valu"] / copies.get(b, 1), 1), "salu": totals[b]["salu"], "lds": totals[b]["lds"],
                                           "vmem": totals[b]["vmem"], "copies": copies.get(b, 1)} for b in BLOCKS},
                            "never_executed_fallbacks": {b: totals[b]["valu"] for b in totals if str(b).endswith("_SLOW")}}
        json.dump(model, open(a.json, "w"), indent=1)
        print("wrote", a.json)


if __name__ == "__main__":
    main()
