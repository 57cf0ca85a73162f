#!/usr/bin/env python
"""Static VALU issue-cost of a kernel's basic blocks (no GPU needed), priced
with the MEASURED cost classes of profiles/r5a_valu_calibration.json
(tools/calib_valu.hip): on gfx950 a plain float32 / shift / logic wave64
instruction occupies its SIMD's issue for 2.4 cycles, packed float32
(v_pk_*), conversions (v_cvt_*), 24-bit multiplies and three-operand integer
logic for 4.3, v_rcp_f32 for 8.3 -- so "instructions" is the wrong unit for a
VALU-bound kernel and this tool prints cycles.

    valu_cost.py <file.hip | file.s> <kernel-name-substring> [min_block_valu]

Per basic block: instructions, VALU instructions, priced cycles, memory
instructions and where its branches go -- enough to follow the integrate
role's issue / apply rounds by eye and add up a frame's path. Opcodes without
a measured class are priced at 2.4 (full rate) and listed at the end, so that
the guess is visible."""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_scan  # noqa: E402

FULL, HALF, QUARTER = 2.4, 4.3, 8.3
# measured (profiles/r5a_valu_calibration.json, 8 waves per SIMD)
MEASURED = {
    "v_mul_f32": FULL, "v_add_f32": FULL, "v_fma_f32": FULL,
    "v_lshrrev_b32": FULL, "v_pk_mul_f32": HALF, "v_pk_fma_f32": HALF,
    "v_pk_add_f32": HALF, "v_cvt_f32_u32": HALF, "v_cvt_u32_f32": HALF,
    "v_rcp_f32": QUARTER, "v_mul_u32_u24": HALF, "v_and_or_b32": HALF,
}
# same pipes by construction (assumed, flagged in the output)
ASSUMED = {
    "v_sub_f32": FULL, "v_fmac_f32": FULL, "v_mac_f32": FULL,
    "v_lshlrev_b32": FULL, "v_and_b32": FULL, "v_or_b32": FULL,
    "v_xor_b32": FULL, "v_mov_b32": FULL, "v_add_u32": FULL,
    "v_sub_u32": FULL, "v_min_u32": FULL, "v_max_u32": FULL,
    "v_min_f32": FULL, "v_max_f32": FULL, "v_trunc_f32": FULL,
    "v_cvt_i32_f32": HALF, "v_cvt_f32_i32": HALF, "v_cvt_f32_ubyte0": HALF,
    "v_cvt_f32_ubyte1": HALF, "v_cvt_f32_ubyte2": HALF,
    "v_cvt_f32_ubyte3": HALF, "v_mad_u32_u24": HALF, "v_mul_lo_u32": QUARTER,
    "v_mad_u64_u32": QUARTER, "v_rsq_f32": QUARTER, "v_sqrt_f32": QUARTER,
    "v_bfe_u32": HALF, "v_lshl_add_u32": HALF, "v_add3_u32": HALF,
    "v_lshl_or_b32": HALF, "v_alignbit_b32": HALF, "v_perm_b32": HALF,
    "v_div_scale_f32": HALF, "v_div_fmas_f32": HALF, "v_div_fixup_f32": HALF,
    # the measured pair v_cmp_lt_f32 + v_cndmask_b32 costs 6.7 for two
    "v_cndmask_b32": 3.35, "v_cmp": 3.35,
}


def price(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in MEASURED:
        return MEASURED[base], "measured"
    if base.startswith("v_pk_"):
        return HALF, "measured"
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        return ASSUMED["v_cmp"], "assumed"
    if base in ASSUMED:
        return ASSUMED[base], "assumed"
    return FULL, "guess"


def blocks_of(body):
    out, cur, name = [], [], "entry"
    for ln in body:
        if ln.endswith(":") and ln.startswith(".L"):
            out.append((name, cur))
            name, cur = ln[:-1], []
        else:
            cur.append(ln)
    out.append((name, cur))
    return out


def main():
    src, want = sys.argv[1], sys.argv[2]
    min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if src.endswith(".s"):
        text = open(src).read()
    else:
        tmp = "/tmp/valu_cost_%d.s" % os.getpid()
        isa_scan.compile_to_asm(src, tmp)
        text = open(tmp).read()
        os.unlink(tmp)
    # keep the labels: isa_scan.kernels() normalises them away
    m = None
    for mm in re.finditer(r"^(_Z\S+):[^\n]*\n", text, re.M):
        if want in mm.group(1):
            m = mm
            break
    if m is None:
        sys.exit("no kernel matching %r" % want)
    end = text.find(".Lfunc_end", m.end())
    body = []
    for ln in text[m.end():end].splitlines():
        ln = ln.split(";")[0].strip()
        if not ln or (ln.startswith(".") and not ln.startswith(".LBB")):
            continue
        body.append(ln)
    blocks = blocks_of(body)
    index = {n: i for i, (n, _) in enumerate(blocks)}
    guesses = collections.Counter()
    total = 0.0
    print("# %s" % m.group(1))
    print("# idx label insts valu cycles  memory-ops  branches->idx")
    for i, (name, b) in enumerate(blocks):
        cyc, nv = 0.0, 0
        mem = collections.Counter()
        br = []
        for ln in b:
            op = ln.split()[0]
            if op.startswith("v_"):
                c, how = price(op)
                cyc += c
                nv += 1
                if how != "measured":
                    guesses[(re.sub(r"_(e32|e64)$", "", op), how)] += 1
            elif op.split("_")[0] in ("global", "flat", "ds", "buffer",
                                      "scratch"):
                mem[op] += 1
            elif op.startswith("s_cbranch") or op == "s_branch":
                br.append("%s->%s" % (op[2:], index.get(ln.split()[1], "?")))
        total += cyc
        if nv >= min_valu:
            print("%4d %-12s %4d %4d %7.1f  %s  %s" % (
                i, name, len(b), nv, cyc,
                ",".join("%s*%d" % kv for kv in mem.most_common(3)),
                " ".join(br)))
    print("# all blocks (static): %.0f VALU cycles" % total)
    print("# opcodes priced without a measurement of their own:")
    for (op, how), n in guesses.most_common():
        print("#   %-22s x%-4d %s (%.2f)" % (op, n, how, price(op)[0]))


if __name__ == "__main__":
    main()
