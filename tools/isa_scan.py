#!/usr/bin/env python
"""Static look at the gfx950 ISA of the library's kernels (no GPU needed):

    isa_scan.py scan [file.hip ...]      per kernel: registers, scratch, and the
                                         instruction kinds that usually mean a
                                         pointer lost its address space (flat_*)
                                         or registers ran out (scratch_*)
    isa_scan.py diff <git-rev> <file.hip> [name-filter]
                                         which kernels of the file changed
                                         against the revision (labels and
                                         comments ignored) -- a change that is
                                         meant to touch one kernel must leave
                                         the others' instruction streams alone

Round 4 found the chunk launch's per-frame table being read by VECTOR loads
(and its images by flat loads) this way -- 526 k -> 644 k frames/s at 8 ranks
once they were scalar / global.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "open3d_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "--cuda-device-only", "-S"]


def compile_to_asm(src, out):
    r = subprocess.run([HIPCC] + FLAGS + [src, "-o", out],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-2000:]))
    return out


def kernels(asm_text):
    """name -> {"body": normalised instruction lines, "meta": {...}}"""
    out = {}
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n", asm_text, re.M):
        name = m.group(1)
        end = asm_text.find(".Lfunc_end", m.end())
        if end < 0:
            continue
        lines = []
        for ln in asm_text[m.end():end].splitlines():
            ln = ln.split(";")[0].strip()
            if not ln or (ln.startswith(".") and not ln.startswith(".LBB")):
                continue
            lines.append(re.sub(r"\.LBB\d+_\d+", ".L", ln))
        out[name] = {"body": lines, "meta": {}}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel",
                         asm_text, re.S):
        if m.group(1) not in out:
            continue
        meta = out[m.group(1)]["meta"]
        for key in ("next_free_vgpr", "next_free_sgpr",
                    "private_segment_fixed_size", "group_segment_fixed_size"):
            k = re.search(r"\.amdhsa_" + key + r"\s+(\d+)", m.group(2))
            if k:
                meta[key] = int(k.group(1))
    return {k: v for k, v in out.items() if v["meta"]}  # kernels only


def short(name):
    name = re.sub(r"^_ZN5o3dmi12_GLOBAL__N_1\d+", "", name)
    return re.sub(r"^_ZN5o3dmi\d+", "", name)


def scan_text(asm_text):
    rows = []
    for name, k in sorted(kernels(asm_text).items()):
        kinds = collections.Counter()
        for ln in k["body"]:
            op = ln.split()[0]
            if op.startswith(("flat_", "scratch_")):
                kinds[op] += 1
        rows.append({"kernel": short(name), "lines": len(k["body"]),
                     "vgpr": k["meta"].get("next_free_vgpr"),
                     "sgpr": k["meta"].get("next_free_sgpr"),
                     "scratch_bytes": k["meta"].get(
                             "private_segment_fixed_size", 0),
                     "lds_bytes": k["meta"].get("group_segment_fixed_size", 0),
                     "flat": sum(v for o, v in kinds.items()
                                 if o.startswith("flat_")),
                     "scratch_ops": sum(v for o, v in kinds.items()
                                        if o.startswith("scratch_"))})
    return rows


def diff_texts(old_text, new_text, name_filter=""):
    a, b = kernels(old_text), kernels(new_text)
    res = {"same": [], "changed": [], "only_old": [], "only_new": []}
    for n in sorted(set(a) | set(b)):
        if name_filter and name_filter not in n:
            continue
        if n not in b:
            res["only_old"].append(short(n))
        elif n not in a:
            res["only_new"].append(short(n))
        elif a[n]["body"] == b[n]["body"]:
            res["same"].append(short(n))
        else:
            # register renames aside: compare the opcode sequences too
            ops_a = [ln.split()[0] for ln in a[n]["body"]]
            ops_b = [ln.split()[0] for ln in b[n]["body"]]
            res["changed"].append({
                "kernel": short(n), "lines_old": len(ops_a),
                "lines_new": len(ops_b),
                "same_opcode_multiset":
                    collections.Counter(ops_a) == collections.Counter(ops_b)})
    return res


def main(argv):
    if len(argv) >= 2 and argv[1] == "scan":
        files = argv[2:] or sorted(
            os.path.join(CSRC, f) for f in os.listdir(CSRC)
            if f.endswith(".hip"))
        with tempfile.TemporaryDirectory() as d:
            for src in files:
                asm = compile_to_asm(src, os.path.join(d, "k.s"))
                for r in scan_text(open(asm).read()):
                    flag = "  <--" if r["flat"] or r["scratch_ops"] else ""
                    print("%-16s %-74s vgpr %3s sgpr %3s scratch %3d B flat %2d "
                          "scratch-ops %2d%s"
                          % (os.path.basename(src), r["kernel"][:74], r["vgpr"],
                             r["sgpr"], r["scratch_bytes"], r["flat"],
                             r["scratch_ops"], flag))
        return 0
    if len(argv) >= 4 and argv[1] == "diff":
        rev, src = argv[2], os.path.abspath(argv[3])
        rel = os.path.relpath(src, ROOT)
        old = subprocess.run(["git", "-C", ROOT, "show", "%s:%s" % (rev, rel)],
                             capture_output=True, text=True, check=True).stdout
        # beside the current file, so that its includes resolve
        tmp_src = os.path.join(os.path.dirname(src), "_isa_old_" +
                               os.path.basename(src))
        with tempfile.TemporaryDirectory() as d:
            try:
                with open(tmp_src, "w") as f:
                    f.write(old)
                old_asm = open(compile_to_asm(
                    tmp_src, os.path.join(d, "old.s"))).read()
            finally:
                if os.path.exists(tmp_src):
                    os.remove(tmp_src)
            new_asm = open(compile_to_asm(src, os.path.join(d, "new.s"))).read()
        res = diff_texts(old_asm, new_asm, argv[4] if len(argv) > 4 else "")
        print("same: %d kernels" % len(res["same"]))
        for c in res["changed"]:
            print("changed: %-80s %d -> %d lines%s"
                  % (c["kernel"][:80], c["lines_old"], c["lines_new"],
                     " (same opcodes: registers / order only)"
                     if c["same_opcode_multiset"] else ""))
        for k in ("only_old", "only_new"):
            for n in res[k]:
                print("%s: %s" % (k, n))
        return 0
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(main(sys.argv))
