#!/usr/bin/env python3
"""Where a kernel's instructions are: static counts per source section from the compiler's own assembly.

    hipcc --offload-arch=gfx950 -std=c++17 -O3 -S --cuda-device-only -gline-tables-only fmsi_amd/csrc/fmsi_dict.hip -o /tmp/dict.s
    python tools/isa_sections.py /tmp/dict.s sk_runs_kernelILb0ELb0ELb0E skdict.inc tools/isa_sections_sk_runs.json

The third argument names the file of the kernel's body; the JSON gives its sections as [name, first line, last line] (lines of that file)
and `helpers` as [name, file, first line, last line].  Every instruction is filed under the section of the LAST kernel-body line the
assembler's .loc directives named before it (an inlined helper's instructions carry the helper's own lines: they stay with the section
that called it) and, separately, under the helper it comes from.  Classes: VALU (v_*), SALU (s_*, without waitcnt / branches / nops),
LDS (ds_*), VMEM (global_* / buffer_* / flat_*), branch, wait.  Static counts: a loop body counts once -- the report says which sections
sit in loops, and the reader multiplies.
"""
import json
import re
import sys
from collections import Counter, defaultdict


def classify(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    asm, kernel, body_file, spec_path = sys.argv[1:5]
    spec = json.load(open(spec_path))
    sections = [(n, a, b) for n, a, b in spec["sections"]]
    helpers = [(n, f, a, b) for n, f, a, b in spec.get("helpers", [])]
    files = {}
    by_section = defaultdict(Counter)
    by_helper = defaultdict(Counter)
    helper_in_section = defaultdict(Counter)
    inside = False
    cur_section, cur_helper = "?", None
    loc_re = re.compile(r"\s*\.loc\s+(\d+)\s+(\d+)")
    file_re = re.compile(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"')
    for line in open(asm, errors="replace"):
        m = file_re.match(line)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if not inside:
            if line.startswith("_Z") and kernel in line and line.rstrip().split(":")[0].endswith("mm") or (line.startswith("_Z") and kernel in line and ":" in line and "@" in line):
                inside = True
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = loc_re.match(line)
        if m:
            f, ln = files.get(int(m.group(1)), "?").rsplit("/", 1)[-1], int(m.group(2))   # (file 0 of a translation unit carries its path)
            cur_helper = None
            if f == body_file:
                hit = [n for n, a, b in sections if a <= ln <= b]
                if hit:
                    cur_section = hit[0]
                    continue
            for n, hf, a, b in helpers:
                if f == hf and a <= ln <= b:
                    cur_helper = n
                    break
            else:
                cur_helper = f"{f}:{ln}" if f != body_file else f"{f}:{ln} (outside the sections)"
            continue
        s = line.strip()
        if not s or s.startswith((".", ";", "#")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c = classify(op)
        by_section[cur_section][c] += 1
        if cur_helper:
            by_helper[cur_helper][c] += 1
            helper_in_section[cur_section][cur_helper] += 1 if c == "VALU" else 0
    cols = ["VALU", "SALU", "LDS", "VMEM", "SMEM", "branch", "wait", "other"]
    print(f"kernel {kernel}: static instruction counts per section of {body_file}")
    print("%-44s" % "section" + "".join("%8s" % c for c in cols))
    tot = Counter()
    for n, a, b in sections + [("?", 0, 0)]:
        if n not in by_section:
            continue
        row = by_section[n]
        tot.update(row)
        print("%-44s" % (f"{n} [{a}-{b}]") + "".join("%8d" % row[c] for c in cols))
        top = helper_in_section[n].most_common(6)
        if top:
            print("    of its VALU, from inlined helpers: " + ", ".join(f"{h} {v}" for h, v in top if v))
    print("%-44s" % "total" + "".join("%8d" % tot[c] for c in cols))
    print("\nby inlined helper (all sections):")
    for h, row in sorted(by_helper.items(), key=lambda kv: -kv[1]["VALU"])[:25]:
        print("  %-58s VALU %5d  SALU %5d  LDS %4d  VMEM %4d" % (h, row["VALU"], row["SALU"], row["LDS"], row["VMEM"]))


if __name__ == "__main__":
    main()
