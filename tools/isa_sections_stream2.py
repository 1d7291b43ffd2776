"""Section table for tools/isa_sections.py of query_stream2_kernel (fmsi_stream.hip), by the source's own markers -- line numbers move with every
edit:  python tools/isa_sections_stream2.py > /tmp/s2.json"""
import json
import re
import sys

src = open(sys.argv[1] if len(sys.argv) > 1 else "fmsi_amd/csrc/fmsi_stream.hip").read().splitlines()
k0 = next(i for i, l in enumerate(src) if "void query_stream2_kernel" in l) + 1


def at(pattern, start=k0):
    return next(i for i in range(start - 1, len(src)) if re.search(pattern, src[i])) + 1


marks = [("set-up before the loop", k0),
         ("loop top: died searches (window_done + advance)", at(r"^    for \(;;\) \{")),
         ("loop top: pass-1 start, flush_orders", at(r"if \(st == SS_PASS1\) begin_pass1_now\(\);")),
         ("loop top: draw (task cut, loads, flush_task)", at(r"idle lanes draw the wave's next tasks")),
         ("loop top: fresh (k-mer of the window)", at(r"if \(st == SS_FRESH_CERT \|\| st == SS_FRESH_PLAIN\) \{")),
         ("reads: table entry, kLCP words", at(r"---- this round's reads ----")),
         ("reads: cooperative line fetch + LDS staging", at(r"const uint64_t line_mask = __ballot\(need_line\);")),
         ("arith: CERT", at(r"---- arithmetic ----")),
         ("arith: LOOKUP", at(r"\} else if \(st == SS_LOOKUP\) \{")),
         ("arith: LF / EXTLF base", at(r"\} else if \(st == SS_LF\) \{")),
         ("arith: HAVE", at(r"\} else if \(st == SS_HAVE\) \{")),
         ("arith: EXTSCAN", at(r"\} else if \(st == SS_EXTSCAN\) \{")),
         ("arith: LF step + its outcomes", at(r"^        if \(do_lf\) \{")),
         ("after the loop: flush, counters", at(r"^    if \(ORD\) flush_orders\(\);"))]
end = at(r"^template <bool COUNT, bool PTABLE, bool WIDE, bool ORD>")
sections = []
lam0 = at(r"auto kmer_base = ")
for (name, a), nxt in zip(marks, [m[1] for m in marks[1:]] + [end]):
    sections.append([name, a, (lam0 - 1) if name.startswith("set-up") else nxt - 1])


def lam(name, pat):
    a = at(pat)
    b = next(i for i in range(a, len(src)) if src[i].startswith("    };") or (i > a and src[i].startswith("    auto ")))
    return [name, "fmsi_stream.hip", a, b + 1]


helpers = [lam("load_kmer", r"auto load_kmer = "), lam("fresh", r"auto fresh = "), lam("flush_task", r"auto flush_task = "), lam("flush_orders", r"auto flush_orders = "),
           lam("begin_pass1_now", r"auto begin_pass1_now = "), lam("window_done", r"auto window_done = "), lam("advance", r"auto advance = "),
           lam("skip_dead", r"auto skip_dead = "), lam("cert_offset", r"auto cert_offset = "), lam("cert_string", r"auto cert_string = "),
           lam("kmer_base", r"auto kmer_base = "), lam("kmer_table_index", r"auto kmer_table_index = "), lam("shift_fwd/rc", r"auto shift_fwd = "),
           ["spiece_*", "fmsi_stream.hip", at(r"struct SPiece", 1), at(r"UNI: reads of one length", 1)],
           ["task_base/task_kmer", "fmsi_stream.hip", at(r"uint32_t task_base\(", 1), at(r"uint64_t task_kmer\(", 1) + 12],
           ["decode_table_entry", "device_fm.h", 1, 100000]]
print(json.dumps({"sections": sections, "helpers": helpers}, indent=1))
