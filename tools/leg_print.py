"""stdin: the output of `python bench.py --leg NAME ...`; prints one short line per variant (name, G units/s, ms per step, kernel ms, parity)."""
import json
import sys

for line in sys.stdin:
    if line.startswith("["):
        for r in json.loads(line):
            print(r["name"], round(r["value"] / 1e9, 2), r["ms"], r["avg_kernel_ms"], r["matches_reference"])
