"""stdin: the output of `python bench.py --leg NAME ...`; prints one short line per variant (name, G units/s, ms per step, kernel ms, fractions,
requests and LF steps per window, parity)."""
import json
import sys

for line in sys.stdin:
    if line.startswith("["):
        for r in json.loads(line):
            print(r["name"], round(r["value"] / 1e9, 2), "G/s  ms", r["ms"], " kernel ms", r["avg_kernel_ms"], " frac", r.get("frac"), " hbm_frac", r.get("hbm_frac"),
                  " traffic/alg", r.get("traffic_over_algorithmic"), " req/window", r.get("requests_per_kmer"), " lf/window", r.get("lf_steps_per_kmer"),
                  " G req/s", r.get("random_requests_per_s"), " positives", r.get("positive_fraction"), " == reference:", r["matches_reference"])
