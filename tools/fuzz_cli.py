#!/usr/bin/env python3
"""Differential fuzzing of the `fmsi` CLI against the real reference, for as many seeds as asked for (tests/test_gpu_fuzz.py runs 40):
python tools/fuzz_cli.py --seeds 500 [--first 100000].  Stops at the first mismatch and says which seed."""
import argparse
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO), str(REPO / "oracle"), str(REPO / "tests")]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=100000)
    a = ap.parse_args()
    import fmsi_amd as fm
    import oracle_py
    import test_gpu_fuzz as fz
    oracle_py.build_oracle()
    checker = oracle_py.REF_CLI if oracle_py.ref_available() else oracle_py.ORACLE_CLI
    for seed in range(a.first, a.first + a.seeds):
        with tempfile.TemporaryDirectory() as d:
            fz.one_case(fm, Path(d), seed, checker)
        if (seed - a.first) % 25 == 24:
            print(f"{seed - a.first + 1} seeds OK (checker: {checker.name})", flush=True)
    print(f"all {a.seeds} seeds OK")


if __name__ == "__main__":
    main()
