#!/usr/bin/env python3
"""Golden outputs of `fmsi query -f and|xor|r-s` (f-masked superstrings: query_mode::general, reference src/fms_index.h:310-327,
src/functions.h) -- made by the REAL reference (oracle/_ref/fmsi_ref, built from /root/reference by oracle/Makefile.ref) on the
query files make_golden.py wrote; run in the build container.  Writes tests/golden/out/<index>.<queries>.fn_<function>.txt and
checks the harness against the reference's own committed golden for `-f xor` (tests/testfiles/result_b_complements_xor.txt,
copied as data to ref_data/)."""
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")
REF_CLI = HERE.parent.parent / "oracle" / "_ref" / "fmsi_ref"
FUNCTIONS = ["and", "xor", "0-0", "1-1", "2-3", "1-1000"]
QUERY_FILES = {"integration_a": ["queries", "edge"], "integration_b": ["queries", "edge"], "sars_k13": ["rand"], "sars_k31": ["rand"],
               "mixed_k11": ["rand"], "repeats_k40": ["rand"], "repeats_k64": ["rand"]}


def main():
    if not REF_CLI.exists():
        sys.exit("build oracle/_ref first: make -C oracle -f Makefile.ref")
    shutil.copy(REF / "tests" / "testfiles" / "result_b_complements_xor.txt", HERE / "ref_data" / "result_b_complements_xor.txt")
    (HERE / "ref_data" / "result_b_complements_xor.txt").chmod(0o644)
    for name, kinds in QUERY_FILES.items():
        for kind in kinds:
            qf = {"queries": HERE / "ref_data" / "queries.txt", "edge": HERE / "edge_queries.fa", "rand": HERE / f"rand_{name}.fa"}[kind]
            for fn in FUNCTIONS:
                res = subprocess.run([str(REF_CLI), "query", "-f", fn, "-q", str(qf), str(HERE / "idx" / name)], capture_output=True)
                assert res.returncode == 0, res.stderr.decode()
                (HERE / "out" / f"{name}.{kind}.fn_{fn}.txt").write_bytes(res.stdout)
    assert (HERE / "out" / "integration_b.queries.fn_xor.txt").read_bytes() == (HERE / "ref_data" / "result_b_complements_xor.txt").read_bytes()
    print("written:", len(FUNCTIONS) * sum(len(v) for v in QUERY_FILES.values()), "files")


if __name__ == "__main__":
    main()
