#!/usr/bin/env python3
"""md5s of an index built by the REAL reference at scale (build container only: needs oracle/_ref/fmsi_ref).

The seeded synthetic masked superstring of the product's generator (fmsi_amd.synthetic_text: the host restatement of
fmsi_hip_index_build_synthetic_spec) -- 10^8 bases, k = 31, 60 % of the positions ON -- is written as FASTA and indexed by the
reference's own construct() + dump_index() (qsufsort, sdsl rrr_vector<63>).  The 60 MB of index files do not travel; their md5s do
(tests/golden/idx/md5_scale.json), and tests/test_gpu_scale.py regenerates the same text on the GPU box, builds the index with
the product's on-device builder, writes the files and compares: byte-identical files = the index the reference built.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
REF_CLI = REPO / "oracle" / "_ref" / "fmsi_ref"

SPEC = {"len": 100_000_000, "seed": 2024, "k": 31, "mask_permille": 600}
# `make_scale_md5.py 1e9`: the same text at 10^9 bases (the reference's qsufsort: ~16 min and 17 GB here) -> md5_scale_1e9.json
OUT = "md5_scale.json"
if len(sys.argv) > 1:
    SPEC = dict(SPEC, len=int(float(sys.argv[1])))
    OUT = f"md5_scale_{sys.argv[1]}.json"


def main():
    import fmsi_amd as fm
    if not REF_CLI.exists():
        sys.exit("build oracle/_ref first: make -C oracle -f Makefile.ref")
    with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
        fa = os.path.join(d, "scale.fa")
        with open(fa, "wb") as f:
            f.write(b">synthetic\n")
            f.write(fm.synthetic_text(SPEC["len"], SPEC["seed"], SPEC["k"], SPEC["mask_permille"]))
            f.write(b"\n")
        t0 = time.time()
        subprocess.run([str(REF_CLI), "index", "-k", str(SPEC["k"]), fa], check=True)
        secs = time.time() - t0
        out = {"spec": SPEC, "reference_index_seconds": round(secs, 1), "md5": {}, "bytes": {}}
        for part in ("ac_gt", "ac", "gt", "mask", "klcp", "misc"):
            p = f"{fa}.fmsi.{part}"
            data = open(p, "rb").read()
            out["md5"][part] = hashlib.md5(data).hexdigest()
            out["bytes"][part] = len(data)
        out["misc"] = open(f"{fa}.fmsi.misc").read().split()
    (HERE / "idx" / OUT).write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
