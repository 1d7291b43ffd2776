#!/usr/bin/env python3
"""The CLI's host pipeline with the device stubbed out (tests/native/cli_pipeline_stub.cpp: answers from a hash of the k-mer), built with
ASan + UBSan, on random query files (every record shape; plain, block-gzip, gzip) in random modes: the output must not depend on batch
size, team size, number of handles, reader, or PEXT/PDEP availability -- and the sanitizers must stay quiet.  Runs without a GPU.
    python tools/stub_pipeline_fuzz.py [cases=150]"""
import gzip
import os
import random
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "tests"), str(REPO / "oracle")]
from conftest import GOLDEN, bgzf_bytes, messy_fasta   # noqa: E402

MODES = [["query"], ["query", "-O"], ["query", "-S", "-O"], ["lookup"], ["lookup", "-S"], ["query", "-f", "xor"]]


def main() -> None:
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    src = REPO / "fmsi_amd" / "csrc"
    with tempfile.TemporaryDirectory() as d:
        exe = Path(d) / "cli_stub"
        res = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", f"-I{REPO / 'include'}",
                              str(REPO / "tests" / "native" / "cli_pipeline_stub.cpp"), *(str(src / f) for f in ("query_cli.cpp", "fastx.cpp", "pgunzip.cpp", "trace.cpp", "host_index.cpp",
                                                                                                                "layout.cpp", "index_writer.cpp")),
                              "-o", str(exe), "-lz", "-lpthread", "-ldl"], capture_output=True, text=True)
        if res.returncode != 0:
            sys.exit(res.stderr[-3000:])
        ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13") if not l.startswith(">"))
        prefix = str(GOLDEN / "idx" / "sars_k13")
        env = {**os.environ, "ASAN_OPTIONS": "detect_leaks=0"}
        q = Path(d) / "q"
        bad = 0
        for seed in range(cases):
            rng = random.Random(seed)
            data = messy_fasta(ms, rng, fastq_at=rng.choice([None, None, 100, 0]))
            if rng.random() < 0.5:
                data = b">first\nACGTTGCA\n" + data
            pack = rng.choice(["plain", "bgzf", "gzip"])
            q.write_bytes(data if pack == "plain" else (bgzf_bytes(data, block=rng.choice([200, 1500, 0xFF00])) if pack == "bgzf" else gzip.compress(data)))
            mode = rng.choice(MODES)
            outs = set()
            for opts, extra in (([], {}),
                                (["--batch", str(rng.choice([1024, 2500, 9000])), "--threads", str(rng.choice([1, 2, 3, 5])), "--devices", rng.choice(["0", "0,0", "0,0,0"])],
                                 {"FMSI_HIP_READER_MIN_BYTES": "1"}),
                                ([], {"FMSI_HIP_SERIAL_READER": "1"}),
                                (["--batch", "1500"], {"FMSI_HIP_READER_MIN_BYTES": "1", "FMSI_HIP_NO_BMI2": "1"})):
                r = subprocess.run([str(exe), *mode, *opts, "-q", str(q), prefix], capture_output=True, env={**env, **extra})
                if r.returncode != 0:
                    bad += 1
                    print("FAIL", seed, mode, opts, extra, r.stderr.decode()[-600:])
                    break
                outs.add(r.stdout)
            if len(outs) > 1:
                bad += 1
                print("DIFF", seed, mode, pack)
        print(f"stub pipeline fuzz, {cases} cases:", "all consistent, no sanitizer report" if not bad else f"{bad} problems")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
