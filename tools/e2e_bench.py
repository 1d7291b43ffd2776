#!/usr/bin/env python3
"""End-to-end `fmsi query` wall time: index load + FASTA parse + device + print.

Builds a synthetic masked superstring index on the GPU, writes it in the reference's .fmsi.* format, draws reads on
the device (half substrings of the superstring, half uniform), writes them as FASTA into /dev/shm, then runs the
product CLI in the requested modes with --stats and reports windows/s (whole process and without the index load).
A sample of the reads also goes through the real reference CLI (oracle/_ref, if it travelled) to (a) check that the
outputs are byte-identical and (b) put its single-threaded rate beside the product's.

    python tools/e2e_bench.py [--n 100000000] [--reads 20000] [--read-len 10000] [--k 31] [--modes or,all,all_fast,all_S]
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

MODES = {
    "or": ["query"],
    "all": ["query", "-O"],
    "all_fast": ["query", "-O", "--fast"],
    "or_S": ["query", "-S"],
    "all_S": ["query", "-S", "-O"],
    "all_S_fast": ["query", "-S", "-O", "--fast"],
    "lookup": ["lookup"],
    "lookup_S": ["lookup", "-S"],
    # f-masked superstrings (query_mode::general): f(ones, total) over both strands, no replay
    "xor": ["query", "-f", "xor"],
    "and": ["query", "-f", "and"],
    "range": ["query", "-f", "1-1"],
    # the same with the k-mer dictionary (--tables full), for comparison; and two handles on the one device (--devices 0,0)
    "or_full": ["query", "--tables", "full"],
    "all_full": ["query", "-O", "--tables", "full"],
    "all_S_full": ["query", "-S", "-O", "--tables", "full"],
    "or_compact": ["query", "--tables", "compact"],       # the super-k-mer dictionary alone
    "all_S_compact": ["query", "-S", "-O", "--tables", "compact"],
    "or_bg": ["query", "--tables", "background"],
    "all_S_bg": ["query", "-S", "-O", "--tables", "background"],
    "or_2dev": ["query", "--devices", "0,0"],
    "all_2dev": ["query", "-O", "--devices", "0,0"],
}


def write_fasta(path: Path, packed: np.ndarray, n_reads: int, read_len: int, first: int) -> None:
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "ab") as f:
        step = max(1, (64 << 20) // read_len)
        for r0 in range(0, n_reads, step):
            r1 = min(n_reads, r0 + step)
            b0, b1 = r0 * read_len, r1 * read_len
            w0, w1 = b0 // 32, (b1 + 31) // 32
            codes = ((packed[w0:w1, None] >> (2 * np.arange(32, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8).reshape(-1)
            codes = codes[b0 - w0 * 32: b0 - w0 * 32 + (b1 - b0)]
            seqs = lut[codes].reshape(r1 - r0, read_len)
            out = bytearray()
            for i in range(r1 - r0):
                out += b">r%d\n" % (first + r0 + i)
                out += seqs[i].tobytes()
                out += b"\n"
            f.write(out)


def write_kmer_fasta(path: Path, kmers: np.ndarray, k: int, first: int) -> None:
    """One k-mer per record (BASELINE configs 2 / 3 / 5's shape: `>INDEX\nKMER\n`), names = the record's number in decimal; rows of one
    name width are built as one 2-d byte array (no Python loop over records)."""
    lut4 = np.frombuffer(b"ACGT", dtype=np.uint8)[(np.arange(256)[:, None] >> (2 * np.arange(4))) & 3].astype(np.uint8)   # byte -> its four bases
    n = len(kmers)
    with open(path, "ab") as f:
        lo = first
        while lo < first + n:
            d = len(str(lo))
            hi = min(first + n, 10 ** d)
            for c0 in range(lo, hi, 4_000_000):
                c1 = min(hi, c0 + 4_000_000)
                m = c1 - c0
                rows = np.empty((m, d + k + 3), dtype=np.uint8)
                rows[:, 0] = ord(">")
                idx = np.arange(c0, c1, dtype=np.int64)
                for j in range(d):
                    rows[:, d - j] = (idx // 10 ** j % 10 + 48).astype(np.uint8)
                rows[:, d + 1] = 10
                rows[:, d + 2:d + 2 + k] = lut4[kmers[c0 - first:c1 - first].view(np.uint8).reshape(m, 8)].reshape(m, 32)[:, :k]
                rows[:, -1] = 10
                f.write(rows.tobytes())
            lo = hi


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--kmer-records", type=int, default=0,
                    help="after the reads: a second query file of this many records of ONE k-mer each (half present, drawn on the device), through --kmer-modes")
    ap.add_argument("--kmer-modes", default="or,all")
    ap.add_argument("--keep", action="store_true", help="leave the index files and the query files in --dir (further runs by hand)")
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--reads", type=int, default=20_000)
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--modes", default="or,all,all_fast,all_S,all_S_fast")
    ap.add_argument("--ref-reads", type=int, default=40, help="reads of the sample that also go through the reference CLI")
    ap.add_argument("--dir", default="/dev/shm/fmsi_e2e")
    ap.add_argument("--out", default=None, help="write the JSON report here")
    ap.add_argument("--compress", choices=["none", "bgzf", "gzip"], default="none",
                    help="query file as block-gzip (members of 64 KiB, level 1: read by several threads) or as one ordinary gzip member")
    ap.add_argument("--compress-level", type=int, default=1, help="zlib's level for --compress gzip / --gz-modes (gzip's own default is 6)")
    ap.add_argument("--gz-modes", default="",
                    help="after the modes on the plain file: the reads file as ONE ordinary gzip member (zlib, --compress-level) through these modes -- the "
                         "team's inflate, pgunzip.h; a mode with FMSI_HIP_NO_PARALLEL_GUNZIP=1+ in front is zlib on one thread, as the reference reads it")
    a = ap.parse_args()

    import torch
    import fmsi_amd as fm

    work = Path(a.dir)
    work.mkdir(parents=True, exist_ok=True)
    prefix = str(work / "idx")
    report = {"n": a.n, "k": a.k, "reads": a.reads, "read_len": a.read_len, "modes": {}}

    t0 = time.time()
    dev = fm.DeviceIndex.build_synthetic(a.n, 12345, a.k, with_klcp=True)
    torch.cuda.synchronize()
    report["build_seconds"] = round(time.time() - t0, 3)
    t0 = time.time()
    dev.write_files(prefix)
    report["write_seconds"] = round(time.time() - t0, 3)
    report["index_bytes"] = sum(os.path.getsize(f"{prefix}.fmsi.{p}") for p in ("ac_gt", "ac", "gt", "mask", "klcp", "misc"))

    fasta = work / "reads.fa"
    fasta.unlink(missing_ok=True)
    fasta.touch()
    t0 = time.time()
    step = max(1, (1 << 30) // a.read_len)           # <= 1 G bases per device draw
    for r0 in range(0, a.reads, step):
        nr = min(step, a.reads - r0)
        words = (nr * a.read_len) // 32 + 2
        d = torch.empty(words, dtype=torch.int64, device="cuda")
        dev.synthetic_reads_dev(a.read_len, 42, r0, nr, d.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        write_fasta(fasta, d.cpu().numpy().view(np.uint64), nr, a.read_len, r0)
        del d
    report["fasta_bytes"] = fasta.stat().st_size
    report["fasta_seconds"] = round(time.time() - t0, 3)
    if a.compress != "none":
        import struct
        import zlib
        t0 = time.time()
        gz = work / "reads.fa.gz"
        with open(fasta, "rb") as f, open(gz, "wb") as g:
            if a.compress == "gzip":
                c = zlib.compressobj(a.compress_level, zlib.DEFLATED, 31)
                while chunk := f.read(1 << 24):
                    g.write(c.compress(chunk))
                g.write(c.flush())
            else:
                while True:
                    chunk = f.read(0xFF00)
                    c = zlib.compressobj(1, zlib.DEFLATED, -15)
                    raw = c.compress(chunk) + c.flush()
                    g.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 12 + 6 + len(raw) + 8 - 1) + raw +
                            struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
                    if not chunk:
                        break
        fasta.unlink()
        fasta = gz
        report["compressed_bytes"] = gz.stat().st_size
        report["compress_seconds"] = round(time.time() - t0, 3)
        report["compress"] = a.compress
    kfasta = work / "kmers.fa"
    if a.kmer_records:
        if a.k > 32:
            raise SystemExit("--kmer-records needs k <= 32")
        t0 = time.time()
        kfasta.unlink(missing_ok=True)
        for r0 in range(0, a.kmer_records, 1 << 25):
            nr = min(1 << 25, a.kmer_records - r0)
            d = torch.empty(nr, dtype=torch.int64, device="cuda")
            dev.synthetic_queries_dev(a.k, 42, r0, nr, d.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            write_kmer_fasta(kfasta, d.cpu().numpy().view(np.uint64), a.k, r0)
            del d
        report["kmer_fasta_bytes"] = kfasta.stat().st_size
        report["kmer_fasta_seconds"] = round(time.time() - t0, 3)
    dev.close()
    del dev
    torch.cuda.empty_cache()

    # sample for the reference
    sample = work / "sample.fa"
    import gzip as _gzip
    with (_gzip.open(fasta, "rb") if a.compress != "none" else open(fasta, "rb")) as f, open(sample, "wb") as g:
        for _ in range(2 * a.ref_reads):
            g.write(f.readline())
    ref_cli = REPO / "oracle" / "_ref" / "fmsi_ref"
    windows_per_read = max(0, a.read_len - a.k + 1)

    def run_modes(mode_list, fasta, sample, units_per_record, n_ref_records, dest):
        for mode in mode_list:
            # "or+--threads+8": mode `or` with extra arguments; "NAME=value+or": with an environment variable
            parts = mode.split("+")
            env = dict(os.environ)
            while "=" in parts[0] and not parts[0].startswith("-"):
                key, _, val = parts.pop(0).partition("=")
                env[key] = val
            args = MODES[parts[0]] + parts[1:]
            outp = work / f"out_{parts[0]}.txt"
            # the checkers first -- the real reference and this binary on the sample, compared below -- and a pause: what is timed is a run on a
            # quiet device.  (The binary leaves the moment its output is written and lets a forked child tear the device context down; a run
            # started within that quarter of a second shares the driver with the dying one, and load times of 0.19 s were 0.3-0.4 s.)
            ref_args = [x for x in MODES[parts[0]] if x not in ("--fast", "--tables", "full", "background", "--devices", "0,0")]
            check = ref_cli.exists() and n_ref_records
            if check:
                tr = time.time()
                r = subprocess.run([str(ref_cli), *ref_args, "-k", str(a.k), "-q", str(sample), prefix], capture_output=True)
                ref_s = time.time() - tr
                g = subprocess.run([str(fm.CLI_PATH), *args, "-q", str(sample), prefix], capture_output=True, env=env)
            time.sleep(0.75)
            t0 = time.time()
            with open(outp, "wb") as so:
                p = subprocess.run([str(fm.CLI_PATH), *args, "--stats", "-q", str(fasta), prefix], stdout=so, stderr=subprocess.PIPE, env=env)
            t1 = time.time()
            wall = t1 - t0
            entry = {"rc": p.returncode, "wall_seconds": round(wall, 3)}
            for line in p.stderr.decode().splitlines():
                if line.startswith("{"):
                    entry.update(json.loads(line))
                elif "FMSI_HIP_VERBOSE" in env:
                    print("   ", line, flush=True)
            if p.returncode != 0:
                entry["stderr"] = p.stderr.decode()[-400:]
            if "epoch_start" in entry:   # what the process spends outside the command's function: start-up (loader, runtime) and teardown
                entry["startup_seconds"] = round(entry.pop("epoch_start") - t0, 3)
                entry["teardown_seconds"] = round(t1 - entry.pop("epoch_end"), 3)
            w = entry.get("windows", 0)
            if w:
                entry["windows_per_s_wall"] = round(w / wall)
                entry["windows_per_s_after_load"] = round(w / max(1e-9, entry["total_seconds"] - entry["load_seconds"]))
            entry["output_bytes"] = outp.stat().st_size
            if check:   # reference on the sample: identical bytes?
                entry["sample_identical_to_reference"] = bool(r.returncode == 0 and r.stdout == g.stdout)
                entry["reference_windows_per_s"] = round(n_ref_records * units_per_record / ref_s)
                with open(outp, "rb") as f:
                    head = f.read(len(r.stdout))
                entry["full_run_prefix_identical"] = bool(head == r.stdout)
            dest[mode] = entry
            print(mode, json.dumps(entry), flush=True)
            outp.unlink()


    if a.reads:
        run_modes(a.modes.split(","), fasta, sample, windows_per_read, a.ref_reads, report["modes"])
    if a.reads and a.gz_modes and a.compress == "none":
        import zlib
        t0 = time.time()
        gz = work / "reads.fa.gz"
        with open(fasta, "rb") as f, open(gz, "wb") as g:
            c = zlib.compressobj(a.compress_level, zlib.DEFLATED, 31)
            while chunk := f.read(1 << 24):
                g.write(c.compress(chunk))
            g.write(c.flush())
        report["gz"] = {"compressed_bytes": gz.stat().st_size, "compress_seconds": round(time.time() - t0, 2), "level": a.compress_level}
        report["gz_modes"] = {}
        run_modes(a.gz_modes.split(","), gz, sample, windows_per_read, a.ref_reads, report["gz_modes"])
        if not a.keep:
            gz.unlink()
    if a.kmer_records:
        ksample = work / "ksample.fa"
        n_ref = min(a.kmer_records, 20000)
        with open(kfasta, "rb") as f, open(ksample, "wb") as g:
            for _ in range(2 * n_ref):
                g.write(f.readline())
        report["kmer_records"] = a.kmer_records
        report["kmer_modes"] = {}
        run_modes(a.kmer_modes.split(","), kfasta, ksample, 1, n_ref, report["kmer_modes"])
    print(json.dumps(report))
    if a.out:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(a.out).write_text(json.dumps(report, indent=1) + "\n")
    if not a.keep:
        for f in work.iterdir():
            f.unlink()
        work.rmdir()


if __name__ == "__main__":
    main()
