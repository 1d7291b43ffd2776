#!/usr/bin/env python3
"""`fmsi index` and `fmsi export` as whole processes at scale: a synthetic masked superstring (mixed mask) is written as a FASTA file, indexed
by the CLI, and the six files are compared with what the library writes from the handle the superstring came from; then exported back.
    python tools/index_cli_bench.py [--n 1000000000] [--k 31]      -> one JSON line"""
import argparse
import hashlib
import json
import subprocess
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def md5(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        while b := f.read(1 << 24):
            h.update(b)
    return h.hexdigest()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=1e9)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--dir", default="/dev/shm/fmsi_index_cli")
    ap.add_argument("--gzip-level", type=int, default=0,
                    help="also: the superstring as one gzip member of this zlib level through `fmsi index` -- inflated by several threads (pgunzip.h) -- and with "
                         "FMSI_HIP_NO_PARALLEL_GUNZIP=1 (zlib on one thread, as the reference reads it): seconds and the files' identity")
    a = ap.parse_args()
    import fmsi_amd as fm
    work = Path(a.dir)
    work.mkdir(parents=True, exist_ok=True)
    n, k = int(a.n), a.k
    dev = fm.DeviceIndex.build_synthetic(n, 99, k, with_klcp=True, mask_permille=700)
    dev.write_files(str(work / "want"))
    ms = dev.export()
    dev.close()
    fm.trim_scratch(0)
    fa = work / "ms.fa"
    with open(fa, "wb") as f:
        f.write(b">ms\n" + ms + b"\n")
    del ms
    time.sleep(3)   # (the driver wipes the device memory this process just released; a process that starts meanwhile waits for it in its first hipMalloc)
    t0 = time.time()
    import os
    p = subprocess.run([str(fm.CLI_PATH), "index", "-k", str(k), str(fa)], capture_output=True, env={**os.environ, "FMSI_HIP_VERBOSE": "1"})
    t_index = time.time() - t0
    assert p.returncode == 0, p.stderr.decode()[-500:]
    phases = {l.split("]")[1].strip().rsplit(None, 2)[0]: float(l.split()[-2]) for l in p.stderr.decode().splitlines() if l.startswith("[fmsi index]")}
    same = {part: md5(f"{fa}.fmsi.{part}") == md5(work / f"want.fmsi.{part}") for part in ("ac_gt", "ac", "gt", "mask", "klcp", "misc")}
    t0 = time.time()
    with open(work / "back.fa", "wb") as out:
        p = subprocess.run([str(fm.CLI_PATH), "export", str(fa)], stdout=out, stderr=subprocess.PIPE)
    t_export = time.time() - t0
    assert p.returncode == 0, p.stderr.decode()[-500:]
    with open(work / "back.fa", "rb") as f:
        f.readline()
        back_md5 = hashlib.md5(f.read()[:-1]).hexdigest()
    with open(fa, "rb") as f:
        f.readline()
        ms_md5 = hashlib.md5(f.read()[:-1]).hexdigest()
    gz_report = None
    if a.gzip_level:
        import zlib
        t0 = time.time()
        gz = work / "msz.fa.gz"
        with open(fa, "rb") as f, open(gz, "wb") as g:
            c = zlib.compressobj(a.gzip_level, zlib.DEFLATED, 31)
            while chunk := f.read(1 << 24):
                g.write(c.compress(chunk))
            g.write(c.flush())
        gz_report = {"level": a.gzip_level, "compressed_bytes": gz.stat().st_size, "compress_seconds": round(time.time() - t0, 1)}
        for name, env in (("team", {}), ("zlib_one_thread", {"FMSI_HIP_NO_PARALLEL_GUNZIP": "1"})):
            time.sleep(2)
            t0 = time.time()
            p = subprocess.run([str(fm.CLI_PATH), "index", "-k", str(k), str(gz)], capture_output=True, env={**os.environ, **env, "FMSI_HIP_VERBOSE": "1"})
            dt = time.time() - t0
            assert p.returncode == 0, p.stderr.decode()[-500:]
            ph = {l.split("]")[1].strip().rsplit(None, 2)[0]: float(l.split()[-2]) for l in p.stderr.decode().splitlines() if l.startswith("[fmsi index]")}
            gz_report[name] = {"fmsi_index_seconds": round(dt, 3), "superstring_read_seconds": ph.get("superstring read"),
                               "files_identical": all(md5(f"{gz}.fmsi.{part}") == md5(work / f"want.fmsi.{part}") for part in ("ac_gt", "ac", "gt", "mask", "klcp", "misc"))}
    print(json.dumps({"superstring_bases": n, "k": k, "fmsi_index_seconds": round(t_index, 3), "bases_per_s": round(n / t_index), "gzip": gz_report,
                      "files_identical_to_the_library_writer": same, "fmsi_export_seconds": round(t_export, 3), "export_identical": back_md5 == ms_md5,
                      "index_phases_seconds": phases}))
    for f in work.iterdir():
        f.unlink()
    work.rmdir()


if __name__ == "__main__":
    main()
