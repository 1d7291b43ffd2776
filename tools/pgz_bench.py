#!/usr/bin/env python3
"""The team's inflate of an ordinary gzip file on its own (fmsi_host_gunzip, pgunzip.h), on this host's cores: text of `--mb` megabytes (reads of
10 kb, or one 31-mer per record), compressed by zlib at the given levels, inflated with 1 ... N threads; zlib's own single-thread rate beside it.
No GPU needed.

    python tools/pgz_bench.py [--mb 512] [--levels 1,6] [--threads 1,4,8,16] [--shape reads|kmers]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=512)
    ap.add_argument("--levels", default="1,6")
    ap.add_argument("--threads", default="1,2,4,8,16")
    ap.add_argument("--shape", choices=["reads", "kmers"], default="reads")
    ap.add_argument("--dir", default="/dev/shm/fmsi_pgz")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import fmsi_amd as fm
    L = fm.lib()
    L.fmsi_host_gunzip.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.fmsi_host_gunzip.restype = C.c_int
    work = Path(a.dir)
    work.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(7)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L_ = 10000 if a.shape == "reads" else 31
    n = (a.mb << 20) // (L_ + 10)
    parts = []
    step = max(1, (1 << 26) // (L_ + 1))
    for r0 in range(0, n, step):
        m = min(step, n - r0)
        seqs = acgt[rng.integers(0, 4, (m, L_), dtype=np.uint8)]
        if a.shape == "reads":
            parts.append(b"".join(b">r%d\n" % (r0 + i) + seqs[i].tobytes() + b"\n" for i in range(m)))
        else:
            names = np.char.add(np.char.add(">", np.arange(r0, r0 + m).astype(str)), "\n").astype("S")
            rows = np.concatenate([seqs, np.full((m, 1), 10, dtype=np.uint8)], axis=1)
            parts.append(b"".join(nm + rw.tobytes() for nm, rw in zip(names.tolist(), rows)))
    text = b"".join(parts)
    report = {"text_bytes": len(text), "shape": a.shape, "cores": len(os.sched_getaffinity(0)), "levels": {}}
    for lvl in [int(x) for x in a.levels.split(",")]:
        t0 = time.time()
        c = zlib.compressobj(lvl, zlib.DEFLATED, 31)
        gz = b"".join([c.compress(text[i:i + (1 << 24)]) for i in range(0, len(text), 1 << 24)] + [c.flush()])
        comp_s = time.time() - t0
        p = work / f"t{lvl}.gz"
        p.write_bytes(gz)
        t0 = time.time()
        ok = zlib.decompressobj(31).decompress(gz) == text
        z_s = time.time() - t0
        entry = {"compressed_bytes": len(gz), "compress_seconds": round(comp_s, 2), "zlib_one_thread_gbs": round(len(text) / z_s / 1e9, 3), "zlib_ok": ok, "threads": {}}
        buf = np.empty(len(text) + 16, dtype=np.uint8)
        buf[:] = 0   # (touched: the timed copies do not fault the pages in)
        for th in [int(x) for x in a.threads.split(",")]:
            best = None
            for _ in range(3):
                nn = C.c_uint64(0)
                st = (C.c_uint64 * 7)()
                t0 = time.time()
                rc = L.fmsi_host_gunzip(str(p).encode(), th, 0, buf.ctypes.data, len(buf), C.byref(nn), st)
                dt = time.time() - t0
                assert rc == 0 and nn.value == len(text), (rc, nn.value)
                best = dt if best is None else min(best, dt)
            same = bytes(buf[:len(text)]) == text
            entry["threads"][th] = {"seconds": round(best, 4), "gbs": round(len(text) / best / 1e9, 3), "identical": same, "rounds": st[0], "chunks": st[1],
                                    "marker_symbols": st[4], "plain_bytes": st[5]}
            print(lvl, th, json.dumps(entry["threads"][th]), flush=True)
        report["levels"][lvl] = entry
        p.unlink()
    print(json.dumps(report))
    if a.out:
        Path(a.out).write_text(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
