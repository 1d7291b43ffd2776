#!/usr/bin/env python3
"""`-f and|xor|r-s` (fmsi_hip_query_runs_fn, fmsi_fn.hip) on resident batches at human scale: windows/s of the kernel, its LF steps and block
reads per window, and the HBM rate those make -- one JSON line; under `rocprofv3 --kernel-trace --stats` the kernel is `query_fn_kernel`.
    python tools/fn_bench.py [--n 2800000000] [--prefix-len 14] [--read-len 10000] [--batch 67108864] [--steps 5]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=2.8e9)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--prefix-len", type=int, default=14)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--batch", type=float, default=2 ** 26)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--subst-ppm", type=int, default=0)
    a = ap.parse_args()
    import torch
    import fmsi_amd as fm

    n, k, L = int(a.n), a.k, a.read_len
    dev = fm.DeviceIndex.build_synthetic(n, 12345, k, prefix_len=a.prefix_len)
    n_reads = max(1, int(a.batch) // (L - k + 1))
    windows = n_reads * (L - k + 1)
    run_start = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    run_windows = np.full(n_reads, L - k + 1, dtype=np.uint32)
    buf = torch.empty(n_reads * L // 32 + 2, dtype=torch.int64, device="cuda")
    out = torch.empty(fm.packed_words(windows, 0), dtype=torch.int64, device="cuda")
    cnt = torch.zeros(fm.N_COUNTERS, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    lib, h = fm.lib(), dev._h

    def launch(fn, counters=False):
        rc = lib.fmsi_hip_query_runs_fn(h, buf.data_ptr(), None, n_reads * L, run_start.ctypes.data, run_windows.ctypes.data, n_reads, windows, k, fn, 1, 1,
                                        fm.OUT_BITS | fm.DEVICE_PTRS, out.data_ptr(), cnt.data_ptr() if counters else None, stream)
        assert rc == 0, lib.fmsi_hip_last_error()

    dev.synthetic_reads_dev(L, 42, 0, n_reads, buf.data_ptr(), a.subst_ppm, stream)
    launch(fm.FN_XOR, counters=True)
    torch.cuda.synchronize()
    c = cnt.cpu().numpy()
    dev.profile(True)
    res = {}
    for name, fn in (("xor", fm.FN_XOR), ("and", fm.FN_AND), ("range 1-1", fm.FN_RANGE)):
        dev.profile_read(reset=True)
        t0 = time.time()
        for s in range(a.steps):
            dev.synthetic_reads_dev(L, 42, (s + 1) * n_reads, n_reads, buf.data_ptr(), a.subst_ppm, stream)
            launch(fn)
        torch.cuda.synchronize()
        wall = time.time() - t0
        launches, ms = dev.profile_read(reset=True)
        res[name] = {"kernel_ms": round(ms / max(1, launches), 3), "windows_per_s": round(windows / (ms / max(1, launches) / 1e3)), "wall_ms_per_step": round(wall / a.steps * 1e3, 3)}
    per = {"lf_steps": c[0] / windows, "block_reads": c[1] / windows, "mask_rank_pairs": c[2] / windows, "table_lookups": c[4] / windows}
    best = res["xor"]
    bytes_per_window = per["block_reads"] * 64 + per["table_lookups"] * 8 + L / 4.0 / (L - k + 1) + 8 + 0.125
    print(json.dumps({"n": n, "k": k, "prefix_len": a.prefix_len, "windows_per_launch": windows, "per_window": {x: round(float(v), 3) for x, v in per.items()},
                      "algorithmic_bytes_per_window": round(float(bytes_per_window), 1), "functions": res,
                      "hbm_rate_TBps": round(float(bytes_per_window) * best["windows_per_s"] / 1e12, 3), "frac_of_8TBps": round(float(bytes_per_window) * best["windows_per_s"] / 8e12, 3)}))


if __name__ == "__main__":
    main()
