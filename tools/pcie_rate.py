#!/usr/bin/env python3
"""The library's host-pointer entry points at human scale, inputs and results in page-locked host memory: what the link makes of the
kernel's rate (bench.py's `value` is measured with inputs resident in HBM).  One JSON line."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main() -> None:
    import torch
    import fmsi_amd as fm
    n, k, batch, L = 2_800_000_000, 31, 1 << 27, 10000
    dev = fm.DeviceIndex.build_synthetic(n, 12345, k).enable_pair_layout()
    dev.enable_dict(chain=True)
    s = torch.cuda.current_stream().cuda_stream
    lib, h = fm.lib(), dev._h
    out = {}
    # isolated k-mers: 8 bytes in, one bit out per k-mer
    d = torch.empty(batch, dtype=torch.int64, device="cuda")
    dev.synthetic_queries_dev(k, 42, 0, batch, d.data_ptr(), s)
    kmers = d.cpu().pin_memory()
    bits = torch.zeros(fm.packed_words(batch, 0), dtype=torch.int64).pin_memory()
    flags = fm.MODE_ALL | fm.KERNEL_DICT | fm.OUT_BITS
    for rep in range(3):
        t0 = time.time()
        rc = lib.fmsi_hip_query_kmers(h, kmers.data_ptr(), batch, k, flags, bits.data_ptr(), None, None)
        assert rc == 0
        dev.sync()
        dt = time.time() - t0
    out["isolated_kmers_host_pointers"] = {"kmers_per_s": round(batch / dt), "ms": round(dt * 1e3, 2), "bytes_in_per_kmer": 8, "h2d_GBps": round(8 * batch / dt / 1e9, 1)}
    # reads as runs: 2 bits per base in, one bit out per window
    n_reads = batch // (L - k + 1)
    windows = n_reads * (L - k + 1)
    dr = torch.empty(n_reads * L // 32 + 2, dtype=torch.int64, device="cuda")
    dev.synthetic_reads_dev(L, 42, 0, n_reads, dr.data_ptr(), 0, s)
    reads = dr.cpu().pin_memory()
    rbits = torch.zeros(fm.packed_words(windows, 0), dtype=torch.int64).pin_memory()
    run_start = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    run_windows = np.full(n_reads, L - k + 1, dtype=np.uint32)
    for rep in range(3):
        t0 = time.time()
        rc = lib.fmsi_hip_query_runs(h, reads.data_ptr(), n_reads * L, run_start.ctypes.data, run_windows.ctypes.data, n_reads, windows, k, fm.MODE_ALL | fm.OUT_BITS,
                                     rbits.data_ptr(), None, None)
        assert rc == 0
        dev.sync()
        dt = time.time() - t0
    out["reads_as_runs_host_pointers"] = {"windows_per_s": round(windows / dt), "ms": round(dt * 1e3, 2), "bytes_in_per_window": round(L / 4 / (L - k + 1), 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
