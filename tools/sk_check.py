"""Super-k-mer dictionary against the k-mer dictionary on seeded reads (GPU): bytes, STRICT, bit-packed; prints the first mismatches.
    python tools/sk_check.py [n] [read_len] [n_reads]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmsi_amd as fm  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 64
k = 31
fm.lib()
dev = fm.DeviceIndex.build_synthetic(n, 1234, k).enable_dict(superkmer=True)
only = fm.DeviceIndex.build_synthetic(n, 1234, k).enable_dict(superkmer_only=True)   # (what bench.py's c4sk leg runs on)
s = torch.cuda.current_stream().cuda_stream
d_reads = torch.empty((n_reads * L) // 32 + 2, dtype=torch.int64, device="cuda")
dev.synthetic_reads_dev(L, 7, 0, n_reads, d_reads.data_ptr(), 10000, s)
torch.cuda.synchronize()
packed = d_reads.cpu().numpy().view(np.uint64)
starts = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
wins = np.full(n_reads, L - k + 1, dtype=np.uint32)
nw = int(wins.sum())


def explain(d, a, b):
    """the mismatching windows' m-mer scores (sk_score of fmsi_amd/csrc/skdict.inc restated): where the ties are"""
    m, bits = 21, 19
    wpr = L - k + 1
    codes = ((packed[:, None] >> (2 * np.arange(32, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8).reshape(-1)
    shown = 0
    last = -10
    for q in d:
        if q == last + 1:
            last = q
            continue
        last = q
        r, off = divmod(int(q), wpr)
        base = r * L + off
        win = codes[base - 12:base + k + 12]
        sc = []
        for p in range(-12, k - m + 1 + 12):
            x = win[p + 12:p + 12 + m]
            xf = sum(int(c) << (2 * i) for i, c in enumerate(x))
            xr = sum((3 - int(c)) << (2 * (m - 1 - i)) for i, c in enumerate(x))
            xc = min(xf, xr)
            lo, hi = xc & ((1 << m) - 1), xc >> m
            sc.append((((hi * 0xC2B2AF) + (lo * 0x9E3779) + 0x7F4A7C15) & 0xFFFFFFFF) >> (32 - bits))
        print("window", q, "read", r, "offset", off, "sk", a[q], "dict", b[q], "scores of the m-mers from 12 before the window:", sc)
        shown += 1
        if shown >= 4:
            break


bad = 0
for flags in (fm.MODE_OR, fm.MODE_ALL, fm.MODE_OR | fm.STRICT, fm.MODE_ALL | fm.STRICT, fm.MODE_OR | fm.OUT_BITS, fm.MODE_ALL | fm.STRICT | fm.OUT_BITS):
    a = dev.query_runs(packed, n_reads * L, starts, wins, k, flags)
    o = only.query_runs(packed, n_reads * L, starts, wins, k, flags)
    if not np.array_equal(a, o):
        d = np.nonzero(a != o)[0]
        print(hex(flags), "the handle with the super-k-mer dictionary alone differs:", d.size, d[:12])
        bad += d.size
    b = dev.query_runs(packed, n_reads * L, starts, wins, k, flags | fm.NO_SUPERKMER)
    if flags & fm.OUT_BITS:
        a, b = fm.unpack_bits(a, nw, flags), fm.unpack_bits(b, nw, flags)
    d = np.nonzero(a != b)[0]
    print(hex(flags), "windows", nw, "mismatches", d.size, d[:12], (d[:12] % (L - k + 1)) if d.size else "")
    bad += d.size
    if d.size and flags == (fm.MODE_ALL | fm.STRICT):
        explain(d, a, b)
sys.exit(1 if bad else 0)
