#!/usr/bin/env python3
"""GPU probe: what query ORDER is worth to the table-less LF walk.  The same 2^27 seeded 31-mers through query_pair_kernel and
query_lockstep_kernel (FMSI_HIP_NO_PREFIX_TABLE) as they come and sorted by value (= by the bases the backward search consumes first)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import fmsi_amd as fm  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_800_000_000
batch = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1 << 27
k = 31
with fm.builder_chain(0):
    ix = fm.DeviceIndex.build_synthetic(n, 12345, k, with_klcp=False, device=0, prefix_len=0)
    ix.enable_pair_layout()
s = torch.cuda.current_stream().cuda_stream
q = torch.empty(batch, dtype=torch.int64, device="cuda")
ix.synthetic_queries_dev(k, 42, 0, batch, q.data_ptr(), s)
torch.cuda.synchronize()
t0 = time.perf_counter()
qs, perm = torch.sort(q)
torch.cuda.synchronize()
print(f"torch.sort of {batch} keys: {(time.perf_counter() - t0) * 1e3:.2f} ms (first call)")
t0 = time.perf_counter()
qs, perm = torch.sort(q)
torch.cuda.synchronize()
print(f"torch.sort of {batch} keys: {(time.perf_counter() - t0) * 1e3:.2f} ms")
out = torch.empty(batch, dtype=torch.uint8, device="cuda")
cnt = torch.zeros(fm.N_COUNTERS, dtype=torch.int64, device="cuda")
ref = None
only_sorted = len(sys.argv) > 3
for name, flags in (("pair", fm.MODE_ALL | fm.KERNEL_PAIR | fm.NO_PREFIX_TABLE), ("lockstep", fm.MODE_ALL | fm.KERNEL_LOCKSTEP | fm.NO_PREFIX_TABLE),
                    ("SORTED", fm.MODE_ALL | fm.KERNEL_SORTED), ("SORTED/or", fm.MODE_OR | fm.KERNEL_SORTED), ("SORTED/strict", fm.MODE_ALL | fm.STRICT | fm.KERNEL_SORTED)):
    for order, buf in (("as generated", q), ("sorted", qs)):
        if name.startswith("SORTED") and order == "sorted":
            continue
        if only_sorted and name != "SORTED":
            continue
        if name != "SORTED" and name.startswith("SORTED"):
            ref = None
        cnt.zero_()
        ix.query_kmers_dev(buf.data_ptr(), batch, k, flags, out.data_ptr(), cnt.data_ptr(), s)
        torch.cuda.synchronize()
        c = cnt.cpu().numpy()
        ix.profile(True)
        ix.profile_read(reset=True)
        for _ in range(2):
            ix.query_kmers_dev(buf.data_ptr(), batch, k, flags, out.data_ptr(), 0, s)
        torch.cuda.synchronize()
        l_n, l_ms = ix.profile_read(reset=True)
        ix.profile(False)
        ms = l_ms / l_n
        got = out if order == "as generated" else torch.empty_like(out).scatter_(0, perm, out)
        if ref is None:
            ref = got.clone()
        same = bool(torch.equal(got, ref))
        print(f"{name:9s} {order:13s} {ms:8.3f} ms  {batch / ms / 1e6:7.3f} G k-mers/s  steps/kmer {c[0] / batch:.2f} block reads/kmer {c[1] / batch:.2f}  same answers {same}")
