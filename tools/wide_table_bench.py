"""The prefix table of a handle with 64-bit rows: 2^24 seeded 31-mers on a 4.5 x 10^9-row index with and without it (GPU).
    python tools/wide_table_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmsi_amd as fm  # noqa: E402

n, k, nq = 4_500_000_000, 31, 1 << 24
fm.lib()
with fm.builder_chain(0):
    t0 = time.time()
    dev = fm.DeviceIndex.build_synthetic(n, 99, k, prefix_len=14)
    print("build incl. the 14-deep table", round(time.time() - t0, 2), "s; handle", round(dev.info.device_bytes / 1e9, 2), "GB", flush=True)
s = torch.cuda.current_stream().cuda_stream
d_k = torch.empty(nq, dtype=torch.int64, device="cuda")
d_o = torch.empty(nq, dtype=torch.uint8, device="cuda")
dev.synthetic_queries_dev(k, 42, 0, nq, d_k.data_ptr(), s)
for name, extra in (("with the table", 0), ("without", fm.NO_PREFIX_TABLE)):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        dev.query_kmers_dev(d_k.data_ptr(), nq, k, fm.MODE_ALL | extra, d_o.data_ptr(), stream=s)
        torch.cuda.synchronize()
        dt = time.time() - t0
    print(name, round(dt * 1e3, 2), "ms", round(nq / dt / 1e9, 3), "G k-mers/s", flush=True)
