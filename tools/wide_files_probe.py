"""An index of more than 2^32 rows through its files: built on the device, written (fmsi_hip_index_write_files), loaded again (the host
path) -- layout planes, scalars and answers of the two handles compared.  python tools/wide_files_probe.py [n]"""
import os
import shutil
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmsi_amd as fm  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_320_000_000
k = 31
d = "/dev/shm/fmsi_wide_probe"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)
try:
    fm.lib()
    with fm.builder_chain(0):
        a = fm.DeviceIndex.build_synthetic(n, 99, k, with_klcp=True, mask_permille=700)
    t0 = time.time()
    a.write_files(d + "/w")
    print("write_files", round(time.time() - t0, 1), "s", {p: os.path.getsize(f"{d}/w.fmsi.{p}") for p in ("ac_gt", "ac", "gt", "mask", "klcp", "misc")}, flush=True)
    t0 = time.time()
    b = fm.DeviceIndex.from_files(d + "/w")
    print("from_files", round(time.time() - t0, 1), "s", flush=True)
    ia, ib = a.info, b.info
    print("n", ia.n, ib.n, "dollar", ia.dollar_position, ib.dollar_position, "counts", list(ia.counts), list(ib.counts), "rows64", ia.rows64, ib.rows64, flush=True)
    ba, ma = a.download_layout()
    bb, mb = b.download_layout()
    print("blocks", ba.shape, bb.shape, "mrank", ma.shape, mb.shape, flush=True)
    for lo, name in ((0, "counts"), (4, "hi plane"), (8, "lo plane"), (12, "mask plane")):
        diff = np.nonzero((ba[:, lo:lo + 4] != bb[:, lo:lo + 4]).any(axis=1))[0]
        print(name, "differing blocks", diff.size, diff[:5], flush=True)
    dm = np.nonzero(ma != mb[: ma.size])[0]
    print("mask ranks differing", dm.size, dm[:5], flush=True)
    import torch
    nq = 1 << 18
    s = torch.cuda.current_stream().cuda_stream
    d_k = torch.empty(nq, dtype=torch.int64, device="cuda")
    a.synthetic_queries_dev(k, 42, 0, nq, d_k.data_ptr(), s)
    torch.cuda.synchronize()
    kmers = d_k.cpu().numpy().view(np.uint64)
    ra = a.query_kmers(kmers, k, fm.MODE_OR | fm.STRICT)
    rb = b.query_kmers(kmers, k, fm.MODE_OR | fm.STRICT)
    print("answers differing", int((ra != rb).sum()), "present in a", int(((ra & 3) == 2).sum() + ((ra >> 2) == 2).sum()), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
