"""Rates of the chained-lookup kernel on all-positive and all-random reads (python tools/chain_probe.py [n]): the benchmark's mix is half and half."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import fmsi_amd as fm

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_800_000_000
k, L = 31, 10000
idx = fm.DeviceIndex.build_synthetic(n, 12345, k)
idx.enable_dict(chain=True)
n_reads = (1 << 27) // (L - k + 1)
words = n_reads * L // 32 + 2
text = torch.from_numpy(idx.download_text(n)[:words].view(np.int64).copy()).cuda()      # reads = consecutive pieces of the superstring
rnd = torch.randint(-2**62, 2**62, (words,), dtype=torch.int64, device="cuda")
run_start = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
run_windows = np.full(n_reads, L - k + 1, dtype=np.uint32)
nw = int(run_windows.sum())
out = torch.zeros(fm.packed_words(nw, 0), dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for name, buf in (("reads from the text", text), ("random reads", rnd)):
    for chain_env in ("chain", "plain"):
        flags = fm.MODE_ALL | fm.OUT_BITS | (fm.NO_PREFIX_TABLE * 0)
        import os
        idx.query_runs_dev(buf.data_ptr(), n_reads * L, run_start, run_windows, k, flags, out.data_ptr(), 0, s)
        torch.cuda.synchronize()
        idx.profile(True); idx.profile_read(reset=True)
        for _ in range(5):
            idx.query_runs_dev(buf.data_ptr(), n_reads * L, run_start, run_windows, k, flags, out.data_ptr(), 0, s)
        torch.cuda.synchronize()
        nl, ms = idx.profile_read(reset=True); idx.profile(False)
        ones = int(fm.unpack_bits(out.cpu().numpy().view(np.uint64), nw, 0).sum())
        print(f"{name}: {ms / nl:.3f} ms per {nw} windows = {nw / (ms / nl) / 1e6:.1f} G windows/s, {ones} present")
        break
