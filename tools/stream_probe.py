"""The kLCP kernel on the benchmark's read mix, one line per environment setting (python tools/stream_probe.py [--n 2.8e9] [--t 15]
[--subst 0,10000] [--flags all,orders] SETTING ...): a SETTING is `NAME=VALUE[+NAME=VALUE...]` or `-` for none; each runs in a process of its
own (the switches are read once).  Per line: ms per launch of 2^27 windows (device profile of the library), G windows/s, requests per window and
-- with FMSI_HIP_STREAM_UTIL in the setting -- lane-rounds per window and the busy share.  Measurement only."""
import argparse
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ".")
    import numpy as np
    import torch
    import fmsi_amd as fm
    n, t, subst, fl = int(float(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    k, L = 31, int(os.environ.get("PROBE_READ_LEN", "10000"))
    idx = fm.DeviceIndex.build_synthetic(n, 12345, k, with_klcp=True, prefix_len=t)
    n_reads = (1 << 27) // (L - k + 1)
    nw = n_reads * (L - k + 1)
    buf = torch.zeros(n_reads * L // 32 + 2, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    run_start = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    run_windows = np.full(n_reads, L - k + 1, dtype=np.uint32)
    flags = {"all": fm.MODE_ALL | fm.OUT_BITS, "or": fm.MODE_OR | fm.OUT_BITS, "orders": fm.ORDERS | fm.STRICT, "orders_fast": fm.ORDERS, "strict": fm.MODE_ALL | fm.STRICT}[fl]
    out = torch.zeros(nw * 2 if fl.startswith("orders") else (nw // 8 + 8) * 8 if fl == "strict" else (nw // 64 + 8) * 8, dtype=torch.int64 if fl.startswith("orders") else torch.uint8, device="cuda")
    cnt = torch.zeros(fm.N_COUNTERS, dtype=torch.int64, device="cuda")
    idx.synthetic_reads_dev(L, 42, 0, n_reads, buf.data_ptr(), subst, s)
    idx.query_stream_dev(buf.data_ptr(), n_reads * L, run_start, run_windows, k, flags, out.data_ptr(), cnt.data_ptr(), s)
    torch.cuda.synchronize()
    c = cnt.cpu().numpy()
    digest = int(out.view(torch.int64).sum().item()) & 0xFFFFFFFFFFFF
    for _ in range(2):
        idx.query_stream_dev(buf.data_ptr(), n_reads * L, run_start, run_windows, k, flags, out.data_ptr(), 0, s)
    torch.cuda.synchronize()
    idx.profile(True); idx.profile_read(reset=True)
    for step in range(8):
        idx.synthetic_reads_dev(L, 42, (step + 1) * n_reads, n_reads, buf.data_ptr(), subst, s)
        idx.query_stream_dev(buf.data_ptr(), n_reads * L, run_start, run_windows, k, flags, out.data_ptr(), 0, s)
    torch.cuda.synchronize()
    nl, ms = idx.profile_read(reset=True); idx.profile(False)
    util = f"  lane-rounds/window {c[6] / nw:.3f} busy {c[7] / max(c[6], 1):.3f}" if os.environ.get("FMSI_HIP_STREAM_UTIL") else ""
    print(f"{ms / nl:7.3f} ms {nw / (ms / nl) / 1e6:6.1f} G/s  req/window {c[1] / nw:.3f} (table {c[4] / nw:.3f} LF {c[0] / nw:.3f} searches {c[3] / nw:.3f}){util}  sum {digest:012x}", flush=True)
    sys.exit(0)

ap = argparse.ArgumentParser()
ap.add_argument("--n", default="2.8e9")
ap.add_argument("--t", default="15")
ap.add_argument("--subst", default="0,10000")
ap.add_argument("--flags", default="all")
ap.add_argument("settings", nargs="*", default=["-"])
a = ap.parse_args()
for fl in a.flags.split(","):
    for subst in a.subst.split(","):
        for t in a.t.split(","):
            for setting in a.settings:
                env = dict(os.environ)
                if setting != "-":
                    for kv in setting.split("+"):
                        name, _, val = kv.partition("=")
                        env[name] = val
                print(f"t={t} {fl:11s} subst={int(subst) // 10000}% {setting:40s} ", end="", flush=True)
                r = subprocess.run([sys.executable, __file__, "--child", a.n, t, subst, fl], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else f"FAILED rc={r.returncode} {r.stderr[-300:]}", flush=True)
