"""The kLCP kernel's absence certificates by read class (python tools/cert_probe.py [n] [subst_ppm]): reads from the text in reading order, the same
reverse-complemented, random reads -- with and without substitutions -- under FMSI_HIP_STREAM_CERT = 0 (none) / 1 (offset 0) / unset (placed by the
run of failing windows): ms, G windows/s, table requests / LF steps / lines per window.  One process per setting (the switch is read once)."""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ".")
    import numpy as np
    import torch
    import fmsi_amd as fm
    n, subst = int(float(sys.argv[2])), int(sys.argv[3])
    k, L = 31, 10000
    idx = fm.DeviceIndex.build_synthetic(n, 12345, k, with_klcp=True, prefix_len=int(os.environ.get("PROBE_T", "14")))
    n_reads = (1 << 25) // (L - k + 1)
    words = n_reads * L // 32 + 2
    text = idx.download_text(n)[:words].copy()
    rng = np.random.default_rng(5)
    codes = ((text[:, None] >> (2 * np.arange(32, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8).reshape(-1)[:n_reads * L]
    if subst:
        hit = rng.random(codes.size) < subst / 1e6
        codes = np.where(hit, (codes + rng.integers(1, 4, codes.size, dtype=np.uint8)) & 3, codes).astype(np.uint8)
    def pack(c):
        c = np.concatenate([c, np.zeros(words * 32 - c.size, dtype=np.uint8)]).reshape(-1, 32).astype(np.uint64)
        return (c << (2 * np.arange(32, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
    fwd = pack(codes)
    rc = pack((3 - codes.reshape(n_reads, L)[:, ::-1]).reshape(-1))
    rnd = rng.integers(0, 2**63, words, dtype=np.int64).view(np.uint64)
    run_start = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    run_windows = np.full(n_reads, L - k + 1, dtype=np.uint32)
    nw = int(run_windows.sum())
    out = torch.zeros(nw * 2, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(fm.N_COUNTERS, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for name, host in (("text reads", fwd), ("text reads, reverse-complemented", rc), ("random reads", rnd)):
        buf = torch.from_numpy(host.view(np.int64).copy()).cuda()
        for fname, flags in (("ALL", fm.MODE_ALL | fm.OUT_BITS), ("ORDERS|STRICT", fm.ORDERS | fm.STRICT)):
            cnt.zero_()
            idx.query_stream_dev(buf.data_ptr(), n_reads * L, run_start, run_windows, k, flags, out.data_ptr(), cnt.data_ptr(), s)
            torch.cuda.synchronize()
            c = cnt.cpu().numpy()
            idx.profile(True); idx.profile_read(reset=True)
            for _ in range(5):
                idx.query_stream_dev(buf.data_ptr(), n_reads * L, run_start, run_windows, k, flags, out.data_ptr(), 0, s)
            torch.cuda.synchronize()
            nl, ms = idx.profile_read(reset=True); idx.profile(False)
            print(f"  {name:34s} {fname:14s} {ms / nl:7.3f} ms  {nw / (ms / nl) / 1e6:6.1f} G windows/s   per window: table requests {c[4] / nw:.3f}  LF steps {c[0] / nw:.3f}  "
                  f"lines {(c[1] - c[4]) / nw:.3f}  searches {c[3] / nw:.3f}" +
                  (f"  rounds per 64 windows {c[6] / nw:.1f}, busy lanes {c[7] / max(c[6], 1):.2f}" if os.environ.get("FMSI_HIP_STREAM_UTIL") else ""), flush=True)
    sys.exit(0)

n = sys.argv[1] if len(sys.argv) > 1 else "4e8"
for subst in ([sys.argv[2]] if len(sys.argv) > 2 else ["0", "10000"]):
    for mode in ("0", "1", None):
        env = dict(os.environ)
        env.pop("FMSI_HIP_STREAM_CERT", None)
        if mode is not None:
            env["FMSI_HIP_STREAM_CERT"] = mode
        print(f"n = {n}, {int(subst) / 1e4:g} % substitutions, certificates: {'none' if mode == '0' else 'offset 0' if mode == '1' else 'placed'}", flush=True)
        subprocess.run([sys.executable, __file__, "--child", n, subst], env=env, check=True)
