"""cpu_baseline leg of bench.py -- the ONLY part of the benchmark that touches oracle/.

It hands the CPU side the very index the GPU built (layout planes -> the reference's on-disk .fmsi.* format),
draws a bounded sample of the same seeded query stream, and times
  kind "reference": the real reference's query_kmers() loop (oracle/_ref, built from /root/reference in the build
                    container and shipped as a binary), or
  kind "port"     : the oracle's C restatement, when oracle/_ref is not present,
single-threaded (the reference has no threading) and, as `socket_value`, with one forked worker per core of one
socket on disjoint query shards over the shared read-only index.  It also checks the sample against the GPU.
"""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "oracle"))


def _planes(index):
    blocks, _ = index.download_layout()
    n = int(index.info.n)
    words = (n + 63) // 64 + 1
    out = []
    for lo in (4, 8, 12):
        plane = np.ascontiguousarray(blocks[:, lo:lo + 4]).reshape(-1).view(np.uint64)
        out.append(np.ascontiguousarray(plane[:words]))
    return out, n


def _kmers_to_text(kmers: np.ndarray, k: int) -> bytes:
    codes = ((kmers[:, None] >> (2 * np.arange(k, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()


def _forked_rate(obj, text: bytes, n_each: int, k: int, mode: int, workers: int):
    """One forked worker per core on disjoint shards; returns aggregate k-mers/s (max worker time)."""
    pipes = []
    t0 = time.perf_counter()
    for w in range(workers):
        r, wr = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            try:
                shard = text[w * n_each * k:(w + 1) * n_each * k]
                sec, _ = obj.time_isolated_kmers(shard, n_each, k, mode)
                os.write(wr, f"{sec}".encode())
            finally:
                os._exit(0)
        os.close(wr)
        pipes.append((pid, r))
    secs = []
    for pid, r in pipes:
        data = os.read(r, 64)
        os.close(r)
        os.waitpid(pid, 0)
        secs.append(float(data.decode() or "nan"))
    wall = time.perf_counter() - t0
    return workers * n_each / max(secs), max(secs), wall


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run(args, index, n, k):
    import ctypes as C
    import torch
    import fmsi_amd as fm
    import oracle_py

    mode = oracle_py.MODE_ALL if args.mode == "all" else oracle_py.MODE_OR
    cores_socket = max(1, (os.cpu_count() or 2) // 4)       # 2 sockets x SMT2 on the gpurun box -> physical cores of one socket
    quota = None
    try:                                                      # the box's container may cap CPU time (cgroup v2 cpu.max)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(q) // int(per))
            cores_socket = min(cores_socket, quota)
    except (OSError, ValueError):
        pass
    sample = args.cpu_sample
    if sample < 0:
        # ~10-20 s of single-thread work: ~15 us per k-mer at human scale, ~3 us when the index is cache resident
        sample = 1_000_000 if n >= 2e8 else 3_000_000
    per_worker = max(20_000, sample // 8)
    total = sample + cores_socket * per_worker

    # the same seeded query stream the GPU steps consume, drawn on the device and copied out
    d = torch.empty(total, dtype=torch.int64, device="cuda")
    d_out = torch.empty(total, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    index.synthetic_queries_dev(k, args.query_seed, 0, total, d.data_ptr(), s)
    gflags = (fm.MODE_ALL if args.mode == "all" else fm.MODE_OR)
    index.query_kmers_dev(d.data_ptr(), total, k, gflags, d_out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    kmers = d.cpu().numpy().view(np.uint64)
    gpu_bits = d_out.cpu().numpy()
    text = _kmers_to_text(kmers, k)

    t0 = time.time()
    (hi, lo, mask), rows = _planes(index)
    L = oracle_py.oracle_lib()
    L.orc_index_from_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
    L.orc_index_from_planes.restype = C.c_void_p
    counts = np.array(list(index.info.counts), dtype=np.uint64)
    h = L.orc_index_from_planes(hi.ctypes.data, lo.ctypes.data, mask.ctypes.data, rows, counts.ctypes.data,
                                int(index.info.dollar_position), k)
    orc = oracle_py.Oracle(L, h)
    del hi, lo, mask
    kind = "port"
    obj = orc
    tmp = None
    if oracle_py.ref_available():
        tmp = tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        prefix = os.path.join(tmp.name, "bench")
        orc.dump(prefix)
        obj = oracle_py.Reference.load(prefix, use_klcp=False)
        kind = "reference"
    prep_s = time.time() - t0

    sec1, bits = obj.time_isolated_kmers(text[:sample * k], sample, k, mode)
    agree = bool(np.array_equal(bits, gpu_bits[:sample]))
    rate1 = sample / sec1
    rate_socket, sec_max, _ = _forked_rate(obj, text[sample * k:], per_worker, k, mode, cores_socket)
    if tmp is not None:
        tmp.cleanup()
    return {
        "value": rate1, "unit": "k-mers/s", "cores": 1, "kind": kind,
        "sample": f"first {sample} k-mers of the benchmark's seeded query stream against the same {rows}-row index "
                  f"(exported from the GPU build in the reference's .fmsi format); query_kmers() per k-mer, -O3-free as-shipped "
                  f"flags (-O2), query loop only",
        "seconds": round(sec1, 3), "matches_gpu": agree,
        "socket_value": rate_socket, "socket_cores": cores_socket,
        "socket_sample": f"{cores_socket} forked workers x {per_worker} further k-mers each, shared read-only index, slowest worker {sec_max:.2f} s",
        "index_export_seconds": round(prep_s, 1), "cpu_model": _cpu_model(), "logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota,
    }
