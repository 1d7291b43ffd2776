"""cpu_baseline leg of bench.py -- the ONLY part of the benchmark that touches oracle/.

It hands the CPU side the very index the GPU built (layout planes -> the reference's on-disk .fmsi.* format),
draws a bounded sample of the same seeded query stream, and times
  kind "reference": the real reference's query_kmers() loop (oracle/_ref, built from /root/reference in the build
                    container and shipped as a binary), or
  kind "port"     : the oracle's C restatement, when oracle/_ref is not present,
single-threaded (the reference has no threading) and, as `socket_value`, with one forked worker per core the box
grants (one socket's physical cores, capped by the container's cgroup CPU quota) on disjoint query shards over the
shared read-only index.  It also checks the sample against the GPU's answers.
"""
import ctypes as C
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "oracle"))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cores():
    """(workers to fork, cgroup quota or None): physical cores of one socket (2 sockets x SMT2 on the gpurun box),
    capped by the container's cgroup v2 CPU quota."""
    cores = max(1, (os.cpu_count() or 2) // 4)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(q) // int(per))
            cores = min(cores, quota)
    except (OSError, ValueError):
        pass
    return cores, quota


def oracle_from_index(index, k, with_klcp):
    """GPU-built index (its downloaded layout planes) -> the oracle's in-memory index (C restatement)."""
    import oracle_py
    blocks, _ = index.download_layout()
    rows = int(index.info.n)
    words = (rows + 63) // 64 + 1
    planes = [np.ascontiguousarray(np.ascontiguousarray(blocks[:, lo:lo + 4]).reshape(-1).view(np.uint64)[:words]) for lo in (4, 8, 12)]
    del blocks
    L = oracle_py.oracle_lib()
    L.orc_index_from_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
    L.orc_index_from_planes.restype = C.c_void_p
    L.orc_index_set_klcp.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.orc_index_set_klcp.restype = None
    counts = np.array(list(index.info.counts), dtype=np.uint64)
    h = L.orc_index_from_planes(planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data, rows, counts.ctypes.data,
                                int(index.info.dollar_position), k)
    if with_klcp:
        kl = index.download_klcp()
        L.orc_index_set_klcp(h, kl.ctypes.data, rows)
    del planes
    return oracle_py.Oracle(L, h)


def _export_index(index, k, with_klcp):
    """GPU-built index -> oracle structure -> (reference-format files ->) the real reference if it is available."""
    import oracle_py
    orc = oracle_from_index(index, k, with_klcp)
    if not oracle_py.ref_available():
        return orc, "port", None
    tmp = tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    prefix = os.path.join(tmp.name, "bench")
    orc.dump(prefix)
    orc.close()
    return oracle_py.Reference.load(prefix, use_klcp=with_klcp), "reference", tmp


def _kmers_to_text(kmers: np.ndarray, k: int) -> bytes:
    codes = ((kmers[:, None] >> (2 * np.arange(k, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()


def _fork_workers(workers, fn):
    """Run fn(w) -> seconds in `workers` forked children; returns the list of seconds."""
    pipes = []
    for w in range(workers):
        r, wr = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            try:
                os.write(wr, f"{fn(w)}".encode())
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
    return secs


def run(args, index, n, k):
    """Isolated k-mers (workload kmers)."""
    import torch
    import fmsi_amd as fm
    import oracle_py

    mode = oracle_py.MODE_ALL if args.mode == "all" else oracle_py.MODE_OR
    cores, quota = _cores()
    sample = args.cpu_sample
    if sample < 0:   # ~10 s of single-thread work: ~9 us per k-mer at human scale, ~1.5 us when the index is cache resident
        sample = 1_000_000 if n >= 2e8 else 3_000_000
    # the all-cores leg: the workers together answer the rest of 10^7 k-mers (human scale: ~7 s per worker at 16 workers)
    per_worker = max(20_000, sample // 8, (10_000_000 - sample) // max(cores, 1) if args.cpu_sample < 0 else 0)
    total = sample + cores * per_worker

    d = torch.empty(total, dtype=torch.int64, device="cuda")
    d_out = torch.empty(total, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    index.synthetic_queries_dev(k, args.query_seed, 0, total, d.data_ptr(), s)   # the stream the GPU steps consume
    index.query_kmers_dev(d.data_ptr(), total, k, fm.MODE_ALL if args.mode == "all" else fm.MODE_OR, d_out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    text = _kmers_to_text(d.cpu().numpy().view(np.uint64), k)
    gpu_bits = d_out.cpu().numpy()

    t0 = time.time()
    obj, kind, tmp = _export_index(index, k, with_klcp=False)
    prep_s = time.time() - t0
    sec1, bits = obj.time_isolated_kmers(text[:sample * k], sample, k, mode)
    agree = bool(np.array_equal(bits, gpu_bits[:sample]))
    rest = text[sample * k:]
    secs = _fork_workers(cores, lambda w: obj.time_isolated_kmers(rest[w * per_worker * k:(w + 1) * per_worker * k], per_worker, k, mode)[0])
    tuned = {}
    if kind == "reference" and oracle_py.REF_TUNED_LIB.exists():
        # the same sources with tuned flags (oracle/Makefile.ref), same files, same k-mers, one thread
        try:
            obj_t = oracle_py.Reference.load_tuned(os.path.join(tmp.name, "bench"))
            sec_t, bits_t = obj_t.time_isolated_kmers(text[:sample * k], sample, k, mode)
            obj_t.close()
            tuned = {"tuned_value": sample / sec_t, "tuned_flags": "-O3 -DNDEBUG -march=x86-64-v3", "tuned_seconds": round(sec_t, 3),
                     "tuned_matches_gpu": bool(np.array_equal(bits_t, gpu_bits[:sample]))}
        except Exception as e:   # a reported extra, never a reason to lose the baseline
            tuned = {"tuned_error": repr(e)}
    if tmp is not None:
        tmp.cleanup()
    return {
        **tuned,
        "value": sample / sec1, "unit": "k-mers/s", "cores": 1, "kind": kind,
        "sample": f"first {sample} k-mers of the benchmark's seeded query stream against the same {int(index.info.n)}-row index "
                  f"(exported from the GPU build in the reference's .fmsi format); query_kmers() per k-mer, as-shipped flags "
                  f"(-g -O2), query loop only",
        "seconds": round(sec1, 3), "matches_gpu": agree,
        "socket_value": cores * per_worker / max(secs), "socket_cores": cores,
        "socket_sample": f"{cores} forked workers x {per_worker} further k-mers each, shared read-only index, slowest worker {max(secs):.2f} s; "
                         f"socket_cores is what this container may use (cgroup CPU quota {quota} of {os.cpu_count()} logical CPUs), not a whole socket",
        "index_export_seconds": round(prep_s, 1), "cpu_model": _cpu_model(), "logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota,
    }


def run_stream(args, index, n, k):
    """Reads with kLCP streaming (workload stream): the reference's record loop with -S over a FASTA of sample reads."""
    import torch
    import fmsi_amd as fm
    import oracle_py

    mode = oracle_py.MODE_ALL if args.mode == "all" else oracle_py.MODE_OR
    cores, quota = _cores()
    L = args.read_len
    wins = L - k + 1
    sample_w = args.cpu_sample
    if sample_w < 0:   # ~10 s single-threaded: ~4 us per window at human scale with -S
        sample_w = 3_000_000
    reads1 = max(1, sample_w // wins)
    per_worker = max(1, reads1 // 8)
    total_reads = reads1 + cores * per_worker

    words = total_reads * L // 32 + 2
    d = torch.empty(words, dtype=torch.int64, device="cuda")
    d_out = torch.empty(total_reads * wins, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    index.synthetic_reads_dev(L, args.query_seed, 0, total_reads, d.data_ptr(), args.subst_ppm, s)
    run_start = np.arange(total_reads, dtype=np.uint64) * np.uint64(L)
    run_windows = np.full(total_reads, wins, dtype=np.uint32)
    # (the device path bench.py is timing: the kLCP kernel, or the run entry point -- dictionary / candidate tables)
    gpu_fn = index.query_runs_dev if getattr(args, "stream_via", "klcp") == "runs" else index.query_stream_dev
    gpu_fn(d.data_ptr(), total_reads * L, run_start, run_windows, k, fm.MODE_ALL if args.mode == "all" else fm.MODE_OR, d_out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    packed = d.cpu().numpy().view(np.uint64)
    codes = ((packed[:, None] >> (2 * np.arange(32, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8).reshape(-1)[:total_reads * L]
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].reshape(total_reads, L)
    gpu_bits = d_out.cpu().numpy().reshape(total_reads, wins)

    t0 = time.time()
    obj, kind, tmp = _export_index(index, k, with_klcp=True)
    prep_s = time.time() - t0
    work = tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)

    def write_fasta(path, lo, hi):
        with open(path, "wb") as f:
            for r in range(lo, hi):
                f.write(b">r%d\n" % r)
                f.write(bases[r].tobytes())
                f.write(b"\n")

    f1 = os.path.join(work.name, "s.fa")
    write_fasta(f1, 0, reads1)
    text, sec1 = obj.query_file(f1, k, has_klcp=True, mode=mode, cap=reads1 * (wins + 32) + 4096)
    got = np.array([np.frombuffer(line.split("\t")[1].encode(), dtype=np.uint8) - ord("0") for line in text.splitlines()])
    agree = bool(np.array_equal(got, gpu_bits[:reads1]))
    shard_files = []
    for w in range(cores):
        p = os.path.join(work.name, f"w{w}.fa")
        write_fasta(p, reads1 + w * per_worker, reads1 + (w + 1) * per_worker)
        shard_files.append(p)
    secs = _fork_workers(cores, lambda w: obj.query_file(shard_files[w], k, has_klcp=True, mode=mode, cap=per_worker * (wins + 32) + 4096)[1])
    work.cleanup()
    if tmp is not None:
        tmp.cleanup()
    return {
        "value": reads1 * wins / sec1, "unit": "k-mers/s", "cores": 1, "kind": kind,
        "sample": f"first {reads1} reads ({reads1 * wins} windows) of the benchmark's seeded read stream against the same "
                  f"{int(index.info.n)}-row index incl. its kLCP array; query_kmers<all>(..., has_klcp=true) through the reference's "
                  f"record loop and chunking, as-shipped flags (-g -O2), query loop only",
        "seconds": round(sec1, 3), "matches_gpu": agree,
        "socket_value": cores * per_worker * wins / max(secs), "socket_cores": cores,
        "socket_sample": f"{cores} forked workers x {per_worker} further reads each, slowest worker {max(secs):.2f} s",
        "index_export_seconds": round(prep_s, 1), "cpu_model": _cpu_model(), "logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota,
    }


def check_legs(c, args, index, paths, names):
    """The `configs` legs of bench.py: a bounded sample of every variant's queries against the real reference (oracle/_ref; the oracle's
    restatement when it did not travel) on the SAME index -- written here by the product's own writer (fmsi_hip_index_write_files: byte
    identical to the reference's files on the goldens and on a 10^8-row index the reference built; 1 s at human scale where the oracle's
    single-threaded RRR encoder takes 18 s) and loaded by the reference's load_index().  Isolated k-mers: query_kmers() per k-mer;
    reads: the reference's record loop with -S (has_klcp) over a FASTA of the first reads of the variant's seeded stream.  The GPU side
    answers the same sample through the variant's own path (same entry point, same flags).  Returns {name: {"matches", "what"}}."""
    import torch
    import fmsi_amd as fm
    import oracle_py
    from bench import QueryPath

    k = args.k
    mode = oracle_py.MODE_ALL if args.mode == "all" else oracle_py.MODE_OR
    n_rows = int(index.info.n)
    has_klcp = bool(index.info.has_klcp)
    tmp = tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    prefix = os.path.join(tmp.name, "leg")
    t0 = time.time()
    index.write_files(prefix)
    write_s = time.time() - t0
    t0 = time.time()
    if oracle_py.ref_available():
        ref, kind = oracle_py.Reference.load(prefix, use_klcp=has_klcp), "the real reference (oracle/_ref)"
    else:
        ref, kind = oracle_py.Oracle.load(prefix, use_klcp=has_klcp), "the oracle's C restatement (oracle/_ref did not travel)"
    load_s = time.time() - t0
    out = {}
    stream_expected = {}   # subst_ppm -> (bits per read, seconds, reads)
    for path, name in zip(paths, names):
        if not path.stream_mode:
            sample = 1_000_000 if n_rows < 2e8 else 200_000
            p = QueryPath(c, args, index, path.via, k, sample)
            d = torch.empty(sample, dtype=torch.int64, device=c.dev_t)
            d_out = torch.empty(sample, dtype=torch.uint8, device=c.dev_t)
            p.make_input(0, d)
            p.launch(d, d_out)
            torch.cuda.synchronize()
            text = _kmers_to_text(d.cpu().numpy().view(np.uint64), k)
            sec, bits = ref.time_isolated_kmers(text, sample, k, mode)
            out[name] = {"matches": bool(np.array_equal(bits, d_out.cpu().numpy())),
                         "what": f"{kind}: query_kmers() on the first {sample} k-mers of the leg's seeded stream, {sample / sec:.0f} k-mers/s on one core, against the "
                                 f"same {n_rows}-row index (files by fmsi_hip_index_write_files {write_s:.1f} s, load_index {load_s:.1f} s)"}
            continue
        L = path.read_len
        wins = L - k + 1
        reads = 30
        p = QueryPath(c, args, index, path.via, k, reads * wins, L, path.subst_ppm)
        d = torch.empty(p.in_words, dtype=torch.int64, device=c.dev_t)
        def fasta_of_reads():
            fa_ = os.path.join(tmp.name, f"s{path.subst_ppm}.fa")
            if not os.path.exists(fa_):
                packed_ = d.cpu().numpy().view(np.uint64)
                codes_ = ((packed_[:, None] >> (2 * np.arange(32, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8).reshape(-1)[:p.n_reads * L]
                bases_ = np.frombuffer(b"ACGT", dtype=np.uint8)[codes_].reshape(p.n_reads, L)
                with open(fa_, "wb") as f_:
                    for r_ in range(p.n_reads):
                        f_.write(b">r%d\n" % r_)
                        f_.write(bases_[r_].tobytes())
                        f_.write(b"\n")
            return fa_

        ow = getattr(p, "order_words", 0)
        if ow:   # `lookup -S`: the reference's orders (its record loop with output_orders) against the device's -- the first strand that has one
            d_ord = torch.empty(p.batch * ow, dtype=torch.int64, device=c.dev_t)
            p.make_input(0, d)
            p.launch(d, d_ord)
            torch.cuda.synchronize()
            got = d_ord.cpu().numpy().reshape(p.batch, ow)
            got = np.where(got[:, 0] >= 0, got[:, 0], got[:, -1]).reshape(p.n_reads, wins)
            fa = fasta_of_reads()
            text, sec = ref.query_file(fa, k, has_klcp=has_klcp, mode=mode, orders=True, cap=p.n_reads * (wins * 11 + 32) + 4096)
            want = np.array([np.array(line.split("\t")[1].split(","), dtype=np.int64) for line in text.splitlines()])
            out[name] = {"matches": bool(want.shape == got.shape and np.array_equal(want, got) and (want >= 0).mean() > 0.2),
                         "what": f"{kind}: its record loop as `lookup -S` (query_kmers_streaming(..., output_orders = true)) over the first {p.n_reads} reads ({p.n_reads * wins} windows) "
                                 f"of the leg's seeded read stream, {p.n_reads * wins / sec:.0f} windows/s on one core, against the same {n_rows}-row index; the device's order of a window "
                                 f"= the first strand's that has one (on this text no k-mer occurs on both strands)"}
            continue
        d_out = torch.empty(p.batch, dtype=torch.uint8, device=c.dev_t)
        p.make_input(0, d)
        p.launch(d, d_out)
        torch.cuda.synchronize()
        gpu_bits = d_out.cpu().numpy().reshape(p.n_reads, wins)
        if path.subst_ppm not in stream_expected:
            fa = fasta_of_reads()
            text, sec = ref.query_file(fa, k, has_klcp=has_klcp, mode=mode, cap=p.n_reads * (wins + 32) + 4096)
            want = np.array([np.frombuffer(line.split("\t")[1].encode(), dtype=np.uint8) - ord("0") for line in text.splitlines()])
            stream_expected[path.subst_ppm] = (want, sec, p.n_reads)
        want, sec, nr = stream_expected[path.subst_ppm]
        out[name] = {"matches": bool(want.shape == gpu_bits.shape and np.array_equal(want, gpu_bits)),
                     "what": f"{kind}: its record loop with {'-S -O (query_kmers<all>(..., has_klcp=true))' if has_klcp else '-O (this handle has no kLCP array; the presence answers are the same)'} "
                             f"over the first {nr} reads ({nr * wins} windows) of the leg's "
                             f"seeded read stream, {nr * wins / sec:.0f} windows/s on one core, against the same {n_rows}-row index "
                             f"(files by fmsi_hip_index_write_files {write_s:.1f} s, load_index {load_s:.1f} s)"}
    ref.close()
    tmp.cleanup()
    return out
