#!/usr/bin/env python3
"""bench.py -- k-mer membership queries/s (k = 31) of the MI355X `fmsi query` hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  One "step" = one pass of the hot path over one batch of
synthetic queries that is already resident in HBM; W untimed steps, then exactly K timed steps bracketed by
barrier + synchronize, MAX over ranks, one JSON line from rank 0.

Workloads (BASELINE.json configs; all on a synthetic max-ones masked superstring of --n bases that is
generated and indexed on the device, replicated per GPU):
  --workload kmers   (default; configs 2/3/5) isolated random 31-mers, half of them present (half of those as the
                     reverse complement), `fmsi query -O` semantics: fmsi_hip_query_kmers, forward strand first.
  --workload stream  (config 4) reads of --read-len bases (half substrings of the superstring, half of those
                     reverse-complemented; half uniform random), `fmsi query -S -O`: fmsi_hip_query_stream (kLCP).
The seeded, counter-based query stream is sharded across ranks by contiguous ranges -- weak scaling, no collective
inside the search.  Results are bit-packed (FMSI_HIP_OUT_BITS: what the reference prints, one symbol per window,
src/fms_index.h:305-309) and their way to the host is INSIDE the timed region: every step's words are gathered to rank 0
over RCCL (N > 1) and copied to page-locked host memory on a copy stream, double-buffered against the next step's kernel
(`--byte-results`: the round-1 form -- bytes left in HBM, gather after the timed region).

Extra objects on the JSON line:
  roofline     -- HBM bound.  `achieved` = algorithmic bytes of one launch / average kernel duration.  The bytes are
                  what this layout must touch, counted by the kernels themselves: 64 B per table line, 128 B per
                  pair-layout block (64 B per single-step block), 32 B per candidate / text read, 8 B per kLCP word
                  when streaming, + query bytes in + 1 B result out per window; the duration comes from HIP events
                  recorded on the launch stream inside the library, around the launch's kernels (the default `phased`
                  form is two: lookup + stepping).  `random_requests_per_s` is the figure the path is really bound by
                  (DESIGN.md 3, 5): random requests, not bytes -- `request_roofline` prices it against the chip's measured
                  ceiling of 48 G random line requests/s (tools/gather_bench entry, profiles/r01k_gather_entry.txt).  A table
                  line counts 64 B (the smallest HBM access) although the entry used is 16 or 32 bytes: `useful_*` counts
                  the entry's bytes instead.
                  `canonical_achieved` applies SURVEY 8(d)'s reference-layout formula (256 B per LF step the REFERENCE
                  algorithm executes for the same queries + 64 B per mask probe + in/out) to the same time; it can
                  exceed the HBM peak because the fused layout and the prefix table touch far fewer bytes.
                  `traffic` = HBM bytes per launch of the timed kernel, MEASURED IN THIS RUN when rocprofv3 is there (default
                  dictionary path, N = 1): two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE / WRITE_SIZE, separate
                  passes, corrected as MI355X_MICROARCH.md prescribes; `traffic_source` says how) -- else from the passes committed
                  under profiles/ when their configuration is the one being run.  `lf_path` = the table-less LF walk on the same
                  index in timed legs of its own (two-base steps; lock-step), the path north_star names.
  cpu_baseline -- the real reference (oracle/_ref, kind "reference") or the C restatement (kind "port") timed on
                  this box's host cores on a bounded sample of the same queries against the same index (bench_cpu.py).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured streaming ceiling


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["kmers", "stream", "e2e"], default="kmers",
                    help="e2e: the whole `fmsi query` process (tools/e2e_bench.py: index files and a FASTA of --reads reads of --read-len bases in "
                         "/dev/shm, the CLI with --stats, a sample against the real reference) -- SURVEY 8(d)(ii); N = 1 only")
    ap.add_argument("--reads", type=int, default=100000, help="e2e workload: reads in the query file")
    ap.add_argument("--n", "--index-bases", dest="n", type=float, default=2.8e9, help="superstring length in bases (human scale: 2.8e9)")
    ap.add_argument("--batch", type=float, default=float(1 << 27), help="query windows per step and GPU")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--prefix-len", type=int, default=-1,
                    help="private prefix table length t (0 = none; default: 16; without candidate tables less while 4^t > 4n)")
    ap.add_argument("--kernel", choices=["phased", "pair", "refill", "lockstep"], default="phased",
                    help="phased = pair layout, table lookups in their own kernel (default); pair = the same in one kernel")
    ap.add_argument("--accel", choices=["dict", "tables"], default="dict",
                    help="private accelerator of the phased workload: dict = the k-mer dictionary (fmsi_hip_index_enable_dict: one 128-byte "
                         "bucket read per k-mer, ~12 bytes of HBM per distinct k-mer); tables = prefix table + candidate tables (177 GB at t = 16)")
    ap.add_argument("--dict-sparse", action="store_true", help="dictionary at half the load (twice the HBM, no second-level reads)")
    ap.add_argument("--no-dict-chain", dest="dict_chain", action="store_false",
                    help="stream workload: the dictionary without the extras of chained lookups (FMSI_HIP_DICT_CHAIN, + about 7 B per k-mer: 58.7 "
                         "instead of 39.0 GB at human scale), i.e. one bucket read per window; default: chained -- a lane follows the superstring "
                         "from the occurrence of its 32-window task's first window")
    ap.add_argument("--hbm-budget", type=float, default=0.0,
                    help="GB of HBM the index and its accelerators may take (0 = no limit): picks the fastest configuration that fits -- "
                         "sparse dictionary, dictionary, or the LF walk behind the deepest prefix table that fits (DESIGN.md 4, footprint table)")
    ap.add_argument("--mode", choices=["all", "or"], default="all")
    ap.add_argument("--read-len", type=int, default=10000, help="stream workload: bases per read")
    ap.add_argument("--subst-ppm", type=int, default=0, help="stream workload: substitutions per million bases of positive reads")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--query-seed", type=int, default=42)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="windows for the CPU baseline (-1 = auto, 0 = skip)")
    ap.add_argument("--no-counters", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="phased kernel without text verification of one-row table intervals")
    ap.add_argument("--repeat-permille", type=int, default=0,
                    help="share (per 1000) of the superstring's 2048-base segments that are diverged copies of a repeat family "
                         "(fmsi_hip_index_build_synthetic_repeats): prefix-table keys with tens of occurrences, as in a real genome")
    ap.add_argument("--repeat-families", type=int, default=1000)
    ap.add_argument("--repeat-divergence-permille", type=int, default=100)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-rank code path (process group over RCCL, barriers, MAX over ranks, gather) with one rank too: "
                         "a check of that path on a 1-GPU box, under torch.distributed.run --nproc-per-node 1")
    ap.add_argument("--abi-gather", action="store_true",
                    help="after the timed region, also gather the results through fmsi_hip_gather (C-ABI, own RCCL communicator) and "
                         "compare with torch.distributed's gather")
    ap.add_argument("--byte-results", action="store_true",
                    help="one result byte per window and no device-to-host copy in the timed region (the round-1 form of the benchmark); "
                         "default: FMSI_HIP_OUT_BITS -- bit-packed results, each step's words copied to page-locked host memory (double "
                         "buffered) and, for N > 1, gathered to rank 0 over RCCL first, all inside the timed region")
    ap.add_argument("--no-lf-leg", action="store_true", help="skip the table-less LF-walk legs (roofline.lf_path)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (two child runs of this script under rocprofv3 --pmc, FETCH_SIZE and "
                         "WRITE_SIZE in separate passes); the figure then comes from the committed profiles/*traffic*.json, if one matches")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the child's mode: build, three launches, nothing printed
    ap.add_argument("--lf-steps", type=int, default=3, help="timed launches per LF-walk leg")
    ap.add_argument("--stream-via", choices=["auto", "klcp", "runs"], default="auto",
                    help="stream workload: klcp = fmsi_hip_query_stream (the kLCP kernel, what -S means in the reference); runs = the same "
                         "windows through fmsi_hip_query_runs (the dictionary, or phased kernels + candidate tables; same answers -- the "
                         "presence answers of -S do not depend on the kLCP array); auto = runs with --accel dict, else klcp")
    args = ap.parse_args()
    if args.stream_via == "auto":
        args.stream_via = "runs" if args.accel == "dict" else "klcp"
    return args


def pmc_traffic(args, n, batch, k):
    """HBM bytes per launch of the timed kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE and WRITE_SIZE in separate passes, calibrated on random 64-B reads: profiles/README.md).
    bench.py cannot collect PMC counters itself; the figure is attached only when the run's configuration is
    the one that was profiled, otherwise null."""
    best = None
    for p in sorted((REPO / "profiles").glob("r*_traffic*.json")):
        try:
            t = json.loads(p.read_text())
        except (OSError, ValueError):
            continue
        c = t.get("config", {})
        if (c.get("workload", "kmers"), c.get("n"), c.get("batch"), c.get("k"), c.get("prefix_len"), c.get("mode")) == \
                (args.workload, n, batch, k, args.prefix_len, args.mode) and c.get("subst_ppm", 0) == args.subst_ppm and \
                args.repeat_permille == 0 and \
                ("dict" if getattr(args, "use_dict", False) else "stream" if (args.workload == "stream" and args.stream_via == "klcp") else
                 ("pair" if (args.kernel == "phased" or args.workload == "stream") else args.kernel)) in t.get("kernel", "") and \
                c.get("stream_via", "klcp") == (args.stream_via if args.workload == "stream" else "klcp"):
            best = t
    return None if best is None else best["hbm_bytes_per_launch"]


def measure_traffic(args, n, batch, k):
    """HBM bytes per launch of the dictionary's lookup kernel, measured NOW: this script is run twice more as a child under
    `rocprofv3 --pmc` -- FETCH_SIZE and WRITE_SIZE cannot share a pass (MI355X_MICROARCH.md, PMC slots) and counters are collected in
    runs of their own -- with the same index, queries and flags (three launches), and the per-dispatch averages of
    dict_lookup_kernel<false, ...> are corrected as that guide prescribes: gfx950 tallies a 128-byte request at 64 B in FETCH_SIZE and
    every read of this kernel is one (buckets by eight lanes x 16 B, k-mers as 512-B wave reads), so fetched = 2 x raw; WRITE_SIZE is
    exact.  Returns (bytes per launch, description) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    child = [sys.executable, str(REPO / "bench.py"), "--pmc-child", "--steps", "3", "--n", repr(float(n)), "--batch", repr(float(batch)), "--k", str(k),
             "--seed", str(args.seed), "--query-seed", str(args.query_seed), "--mode", args.mode, "--accel", "dict", "--workload", args.workload,
             "--read-len", str(args.read_len), "--subst-ppm", str(args.subst_ppm)] + (["--dict-sparse"] if args.dict_sparse else []) + \
            (["--repeat-permille", str(args.repeat_permille), "--repeat-families", str(args.repeat_families),
              "--repeat-divergence-permille", str(args.repeat_divergence_permille)] if args.repeat_permille else [])
    import fmsi_amd as fm
    fm.trim_scratch(int(os.environ.get("LOCAL_RANK", "0")))   # the child builds the same index beside this process: give it the room
    # the timed kernel's name as rocprofv3 prints it (the construction of a chained dictionary also runs dict_lookup_kernel<false, true>
    # once, over the text: not that one)
    if args.workload == "stream":
        kernel_tag = "dict_chain_kernel<false>" if args.dict_chain else "dict_lookup_kernel<false, true>"
        child += [] if args.dict_chain else ["--no-dict-chain"]
    else:
        kernel_tag = "dict_lookup_kernel<false, false>"
    got = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            try:
                res = subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "run", "--"] + child, cwd="/tmp", env=env,
                                     stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, f"rocprofv3 --pmc {counter}: {e!r}"
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for r in csv.DictReader(fh):
                        if kernel_tag in r["Kernel_Name"] and r["Counter_Name"] == counter:
                            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            if res.returncode != 0 or not per:
                return None, f"rocprofv3 --pmc {counter}: rc {res.returncode}, {len(per)} dispatches ({' '.join(l for l in res.stderr.decode(errors='replace').splitlines() if not l[:1] in 'WIE' or 'rror' in l)[-600:]})"
            got[counter] = sum(per.values()) / len(per) * 1024.0   # the counters are in KiB
    return 2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"], (
        f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (a child run each, 3 launches of {kernel_tag}), "
        f"2 x raw FETCH_SIZE {got['FETCH_SIZE'] / 1e9:.3f} GB (gfx950 tallies a 128-B request at 64 B) + WRITE_SIZE {got['WRITE_SIZE'] / 1e9:.3f} GB")


def run_e2e(args):
    """--workload e2e: one JSON line for the whole-process rate of the drop-in CLI (default mode), the other modes in `config`."""
    import subprocess
    cmd = [sys.executable, str(REPO / "tools" / "e2e_bench.py"), "--n", str(int(args.n)), "--reads", str(args.reads), "--read-len", str(args.read_len),
           "--k", str(args.k), "--modes", "or,all,all_S,xor,or_full", "--ref-reads", "20"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    report = None
    for l in res.stdout.splitlines():
        if l.startswith("{"):
            report = json.loads(l)
    if res.returncode != 0 or report is None:
        sys.stderr.write(res.stderr[-2000:])
        raise RuntimeError("tools/e2e_bench.py failed")
    m = report["modes"]["or"]
    modes = {name: {"wall_seconds": e["wall_seconds"], "process_seconds": e.get("total_seconds"), "load_seconds": e.get("load_seconds"),
                    "windows_per_s_whole_process": e.get("windows_per_s_wall"), "windows_per_s_after_load": e.get("windows_per_s_after_load"),
                    "identical_to_reference_on_sample": e.get("sample_identical_to_reference"), "reference_windows_per_s": e.get("reference_windows_per_s")}
             for name, e in report["modes"].items()}
    line = {"metric": "k-mer membership queries/sec (k=31), whole `fmsi query` process", "value": m["windows_per_s_wall"], "unit": "k-mers/s", "n_gpus": 1,
            "steps": 1, "warmup": 0, "ms_per_step": m["wall_seconds"] * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"whole process: fmsi query -q <{report['fasta_bytes'] / 1e9:.2f} GB FASTA, {args.reads} reads of {args.read_len} bases> on a synthetic "
                                   f"index of {int(args.n)} rows written in the reference's format ({report['index_bytes'] / 1e9:.2f} GB of files in /dev/shm), "
                                   f"output to a file; wall time of the process incl. index load, parse, device, print",
                       "windows": m["windows"], "modes": modes},
            "roofline": None, "cpu_baseline": {"value": m.get("reference_windows_per_s"), "unit": "k-mers/s", "cores": 1, "kind": "reference",
                                               "sample": "the reference CLI (oracle/_ref/fmsi_ref) on the first 20 reads of the same file, same index files"}}
    print(json.dumps(line))


def main():
    args = parse_args()
    if args.workload == "e2e":
        return run_e2e(args)
    # stdout carries the ONE JSON line and nothing else: libraries that print to the C stdout (RCCL's version banner at
    # communicator creation, flushed at exit) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    import fmsi_amd as fm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1 or args.force_dist   # the multi-rank code path
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev_t = torch.device("cuda", local_rank)
    stream_mode = args.workload == "stream"

    n = int(args.n)
    k = args.k
    # the dictionary answers presence queries of the phased workload and of `--stream-via runs`; the kLCP kernel and the A/B kernels
    # (--kernel pair | refill | lockstep) run without it
    use_dict = args.accel == "dict" and k <= 32 and ((not stream_mode and args.kernel == "phased") or (stream_mode and args.stream_via == "runs"))
    if args.hbm_budget > 0 and use_dict:
        # footprints: 4 bits per row (single-step blocks) + 8 bits per row (pair layout) + 2 bits per base (text) + the accelerator;
        # the dictionary: 128 bytes x 2^b buckets, b the smallest with <= 10.5 (sparse: 5.25) distinct k-mers per bucket, + ~3 % overflow
        base_gb = (n * 0.5 + n * 1.0 + n * 0.25) / 1e9
        b = 4
        while n / float(1 << b) > 10.5:
            b += 1
        dense_gb, sparse_gb = (1 << b) * 128 * 1.03 / 1e9, (1 << (b + 1)) * 128 * 1.01 / 1e9
        if args.hbm_budget >= base_gb + sparse_gb:
            args.dict_sparse = True
        elif args.hbm_budget < base_gb + dense_gb:   # no room for the dictionary: steps behind a prefix table of 8 x 4^t bytes
            use_dict = False
            args.no_verify = True
            t_fit = 0
            while t_fit < min(k, 16) and base_gb + 8 * 4 ** (t_fit + 1) / 1e9 <= args.hbm_budget:
                t_fit += 1
            if args.prefix_len < 0:
                args.prefix_len = t_fit
    args.use_dict = use_dict
    if use_dict and args.prefix_len < 0:
        args.prefix_len = 0   # no prefix table: the dictionary is the accelerator, the LF-walk legs run table-less anyway
    if args.prefix_len < 0:
        # Measured: the largest table, t = 16, is the fastest, and with the candidate tables (k - t <= 15: one
        # line settles a k-mer) that holds for ANY index size -- E. coli scale: 44 G k-mers/s with the 172 GB of t = 16 tables
        # against 26 G/s with cache-resident t = 12 ones.  The kLCP kernel and the table-less kernels keep the rule "no more
        # entries than 4 x rows".
        args.prefix_len = min(k, 16)
        candidates = args.kernel == "phased" and not args.no_verify and (args.workload == "kmers" or args.stream_via == "runs") and \
            1 <= k - args.prefix_len <= 15
        while not candidates and args.prefix_len > 0 and 4 ** args.prefix_len > 4 * n:
            args.prefix_len -= 1
    t0 = time.time()
    index = fm.DeviceIndex.build_synthetic(n, args.seed, k, with_klcp=stream_mode, device=local_rank, prefix_len=args.prefix_len,
                                           repeats=(args.repeat_permille, args.repeat_families, args.repeat_divergence_permille))
    build_s = time.time() - t0
    flags = fm.MODE_ALL if args.mode == "all" else fm.MODE_OR
    accel_s = 0.0
    if use_dict:
        t0 = time.time()
        if not stream_mode:
            index.enable_pair_layout()   # (the table-less LF-walk leg and the canonical-step count use it)
        index.enable_dict(sparse=args.dict_sparse, chain=bool(args.dict_chain and stream_mode))
        accel_s = time.time() - t0
        build_s += accel_s
    if stream_mode and args.stream_via == "runs" and not use_dict:
        index.enable_pair_layout()
        if 0 < args.prefix_len < k and not args.no_verify:
            index.enable_verify()
    if not stream_mode and not use_dict:
        if args.kernel in ("pair", "phased"):
            t0 = time.time()
            index.enable_pair_layout()
            if args.kernel == "phased" and 0 < args.prefix_len < k and not args.no_verify:
                try:
                    index.enable_verify()
                except fm.FmsiError as e:   # an accelerator: without it the run is slower, not wrong
                    print(f"warning: text verification is off ({e})", file=sys.stderr)
            build_s += time.time() - t0
        flags |= {"phased": fm.KERNEL_PHASED, "pair": fm.KERNEL_PAIR, "refill": fm.KERNEL_REFILL, "lockstep": fm.KERNEL_LOCKSTEP}[args.kernel]
    if use_dict:
        flags |= fm.KERNEL_DICT
    stream = torch.cuda.current_stream().cuda_stream

    # ---- one step's input, resident in HBM ----------------------------------------------------------------
    if stream_mode:
        L = args.read_len
        n_reads = max(1, int(args.batch) // (L - k + 1))
        batch = n_reads * (L - k + 1)                       # windows per step and GPU
        run_start = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
        run_windows = np.full(n_reads, L - k + 1, dtype=np.uint32)
        in_words = n_reads * L // 32 + 2
        in_bytes = n_reads * L / 4.0

        def make_input(step, buf):
            index.synthetic_reads_dev(L, args.query_seed, (step * world + rank) * n_reads, n_reads, buf.data_ptr(), args.subst_ppm, stream)

        def launch(buf, out, counters_ptr=0, extra=0):
            fn = index.query_runs_dev if args.stream_via == "runs" else index.query_stream_dev
            fn(buf.data_ptr(), n_reads * L, run_start, run_windows, k, flags | extra, out.data_ptr(), counters_ptr, stream)
    else:
        batch = int(args.batch)
        in_words = batch
        in_bytes = 8.0 * batch

        def make_input(step, buf):
            # rank r owns the contiguous range [r*batch, (r+1)*batch) of every step's global batch (weak scaling)
            index.synthetic_queries_dev(k, args.query_seed, (step * world + rank) * batch, batch, buf.data_ptr(), stream)

        def launch(buf, out, counters_ptr=0, extra=0):
            index.query_kmers_dev(buf.data_ptr(), batch, k, flags | extra, out.data_ptr(), counters_ptr, stream)

    # results: bit-packed words (default) -- two device buffers, two page-locked host buffers, a copy stream -- or bytes
    bits = not args.byte_results
    out_flag = fm.OUT_BITS if bits else 0
    d_out = torch.empty(batch, dtype=torch.uint8, device=dev_t)                      # byte results (counted launches, checks)
    n_words = fm.packed_words(batch, 0)
    d_bits = [torch.zeros(n_words, dtype=torch.int64, device=dev_t) for _ in range(2)] if bits else None
    root_words = n_words * (world if multi else 1)
    d_gather = [torch.zeros(root_words, dtype=torch.int64, device=dev_t) for _ in range(2)] if (bits and multi and rank == 0) else None
    h_bits = [torch.zeros(root_words, dtype=torch.int64).pin_memory() for _ in range(2)] if (bits and rank == 0) else None
    copy_stream = torch.cuda.Stream(device=dev_t) if bits else None
    ev_kernel = [torch.cuda.Event() for _ in range(2)]
    ev_copied = [torch.cuda.Event() for _ in range(2)]
    d_cnt = torch.zeros(fm.N_COUNTERS, dtype=torch.int64, device=dev_t)
    n_buf = max(1, min(args.steps, 4))
    bufs = [torch.empty(in_words, dtype=torch.int64, device=dev_t) for _ in range(n_buf)]

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pmc_child:   # under rocprofv3 --pmc: the timed kernel's launches and nothing else
        make_input(0, bufs[0])
        for _ in range(args.steps):
            launch(bufs[0], d_bits[0] if bits else d_out, extra=out_flag)
        torch.cuda.synchronize()
        os.close(json_fd)
        return
    # ---- warmup, plus counted launches that supply the algorithmic byte counts ----------------------------
    make_input(0, bufs[0])
    for _ in range(max(args.warmup, 1)):
        launch(bufs[0], d_out)
    counters = canonical = None
    if not args.no_counters:
        d_cnt.zero_()
        launch(bufs[0], d_out, d_cnt.data_ptr())
        torch.cuda.synchronize()
        counters = d_cnt.cpu().numpy().astype(np.int64)
        # what the reference algorithm executes for the same queries: no prefix table (and, for the stream workload,
        # its own chunking is longer than a 64-window task, so this is an upper bound on its steps)
        d_cnt.zero_()
        launch(bufs[0], d_out, d_cnt.data_ptr(), fm.NO_PREFIX_TABLE)
        torch.cuda.synchronize()
        canonical = d_cnt.cpu().numpy().astype(np.int64)
    positives = float(d_out.to(torch.float32).mean().item())

    for s in range(n_buf):
        make_input(s + 1, bufs[s])
    compute_stream = torch.cuda.current_stream()

    def step(s):
        """One pass of the hot path over one resident batch; with packed results also their way to the host: the kernel writes
        d_bits[b], the copy stream -- behind the kernel's event -- gathers the ranks' words to rank 0 (N > 1) and copies them
        to page-locked memory while the next step's kernel runs; a buffer is reused when its copy is done."""
        if not bits:
            launch(bufs[s % n_buf], d_out)
            return
        b = s & 1
        compute_stream.wait_event(ev_copied[b])
        launch(bufs[s % n_buf], d_bits[b], extra=out_flag)
        ev_kernel[b].record(compute_stream)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ev_kernel[b])
            if multi:
                dist.gather(d_bits[b], list(d_gather[b].chunk(world)) if rank == 0 else None, dst=0)
                if rank == 0:
                    h_bits[b].copy_(d_gather[b], non_blocking=True)
            else:
                h_bits[b].copy_(d_bits[b], non_blocking=True)
            ev_copied[b].record(copy_stream)
        # (hipStreamQuery: hands the copy to the hardware now -- the runtime batches enqueued commands, and a host thread that next
        # blocks inside a library call would leave the copy unsubmitted until it wakes; measured: 0.5 ms per step in the stream workload)
        copy_stream.query()

    if bits:   # one untimed pass through the copy path (first use of the copy stream, of the gather's communicator, of the pinned pages)
        for b in range(2):
            ev_copied[b].record(compute_stream)
        step(0)
        step(1)
    index.profile(True)
    index.profile_read(reset=True)
    barrier()   # (the first collective of the run also builds RCCL's communicator: twice, so that the timed one is warm)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s)
    enqueue_s = time.perf_counter() - t0   # (host time to enqueue the steps: the launches are asynchronous)
    barrier()   # (torch.cuda.synchronize() inside: every stream of the device, the copy stream included)
    elapsed = time.perf_counter() - t0
    launches, kernel_ms = index.profile_read(reset=True)
    index.profile(False)
    # the packed words that reached the host are the byte results of the same batch
    bits_check = None
    if bits and rank == 0:
        last = args.steps - 1
        launch(bufs[last % n_buf], d_out)
        torch.cuda.synchronize()
        got = fm.unpack_bits(h_bits[last & 1][:n_words].numpy().view(np.uint64), batch, 0)
        bits_check = bool(np.array_equal(got, d_out.cpu().numpy()))
        if not bits_check:
            raise RuntimeError("bit-packed results on the host differ from the byte results of the same batch")

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
    if multi:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed_max = float(el.item())

    # gather the last step's results to rank 0 (rank order = output order); outside the timed region
    gather_note = None
    if multi and bits and rank == 0:   # the gather ran inside the timed region: rank 0's own words must have come through it
        if not torch.equal(d_gather[(args.steps - 1) & 1][:n_words], d_bits[(args.steps - 1) & 1]):
            raise RuntimeError("gathered results differ from rank 0's own")
        gather_note = {"in_timed_region": True, "bytes_per_step": int(root_words * 8), "via": "torch.distributed.gather (RCCL)"}
    if multi and not bits:
        parts = [torch.empty_like(d_out) for _ in range(world)] if rank == 0 else None
        dist.gather(d_out, parts, dst=0)
        if rank == 0 and not torch.equal(parts[0], d_out):
            raise RuntimeError("gathered results differ from rank 0's own")
    if args.abi_gather:
        # the same gather through the C-ABI (fmsi_hip_gather over its own RCCL communicator), checked against the above;
        # a mismatch or a failure ends the run with an error (the flag is an explicit request for this check)
        from fmsi_amd import sharding
        last_b = (args.steps - 1) & 1
        src = d_bits[last_b].view(torch.uint8) if bits else d_out
        comm = sharding.comm_from_process_group(local_rank) if multi else fm.Comm(fm.Comm.unique_id(), 1, 0, local_rank)
        t_g = time.perf_counter()
        whole = sharding.gather_to_root_abi(comm, src, src.numel() * world, dst=0)
        torch.cuda.synchronize()
        t_g = time.perf_counter() - t_g
        if rank == 0:
            if bits:
                ref = d_gather[last_b].view(torch.uint8) if multi else src
            else:
                ref = torch.cat(parts) if multi else d_out
            if not torch.equal(whole, ref):
                raise RuntimeError("fmsi_hip_gather: MISMATCH with torch.distributed.gather")
            gather_note = dict(gather_note or {}, fmsi_hip_gather="matches torch.distributed.gather", bytes=int(whole.numel()),
                               seconds=round(t_g, 4))
        comm.close()

    # ---- the table-less LF walk (north_star's form: update_range steps only, src/fms_index.h:98-124) on the same index --------
    # Timed legs of their own, rank 0 of the 1-GPU run: the two-base-step pair kernel and the lock-step baseline, both with
    # FMSI_HIP_NO_PREFIX_TABLE, same resident batch, kernel time from the library's HIP events on the launch stream.
    lf_path = None
    if not args.no_lf_leg and not stream_mode and world == 1 and not args.no_counters and index.has_pair_layout:
        lf_path = []
        lf_out = d_bits[0] if bits else d_out
        for name, kflag, block_b in (("query_pair_kernel<false,false,false> (two-base steps, 128-B blocks, no table)", fm.KERNEL_PAIR, 128),
                                     ("query_lockstep_kernel<false> (one lane per k-mer, 64-B blocks, no table)", fm.KERNEL_LOCKSTEP, 64)):
            base_flags = (flags & ~0xF0) | kflag | fm.NO_PREFIX_TABLE
            d_cnt.zero_()
            index.query_kmers_dev(bufs[0].data_ptr(), batch, k, base_flags, d_out.data_ptr(), d_cnt.data_ptr(), stream)
            torch.cuda.synchronize()
            c = d_cnt.cpu().numpy().astype(np.int64)
            index.profile(True)
            index.profile_read(reset=True)
            torch.cuda.synchronize()
            t_l = time.perf_counter()
            for s_ in range(args.lf_steps):
                index.query_kmers_dev(bufs[s_ % n_buf].data_ptr(), batch, k, base_flags | out_flag, lf_out.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            t_l = time.perf_counter() - t_l
            l_n, l_ms = index.profile_read(reset=True)
            index.profile(False)
            k_s = l_ms / max(l_n, 1) / 1e3
            # the layout's own bytes: one block per boundary read (128 B pair / 64 B single; [1] counts them, explicit mask
            # probes included), the k-mers in, the packed results out; and SURVEY 8(d)'s canonical 256 B per LF step
            alg = block_b * int(c[1]) + 8.0 * batch + (batch / 8.0 if bits else batch)
            canon = 256 * int(c[0]) + 64 * int(c[2]) + 8.0 * batch + batch / 8.0
            lf_path.append({"kernel": name, "prefix_table": 0, "ms": round(k_s * 1e3, 3), "launches": int(l_n),
                            "wall_ms_per_step": round(t_l / args.lf_steps * 1e3, 3),
                            "kmers_per_s": round(batch / k_s, 1), "lf_steps_per_kmer": round(int(c[0]) / batch, 3),
                            "block_reads_per_kmer": round(int(c[1]) / batch, 3), "block_bytes": block_b,
                            "algorithmic_bytes": int(alg), "achieved": round(alg / k_s / 1e9, 1), "frac": round(alg / k_s / 1e9 / HBM_PEAK_GBS, 4),
                            "random_requests_per_s": round(int(c[1]) / k_s / 1e9, 2),
                            "canonical_bytes": int(canon), "canonical_achieved": round(canon / k_s / 1e9, 1),
                            "canonical_frac": round(canon / k_s / 1e9 / HBM_PEAK_GBS, 4)})

    if rank == 0:
        total_q = batch * world * args.steps
        value = total_q / elapsed_max
        avg_kernel_s = kernel_ms / max(launches, 1) / 1e3
        roof = None
        if counters is not None:
            blocks_read, lookups, klcp_words = int(counters[1]), int(counters[4]), int(counters[5])
            verify_reads = int(counters[7])   # text verification: 4-byte position + 8..16 bytes of text, two requests per verified search
            # blocks_read already contains the prefix-table lines (one 64-B line per lookup); the pair layout reads 128-B blocks
            pair_blocks = (args.kernel in ("pair", "phased") and not stream_mode) or (stream_mode and args.stream_via == "runs")
            block_b = 128 if pair_blocks else 64
            # a table lookup is one 64-byte HBM access whatever the entry's size (8, 16 or 32 bytes): the memory delivers
            # nothing smaller (PMC, profiles/r01m_traffic_kmers.json: 60 B fetched per lookup of a 32-byte entry), so the line is
            # what is counted; useful_* below counts the entry's own bytes instead
            entry_b = int(index.info.candidate_entry_bytes)
            lane_lookups = (not stream_mode or args.stream_via == "runs") and args.kernel == "phased" and entry_b and k - int(index.info.prefix_len) <= 15
            lookup_b = 64
            entry_used_b = entry_b if lane_lookups else 64
            if use_dict:   # a lookup is one 128-byte bucket (sixteen 8-byte slots, eight lanes x 16 bytes in one instruction)
                lookup_b, entry_used_b, lane_lookups = 128, 128, False
            out_b = batch / 8.0 if bits else 1.0 * batch   # packed: one bit per window
            bytes_launch = block_b * (blocks_read - lookups) + lookup_b * lookups + 8 * klcp_words + in_bytes + out_b
            bytes_launch += 32 * verify_reads   # counted as two 32-byte sectors per verification
            # (phased form: the work items of the searches left to the stepping kernel -- 48 bytes for about 3 % of the k-mers -- are not counted)
            canon_bytes = 256 * int(canonical[0]) + 64 * int(canonical[2]) + 8 * int(canonical[5]) + in_bytes + batch // 8
            ach = bytes_launch / avg_kernel_s / 1e9
            traffic, traffic_source = None, None
            if use_dict and world == 1 and not args.no_pmc and not os.environ.get("FMSI_BENCH_NO_PMC"):
                traffic, traffic_source = measure_traffic(args, n, batch, k)
            if traffic is None:
                traffic_source = ("profiles/*traffic*.json (rocprofv3 PMC passes of the same configuration, committed)" +
                                  (f"; not measured in this run: {traffic_source}" if traffic_source else ""))
                traffic = pmc_traffic(args, n, batch, k)
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source if traffic is not None else None,
                    "traffic_over_algorithmic": round(traffic / bytes_launch, 4) if traffic else None,
                    "kernel": ("dict_chain_kernel" if (args.dict_chain and stream_mode) else "dict_lookup_kernel") if use_dict else ("query_stream_kernel" if stream_mode and args.stream_via == "klcp" else
                               ("lookup_lane_kernel" if lane_lookups else "pair_lookup_kernel") + " + query_pair_kernel"
                               if args.kernel == "phased" or stream_mode else f"query_{args.kernel}_kernel"),
                    "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": launches,
                    "algorithmic_bytes_per_launch": int(bytes_launch),
                    "bytes_per_kmer": round(bytes_launch / batch, 1),
                    "lf_steps_per_kmer": round(int(counters[0]) / batch, 3),
                    "blocks_per_kmer": round(blocks_read / batch, 3), "table_lookups_per_kmer": round(lookups / batch, 3),
                    "klcp_words_per_kmer": round(klcp_words / batch, 3), "verify_reads_per_kmer": round(verify_reads / batch, 3),
                    "random_requests_per_s": round((blocks_read + klcp_words + verify_reads) / avg_kernel_s / 1e9, 2),
                    # the ceiling this path runs against: random line requests per second, measured with
                    # tools/gather_bench on the same footprint (profiles/r01k_gather_entry.txt: 48 G/s for 16..64-byte reads on 69-137 GB)
                    # (an index that fits the caches -- E. coli scale -- is not bound by it: null)
                    "request_roofline": None if int(index.info.device_bytes) < (2 << 30) else
                                        {"achieved": round((blocks_read + klcp_words + verify_reads) / avg_kernel_s / 1e9, 2), "peak": 48.0,
                                         "unit": "G random requests/s",
                                         "frac": round((blocks_read + klcp_words + verify_reads) / avg_kernel_s / 1e9 / 48.0, 4)},
                    "table_entry_bytes": entry_used_b,
                    "useful_bytes_per_kmer": round((bytes_launch - (lookup_b - entry_used_b) * lookups) / batch, 1),
                    "useful_frac": round((bytes_launch - (lookup_b - entry_used_b) * lookups) / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 4),
                    "canonical_lf_steps_per_kmer": round(int(canonical[0]) / batch, 3),
                    "canonical_achieved": round(canon_bytes / avg_kernel_s / 1e9, 1),
                    "canonical_frac": round(canon_bytes / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 4),
                    "lf_path": lf_path}
        info = index.info
        if stream_mode:
            scale = "human-genome scale" if n >= 2e9 else ("pangenome scale" if n >= 5e8 else ("E. coli scale" if n <= 1e7 else "synthetic scale"))
            wl = (f"synthetic max-ones masked superstring n={n} ({scale}), reads of {args.read_len} bases (50% from the "
                  f"superstring, half of those reverse-complemented, {args.subst_ppm} substitutions per million), every {k}-mer window, "
                  f"fmsi query -S -O semantics (kLCP streaming)")
        else:
            scale = "human-genome scale" if n >= 2e9 else ("pangenome scale" if n >= 5e8 else ("E. coli scale" if n <= 1e7 else "synthetic scale"))
            wl = (f"synthetic max-ones masked superstring n={n} ({scale}), isolated random {k}-mers 50% present, "
                  f"fmsi query -O semantics")
        if args.repeat_permille:
            wl += (f"; {args.repeat_permille / 10:.0f}% of the superstring's 2048-base segments are copies of {args.repeat_families} repeat "
                   f"families at {args.repeat_divergence_permille / 10:.0f}% divergence")
        line = {
            "metric": "k-mer membership queries/sec (k=31)", "value": value, "unit": "k-mers/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": wl, "index_rows": int(info.n), "k": k, "queries_per_step_per_gpu": batch, "mode": args.mode,
                       "kernel": (("dictionary-chained-runs" if args.dict_chain else "dictionary-runs") if stream_mode else "dictionary") if use_dict else
                                 (("phased-runs" if args.stream_via == "runs" else "stream") if stream_mode else args.kernel),
                       "accelerator": ({"kind": "k-mer dictionary", "bytes": int(info.dict_bytes), "levels": int(info.dict_levels),
                                        "log2_buckets": int(info.dict_log2_buckets), "keys_per_level": [int(x) for x in info.dict_keys][:int(info.dict_levels)],
                                        "chain_bytes": int(info.dict_chain_bytes),
                                        "build_seconds": round(accel_s, 2)} if use_dict else
                                       {"kind": "prefix table + candidate tables", "prefix_table_len": int(info.prefix_len),
                                        "candidate_entry_bytes": int(info.candidate_entry_bytes)}),
                       "prefix_table_len": int(info.prefix_len), "index_bytes_hbm": int(info.device_bytes),
                       "candidate_entry_bytes": int(info.candidate_entry_bytes),
                       "block_bytes": 128 if (args.kernel in ("pair", "phased") and not stream_mode) or (stream_mode and args.stream_via == "runs") else 64,
                       "parallelism": f"query-sharded x{world}, index replicated", "positive_fraction": round(positives, 4),
                       "results": ("bit-packed (FMSI_HIP_OUT_BITS), copied to page-locked host memory inside the timed region"
                                   + (", after the RCCL gather to rank 0" if multi else "")) if bits else "one byte per window, left in HBM",
                       "results_on_host_match_bytes": bits_check, "host_enqueue_ms_per_step": round(enqueue_s / args.steps * 1e3, 3),
                       "index_build_seconds": round(build_s, 2)},
            "roofline": roof,
            "cpu_baseline": None,
        }
        if gather_note is not None:
            line["config"]["gather"] = gather_note
        if args.cpu_sample != 0 and world == 1:   # reported baseline: rank 0 of the 1-GPU run only
            try:
                import bench_cpu   # the only part of the benchmark that touches oracle/
                line["cpu_baseline"] = bench_cpu.run_stream(args, index, n, k) if stream_mode else bench_cpu.run(args, index, n, k)
            except Exception as e:   # the baseline is a reported figure, never a reason to lose the GPU line
                line["cpu_baseline"] = {"error": repr(e)}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
