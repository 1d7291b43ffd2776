#!/usr/bin/env python3
"""bench.py -- k-mer membership queries/s (k = 31) of the MI355X `fmsi query` hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  One "step" = one pass of the hot path over one batch of
synthetic queries that is already resident in HBM; W untimed steps, then exactly K timed steps bracketed by
barrier + synchronize, MAX over ranks, one JSON line from rank 0.

The line's `value` (headline; BASELINE.json configs 3/5's workload at N GPUs): a synthetic max-ones masked superstring of 2.8 x 10^9
bases, generated and indexed on the device and replicated per GPU; 2^27 isolated random 31-mers per step and GPU, half of them present
(half of those as the reverse complement), `fmsi query -O` semantics, answered by the handle's k-mer dictionary -- a hash table of the
answers the index gives, ONE 128-byte read and NO LF step per k-mer; the LF walk itself is timed in `roofline.lf_path`.
The seeded, counter-based query stream is sharded across ranks by contiguous ranges -- weak scaling, no collective inside the search.
Results are bit-packed (FMSI_HIP_OUT_BITS: what the reference prints, one symbol per window, src/fms_index.h:305-309) and their way to the
host is INSIDE the timed region: every step's words are gathered to rank 0's HBM over RCCL (N > 1) and copied to page-locked host memory on
a copy stream, double-buffered against the next step's kernel (`--host-copy per-rank`, default: every rank copies its own words over its
own PCIe link; `root`: rank 0 copies all N ranks' words).

Extra objects on the JSON line:
  roofline     -- HBM bound, for the headline's kernel.  `achieved` = algorithmic bytes of one launch / average kernel duration (HIP events
                  recorded on the launch stream inside the library).  `traffic` = HBM bytes per launch MEASURED IN THIS RUN: this script is
                  run twice more as a child under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in passes of their own, corrected as
                  MI355X_MICROARCH.md prescribes; `traffic_source` says how).  `lf_path` = the table-less LF walk on the same index in
                  timed legs of its own -- two-base steps (query_pair_kernel) and one lane per k-mer in lock-step (query_lockstep_kernel),
                  the path north_star names -- each with `algorithmic_incl_cache` (block reads x block bytes / time: includes the top of
                  the search tree, which the caches serve) and `hbm_frac` (PMC bytes of the same child runs / time / 8 TB/s).
  cpu_baseline -- the real reference (oracle/_ref, kind "reference") or the C restatement (kind "port") timed on this box's host cores
                  on a bounded sample of the same queries against the same index (bench_cpu.py).
  configs      -- (N = 1, default workload) one entry per further BASELINE configuration, each measured in a child process of this
                  script on a freshly built index, with kernel time, algorithmic bytes, PMC traffic (in-run child passes) and a sample
                  checked against the real reference:
                    c2  E. coli scale (n = 4.7 x 10^6), 10^8 isolated 31-mers per step: cache resident, L2-level GB/s, frac null
                    c3  pangenome scale (n = 1.4 x 10^9), 2^27 per step x 8 steps >= 10^9 queries
                    c4  human scale, reads of 10 kb with `-S -O` answers, three ways -- chained dictionary lookups, plain dictionary
                        lookups, the kLCP kernel (what -S means in the reference) -- each without and with 1 % substitutions
                    e2e the whole `fmsi query` process (index load + parse + device + print), default / -O / -S -O
                  `--no-config-legs` skips them; `--legs c2,c4` picks.
"""
import argparse
import json
import os
import re
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured streaming ceiling
REQUEST_PEAK_G = 48.0   # random line requests/s the chip serves (tools/gather_bench entry, profiles/r01k_gather_entry.txt)
MARKER_KERNELS = ("synth_queries_kernel", "synth_reads_kernel")   # every variant of a PMC child run starts by drawing its input


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["kmers", "stream", "e2e"], default="kmers",
                    help="e2e: the whole `fmsi query` process (tools/e2e_bench.py: index files and a FASTA of --reads reads of --read-len bases in "
                         "/dev/shm, the CLI with --stats, a sample against the real reference) -- SURVEY 8(d)(ii); N = 1 only")
    ap.add_argument("--reads", type=int, default=100000, help="e2e workload: reads in the query file")
    ap.add_argument("--e2e-kmer-records", type=int, default=100_000_000,
                    help="e2e legs: records of the second query file, ONE k-mer each (BASELINE configs 2 / 3 / 5's shape: 100 M random queries); 0 = none")
    ap.add_argument("--n", "--index-bases", dest="n", type=float, default=2.8e9, help="superstring length in bases (human scale: 2.8e9)")
    ap.add_argument("--batch", type=float, default=float(1 << 27), help="query windows per step and GPU")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--prefix-len", type=int, default=-1,
                    help="private prefix table length t (0 = none; default: 16; without candidate tables less while 4^t > 4n)")
    ap.add_argument("--kernel", choices=["phased", "pair", "refill", "lockstep"], default="phased",
                    help="phased = pair layout, table lookups in their own kernel (default); pair = the same in one kernel")
    ap.add_argument("--accel", choices=["compact", "dict", "tables"], default="compact",
                    help="private accelerator of the phased workload: compact = the super-k-mer dictionary ALONE (FMSI_HIP_DICT_SUPERKMER_ONLY: 16-byte "
                         "records filed by minimizer, one 128-byte bucket read per isolated k-mer, ~7 bytes of HBM per distinct k-mer; 23 <= k <= 32, "
                         "else dict); dict = the k-mer dictionary (fmsi_hip_index_enable_dict: one 128-byte bucket read per k-mer, ~13 bytes of HBM "
                         "per distinct k-mer); tables = prefix table + candidate tables (177 GB at t = 16)")
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
    ap.add_argument("--host-copy", choices=["per-rank", "root"], default="per-rank",
                    help="N > 1, packed results: per-rank = every rank copies ITS OWN step's words to its page-locked host buffer over its own "
                         "PCIe link while the RCCL gather brings all ranks' words to rank 0's HBM (both inside the timed region); root = rank 0 "
                         "copies all N ranks' words after the gather (one PCIe link carries N x 16.8 MB per step).  The other form is timed in a "
                         "short region of its own afterwards and reported in config.gather")
    ap.add_argument("--stagger-build", type=float, default=0.0, help="N > 1: rank r starts its index build r x this many seconds late")
    ap.add_argument("--no-lf-leg", action="store_true", help="skip the table-less LF-walk legs (roofline.lf_path)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (child runs of this script under rocprofv3 --pmc, FETCH_SIZE and "
                         "WRITE_SIZE in separate passes)")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the `configs` legs (BASELINE configs 2, 3, 4 and the whole process)")
    # (the whole-process legs first: a process that starts right after others released tens of GB of HBM sits out the driver's wipe in its first
    # hipMalloc -- DESIGN.md 3.7 / 8 -- and behind twenty leg processes that was 0.1-0.2 s of a 0.4 s run on some boxes)
    ap.add_argument("--legs", default="e2e,c2,c3,c3sk,c4,c4k,c4k16,c4k48,c4sk,c5dict", help="which `configs` legs to run, in this order")
    ap.add_argument("--leg", default=None, help=argparse.SUPPRESS)          # child mode: one leg, a JSON list on stdout
    ap.add_argument("--variants", default=None, help="with --leg: only these variants of the leg (comma-separated names), e.g. for a rocprofv3 run of one")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # child mode under rocprofv3: the plan's launches, nothing printed
    ap.add_argument("--pmc-plan", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--leg-handle", default=None, help=argparse.SUPPRESS)   # (a PMC child of a leg: which leg's handle it rebuilds)
    ap.add_argument("--lf-steps", type=int, default=3, help="timed launches per LF-walk leg")
    ap.add_argument("--stream-via", choices=["auto", "klcp", "runs"], default="auto",
                    help="stream workload: klcp = fmsi_hip_query_stream (the kLCP kernel, what -S means in the reference); runs = the same "
                         "windows through fmsi_hip_query_runs (the dictionary, or phased kernels + candidate tables; same answers -- the "
                         "presence answers of -S do not depend on the kLCP array); auto = runs with --accel dict, else klcp")
    args = ap.parse_args(argv)
    if args.stream_via == "auto":
        args.stream_via = "runs" if args.accel in ("dict", "compact") else "klcp"
    return args


# ======================================================================================================================================
# process / device context
# ======================================================================================================================================
class Ctx:
    pass


def open_ctx(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    import fmsi_amd as fm
    c = Ctx()
    c.np, c.torch, c.dist, c.fm = np, torch, dist, fm
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    c.rank = int(os.environ.get("RANK", "0"))
    c.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    c.multi = c.world > 1 or args.force_dist   # the multi-rank code path
    # one process per GPU of ONE node: more ranks than devices is a launch error, said once and before any rank waits for another
    # (a rank without a device of its own would otherwise die inside the process group's set-up, or -- worse -- share GPU 0 and report
    # N times the rate of one GPU).  FMSI_BENCH_SHARE_GPU=1: the tests' two ranks on one device.
    n_dev = fm.device_count()
    if (c.world > n_dev or c.local_rank >= n_dev) and not os.environ.get("FMSI_BENCH_SHARE_GPU"):
        if c.rank == 0:
            print(f"bench.py: {c.world} ranks were launched but this node has {n_dev} HIP device(s): one rank per GPU -- run with --gpus <= {n_dev}", file=sys.stderr)
        sys.exit(2)
    if c.multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", c.local_rank))
    if args.gpus != c.world and c.rank == 0 and c.world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {c.world}", file=sys.stderr)
    torch.cuda.set_device(c.local_rank)
    c.dev_t = torch.device("cuda", c.local_rank)
    # The queries run on a stream of their own, not on the legacy default stream: whatever a library puts on stream 0 -- RCCL's gather
    # does (measured: with the kernels on the default stream every step's gather waited for the NEXT step's kernel, and kernel, gather
    # and copy ran one after the other: 3.30 instead of 2.95 ms per step) -- orders itself against EVERY blocking stream's work.
    c.compute_stream = torch.cuda.Stream(device=c.dev_t)
    torch.cuda.set_stream(c.compute_stream)
    c.stream = c.compute_stream.cuda_stream
    return c


def barrier(c):
    if c.multi:
        c.dist.barrier()
    c.torch.cuda.synchronize()


# ======================================================================================================================================
# the handle: index + accelerators for what the run measures
# ======================================================================================================================================
def resolve_accelerators(args, n, k, stream_mode):
    """Which private accelerator answers the timed queries (sets args.use_dict, args.prefix_len): the dictionary answers presence queries of
    the phased workload and of `--stream-via runs`; the kLCP kernel and the A/B kernels (--kernel pair | refill | lockstep) run without."""
    use_dict = args.accel in ("dict", "compact") and k <= 32 and ((not stream_mode and args.kernel == "phased") or (stream_mode and args.stream_via == "runs"))
    # the compact structure answers isolated k-mers (the headline) at the k-mer dictionary's rate and half its HBM; reads of the stream
    # workload keep the k-mer dictionary with chained lookups unless --accel compact is given explicitly with --no-dict-chain
    args.compact = use_dict and args.accel == "compact" and 23 <= k <= 32 and not args.dict_sparse and not args.hbm_budget and \
        (not stream_mode or not args.dict_chain)
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
        # line settles a k-mer) that holds for ANY index size.  The kLCP kernel and the table-less kernels keep the rule "no more
        # entries than 4 x rows".
        args.prefix_len = min(k, 16)
        candidates = args.kernel == "phased" and not args.no_verify and (args.workload == "kmers" or args.stream_via == "runs") and \
            1 <= k - args.prefix_len <= 15
        while not candidates and args.prefix_len > 0 and 4 ** args.prefix_len > 4 * n:
            args.prefix_len -= 1


def build_handle(c, args, n, k, stream_mode, with_klcp=None, want_pair=None, chain=None, superkmer_only=False):
    """Synthetic superstring -> index -> accelerators, as one chain of builders (they share the library's block cache).
    Returns (index, seconds for everything, seconds for the accelerators)."""
    fm = c.fm
    if c.multi and args.stagger_build > 0:
        time.sleep(args.stagger_build * c.rank)
    with_klcp = stream_mode if with_klcp is None else with_klcp
    chain = bool(args.dict_chain and stream_mode) if chain is None else chain
    t0 = time.time()
    accel_s = 0.0
    with fm.builder_chain(c.local_rank):
        index = fm.DeviceIndex.build_synthetic(n, args.seed, k, with_klcp=with_klcp, device=c.local_rank, prefix_len=args.prefix_len,
                                               repeats=(args.repeat_permille, args.repeat_families, args.repeat_divergence_permille))
        if args.use_dict:
            t1 = time.time()
            if (not stream_mode) if want_pair is None else want_pair:
                index.enable_pair_layout()   # (the table-less LF-walk leg and the canonical-step count use it)
            try:
                index.enable_dict(sparse=args.dict_sparse, chain=chain and not (superkmer_only or args.compact), superkmer_only=superkmer_only or args.compact)
            except fm.FmsiError as e:
                if not args.compact or e.code != fm.E_RANGE:
                    raise
                # (a text whose minimizers are too heavy for the records: the k-mer dictionary the construction went through stays)
                print(f"warning: no super-k-mer dictionary for this text ({e}); the k-mer dictionary answers", file=sys.stderr)
                args.compact = False
            accel_s = time.time() - t1
        elif stream_mode and args.stream_via == "runs":
            index.enable_pair_layout()
            if 0 < args.prefix_len < k and not args.no_verify:
                index.enable_verify()
        elif not stream_mode and args.kernel in ("pair", "phased"):
            index.enable_pair_layout()
            if args.kernel == "phased" and 0 < args.prefix_len < k and not args.no_verify:
                try:
                    index.enable_verify()
                except fm.FmsiError as e:   # an accelerator: without it the run is slower, not wrong
                    print(f"warning: text verification is off ({e})", file=sys.stderr)
    c.torch.cuda.synchronize()
    return index, time.time() - t0, accel_s


# ======================================================================================================================================
# a "path": one way of answering one kind of batch on one handle
# ======================================================================================================================================
class QueryPath:
    """What one timed variant launches.  via:
         dict        isolated k-mers, fmsi_hip_query_kmers, the k-mer dictionary (headline)
         legacy      isolated k-mers, the kernel --kernel / --accel tables select
         lf_pair     isolated k-mers, two-base LF steps without any table     lf_lockstep  one lane per k-mer, lock-step, no table
         runs_chain  reads, fmsi_hip_query_runs, chained dictionary lookups   runs_plain   the same, one bucket read per window
         runs_tables reads, fmsi_hip_query_runs, candidate tables             klcp         reads, fmsi_hip_query_stream (kLCP kernel)"""

    def __init__(self, c, args, index, via, k, batch, read_len=10000, subst_ppm=0):
        fm, np = c.fm, c.np
        self.c, self.args, self.index, self.via, self.k, self.subst_ppm = c, args, index, via, k, subst_ppm
        self.stream_mode = via in ("runs_chain", "runs_plain", "runs_tables", "runs_sk", "klcp", "klcp_orders", "klcp_orders_fast")
        # `lookup -S`: 8-byte orders, one per window (--fast) or one per window and strand (the CLI's default: it replays the strand predictor)
        self.order_words = {"klcp_orders": 2, "klcp_orders_fast": 1}.get(via, 0)
        mode = fm.MODE_ALL if args.mode == "all" else fm.MODE_OR
        # (the kLCP kernel reads 128-byte S-blocks since round 3; FMSI_HIP_STREAM_V1 brings the 64-byte-block kernel back)
        self.block_b = 64 if via == "lf_lockstep" or (via.startswith("klcp") and os.environ.get("FMSI_HIP_STREAM_V1")) or \
            (via == "legacy" and args.kernel in ("refill", "lockstep")) else 128
        self.is_dict = via in ("dict", "dict_sk", "runs_chain", "runs_plain", "runs_sk")
        if self.stream_mode:
            L = self.read_len = read_len
            self.n_reads = max(1, int(batch) // (L - k + 1))
            self.batch = self.n_reads * (L - k + 1)                       # windows per step and GPU
            self.run_start = np.arange(self.n_reads, dtype=np.uint64) * np.uint64(L)
            self.run_windows = np.full(self.n_reads, L - k + 1, dtype=np.uint32)
            self.in_words = self.n_reads * L // 32 + 2
            self.in_bytes = self.n_reads * L / 4.0
            self.flags = mode | {"runs_chain": fm.KERNEL_DICT | fm.NO_SUPERKMER, "runs_plain": fm.KERNEL_DICT | fm.NO_DICT_CHAIN | fm.NO_SUPERKMER, "runs_tables": 0,
                                 "runs_sk": fm.KERNEL_DICT, "klcp": 0, "klcp_orders": 0, "klcp_orders_fast": 0}[via]
            if self.order_words:
                self.flags = fm.ORDERS | (fm.STRICT if self.order_words == 2 else 0)
            self.kernel = {"runs_chain": "dict_chain_kernel", "runs_plain": "dict_lookup_kernel", "runs_tables": "lookup_lane_kernel + query_pair_kernel", "runs_sk": "sk_runs_kernel",
                           "klcp": "query_stream_kernel" if os.environ.get("FMSI_HIP_STREAM_V1") else "query_stream2_kernel"}[via.split("_orders")[0]]
            self.pmc_kernel = {"runs_chain": "dict_chain_kernel<false>", "runs_plain": "dict_lookup_kernel<false, true>", "runs_tables": "lookup_lane_kernel<false", "runs_sk": "sk_runs_kernel<false>",
                               "klcp": "query_stream2_kernel<false"}[via.split("_orders")[0]]
        else:
            self.batch = int(batch)
            self.in_words = self.batch
            self.in_bytes = 8.0 * self.batch
            if via == "dict":
                self.flags, self.kernel, self.pmc_kernel = mode | fm.KERNEL_DICT, "dict_lookup_kernel", "dict_lookup_kernel<false, false>"
            elif via == "dict_sk":   # a handle that keeps the super-k-mer dictionary alone: every k-mer a one-window read (fmsi_hip.hip, sk_iso)
                self.flags, self.kernel, self.pmc_kernel = mode | fm.KERNEL_DICT, "sk_kmers_kernel", "sk_kmers_kernel<false>"
            elif via == "lf_pair":
                self.flags, self.kernel, self.pmc_kernel = mode | fm.KERNEL_PAIR | fm.NO_PREFIX_TABLE, "query_pair_kernel", "query_pair_kernel<false, false, false>"
            elif via == "lf_lockstep":
                self.flags, self.kernel, self.pmc_kernel = mode | fm.KERNEL_LOCKSTEP | fm.NO_PREFIX_TABLE, "query_lockstep_kernel", "query_lockstep_kernel<false>"
            elif via == "lf_sorted":
                self.flags, self.kernel, self.pmc_kernel = mode | fm.KERNEL_SORTED | fm.NO_PREFIX_TABLE, "sorted_walk_kernel", "sorted_walk_kernel<false>"
            else:
                self.flags = mode | {"phased": fm.KERNEL_PHASED, "pair": fm.KERNEL_PAIR, "refill": fm.KERNEL_REFILL, "lockstep": fm.KERNEL_LOCKSTEP}[args.kernel]
                entry_b = int(index.info.candidate_entry_bytes)
                lane = args.kernel == "phased" and entry_b and k - int(index.info.prefix_len) <= 15
                self.kernel = (("lookup_lane_kernel" if lane else "pair_lookup_kernel") + " + query_pair_kernel") if args.kernel == "phased" else f"query_{args.kernel}_kernel"
                self.pmc_kernel = "query_pair_kernel<false" if args.kernel in ("phased", "pair") else f"query_{args.kernel}_kernel<false"

    def make_input(self, step, buf):
        """rank r owns the contiguous range [r*batch, (r+1)*batch) of every step's global batch (weak scaling)"""
        c, a = self.c, self.args
        if self.stream_mode:
            self.index.synthetic_reads_dev(self.read_len, a.query_seed, (step * c.world + c.rank) * self.n_reads, self.n_reads, buf.data_ptr(), self.subst_ppm, c.stream)
        else:
            self.index.synthetic_queries_dev(self.k, a.query_seed, (step * c.world + c.rank) * self.batch, self.batch, buf.data_ptr(), c.stream)

    def launch(self, buf, out, counters_ptr=0, extra=0, kernel=None):
        """kernel: replaces the path's kernel selection bits for this launch (the canonical-step count: the per-k-mer walk)"""
        c = self.c
        flags = self.flags | extra
        if kernel is not None:
            flags = (flags & ~0xF0) | kernel
        if self.stream_mode:
            fn = self.index.query_stream_dev if self.via.startswith("klcp") else self.index.query_runs_dev
            fn(buf.data_ptr(), self.n_reads * self.read_len, self.run_start, self.run_windows, self.k, flags, out.data_ptr(), counters_ptr, c.stream)
        else:
            self.index.query_kmers_dev(buf.data_ptr(), self.batch, self.k, flags, out.data_ptr(), counters_ptr, c.stream)

    def describe(self):
        return {"via": self.via, "workload": "stream" if self.stream_mode else "kmers", "subst_ppm": self.subst_ppm}


def timed_region(c, args, path, steps, warmup, bits=True, host_copy="per-rank", want_counters=True, want_canonical=False, alt_region=False):
    """W untimed + exactly K timed steps of `path` on resident batches, bracketed by barrier + synchronize; with packed results the
    words' way to the host is inside the region (module docstring).  Returns a dict of raw measurements."""
    torch, np, fm, dist = c.torch, c.np, c.fm, c.dist
    index, batch = path.index, path.batch
    multi, world, rank = c.multi, c.world, c.rank
    orders = getattr(path, "order_words", 0)
    if orders:
        bits = False   # (8-byte orders stay in HBM: their way to the host would be PCIe's rate, not the kernel's)
    out_flag = fm.OUT_BITS if bits else 0
    d_out = torch.empty(batch * orders, dtype=torch.int64, device=c.dev_t) if orders else \
        torch.empty(batch, dtype=torch.uint8, device=c.dev_t)                          # byte results (counted launches, checks)
    n_words = fm.packed_words(batch, 0)
    d_bits = [torch.zeros(n_words, dtype=torch.int64, device=c.dev_t) for _ in range(2)] if bits else None
    root_words = n_words * (world if multi else 1)
    d_gather = [torch.zeros(root_words, dtype=torch.int64, device=c.dev_t) for _ in range(2)] if (bits and multi and rank == 0) else None
    # page-locked host buffers: rank 0 can take every rank's words (root funnel), every other rank its own (per-rank copies)
    h_bits = [torch.zeros(root_words if rank == 0 else n_words, dtype=torch.int64).pin_memory() for _ in range(2)] if bits else None
    copy_stream = torch.cuda.Stream(device=c.dev_t) if bits else None    # the gather (N > 1) and, without one, the copy to the host
    d2h_stream = torch.cuda.Stream(device=c.dev_t) if (bits and multi) else None   # per-rank copies run beside the gather
    ev_kernel = [torch.cuda.Event() for _ in range(2)]
    ev_copied = [torch.cuda.Event() for _ in range(2)]
    ev_d2h = [torch.cuda.Event() for _ in range(2)]
    d_cnt = torch.zeros(fm.N_COUNTERS, dtype=torch.int64, device=c.dev_t)
    n_buf = max(1, min(steps, 4))
    bufs = [torch.empty(path.in_words, dtype=torch.int64, device=c.dev_t) for _ in range(n_buf)]

    # ---- warmup, plus counted launches that supply the algorithmic byte counts ----------------------------
    path.make_input(0, bufs[0])
    for _ in range(max(warmup, 1)):
        path.launch(bufs[0], d_out)
    counters = canonical = None
    if want_counters:
        d_cnt.zero_()
        path.launch(bufs[0], d_out, d_cnt.data_ptr())
        torch.cuda.synchronize()
        counters = d_cnt.cpu().numpy().astype(np.int64)
        if want_canonical:
            # what the reference algorithm executes for the same queries: no prefix table (and, for the stream workload,
            # its own chunking is longer than a 64-window task, so this is an upper bound on its steps)
            d_cnt.zero_()
            path.launch(bufs[0], d_out, d_cnt.data_ptr(), fm.NO_PREFIX_TABLE,
                        kernel=fm.KERNEL_PAIR if (index.has_pair_layout and path.via != "klcp") else None)   # (one search per k-mer and strand: not the batch-sorted walk AUTO would take)
            torch.cuda.synchronize()
            canonical = d_cnt.cpu().numpy().astype(np.int64)
    if orders:
        positives = float((d_out.view(batch, orders).max(dim=1).values >= 0).to(torch.float32).mean().item())
    else:
        positives = float(d_out.to(torch.float32).mean().item())
    for s in range(n_buf):
        path.make_input(s + 1, bufs[s])
    compute_stream = torch.cuda.current_stream()
    ev_times = []   # per timed step: gather start / end on the copy stream, D2H start / end on the stream that copies

    def step(s, mode, record=False):
        """One pass of the hot path over one resident batch; with packed results also their way to the host: the kernel writes
        d_bits[b], the copy stream -- behind the kernel's event -- gathers the ranks' words to rank 0's HBM (N > 1) and copies them
        to page-locked memory while the next step's kernel runs; a buffer is reused when its copy is done."""
        if not bits:
            path.launch(bufs[s % n_buf], d_out)
            return
        b = s & 1
        compute_stream.wait_event(ev_copied[b])
        path.launch(bufs[s % n_buf], d_bits[b], extra=out_flag)
        ev_kernel[b].record(compute_stream)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if record else None   # gather start / end, D2H start / end
        if multi and mode == "per-rank":   # own words over the own PCIe link, beside the gather (the path's collective) on the copy stream
            with torch.cuda.stream(d2h_stream):
                d2h_stream.wait_event(ev_kernel[b])
                if record:
                    e[2].record(d2h_stream)
                h_bits[b][:n_words].copy_(d_bits[b], non_blocking=True)
                if record:
                    e[3].record(d2h_stream)
                ev_d2h[b].record(d2h_stream)
            d2h_stream.query()
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ev_kernel[b])
            if record:
                e[0].record(copy_stream)
            if multi:
                dist.gather(d_bits[b], list(d_gather[b].chunk(world)) if rank == 0 else None, dst=0)
            if record:
                e[1].record(copy_stream)
            if multi and mode == "per-rank":
                copy_stream.wait_event(ev_d2h[b])
            else:
                if record:
                    e[2].record(copy_stream)
                if not multi:
                    h_bits[b][:n_words].copy_(d_bits[b], non_blocking=True)
                elif rank == 0:   # root funnel: all N ranks' words over rank 0's PCIe link
                    h_bits[b].copy_(d_gather[b], non_blocking=True)
                if record:
                    e[3].record(copy_stream)
            if record:
                ev_times.append(e)
            ev_copied[b].record(copy_stream)
        # (hipStreamQuery: hands the copy to the hardware now -- the runtime batches enqueued commands, and a host thread that next
        # blocks inside a library call would leave the copy unsubmitted until it wakes; measured: 0.5 ms per step in the stream workload)
        copy_stream.query()

    def region(k_steps, mode, record):
        if bits:   # one untimed pass through the copy path (first use of the copy stream, of the gather's communicator, of the pinned pages)
            for b in range(2):
                ev_copied[b].record(compute_stream)
            step(0, mode)
            step(1, mode)
        index.profile(True)
        index.profile_read(reset=True)
        barrier(c)   # (the first collective of the run also builds RCCL's communicator: twice, so that the timed one is warm)
        barrier(c)
        t0 = time.perf_counter()
        for s in range(k_steps):
            step(s, mode, record)
        enqueue_s = time.perf_counter() - t0   # (host time to enqueue the steps: the launches are asynchronous)
        barrier(c)   # (torch.cuda.synchronize() inside: every stream of the device, the copy stream included)
        elapsed = time.perf_counter() - t0
        launches, kernel_ms = index.profile_read(reset=True)
        index.profile(False)
        el = torch.tensor([elapsed], dtype=torch.float64, device=c.dev_t)
        if multi:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), enqueue_s, launches, kernel_ms

    elapsed_max, enqueue_s, launches, kernel_ms = region(steps, host_copy, True)
    m = {"batch": batch, "steps": steps, "elapsed_max": elapsed_max, "enqueue_s": enqueue_s, "launches": launches, "kernel_ms": kernel_ms,
         "counters": counters, "canonical": canonical, "positives": positives, "bits": bits, "order_words": orders, "n_words": n_words, "root_words": root_words,
         "bits_check": None, "gather": None}
    # the packed words that reached the host are the byte results of the same batch
    if bits:
        last = steps - 1
        path.launch(bufs[last % n_buf], d_out)
        torch.cuda.synchronize()
        own_ok = multi and host_copy == "root" and rank != 0   # (root funnel: only rank 0 holds host copies)
        if not own_ok:
            got = fm.unpack_bits(h_bits[last & 1][:n_words].numpy().view(np.uint64), batch, 0)
            m["bits_check"] = bool(np.array_equal(got, d_out.cpu().numpy()))
            if not m["bits_check"]:
                raise RuntimeError("bit-packed results on the host differ from the byte results of the same batch")
    if multi and bits:
        ones = torch.ones(1, dtype=torch.int64, device=c.dev_t)
        dist.all_reduce(ones)
        ms_g = sum(e[0].elapsed_time(e[1]) for e in ev_times) / max(len(ev_times), 1)
        ms_c = sum(e[2].elapsed_time(e[3]) for e in ev_times) / max(len(ev_times), 1)
        g = {"in_timed_region": True, "ranks_seen": int(ones.item()), "world": world, "via": "torch.distributed.gather (RCCL) to rank 0's HBM",
             "host_copy": host_copy, "bytes_gathered_per_step": int(root_words * 8), "bytes_to_host_per_step_rank0": int((n_words if host_copy == "per-rank" else root_words) * 8),
             "gather_ms_per_step_rank0": round(ms_g, 4), "d2h_ms_per_step_rank0": round(ms_c, 4)}
        if rank == 0:   # the gather ran inside the timed region: rank 0's own words must have come through it
            if not torch.equal(d_gather[(steps - 1) & 1][:n_words], d_bits[(steps - 1) & 1]):
                raise RuntimeError("gathered results differ from rank 0's own")
        if alt_region:   # the other host-copy form, timed the same way over a short region of its own
            other = "root" if host_copy == "per-rank" else "per-rank"
            k2 = max(2, min(steps, 10))
            el2, _, _, _ = region(k2, other, False)
            g["alt"] = {"host_copy": other, "steps": k2, "ms_per_step": round(el2 / k2 * 1e3, 4), "value": batch * world * k2 / el2}
        m["gather"] = g
    m["_keep"] = (d_out, d_bits, d_gather, h_bits, bufs, d_cnt)   # (the caller's further checks use them)
    return m


def roofline_of(path, m, info):
    """The HBM roofline object of one measured path (without PMC traffic): the layout's own algorithmic bytes over the kernel's time."""
    batch, counters = m["batch"], m["counters"]
    avg_kernel_s = m["kernel_ms"] / max(m["launches"], 1) / 1e3
    if counters is None:
        return {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": path.kernel,
                "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": m["launches"]}
    blocks_read, lookups, klcp_words = int(counters[1]), int(counters[4]), int(counters[5])
    verify_reads = int(counters[7])   # text verification / chained side reads: counted as 32-byte sectors
    # blocks_read already contains the table lines / bucket reads (one per lookup)
    entry_b = int(info.candidate_entry_bytes)
    lookup_b, entry_used_b = 64, 64
    if path.is_dict:      # a lookup is one 128-byte bucket (sixteen 8-byte slots, eight lanes x 16 bytes in one instruction)
        lookup_b = entry_used_b = 128
    elif "lookup_lane" in path.kernel and entry_b:
        entry_used_b = entry_b   # a table lookup is one 64-byte HBM access whatever the entry's size; useful_* counts the entry's bytes
    out_b = batch / 8.0 if m["bits"] else 1.0 * batch   # packed: one bit per window
    if m.get("order_words"):
        out_b = 8.0 * m["order_words"] * batch              # the array of orders (filled with -1 on the device, then the orders found)
    bytes_launch = path.block_b * (blocks_read - lookups) + lookup_b * lookups + 8 * klcp_words + path.in_bytes + out_b + 32 * verify_reads
    ach = bytes_launch / avg_kernel_s / 1e9
    requests = blocks_read + klcp_words + verify_reads
    roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": None, "traffic_source": None, "traffic_over_algorithmic": None,
            "kernel": path.kernel, "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": m["launches"],
            "algorithmic_bytes_per_launch": int(bytes_launch), "bytes_per_kmer": round(bytes_launch / batch, 1),
            "lf_steps_per_kmer": round(int(counters[0]) / batch, 3), "blocks_per_kmer": round(blocks_read / batch, 3),
            "table_lookups_per_kmer": round(lookups / batch, 3), "klcp_words_per_kmer": round(klcp_words / batch, 3),
            "side_reads_per_kmer": round(verify_reads / batch, 3),
            "random_requests_per_s": round(requests / avg_kernel_s / 1e9, 2),
            # the ceiling the lookups run against: random line requests per second, measured with tools/gather_bench on the
            # same footprint (an index that fits the caches -- E. coli scale -- is not bound by it: null)
            "request_roofline": None if int(info.device_bytes) < (2 << 30) else
                                {"achieved": round(requests / avg_kernel_s / 1e9, 2), "peak": REQUEST_PEAK_G, "unit": "G random requests/s",
                                 "frac": round(requests / avg_kernel_s / 1e9 / REQUEST_PEAK_G, 4)},
            "table_entry_bytes": entry_used_b,
            "useful_bytes_per_kmer": round((bytes_launch - (lookup_b - entry_used_b) * lookups) / batch, 1)}
    if m["canonical"] is not None:   # SURVEY 8(d)'s reference-layout accounting: 256 B per LF step the reference would execute + 64 B per probe
        canonical = m["canonical"]
        canon_bytes = 256 * int(canonical[0]) + 64 * int(canonical[2]) + 8 * int(canonical[5]) + path.in_bytes + batch // 8
        roof["canonical_lf_steps_per_kmer"] = round(int(canonical[0]) / batch, 3)
        roof["canonical_bytes_per_launch"] = int(canon_bytes)
    return roof


# ======================================================================================================================================
# PMC traffic: child runs of this script under rocprofv3 --pmc
# ======================================================================================================================================
def fetch_factor(kernel_name):
    """FETCH_SIZE correction per kernel (MI355X_MICROARCH.md, HBM: gfx950 tallies a 128-byte request at 64 B -- 0.50x -- as it does wide
    coalesced streaming reads; random 64-byte reads are tallied at 1.015x: tools/gather_bench calibration, profiles/README.md r01).
    Returns (factor, 128-byte lines counted at half): kernels that MIX 128-byte block reads with small reads (the streaming kernel: S-blocks
    beside 8-byte table entries and task records) get raw + 64 B x their 128-byte line reads per launch instead of a factor."""
    if any(t in kernel_name for t in ("query_lockstep_kernel", "query_refill_kernel", "query_fn_kernel", "query_longk_kernel", "query_stream_kernel")):
        return 1.0 / 1.015, False
    if "query_stream2_kernel" in kernel_name:
        return 1.0, True
    return 2.0, False


def pmc_child(c, args, index, plan):
    """Under rocprofv3 --pmc: per variant of the plan its input (the marker dispatch) and three launches, nothing else."""
    torch, fm = c.torch, c.fm
    for v in plan:
        path = QueryPath(c, args, index, v["via"], args.k, v.get("batch", args.batch), args.read_len, v.get("subst_ppm", 0))
        buf = torch.empty(path.in_words, dtype=torch.int64, device=c.dev_t)
        out = torch.zeros(max(path.batch * path.order_words if getattr(path, "order_words", 0) else fm.packed_words(path.batch, 0), 1), dtype=torch.int64, device=c.dev_t)
        torch.cuda.synchronize()
        path.make_input(1, buf)
        for _ in range(3):
            path.launch(buf, out, extra=0 if getattr(path, "order_words", 0) else fm.OUT_BITS)
        torch.cuda.synchronize()
        del buf, out


def measure_traffic(args, plan, child_args):
    """HBM bytes per launch of every variant of `plan`, measured NOW: this script runs twice more as a child under `rocprofv3 --pmc` --
    FETCH_SIZE and WRITE_SIZE cannot share a pass (MI355X_MICROARCH.md, PMC slots) and counters are collected in runs of their own -- with
    the same index and flags; a variant's dispatches are those between its marker (the generator kernel that draws its input) and the
    next one.  Returns ({variant name: {...}}, None) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    child = [sys.executable, str(REPO / "bench.py"), "--pmc-child", "--pmc-plan", json.dumps(plan)] + child_args
    per_variant = [dict(name=v["name"], kernels={}, lines128=v.get("lines128")) for v in plan]
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            try:
                res = subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "run", "--"] + child, cwd="/tmp", env=env,
                                     stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=420)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, f"rocprofv3 --pmc {counter}: {e!r}"
            rows = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for r in csv.DictReader(fh):
                        if r["Counter_Name"] != counter:
                            continue
                        key = int(r["Dispatch_Id"])
                        name, val = rows.get(key, (r["Kernel_Name"], 0.0))
                        rows[key] = (name, val + float(r["Counter_Value"]))
            group = -1
            seen = 0
            for key in sorted(rows):
                name, val = rows[key]
                if any(mk in name for mk in MARKER_KERNELS):
                    group += 1
                    continue
                if "sblock_build_kernel" in name:   # (the streaming layout, built once on the handle's first -S call: construction, not a query's traffic)
                    continue
                if 0 <= group < len(plan) and ("fmsi::" in name or "rocprim::" in name):   # (the library's kernels and the sorts it calls; not torch's fill of the next variant's buffer)
                    kk = per_variant[group]["kernels"].setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "dispatches": 0})
                    kk[counter] += val * 1024.0   # the counters are in KiB
                    if counter == "FETCH_SIZE":
                        kk["dispatches"] += 1
                    seen += 1
            if res.returncode != 0 or group + 1 != len(plan) or not seen:
                return None, (f"rocprofv3 --pmc {counter}: rc {res.returncode}, {group + 1} of {len(plan)} variants seen "
                              f"({' '.join(l for l in res.stderr.decode(errors='replace').splitlines() if not l[:1] in 'WIE' or 'rror' in l)[-600:]})")
    out = {}
    for pv in per_variant:
        fetched = written = raw = 0.0
        by_kernel = {}
        for name, kk in pv["kernels"].items():
            f, add_lines = fetch_factor(name)
            if add_lines:
                fetched += 64.0 * float(pv.get("lines128") or 0)
            fetched += f * kk["FETCH_SIZE"] / 3.0
            raw += kk["FETCH_SIZE"] / 3.0
            written += kk["WRITE_SIZE"] / 3.0
            mk = re.search(r"(\w+_kernel(?:<[^(]*>)?)", name)
            short = mk.group(1) if mk else name[:70]
            by_kernel[short] = {"fetch_raw": round(kk["FETCH_SIZE"] / 3.0), "fetch_factor": round(f, 4), "write": round(kk["WRITE_SIZE"] / 3.0), "dispatches_per_launch": kk["dispatches"] / 3.0}
        out[pv["name"]] = {"traffic": fetched + written, "fetch_raw": raw, "fetch": fetched, "write": written, "by_kernel": by_kernel,
                           "source": (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (a child run each, 3 launches per variant); raw FETCH_SIZE "
                                      f"{raw / 1e9:.3f} GB -> {fetched / 1e9:.3f} GB (x2 where the kernel reads 128-B requests or wide coalesced streams, which gfx950 "
                                      f"tallies at half; /1.015 for random 64-B block reads; raw + 64 B per 128-B block read where a kernel mixes them with small reads) "
                                      f"+ WRITE_SIZE {written / 1e9:.3f} GB")}
    return out, None


def base_child_args(args, n, batch):
    a = ["--n", repr(float(n)), "--batch", repr(float(batch)), "--k", str(args.k), "--seed", str(args.seed), "--query-seed", str(args.query_seed),
         "--mode", args.mode, "--read-len", str(args.read_len), "--prefix-len", str(args.prefix_len), "--accel", args.accel]
    if args.dict_sparse:
        a.append("--dict-sparse")
    if args.repeat_permille:
        a += ["--repeat-permille", str(args.repeat_permille), "--repeat-families", str(args.repeat_families),
              "--repeat-divergence-permille", str(args.repeat_divergence_permille)]
    return a


# ======================================================================================================================================
# the whole process
# ======================================================================================================================================
def e2e_report(args, modes="or,all,all_S", kmer_records=0, kmer_modes="or,all", gz_modes=""):
    import subprocess
    cmd = [sys.executable, str(REPO / "tools" / "e2e_bench.py"), "--n", str(int(args.n)), "--reads", str(args.reads), "--read-len", str(args.read_len),
           "--k", str(args.k), "--modes", modes, "--ref-reads", "20"]
    if gz_modes:
        cmd += ["--gz-modes", gz_modes, "--compress-level", "1"]
    if kmer_records and args.k <= 32:
        cmd += ["--kmer-records", str(int(kmer_records)), "--kmer-modes", kmer_modes]
    res = subprocess.run(cmd, capture_output=True, text=True)
    report = None
    for l in res.stdout.splitlines():
        if l.startswith("{"):
            report = json.loads(l)
    if res.returncode != 0 or report is None:
        sys.stderr.write(res.stderr[-2000:])
        raise RuntimeError("tools/e2e_bench.py failed")
    def digest(e):
        return {"wall_seconds": e["wall_seconds"], "process_seconds": e.get("total_seconds"), "load_seconds": e.get("load_seconds"),
                "windows_per_s_whole_process": e.get("windows_per_s_wall"), "windows_per_s_after_load": e.get("windows_per_s_after_load"),
                "stages": {s: e.get(s) for s in ("read_seconds", "prepare_seconds", "device_seconds", "replay_seconds", "format_seconds", "write_seconds",
                                                 "startup_seconds", "teardown_seconds") if e.get(s) is not None},
                "dictionary": e.get("dictionary"), "kmer_batches": e.get("kmer_batches"), "windows": e.get("windows"),
                "identical_to_reference_on_sample": e.get("sample_identical_to_reference") and e.get("full_run_prefix_identical"),
                "reference_windows_per_s": e.get("reference_windows_per_s")}
    modes_out = {name: digest(e) for name, e in report["modes"].items()}
    for name, e in (report.get("kmer_modes") or {}).items():   # the query file of one k-mer per record (configs 2 / 3 / 5's shape)
        modes_out["isolated_" + name] = digest(e)
    for name, e in (report.get("gz_modes") or {}).items():     # the reads file as one ordinary gzip member
        modes_out[("zlib_" if "NO_PARALLEL_GUNZIP" in name else "") + name.split("+")[-1] + "_gz"] = digest(e)
    return report, modes_out


def run_e2e(args):
    """--workload e2e: one JSON line for the whole-process rate of the drop-in CLI (default mode), the other modes in `config`."""
    report, modes = e2e_report(args, "or,all,all_S,xor,or_full")
    m = report["modes"]["or"]
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


# ======================================================================================================================================
# the `configs` legs (children of the default run)
# ======================================================================================================================================
LEG_SPECS = {
    # name: (n, workload, batch per step, steps, variants [(variant name, via, subst_ppm)])
    "c2": (4.7e6, "kmers", 1e8, 5, [("c2", "dict", 0)]),
    "c3": (1.4e9, "kmers", float(1 << 27), 8, [("c3", "dict", 0)]),
    "c4": (2.8e9, "stream", float(1 << 27), 8, [("c4_dict_chain", "runs_chain", 0), ("c4_dict_chain_subst1pct", "runs_chain", 10000),
                                                 ("c4_dict_plain", "runs_plain", 0), ("c4_dict_plain_subst1pct", "runs_plain", 10000)]),
    # the kLCP kernel on a handle of its own: blocks, kLCP array, S-blocks and the prefix table at the depth the `fmsi` binary takes for a
    # streaming run at this scale (query_cli.cpp: t = 15, 8.6 GB -- an 11 GB handle) -- no dictionary
    "c4k": (2.8e9, "stream", float(1 << 27), 8, [("c4_klcp", "klcp", 0), ("c4_klcp_subst1pct", "klcp", 10000),
                                                  # `fmsi lookup -S` (k-mer orders: src/fms_index.h:147-157 inside :181-254) on the same handle: both strands'
                                                  # orders as the CLI asks for them (it replays the strand predictor), and --fast's one order per window
                                                  ("c4_klcp_lookup", "klcp_orders", 0), ("c4_klcp_lookup_fast", "klcp_orders_fast", 0)]),
    # the same kernel on the 16-deep table (34 GB; rounds 3-4 timed all kLCP legs there): what the extra 26 GB buy over the depth the `fmsi` binary takes
    "c4k16": (2.8e9, "stream", float(1 << 27), 8, [("c4_klcp_t16", "klcp", 0), ("c4_klcp_t16_subst1pct", "klcp", 10000)]),
    # `fmsi query -S -O` with k = 48 (the reference's __uint128_t k-mers, src/main.cpp:223-230): an index of its own, the same read mix
    "c4k48": (2.8e9, "stream", float(1 << 27), 8, [("c4_klcp_k48", "klcp", 0)]),
    # the same reads through the super-k-mer dictionary ALONE (FMSI_HIP_DICT_SUPERKMER_ONLY: the k-mer dictionary is released after the
    # construction -- the handle is the index's blocks + the records + the small aux dictionary)
    # (isolated 31-mers on the compact structure alone, config 5's per-GPU work, are the headline itself since round 4)
    "c4sk": (2.8e9, "stream", float(1 << 27), 8, [("c4_superkmer", "runs_sk", 0), ("c4_superkmer_subst1pct", "runs_sk", 10000)]),
    # config 5's per-GPU work on the k-mer dictionary (rounds 2-3's headline; round 4's headline is the compact structure): n = 2.8e9, isolated 31-mers
    "c5dict": (2.8e9, "kmers", float(1 << 27), 8, [("c5_kmer_dictionary", "dict", 0)]),
    # config 3 (isolated 31-mers, pangenome scale) on the compact structure alone
    "c3sk": (1.4e9, "kmers", float(1 << 27), 8, [("c3_superkmer", "dict_sk", 0)]),
}
LEG_TEXT = {
    "c2": "BASELINE config 2: E. coli single-genome scale (synthetic max-ones masked superstring n=4700000), k=31, 10^8 isolated random 31-mers per step "
          "(50% present), fmsi query -O semantics, k-mer dictionary; the handle (a few tens of MB) is cache resident: `l2_level_gbs` is the algorithmic "
          "rate, `frac` (of HBM) is not meaningful and null",
    "c3": "BASELINE config 3: E. coli pangenome scale (n=1400000000, ~1.4 G distinct 31-mers), k=31, 2^27 isolated random 31-mers per step x 8 steps "
          "(>= 10^9 queries, 50% present), fmsi query -O semantics, k-mer dictionary",
    "c4": "BASELINE config 4: human-genome scale (n=2800000000), k=31, reads of 10000 bases (50% from the superstring, half of those reverse-complemented), "
          "every 31-mer window, fmsi query -S -O answers",
}
LEG_TEXT["c4sk"] = LEG_TEXT["c4"]
LEG_TEXT["c4k"] = LEG_TEXT["c4"] + "; prefix table at the depth the `fmsi` binary takes for a streaming run (15: an 11 GB handle)"
LEG_TEXT["c4k16"] = LEG_TEXT["c4"] + "; 16-deep prefix table (a 37 GB handle: rounds 3-4's form of the kLCP legs)"
LEG_TEXT["c4k48"] = LEG_TEXT["c4"].replace("k=31", "k=48").replace("31-mer", "48-mer")
LEG_K = {"c4k48": 48}
LEG_TEXT["c3sk"] = LEG_TEXT["c3"].replace("k-mer dictionary", "super-k-mer dictionary ALONE (the handle holds no k-mer dictionary: every k-mer is a one-window read, its minimizer's bucket "
                                          "one 128-byte read, heavy / overflowed ones the small aux dictionary)")
LEG_TEXT["c5dict"] = ("BASELINE config 5's per-GPU work on the K-MER dictionary (a hash table of the index's answers, 13 B per k-mer: what rounds 2-3 quoted as the headline): "
                      "human-genome scale (n=2800000000), k=31, 2^27 isolated random 31-mers per step x 8 steps (50% present), fmsi query -O semantics")
LEG_TEXT["c5_superkmer"] = ("BASELINE config 5's per-GPU work (and the headline's): human-genome scale (n=2800000000), k=31, 2^27 isolated random 31-mers per step x 8 steps (50% present), "
                    "fmsi query -O semantics, super-k-mer dictionary ALONE (no k-mer dictionary in the handle)")
VIA_TEXT = {"runs_chain": "fmsi_hip_query_runs, chained dictionary lookups (58.7 GB handle): a lane follows the superstring from its task's first window",
            "runs_plain": "fmsi_hip_query_runs, one dictionary bucket read per window (FMSI_HIP_NO_DICT_CHAIN)",
            "klcp": "fmsi_hip_query_stream: the kLCP kernel (extend_range_with_klcp + one LF step per positive window, src/fms_index.h:106-115,181-254)",
            "klcp_orders": "fmsi_hip_query_stream with FMSI_HIP_ORDERS | FMSI_HIP_STRICT -- what `fmsi lookup -S` runs: the kLCP kernel's ORD instantiation, an 8-byte order per "
                           "window AND strand (16 B per window, left in HBM: not bit-packed, no host copy in the region)",
            "klcp_orders_fast": "fmsi_hip_query_stream with FMSI_HIP_ORDERS -- `fmsi lookup -S --fast`: one 8-byte order per window (left in HBM)",
            "runs_sk": "fmsi_hip_query_runs, super-k-mer dictionary alone (records filed by minimizer: a bucket read when a lane's minimizer changes; the handle holds no k-mer dictionary)",
            "dict": "fmsi_hip_query_kmers, k-mer dictionary"}


def run_leg(args):
    """Child mode (--leg NAME): builds the leg's handle once, times every variant on it, checks a sample of each against the real
    reference, frees the handle, measures the variants' HBM traffic (PMC grandchildren) and prints ONE JSON list."""
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    name = args.leg
    n, workload, batch, steps, variants = LEG_SPECS[name]
    if args.variants:
        variants = [v for v in variants if v[0] in args.variants.split(",")]
    n = int(n)
    args.n, args.batch, args.workload = float(n), batch, workload
    stream_mode = workload == "stream"
    sk_only = name.endswith("sk")
    klcp_only = name in ("c4k", "c4k16", "c4k48")
    if name in LEG_K:
        args.k = LEG_K[name]
    args.stream_via = "klcp" if klcp_only else "runs"
    args.accel = "tables" if klcp_only else "dict"   # (the legs say themselves which structures their handle keeps)
    if klcp_only:
        # (the kLCP kernel's fresh searches and certificates go through the prefix table: at the CLI's depth -- what ships --, c4k16 at 16)
        args.prefix_len = int(os.environ.get("FMSI_BENCH_KLCP_TABLE", "16" if name == "c4k16" else "15"))
    c = open_ctx(args)
    k = args.k
    resolve_accelerators(args, n, k, stream_mode)
    index, build_s, accel_s = build_handle(c, args, n, k, stream_mode, with_klcp=klcp_only, want_pair=False, chain=stream_mode and not sk_only and not klcp_only,
                                           superkmer_only=sk_only)
    info = index.info
    legs = []
    paths = []
    for vname, via, subst in variants:
        path = QueryPath(c, args, index, via, k, batch, args.read_len, subst)
        m = timed_region(c, args, path, steps, args.warmup, bits=True)
        roof = roofline_of(path, m, info)
        m.pop("_keep")
        value = m["batch"] * steps / m["elapsed_max"]
        resident = int(info.device_bytes) < (1 << 28)   # fits the Infinity Cache: HBM % is not meaningful
        legs.append({"name": vname, "config": LEG_TEXT.get(vname) or (LEG_TEXT[name] + ("; " + VIA_TEXT[via] if path.stream_mode else "") +
                                                                      (f"; {subst // 10000}% substitutions in the reads drawn from the superstring" if subst else "")),
                     "value": value, "unit": "k-mers/s", "steps": steps, "queries_per_step": m["batch"], "ms": round(m["elapsed_max"] / steps * 1e3, 4),
                     "kernel": roof["kernel"], "avg_kernel_ms": roof["avg_kernel_ms"], "algorithmic_bytes": roof.get("algorithmic_bytes_per_launch"),
                     "achieved_gbs": None if resident else roof["achieved"], "l2_level_gbs": roof["achieved"] if resident else None,
                     "frac": None if resident else roof["frac"], "traffic": None, "traffic_over_algorithmic": None, "traffic_source": None,
                     "requests_per_kmer": round(roof.get("blocks_per_kmer", 0) + roof.get("klcp_words_per_kmer", 0) + roof.get("side_reads_per_kmer", 0), 3),
                     "lf_steps_per_kmer": roof.get("lf_steps_per_kmer"), "random_requests_per_s": roof.get("random_requests_per_s"),
                     "positive_fraction": round(m["positives"], 4), "results_on_host_match_bytes": m["bits_check"],
                     "index_rows": int(info.n), "handle_bytes_hbm": int(info.device_bytes), "dict_bytes": int(info.dict_bytes), "dict_chain_bytes": int(info.dict_chain_bytes),
                     "superkmer": None if not int(info.sk_bytes) else {"bytes": int(info.sk_bytes), "records": int(info.sk_records), "log2_buckets": int(info.sk_log2_buckets),
                                                                        "minimizer_len": int(info.sk_minimizer_len), "heavy_minimizers": int(info.sk_heavy_minimizers),
                                                                        "overflowed_buckets": int(info.sk_overflowed_buckets), "aux_dictionary_keys": int(info.sk_aux_keys)},
                     "index_build_seconds": round(build_s, 2), "matches_reference": None, "reference": None,
                     "_lines128": int(m["counters"][1]) - int(m["counters"][4]) if path.block_b == 128 and not path.is_dict else 0})
        paths.append(path)
        del m
    # ---- a sample of every variant against the real reference (bench_cpu.py: the only user of oracle/) ----
    try:
        import bench_cpu
        checks = bench_cpu.check_legs(c, args, index, paths, [v[0] for v in variants])
        for leg in legs:
            ch = checks.get(leg["name"])
            if ch:
                leg["matches_reference"], leg["reference"] = ch["matches"], ch["what"]
    except Exception as e:   # a check that could not run is reported as such, never as a match
        for leg in legs:
            leg["reference"] = f"not checked: {e!r}"
    del paths
    index.close()
    del index
    c.torch.cuda.empty_cache()
    c.fm.trim_scratch(c.local_rank)
    if not args.no_pmc and not os.environ.get("FMSI_BENCH_NO_PMC"):
        plan = [{"name": vn, "via": via, "subst_ppm": s, "batch": batch, "lines128": leg["_lines128"]} for (vn, via, s), leg in zip(variants, legs)]
        extra = ["--workload", workload, "--stream-via", args.stream_via, "--leg-handle", name] + base_child_args(args, n, batch)
        got, why = measure_traffic(args, plan, extra)
        for leg in legs:
            if got and leg["name"] in got:
                t = got[leg["name"]]
                leg["traffic"] = t["traffic"]
                leg["traffic_over_algorithmic"] = round(t["traffic"] / leg["algorithmic_bytes"], 4) if leg["algorithmic_bytes"] else None
                leg["traffic_source"] = t["source"]
                leg["traffic_by_kernel"] = t["by_kernel"]
                if leg["frac"] is not None:
                    leg["hbm_frac"] = round(t["traffic"] / (leg["avg_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            else:
                leg["traffic_source"] = f"not measured: {why}"
    for leg in legs:
        leg.pop("_lines128", None)
    os.write(json_fd, (json.dumps(legs) + "\n").encode())
    os.close(json_fd)


def run_config_legs(args):
    """The parent's side: one child process per leg (a fresh handle each; the parent has released its own)."""
    import subprocess
    out = []
    for name in [x for x in args.legs.split(",") if x]:
        t0 = time.time()
        if name == "e2e":
            try:
                report, modes = e2e_report(args, kmer_records=args.e2e_kmer_records, gz_modes="or,FMSI_HIP_NO_PARALLEL_GUNZIP=1+or")
                for mode_name, cli in (("or", "fmsi query"), ("all", "fmsi query -O"), ("all_S", "fmsi query -S -O"), ("isolated_or", "fmsi query"), ("isolated_all", "fmsi query -O"),
                                       ("or_gz", "fmsi query"), ("zlib_or_gz", "FMSI_HIP_NO_PARALLEL_GUNZIP=1 fmsi query")):
                    e = modes.get(mode_name)
                    if not e:
                        continue
                    isolated = mode_name.startswith("isolated_")
                    gz = mode_name.endswith("_gz")
                    what = (f"<{report['kmer_fasta_bytes'] / 1e9:.2f} GB FASTA: {report['kmer_records']} records of ONE {args.k}-mer each, half present (configs 2 / 3 / 5's query shape)>"
                            if isolated else f"<{report['fasta_bytes'] / 1e9:.2f} GB FASTA: {args.reads} reads of {args.read_len} bases>")
                    if gz:
                        what = (f"<the same reads as ONE ordinary gzip member (zlib level {report['gz']['level']}, {report['gz']['compressed_bytes'] / 1e9:.2f} GB): "
                                + ("zlib on one thread, as the reference reads it (src/parser.h:14-27)>" if mode_name.startswith("zlib_") else "inflated by the team, pgunzip.h>"))
                    out.append({"name": f"e2e_{mode_name}", "config": f"whole process `{cli} -q {what} "
                                                                      f"<index of {int(args.n)} rows, {report['index_bytes'] / 1e9:.2f} GB of .fmsi.* files in /dev/shm>`, output to a file, default options; "
                                                                      f"wall time of the process incl. index load, parse, device, print (SURVEY 8(d)(ii), src/main.cpp:238-375)",
                                "value": e["windows_per_s_whole_process"], "unit": "k-mers/s", "ms": round(e["wall_seconds"] * 1e3, 1), "windows": e.get("windows"),
                                "load_seconds": e["load_seconds"], "windows_per_s_after_load": e["windows_per_s_after_load"], "stages": e["stages"], "dictionary": e["dictionary"],
                                "kmer_batches": e.get("kmer_batches"), **({"compact_last": True} if gz else {}),
                                "kernel": None, "avg_kernel_ms": None, "algorithmic_bytes": None, "frac": None, "traffic": None,
                                "matches_reference": e["identical_to_reference_on_sample"],
                                "reference": f"the reference CLI on the first {'20000 records' if isolated else '20 reads'} of the same file, same index files: {e['reference_windows_per_s']} windows/s on one core; "
                                             f"byte-identical output (that sample, and the same prefix of the full run's output)"})
            except Exception as e:
                out.append({"name": "e2e", "error": repr(e)})
            out[-1]["leg_seconds"] = round(time.time() - t0, 1)
            continue
        if name not in LEG_SPECS:
            out.append({"name": name, "error": "unknown leg"})
            continue
        cmd = [sys.executable, str(REPO / "bench.py"), "--leg", name, "--warmup", str(args.warmup), "--seed", str(args.seed), "--query-seed", str(args.query_seed),
               "--mode", args.mode, "--k", str(args.k)] + (["--no-pmc"] if args.no_pmc else [])
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            got = None
            for l in res.stdout.splitlines():
                if l.startswith("["):
                    got = json.loads(l)
            if res.returncode != 0 or got is None:
                raise RuntimeError(f"rc {res.returncode}: {res.stderr[-800:]}")
            for g in got:
                g["leg_seconds"] = round(time.time() - t0, 1)
            out += got
        except Exception as e:
            out.append({"name": name, "error": repr(e)[:1200]})
    return out


# ======================================================================================================================================
# ---- the line the driver parses ---------------------------------------------------------------------------------------------
# The driver keeps the last 8 KB of stdout (its stderr trailer included) and parses the LAST line: that line is a compact summary of
# at most COMPACT_LIMIT bytes -- numbers and short names only.  The full record (every leg's counters, sources and prose) is written
# one line earlier on stdout and to bench_details.json beside this file (and under gpurun_out/ when that exists).
COMPACT_LIMIT = 4000


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def _short_kernel(name):
    """'query_pair_kernel<false,false,false> (two-base ...)' -> 'query_pair_kernel<false,false,false>'"""
    return None if name is None else str(name).split(" (")[0]


def compact_line(line):
    cfg, roof, cpu = line.get("config") or {}, line.get("roofline"), line.get("cpu_baseline")
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in line}
    out["value"], out["ms_per_step"] = _r(out["value"], 1), _r(out["ms_per_step"], 5)
    acc = cfg.get("accelerator") or {}
    c2 = {"workload": cfg.get("workload_short") or str(cfg.get("workload", ""))[:96], "index_rows": cfg.get("index_rows"), "k": cfg.get("k"),
          "queries_per_step_per_gpu": cfg.get("queries_per_step_per_gpu"), "mode": cfg.get("mode"), "kernel": cfg.get("kernel"),
          "accelerator": acc.get("kind"), "accelerator_bytes": acc.get("bytes"), "index_bytes_hbm": cfg.get("index_bytes_hbm"),
          "parallelism": cfg.get("parallelism"), "positive_fraction": cfg.get("positive_fraction"),
          "results_on_host_match_bytes": cfg.get("results_on_host_match_bytes"), "index_build_seconds": cfg.get("index_build_seconds")}
    g = cfg.get("gather")
    if g:
        c2["gather"] = {k: g[k] for k in ("in_timed_region", "ranks_seen", "world", "host_copy", "bytes_gathered_per_step", "gather_ms_per_step_rank0",
                                            "d2h_ms_per_step_rank0", "fmsi_hip_gather") if k in g}
        if "alt" in g:
            c2["gather"]["alt"] = {k: _r(g["alt"][k], 1) for k in ("host_copy", "ms_per_step", "value") if k in g["alt"]}
    out["config"] = c2
    if roof:
        r2 = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac", "traffic_over_algorithmic", "kernel",
                                       "avg_kernel_ms", "launches", "algorithmic_bytes_per_launch", "bytes_per_kmer", "blocks_per_kmer")}
        rr = roof.get("request_roofline")
        if rr:
            r2["requests_gps"], r2["requests_frac"] = rr.get("achieved"), rr.get("frac")
        if roof.get("lf_path"):
            r2["lf_path"] = [{"kernel": _short_kernel(l.get("kernel")), "ms": l.get("ms"), "kmers_per_s": l.get("kmers_per_s"),
                              "lf_steps_per_kmer": l.get("lf_steps_per_kmer"), "hbm_frac": l.get("hbm_frac"),
                              "hbm_lines_per_kmer": l.get("hbm_lines_per_kmer"), "hbm_glines_per_s": l.get("hbm_glines_per_s")} for l in roof["lf_path"]]
            r2["lf_path"] = [{k: v for k, v in l.items() if v is not None} for l in r2["lf_path"]]
        if isinstance(r2.get("traffic"), float):
            r2["traffic"] = int(r2["traffic"])
        for l in r2.get("lf_path") or []:
            l["kmers_per_s"] = int(l["kmers_per_s"])
        out["roofline"] = r2
    else:
        out["roofline"] = None
    if cpu:
        if "error" in cpu:
            out["cpu_baseline"] = {"error": str(cpu["error"])[:120]}
        else:
            c3 = {k: _r(cpu.get(k), 1) for k in ("value", "unit", "cores", "kind", "matches_gpu", "socket_value", "socket_cores", "tuned_value", "cpu_model")}
            c3["sample"] = cpu.get("sample_short") or str(cpu.get("sample", ""))[:80]
            out["cpu_baseline"] = c3
    else:
        out["cpu_baseline"] = None
    extra_legs = []   # (legs that enter the compact line only where it has room left: the full record has them)
    if line.get("configs"):
        legs = []
        for l in line["configs"]:
            if not isinstance(l, dict):
                continue
            if l.get("compact_last"):
                extra_legs.append({"name": l.get("name"), "value": None if l.get("value") is None else int(float("%.5g" % l["value"])), "matches_reference": l.get("matches_reference")})
                continue
            if "error" in l and "value" not in l:
                legs.append({"name": l.get("name"), "error": str(l["error"])[:60]})
                continue
            leg = {"name": l.get("name"), "value": None if l.get("value") is None else int(float("%.5g" % l["value"])), "ms": l.get("ms"),
                   "kernel": _short_kernel(l.get("kernel")),
                   "kernel_ms": l.get("avg_kernel_ms"), "frac": l.get("frac"), "hbm_frac": l.get("hbm_frac"),
                   "hbm_gb": None if not l.get("handle_bytes_hbm") else round(l["handle_bytes_hbm"] / 1e9, 1),
                   "after_load": None if not l.get("windows_per_s_after_load") else int(float("%.4g" % l["windows_per_s_after_load"])),
                   "load_s": None if l.get("load_seconds") is None else round(l["load_seconds"], 3), "dictionary": l.get("dictionary"),
                   "matches_reference": l.get("matches_reference")}
            legs.append({k: v for k, v in leg.items() if v is not None or k in ("name", "value", "matches_reference")})
        out["configs"] = legs

    def size(o):
        return len(json.dumps(o, separators=(",", ":")))
    # never above the limit: shed the least important fields first
    # (a leg keeps name, value, hbm_frac, hbm_gb -- the structure it was measured on --, the whole-process legs after_load / load_s, and matches_reference to the last)
    for drop in (("configs", "kernel"), ("configs", "kernel_ms"), ("configs", "dictionary"), ("roofline", "bytes_per_kmer"), ("roofline", "blocks_per_kmer"),
                 ("config", "gather", "alt"), ("cpu_baseline", "cpu_model"), ("cpu_baseline", "tuned_value"), ("configs", "ms"), ("configs", "frac"),
                 ("roofline", "launches"), ("config", "positive_fraction"), ("config", "index_build_seconds"), ("config", "accelerator_bytes"),
                 ("configs", "load_s"), ("configs", "hbm_gb")):
        if size(out) <= COMPACT_LIMIT:
            break
        tgt = out.get(drop[0])
        if isinstance(tgt, list):
            for t in tgt:
                t.pop(drop[1], None)
        elif isinstance(tgt, dict):
            for key in drop[1:-1]:
                tgt = tgt.get(key) or {}
            tgt.pop(drop[-1], None)
    while size(out) > COMPACT_LIMIT and out.get("configs"):
        out["configs"].pop()
    for leg in extra_legs:
        out["configs"].append(leg)
        if size(out) > COMPACT_LIMIT:
            out["configs"].pop()
            break
    return json.dumps(out, separators=(",", ":"))


def emit(fd, line):
    """stdout: the full record, then -- last -- the compact line the driver parses."""
    full = json.dumps(line)
    here = os.path.dirname(os.path.abspath(__file__))
    for d in (here, os.path.join(here, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_details.json"), "w") as f:
                    f.write(full + "\n")
        except OSError:
            pass
    small = compact_line(line)
    assert len(small) <= COMPACT_LIMIT and "\n" not in small
    os.write(fd, (full + "\n").encode())
    os.write(fd, (small + "\n").encode())


def main():
    args = parse_args()
    if args.workload == "e2e":
        return run_e2e(args)
    if args.leg:
        return run_leg(args)
    # stdout carries the ONE JSON line and nothing else: libraries that print to the C stdout (RCCL's version banner at
    # communicator creation, flushed at exit) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    c = open_ctx(args)
    np, torch, dist, fm = c.np, c.torch, c.dist, c.fm
    world, rank, multi = c.world, c.rank, c.multi
    stream_mode = args.workload == "stream"
    n, k = int(args.n), args.k
    resolve_accelerators(args, n, k, stream_mode)
    use_dict = args.use_dict

    if args.pmc_child:   # under rocprofv3 --pmc: the plan's launches and nothing else (the handle is built as the measured run built it)
        plan = json.loads(args.pmc_plan)
        index, _, _ = build_handle(c, args, n, k, stream_mode, with_klcp=any(v["via"].startswith("klcp") for v in plan),
                                   want_pair=any(v["via"].startswith("lf_") for v in plan), chain=any(v["via"] == "runs_chain" for v in plan),
                                   superkmer_only=any(v["via"] in ("runs_sk", "dict_sk") for v in plan))
        pmc_child(c, args, index, plan)
        os.close(json_fd)
        return

    index, build_s, accel_s = build_handle(c, args, n, k, stream_mode)
    # per-rank build time: with N ranks building at once (each on its own GPU) the slowest is what the run waits for
    build_all = [build_s]
    if multi:
        t = torch.tensor([build_s], dtype=torch.float64, device=c.dev_t)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        build_all = [float(p.item()) for p in parts]
    if stream_mode:
        via = ("runs_sk" if args.compact else ("runs_chain" if args.dict_chain else "runs_plain")) if use_dict else ("runs_tables" if args.stream_via == "runs" else "klcp")
    else:
        via = ("dict_sk" if args.compact else "dict") if use_dict else "legacy"
    path = QueryPath(c, args, index, via, k, args.batch, args.read_len, args.subst_ppm)
    batch = path.batch
    bits = not args.byte_results
    m = timed_region(c, args, path, args.steps, args.warmup, bits=bits, host_copy=args.host_copy, want_counters=not args.no_counters,
                     want_canonical=not args.no_counters, alt_region=world > 1)
    d_out, d_bits, d_gather, h_bits, bufs, d_cnt = m.pop("_keep")
    elapsed_max = m["elapsed_max"]
    gather_note = m["gather"]

    torch.cuda.synchronize()
    if multi and not bits:   # byte results: gathered after the timed region (the round-1 form)
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
        comm = sharding.comm_from_process_group(c.local_rank) if multi else fm.Comm(fm.Comm.unique_id(), 1, 0, c.local_rank)
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
            gather_note = dict(gather_note or {}, fmsi_hip_gather="matches torch.distributed.gather", bytes=int(whole.numel()), seconds=round(t_g, 4))
        comm.close()

    # ---- the table-less LF walk (north_star's form: update_range steps only, src/fms_index.h:98-124) on the same index --------
    # Timed legs of their own, rank 0 of the 1-GPU run: the two-base-step pair kernel and the lock-step baseline, both with
    # FMSI_HIP_NO_PREFIX_TABLE, same resident batches, kernel time from the library's HIP events on the launch stream.
    lf_path = None
    if not args.no_lf_leg and not stream_mode and world == 1 and not args.no_counters and index.has_pair_layout:
        lf_path = []
        lf_out = d_bits[0] if bits else d_out
        out_flag = fm.OUT_BITS if bits else 0
        for lf_via, label in (("lf_sorted", "sorted_walk_kernel<false> (the batch's strand searches sorted by their first bases, one lane per search in lock-step, two-base "
                                            "steps, no index-side table; items + radix sort + per-batch top of the search tree + walk + combine, all inside the timed launches)"),
                              ("lf_pair", "query_pair_kernel<false,false,false> (two-base steps, 128-B blocks, no table)"),
                              ("lf_lockstep", "query_lockstep_kernel<false> (one lane per k-mer walked in lock-step, 64-B blocks, no table)")):
            p = QueryPath(c, args, index, lf_via, k, batch)
            d_cnt.zero_()
            p.launch(bufs[0], d_out, d_cnt.data_ptr())
            torch.cuda.synchronize()
            cn = d_cnt.cpu().numpy().astype(np.int64)
            index.profile(True)
            index.profile_read(reset=True)
            torch.cuda.synchronize()
            t_l = time.perf_counter()
            for s_ in range(args.lf_steps):
                p.launch(bufs[s_ % len(bufs)], lf_out, 0, out_flag)
            torch.cuda.synchronize()
            t_l = time.perf_counter() - t_l
            l_n, l_ms = index.profile_read(reset=True)
            index.profile(False)
            k_s = l_ms / max(l_n, 1) / 1e3
            # the layout's own bytes: one block per boundary read (128 B pair / 64 B single; [1] counts them, explicit mask
            # probes included), the k-mers in, the packed results out; and SURVEY 8(d)'s canonical 256 B per LF step
            alg = p.block_b * int(cn[1]) + 8.0 * batch + (batch / 8.0 if bits else batch)
            canon = 256 * int(cn[0]) + 64 * int(cn[2]) + 8.0 * batch + batch / 8.0
            lf_path.append({"kernel": label, "via": lf_via, "prefix_table": 0, "ms": round(k_s * 1e3, 3), "launches": int(l_n),
                            "wall_ms_per_step": round(t_l / args.lf_steps * 1e3, 3),
                            "kmers_per_s": round(batch / k_s, 1), "lf_steps_per_kmer": round(int(cn[0]) / batch, 3),
                            "block_reads_per_kmer": round(int(cn[1]) / batch, 3), "block_bytes": p.block_b,
                            "algorithmic_bytes": int(alg),
                            # block reads x block bytes over the kernel's time: INCLUDES the reads the caches serve (the first four to five
                            # steps of every search touch 4^9 distinct blocks and fewer) -- not an HBM figure, may exceed the HBM peak
                            "algorithmic_incl_cache": {"gbs": round(alg / k_s / 1e9, 1), "over_hbm_peak": round(alg / k_s / 1e9 / HBM_PEAK_GBS, 4)},
                            "random_requests_per_s": round(int(cn[1]) / k_s / 1e9, 2),
                            "canonical_bytes": int(canon), "canonical_gbs": round(canon / k_s / 1e9, 1),
                            "hbm_traffic": None, "hbm_gbs": None, "hbm_frac": None})

    if rank != 0:
        os.close(json_fd)
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return

    total_q = batch * world * args.steps
    value = total_q / elapsed_max
    info = index.info
    roof = roofline_of(path, m, info) if m["counters"] is not None else None
    cpu_baseline = None
    if args.cpu_sample != 0 and world == 1:   # reported baseline: rank 0 of the 1-GPU run only
        try:
            import bench_cpu   # the only part of the benchmark that touches oracle/
            cpu_baseline = bench_cpu.run_stream(args, index, n, k) if stream_mode else bench_cpu.run(args, index, n, k)
        except Exception as e:   # the baseline is a reported figure, never a reason to lose the GPU line
            cpu_baseline = {"error": repr(e)}
    info_d = {"n": int(info.n), "device_bytes": int(info.device_bytes), "dict_bytes": int(info.dict_bytes), "dict_levels": int(info.dict_levels),
              "sk_bytes": int(info.sk_bytes), "sk_records": int(info.sk_records), "sk_log2_buckets": int(info.sk_log2_buckets), "sk_minimizer_len": int(info.sk_minimizer_len),
              "sk_heavy_minimizers": int(info.sk_heavy_minimizers), "sk_overflowed_buckets": int(info.sk_overflowed_buckets), "sk_aux_keys": int(info.sk_aux_keys),
              "dict_log2_buckets": int(info.dict_log2_buckets), "dict_keys": [int(x) for x in info.dict_keys][:int(info.dict_levels)],
              "dict_chain_bytes": int(info.dict_chain_bytes), "prefix_len": int(info.prefix_len), "candidate_entry_bytes": int(info.candidate_entry_bytes)}
    # ---- everything below runs in child processes: the handle goes first ----
    del path, d_out, d_bits, d_gather, h_bits, bufs, d_cnt
    index.close()
    del index
    torch.cuda.empty_cache()
    fm.trim_scratch(c.local_rank)

    if roof is not None and world == 1 and not args.no_pmc and not os.environ.get("FMSI_BENCH_NO_PMC") and via in ("dict", "dict_sk", "runs_sk", "runs_chain", "runs_plain", "klcp"):
        plan = [{"name": "headline", "via": via, "subst_ppm": args.subst_ppm, "batch": batch,
                 "lines128": int(m["counters"][1]) - int(m["counters"][4]) if via == "klcp" else 0}]
        for leg in (lf_path or []):
            plan.append({"name": leg["via"], "via": leg["via"], "batch": batch})
        extra = ["--workload", args.workload, "--stream-via", args.stream_via, "--subst-ppm", str(args.subst_ppm)] + ([] if args.dict_chain else ["--no-dict-chain"]) + \
            base_child_args(args, n, args.batch)
        got, why = measure_traffic(args, plan, extra)
        if got:
            t = got["headline"]
            roof["traffic"], roof["traffic_source"], roof["traffic_by_kernel"] = t["traffic"], t["source"], t["by_kernel"]
            roof["traffic_over_algorithmic"] = round(t["traffic"] / roof["algorithmic_bytes_per_launch"], 4)
            roof["hbm_frac"] = round(t["traffic"] / (roof["avg_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            for leg in (lf_path or []):
                tt = got.get(leg["via"])
                if tt:
                    leg["hbm_traffic"] = tt["traffic"]
                    leg["hbm_gbs"] = round(tt["traffic"] / (leg["ms"] * 1e-3) / 1e9, 1)
                    leg["hbm_frac"] = round(tt["traffic"] / (leg["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    # requests that reached the memory side, in the kernel's own line size, per k-mer and per second (the ceiling is ~48 G/s: DESIGN.md 3)
                    leg["hbm_lines_per_kmer"] = round(tt["traffic"] / leg["block_bytes"] / batch, 2)
                    leg["hbm_glines_per_s"] = round(tt["traffic"] / leg["block_bytes"] / (leg["ms"] * 1e-3) / 1e9, 2)
                    leg["traffic_by_kernel"] = tt["by_kernel"]
        else:
            roof["traffic_source"] = f"not measured in this run: {why}"
    if roof is not None:
        roof["lf_path"] = lf_path

    scale = "human-genome scale" if n >= 2e9 else ("pangenome scale" if n >= 5e8 else ("E. coli scale" if n <= 1e7 else "synthetic scale"))
    if stream_mode:
        wl = (f"synthetic max-ones masked superstring n={n} ({scale}), reads of {args.read_len} bases (50% from the "
              f"superstring, half of those reverse-complemented, {args.subst_ppm} substitutions per million), every {k}-mer window, "
              f"fmsi query -S -O semantics")
    else:
        wl = (f"synthetic max-ones masked superstring n={n} ({scale}), isolated random {k}-mers 50% present, fmsi query -O semantics")
    if use_dict and args.compact:
        wl += ("; answered by the handle's super-k-mer dictionary ALONE (the index's answers filed by minimizer in 16-byte records: one 128-byte bucket read and no LF "
               "step per k-mer, no k-mer dictionary in the handle; the LF walk is roofline.lf_path)")
    elif use_dict:
        wl += "; answered by the handle's k-mer dictionary (a hash table of the index's answers: one 128-byte read and no LF step per k-mer; the LF walk is roofline.lf_path)"
    if args.repeat_permille:
        wl += (f"; {args.repeat_permille / 10:.0f}% of the superstring's 2048-base segments are copies of {args.repeat_families} repeat "
               f"families at {args.repeat_divergence_permille / 10:.0f}% divergence")
    line = {
        "metric": "k-mer membership queries/sec (k=31)", "value": value, "unit": "k-mers/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": wl, "index_rows": info_d["n"], "k": k, "queries_per_step_per_gpu": batch, "mode": args.mode,
                   "kernel": {"dict": "dictionary", "dict_sk": "super-k-mer dictionary", "runs_sk": "super-k-mer-runs", "runs_chain": "dictionary-chained-runs",
                              "runs_plain": "dictionary-runs", "runs_tables": "phased-runs", "klcp": "stream"}.get(via, args.kernel),
                   "accelerator": ({"kind": "super-k-mer dictionary alone", "bytes": info_d["sk_bytes"], "records": info_d["sk_records"],
                                    "log2_buckets": info_d["sk_log2_buckets"], "minimizer_len": info_d["sk_minimizer_len"],
                                    "heavy_minimizers": info_d["sk_heavy_minimizers"], "overflowed_buckets": info_d["sk_overflowed_buckets"],
                                    "aux_dictionary_keys": info_d["sk_aux_keys"], "build_seconds": round(accel_s, 2), "lf_steps_per_query": 0} if (use_dict and args.compact) else
                                   {"kind": "k-mer dictionary", "bytes": info_d["dict_bytes"], "levels": info_d["dict_levels"],
                                    "log2_buckets": info_d["dict_log2_buckets"], "keys_per_level": info_d["dict_keys"],
                                    "chain_bytes": info_d["dict_chain_bytes"], "build_seconds": round(accel_s, 2),
                                    "lf_steps_per_query": 0} if use_dict else
                                   {"kind": "prefix table + candidate tables", "prefix_table_len": info_d["prefix_len"],
                                    "candidate_entry_bytes": info_d["candidate_entry_bytes"]}),
                   "prefix_table_len": info_d["prefix_len"], "index_bytes_hbm": info_d["device_bytes"],
                   "candidate_entry_bytes": info_d["candidate_entry_bytes"],
                   "parallelism": f"query-sharded x{world}, index replicated", "positive_fraction": round(m["positives"], 4),
                   "results": ("bit-packed (FMSI_HIP_OUT_BITS), copied to page-locked host memory inside the timed region"
                               + (f", {args.host_copy} copies beside the RCCL gather to rank 0's HBM" if multi else "")) if bits else "one byte per window, left in HBM",
                   "results_on_host_match_bytes": m["bits_check"], "host_enqueue_ms_per_step": round(m["enqueue_s"] / args.steps * 1e3, 3),
                   "index_build_seconds": round(max(build_all), 2), "index_build_seconds_per_rank": [round(x, 2) for x in build_all]},
        "roofline": roof,
        "cpu_baseline": cpu_baseline,
    }
    if gather_note is not None:
        line["config"]["gather"] = gather_note
    if world == 1 and not multi and not stream_mode and use_dict and not args.no_config_legs and not args.repeat_permille:
        t0 = time.time()
        line["configs"] = run_config_legs(args)
        line["config"]["config_legs_seconds"] = round(time.time() - t0, 1)
    emit(json_fd, line)
    os.close(json_fd)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
