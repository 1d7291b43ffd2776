#!/usr/bin/env python3
"""bench.py -- k-mer membership queries/s (k = 31) of the MI355X `fmsi query` hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  One "step" = one pass of the hot path (fmsi_hip_query_kmers,
`-O` semantics, forward strand first) over one batch of synthetic isolated 31-mers that is already
resident in HBM; W untimed steps, then exactly K timed steps bracketed by barrier + synchronize, MAX over
ranks, one JSON line from rank 0.

Workload (BASELINE.json: human-genome-scale masked superstring, isolated random 31-mers, half of them
present): a synthetic max-ones superstring of --n bases generated and indexed on the device
(fmsi_hip_index_build_synthetic), replicated per GPU; the query stream (seeded, counter based) is sharded
across ranks by contiguous ranges -- weak scaling, no collective on the data path; the results of the
last step are gathered to rank 0 over RCCL after the timed region (that is the reference's stdout order).

Extra objects on the JSON line:
  roofline     -- HBM bound.  `achieved` = algorithmic bytes of one launch / average kernel duration, where the
                  bytes are 64 B x (interleaved blocks + prefix-table lines the kernel's own counters report for
                  this layout) + 8 B k-mer in + 1 B result out per window, and the duration comes from HIP events
                  recorded on the launch stream inside the library.  `canonical_achieved` applies SURVEY 8(d)'s
                  reference-layout formula (256 B per LF step the REFERENCE algorithm would execute for the same
                  queries + 64 B per mask probe + in/out) to the same time; it can exceed the HBM peak because the
                  fused layout and the prefix table touch far fewer bytes than the reference's structures.
  cpu_baseline -- the real reference (oracle/_ref, kind "reference") or the C restatement (kind "port") timed on
                  this box's host cores on a bounded sample of the same queries against the same index.
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
    ap.add_argument("--n", type=float, default=2.8e9, help="superstring length in bases (human scale: 2.8e9)")
    ap.add_argument("--batch", type=float, default=float(1 << 27), help="query k-mers per step and GPU")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--prefix-len", type=int, default=12, help="private prefix table length t (0 = none)")
    ap.add_argument("--kernel", choices=["refill", "lockstep"], default="refill")
    ap.add_argument("--mode", choices=["all", "or"], default="all")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--query-seed", type=int, default=42)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="k-mers for the CPU baseline (-1 = auto, 0 = skip)")
    ap.add_argument("--no-counters", action="store_true")
    return ap.parse_args()


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import fmsi_amd as fm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev_t = torch.device("cuda", local_rank)

    n = int(args.n)
    k = args.k
    batch = int(args.batch)
    t0 = time.time()
    index = fm.DeviceIndex.build_synthetic(n, args.seed, k, with_klcp=False, device=local_rank, prefix_len=args.prefix_len)
    build_s = time.time() - t0
    flags = fm.MODE_ALL if args.mode == "all" else fm.MODE_OR
    flags |= fm.KERNEL_LOCKSTEP if args.kernel == "lockstep" else fm.KERNEL_REFILL

    stream = torch.cuda.current_stream().cuda_stream
    d_kmers = torch.empty(batch, dtype=torch.int64, device=dev_t)
    d_out = torch.empty(batch, dtype=torch.uint8, device=dev_t)
    d_cnt = torch.zeros(fm.N_COUNTERS, dtype=torch.int64, device=dev_t)

    def gen(step):
        # rank r owns the contiguous range [r*batch, (r+1)*batch) of every step's global batch (weak scaling)
        first = (step * world + rank) * batch
        index.synthetic_queries_dev(k, args.query_seed, first, batch, d_kmers.data_ptr(), stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup (also: one counted launch to obtain the algorithmic byte counts for this query distribution)
    gen(0)
    for w in range(max(args.warmup, 1)):
        index.query_kmers_dev(d_kmers.data_ptr(), batch, k, flags, d_out.data_ptr(), stream=stream)
    counters = None
    if not args.no_counters:
        d_cnt.zero_()
        index.query_kmers_dev(d_kmers.data_ptr(), batch, k, flags, d_out.data_ptr(), counters_ptr=d_cnt.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        counters = d_cnt.cpu().numpy().astype(np.int64)
    # canonical step count: what the reference algorithm (no prefix table) executes for the same queries
    canonical = None
    if not args.no_counters:
        d_cnt.zero_()
        index.query_kmers_dev(d_kmers.data_ptr(), batch, k, flags | fm.NO_PREFIX_TABLE, d_out.data_ptr(), counters_ptr=d_cnt.data_ptr(),
                              stream=stream)
        torch.cuda.synchronize()
        canonical = d_cnt.cpu().numpy().astype(np.int64)
    positives = float(d_out.to(torch.float32).mean().item())

    # the query batches of the timed steps are generated before the timed region (inputs resident in HBM)
    n_buf = min(args.steps, 4)
    bufs = [torch.empty(batch, dtype=torch.int64, device=dev_t) for _ in range(n_buf)]
    for s in range(n_buf):
        index.synthetic_queries_dev(k, args.query_seed, ((s + 1) * world + rank) * batch, batch, bufs[s].data_ptr(), stream)
    index.profile(True)
    index.profile_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        index.query_kmers_dev(bufs[s % n_buf].data_ptr(), batch, k, flags, d_out.data_ptr(), stream=stream)
    barrier()
    elapsed = time.perf_counter() - t0
    launches, kernel_ms = index.profile_read(reset=True)
    index.profile(False)

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed_max = float(el.item())

    # gather the last step's results to rank 0 (rank order = output order); outside the timed region
    gathered = None
    if world > 1:
        parts = [torch.empty_like(d_out) for _ in range(world)] if rank == 0 else None
        dist.gather(d_out, parts, dst=0)
        if rank == 0:
            gathered = torch.cat(parts)
    else:
        gathered = d_out

    if rank == 0:
        total_q = batch * world * args.steps
        value = total_q / elapsed_max
        avg_kernel_s = kernel_ms / max(launches, 1) / 1e3
        roof = None
        if counters is not None:
            blocks_read, lookups = int(counters[1]), int(counters[4])
            # blocks_read already contains the prefix-table lines (one 64-B line per lookup)
            bytes_launch = 64 * blocks_read + 8 * batch + 1 * batch
            canon_bytes = 256 * int(canonical[0]) + 64 * int(canonical[2]) + 8 * batch + batch // 8
            ach = bytes_launch / avg_kernel_s / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args, n, batch, k),
                    "kernel": f"query_{args.kernel}_kernel", "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": launches,
                    "algorithmic_bytes_per_launch": bytes_launch,
                    "bytes_per_kmer": round(bytes_launch / batch, 1),
                    "lf_steps_per_kmer": round(int(counters[0]) / batch, 3),
                    "blocks_per_kmer": round(blocks_read / batch, 3), "table_lookups_per_kmer": round(lookups / batch, 3),
                    "random_requests_per_s": round(blocks_read / avg_kernel_s / 1e9, 2),
                    "canonical_lf_steps_per_kmer": round(int(canonical[0]) / batch, 3),
                    "canonical_achieved": round(canon_bytes / avg_kernel_s / 1e9, 1),
                    "canonical_frac": round(canon_bytes / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 4)}
        info = index.info
        line = {
            "metric": "k-mer membership queries/sec (k=31)", "value": value, "unit": "k-mers/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"synthetic max-ones masked superstring n={n} (human-genome scale), isolated random {k}-mers "
                                   f"50% present, fmsi query -O semantics", "index_rows": int(info.n), "k": k,
                       "queries_per_step_per_gpu": batch, "mode": args.mode, "kernel": args.kernel,
                       "prefix_table_len": int(info.prefix_len), "index_bytes_hbm": int(info.device_bytes),
                       "block_bytes": int(info.block_bytes), "parallelism": f"query-sharded x{world}, index replicated",
                       "positive_fraction": round(positives, 4), "index_build_seconds": round(build_s, 2)},
            "roofline": roof,
            "cpu_baseline": None,
        }
        if args.cpu_sample != 0:
            try:
                line["cpu_baseline"] = cpu_baseline(args, index, n, k)
            except Exception as e:   # the baseline is a reported figure, never a reason to lose the GPU line
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(args, n, batch, k):
    """HBM bytes per launch of the timed kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE and WRITE_SIZE in separate passes, calibrated on random 64-B reads: profiles/README.md).
    bench.py cannot collect PMC counters itself; the figure is attached only when the run's configuration is
    the one that was profiled, otherwise null."""
    best = None   # returns bytes per launch (FETCH_SIZE + WRITE_SIZE) or None
    for p in sorted((REPO / "profiles").glob("r*_traffic.json")):
        try:
            t = json.loads(p.read_text())
        except (OSError, ValueError):
            continue
        c = t.get("config", {})
        if (c.get("n"), c.get("batch"), c.get("k"), c.get("prefix_len"), c.get("mode")) == (n, batch, k, args.prefix_len, args.mode) \
                and args.kernel in t.get("kernel", ""):
            best = t
    if best is None:
        return None
    return best["hbm_bytes_per_launch"]


def cpu_baseline(args, index, n, k):
    """Filled in by oracle-side code (bench_cpu.py); kept separate so the GPU path never imports oracle/."""
    import bench_cpu
    return bench_cpu.run(args, index, n, k)


if __name__ == "__main__":
    main()
