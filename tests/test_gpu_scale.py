"""At-scale GPU parity (run with `-m gpu` on an MI355X): BASELINE.json's configurations at their full sizes, mixed masks and the
default (OR) mode at >= 10^8 rows, and an index that is byte-identical to one the REAL reference built at 10^8 rows.

What checks what:
  * kernels against each other: lookups + candidates (phased), lookups + two-base steps (pair), lock-step single steps without
    any table -- three independent algorithms, strand by strand (STRICT), both modes;
  * against the CPU side on a bounded sample of the same seeded queries: the GPU-built index is handed to the oracle (C
    restatement, both strands' ternaries: src/fms_index.h:117-144 per strand) and, when oracle/_ref travelled, to the real
    reference's own query_kmers() loop (printed bits, OR mode: the reference's default, whose output does not depend on the
    strand predictor);
  * against the real reference's index files: md5s recorded in the build container (tests/golden/make_scale_md5.py).
"""
import hashlib
import os
import shutil
import tempfile
from pathlib import Path
import json
import sys

import numpy as np
import pytest

import oracle_py
from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu

sys.path.insert(0, str(REPO))


@pytest.fixture
def fm(product_built):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on an MI355X")
    # every test here runs a chain of builders over tens of GB of temporaries: they share the library's block cache (devpool.h) and
    # return it once, when the test ends
    with product_built.builder_chain(0):
        yield product_built


def kmers_text(kmers: np.ndarray, k: int) -> bytes:
    codes = ((kmers[:, None] >> (2 * np.arange(k, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()


def device_answers(fm, dev, d_kmers, nq, k, flags):
    import torch
    d_out = torch.empty(nq, dtype=torch.uint8, device="cuda")
    dev.query_kmers_dev(d_kmers.data_ptr(), nq, k, flags, d_out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


def cross_check(fm, dev, k, nq, sample, seed=42, or_reference=True, with_dict=True, chain=False):
    """nq seeded queries (half drawn from the superstring): kernels agree strand by strand in both modes; the first `sample`
    of them equal the oracle's ternaries on the exported index and (OR mode) the real reference's printed bits."""
    import torch
    import bench_cpu
    s = torch.cuda.current_stream().cuda_stream
    d_kmers = torch.empty(nq, dtype=torch.int64, device="cuda")
    dev.synthetic_queries_dev(k, seed, 0, nq, d_kmers.data_ptr(), s)
    torch.cuda.synchronize()
    text = kmers_text(d_kmers[:sample].cpu().numpy().view(np.uint64), k)
    orc = bench_cpu.oracle_from_index(dev, k, with_klcp=False)
    strict = {}
    for mode_name, mode in (("or", fm.MODE_OR), ("all", fm.MODE_ALL)):
        outs = {name: device_answers(fm, dev, d_kmers, nq, k, mode | fm.STRICT | kern)
                for name, kern in (("phased", fm.KERNEL_PHASED), ("pair", fm.KERNEL_PAIR | fm.NO_PREFIX_TABLE),
                                   ("lockstep", fm.KERNEL_LOCKSTEP | fm.NO_PREFIX_TABLE), ("sorted", fm.KERNEL_SORTED))}
        assert np.array_equal(outs["phased"], outs["pair"]), mode_name
        assert np.array_equal(outs["phased"], outs["lockstep"]), mode_name
        assert np.array_equal(outs["phased"], outs["sorted"]), mode_name
        f, r = fm.decode_strict(outs["phased"])
        of, orr = orc.ternaries_bulk(text, sample, k, mode_name == "all")
        assert np.array_equal(f[:sample], of) and np.array_equal(r[:sample], orr), mode_name
        # the cheaper schedule (forward first, reverse complement when needed) and its bit-packed form
        fast = device_answers(fm, dev, d_kmers, nq, k, mode | fm.KERNEL_PHASED)
        want = (np.where(f != -1, f, r) == 1) if mode_name == "all" else ((f == 1) | (r == 1))
        assert np.array_equal(fast.astype(bool), want), mode_name
        d_bits = torch.zeros(fm.packed_words(nq, 0), dtype=torch.int64, device="cuda")
        dev.query_kmers_dev(d_kmers.data_ptr(), nq, k, mode | fm.KERNEL_PHASED | fm.OUT_BITS, d_bits.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert np.array_equal(fm.unpack_bits(d_bits.cpu().numpy().view(np.uint64), nq, 0), fast)
        strict[mode_name] = (f, r)
    if with_dict:   # the k-mer dictionary, added to the same handle: the same answers from one bucket read per k-mer
        dev.enable_dict(chain=chain)
        for mode_name, mode in (("or", fm.MODE_OR), ("all", fm.MODE_ALL)):
            f, r = fm.decode_strict(device_answers(fm, dev, d_kmers, nq, k, mode | fm.STRICT | fm.KERNEL_DICT))
            assert np.array_equal(f, strict[mode_name][0]) and np.array_equal(r, strict[mode_name][1]), ("dictionary", mode_name)
    if or_reference and oracle_py.ref_available():
        import tempfile
        with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
            orc.dump(f"{d}/x")
            ref = oracle_py.Reference.load(f"{d}/x")
            _, bits = ref.time_isolated_kmers(text, sample, k, oracle_py.MODE_OR)
            f, r = strict["or"]
            assert np.array_equal(bits.astype(bool), ((f == 1) | (r == 1))[:sample])
            ref.close()
    orc.close()
    return strict


def test_config3_pangenome_scale(fm):
    """BASELINE.json config 3: E. coli pangenome scale, n = 1.4 x 10^9 rows, k = 31, the production tables (t = 16, 32-byte
    candidate entries)."""
    n, k = 1_400_000_000, 31
    dev = fm.DeviceIndex.build_synthetic(n, 12345, k, prefix_len=16).enable_pair_layout().enable_verify()
    assert int(dev.info.candidate_entry_bytes) == 32
    strict = cross_check(fm, dev, k, nq=1 << 22, sample=150_000)
    f, r = strict["all"]
    present = (f == 1) | (r == 1)
    assert 0.49 < present.mean() < 0.51 and ((f == 0) | (r == 0)).sum() == 0    # max-ones: an occurrence is always ON


def test_config2_ecoli_full_size(fm):
    """BASELINE.json config 2 at its full size: n = 4.7 x 10^6 rows, 2^20 queries against the oracle, with the tables bench.py
    uses at this size (t = 16 + candidates: 172 GB for a 4.7 Mbp genome) and with tables that fit the rule 4^t <= 4n (t = 12)."""
    n, k = 4_700_000, 31
    for t in (16, 12):
        dev = fm.DeviceIndex.build_synthetic(n, 12345, k, prefix_len=t).enable_pair_layout().enable_verify()
        cross_check(fm, dev, k, nq=1 << 20, sample=1 << 20, or_reference=(t == 16))
        dev.close()


def test_mixed_mask_at_1e8_rows(fm):
    """A non-max-ones mask at 1.2 x 10^8 rows (60 % of the positions ON) with repeat families, so that intervals of several rows
    with differing mask bits occur: OR mode scans the interval (src/fms_index.h:133-142), `-O` takes its first row."""
    n, k = 120_000_000, 31
    dev = fm.DeviceIndex.build_synthetic(n, 777, k, prefix_len=14, repeats=(200, 50, 10), mask_permille=600).enable_pair_layout().enable_verify()
    strict = cross_check(fm, dev, k, nq=1 << 22, sample=300_000)
    fo, ro = strict["or"]
    fa, ra = strict["all"]
    assert ((fo == 0) | (ro == 0)).mean() > 0.1 and ((fo == 1) | (ro == 1)).mean() > 0.2     # both answers occur among the present
    assert (fo != fa).sum() + (ro != ra).sum() > 1000                                          # and the modes really differ
    assert ((fo == -1) == (fa == -1)).all() and ((ro == -1) == (ra == -1)).all()               # absence does not depend on the mode
    assert ((fa == 1) <= (fo == 1)).all()                                                      # first row ON => some row ON
    assert dev.info.dict_ambiguous > 1000      # keys whose `-O` answer the dictionary took from the index's own search


def test_dictionary_at_human_scale(fm):
    """BASELINE.json's headline size (n = 2.8 x 10^9 rows) in bench.py's default configuration: pair layout + k-mer dictionary
    (36.5 GB), no tables.  Dictionary == two-base LF walk == lock-step LF walk, strand by strand, both modes, on 4 M seeded k-mers;
    a sample against the oracle on the exported index; reads through the run entry point == their isolated windows."""
    import torch
    n, k = 2_800_000_000, 31
    dev = fm.DeviceIndex.build_synthetic(n, 12345, k).enable_pair_layout()
    strict = cross_check(fm, dev, k, nq=1 << 22, sample=100_000, or_reference=True, chain=True)   # (the sample also through the real reference's query_kmers())
    info = dev.info
    assert info.dict_levels == 2 and info.dict_log2_buckets == 28 and info.dict_keys[0] > 2_799_000_000
    assert info.dict_bytes - info.dict_chain_bytes < 40e9 and 15e9 < info.dict_chain_bytes < 25e9
    f, r = strict["all"]
    assert 0.49 < ((f == 1) | (r == 1)).mean() < 0.51
    L, n_reads = 3000, 512
    s = torch.cuda.current_stream().cuda_stream
    d_reads = torch.empty((n_reads * L) // 32 + 2, dtype=torch.int64, device="cuda")
    dev.synthetic_reads_dev(L, 7, 0, n_reads, d_reads.data_ptr(), 10000, s)    # 1 % substitutions in the reads drawn from the text
    torch.cuda.synchronize()
    packed = d_reads.cpu().numpy().view(np.uint64)
    starts = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    wins = np.full(n_reads, L - k + 1, dtype=np.uint32)
    a, cn = dev.query_runs(packed, n_reads * L, starts, wins, k, fm.MODE_ALL | fm.STRICT, counters=True)               # dictionary, chained
    b = dev.query_runs(packed, n_reads * L, starts, wins, k, fm.MODE_ALL | fm.STRICT | fm.NO_PREFIX_TABLE)            # LF walk
    assert np.array_equal(a, b)
    assert int(cn[4]) < 0.8 * a.size                       # (fewer bucket reads than windows: the chains ran)
    present = (((a & 3) == 2) | (((a >> 2) & 3) == 2)).reshape(n_reads, -1).mean(axis=1)
    assert 0.3 < (present > 0.5).mean() < 0.7 and ((present > 0.02) & (present < 0.5)).sum() == 0    # reads from the text keep ~73 % of their windows


def _reads_agree(fm, dev, k, L, n_reads, subst_ppm, flags_a, flags_b, seed=7):
    """n_reads seeded reads of L bases (half from the text, with substitutions) through fmsi_hip_query_runs under two flag sets: same bytes."""
    import torch
    s = torch.cuda.current_stream().cuda_stream
    d_reads = torch.empty((n_reads * L) // 32 + 2, dtype=torch.int64, device="cuda")
    dev.synthetic_reads_dev(L, seed, 0, n_reads, d_reads.data_ptr(), subst_ppm, s)
    torch.cuda.synchronize()
    packed = d_reads.cpu().numpy().view(np.uint64)
    starts = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    wins = np.full(n_reads, L - k + 1, dtype=np.uint32)
    a, cn = dev.query_runs(packed, n_reads * L, starts, wins, k, flags_a, counters=True)
    b = dev.query_runs(packed, n_reads * L, starts, wins, k, flags_b)
    assert np.array_equal(a, b)
    return a, cn


@pytest.fixture(scope="module")
def human_mixed(product_built):
    """ONE handle for the at-scale tests that want the hard text: n = 2.8 x 10^9 rows, 60 % of the positions ON, a fifth of the 2048-base
    segments copies of 50 consensus segments at 1 % divergence, with the kLCP array, a 16-deep prefix table (what bench.py's kLCP legs use: entries
    of <= 3 rows carry signatures, so the S-block kernel's absence certificates fire) and the pair layout (building it takes longer than
    either test's queries; the dictionaries a test adds stay on it)."""
    fm = product_built
    n, k = 2_800_000_000, 31
    with fm.builder_chain(0):
        dev = fm.DeviceIndex.build_synthetic(n, 777, k, with_klcp=True, prefix_len=16, repeats=(200, 50, 10), mask_permille=600).enable_pair_layout()
    yield dev
    dev.close()
    fm.trim_scratch(0)


def _seeded_reads(dev, L, n_reads, subst_ppm, seed):
    import torch
    s = torch.cuda.current_stream().cuda_stream
    d_reads = torch.empty((n_reads * L) // 32 + 2, dtype=torch.int64, device="cuda")
    dev.synthetic_reads_dev(L, seed, 0, n_reads, d_reads.data_ptr(), subst_ppm, s)
    torch.cuda.synchronize()
    return d_reads.cpu().numpy().view(np.uint64)


def test_stream_kernel_at_human_scale(fm, human_mixed):
    """The S-block kLCP kernel (query_stream2_kernel, what `fmsi query -S` runs; src/fms_index.h:106-115,181-254) held against CHECKERS at
    the headline size, on the configuration bench.py times: n = 2.8 x 10^9 rows (mixed mask + repeat families: kLCP runs of thousands of
    rows, intervals with differing mask bits), 512 reads of 3000 bases, half from the text, 1 % substitutions.
      * every window, both modes: STRICT ternaries == the table-less LF walk (fmsi_hip_query_runs | NO_PREFIX_TABLE: fresh searches);
      * the fast schedule (first strand decides if it can), as bytes and bit-packed (OUT_BITS), == what the ternaries imply;
      * uniform runs (three numbers in the kernel's arguments) and ragged runs (the run table, tasks that end inside reads);
      * the first 34 reads (10^5 windows): == the ORACLE's ternaries (C restatement on the downloaded index incl. kLCP) and == the REAL
        reference's `-S` record loop (query_kmers<or>(..., has_klcp=true)) on the index written in its own file format;
      * ORDERS (`lookup -S`: the kernel's ORD instantiation) == the oracle's orders on the sample, == the LF walk on all windows."""
    import os
    import tempfile
    import bench_cpu
    from conftest import stream_expected_fast, stream_expected_orders
    dev, k, L, n_reads = human_mixed, 31, 3000, 512
    assert int(dev.info.has_klcp) == 1
    packed = _seeded_reads(dev, L, n_reads, 10000, seed=11)
    w = L - k + 1
    starts = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    wins = np.full(n_reads, w, dtype=np.uint32)
    wins_r = wins.copy()                     # ragged: every third read cut short, one read of a single window, one of 65 (a task + 1)
    wins_r[::3] = w // 2 + 7
    wins_r[5], wins_r[8] = 1, 65
    sample_reads = 34
    # ---- the checker's side: oracle on the downloaded index; the real reference on the files the product writes ----
    orc = bench_cpu.oracle_from_index(dev, k, with_klcp=True)
    codes = ((packed[:, None] >> (2 * np.arange(32, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8).reshape(-1)[:n_reads * L]
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].reshape(n_reads, L)
    idx = np.arange(w)[:, None] + np.arange(k)[None, :]
    sample_text = b"".join(bases[r][idx].tobytes() for r in range(sample_reads))
    n_sample = sample_reads * w
    strict = {}
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        for rw, tag in ((wins, "uniform"), (wins_r, "ragged")):
            walk = dev.query_runs(packed, n_reads * L, starts, rw, k, flag | fm.STRICT | fm.NO_PREFIX_TABLE)
            got, cn = dev.query_stream(packed, n_reads * L, starts, rw, k, flag | fm.STRICT, counters=True)     # (16-deep table: certificates)
            assert np.array_equal(got, walk), (mode_all, tag, np.nonzero(got != walk)[0][:10])
            # both strands of every window, yet far fewer table requests than windows: absent (t + 4)-mers settled a dozen windows each
            assert int(cn[4]) < 0.6 * int(rw.sum()), (int(cn[4]), int(rw.sum()))
            assert np.array_equal(dev.query_stream(packed, n_reads * L, starts, rw, k, flag | fm.STRICT | fm.KERNEL_REFILL), walk), (mode_all, tag)   # (the first kernel: no certificates)
            assert np.array_equal(dev.query_stream(packed, n_reads * L, starts, rw, k, flag | fm.STRICT | fm.NO_PREFIX_TABLE), walk), (mode_all, tag)
            f, r = fm.decode_strict(got)
            # a streamed window costs about one LF step where the read follows the text: far fewer steps than fresh searches of k bases
            assert 0 < int(cn[0]) < 8 * int(rw.sum())
            fast = dev.query_stream(packed, n_reads * L, starts, rw, k, flag)
            assert np.array_equal(fast.astype(bool), stream_expected_fast(f, r, rw, mode_all)), (mode_all, tag)
            bits = dev.query_stream(packed, n_reads * L, starts, rw, k, flag | fm.OUT_BITS)
            assert np.array_equal(fm.unpack_bits(bits, int(rw.sum()), 0), fast), (mode_all, tag)
            if tag == "uniform":
                strict[mode_all] = (f, r)
                of, orr = orc.ternaries_bulk(sample_text, n_sample, k, mode_all)
                assert np.array_equal(f[:n_sample], of) and np.array_equal(r[:n_sample], orr), mode_all
    fo, ro = strict[False]
    fa, ra = strict[True]
    assert ((fo == 1) | (ro == 1)).mean() > 0.15 and ((fo == 0) | (ro == 0)).mean() > 0.05 and ((fo == -1) & (ro == -1)).mean() > 0.4
    assert (fo != fa).sum() + (ro != ra).sum() > 100          # the modes differ on this mask (intervals with mixed bits)
    # ---- lookup -S (ORDERS) ----
    _, _, of, orr = orc.ternaries_bulk(sample_text, n_sample, k, False, orders=True)
    got = dev.query_stream(packed, n_reads * L, starts, wins, k, fm.ORDERS | fm.STRICT)
    assert np.array_equal(got[:n_sample, 0], of) and np.array_equal(got[:n_sample, 1], orr)
    walk = dev.query_runs(packed, n_reads * L, starts, wins, k, fm.ORDERS | fm.STRICT | fm.NO_PREFIX_TABLE)
    assert np.array_equal(got, walk)
    got1 = dev.query_stream(packed, n_reads * L, starts, wins, k, fm.ORDERS)
    assert np.array_equal(got1, stream_expected_orders(fo, ro, got[:, 0], got[:, 1], wins))
    orc.close()
    # ---- the real reference's own -S loop on the sample ----
    if oracle_py.ref_available():
        with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
            dev.write_files(f"{d}/x")
            ref = oracle_py.Reference.load(f"{d}/x", use_klcp=True)
            fa_path = os.path.join(d, "s.fa")
            with open(fa_path, "wb") as fh:
                for r_ in range(sample_reads):
                    fh.write(b">r%d\n" % r_ + bases[r_].tobytes() + b"\n")
            text, _ = ref.query_file(fa_path, k, has_klcp=True, mode=oracle_py.MODE_OR, cap=sample_reads * (w + 32) + 4096)
            want = np.concatenate([np.frombuffer(line.split("\t")[1].encode(), dtype=np.uint8) - ord("0") for line in text.splitlines()])
            assert np.array_equal(want.astype(bool), ((fo == 1) | (ro == 1))[:n_sample])
            ref.close()


def test_mixed_mask_at_human_scale(fm, human_mixed):
    """The headline size with a NON-max-ones mask (60 % of the positions ON) and repeat families (a fifth of the 2048-base segments are
    copies of 50 consensus segments at 1 % divergence: thousands of occurrences per k-mer with differing mask bits): OR mode scans the
    interval (src/fms_index.h:133-142), `-O` takes its first row, STRICT both strands.  Dictionary == two-base LF walk == lock-step LF
    walk on 4 M seeded k-mers in both modes, the oracle's ternaries and the real reference's printed bits on a sample of the exported
    index; then the super-k-mer dictionary (the families' minimizers are heavy: the aux dictionary) against the LF walk on reads."""
    k = 31
    dev = human_mixed
    strict = cross_check(fm, dev, k, nq=1 << 22, sample=100_000, or_reference=True)
    fo, ro = strict["or"]
    fa, ra = strict["all"]
    assert ((fo == 0) | (ro == 0)).mean() > 0.1 and ((fo == 1) | (ro == 1)).mean() > 0.2     # both answers occur among the present
    assert (fo != fa).sum() + (ro != ra).sum() > 1000                                          # and the modes really differ
    assert ((fo == -1) == (fa == -1)).all() and ((ro == -1) == (ra == -1)).all()               # absence does not depend on the mode
    assert ((fa == 1) <= (fo == 1)).all()                                                      # first row ON => some row ON
    assert dev.info.dict_ambiguous > 100_000     # keys whose `-O` answer the dictionary took from the index's own search
    dev.enable_dict(superkmer=True)
    info = dev.info
    assert info.sk_bytes > 0 and info.sk_heavy_minimizers > 1000 and info.sk_aux_keys > 100_000
    for mode in (fm.MODE_OR, fm.MODE_ALL):
        a, cn = _reads_agree(fm, dev, k, 3000, 512, 10000, mode | fm.STRICT, mode | fm.STRICT | fm.NO_PREFIX_TABLE)       # records (+ aux) == LF walk
        assert int(cn[0]) == 0 and int(cn[4]) < 0.6 * a.size
        _reads_agree(fm, dev, k, 3000, 512, 10000, mode | fm.STRICT | fm.NO_SUPERKMER, mode | fm.STRICT)                   # == k-mer dictionary


def test_headline_kernel_at_headline_size(fm, human_mixed):
    """What bench.py's headline runs -- isolated 31-mers on a handle that keeps the super-k-mer dictionary ALONE (`sk_kmers_kernel`: a lane
    per k-mer, one bucket read, no LF step) -- held against the checkers at the headline's size and on the hard text: n = 2.8 x 10^9 rows,
    60 % mask, repeat families (their minimizers are heavy, their buckets overflow: the aux dictionary's path).  2^22 seeded k-mers through
    fmsi_hip_query_kmers in OR and ALL, STRICT: the bytes == the table-less two-base LF walk strand by strand on all of them, == the
    oracle's ternaries (src/fms_index.h:263-331 both strands) and the real reference's printed bits on 10^5; the cheaper schedule and the
    bit-packed form agree; and the counters say which path answered: no LF step, more bucket reads than k-mers (the aux lookups)."""
    import torch
    import bench_cpu
    k, nq, sample = 31, 1 << 22, 100_000
    dev = human_mixed
    dev.enable_dict(superkmer=True)            # (a no-op when test_mixed_mask_at_human_scale ran before this one)
    dev.enable_dict(superkmer_only=True)       # the k-mer dictionary goes: the records answer isolated k-mers, too
    info = dev.info
    assert int(info.dict_bytes) == 0 and int(info.sk_bytes) > 0
    assert int(info.sk_aux_keys) > 100_000 and int(info.sk_overflowed_buckets) > 0 and int(info.sk_heavy_minimizers) > 1000
    s = torch.cuda.current_stream().cuda_stream
    d_kmers = torch.empty(nq, dtype=torch.int64, device="cuda")
    dev.synthetic_queries_dev(k, 4242, 0, nq, d_kmers.data_ptr(), s)
    torch.cuda.synchronize()
    text = kmers_text(d_kmers[:sample].cpu().numpy().view(np.uint64), k)
    orc = bench_cpu.oracle_from_index(dev, k, with_klcp=False)
    strict = {}
    for mode_name, mode in (("or", fm.MODE_OR), ("all", fm.MODE_ALL)):
        d_out = torch.empty(nq, dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros(fm.N_COUNTERS, dtype=torch.int64, device="cuda")
        dev.query_kmers_dev(d_kmers.data_ptr(), nq, k, mode | fm.STRICT, d_out.data_ptr(), d_cnt.data_ptr(), s)
        torch.cuda.synchronize()
        got, cn = d_out.cpu().numpy(), d_cnt.cpu().numpy()
        assert int(cn[0]) == 0 and int(cn[6]) == nq and int(cn[4]) > nq + 1000, cn      # no LF step; bucket reads = records' + the aux dictionary's
        walk = device_answers(fm, dev, d_kmers, nq, k, mode | fm.STRICT | fm.KERNEL_PAIR | fm.NO_PREFIX_TABLE)
        assert np.array_equal(got, walk), (mode_name, np.nonzero(got != walk)[0][:10])
        f, r = fm.decode_strict(got)
        of, orr = orc.ternaries_bulk(text, sample, k, mode_name == "all")
        assert np.array_equal(f[:sample], of) and np.array_equal(r[:sample], orr), mode_name
        fast = device_answers(fm, dev, d_kmers, nq, k, mode)
        want = (np.where(f != -1, f, r) == 1) if mode_name == "all" else ((f == 1) | (r == 1))
        assert np.array_equal(fast.astype(bool), want), mode_name
        for fl, st in ((mode | fm.OUT_BITS, 0), (mode | fm.STRICT | fm.OUT_BITS, fm.STRICT)):
            d_bits = torch.zeros(fm.packed_words(nq, st), dtype=torch.int64, device="cuda")
            dev.query_kmers_dev(d_kmers.data_ptr(), nq, k, fl, d_bits.data_ptr(), stream=s)
            torch.cuda.synchronize()
            assert np.array_equal(fm.unpack_bits(d_bits.cpu().numpy().view(np.uint64), nq, st), got if st else fast), (mode_name, st)
        strict[mode_name] = (f, r)
    fo, ro = strict["or"]
    fa, ra = strict["all"]
    assert 0.3 < ((fo == 1) | (ro == 1) | (fo == 0) | (ro == 0)).mean() < 0.7 and (fo != fa).sum() + (ro != ra).sum() > 1000   # present and absent, and the modes differ
    if oracle_py.ref_available():
        with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
            orc.dump(f"{d}/x")
            ref = oracle_py.Reference.load(f"{d}/x")
            for mode_name, omode in (("or", oracle_py.MODE_OR), ("all", oracle_py.MODE_ALL)):
                f, r = strict[mode_name]
                _, bits = ref.time_isolated_kmers(text, sample, k, omode)
                if mode_name == "or":
                    assert np.array_equal(bits.astype(bool), ((f == 1) | (r == 1))[:sample])
                else:   # -O on a non-max-ones mask: what is printed depends on the strand searched first -- where the strands agree it does not
                    same = (f[:sample] == r[:sample]) | (f[:sample] == -1) | (r[:sample] == -1)
                    assert np.array_equal(bits.astype(bool)[same], (np.where(f != -1, f, r) == 1)[:sample][same])
            ref.close()
    orc.close()


def test_adversarial_text_at_1e9_rows(fm):
    """A text of 10^9 bases that is NOT i.i.d.: megabase homopolymers, tandem repeats of 2..6-base units, a family of 10^5 copies of one
    300-base consensus at 1 % divergence, on a random background with a 60 % mask -- through the on-device builder (prefix doubling over
    millions of tied rows), the k-mer dictionary (keys with 10^6 occurrences and mixed mask bits: the index's own search decides), the
    OR mode's interval scan over intervals of 10^6 rows, and the super-k-mer dictionary (ties in every window of the low-complexity
    stretches).  All kernels agree strand by strand in both modes on seeded queries AND on k-mers cut from the adversarial stretches and
    their borders; a sample of both against the oracle and the real reference."""
    import torch
    n, k = 1_000_000_000, 31
    rng = np.random.default_rng(2024)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    regions = []

    def put(at, block):
        codes[at:at + block.size] = block
        regions.append((at, block.size))

    put(50_000_000, np.zeros(1_500_000, dtype=np.uint8))                        # poly-A, 1.5 Mb
    put(150_000_000, np.full(1_000_000, 3, dtype=np.uint8))                     # poly-T (its reverse complement: the two strands meet)
    put(250_000_000, np.full(700_000, 1, dtype=np.uint8))                       # poly-C
    at = 300_000_000
    for unit in ([0, 3], [0, 1, 2], [0, 1, 2, 3], [2, 2, 0, 1, 3], [0, 0, 1, 3, 2, 2]):   # AT, ACG, ACGT (= its own reverse complement), ...
        put(at, np.tile(np.array(unit, dtype=np.uint8), 500_000 // len(unit)))
        at += 7_000_000
    consensus = rng.integers(0, 4, size=300, dtype=np.uint8)
    fam = np.tile(consensus, 100_000)
    hits = rng.random(fam.size) < 0.01
    fam[hits] = rng.integers(0, 4, size=int(hits.sum()), dtype=np.uint8)
    put(600_000_000, fam)                                                        # 10^5 copies back to back (30 Mb)
    for c in range(2000):                                                        # ... and 2000 more scattered over the text
        cp = consensus.copy()
        cp[rng.integers(0, 300, size=3)] = rng.integers(0, 4, size=3, dtype=np.uint8)
        codes[700_000_000 + c * 100_000: 700_000_000 + c * 100_000 + 300] = cp
    on = rng.random(n) < 0.6
    on[n - (k - 1):] = False
    ms = lut[codes] + np.where(on, 0, 32).astype(np.uint8)                       # lower case = OFF
    del on
    dev = fm.DeviceIndex.build(ms.tobytes(), k).enable_pair_layout()
    del ms
    assert int(dev.info.n) == n + 1
    strict = cross_check(fm, dev, k, nq=1 << 21, sample=50_000, or_reference=True)
    assert dev.info.dict_ambiguous > 1000
    # k-mers cut from the adversarial stretches, their borders and the family (both orientations, a few substitutions)
    picks = []
    for at, size in regions:
        for d in (-40, -31, -30, -16, -1, 0, 1, 15, 100, size // 2, size - 32, size - 31, size - 16, size - 1, size):
            picks.append(at + d)
        picks += [int(x) for x in rng.integers(at, at + max(1, size - k), size=300)]
    picks += [700_000_000 + c * 100_000 + int(o) for c in range(0, 2000, 7) for o in (0, 100, 269)]
    picks = np.array([p for p in picks if 0 <= p <= n - k], dtype=np.int64)
    win = codes[picks[:, None] + np.arange(k)[None, :]].astype(np.uint64)
    flip = rng.random(picks.size) < 0.5
    win[flip] = 3 - win[flip][:, ::-1]
    sub = rng.random(picks.size) < 0.2
    win[sub, rng.integers(0, k, size=int(sub.sum()))] = rng.integers(0, 4, size=int(sub.sum()))
    kmers = (win << (2 * np.arange(k, dtype=np.uint64))[None, :]).sum(axis=1).astype(np.uint64)
    del codes
    text = kmers_text(kmers, k)
    import bench_cpu
    orc = bench_cpu.oracle_from_index(dev, k, with_klcp=False)
    for mode_name, mode in (("or", fm.MODE_OR), ("all", fm.MODE_ALL)):
        of, orr = orc.ternaries_bulk(text, kmers.size, k, mode_name == "all")
        for kern in (fm.KERNEL_DICT, fm.KERNEL_PAIR | fm.NO_PREFIX_TABLE, fm.KERNEL_LOCKSTEP | fm.NO_PREFIX_TABLE, fm.KERNEL_SORTED):
            f, r = fm.decode_strict(dev.query_kmers(kmers, k, mode | fm.STRICT | kern))
            assert np.array_equal(f, of) and np.array_equal(r, orr), (mode_name, kern)
        assert (of == 1).sum() > 100 and (of == 0).sum() > 0 and (of == -1).sum() > 100   # (present ON, present OFF, absent all occur)
    orc.close()
    # the super-k-mer dictionary on such a text: either it is built (and agrees with the LF walk on reads), or it says why not
    try:
        dev.enable_dict(superkmer=True)
    except fm.FmsiError as e:
        assert e.code == fm.E_RANGE
        print(f"adversarial text: no super-k-mer dictionary (FMSI_HIP_ERANGE: {e})")
        branch = "refused"
    else:
        info = dev.info
        print(f"adversarial text: super-k-mer dictionary built ({info.sk_records} records, {info.sk_heavy_minimizers} heavy minimizers, "
              f"{info.sk_aux_keys} k-mers in the aux dictionary)")
        branch = "built"
        for mode in (fm.MODE_OR, fm.MODE_ALL):
            _reads_agree(fm, dev, k, 2000, 512, 10000, mode | fm.STRICT, mode | fm.STRICT | fm.NO_PREFIX_TABLE)
    # which branch this text takes is a property of the construction, not of the box: it is BUILT (the megabase homopolymers and tandem
    # repeats are a few heavy minimizers whose k-mers go to the aux dictionary) -- a change of that is a change to look at
    assert branch == "built"


def test_largest_index_with_32_bit_rows(fm):
    """A superstring of 2^32 - 8192 bases, the largest the 32-bit on-device builder takes (rocPRIM's sort ends there, 32-bit rows 8 000 rows later): built on
    the device, then dictionary == lookups + two-base LF walk == lock-step LF walk strand by strand in both modes on 2^20 seeded k-mers,
    a sample against the oracle on the downloaded index -- rows, counts, suffix positions and dictionary indices all at the edge of
    what 32 bits hold."""
    length, k = (1 << 32) - 8192, 31
    dev = fm.DeviceIndex.build_synthetic(length, 4242, k, prefix_len=14).enable_pair_layout()
    assert int(dev.info.n) == length + 1 and not int(dev.info.rows64)
    strict = cross_check(fm, dev, k, nq=1 << 20, sample=20_000, or_reference=False)
    f, r = strict["all"]
    assert 0.49 < ((f == 1) | (r == 1)).mean() < 0.51
    assert int(dev.info.dict_keys[0]) > 4_290_000_000 and int(dev.info.dict_log2_buckets) == 29
    dev.close()


@pytest.mark.parametrize("gold_file", ["md5_scale.json", "md5_scale_1e9.json"])
def test_index_equals_the_reference_built_one(fm, tmp_path, gold_file):
    """The real reference built these indexes (10^8 and 10^9 rows, mixed mask, kLCP array: src/fms_index.h:397-460 -- qsufsort, 57 s and 986 s
    in the build container, tests/golden/make_scale_md5.py) and their files' md5s are committed.  The on-device builder + the product's
    writer reproduce the six files byte for byte, and the files -- loaded through fmsi_host_index_load like any reference-made index --
    answer like the handle they were written from (10^8)."""
    gold = json.loads((GOLDEN / "idx" / gold_file).read_text())
    spec = gold["spec"]
    n, k = spec["len"], spec["k"]
    dev = fm.DeviceIndex.build_synthetic(n, spec["seed"], k, with_klcp=True, mask_permille=spec["mask_permille"])
    work = Path(tempfile.mkdtemp(dir="/dev/shm")) if n > 200_000_000 and os.path.isdir("/dev/shm") else tmp_path   # (half a gigabyte of files)
    prefix = str(work / "scale")
    try:
        dev.write_files(prefix)
        for part, want in gold["md5"].items():
            h, size = hashlib.md5(), 0
            with open(f"{prefix}.fmsi.{part}", "rb") as f:
                while chunk := f.read(1 << 24):
                    h.update(chunk)
                    size += len(chunk)
            assert size == gold["bytes"][part] and h.hexdigest() == want, part
        if n > 200_000_000:
            dev.close()
            return
    finally:
        if work is not tmp_path and n > 200_000_000:
            shutil.rmtree(work, ignore_errors=True)
    import torch
    loaded = fm.DeviceIndex.from_files(prefix, use_klcp=True, prefix_len=13).enable_pair_layout().enable_verify()
    nq = 1 << 21
    d_kmers = torch.empty(nq, dtype=torch.int64, device="cuda")
    dev.synthetic_queries_dev(k, 5, 0, nq, d_kmers.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for mode in (fm.MODE_OR, fm.MODE_ALL):
        a = device_answers(fm, loaded, d_kmers, nq, k, mode | fm.STRICT | fm.KERNEL_PHASED)
        b = device_answers(fm, dev, d_kmers, nq, k, mode | fm.STRICT | fm.KERNEL_LOCKSTEP)
        assert np.array_equal(a, b)
    # -S on the loaded files: streaming == isolated windows
    L, n_reads = 2000, 256
    d_reads = torch.empty((n_reads * L) // 32 + 2, dtype=torch.int64, device="cuda")
    dev.synthetic_reads_dev(L, 3, 0, n_reads, d_reads.data_ptr(), 2000, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    packed = d_reads.cpu().numpy().view(np.uint64)
    starts = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    wins = np.full(n_reads, L - k + 1, dtype=np.uint32)
    assert np.array_equal(loaded.query_stream(packed, n_reads * L, starts, wins, k, fm.MODE_OR | fm.STRICT),
                          loaded.query_runs(packed, n_reads * L, starts, wins, k, fm.MODE_OR | fm.STRICT))


def test_auto_takes_the_sorted_walk_for_large_batches(fm):
    """FMSI_HIP_KERNEL_AUTO on a handle with the pair layout and no table: a batch of 2^22 k-mers and more goes through the batch-sorted walk
    (fmsi_sorted.hip) -- recognisable by its counters: the shared top of the search tree is walked once, far fewer LF steps than one search per
    k-mer and strand executes -- a smaller one through the per-k-mer pair kernel; the answers are the same either way."""
    import torch
    n, k = 100_000_000, 31
    dev = fm.DeviceIndex.build_synthetic(n, 4321, k, mask_permille=700).enable_pair_layout()
    s = torch.cuda.current_stream().cuda_stream
    for nq, expect_sorted in ((1 << 22, True), ((1 << 22) - 64, False)):
        d_kmers = torch.empty(nq, dtype=torch.int64, device="cuda")
        dev.synthetic_queries_dev(k, 9, 0, nq, d_kmers.data_ptr(), s)
        outs, steps = {}, {}
        for name, kern in (("auto", 0), ("pair", fm.KERNEL_PAIR), ("sorted", fm.KERNEL_SORTED)):
            d_out = torch.empty(nq, dtype=torch.uint8, device="cuda")
            d_cnt = torch.zeros(fm.N_COUNTERS, dtype=torch.int64, device="cuda")
            dev.query_kmers_dev(d_kmers.data_ptr(), nq, k, fm.MODE_ALL | fm.STRICT | kern, d_out.data_ptr(), d_cnt.data_ptr(), s)
            torch.cuda.synchronize()
            outs[name], steps[name] = d_out.cpu().numpy(), int(d_cnt[0].item())
        assert np.array_equal(outs["auto"], outs["pair"]) and np.array_equal(outs["sorted"], outs["pair"])
        assert steps["sorted"] < 0.7 * steps["pair"]
        assert steps["auto"] == (steps["sorted"] if expect_sorted else steps["pair"]), (nq, steps)
