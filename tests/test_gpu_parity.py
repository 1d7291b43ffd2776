"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C-ABI of
libfmsi_hip.so (and through the `fmsi` CLI), against the CPU oracle and the committed golden outputs of
the real reference.  Everything is integer work: the bar is bit-exact.
"""
import json
import random

import numpy as np
import pytest

import oracle_py
from conftest import CLI_DATASETS, DATASETS, FN_CASES, GOLDEN, MODES, REPO, bgzf_bytes, ms_of, golden_fn_out, golden_out, index_prefix, messy_fasta, mixed_kmers, query_file, revcomp, synthetic_ms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(product_built):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on an MI355X")
    return product_built


def expected_fast(fwd, rc, mode_all):
    """Forward strand first, reverse complement only when needed (src/fms_index.h:289-299 without swap)."""
    if mode_all:
        return np.where(fwd != -1, fwd, rc) == 1
    return (fwd == 1) | (rc == 1)


@pytest.mark.parametrize("name", list(DATASETS))
def test_both_strands_equal_oracle(fm, oracle_built, name):
    k = DATASETS[name][0]
    orc = oracle_py.Oracle.load(index_prefix(name))
    dev = fm.DeviceIndex.from_files(index_prefix(name))
    ms_file = {"integration_a": "ref_data/integration_a.fa", "integration_b": "ref_data/integration_b.fa",
               "sars_k13": "ref_data/GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13",
               "sars_k31": "ref_data/GCF_test2_ones.fna.ms.k31", "mixed_k11": "mixed.fa"}[name]
    ms = "".join(l.strip() for l in open(GOLDEN / ms_file) if not l.startswith(">"))
    kmers = mixed_kmers(ms, k, 3000, seed=17)
    if k <= 6:   # exhaustive for tiny k
        import itertools
        kmers = ["".join(p) for p in itertools.product("ACGT", repeat=k)]
    raw = "".join(kmers).encode()
    packed = fm.pack_kmers(kmers, k)
    dev_t = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=min(k, 6))   # private prefix table of min(k, 6) steps
    variants = [(dev, fm.KERNEL_LOCKSTEP), (dev, fm.KERNEL_REFILL), (dev_t, fm.KERNEL_REFILL)]
    # two-base-step kernel over the 128-B pair layout: no table, and tables of both parities of k - t
    dev_p = fm.DeviceIndex.from_files(index_prefix(name)).enable_pair_layout()
    dev_pe = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=min(k, 6)).enable_pair_layout()
    dev_po = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=max(1, min(k, 6) - 1)).enable_pair_layout()
    variants += [(dev_p, fm.KERNEL_PAIR), (dev_pe, fm.KERNEL_PAIR), (dev_po, fm.KERNEL_PAIR), (dev_pe, fm.KERNEL_REFILL)]
    # the batch-sorted lock-step walk (fmsi_sorted.hip): no table read, whatever the handle has
    variants += [(dev_p, fm.KERNEL_SORTED), (dev_pe, fm.KERNEL_SORTED)]
    # phased form (table lookups in a kernel of their own; what AUTO picks with a pair layout and a table)
    variants += [(dev_pe, fm.KERNEL_PHASED), (dev_po, fm.KERNEL_PHASED), (dev_po, 0)]
    # ... with text verification of one-row table intervals (needs t < k)
    dev_pv = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=max(1, min(k, 6) - 1)).enable_pair_layout().enable_verify()
    variants += [(dev_pv, fm.KERNEL_PHASED), (dev_pv, 0), (dev_pv, fm.KERNEL_PAIR)]
    if k > 8:
        dev_pv8 = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=8).enable_pair_layout().enable_verify()
        variants += [(dev_pv8, fm.KERNEL_PHASED)]
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk(raw, len(kmers), k, mode_all)
        for d, kern in variants:
            got_f, got_r = fm.decode_strict(d.query_kmers(packed, k, flag | fm.STRICT | kern))
            assert np.array_equal(got_f, fwd) and np.array_equal(got_r, rc)
            fast = d.query_kmers(packed, k, flag | kern)
            assert np.array_equal(fast.astype(bool), expected_fast(fwd, rc, mode_all))
    # lookup (orders)
    fwd, rc, of, orc_r = orc.ternaries_bulk(raw, len(kmers), k, False, orders=True)
    for d, kern in variants:
        got = d.query_kmers(packed, k, fm.ORDERS | fm.STRICT | kern)
        assert np.array_equal(got[:, 0], of) and np.array_equal(got[:, 1], orc_r)
        got1 = d.query_kmers(packed, k, fm.ORDERS | kern)
        assert np.array_equal(got1, np.where(of >= 0, of, orc_r))


def test_windows_equal_kmers_and_oracle(fm, oracle_built):
    name, k = "sars_k31", 31
    orc = oracle_py.Oracle.load(index_prefix(name))
    dev = fm.DeviceIndex.from_files(index_prefix(name))
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_test2_ones.fna.ms.k31") if not l.startswith(">")).upper()
    rng = random.Random(5)
    seq = ms[2000:6000] + "".join(rng.choice("ACGT") for _ in range(500)) + revcomp(ms[9000:12000])
    nwin = len(seq) - k + 1
    kmers = [seq[i:i + k] for i in range(nwin)]
    fwd, rc = orc.ternaries_bulk("".join(kmers).encode(), nwin, k, True)
    words = fm.pack_sequence(seq.encode())
    got = dev.query_windows(words, len(seq), k, fm.MODE_ALL | fm.STRICT)
    gf, gr = fm.decode_strict(got)
    assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
    starts = np.array(rng.sample(range(nwin), 1000), dtype=np.uint64)
    got2 = dev.query_windows(words, len(seq), k, fm.MODE_ALL | fm.STRICT, window_start=starts)
    assert np.array_equal(got2, got[starts.astype(np.int64)])
    # the same windows as isolated k-mers
    got3 = dev.query_kmers(fm.pack_kmers(kmers, k), k, fm.MODE_ALL | fm.STRICT)
    assert np.array_equal(got3, got)


def test_empty_and_argument_errors(fm):
    dev = fm.DeviceIndex.from_files(index_prefix("sars_k31"))
    assert dev.query_kmers(np.zeros(0, dtype=np.uint64), 31).size == 0
    with pytest.raises(fm.FmsiError) as e:
        dev.query_kmers(np.zeros(4, dtype=np.uint64), 33)
    assert e.value.code == fm.E_RANGE
    with pytest.raises(fm.FmsiError):
        dev.query_windows(np.zeros(4, dtype=np.uint64), 40, 31, n_windows=100)   # runs past the sequence
    info = dev.info
    assert info.n == 29900 and info.k == 31 and info.block_bytes == 64 and info.block_symbols == 128


def test_lf_step_counter_equals_oracle(fm, oracle_built):
    """The in-kernel LF-step counter (roofline input) equals the oracle's count of
    update_range calls with i != j under the same strand schedule (both strands of every k-mer)."""
    name, k = "sars_k13", 13
    orc = oracle_py.Oracle.load(index_prefix(name))
    dev = fm.DeviceIndex.from_files(index_prefix(name))
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13") if not l.startswith(">"))
    kmers = mixed_kmers(ms, k, 5000, seed=3)
    orc.counters_reset()
    orc.ternaries_bulk("".join(kmers).encode(), len(kmers), k, True)
    want = orc.counters()
    for kern in (fm.KERNEL_LOCKSTEP, fm.KERNEL_REFILL):
        out, cn = dev.query_kmers(fm.pack_kmers(kmers, k), k, fm.MODE_ALL | fm.STRICT | kern, counters=True)
        assert int(cn[0]) == want["lf_steps"]
        assert int(cn[3]) == want["strand_searches"] == 2 * len(kmers)
        assert int(cn[2]) == want["mask_probes"]
        assert int(cn[6]) == len(kmers)
    # with a t-step prefix table the walk starts t steps in: fewer LF steps, one lookup per strand search
    dev_t = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=5)
    out_t, cn_t = dev_t.query_kmers(fm.pack_kmers(kmers, k), k, fm.MODE_ALL | fm.STRICT, counters=True)
    assert np.array_equal(out_t, out)
    assert int(cn_t[4]) == 2 * len(kmers) and int(cn_t[0]) < want["lf_steps"] - 4 * len(kmers)
    out_n, cn_n = dev_t.query_kmers(fm.pack_kmers(kmers, k), k, fm.MODE_ALL | fm.STRICT | fm.NO_PREFIX_TABLE, counters=True)
    assert np.array_equal(out_n, out) and int(cn_n[0]) == want["lf_steps"] and int(cn_n[4]) == 0


# (the three thinned families below -- 80 cases -- cost 15 s of the suite: they run in full again; FMSI_TEST_EXTENDED=0 thins them)
EXTENDED = __import__("os").environ.get("FMSI_TEST_EXTENDED", "1") != "0"
CASES = [(n, kind, mode) for n, (k, kinds) in CLI_DATASETS.items() for kind in kinds for mode in MODES]


@pytest.mark.parametrize("name,kind,mode", CASES)
def test_cli_equals_reference_golden(fm, name, kind, mode):
    """`fmsi query|lookup` on the GPU reproduces the real reference's stdout byte for byte (default = exact
    replay of the strand predictor), incl. the non-max-ones `-O` cases whose output is order dependent."""
    k = CLI_DATASETS[name][0]
    args = MODES[mode][0]
    rc, out, err = fm.run_cli([*args, "-k", str(k), "-q", str(query_file(name, kind)), index_prefix(name)])
    assert rc == 0, err.decode()
    assert out == golden_out(name, kind, mode)


def test_cli_output_to_a_file_on_tmpfs(fm, tmp_path):
    """stdout redirected to a regular file (round 6, OutPrealloc in query_cli.cpp: the file's pages are allocated beyond its end while the index
    loads, and what is not written is given back): on tmpfs -- where the whole-process benchmarks write -- and on the test directory's file system the
    bytes are the pipe's, truncating and appending, and no unwritten pages stay allocated behind the end of the file."""
    import os
    import subprocess
    import tempfile
    name, kind = "sars_k31", CLI_DATASETS["sars_k31"][1][0]
    k = CLI_DATASETS[name][0]
    q = str(query_file(name, kind))
    for mode in ("or", "all"):
        want = golden_out(name, kind, mode)
        for where in ("/dev/shm", str(tmp_path)):
            if not os.path.isdir(where):
                continue
            for how in ("wb", "ab"):
                with tempfile.NamedTemporaryFile(dir=where, delete=False) as f:
                    path = f.name
                try:
                    head = b"kept\n" if how == "ab" else b""
                    with open(path, "wb") as f:
                        f.write(head)
                    with open(path, how) as so:
                        res = subprocess.run([str(fm.CLI_PATH), *MODES[mode][0], "-k", str(k), "-q", q, index_prefix(name)], stdout=so, stderr=subprocess.PIPE)
                    assert res.returncode == 0, res.stderr.decode()[-400:]
                    with open(path, "rb") as f:
                        assert f.read() == head + want, (mode, where, how)
                    st = os.stat(path)
                    assert st.st_blocks * 512 <= st.st_size + (1 << 20), (mode, where, how, st.st_blocks * 512, st.st_size)
                finally:
                    os.unlink(path)


@pytest.mark.parametrize("name", ["sars_k31", "sars_k13", "repeats_k40"])
def test_cli_streams_without_S_and_survives_a_bad_klcp_file(fm, tmp_path, monkeypatch, name):
    """Since round 4 `fmsi query` / `lookup` WITHOUT -S stream their windows through the kLCP kernel when the index has its .klcp file
    (the interval a streamed step reaches is the fresh search's: src/fms_index.h:106-115) -- the bytes stay the reference's, with the
    streaming on (default), off (FMSI_HIP_CLI_STREAM=0), and with a .klcp file that does not load: the reference never opens that file
    without -S, so such a run must answer all the same (by the searches), while -S on it fails."""
    import shutil
    k, kinds = CLI_DATASETS[name]
    kind = kinds[0]
    q = str(query_file(name, kind))
    for mode in ("or", "all", "lookup"):
        want = golden_out(name, kind, mode)
        for env in (None, "0"):
            if env is None:
                monkeypatch.delenv("FMSI_HIP_CLI_STREAM", raising=False)
            else:
                monkeypatch.setenv("FMSI_HIP_CLI_STREAM", env)
            rc, out, err = fm.run_cli([*MODES[mode][0], "--stats", "-k", str(k), "-q", q, index_prefix(name)])
            assert rc == 0 and out == want, (mode, env, err.decode()[-300:])
    monkeypatch.delenv("FMSI_HIP_CLI_STREAM", raising=False)
    bad = str(tmp_path / "bad")
    for part in ("ac_gt", "ac", "gt", "mask", "klcp", "misc"):
        shutil.copy(f"{index_prefix(name)}.fmsi.{part}", f"{bad}.fmsi.{part}")
    with open(f"{bad}.fmsi.klcp", "r+b") as f:
        f.truncate(9)
    for mode in ("or", "all", "lookup"):
        rc, out, err = fm.run_cli([*MODES[mode][0], "-k", str(k), "-q", q, bad])
        assert rc == 0 and out == golden_out(name, kind, mode), (mode, err.decode()[-300:])
    rc, out, err = fm.run_cli(["query", "-S", "-k", str(k), "-q", q, bad])
    assert rc != 0


# (`full` in every presence mode; `compact`, `background` and `candidates` in the two that differ most -- the default mode and `-S -O` resp. `-O`: the suite has a time limit)
# (FMSI_TEST_EXTENDED=1: every query file of every data set for `compact` / `background` / `candidates` as well; by default those take each data set's first)
@pytest.mark.parametrize("name,kind,mode,tables", [(*c, t) for c in CASES for t in ("full", "compact", "candidates", "background")
                                                   if ((t == "full" and c[2] in ("or", "all", "or_S", "all_S")) or (t in ("compact", "background") and c[2] in ("or", "all_S")) or
                                                       (t == "candidates" and c[2] in ("or", "all"))) and (EXTENDED or t == "full" or c[1] == CLI_DATASETS[c[0]][1][0])])
def test_cli_with_full_tables_equals_reference_golden(fm, name, kind, mode, tables):
    """--tables full (the k-mer dictionary, the configuration bench.py measures; with it `-S` presence queries take the dictionary
    too), --tables compact (the super-k-mer dictionary alone where 23 <= k <= 32, the k-mer dictionary elsewhere) and --tables candidates (16-deep prefix table + candidate tables) change the device path of the presence modes and not
    a byte of the output -- incl. the exact replay of the strand predictor from the dictionary's both-strand answers; an unknown
    value is a usage error."""
    k = CLI_DATASETS[name][0]
    rc, out, err = fm.run_cli([*MODES[mode][0], "--tables", tables, "-k", str(k), "-q", str(query_file(name, kind)), index_prefix(name)])
    assert rc == 0, err.decode()
    assert out == golden_out(name, kind, mode)
    if tables == "full" and mode == "or":
        rc, out, err = fm.run_cli([*MODES[mode][0], "--tables", "huge", "-k", str(k), "-q", str(query_file(name, kind)), index_prefix(name)])
        assert rc == 1 and out == b""


@pytest.mark.parametrize("mode", ["or", "or_S", "all", "all_S"])
def test_cli_fast_mode(fm, mode):
    """--fast is exact for OR always and for -O on the max-ones fixture."""
    name, k = "sars_k31", 31
    rc, out, err = fm.run_cli([*MODES[mode][0], "--fast", "-k", str(k), "-q", str(query_file(name, "rand")), index_prefix(name)])
    assert rc == 0, err.decode()
    assert out == golden_out(name, "rand", mode)
    rc, out, err = fm.run_cli([*MODES["or"][0], "--fast", "-q", str(query_file("mixed_k11", "rand")), index_prefix("mixed_k11")])
    assert rc == 0 and out == golden_out("mixed_k11", "rand", "or")


# (one thread and 1 K batches in the two modes whose replay crosses batches the hardest, every mode with five threads: the suite has a time limit;
# FMSI_TEST_EXTENDED=1 runs the full product)
@pytest.mark.parametrize("name,kind,mode,threads,batch",
                         [(n, kd, m, t, b) for m in MODES for (n, kd) in [("sars_k13", "rand"), ("mixed_k11", "rand"), ("repeats_k40", "rand"), ("integration_a", "edge")]
                          for (t, b) in [(1, 1024), (5, 5000)] if EXTENDED or t == 5 or m in ("all_S", "lookup")])
def test_cli_pipeline_is_batch_and_thread_invariant(fm, name, kind, mode, threads, batch):
    """The host pipeline (reader -> team packing -> device -> replay -> team formatting) must not change a byte whatever
    the batch cut and the team size: the predictor state crosses batches, runs are re-packed word aligned, and the
    output lines are assembled by several threads."""
    k = CLI_DATASETS[name][0]
    rc, out, err = fm.run_cli([*MODES[mode][0], "-k", str(k), "--threads", str(threads), "--batch", str(batch), "-q",
                               str(query_file(name, kind)), index_prefix(name)])
    assert rc == 0, err.decode()
    assert out == golden_out(name, kind, mode)


def test_cli_stdin_gzip_and_small_batches(fm, tmp_path):
    import gzip
    name, k = "sars_k13", 13
    data = query_file(name, "rand").read_bytes()
    gz = tmp_path / "q.fa.gz"
    gz.write_bytes(gzip.compress(data))
    want = golden_out(name, "rand", "all_S")
    rc, out, err = fm.run_cli(["query", "-S", "-O", "-q", str(gz), "--batch", "1024", index_prefix(name)])
    assert rc == 0 and out == want
    rc, out, err = fm.run_cli(["query", "-S", "-O", index_prefix(name)], stdin=data)
    assert rc == 0 and out == want
    rc, out, err = fm.run_cli(["query", "-q", str(tmp_path / "missing.fa"), index_prefix(name)])
    assert rc == -6 or rc == 134   # uncaught std::invalid_argument -> abort, like the reference


@pytest.fixture(scope="module")
def synthetic_index(oracle_built, tmp_path_factory):
    """1.2 M-row max-ones index built by the oracle (oracle index bytes == reference bytes, test_oracle.py)."""
    k = 31
    ms = synthetic_ms(1_200_000, k, seed=12345)
    d = tmp_path_factory.mktemp("syn")
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=True)
    orc.dump(str(d / "syn"))
    return ms, k, str(d / "syn"), orc


def test_synthetic_random_queries_equal_oracle(fm, synthetic_index):
    ms, k, prefix, orc = synthetic_index
    dev = fm.DeviceIndex.from_files(prefix)
    kmers = mixed_kmers(ms, k, 200_000, seed=42)
    raw = "".join(kmers).encode()
    orc.predictor_reset()
    _, want = orc.time_isolated_kmers(raw, len(kmers), k, oracle_py.MODE_ALL)   # the reference loop incl. predictor
    packed = fm.pack_kmers(kmers, k)
    got = dev.query_kmers(packed, k, fm.MODE_ALL)
    assert np.array_equal(got, want)
    assert 0.45 < got.mean() < 0.55
    orc.predictor_reset()
    _, want_or = orc.time_isolated_kmers(raw, len(kmers), k, oracle_py.MODE_OR)
    assert np.array_equal(dev.query_kmers(packed, k, fm.MODE_OR), want_or)


def test_properties_at_scale(fm, synthetic_index):
    """Size-independent properties on 16 M queries (no oracle needed):
    every k-mer of a max-ones superstring is present on both strands; reverse-complement symmetry;
    launches are deterministic; windows == isolated k-mers."""
    ms, k, prefix, _ = synthetic_index
    dev = fm.DeviceIndex.from_files(prefix)
    up = ms.upper().encode()
    words = fm.pack_sequence(up)
    nwin = len(up) - k + 1
    pos = dev.query_windows(words, len(up), k, fm.MODE_ALL | fm.STRICT)
    f, r = fm.decode_strict(pos)
    assert (f == 1).all()                       # all superstring k-mers are ON occurrences
    assert ((r == 1) | (r == -1)).all()         # the RC strand is either absent or ON (max-ones)
    rng = np.random.default_rng(7)
    n = 16_000_000
    kmers = rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
    starts = rng.integers(0, nwin, size=n // 2, dtype=np.uint64)
    a = dev.query_kmers(kmers, k, fm.MODE_OR | fm.STRICT)
    dev.profile(True)
    b = dev.query_kmers(kmers, k, fm.MODE_OR | fm.STRICT)
    launches, ms_total = dev.profile_read()
    dev.profile(False)
    print(f"\n[perf] refill strict: {n} random 31-mers in {ms_total:.3f} ms ({launches} launch) = {n / ms_total / 1e3:.1f} M k-mers/s")
    assert np.array_equal(a, b)
    # reverse complement of a packed k-mer, on the host
    sub = kmers[:500_000]
    c = ((sub[:, None] >> (2 * np.arange(k, dtype=np.uint64))) & np.uint64(3))
    rcodes = (np.uint64(3) - c)[:, ::-1]
    rck = np.bitwise_or.reduce(rcodes << (2 * np.arange(k, dtype=np.uint64)), axis=1)
    fa, ra = fm.decode_strict(a[:500_000])
    fb, rb = fm.decode_strict(dev.query_kmers(rck, k, fm.MODE_OR | fm.STRICT))
    assert np.array_equal(fa, rb) and np.array_equal(ra, fb)
    # random 31-mers are (almost surely) absent from a 1.2 M-char text
    assert (fa == 1).mean() < 1e-3
    w = dev.query_windows(words, len(up), k, fm.MODE_ALL, window_start=starts)
    assert (w == 1).all()


@pytest.mark.parametrize("t", [13, 16])
def test_large_prefix_tables(fm, oracle_built, t):
    """Prefix tables far larger than the index (4^13 = 67 M and 4^16 = 4.3 G entries, 34 GB): same answers as the oracle
    from every kernel that takes the table -- single steps, pair steps (both parities of k - t) and streaming."""
    name, k = "sars_k31", 31
    orc = oracle_py.Oracle.load(index_prefix(name), use_klcp=True)
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_test2_ones.fna.ms.k31") if not l.startswith(">"))
    kmers = mixed_kmers(ms, k, 20000, seed=t)
    raw = "".join(kmers).encode()
    packed = fm.pack_kmers(kmers, k)
    dev = fm.DeviceIndex.from_files(index_prefix(name), use_klcp=True, prefix_len=t).enable_pair_layout().enable_verify()
    assert dev.info.prefix_len == t
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk(raw, len(kmers), k, mode_all)
        for kern in (fm.KERNEL_PHASED, fm.KERNEL_PAIR, fm.KERNEL_REFILL):
            gf, gr = fm.decode_strict(dev.query_kmers(packed, k, flag | fm.STRICT | kern))
            assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
            fast = dev.query_kmers(packed, k, flag | kern)
            assert np.array_equal(fast.astype(bool), expected_fast(fwd, rc, mode_all))
    up = ms.upper()
    reads = [up[100:700], revcomp(up[2000:2900]), up[5000:5031]]
    seq = "".join(reads)
    starts = np.cumsum([0] + [len(r) for r in reads[:-1]])
    wins = [len(r) - k + 1 for r in reads]
    wk = [r[i:i + k] for r in reads for i in range(len(r) - k + 1)]
    fwd, rc = orc.ternaries_bulk("".join(wk).encode(), len(wk), k, True)
    gf, gr = fm.decode_strict(dev.query_stream(fm.pack_sequence(seq.encode()), len(seq), starts, wins, k, fm.MODE_ALL | fm.STRICT))
    assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
    gf, gr = fm.decode_strict(dev.query_runs(fm.pack_sequence(seq.encode()), len(seq), starts, wins, k, fm.MODE_ALL | fm.STRICT))
    assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)


@pytest.mark.parametrize("t,k", [(3, 9), (4, 12), (5, 7), (6, 31), (7, 8)])
def test_text_verification_on_repeats_and_text_start(fm, oracle_built, t, k):
    """Text verification (fmsi_hip_index_enable_verify) where it is delicate: k-mers that occur two and three times with
    DIFFERENT mask bits (the first match decides -O, any match the default mode), candidates that fail only in their
    last base, k-mers at the very start of the text (position < k - t), reverse complements, and table depths that leave
    1..3 rows often.  Every mode against the oracle, strand by strand."""
    rng = random.Random(100 * t + k)
    core = "".join(rng.choice("ACGT") for _ in range(k + 6))
    parts = []
    for rep in range(3):   # the same stretch three times, with different case (mask) each time and different neighbours
        piece = core.upper() if rep == 0 else (core.lower() if rep == 1 else "".join(c.upper() if rng.random() < 0.5 else c.lower() for c in core))
        parts.append("".join(rng.choice("ACGTacgt") for _ in range(rng.randrange(20, 60))) + piece)
    near = core[:k - 1] + ("A" if core[k - 1] != "A" else "C") + core[k:]   # equal but for one base
    ms = core[:k + 2].upper() + "".join(parts) + near.lower() + "".join(rng.choice("ACGTacgt") for _ in range(400))
    ms = ms[:len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    dev = fm.DeviceIndex.build(ms.encode(), k, prefix_len=t).enable_pair_layout().enable_verify()
    up = ms.upper()
    kmers = [up[p:p + k] for p in range(len(up) - k + 1)]
    kmers += [revcomp(x) for x in kmers]
    kmers += ["".join(rng.choice("ACGT") for _ in range(k)) for _ in range(500)]
    kmers += [x[:-1] + rng.choice("ACGT") for x in kmers[:300]] + [rng.choice("ACGT") + x[1:] for x in kmers[:300]]
    raw = "".join(kmers).encode()
    packed = fm.pack_kmers(kmers, k)
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk(raw, len(kmers), k, mode_all)
        for kern in (fm.KERNEL_PHASED, 0):
            gf, gr = fm.decode_strict(dev.query_kmers(packed, k, flag | fm.STRICT | kern))
            assert np.array_equal(gf, fwd), np.nonzero(gf != fwd)[0][:10]
            assert np.array_equal(gr, rc), np.nonzero(gr != rc)[0][:10]
            fast = dev.query_kmers(packed, k, flag | kern)
            assert np.array_equal(fast.astype(bool), expected_fast(fwd, rc, mode_all))
    assert (fwd == 0).any() and (fwd == 1).any() and (rc == 1).any()
    _, cn = dev.query_kmers(packed, k, fm.MODE_ALL | fm.STRICT, counters=True)
    _, cn_steps = dev.query_kmers(packed, k, fm.MODE_ALL | fm.STRICT | fm.KERNEL_PAIR, counters=True)
    assert int(cn[6]) == len(kmers) and int(cn[0]) < int(cn_steps[0])   # candidates settled searches that otherwise take steps
    fwd, rc, of, orr = orc.ternaries_bulk(raw, len(kmers), k, False, orders=True)
    got = dev.query_kmers(packed, k, fm.ORDERS | fm.STRICT)   # lookup never verifies: ranks come from the steps
    assert np.array_equal(got[:, 0], of) and np.array_equal(got[:, 1], orr)


@pytest.mark.parametrize("narrow", [False, True])
@pytest.mark.parametrize("t,k", [(4, 11), (6, 21), (8, 23)])
def test_candidate_entries_of_up_to_seven_occurrences(fm, oracle_built, t, k, narrow, monkeypatch):
    """The lookup kernel's one-lane form (k - t <= 15) on keys that occur 1..9 times over both strands with mixed mask
    bits: the 32-byte entries list up to seven occurrences (forward copies, then reverse-complement copies), the
    16-byte ones (FMSI_HIP_NARROW_CANDIDATES) three, and what exceeds them goes to the stepping kernel.  Every mode
    against the oracle strand by strand, isolated k-mers and runs."""
    if narrow:
        monkeypatch.setenv("FMSI_HIP_NARROW_CANDIDATES", "1")
    rng = random.Random(1000 * t + k + (7 if narrow else 0))
    rnd = lambda a, b: "".join(rng.choice("ACGTacgt") for _ in range(rng.randrange(a, b)))
    ms = rnd(30, 50)
    cores = []
    for m_f, m_r in [(4, 0), (2, 3), (7, 0), (3, 4), (5, 4), (0, 5), (1, 1), (6, 1), (8, 0)]:
        core = "".join(rng.choice("ACGT") for _ in range(k + 5))
        cores.append(core)
        copies = [core] * m_f + [revcomp(core)] * m_r
        rng.shuffle(copies)
        for c in copies:   # each copy with its own case pattern (mask bits) and neighbours
            ms += "".join(ch.upper() if rng.random() < 0.5 else ch.lower() for ch in c) + rnd(3, 25)
    ms += rnd(300, 400)
    ms = ms[:len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    dev = fm.DeviceIndex.build(ms.encode(), k, prefix_len=t).enable_pair_layout().enable_verify()
    assert int(dev.info.candidate_entry_bytes) == (16 if narrow else 32)
    up = ms.upper()
    kmers = [up[p:p + k] for p in range(len(up) - k + 1)]
    kmers += [revcomp(x) for x in kmers]
    kmers += ["".join(rng.choice("ACGT") for _ in range(k)) for _ in range(400)]
    kmers += [x[:-1] + rng.choice("ACGT") for x in kmers[:400]] + [rng.choice("ACGT") + x[1:] for x in kmers[:400]]
    raw = "".join(kmers).encode()
    packed = fm.pack_kmers(kmers, k)
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk(raw, len(kmers), k, mode_all)
        gf, gr = fm.decode_strict(dev.query_kmers(packed, k, flag | fm.STRICT | fm.KERNEL_PHASED))
        assert np.array_equal(gf, fwd), np.nonzero(gf != fwd)[0][:10]
        assert np.array_equal(gr, rc), np.nonzero(gr != rc)[0][:10]
        fast = dev.query_kmers(packed, k, flag | fm.KERNEL_PHASED)
        assert np.array_equal(fast.astype(bool), expected_fast(fwd, rc, mode_all))
    assert (fwd == 0).any() and (fwd == 1).any() and (rc == 0).any() and (rc == 1).any()
    # the same windows as runs of the superstring itself and of its reverse complement
    seq = up + revcomp(up)
    starts = np.array([0, len(up)], dtype=np.uint64)
    wins = np.array([len(up) - k + 1] * 2, dtype=np.uint32)
    wk = [seq[s:s + k] for s0 in (0, len(up)) for s in range(s0, s0 + len(up) - k + 1)]
    fwd, rc = orc.ternaries_bulk("".join(wk).encode(), len(wk), k, True)
    gf, gr = fm.decode_strict(dev.query_runs(fm.pack_sequence(seq.encode()), len(seq), starts, wins, k, fm.MODE_ALL | fm.STRICT))
    assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
    # the candidates settle what would otherwise take steps; with seven slots more than with three
    _, cn = dev.query_kmers(packed, k, fm.MODE_ALL | fm.STRICT, counters=True)
    _, cn_steps = dev.query_kmers(packed, k, fm.MODE_ALL | fm.STRICT | fm.KERNEL_PAIR, counters=True)
    assert int(cn[6]) == len(kmers) and int(cn[0]) < int(cn_steps[0])


def test_full_scale_cross_kernel_agreement(fm):
    """BASELINE.json's full size (human-genome scale, n = 2.8 G rows) with the production configuration (t = 16 signature
    table, pair layout, text verification): no CPU checker finishes here, so the size-independent property is that four
    independent algorithms agree bit for bit, strand by strand, on 4 M seeded k-mers (half present) -- lookups + candidate
    comparison (phased), lookups + two-base steps (one-kernel pair), single steps over the 64-byte layout (refill) and
    plain lock-step search without any table -- plus: every k-mer drawn from the superstring is found on its strand, and
    streaming a read gives what its isolated windows give.  (bench.py checks a sample of the same stream against the
    real reference on the host: cpu_baseline.matches_gpu.)"""
    import torch
    n, k, nq = 2_800_000_000, 31, 1 << 22
    dev = fm.DeviceIndex.build_synthetic(n, 12345, k, with_klcp=True, prefix_len=16).enable_pair_layout().enable_verify()
    s = torch.cuda.current_stream().cuda_stream
    d_kmers = torch.empty(nq, dtype=torch.int64, device="cuda")
    dev.synthetic_queries_dev(k, 42, 0, nq, d_kmers.data_ptr(), s)
    outs = {}
    for name, kern in (("phased", fm.KERNEL_PHASED), ("pair", fm.KERNEL_PAIR), ("refill", fm.KERNEL_REFILL),
                       ("lockstep", fm.KERNEL_LOCKSTEP | fm.NO_PREFIX_TABLE)):
        d_out = torch.empty(nq, dtype=torch.uint8, device="cuda")
        dev.query_kmers_dev(d_kmers.data_ptr(), nq, k, fm.MODE_ALL | fm.STRICT | kern, d_out.data_ptr(), stream=s)
        torch.cuda.synchronize()
        outs[name] = d_out.cpu().numpy()
    for name in ("pair", "refill", "lockstep"):
        assert np.array_equal(outs["phased"], outs[name]), name
    f, r = fm.decode_strict(outs["phased"])
    present = (f == 1) | (r == 1)
    assert 0.49 < present.mean() < 0.51 and not ((f == 1) & (r == 1)).any()
    assert ((f == 0) | (r == 0)).sum() == 0          # max-ones superstring: an occurrence is always ON
    # fast schedule == what the strict results imply
    d_out = torch.empty(nq, dtype=torch.uint8, device="cuda")
    dev.query_kmers_dev(d_kmers.data_ptr(), nq, k, fm.MODE_ALL, d_out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().astype(bool), present)
    # reads from the superstring (both orientations) and random reads: streaming == runs == isolated windows
    L, n_reads = 1000, 512
    words = (n_reads * L) // 32 + 2
    d_reads = torch.empty(words, dtype=torch.int64, device="cuda")
    dev.synthetic_reads_dev(L, 7, 0, n_reads, d_reads.data_ptr(), 0, s)
    torch.cuda.synchronize()
    packed = d_reads.cpu().numpy().view(np.uint64)
    starts = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    wins = np.full(n_reads, L - k + 1, dtype=np.uint32)
    a = dev.query_stream(packed, n_reads * L, starts, wins, k, fm.MODE_ALL | fm.STRICT)
    b = dev.query_runs(packed, n_reads * L, starts, wins, k, fm.MODE_ALL | fm.STRICT)
    assert np.array_equal(a, b)
    per_read = ((a & 3) == 2) | (((a >> 2) & 3) == 2)
    frac = per_read.reshape(n_reads, -1).mean(axis=1)
    assert ((frac == 0) | (frac == 1)).all() and 0.35 < (frac == 1).mean() < 0.65   # a read is from the superstring or it is not


def test_cross_kernel_agreement_on_repeat_families(fm):
    """The same property on a text with repeat families (2 x 10^8 rows, 30 % of its segments copies of 100 families at 2 %
    divergence: prefix-table keys with tens of occurrences, so the candidate entries overflow and the stepping kernel
    answers a tenth of the k-mers): the four algorithms agree strand by strand, and the two candidate formats agree."""
    import torch
    n, k, nq = 200_000_000, 31, 1 << 21
    rep = (300, 100, 20)
    dev = fm.DeviceIndex.build_synthetic(n, 99, k, prefix_len=16, repeats=rep).enable_pair_layout().enable_verify()
    s = torch.cuda.current_stream().cuda_stream
    d_kmers = torch.empty(nq, dtype=torch.int64, device="cuda")
    dev.synthetic_queries_dev(k, 42, 0, nq, d_kmers.data_ptr(), s)
    d_cnt = torch.zeros(fm.N_COUNTERS, dtype=torch.int64, device="cuda")
    outs = {}
    for name, kern in (("phased", fm.KERNEL_PHASED), ("pair", fm.KERNEL_PAIR), ("refill", fm.KERNEL_REFILL),
                       ("lockstep", fm.KERNEL_LOCKSTEP | fm.NO_PREFIX_TABLE)):
        d_out = torch.empty(nq, dtype=torch.uint8, device="cuda")
        dev.query_kmers_dev(d_kmers.data_ptr(), nq, k, fm.MODE_ALL | fm.STRICT | kern, d_out.data_ptr(),
                            counters_ptr=d_cnt.data_ptr() if name == "phased" else 0, stream=s)
        torch.cuda.synchronize()
        outs[name] = d_out.cpu().numpy()
    for name in ("pair", "refill", "lockstep"):
        assert np.array_equal(outs["phased"], outs[name]), name
    cn = d_cnt.cpu().numpy()
    assert int(cn[0]) > 0.3 * nq            # LF steps: the stepping kernel had real work
    f, r = fm.decode_strict(outs["phased"])
    present = (f == 1) | (r == 1)
    assert 0.49 < present.mean() < 0.52     # (a random k-mer is never in the text; a drawn one always)
    for mode in (fm.MODE_ALL, fm.MODE_OR):
        d_out = torch.empty(nq, dtype=torch.uint8, device="cuda")
        dev.query_kmers_dev(d_kmers.data_ptr(), nq, k, mode, d_out.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().astype(bool), present)


@pytest.mark.parametrize("verify", [False, True])
@pytest.mark.parametrize("k,t", [(31, 8), (31, 10), (19, 7), (32, 9)])
def test_runs_cut_inside_the_lookup_kernel(fm, oracle_built, k, t, verify):
    """fmsi_hip_query_runs on the phased kernels cuts the windows inside the lookup kernel (no expanded k-mer array):
    ragged runs -- single windows, runs that straddle the kernel's chunks, overlapping and out-of-order starts -- must give,
    window by window, what the same k-mers give as isolated packed k-mers and what the oracle gives, in every mode."""
    rng = np.random.default_rng(1000 * k + t + verify)
    ms = synthetic_ms(60_000, k, seed=k + t, mask="mixed")
    orc = oracle_py.Oracle.construct(ms, k)
    dev = fm.DeviceIndex.build(ms.encode(), k, prefix_len=t).enable_pair_layout()
    if verify:
        dev.enable_verify()
    up = ms.upper()
    pieces = []
    while sum(map(len, pieces)) < 150_000:
        L = int(rng.choice([k, k + 1, k + 5, 200, 3000, 9000]))
        if rng.random() < 0.5:
            p0 = int(rng.integers(0, len(up) - L))
            piece = up[p0:p0 + L]
            if rng.random() < 0.5:
                piece = revcomp(piece)
        else:
            piece = "".join(np.array(list("ACGT"))[rng.integers(0, 4, L)])
        pieces.append(piece)
    seq = "".join(pieces)
    bounds = np.cumsum([0] + [len(x) for x in pieces])
    starts, wins = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        # a run per piece -- sometimes split into two runs that overlap by a few windows, sometimes cut short
        w = int(b - a) - k + 1
        if w > 10 and rng.random() < 0.3:
            cut = int(rng.integers(1, w))
            starts += [int(a), int(a) + max(0, cut - 3)]
            wins += [cut, w - max(0, cut - 3)]
        else:
            starts.append(int(a))
            wins.append(w if rng.random() < 0.8 else max(1, w // 2))
    order = rng.permutation(len(starts))
    starts = np.asarray(starts, dtype=np.uint64)[order]
    wins = np.asarray(wins, dtype=np.uint32)[order]
    wk = [seq[int(s0) + i:int(s0) + i + k] for s0, w in zip(starts, wins) for i in range(int(w))]
    fwd, rc = orc.ternaries_bulk("".join(wk).encode(), len(wk), k, True)
    fwd_or, rc_or = orc.ternaries_bulk("".join(wk).encode(), len(wk), k, False)
    packed_seq = fm.pack_sequence(seq.encode())
    packed_kmers = fm.pack_kmers(wk, k)
    for kern in (0, fm.KERNEL_PHASED):
        gf, gr = fm.decode_strict(dev.query_runs(packed_seq, len(seq), starts, wins, k, fm.MODE_ALL | fm.STRICT | kern))
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
        gf, gr = fm.decode_strict(dev.query_runs(packed_seq, len(seq), starts, wins, k, fm.MODE_OR | fm.STRICT | kern))
        assert np.array_equal(gf, fwd_or) and np.array_equal(gr, rc_or)
        for flag, mode_all, f_, r_ in ((fm.MODE_ALL, True, fwd, rc), (fm.MODE_OR, False, fwd_or, rc_or)):
            got = dev.query_runs(packed_seq, len(seq), starts, wins, k, flag | kern)
            assert np.array_equal(got.astype(bool), expected_fast(f_, r_, mode_all))
            assert np.array_equal(got, dev.query_kmers(packed_kmers, k, flag | kern))
    a, cn = dev.query_runs(packed_seq, len(seq), starts, wins, k, fm.MODE_ALL, counters=True)
    assert cn[6] == len(wk) and np.array_equal(a.astype(bool), expected_fast(fwd, rc, True))


def test_candidate_tables_are_for_the_index_k_only(fm, oracle_built):
    """The search itself is k-agnostic (any pattern length can be asked of a handle); the candidate tables are made for
    the index's k, so a query with another k must not touch them."""
    name, k = "sars_k31", 31
    orc = oracle_py.Oracle.load(index_prefix(name))
    dev = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=8).enable_pair_layout().enable_verify()
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_test2_ones.fna.ms.k31") if not l.startswith(">"))
    for kq in (31, 27, 19, 9):
        kmers = mixed_kmers(ms, kq, 3000, seed=kq)
        fwd, rc = orc.ternaries_bulk("".join(kmers).encode(), len(kmers), kq, True)
        gf, gr = fm.decode_strict(dev.query_kmers(fm.pack_kmers(kmers, kq), kq, fm.MODE_ALL | fm.STRICT))
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc), kq


@pytest.mark.parametrize("fastq_at", [None, 150, 0])
@pytest.mark.parametrize("mode", ["or", "all_S", "lookup"])
def test_block_parallel_reader_equals_the_ordinary_one(fm, oracle_built, tmp_path, monkeypatch, mode, fastq_at):
    """The CLI's block-parallel FASTA reader (mapped file cut at '>' lines, one ordinary reader per part) against the
    ordinary reader and against the oracle's restatement of kseq, on a file of awkward records and with ranges of a few
    kilobytes so that cuts fall everywhere; with FASTQ records from some point on it has to hand over to the ordinary
    reader at the right record."""
    name, k = "sars_k13", 13
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13") if not l.startswith(">"))
    data = messy_fasta(ms, random.Random(31 + (fastq_at or 7)), fastq_at)
    if fastq_at == 0:
        data = data[data.index(b"@"):]        # a file that starts as FASTQ: cut at '@' lines whose next-but-one line starts with '+'
                                              # (these records have '>' and '@' at the start of quality lines: parts that do not end clean)
    q = tmp_path / "messy.fa"
    q.write_bytes(data)
    args = [*MODES[mode][0], "-k", str(k), "--batch", "4096", "-q", str(q), index_prefix(name)]
    monkeypatch.setenv("FMSI_HIP_SERIAL_READER", "1")
    rc, want, err = fm.run_cli(args)
    assert rc == 0, err.decode()
    monkeypatch.delenv("FMSI_HIP_SERIAL_READER")
    monkeypatch.setenv("FMSI_HIP_READER_MIN_BYTES", "1")
    rc, got, err = fm.run_cli(args)
    assert rc == 0, err.decode()
    assert got == want
    checker = oracle_py.REF_CLI if oracle_py.ref_available() else oracle_py.ORACLE_CLI   # the real reference when it is there
    rc, ref, err = oracle_py.run_cli(checker, [a for a in args if a not in ("--batch", "4096")])
    assert rc == 0 and got == ref
    assert got.count(b"\n") > (100 if fastq_at != 0 else 50)


@pytest.mark.parametrize("mode", ["or", "all", "or_S", "all_S"])
@pytest.mark.parametrize("name,k,shape", [("sars_k31", 31, "pure"), ("mixed_k11", 11, "no_tail"), ("both_strands", 13, "odd")])
def test_cli_kmer_batches_equal_the_reference(fm, oracle_built, tmp_path, monkeypatch, name, k, shape, mode):
    """A query file of ONE k-mer per record -- the shape of BASELINE's isolated-k-mer configurations -- takes a path of its own through
    `fmsi query` (cli_reader.h KmerPart: parsed, validated, packed and its output line written in one pass; fmsi_hip_query_kmers; the
    predictor's recurrence as a table): byte-identical to the real reference's record loop (src/main.cpp:328-373) and to the general
    path, in the four presence modes -- `-S` on a one-window chunk is the single step, src/fms_index.h:181-254 -- incl. a non-max-ones
    mask under -O, where what is printed depends on the predictor's state, and files where records of other shapes (CRLF, N, shorter,
    longer, two lines, blank lines, FASTQ) send their range back to the ordinary rules."""
    if name == "both_strands":   # a text with reverse-complemented copies of its own pieces under a random mask: k-mers on both strands with different bits
        rng = random.Random(77)
        base = "".join(rng.choice("ACGT") for _ in range(30000))
        pieces = [base]
        for _ in range(12):
            p0 = rng.randrange(0, len(base) - 3000)
            pieces.append(revcomp(base[p0:p0 + rng.choice([500, 1500, 3000])]))
        ms = "".join(ch if rng.random() < 0.6 else ch.lower() for ch in "".join(pieces))
        ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
        prefix = str(tmp_path / "both")
        oracle_py.Oracle.construct(ms, k, use_klcp=True).dump(prefix)
    else:
        prefix = index_prefix(name)
        ms = ms_of(name)
    up = ms.upper()
    rng = random.Random(k * 7 + len(shape))
    recs = []
    for i in range(20000):
        p0 = rng.randrange(0, len(up) - k)
        s = up[p0:p0 + k] if rng.random() < 0.6 else "".join(rng.choice("ACGT") for _ in range(k))
        if rng.random() < 0.3:
            s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        if rng.random() < 0.1:
            s = s.lower()
        head = f"{i}" if rng.random() < 0.7 else f"q{i} comment >x"
        rec = f">{head}\n{s}\n"
        if shape == "odd" and i % 1511 == 1510:
            rec = rng.choice([f">{head}\r\n{s}\r\n", f">{head}\n{s[:3]}N{s[4:]}\n", f">{head}\n{s[:-1]}\n", f">{head}\n{s}AC\n", f">{head}\n{s[:6]}\n{s[6:]}\n",
                              f">{head}\n{s}\n\n", f"@{head}\n{s}\n+\n{'I' * k}\n"])
        recs.append(rec)
    text = "".join(recs)
    q = tmp_path / "kmers.fa"
    q.write_bytes((text.rstrip("\n") if shape == "no_tail" else text).encode())
    args = [*MODES[mode][0], "-k", str(k), "--stats", "-q", str(q), prefix]
    checker = oracle_py.REF_CLI if oracle_py.ref_available() else oracle_py.ORACLE_CLI
    rc, ref, err = oracle_py.run_cli(checker, [a for a in args if a != "--stats"])
    assert rc == 0, err.decode()
    monkeypatch.setenv("FMSI_HIP_READER_MIN_BYTES", "1")
    for opts in ([], ["--batch", "20000", "--threads", "5"]) + ((["--tables", "full"],) if mode == "all" else ()):
        rc, got, err = fm.run_cli(args[:1] + opts + args[1:])
        assert rc == 0, err.decode()
        assert got == ref, (opts, mode)
        stats = json.loads([l for l in err.decode().splitlines() if l.startswith("{")][-1])
        assert stats["windows"] >= 19900
        if shape != "odd" or opts[:1] == ["--batch"]:   # (the path was taken -- with small ranges also by the ranges between two odd records)
            assert stats["kmer_batches"] > 0, (opts, mode)
    monkeypatch.setenv("FMSI_HIP_CLI_KMER_BATCHES", "0")
    rc, general, err = fm.run_cli(args)
    assert rc == 0 and general == ref and b'"kmer_batches": 0' in err


@pytest.mark.parametrize("shape", ["four_line", "four_line_crlf", "wrapped_adversarial", "truncated_tail", "short_quality_inside", "no_final_newline"])
def test_block_parallel_fastq_reader(fm, oracle_built, tmp_path, monkeypatch, shape):
    """FASTQ files through the block-parallel reader: 4-line records are cut at header lines ('@' line whose next-but-one line starts
    with '+'; quality lines that start with '@' or '+' do not fool it), every part is read by the ordinary reader (the reference's
    kseq rules) and must end between two records -- where it does not (wrapped records with adversarial quality lines, a truncated
    or corrupt record) the ordinary reader takes over at that range, so the output is the sequential parse's in every case."""
    name, k = "sars_k13", 13
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13") if not l.startswith(">")).upper()
    rng = random.Random(hash(shape) & 0xFFFF)
    nl = b"\r\n" if shape == "four_line_crlf" else b"\n"
    out = bytearray()
    for r in range(1500):
        L = rng.choice([13, 14, 40, 75, 151, 300])
        p = rng.randrange(0, len(ms) - L)
        seq = ms[p:p + L] if rng.random() < 0.6 else "".join(rng.choice("ACGTN") for _ in range(L))
        qual = "".join(rng.choice("@+>IIIIFFF#5") for _ in range(L))
        if r % 7 == 0:
            qual = "@" + qual[1:]
        if r % 11 == 0:
            qual = "+" + qual[1:]
        if shape == "short_quality_inside" and r == 700:
            qual = qual[: L // 2]
        if shape == "wrapped_adversarial":
            w = 30
            sl = [seq[i:i + w] for i in range(0, L, w)]
            ql = [qual[i:i + w] for i in range(0, L, w)]
            if len(ql) >= 3 and r % 3 == 0:     # a quality line starting with '@' whose next-but-one line starts with '+': a false header
                ql[0] = "@" + ql[0][1:]
                ql[2] = "+" + ql[2][1:]
            out += f"@r{r} c".encode() + nl + nl.join(x.encode() for x in sl) + nl + b"+" + nl + nl.join(x.encode() for x in ql) + nl
        else:
            out += f"@r{r} comment".encode() + nl + seq.encode() + nl + (b"+" if r % 2 else f"+r{r}".encode()) + nl + qual.encode() + nl
    data = bytes(out)
    if shape == "truncated_tail":
        data = data[: len(data) - 40]
    if shape == "no_final_newline":
        data = data.rstrip(b"\n")
    q = tmp_path / "q.fq"
    q.write_bytes(data)
    for mode in ("or", "all_S"):
        args = [*MODES[mode][0], "-k", str(k), "--batch", "4096", "-q", str(q), index_prefix(name)]
        monkeypatch.setenv("FMSI_HIP_SERIAL_READER", "1")
        rc, want, err = fm.run_cli(args)
        assert rc == 0, err.decode()
        monkeypatch.delenv("FMSI_HIP_SERIAL_READER")
        monkeypatch.setenv("FMSI_HIP_READER_MIN_BYTES", "1")
        rc, got, err = fm.run_cli([*args[:-1], "--stats", args[-1]])
        assert rc == 0, err.decode()
        monkeypatch.delenv("FMSI_HIP_READER_MIN_BYTES")
        assert got == want
        import json as _json
        par = _json.loads([l for l in err.decode().splitlines() if l.startswith("{")][-1])["block_parallel_read_bytes"]
        if shape in ("four_line", "four_line_crlf", "no_final_newline"):
            assert par == len(data)                     # the whole file went through the parallel parts
        elif shape == "wrapped_adversarial":
            assert par < len(data) // 4                 # a false header early on: the ordinary reader took over
        else:
            assert 0 < par < len(data)                  # up to the range with the broken record
        checker = oracle_py.REF_CLI if oracle_py.ref_available() else oracle_py.ORACLE_CLI
        rc, ref, err = oracle_py.run_cli(checker, [a for a in args if a not in ("--batch", "4096")])
        assert rc == 0 and got == ref
        assert got.count(b"\n") > (600 if shape == "short_quality_inside" else 1400)


@pytest.mark.parametrize("name,k", [("mixed_k11", 11), ("sars_k13", 13), ("both_strands", 13)])
def test_lookup_replay_on_large_batches(fm, oracle_built, tmp_path, name, k):
    """`fmsi lookup` on batches of a million windows and more: the replay of the strand predictor skips runs of windows with the
    same signs behind a step that left it unchanged, and must still print the other strand's order where it had swapped the
    strands -- `both_strands`: a text that holds reverse-complemented copies of its own pieces under a random mask, so that windows
    whose two strands both have the k-mer come in long runs.  Byte for byte the reference's output, in one batch and in many."""
    if name == "both_strands":
        rng = random.Random(77)
        base = "".join(rng.choice("ACGT") for _ in range(30000))
        pieces = [base]
        for _ in range(12):
            p = rng.randrange(0, len(base) - 3000)
            pieces.append(revcomp(base[p:p + rng.choice([500, 1500, 3000])]))
        text = "".join(pieces)
        ms = "".join(ch if rng.random() < 0.6 else ch.lower() for ch in text)
        ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
        prefix = str(tmp_path / "both")
        oracle_py.Oracle.construct(ms, k, use_klcp=False).dump(prefix)
    else:
        src = GOLDEN / ("mixed.fa" if name == "mixed_k11" else "ref_data/GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13")
        ms = "".join(l.strip() for l in open(src) if not l.startswith(">"))
        prefix = index_prefix(name)
    rng = random.Random(5)
    q = tmp_path / "q.fa"
    with open(q, "w") as f:
        for r in range(700):
            L = rng.choice([1500, 2500, 4000])
            p = rng.randrange(0, len(ms) - L)
            seq = ms[p:p + L]
            if r % 3 == 1:
                seq = revcomp(seq.upper())
            if r % 5 == 4:
                seq = "".join(rng.choice("ACGT") for _ in range(L))
            if r % 7 == 0:
                seq = seq[:L // 2] + "N" + seq[L // 2:]
            f.write(f">r{r}\n{seq}\n")
    args = ["lookup", "-k", str(k), "-q", str(q), prefix]
    checker = oracle_py.REF_CLI if oracle_py.ref_available() else oracle_py.ORACLE_CLI
    rc, ref, err = oracle_py.run_cli(checker, args)
    assert rc == 0, err
    if name == "both_strands":   # many windows whose two strands both have the k-mer, with different orders
        both = fm.DeviceIndex.from_files(prefix, device=0)
        kms = [ms.upper()[p:p + k] for p in range(0, 30000, 3)]
        o = both.query_kmers(fm.pack_kmers(kms, k), k, fm.ORDERS | fm.STRICT).reshape(-1, 2)
        assert int(((o[:, 0] >= 0) & (o[:, 1] >= 0) & (o[:, 0] != o[:, 1])).sum()) > 500
        both.close()
    rc, got, err = fm.run_cli(args)
    assert rc == 0, err.decode()
    assert got == ref
    rc, got, err = fm.run_cli([*args[:-1], "--batch", "50000", "--threads", "5", args[-1]])
    assert rc == 0 and got == ref
    assert sum(len(l.split(b"\t")[1].split(b",")) for l in ref.splitlines()) > 1_500_000


# (FMSI_TEST_EXTENDED=1: all six functions on every query file; by default `0-0` and `1-1000` -- the ranges' two ends, whose code is `1-1` / `2-3`'s -- take each data set's first file)
@pytest.mark.parametrize("name,kind,fn", [c for c in FN_CASES if EXTENDED or c[2] not in ("0-0", "1-1000") or c[1] == CLI_DATASETS[c[0]][1][0]])
def test_cli_general_functions_equal_reference_golden(fm, name, kind, fn):
    """`fmsi query -f and|xor|r-s` (f-masked superstrings: query_mode::general, src/fms_index.h:316-327) reproduces the real
    reference's stdout byte for byte -- k = 3 .. 64, masks of every kind, repeats (totals above 1), the reference's own golden
    for `-f xor` among them -- also with -S (which changes nothing in this mode), in small batches and on two handles."""
    k, _ = CLI_DATASETS[name]
    want = golden_fn_out(name, kind, fn)
    base = ["query", "-f", fn, "-k", str(k), "-q", str(query_file(name, kind))]
    rc, out, err = fm.run_cli([*base, index_prefix(name)])
    assert rc == 0, err.decode()
    assert out == want
    if fn in ("xor", "2-3"):
        for extra in (["-S"], ["--batch", "1500", "--devices", "0,0"], ["--tables", "full"]):
            rc, out, err = fm.run_cli([*base, *extra, index_prefix(name)])
            assert rc == 0, err.decode()
            assert out == want, extra


def test_cli_function_names_and_errors(fm):
    """mask_function() (src/functions.h:23-57) and ms_query's handling of -f (src/main.cpp:255-263, 273-280, 297-300): unknown names,
    -O after -f, lookup with a function, `or` / `all` as names of the fast paths."""
    q, prefix = str(query_file("integration_b", "queries")), index_prefix("integration_b")
    for name in ("nand", "5", "1-", "-1-2", "1-x", "a-3", "1--2", ""):
        rc, out, err = fm.run_cli(["query", "-f", name, "-q", q, prefix])
        assert rc == 1 and out == b"" and f"ERROR: Function '{name}' not recognized.".encode() in err and b"Usage:   fmsi query" in err, name
    rc, out, err = fm.run_cli(["query", "-f", "xor", "-O", "-q", q, prefix])
    assert rc == 0 and out == golden_fn_out("integration_b", "queries", "xor") and b"WARNING: Parameter -O is ignored when parameter -f is specified." in err
    rc, out, err = fm.run_cli(["query", "-f", "all", "-q", q, prefix])
    assert rc == 0 and out == golden_out("integration_b", "queries", "all")
    rc, out, err = fm.run_cli(["query", "-f", "xor", "-f", "or", "-q", q, prefix])
    assert rc == 0 and out == golden_out("integration_b", "queries", "or")
    rc, out, err = fm.run_cli(["query", "-f", "007-9", "-q", q, prefix])
    rc2, out2, _ = fm.run_cli(["query", "-f", "7-9", "-q", q, prefix])
    assert rc == 0 and rc2 == 0 and out == out2
    rc, out, err = fm.run_cli(["lookup", "-f", "xor", "-q", q, prefix])
    assert rc == 1 and out == b"" and b"not allowed with f-masked superstrings" in err and b"Usage:   fmsi lookup" in err
    rc, out, err = fm.run_cli(["lookup", "-f", "all", "-q", q, prefix])
    assert rc == 0 and out == golden_out("integration_b", "queries", "lookup") and b"WARNING: The current version of FMSI has speed benefits" in err
    # a bound std::stoi rejects kills the reference (uncaught std::out_of_range): the same here
    rc, out, err = fm.run_cli(["query", "-f", "1-99999999999", "-q", q, prefix])
    assert rc in (134, -6) and out == b""


@pytest.mark.parametrize("k", [4, 12, 31, 32, 33, 40])
def test_general_function_counts_equal_oracle(fm, oracle_built, tmp_path, k):
    """fmsi_hip_query_runs_fn against the oracle's single_query_general() of both strands, window by window: a text with repeats
    (totals up to dozens), reverse-complemented copies (both strands occur) and -- for even k -- self-complementary k-mers, which
    count once; every function, with and without the prefix table, byte and bit-packed results, k on both sides of 32."""
    rng = random.Random(100 + k)
    unit = "".join(rng.choice("ACGT") for _ in range(300))
    pal = "".join(rng.choice("ACGT") for _ in range(k // 2))
    pal = pal + revcomp(pal)                                   # self-complementary when k is even
    pieces = []
    for i in range(60):
        c = rng.random()
        if c < 0.3:
            pieces.append(unit[rng.randrange(0, 100):][: rng.randrange(k, 200)])
        elif c < 0.45:
            pieces.append(revcomp(unit[rng.randrange(0, 100):][: rng.randrange(k, 200)]))
        elif c < 0.6:
            pieces.append(pal)
        else:
            pieces.append("".join(rng.choice("ACGT") for _ in range(rng.randrange(k, 400))))
    text = "".join(pieces)
    ms = "".join(ch if rng.random() < 0.55 else ch.lower() for ch in text)
    ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    prefix = str(tmp_path / "fn")
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    orc.dump(prefix)
    # reads: pieces of the text, of its reverse complement, random ones, the palindrome
    reads = []
    for i in range(120):
        L = rng.randrange(k, 3 * k + 60)
        p = rng.randrange(0, len(text) - L)
        r = text[p:p + L] if i % 4 else "".join(rng.choice("ACGT") for _ in range(L))
        reads.append(revcomp(r) if i % 3 == 1 else r)
    reads.append(pal + pal)
    # ... in mixed case: the reference compares CHARACTERS with their case-preserving reverse complement (src/kmers.h:53-68), so a
    # self-complementary k-mer typed in asymmetric case is counted twice
    reads = [r if i % 2 else "".join(c.lower() if rng.random() < 0.3 else c for c in r) for i, r in enumerate(reads)]
    if k % 2 == 0:
        reads.append(pal[: k // 2].lower() + pal[k // 2:])
        reads.append(pal.lower())
    seq = "".join(reads)
    starts, wins, off = [], [], 0
    for r in reads:
        starts.append(off)
        wins.append(len(r) - k + 1)
        off += len(r)
    kmers = [r[i:i + k] for r in reads for i in range(len(r) - k + 1)]
    counts = [orc.kmer_general(km.encode(), k) for km in kmers]
    assert max(t for _, t in counts) >= 8 and any(o not in (0, t) for o, t in counts)
    if k % 2 == 0:
        assert orc.kmer_general(pal.encode(), k)[1] >= 1
    packed = fm.pack_sequence(seq.encode())
    lower = fm.lower_case_bits(seq.encode())
    if k % 2 == 0:   # the asymmetric palindrome counts twice, the symmetric ones once
        o1, t1 = orc.kmer_general(pal.encode(), k)
        assert orc.kmer_general((pal[: k // 2].lower() + pal[k // 2:]).encode(), k) == (2 * o1, 2 * t1) and orc.kmer_general(pal.lower().encode(), k) == (o1, t1)
    for t in (0, min(k, 6)):
        dev = fm.DeviceIndex.from_files(prefix, device=0, prefix_len=t)
        for fn, r, s in ((fm.FN_AND, 0, 0), (fm.FN_XOR, 0, 0), (fm.FN_RANGE, 0, 0), (fm.FN_RANGE, 1, 1), (fm.FN_RANGE, 2, 5), (fm.FN_RANGE, 3, 2)):
            want = np.array([orc.apply_function(fn, r, s, o, tt) for o, tt in counts], dtype=np.uint8)
            got, cn = dev.query_runs_fn(packed, len(seq), starts, wins, k, fn, r, s, counters=True, lower_case=lower)
            assert np.array_equal(got, want), (t, fn, r, s)
            assert int(cn[6]) == len(kmers) and int(cn[3]) == 2 * len(kmers) and (int(cn[4]) == 2 * len(kmers) if t else int(cn[4]) == 0)
            bits = dev.query_runs_fn(packed, len(seq), starts, wins, k, fn, r, s, flags=fm.OUT_BITS, lower_case=lower)
            assert np.array_equal(fm.unpack_bits(bits, len(kmers), 0), want)
        assert 0 < int(want.sum()) or True
        dev.close()


@pytest.mark.parametrize("shape", ["fasta_messy", "fastq_4line", "fasta_long_records", "fasta_one_giant_record", "then_plain_gzip", "bad_crc", "not_bgzf_gzip"])
def test_block_gzip_reader(fm, oracle_built, tmp_path, monkeypatch, shape):
    """Block-gzip (BGZF) query files: the members are found by their headers, inflated by several threads and parsed in stretches that
    end where a record starts -- the output must be what the ordinary reader (zlib's gzread, as in the reference) makes of the same
    file, for records of every shape, records far longer than a stretch, and files that stop being BGZF half way (an ordinary gzip
    member appended, a member whose CRC does not match): there the ordinary reader takes over at the byte the parallel part reached.
    A file that is ONE ordinary gzip member is read through zlib's handle in stretches that the team parses."""
    import gzip
    import json as _json
    name, k = "sars_k13", 13
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13") if not l.startswith(">"))
    rng = random.Random(hash(shape) & 0xFFFF)
    if shape == "fastq_4line":
        up = ms.upper()
        data = bytearray()
        for r in range(1500):
            L = rng.choice([13, 40, 75, 151, 300])
            p = rng.randrange(0, len(up) - L)
            qual = "".join(rng.choice("@+>IIIIFFF#5") for _ in range(L))
            data += f"@r{r} c\n{up[p:p + L]}\n+\n{qual}\n".encode()
        data = bytes(data)
    elif shape == "fasta_one_giant_record":      # hundreds of stretches long: the stretches double until it fits
        up = ms.upper()
        giant = "".join(up[rng.randrange(0, 20000):][:5000] for _ in range(120))
        data = (f">small0\n{up[100:400]}\n>giant\n{giant[:300000]}N{giant[300000:]}\n>small1\n{up[500:900]}\n").encode()
    elif shape == "fasta_long_records":
        up = ms.upper()
        data = b"".join(f">long{r}\n".encode() + (up[rng.randrange(0, 1000):][:rng.choice([9000, 20000])] + "N" + up[:5000]).encode() + b"\n" for r in range(12))
    else:
        data = b">first\nACGTTGCA\n" + messy_fasta(ms, rng)     # (messy_fasta may open with an '@' header: that would make it a FASTQ file)
    blob = bgzf_bytes(data, block=rng.choice([700, 3000, 0xFF00]))
    expect_parallel = len(data)
    if shape == "then_plain_gzip":
        extra = messy_fasta(ms, random.Random(9))
        blob = bgzf_bytes(data, block=1500, eof=False) + gzip.compress(extra)
        data, expect_parallel = data + extra, None
    elif shape == "bad_crc":
        blob = bytearray(bgzf_bytes(data, block=1500))
        at, members = 0, []
        while at < len(blob):
            bsize = int.from_bytes(blob[at + 16:at + 18], "little") + 1
            members.append((at, bsize))
            at += bsize
        at, bsize = members[len(members) // 2]
        good = tmp_path / "good.fa.gz"
        good.write_bytes(bytes(blob))
        blob[at + bsize - 8] ^= 0x5A                     # the CRC of a member in the middle
        blob, expect_parallel = bytes(blob), None
    elif shape == "not_bgzf_gzip":
        blob, expect_parallel = gzip.compress(data), "stream"   # (one ordinary gzip member: stretches of the stream parsed by the team since round 5)
    q = tmp_path / "q.fa.gz"
    q.write_bytes(blob)
    for mode in ("or", "all_S", "lookup"):
        args = [*MODES[mode][0], "-k", str(k), "--batch", "4096", "-q", str(q), index_prefix(name)]
        monkeypatch.setenv("FMSI_HIP_SERIAL_READER", "1")
        rc0, want, err0 = fm.run_cli(args)
        monkeypatch.delenv("FMSI_HIP_SERIAL_READER")
        monkeypatch.setenv("FMSI_HIP_READER_MIN_BYTES", "1")
        rc, got, err = fm.run_cli([*args[:-1], "--stats", args[-1]])
        monkeypatch.delenv("FMSI_HIP_READER_MIN_BYTES")
        assert rc == rc0 == 0, err.decode()
        if shape == "bad_crc":
            # zlib's gzread drops what it had inflated in the call that meets the error, so how much comes out in front of a corrupt
            # member depends on buffer sizes (the reference's too): whole lines of the intact file's output, up to short of the damage
            rc, whole, _ = fm.run_cli([*args[:-2], str(good), args[-1]])
            assert rc == 0 and whole.startswith(got) and (got == b"" or got.endswith(b"\n")) and 0 < len(got) < len(whole) and whole.startswith(want)
        else:
            assert got == want
        par = _json.loads([l for l in err.decode().splitlines() if l.startswith("{")][-1])["block_parallel_read_bytes"]
        if expect_parallel == "stream":
            assert 0 < par <= len(data)
        elif expect_parallel is not None:
            assert par == expect_parallel
        else:
            assert 0 < par < len(data)
        if shape != "bad_crc":       # (zlib's error handling on a corrupt stream is the serial reader's business; the reference's is the same zlib)
            checker = oracle_py.REF_CLI if oracle_py.ref_available() else oracle_py.ORACLE_CLI
            rc, ref, _ = oracle_py.run_cli(checker, [a for a in args if a not in ("--batch", "4096")])
            assert rc == 0 and got == ref
        assert got.count(b"\n") >= (3 if shape == "fasta_one_giant_record" else 12)


DEGENERATE_INPUTS = {
    "empty": b"", "newline": b"\n", "gt_only": b">", "name_only": b">name", "name_nl": b">name\n", "junk": b"hello world\nno header here\n",
    "junk_then_rec": b"garbage\n>r\nACGTACGTACGTACGTACGT\n", "nul_bytes": b">r\nACGT\x00ACGTACGTACGTACGTAC\n", "only_at": b"@",
    "fastq_no_qual": b"@r\nACGTACGTACGTACGTACGT\n+\n", "cr_only": b">r\rACGTACGTACGTACGTACGT\r", "tabs": b">r\tc\nACGTACGTACGTACGTACGT\n",
    "long_name": b">" + b"n" * 100000 + b"\nACGTACGTACGTACGTACGT\n", "high_bytes": b">r\n\xff\xfeACGTACGTACGTACGTACGT\n",
    "two_gt": b">>r\nACGTACGTACGTACGTACGT\n", "space_name": b"> r\nACGTACGTACGTACGTACGT\n",
}


@pytest.mark.parametrize("case", list(DEGENERATE_INPUTS))
def test_degenerate_query_files(fm, oracle_built, tmp_path, monkeypatch, case):
    """Query files that are hardly files: empty, a lone '>', a header without a sequence, no header at all, NUL and high bytes, CR line
    ends only, a FASTQ record without its quality, a 100-kB name -- plain and block-gzip, through the ordinary and the block-parallel
    readers, and over stdin: exit status and bytes of the real reference."""
    import subprocess
    prefix = index_prefix("sars_k13")
    checker = oracle_py.REF_CLI if oracle_py.ref_available() else oracle_py.ORACLE_CLI
    data = DEGENERATE_INPUTS[case]
    for pack in ("plain", "bgzf"):
        q = tmp_path / f"q_{pack}"
        q.write_bytes(data if pack == "plain" else bgzf_bytes(data))
        for mode in (["query", "-O"], ["lookup", "-S"]):
            rc0, ref, _ = oracle_py.run_cli(checker, [*mode, "-q", str(q), prefix])
            for min_bytes in (None, "1"):
                if min_bytes:
                    monkeypatch.setenv("FMSI_HIP_READER_MIN_BYTES", min_bytes)
                rc, got, err = fm.run_cli([*mode, "-q", str(q), prefix])
                monkeypatch.delenv("FMSI_HIP_READER_MIN_BYTES", raising=False)
                assert rc == rc0 and got == ref, (pack, mode, min_bytes, err.decode()[-300:])
    ref = subprocess.run([str(checker), "query", prefix], input=data, capture_output=True)
    got = subprocess.run([str(fm.CLI_PATH), "query", prefix], input=data, capture_output=True)
    assert got.returncode == ref.returncode and got.stdout == ref.stdout


def test_cli_exits_when_stdout_fails(fm, tmp_path):
    """A writer that fails (ENOSPC: stdout = /dev/full) must end the process with exit status 1 -- not leave the reader thread
    blocked on a full queue and the CLI hanging in join().  Many small batches, so that input is still pending when it happens."""
    import subprocess
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_test2_ones.fna.ms.k31") if not l.startswith(">")).upper()
    q = tmp_path / "many.fa"
    with open(q, "w") as f:
        for r in range(4000):
            p = (r * 37) % (len(ms) - 400)
            f.write(f">r{r}\n{ms[p:p + 400]}\n")
    with open("/dev/full", "wb") as full:
        res = subprocess.run([str(fm.CLI_PATH), "query", "-q", str(q), "--batch", "1024", index_prefix("sars_k31")], stdout=full,
                             stderr=subprocess.PIPE, timeout=120)
    assert res.returncode == 1 and b"write to stdout failed" in res.stderr
    # and the same input to a healthy stdout still works
    rc, out, _ = fm.run_cli(["query", "-q", str(q), "--batch", "1024", index_prefix("sars_k31")])
    assert rc == 0 and out.count(b"\n") == 4000


@pytest.mark.parametrize("mode", ["or", "all", "all_S"])
def test_cli_output_to_a_regular_file(fm, tmp_path, mode):
    """stdout = a regular file instead of a pipe: the bytes a pipe receives -- behind whatever was in the file before the offset (a
    shell's `{ echo; fmsi query; } > out`), leaving alone what lies behind the range when the file is already longer (`1<> out`),
    with the shared offset behind the output afterwards; and appended to a file opened for appending."""
    import os
    import subprocess
    name, k = "mixed_k11", 11
    ms = "".join(l.strip() for l in open(GOLDEN / "mixed.fa") if not l.startswith(">"))
    rng = random.Random(11)
    q = tmp_path / "q.fa"
    q.write_bytes(b"".join(messy_fasta(ms, rng) for _ in range(8)))
    args = [str(fm.CLI_PATH), *MODES[mode][0], "-k", str(k), "-q", str(q), "--batch", "4096", index_prefix(name)]
    want = subprocess.run(args, capture_output=True, timeout=300)
    assert want.returncode == 0 and len(want.stdout) > 10000, want.stderr.decode()
    # (1) a fresh write-only file with a header line before the output and a trailer after it, through the one shared offset
    out = tmp_path / "out1.txt"
    with open(out, "wb") as f:
        os.write(f.fileno(), b"header line\n")
        res = subprocess.run(args, stdout=f, stderr=subprocess.PIPE, timeout=300)
        assert res.returncode == 0, res.stderr.decode()
        os.write(f.fileno(), b"trailer\n")
    assert out.read_bytes() == b"header line\n" + want.stdout + b"trailer\n"
    # (2) a longer file opened read-write at offset 5: the output overwrites its range, the rest stays
    out = tmp_path / "out2.txt"
    filler = b"x" * (len(want.stdout) + 1000)
    out.write_bytes(filler)
    fd = os.open(out, os.O_RDWR)
    os.lseek(fd, 5, os.SEEK_SET)
    res = subprocess.run(args, stdout=fd, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0, res.stderr.decode()
    assert os.lseek(fd, 0, os.SEEK_CUR) == 5 + len(want.stdout)
    os.close(fd)
    assert out.read_bytes() == filler[:5] + want.stdout + filler[5 + len(want.stdout):]
    # (3) append mode
    out = tmp_path / "out3.txt"
    out.write_bytes(b"before\n")
    with open(out, "ab") as f:
        res = subprocess.run(args, stdout=f, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0 and out.read_bytes() == b"before\n" + want.stdout


@pytest.mark.parametrize("mode", ["or", "all", "all_S", "lookup"])
def test_cli_with_several_device_handles(fm, tmp_path, mode):
    """--devices a,b,...: one replicated handle per device, batches dealt out to worker threads, output in input order and the
    strand predictor replayed across the batches as if there were one device.  The GPU box has one GPU, so the list names it twice
    (two handles, two workers -- the same code path); many small batches, so that batches really finish out of order."""
    name, k = "mixed_k11", 11      # non-max-ones mask: -O output depends on the predictor's state across batches
    ms = "".join(l.strip() for l in open(GOLDEN / "mixed.fa") if not l.startswith(">"))
    rng = random.Random(3)
    q = tmp_path / "q.fa"
    q.write_bytes(b"".join(messy_fasta(ms, rng) for _ in range(12)))
    args = [*MODES[mode][0], "-k", str(k), "-q", str(q)]
    rc1, out1, err1 = fm.run_cli([*args, "--batch", "2048", index_prefix(name)])
    assert rc1 == 0, err1.decode()
    for devices in ("0,0", "0,0,0"):
        rc, out, err = fm.run_cli([*args, "--batch", "2048", "--devices", devices, "--stats", index_prefix(name)])
        assert rc == 0, err.decode()
        assert out == out1
        assert f'"devices": {len(devices.split(","))}'.encode() in err
    # and the reference itself says the same (the oracle's record loop over the same file)
    orc = oracle_py.Oracle.load(index_prefix(name), use_klcp=MODES[mode][2])
    want, _ = orc.query_file(str(q), k, has_klcp=MODES[mode][2], mode=MODES[mode][3], orders=MODES[mode][1], cap=len(out1) + 4096)
    assert out1.decode() == want
    rc, out, err = fm.run_cli([*args, "--devices", "0,x", index_prefix(name)])
    assert rc == 1 and out == b""


@pytest.mark.parametrize("mode", ["or", "all", "or_S", "all_S"])
def test_cli_background_dictionary_switch_over(fm, tmp_path, mode):
    """--tables background (what `--tables auto` does from 8 GB of queries on): the first batches are answered by the LF walk / the kLCP
    kernel while a second handle builds the k-mer dictionary in a thread of its own; every batch from the moment it stands goes to the
    dictionary.  The bytes do not depend on where the switch falls -- many small batches, two handles, a non-max-ones mask whose -O output
    depends on the predictor's state across batches -- and equal the one-path runs and the oracle's record loop; --stats says how many
    batches the dictionary answered."""
    import json as _json
    name, k = "mixed_k11", 11
    ms = "".join(l.strip() for l in open(GOLDEN / "mixed.fa") if not l.startswith(">"))
    rng = random.Random(17)
    q = tmp_path / "q.fa"
    q.write_bytes(b"".join(messy_fasta(ms, rng) for _ in range(60)))
    args = [*MODES[mode][0], "-k", str(k), "-q", str(q)]
    rc0, out0, err0 = fm.run_cli([*args, "--batch", "2048", index_prefix(name)])
    assert rc0 == 0, err0.decode()
    switched = 0
    for extra in (["--tables", "background"], ["--tables", "background", "--devices", "0,0"], ["--tables", "background", "--threads", "3"]):
        rc, out, err = fm.run_cli([*args, "--batch", "2048", "--stats", *extra, index_prefix(name)])
        assert rc == 0, err.decode()
        assert out == out0, extra
        stats = _json.loads([l for l in err.decode().splitlines() if l.startswith("{")][-1])
        assert stats["dictionary"] == "beside the queries"
        switched += stats["dictionary_batches"]
    assert switched > 0   # (a 12-kb index: the dictionary stands within the first batches of at least one of the runs)
    orc = oracle_py.Oracle.load(index_prefix(name), use_klcp=MODES[mode][2])
    want, _ = orc.query_file(str(q), k, has_klcp=MODES[mode][2], mode=MODES[mode][3], orders=MODES[mode][1], cap=len(out0) + 4096)
    assert out0.decode() == want


def test_sorted_walk_in_sub_batches(fm, oracle_built, monkeypatch):
    """FMSI_HIP_KERNEL_SORTED cuts a call into sub-batches (item values are 32 bits, one radix sort takes < 2^32 items: 2^28 k-mers each);
    forced down to 1000 and to 7 k-mers per sub-batch here: the same bytes as one batch, as the oracle, bytes and bit-packed, odd and even
    k, with the cut between the two launches at every depth the environment can ask for."""
    import subprocess
    import sys
    code = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import fmsi_amd as fm, oracle_py
from conftest import synthetic_ms, mixed_kmers
for k, n in ((31, 20000), (32, 5000), (12, 3000)):
    ms = synthetic_ms(n, k, seed=k, mask="mixed")
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    dev = fm.DeviceIndex.build(ms.encode(), k).enable_pair_layout()
    kmers = mixed_kmers(ms, k, 4001, seed=3)
    packed = fm.pack_kmers(kmers, k)
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk("".join(kmers).encode(), len(kmers), k, mode_all)
        gf, gr = fm.decode_strict(dev.query_kmers(packed, k, flag | fm.STRICT | fm.KERNEL_SORTED))
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc), (k, mode_all)
        fast = dev.query_kmers(packed, k, flag | fm.KERNEL_SORTED)
        want = (np.where(fwd != -1, fwd, rc) == 1) if mode_all else ((fwd == 1) | (rc == 1))
        assert np.array_equal(fast.astype(bool), want)
        bits = dev.query_kmers(packed, k, flag | fm.KERNEL_SORTED | fm.OUT_BITS)
        assert np.array_equal(fm.unpack_bits(bits, len(kmers), 0), fast)
print("ok")
""" % (str(REPO), str(REPO / "oracle"), str(REPO / "tests"))
    import os
    for env in ({"FMSI_HIP_SORTED_SUB": "1000"}, {"FMSI_HIP_SORTED_SUB": "7"}, {"FMSI_HIP_SORTED_CUT": "0"}, {"FMSI_HIP_SORTED_CUT": "5"}, {"FMSI_HIP_SORTED_CUT": "24"},
                {"FMSI_HIP_SORTED_NO_TOP": "1"}, {"FMSI_HIP_SORTED_BASES": "3"}):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert res.returncode == 0 and res.stdout.strip().endswith("ok"), (env, res.stderr[-1500:])
