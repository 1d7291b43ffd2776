"""GPU tests for k > 32 (the reference walks query characters, so it is k-agnostic; its kLCP array exists up to k = 64):
fmsi_hip_query_windows (fmsi_longk.hip) and fmsi_hip_query_stream (WIDE instantiation) against the oracle, per window
and per strand; the CLI goldens for k = 40 / 64 are in test_gpu_parity.py::test_cli_equals_reference_golden."""
import random

import numpy as np
import pytest

import oracle_py
from conftest import LONGK_DATASETS, index_prefix, ms_of, revcomp, stream_expected_fast, stream_expected_orders, synthetic_ms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(product_built):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device visible")
    return product_built


def expected_fast(fwd, rc, mode_all):
    if mode_all:
        return np.where(fwd != -1, fwd, rc) == 1
    return (fwd == 1) | (rc == 1)


def make_reads(ms, k, seed, n_reads=50):
    rng = random.Random(seed)
    up = ms.upper()
    reads = []
    for _ in range(n_reads):
        L = rng.choice([k, k + 1, k + 7, 63 + k, 64 + k, 65 + k, 2 * k + 30, 500])
        if rng.random() < 0.8:
            p = rng.randrange(0, len(up) - L)
            r = up[p:p + L]
            if rng.random() < 0.5:
                r = revcomp(r)
            if rng.random() < 0.5:
                r = "".join(rng.choice("ACGT") if rng.random() < 0.01 else ch for ch in r)
        else:
            r = "".join(rng.choice("ACGT") for _ in range(L))
        reads.append(r)
    reads.append(up[50:50 + 2000])
    reads.append(revcomp(up[3000:3000 + 1500]))
    return reads


def check(fm, orc, dev, reads, k, stream):
    seq = "".join(reads)
    words = fm.pack_sequence(seq.encode())
    starts, wins, kmers, wstart = [], [], [], []
    pos = 0
    for r in reads:
        starts.append(pos)
        wins.append(len(r) - k + 1)
        kmers += [r[i:i + k] for i in range(len(r) - k + 1)]
        wstart += list(range(pos, pos + len(r) - k + 1))
        pos += len(r)
    raw = "".join(kmers).encode()
    n = len(kmers)
    wstart = np.asarray(wstart, dtype=np.uint64)

    def run(flags, counters=False):
        if stream:
            return dev.query_stream(words, len(seq), starts, wins, k, flags, counters=counters)
        return dev.query_windows(words, len(seq), k, flags, window_start=wstart, counters=counters)

    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk(raw, n, k, mode_all)
        gf, gr = fm.decode_strict(run(flag | fm.STRICT))
        assert np.array_equal(gf, fwd), (mode_all, np.nonzero(gf != fwd)[0][:10])
        assert np.array_equal(gr, rc), (mode_all, np.nonzero(gr != rc)[0][:10])
        fast = run(flag)
        want_fast = stream_expected_fast(fwd, rc, wins, mode_all) if stream else expected_fast(fwd, rc, mode_all)
        assert np.array_equal(fast.astype(bool), want_fast)
        assert np.array_equal(run(flag | fm.NO_PREFIX_TABLE), fast)
        if stream:   # the S-block kernel's WIDE instantiation (round 4) against the first kernel's, and its bit-packed results
            assert np.array_equal(run(flag | fm.KERNEL_REFILL), fast)
            assert np.array_equal(run(flag | fm.STRICT | fm.KERNEL_REFILL), run(flag | fm.STRICT))
            assert np.array_equal(fm.unpack_bits(run(flag | fm.OUT_BITS), n, 0), fast)
        assert 0.05 < fast.mean() < 0.95
    fwd, rc, of, orr = orc.ternaries_bulk(raw, n, k, False, orders=True)
    got = run(fm.ORDERS | fm.STRICT)
    assert np.array_equal(got[:, 0], of) and np.array_equal(got[:, 1], orr)
    assert np.array_equal(run(fm.ORDERS), stream_expected_orders(fwd, rc, of, orr, wins) if stream else np.where(of >= 0, of, orr))
    if stream:
        assert np.array_equal(run(fm.ORDERS | fm.STRICT | fm.KERNEL_REFILL), got) and np.array_equal(run(fm.ORDERS | fm.KERNEL_REFILL), run(fm.ORDERS))
    if not stream:   # LF-step accounting of the both-strands schedule equals the oracle's
        orc.counters_reset()
        orc.ternaries_bulk(raw, n, k, True)
        want = orc.counters()
        _, cn = run(fm.MODE_ALL | fm.STRICT, counters=True)
        assert int(cn[0]) == want["lf_steps"] and int(cn[3]) == want["strand_searches"] and int(cn[6]) == n


@pytest.mark.parametrize("name", list(LONGK_DATASETS))
@pytest.mark.parametrize("stream", [False, True])
def test_longk_golden_indexes_equal_oracle(fm, oracle_built, name, stream):
    k = LONGK_DATASETS[name][0]
    orc = oracle_py.Oracle.load(index_prefix(name), use_klcp=True)
    dev = fm.DeviceIndex.from_files(index_prefix(name), use_klcp=True, prefix_len=5)
    check(fm, orc, dev, make_reads(ms_of(name), k, seed=k), k, stream)


@pytest.mark.parametrize("k,stream", [(33, False), (33, True), (48, True), (64, True), (64, False), (65, False), (100, False), (200, False)])
def test_longk_synthetic_equals_oracle(fm, oracle_built, tmp_path, k, stream):
    ms = synthetic_ms(150_000, k, seed=k, mask="mixed")
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=k <= 64)
    dev = fm.DeviceIndex.build(ms.encode(), k, with_klcp=k <= 64, prefix_len=7)
    check(fm, orc, dev, make_reads(ms, k, seed=3 * k, n_reads=40), k, stream)


@pytest.mark.parametrize("k,length", [(33, 96), (48, 150), (64, 64), (64, 191), (40, 400)])
def test_longk_stream_runs_of_one_length(fm, oracle_built, k, length):
    """Reads of ONE length (no run table: the S-block kernel cuts its tasks itself) for 33 <= k <= 64: results and orders against the oracle."""
    ms = synthetic_ms(60_000, k, seed=100 + k, mask="mixed")
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=True)
    dev = fm.DeviceIndex.build(ms.encode(), k, with_klcp=True, prefix_len=6)
    up = ms.upper()
    rng = random.Random(k * length)
    reads = []
    for i in range(90):
        if i % 5 == 4:
            r = "".join(rng.choice("ACGT") for _ in range(length))
        else:
            p = rng.randrange(0, len(up) - length)
            r = up[p:p + length]
            if i % 3 == 1:
                r = revcomp(r)
            if i % 4 == 2:
                r = "".join(rng.choice("ACGT") if rng.random() < 0.01 else ch for ch in r)
        reads.append(r)
    check(fm, orc, dev, reads, k, True)


def test_longk_entry_point_limits(fm):
    ms = synthetic_ms(5000, 40, seed=1)
    dev = fm.DeviceIndex.build(ms.encode(), 40, with_klcp=False)
    with pytest.raises(fm.FmsiError):       # one u64 per k-mer cannot hold 40 bases
        dev.query_kmers(np.zeros(4, dtype=np.uint64), 40)
    words = fm.pack_sequence(ms.upper().encode())
    with pytest.raises(fm.FmsiError):       # no kLCP array on this handle
        dev.query_stream(words, len(ms), [0], [len(ms) - 39], 40, fm.MODE_OR)
    dev65 = fm.DeviceIndex.build(synthetic_ms(5000, 65, seed=2).encode(), 65, with_klcp=False)
    assert dev65.info.k == 65
