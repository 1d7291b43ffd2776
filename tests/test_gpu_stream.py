"""GPU tests of the kLCP streaming kernel (fmsi_hip_query_stream, counterpart of query_kmers_streaming<>(),
reference src/fms_index.h:181-254): per-window results must equal fresh searches of both strands (oracle), for
any cut of the sequences into runs, incl. indexes with long kLCP runs (repeats) that exercise the multi-word scan."""
import random

import numpy as np
import pytest

import oracle_py
from conftest import (DATASETS, GOLDEN, index_prefix, mixed_kmers, revcomp, stream_expected_fast, stream_expected_orders,
                      synthetic_ms)

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


def run_case(fm, orc, dev, reads, k, check_orders=True):
    """reads: list of ACGT strings (len >= k).  Packs them back to back, one run per read."""
    seq = "".join(reads)
    words = fm.pack_sequence(seq.encode())
    starts, wins, kmers = [], [], []
    pos = 0
    for r in reads:
        starts.append(pos)
        wins.append(len(r) - k + 1)
        kmers += [r[i:i + k] for i in range(len(r) - k + 1)]
        pos += len(r)
    raw = "".join(kmers).encode()
    n = len(kmers)
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk(raw, n, k, mode_all)
        gf, gr = fm.decode_strict(dev.query_stream(words, len(seq), starts, wins, k, flag | fm.STRICT))
        assert np.array_equal(gf, fwd), (mode_all, np.nonzero(gf != fwd)[0][:10])
        assert np.array_equal(gr, rc), (mode_all, np.nonzero(gr != rc)[0][:10])
        fast = dev.query_stream(words, len(seq), starts, wins, k, flag)
        assert np.array_equal(fast.astype(bool), stream_expected_fast(fwd, rc, wins, mode_all))
        # same answers without the prefix table and (forward strand first everywhere) from the non-streaming kernel
        assert np.array_equal(dev.query_stream(words, len(seq), starts, wins, k, flag | fm.NO_PREFIX_TABLE), fast)
        # ... and from the first streaming kernel (64-byte blocks + separate kLCP words)
        assert np.array_equal(dev.query_stream(words, len(seq), starts, wins, k, flag | fm.KERNEL_REFILL), fast)
        v1f, v1r = fm.decode_strict(dev.query_stream(words, len(seq), starts, wins, k, flag | fm.STRICT | fm.KERNEL_REFILL | fm.NO_PREFIX_TABLE))
        assert np.array_equal(v1f, fwd) and np.array_equal(v1r, rc)
        packed = dev.query_stream(words, len(seq), starts, wins, k, flag | fm.OUT_BITS)
        assert np.array_equal(fm.unpack_bits(packed, n, 0), fast)
        assert np.array_equal(dev.query_kmers(fm.pack_kmers(kmers, k), k, flag).astype(bool), expected_fast(fwd, rc, mode_all))
    if check_orders:
        fwd, rc, of, orr = orc.ternaries_bulk(raw, n, k, False, orders=True)
        got = dev.query_stream(words, len(seq), starts, wins, k, fm.ORDERS | fm.STRICT)
        assert np.array_equal(got[:, 0], of) and np.array_equal(got[:, 1], orr)
        got1 = dev.query_stream(words, len(seq), starts, wins, k, fm.ORDERS)
        assert np.array_equal(got1, stream_expected_orders(fwd, rc, of, orr, wins))
        # (since round 4 the S-block kernel's ORD instantiation; the first kernel and the table-less start give the same orders)
        for extra in (fm.KERNEL_REFILL, fm.NO_PREFIX_TABLE):
            assert np.array_equal(dev.query_stream(words, len(seq), starts, wins, k, fm.ORDERS | fm.STRICT | extra), got)
            assert np.array_equal(dev.query_stream(words, len(seq), starts, wins, k, fm.ORDERS | extra), got1)


def ms_of(name):
    f = {"integration_a": "ref_data/integration_a.fa", "integration_b": "ref_data/integration_b.fa",
         "sars_k13": "ref_data/GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13", "sars_k31": "ref_data/GCF_test2_ones.fna.ms.k31",
         "mixed_k11": "mixed.fa"}[name]
    return "".join(l.strip() for l in open(GOLDEN / f) if not l.startswith(">"))


@pytest.mark.parametrize("name", ["sars_k13", "sars_k31", "mixed_k11", "integration_a"])
def test_stream_equals_oracle(fm, oracle_built, name):
    k = DATASETS[name][0]
    orc = oracle_py.Oracle.load(index_prefix(name))
    dev = fm.DeviceIndex.from_files(index_prefix(name), use_klcp=True, prefix_len=min(k, 5))
    ms = ms_of(name).upper()
    rng = random.Random(11)
    reads = []
    if len(ms) > 500:
        for _ in range(60):
            L = rng.choice([k, k + 1, k + 5, 63 + k, 64 + k, 65 + k, 150, 400])
            p = rng.randrange(0, len(ms) - L)
            r = ms[p:p + L]
            if rng.random() < 0.5:
                r = revcomp(r)
            if rng.random() < 0.5:   # a few substitutions: mixes hit and miss stretches
                r = "".join(rng.choice("ACGT") if rng.random() < 0.02 else ch for ch in r)
            reads.append(r)
        reads += ["".join(rng.choice("ACGT") for _ in range(200)) for _ in range(5)]
        reads.append(ms[100:100 + 3000])              # a long run: many tasks
        reads.append(revcomp(ms[4000:4000 + 2500]))
    else:
        reads = ["ACGTACGTACG", "TACGTAC", "ACG", "GGGGGGG", "CGTACGTA" * 20]
    run_case(fm, orc, dev, reads, k)


@pytest.mark.parametrize("length", [31, 95, 96, 400])
def test_stream_runs_of_one_length(fm, oracle_built, length):
    """Reads of ONE length back to back (and one read alone): the library takes these without a run table (three numbers in the
    kernel's arguments) -- answers must be those of the same reads handed over with a table (reads of two lengths)."""
    name = "sars_k31"
    k = DATASETS[name][0]
    orc = oracle_py.Oracle.load(index_prefix(name))
    dev = fm.DeviceIndex.from_files(index_prefix(name), use_klcp=True, prefix_len=5)
    ms = ms_of(name).upper()
    rng = random.Random(length)
    reads = []
    for i in range(70):
        p = rng.randrange(0, len(ms) - length)
        r = ms[p:p + length]
        if i % 3 == 1:
            r = revcomp(r)
        if i % 4 == 2:
            r = "".join(rng.choice("ACGT") if rng.random() < 0.03 else ch for ch in r)
        reads.append(r)
    run_case(fm, orc, dev, reads, k)
    run_case(fm, orc, dev, reads[:1], k, check_orders=False)
    # equal runs at a stride with gaps (windows over the junctions left out), against the same call with one run made ragged
    seq = "".join(reads)
    words = fm.pack_sequence(seq.encode())
    wpr = length - k + 1
    starts = [i * length for i in range(len(reads))]
    a = dev.query_stream(words, len(seq), starts, [wpr] * len(reads), k, fm.MODE_ALL)
    if wpr > 1:
        b = dev.query_stream(words, len(seq), starts, [wpr] * (len(reads) - 1) + [wpr - 1], k, fm.MODE_ALL)
        assert np.array_equal(a[:-1], b)
    # runs that overlap: each run is answered as a read of its own
    half = dev.query_stream(words, len(seq), [0, length // 2 + 1], [wpr, wpr], k, fm.MODE_ALL)
    one = dev.query_stream(words, len(seq), [length // 2 + 1], [wpr], k, fm.MODE_ALL)
    assert np.array_equal(half[wpr:], one) and np.array_equal(half[:wpr], a[:wpr])


def test_stream_on_repetitive_index(fm, oracle_built, tmp_path):
    """Long runs of set kLCP bits (many rows sharing a (k-1)-mer): the extension crosses 64-bit words and blocks."""
    k = 9
    rng = random.Random(3)
    unit = "".join(rng.choice("ACGT") for _ in range(40))
    ms = unit * 150 + "".join(rng.choice("ACGT") for _ in range(3000)) + "A" * 700 + unit[:20] * 30
    ms = "".join(ch if rng.random() < 0.7 else ch.lower() for ch in ms)
    ms = ms[: len(ms) - (k - 1)].rstrip() + ms[len(ms) - (k - 1):].lower()
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=True)
    prefix = str(tmp_path / "rep")
    orc.dump(prefix)
    dev = fm.DeviceIndex.from_files(prefix, use_klcp=True, prefix_len=4)
    up = ms.upper()
    reads = [up[0:400], up[5990:6300], revcomp(up[100:500]), "A" * 100, up[9000:9400], revcomp(up[9100:9700]), unit * 3,
             "".join(rng.choice("ACGT") for _ in range(300))]
    run_case(fm, orc, dev, reads, k)


@pytest.mark.parametrize("k,t", [(7, 4), (8, 4), (9, 4), (10, 4), (12, 6), (31, 9), (32, 8)])
def test_stream_absence_certificates(fm, oracle_built, tmp_path, k, t):
    """The certificate lookups of the S-block kernel (a (t + 4)-mer that the text does not hold settles k - t - 3 windows at once): k below
    t + 4 (none possible), k = t + 4 (a certificate settles its own window only), k just above, table keys with MANY rows (a short table
    on a repetitive text: entries without signatures cannot certify), reads that leave and rejoin the text at substitutions a few bases
    apart, at the first and last bases of tasks, reads of random bases, reads shorter than a certificate's reach -- strand by strand
    against the oracle, both modes, orders, and against the first kernel."""
    rng = random.Random(1000 * k + t)
    body = "".join(rng.choice("ACGT") for _ in range(6000))
    fam = "".join(rng.choice("ACGT") for _ in range(300))
    ms = body[:2000] + fam + body[2000:4000] + "".join(rng.choice("ACGT") if rng.random() < 0.03 else ch for ch in fam) + body[4000:] + "AC" * 200 + "T" * 150
    ms = "".join(ch if rng.random() < 0.7 else ch.lower() for ch in ms)
    ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=True)
    prefix = str(tmp_path / "c")
    orc.dump(prefix)
    dev = fm.DeviceIndex.load(prefix, use_klcp=True, prefix_len=t)
    up = ms.upper()
    reads = []
    for L in (k, k + 1, 64 + k - 1, 64 + k, 200, 700):
        for _ in range(6):
            p = rng.randrange(0, len(up) - L + 1)
            r = up[p:p + L]
            if rng.random() < 0.5:
                r = revcomp(r)
            reads.append(r)
            sub = list(r)
            for x in sorted({0, L - 1, L // 2, rng.randrange(L), min(L - 1, L // 2 + rng.randrange(1, 6)), rng.randrange(L)}):
                if rng.random() < 0.7:
                    sub[x] = rng.choice([c for c in "ACGT" if c != sub[x]])
            reads.append("".join(sub))
        reads.append("".join(rng.choice("ACGT") for _ in range(L)))
    reads.append(up[:900])
    reads.append(revcomp(up[-900:]))
    reads.append("ACGT" * 60 + "T" * 100)
    run_case(fm, orc, dev, reads, k)
    one = [r[:150 + k - 1].ljust(150 + k - 1, "A") for r in reads if len(r) >= 150]     # reads of one length: tasks cut inside the kernel
    run_case(fm, orc, dev, one, k)


def test_stream_argument_errors(fm):
    dev_no = fm.DeviceIndex.from_files(index_prefix("sars_k31"), use_klcp=False)
    words = fm.pack_sequence(b"ACGT" * 20)
    with pytest.raises(fm.FmsiError):
        dev_no.query_stream(words, 80, [0], [50], 31)            # no kLCP in the handle
    dev = fm.DeviceIndex.from_files(index_prefix("sars_k31"), use_klcp=True)
    with pytest.raises(fm.FmsiError):
        dev.query_stream(words, 80, [0], [51], 31)               # run past the end
    assert dev.query_stream(words, 80, [], [], 31).size == 0


def test_stream_counters_positive_read(fm, oracle_built):
    """On a read taken from the superstring a streamed window costs one LF step: the kernel's step counter must
    be close to one per window (plus one fresh search per 64-window task and strand)."""
    name, k = "sars_k31", 31
    dev = fm.DeviceIndex.from_files(index_prefix(name), use_klcp=True)
    ms = ms_of(name).upper()
    read = ms[1000:1000 + 6400 + k - 1]
    words = fm.pack_sequence(read.encode())
    out, cn = dev.query_stream(words, len(read), [0], [6400], k, fm.MODE_ALL, counters=True)
    assert (out == 1).all()
    assert 6400 <= int(cn[0]) <= 6400 + 100 * (k + 1)
    out2, cn2 = dev.query_windows(words, len(read), k, fm.MODE_ALL, counters=True)
    assert np.array_equal(out, out2) and int(cn2[0]) == 6400 * k


@pytest.mark.parametrize("k", [5, 12])
def test_stream_layout_boundaries(fm, oracle_built, tmp_path, k):
    """The streaming layout's blocks hold 192 rows: row counts around its multiples (and around the 64- and 128-row marks inside a block),
    mixed masks, repeats (intervals of many rows that straddle blocks, kLCP runs across them), reads in both orientations with
    substitutions -- against the oracle strand by strand, in both modes, with and without the prefix table."""
    rng = random.Random(100 + k)
    for n in (63, 64, 65, 127, 128, 129, 190, 191, 192, 193, 255, 256, 383, 384, 385, 575, 576, 577, 767, 768, 769, 1151, 1152, 1153, 4000):
        body = "".join(rng.choice("ACGT") for _ in range(max(8, n // 3)))
        ms = (body * 4)[:n] if n > 40 else body[:n]
        ms = "".join(rng.choice("ACGT") if rng.random() < 0.05 else ch for ch in ms)
        ms = "".join(ch if rng.random() < 0.6 else ch.lower() for ch in ms)
        ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
        orc = oracle_py.Oracle.construct(ms, k, use_klcp=True)
        prefix = str(tmp_path / f"b{n}")
        orc.dump(prefix)
        dev = fm.DeviceIndex.load(prefix, use_klcp=True, prefix_len=3)
        up = ms.upper()
        reads = [up, revcomp(up)]
        for _ in range(6):
            L = rng.randrange(k, min(len(up), 150) + 1)
            p = rng.randrange(0, len(up) - L + 1)
            r = up[p:p + L]
            if rng.random() < 0.5:
                r = revcomp(r)
            if rng.random() < 0.5:
                r = "".join(rng.choice("ACGT") if rng.random() < 0.03 else ch for ch in r)
            reads.append(r)
        run_case(fm, orc, dev, reads, k, check_orders=False)


def test_stream_orders_into_an_output_that_is_not_16_byte_aligned(fm, oracle_built):
    """`lookup -S` on the device stores a window's two strand orders as ONE 16-byte store where it can; an output array whose pairs are
    not 16-byte aligned (a device pointer 8 bytes into an allocation) takes the 8-byte stores, and the same orders arrive."""
    import torch
    name = "sars_k31"
    k = DATASETS[name][0]
    dev = fm.DeviceIndex.from_files(index_prefix(name), use_klcp=True, prefix_len=5)
    ms = ms_of(name).upper()
    rng = random.Random(5)
    reads = []
    for i in range(40):
        L = rng.choice([k, 64 + k, 150, 400])
        p = rng.randrange(0, len(ms) - L)
        r = ms[p:p + L]
        if i % 2:
            r = revcomp(r)
        if i % 3 == 0:
            r = "".join(rng.choice("ACGT") if rng.random() < 0.03 else ch for ch in r)
        reads.append(r)
    reads += ["".join(rng.choice("ACGT") for _ in range(300)) for _ in range(3)]
    seq = "".join(reads)
    words = fm.pack_sequence(seq.encode())
    starts, wins, pos = [], [], 0
    for r in reads:
        starts.append(pos)
        wins.append(len(r) - k + 1)
        pos += len(r)
    n = sum(wins)
    want = dev.query_stream(words, len(seq), starts, wins, k, fm.ORDERS | fm.STRICT)
    d_seq = torch.from_numpy(words.view(np.int64).copy()).cuda()
    s = torch.cuda.current_stream().cuda_stream
    for shift in (0, 1):   # int64 elements into a 16-byte aligned allocation
        out = torch.full((2 * n + 4,), 7, dtype=torch.int64, device="cuda")
        assert out.data_ptr() % 16 == 0
        dev.query_stream_dev(d_seq.data_ptr(), len(seq), np.asarray(starts, dtype=np.uint64), np.asarray(wins, dtype=np.uint32), k, fm.ORDERS | fm.STRICT,
                             out.data_ptr() + 8 * shift, 0, s)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got[shift:shift + 2 * n].reshape(n, 2), want), shift
        assert np.all(got[:shift] == 7) and np.all(got[shift + 2 * n:] == 7)   # (nothing written outside the array)
