"""GPU tests of the on-device index builder (fmsi_hip_index_build*, counterpart of construct<T>(),
reference src/fms_index.h:397-460): the layout it leaves in HBM must equal, dword for dword, the layout
the host builds from the oracle's (== the reference's) index files."""
import random

import numpy as np
import pytest

import oracle_py
from conftest import GOLDEN, mixed_kmers, synthetic_ms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(product_built):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device visible")
    return product_built


def host_layout_from_oracle(fm, tmp_path, ms: str, k: int, tag: str):
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=True)
    prefix = str(tmp_path / tag)
    orc.dump(prefix)
    host = fm.HostIndex(prefix, use_klcp=True)
    return orc, host, prefix


def check_equal_layout(fm, dev, host):
    want_blocks, want_mrank = host.layout()
    got_blocks, got_mrank = dev.download_layout()
    assert got_blocks.shape == want_blocks.shape
    assert np.array_equal(got_mrank, want_mrank)
    bad = np.nonzero((got_blocks != want_blocks).any(axis=1))[0]
    assert bad.size == 0, f"first differing block {bad[:5]}"
    assert dev.info.n == host.n


CASES = [
    ("tiny", "CaGGTag", 3),
    ("repeat", "ACGT" * 300 + "acg", 4),                       # every 32-mer key tied: host tie-break path
    ("polyA", "A" * 500 + "C" + "A" * 300 + "ca", 3),
    ("n127", None, 5), ("n128", None, 5), ("n255", None, 7), ("n256", None, 7), ("n4096", None, 9),
]


@pytest.mark.parametrize("tag,ms,k", CASES)
def test_build_from_text_equals_oracle_layout(fm, oracle_built, tmp_path, tag, ms, k):
    if ms is None:
        ms = synthetic_ms(int(tag[1:]) - 1, k, seed=len(tag) * 7 + k, mask="mixed")
    orc, host, prefix = host_layout_from_oracle(fm, tmp_path, ms, k, tag)
    dev = fm.DeviceIndex.build(ms.encode(), k, with_klcp=True)
    check_equal_layout(fm, dev, host)
    # and it answers like the oracle
    kmers = mixed_kmers(ms, k, 500, seed=1) if len(ms) > 2 * k else ["ACG", "CAG", "GGT", "TAG", "AAA"]
    fwd, rc = orc.ternaries_bulk("".join(kmers).encode(), len(kmers), k, False)
    gf, gr = fm.decode_strict(dev.query_kmers(fm.pack_kmers(kmers, k), k, fm.MODE_OR | fm.STRICT))
    assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)


def test_build_mixed_fixture_equals_reference_index(fm):
    """Device-built layout of the mixed-mask fixture == layout from the REAL reference's index files."""
    ms = "".join(l.strip() for l in open(GOLDEN / "mixed.fa") if not l.startswith(">"))
    host = fm.HostIndex(str(GOLDEN / "idx" / "mixed_k11"), use_klcp=True)
    dev = fm.DeviceIndex.build(ms.encode(), 11, with_klcp=True)
    check_equal_layout(fm, dev, host)
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_test2_ones.fna.ms.k31") if not l.startswith(">"))
    host = fm.HostIndex(str(GOLDEN / "idx" / "sars_k31"), use_klcp=True)
    dev = fm.DeviceIndex.build(ms.encode(), 31, with_klcp=True)
    check_equal_layout(fm, dev, host)


def test_synthetic_build_equals_oracle(fm, oracle_built, tmp_path):
    n, seed, k = 300_000, 12345, 31
    dev = fm.DeviceIndex.build_synthetic(n, seed, k, with_klcp=True, prefix_len=8)
    text = fm.synthetic_text(n, seed, k)
    assert np.array_equal(dev.download_text(n), fm.synthetic_text_words(n, seed))
    orc, host, prefix = host_layout_from_oracle(fm, tmp_path, text.decode(), k, "syn")
    check_equal_layout(fm, dev, host)
    # device-generated queries: ~half positive, bit-exact vs the oracle on the same packed k-mers
    import torch
    nq = 100_000
    d_kmers = torch.empty(nq, dtype=torch.int64, device="cuda")
    d_out = torch.empty(nq, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    dev.synthetic_queries_dev(k, 42, 0, nq, d_kmers.data_ptr(), s)
    dev.query_kmers_dev(d_kmers.data_ptr(), nq, k, fm.MODE_ALL | fm.STRICT, d_out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    kmers = d_kmers.cpu().numpy().view(np.uint64)
    got_f, got_r = fm.decode_strict(d_out.cpu().numpy())
    codes = ((kmers[:, None] >> (2 * np.arange(k, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8)
    raw = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()
    fwd, rc = orc.ternaries_bulk(raw, nq, k, True)
    assert np.array_equal(got_f, fwd) and np.array_equal(got_r, rc)
    pos = ((fwd == 1) | (rc == 1)).mean()
    assert 0.45 < pos < 0.55
    # the stream of queries is a pure function of (seed, index): a second draw of a sub-range agrees
    d2 = torch.empty(1000, dtype=torch.int64, device="cuda")
    dev.synthetic_queries_dev(k, 42, 5000, 1000, d2.data_ptr(), s)
    torch.cuda.synchronize()
    assert np.array_equal(d2.cpu().numpy().view(np.uint64), kmers[5000:6000])


def klcp_bits(words, nbits):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:nbits]


@pytest.mark.parametrize("n,k,mask", [(5_000, 7, "mixed"), (100_000, 9, "mixed"), (300_000, 9, "mixed"), (300_000, 13, "mixed"),
                                      (700_001, 12, "ones"), (200_000, 40, "mixed"), (200_000, 64, "ones")])
def test_klcp_equals_oracle(fm, oracle_built, n, k, mask):
    """The device-built kLCP array (klcp_kernel, construct_klcp src/fms_index.h:357-385) bit for bit against the oracle, on
    texts large enough to put many workgroups on every CU (dense at k = 9: ~half of the bits set) -- and the same bits again
    on a second build (an earlier fused version of the kernel was not run-to-run stable beyond 2^16 rows)."""
    ms = synthetic_ms(n, k, seed=n % 1000 + k, mask=mask)
    if k >= 33:      # plant repeats so that long (k-1)-mers are shared at all
        rng = random.Random(k)
        parts = [ms[:5000]]
        while sum(map(len, parts)) < n:
            a = rng.randrange(0, 4900)
            parts.append(ms[a:a + rng.randrange(30, 100)].upper() if rng.random() < 0.7 else
                         "".join(rng.choice("ACGT") for _ in range(rng.randrange(30, 200))))
        ms = "".join(parts)
        ms = ms[:len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=True)
    words, nbits = orc.words(3)
    want = klcp_bits(words, nbits)
    assert want.sum() > 100
    for rep in range(2):
        dev = fm.DeviceIndex.build(ms.encode(), k, with_klcp=True)
        got = klcp_bits(dev.download_klcp(), nbits)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"rep {rep}: {bad.size} kLCP bits differ, first rows {bad[:8]}"


def test_build_is_deterministic_at_scale(fm):
    """Two builds of a 5 M-base synthetic superstring leave identical bytes in HBM (layout, mask ranks, kLCP)."""
    a = fm.DeviceIndex.build_synthetic(5_000_000, 3, 23, with_klcp=True)
    b = fm.DeviceIndex.build_synthetic(5_000_000, 3, 23, with_klcp=True)
    (ba, ma), (bb, mb) = a.download_layout(), b.download_layout()
    assert np.array_equal(ba, bb) and np.array_equal(ma, mb)
    assert np.array_equal(a.download_klcp(), b.download_klcp())


def repetitive_cases():
    rng = random.Random(99)
    unit = "".join(rng.choice("ACGT") for _ in range(5000))
    fib_a, fib_b = "A", "AC"
    while len(fib_b) < 150_000:
        fib_a, fib_b = fib_b, fib_b + fib_a
    mutated = lambda s: "".join(rng.choice("ACGT") if rng.random() < 0.001 else ch for ch in s)
    return {
        "period4": ("ACGT" * 60_000, 5),                                   # every suffix tied until its end: log2(n/29) rounds
        "polyA": ("A" * 100_000 + "C" + "A" * 70_000, 4),
        "fibonacci": (fib_b[:150_000], 6),
        "tandem_5k": (unit * 30 + "".join(rng.choice("ACGT") for _ in range(20_000)) + unit[:2500] * 8, 11),
        "diverged_copies": ("".join(mutated(unit * 6) for _ in range(5)), 13),     # long repeats broken by rare substitutions
        "short_then_long": ("".join(rng.choice("ACGT") for _ in range(200_000)) + unit[:700] * 2, 31),
    }


@pytest.mark.parametrize("tag", list(repetitive_cases()))
def test_build_repetitive_text_equals_reference(fm, oracle_built, tmp_path, tag):
    """Repeats far longer than the 29 bases of the first sort key: the tied rows are ordered by prefix doubling on the
    device.  Checked dword for dword against the layout of the index the REAL reference (Larsson-Sadakane qsufsort)
    writes for the same superstring -- or the oracle's comparison sort where oracle/_ref did not travel."""
    ms, k = repetitive_cases()[tag]
    rng = random.Random(len(ms))
    ms = "".join(ch if rng.random() < 0.7 else ch.lower() for ch in ms)
    ms = ms[:len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    fa = tmp_path / tag
    fa.write_text(f">{tag}\n{ms}\n")
    if oracle_py.REF_CLI.exists():
        rc, _, err = oracle_py.run_cli(oracle_py.REF_CLI, ["index", "-k", str(k), str(fa)])
    else:
        rc, _, err = oracle_py.run_cli(oracle_py.ORACLE_CLI, ["index", "-k", str(k), str(fa)])
    assert rc == 0, err
    host = fm.HostIndex(str(fa), use_klcp=True)
    dev = fm.DeviceIndex.build(ms.encode(), k, with_klcp=True)
    check_equal_layout(fm, dev, host)
    assert np.array_equal(klcp_bits(dev.download_klcp(), host.n), host.bits("klcp"))
    # and the product's own `fmsi index` writes the same files
    fb = tmp_path / (tag + "_gpu")
    fb.write_text(f">{tag}\n{ms}\n")
    rc, _, err = fm.run_cli(["index", "-k", str(k), str(fb)])
    assert rc == 0, err
    for part in ("ac_gt", "ac", "gt", "mask", "klcp", "misc"):
        assert (tmp_path / f"{tag}_gpu.fmsi.{part}").read_bytes() == (tmp_path / f"{tag}.fmsi.{part}").read_bytes(), part


def test_synthetic_repeat_families_equal_oracle(fm, oracle_built, tmp_path):
    """fmsi_hip_index_build_synthetic_repeats (bench.py --repeat-permille): a text with repeat families -- prefix-table
    keys with tens of occurrences -- is indexed like any other (layout == the oracle's from the same text), the host
    restatement of a word agrees with the device's, and every query path answers it like the oracle, candidate tables
    included (their entries overflow here, so the stepping kernel gets real work)."""
    import torch
    n, seed, k = 400_000, 777, 25
    rep = (400, 5, 80)   # 40 % of the segments from 5 families at 8 % divergence: ~16 copies per family
    dev = fm.DeviceIndex.build_synthetic(n, seed, k, with_klcp=True, prefix_len=8, repeats=rep)
    words = dev.download_text(n)
    L = fm.lib()
    for w in (0, 63, 64, 1000, 5000, n // 32 - 1):
        assert int(words[w]) == int(L.fmsi_hip_synthetic_text_word_repeats(seed, w, *rep))
    plain = fm.synthetic_text_words(n, seed)
    assert 0.25 < float((words[:n // 32] != plain[:n // 32]).mean()) < 0.55          # the repeat segments differ from the uniform text
    codes = ((words[:, None] >> (2 * np.arange(32, dtype=np.uint64))) & np.uint64(3)).reshape(-1)[:n].astype(np.uint8)
    up = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    lo = np.frombuffer(b"acgt", dtype=np.uint8)[codes]
    text = np.where(np.arange(n) + k <= n, up, lo).tobytes().decode()
    orc, host, prefix = host_layout_from_oracle(fm, tmp_path, text, k, "rep")
    check_equal_layout(fm, dev, host)
    nq = 60_000
    d_kmers = torch.empty(nq, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    dev.synthetic_queries_dev(k, 42, 0, nq, d_kmers.data_ptr(), s)
    torch.cuda.synchronize()
    kmers = d_kmers.cpu().numpy().view(np.uint64)
    kc = ((kmers[:, None] >> (2 * np.arange(k, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8)
    raw = np.frombuffer(b"ACGT", dtype=np.uint8)[kc].tobytes()
    fwd, rc = orc.ternaries_bulk(raw, nq, k, True)
    # k - t = 15: the one-lane lookup kernel with 32-byte entries (4^10 keys for 4 x 10^5 rows: uniform keys fit their
    # entry, the keys of the families -- about 16 copies x 0.92^10 -- overflow it)
    dev_c = fm.DeviceIndex.build_synthetic(n, seed, k, prefix_len=10, repeats=rep).enable_pair_layout().enable_verify()
    assert int(dev_c.info.candidate_entry_bytes) == 32
    for d in (dev, dev_c):
        got, cn = d.query_kmers(kmers, k, fm.MODE_ALL | fm.STRICT, counters=True)
        gf, gr = fm.decode_strict(got)
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
    assert int(cn[0]) > 0.02 * nq          # LF steps were needed: the candidate entries did not hold every key


@pytest.mark.parametrize("name", ["integration_a", "integration_b", "sars_k13", "sars_k31", "mixed_k11", "repeats_k40", "repeats_k64"])
def test_device_load_equals_host_load(fm, name):
    """fmsi_hip_index_load (RRR-63 decode + block interleaving on the device, from the files' raw payloads) gives dword for dword
    the layout fmsi_host_index_load + fmsi_hip_index_create build on the host -- blocks, mask ranks, kLCP, scalars -- on every
    golden index (incl. inverted RRR superblocks in mixed_k11 and n % 63 == 0 cases) and answers the same."""
    from conftest import index_prefix
    prefix = index_prefix(name) if name in ("integration_a", "integration_b", "sars_k13", "sars_k31", "mixed_k11") else str(GOLDEN / "idx" / name)
    a = fm.DeviceIndex.from_files(prefix, use_klcp=True, prefix_len=4)
    b = fm.DeviceIndex.load(prefix, use_klcp=True, prefix_len=4)
    ba, ma = a.download_layout()
    bb, mb = b.download_layout()
    assert np.array_equal(ba, bb) and np.array_equal(ma, mb)
    ia, ib = a.info, b.info
    assert (ia.n, ia.k, ia.has_klcp, ia.prefix_len, list(ia.counts), ia.dollar_position) == (ib.n, ib.k, ib.has_klcp, ib.prefix_len, list(ib.counts), ib.dollar_position)
    assert np.array_equal(a.download_klcp()[: (ia.n + 63) // 64], b.download_klcp()[: (ia.n + 63) // 64])
    nok = fm.DeviceIndex.load(prefix, use_klcp=False)
    assert nok.info.has_klcp == 0
    with pytest.raises(fm.FmsiError) as e:
        fm.DeviceIndex.load(prefix + "_missing")
    assert e.value.code == fm.E_IO


def test_load_with_the_table_signed_beside_the_queries(fm, tmp_path):
    """FMSI_HIP_LOAD_ASYNC_TABLE (what the `fmsi` binary asks for): the handle comes back when the prefix table's intervals stand and the
    signatures of its entries of 1..5 rows are written on a stream of the library's own while the caller already queries -- an entry is
    rewritten by one aligned 8-byte store and both of its forms give the same interval.  10^8 rows, a 13-deep table (1.5 rows per key:
    most entries get signatures): isolated k-mers, reads through the kLCP kernel (its certificates read the entries) and `lookup` orders
    sent the moment the load returns, again after wait_background(), and through a handle loaded the ordinary way -- the same bytes; then
    release_scratch() gives the staging buffers back and the handle answers as before."""
    import torch
    n, k = 100_000_000, 31
    built = fm.DeviceIndex.build_synthetic(n, 4242, k, with_klcp=True, mask_permille=700)
    prefix = str(tmp_path / "idx")
    built.write_files(prefix)
    s = torch.cuda.current_stream().cuda_stream
    nq, L, n_reads = 1 << 21, 3000, 512
    d_kmers = torch.empty(nq, dtype=torch.int64, device="cuda")
    built.synthetic_queries_dev(k, 11, 0, nq, d_kmers.data_ptr(), s)
    d_reads = torch.empty((n_reads * L) // 32 + 2, dtype=torch.int64, device="cuda")
    built.synthetic_reads_dev(L, 5, 0, n_reads, d_reads.data_ptr(), 10000, s)
    torch.cuda.synchronize()
    packed = d_reads.cpu().numpy().view(np.uint64)
    kmers = d_kmers.cpu().numpy().view(np.uint64)
    starts = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    wins = np.full(n_reads, L - k + 1, dtype=np.uint32)
    built.close()

    def answers(dev):
        return (dev.query_kmers(kmers, k, fm.MODE_ALL | fm.STRICT), dev.query_stream(packed, n_reads * L, starts, wins, k, fm.MODE_OR | fm.STRICT),
                dev.query_kmers(kmers[:200_000], k, fm.ORDERS | fm.STRICT))

    quick = fm.DeviceIndex.load(prefix, use_klcp=True, prefix_len=13, async_table=True)
    first = answers(quick)            # (the signature pass is most likely still running)
    quick.wait_background()
    second = answers(quick)
    plain = fm.DeviceIndex.load(prefix, use_klcp=True, prefix_len=13)
    want = answers(plain)
    for a, b, c in zip(first, second, want):
        assert np.array_equal(a, c) and np.array_equal(b, c)
    f, r = fm.decode_strict(want[0])
    assert 0.4 < ((f == 1) | (r == 1) | (f == 0) | (r == 0)).mean() < 0.6
    before = int(quick.info.device_bytes)
    quick.release_scratch()
    assert int(quick.info.device_bytes) == before
    assert np.array_equal(quick.query_kmers(kmers[:50_000], k, fm.MODE_ALL | fm.STRICT), want[0][:50_000])
    quick.close()
    plain.close()


def test_device_load_at_scale_and_corrupt_files(fm, oracle_built, tmp_path):
    """1.2 M rows with a mixed mask (many superblocks, threaded host decode vs the device's), every row count n % 63 / n % 64 /
    n % 128 boundary in a sweep of small sizes, and corrupt .mask images are FMSI_HIP_EIO on this path too."""
    import struct
    from conftest import synthetic_ms
    for n, seed in [(1_200_000, 1)] + [(m, m) for m in (62, 63, 64, 125, 126, 127, 128, 129, 189, 190, 4031, 4032, 4033, 8063, 8064, 8065)]:
        k = 5 if n < 1000 else 31
        ms = synthetic_ms(n, k, seed=seed, mask="mixed")
        prefix = str(tmp_path / f"i{n}")
        oracle_py.Oracle.construct(ms, k, use_klcp=True).dump(prefix)
        a = fm.DeviceIndex.from_files(prefix, use_klcp=True)
        b = fm.DeviceIndex.load(prefix, use_klcp=True)
        ba, ma = a.download_layout()
        bb, mb = b.download_layout()
        assert np.array_equal(ba, bb) and np.array_equal(ma, mb), n
        assert np.array_equal(a.download_klcp()[: (n + 64) // 64], b.download_klcp()[: (n + 64) // 64]), n
    good = open(prefix + ".fmsi.mask", "rb").read()
    n_, bt_bits = struct.unpack_from("<QQ", good, 0)
    off_btnr = 8 + 9 + ((bt_bits + 63) // 64) * 8
    btnr_bits = struct.unpack_from("<Q", good, off_btnr)[0]
    off_btnrp = off_btnr + 8 + ((btnr_bits + 63) // 64) * 8
    bad = bytearray(good)
    bad[off_btnrp + 9: off_btnrp + 17] = b"\xff" * 8
    for image in (bytes(bad), good[: len(good) // 2]):
        open(prefix + ".fmsi.mask", "wb").write(image)
        with pytest.raises(fm.FmsiError) as e:
            fm.DeviceIndex.load(prefix)
        assert e.value.code == fm.E_IO


def test_device_load_rejects_parts_of_different_indexes(fm, oracle_built, tmp_path):
    """Header-consistent but MISMATCHED index parts -- .ac / .gt of another index of the same length, a .klcp of another length, a .misc
    whose `$` row or C[] lie outside the rows or belong to another text -- are FMSI_HIP_EIO ("index not correctly loaded") on the
    device-load path, decided BEFORE any kernel indexes gt[] / ac[] by ranks taken from ac_gt (the layout kernel used to run first and
    read past the device allocations), and the process goes on to load and query a good index afterwards."""
    import shutil
    from conftest import synthetic_ms
    n, k = 300_000, 21
    pa, pb, pc = (str(tmp_path / t) for t in "abc")
    oracle_py.Oracle.construct(synthetic_ms(n, k, seed=11, mask="mixed"), k, use_klcp=True).dump(pa)
    # another text of the same length with a skewed composition: popcount(ac_gt) differs by thousands
    rng = np.random.default_rng(3)
    skew = "".join(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=n, p=[0.4, 0.3, 0.2, 0.1])].tobytes().decode())
    skew = skew[: n - (k - 1)] + skew[n - (k - 1):].lower()
    oracle_py.Oracle.construct(skew, k, use_klcp=True).dump(pb)
    oracle_py.Oracle.construct(synthetic_ms(n // 2, k, seed=12, mask="mixed"), k, use_klcp=True).dump(pc)
    parts = ("ac_gt", "ac", "gt", "mask", "klcp", "misc")

    def mixed(tag, **take):
        dst = str(tmp_path / tag)
        for part in parts:
            shutil.copy(f"{take.get(part, pa)}.fmsi.{part}", f"{dst}.fmsi.{part}")
        return dst

    cases = {
        "gt_ac_of_b": mixed("m1", ac=pb, gt=pb),          # |ac| + |gt| = n still holds
        "acgt_of_b": mixed("m2", ac_gt=pb),
        "klcp_of_c": mixed("m3", klcp=pc),
        "misc_of_b": mixed("m4", misc=pb),
    }
    d = mixed("m5")
    lines = open(d + ".fmsi.misc").read().split()
    open(d + ".fmsi.misc", "w").write("\n".join([str(n + 5)] + lines[1:]) + "\n")   # `$` row outside the rows
    cases["dollar_outside"] = d
    d = mixed("m6")
    open(d + ".fmsi.misc", "w").write("\n".join([lines[0], lines[1], lines[3], lines[2], lines[4], lines[5]]) + "\n")   # C[] not monotone
    cases["counts_not_monotone"] = d
    for tag, prefix in cases.items():
        with pytest.raises(fm.FmsiError) as e:
            fm.DeviceIndex.load(prefix, use_klcp=True)
        assert e.value.code == fm.E_IO, tag
        with pytest.raises(fm.FmsiError):   # the host path refuses them too (build_layout's checks)
            fm.DeviceIndex.from_files(prefix, use_klcp=True)
    good = fm.DeviceIndex.load(pa, use_klcp=True, prefix_len=6)
    ms = synthetic_ms(n, k, seed=11, mask="mixed").upper()
    kmers = [ms[p:p + k] for p in range(0, 4000, 7)]
    got = good.query_kmers(fm.pack_kmers(kmers, k), k, fm.MODE_OR)
    orc = oracle_py.Oracle.load(pa)
    wf, wr = orc.ternaries_bulk("".join(kmers).encode(), len(kmers), k, False)
    assert np.array_equal(got, (np.maximum(wf, wr) == 1).astype(np.uint8))


def test_block_cache_rules(fm):
    """devpool.h: by default a builder returns the cache of large freed blocks to the driver when it ends (nothing the handle's info does
    not count stays on the device); inside builder_chain the blocks stay for the next builder and go back when the chain ends."""
    n, k = 300_000_000, 31   # (temporaries of 0.3-2.4 GB: well above the cache's 64 MB threshold)
    assert fm.scratch_bytes(0) == 0
    dev = fm.DeviceIndex.build_synthetic(n, 5, k)
    assert fm.scratch_bytes(0) == 0
    dev.enable_pair_layout()
    assert fm.scratch_bytes(0) == 0
    del dev
    with fm.builder_chain(0):
        dev = fm.DeviceIndex.build_synthetic(n, 5, k)
        kept = fm.scratch_bytes(0)
        assert kept > (1 << 30)
        dev.enable_pair_layout().enable_dict()
        assert fm.scratch_bytes(0) > 0
    assert fm.scratch_bytes(0) == 0
    import torch
    d_k = torch.empty(1 << 20, dtype=torch.int64, device="cuda")
    d_o = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    dev.synthetic_queries_dev(k, 42, 0, 1 << 20, d_k.data_ptr(), s)
    dev.query_kmers_dev(d_k.data_ptr(), 1 << 20, k, fm.MODE_ALL | fm.KERNEL_DICT, d_o.data_ptr(), stream=s)
    a = d_o.cpu().numpy().copy()
    dev.query_kmers_dev(d_k.data_ptr(), 1 << 20, k, fm.MODE_ALL | fm.KERNEL_PAIR | fm.NO_PREFIX_TABLE, d_o.data_ptr(), stream=s)
    torch.cuda.synchronize()
    assert np.array_equal(a, d_o.cpu().numpy()) and 0.45 < a.mean() < 0.55
