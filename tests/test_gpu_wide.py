"""64-bit rows (fmsi_wide.hip): the reference's intervals are size_t (src/fms_index.h:98-103), and an index of 2^32 rows or more
loads into the same blocks with a table of absolute counts per superblock.  No such index can be built here, so
FMSI_HIP_FORCE_ROWS64 sends the golden indexes -- and a 1.2 M-row one, 1170 superblocks of 8 blocks -- through that
instantiation: every mode against the oracle, the CLI against the reference's golden output, and the entry points that do not
exist for such handles refuse with FMSI_HIP_ERANGE instead of answering wrongly."""
import os
import random

import numpy as np
import pytest

import oracle_py
from conftest import (CLI_DATASETS, DATASETS, FN_CASES, GOLDEN, MODES, golden_fn_out, golden_out, index_prefix, mixed_kmers, query_file,
                      synthetic_ms)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fm(product_built, monkeypatch):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on an MI355X")
    monkeypatch.setenv("FMSI_HIP_FORCE_ROWS64", "1")
    return product_built


def check(fm, dev, orc, kmers, k):
    assert dev.info.rows64 == 1
    raw = "".join(kmers).encode()
    packed = fm.pack_kmers(kmers, k)
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk(raw, len(kmers), k, mode_all)
        got, cn = dev.query_kmers(packed, k, flag | fm.STRICT, counters=True)
        gf, gr = fm.decode_strict(got)
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
        # the same from the search without the handle's prefix table (16-byte [i, j) entries for 64-bit rows) -- which saves steps when it is there
        plain, cn0 = dev.query_kmers(packed, k, flag | fm.STRICT | fm.NO_PREFIX_TABLE, counters=True)
        assert np.array_equal(plain, got) and int(cn0[4]) == 0
        if dev.info.prefix_len and k >= dev.info.prefix_len:
            assert int(cn[4]) == int(cn[3]) and int(cn[0]) < int(cn0[0])      # a lookup per strand search, fewer LF steps
        fast = dev.query_kmers(packed, k, flag | fm.KERNEL_PHASED)      # (kernel selection flags do not matter here)
        assert np.array_equal(fast.astype(bool), (np.where(fwd != -1, fwd, rc) == 1) if mode_all else ((fwd == 1) | (rc == 1)))
        assert np.array_equal(fm.unpack_bits(dev.query_kmers(packed, k, flag | fm.STRICT | fm.OUT_BITS), len(kmers), fm.STRICT), got)
    _, _, of, orr = orc.ternaries_bulk(raw, len(kmers), k, False, orders=True)
    got = dev.query_kmers(packed, k, fm.ORDERS | fm.STRICT)
    assert np.array_equal(got[:, 0], of) and np.array_equal(got[:, 1], orr)
    assert np.array_equal(dev.query_kmers(packed, k, fm.ORDERS), np.where(of >= 0, of, orr))


@pytest.mark.parametrize("name", [n for n in DATASETS if DATASETS[n][0] <= 32])
def test_wide_rows_equal_oracle_on_the_golden_indexes(fm, oracle_built, name):
    k = DATASETS[name][0]
    orc = oracle_py.Oracle.load(index_prefix(name))
    dev = fm.DeviceIndex.from_files(index_prefix(name), use_klcp=True, prefix_len=6)     # (the kLCP array is not taken; the table is, as deep as the index is large)
    ms_file = {"integration_a": "ref_data/integration_a.fa", "integration_b": "ref_data/integration_b.fa",
               "sars_k13": "ref_data/GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13",
               "sars_k31": "ref_data/GCF_test2_ones.fna.ms.k31", "mixed_k11": "mixed.fa"}[name]
    ms = "".join(l.strip() for l in open(GOLDEN / ms_file) if not l.startswith(">"))
    kmers = mixed_kmers(ms, k, 4000, seed=5)
    if k <= 6:
        import itertools
        kmers = ["".join(p) for p in itertools.product("ACGT", repeat=k)]
    check(fm, dev, orc, kmers, k)
    info = dev.info
    assert 1 <= info.prefix_len <= 6 and info.has_klcp == 0
    for call in (dev.enable_pair_layout, dev.enable_dict, dev.enable_verify):
        with pytest.raises(fm.FmsiError) as e:
            call()
        assert e.value.code == fm.E_RANGE
    assert dev.export() == ms.encode()        # export_ms() over 64-bit rows: the superstring the index was made of, mask as case
    # windows and runs go through the same kernel
    up = ms.upper()[:3000]
    if len(up) > k + 10:
        words = fm.pack_sequence(up.encode())
        n = len(up) - k + 1
        wk = [up[i:i + k] for i in range(n)]
        fwd, rc = orc.ternaries_bulk("".join(wk).encode(), n, k, True)
        gf, gr = fm.decode_strict(dev.query_runs(words, len(up), [0], [n], k, fm.MODE_ALL | fm.STRICT))
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
        # the streaming entry point on 64-bit rows: served by the plain search (the kLCP extension is a short cut, not another answer)
        sf, sr = fm.decode_strict(dev.query_stream(words, len(up), [0], [n], k, fm.MODE_ALL | fm.STRICT))
        assert np.array_equal(sf, fwd) and np.array_equal(sr, rc)


def test_wide_rows_across_many_superblocks(fm, oracle_built, tmp_path):
    """1.2 M rows = 9 376 blocks = 1 172 superblocks of 8: the wrapped block counts are restored from the superblock table
    everywhere; mixed mask, so ORDERS and the OR-mode interval scan use the mask counts too."""
    k = 31
    ms = synthetic_ms(1_200_000, k, seed=4242, mask="mixed")
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    orc.dump(str(tmp_path / "w"))
    dev = fm.DeviceIndex.from_files(str(tmp_path / "w"), prefix_len=9)
    assert dev.info.prefix_len == 9
    check(fm, dev, orc, mixed_kmers(ms, k, 100_000, seed=6), k)
    # and the written files are the loaded ones (the writer works from the same blocks)
    dev.write_files(str(tmp_path / "w2"))
    for part in ("ac_gt", "ac", "gt", "mask", "misc"):
        assert open(f"{tmp_path}/w.fmsi.{part}", "rb").read() == open(f"{tmp_path}/w2.fmsi.{part}", "rb").read()
    assert dev.export() == ms.encode()        # (the walk crosses superblocks at every other step)


@pytest.mark.parametrize("name,kind,mode", [(n, kind, mode) for n, (k, kinds) in CLI_DATASETS.items() for kind in kinds[:1]
                                            for mode in ("or", "all", "lookup")])
def test_cli_with_wide_rows(fm, name, kind, mode):
    """(k <= 32: query_wide_kernel; k = 40, 64: the k-agnostic walk of fmsi_longk.hip, a template over the row width)"""
    k = CLI_DATASETS[name][0]
    rc, out, err = fm.run_cli([*MODES[mode][0], "-k", str(k), "-q", str(query_file(name, kind)), index_prefix(name)])
    assert rc == 0, err.decode()
    assert out == golden_out(name, kind, mode)
    # -S on such an index: the kLCP array is the reference's short cut, not another answer -- the plain search over 64-bit rows serves it
    # (and the record loop's strand bookkeeping is replayed as for any -S run): the reference's -S output
    s_mode = {"or": "or_S", "all": "all_S", "lookup": "lookup_S"}.get(mode)
    if s_mode in MODES:
        rc, out, err = fm.run_cli([*MODES[s_mode][0], "-k", str(k), "-q", str(query_file(name, kind)), index_prefix(name)])
        assert rc == 0, err.decode()
        assert out == golden_out(name, kind, s_mode)
    if mode == "or":     # `fmsi export` walks 64-bit rows, too: the same FASTA as with 32-bit rows (tests/test_gpu_export.py holds that one against the reference)
        rc, out, err = fm.run_cli(["export", index_prefix(name)])
        assert rc == 0, err.decode()
        saved = os.environ.pop("FMSI_HIP_FORCE_ROWS64")
        try:
            rc2, out2, err2 = fm.run_cli(["export", index_prefix(name)])
        finally:   # (whatever happens in between: the later tests of this module must still run on 64-bit rows)
            os.environ["FMSI_HIP_FORCE_ROWS64"] = saved
        assert rc2 == 0 and out == out2 and out.startswith(b">exported masked superstring\n")


@pytest.mark.parametrize("shape", ["random", "mixed_mask", "small_repeats", "small_repeats_doubling", "big_repeats", "tandem", "homopolymer"])
def test_wide_builder_equals_the_narrow_one(product_built, oracle_built, monkeypatch, shape):
    """The on-device builder for superstrings of 2^32 bases and more (suffixes dealt into 16 buckets by their first two bases, a rocPRIM
    sort per bucket with 64-bit positions, tied suffixes ordered group by group) forced onto small texts (FMSI_HIP_FORCE_WIDE_BUILD): dword
    for dword the blocks and mask ranks of the 32-bit builder, the same scalars, and -- through query_wide_kernel and its superblock
    table -- the oracle's answers."""
    fm = product_built
    k = 31
    rng = np.random.default_rng(5)
    if shape.startswith("small_repeats"):
        base = synthetic_ms(60_000, k, seed=9, mask="mixed")
        ms = base + base[1000:1400] * 20 + base[5000:5100].upper() * 8 + base[:30_000]     # tied groups of up to a few dozen suffixes
        ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
        if shape.endswith("doubling"):      # ... through the prefix doubling all the same (FMSI_HIP_WIDE_DOUBLING)
            monkeypatch.setenv("FMSI_HIP_WIDE_DOUBLING", "1")
    elif shape == "big_repeats":            # tied groups of hundreds and thousands of suffixes: prefix doubling with 64-bit ranks (round 4)
        base = synthetic_ms(80_000, k, seed=11, mask="mixed")
        fam = base[2000:2600]
        copies = []
        r2 = random.Random(7)
        for _ in range(300):
            c = list(fam)
            for _ in range(3):
                c[r2.randrange(len(c))] = r2.choice("ACGTacgt")
            copies.append("".join(c))
        ms = base + "".join(copies) + base[10_000:40_000] + fam * 40
        ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    elif shape == "tandem":                 # one group of 32 000 suffixes that differ only beyond tens of thousands of bases: many rounds
        ms = ("ACGTTGCA" * 4000) + "acgttgcaacgttgcaacgttgcaacgttg"
    elif shape == "homopolymer":
        ms = synthetic_ms(5000, k, seed=2, mask="ones").upper() + "A" * 50_000 + synthetic_ms(5000, k, seed=4, mask="ones")
        ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    else:
        ms = synthetic_ms(300_000, k, seed=3, mask="mixed" if shape == "mixed_mask" else "ones")
    narrow = fm.DeviceIndex.build(ms.encode(), k)
    monkeypatch.setenv("FMSI_HIP_FORCE_WIDE_BUILD", "1")
    wide = fm.DeviceIndex.build(ms.encode(), k)
    monkeypatch.delenv("FMSI_HIP_FORCE_WIDE_BUILD")
    assert wide.info.rows64 == 1 and narrow.info.rows64 == 0
    bn, mn = narrow.download_layout()
    bw, mw = wide.download_layout()
    assert np.array_equal(bn, bw) and np.array_equal(mn, mw)
    assert (list(wide.info.counts), wide.info.dollar_position, wide.info.n) == (list(narrow.info.counts), narrow.info.dollar_position, narrow.info.n)
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    check(fm, wide, orc, mixed_kmers(ms, k, 20_000, seed=int(rng.integers(1, 1000))), k)
    # the kLCP array (what `fmsi index` writes as .klcp): bit for bit the 32-bit builder's
    kn = fm.DeviceIndex.build(ms.encode(), k, with_klcp=True).download_klcp()
    monkeypatch.setenv("FMSI_HIP_FORCE_WIDE_BUILD", "1")
    kw = fm.DeviceIndex.build(ms.encode(), k, with_klcp=True)
    assert kw.info.rows64 == 1 and kw.info.has_klcp == 1
    assert np.array_equal(kw.download_klcp(), kn)


def test_wide_builder_takes_repetitive_texts(product_built, oracle_built, monkeypatch):
    """(Until round 4 this text was refused with FMSI_HIP_ERANGE: a group of 32 000 suffixes that agree on 29 bases.)  Prefix doubling with
    64-bit ranks orders it: the index answers like the oracle's, and `fmsi export`'s walk gives the superstring back."""
    fm = product_built
    monkeypatch.setenv("FMSI_HIP_FORCE_WIDE_BUILD", "1")
    ms = ("ACGTTGCA" * 4000) + "acgttgcaacgttgcaacgttgcaacgttg"
    dev = fm.DeviceIndex.build(ms.encode(), 31)
    assert dev.info.rows64 == 1
    orc = oracle_py.Oracle.construct(ms, 31, use_klcp=False)
    check(fm, dev, orc, mixed_kmers(ms, 31, 2000, seed=5), 31)
    assert dev.export().decode() == ms


# (FMSI_TEST_EXTENDED=1 runs these: 41 s for the uniform text, 44 s more for the one with repeat families -- ~10^9 tied suffixes through the 64-bit prefix
# doubling.  No BASELINE configuration has 2^32 rows, the suite has a time limit of which it took 580 s, and the 64-bit paths -- builder, kernels, CLI --
# are held against the 32-bit ones and the oracle on small texts in every other test of this file (FMSI_HIP_FORCE_ROWS64).  Last run in full at the end of round 5.)
@pytest.mark.skipif(not os.environ.get("FMSI_TEST_EXTENDED"), reason="41 + 44 s for sizes no BASELINE configuration has (FMSI_TEST_EXTENDED=1 runs them)")
@pytest.mark.parametrize("repeats", [(0, 0, 0), (200, 50, 10)], ids=lambda r: "repeat_families" if r[0] else "uniform")
def test_index_of_more_than_2_to_32_rows(product_built, oracle_built, repeats):
    """A REAL index with 64-bit rows: 4.5 x 10^9 bases generated and indexed on the device (2.25 GB of blocks, three superblocks of 2^31
    rows), queried by query_wide_kernel: strand by strand in both modes against the oracle on the downloaded planes (2^15 k-mers), present
    k-mers present and absent ones absent on 2^20 seeded queries, windows through the run entry point, ORDERS against the oracle.
    repeat_families (round 4): a fifth of the text's 2048-base segments are copies of 50 consensus segments at 1 % divergence -- tied groups of
    thousands of suffixes, ~10^9 tied suffixes in all: the builder's prefix doubling with 64-bit ranks (until round 4: FMSI_HIP_ERANGE)."""
    import torch
    import bench_cpu
    fm = product_built
    n, k = 4_500_000_000, 31
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 170e9:   # (the builder's temporaries at this size: ~150 GB)
        pytest.skip(f"needs ~170 GB of free HBM for the builder's temporaries, this device has {free_b / 1e9:.0f} GB free")
    with fm.builder_chain(0):
        dev = fm.DeviceIndex.build_synthetic(n, 99, k, prefix_len=14, repeats=repeats)
    info = dev.info
    assert info.rows64 == 1 and info.n == n + 1 and info.n > (1 << 32) and info.prefix_len == 14
    nq = 1 << 20
    s = torch.cuda.current_stream().cuda_stream
    d_k = torch.empty(nq, dtype=torch.int64, device="cuda")
    d_o = torch.empty(nq, dtype=torch.uint8, device="cuda")
    dev.synthetic_queries_dev(k, 42, 0, nq, d_k.data_ptr(), s)
    outs = {}
    for name, mode in (("or", fm.MODE_OR), ("all", fm.MODE_ALL)):
        dev.query_kmers_dev(d_k.data_ptr(), nq, k, mode | fm.STRICT, d_o.data_ptr(), stream=s)
        torch.cuda.synchronize()
        outs[name] = d_o.cpu().numpy().copy()
    f, r = fm.decode_strict(outs["all"])
    present = (f == 1) | (r == 1)
    assert 0.49 < present.mean() < 0.51 and ((f == 0) | (r == 0)).sum() == 0     # max-ones: half the stream is drawn from the text, and ON
    assert np.array_equal(outs["or"], outs["all"])                                 # (max-ones: the modes agree)
    dev.query_kmers_dev(d_k.data_ptr(), nq, k, fm.MODE_ALL | fm.STRICT | fm.NO_PREFIX_TABLE, d_o.data_ptr(), stream=s)
    torch.cuda.synchronize()
    assert np.array_equal(d_o.cpu().numpy(), outs["all"])                          # the 14-deep table (4.3 GB of 16-byte entries) changes no answer
    sample = 1 << 15
    kmers = d_k[:sample].cpu().numpy().view(np.uint64)
    codes = ((kmers[:, None] >> (2 * np.arange(k, dtype=np.uint64))) & np.uint64(3)).astype(np.uint8)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()
    orc = bench_cpu.oracle_from_index(dev, k, with_klcp=False)
    assert orc.n == n + 1
    for name, mode_all in (("or", False), ("all", True)):
        of, orr = orc.ternaries_bulk(text, sample, k, mode_all)
        gf, gr = fm.decode_strict(outs[name][:sample])
        assert np.array_equal(gf, of) and np.array_equal(gr, orr), name
    _, _, oo_f, oo_r = orc.ternaries_bulk(text, sample, k, False, orders=True)
    got = dev.query_kmers(kmers, k, fm.ORDERS | fm.STRICT)
    assert np.array_equal(got[:, 0], oo_f) and np.array_equal(got[:, 1], oo_r)
    assert int(oo_f.max()) > (1 << 32) or int(oo_r.max()) > (1 << 32)              # ranks beyond 32 bits really occur
    orc.close()
    # `fmsi export` of such an index: the LF walk over 64-bit rows gives back the text the generator made -- all of it (max-ones: upper case
    # but for the last k - 1 positions), compared word by word with fmsi_hip_synthetic_text_word
    got = np.frombuffer(dev.export(), dtype=np.uint8)
    assert got.size == n
    lower = (got & 32) != 0
    assert not lower[: n - (k - 1)].any() and lower[n - (k - 1):].all()
    code = np.zeros(256, dtype=np.uint8)
    code[ord("C")] = code[ord("c")] = 1
    code[ord("G")] = code[ord("g")] = 2
    code[ord("T")] = code[ord("t")] = 3
    lib = fm.lib()
    rng = np.random.default_rng(5)
    words = np.unique(np.concatenate([np.arange(0, 4096), (1 << 27) + np.arange(-2048, 2048), rng.integers(0, n // 32 - 1, size=200_000), [n // 32 - 1]]))
    # (sampled words incl. those around the text's 2^32nd base = word 2^27)
    pos = words[:, None] * 32 + np.arange(32)
    have = (code[got[pos]].astype(np.uint64) << (2 * np.arange(32, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
    want = np.array([lib.fmsi_hip_synthetic_text_word_repeats(99, int(w), *repeats) if repeats[0] else lib.fmsi_hip_synthetic_text_word(99, int(w)) for w in words],
                    dtype=np.uint64)
    assert np.array_equal(have, want), words[np.nonzero(have != want)[0][:5]]


@pytest.mark.skipif(not os.environ.get("FMSI_TEST_EXTENDED"), reason="33 s for a size no BASELINE configuration has (FMSI_TEST_EXTENDED=1 runs it); the library-level test above "
                                                                    "holds an index of 4.5 x 10^9 rows against the oracle, test_gpu_index_cli.py the `fmsi index` command")
def test_cli_index_query_export_beyond_2_to_32_bases(product_built):
    """The three commands as a user of the reference would chain them, on a masked superstring of 4.32 x 10^9 bases (> 2^32): `fmsi index`
    (FASTA -> the on-device builder for 64-bit rows -> the reference's files), `fmsi query` on those files (k-mers cut from the text answer
    their mask bit, random ones 0; with -S, too) and `fmsi export` (the walk over 64-bit rows): the exported FASTA is the input superstring, byte for byte."""
    import shutil
    import subprocess
    fm = product_built
    n, k = 4_320_000_000, 31
    d = "/dev/shm/fmsi_wide_cli"
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    try:
        fa, out_fa, q_fa = f"{d}/ms.fa", f"{d}/exported.fa", f"{d}/q.fa"
        rng = np.random.default_rng(77)
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        picks = []          # (k-mer, its mask bit) cut from the text while it is written
        step = 1 << 28
        with open(fa, "wb") as f:
            f.write(b">ms\n")
            for at in range(0, n, step):
                m = min(step, n - at)
                chunk = lut[rng.integers(0, 4, size=m, dtype=np.uint8)]
                off = rng.integers(0, 10, size=m, dtype=np.uint8) >= 7          # 30 % of the k-mers OFF (lower case)
                if at + m > n - (k - 1):
                    off[max(0, n - (k - 1) - at):] = True                          # the last k - 1 positions carry no k-mer
                chunk = chunk | (off.astype(np.uint8) << 5)
                for p in rng.integers(0, m - k, size=120):
                    picks.append((bytes(chunk[p:p + k]).upper(), not bool(off[p])))
                chunk.tofile(f)
            f.write(b"\n")
        rc, _, err = fm.run_cli(["index", "-k", str(k), fa])      # (as the reference's default: with the kLCP array)
        assert rc == 0, err.decode()[-2000:]
        assert os.path.getsize(fa + ".fmsi.klcp") > n // 8
        with open(q_fa, "wb") as f:
            for i, (kmer, _) in enumerate(picks):
                f.write(b">p%d\n%s\n" % (i, kmer))
            for i in range(500):
                f.write(b">r%d\n%s\n" % (i, bytes(lut[rng.integers(0, 4, size=k, dtype=np.uint8)])))
        rc, out, err = fm.run_cli(["query", "-k", str(k), "-q", q_fa, fa])
        assert rc == 0, err.decode()[-2000:]
        answers = [l.split(b"\t")[-1] for l in out.split(b"\n") if l]      # (one line per record: name, tab, one digit per window)
        assert len(answers) == len(picks) + 500
        assert [a == b"1" for a in answers[:len(picks)]] == [on for _, on in picks]     # (random text: a k-mer occurs once, on one strand)
        assert all(a == b"0" for a in answers[len(picks):])
        rc, out_s, err = fm.run_cli(["query", "-S", "-k", str(k), "-q", q_fa, fa])      # -S on 64-bit rows: the plain search, the same answers
        assert rc == 0 and out_s == out, err.decode()[-2000:]
        with open(out_fa, "wb") as f:
            res = subprocess.run([str(fm.CLI_PATH), "export", fa], stdout=f, stderr=subprocess.PIPE)
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        head = b">exported masked superstring\n"
        assert os.path.getsize(out_fa) == len(head) + n + 1
        a = np.memmap(fa, dtype=np.uint8, mode="r", offset=4)
        b = np.memmap(out_fa, dtype=np.uint8, mode="r", offset=len(head))
        assert open(out_fa, "rb").read(len(head)) == head
        for at in range(0, n + 1, step):
            assert np.array_equal(a[at:at + step], b[at:at + step]), at
        del a, b
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("name,kind,fn", [c for c in FN_CASES if c[2] in ("and", "xor", "1-1", "2-3")])
def test_cli_demasking_functions_with_wide_rows(fm, name, kind, fn):
    """`fmsi query -f and|xor|r-s` over 64-bit rows (the kernel of fmsi_fn.hip is a template over the row width, too): the reference's
    golden output, k <= 32 and k = 40, 64 alike."""
    k = CLI_DATASETS[name][0]
    rc, out, err = fm.run_cli(["query", "-f", fn, "-k", str(k), "-q", str(query_file(name, kind)), index_prefix(name)])
    assert rc == 0, err.decode()
    assert out == golden_fn_out(name, kind, fn)

