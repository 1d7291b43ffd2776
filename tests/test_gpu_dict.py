"""The k-mer dictionary (fmsi_hip_index_enable_dict, fmsi_dict.hip) against the oracle: a private accelerator must give, strand by
strand and in both modes, exactly what get_range_with_pattern() + infer_presence() give (src/fms_index.h:117-144).  Covered: every
golden index with k <= 32 (mixed masks: answers that depend on the suffix order come from the index's own search), exhaustive
k-mer sets incl. palindromes (even k), the overflow levels, runs / windows / bit-packed output, handles built on the device and
handles loaded from the reference's files, other k and ORDERS falling back to the other kernels.
"""
import itertools
import random

import numpy as np
import pytest

import oracle_py
from conftest import DATASETS, GOLDEN, index_prefix, mixed_kmers, revcomp, synthetic_ms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(product_built):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on an MI355X")
    return product_built


def check_against_oracle(fm, dev, orc, kmers, k):
    raw = "".join(kmers).encode()
    packed = fm.pack_kmers(kmers, k)
    n = len(kmers)
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk(raw, n, k, mode_all)
        got = dev.query_kmers(packed, k, flag | fm.STRICT | fm.KERNEL_DICT)
        gf, gr = fm.decode_strict(got)
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc), ("strict", mode_all)
        assert np.array_equal(dev.query_kmers(packed, k, flag | fm.STRICT), got)                      # AUTO picks the dictionary
        want_fast = (np.where(fwd != -1, fwd, rc) == 1) if mode_all else ((fwd == 1) | (rc == 1))
        fast = dev.query_kmers(packed, k, flag | fm.KERNEL_DICT)
        assert np.array_equal(fast.astype(bool), want_fast), ("fast", mode_all)
        for extra, ref in ((0, fast), (fm.STRICT, got)):
            words = dev.query_kmers(packed, k, flag | fm.KERNEL_DICT | extra | fm.OUT_BITS)
            assert np.array_equal(fm.unpack_bits(words, n, extra), ref), ("bits", mode_all, extra)
        # the LF walk on the same handle (no accelerators) agrees
        assert np.array_equal(dev.query_kmers(packed, k, flag | fm.STRICT | fm.NO_PREFIX_TABLE), got)


@pytest.mark.parametrize("name", [n for n in DATASETS if DATASETS[n][0] <= 32])
def test_dictionary_equals_oracle_on_the_golden_indexes(fm, oracle_built, name):
    k = DATASETS[name][0]
    orc = oracle_py.Oracle.load(index_prefix(name))
    dev = fm.DeviceIndex.from_files(index_prefix(name)).enable_dict()
    assert dev.info.dict_bytes > 0 and dev.info.dict_levels >= 1
    ms_file = {"integration_a": "ref_data/integration_a.fa", "integration_b": "ref_data/integration_b.fa",
               "sars_k13": "ref_data/GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13",
               "sars_k31": "ref_data/GCF_test2_ones.fna.ms.k31", "mixed_k11": "mixed.fa"}[name]
    ms = "".join(l.strip() for l in open(GOLDEN / ms_file) if not l.startswith(">"))
    kmers = mixed_kmers(ms, k, 5000, seed=31)
    if k <= 6:
        kmers = ["".join(p) for p in itertools.product("ACGT", repeat=k)]
    check_against_oracle(fm, dev, orc, kmers, k)
    # every k-mer of the superstring, both orientations (all of them are keys)
    up = ms.upper()
    wins = [up[i:i + k] for i in range(0, len(up) - k + 1, max(1, (len(up) - k + 1) // 4000))]
    check_against_oracle(fm, dev, orc, wins + [revcomp(w) for w in wins], k)
    # the dictionary also sits next to the other accelerators
    both = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=min(k - 1, 6)).enable_pair_layout().enable_verify().enable_dict()
    check_against_oracle(fm, both, orc, kmers[:2000], k)


@pytest.mark.parametrize("k,mask", [(6, "mixed"), (8, "mixed"), (12, "ones"), (32, "mixed"), (1, "mixed"), (2, "mixed")])
def test_even_k_palindromes_and_repeats(fm, oracle_built, k, mask):
    """Even k has palindromic k-mers (q == rc(q): one strand, one answer); a short random text over a small k makes every key
    occur many times on both strands with ON and OFF occurrences mixed (`-O` = the first row of the interval: the index's own
    search resolves those); k = 32 uses all 64 bits of the key."""
    rng = random.Random(k)
    n = 3000 if k <= 12 else 60_000
    ms = synthetic_ms(n, k, seed=100 + k, mask=mask)
    if k >= 12:   # plant reverse complements and exact repeats
        ms = ms[:1000] + revcomp(ms[200:700]).upper() + ms[1000:2000] + ms[300:900] + ms[2000:]
        ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    dev = fm.DeviceIndex.build(ms.encode(), k).enable_dict()
    if k <= 8:
        kmers = ["".join(p) for p in itertools.product("ACGT", repeat=k)]
        if mask == "mixed" and k >= 6:
            assert dev.info.dict_ambiguous > 0      # the suffix-order path was exercised
    else:
        kmers = mixed_kmers(ms, k, 6000, seed=k)
        pal = []
        for _ in range(300):                          # palindromes: half + its reverse complement
            h = "".join(rng.choice("ACGT") for _ in range(k // 2))
            pal.append(h + revcomp(h))
        kmers += pal
        up = ms.upper()
        kmers += [up[i:i + k] for i in range(0, len(up) - k + 1, 7)][:4000]
    check_against_oracle(fm, dev, orc, kmers, k)


def test_overflow_levels_runs_and_other_k(fm, oracle_built):
    """1.2 M distinct k-mers at 9.2 keys per bucket: some 2000 buckets overflow into level 1.  Runs of windows through the
    dictionary; a query with another k, and ORDERS, take the other kernels and still agree with the oracle."""
    k = 31
    ms = synthetic_ms(1_200_000, k, seed=12345, mask="mixed")
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    dev = fm.DeviceIndex.build(ms.encode(), k).enable_dict()
    info = dev.info
    assert info.dict_levels >= 2 and info.dict_keys[1] > 100 and info.dict_keys[0] > 1_190_000
    assert info.dict_bytes >= (1 << info.dict_log2_buckets) * 128
    sparse = fm.DeviceIndex.build(ms.encode(), k).enable_dict(sparse=True)
    assert sparse.info.dict_log2_buckets == info.dict_log2_buckets + 1 and sparse.info.dict_keys[1] < info.dict_keys[1]
    kmers = mixed_kmers(ms, k, 60_000, seed=8)
    up = ms.upper()
    kmers += [up[i:i + k] for i in range(0, len(up) - k + 1, 17)]       # 70 k keys: the spilled ones among them
    check_against_oracle(fm, dev, orc, kmers, k)
    check_against_oracle(fm, sparse, orc, kmers[:20_000], k)
    # runs of windows (reads): both orientations and random stretches, ragged run lengths
    rng = random.Random(4)
    seq = up[5000:9000] + "".join(rng.choice("ACGT") for _ in range(600)) + revcomp(up[20000:23000]) + up[100:1100]
    words = fm.pack_sequence(seq.encode())
    run_start, run_windows, pos = [], [], 0
    while pos + k + 2 < len(seq):
        w = min(rng.choice([1, 5, 64, 130, 1000]), len(seq) - pos - k + 1)
        run_start.append(pos)
        run_windows.append(w)
        pos += w + rng.choice([0, 1, 29])
    wk = [seq[s + i: s + i + k] for s, w in zip(run_start, run_windows) for i in range(w)]
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk("".join(wk).encode(), len(wk), k, mode_all)
        got = dev.query_runs(words, len(seq), run_start, run_windows, k, flag | fm.STRICT)
        gf, gr = fm.decode_strict(got)
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
        bits = fm.unpack_bits(dev.query_runs(words, len(seq), run_start, run_windows, k, flag | fm.STRICT | fm.OUT_BITS), len(wk), fm.STRICT)
        assert np.array_equal(bits, got)
        starts = np.array([s + i for s, w in zip(run_start, run_windows) for i in range(w)], dtype=np.uint64)
        assert np.array_equal(dev.query_windows(words, len(seq), k, flag | fm.STRICT, window_start=starts), got)
    # another k: not the dictionary's business
    k2 = 21
    km2 = [s[:k2] for s in kmers[:3000]]
    f2, r2 = orc.ternaries_bulk("".join(km2).encode(), len(km2), k2, False)
    g2f, g2r = fm.decode_strict(dev.query_kmers(fm.pack_kmers(km2, k2), k2, fm.MODE_OR | fm.STRICT | fm.KERNEL_DICT))
    assert np.array_equal(g2f, f2) and np.array_equal(g2r, r2)
    # lookup (orders)
    _, _, of, orr = orc.ternaries_bulk("".join(kmers[:3000]).encode(), 3000, k, False, orders=True)
    got = dev.query_kmers(fm.pack_kmers(kmers[:3000], k), k, fm.ORDERS | fm.STRICT)
    assert np.array_equal(got[:, 0], of) and np.array_equal(got[:, 1], orr)


def test_dictionary_counters(fm):
    """One bucket read per k-mer (+ the second level's for the few in overflowed buckets), no LF steps."""
    k = 31
    ms = synthetic_ms(300_000, k, seed=3)
    dev = fm.DeviceIndex.build(ms.encode(), k).enable_dict()
    kmers = mixed_kmers(ms, k, 50_000, seed=1)
    _, cn = dev.query_kmers(fm.pack_kmers(kmers, k), k, fm.MODE_ALL, counters=True)
    assert int(cn[0]) == 0 and int(cn[6]) == len(kmers)
    assert len(kmers) <= int(cn[4]) <= int(1.1 * len(kmers)) and int(cn[1]) == int(cn[4])


@pytest.mark.parametrize("read_len,k,mask", [(150, 31, "mixed"), (62, 31, "mixed"), (45, 31, "ones"), (31, 31, "mixed"), (500, 13, "mixed"), (96, 32, "mixed"), (40, 8, "mixed")])
def test_chained_lookups_of_reads(fm, oracle_built, read_len, k, mask):
    """FMSI_HIP_DICT_CHAIN: reads of one length through fmsi_hip_query_runs take dict_chain_kernel -- a lane follows the superstring
    from the occurrence of its task's first window.  Reads from the text in both orientations, with substitutions (the chain breaks and
    the rest of the task takes ordinary lookups), random reads, reads that run to the text's two ends, tasks that straddle reads
    (windows per read not a multiple of 32, and reads of fewer than 32 windows), small k (every k-mer occurs many times, on both
    strands, with mixed mask bits): strand by strand what the oracle says, in both modes, as bytes and bit-packed, and equal to the
    unchained dictionary."""
    rng = random.Random(read_len * 100 + k)
    n = 40_000 if k > 12 else 3_000
    ms = synthetic_ms(n, k, seed=read_len + k, mask=mask)
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    dev = fm.DeviceIndex.build(ms.encode(), k).enable_dict(chain=True)
    plain = fm.DeviceIndex.build(ms.encode(), k).enable_dict()
    assert dev.info.dict_chain_bytes > 0 and plain.info.dict_chain_bytes == 0
    up = ms.upper()
    reads = []
    for r in range(700):
        kind = r % 7
        if kind in (0, 1, 2):
            p = rng.randrange(0, len(up) - read_len + 1)
            s = up[p:p + read_len]
        elif kind == 3:
            s = up[:read_len] if r % 2 else up[len(up) - read_len:]                   # the text's two ends
        elif kind == 4:
            s = "".join(rng.choice("ACGT") for _ in range(read_len))
        else:
            p = rng.randrange(0, len(up) - read_len + 1)
            s = list(up[p:p + read_len])
            for _ in range(1 + read_len // 60):
                s[rng.randrange(read_len)] = rng.choice("ACGT")                        # substitutions
            s = "".join(s)
        if kind in (1, 5) or (kind == 3 and r % 4 == 3):
            s = revcomp(s)
        reads.append(s)
    seq = "".join(reads)
    words = fm.pack_sequence(seq.encode())
    w = read_len - k + 1
    starts = np.arange(len(reads), dtype=np.uint64) * np.uint64(read_len)
    wins = np.full(len(reads), w, dtype=np.uint32)
    kmers = [r[i:i + k] for r in reads for i in range(w)]
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk("".join(kmers).encode(), len(kmers), k, mode_all)
        got, cn = dev.query_runs(words, len(seq), starts, wins, k, flag | fm.STRICT, counters=True)
        gf, gr = fm.decode_strict(got)
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc), mode_all
        assert np.array_equal(plain.query_runs(words, len(seq), starts, wins, k, flag | fm.STRICT), got)
        bits = dev.query_runs(words, len(seq), starts, wins, k, flag | fm.OUT_BITS)
        want_fast = (np.where(fwd != -1, fwd, rc) == 1) if mode_all else ((fwd == 1) | (rc == 1))
        assert np.array_equal(fm.unpack_bits(bits, len(kmers), 0).astype(bool), want_fast)
        assert int(cn[6]) == len(kmers)
        if w >= 64 and k > 12:
            assert int(cn[4]) < 0.6 * len(kmers)          # fewer bucket reads than windows: the chains really ran
    # the extras can be added to a dictionary that is already there (a handle from files keeps no text: the LF walk supplies it again)
    plain.enable_dict(chain=True)
    assert plain.info.dict_chain_bytes > 0
    assert np.array_equal(plain.query_runs(words, len(seq), starts, wins, k, fm.MODE_ALL | fm.STRICT), dev.query_runs(words, len(seq), starts, wins, k, fm.MODE_ALL | fm.STRICT))
    # ragged runs are not chained and still right
    starts2 = starts[:50].copy()
    wins2 = wins[:50].copy()
    wins2[::3] = max(1, w // 2)
    km2 = [reads[r][i:i + k] for r in range(50) for i in range(int(wins2[r]))]
    f2, r2 = orc.ternaries_bulk("".join(km2).encode(), len(km2), k, True)
    g2f, g2r = fm.decode_strict(dev.query_runs(words, len(seq), starts2, wins2, k, fm.MODE_ALL | fm.STRICT))
    assert np.array_equal(g2f, f2) and np.array_equal(g2r, r2)


@pytest.mark.parametrize("shape", ["poly_a", "period_4", "tandem_unit", "two_keys"])
def test_dictionary_on_degenerate_texts(fm, oracle_built, shape):
    """Texts whose k-mers are a handful of keys with 10^5 occurrences each (the construction folds occurrences with atomics, never
    with a thread that walks them), mixed masks on top so that nearly every key is ambiguous for `-O` and the index's own search
    decides; chained lookups on the same."""
    k = 21
    rng = random.Random(7)
    unit = {"poly_a": "A", "period_4": "ACGT", "tandem_unit": "".join(rng.choice("ACGT") for _ in range(997)), "two_keys": "AT"}[shape]
    n = 200_000
    up = (unit * (n // len(unit) + 1))[:n]
    ms = "".join(c if rng.random() < 0.5 else c.lower() for c in up[: n - k + 1]) + up[n - k + 1:].lower()
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    dev = fm.DeviceIndex.build(ms.encode(), k).enable_dict(chain=True)
    assert dev.info.dict_keys[0] <= 2 * len(unit) + 2 * k and dev.info.dict_ambiguous >= 1
    kmers = [up[i:i + k] for i in range(0, 3000)] + [revcomp(up[i:i + k]) for i in range(0, 3000, 3)] + mixed_kmers(ms, k, 3000, seed=2)
    check_against_oracle(fm, dev, orc, kmers, k)
    L = 300
    reads = [up[p:p + L] for p in range(0, 30_000, 997)] + [revcomp(up[p:p + L]) for p in range(5, 30_000, 1499)]
    seq = "".join(reads)
    w = L - k + 1
    wk = [r[i:i + k] for r in reads for i in range(w)]
    fwd, rc = orc.ternaries_bulk("".join(wk).encode(), len(wk), k, True)
    gf, gr = fm.decode_strict(dev.query_runs(fm.pack_sequence(seq.encode()), len(seq), np.arange(len(reads), dtype=np.uint64) * np.uint64(L),
                                             np.full(len(reads), w, dtype=np.uint32), k, fm.MODE_ALL | fm.STRICT))
    assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)


def test_handle_lifecycle_returns_its_memory(fm, tmp_path):
    """Handles created, given every accelerator, queried and destroyed over and over: after the library's block cache is trimmed the device
    has its memory back (nothing a handle allocated outlives it), and a destroyed handle's staging buffers do not pile up."""
    import torch
    ms = synthetic_text_for_lifecycle()
    free0 = None
    for cycle in range(7):
        if cycle == 1:      # (the first cycle also pays for what the runtime and rocPRIM keep for themselves)
            fm.trim_scratch(0)
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
        dev = fm.DeviceIndex.build(ms.encode(), 21, with_klcp=True, prefix_len=8).enable_pair_layout()
        dev.enable_verify()
        dev.enable_dict(chain=bool(cycle % 2))
        kmers = [ms[i:i + 21].upper() for i in range(0, 4000, 7)]
        a = dev.query_kmers(fm.pack_kmers(kmers, 21), 21, fm.MODE_ALL | fm.STRICT)
        b = dev.query_kmers(fm.pack_kmers(kmers, 21), 21, fm.MODE_ALL | fm.STRICT | fm.KERNEL_LOCKSTEP | fm.NO_PREFIX_TABLE)
        assert np.array_equal(a, b)
        dev.close()
    fm.trim_scratch(0)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (64 << 20), (free0, free1)


def synthetic_text_for_lifecycle():
    rng = random.Random(12)
    s = "".join(rng.choice("ACGT") for _ in range(3_000_000))
    return s[:-20] + s[-20:].lower()


def _sk_reads(rng, up, read_len, n_reads):
    reads = []
    for r in range(n_reads):
        kind = r % 7
        if kind in (0, 1, 2):
            p = rng.randrange(0, len(up) - read_len + 1)
            s = up[p:p + read_len]
        elif kind == 3:
            s = up[:read_len] if r % 2 else up[len(up) - read_len:]                   # the text's two ends
        elif kind == 4:
            s = "".join(rng.choice("ACGT") for _ in range(read_len))
        else:
            p = rng.randrange(0, len(up) - read_len + 1)
            s = list(up[p:p + read_len])
            for _ in range(1 + read_len // 60):
                s[rng.randrange(read_len)] = rng.choice("ACGT")                        # substitutions
            s = "".join(s)
        if kind in (1, 5) or (kind == 3 and r % 4 == 3):
            s = revcomp(s)
        reads.append(s)
    return reads


@pytest.mark.parametrize("read_len,k,mask,shape", [(150, 31, "ones", "random"), (100, 31, "mixed", "repeats"), (40, 27, "mixed", "random"),
                                                  (250, 32, "mixed", "repeats"), (64, 23, "ones", "lowcomplexity"), (33, 31, "mixed", "random")])
def test_super_kmer_dictionary(fm, oracle_built, read_len, k, mask, shape):
    """FMSI_HIP_DICT_SUPERKMER: reads of one length through fmsi_hip_query_runs take sk_runs_kernel -- a bucket read when a lane's
    minimizer changes, the window compared base by base with the records of its minimizer.  Texts with repeats (minimizers with many
    neighbourhoods: heavy records, the aux dictionary), low-complexity stretches (ties between equal m-mers and between an m-mer and its
    reverse complement: the leftmost / rightmost records), reads in both orientations, with substitutions, random reads, the text's two
    ends, tasks that straddle reads and reads shorter than a task, even k (self-complementary k-mers), mixed masks: strand by strand what
    the oracle says, in both modes, as bytes and bit-packed, equal to the k-mer dictionary -- also when the k-mer dictionary is gone."""
    rng = random.Random(read_len * 1000 + k)
    n = 40_000
    if shape == "random":
        ms = synthetic_ms(n, k, seed=read_len + k, mask=mask)
    else:
        body = synthetic_ms(n // 2, k, seed=read_len + k + 1, mask=mask)
        pieces = [body]
        unit = body[1000:1400]
        for c in range(30):      # a repeat family: copies with a few substitutions each
            cp = list(unit)
            for _ in range(4):
                cp[rng.randrange(len(cp))] = rng.choice("ACGTacgt")
            pieces.append("".join(cp))
            pieces.append(body[rng.randrange(0, len(body) - 300):][:300])
        if shape == "lowcomplexity":
            pieces += ["A" * 300, "AT" * 200, "ACG" * 150, "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT" * 3, revcomp(unit) * 2]
        pieces.append(revcomp(body[3000:3500]))
        ms = "".join(pieces)
        ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    dev = fm.DeviceIndex.build(ms.encode(), k).enable_dict(superkmer=True)
    only = fm.DeviceIndex.build(ms.encode(), k).enable_dict(superkmer_only=True)
    plain = fm.DeviceIndex.build(ms.encode(), k).enable_dict()
    assert dev.info.sk_bytes > 0 and dev.info.dict_bytes > 0 and only.info.sk_bytes > 0 and only.info.dict_bytes == 0 and plain.info.sk_bytes == 0
    assert dev.info.sk_minimizer_len == (21 if k <= 31 else 23)
    if shape != "random":
        assert dev.info.sk_aux_keys > 0      # the repeat family's minimizers are heavy
    up = ms.upper()
    reads = _sk_reads(rng, up, read_len, 700)
    seq = "".join(reads)
    words = fm.pack_sequence(seq.encode())
    w = read_len - k + 1
    starts = np.arange(len(reads), dtype=np.uint64) * np.uint64(read_len)
    wins = np.full(len(reads), w, dtype=np.uint32)
    kmers = [r[i:i + k] for r in reads for i in range(w)]
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk("".join(kmers).encode(), len(kmers), k, mode_all)
        got, cn = dev.query_runs(words, len(seq), starts, wins, k, flag | fm.STRICT, counters=True)
        gf, gr = fm.decode_strict(got)
        assert np.array_equal(gf, fwd), (mode_all, np.nonzero(gf != fwd)[0][:10])
        assert np.array_equal(gr, rc), (mode_all, np.nonzero(gr != rc)[0][:10])
        assert np.array_equal(only.query_runs(words, len(seq), starts, wins, k, flag | fm.STRICT), got)
        assert np.array_equal(plain.query_runs(words, len(seq), starts, wins, k, flag | fm.STRICT), got)
        assert np.array_equal(dev.query_runs(words, len(seq), starts, wins, k, flag | fm.STRICT | fm.NO_SUPERKMER), got)
        bits = only.query_runs(words, len(seq), starts, wins, k, flag | fm.OUT_BITS)
        want_fast = (np.where(fwd != -1, fwd, rc) == 1) if mode_all else ((fwd == 1) | (rc == 1))
        assert np.array_equal(fm.unpack_bits(bits, len(kmers), 0).astype(bool), want_fast)
        assert int(cn[6]) == len(kmers)
        if w >= 64 and shape == "random":
            assert int(cn[4]) < 0.4 * len(kmers)          # far fewer bucket reads than windows
    # isolated k-mers (when the k-mer dictionary is gone) and runs of differing lengths (the run table instead of three numbers)
    km = kmers[:3000]
    f1, r1 = orc.ternaries_bulk("".join(km).encode(), len(km), k, True)
    g1, cn1 = only.query_kmers(fm.pack_kmers(km, k), k, fm.MODE_ALL | fm.STRICT, counters=True)
    g1f, g1r = fm.decode_strict(g1)
    assert np.array_equal(g1f, f1) and np.array_equal(g1r, r1)
    assert int(cn1[0]) == 0 and int(cn1[6]) == len(km)     # (the records answer isolated k-mers, too: one-window reads, no LF step)
    assert np.array_equal(fm.unpack_bits(only.query_kmers(fm.pack_kmers(km, k), k, fm.MODE_ALL | fm.OUT_BITS), len(km), 0).astype(bool), np.where(f1 != -1, f1, r1) == 1)
    g1n = only.query_kmers(fm.pack_kmers(km, k), k, fm.MODE_ALL | fm.STRICT | fm.NO_SUPERKMER)             # ... unless told not to: the index's own search
    assert np.array_equal(g1n, g1)
    starts2 = starts[:50].copy()
    wins2 = wins[:50].copy()
    wins2[::3] = max(1, w // 2)
    km2 = [reads[r][i:i + k] for r in range(50) for i in range(int(wins2[r]))]
    f2, r2 = orc.ternaries_bulk("".join(km2).encode(), len(km2), k, True)
    for h in (dev, only):
        g2, cn2 = h.query_runs(words, len(seq), starts2, wins2, k, fm.MODE_ALL | fm.STRICT, counters=True)
        g2f, g2r = fm.decode_strict(g2)
        assert np.array_equal(g2f, f2) and np.array_equal(g2r, r2)
        assert int(cn2[0]) == 0 and int(cn2[6]) == len(km2)     # no LF step: the records answered


@pytest.mark.parametrize("n,k", [(3000, 32), (1200, 32), (3000, 31), (400, 24)])
def test_super_kmer_dictionary_on_a_tiny_text(fm, oracle_built, n, k):
    """Few records: the bucket count's floor.  A record keeps 36 bits of its hashed minimizer beside the bucket number, so at least
    mbits - 36 bits must pick the bucket however small the text is (k = 32: m = 23, 46 bits -> >= 2^10 buckets; the construction used to
    start at 2^6 for texts below ~10 kb, dropped the top bits of the remainder and failed its own self-check)."""
    ms = synthetic_ms(n, k, seed=n + k, mask="mixed")
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    only = fm.DeviceIndex.build(ms.encode(), k).enable_dict(superkmer_only=True)
    assert only.info.sk_bytes > 0 and only.info.dict_bytes == 0
    rng = random.Random(n)
    up = ms.upper()
    reads = _sk_reads(rng, up, min(150, n // 3), 60)
    read_len = len(reads[0])
    seq = "".join(reads)
    words = fm.pack_sequence(seq.encode())
    w = read_len - k + 1
    starts = np.arange(len(reads), dtype=np.uint64) * np.uint64(read_len)
    wins = np.full(len(reads), w, dtype=np.uint32)
    kmers = [r[i:i + k] for r in reads for i in range(w)]
    for mode_all in (False, True):
        fwd, rc = orc.ternaries_bulk("".join(kmers).encode(), len(kmers), k, mode_all)
        got, cn = only.query_runs(words, len(seq), starts, wins, k, (fm.MODE_ALL if mode_all else fm.MODE_OR) | fm.STRICT, counters=True)
        gf, gr = fm.decode_strict(got)
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc), mode_all
        assert int(cn[0]) == 0


def test_super_kmer_dictionary_score_ties(fm):
    """Two DIFFERENT m-mers of a window can share the smallest (19-bit) score: the window is then filed under the leftmost AND the rightmost
    of them, and a query of the other strand takes the one the text's own orientation does not -- whose records may be heavy or pushed out
    of an overflowed bucket on their own.  The construction therefore runs the text through the records twice (ties to the left, ties to
    the right).  3 x 10^7 windows of seeded 10-kb reads (a few hundred such ties; half the reads reverse-complemented): the super-k-mer
    dictionary, alone and beside the k-mer dictionary, == the k-mer dictionary, as bytes, STRICT and bit-packed."""
    import torch
    n, k, L, n_reads = 50_000_000, 31, 10_000, 3000
    dev = fm.DeviceIndex.build_synthetic(n, 1234, k).enable_dict(superkmer=True)
    only = fm.DeviceIndex.build_synthetic(n, 1234, k).enable_dict(superkmer_only=True)
    s = torch.cuda.current_stream().cuda_stream
    d_reads = torch.empty((n_reads * L) // 32 + 2, dtype=torch.int64, device="cuda")
    dev.synthetic_reads_dev(L, 7, 0, n_reads, d_reads.data_ptr(), 10000, s)
    torch.cuda.synchronize()
    packed = d_reads.cpu().numpy().view(np.uint64)
    starts = np.arange(n_reads, dtype=np.uint64) * np.uint64(L)
    wins = np.full(n_reads, L - k + 1, dtype=np.uint32)
    for flags in (fm.MODE_OR, fm.MODE_ALL | fm.STRICT, fm.MODE_ALL | fm.OUT_BITS):
        want = dev.query_runs(packed, n_reads * L, starts, wins, k, flags | fm.NO_SUPERKMER)
        got, cn = dev.query_runs(packed, n_reads * L, starts, wins, k, flags, counters=True)
        assert int(cn[0]) == 0 and int(cn[4]) < 0.3 * int(wins.sum())      # (the records answered: a bucket read per ~6 windows)
        assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
        assert np.array_equal(only.query_runs(packed, n_reads * L, starts, wins, k, flags), want)


def test_super_kmer_dictionary_needs_its_k(fm):
    """k outside 23..32: FMSI_HIP_ERANGE, and the handle keeps the k-mer dictionary the construction went through."""
    k = 15
    ms = synthetic_ms(5000, k, seed=3, mask="ones")
    dev = fm.DeviceIndex.build(ms.encode(), k)
    with pytest.raises(fm.FmsiError) as e:
        dev.enable_dict(superkmer=True)
    assert e.value.code == fm.E_RANGE
