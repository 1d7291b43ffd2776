"""FMSI_HIP_OUT_BITS (bit-packed results, SURVEY.md 8(b) `out /*bit-packed ...*/`): for every kernel, mode and entry point the
packed words unpack to exactly the one-byte-per-window results of the same call without the flag -- which the other GPU tests
pin against the oracle -- and to the oracle's answers directly.  The reference emits one symbol per window
(src/fms_index.h:305-309); one bit per window is that symbol.
"""
import random

import numpy as np
import pytest

import oracle_py
from conftest import GOLDEN, index_prefix, mixed_kmers, revcomp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(product_built):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on an MI355X")
    return product_built


def ms_k13():
    return "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13") if not l.startswith(">"))


def test_unpack_helper():
    import fmsi_amd as fm
    w = np.array([0b1011, 1 << 63], dtype=np.uint64)
    got = fm.unpack_bits(w, 70, 0)
    assert got[:4].tolist() == [1, 1, 0, 1] and got[63] == 0 and got.size == 70 and got[4:63].sum() == 0
    s = fm.unpack_bits(np.array([0xA5], dtype=np.uint64), 3, fm.STRICT)
    assert s.tolist() == [5, 10, 0]


@pytest.mark.parametrize("n_kmers", [1, 63, 64, 65, 4096, 4097, 20011])
def test_packed_equals_bytes_every_kernel(fm, oracle_built, n_kmers):
    """k = 13 SARS-CoV-2 fixture (non-max-ones mask): every kernel x {OR, ALL} x {plain, STRICT}; sizes around the word and
    chunk boundaries (the tail word's unused bits must be zero)."""
    name, k = "sars_k13", 13
    kmers = mixed_kmers(ms_k13(), k, n_kmers, seed=23)
    packed = fm.pack_kmers(kmers, k)
    orc = oracle_py.Oracle.load(index_prefix(name))
    plain = fm.DeviceIndex.from_files(index_prefix(name))
    tab = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=6).enable_pair_layout()
    cand = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=6).enable_pair_layout().enable_verify()    # one-lane lookups pack in-kernel
    variants = [(plain, fm.KERNEL_LOCKSTEP), (plain, fm.KERNEL_REFILL), (tab, fm.KERNEL_PAIR), (tab, fm.KERNEL_PHASED), (cand, fm.KERNEL_PHASED),
                (cand, 0), (cand, fm.NO_PREFIX_TABLE | fm.KERNEL_PAIR), (tab, fm.KERNEL_SORTED)]
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk("".join(kmers).encode(), n_kmers, k, mode_all)
        want_strict = ((fwd + 1) | ((rc + 1) << 2)).astype(np.uint8)
        want_fast = (np.where(fwd != -1, fwd, rc) == 1) if mode_all else ((fwd == 1) | (rc == 1))
        for d, kern in variants:
            for extra, want in ((0, want_fast.astype(np.uint8)), (fm.STRICT, want_strict)):
                words = d.query_kmers(packed, k, flag | kern | extra | fm.OUT_BITS)
                assert words.dtype == np.uint64 and words.size == fm.packed_words(n_kmers, extra)
                got = fm.unpack_bits(words, n_kmers, extra)
                assert np.array_equal(got, want), (mode_all, kern, extra)
                assert np.array_equal(got, d.query_kmers(packed, k, flag | kern | extra))
                # nothing beyond the last window
                per = 16 if extra else 64
                tail = n_kmers % per
                if tail:
                    assert int(words[-1]) >> ((4 if extra else 1) * tail) == 0


def test_packed_runs_stream_and_windows(fm, oracle_built):
    """The run / window / -S entry points: packed == bytes == oracle, k = 31 max-ones fixture, ragged runs."""
    name, k = "sars_k31", 31
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_test2_ones.fna.ms.k31") if not l.startswith(">")).upper()
    rng = random.Random(9)
    seq = ms[100:5200] + "".join(rng.choice("ACGT") for _ in range(700)) + revcomp(ms[7000:11000])
    words = fm.pack_sequence(seq.encode())
    run_start, run_windows, pos = [], [], 0
    while pos + k + 5 < len(seq):
        w = rng.choice([1, 2, 63, 64, 65, 200, 777])
        w = min(w, len(seq) - pos - k + 1)
        run_start.append(pos)
        run_windows.append(w)
        pos += w + rng.choice([0, 3, 40])
    n = sum(run_windows)
    kmers = [seq[s + i: s + i + k] for s, w in zip(run_start, run_windows) for i in range(w)]
    orc = oracle_py.Oracle.load(index_prefix(name))
    fwd, rc = orc.ternaries_bulk("".join(kmers).encode(), n, k, True)
    want = ((fwd + 1) | ((rc + 1) << 2)).astype(np.uint8)
    klcp = fm.DeviceIndex.from_files(index_prefix(name), use_klcp=True, prefix_len=8)
    cand = fm.DeviceIndex.from_files(index_prefix(name), prefix_len=16).enable_pair_layout().enable_verify()   # k - t = 15: base candidates
    flag = fm.MODE_ALL | fm.STRICT
    for call in (lambda f: klcp.query_stream(words, len(seq), run_start, run_windows, k, f),
                 lambda f: klcp.query_runs(words, len(seq), run_start, run_windows, k, f),
                 lambda f: cand.query_runs(words, len(seq), run_start, run_windows, k, f)):
        got = fm.unpack_bits(call(flag | fm.OUT_BITS), n, fm.STRICT)
        assert np.array_equal(got, want)
        assert np.array_equal(got, call(flag))
    starts = np.array([s + i for s, w in zip(run_start, run_windows) for i in range(w)], dtype=np.uint64)
    gw = fm.unpack_bits(cand.query_windows(words, len(seq), k, flag | fm.OUT_BITS, window_start=starts), n, fm.STRICT)
    assert np.array_equal(gw, want)
    # non-STRICT one bit per window, OR mode
    f1, r1 = orc.ternaries_bulk("".join(kmers).encode(), n, k, False)
    bits = fm.unpack_bits(cand.query_runs(words, len(seq), run_start, run_windows, k, fm.MODE_OR | fm.OUT_BITS), n, 0)
    assert np.array_equal(bits.astype(bool), (f1 == 1) | (r1 == 1))


def test_packed_long_k(fm, oracle_built):
    """k > 32 (query_longk_kernel / the WIDE streaming kernel) through the packing pass."""
    name, k = "repeats_k40", 40
    prefix = str(GOLDEN / "idx" / name)
    ms = "".join(l.strip() for l in open(GOLDEN / "repeats.fa") if not l.startswith(">"))
    seq = ms[50:900].upper()
    n = len(seq) - k + 1
    words = fm.pack_sequence(seq.encode())
    dev = fm.DeviceIndex.from_files(prefix, use_klcp=True)
    for fn in (dev.query_runs, dev.query_stream):
        a = fn(words, len(seq), [0], [n], k, fm.MODE_OR | fm.STRICT)
        b = fm.unpack_bits(fn(words, len(seq), [0], [n], k, fm.MODE_OR | fm.STRICT | fm.OUT_BITS), n, fm.STRICT)
        assert np.array_equal(a, b) and (a & 3).max() == 2   # (some windows present)


def test_packed_needs_aligned_output(fm):
    import torch
    dev = fm.DeviceIndex.from_files(index_prefix("sars_k31"))
    km = torch.zeros(128, dtype=torch.int64, device="cuda")
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(fm.FmsiError):
        dev.query_kmers_dev(km.data_ptr(), 128, 31, fm.MODE_OR | fm.OUT_BITS, out.data_ptr() + 4)
    dev.query_kmers_dev(km.data_ptr(), 128, 31, fm.MODE_OR | fm.OUT_BITS, out.data_ptr())
    dev.sync()
