"""Randomised parity sweep on the GPU: many small indexes with random length, k, mask, repeat structure and prefix-table
depth; every query path the library has (lock-step, refill, one-kernel pair, phased with and without text verification,
runs, streaming, long k) must return the oracle's per-strand results for every k-mer of the text, its reverse
complements, near misses and random k-mers.  Catches corner cases no hand-written case thought of (n = 1, k = 1, t = k,
intervals at block boundaries, '$' row and suffix-1 row inside the matched range, ...)."""
import random

import numpy as np
import pytest

import oracle_py
from conftest import revcomp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(product_built):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device visible")
    return product_built


def random_ms(rng, n, k):
    style = rng.choice(["uniform", "repeats", "lowcomplexity", "periodic"])
    if style == "uniform":
        s = [rng.choice("ACGT") for _ in range(n)]
    elif style == "lowcomplexity":
        s = [rng.choice("AAAC") for _ in range(n)]
    elif style == "periodic":
        unit = [rng.choice("ACGT") for _ in range(rng.randrange(1, 9))]
        s = [unit[i % len(unit)] for i in range(n)]
    else:
        s = []
        while len(s) < n:
            if s and rng.random() < 0.5:
                a = rng.randrange(0, len(s))
                s += s[a:a + rng.randrange(1, 3 * k + 2)]
            else:
                s += [rng.choice("ACGT") for _ in range(rng.randrange(1, 2 * k + 2))]
        s = s[:n]
    return "".join(c if rng.random() < 0.6 else c.lower() for c in s)


CONFIGS = list(range(150))


@pytest.mark.parametrize("seed", CONFIGS)
def test_random_config(fm, oracle_built, seed):
    rng = random.Random(9000 + seed)
    k = rng.choice([1, 2, 3, 4, 5, 7, 8, 11, 13, 16, 21, 31, 32])
    n = max(k, rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 3000, 20000]))
    t = rng.randrange(0, min(k, 9) + 1)
    ms = random_ms(rng, n, k)
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=True)
    dev = fm.DeviceIndex.build(ms.encode(), k, with_klcp=True, prefix_len=t)
    dev_p = fm.DeviceIndex.build(ms.encode(), k, with_klcp=True, prefix_len=t).enable_pair_layout()
    variants = [(dev, fm.KERNEL_LOCKSTEP), (dev, fm.KERNEL_REFILL), (dev_p, fm.KERNEL_PAIR), (dev_p, fm.KERNEL_PHASED), (dev_p, 0), (dev_p, fm.KERNEL_SORTED)]
    if 0 < t < k:
        dev_v = fm.DeviceIndex.build(ms.encode(), k, with_klcp=True, prefix_len=t).enable_pair_layout().enable_verify()
        variants += [(dev_v, fm.KERNEL_PHASED), (dev_v, 0), (dev_v, fm.KERNEL_REFILL)]
    up = ms.upper()
    kmers = [up[p:p + k] for p in range(len(up) - k + 1)][:4000]
    kmers += [revcomp(x) for x in kmers[:2000]]
    kmers += ["".join(rng.choice("ACGT") for _ in range(k)) for _ in range(300)]
    kmers += [x[:-1] + rng.choice("ACGT") for x in kmers[:200]] + [rng.choice("ACGT") + x[1:] for x in kmers[:200]]
    raw = "".join(kmers).encode()
    packed = fm.pack_kmers(kmers, k)
    for mode_all in (False, True):
        flag = fm.MODE_ALL if mode_all else fm.MODE_OR
        fwd, rc = orc.ternaries_bulk(raw, len(kmers), k, mode_all)
        for d, kern in variants:
            gf, gr = fm.decode_strict(d.query_kmers(packed, k, flag | fm.STRICT | kern))
            assert np.array_equal(gf, fwd) and np.array_equal(gr, rc), (seed, k, n, t, kern, mode_all)
    fwd, rc, of, orr = orc.ternaries_bulk(raw, len(kmers), k, False, orders=True)
    for d, kern in variants:
        got = d.query_kmers(packed, k, fm.ORDERS | fm.STRICT | kern)
        assert np.array_equal(got[:, 0], of) and np.array_equal(got[:, 1], orr), (seed, k, n, t, kern)
    # the text itself as one run: runs and streaming entry points, both strands
    if len(up) >= k:
        words = fm.pack_sequence(up.encode())
        nw = len(up) - k + 1
        wk = [up[p:p + k] for p in range(nw)]
        fwd, rc = orc.ternaries_bulk("".join(wk).encode(), nw, k, True)
        for d in {v[0] for v in variants}:
            gf, gr = fm.decode_strict(d.query_runs(words, len(up), [0], [nw], k, fm.MODE_ALL | fm.STRICT))
            assert np.array_equal(gf, fwd) and np.array_equal(gr, rc), (seed, "runs")
            gf, gr = fm.decode_strict(d.query_stream(words, len(up), [0], [nw], k, fm.MODE_ALL | fm.STRICT))
            assert np.array_equal(gf, fwd) and np.array_equal(gr, rc), (seed, "stream")
    # and export gives the superstring back
    assert dev.export().decode() == ms
