"""Counterpart of the reference's verification test (tests/verify.py driven by tests/Makefile:21-25):
presence is re-derived by brute force from the masked superstring itself -- a k-mer is present iff it, or its reverse
complement, occurs somewhere with the mask bit of that occurrence ON (verify.py:11-20, 45-46) -- and compared with what
the programs print for seeded reads sent over a pipe (`-q -`), half of them substrings of the superstring and half
uniform (verify.py:98-107).  Same data sets and read shapes as the reference's runs (SARS-CoV-2 k = 13, 100 reads of 62;
GCF_test2_ones k = 31 with -O, 100 reads of 90), plus what it leaves out: `-S`, and reads with lower case and N.

This checker shares nothing with the oracle or the product: it pins the ORACLE on CPU and the PRODUCT CLI on the GPU.
"""
import random

import pytest

import oracle_py
from conftest import GOLDEN, index_prefix, revcomp

RUNS = [
    # name, masked superstring, k, reads, read length, extra CLI args
    ("sars_k13", "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13", 13, 100, 62, []),
    ("sars_k31", "GCF_test2_ones.fna.ms.k31", 31, 100, 90, ["-O"]),
    ("sars_k13", "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13", 13, 60, 200, ["-S"]),
    ("sars_k31", "GCF_test2_ones.fna.ms.k31", 31, 60, 300, ["-S", "-O"]),
]


def masked_superstring(fname: str) -> str:
    return "".join(l.strip() for l in open(GOLDEN / "ref_data" / fname) if not l.startswith(">"))


def on_kmers(ms: str, k: int) -> set:
    """k-mers with at least one ON occurrence (upper case first letter = mask 1, src/kmers.h:71-73)."""
    up = ms.upper()
    return {up[p:p + k] for p in range(len(ms) - k + 1) if ms[p].isupper()}


def brute_force_line(read: str, on: set, k: int) -> str:
    """What `fmsi query` must print for one record: one character per window; a window with a non-ACGT letter is 0
    (and the reference over-prints after an invalid character exactly as many zeros, src/main.cpp:351-360, so the
    line has one character per window there too for reads longer than k)."""
    out = []
    up = read.upper()
    for i in range(len(up) - k + 1):
        w = up[i:i + k]
        if any(c not in "ACGT" for c in w):
            out.append("0")
        else:
            out.append("1" if (w in on or revcomp(w) in on) else "0")
    return "".join(out)


def seeded_reads(ms: str, n: int, length: int, seed: int, dirty: bool = False):
    rng = random.Random(seed)
    up = ms.upper()
    reads = []
    for _ in range(n):
        if rng.random() < 0.5:
            s = rng.randrange(0, len(up) - length + 1)
            r = up[s:s + length]
            if rng.random() < 0.5:
                r = revcomp(r)
        else:
            r = "".join(rng.choice("ACGT") for _ in range(length))
        if dirty:   # lower case is valid; an N kills the windows that contain it
            r = "".join(c.lower() if rng.random() < 0.3 else c for c in r)
            if rng.random() < 0.5:
                p = rng.randrange(0, length)
                r = r[:p] + "N" + r[p + 1:]
        reads.append(r)
    return reads


def check_cli(run, name, fname, k, n_reads, length, extra, dirty=False):
    ms = masked_superstring(fname)
    on = on_kmers(ms, k)
    reads = seeded_reads(ms, n_reads, length, seed=42, dirty=dirty)
    stdin = "".join(f">header some comment\n{r}\n" for r in reads).encode()
    rc, out, err = run(["query", "-q", "-", "-k", str(k), *extra, index_prefix(name)], stdin)
    assert rc == 0, err.decode()
    lines = out.decode().split("\n")
    assert lines[-1] == "" and len(lines) == n_reads + 1
    n_pos = 0
    for read, line in zip(reads, lines):
        header, _, bits = line.partition("\t")
        assert header == "header"
        want = brute_force_line(read, on, k)
        if "N" in read:   # the over-print quirk can add zeros after the last invalid character: compare the windows
            assert bits[:len(want)] == want and set(bits[len(want):]) <= {"0"}
        else:
            assert bits == want
        n_pos += want.count("1")
    assert n_pos > 0.2 * n_reads * (length - k + 1)   # the positives were really there


@pytest.mark.parametrize("name,fname,k,n_reads,length,extra", RUNS)
@pytest.mark.parametrize("dirty", [False, True])
def test_oracle_cli_against_brute_force(oracle_built, name, fname, k, n_reads, length, extra, dirty):
    check_cli(lambda args, stdin: oracle_py.run_cli(oracle_py.ORACLE_CLI, args, stdin), name, fname, k, n_reads, length, extra, dirty)


@pytest.mark.skipif(not oracle_py.ref_available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,fname,k,n_reads,length,extra", RUNS[:2])
def test_reference_cli_against_brute_force(name, fname, k, n_reads, length, extra):
    """(the checker itself, held against the real reference on the reference's own two runs)"""
    check_cli(lambda args, stdin: oracle_py.run_cli(oracle_py.REF_CLI, args, stdin), name, fname, k, n_reads, length, extra)


# (--tables full: the k-mer dictionary answers the presence queries, -S ones included; candidates: the non-streaming ones only)
PRODUCT_RUNS = [(*r, t) for r in RUNS for t in ([], ["--tables", "full"], ["--tables", "candidates"]) if not (t and t[1] == "candidates" and "-S" in r[5])]


@pytest.mark.gpu
@pytest.mark.parametrize("name,fname,k,n_reads,length,extra,tables", PRODUCT_RUNS)
@pytest.mark.parametrize("dirty", [False, True])
def test_product_cli_against_brute_force(product_built, name, fname, k, n_reads, length, extra, dirty, tables):
    fm = product_built
    check_cli(lambda args, stdin: fm.run_cli([*args[:-1], *tables, args[-1]], stdin), name, fname, k, n_reads, length, extra, dirty)


# ---- `-f and|xor|r-s` (f-masked superstrings, query_mode::general): the same kind of checker -------------------------------------------
# f(ones, total) with total = occurrences of the k-mer and of its reverse complement in the superstring and ones = those with the
# mask ON, counted by brute force; a k-mer that reads the same as its reverse complement CHARACTER BY CHARACTER (the reference compares
# the query's characters, case included, src/kmers.h:53-68) is counted once.
FN_RUNS = [
    # index, masked superstring file (under tests/golden), k, function
    ("mixed_k11", "mixed.fa", 11, "and"), ("mixed_k11", "mixed.fa", 11, "xor"), ("mixed_k11", "mixed.fa", 11, "1-1"), ("mixed_k11", "mixed.fa", 11, "2-9"),
    ("integration_b", "ref_data/integration_b.fa", 3, "xor"), ("integration_b", "ref_data/integration_b.fa", 3, "and"),
    ("sars_k13", "ref_data/GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13", 13, "xor"), ("sars_k13", "ref_data/GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13", 13, "0-0"),
]


def occurrence_counts(ms: str, k: int) -> dict:
    """k-mer -> [ON occurrences, occurrences] over the superstring's own strand."""
    up = ms.upper()
    counts = {}
    for p in range(len(ms) - k + 1):
        c = counts.setdefault(up[p:p + k], [0, 0])
        c[0] += ms[p].isupper()
        c[1] += 1
    return counts


def apply_fn(fn: str, ones: int, total: int) -> bool:
    if fn == "and":
        return total > 0 and ones == total
    if fn == "xor":
        return ones % 2 == 1
    r, s = (int(x) for x in fn.split("-"))
    return r <= ones <= s


def brute_force_fn_line(read: str, counts: dict, k: int, fn: str) -> str:
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    out = []
    for i in range(len(read) - k + 1):
        w = read[i:i + k]
        if any(c not in "ACGTacgt" for c in w):
            out.append("0")
            continue
        rc = w[::-1].translate(comp)                        # case-preserving, as ReverseComplementString is
        ones, total = counts.get(w.upper(), (0, 0))
        if w != rc:                                           # AreStringsEqual on the characters
            o2, t2 = counts.get(rc.upper(), (0, 0))
            ones, total = ones + o2, total + t2
        out.append("1" if apply_fn(fn, ones, total) else "0")
    return "".join(out)


def check_fn_cli(run, name, fname, k, fn):
    ms = "".join(l.strip() for l in open(GOLDEN / fname) if not l.startswith(">"))
    counts = occurrence_counts(ms, k)
    length = max(4 * k, 40)
    reads = seeded_reads(ms * (1 + 2 * length // len(ms)), 80, length, seed=7, dirty=True)
    # (these k are odd: no k-mer is its own reverse complement here -- test_self_complementary_kmers_by_brute_force has an even k)
    stdin = "".join(f">r{i}\n{r}\n" for i, r in enumerate(reads)).encode()
    rc, out, err = run(["query", "-f", fn, "-q", "-", "-k", str(k), index_prefix(name)], stdin)
    assert rc == 0, err.decode()
    lines = out.decode().split("\n")
    assert lines[-1] == "" and len(lines) == len(reads) + 1
    ones = 0
    for read, line in zip(reads, lines):
        bits = line.partition("\t")[2]
        want = brute_force_fn_line(read, counts, k, fn)
        if "N" in read:
            assert bits[:len(want)] == want and set(bits[len(want):]) <= {"0"}
        else:
            assert bits == want
        ones += want.count("1")
    assert ones > 0


def test_self_complementary_kmers_by_brute_force(oracle_built, tmp_path):
    """Even k: a k-mer equal to its own reverse complement counts once -- unless its CASE is asymmetric, then the reference's character
    comparison does not see the equality and counts it twice.  Oracle CLI and (when built) the real reference against the brute force."""
    k = 6
    rng = random.Random(3)
    pal = "ACGCGT"                                            # its own reverse complement
    body = "".join(rng.choice("ACGT") for _ in range(300))
    text = body[:100] + pal + body[100:200] + pal + body[200:] + pal
    ms = "".join(c if rng.random() < 0.6 else c.lower() for c in text)
    ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    prefix = str(tmp_path / "pal")
    oracle_py.Oracle.construct(ms, k, use_klcp=False).dump(prefix)
    counts = occurrence_counts(ms, k)
    assert counts[pal][1] >= 3
    reads = [pal, pal.lower(), "acgCGT", "ACGcgt", "AcGcGt", "TT" + pal + "GG" + "acGCGt" + "A"]
    stdin = "".join(f">r{i}\n{r}\n" for i, r in enumerate(reads)).encode()
    clis = [oracle_py.ORACLE_CLI] + ([oracle_py.REF_CLI] if oracle_py.ref_available() else [])
    for fn in ("xor", "and", "1-3", "4-100"):
        want = "".join(f"r{i}\t{brute_force_fn_line(r, counts, k, fn)}\n" for i, r in enumerate(reads))
        for cli in clis:
            rc, out, err = oracle_py.run_cli(cli, ["query", "-f", fn, "-q", "-", prefix], stdin)
            assert rc == 0 and out.decode() == want, (fn, cli)
    # the rule makes a difference on these reads: symmetric and asymmetric case answer differently for some function
    assert brute_force_fn_line("acgCGT", counts, k, "1-3") != brute_force_fn_line(pal, counts, k, "1-3") or \
        brute_force_fn_line("acgCGT", counts, k, "xor") != brute_force_fn_line(pal, counts, k, "xor")


@pytest.mark.parametrize("name,fname,k,fn", FN_RUNS)
def test_oracle_cli_general_functions_against_brute_force(oracle_built, name, fname, k, fn):
    check_fn_cli(lambda args, stdin: oracle_py.run_cli(oracle_py.ORACLE_CLI, args, stdin), name, fname, k, fn)


@pytest.mark.skipif(not oracle_py.ref_available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,fname,k,fn", FN_RUNS[:4])
def test_reference_cli_general_functions_against_brute_force(name, fname, k, fn):
    check_fn_cli(lambda args, stdin: oracle_py.run_cli(oracle_py.REF_CLI, args, stdin), name, fname, k, fn)


@pytest.mark.gpu
@pytest.mark.parametrize("name,fname,k,fn", FN_RUNS)
def test_product_cli_general_functions_against_brute_force(product_built, name, fname, k, fn):
    fm = product_built
    check_fn_cli(lambda args, stdin: fm.run_cli(args, stdin), name, fname, k, fn)


@pytest.mark.gpu
def test_product_self_complementary_kmers_by_brute_force(product_built, oracle_built, tmp_path):
    fm = product_built
    k = 6
    rng = random.Random(3)
    pal = "ACGCGT"
    body = "".join(rng.choice("ACGT") for _ in range(300))
    text = body[:100] + pal + body[100:200] + pal + body[200:] + pal
    ms = "".join(c if rng.random() < 0.6 else c.lower() for c in text)
    ms = ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower()
    prefix = str(tmp_path / "pal")
    oracle_py.Oracle.construct(ms, k, use_klcp=False).dump(prefix)
    counts = occurrence_counts(ms, k)
    reads = [pal, pal.lower(), "acgCGT", "ACGcgt", "AcGcGt", "TT" + pal + "GG" + "acGCGt" + "A"]
    stdin = "".join(f">r{i}\n{r}\n" for i, r in enumerate(reads)).encode()
    for fn in ("xor", "and", "1-3", "4-100"):
        want = "".join(f"r{i}\t{brute_force_fn_line(r, counts, k, fn)}\n" for i, r in enumerate(reads))
        rc, out, err = fm.run_cli(["query", "-f", fn, "-q", "-", prefix], stdin)
        assert rc == 0 and out.decode() == want, fn

