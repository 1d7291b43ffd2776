"""CPU tests that PIN the oracle (oracle/fmsi_oracle.c) to the reference:
  - the reference's known-answer tables (tests/fms_index_test.h -> tests/golden/kat_fms_index.json),
  - index bytes produced by the real reference (tests/golden/idx, md5 in SURVEY.md 8(c)),
  - the reference CLI's stdout on every golden query file, in all six modes,
  - when oracle/_ref is present (build container, or travelled to the GPU box): the real reference's
    functions on random inputs.
"""
import hashlib
import json
import random

import numpy as np
import pytest

import oracle_py
from conftest import (CLI_DATASETS, DATASETS, FN_CASES, GOLDEN, golden_fn_out, INDEX_CASES, INDEX_PARTS, MODES, golden_index_part, golden_out, index_prefix, mixed_kmers,
                      query_file, synthetic_ms)


def make(cls, kat, name):
    d = kat["indexes"][name]
    return cls.literal(d["ac_gt"], d["ac"], d["gt"], d["mask"], d["counts"], d["dollar"], d["klcp"], k=0)


def impls():
    out = [oracle_py.Oracle]
    if oracle_py.ref_available():
        out.append(oracle_py.Reference)
    return out


@pytest.mark.parametrize("cls", impls(), ids=lambda c: c.__name__)
class TestKnownAnswers:
    def test_rank(self, oracle_built, kat, cls):
        for name, i, c, want in kat["rank"]:
            assert make(cls, kat, name).rank(i, c) == want

    def test_update_range(self, oracle_built, kat, cls):
        for name, i, j, c, wi, wj in kat["update_range"]:
            assert make(cls, kat, name).update_range(i, j, c) == (wi, wj)

    def test_extend_range_with_klcp(self, oracle_built, kat, cls):
        for name, i, j, wi, wj in kat["extend_range_with_klcp"]:
            assert make(cls, kat, name).extend_range_with_klcp(i, j) == (wi, wj)

    def test_get_range_with_pattern(self, oracle_built, kat, cls):
        for name, pat, wi, wj in kat["get_range_with_pattern"]:
            assert make(cls, kat, name).get_range_with_pattern(pat) == (wi, wj)

    def test_kmer_order_if_present(self, oracle_built, kat, cls):
        for name, a, b, want in kat["kmer_order_if_present"]:
            assert make(cls, kat, name).kmer_order_if_present(a, b) == want

    def test_query_kmers_streaming(self, oracle_built, kat, cls):
        ix = make(cls, kat, "dummy3")   # one index for the whole table, as in the reference test
        for name, q, k, mo, want in kat["query_kmers_streaming"]:
            assert ix.query_kmers_streaming(q, k, maximized_ones=mo) == want
        ix = make(cls, kat, "dummy3")
        for name, q, k, mo, want in kat["query_kmers_streaming_orders"]:
            assert ix.query_kmers_streaming(q, k, maximized_ones=mo, orders=True) == want

    def test_query(self, oracle_built, kat, cls):
        ixs = {}
        for name, q, k, want in kat["query_orders"]:
            ix = ixs.setdefault(("o", name), make(cls, kat, name))
            assert ix.query_kmers(q, k, orders=True) == want
        ixs = {}
        for name, q, want in kat["query"]:
            ix = ixs.setdefault(name, make(cls, kat, name))
            assert ix.query_kmers(q, len(q)) == want


def test_construct_kat(oracle_built, kat):
    c = kat["construct"]
    ix = oracle_py.Oracle.construct(c["ms"], c["k"], use_klcp=False)
    want = kat["indexes"][c["want"]]
    for which, key in ((0, "ac_gt"), (1, "ac"), (2, "gt")):
        words, nbits = ix.words(which)
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:nbits]
        assert bits.tolist() == want[key]
    assert ix.mask_bits().tolist() == want["mask"]
    assert ix.counts() == want["counts"] and ix.dollar() == want["dollar"]
    for ms, k, bits in kat["construct_klcp"]:
        ix = oracle_py.Oracle.construct(ms, k, use_klcp=True)
        words, nbits = ix.words(3)
        assert np.unpackbits(words.view(np.uint8), bitorder="little")[:nbits].tolist() == bits


@pytest.mark.parametrize("name", list(INDEX_CASES))
def test_index_bytes_equal_reference(oracle_built, tmp_path, name):
    """oracle construct()+dump == bytes written by the real reference (tests/golden/idx, md5.json), incl. k > 32 and -x."""
    md5 = json.loads((GOLDEN / "idx" / "md5.json").read_text())
    src, extra = INDEX_CASES[name]
    dst = tmp_path / name
    dst.write_bytes((GOLDEN / src).read_bytes())
    rc, _, err = oracle_py.run_cli(oracle_py.ORACLE_CLI, ["index", *extra, str(dst)])
    assert rc == 0, err
    for part in INDEX_PARTS:
        want = golden_index_part(name, part)
        f = tmp_path / f"{name}.fmsi.{part}"
        if want is None:
            assert not f.exists()
            continue
        assert f.read_bytes() == want, part
        assert hashlib.md5(want).hexdigest() == md5[f"{name}.fmsi.{part}"], part


def test_survey_md5_of_k31_fixture():
    """The values recorded from the real reference CLI in SURVEY.md 8(c)."""
    md5 = json.loads((GOLDEN / "idx" / "md5.json").read_text())
    assert md5["sars_k31.fmsi.ac_gt"].startswith("4d4d4a26") and md5["sars_k31.fmsi.ac_gt"].endswith("f54a")
    assert md5["sars_k31.fmsi.ac"].startswith("86c3b8fc") and md5["sars_k31.fmsi.gt"].startswith("130bfcbd")
    assert md5["sars_k31.fmsi.klcp"].startswith("87c542a9") and md5["sars_k31.fmsi.mask"].startswith("59d222ee")
    assert (GOLDEN / "idx" / "sars_k31.fmsi.misc").read_text().split() == ["27616", "1", "9600", "15466", "20960", "31"]
    assert (GOLDEN / "idx" / "integration_a.fmsi.misc").read_text().split() == ["2", "1", "3", "5", "7", "3"]
    assert (GOLDEN / "idx" / "integration_a.fmsi.mask").stat().st_size == 91


CASES = [(n, kind, mode) for n, (k, kinds) in CLI_DATASETS.items() for kind in kinds for mode in MODES]


@pytest.mark.parametrize("name,kind,mode", CASES)
def test_cli_output_equals_reference(oracle_built, name, kind, mode):
    k, _ = CLI_DATASETS[name]
    args, _, _, _ = MODES[mode]
    rc, out, err = oracle_py.run_cli(oracle_py.ORACLE_CLI, [*args, "-k", str(k), "-q", str(query_file(name, kind)), index_prefix(name)])
    assert rc == 0, err
    assert out == golden_out(name, kind, mode)


@pytest.mark.parametrize("name,kind,fn", FN_CASES)
def test_cli_general_functions_equal_reference(oracle_built, name, kind, fn):
    """`query -f and|xor|r-s` (query_mode::general, src/fms_index.h:316-327): the restatement against the real reference's output."""
    k, _ = CLI_DATASETS[name]
    rc, out, err = oracle_py.run_cli(oracle_py.ORACLE_CLI, ["query", "-f", fn, "-k", str(k), "-q", str(query_file(name, kind)), index_prefix(name)])
    assert rc == 0, err
    assert out == golden_fn_out(name, kind, fn)
    if fn == "xor":   # -S changes nothing in this mode (:336), -O is ignored with a warning (src/main.cpp:273-276)
        rc, out2, err = oracle_py.run_cli(oracle_py.ORACLE_CLI, ["query", "-S", "-f", fn, "-O", "-q", str(query_file(name, kind)), index_prefix(name)])
        assert rc == 0 and out2 == out and b"WARNING: Parameter -O is ignored" in err


def test_function_names_parse_like_the_reference(oracle_built):
    """mask_function(), src/functions.h:23-57: which names are functions, and what the unknown ones print."""
    q, prefix = str(query_file("integration_b", "queries")), index_prefix("integration_b")
    for name in ("nand", "5", "1-", "-1-2", "1-x", "a-3", "1--2", ""):
        for cli in ([oracle_py.ORACLE_CLI] + ([oracle_py.REF_CLI] if oracle_py.ref_available() else [])):
            rc, out, err = oracle_py.run_cli(cli, ["query", "-f", name, "-q", q, prefix])
            assert rc == 1 and out == b"" and f"ERROR: Function '{name}' not recognized.".encode() in err, (cli, name, err)
    for name, same_as in (("or", []), ("all", ["-O"]), ("0-1000", ["-f", "0-2000"]), ("007-9", ["-f", "7-9"])):
        rc, out, err = oracle_py.run_cli(oracle_py.ORACLE_CLI, ["query", "-f", name, "-q", q, prefix])
        rc2, out2, _ = oracle_py.run_cli(oracle_py.ORACLE_CLI, ["query", *same_as, "-q", q, prefix])
        assert rc == 0 and rc2 == 0 and out == out2, name
    rc, out, err = oracle_py.run_cli(oracle_py.ORACLE_CLI, ["lookup", "-f", "xor", "-q", q, prefix])
    assert rc == 1 and b"not allowed with f-masked superstrings" in err


def test_reference_goldens_unchanged():
    """The reference's own committed goldens (tests/testfiles/result_*_complements*.txt)."""
    rd = GOLDEN / "ref_data"
    assert golden_out("integration_a", "queries", "or") == (rd / "result_a_complements.txt").read_bytes()
    assert golden_out("integration_a", "queries", "lookup") == (rd / "result_a_complements_hash.txt").read_bytes()
    assert golden_out("integration_b", "queries", "or") == (rd / "result_b_complements.txt").read_bytes()
    assert golden_fn_out("integration_b", "queries", "xor") == (rd / "result_b_complements_xor.txt").read_bytes()


def test_rrr_roundtrip_densities(oracle_built, tmp_path):
    """RRR-63 encode (oracle) -> file -> access/rank, incl. inverted superblocks, n % 63 == 0 and tiny n."""
    rng = random.Random(7)
    for n in (1, 62, 63, 64, 126, 2016, 2017, 4032, 20001):
        for dens in (0.0, 0.02, 0.5, 0.97, 1.0):
            bits = [1 if rng.random() < dens else 0 for _ in range(n)]
            ix = oracle_py.Oracle.literal([0] * n, [0] * n, [], bits, [1, 1, 1, 1], 0, [], k=3)
            got = ix.mask_bits().tolist()
            assert got == bits, (n, dens)
            pref = np.concatenate([[0], np.cumsum(bits)])
            for i in random.Random(n).sample(range(n + 1), min(n + 1, 50)):
                assert ix.mask_rank(i) == pref[i]


@pytest.mark.skipif(not oracle_py.ref_available(), reason="oracle/_ref not built")
def test_oracle_equals_real_reference_on_random_input(oracle_built, tmp_path):
    """Same index files, same seeded queries: oracle vs the real reference, function by function."""
    k = 15
    ms = synthetic_ms(60000, k, seed=11, mask="mixed")
    p = tmp_path / "ms.fa"
    p.write_text(">ms\n" + ms + "\n")
    rc, _, err = oracle_py.run_cli(oracle_py.REF_CLI, ["index", "-k", str(k), str(p)])
    assert rc == 0, err
    orc = oracle_py.Oracle.load(str(p), use_klcp=True)
    ref = oracle_py.Reference.load(str(p), use_klcp=True)
    rng = random.Random(3)
    n = orc.n
    assert n == ref.n == len(ms) + 1
    for _ in range(3000):
        i = rng.randrange(0, n + 1)
        for c in range(4):
            assert orc.rank(i, c) == ref.rank(i, c)
        assert orc.rank1(0, i) == ref.rank1(0, i)
        assert orc.mask_rank(i) == ref.mask_rank(i)
        if i < n:
            assert orc.mask_at(i) == ref.mask_at(i)
    for kmer in mixed_kmers(ms, k, 2000, seed=5):
        assert orc.get_range_with_pattern(kmer) == ref.get_range_with_pattern(kmer)
    reads = ["".join(mixed_kmers(ms, 60, 1, seed=s)) for s in range(200)]
    for has_klcp in (False, True):
        for mode in (oracle_py.MODE_OR, oracle_py.MODE_ALL):
            for orders in (False, True):
                orc.predictor_reset(); ref.predictor_reset()
                for r in reads:
                    assert orc.query_kmers(r, k, has_klcp, mode, orders) == ref.query_kmers(r, k, has_klcp, mode, orders)


@pytest.mark.skipif(not (oracle_py.ref_available() and oracle_py.REF_TUNED_LIB.exists()), reason="oracle/_ref not built")
def test_tuned_reference_build_answers_like_the_shipped_one(oracle_built, tmp_path):
    """bench_cpu.py's tuned_value times the reference's sources with -O3 -DNDEBUG -march=x86-64-v3 (oracle/Makefile.ref):
    the flags must not change a single answer."""
    k = 17
    ms = synthetic_ms(40000, k, seed=23, mask="mixed")
    p = tmp_path / "ms.fa"
    p.write_text(">ms\n" + ms + "\n")
    rc, _, err = oracle_py.run_cli(oracle_py.REF_CLI, ["index", "-k", str(k), str(p)])
    assert rc == 0, err
    ref = oracle_py.Reference.load(str(p))
    tuned = oracle_py.Reference.load_tuned(str(p))
    kmers = mixed_kmers(ms, k, 20000, seed=9)
    raw = "".join(kmers).encode()
    for mode in (oracle_py.MODE_OR, oracle_py.MODE_ALL):
        _, a = ref.time_isolated_kmers(raw, len(kmers), k, mode)
        _, b = tuned.time_isolated_kmers(raw, len(kmers), k, mode)
        assert np.array_equal(a, b) and a.any() and not a.all()
    ref.close()
    tuned.close()
