"""Differential runs of the whole CLI against the real reference (oracle/_ref; the oracle's CLI where that is not built): random
masked superstrings -- any k from 1 to 70, repeats, reverse-complemented copies, poly-A stretches, random masks -- indexed by the
oracle's restatement of `fmsi index`, random query files of every record shape, every mode (`query`, `-O`, `-S`, `lookup`,
`-f and|xor|r-s`) under random GPU-side options (batch size, threads, tables, two handles).  stdout must be the reference's, byte for byte;
every third superstring also goes through `fmsi index` on the device (same files as the restatement's, which the reference's pin) and `fmsi export`.
tools/fuzz_cli.py runs the same generator for as long as one likes."""
import os
import random

import pytest

import oracle_py
from conftest import bgzf_bytes, messy_fasta, revcomp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(product_built):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests need an MI355X (the library has no CPU fallback)")
    return product_built

MODES = [["query"], ["query", "-O"], ["query", "-S"], ["query", "-S", "-O"], ["lookup"], ["lookup", "-S"], ["query", "-f", "xor"], ["query", "-f", "and"],
         ["query", "-f", "1-2"], ["query", "-S", "-f", "0-0"]]


def random_superstring(rng: random.Random, k: int) -> str:
    n = rng.choice([k + 1, k + 5, 60, 300, 1500, 6000])
    n = max(n, k + 1)
    alphabet = rng.choice(["ACGT", "ACGT", "AC", "A", "ACGTT"])
    base = "".join(rng.choice(alphabet) for _ in range(n))
    pieces = [base]
    for _ in range(rng.randrange(0, 6)):
        L = rng.randrange(1, max(2, min(len(base), 400)))
        p = rng.randrange(0, len(base) - L + 1)
        piece = base[p:p + L]
        pieces.append(revcomp(piece) if rng.random() < 0.5 else piece)
        if rng.random() < 0.2:
            pieces.append(rng.choice("ACGT") * rng.randrange(1, 3 * k + 2))
    text = "".join(pieces)
    on = rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])
    ms = "".join(ch if rng.random() < on else ch.lower() for ch in text)
    return ms[: len(ms) - (k - 1)] + ms[len(ms) - (k - 1):].lower() if k > 1 else ms


def one_case(fm, tmp_path, seed: int, checker):
    rng = random.Random(seed)
    k = rng.choice([1, 2, 3, 4, 7, 11, 13, 16, 21, 31, 32, 33, 40, 63, 64, 65, 70])
    ms = random_superstring(rng, k)
    prefix = str(tmp_path / f"fz{seed}")
    use_klcp = k <= 64
    oracle_py.Oracle.construct(ms, k, use_klcp=use_klcp).dump(prefix)
    if seed % 3 == 0:   # `fmsi index` on the device writes the same bytes, and `fmsi export` gives the superstring back
        fa = tmp_path / f"ms{seed}.fa"
        fa.write_text(f">ms\n{ms}\n")
        rc, out, err = fm.run_cli(["index", "-k", str(k), str(fa)])
        assert rc == 0, (seed, k, err.decode()[-400:])
        for part in ("ac_gt", "ac", "gt", "mask", "misc") + (("klcp",) if use_klcp else ()):
            assert (tmp_path / f"ms{seed}.fa.fmsi.{part}").read_bytes() == (tmp_path / f"fz{seed}.fmsi.{part}").read_bytes(), (seed, k, len(ms), part)
        rc, out, err = fm.run_cli(["export", str(fa)])
        assert rc == 0 and out.decode() == ">exported masked superstring\n" + ms + "\n", (seed, k, len(ms))
    q = tmp_path / f"fz{seed}.fa"
    src = ms if len(ms) > 800 else ms * (800 // len(ms) + 2)      # (messy_fasta draws reads of up to 700 bases)
    data = messy_fasta(src, rng, fastq_at=rng.choice([None, None, 150, 0]))
    packing = rng.choice(["plain", "plain", "bgzf", "gzip"])
    if packing == "bgzf":
        data = bgzf_bytes(data, block=rng.choice([300, 2000, 0xFF00]))
    elif packing == "gzip":
        import gzip
        data = gzip.compress(data)
    q.write_bytes(data)
    # odd seeds: also files this small go through the block-parallel readers (mapped plain files, block-gzip members, and -- pgunzip.h -- the
    # one member of an ordinary gzip file, inflated in many chunks on two thirds of those seeds)
    os.environ.pop("FMSI_HIP_PGZ_CHUNK_BYTES", None)
    if seed % 2:
        os.environ["FMSI_HIP_READER_MIN_BYTES"] = "1"
        chunk = random.Random(seed ^ 0x5A17).choice([None, "200", "1500"])
        if chunk:
            os.environ["FMSI_HIP_PGZ_CHUNK_BYTES"] = chunk
    else:
        os.environ.pop("FMSI_HIP_READER_MIN_BYTES", None)
    modes = [m for m in MODES if use_klcp or "-S" not in m]
    for mode in rng.sample(modes, 3):
        opts = []
        if rng.random() < 0.5:
            opts += ["--batch", str(rng.choice([1024, 3000, 20000]))]
        if rng.random() < 0.3:
            opts += ["--threads", str(rng.choice([1, 3, 7]))]
        if rng.random() < 0.4:
            opts += ["--tables", rng.choice(["full", "compact", "candidates", "auto", "background", "background"])]   # (background: the switch to the dictionary falls on any batch)
        if rng.random() < 0.25:
            opts += ["--devices", "0,0"]
        rc, want, err = oracle_py.run_cli(checker, [*mode, "-q", str(q), prefix])
        if rc != 0:     # the reference itself dies on this input (an sdsl assertion in its as-shipped build): nothing to compare with
            assert b"Assertion" in err, (seed, mode, err)
            continue
        rc, got, err = fm.run_cli([*mode, *opts, "-q", str(q), prefix])
        assert rc == 0, (seed, mode, opts, err.decode()[-600:])
        assert got == want, (seed, k, len(ms), mode, opts, packing)
    os.environ.pop("FMSI_HIP_READER_MIN_BYTES", None)
    os.environ.pop("FMSI_HIP_PGZ_CHUNK_BYTES", None)


@pytest.mark.parametrize("seed", range(12))   # (tools/fuzz_cli.py runs as many as one likes: 5 575 seeds were clean over rounds 2-4; the suite has a time limit)
def test_cli_against_reference_on_random_inputs(fm, oracle_built, tmp_path, seed):
    checker = oracle_py.REF_CLI if oracle_py.ref_available() else oracle_py.ORACLE_CLI
    one_case(fm, tmp_path, 9000 + seed, checker)
