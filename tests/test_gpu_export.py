"""GPU tests of `fmsi export` (fmsi_hip_index_export / fmsi_cli_export, counterpart of export_ms + ms_export, reference
src/fms_index.h:462-473, src/main.cpp:508-541).  The real reference's export of every golden index equals the masked
superstring the index was built from (asserted when the fixtures are generated, tests/golden/make_golden.py step 3c),
so the superstring files are the goldens; beyond them the round trip index -> export must be the identity."""
import random

import numpy as np
import pytest

import oracle_py
from conftest import MS_FILES, index_prefix, ms_of, synthetic_ms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(product_built):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device visible")
    return product_built


@pytest.mark.parametrize("name", list(MS_FILES))
def test_export_cli_equals_reference(fm, name):
    rc, out, err = fm.run_cli(["export", index_prefix(name)])
    assert rc == 0, err
    assert out.decode() == ">exported masked superstring\n" + ms_of(name) + "\n"


def test_export_cli_errors(fm, tmp_path):
    rc, out, err = fm.run_cli(["export", "-h"])
    assert rc == 0 and b"Usage:   fmsi export <index-prefix>" in err
    rc, out, err = fm.run_cli(["export", str(tmp_path / "nope")])
    assert rc == 1 and b"index not correctly loaded" in err and out == b""


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 127, 128, 129, 4096, 100_003, 2_000_000])
def test_export_round_trip(fm, n):
    k = min(5, n)
    ms = synthetic_ms(n, k, seed=n, mask="mixed")
    dev = fm.DeviceIndex.build(ms.encode(), k)
    assert dev.export().decode() == ms


def test_export_round_trip_repetitive_and_file_index(fm, oracle_built, tmp_path):
    rng = random.Random(4)
    unit = "".join(rng.choice("ACGT") for _ in range(300))
    ms = "".join(ch if rng.random() < 0.5 else ch.lower() for ch in unit * 200 + "A" * 5000 + "ACGT" * 3000)
    ms = ms[:-8] + ms[-8:].lower()
    dev = fm.DeviceIndex.build(ms.encode(), 9)
    assert dev.export().decode() == ms
    # an index that went through the reference's file format (written by the oracle) exports the same text
    orc = oracle_py.Oracle.construct(ms, 9, use_klcp=False)
    orc.dump(str(tmp_path / "x"))
    assert fm.DeviceIndex.from_files(str(tmp_path / "x")).export().decode() == ms
    if oracle_py.REF_CLI.exists():      # and so does the real reference from the files the product wrote
        dev.write_files(str(tmp_path / "y"))
        rc, out, err = oracle_py.run_cli(oracle_py.REF_CLI, ["export", str(tmp_path / "y")])
        assert rc == 0 and out.decode() == ">exported masked superstring\n" + ms + "\n"


def test_export_at_scale(fm):
    """200 M bases: the text the builder kept in HBM comes back from the LF walk; all k-mers ON => upper case but the tail."""
    n, seed, k = 200_000_000, 77, 31
    dev = fm.DeviceIndex.build_synthetic(n, seed, k)
    got = np.frombuffer(dev.export(), dtype=np.uint8)
    assert got.size == n
    want = np.frombuffer(fm.synthetic_text(n, seed, k), dtype=np.uint8)
    assert np.array_equal(got, want)
