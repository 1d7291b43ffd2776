"""64-bit rows (fmsi_wide.hip): the reference's intervals are size_t (src/fms_index.h:98-103), and an index of 2^32 rows or more
loads into the same blocks with a table of absolute counts per superblock.  No such index can be built here, so
FMSI_HIP_FORCE_ROWS64 sends the golden indexes -- and a 1.2 M-row one, 1170 superblocks of 8 blocks -- through that
instantiation: every mode against the oracle, the CLI against the reference's golden output, and the entry points that do not
exist for such handles refuse with FMSI_HIP_ERANGE instead of answering wrongly."""
import numpy as np
import pytest

import oracle_py
from conftest import CLI_DATASETS, DATASETS, GOLDEN, MODES, golden_out, index_prefix, mixed_kmers, query_file, synthetic_ms

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
        got = dev.query_kmers(packed, k, flag | fm.STRICT)
        gf, gr = fm.decode_strict(got)
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
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
    dev = fm.DeviceIndex.from_files(index_prefix(name), use_klcp=True, prefix_len=6)     # (table and kLCP are not taken)
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
    assert info.prefix_len == 0 and info.has_klcp == 0
    for call in (dev.enable_pair_layout, dev.enable_dict, dev.enable_verify, dev.export):
        with pytest.raises(fm.FmsiError) as e:
            call()
        assert e.value.code == fm.E_RANGE
    # windows and runs go through the same kernel
    up = ms.upper()[:3000]
    if len(up) > k + 10:
        words = fm.pack_sequence(up.encode())
        n = len(up) - k + 1
        wk = [up[i:i + k] for i in range(n)]
        fwd, rc = orc.ternaries_bulk("".join(wk).encode(), n, k, True)
        gf, gr = fm.decode_strict(dev.query_runs(words, len(up), [0], [n], k, fm.MODE_ALL | fm.STRICT))
        assert np.array_equal(gf, fwd) and np.array_equal(gr, rc)
        with pytest.raises(fm.FmsiError):
            dev.query_stream(words, len(up), [0], [n], k, fm.MODE_ALL | fm.STRICT)


def test_wide_rows_across_many_superblocks(fm, oracle_built, tmp_path):
    """1.2 M rows = 9 376 blocks = 1 172 superblocks of 8: the wrapped block counts are restored from the superblock table
    everywhere; mixed mask, so ORDERS and the OR-mode interval scan use the mask counts too."""
    k = 31
    ms = synthetic_ms(1_200_000, k, seed=4242, mask="mixed")
    orc = oracle_py.Oracle.construct(ms, k, use_klcp=False)
    orc.dump(str(tmp_path / "w"))
    dev = fm.DeviceIndex.from_files(str(tmp_path / "w"))
    check(fm, dev, orc, mixed_kmers(ms, k, 100_000, seed=6), k)
    # and the written files are the loaded ones (the writer works from the same blocks)
    dev.write_files(str(tmp_path / "w2"))
    for part in ("ac_gt", "ac", "gt", "mask", "misc"):
        assert open(f"{tmp_path}/w.fmsi.{part}", "rb").read() == open(f"{tmp_path}/w2.fmsi.{part}", "rb").read()


@pytest.mark.parametrize("name,kind,mode", [(n, kind, mode) for n, (k, kinds) in CLI_DATASETS.items() if k <= 32 for kind in kinds[:1]
                                            for mode in ("or", "all", "lookup")])
def test_cli_with_wide_rows(fm, name, kind, mode):
    k = CLI_DATASETS[name][0]
    rc, out, err = fm.run_cli([*MODES[mode][0], "-k", str(k), "-q", str(query_file(name, kind)), index_prefix(name)])
    assert rc == 0, err.decode()
    assert out == golden_out(name, kind, mode)
    rc, out, err = fm.run_cli([*MODES["or_S"][0], "-k", str(k), "-q", str(query_file(name, kind)), index_prefix(name)])
    assert rc == 1 and b"2^32 rows" in err
