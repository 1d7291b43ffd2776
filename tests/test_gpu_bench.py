"""bench.py as the driver runs it: the LAST stdout line must be the compact record the driver can parse (it keeps an 8-KB tail)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


@pytest.fixture
def fm(product_built):
    if product_built.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on an MI355X")
    return product_built


def test_default_bench_line_is_parseable_by_the_driver(fm):
    """`python bench.py --steps 2 --warmup 1` -- the driver's command but for the step counts and with one leg of every kind instead of
    all twenty, on an index of 5 x 10^8 rows (the suite has a time limit; the driver's own run is at 2.8 x 10^9 and carries them all): the headline, the LF-walk legs, the CPU baseline, a
    cache-resident leg and the whole-process legs incl. the query file of one k-mer per record.  The last line
    is below 4 KB, parses, and carries `roofline` (with the LF-walk legs) and `cpu_baseline`; the line before it is the full record and
    agrees with it."""
    # (the rocprofv3 --pmc grandchildren are switched off here: they rebuild every leg's handle twice more -- a minute of the suite's time --
    # and change nothing about the line's shape but `roofline.traffic`, which the driver's own run carries)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FMSI_BENCH_NO_PMC="1")
    res = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "2", "--warmup", "1", "--n", "500000000", "--legs", "c2,e2e", "--reads", "20000", "--e2e-kmer-records", "4000000", "--cpu-sample", "200000"],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=str(REPO))
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 2, [l[:80] for l in lines]
    last = lines[-1]
    assert len(last) < 4096, len(last)
    tail = (res.stdout + "\n---- stderr ----\n" + res.stderr)[-8192:]     # what the driver keeps
    assert tail.count(last) == 1 or len(res.stderr) > 4000, "the compact line does not survive the driver's 8-KB tail"
    d = json.loads(last)
    full = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 1e9 and d["unit"] == "k-mers/s"
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["queries_per_step_per_gpu"] - 1) < 1e-3
    r = d["roofline"]
    assert r and r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and r["kernel"] and r["avg_kernel_ms"] <= d["ms_per_step"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    assert [l["kernel"].split("<")[0] for l in r["lf_path"]] == ["sorted_walk_kernel", "query_pair_kernel", "query_lockstep_kernel"]
    assert all(l["kmers_per_s"] > 1e9 for l in r["lf_path"])
    assert r["lf_path"][0]["kmers_per_s"] > r["lf_path"][1]["kmers_per_s"] > r["lf_path"][2]["kmers_per_s"]
    c = d["cpu_baseline"]
    assert c and c["value"] > 0 and c["cores"] == 1 and c["kind"] == "reference" and c["matches_gpu"] is True and c["sample"]
    names = [l["name"] for l in d["configs"]]
    assert names == [l["name"] for l in full["configs"]] and "c2" in names
    assert {"e2e_or", "e2e_all", "e2e_all_S", "e2e_isolated_or", "e2e_isolated_all"} <= set(names)
    iso = {l["name"]: l for l in full["configs"] if l["name"].startswith("e2e_isolated")}
    assert all(l["kmer_batches"] > 0 and l["windows"] == 4000000 for l in iso.values())
    bad = [l for l in d["configs"] if l.get("matches_reference") is not True]
    assert not bad, bad
    assert full["value"] == pytest.approx(d["value"], rel=1e-9) and full["roofline"]["frac"] == r["frac"]
