"""Host-side pieces of the bench record (gt4py_dwarf_p_cloudsc2_tl_ad_amd/benchlib.py + bench.py), checked without a GPU: the
roofline arithmetic (algorithmic words per column x itemsize x columns / time / 8 TB/s), the PMC-traffic lookup
with the gfx950 FETCH_SIZE correction, the record skeleton of the driver's contract, the single-process forms of the N-rank
end-of-run protocol, and what `bench.py` takes as "the host's cores"."""
import argparse
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gt4py_dwarf_p_cloudsc2_tl_ad_amd import benchlib  # noqa: E402


def test_roofline_entry_is_algorithmic_bytes_over_time_over_the_hbm_peak():
    r = benchlib.roofline_entry("cs2::nl_ring_kernel", benchlib.NL_WORDS_PER_COL, 8, 65536, "double", 0.300)
    assert r["bytes_per_launch"] == 3567 * 8 * 65536 == 1870135296            # 28 536 B per column
    assert r["achieved"] == pytest.approx(1870135296 / 0.300e-3 / 1e9) and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["frac"] == pytest.approx(r["achieved"] / 8000.0) and 0.77 < r["frac"] < 0.79
    assert r["bound"] == "hbm" and r["dtype"] == "f64" and r["columns"] == 65536
    assert r["kernel_columns_per_s"] == pytest.approx(65536 / 0.300e-3)
    t = benchlib.roofline_entry("cs2::ad_kernel", benchlib.TLAD_WORDS_PER_COL, 4, 524288, "single", 3.28, placement={"mode": "tuned"})
    assert t["bytes_per_launch"] == 7134 * 4 * 524288 and t["dtype"] == "f32" and t["placement"] == {"mode": "tuned"}


def test_pmc_traffic_doubles_fetch_size_and_reads_the_newest_committed_round():
    """MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE tallies a wide streaming read at half its size on gfx950 -> x 2;
    WRITE_SIZE is exact; both in KiB.  The lookup takes the newest round's summary under profiles/ that has the kernel."""
    traffic, src = benchlib.pmc_traffic("nl_ring_kernel", 65536, "double")
    assert src and "profiles/r04/" in src and "FETCH_SIZE" in src
    pm = json.load(open(os.path.join(ROOT, src.split(", ")[1].split(":")[0])))
    k = [v for n, v in pm.items() if "nl_ring_kernel" in n][0]
    want = 2.0 * k["FETCH_SIZE"]["mean_per_dispatch"] * 1024 + k["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    assert traffic == pytest.approx(want) and 1.0 <= traffic / 1870135296 < 1.05          # 1.02 x the algorithmic bytes
    ad, _ = benchlib.pmc_traffic("ad_kernel", 524288, "single")
    assert 1.28 < ad / (7134 * 4 * 524288) < 1.35                                           # the recompute design's 1.31 x
    assert benchlib.pmc_traffic("nl_ring_kernel", 12345, "double") == (None, None)          # no committed summary: null


def test_base_record_carries_the_contract_keys():
    args = argparse.Namespace(config=2, steps=20, warmup=5, precision="double")
    d = benchlib.base_record(args, 8, 65536, 137, value=1.0, ms_per_step=2.0, ranks=8, backend="nccl (RCCL)")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["columns_total"] == 8 * 65536 and "model" not in d["config"] and "configs[1]" in d["config"]["workload"]
    args5 = argparse.Namespace(config=5, steps=20, warmup=5, precision="single")
    d5 = benchlib.base_record(args5, 8, 524288, 137, value=1.0, ms_per_step=2.0, ranks=8, backend="nccl (RCCL)")
    assert d5["scaling"] == "strong" and d5["dtype"] == "f32" and d5["config"]["columns_total"] == benchlib.CONFIG5_COLUMNS


def test_single_process_forms_of_the_end_of_run_protocol():
    assert benchlib.gather_rank_reports(None, 1, {"rank": 0}) == [{"rank": 0}]
    calls = []
    assert benchlib.rank0_then_everyone(None, 0, lambda: calls.append(1) or "record") == "record" and calls == [1]


def test_host_cpu_share_reads_affinity_and_cgroup_quota(tmp_path, monkeypatch):
    import bench

    aff, quota = bench.host_cpu_share()
    assert aff == len(os.sched_getaffinity(0)) and (quota is None or quota > 0)
    # a cgroup v2 `cpu.max` of "1600000 100000" is a 16-core quota (what the builder's GPU boxes report beside a 256-core mask)
    fake = tmp_path / "cpu.max"
    fake.write_text("1600000 100000\n")
    real_open = open

    def fake_open(path, *a, **k):
        return real_open(str(fake) if path == "/sys/fs/cgroup/cpu.max" else path, *a, **k)

    monkeypatch.setattr("builtins.open", fake_open)
    assert bench.host_cpu_share() == (aff, 16.0)
    fake.write_text("max 100000\n")
    assert bench.host_cpu_share() == (aff, None)
