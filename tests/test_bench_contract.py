"""The record `bench.py` produces, checked on what a default run of the committed build produced on an MI355X
(profiles/r<round>_bench_full_*.json = bench_full.json of that run): the driver's contract fields, BASELINE.json's metric, the
two objects the hot-path tier asks for (`roofline`, `cpu_baseline`), a `per_config` entry for every other BASELINE
configuration -- and the ONE LINE of it that goes to stdout (bench.compact): short enough for the driver's parser
(round 3's 24.5 KB line came back `parsed: null`), at 1 and at 8 ranks."""
import copy
import glob
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def latest_file():
    files = glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_bench_full_*.json"))
    assert files, "no committed bench record under profiles/"
    return max(files, key=lambda f: (int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)), os.path.getmtime(f)))


def latest_line():
    with open(latest_file()) as f:
        return json.load(f)


def record_round():
    return int(re.match(r"r(\d+)_", os.path.basename(latest_file())).group(1))


def stdout_line(d):
    import bench
    return bench.compact(d)


def test_the_stdout_line_is_short_at_1_and_8_ranks():
    import bench
    d = latest_line()
    line = stdout_line(d)
    assert "per_config" not in line and list(line.keys())[-1] == "per_config_summary"
    assert len(json.dumps(line)) < bench.LINE_LIMIT <= 12000
    # the same line as 8 ranks print it: 8 bases of a 4 GB container, per-rank NUMA fields, world-size figures
    d8 = copy.deepcopy(d)
    d8["n_gpus"] = 8
    d8["rank_bases"] = [473_000_000 * r for r in range(8)]
    d8["rccl_ranks_seen"] = 8
    d8["container_bytes_all_ranks"] = 8 * 473_000_000
    d8["host"].update({"numa_node": 1, "cpus_pinned": 96, "pci": "0000:c5:00.0"})
    d8["compress"]["layout_gather"] = "rccl-c-abi (sprintz_mi355x_gather_layout: ncclAllGather, in-stream)"
    for e in d8["per_config"]:
        e["job"] = {"raw_bytes_all_ranks": 8192000000.0, "ratio": 2.9971, "decompress_ms_max_rank": 0.7123, "decompress_MBps": 11500000.0,
                    "compress_ms_max_rank": 1.2345, "compress_MBps": 6640000.0, "n_gpus": 8}
    assert len(json.dumps(stdout_line(d8))) < bench.LINE_LIMIT
    # every key the driver and the judge read survives the compaction
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "ratio", "kernel_ms", "roofline", "cpu_baseline", "compress", "rccl_ranks_seen", "per_config_summary"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k


def test_data_sweep_is_on_the_line():
    """SURVEY 8d's other generators on the headline shape (round 4 on)"""
    if record_round() < 4:
        import pytest
        pytest.skip("the committed record predates data_sweep")
    s = stdout_line(latest_line())["data_sweep"]
    for kind in ("uniform", "walk300", "walkflat"):
        dec_ms, dec_frac, enc_ms, enc_frac, ratio = s[kind]
        assert dec_ms > 0 and enc_ms > 0 and 0 < dec_frac <= 1 and 0 < enc_frac <= 1
    assert 0.96 < s["uniform"][4] < 0.98           # 10 240 B in, 10 568 B out (SURVEY 8d's worked example)
    assert 1.4 < s["walk300"][4] < 1.6


def test_contract_fields():
    d = latest_line()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert d["metric"] == base["metric"]
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "MB/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "u16"
    assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["data"].startswith("synthetic")
    # value = decompressed bytes of the whole job / wall time of the timed steps
    raw = 131072 * 5120 * 2 * d["n_gpus"]
    assert abs(d["value"] - raw / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01


def test_roofline_and_cpu_baseline():
    d = latest_line()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.3 < r["frac"] <= 1.0
    if "frac_kernel_events" in r:
        # round 6 on: achieved / frac follow from the line's own ms_per_step (the clock the driver re-derives them with);
        # the HIP-event average of the same launches rides beside them
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (d["ms_per_step"] * 1e-3) / 1e9) / r["achieved"] < 0.01
        assert abs(r["achieved_kernel_events"] - r["algorithmic_bytes_per_launch"] / (d["kernel_ms"] * 1e-3) / 1e9) / r["achieved_kernel_events"] < 0.01
        assert abs(r["frac_kernel_events"] - r["achieved_kernel_events"] / r["peak"]) < 1e-3
        assert "configs_8gpu_sharding" in d and "layout_gather_fallback" in d and len(d["ms_per_step_ranks"]) == d["n_gpus"]
    else:
        # (records up to round 5) achieved = algorithmic bytes per launch / the HIP-event launch duration
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (d["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    if r["traffic"] is not None:
        assert 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.5      # no wasted re-reads
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "value_1thread", "physical_cores", "logical_cpus", "value_cache_resident"):
        assert k in c, k
    assert c["kind"] == "reference" and c["cores"] >= 1 and c["value"] > 0
    # the many-thread leg runs in C (oracle/mt_bench.c): one thread per PHYSICAL core, and it must beat one thread by far more
    # than the 4x the interpreter-lock-bound harness of round 2 managed on 256 threads
    # ... and never more threads than the container's CPU quota allows (threads_limit says which bound applied)
    assert 1 <= c["cores"] <= c["physical_cores"] <= c["logical_cpus"]
    assert c["value"] > 0.4 * min(c["cores"], 16) * c["value_1thread"]


def test_every_baseline_configuration_has_an_entry():
    d = latest_line()
    names = [v["name"] for v in d["per_config"]]
    for want in ("cfg1", "cfg3_1k", "cfg3_10k", "cfg4_10000", "cfg4_80000", "cfg4_800000", "cfg5"):
        assert want in names, want
    for v in d["per_config"]:
        if v["name"] == "cfg2":
            continue
        assert v["decompress_ms"] > 0 and v["compress_ms"] > 0, v["name"]
        assert 0 < v["roofline"]["frac"] <= 1.0, v["name"]
        assert v["cpu_baseline"]["kind"] == "reference", v["name"]
        if v["name"].startswith("cfg4_"):
            assert "Huff0" in v["entropy_stage"] and v["huff0_decode_ms"] > 0
            # algorithmic bytes of the chain = Huff0 blocks in + samples out + offset tables: NOT the Sprintz streams
            # that cross HBM between the two stages -- those show up as traffic_ratio_by_design
            r = v["roofline"]
            assert r["algorithmic_bytes_per_launch"] < v["raw_bytes"] * (1 + 1 / 2.5)
            assert 1.3 < r["traffic_ratio_by_design"] < 1.8


def test_the_tail_of_the_line_names_every_configuration():
    """the driver keeps the last 2 000 characters of the line: per_config_summary is the LAST key and fits in them"""
    d = stdout_line(latest_line())
    assert list(d.keys())[-1] == "per_config_summary"
    tail = json.dumps(d)[-2000:]
    s = d["per_config_summary"]
    for name in ("cfg2", "cfg1", "cfg3_1k", "cfg3_10k", "cfg4_10000", "cfg4_80000", "cfg4_800000", "cfg5"):
        assert name in s and f'"{name}"' in tail, name
        dec_ms, dec_frac = s[name][0], s[name][1]
        assert dec_ms > 0 and 0 < dec_frac <= 1
    assert len(json.dumps(s)) < 1900
