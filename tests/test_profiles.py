"""The committed benchmark lines keep the driver's contract: the keys it reads, a roofline whose
fraction is at most 1 and can be recomputed from the committed rocprof kernel statistics and PMC
record, the CPU-baseline object, and consistency between the figures inside a line."""
import csv
import glob
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

# the newest set only: earlier sets are kept for the history of the design and predate some fields
_ALL = glob.glob(os.path.join(ROOT, "profiles", "r*_v*_bench.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*_v*_serial_bench.json")) \
    + glob.glob(os.path.join(ROOT, "profiles", "r*_v*_cfg?_*.json"))
_TAG = max((re.match(r"(r\d+)_v(\d+)_", os.path.basename(p)).groups() for p in _ALL), key=lambda t: (int(t[0][1:]), int(t[1])))
PREFIX = f"{_TAG[0]}_v{_TAG[1]}_"
LINES = sorted(set(p for p in _ALL if os.path.basename(p).startswith(PREFIX)))


def _line(path):
    rows = [l for l in open(path).read().splitlines() if l.startswith('{"metric"')]
    assert rows, path
    return json.loads(rows[-1])


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_bench_line_contract(path):
    j = _line(path)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["metric"] == "rays/s" and j["unit"] == "rays/s" and j["higher_is_better"] is True
    assert j["scaling"] in ("weak", "strong") and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    par = j["config"]["parallelism"]
    assert par["world_size"] == j["n_gpus"] and par["steps_total"] == j["frames_accumulated"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm"):
        assert key in r, key
    assert r["bound"] == "valu" and r["peak"] == 0.5 * r["simds"] * r["clock_ghz"]
    assert r["frames_in_flight"] == 1, "per-kernel durations must come from launches that had the GPU to themselves"
    assert r["frac"] is not None, "every configuration has a PMC record of its own (tools/profile_round.sh)"
    assert r["pmc_source"]["stale"] is False
    assert 0.0 < r["lane_utilisation"] <= 1.0 and abs(r["useful_frac"] - r["frac"] * r["lane_utilisation"]) < 1e-12
    for k in ("k_primary", "k_intersect"):
        pk = r["per_kernel"][k]
        assert 0.0 < pk["frac"] <= 1.0 and 0.0 < pk["lane_utilisation"] <= 1.0 and 0.0 < pk["hbm_frac"] <= 1.0
        assert abs(pk["useful_frac"] - pk["frac"] * pk["lane_utilisation"]) < 1e-12
    if r["frac"] is not None:
        assert 0.0 < r["frac"] <= 1.0 and 0.0 < r["hbm"]["frac"] <= 1.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert abs(r["achieved"] - r["valu_insts_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
        assert abs(r["hbm"]["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["hbm"]["achieved"] < 1e-9
        assert r["hbm"]["traffic_over_compulsory"] > 1.0
    per = r["per_kernel"]
    n = per["k_primary"]["launches"] + per["k_intersect"]["launches"]
    assert n == r["launches"]
    ms = per["k_primary"]["launches"] * per["k_primary"]["avg_launch_ms"] + per["k_intersect"]["launches"] * per["k_intersect"]["avg_launch_ms"]
    assert abs(ms / n - r["avg_launch_ms"]) < 1e-6
    # whole-job throughput and the step time describe the same run
    assert abs(j["value"] * j["time_to_image_s"] / (j["paths_per_s"] * j["time_to_image_s"] * j["rays_per_path"]) - 1.0) < 1e-9
    if "cpu_baseline" in j:
        c = j["cpu_baseline"]
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in c, key
        assert c["kind"] == "port" and c["unit"] == "rays/s" and c["cores"] >= 1


def _default_line():
    default = [p for p in LINES if p.endswith(PREFIX + "bench.json")]
    assert len(default) == 1, default
    return _line(default[0])


def test_default_line_has_both_scenes_of_the_metric_and_the_cpu_baseline():
    j = _default_line()
    assert "cpu_baseline" in j and j["n_gpus"] == 1 and j["scaling"] == "weak"
    assert j["config"]["workload"].startswith("dragon 1920x1080") and j["cornell"]["rays_per_s"] > 0


def test_roofline_can_be_recomputed_from_the_committed_profiles():
    """frac = SQ_INSTS_VALU per launch (profiles/traffic.json, PMC passes of these sources) / launch duration
    (rocprofv3 kernel statistics of the one-frame-at-a-time run) / (0.5 x SIMDs x 2.4 GHz)."""
    j = _default_line()
    r = j["roofline"]
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    m = re.match(r"(\w+) (\d+x\d+), (\d+) spp per step, max depth (\d+)", j["config"]["workload"])
    rec = t[f"{m.group(1)} {m.group(2)} {m.group(3)}spp depth{m.group(4)}"]
    assert abs(rec["bytes_per_launch"] - r["traffic"]) < 1.0 and abs(rec["valu_insts_per_launch"] - r["valu_insts_per_launch"]) < 1.0
    assert rec["source_hash"] == bench.source_hash(), "the tracing kernels changed after the PMC passes: run tools/profile_round.sh again"
    assert r["pmc_source"]["stale"] is False
    ns, calls = 0.0, 0
    for row in csv.DictReader(open(os.path.join(ROOT, "profiles", PREFIX + "serial_kernel_stats.csv"))):
        if "k_intersect" in row["Name"] or "k_primary" in row["Name"]:
            ns += float(row["TotalDurationNs"]); calls += int(row["Calls"])
    avg_ms = ns / calls / 1e6
    assert abs(avg_ms - r["avg_launch_ms"]) / avg_ms < 0.03       # two separate runs of the same command
    frac = rec["valu_insts_per_launch"] / (avg_ms * 1e-3) / 1e9 / r["peak"]
    assert abs(frac - r["frac"]) < 0.03 and frac <= 1.0
    hbm = rec["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / 8000.0
    assert abs(hbm - r["hbm"]["frac"]) < 0.01 and hbm <= 1.0


def test_cornell_roofline_can_be_recomputed_from_the_committed_profiles():
    """The metric's second scene: the `cornell` object of the default line carries its own roofline, and it follows
    from the Cornell PMC record and the rocprofv3 kernel statistics of a separate Cornell run."""
    j = _default_line()
    r = j["cornell"]["roofline"]
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["cornell 1920x1080 256spp depth10"]
    assert rec["source_hash"] == bench.source_hash() and r["pmc_source"]["stale"] is False
    assert abs(rec["valu_insts_per_launch"] - r["valu_insts_per_launch"]) < 1.0 and abs(rec["lane_utilisation"] - r["lane_utilisation"]) < 1e-12
    ns, calls = 0.0, 0
    for row in csv.DictReader(open(os.path.join(ROOT, "profiles", PREFIX + "cfg_cornell_kernel_stats.csv"))):
        if "k_intersect" in row["Name"] or "k_primary" in row["Name"]:
            ns += float(row["TotalDurationNs"]); calls += int(row["Calls"])
    avg_ms = ns / calls / 1e6
    assert abs(avg_ms - r["avg_launch_ms"]) / avg_ms < 0.05       # two separate runs
    frac = rec["valu_insts_per_launch"] / (avg_ms * 1e-3) / 1e9 / r["peak"]
    assert abs(frac - r["frac"]) < 0.03 and frac <= 1.0
    assert abs(r["useful_frac"] - r["frac"] * r["lane_utilisation"]) < 1e-12
    # and the Cornell configuration line of its own agrees with the object of the default line
    cfg2 = _line(os.path.join(ROOT, "profiles", PREFIX + "cfg2_cornell_1080p.json"))["roofline"]
    assert abs(cfg2["frac"] - r["frac"]) < 0.03


def test_lane_utilisation_follows_from_the_counters():
    """lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), recomputed from the committed PMC summary."""
    vals = {}
    kernel = None
    for line in open(os.path.join(ROOT, "profiles", PREFIX + "pmc.txt")):
        if line.startswith("== "):
            kernel = line.split()[1]
        elif kernel and "sum=" in line:
            vals.setdefault(kernel, {})[line.split()[0]] = float(line.split("sum=")[1].split()[0])
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["dragon 1920x1080 256spp depth10"]
    for k in ("k_primary", "k_intersect"):
        lanes = vals[k]["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals[k]["SQ_ACTIVE_INST_VALU"])
        assert abs(lanes - rec["kernels"][k]["lane_utilisation"]) < 2e-3          # the summary prints four digits


def test_make_roofline_fractions():
    """Unit check of the arithmetic in bench.make_roofline, no GPU: 1024 SIMDs at 2.4 GHz issue at most 1228.8 G
    vector instructions per second."""
    pmc = {"valu_insts_per_launch": 6.144e9, "bytes_per_launch": 8.0e9, "lane_utilisation": 0.6, "source_hash": bench.source_hash(), "command": "x"}
    r = bench.make_roofline(launches=20, total_ms=200.0, rays=2.0e9, n_simds=1024, pmc=pmc, compulsory_bytes=8.0e10,
                            survey_bytes_per_ray=1800.0, frames_in_flight=1, per_kernel={})
    assert r["peak"] == 1228.8 and abs(r["achieved"] - 614.4) < 1e-9 and abs(r["frac"] - 0.5) < 1e-12
    assert abs(r["hbm"]["achieved"] - 800.0) < 1e-9 and abs(r["hbm"]["frac"] - 0.1) < 1e-12 and r["hbm"]["traffic_over_compulsory"] == 2.0
    assert r["pmc_source"]["stale"] is False and abs(r["insts_per_cycle_per_simd"] - 0.25) < 1e-12
    assert r["lane_utilisation"] == 0.6 and abs(r["useful_frac"] - 0.3) < 1e-12
    none = bench.make_roofline(20, 200.0, 2.0e9, 1024, None, 8.0e10, None, 1, {})
    assert none["frac"] is None and none["hbm"]["frac"] is None and "survey_model" not in none
