"""The committed benchmark lines keep the driver's contract: the keys it reads, the roofline and
CPU-baseline objects, and consistency between the figures inside a line."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import re

# the newest set only: earlier sets are kept for the history of the design and predate some fields
_ALL = glob.glob(os.path.join(ROOT, "profiles", "r*_v*_bench*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*_v*_serial_bench.json"))
_TAG = max((re.match(r"(r\d+)_v(\d+)_", os.path.basename(p)).groups() for p in _ALL), key=lambda t: (int(t[0][1:]), int(t[1])))
LINES = sorted(set(p for p in _ALL if os.path.basename(p).startswith(f"{_TAG[0]}_v{_TAG[1]}_")))


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
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = rays of the timed traversal launches x algorithmic bytes per ray / their summed duration
    want = r["rays_per_launch"] * r["bytes_per_ray"] / (r["avg_launch_ms"] * 1e-3) / 1e9
    assert abs(want - r["achieved"]) / r["achieved"] < 1e-6
    per = r["per_kernel"]
    n = per["k_primary"]["launches"] + per["k_intersect"]["launches"]
    assert n == r["launches"] == j["steps"] * 10
    ms = per["k_primary"]["launches"] * per["k_primary"]["avg_launch_ms"] + per["k_intersect"]["launches"] * per["k_intersect"]["avg_launch_ms"]
    assert abs(ms / n - r["avg_launch_ms"]) < 1e-6
    # whole-job throughput and the step time describe the same run
    rays_per_step = j["value"] * j["ms_per_step"] * 1e-3
    assert abs(rays_per_step / (r["rays_per_launch"] * 10) - 1.0) < 1e-3
    if "cpu_baseline" in j:
        c = j["cpu_baseline"]
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in c, key
        assert c["kind"] == "port" and c["unit"] == "rays/s" and c["cores"] >= 1


def test_default_line_is_there_with_its_cpu_baseline():
    default = [p for p in LINES if p.endswith("_bench.json") and "serial" not in p and "cornell" not in p]
    assert len(default) == 1, default
    j = _line(default[0])
    assert "cpu_baseline" in j and j["n_gpus"] == 1
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    m = re.match(r"(\w+) (\d+x\d+), (\d+) spp per step, max depth (\d+)", j["config"]["workload"])
    key = f"{m.group(1)} {m.group(2)} {m.group(3)}spp depth{m.group(4)}"
    assert key in t and abs(t[key]["bytes_per_launch"] - j["roofline"]["traffic"]) < 1.0
