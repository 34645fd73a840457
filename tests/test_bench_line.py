"""The ONE stdout line of bench.py must stay short: the driver keeps a bounded tail of stdout, and round 5's 20 KB line was recorded as
`"parsed": null` — the round's headline went unmeasured.  short_line() is fed the largest record a run has produced (round 5's own,
profiles/r05_bench.json.txt) and a record bloated far beyond it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (top level of bench.py imports nothing that needs a GPU)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")
ROOF = ("bound", "achieved", "peak", "unit", "frac", "traffic")


def _round5_record():
    for ln in open(os.path.join(ROOT, "profiles", "r05_bench.json.txt")):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError("profiles/r05_bench.json.txt holds no JSON line")


def _check(line):
    assert "\n" not in line and len(line) < bench.SHORT_LINE_MAX, len(line)
    out = json.loads(line)
    for k in REQUIRED:
        assert k in out, k
    for k in ROOF:
        assert k in out["roofline"], k
    assert out["roofline"]["bound"] in ("hbm", "mfma")
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in out["cpu_baseline"], k
    assert "workload" in out["config"] and "model" not in out["config"]
    return out


def test_short_line_of_the_round_5_record_fits():
    full = _round5_record()
    assert len(json.dumps(full)) > 15000          # the record that cost round 5 its driver measurement
    out = _check(bench.short_line(full))
    assert abs(out["value"] - full["value"]) / full["value"] < 1e-5
    assert abs(out["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert out["config"]["value_clustered"] > 0
    assert out["cpu_baseline"]["speedup_vs_1_core"]["seq_push"] > 1
    assert out["legs"]["orb_images_per_s"] > 0 and out["legs"]["lsd_one_image_ms"] > 0 and out["legs"]["configs3_gn_frame_pairs_per_s"] > 0


def test_short_line_survives_bloat_and_failed_legs():
    full = _round5_record()
    for k in list(full):
        if isinstance(full[k], dict):
            full[k]["note"] = "x" * 5000            # prose never reaches the line
            full[k]["another_leg"] = {"rows": list(range(500))}
    full["config"]["workload"] = full["config"]["workload"][:300]
    full["orb_front_end"] = {"error": "StvoError: out of memory"}
    full["per_rank_frame_pairs_per_s"] = [1.25e6] * 8
    full["n_gpus"] = full["rccl_ranks"] = 8
    out = _check(bench.short_line(full))
    assert out["legs"]["orb_images_per_s"] == "error"
    assert len(out["per_rank_frame_pairs_per_s"]) == 8


def test_short_line_without_extras_or_cpu_baseline():
    full = _round5_record()
    for k in ("cpu_baseline", "latency", "configs1", "configs3", "orb_front_end", "lsd_front_end", "images_to_poses", "images_to_poses_with_lines",
              "reverse_check_correlated", "headline_clustered"):
        full.pop(k, None)
    line = bench.short_line(full)
    out = json.loads(line)
    assert len(line) < bench.SHORT_LINE_MAX and "cpu_baseline" not in out and "legs" not in out and out["roofline"]["frac"] > 0


def test_write_extras_round_trip(tmp_path):
    full = _round5_record()
    p = tmp_path / "bench_extras.json"
    bench.write_extras(full, str(p))
    assert json.load(open(p))["value"] == full["value"]
