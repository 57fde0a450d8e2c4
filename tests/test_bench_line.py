"""The driver reads bench.py's LAST stdout line; round 5's 20.7 KB line was not parsed.
The line is now compact_line(out): < 4 KB, json.loads-able, carrying roofline.frac and cpu_baseline.value; the rest
goes to bench_detail.json.  The fixture is round 5's full (unparsed) line, i.e. the largest `out` the script builds."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FULL = os.path.join(ROOT, "tests", "golden", "bench", "r5_full_line.json")


def _full():
    with open(FULL) as f:
        return json.loads(f.read())


def test_compact_line_is_small_and_complete():
    out = _full()
    assert len(json.dumps(out)) > 16000          # the fixture really is the oversized line
    line = bench.compact_line(out)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT_BYTES
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["config"]["workload"] and back["config"]["global_batch"] == 64
    rl = back["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_ms"):
        assert rl.get(k) is not None, k
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    cb = back["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and cb["sample"]
    assert back["int8"]["value"] > 0 and back["int8"]["roofline"]["frac"] > 0
    assert abs(back["value"] - out["value"]) / out["value"] < 1e-4
    assert all(v for v in back["batch_sweep"].values())
    # no nested per-kernel dictionaries or prose blocks leak into the line
    assert "by_kernel" not in rl and "per_pipe" not in rl and "achieved_is" not in rl


def test_emit_prints_the_compact_line_last_and_writes_the_detail(tmp_path, monkeypatch):
    out = _full()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    so, se = io.StringIO(), io.StringIO()
    with redirect_stdout(so), redirect_stderr(se):
        bench.emit(out)
    lines = [ln for ln in so.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_LIMIT_BYTES
    assert json.loads(lines[0])["roofline"]["frac"] > 0
    with open(tmp_path / "bench_detail.json") as f:
        detail = json.load(f)
    assert detail["roofline"]["by_kernel"] and detail["config2_yolov3_tiny_416_b32_fp32"]["roofline"]


def test_line_survives_a_failed_cpu_baseline_and_missing_legs():
    out = _full()
    out["cpu_baseline"] = {"error": "x" * 5000}
    out.pop("int8")
    out["roofline"]["traffic"] = None
    line = bench.compact_line(out)
    assert len(json.dumps(line)) < bench.LINE_LIMIT_BYTES
    assert line["roofline"]["traffic"] is None and "error" in line["cpu_baseline"]
