"""bench.py's one-line contract on a small workload: the keys the driver reads, the roofline and CPU-baseline objects,
and kernel time per step no larger than the step itself."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--clips", "96",
           "--cpu-clips", "8", "--cpu-pool-seconds", "0"] + list(extra)
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [(), ("--inflight", "1"), ("--streams", "2"), ("--config", "3"), ("--config", "5")])
def test_bench_line(extra):
    d = _run(*extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "traffic" in r and r["fp32"]["peak"] == 157.3
    assert r["kernel_ms_per_step"] <= d["ms_per_step"] * 1.001          # union of the launch intervals, never more than the step
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
