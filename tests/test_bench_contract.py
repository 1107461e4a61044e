"""bench.py prints ONE JSON line with the fields the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract(built):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--systems", "4000",
                          "--cpu-sample", "2000"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1e5  # the north-star's floor: 1e5 converged 32-constraint systems per second
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    # the fraction is quoted on bytes that move: it cannot exceed what a streaming copy reaches on this part, and it agrees
    # with the counters' bytes over the same time
    assert r["frac_of_streaming_copy_rate"] <= 1.0
    assert r["algorithmic_bytes_per_system"] == 1920 and r["by_survey_8d_bytes"]["bytes_per_system"] == 2560
    if r["traffic"] is not None:
        assert abs(r["frac"] - r["frac_by_counter_bytes"]) <= 0.05 * r["frac_by_counter_bytes"]
    assert r["mixed_structure"]["algorithmic_bytes_per_system"] == 2560 and r["mixed_structure"]["frac"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
