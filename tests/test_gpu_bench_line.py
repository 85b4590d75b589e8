"""bench.py's one JSON line on a small grid: the keys the driver's contract names are there, with the roofline and the CPU
baseline objects, and the numbers hang together."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "24", "--steps", "4", "--warmup", "2", "--steady-after", "8",
                        "--steady-steps", "3"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert ro["traffic"] is None     # PMC traffic is only quoted for the 100^3 configuration it was measured on
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["host"]["physical_cores_in_mask"] >= 1 and cb["linear_solve_GBps"] > 0
    assert cb["cores"] == 1 or str(cb["cores"]) in cb["host"]["thread_sweep_newton_its_per_s"]
    assert d["steady_state"]["steps"] == 3 and d["stream_ceiling"]["read_GBps"] > 0
    # Flow's "cpr" (true-IMPES weights) and the quasi-IMPES variant, side by side under their reference names
    assert d["cpr"]["value"] > 0 and d["cpr_quasiimpes"]["value"] > 0 and d["cpr_reuse_setup_2"]["value"] > 0 and d["rccl"] is None
