"""bench.py's one JSON line on a small grid: the keys the driver's contract names are there, with the roofline and the CPU
baseline objects, the numbers hang together, and the line stays short enough for the driver to parse from the tail of stdout (round 4's
22 KB line was cut in half); the full record is in the file the line names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract(tmp_path):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "24", "--steps", "4", "--warmup", "2", "--steady-after", "8",
                        "--steady-steps", "3", "--detail", detail], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.strip().splitlines()
    lines = [ln for ln in out if ln.startswith("{")]
    assert len(lines) == 1 and out[-1] == lines[0]       # the LAST line of stdout is the one JSON object
    assert len(lines[0]) < 4096, len(lines[0])
    d = json.loads(lines[0])
    full = json.load(open(detail))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]     # both rounded to four decimals in the line
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-4
    assert ro["traffic"] is None     # PMC traffic is only quoted for the 100^3 configuration it was measured on
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["linear_solve_GBps"] > 0
    fcb = full["cpu_baseline"]
    assert fcb["host"]["physical_cores_in_mask"] >= 1 and abs(fcb["value"] - cb["value"]) < 1e-3
    assert cb["cores"] == 1 or str(cb["cores"]) in fcb["host"]["thread_sweep_newton_its_per_s"]
    assert d["steady_state"]["value"] > 0 and full["steady_state"]["steps"] == 3 and d["stream_read_GBps"] > 0
    # no ordering / smoother flag was passed: the line reports what the library chose (a 24^3 grid: the greedy colouring, red-black)
    assert d["config"]["ilu_ordering"] == "graph_coloring_greedy" and d["config"]["ilu_colors"] == 2 and d["config"]["ilu_chain_length"] == 0
    assert d["config"]["ilu_ordering_chosen_by"].startswith("library default") and d["cpr_amg_ilu_levels"] == 1
    # Flow's "cpr" (true-IMPES weights) and the quasi-IMPES variant, side by side under their reference names
    assert d["cpr"]["value"] > 0 and d["cpr_quasiimpes"]["value"] > 0 and d["cpr_reuse_setup_2"]["value"] > 0 and "rccl" not in d
    assert full["cpr"]["kernels"]["cpr_amg"]["launches"] > 0 and full["kernels"]["spmv"]["launches"] > 0
