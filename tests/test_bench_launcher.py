"""bench.py's launcher logic (CPU only): `--gpus N` either runs inline (N = 1 or a rank of torch.distributed.run), starts N
ranks as children before any GPU call, or refuses - it never silently runs fewer GPUs than asked for."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launch_plan(bench):
    assert bench.launch_plan(1, {}, 0) == ("inline", 1)
    assert bench.launch_plan(1, {}, 8) == ("inline", 1)
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, 8) == ("inline", 8)      # the driver's torch.distributed.run form
    assert bench.launch_plan(4, {}, 8) == ("spawn", 4)                        # plain `python bench.py --gpus 4`
    with pytest.raises(SystemExit, match="only 2 GPU"):
        bench.launch_plan(8, {}, 2)
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.launch_plan(8, {"WORLD_SIZE": "2"}, 8)
    with pytest.raises(SystemExit):
        bench.launch_plan(0, {}, 8)


def test_spawn_command_is_the_drivers(bench):
    cmd = bench.spawn_command(4, ["--gpus", "4", "--steps", "7"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7"]


def test_refuses_without_enough_gpus():
    """no GPU in the CPU container: --gpus 2 must fail loudly (and before touching any device)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the run would start")
    assert r.returncode != 0 and "refusing to run a smaller job" in (r.stderr + r.stdout)


def test_traffic_is_tied_to_the_kernel_sources(bench):
    t, src = bench.pmc_traffic("k_spmv")
    h = bench.kernel_source_hash()
    assert len(h) == 16
    if t is None:
        assert h in src          # says which sources it looked for
    else:
        import json
        assert json.load(open(os.path.join(ROOT, src)))["kernel_source_sha16"] == h


def test_comm_summary_and_the_line_with_eight_ranks(bench):
    """what `bench.py --gpus 8` adds to its line - the communication spans per rank as max / mean, the RCCL communicator - from synthetic
    per-rank records (no run with more than one GPU has been possible): the arithmetic, and the line still under 4 KB"""
    every = [{"halo": (70, 7.0 + r), "allreduce": (141, 14.1), "cpr_gather": (0, 0.0)} for r in range(8)]
    c = bench.comm_summary(every)
    assert set(c) == {"halo", "allreduce"}                       # no rank recorded a gather span: left out
    assert c["halo"]["launches_per_rank"] == 70 and abs(c["halo"]["avg_ms_max"] - 14.0 / 70) < 1e-5 and abs(c["halo"]["avg_ms_mean"] - 10.5 / 70) < 1e-5
    assert c["halo"]["total_ms_max"] == 14.0 and c["halo"]["total_ms_mean"] == 10.5 and c["allreduce"]["avg_ms_max"] == 0.1
    kern = {k: {"algorithmic_GBps": 4321.0} for k in ("spmv", "ilu_apply", "ilu_factor", "vector", "assemble", "iq_update", "convergence")}
    out = {"metric": "m" * 90, "value": 640.123456, "unit": "u" * 100, "n_gpus": 8, "steps": 20, "warmup": 3, "ms_per_step": 12.5, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "w" * 300, "parallelism": "p" * 120, "cells_per_gpu": 1000000},
           "newton_iterations_per_s_global": 80.0, "linear_iterations_per_newton": 21.3, "linear_solve_GBps": 4000.0, "kernels": kern,
           "roofline": {"bound": "hbm", "kernel": "k_spmv", "achieved": 5000.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.625, "traffic": None, "avg_launch_ms": 0.11,
                        "algorithmic_bytes_per_launch": 579440004, "frac_of_stream_ceiling": 0.9, "traffic_source": "s" * 200},
           "steady_state": {"value": 400.0, "linear_iterations_per_newton": 30.0}, "stream_ceiling": {"read_GBps": 5900.0},
           "rccl": {"nranks": 8, "kind": "rccl", "rank_devices": list(range(8)), "hosts": ["h"]},
           "comm": dict(c, profiled_solves=7, profiled_every=3, linear_iterations_per_newton=21.3, single_domain_linear_iterations_per_newton=17.5),
           "device": {"name": "AMD Instinct MI355X"}}
    import json
    line = bench.compact_line(out, "gpurun_out/bench_detail.json")
    assert len(json.dumps(line)) < bench.LINE_LIMIT and line["comm"]["halo"]["avg_ms_max"] == c["halo"]["avg_ms_max"] and line["rccl"] == {"nranks": 8, "kind": "rccl"}
    assert "cpu_baseline" not in line and line["n_gpus"] == 8 and line["steady_state"]["value"] == 400.0


def test_the_line_stays_short_whatever_an_extra_window_says(bench):
    """a long error text in an extra window must not cost the run its line: extras are dropped (and named), the contract's keys stay"""
    import json
    base = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "config": {"workload": "w" * 300}, "roofline": {"bound": "hbm", "frac": 0.6}, "cpu_baseline": {"value": 1.0, "cores": 16, "kind": "port", "sample": "s" * 300, "cpu_model": "c"},
            "device": "d" * 5000, "cpr": {"error": "e" * 120}, "kernel_GBps": {str(i): 1.0 for i in range(200)}}
    txt = bench.fit_line(dict(base))
    d = json.loads(txt)
    assert len(txt) < bench.LINE_LIMIT and d["value"] == 1.0 and d["roofline"]["frac"] == 0.6 and d["cpu_baseline"]["cores"] == 16 and "workload" in d["config"]
    assert "device" in d["dropped_for_length"] and "device" not in d
    small = bench.fit_line({k: v for k, v in base.items() if k not in ("device", "kernel_GBps")})
    assert "dropped_for_length" not in json.loads(small)


def test_single_domain_comparator_comes_from_a_measurement_or_not_at_all(bench, tmp_path):
    """the N > 1 line's single-domain iteration count: read from the detail file an N = 1 run of the SAME case left on the box, None when
    there is none, when it was another case, another window or non-default flags - never a typed-in constant"""
    import json
    f = tmp_path / "bench_detail.json"
    assert bench.single_domain_comparator(str(f), 100, 20, 3, "ilu0", True) is None                 # no file
    rec = {"n_gpus": 1, "steps": 20, "warmup": 3, "preconditioner": "ilu0", "linear_iterations_per_newton": 17.049,
           "config": {"cells_per_gpu": 10 ** 6, "ilu_ordering_chosen_by": "library default (auto)"},
           "cpr": {"linear_iterations_per_newton": 4.45}, "cpr_quasiimpes": {"error": "x"}}
    f.write_text(json.dumps(rec))
    assert bench.single_domain_comparator(str(f), 100, 20, 3, "ilu0", True) == 17.05
    assert bench.single_domain_comparator(str(f), 100, 20, 3, "cpr", True) == 4.45                  # from the N = 1 run's CPR side window
    assert bench.single_domain_comparator(str(f), 100, 20, 3, "cpr_trueimpes", True) == 4.45
    assert bench.single_domain_comparator(str(f), 100, 20, 3, "cpr_quasiimpes", True) is None       # that side window failed: nothing to quote
    assert bench.single_domain_comparator(str(f), 64, 20, 3, "ilu0", True) is None                  # another size
    assert bench.single_domain_comparator(str(f), 100, 40, 3, "ilu0", True) is None                 # another window
    assert bench.single_domain_comparator(str(f), 100, 20, 3, "ilu0", False) is None                # this run has ordering / smoother flags
    rec["config"]["ilu_ordering_chosen_by"] = "--reorder"
    f.write_text(json.dumps(rec))
    assert bench.single_domain_comparator(str(f), 100, 20, 3, "ilu0", True) is None                 # ... or the N = 1 run had
    rec["n_gpus"] = 2
    f.write_text(json.dumps(rec))
    assert bench.single_domain_comparator(str(f), 100, 20, 3, "ilu0", True) is None
    f.write_text("not json")
    assert bench.single_domain_comparator(str(f), 100, 20, 3, "ilu0", True) is None
    assert not hasattr(bench, "SINGLE_DOMAIN_LIN_ITS")
