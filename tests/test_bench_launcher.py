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
