"""Host logic of the domain-decomposed path on the CPU: partition, owner-first local numbering, halo lists, exercised
with a 2-rank torch.distributed run over gloo (the multi-GPU path uses RCCL for the same exchanges).  No GPU needed."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, shape, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("opm-autodiff_amd")
    nx, ny, nz = shape
    pat = pkg.grid.cartesian_pattern(nx, ny, nz)
    Nb = pat["Nb"]
    px, py, pz = pkg.ras.block_layout(world)
    owner = pkg.ras.cartesian_owner(nx, ny, nz, px, py, pz)
    lp = pkg.ras.local_problem(pat["rowptr"], pat["col"], owner, rank)
    rng = np.random.default_rng(5)
    xg = rng.standard_normal((Nb, 3))
    vals = pkg.grid.synthetic_block_values(pat, seed=3).reshape(-1, 3, 3)
    Nown, Ngh = lp["Nown"], lp["Nghost"]
    x = np.zeros((Nown + Ngh, 3))
    x[:Nown] = xg[lp["gids"][:Nown]]
    # halo exchange exactly as libopmhip does it: pack send cells per neighbour, send/recv, ghosts land contiguously
    reqs = []
    recv = []
    for qi, nb in enumerate(lp["neigh"]):
        s = torch.from_numpy(np.ascontiguousarray(x[lp["send_cells"][lp["send_ptr"][qi]:lp["send_ptr"][qi + 1]]]))
        r = torch.empty((int(lp["recv_ptr"][qi + 1] - lp["recv_ptr"][qi]), 3), dtype=torch.float64)
        reqs.append(dist.isend(s, int(nb)))
        reqs.append(dist.irecv(r, int(nb)))
        recv.append(r)
    for r_ in reqs:
        r_.wait()
    for qi, r in enumerate(recv):
        x[Nown + lp["recv_ptr"][qi]:Nown + lp["recv_ptr"][qi + 1]] = r.numpy()
    ok_halo = bool(np.array_equal(x[Nown:], xg[lp["gids"][Nown:]]))
    # distributed SpMV on the owned rows
    lv = vals[lp["entry"]]
    y = np.zeros((Nown, 3))
    rowid = np.repeat(np.arange(Nown), np.diff(lp["rows"]))
    np.add.at(y, rowid, np.einsum("kij,kj->ki", lv, x[lp["cols"]]))
    row_g = pkg.grid.row_of_entries(pat["rowptr"])
    yg = np.zeros((Nb, 3))
    np.add.at(yg, row_g, np.einsum("kij,kj->ki", vals, xg[pat["col"]]))
    ok_spmv = bool(np.allclose(y, yg[lp["gids"][:Nown]], rtol=1e-13, atol=1e-13))
    # distributed dot
    d = torch.tensor([float((x[:Nown] * x[:Nown]).sum())], dtype=torch.float64)
    dist.all_reduce(d)
    ok_dot = bool(abs(d.item() - float((xg * xg).sum())) < 1e-10 * d.item())
    # structure: columns ascend, ghosts last, every row keeps its diagonal, ghost ranges cover all ghosts
    ok_struct = True
    for i in range(Nown):
        c = lp["cols"][lp["rows"][i]:lp["rows"][i + 1]]
        ok_struct &= bool(np.all(np.diff(c) > 0)) and (i in c)
    ok_struct &= int(lp["recv_ptr"][-1]) == Ngh and bool(np.all(owner[lp["cells"][:Nown]] == rank))
    ok_struct &= bool(np.all(np.diff(lp["gids"][:Nown]) > 0))
    q.put((rank, ok_halo, ok_spmv, ok_dot, ok_struct, Nown, Ngh))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (8, 6, 4)), (4, (8, 6, 4))])
def test_partition_halo_spmv_over_gloo(world, shape):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[5] for r in res) == shape[0] * shape[1] * shape[2]
    for r in res:
        assert r[1] and r[2] and r[3] and r[4], r
        assert r[6] > 0


def test_window_built_subdomain_equals_global_slice(pkg):
    """ras.cartesian_subdomain_case (never builds the global pattern) against slicing the global case."""
    n, world = 4, 4
    px, py, pz = pkg.ras.block_layout(world)
    NX, NY, NZ = px * n, py * n, pz * n
    g = pkg.decks.cartesian_case(NX, NY, NZ, state="mixed", heterogeneous=True)
    owner = pkg.ras.cartesian_owner(NX, NY, NZ, px, py, pz)
    for rank in range(world):
        c = pkg.ras.cartesian_subdomain_case(n, world, rank, state="mixed", heterogeneous=True)
        lp = pkg.ras.local_problem(g["rowptr"], g["col"], owner, rank)
        assert c["Nb"] == n ** 3 and np.array_equal(c["gids"], lp["gids"])
        assert np.array_equal(c["rowptr"], lp["rows"]) and np.array_equal(c["col"], lp["cols"])
        assert np.array_equal(c["trans"], g["trans"][lp["entry"]]) and np.array_equal(c["area"], g["area"][lp["entry"]])
        gi = lp["gids"]
        assert np.array_equal(c["pv"].reshape(-1, 3), g["pv"].reshape(-1, 3)[gi]) and np.array_equal(c["meaning"], g["meaning"][gi])
        assert np.array_equal(c["depth"], g["depth"][gi])
        for k in ("neigh", "send_ptr", "send_cells", "recv_ptr"):
            assert np.array_equal(c["halo"][k], lp[k])


def _gather_worker(rank, world, port, q):
    """the exchange pattern of the CPR pressure stage that spans the ranks (csrc/cpr.hip: cpr_gather_setup / cpr_gathered_cycle), over
    gloo: slices of different lengths, padded to the longest, all-gathered, unpadded through the index list r * maxn + i.
    What this does NOT do: run a line of csrc/comm.hip or csrc/cpr.hip - it re-enacts the protocol's index arithmetic in Python between
    real processes (a world-size-2/3 rendezvous, a real all-gather of ragged slices), so a mistake in the PROTOCOL shows here without a
    GPU; the library's own implementation of it is compared with the oracle on the loopback communicator (tests/test_gpu_dd.py::
    test_dd_cpr_pressure_stage_across_the_ranks) and has never run over RCCL with more than one rank."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nloc = 5 + 3 * rank                                   # my rows of the joined level
    mine = torch.tensor([nloc], dtype=torch.int64)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = [int(s.item()) for s in sizes]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    maxn = max(sizes)
    rng = np.random.default_rng(100 + rank)
    b = rng.standard_normal(nloc)
    send = torch.zeros(maxn, dtype=torch.float64)
    send[:nloc] = torch.from_numpy(b)
    recv = [torch.zeros(maxn, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(recv, send)
    flat = torch.cat(recv).numpy()
    unpad = np.concatenate([r * maxn + np.arange(sizes[r]) for r in range(world)])
    joined = flat[unpad]
    expect = np.concatenate([np.random.default_rng(100 + r).standard_normal(sizes[r]) for r in range(world)])
    ok = bool(np.array_equal(joined, expect)) and len(joined) == offs[-1]
    # every rank cycles on the same joined vector: its slice of the (here: doubled) result is what it keeps
    mine_back = (2.0 * joined)[offs[rank]:offs[rank + 1]]
    ok &= bool(np.array_equal(mine_back, 2.0 * b))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_joined_level_gather_protocol_over_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_bench_two_ranks_control_plane_over_gloo(tmp_path):
    """`bench.py --gpus 2` the way the driver starts it (torch.distributed.run, two ranks, 127.0.0.1) on the CPU: the device context is a
    stand-in (tests/bench_fake_rank.py), everything else is bench.py's own multi-rank code - rendezvous over gloo, the communicator id
    broadcast from rank 0, the gathered RCCL record (and the rule that refuses a communicator that does not span --gpus ranks), barriers, the
    MAX of the elapsed times, the communication spans gathered over the ranks as max / mean, ONE line of < 4 KB from rank 0 with `value` =
    steps x ranks / time.  No run with more than one GPU has been possible: this is what can be checked of it here."""
    import json
    import subprocess
    import sys
    import socket

    def free_port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]
    port = free_port()
    detail = str(tmp_path / "detail.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "bench_fake_rank.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--cells-per-edge", "10", "--steady-after", "12", "--steady-steps", "4",
           "--detail", detail]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096          # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and "cpu_baseline" not in d and d["rccl"] == {"nranks": 2, "kind": "rccl"}
    assert abs(d["value"] - 2 * d["newton_iterations_per_s_global"]) < 1e-3 * d["value"] and "2x1x1" in d["config"]["parallelism"]
    assert d["unit"].startswith("Newton iterations/s x 1M-cell subdomains")
    c = d["comm"]
    assert c["halo"]["launches_per_rank"] > 0 and c["halo"]["avg_ms_max"] > c["halo"]["avg_ms_mean"] > 0     # rank 1's stand-in spans are twice rank 0's
    assert abs(c["halo"]["avg_ms_max"] / c["halo"]["avg_ms_mean"] - 4.0 / 3.0) < 1e-3 and "cpr_gather" not in c
    # the one-domain comparator comes from an N = 1 run of the SAME case on this box or not at all (tests/test_bench_launcher.py): no 10^3-cell
    # N = 1 record exists here, so it is null - not a typed-in figure
    assert c["single_domain_linear_iterations_per_newton"] is None and c["linear_iterations_per_newton"] > 0
    full = json.load(open(detail))
    assert full["rccl"]["rank_devices"] == [0, 1] and full["rccl"]["selftest_sum"] == full["rccl"]["selftest_expected"] == [3.0, 4.0]
    assert full["steady_state"]["steps"] == 4 and full["spmv_boundary_share_of_time"] == 0.2
    # a communicator that does not span the ranks that were asked for: every rank exits non-zero, no line
    bad = subprocess.run(cmd[:8] + [str(free_port()), cmd[9]] + ["--gpus", "2", "--steps", "2", "--warmup", "1", "--cells-per-edge", "10", "--steady-after", "0"], capture_output=True, text=True, timeout=600,
                         env=dict(env, OPMHIP_FAKE_NRANKS="1"), cwd=ROOT)
    assert bad.returncode != 0 and "RCCL communicator spans 1 rank" in (bad.stderr + bad.stdout)
