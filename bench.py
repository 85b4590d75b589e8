#!/usr/bin/env python3
"""bench.py — Newton iterations per second of the per-Newton-iteration hot path on MI355X.

Metric (BASELINE.json): "Newton iterations/sec + linear-solve GB/s, 1M-cell 3-phase black-oil".
Workload (BASELINE.json configs[1]): synthetic 100 x 100 x 100 Cartesian three-phase black-oil grid, homogeneous rock,
SPE1 fluid, per GPU.  A "step" is one Newton iteration of BlackoilModelEbos::nonlinearIteration that does real work:
assembleReservoir (AD linearisation into block-CSR) -> getReservoirConvergence -> solveJacobianSystem (block-ILU0
factorisation + BiCGStab to 1e-2) -> updateSolution (chopped update, primary-variable switching, intensive
quantities).  Time steps follow one another under Flow's sub-step control (newton.AdaptiveTimeStepping: 1 day first,
grown by the Newton-iteration-count rule up to the 10-day report step of SURVEY.md §8d, rolled back and chopped by
0.33 when the Newton method gives up) with a fixed-rate five-spot source pair; failed iterations count as work.
Everything is resident in HBM when the timed region starts; nothing crosses PCIe inside it except ~200 bytes of
scalars per Newton iteration.

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU, either launched by the driver through torch.distributed.run (WORLD_SIZE set) or, when bench.py
is started plainly with --gpus N, by bench.py itself (it starts torch.distributed.run as a child BEFORE anything touches
a GPU and fails loudly when fewer than N devices are visible).  Restricted-additive-Schwarz domain decomposition of a
(px 100) x (py 100) x (pz 100) grid (opm-autodiff_amd/ras.py), 10^6 cells per GPU, halo exchange and scalar all-reduces
over RCCL inside libopmhip (weak scaling; 8 GPUs = BASELINE configs[3], 200^3).  Prints ONE JSON line on rank 0.

After the K timed steps the run goes on (untimed) to Newton iteration `--steady-after` and times `--steady-steps` more:
the steady-state phase of the simulation (10-day steps, ~30 linear iterations per Newton iteration) beside the easy
start-up phase the K steps measure.  `value` is the K-step figure the contract asks for; `steady_state` is an extra key.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")      # the CPU baseline sets its own thread count (orc_set_threads)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle OpenMP threads sleep: the box's CPU share may be a quota
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL between processes: the host driver supports dmabuf IPC only (exported on the boxes already; kept for an environment that lost it)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

DAY = 86400.0
PROFILE_EVERY = 3   # kernel scopes are recorded in every 3rd linear solve: the steady phase runs 2 Newton iterations per time step, a stride of 4 sampled only its first (harder) solves
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def device_info(torch, index):
    """what the box says about its GPU (the kernel times of this bench vary by box: see profiles/README.md)"""
    try:
        p = torch.cuda.get_device_properties(index)
        info = {"name": p.name, "compute_units": p.multi_processor_count, "memory_GiB": round(p.total_memory / 2**30, 1)}
        for k in ("clock_rate", "memory_clock_rate", "gcnArchName", "L2_cache_size"):
            if hasattr(p, k):
                info[k] = getattr(p, k)
        return info
    except Exception as e:   # never let a diagnostic take the bench line down
        return {"error": str(e)}


def alg_bytes(Nb, nnzb, rest_blocks=0):
    """Algorithmic bytes per launch (SURVEY.md §8d / BASELINE.md §4), block size 3, double values, int32 indices.
    rest_blocks > 0: the half-product form of ILU0-BiCGStab is in force (opmhip_get_product_form: U == upper(A)) - the product inside a solve
    streams the matrix WITHOUT its U part (that many blocks: lower entries and diagonal) plus the backward sweep's row sums, the ILU0
    application stores those sums, the factorisation writes the rest stream.  The same accounting rules as SURVEY 8d: every array once
    per kernel, 4-byte column indices, row pointers, vectors 24 B per row."""
    B = alg_bytes_plain(Nb, nnzb)
    if rest_blocks:
        B["spmv_full"], B["spmv_full_operands"] = B["spmv"], B["spmv_operands"]
        B["spmv"] = 76 * rest_blocks + 4 * (Nb + 1) + 72 * Nb                 # values + indices of the rest, x (gathered), the row sums, y
        B["spmv_operands"] = B["spmv"] + 24 * Nb                              # + the second operand of the folded scalar products
        B["ilu_apply"] += 24 * Nb                                             # the row sums written
        B["ilu_factor"] += 72 * rest_blocks                                   # the rest stream written
    return B


def alg_bytes_plain(Nb, nnzb):
    return {
        # block-CSR SpMV exactly as SURVEY 8d counts it: 72-B blocks + 4-B column indices, row pointers, x and y (579.44 MB at 100^3)
        "spmv": 76 * nnzb + 4 * (Nb + 1) + 48 * Nb,
        # ... + the second operand of the BiCGStab scalar products when they ride in the kernel (24 B per row; no wells, no OPMHIP_DOTS_SEPARATE)
        "spmv_operands": 76 * nnzb + 4 * (Nb + 1) + 48 * Nb + 24 * Nb,
        "ilu_apply": 76 * nnzb + 4 * (Nb + 1) + 4 * Nb + 72 * Nb,
        "ilu_factor": 2 * 72 * nnzb + 4 * nnzb + 4 * Nb,
        # per scope, three scopes per iteration: the p-update (4 passes), r-update (3), k_bicg_upd2 with both updates of x (8);
        # the scalar products behind a product ride in the product's kernel (with wells: k_dots, a scope of its own)
        "vector": 24 * Nb * 15 / 3,
        "assemble": 85 * Nb + 12 * nnzb + 72 * nnzb + 24 * Nb,
        "iq_update": 24 * Nb + 544 * Nb,
        "convergence": 56 * Nb,
        # CPR V-cycle: cpr_amg_bytes() from the hierarchy's actual level sizes (set per context once the hierarchy exists)
        "cpr_amg": None,
    }


def cpr_amg_bytes(Nb, level_n, level_nnz, ilu_levels=0):
    """Algorithmic bytes of one application of the pressure AMG (csrc/cpr.hip: k_cpr_restrict_fine + cpr_vcycle, the scope the
    profiler books under `cpr_amg`), from the hierarchy's own level sizes (opmhip_cpr_levels): per level that is not the coarsest
    TWO matrix passes (residual on the way down, post-smoothing on the way up) over 8-byte values and 4-byte column indices of
    the level's entries, and its vector passes - residual (b, x in, r out), restriction (r in; b and x of the coarser level out),
    prolongation (aggregate map, x in, x' out, the coarse x in), post-smoothing (1/diag, b, x' in, x out): 84 B per row + 24 B per
    coarse row; the restriction of the block residual in front (d and the weights in, b and x out: 64 B per cell); the coarsest
    level: a dense triangular solve (8 n^2) when it has <= 128 rows, else 1 + 4 Jacobi sweeps (12 B per entry + 32 B per row each).
    A level smoothed with ILU0 (--cpr-amg-ilu-levels): two applications of its factors (every off-diagonal entry once per application
    as 8-byte value + 4-byte column, 1 / U_ii, d in, v out twice) and two residual passes instead of one residual and one Jacobi pass."""
    total = 64 * Nb
    L = len(level_n)
    for l in range(L - 1):
        total += 2 * 12 * level_nnz[l] + 84 * level_n[l] + 24 * level_n[l + 1]
        if l < ilu_levels:
            total += 2 * (12 * (level_nnz[l] - level_n[l]) + 40 * level_n[l]) + 16 * level_n[l]
    nl, zl = level_n[-1], level_nnz[-1]
    total += 8 * nl * nl if nl <= 128 else 5 * (12 * zl + 32 * nl)
    return total


def make_simulation(pkg, model, report_step=10 * DAY):
    """Flow's sub-stepping (newton.AdaptiveTimeStepping: 1 day first, then grown by the Newton-iteration-count rule up to
    the report-step length, chopped by 0.33 and rolled back when the Newton method gives up) over the Newton loop."""
    nm = pkg.newton.BlackoilModelHip(model)
    return pkg.newton.AdaptiveTimeStepping(nm, pkg.newton.TimeSteppingParameters(initial_dt=1 * DAY, max_dt=report_step))


def cpu_baseline_run(pkg, case, src, threads, budget_s, max_newton):
    import oracle_bind
    import subprocess
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    orc = oracle_bind.Oracle(so)
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    o.set_source(src)
    sim = make_simulation(pkg, oracle_bind.OracleAsHipModel(o, tol=1e-2, maxit=200, w=0.9, mode="post_scale", reorder="none", threads=threads))
    newton = 0
    t_start = time.perf_counter()
    while newton < max_newton and time.perf_counter() - t_start < budget_s:
        sim.next_newton_iteration()
        newton += 1
    rep = sim.report
    total = rep.solver_time()
    return {"value": newton / total, "newton_iterations": newton, "linear_iterations": int(rep.total_linear_iterations), "cpu_seconds": total,
            "seconds": {"assemble": rep.assemble_time, "linear_setup": rep.linear_solve_setup_time, "linear_solve": rep.linear_solve_time,
                        "update": rep.update_time}}


def host_cpu_share():
    """what this process may use of the host: CPUs in the affinity mask, physical cores among them, and the cgroup's CPU
    quota (cpu.max: a box may show 256 logical CPUs and still be throttled to a share of them)"""
    try:
        mask = sorted(os.sched_getaffinity(0))
    except AttributeError:
        mask = list(range(os.cpu_count() or 1))
    cores = set()
    for cpu in mask:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu) as f:
                cores.add(f.read().strip())
        except OSError:
            cores.add(str(cpu))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                t = f.read().split()
            if path.endswith("cpu.max"):
                quota = None if t[0] == "max" else float(t[0]) / float(t[1])
            else:
                q = float(t[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    quota = None if q <= 0 else q / float(g.read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return {"cpus_in_affinity_mask": len(mask), "physical_cores_in_mask": len(cores), "cgroup_cpu_quota": quota}


def interleave_new_pages():
    """set_mempolicy(MPOL_INTERLEAVE, all nodes): pages touched from here on are spread round-robin over the NUMA nodes,
    so that the threads of both sockets of the host find half of every array in local memory whoever touched it first
    (the oracle's std::vectors are value-initialised by the thread that resizes them).  Best effort: False if refused."""
    import ctypes
    try:
        nodes = [int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]
        if len(nodes) < 2:
            return False
        mask = (ctypes.c_ulong * 16)()
        for nd in nodes:
            mask[nd // 64] |= 1 << (nd % 64)
        libc = ctypes.CDLL(None, use_errno=True)
        return libc.syscall(238, 3, mask, 16 * 64 + 1) == 0   # x86-64: SYS_set_mempolicy = 238, MPOL_INTERLEAVE = 3
    except Exception:   # noqa: BLE001
        return False


def cpu_baseline(pkg, case, src, nnzb):
    """The CPU port (oracle/) timed on the host cores on a bounded sample of the same workload: the same Newton and
    time-stepping loop from the same initial state, run the way N MPI ranks of Flow would: OpenMP-threaded assembly,
    block-Jacobi ILU0 over N contiguous row ranges, threaded SpMV / scalar products, BiCGStab to 1e-2, relaxation 0.9.
    N is MEASURED: a short sample at 16, 32, 64 threads and at every physical core this process may use (bounded by the
    cgroup's CPU quota when there is one), the headline run with the fastest.  The 1-thread run (natural-order ILU0 = one
    Flow rank) is reported beside it."""
    share = host_cpu_share()
    limit = share["physical_cores_in_mask"]
    if share["cgroup_cpu_quota"]:
        limit = max(1, min(limit, int(share["cgroup_cpu_quota"] + 0.5)))
    interleaved = interleave_new_pages()
    cand = sorted({t for t in (16, 32, 64, limit) if 1 < t <= limit} | ({limit} if limit > 1 else set()))
    sweep = {}
    for t in cand:   # 3 Newton iterations each (the first carries the one-time allocations): a few seconds per candidate
        r = cpu_baseline_run(pkg, case, src, t, budget_s=6.0, max_newton=3)
        sweep[t] = r["value"]
    threads = max(sweep, key=sweep.get) if sweep else 1
    mt = cpu_baseline_run(pkg, case, src, threads, budget_s=18.0, max_newton=24) if threads > 1 else None
    st = cpu_baseline_run(pkg, case, src, 1, budget_s=12.0, max_newton=8)
    head = mt if (mt is not None and mt["value"] > st["value"]) else st
    cores = threads if head is mt else 1
    # algorithmic bytes of one BiCGStab iteration (2 SpMV + 2 M^-1 + the vector passes), as for the GPU's linear_solve_GBps
    B = alg_bytes(case["Nb"], nnzb)
    it_bytes = 2 * B["spmv"] + 2 * B["ilu_apply"] + 3 * B["vector"]
    ls_gbps = lambda r: round(r["linear_iterations"] * it_bytes / r["seconds"]["linear_solve"] / 1e9, 1) if r["seconds"]["linear_solve"] > 0 else None
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = {
        "value": head["value"], "unit": "Newton iterations/s", "cores": cores, "kind": "port",
        "host": dict(share, cpu_model=cpu_model, logical_cpus=os.cpu_count(), numa_interleave=interleaved,
                     thread_sweep_newton_its_per_s={str(k): round(v, 3) for k, v in sweep.items()},
                     cores_note="threads = the fastest of the sweep; candidates are bounded by the physical cores in the affinity mask and by the cgroup CPU quota"),
        "linear_solve_GBps": ls_gbps(head),
        "multi_thread": None if mt is None else {"threads": threads, "value": mt["value"], "seconds": mt["seconds"], "newton_iterations": mt["newton_iterations"],
                                                 "linear_solve_GBps": ls_gbps(mt)},
        "sample": "the first %d Newton iterations of the same %d-cell case and time-step control on the CPU restatement "
                  "(oracle/), %d thread(s): %s, BiCGStab to 1e-2; %d linear iterations in all; %.1f s of wall time"
                  % (head["newton_iterations"], case["Nb"], cores,
                     "OpenMP assembly, block-Jacobi ILU0 over %d row ranges (what %d Flow MPI ranks factor)" % (cores, cores) if cores > 1
                     else "natural-order block ILU0 (what one Flow rank factors)",
                     head["linear_iterations"], head["cpu_seconds"]),
        "newton_iterations": head["newton_iterations"], "linear_iterations": head["linear_iterations"], "seconds": head["seconds"],
        "single_thread": {"value": st["value"], "newton_iterations": st["newton_iterations"], "linear_iterations": st["linear_iterations"],
                          "seconds": st["seconds"], "linear_solve_GBps": ls_gbps(st)},
    }
    return out


COMM_SCOPES = ("halo", "allreduce", "cpr_gather")


def single_domain_comparator(detail_path, n, steps, warmup, preconditioner, flags_default):
    """Linear iterations per Newton iteration of the ONE-domain run of this bench that a decomposed run's count is put beside - read from the
    detail file an N = 1 run left on this box (the driver runs N = 1 first), and only if that run was the same case: same cells per GPU, same
    window, same preconditioner, default ordering / smoother flags on both sides.  None otherwise: no typed-in figure stands in for a
    measurement nobody made on this box."""
    try:
        with open(detail_path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    try:
        if d.get("n_gpus") != 1 or d["config"]["cells_per_gpu"] != n ** 3 or d["steps"] != steps or d["warmup"] != warmup:
            return None
        if not flags_default or d["config"].get("ilu_ordering_chosen_by") != "library default (auto)":
            return None
        if d.get("preconditioner") == preconditioner:
            return round(float(d["linear_iterations_per_newton"]), 2)
        side = d.get("cpr" if preconditioner in ("cpr", "cpr_trueimpes") else preconditioner)
        return round(float(side["linear_iterations_per_newton"]), 2) if side and "linear_iterations_per_newton" in side else None
    except (KeyError, TypeError, ValueError):
        return None


def comm_summary(every):
    """every: per rank {scope: (launches, total_ms)} of the profiler's communication spans -> per scope the launches per rank and the average /
    total milliseconds as max and mean over the ranks (a scope no rank recorded is left out)"""
    out = {}
    for k in COMM_SCOPES:
        avg = [e[k][1] / e[k][0] for e in every if e.get(k, (0, 0.0))[0]]
        tot = [e.get(k, (0, 0.0))[1] for e in every]
        if avg:
            out[k] = {"launches_per_rank": max(e.get(k, (0, 0.0))[0] for e in every), "avg_ms_max": round(max(avg), 5), "avg_ms_mean": round(sum(avg) / len(avg), 5),
                      "total_ms_max": round(max(tot), 3), "total_ms_mean": round(sum(tot) / len(tot), 3)}
    return out


LINE_LIMIT = 4096   # bytes of the one line on stdout (round 4's 22 KB line could not be parsed from the driver's 8 KB tail)


def _r(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def compact_line(out, detail_path):
    """The ONE line on stdout: the contract's keys, `config`, `roofline`, `cpu_baseline`, one number per extra window.  Everything
    else (kernel scopes per window, reports, time steps, thread sweep, device) is in the file `detail` names."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _r(out[k]) for k in keep}
    line["config"] = out["config"]
    line["newton_iterations_per_s_global"] = _r(out["newton_iterations_per_s_global"])
    line["linear_iterations_per_newton"] = _r(out["linear_iterations_per_newton"], 2)
    line["linear_solve_GBps"] = out["linear_solve_GBps"]
    ro = out["roofline"]
    line["roofline"] = {k: _r(ro[k]) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "algorithmic_bytes_per_launch",
                                               "frac_of_stream_ceiling")}
    line["roofline"]["traffic_source"] = ro["traffic_source"] if len(str(ro["traffic_source"])) < 80 else str(ro["traffic_source"])[:77] + "..."
    if ro.get("full_spmv"):
        line["roofline"]["full_spmv"] = {k: ro["full_spmv"][k] for k in ("avg_launch_ms", "achieved", "frac")}
    if isinstance(line["roofline"].get("kernel"), str) and len(line["roofline"]["kernel"]) > 120:
        line["roofline"]["kernel"] = line["roofline"]["kernel"][:117] + "..."
    if out.get("product_form"):
        line["half_product"] = bool(out["product_form"]["half_product"])
    if out.get("plugin"):
        line["plugin"] = {"error": out["plugin"]["error"][:120]} if "error" in out["plugin"] else {k: out["plugin"][k] for k in ("t_copy_ms", "t_copy_ms_pageable", "t_factor_ms", "t_solve_ms", "GBps_h2d")}
    # the other kernels of the window behind `value`: algorithmic GB/s (bytes per launch in DESIGN.md section 4)
    line["kernel_GBps"] = {k: v["algorithmic_GBps"] for k, v in out["kernels"].items()}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb["sample"], "single_thread_value": _r(cb["single_thread"]["value"]),
                                "cpu_model": cb["host"]["cpu_model"], "linear_solve_GBps": cb["linear_solve_GBps"]}
        line["gpu_over_cpu"] = _r(out["gpu_over_cpu"], 2)

    def one(w):   # one number (or the error) per extra window
        if not w:
            return None
        if "error" in w:
            return {"error": w["error"][:120]}
        o = {"value": _r(w["value"], 2), "lin_its": _r(w["linear_iterations_per_newton"], 2)}
        if "steady_state" in w:
            o["steady"] = _r(w["steady_state"]["value"], 2)
        return o
    line["steady_state"] = one(out.get("steady_state"))
    for k in ("cpr", "cpr_quasiimpes", "cpr_amg_jacobi_smoother", "cpr_reuse_setup_2", "cpr_reuse_setup_2_sync"):
        if out.get(k) is not None:
            line[k] = one(out[k])
    line["cpr_amg_ilu_levels"] = out.get("cpr_amg_ilu_levels")
    line["stream_read_GBps"] = out["stream_ceiling"]["read_GBps"]
    if out.get("rccl"):
        line["rccl"] = {"nranks": out["rccl"]["nranks"], "kind": out["rccl"]["kind"]}
    if out.get("comm"):
        line["comm"] = out["comm"]
    line["device"] = out["device"].get("name")
    line["detail"] = detail_path
    return line


def fit_line(line):
    """the line as JSON text, under LINE_LIMIT whatever happened: should something unforeseen make it long (an error text in an extra window,
    a long device name), the extras go first - never the contract's keys, `config.workload`, `roofline` or `cpu_baseline` - and the line
    says what was dropped; the detail file keeps everything"""
    txt = json.dumps(line)
    dropped = []
    for k in ("cpr_reuse_setup_2_sync", "cpr_amg_jacobi_smoother", "cpr_reuse_setup_2", "cpr_quasiimpes", "plugin", "cpr", "comm", "kernel_GBps", "steady_state", "rccl", "device"):
        if len(txt) < LINE_LIMIT:
            break
        if k in line:
            del line[k]
            dropped.append(k)
            line["dropped_for_length"] = dropped
            txt = json.dumps(line)
    if len(txt) >= LINE_LIMIT:   # last resort: the free texts
        line["cpu_baseline"] = {k: v for k, v in (line.get("cpu_baseline") or {}).items() if k != "cpu_model"} or None
        if isinstance(line.get("cpu_baseline"), dict) and "sample" in line["cpu_baseline"]:
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:200]
        line["config"] = dict(line["config"], workload=line["config"]["workload"][:200])
        txt = json.dumps(line)
    return txt


def launch_plan(gpus, env, device_count):
    """What `python bench.py --gpus N` has to do before any GPU call: ("inline", world) - run in this process (N = 1, or
    a rank started by torch.distributed.run) - or ("spawn", N) - start N ranks as children.  Raises SystemExit on a
    request that cannot be met; never silently runs fewer GPUs than asked for."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d" % (gpus, world))
        return ("inline", world)
    if gpus == 1:
        return ("inline", 1)
    if device_count < gpus:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible - refusing to run a smaller job" % (gpus, device_count))
    return ("spawn", gpus)


def spawn_command(gpus, argv, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus, "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def kernel_source_hash():
    """sha256 over the kernel sources: ties a committed PMC traffic summary to the build it was measured on"""
    h = hashlib.sha256()
    for f in ("solver.hip", "assemble.hip", "internal.hpp"):
        with open(os.path.join(ROOT, "opm-autodiff_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(*kernel_substrs):
    """HBM-side traffic of a kernel per launch from the newest committed PMC summary (PMC passes need their own
    rocprofv3 runs, tools/pmc_quick.sh + tools/pmc_to_json.py) - only if it was measured on THESE kernel sources."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    for tj in reversed(files):
        with open(tj) as f:
            d = json.load(f)
        if d.get("kernel_source_sha16") != kernel_source_hash():
            continue
        vals = [v["traffic_bytes_per_launch"] for k, v in d["kernels"].items() if all(sub in k for sub in kernel_substrs)]
        if vals:
            return sum(vals) / len(vals), os.path.relpath(tj, ROOT)
    return None, "no PMC summary under profiles/ was measured on the present kernel sources (sha16 %s)" % kernel_source_hash()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--cells-per-edge", dest="n", type=int, default=100, help="cells per edge of the per-GPU cube (behind torch.distributed.run use the long form: its own parser takes a bare --n for an abbreviation of its options)")
    ap.add_argument("--reorder", default=None, help="ILU0 ordering; default: the library's own choice (opmhip_default_config: auto), reported in config.ilu_ordering")
    ap.add_argument("--chain-length", type=int, default=0, help="rows per chain of the line-coloured ILU0 ordering; 0: the library's choice (10 at 10^6 cells)")
    ap.add_argument("--full-line", action="store_true", help="tools/ only: print the full record as the one line (tens of KB: the driver could not parse that from its tail of stdout)")
    ap.add_argument("--detail", default=None, help="file for the full record (per-window kernel scopes, reports, time steps); default gpurun_out/bench_detail.json (N > 1: bench_detail_nN.json)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steady-after", type=int, default=200, help="second timed window starts at this Newton iteration (0: none)")
    ap.add_argument("--steady-steps", type=int, default=100)
    ap.add_argument("--preconditioner", default="ilu0", choices=["ilu0", "cpr", "cpr_trueimpes", "cpr_quasiimpes"], help="--linear-solver-configuration of the run behind `value`")
    ap.add_argument("--cpr-reuse-setup", type=int, default=3, choices=[0, 1, 2, 3], help="Flow's --cpr-reuse-setup for the CPR runs: when the hierarchy's structure is built anew (3 = never, the default of Flow)")
    ap.add_argument("--cpr-amg-ilu-levels", type=int, default=None, help="CPR runs: this many of the pressure AMG's finest levels smooth with ILU0 (the reference's AMG smoother) instead of damped Jacobi; default: the library's choice (opmhip_default_config: -1 = level 0 where the block ordering has at most three colours), reported in cpr_amg_ilu_levels")
    ap.add_argument("--fused-reductions", dest="fused_reductions", action="store_true", default=None,
                    help="BiCGStab with one reduction per half iteration (opmhip_config.fused_reductions). Default: off on one GPU (there it is even, profiles/r06_fused_ab.txt), "
                         "on for --gpus N > 1 with ILU0, where every reduction is an all-reduce over xGMI (two per iteration instead of four); --no-fused-reductions: off")
    ap.add_argument("--no-fused-reductions", dest="fused_reductions", action="store_false")
    ap.add_argument("--no-cpr-side-run", action="store_true", help="skip the side runs with the CPR preconditioners (extra keys `cpr`, `cpr_quasiimpes`)")
    a = ap.parse_args()

    import torch
    plan, world = launch_plan(a.gpus, os.environ, torch.cuda.device_count())  # device_count() does not initialise the GPU
    if plan == "spawn":
        # children first, the GPU never: this process only waits for torch.distributed.run and hands on its exit code
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        raise SystemExit(subprocess.call(spawn_command(a.gpus, sys.argv[1:], port)))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        # control plane (rendezvous, barrier, max of the elapsed time, broadcast of the RCCL id) over gloo; the data path
        # - halos and scalar all-reduces inside every Newton iteration - runs on RCCL inside libopmhip, on its own stream
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    # The product never goes through torch: libopmhip selects device `local_rank` itself and opmhip_create fails loudly
    # (OPMHIP_NO_DEVICE) when there is none - no CPU fallback.  torch only brackets the timed region when it sees the card.
    torch_sees_gpu = torch.cuda.is_available() and torch.cuda.device_count() > local_rank
    if torch_sees_gpu:
        torch.cuda.set_device(local_rank)
    sync_models = []   # every context with work in flight; all of the product's work runs on its contexts' own streams

    def device_sync():
        if torch_sees_gpu:
            torch.cuda.synchronize()
        for mdl in sync_models:
            mdl.synchronize()   # opmhip_synchronize: raises on a device error

    pkg = importlib.import_module("opm-autodiff_amd")
    n = a.n
    fused_side = bool(a.fused_reductions)                                       # the CPR side runs: only when asked for
    a.fused_reductions = (world > 1 and a.preconditioner == "ilu0") if a.fused_reductions is None else a.fused_reductions
    skw = dict(device_id=local_rank, reorder=a.reorder, tolerance=1e-2, maxit=200, ilu_relaxation=0.9, chain_length=a.chain_length,
               preconditioner=a.preconditioner, cpr_reuse_setup=a.cpr_reuse_setup, cpr_amg_ilu_levels=a.cpr_amg_ilu_levels,
               fused_reductions=int(a.fused_reductions))
    if world == 1:
        case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
        src = pkg.decks.five_spot_source(case, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY * (n / 100.0) ** 2)
        model = pkg.capi.HipModel(case, **skw)
        layout = (1, 1, 1)
    else:
        # weak scaling: n^3 cells per GPU, global grid (px n) x (py n) x (pz n), restricted additive Schwarz
        layout = pkg.ras.block_layout(world)
        case = pkg.ras.cartesian_subdomain_case(n, world, rank, state="mixed", heterogeneous=False)
        src = case["source"]
        uid = [pkg.capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        model = pkg.capi.HipModel(case, comm=("rccl", world, rank, uid[0]), **skw)
    sync_models.append(model)
    # N > 1: what RCCL itself says about the communicator the data path runs on, and one all-reduce through it.  A run whose
    # communicator does not span --gpus ranks is not the run that was asked for: every rank exits non-zero.
    rccl = None
    if world > 1:
        info = model.comm_info()
        s0, s1 = model.comm_selftest()
        gathered = [None] * world
        dist.all_gather_object(gathered, (info["rank"], info["device"], socket.gethostname()))
        rccl = {"nranks": info["nranks"], "kind": info["kind"], "rank_devices": [g[1] for g in sorted(gathered)],
                "hosts": sorted({g[2] for g in gathered}), "selftest_sum": [s0, s1], "selftest_expected": [world * (world + 1) / 2.0, 2.0 * world]}
        if info["kind"] != "rccl" or info["nranks"] != a.gpus or rccl["selftest_sum"] != rccl["selftest_expected"]:
            raise SystemExit("bench.py: RCCL communicator spans %d rank(s) (kind %s, self-test %r), --gpus %d was asked for" % (info["nranks"], info["kind"], rccl["selftest_sum"], a.gpus))
    model.set_state(case["pv"], case["meaning"])
    model.set_source(src)
    sim = make_simulation(pkg, model)
    Nb, nnzb = case["Nb"], len(case["col"])
    product_form = model.product_form()     # what opmhip_config.half_product resolved to on this pattern
    B = alg_bytes(Nb, nnzb, product_form["rest_blocks"] if product_form["half_product"] else 0)

    def use_cpr_of(mdl):
        return getattr(mdl, "_bench_preconditioner", "ilu0") != "ilu0"
    model._bench_preconditioner = a.preconditioner

    def barrier():
        device_sync()
        if dist is not None:
            dist.barrier()
        device_sync()

    def timed_window(steps):
        """EXACTLY `steps` Newton iterations between two barriers; max over ranks; kernel scopes of every 4th solve"""
        barrier()
        model.profile_enable(PROFILE_EVERY)
        rep0 = pkg.newton.SimulatorReportSingle()
        rep0 += sim.report
        ts0, tf0, h0 = sim.timesteps_done, sim.timesteps_failed, len(sim.history)
        t0 = time.perf_counter()
        for _ in range(steps):
            sim.next_newton_iteration()
        barrier()
        elapsed = time.perf_counter() - t0
        prof = model.profile()
        model.profile_enable(False)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        rep = sim.report
        kernels = {}
        boundary_share = None
        if prof.get("spmv_boundary", (0, 0.0))[0]:   # decomposed runs: one product = interior launch + boundary launch
            prof["spmv"] = (prof["spmv"][0], prof["spmv"][1] + prof["spmv_boundary"][1])
            boundary_share = round(prof["spmv_boundary"][1] / prof["spmv"][1], 4)
        prof.pop("spmv_boundary", None)
        pf = model.product_form()   # this window's model: a CPR context keeps the whole product, whatever the headline context took
        Bm = dict(alg_bytes(Nb, nnzb, pf["rest_blocks"] if pf["half_product"] else 0))
        if use_cpr_of(model):
            lv = model.cpr_levels()
            Bm["cpr_amg"] = cpr_amg_bytes(Nb, [int(v) for v in lv[0]], [int(v) for v in lv[1]], model.ordering_info()["cpr_amg_ilu_levels"])
        # what the profiled solves were: solves = factorisations, BiCGStab iterations = products / 2 (two per iteration; a solve that
        # stops on a first half adds one) - so that launches x avg_ms can be put beside the window's ms_per_step
        solves_profiled = prof.get("ilu_factor", (0, 0.0))[0] // (2 if use_cpr_of(model) else 1)   # CPR: the value set-up of the hierarchy is a second scope of that class
        its_profiled = prof.get("spmv", (0, 0.0))[0] / 2.0
        for name, (cnt, ms) in prof.items():
            if cnt and Bm.get(name):
                avg = ms / cnt
                kernels[name] = {"launches": cnt, "avg_ms": round(avg, 5), "algorithmic_GBps": round(Bm[name] / avg / 1e6, 1),
                                 "algorithmic_bytes_per_launch": int(Bm[name])}
                if name in ("spmv", "ilu_apply", "vector", "cpr_amg"):
                    kernels[name]["iterations_profiled"] = its_profiled
                    kernels[name]["solves_profiled"] = solves_profiled
        # communication spans (decomposed runs; they overlap the kernel scopes): per rank launches and milliseconds, then max / mean over
        # the ranks - with the iteration count beside them, so that "slower because of more iterations" and "slower because of the
        # exchanges" come apart (parity caveat)
        comm = None
        if dist is not None:
            mine = {k: (prof.get(k, (0, 0.0))[0], prof.get(k, (0, 0.0))[1]) for k in COMM_SCOPES}
            every = [None] * world
            dist.all_gather_object(every, mine)
            comm = comm_summary(every)
        ls_bytes = sum(Bm[k] * prof[k][0] for k in ("spmv", "ilu_apply", "ilu_factor", "vector"))
        ls_ms = sum(prof[k][1] for k in ("spmv", "ilu_apply", "ilu_factor", "vector"))
        return {"elapsed": elapsed, "steps": steps, "kernels": kernels, "spmv_boundary_share_of_time": boundary_share, "comm": comm,
                "profiled": {"every": PROFILE_EVERY, "solves": solves_profiled, "linear_iterations": its_profiled,
                             "linear_iterations_per_profiled_solve": (its_profiled / solves_profiled) if solves_profiled else None},
                "linear_iterations_per_newton": (rep.total_linear_iterations - rep0.total_linear_iterations) / steps,
                "timesteps_completed": sim.timesteps_done - ts0, "timesteps_chopped": sim.timesteps_failed - tf0,
                "time_steps_days": [round(h[0] / DAY, 3) for h in sim.history[h0:]],
                "linear_solve_GBps": round(ls_bytes / ls_ms / 1e6, 1) if ls_ms > 0 else None,
                "report": {"assemble_time": rep.assemble_time - rep0.assemble_time,
                           "linear_solve_setup_time": rep.linear_solve_setup_time - rep0.linear_solve_setup_time,
                           "linear_solve_time": rep.linear_solve_time - rep0.linear_solve_time,
                           "update_time": rep.update_time - rep0.update_time}}

    for _ in range(a.warmup):
        sim.next_newton_iteration()
    W = timed_window(a.steps)
    elapsed, kernels = W["elapsed"], W["kernels"]
    steady = None
    def guarded(what, fn):
        """the extra windows must not cost the run its line: a failure in one of them is reported in its place (all ranks
        take the same branch: the Newton and time-step decisions hang on all-reduced numbers)"""
        try:
            return fn()
        except Exception as e:   # noqa: BLE001
            return {"error": "%s: %s: %s" % (what, type(e).__name__, e)}

    def steady_window():
        done = a.warmup + a.steps
        while done < a.steady_after:   # untimed: carries the simulation into its steady phase
            sim.next_newton_iteration()
            done += 1
        S = timed_window(a.steady_steps)
        return {"from_newton_iteration": done, "steps": S["steps"], "ms_per_step": 1e3 * S["elapsed"] / S["steps"],
                  "newton_iterations_per_s_global": S["steps"] / S["elapsed"], "value": S["steps"] * world / S["elapsed"],
                  "linear_iterations_per_newton": S["linear_iterations_per_newton"], "time_steps_days": S["time_steps_days"],
                  "timesteps_chopped": S["timesteps_chopped"], "linear_solve_GBps": S["linear_solve_GBps"], "report": S["report"],
                  "kernels": S["kernels"], "profiled": S["profiled"]}

    if a.steady_after > 0 and a.steady_steps > 0:
        steady = guarded("steady-state window", steady_window)
    # the same workload with the CPR preconditioners, side by side, each in its own context with its own warm-up and timed
    # windows: "cpr" = cpr_trueimpes (what the name means in Flow, setupPropertyTree.cpp:62-76) and "cpr_quasiimpes"
    cpr_sides = {}
    def cpr_window(prec, **over):
        def run():
            nonlocal sim, model
            model2 = pkg.capi.HipModel(case, **dict(skw, preconditioner=prec, fused_reductions=int(fused_side), **over))
            model2._bench_preconditioner = prec
            model2.set_state(case["pv"], case["meaning"])
            model2.set_source(src)
            sim_main, model_main = sim, model
            sim, model = make_simulation(pkg, model2), model2
            sync_models.append(model2)
            try:
                for _ in range(a.warmup):
                    sim.next_newton_iteration()      # includes the one-time host-side aggregation of the pressure AMG
                C1 = timed_window(a.steps)
                side = {"value": C1["steps"] / C1["elapsed"], "ms_per_step": 1e3 * C1["elapsed"] / C1["steps"], "steps": C1["steps"],
                        "linear_iterations_per_newton": C1["linear_iterations_per_newton"], "report": C1["report"], "kernels": C1["kernels"], "profiled": C1["profiled"],
                        "amg_levels": [int(v) for v in model2.cpr_levels()[0]], "cpr_amg_ilu_levels": model2.ordering_info()["cpr_amg_ilu_levels"]}
                if a.steady_after > 0 and a.steady_steps > 0:
                    done = a.warmup + a.steps
                    while done < a.steady_after:
                        sim.next_newton_iteration()
                        done += 1
                    C2 = timed_window(a.steady_steps)
                    side["steady_state"] = {"from_newton_iteration": done, "steps": C2["steps"], "value": C2["steps"] / C2["elapsed"],
                                            "ms_per_step": 1e3 * C2["elapsed"] / C2["steps"],
                                            "linear_iterations_per_newton": C2["linear_iterations_per_newton"], "timesteps_chopped": C2["timesteps_chopped"],
                                            "kernels": C2["kernels"], "profiled": C2["profiled"]}
                return side
            finally:
                sim, model = sim_main, model_main
                sync_models.remove(model2)
                del model2
        return run

    if a.preconditioner == "ilu0" and world == 1 and not a.no_cpr_side_run:
        for prec in ("cpr", "cpr_quasiimpes"):
            cpr_sides[prec] = guarded("CPR side run (%s)" % prec, cpr_window(prec))
        if a.cpr_amg_ilu_levels != 0:   # the same with damped Jacobi on every level of the pressure AMG
            cpr_sides["cpr_amg_jacobi_smoother"] = guarded("CPR side run (cpr, Jacobi-smoothed AMG)", cpr_window("cpr", cpr_amg_ilu_levels=0))
        if a.cpr_reuse_setup == 3:   # Flow's other --cpr-reuse-setup worth a line: the hierarchy's structure follows the state
            # ... rebuilt on a host thread beside the solves (opmhip_config.cpr_async_setup), and - the reference's rule to the letter -
            # by the solve that finds the rule met (0.28 s of host work inside the window where it happens)
            cpr_sides["cpr_reuse_setup_2"] = guarded("CPR side run (cpr, --cpr-reuse-setup=2, rebuild beside the solves)", cpr_window("cpr", cpr_reuse_setup=2, cpr_async_setup=1))
            cpr_sides["cpr_reuse_setup_2_sync"] = guarded("CPR side run (cpr, --cpr-reuse-setup=2)", cpr_window("cpr", cpr_reuse_setup=2))
    # what a kernel that only streams reaches on THIS card (reads the Jacobian's values once per launch)
    stream = guarded("stream_read probe", lambda: {"ms": model.time_kernel("stream_read", reps=20)})
    # the whole block-CSR product (SURVEY 8d's 579.44 MB at 100^3) back to back, 20 launches: with the half-product form in force no kernel
    # of a solve computes it any more, so it is timed here, outside the solve, for the record north_star asks for
    full_spmv = guarded("full SpMV probe", lambda: {"ms": model.time_kernel("spmv", reps=20)}) if a.preconditioner == "ilu0" else {}
    stream_ms = stream.get("ms")
    stream_GBps = 72.0 * nnzb / stream_ms / 1e6 if stream_ms else None

    # The drop-in at its own speed: opmhip_solve_system with HOST pointers - what a Flow that applies INTEGRATION.md
    # section 1 only (--accelerator-mode=hip, assembly on the host) gets per linear solve: the copy of 500 MB of values over PCIe, the
    # factorisation, BiCGStab.  The Jacobian and residual of the present state are fetched once and handed to a second context three times,
    # with the arrays registered for DMA (opmhip_config.pin_host_arrays, what host/hipSolverBackend.hpp asks for) and without.
    def plugin_window():
        jac, res = model.assemble(DAY, 0)
        rp, ci = case["rowptr"], case["col"]
        o = {}
        for pin in (1, 0):
            sv = pkg.capi.HipSolver(device_id=local_rank, tolerance=1e-2, maxit=200, ilu_relaxation=0.9, pin_host_arrays=pin)
            runs = []
            for k in range(4):
                r = sv.solve_system(Nb, rp if k == 0 else None, ci if k == 0 else None, jac, res)
                x = sv.get_result()
                runs.append((r.t_copy, r.t_factor, r.t_solve, r.iterations, bool(r.converged)))
            later = runs[1:]      # the first call orders the pattern and (pin = 1) registers the arrays
            key = "" if pin else "_pageable"
            o["t_copy_ms" + key] = round(1e3 * min(t[0] for t in later), 3)
            if pin:
                o["t_factor_ms"] = round(1e3 * min(t[1] for t in later), 3)
                o["t_solve_ms"] = round(1e3 * min(t[2] for t in later), 3)
                o["iterations"] = later[0][3]
                o["converged"] = later[0][4]
                o["first_call_t_copy_ms"] = round(1e3 * runs[0][0], 1)
                o["GBps_h2d"] = round((72.0 * nnzb + 24.0 * Nb) / min(t[0] for t in later) / 1e9, 1)
            del sv
        o["bytes_h2d"] = 72 * nnzb + 24 * Nb
        return o
    plugin = guarded("plug-in window", plugin_window) if (world == 1 and a.preconditioner == "ilu0" and not a.no_cpr_side_run) else None

    dots_separate = os.environ.get("OPMHIP_TUNING", "") == "1" and os.environ.get("OPMHIP_DOTS_SEPARATE", "0") not in ("", "0")
    sp = kernels.get("spmv", {"avg_ms": float("nan"), "algorithmic_GBps": float("nan")})
    ok = sp["algorithmic_GBps"] == sp["algorithmic_GBps"]
    traffic, traffic_src = (None, "single-GPU 100^3 line-colouring runs only")
    chosen = model.ordering_info()   # what the library's defaults resolved to (or what the flags forced)
    if world == 1 and n == 100 and chosen["ilu_ordering"] == "line_coloring" and chosen["chain_length"] == 10:
        # (half-product form: the product kernels of the solve are the <.., true> instantiations; the whole product timed by the probe is not)
        traffic, traffic_src = pmc_traffic("k_spmv_pipe_st<", ", true>") if product_form["half_product"] else pmc_traffic("k_spmv")
    out = {
        "metric": "Newton iterations/sec, 1M-cell 3-phase black-oil (assembly + ILU0/BiCGStab solve + update)",
        # Whole-job aggregate.  Weak scaling: every rank advances the SAME coupled Newton iteration on its own 1M-cell
        # subdomain, so the work done per second is (global iterations/s) x (number of 1M-cell subdomains); the plain
        # global rate of the coupled (N x 1M-cell) problem is given beside it.
        "value": a.steps * world / elapsed,
        "unit": "Newton iterations/s" if world == 1 else "Newton iterations/s x 1M-cell subdomains (= newton_iterations_per_s_global x n_gpus)",
        "newton_iterations_per_s_global": a.steps / elapsed,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "synthetic %dx%dx%d Cartesian 3-phase black-oil (BASELINE configs[1]), SPE1 fluid, homogeneous "
                               "100 mD, gas cap + undersaturated oil, five-spot rate sources, adaptive time steps 1 d -> 10 d (Flow's "
                               "pid+newtoniteration control and 0.33 chop)" % (n, n, n),
                   "cells_per_gpu": Nb, "blocks_per_gpu": nnzb, "ilu_ordering": chosen["ilu_ordering"], "ilu_chain_length": chosen["chain_length"],
                   "ilu_colors": chosen["colors"], "ilu_ordering_chosen_by": "library default (auto)" if a.reorder is None else "--reorder", "linear_tol": 1e-2,
                   "ilu_relaxation": 0.9, "bicgstab_reductions_per_iteration": 2 if a.fused_reductions else 4, "parallelism": "1 GPU" if world == 1 else "RAS domain decomposition %dx%dx%d, %s per GPU, halos + all-reduces over RCCL" % (layout + ("block-Jacobi ILU0" if a.preconditioner == "ilu0" else "one CPR (%s) per subdomain" % a.preconditioner,))},
        "linear_iterations_per_newton": W["linear_iterations_per_newton"],
        "timesteps_completed": W["timesteps_completed"], "timesteps_chopped": W["timesteps_chopped"],
        "time_steps_days": W["time_steps_days"],
        "linear_solve_GBps": W["linear_solve_GBps"],
        "report": W["report"],
        "kernels": kernels,
        "profiled": W["profiled"],
        "spmv_boundary_share_of_time": W["spmv_boundary_share_of_time"],
        "steady_state": steady,
        # same case, CPR instead of ILU0 as the preconditioner of BiCGStab (extra information; `value` is the run above):
        # "cpr" is Flow's cpr = cpr_trueimpes, "cpr_quasiimpes" the quasi-IMPES variant
        "cpr": cpr_sides.get("cpr"),
        "cpr_quasiimpes": cpr_sides.get("cpr_quasiimpes"),
        # "cpr" with --cpr-reuse-setup=2: the structure of the pressure hierarchy is built anew whenever a solve took more than 10
        # iterations - on a host thread beside the solves (cpr_async_setup = 1: the new structure takes over at the first solve
        # boundary after it is ready) and, "_sync", by the solve itself (0.3 s of host time inside the windows where it happens);
        # the two runs above keep Flow's default 3 (never)
        # the CPR runs smooth level 0 of the pressure AMG with ILU0, the reference's AMG smoother (opmhip_config.cpr_amg_ilu_levels =
        # --cpr-amg-ilu-levels, default 1 here); "cpr_amg_jacobi_smoother": "cpr" with damped Jacobi on every level instead (the library's default)
        "cpr_amg_ilu_levels": (cpr_sides.get("cpr") or {}).get("cpr_amg_ilu_levels"),   # in force in the CPR side runs (library's choice unless --cpr-amg-ilu-levels)
        "cpr_amg_jacobi_smoother": cpr_sides.get("cpr_amg_jacobi_smoother"),
        "cpr_reuse_setup_2": cpr_sides.get("cpr_reuse_setup_2"),
        "cpr_reuse_setup_2_sync": cpr_sides.get("cpr_reuse_setup_2_sync"),
        "preconditioner": a.preconditioner,
        # opmhip_solve_system with host pointers on this case's Jacobian (second context): milliseconds per linear solve by phase
        "plugin": plugin,
        "rccl": rccl,
        # decomposed runs: the communication spans of the profiled solves of the window behind `value` (halo: pack -> exchange -> ghosts in, on
        # the halo stream beside the interior tiles; allreduce: local sums -> all-reduce; cpr_gather: the joined level's all-gather + cycle),
        # max / mean over the ranks, and the iteration count next to the one-domain figure of the same case (N = 1 run of this bench)
        "comm": None if W["comm"] is None else dict(W["comm"], profiled_solves=W["profiled"]["solves"], profiled_every=PROFILE_EVERY,
                                                    linear_iterations_per_newton=round(W["linear_iterations_per_newton"], 2),
                                                    single_domain_linear_iterations_per_newton=single_domain_comparator(
                                                        os.path.join(ROOT, "gpurun_out", "bench_detail.json"), n, a.steps, a.warmup, a.preconditioner,
                                                        a.reorder is None and a.chain_length == 0 and a.cpr_amg_ilu_levels is None and a.cpr_reuse_setup == 3)),
        "device": device_info(torch, local_rank),
        "stream_ceiling": {"read_GBps": round(stream_GBps, 1) if stream_GBps else None, "bytes_per_launch": 72 * nnzb,
                           "avg_launch_ms": round(stream_ms, 5) if stream_ms else None, "error": stream.get("error"),
                           "kernel": "k_stream_read: the Jacobian's value array read once, 16-B loads, nothing else (back to back, 20 launches)"},
        "product_form": product_form,
        "roofline": {"bound": "hbm", "kernel": ("k_spmv_pipe_st<UADD>: the block-CSR product in its half-product form - the matrix without its U part "
                                                "(%d of %d blocks) + the backward sweep's row sums" % (product_form["rest_blocks"], nnzb))
                                               if product_form["half_product"] else "k_spmv (block-CSR SpMV, 3x3 double blocks)",
                     "achieved": sp["algorithmic_GBps"],
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (sp["algorithmic_GBps"] / HBM_PEAK_GBS) if ok else None,
                     "frac_of_stream_ceiling": (sp["algorithmic_GBps"] / stream_GBps) if (ok and stream_GBps) else None,
                     "traffic": traffic, "traffic_unit": "bytes per launch (L2 fabric side; FETCH_SIZE x 2 + WRITE_SIZE, KiB -> B)",
                     "traffic_source": traffic_src, "avg_launch_ms": sp["avg_ms"],
                     # achieved / frac are quoted on SURVEY 8d's plain block-SpMV bytes (579.44 MB at 100^3).  The BiCGStab scalar
                     # products behind a product ride in the kernel (no wells here, unless OPMHIP_DOTS_SEPARATE is set): their
                     # second operand (24 B per row) is part of what the kernel reads but not of the figure the fraction is taken of
                     "algorithmic_bytes_per_launch": B["spmv"],
                     # the whole product (579.44 MB), timed back to back outside the solve: {avg_launch_ms, achieved GB/s, frac of 8 TB/s}
                     "full_spmv": None if not full_spmv.get("ms") else {"avg_launch_ms": round(full_spmv["ms"], 5), "achieved": round(alg_bytes_plain(Nb, nnzb)["spmv"] / full_spmv["ms"] / 1e6, 1),
                                                                         "frac": round(alg_bytes_plain(Nb, nnzb)["spmv"] / full_spmv["ms"] / 1e6 / HBM_PEAK_GBS, 4),
                                                                         "algorithmic_bytes_per_launch": alg_bytes_plain(Nb, nnzb)["spmv"]},
                     "operand_bytes_per_launch": B["spmv"] if dots_separate else B["spmv_operands"],
                     "scalar_products": "k_dots behind every product (OPMHIP_DOTS_SEPARATE)" if dots_separate else "folded into the kernel (one partial sum per workgroup)"},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, case, src, nnzb)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        # the full record goes to a file; the line on stdout is what the driver parses and stays under 4 KB
        # (N > 1 runs write a file of their own: the N = 1 record stays where the later runs of a scaling series look for their comparators)
        detail = a.detail or os.path.join(ROOT, "gpurun_out", "bench_detail.json" if world == 1 else "bench_detail_n%d.json" % world)
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail)), exist_ok=True)
            with open(detail, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            detail = "not written: %s" % e
        line = fit_line(compact_line(out, os.path.relpath(detail, ROOT) if os.path.isabs(detail) and detail.startswith(ROOT) else detail))
        print(json.dumps(out) if a.full_line else line)
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
