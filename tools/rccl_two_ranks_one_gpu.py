#!/usr/bin/env python3
"""Two RCCL ranks on ONE GPU (two processes, same device): exercises the real ncclSend/ncclRecv/ncclAllReduce path of
libopmhip's decomposed solver where only a single GPU is available.  RCCL may refuse two ranks on one device
("Duplicate GPU detected"); the script reports that instead of hanging (run it under `timeout`).
    python tools/rccl_two_ranks_one_gpu.py            # parent: spawns rank 0 and rank 1
"""
import importlib, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(rank, idfile):
    import numpy as np
    pkg = importlib.import_module("opm-autodiff_amd")
    if rank == 0:
        uid = pkg.capi.comm_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                raise SystemExit("rank 1: no id file")
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    case = pkg.ras.cartesian_subdomain_case(8, 2, rank, state="mixed", heterogeneous=False, rate_scale=40.0)
    m = pkg.capi.HipModel(case, comm=("rccl", 2, rank, uid), reorder="line_coloring")
    m.set_state(case["pv"], case["meaning"])
    m.set_source(case["source"])
    drv = pkg.newton.BlackoilModelHip(m)
    rep = drv.step(2 * 86400.0)
    pv, mean = m.get_state()
    print("rank %d: newton %d linear %d  sum(p) over owned cells %.12e" % (rank, rep.total_newton_iterations, rep.total_linear_iterations,
                                                                         float(np.sum(pv.reshape(-1, 3)[:case["Nb"], 1]))), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3:
        child(int(sys.argv[1]), sys.argv[2])
        sys.exit(0)
    d = tempfile.mkdtemp()
    idfile = os.path.join(d, "rccl_id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), idfile], env=env) for r in range(2)]
    rc = 0
    t0 = time.time()
    for p in ps:
        try:
            rc |= p.wait(timeout=max(1.0, 90 - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            p.kill()
            rc |= 124
    print("exit", rc)
    sys.exit(rc)
