import importlib, os, sys, threading, uuid, ctypes
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("opm-autodiff_amd")
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 2**20
case = pkg.decks.cartesian_case(30, 30, 30, state="mixed", heterogeneous=True)
src = pkg.decks.five_spot_source(case, rate_sm3_per_day=30.0)
m = pkg.capi.HipModel(case, reorder="line_coloring", preconditioner="cpr", cpr_reuse_setup=0, cpr_amg_ilu_levels=2)
m.set_state(case["pv"], case["meaning"]); m.set_source(src)
m.assemble(86400.0, 0, fetch=False); m.solve_jacobian_system()
f0 = free_mb()
for it in range(150):
    m.assemble(86400.0, 1, fetch=False)
    assert m.solve_jacobian_system().converged
f1 = free_mb()
print("single rank, 150 rebuilds with 2 ILU0 levels: free device memory %.1f -> %.1f MiB" % (f0, f1))
world, n = 2, 16
parts = [pkg.ras.cartesian_subdomain_case(n, world, r, state="mixed", heterogeneous=True, rate_scale=30.0) for r in range(world)]
group = "leak" + uuid.uuid4().hex
res = [None] * world
def body(r):
    c = parts[r]
    mm = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring", preconditioner="cpr_quasiimpes", cpr_reuse_setup=0, cpr_gather_rows=300)
    mm.set_state(c["pv"], c["meaning"]); mm.set_source(c["source"])
    mm.assemble(86400.0, 0, fetch=False); mm.solve_jacobian_system()
    a = free_mb()
    for it in range(60):
        mm.assemble(86400.0, 1, fetch=False)
        assert mm.solve_jacobian_system().converged
    res[r] = (a, free_mb())
ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
print("two ranks, 60 rebuilds of the joined level each: free device memory", res)
