import importlib, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench, helpers
pkg = importlib.import_module("opm-autodiff_amd")
c4, _, _ = helpers.norne_shaped_case(pkg)
rng = np.random.default_rng(3)
cells = rng.choice(c4["Nb"], 24, replace=False)
s = np.zeros((c4["Nb"], 3)); q = 200.0 / 86400.0
s[cells[:12], 1] = q; s[cells[12:], 0] = -q
src = np.ascontiguousarray(s.reshape(-1))
reorder = sys.argv[1] if len(sys.argv) > 1 else "line_coloring"
m = pkg.capi.HipModel(c4, reorder=reorder, tolerance=1e-2, maxit=200, ilu_relaxation=0.9, chain_length=10)
m.set_state(c4["pv"], c4["meaning"]); m.set_source(src)
sim = bench.make_simulation(pkg, m)
for _ in range(3): sim.next_newton_iteration()
to, fr, rpc = m.ordering()
print("colours:", len(rpc), "rows per colour (first 12):", list(rpc[:12]), file=sys.stderr)
m.synchronize(); t0 = time.perf_counter(); lin = 0
for _ in range(20): lin += sim.next_newton_iteration().total_linear_iterations
m.synchronize(); el = time.perf_counter() - t0
print("%s: %.1f Newton its/s, %.1f lin/newton, %.3f ms per linear iteration (all-in)" % (reorder, 20 / el, lin / 20, 1e3 * el / max(lin, 1)), file=sys.stderr)
