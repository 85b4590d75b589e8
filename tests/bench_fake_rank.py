"""One rank of `bench.py --gpus 2` WITHOUT a GPU: the product's device context (capi.HipModel) is replaced by a stand-in that answers the
calls bench.py and newton.py make of it with plausible numbers, so that bench.py's own multi-rank control plane runs for real - the gloo
rendezvous, the broadcast of the communicator id, the gathered RCCL record and its refusal rule, barriers, the MAX over the ranks' elapsed
times, the communication spans gathered over the ranks, the one line from rank 0.  Started by tests/test_dd_gloo.py through
torch.distributed.run; not product code, and nothing of libopmhip runs in it (that half is tests/test_gpu_dd.py)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("opm-autodiff_amd")
import bench  # noqa: E402


class FakeResult:
    def __init__(self, its):
        self.iterations, self.it, self.converged, self.reduction = its, float(its), 1, 5e-3
        self.t_factor, self.t_solve, self.t_copy = 1e-4, 1e-3, 1e-5


class FakeModel:
    """the call surface of capi.HipModel that bench.py / newton.py use"""
    PROF = pkg.capi.HipSolver.PROF

    def __init__(self, case, comm=None, **kw):
        assert comm is not None and comm[0] == "rccl" and len(comm[3]) == 128, comm
        self.kind, self.world, self.rank, self.uid = comm[0], comm[1], comm[2], comm[3]
        self.case, self.kw, self.it, self.solves, self.on = case, kw, 0, 0, False
        assert case["Nghost"] > 0 and "halo" in case and case["global_cells"] == self.world * case["Nb"]

    def set_state(self, pv, meaning): self.it = 0
    def set_source(self, src): assert len(src) == 3 * (self.case["Nb"] + self.case["Nghost"]) or len(src) == 3 * self.case["Nb"]
    def assemble(self, dt, iteration, fetch=False): self.it = iteration
    def convergence(self, dt, tol):
        c = np.zeros(17)
        c[9], c[10] = 1.0, 0.0
        c[11:17] = 1e-9 if self.it >= 2 else 1.0     # converges at the third iteration of every time step
        return c

    def solve_jacobian_system(self):
        self.solves += 1
        time.sleep(0.002)
        return FakeResult(5 + self.it)

    def update(self, dx, relax): return 0
    def advance_time_level(self): pass
    def update_failed(self): pass
    def relative_change(self): return 1e-3
    def begin_time_step(self, dt): pass
    def end_time_step(self, dt): pass
    def synchronize(self): pass
    def profile_enable(self, on): self.on = bool(on); self.solves = 0
    def profile(self):
        p = {k: (0, 0.0) for k in self.PROF}
        n = max(self.solves, 1)
        p.update(spmv=(12 * n, 1.2 * n), spmv_boundary=(12 * n, 0.3 * n), ilu_apply=(12 * n, 1.7 * n), ilu_factor=(n, 0.35 * n), vector=(18 * n, 0.5 * n),
                 assemble=(n, 0.6 * n), iq_update=(n, 0.18 * n), convergence=(n, 0.05 * n), halo=(12 * n, 0.1 * n * (1 + self.rank)), allreduce=(25 * n, 0.2 * n))
        return p

    def comm_info(self): return {"nranks": int(os.environ.get("OPMHIP_FAKE_NRANKS", self.world)), "rank": self.rank, "device": self.rank, "kind": "rccl"}
    def comm_selftest(self): return self.world * (self.world + 1) / 2.0, 2.0 * self.world
    def ordering_info(self): return {"ilu_ordering": "graph_coloring_greedy", "chain_length": 0, "colors": 2, "cpr_amg_ilu_levels": 0}
    def product_form(self): return {"half_product": False, "u_is_upper_a": True, "rest_blocks": 0, "rest_positions": 0}   # subdomains with ghost columns: the plain form
    def time_kernel(self, which, reps=20): return 0.1
    def cpr_levels(self): return [self.case["Nb"]], [len(self.case["col"])]


pkg.capi.HipModel = FakeModel
pkg.capi.comm_unique_id = lambda: bytes(range(128))
if __name__ == "__main__":
    bench.main()
