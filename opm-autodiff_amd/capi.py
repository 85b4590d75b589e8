"""ctypes binding of libopmhip.so — the same C-ABI a Flow-side shim binds (include/opmhip.h).

No fallback: if the shared library is missing this module raises at import of the library handle, and every
compute entry point returns OPMHIP_NO_DEVICE from opmhip_create when no gfx950 device is visible.
"""
import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OPMHIP_LIB", os.path.join(HERE, "libopmhip.so"))  # OPMHIP_LIB: tuning variants only
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "opmhip.h")

SUCCESS = 0
ANALYSIS_FAILED, CREATE_PRECONDITIONER_FAILED, UNKNOWN_ERROR = -1, -2, -3
INVALID_ARGUMENT, NOT_READY, DEVICE_ERROR, NO_DEVICE = -4, -5, -6, -7
REORDER = {"level_scheduling": 1, "graph_coloring": 2, "graph_coloring_greedy": 3, "line_coloring": 4, "auto": 5}
RELAX = {"post_scale": 0, "in_sweep": 1}
PRECONDITIONER = {"ilu0": 0, "cpr_quasiimpes": 1, "cpr": 2, "cpr_trueimpes": 2}   # opmhip_preconditioner


class OpmHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("opmhip status %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int), ("device_id", C.c_int), ("verbosity", C.c_int), ("maxit", C.c_int),
                ("tolerance", C.c_double), ("ilu_relaxation", C.c_double), ("relax_mode", C.c_int),
                ("reorder", C.c_int), ("zero_diag_fix", C.c_int), ("chain_length", C.c_int), ("spmv_pipe_wgs", C.c_int),
                ("preconditioner", C.c_int), ("cpr_reuse_setup", C.c_int), ("cpr_async_setup", C.c_int), ("cpr_amg_ilu_levels", C.c_int), ("cpr_gather_rows", C.c_int),
                ("half_product", C.c_int), ("pin_host_arrays", C.c_int), ("fused_reductions", C.c_int)]


class Result(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("reduction", C.c_double),
                ("conv_rate", C.c_double), ("elapsed", C.c_double), ("it", C.c_double), ("t_copy", C.c_double),
                ("t_factor", C.c_double), ("t_solve", C.c_double), ("num_colors", C.c_int),
                ("reserved", C.c_int * 3)]


EPS_FIELDS = ("swl", "swcr", "swu", "sowcr", "sgl", "sgcr", "sgu", "sogcr", "max_pcow", "max_pcgo", "max_krw", "max_krow", "max_krg", "max_krog",
              "krwr", "krorw", "krgr", "krorg")   # OPMHIP_EPS_* order


class EndpointScaling(C.Structure):
    _fields_ = [("sat_scaling", C.c_int), ("three_point_kr", C.c_int), ("krw", C.c_int), ("kro", C.c_int), ("krg", C.c_int),
                ("pcw", C.c_int), ("pcg", C.c_int), ("points", C.c_void_p * 18)]


def endpoint_scaling_struct(es):
    """dict(sat_scaling, three_point_kr, krw, kro, krg, pcw, pcg, <EPS_FIELDS>: per-cell arrays or absent) -> (struct, keep-alive)"""
    s = EndpointScaling()
    for k in ("sat_scaling", "three_point_kr", "krw", "kro", "krg", "pcw", "pcg"):
        setattr(s, k, int(es.get(k, 0)))
    keep = []
    for f, name in enumerate(EPS_FIELDS):
        a = _f64(es.get(name))
        keep.append(a)
        s.points[f] = None if a is None else a.ctypes.data
    return s, keep


MS_APPLY_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))   # opmhip_ms_apply_fn


class Wells(C.Structure):
    _fields_ = [("num_wells", C.c_int), ("val_pointers", C.c_void_p), ("Ccols", C.c_void_p),
                ("Bcols", C.c_void_p), ("Cnnzs", C.c_void_p), ("Dnnzs", C.c_void_p), ("Bnnzs", C.c_void_p),
                ("num_ms_wells", C.c_int), ("ms_apply", MS_APPLY_FN), ("ms_user", C.c_void_p), ("distributed", C.c_int)]


def declared_symbols():
    """Every function name include/opmhip.h declares."""
    with open(HEADER_PATH) as f:
        txt = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(opmhip_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OpmHipError(NO_DEVICE, "libopmhip.so not built (%s); run `make -C %s` or "
                              "__graft_entry__.build() - there is no CPU fallback" % (LIB_PATH, HERE))
        L = C.CDLL(LIB_PATH)
        vp, ip, dp = C.c_void_p, C.c_void_p, C.c_void_p
        L.opmhip_abi_version.restype = C.c_int
        L.opmhip_default_config.argtypes = [C.POINTER(Config)]
        L.opmhip_default_config.restype = None
        L.opmhip_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        L.opmhip_destroy.argtypes = [vp]
        L.opmhip_destroy.restype = None
        L.opmhip_last_error.argtypes = [vp]
        L.opmhip_last_error.restype = C.c_char_p
        L.opmhip_set_pattern.argtypes = [vp, C.c_int, C.c_int, ip, ip]
        L.opmhip_solve_system.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, ip, ip, dp, C.POINTER(Wells),
                                          C.POINTER(Result)]
        L.opmhip_get_result.argtypes = [vp, dp]
        L.opmhip_get_rhs.argtypes = [vp, dp]
        L.opmhip_add_well_contributions.argtypes = [vp, C.POINTER(Wells)]
        L.opmhip_wells_apply_residual.argtypes = [vp, C.POINTER(Wells), dp]
        L.opmhip_wells_recover_solution.argtypes = [vp, C.POINTER(Wells), dp, dp]
        L.opmhip_upload_system.argtypes = [vp, dp, dp]
        L.opmhip_spmv.argtypes = [vp, dp, dp]
        L.opmhip_ilu0_factor.argtypes = [vp, dp]
        L.opmhip_ilu0_apply.argtypes = [vp, dp, dp]
        L.opmhip_cpr_apply.argtypes = [vp, dp, dp]
        L.opmhip_preconditioned_product.argtypes = [vp, dp, dp, dp]
        L.opmhip_cpr_recreate.argtypes = [vp]
        L.opmhip_set_cpr_weights.argtypes = [vp, dp]
        L.opmhip_get_cpr_weights.argtypes = [vp, dp]
        L.opmhip_get_ordering.argtypes = [vp, ip, ip, ip]
        L.opmhip_get_ordering_info.argtypes = [vp, C.POINTER(C.c_int * 4)]
        L.opmhip_get_product_form.argtypes = [vp, C.POINTER(C.c_int * 4)]
        L.opmhip_time_kernel.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.opmhip_cpr_levels.argtypes = [vp, ip, ip, C.c_int]
        L.opmhip_profile_enable.argtypes = [vp, C.c_int]
        L.opmhip_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
        L.opmhip_synchronize.argtypes = [vp]
        L.opmhip_comm_info.argtypes = [vp, ip]
        L.opmhip_comm_selftest.argtypes = [vp, dp]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def make_wells(w):
    """dict(numWells, val_pointers, Ccols, Bcols, Cnnzs, Dnnzs, Bnnzs[, numMsWells, ms_apply][, distributed]) -> (Wells struct, keep-alive list).
    numWells counts the standard wells; ms_apply(x, y) - numpy views of the pinned host vectors, natural order - performs
    y -= C^T (D^-1 (B x)) for the numMsWells multisegment wells in place (opmhip_wells.ms_apply)."""
    if not w:
        return None, []
    nstd = int(w.get("numWells", 0))
    if nstd > 0:
        keep = [_i32(w["val_pointers"]), _i32(w["Ccols"]), _i32(w["Bcols"]), _f64(w["Cnnzs"]), _f64(w["Dnnzs"]), _f64(w["Bnnzs"])]
        s = Wells(nstd, *[_ptr(k) for k in keep])
    else:
        keep, s = [], Wells(0)
    nms = int(w.get("numMsWells", 0))
    if nms > 0:
        fn, n = w["ms_apply"], int(w["N"])

        def tramp(_user, hx, hy):
            fn(np.ctypeslib.as_array(hx, shape=(n,)), np.ctypeslib.as_array(hy, shape=(n,)))
        cb = MS_APPLY_FN(tramp)
        keep.append(cb)
        s.num_ms_wells, s.ms_apply = nms, cb
    s.distributed = int(w.get("distributed", 0))   # decomposed runs: the same list on every rank, each with its own perforations (opmhip_wells.distributed)
    return s, keep


class HipSolver:
    """Thin object wrapper over one opmhip context (mirrors how bda::BdaSolver<3> is used:
    ctor(verbosity, maxit, tolerance, deviceID), solve_system(...), get_result(x))."""

    def __init__(self, verbosity=0, maxit=200, tolerance=1e-2, device_id=0, ilu_relaxation=0.9,
                 relax_mode="post_scale", reorder=None, zero_diag_fix=True, chain_length=0, spmv_pipe_wgs=0,
                 preconditioner="ilu0", cpr_reuse_setup=3, cpr_async_setup=0, cpr_amg_ilu_levels=None, cpr_gather_rows=None, half_product=0, pin_host_arrays=0, fused_reductions=0):
        """reorder / cpr_amg_ilu_levels / cpr_gather_rows = None: what opmhip_default_config says (reorder "auto", the library's choice of
        the AMG smoother, the pressure stage across the ranks as the communicator's kind allows).  half_product: ILU0-BiCGStab forms the
        product after M^-1 from the backward sweep's row sums (0 the library's choice, > 0 wherever the pattern allows, < 0 never)"""
        L = lib()
        cfg = Config()
        L.opmhip_default_config(C.byref(cfg))
        cfg.verbosity, cfg.maxit, cfg.tolerance, cfg.device_id = verbosity, maxit, tolerance, device_id
        cfg.ilu_relaxation = ilu_relaxation
        cfg.relax_mode = RELAX[relax_mode]
        if reorder is not None:
            cfg.reorder = REORDER[reorder]
        cfg.zero_diag_fix = int(zero_diag_fix)
        cfg.chain_length = int(chain_length)  # line colouring: rows per chain (0: the library's default, 8; 10 with reorder="auto")
        cfg.spmv_pipe_wgs = int(spmv_pipe_wgs)  # pipelined SpMV: workgroups it is sized for (0 default, < 0 off; tests use small values)
        # --linear-solver-configuration (setupPropertyTree.cpp:62-76): "cpr" is short for cpr_trueimpes, as in Flow
        cfg.preconditioner = PRECONDITIONER[preconditioner]
        if cpr_amg_ilu_levels is not None:
            cfg.cpr_amg_ilu_levels = int(cpr_amg_ilu_levels)   # finest levels of the pressure AMG that smooth with ILU0 (0: Jacobi everywhere, < 0: the library's choice)
        if cpr_gather_rows is not None:
            cfg.cpr_gather_rows = int(cpr_gather_rows)         # decomposed runs: the hierarchy is continued across the ranks from the first level this small (0 default, < 0 off)
        cfg.half_product = int(half_product)
        cfg.fused_reductions = int(fused_reductions)   # 1: one reduction (three sums) per BiCGStab half iteration, recurred norms
        self._pin = bool(pin_host_arrays)
        cfg.pin_host_arrays = int(pin_host_arrays)   # 1: vals / b / x keep their addresses (Flow's do): registered for DMA on first sight
        cfg.cpr_async_setup = int(cpr_async_setup)   # mode 2 only: the rebuild on a host thread beside the solves
        cfg.cpr_reuse_setup = int(cpr_reuse_setup)   # --cpr-reuse-setup: 0 every solve, 1 every time step, 2 after > 10 iterations, 3 never
        self._h = C.c_void_p()
        rc = L.opmhip_create(C.byref(cfg), C.byref(self._h))
        if rc != SUCCESS:
            raise OpmHipError(rc, L.opmhip_last_error(None).decode())
        self.Nb = self.nnzb = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().opmhip_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise OpmHipError(rc, lib().opmhip_last_error(self._h).decode())
        return rc

    def set_pattern(self, Nb, rows, cols):
        rows, cols = _i32(rows), _i32(cols)
        self._check(lib().opmhip_set_pattern(self._h, Nb, len(cols), _ptr(rows), _ptr(cols)))
        self.Nb, self.nnzb = Nb, len(cols)

    def solve_system(self, Nb, rows, cols, vals, b, wells=None):
        """bda::BdaSolver::solve_system; returns the Result struct (check .converged like the reference)."""
        rows, cols, vals, b = _i32(rows), _i32(cols), _f64(vals), _f64(b)
        nnzb = self.nnzb if cols is None else len(cols)
        res = Result()
        ws, keep = make_wells(wells)
        self._check(lib().opmhip_solve_system(self._h, 3 * Nb, 9 * nnzb, 3, _ptr(vals), _ptr(rows), _ptr(cols),
                                              _ptr(b), C.byref(ws) if ws else None, C.byref(res)))
        self.Nb, self.nnzb = Nb, nnzb
        return res

    def get_result(self):
        if getattr(self, "_pin", False):
            # registered arrays must keep their addresses (opmhip_config.pin_host_arrays): one result buffer per context, handed out as a copy
            if getattr(self, "_xbuf", None) is None or len(self._xbuf) != 3 * self.Nb:
                self._xbuf = np.empty(3 * self.Nb)
            self._check(lib().opmhip_get_result(self._h, _ptr(self._xbuf)))
            return self._xbuf.copy()
        x = np.empty(3 * self.Nb)
        self._check(lib().opmhip_get_result(self._h, _ptr(x)))
        return x


    def get_rhs(self):
        """the right-hand side / residual currently on the device, natural order"""
        b = np.empty(3 * self.Nb)
        self._check(lib().opmhip_get_rhs(self._h, _ptr(b)))
        return b

    def add_well_contributions(self, wells):
        """A -= C^T D^-1 B into the device-resident matrix (StandardWell::addWellContributions); the pattern must hold
        the well cliques"""
        ws, keep = make_wells(wells)
        self._check(lib().opmhip_add_well_contributions(self._h, C.byref(ws) if ws else None))

    def wells_apply_residual(self, wells, res_well):
        """r -= C^T (D^-1 resWell) on the device-resident residual (StandardWell::apply(BVector& r))"""
        ws, keep = make_wells(wells)
        rw = _f64(res_well)
        self._check(lib().opmhip_wells_apply_residual(self._h, C.byref(ws) if ws else None, _ptr(rw)))

    def wells_recover_solution(self, wells, res_well):
        """xw = D^-1 (resWell - B x) from the device-resident solution (StandardWell::recoverSolutionWell)"""
        ws, keep = make_wells(wells)
        rw = _f64(res_well)
        xw = np.empty(4 * int(wells["numWells"]))
        self._check(lib().opmhip_wells_recover_solution(self._h, C.byref(ws) if ws else None, _ptr(rw), _ptr(xw)))
        return xw

    def upload_system(self, vals, b=None):
        vals, b = _f64(vals), _f64(b)
        self._check(lib().opmhip_upload_system(self._h, _ptr(vals), _ptr(b)))

    def spmv(self, x):
        x = _f64(x)
        y = np.empty_like(x)
        self._check(lib().opmhip_spmv(self._h, _ptr(x), _ptr(y)))
        return y

    def ilu0_factor(self, want_factors=True):
        lu = np.empty(9 * self.nnzb) if want_factors else None
        self._check(lib().opmhip_ilu0_factor(self._h, _ptr(lu)))
        return lu

    def ilu0_apply(self, d):
        d = _f64(d)
        v = np.empty_like(d)
        self._check(lib().opmhip_ilu0_apply(self._h, _ptr(d), _ptr(v)))
        return v

    def set_cpr_weights(self, w=None):
        """CPR weights from outside (3 per block row, natural order - e.g. Flow's getTrueImpesWeights); None: computed again"""
        w = None if w is None else _f64(np.asarray(w).reshape(-1))
        self._w_keep = w
        self._check(lib().opmhip_set_cpr_weights(self._h, _ptr(w)))

    def cpr_weights(self):
        w = np.empty(3 * self.Nb)
        self._check(lib().opmhip_get_cpr_weights(self._h, _ptr(w)))
        return w.reshape(-1, 3)

    def cpr_recreate(self):
        """the next solve builds the CPR hierarchy's structure anew (the host's ISTLSolverEbos::shouldCreateSolver said so)"""
        self._check(lib().opmhip_cpr_recreate(self._h))

    def cpr_apply(self, d):
        d = _f64(d)
        v = np.empty_like(d)
        self._check(lib().opmhip_cpr_apply(self._h, _ptr(d), _ptr(v)))
        return v

    def preconditioned_product(self, d):
        """(A (M^-1 d), M^-1 d) in the form ILU0-BiCGStab uses inside a solve (opmhip_preconditioned_product)"""
        d = _f64(d)
        t, z = np.empty_like(d), np.empty_like(d)
        self._check(lib().opmhip_preconditioned_product(self._h, _ptr(d), _ptr(t), _ptr(z)))
        return t, z

    def ordering(self):
        to = np.empty(self.Nb, np.int32)
        fr = np.empty(self.Nb, np.int32)
        rpc = np.zeros(self.Nb, np.int32)
        nc = self._check(lib().opmhip_get_ordering(self._h, _ptr(to), _ptr(fr), _ptr(rpc)))
        return to, fr, rpc[:nc].copy()

    def ordering_info(self):
        """what the library's own choices resolved to (opmhip_get_ordering_info)"""
        info = (C.c_int * 4)()
        self._check(lib().opmhip_get_ordering_info(self._h, C.byref(info)))
        names = {v: k for k, v in REORDER.items()}
        return {"ilu_ordering": names[info[0]], "chain_length": int(info[1]), "colors": int(info[2]), "cpr_amg_ilu_levels": int(info[3])}

    def product_form(self):
        """what opmhip_config.half_product resolved to (opmhip_get_product_form)"""
        info = (C.c_int * 4)()
        self._check(lib().opmhip_get_product_form(self._h, C.byref(info)))
        return {"half_product": bool(info[0]), "u_is_upper_a": bool(info[1]), "rest_blocks": int(info[2]), "rest_positions": int(info[3])}

    PROF = ["spmv", "ilu_apply", "ilu_factor", "vector", "assemble", "iq_update", "convergence", "cpr_amg", "spmv_boundary",
            "halo", "allreduce", "cpr_gather"]   # the last three: communication spans of decomposed runs (they overlap the kernel scopes)

    def profile_enable(self, on=True):
        self._check(lib().opmhip_profile_enable(self._h, int(on)))

    def profile(self):
        """-> {class: (launches, total_ms)} from HIP events on the context's stream"""
        out = {}
        for i, name in enumerate(self.PROF):
            n, ms = C.c_longlong(0), C.c_double(0.0)
            self._check(lib().opmhip_profile_get(self._h, i, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out

    def cpr_levels(self):
        """(unknowns, entries) of the pressure-AMG levels (preconditioner="cpr", after the first solve)"""
        n, nnz = np.zeros(32, np.int32), np.zeros(32, np.int32)
        L = self._check(lib().opmhip_cpr_levels(self._h, _ptr(n), _ptr(nnz), 32))
        return [int(v) for v in n[:L]], [int(v) for v in nnz[:L]]

    def synchronize(self):
        """everything enqueued on the context's stream is done"""
        self._check(lib().opmhip_synchronize(self._h))

    def comm_info(self):
        """what the communicator itself reports: dict(nranks = ncclCommCount, rank, device, kind)"""
        a = np.zeros(4, np.int32)
        self._check(lib().opmhip_comm_info(self._h, _ptr(a)))
        return {"nranks": int(a[0]), "rank": int(a[1]), "device": int(a[2]), "kind": ["none", "loopback", "rccl"][int(a[3])]}

    def comm_selftest(self):
        """one RCCL all-reduce of (1 + rank, 2): -> (sum over ranks of 1 + rank, 2 * nranks)"""
        a = np.zeros(2)
        self._check(lib().opmhip_comm_selftest(self._h, _ptr(a)))
        return float(a[0]), float(a[1])

    def time_kernel(self, which, reps=20):
        ms = C.c_double()
        self._check(lib().opmhip_time_kernel(self._h, {"spmv": 0, "ilu_apply": 1, "ilu_factor": 2, "vector": 3, "stream_read": 4, "spmv_dot1": 5, "spmv_dot2": 6,
                                                              "rest_product": 7, "rest_product_dot2": 8, "ilu_apply_rowsums": 9}[which],
                                             reps, C.byref(ms)))
        return ms.value


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 creates it, the launcher broadcasts it)."""
    L = lib()
    if not getattr(L, "_asm_bound", False):
        _bind_assembly(L)
        L._asm_bound = True
    buf = C.create_string_buffer(128)
    rc = L.opmhip_comm_unique_id(buf)
    if rc != SUCCESS:
        raise OpmHipError(rc, "opmhip_comm_unique_id failed (librccl not loadable?)")
    return buf.raw


def _bind_assembly(L):
    vp = C.c_void_p
    L.opmhip_set_fluid.argtypes = [vp, vp]
    L.opmhip_set_static.argtypes = [vp] + [vp] * 9
    L.opmhip_set_state.argtypes = [vp, vp, vp]
    L.opmhip_get_state.argtypes = [vp, vp, vp]
    L.opmhip_advance_time_level.argtypes = [vp]
    L.opmhip_update_failed.argtypes = [vp]
    L.opmhip_end_time_step.argtypes = [vp, C.c_double]
    L.opmhip_set_drift_compensation.argtypes = [vp, C.c_int, C.c_double]
    L.opmhip_set_source.argtypes = [vp, vp, vp]
    L.opmhip_assemble.argtypes = [vp, C.c_double, C.c_int, vp, vp]
    L.opmhip_get_iq.argtypes = [vp, vp]
    L.opmhip_get_iq_cells.argtypes = [vp, C.c_int, vp, vp]
    L.opmhip_set_source_cells.argtypes = [vp, C.c_int, vp, vp, vp]
    L.opmhip_convergence.argtypes = [vp, C.c_double, C.c_double, vp]
    L.opmhip_update.argtypes = [vp, vp, C.c_double, C.POINTER(C.c_int)]
    L.opmhip_set_pattern_dd.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.opmhip_comm_unique_id.argtypes = [C.c_char_p]
    L.opmhip_comm_init_rccl.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.opmhip_comm_init_loopback.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.opmhip_set_halo.argtypes = [vp, C.c_longlong, C.c_int, vp, vp, vp, vp]
    L.opmhip_set_cell_global_ids.argtypes = [vp, vp]
    L.opmhip_fluid_probe.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.opmhip_set_problem_extras.argtypes = [vp, vp, vp, vp]
    L.opmhip_set_pcw.argtypes = [vp, vp]
    L.opmhip_set_endpoint_scaling.argtypes = [vp, vp]
    L.opmhip_set_composition_change_limits.argtypes = [vp, vp, vp, vp]
    L.opmhip_set_irreversible_compaction.argtypes = [vp, C.c_int]
    L.opmhip_begin_time_step.argtypes = [vp, C.c_double]
    L.opmhip_get_trackers.argtypes = [vp, vp, vp, vp, vp]
    L.opmhip_set_vappars.argtypes = [vp, C.c_int, C.c_double, C.c_double]
    L.opmhip_set_water_compaction.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.opmhip_get_max_water_saturation.argtypes = [vp, vp]
    L.opmhip_relative_change.argtypes = [vp, C.POINTER(C.c_double)]
    L.opmhip_sat_end_points.argtypes = [vp, C.c_int, vp]
    L.opmhip_sat_probe.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp]
    L.opmhip_gas_probe.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.opmhip_iq_fields.argtypes = [vp]
    L.opmhip_set_hysteresis.argtypes = [vp, C.c_int, vp, vp]
    L.opmhip_get_hysteresis.argtypes = [vp, vp, vp, vp, vp]
    L.opmhip_set_hysteresis_params.argtypes = [vp, vp, vp]


class HipFluid(HipSolver):
    """A context that only holds the fluid tables: point evaluation of the device's property functions
    (opmhip_fluid_probe) for host-side setup such as equilibration (equil.py)."""

    COLUMNS = ("invBw", "invBg", "invBo", "rsSat", "pcow", "pcgo", "muo", "mug")

    def __init__(self, fluid, **solver_kw):
        super().__init__(**solver_kw)
        L = lib()
        if not getattr(L, "_asm_bound", False):
            _bind_assembly(L)
            L._asm_bound = True
        self.fluid = fluid
        self._fd = fluid.desc()
        self._check(L.opmhip_set_fluid(self._h, C.addressof(self._fd)))

    def probe(self, p, rs=0.0, sw=0.0, sg=0.0, pvt_region=0, sat_region=0):
        """-> array (n, 8): 1/B_w(p), 1/B_g(p), 1/B_o(p, rs), RsSat(p), pcow(sw), pcgo(sg), mu_o(p, rs), mu_g(p)"""
        p = np.atleast_1d(np.asarray(p, np.float64))
        n = len(p)
        a = [np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float64), (n,))) for v in (p, rs, sw, sg)]
        out = np.empty((n, 8))
        self._check(lib().opmhip_fluid_probe(self._h, pvt_region, sat_region, n, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(out)))
        return out


def _probe_gas(self, p, rv=0.0, pvt_region=0):
    """-> array (n, 3): 1/B_g(p, rv) and mu_g(p, rv) (saturated curve where rv >= RvSat(p)), RvSat(p)"""
    p = np.atleast_1d(np.asarray(p, np.float64))
    n = len(p)
    a = [np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float64), (n,))) for v in (p, rv)]
    out = np.empty((n, 3))
    self._check(lib().opmhip_gas_probe(self._h, pvt_region, n, _ptr(a[0]), _ptr(a[1]), _ptr(out)))
    return out


HipFluid.probe_gas = _probe_gas


def _sat_end_points(self, sat_region=0):
    """the end points of a saturation region's own tables: dict over capi.EPS_FIELDS (opmhip_sat_end_points)"""
    out = np.empty(18)
    self._check(lib().opmhip_sat_end_points(self._h, sat_region, _ptr(out)))
    return dict(zip(EPS_FIELDS, out.tolist()))


HipFluid.sat_end_points = _sat_end_points


def _sat_probe(self, sw, sg, endscale=None, sat_region=0):
    """(n, 5): krw, kro, krg, pcow, pcgo at (sw, sg); endscale: dict(sat_scaling, three_point_kr, krw, kro, krg, pcw, pcg and any
    of EPS_FIELDS as ONE value each; absent = the table's own end point) or None = unscaled"""
    sw = np.atleast_1d(np.asarray(sw, np.float64))
    n = len(sw)
    sw = np.ascontiguousarray(sw)
    sg = np.ascontiguousarray(np.broadcast_to(np.asarray(sg, np.float64), (n,)))
    out = np.empty(5 * n)
    if endscale is None:
        self._check(lib().opmhip_sat_probe(self._h, sat_region, None, n, _ptr(sw), _ptr(sg), _ptr(out)))
    else:
        es = {k: (np.array([v], np.float64) if k in EPS_FIELDS else v) for k, v in endscale.items()}
        s, keep = endpoint_scaling_struct(es)
        self._check(lib().opmhip_sat_probe(self._h, sat_region, C.byref(s), n, _ptr(sw), _ptr(sg), _ptr(out)))
    return out.reshape(n, 5)


HipFluid.sat_probe = _sat_probe


class HipModel(HipSolver):
    """One context holding the whole per-Newton-iteration path on the device: the call surface follows
    BlackoilModelEbos (assembleReservoir -> getReservoirConvergence -> solveJacobianSystem -> updateSolution,
    opm/simulators/flow/BlackoilModelEbos.hpp:274-392) with every array resident in HBM between the calls."""

    def __init__(self, case, comm=None, **solver_kw):
        """comm (decomposed runs; case from ras.cartesian_subdomain_case / ras.local_problem):
        ("rccl", nranks, rank, id128) or ("loopback", nranks, rank, group_name)."""
        super().__init__(**solver_kw)
        L = lib()
        if not getattr(L, "_asm_bound", False):
            _bind_assembly(L)
            L._asm_bound = True
        self.case = case
        self.Nghost = int(case.get("Nghost", 0))
        if self.Nghost or comm:
            rows, cols = _i32(case["rowptr"]), _i32(case["col"])
            self._check(L.opmhip_set_pattern_dd(self._h, case["Nb"], self.Nghost, len(cols), _ptr(rows), _ptr(cols)))
            self.Nb, self.nnzb = case["Nb"], len(cols)
            if case.get("gids") is not None:
                g64 = np.ascontiguousarray(case["gids"], np.int64)
                self._check(L.opmhip_set_cell_global_ids(self._h, _ptr(g64)))
            if comm:
                kind, nranks, rank, tok = comm
                if kind == "rccl":
                    self._check(L.opmhip_comm_init_rccl(self._h, nranks, rank, tok))
                else:
                    self._check(L.opmhip_comm_init_loopback(self._h, nranks, rank, tok.encode()))
                h = case["halo"]
                keep = [_i32(h["neigh"]), _i32(h["send_ptr"]), _i32(h["send_cells"]), _i32(h["recv_ptr"])]
                self._check(L.opmhip_set_halo(self._h, int(case["global_cells"]), len(keep[0]), *[_ptr(k) for k in keep]))
        else:
            self.set_pattern(case["Nb"], case["rowptr"], case["col"])
        self.Nloc = self.Nb + self.Nghost
        self._fd = case["fluid"].desc()
        self._check(L.opmhip_set_fluid(self._h, C.addressof(self._fd)))
        g = lambda k, f: _ptr(f(case[k])) if case.get(k) is not None else None
        keep = {k: (_f64(case[k]) if k not in ("pvtnum", "satnum") else _i32(case[k]))
                for k in ("trans", "area", "thpres", "poro", "volume", "depth", "pvtnum", "satnum", "rsmax")
                if case.get(k) is not None}
        p = lambda k: _ptr(keep.get(k))
        self._check(L.opmhip_set_static(self._h, p("trans"), p("area"), p("thpres"), p("poro"), p("volume"), p("depth"),
                                        p("pvtnum"), p("satnum"), p("rsmax")))
        if any(case.get(k) is not None for k in ("rvmax", "rocknum", "overburden")):
            self.set_problem_extras(case.get("rvmax"), case.get("rocknum"), case.get("overburden"))
        if case.get("pcw") is not None:
            self.set_pcw(case["pcw"])
        if case.get("endscale") is not None:
            self.set_endpoint_scaling(case["endscale"])

    def set_endpoint_scaling(self, es):
        """saturation end-point scaling (ENDSCALE family): es = dict(sat_scaling, three_point_kr, krw, kro, krg, pcw, pcg and any
        of capi.EPS_FIELDS as per-cell arrays; absent = the end point of the cell's SATNUM tables); None = off.  The fluid needs
        pc_scaling=True."""
        if es is None:
            self._check(lib().opmhip_set_endpoint_scaling(self._h, None))
            return
        s, keep = endpoint_scaling_struct(es)
        self._check(lib().opmhip_set_endpoint_scaling(self._h, C.byref(s)))

    def set_hysteresis(self, kr_model, imbnum=None, imb_endscale=None):
        """relative-permeability hysteresis (SATOPTS HYSTER): kr_model = EHYSTR item 2 (0 | 1; None / negative = off), imbnum = per
        cell imbibition saturation region (0-based), imb_endscale = dict of per-cell scaled end points of the imbibition curves (any
        of capi.EPS_FIELDS, absent = the imbibition tables' own; only with set_endpoint_scaling in force)"""
        if kr_model is None or kr_model < 0:
            self._check(lib().opmhip_set_hysteresis(self._h, -1, None, None))
            return
        imb = _i32(imbnum)
        if imb_endscale is None:
            self._check(lib().opmhip_set_hysteresis(self._h, int(kr_model), _ptr(imb), None))
            return
        s, keep = endpoint_scaling_struct(imb_endscale)
        self._check(lib().opmhip_set_hysteresis(self._h, int(kr_model), _ptr(imb), C.byref(s)))

    def hysteresis(self):
        """(krnSwMdc, deltaSwImbKrn) of the oil-water system, then of the gas-oil system: four per-cell arrays (natural order)"""
        out = [np.empty(self.Nloc) for _ in range(4)]
        self._check(lib().opmhip_get_hysteresis(self._h, *[_ptr(a) for a in out]))
        return tuple(a[:self.Nb] for a in out) if self.Nghost == 0 else tuple(out)

    def set_hysteresis_params(self, sw_ow, sw_go):
        a, b = _f64(sw_ow), _f64(sw_go)
        self._check(lib().opmhip_set_hysteresis_params(self._h, _ptr(a), _ptr(b)))

    def set_pcw(self, pcw):
        """per-cell scaled maximum of the oil-water capillary pressure (PCW, or SWATINIT through equil.equilibrate's pcw_scale
        x the table's pcow(Swl)); None = the tables' own.  The fluid needs pc_scaling=True."""
        a = _f64(pcw)
        self._check(lib().opmhip_set_pcw(self._h, _ptr(a)))

    def set_problem_extras(self, rvmax=None, rocknum=None, overburden=None):
        """DRVDT cap on Rv, rock-table index (ROCKNUM), overburden pressure - per cell, any None"""
        a, b, d = _f64(rvmax), _i32(rocknum), _f64(overburden)
        self._check(lib().opmhip_set_problem_extras(self._h, _ptr(a), _ptr(b), _ptr(d)))

    def set_state(self, pv, meaning):
        pv, meaning = _f64(pv), np.ascontiguousarray(meaning, np.uint8)
        self._check(lib().opmhip_set_state(self._h, _ptr(pv), _ptr(meaning)))

    def get_state(self):
        pv = np.empty(3 * self.Nloc)
        m = np.empty(self.Nloc, np.uint8)
        self._check(lib().opmhip_get_state(self._h, _ptr(pv), _ptr(m)))
        return pv, m

    def advance_time_level(self):
        """solution(1) = solution(0): call when a time step starts (FvBaseDiscretization::advanceTimeLevel)."""
        self._check(lib().opmhip_advance_time_level(self._h))

    def update_failed(self):
        """solution(0) = solution(1) + intensive quantities: the Newton method gave up on this time step."""
        self._check(lib().opmhip_update_failed(self._h))

    def relative_change(self):
        """BlackoilModelEbos::relativeChange: squared change of pressure and saturations since advance_time_level over their
        squared new values (the PID time-step control's error measure); summed over the ranks of a decomposed run."""
        out = C.c_double(0.0)
        self._check(lib().opmhip_relative_change(self._h, C.byref(out)))
        return out.value

    def set_composition_change_limits(self, drsdt=None, drsdt_all_cells=None, drvdt=None):
        """DRSDT / DRVDT: rates per PVT region [1/s] (negative: none), None = keyword not in force; drsdt_all_cells: the OILVAP
        option per region.  After set_state (lastRs / lastRv start from the state now present)."""
        a, b, d = _f64(drsdt), _i32(drsdt_all_cells), _f64(drvdt)
        self._check(lib().opmhip_set_composition_change_limits(self._h, _ptr(a), _ptr(b), _ptr(d)))

    def set_irreversible_compaction(self, enable=True):
        """ROCKCOMP IRREVERS: rock tables read at the lowest oil pressure a cell has seen.  After set_state."""
        self._check(lib().opmhip_set_irreversible_compaction(self._h, int(enable)))

    def begin_time_step(self, dt):
        """EclProblem::beginTimeStep, per-cell part: minimum pressure, DRSDT / DRVDT caps of a step of size dt"""
        self._check(lib().opmhip_begin_time_step(self._h, dt))

    def set_vappars(self, vap1, vap2, enable=True):
        """VAPPARS: exponents on the saturated Rv / Rs below the largest oil saturation seen.  After set_state."""
        self._check(lib().opmhip_set_vappars(self._h, int(enable), float(vap1), float(vap2)))

    def set_water_compaction(self, tables):
        """Water-induced rock compaction (ROCK2D / ROCK2DTR / ROCKWNOD): one dict per rock region with "pressure" [Pa] and "sw"
        nodes (ascending), "pv_mult" [len(pressure) x len(sw)] and optionally "trans_mult" of the same shape (all tables or
        none).  None / []: off.  After set_state."""
        tables = list(tables or [])
        if not tables:
            self._check(lib().opmhip_set_water_compaction(self._h, 0, None, None, None, None, None, None))
            return
        npv = _i32([len(t["pressure"]) for t in tables])
        nsw = _i32([len(t["sw"]) for t in tables])
        pr = _f64(np.concatenate([np.asarray(t["pressure"], float) for t in tables]))
        sw = _f64(np.concatenate([np.asarray(t["sw"], float) for t in tables]))
        shaped = lambda t, k: np.asarray(t[k], float).reshape(len(t["pressure"]), len(t["sw"])).ravel()
        pv = _f64(np.concatenate([shaped(t, "pv_mult") for t in tables]))
        has_tr = [t.get("trans_mult") is not None for t in tables]
        if any(has_tr) and not all(has_tr):
            raise ValueError("set_water_compaction: trans_mult for all tables or for none")
        tr = _f64(np.concatenate([shaped(t, "trans_mult") for t in tables])) if all(has_tr) else None
        self._check(lib().opmhip_set_water_compaction(self._h, len(tables), _ptr(npv), _ptr(nsw), _ptr(pr), _ptr(sw), _ptr(pv), _ptr(tr) if tr is not None else None))

    def max_water_saturation(self):
        """the tracker of the water-induced compaction per cell, natural order (zeros when the feature is off)"""
        a = np.empty(self.Nloc)
        self._check(lib().opmhip_get_max_water_saturation(self._h, _ptr(a)))
        return a

    def trackers(self):
        """-> (lastRs, lastRv, minimum oil pressure, maximum oil saturation) per cell, natural order; zeros where not kept"""
        a, b, d, e = (np.empty(self.Nloc) for _ in range(4))
        self._check(lib().opmhip_get_trackers(self._h, _ptr(a), _ptr(b), _ptr(d), _ptr(e)))
        return a, b, d, e

    def end_time_step(self, dt):
        """EclProblem::endTimeStep (drift part): remember residual * dt of the time step that was just accepted."""
        self._check(lib().opmhip_end_time_step(self._h, dt))

    def set_drift_compensation(self, enable=True, max_compensation=0.1):
        self._check(lib().opmhip_set_drift_compensation(self._h, int(enable), max_compensation))

    def set_source(self, source, dsource=None):
        s, d = _f64(source), _f64(dsource)
        self._check(lib().opmhip_set_source(self._h, _ptr(s), _ptr(d)))

    def set_source_cells(self, cells, source, dsource=None):
        """the source terms of the cells named (a well model's connection rates), zero everywhere else: opmhip_set_source_cells"""
        cl, s, d = _i32(cells), _f64(source), _f64(dsource)
        self._check(lib().opmhip_set_source_cells(self._h, len(cl), _ptr(cl), _ptr(s), _ptr(d)))

    def assemble(self, dt, iteration, fetch=True):
        """linearizeDomain(); with fetch the Jacobian values and residual are copied back (natural order)."""
        jac = np.empty(9 * self.nnzb) if fetch else None
        res = np.empty(3 * self.Nb) if fetch else None
        self._check(lib().opmhip_assemble(self._h, dt, iteration, _ptr(jac), _ptr(res)))
        return jac, res

    def iq(self):
        nf = self._check(lib().opmhip_iq_fields(self._h))
        out = np.empty(self.Nloc * nf * 4)
        self._check(lib().opmhip_get_iq(self._h, _ptr(out)))
        return out.reshape(self.Nloc, nf, 4)

    def iq_cells(self, cells):
        """the records of the cells named (a well model's perforated cells), gathered on the device: opmhip_get_iq_cells"""
        cl = _i32(cells)
        nf = self._check(lib().opmhip_iq_fields(self._h))
        out = np.empty(len(cl) * nf * 4)
        self._check(lib().opmhip_get_iq_cells(self._h, len(cl), _ptr(cl), _ptr(out)))
        return out.reshape(len(cl), nf, 4)

    def convergence(self, dt, tol_cnv=1e-2):
        out = np.empty(17)
        self._check(lib().opmhip_convergence(self._h, dt, tol_cnv, _ptr(out)))
        return out

    def solve_jacobian_system(self, wells=None):
        """solveJacobianSystem on the device-resident Jacobian and residual (no host copies)."""
        res = Result()
        ws, keep = make_wells(wells)
        self._check(lib().opmhip_solve_system(self._h, 3 * self.Nb, 9 * self.nnzb, 3, None, None, None, None,
                                              C.byref(ws) if ws else None, C.byref(res)))
        return res

    def update(self, dx=None, relax=1.0):
        dx = _f64(dx)
        n = C.c_int(0)
        self._check(lib().opmhip_update(self._h, _ptr(dx), relax, C.byref(n)))
        return n.value
