"""Synthetic Cartesian grids for the hot path: block-CSR pattern of the element-centred finite-volume stencil and
the per-connection two-point-flux transmissibilities.

Cell order is the compressed natural Cartesian order i + nx*(j + ny*k) (CpGrid, in-tree hint
wells/WellConnectionAuxiliaryModule.hpp:47-59).  The sparsity pattern is stencil(I) = {I} + face neighbours,
columns ascending.  Transmissibilities restate ebos/ecltransmissibility.cc:
half transmissibility K * |A.d| / |d|^2 (computeHalfTrans_, :928-944), harmonic combination
T = 1 / (1/T1 + 1/T2) with T = 0 below 1e-30 (:352-356), NTG on x/y faces (:1016-1043).
"""
import numpy as np


def cartesian_pattern(nx, ny, nz):
    """-> dict(Nb, rowptr, col, face_dir) ; face_dir[k] in {-3,-2,-1,0,1,2,3}: 0 diagonal, +-1 x, +-2 y, +-3 z."""
    Nb = nx * ny * nz
    idx = np.arange(Nb, dtype=np.int64)
    i = idx % nx
    j = (idx // nx) % ny
    k = idx // (nx * ny)
    # candidate neighbours in ascending column order: -z, -y, -x, self, +x, +y, +z
    offs = np.array([-nx * ny, -nx, -1, 0, 1, nx, nx * ny], dtype=np.int64)
    dirs = np.array([-3, -2, -1, 0, 1, 2, 3], dtype=np.int8)
    valid = np.stack([k > 0, j > 0, i > 0, np.ones(Nb, bool), i < nx - 1, j < ny - 1, k < nz - 1], axis=1)
    cols = idx[:, None] + offs[None, :]
    rowlen = valid.sum(axis=1)
    rowptr = np.zeros(Nb + 1, dtype=np.int64)
    np.cumsum(rowlen, out=rowptr[1:])
    col = cols[valid].astype(np.int32)
    face_dir = np.broadcast_to(dirs[None, :], valid.shape)[valid].astype(np.int8)
    assert rowptr[-1] < 2 ** 31
    return dict(Nb=Nb, nx=nx, ny=ny, nz=nz, rowptr=rowptr.astype(np.int32), col=col, face_dir=face_dir)


def row_of_entries(rowptr):
    return np.repeat(np.arange(len(rowptr) - 1, dtype=np.int32), np.diff(rowptr))


def cartesian_geometry(pat, dx, dy, dz, top=2500.0):
    """Cell volumes, centre depths and per-entry face areas for a box grid with uniform spacings (metres)."""
    nx, ny, nz, Nb = pat["nx"], pat["ny"], pat["nz"], pat["Nb"]
    k = np.arange(Nb) // (nx * ny)
    volume = np.full(Nb, dx * dy * dz)
    depth = top + (k + 0.5) * dz
    a = {1: dy * dz, 2: dx * dz, 3: dx * dy}
    area = np.zeros(len(pat["col"]))
    for d, A in a.items():
        area[np.abs(pat["face_dir"]) == d] = A
    return volume, depth, area


def tpfa_transmissibility(pat, permx, permy, permz, dx, dy, dz, ntg=None):
    """Per block-CSR entry T_IJ in SI (m^3): 0 on the diagonal.  perm* in m^2, per cell."""
    Nb = pat["Nb"]
    row = row_of_entries(pat["rowptr"])
    col = pat["col"]
    fd = pat["face_dir"].astype(np.int64)
    ntg = np.ones(Nb) if ntg is None else ntg
    T = np.zeros(len(col))
    spec = {1: (permx, dx, dy * dz, True), 2: (permy, dy, dx * dz, True), 3: (permz, dz, dx * dy, False)}
    for d, (perm, h, area, use_ntg) in spec.items():
        m = np.abs(fd) == d
        # half trans: K * |A . d| / |d|^2 with d = face centre - cell centre, |d| = h/2, A parallel to d
        half = lambda c: perm[c] * (area * (h / 2.0)) / ((h / 2.0) ** 2) * (ntg[c] if use_ntg else 1.0)
        t1, t2 = half(row[m]), half(col[m])
        with np.errstate(divide="ignore"):
            t = 1.0 / (1.0 / t1 + 1.0 / t2)
        t[(np.abs(t1) < 1e-30) | (np.abs(t2) < 1e-30)] = 0.0
        T[m] = t
    return T


def synthetic_block_values(pat, seed=0, dominance=1.5):
    """Random block values on a pattern, block-diagonally dominant (solver-only experiments; NOT a Jacobian)."""
    rng = np.random.default_rng(seed)
    nnzb = len(pat["col"])
    row = row_of_entries(pat["rowptr"])
    val = rng.uniform(-0.3, 0.0, size=(nnzb, 3, 3))
    isd = pat["col"] == row
    rs = np.zeros((pat["Nb"], 3))
    np.add.at(rs, row[~isd], np.abs(val[~isd]).sum(axis=2))
    dblk = rng.uniform(-0.1, 0.1, size=(pat["Nb"], 3, 3))
    ar = np.arange(3)
    dblk[:, ar, ar] = dominance * (rs + 0.5) + rng.uniform(0, 0.2, size=(pat["Nb"], 3))
    val[isd] = dblk
    return np.ascontiguousarray(val.reshape(-1))
