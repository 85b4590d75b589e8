"""Domain decomposition of a grid over the GPUs of a node (restricted additive Schwarz with block-Jacobi ILU0).

Host-side logic only: who owns which cell, the owner-cells-first local numbering with ghost cells last that the
reference demands of parallel runs (opm/simulators/linalg/ISTLSolverEbos.hpp:171-178), and the halo lists.  The device
side is libopmhip's set_pattern_dd / set_halo / comm_* entry points (include/opmhip.h); the reference's counterpart is
the MPI machinery of Dune::OwnerOverlapCopyCommunication fed by ExtractParallelGridInformationToISTL
(opm/simulators/linalg/ExtractParallelGridInformationToISTL.cpp) and findOverlapRowsAndColumns.hpp.

Local numbering of rank r: owned cells in ascending global id, then the ghost cells (cells of other ranks that an owned
row couples to) grouped by owner rank ascending and by global id inside a group.  By symmetry of the pattern the cells
rank r must send to neighbour q are exactly r's owned cells that couple to a cell of q, in ascending global id - which is
the order in which q numbers them as ghosts.
"""
import numpy as np

from . import decks as _decks
from . import grid as _grid


def block_layout(nranks):
    """(px, py, pz) for 1, 2, 4, 8 ... ranks: split x, then y, then z (8 -> 2 x 2 x 2: three face neighbours each, one
    xGMI link per neighbour)."""
    p = [1, 1, 1]
    d = 0
    n = nranks
    while n > 1:
        if n % 2:
            raise ValueError("rank count must be a power of two")
        p[d % 3] *= 2
        d += 1
        n //= 2
    return tuple(p)


def cartesian_owner(nx, ny, nz, px, py, pz):
    """owner rank of every cell of an nx x ny x nz grid cut into px x py x pz equal boxes (rank = cx + px*(cy + py*cz))"""
    idx = np.arange(nx * ny * nz)
    i, j, k = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    return ((i * px) // nx + px * ((j * py) // ny + py * ((k * pz) // nz))).astype(np.int32)


def local_problem(rowptr, col, owner, rank, gid=None):
    """Subdomain of `rank` from a (window of a) global pattern.

    rowptr/col: block-CSR pattern over the window's cells (window-local ids); owner[cell]; gid[cell] = global id (default:
    the window is the whole grid).  Returns dict(Nown, Nghost, rows, cols (local ids), entry (window entry index of each
    local entry), cells (window ids of the local cells: owned then ghosts), gids, neigh, send_ptr, send_cells, recv_ptr).
    """
    rowptr = np.asarray(rowptr, np.int64)
    col = np.asarray(col, np.int64)
    n = len(rowptr) - 1
    gid = np.arange(n, dtype=np.int64) if gid is None else np.asarray(gid, np.int64)
    owned = np.flatnonzero(owner == rank)
    owned = owned[np.argsort(gid[owned], kind="stable")]
    Nown = len(owned)
    # entries of the owned rows
    lens = (rowptr[owned + 1] - rowptr[owned])
    starts = rowptr[owned]
    ent = np.repeat(starts - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(lens.sum())
    c = col[ent]
    isghost = owner[c] != rank
    gcells = np.unique(c[isghost])
    order = np.lexsort((gid[gcells], owner[gcells]))
    gcells = gcells[order]
    Nghost = len(gcells)
    loc = np.full(n, -1, np.int64)
    loc[owned] = np.arange(Nown)
    loc[gcells] = Nown + np.arange(Nghost)
    cols = loc[c]
    rows = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    # columns must ascend inside a row in LOCAL ids: ghosts moved to the row's end -> re-sort each row
    rowid = np.repeat(np.arange(Nown), lens)
    perm = np.lexsort((cols, rowid))
    cols, ent = cols[perm], ent[perm]
    neigh = np.unique(owner[gcells]).astype(np.int32)
    recv_ptr = np.concatenate([[0], np.cumsum([np.count_nonzero(owner[gcells] == q) for q in neigh])]).astype(np.int32)
    send_cells, send_ptr = [], [0]
    rowcell = owned[rowid]
    for q in neigh:
        m = owner[col[ent]] == q
        mine = np.unique(rowcell[m])
        mine = mine[np.argsort(gid[mine], kind="stable")]
        send_cells.append(loc[mine])
        send_ptr.append(send_ptr[-1] + len(mine))
    send_cells = np.concatenate(send_cells).astype(np.int32) if send_cells else np.zeros(0, np.int32)
    cells = np.concatenate([owned, gcells])
    return dict(Nown=Nown, Nghost=Nghost, rows=rows, cols=cols.astype(np.int32), entry=ent, cells=cells, gids=gid[cells],
                neigh=neigh, send_ptr=np.array(send_ptr, np.int32), send_cells=send_cells, recv_ptr=recv_ptr)


def cartesian_subdomain_case(n_per_rank, nranks, rank, state="mixed", heterogeneous=False, rate_scale=None, **kw):
    """Local case of `rank` for a global grid of (px*n) x (py*n) x (pz*n) cells (weak scaling: n^3 cells per rank), built
    from a window one cell larger than the rank's box so that the global pattern is never materialised.  Cell data
    (state, porosity, depth ...) are those of the global synthetic case of opm-autodiff_amd.decks."""
    px, py, pz = block_layout(nranks)
    n = n_per_rank
    NX, NY, NZ = px * n, py * n, pz * n
    cx, cy, cz = rank % px, (rank // px) % py, rank // (px * py)
    lo = np.array([cx * n, cy * n, cz * n])
    hi = lo + n
    wlo = np.maximum(lo - 1, 0)
    whi = np.minimum(hi + 1, [NX, NY, NZ])
    wn = whi - wlo
    pat = _grid.cartesian_pattern(int(wn[0]), int(wn[1]), int(wn[2]))
    widx = np.arange(pat["Nb"])
    wi, wj, wk = widx % wn[0] + wlo[0], (widx // wn[0]) % wn[1] + wlo[1], widx // (wn[0] * wn[1]) + wlo[2]
    gid = wi + NX * (wj + NY * wk)
    owner = ((wi * px) // NX + px * ((wj * py) // NY + py * ((wk * pz) // NZ))).astype(np.int32)
    lp = local_problem(pat["rowptr"], pat["col"], owner, rank, gid=gid)
    # position of every local entry in the GLOBAL natural block-CSR (tests compare Jacobian blocks through it):
    # global row-major 7-point pattern -> entry index = rowptr_global[row] + rank of the column inside that row
    row_w = _grid.row_of_entries(pat["rowptr"])[lp["entry"]]
    gi_, gj_ = gid[row_w], gid[pat["col"][lp["entry"]]]
    lp["entry_global"] = _global_entry_index(NX, NY, NZ, gi_, gj_)
    cells = _decks.cartesian_cells(NX, NY, NZ, state=state, heterogeneous=heterogeneous, **kw)   # global per-cell arrays
    g = lp["gids"]
    # per-entry data from the window (transmissibility needs the permeability of both cells of a face)
    permw = cells["perm"][gid]
    dx, dy, dz = cells["dx"], cells["dy"], cells["dz"]
    transw = _grid.tpfa_transmissibility(pat, permw, permw, permw, dx, dy, dz)
    _, _, areaw = _grid.cartesian_geometry(pat, dx, dy, dz, cells["top"])
    case = dict(Nb=lp["Nown"], Nghost=lp["Nghost"], Nloc=lp["Nown"] + lp["Nghost"], rowptr=lp["rows"], col=lp["cols"],
                trans=np.ascontiguousarray(transw[lp["entry"]]), area=np.ascontiguousarray(areaw[lp["entry"]]),
                poro=np.ascontiguousarray(cells["poro"][g]), volume=np.ascontiguousarray(cells["volume"][g]),
                depth=np.ascontiguousarray(cells["depth"][g]), fluid=cells["fluid"],
                pv=np.ascontiguousarray(cells["pv"].reshape(-1, 3)[g].reshape(-1)), meaning=np.ascontiguousarray(cells["meaning"][g]),
                gids=g, halo=lp, global_cells=NX * NY * NZ, nx=NX, ny=NY, nz=NZ)
    src_global = _decks.five_spot_source(dict(Nb=NX * NY * NZ, nx=NX, ny=NY, nz=NZ, pv=cells["pv"], meaning=cells["meaning"], fluid=cells["fluid"]),
                                         rate_sm3_per_day=(rate_scale if rate_scale is not None else _decks.BENCH_RATE_SM3_PER_DAY * (n / 100.0) ** 2) * nranks ** (2.0 / 3.0))
    case["source"] = np.ascontiguousarray(src_global.reshape(-1, 3)[g].reshape(-1))
    return case


def _global_entry_index(NX, NY, NZ, gi, gj):
    """index of entry (gi, gj) in cartesian_pattern(NX, NY, NZ)'s block-CSR, without building it"""
    gi = np.asarray(gi, np.int64)
    gj = np.asarray(gj, np.int64)
    i, j, k = gi % NX, (gi // NX) % NY, gi // (NX * NY)
    # row length and row start: count of valid neighbours of all earlier cells
    def rowlen(i, j, k):
        return 1 + (i > 0) + (i < NX - 1) + (j > 0) + (j < NY - 1) + (k > 0) + (k < NZ - 1)
    # prefix: sum over cells < gi of rowlen = 7*gi - (boundary deficits); compute exactly with per-axis counts
    def deficit_prefix(g):
        # number of missing neighbours among cells [0, g)
        kk, rem = g // (NX * NY), g % (NX * NY)
        jj, ii = rem // NX, rem % NX
        full_planes = kk
        # x deficits: 2 per (row of NX cells) -> cells with i == 0 or i == NX-1
        rows_done = kk * NY + jj
        x_def = 2 * rows_done + (ii > 0) * 1 + (ii > NX - 1) * 1
        # y deficits: per plane 2*NX (j == 0 and j == NY-1 rows)
        y_def = full_planes * 2 * NX + np.where(jj > 0, NX, ii) + np.where(jj > NY - 1, 0, 0) + np.where(jj == NY - 1, ii, 0)
        # z deficits: plane 0 and plane NZ-1
        z_def = np.where(kk > 0, NX * NY, rem) + np.where(kk == NZ - 1, rem, 0)
        return x_def + y_def + z_def
    start = 7 * gi - deficit_prefix(gi)
    # rank of column inside the row: columns ascend: -z, -y, -x, self, +x, +y, +z
    d = gj - gi
    before = np.zeros_like(gi)
    before += (k > 0) & (d > -NX * NY)
    before += (j > 0) & (d > -NX)
    before += (i > 0) & (d > -1)
    before += (d > 0)
    before += (i < NX - 1) & (d > 1)
    before += (j < NY - 1) & (d > NX)
    return start + before
