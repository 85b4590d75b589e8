"""Two-point-flux transmissibilities of the connections of a grid: the input `opmhip_set_static` takes per block-CSR entry.

Host-side restatement of EclTransmissibility::update (ebos/ecltransmissibility.cc:146-500) for what
the per-Newton hot path consumes.  The grid library is not part of the path: the caller hands over the faces with their
geometry (for a corner-point grid: what Dune::CpGrid's faceCenterEcl / faceAreaNormalEcl return, :760-811); `cartesian_faces`
produces them for block-centred grids (DX / DY / DZ / TOPS per cell).

  half transmissibility   K_dd |A . d| / |d|^2, d = face centre - cell centre, K_dd the permeability along the
                          axis of the cell's face (computeHalfTrans_ :928-944, distanceVector_ :962-976)
  NTG                     multiplies the half transmissibilities of x and y faces (applyNtg_ :1016-1043)
  face transmissibility   1 / (1/T1 + 1/T2), zero if a half is below 1e-30 (:352-356)
  MULTX / MULTX- ...      the multiplier of the inside cell's face, then of the outside cell's face (applyMultipliers_
                          :982-1013, called :375-377)
  MULTREGT                a region-pair multiplier per face direction (:381-404)
  NNC / EDITNNC           EDITNNC scales existing connections, NNC adds to a connection of the grid or creates one
                          (applyEditNncToGridTrans_ :864-925, applyNncToGridTrans_ :814-862, order :487-488)

  PINCH option ALL        vertical connections (also those that bridge pinched-out cells) take the SMALLEST MULTZ met going down
                          the pillar from the upper cell to the cell above the lower one, then the lower cell's MULTZ-
                          (applyAllZMultipliers_ :575-612, selected :219-227)

Not restated: the search for PINCH / MINPV connections themselves (the grid library makes them; they arrive here as faces),
boundary and thermal half transmissibilities, diffusivities.  Faces must be given once, cell1 = the cell with the lower Cartesian index
(the reference skips the other orientation, :296-299).  Units: SI (perm m^2, lengths m) -> m^3.
"""
import numpy as np

# indexInInside of the reference element that holds the intersection (:982-1013): left, right, front, back, bottom, top
XM, XP, YM, YP, ZM, ZP = range(6)
_MULT_KEY = {XM: "X-", XP: "X+", YM: "Y-", YP: "Y+", ZM: "Z-", ZP: "Z+"}


def half_transmissibility(perm_dd, area_normal, distance):
    """computeHalfTrans_ (:928-944): halfTrans = K; val = sum_i A_i d_i; halfTrans *= |val|; halfTrans /= |d|^2 - in that order"""
    an, d = np.atleast_2d(area_normal), np.atleast_2d(distance)
    val = np.zeros(len(an))
    for i in range(an.shape[1]):
        val = val + an[:, i] * d[:, i]
    norm2 = np.zeros(len(d))
    for i in range(d.shape[1]):
        norm2 = norm2 + d[:, i] * d[:, i]
    h = np.asarray(perm_dd, float) * np.abs(val)
    return h / norm2


def face_transmissibilities(faces, centroid, perm, ntg=None, mult=None, region_mult=None, multz_all=None):
    """faces: dict(cell1, cell2, face1, face2, center1 (nf, 3), center2 (nf, 3), area_normal (nf, 3)); cell1 / cell2 compressed
    cell indices, face1 / face2 in XM..ZP; center1 / center2: the face centre seen from either cell (they differ across
    faults of a corner-point grid).  centroid (n, 3): cell centres as the input grid computes them (axisCentroids, :163-189).
    perm (n, 3): diagonal of the permeability tensor.  ntg (n) or None.  mult: dict 'X-','X+','Y-','Y+','Z-','Z+' -> (n)
    arrays (MULTX- ... MULTZ), missing = 1.  region_mult(cell1[], cell2[], axis[]) -> factors (MULTREGT) or None.
    multz_all = dict(cart (n: compressed -> Cartesian index), nxny, multz (MULTZ over ALL Cartesian cells)): the PINCH "ALL"
    option for the vertical faces (they must then run top to bottom: cell1 above cell2).
    -> transmissibility per face"""
    c1, c2 = np.asarray(faces["cell1"], np.int64), np.asarray(faces["cell2"], np.int64)
    f1, f2 = np.asarray(faces["face1"], np.int64), np.asarray(faces["face2"], np.int64)
    an = np.asarray(faces["area_normal"], float)
    centroid, perm = np.asarray(centroid, float), np.asarray(perm, float)
    h1 = half_transmissibility(perm[c1, f1 // 2], an, np.asarray(faces["center1"], float) - centroid[c1])
    h2 = half_transmissibility(perm[c2, f2 // 2], an, np.asarray(faces["center2"], float) - centroid[c2])
    if ntg is not None:
        ntg = np.asarray(ntg, float)
        h1 = np.where(f1 < ZM, h1 * ntg[c1], h1)   # NTG does not apply to top and bottom faces
        h2 = np.where(f2 < ZM, h2 * ntg[c2], h2)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = 1.0 / (1.0 / h1 + 1.0 / h2)
    t = np.where((np.abs(h1) < 1e-30) | (np.abs(h2) < 1e-30), 0.0, t)
    zall = np.zeros(len(t), bool)
    if multz_all is not None:
        # applyAllZMultipliers_: the smallest MULTZ between the two cells of a vertical connection, pinched-out cells included
        cart, nxny, mz = np.asarray(multz_all["cart"], np.int64), int(multz_all["nxny"]), np.asarray(multz_all["multz"], float)
        zall = f1 > YP
        if np.any(f1[zall] != ZP):
            raise ValueError("multz_all: vertical faces must be given from the upper cell (face1 = ZP)")
        for q in np.nonzero(zall)[0]:
            a, b = cart[c1[q]], cart[c2[q]]
            last = b - nxny
            m = mz[last]
            for cc in range(a, last, nxny):
                m = min(m, mz[cc])
            t[q] = t[q] * m
        if mult and "Z-" in mult:
            t = np.where(zall, t * np.asarray(mult["Z-"], float)[c2], t)     # the outside element's face (:606)
    if mult:
        for cells, fidx in ((c1, f1), (c2, f2)):   # the inside element's face first, then the outside element's
            m = np.ones(len(t))
            for f, key in _MULT_KEY.items():
                if key in mult:
                    sel = (fidx == f) & ~zall
                    m[sel] = np.asarray(mult[key], float)[cells[sel]]
            t = t * m
    if region_mult is not None:
        t = t * region_mult(c1, c2, f1 // 2)
    return t


def apply_nnc(cell1, cell2, trans, nnc=(), editnnc=()):
    """EDITNNC first, then NNC (:487-488).  nnc / editnnc: iterables of (cell a, cell b, value) on compressed indices (-1 =
    inactive).  EDITNNC multiplies an existing connection (unknown pairs are reported back); an NNC between active cells adds
    to an existing connection or - "not resembled by the grid" - becomes a connection of its own with that transmissibility.
    -> cell1, cell2, trans (grid connections first, new ones appended), list of EDITNNC entries without a connection"""
    key = {}
    c1, c2, t = list(map(int, cell1)), list(map(int, cell2)), list(map(float, trans))
    for q, (a, b) in enumerate(zip(c1, c2)):
        key[(min(a, b), max(a, b))] = q
    unmatched = []
    for a, b, v in editnnc:
        q = key.get((min(a, b), max(a, b)))
        if q is None or a < 0 or b < 0:
            unmatched.append((a, b, v))
        else:
            t[q] *= v
    for a, b, v in nnc:
        lo, hi = min(a, b), max(a, b)
        if lo < 0:          # both inactive: silently dropped; one inactive: dropped with a warning in the reference
            continue
        q = key.get((lo, hi))
        if q is None:
            key[(lo, hi)] = len(t)
            c1.append(lo); c2.append(hi); t.append(float(v))
        else:
            t[q] += v
    return np.array(c1, np.int64), np.array(c2, np.int64), np.array(t), unmatched


def cartesian_faces(nx, ny, nz, dx, dy, dz, tops, actnum=None):
    """Block-centred grid (DX / DY / DZ per cell, TOPS for the top layer or for every cell): the interior faces between active
    cells with the geometry the transmissibilities want, plus per-cell centres, volumes and depths.  A face between two
    cells of different size takes the overlap-free convention of a block-centred grid: area and centre of the inside
    (lower-index) cell's face, as seen from each cell at its own depth.
    -> dict(n, cart (compressed -> Cartesian), faces, centroid, volume, depth, face_area)"""
    N = nx * ny * nz
    def full(a):
        a = np.asarray(a, float).reshape(-1)
        if a.size not in (1, N):
            raise ValueError("DX / DY / DZ: one value or one per cell")
        return np.full(N, a[0]) if a.size == 1 else a.copy()
    DX, DY, DZ = full(dx), full(dy), full(dz)
    idx = np.arange(N)
    i, j, k = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    tops = np.asarray(tops, float).reshape(-1)
    if tops.size == N:
        ztop = tops.copy()
    else:
        if tops.size not in (1, nx * ny):
            raise ValueError("TOPS: one value, one per column or one per cell")
        ztop = np.zeros(N)
        ztop[: nx * ny] = tops
        for kk in range(1, nz):
            lo, up = slice(kk * nx * ny, (kk + 1) * nx * ny), slice((kk - 1) * nx * ny, kk * nx * ny)
            ztop[lo] = ztop[up] + DZ[up]
    # x / y positions: cumulative sums along each row of cells
    X = DX.reshape(nz, ny, nx); Y = DY.reshape(nz, ny, nx)
    x0 = (np.cumsum(X, axis=2) - X).reshape(-1)
    y0 = (np.cumsum(Y, axis=1) - Y).reshape(-1)
    cen = np.stack([x0 + 0.5 * DX, y0 + 0.5 * DY, ztop + 0.5 * DZ], axis=1)
    act = np.ones(N, bool) if actnum is None else np.asarray(actnum).reshape(-1) != 0
    comp = np.full(N, -1, np.int64)
    comp[act] = np.arange(act.sum())
    F = dict(cell1=[], cell2=[], face1=[], face2=[], center1=[], center2=[], area_normal=[])
    for axis, (stride, has, fa, fb) in enumerate(((1, i < nx - 1, XP, XM), (nx, j < ny - 1, YP, YM), (nx * ny, k < nz - 1, ZP, ZM))):
        a = idx[has]
        b = a + stride
        m = act[a] & act[b]
        a, b = a[m], b[m]
        size = [DX, DY, DZ]
        area = np.ones(len(a))
        for o in range(3):
            if o != axis:
                area = area * size[o][a]
        nrm = np.zeros((len(a), 3)); nrm[:, axis] = area
        ca, cb = cen[a].copy(), cen[b].copy()
        ca[:, axis] += 0.5 * size[axis][a]
        cb[:, axis] -= 0.5 * size[axis][b]
        F["cell1"].append(comp[a]); F["cell2"].append(comp[b])
        F["face1"].append(np.full(len(a), fa)); F["face2"].append(np.full(len(a), fb))
        F["center1"].append(ca); F["center2"].append(cb); F["area_normal"].append(nrm)
    faces = {key: np.concatenate(v) for key, v in F.items()}
    face_area = np.abs(faces["area_normal"]).sum(axis=1)
    return dict(n=int(act.sum()), cart=idx[act], faces=faces, centroid=cen[act], volume=(DX * DY * DZ)[act], depth=cen[act, 2], face_area=face_area)


def connections_to_pattern(n, cell1, cell2, trans, area=None):
    """Connections -> the block-CSR pattern of the element-centred stencil (diagonal + both orientations of every connection,
    columns ascending) with per-entry transmissibility and face area (1 where none is known: NNCs) - the arrays
    opmhip_set_pattern / opmhip_set_static take.  -> dict(Nb, rowptr, col, trans, area, conn (entry -> connection, -1 diagonal))"""
    c1, c2 = np.asarray(cell1, np.int64), np.asarray(cell2, np.int64)
    nc = len(c1)
    area = np.ones(nc) if area is None else np.concatenate([np.asarray(area, float), np.ones(nc - len(area))])
    rows = np.concatenate([np.arange(n), c1, c2])
    cols = np.concatenate([np.arange(n), c2, c1])
    conn = np.concatenate([np.full(n, -1), np.arange(nc), np.arange(nc)])
    order = np.lexsort((cols, rows))
    rows, cols, conn = rows[order], cols[order], conn[order]
    if np.any((rows[1:] == rows[:-1]) & (cols[1:] == cols[:-1])):
        raise ValueError("duplicate connection: merge it first (apply_nnc adds an NNC to the connection of the grid)")
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    t = np.where(conn >= 0, np.asarray(trans, float)[np.maximum(conn, 0)], 0.0)
    a = np.where(conn >= 0, area[np.maximum(conn, 0)], 1.0)
    return dict(Nb=n, rowptr=rowptr.astype(np.int32), col=cols.astype(np.int32), trans=np.ascontiguousarray(t), area=np.ascontiguousarray(a), conn=conn)
