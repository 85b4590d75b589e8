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

Not restated: MINPV (which cells are pinched out is the caller's ACTNUM; `cornerpoint_faces(pinch=...)` then makes the
connections across them), boundary and thermal half transmissibilities, diffusivities.  Faces must be given once, cell1 = the cell with the lower Cartesian index
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


# ---- corner-point grids (COORD / ZCORN) -----------------------------------------------------------------------------------
# The reference leaves this to its grid library (Dune::CpGrid of opm-grid, absent from the reference tree: processEclipseFormat
# builds the cells and the faces, faceCenterEcl / faceAreaNormalEcl serve them to EclTransmissibility, :760-811) and to
# EclipseGrid of opm-common (getCellCenter, getCellVolume, getCellDepth, :174-182).  What those return is restated here from
# the published definitions - UNVERIFIED against upstream, no reference numbers exist in the tree (tests/test_transmissibility.py
# holds the properties: Cartesian equivalence, area and volume conservation across faults and under shear):
#   cell centre   mean of the 8 corner points; depth its z; volume of the hexahedron (faces fanned about their centres)
#   face centre   mean of the cell's OWN 4 corners of that face (so the two sides of a fault see different centres)
#   face area     area-weighted normal of the INTERSECTION of the two cells' faces: across a fault the overlap of two
#                 quadrilaterals that share a pillar pair, and a cell then meets every cell of the neighbour column it overlaps
_GAUSS3 = (np.array([-np.sqrt(0.6), 0.0, np.sqrt(0.6)]), np.array([5.0, 8.0, 5.0]) / 9.0)


def cornerpoint_corners(nx, ny, nz, coord, zcorn):
    """-> (N, 2, 2, 2, 3): the corner points of every cell, indexed [cell, kk (0 top, 1 bottom), jj, ii, xyz], cells in natural
    order i + nx (j + ny k).  COORD: (nx+1)(ny+1) pillars x (x, y, z of the top point, x, y, z of the bottom point), the
    pillar of node (ip, jp) at ip + (nx+1) jp; ZCORN: 8 nx ny nz depths, layer by layer top face then bottom face, each a
    (2 ny) x (2 nx) array of the corner depths.  x, y of a corner: on its pillar at its depth (a vertical or degenerate
    pillar keeps its top x, y)."""
    coord = np.asarray(coord, float).reshape(ny + 1, nx + 1, 6)
    z = np.asarray(zcorn, float).reshape(nz, 2, ny, 2, nx, 2)       # [k, kk, j, jj, i, ii]
    z = np.transpose(z, (0, 2, 4, 1, 3, 5))                        # [k, j, i, kk, jj, ii]
    out = np.empty((nz, ny, nx, 2, 2, 2, 3))
    out[..., 2] = z
    for jj in range(2):
        for ii in range(2):
            p = coord[jj:jj + ny, ii:ii + nx]                      # pillar of this corner, per (j, i)
            zt, zb = p[..., 2], p[..., 5]
            dz = zb - zt
            safe = np.where(dz != 0.0, dz, 1.0)
            for kk in range(2):
                zz = z[:, :, :, kk, jj, ii]
                w = np.where(dz != 0.0, (zz - zt) / safe, 0.0)
                out[:, :, :, kk, jj, ii, 0] = p[..., 0] + w * (p[..., 3] - p[..., 0])
                out[:, :, :, kk, jj, ii, 1] = p[..., 1] + w * (p[..., 4] - p[..., 1])
    return out.reshape(nx * ny * nz, 2, 2, 2, 3)


def _quad_area_vector(p00, p01, p11, p10):
    """area-weighted normal of the (possibly non-planar) quadrilateral p00 -> p01 -> p11 -> p10: half the cross product of its
    diagonals, which is the vector area of EVERY surface spanned by that edge loop"""
    return 0.5 * np.cross(p11 - p00, p10 - p01)


def hexahedron_volume(c):
    """c (N, 2, 2, 2, 3) -> volumes: divergence theorem over the six faces, each fanned into four triangles about the mean of
    its corners"""
    ctr = c.reshape(len(c), 8, 3).mean(axis=1)
    quads = (  # outward orientation for a cell with x to the right, y to the back, z DOWN (kk = 1 is deeper)
        (c[:, 0, 0, 0], c[:, 0, 1, 0], c[:, 0, 1, 1], c[:, 0, 0, 1]),   # top (kk = 0)
        (c[:, 1, 0, 0], c[:, 1, 0, 1], c[:, 1, 1, 1], c[:, 1, 1, 0]),   # bottom
        (c[:, 0, 0, 0], c[:, 1, 0, 0], c[:, 1, 1, 0], c[:, 0, 1, 0]),   # x- (ii = 0)
        (c[:, 0, 0, 1], c[:, 0, 1, 1], c[:, 1, 1, 1], c[:, 1, 0, 1]),   # x+
        (c[:, 0, 0, 0], c[:, 0, 0, 1], c[:, 1, 0, 1], c[:, 1, 0, 0]),   # y- (jj = 0)
        (c[:, 0, 1, 0], c[:, 1, 1, 0], c[:, 1, 1, 1], c[:, 0, 1, 1]),   # y+
    )
    vol = np.zeros(len(c))
    for q in quads:
        m = (q[0] + q[1] + q[2] + q[3]) / 4.0
        for a, b in ((0, 1), (1, 2), (2, 3), (3, 0)):
            # tetrahedron (cell centre, face centre, q[a], q[b]); the sign of the whole sum fixes the orientation
            vol += np.einsum("ij,ij->i", m - ctr, np.cross(q[a] - ctr, q[b] - ctr)) / 6.0
    return np.abs(vol)


def _pillar_overlap_area(a_t, a_b, b_t, b_b, p1, p2):
    """The vector area of the overlap of two quadrilaterals that hang between the same two pillars.  a_t, a_b, b_t, b_b: (n, 2)
    depths of the top / bottom edge of side A / side B at pillar 1 and pillar 2 (edges are straight between the pillars in the
    (s, z) chart of the pillar pair, s in [0, 1]); p1, p2: (n, 6) the pillars (top point, bottom point).
    The surface: X(s, z) = (1 - s) P1(z) + s P2(z), P(z) the pillar's point at depth z; its area element dX/ds x dX/dz is
    integrated over { max(tA, tB) <= z <= min(bA, bB) } - between the crossing points of the four edge lines the bounds are
    linear in s and the integrand is a polynomial of degree 3, so 3-point Gauss rules in s and 2 in z are exact."""
    n = len(a_t)
    lin = lambda e, s: e[:, 0:1] + s * (e[:, 1:2] - e[:, 0:1])
    # crossing points of (upper candidates) x (lower candidates) and among each other: tA=tB, bA=bB, tA=bB, tB=bA
    def cross(e, f):
        d0, d1 = e[:, 0] - f[:, 0], e[:, 1] - f[:, 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            s = d0 / (d0 - d1)
        return np.where((d0 * d1 < 0.0), s, 0.0)
    brk = np.stack([np.zeros(n), cross(a_t, b_t), cross(a_b, b_b), cross(a_t, b_b), cross(b_t, a_b), np.ones(n)], axis=1)
    brk = np.sort(np.clip(brk, 0.0, 1.0), axis=1)
    def pillar(p, zz):      # point of pillar p (n, 6) at depth zz (n, m) -> x, y ; and d(x, y)/dz
        dz = p[:, 5] - p[:, 2]
        safe = np.where(dz != 0.0, dz, 1.0)
        gx = np.where(dz != 0.0, (p[:, 3] - p[:, 0]) / safe, 0.0)[:, None]
        gy = np.where(dz != 0.0, (p[:, 4] - p[:, 1]) / safe, 0.0)[:, None]
        return p[:, 0:1] + gx * (zz - p[:, 2:3]), p[:, 1:2] + gy * (zz - p[:, 2:3]), gx, gy
    area = np.zeros((n, 3))
    gs, gw = _GAUSS3
    gz, gzw = np.array([-1.0, 1.0]) / np.sqrt(3.0), np.array([1.0, 1.0])
    for q in range(5):
        s0, s1 = brk[:, q:q + 1], brk[:, q + 1:q + 2]
        hs = 0.5 * (s1 - s0)
        s = 0.5 * (s0 + s1) + hs * gs[None, :]                     # (n, 3)
        up = np.maximum(lin(a_t, s), lin(b_t, s))
        lo = np.minimum(lin(a_b, s), lin(b_b, s))
        h = np.maximum(lo - up, 0.0)
        for zq, zw in zip(gz, gzw):
            zz = 0.5 * (up + lo) + 0.5 * h * zq
            x1, y1, gx1, gy1 = pillar(p1, zz)
            x2, y2, gx2, gy2 = pillar(p2, zz)
            ds = np.stack([x2 - x1, y2 - y1, np.zeros_like(zz)], axis=2)                       # dX/ds
            dzv = np.stack([(1 - s) * gx1 + s * gx2, (1 - s) * gy1 + s * gy2, np.ones_like(zz)], axis=2)   # dX/dz
            wgt = (hs * gw[None, :]) * (0.5 * h * zw)
            area += (np.cross(ds, dzv) * wgt[:, :, None]).sum(axis=1)
    return area


def cornerpoint_faces(nx, ny, nz, coord, zcorn, actnum=None, max_fault_throw=None, pinch=None):
    """Corner-point grid -> what `face_transmissibilities` and `opmhip_set_static` want: the connections between active cells
    with their geometry, plus per-cell centres, volumes and depths (see the comment block above for the definitions).
    Vertical connections: between k and k + 1 of a column (connections ACROSS pinched-out or inactive layers are the grid
    library's PINCH search, not built: hand them over as NNCs).  Lateral connections: every pair of cells of neighbouring
    columns whose faces on the shared pillar pair overlap - k to k for a conforming grid, k to k' across a fault
    (max_fault_throw: limit on |k - k'|, default nz - 1).
    pinch (PINCH item 1, a thickness; None = no PINCH keyword): an active cell whose lower neighbours are inactive is connected
    to the next active cell of its column when the inactive cells in between are together no thicker than `pinch` (mean of the
    four pillar thicknesses) - the connection the grid library creates across pinched-out layers (option GAP of item 2: the
    thickness test applies; TOPBOT geometry: the upper cell's bottom face, each cell's own face centre).  Combine with
    face_transmissibilities(multz_all=...) for item 4 = ALL.
    -> dict(n, cart, faces, centroid, volume, depth, face_area) like cartesian_faces"""
    N = nx * ny * nz
    c = cornerpoint_corners(nx, ny, nz, coord, zcorn)
    coordp = np.asarray(coord, float).reshape(ny + 1, nx + 1, 6)
    cen = c.reshape(N, 8, 3).mean(axis=1)
    vol = hexahedron_volume(c)
    act = np.ones(N, bool) if actnum is None else np.asarray(actnum).reshape(-1) != 0
    comp = np.full(N, -1, np.int64)
    comp[act] = np.arange(act.sum())
    idx = np.arange(N)
    i, j, k = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    F = dict(cell1=[], cell2=[], face1=[], face2=[], center1=[], center2=[], area_normal=[])

    def add(a, b, fa, fb, ca, cb, nrm):
        F["cell1"].append(comp[a]); F["cell2"].append(comp[b])
        F["face1"].append(np.full(len(a), fa)); F["face2"].append(np.full(len(a), fb))
        F["center1"].append(ca); F["center2"].append(cb); F["area_normal"].append(nrm)

    # lateral faces: side S+ of cell a (the corners with ii = 1, or jj = 1) against side S- of the cells of the next column
    throw = nz - 1 if max_fault_throw is None else int(max_fault_throw)
    for axis, (stride, has, fa, fb) in enumerate(((1, i < nx - 1, XP, XM), (nx, j < ny - 1, YP, YM))):
        a0 = idx[has]
        # side corners [cell, kk, e] with e = 0 / 1 the two pillars of the side, ordered so that dX/ds x dX/dz points from a to b
        if axis == 0:
            sa, sb = c[:, :, :, 1], c[:, :, :, 0]                   # [cell, kk, jj]: pillars (i+1, j), (i+1, j+1)
            pil = lambda cells: (coordp[j[cells], i[cells] + 1], coordp[j[cells] + 1, i[cells] + 1])
            sign = 1.0
        else:
            sa, sb = c[:, :, 1, :], c[:, :, 0, :]                   # [cell, kk, ii]: pillars (i, j+1), (i+1, j+1)
            pil = lambda cells: (coordp[j[cells] + 1, i[cells]], coordp[j[cells] + 1, i[cells] + 1])
            sign = -1.0
        for dk in range(-throw, throw + 1):
            m = (k[a0] + dk >= 0) & (k[a0] + dk < nz)
            a = a0[m]
            b = a + stride + dk * nx * ny
            m = act[a] & act[b]
            a, b = a[m], b[m]
            if len(a) == 0:
                continue
            za_t, za_b = sa[a, 0, :, 2], sa[a, 1, :, 2]
            zb_t, zb_b = sb[b, 0, :, 2], sb[b, 1, :, 2]
            # overlap possible at all?  (depths grow downwards)
            m = (np.minimum(za_b, zb_b).max(axis=1) > np.maximum(za_t, zb_t).min(axis=1))
            if not m.any():
                continue
            a, b = a[m], b[m]
            p1, p2 = pil(a)
            nrm = sign * _pillar_overlap_area(za_t[m], za_b[m], zb_t[m], zb_b[m], p1, p2)
            keep = np.abs(nrm).sum(axis=1) > 0.0
            a, b, nrm = a[keep], b[keep], nrm[keep]
            add(a, b, fa, fb, sa[a].reshape(len(a), 4, 3).mean(axis=1), sb[b].reshape(len(b), 4, 3).mean(axis=1), nrm)
    # vertical faces: bottom of a against the top of the cell below
    a = idx[k < nz - 1]
    b = a + nx * ny
    m = act[a] & act[b]
    a, b = a[m], b[m]
    bot = c[a, 1]                                                   # [cell, jj, ii]
    nrm = _quad_area_vector(bot[:, 0, 0], bot[:, 0, 1], bot[:, 1, 1], bot[:, 1, 0])
    add(a, b, ZP, ZM, bot.reshape(len(a), 4, 3).mean(axis=1), c[b, 0].reshape(len(b), 4, 3).mean(axis=1), nrm)
    if pinch is not None:
        thick = (c[:, 1, :, :, 2] - c[:, 0, :, :, 2]).reshape(N, 4).mean(axis=1)
        pa, pb = [], []
        for col in range(nx * ny):
            cells = col + nx * ny * np.arange(nz)
            up = -1          # the last active cell above, and the thickness of the inactive cells passed since
            gap, passed = 0.0, 0
            for q in cells:
                if act[q]:
                    if up >= 0 and passed > 0 and gap <= pinch:
                        pa.append(up); pb.append(q)
                    up, gap, passed = q, 0.0, 0
                else:
                    gap += thick[q]; passed += 1
        if pa:
            a, b = np.array(pa), np.array(pb)
            bot = c[a, 1]
            add(a, b, ZP, ZM, bot.reshape(len(a), 4, 3).mean(axis=1), c[b, 0].reshape(len(b), 4, 3).mean(axis=1),
                _quad_area_vector(bot[:, 0, 0], bot[:, 0, 1], bot[:, 1, 1], bot[:, 1, 0]))
    faces = {key: np.concatenate(v) for key, v in F.items()}
    # one orientation per pair, cell1 the lower Cartesian index (fault connections with dk < 0 may come out reversed)
    swap = faces["cell1"] > faces["cell2"]
    for x, y in (("cell1", "cell2"), ("face1", "face2"), ("center1", "center2")):
        tx = faces[x].copy()
        faces[x] = np.where(swap[(...,) + (None,) * (faces[x].ndim - 1)], faces[y], faces[x])
        faces[y] = np.where(swap[(...,) + (None,) * (faces[y].ndim - 1)], tx, faces[y])
    faces["area_normal"] = np.where(swap[:, None], -faces["area_normal"], faces["area_normal"])
    face_area = np.sqrt((faces["area_normal"] ** 2).sum(axis=1))
    return dict(n=int(act.sum()), cart=idx[act], faces=faces, centroid=cen[act], volume=vol[act], depth=cen[act, 2], face_area=face_area)


def cartesian_cornerpoint(nx, ny, nz, dx, dy, dz, top=0.0, shear=(0.0, 0.0), fault_i=None, throw=0.0, dip=0.0):
    """COORD / ZCORN of a box of nx x ny x nz cells of size dx x dy x dz (test and example input): pillars sheared by
    `shear` = (d x / d z, d y / d z), layers dipping by `dip` = d z / d x, and - fault_i given - the columns i >= fault_i
    thrown down by `throw`.  -> coord ((nx+1)(ny+1), 6), zcorn (8 nx ny nz)"""
    ztot = nz * dz + abs(throw) + abs(dip) * nx * dx
    coord = np.zeros((ny + 1, nx + 1, 6))
    for jp in range(ny + 1):
        for ip in range(nx + 1):
            z0, z1 = top - ztot, top + 2 * ztot
            coord[jp, ip] = [ip * dx + shear[0] * (z0 - top), jp * dy + shear[1] * (z0 - top), z0,
                             ip * dx + shear[0] * (z1 - top), jp * dy + shear[1] * (z1 - top), z1]
    z = np.zeros((nz, 2, ny, 2, nx, 2))
    for kk in range(2):
        for ii in range(2):
            ip = np.arange(nx) + ii
            zz = top + (np.arange(nz)[:, None] + kk) * dz + dip * (ip * dx)[None, :]
            if fault_i is not None:
                zz = zz + np.where(np.arange(nx) >= fault_i, throw, 0.0)[None, :]
            z[:, kk, :, :, :, ii] = zz[:, None, None, :]
    return coord.reshape(-1, 6), z.reshape(-1)
