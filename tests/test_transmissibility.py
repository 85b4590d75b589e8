"""Deck-side inputs of the hot path: transmissibilities (ebos/ecltransmissibility.cc) and
threshold pressures (ebos/eclthresholdpressure.hh, eclgenericthresholdpressure.cc).  The reference holds no numbers for
either (tests/test_thresholdpressure.cpp only checks that a vector comes back), so these are property tests of the
restated formulas: unpinned, as DESIGN.md says."""
import importlib

import numpy as np
import pytest

pkg = importlib.import_module("opm-autodiff_amd")
T = pkg.transmissibility


def uniform(nx=4, ny=3, nz=5, dx=20.0, dy=30.0, dz=4.0):
    g = T.cartesian_faces(nx, ny, nz, dx, dy, dz, 1000.0)
    rng = np.random.default_rng(3)
    perm = rng.uniform(1e-14, 1e-12, (g["n"], 3))
    return g, perm


def test_uniform_grid_reproduces_the_cartesian_formula_bit_for_bit():
    nx, ny, nz, dx, dy, dz = 4, 3, 5, 20.0, 30.0, 4.0
    g, perm = uniform(nx, ny, nz, dx, dy, dz)
    t = T.face_transmissibilities(g["faces"], g["centroid"], perm)
    pat = T.connections_to_pattern(g["n"], g["faces"]["cell1"], g["faces"]["cell2"], t, g["face_area"])
    ref_pat = pkg.grid.cartesian_pattern(nx, ny, nz)
    assert np.array_equal(pat["rowptr"], ref_pat["rowptr"]) and np.array_equal(pat["col"], ref_pat["col"])
    ref_t = pkg.grid.tpfa_transmissibility(ref_pat, perm[:, 0].copy(), perm[:, 1].copy(), perm[:, 2].copy(), dx, dy, dz)
    _, depth, ref_area = pkg.grid.cartesian_geometry(ref_pat, dx, dy, dz, 1000.0)
    # the general formula sums A.d over three components (two of them zero) and |d|^2 likewise: same roundings
    np.testing.assert_allclose(pat["trans"], ref_t, rtol=4e-16, atol=0.0)
    off = pat["conn"] >= 0
    assert np.array_equal(pat["area"][off], ref_area[off])   # the diagonal entries carry no face
    np.testing.assert_allclose(g["depth"], depth, rtol=1e-15)
    assert np.array_equal(g["volume"], np.full(g["n"], dx * dy * dz))


def test_ntg_multipliers_and_region_multipliers():
    g, perm = uniform()
    F = g["faces"]
    base = T.face_transmissibilities(F, g["centroid"], perm)
    ntg = np.random.default_rng(1).uniform(0.2, 1.0, g["n"])
    t = T.face_transmissibilities(F, g["centroid"], perm, ntg=ntg)
    z = F["face1"] >= T.ZM
    assert np.array_equal(t[z], base[z])                       # NTG leaves vertical transmissibilities alone (:1016-1043)
    c1, c2 = F["cell1"][~z], F["cell2"][~z]
    ax = F["face1"][~z] // 2
    h1 = perm[c1, ax] * ntg[c1]
    h2 = perm[c2, ax] * ntg[c2]
    # the geometric factor of both halves is the same on a uniform grid: T ~ harmonic mean of K * NTG
    np.testing.assert_allclose(t[~z] / base[~z], (1.0 / (1.0 / h1 + 1.0 / h2)) / (1.0 / (1.0 / perm[c1, ax] + 1.0 / perm[c2, ax])), rtol=1e-13)
    # MULTX acts on the + face of its own cell, MULTX- on the - face (:982-1013): the face between i and i+1 takes both
    mx, mxm = np.full(g["n"], 1.0), np.full(g["n"], 1.0)
    mx[5], mxm[6] = 0.5, 0.25
    t = T.face_transmissibilities(F, g["centroid"], perm, mult={"X+": mx, "X-": mxm})
    ratio = t / base
    hit = (F["cell1"] == 5) & (F["cell2"] == 6)
    assert hit.sum() == 1 and t[hit][0] == base[hit][0] * 0.5 * 0.25
    other = (F["face1"] == T.XP) & (((F["cell1"] == 5) | (F["cell2"] == 6)) & ~hit)
    assert np.all(ratio[~hit & ~other] == 1.0)
    # zero permeability closes a face without a division by zero (:352-356)
    p0 = perm.copy(); p0[7] = 0.0
    t = T.face_transmissibilities(F, g["centroid"], p0)
    assert np.all(t[(F["cell1"] == 7) | (F["cell2"] == 7)] == 0.0) and np.all(np.isfinite(t))
    # MULTREGT: a factor per region pair and direction
    reg = (np.arange(g["n"]) % 2)
    t = T.face_transmissibilities(F, g["centroid"], perm, region_mult=lambda a, b, axis: np.where((reg[a] != reg[b]) & (axis == 0), 0.1, 1.0))
    sel = (reg[F["cell1"]] != reg[F["cell2"]]) & (F["face1"] // 2 == 0)
    assert np.array_equal(t[sel], base[sel] * 0.1) and np.array_equal(t[~sel], base[~sel])


def test_layered_grid_and_inactive_cells():
    nx, ny, nz = 3, 2, 4
    dz = np.repeat([2.0, 6.0, 3.0, 5.0], nx * ny)
    act = np.ones(nx * ny * nz, int); act[7] = 0
    g = T.cartesian_faces(nx, ny, nz, 10.0, 10.0, dz, 2000.0, actnum=act)
    assert g["n"] == nx * ny * nz - 1 and 7 not in g["cart"]
    np.testing.assert_allclose(g["depth"][0], 2001.0)
    np.testing.assert_allclose(g["depth"][-1], 2000.0 + 2 + 6 + 3 + 2.5)
    perm = np.full((g["n"], 3), 1e-13)
    t = T.face_transmissibilities(g["faces"], g["centroid"], perm)
    F = g["faces"]
    zf = F["face1"] == T.ZP
    # vertical face between layers of thickness a and b: K A / ((a + b) / 2)
    a, b = g["volume"][F["cell1"][zf]] / 100.0, g["volume"][F["cell2"][zf]] / 100.0
    np.testing.assert_allclose(t[zf], 1e-13 * 100.0 / (0.5 * (a + b)), rtol=1e-13)
    assert not np.any((g["cart"][F["cell1"]] == 7) | (g["cart"][F["cell2"]] == 7))


def test_nnc_and_editnnc():
    g, perm = uniform(3, 1, 2)
    F = g["faces"]
    t = T.face_transmissibilities(F, g["centroid"], perm)
    c1, c2, t2, bad = T.apply_nnc(F["cell1"], F["cell2"], t, nnc=[(0, 5, 7.0), (1, 0, 2.0), (-1, 3, 1.0), (-1, -1, 1.0)], editnnc=[(0, 1, 3.0), (0, 4, 9.0)])
    q = np.nonzero((F["cell1"] == 0) & (F["cell2"] == 1))[0][0]
    assert t2[q] == t[q] * 3.0 + 2.0                      # EDITNNC scales first, the NNC then adds (:487-488)
    assert (c1[-1], c2[-1], t2[-1]) == (0, 5, 7.0)        # not resembled by the grid: a connection of its own
    assert len(t2) == len(t) + 1 and bad == [(0, 4, 9.0)]
    pat = T.connections_to_pattern(g["n"], c1, c2, t2, g["face_area"])
    row = np.repeat(np.arange(g["n"]), np.diff(pat["rowptr"]))
    k = np.nonzero((row == 5) & (pat["col"] == 0))[0][0]
    assert pat["trans"][k] == 7.0 and pat["area"][k] == 1.0
    # symmetric pattern: every entry has its transpose
    s = set(zip(row.tolist(), pat["col"].tolist()))
    assert all((b, a) in s for a, b in s)


def test_threshold_pressures():
    H = pkg.thpres
    g, perm = uniform(4, 1, 3)
    F = g["faces"]
    t = T.face_transmissibilities(F, g["centroid"], perm)
    n = g["n"]
    eql = np.array([0, 0, 1, 2] * 3)
    rng = np.random.default_rng(0)
    iq = np.zeros((n, 17, 4))
    iq[:, H.F_P:H.F_P + 3, 0] = rng.uniform(200e5, 210e5, (n, 3))
    iq[:, H.F_RHO:H.F_RHO + 3, 0] = np.array([1000.0, 800.0, 150.0])
    iq[:, H.F_MOB:H.F_MOB + 3, 0] = rng.uniform(0.0, 1.0, (n, 3))
    iq[:, H.F_MOB + 2, 0] = 0.0                             # gas immobile everywhere: it must not set a default
    d = H.default_threshold_pressures(eql, 3, F["cell1"], F["cell2"], t, g["face_area"], iq, g["depth"])
    assert np.array_equal(d, d.T) and np.all(np.diag(d) == 0.0) and d[0, 2] == 0.0   # regions 0 and 2 never meet
    # brute force over the faces, both orientations, as computeDefaultThresholdPressures_ walks them
    exp = np.zeros((3, 3))
    for a, b in zip(F["cell1"], F["cell2"]):
        if eql[a] == eql[b]:
            continue
        for i, j in ((a, b), (b, a)):
            for ph in range(2):
                rho = (iq[i, H.F_RHO + ph, 0] + iq[j, H.F_RHO + ph, 0]) / 2.0
                dp = iq[j, H.F_P + ph, 0] + rho * ((g["depth"][i] - g["depth"][j]) * H.GRAVITY) - iq[i, H.F_P + ph, 0]
                up = j if dp > 0 else i
                if iq[up, H.F_MOB + ph, 0] > 0.0:
                    exp[eql[a], eql[b]] = exp[eql[b], eql[a]] = max(exp[eql[a], eql[b]], abs(dp))
    np.testing.assert_allclose(d, exp, rtol=1e-15)
    m = H.threshold_pressure_matrix(3, [(1, 2, 12.0e5), (2, 3, None), (1, 3, 5.0e5)], eql, F["cell1"], F["cell2"], defaults=d)
    assert m[0, 1] == m[1, 0] == 12.0e5 and m[1, 2] == d[1, 2] and m[0, 2] == 0.0   # 1-3 has a record but no common face
    with pytest.raises(ValueError):
        H.threshold_pressure_matrix(3, [(2, 3, None)], eql, F["cell1"], F["cell2"])
    pat = T.connections_to_pattern(n, F["cell1"], F["cell2"], t, g["face_area"])
    e = H.per_entry(pat["rowptr"], pat["col"], eql, m)
    row = np.repeat(np.arange(n), np.diff(pat["rowptr"]))
    assert np.all(e[eql[row] == eql[pat["col"]]] == 0.0)
    assert np.all(e[(eql[row] == 0) & (eql[pat["col"]] == 1)] == 12.0e5)


def test_multz_all_takes_the_smallest_multiplier_down_the_pillar():
    """PINCH option ALL (ecltransmissibility.cc:575-612): one column of 5 layers, layers 1 and 2 pinched out, the connection
    0 -> 3 takes min(MULTZ[0], MULTZ[1], MULTZ[2]) and MULTZ-[3]; a plain neighbour connection its upper cell's MULTZ only"""
    act = np.array([1, 0, 0, 1, 1])
    g = T.cartesian_faces(1, 1, 5, 10.0, 10.0, 2.0, 0.0, actnum=act)
    assert g["n"] == 3 and len(g["faces"]["cell1"]) == 1            # only 3 -> 4 touch; the pinch connection is added by hand
    F = {k: v.copy() for k, v in g["faces"].items()}
    F["cell1"] = np.concatenate([[0], F["cell1"]]); F["cell2"] = np.concatenate([[1], F["cell2"]])
    F["face1"] = np.concatenate([[T.ZP], F["face1"]]); F["face2"] = np.concatenate([[T.ZM], F["face2"]])
    F["center1"] = np.vstack([[[5.0, 5.0, 2.0]], F["center1"]]); F["center2"] = np.vstack([[[5.0, 5.0, 6.0]], F["center2"]])
    F["area_normal"] = np.vstack([[[0.0, 0.0, 100.0]], F["area_normal"]])
    perm = np.full((3, 3), 1e-13)
    base = T.face_transmissibilities(F, g["centroid"], perm)
    mz_cart = np.array([0.8, 0.3, 0.5, 0.9, 1.0])
    mzm = np.array([1.0, 0.5, 1.0])                                   # MULTZ- per compressed cell
    t = T.face_transmissibilities(F, g["centroid"], perm, mult={"Z+": mz_cart[g["cart"]], "Z-": mzm},
                                  multz_all=dict(cart=g["cart"], nxny=1, multz=mz_cart))
    assert t[0] == base[0] * 0.3 * 0.5                                # min(0.8, 0.3, 0.5) and MULTZ- of cell 3
    assert t[1] == base[1] * 0.9 * 1.0
    # without the option: the upper cell's own MULTZ and the lower cell's MULTZ-
    t = T.face_transmissibilities(F, g["centroid"], perm, mult={"Z+": mz_cart[g["cart"]], "Z-": mzm})
    assert t[0] == base[0] * 0.8 * 0.5


# ---- corner-point geometry (cornerpoint_faces): no reference numbers exist for it, so properties -----------------------------
def _by_pair(g):
    f = g["faces"]
    return {(int(a), int(b)): q for q, (a, b) in enumerate(zip(f["cell1"], f["cell2"]))}


def test_cornerpoint_box_equals_the_block_centred_grid():
    nx, ny, nz, dx, dy, dz = 4, 3, 5, 20.0, 30.0, 4.0
    coord, zcorn = T.cartesian_cornerpoint(nx, ny, nz, dx, dy, dz, top=1000.0)
    act = np.ones(nx * ny * nz, int); act[[7, 31]] = 0
    g = T.cornerpoint_faces(nx, ny, nz, coord, zcorn, actnum=act)
    h = T.cartesian_faces(nx, ny, nz, dx, dy, dz, 1000.0, actnum=act)
    assert g["n"] == h["n"] == nx * ny * nz - 2 and np.array_equal(g["cart"], h["cart"])
    np.testing.assert_allclose(g["volume"], h["volume"], rtol=1e-14)
    np.testing.assert_allclose(g["centroid"], h["centroid"], rtol=1e-14)
    pg, ph = _by_pair(g), _by_pair(h)
    assert set(pg) == set(ph) and all(a < b for a, b in pg)
    og, oh = [pg[k] for k in sorted(pg)], [ph[k] for k in sorted(ph)]
    for key in ("face1", "face2"):
        assert np.array_equal(g["faces"][key][og], h["faces"][key][oh])
    for key in ("center1", "center2", "area_normal"):
        np.testing.assert_allclose(g["faces"][key][og], h["faces"][key][oh], rtol=1e-13, atol=1e-9)
    perm = np.random.default_rng(2).uniform(1e-14, 1e-12, (g["n"], 3))
    np.testing.assert_allclose(T.face_transmissibilities(g["faces"], g["centroid"], perm)[og],
                               T.face_transmissibilities(h["faces"], h["centroid"], perm)[oh], rtol=1e-12)


def test_cornerpoint_fault_splits_faces_and_conserves_area():
    """columns i >= 2 thrown down by 1.5 layers: a cell left of the fault meets the two cells it overlaps, half a face each"""
    nx, ny, nz, dx, dy, dz = 4, 2, 6, 10.0, 10.0, 2.0
    coord, zcorn = T.cartesian_cornerpoint(nx, ny, nz, dx, dy, dz, top=0.0, fault_i=2, throw=1.5 * dz)
    g = T.cornerpoint_faces(nx, ny, nz, coord, zcorn)
    f = g["faces"]
    cell = lambda i, j, k: i + nx * (j + ny * k)
    pairs = _by_pair(g)
    for j in range(ny):
        for k in range(nz):
            a = cell(1, j, k)
            partners = {b: q for (x, b), q in pairs.items() if x == a and b % nx == 2} | {x: q for (x, b), q in pairs.items() if b == a and x % nx == 2}
            expect = {cell(2, j, kk) for kk in (k - 2, k - 1) if 0 <= kk < nz}
            assert set(partners) == expect
            for b, q in partners.items():
                nrm = f["area_normal"][q] * (1.0 if f["cell1"][q] == a else -1.0)
                np.testing.assert_allclose(nrm, [0.5 * dz * dy, 0.0, 0.0], rtol=1e-13, atol=1e-12)
                # the centres are those of the cells' own faces, not of the overlap: they differ by the throw
                ca, cb = (f["center1"][q], f["center2"][q]) if f["cell1"][q] == a else (f["center2"][q], f["center1"][q])
                np.testing.assert_allclose(ca, [2 * dx, (j + 0.5) * dy, (k + 0.5) * dz], atol=1e-12)
                np.testing.assert_allclose(cb, [2 * dx, (j + 0.5) * dy, (b // (nx * ny) + 0.5) * dz + 1.5 * dz], atol=1e-12)
    # away from the fault nothing changed; the thrown block kept its volumes
    np.testing.assert_allclose(g["volume"], dx * dy * dz, rtol=1e-14)
    assert (cell(0, 0, 0), cell(1, 0, 0)) in pairs and (cell(2, 0, 0), cell(3, 0, 0)) in pairs
    # max_fault_throw limits the search
    g1 = T.cornerpoint_faces(nx, ny, nz, coord, zcorn, max_fault_throw=1)
    assert all(abs(a // (nx * ny) - b // (nx * ny)) <= 1 for a, b in _by_pair(g1))


def test_cornerpoint_scissor_fault_with_crossing_edges():
    """the throw grows along the fault from -0.7 to +1.9 layers, so that edges of the two sides cross between the pillars: the
    overlaps of one cell's face with all the cells of the other column still add up to that face (where the column covers
    it), and one of them is checked against a brute-force count in the (s, z) chart"""
    nx, ny, nz, dx, dy, dz = 2, 3, 8, 10.0, 12.0, 2.0
    coord, zcorn = T.cartesian_cornerpoint(nx, ny, nz, dx, dy, dz, top=0.0)
    z = zcorn.reshape(nz, 2, ny, 2, nx, 2).copy()
    throw = lambda jp: (-0.7 + 2.6 * jp / ny) * dz
    for jj in range(2):
        for j in range(ny):
            z[:, :, j, jj, 1, :] += throw(j + jj)
    g = T.cornerpoint_faces(nx, ny, nz, coord, z.reshape(-1))
    f = g["faces"]
    cell = lambda i, j, k: i + nx * (j + ny * k)
    for j in range(ny):
        for k in range(3, nz - 3):       # faces fully covered by the other column
            a = cell(0, j, k)
            tot = sum(f["area_normal"][q][0] for (x, b), q in _by_pair(g).items() if x == a and b % nx == 1)
            tot += sum(-f["area_normal"][q][0] for (x, b), q in _by_pair(g).items() if b == a and x % nx == 1)
            np.testing.assert_allclose(tot, dy * dz, rtol=1e-12)
    # brute force in the (s, z) chart for every partner of cell (0, 1, 4)
    a = cell(0, 1, 4)
    s = (np.arange(4000) + 0.5) / 4000
    tA, bA = 4 * dz, 5 * dz
    shapes = set()
    for kb in range(nz):
        b = cell(1, 1, kb)
        tB = kb * dz + throw(1) + s * (throw(2) - throw(1)); bB = tB + dz
        h = np.clip(np.minimum(bA, bB) - np.maximum(tA, tB), 0.0, None)
        q = _by_pair(g).get((min(a, b), max(a, b)))
        if h.max() == 0.0:
            assert q is None
            continue
        np.testing.assert_allclose(abs(f["area_normal"][q][0]), dy * h.mean(), rtol=1e-5, atol=1e-6 * dy * dz)   # midpoint rule of the check
        shapes.add(("partly uncovered" if (h == 0.0).any() else "covered", "edges cross" if np.ptp(np.sign(tB - tA)) + np.ptp(np.sign(bB - bA)) > 0 else "no crossing"))
    assert ("partly uncovered", "no crossing") in shapes or ("partly uncovered", "edges cross") in shapes
    assert any(sh[1] == "edges cross" for sh in shapes)


def test_cornerpoint_shear_and_dip():
    nx, ny, nz, dx, dy, dz = 3, 3, 4, 10.0, 8.0, 2.0
    # sheared pillars: volumes unchanged, side faces are parallelograms spanned by the pillar direction
    coord, zcorn = T.cartesian_cornerpoint(nx, ny, nz, dx, dy, dz, top=100.0, shear=(0.3, -0.2))
    g = T.cornerpoint_faces(nx, ny, nz, coord, zcorn)
    np.testing.assert_allclose(g["volume"], dx * dy * dz, rtol=1e-13)
    f = g["faces"]
    pillar = np.array([0.3 * dz, -0.2 * dz, dz])
    for q in range(len(f["cell1"])):
        if f["face1"][q] == T.XP:
            exp = np.cross([0.0, dy, 0.0], pillar)
        elif f["face1"][q] == T.YP:
            exp = -np.cross([dx, 0.0, 0.0], pillar)
        else:
            exp = np.array([0.0, 0.0, dx * dy])
        np.testing.assert_allclose(f["area_normal"][q], exp, rtol=1e-12, atol=1e-10)
    # the centre of a sheared cell sits on the sheared axis
    np.testing.assert_allclose(g["centroid"][0], [0.5 * dx + 0.3 * 0.5 * dz, 0.5 * dy - 0.2 * 0.5 * dz, 100.0 + 0.5 * dz], rtol=1e-13)
    # dipping layers (dz/dx = 0.1), vertical pillars: conforming, lateral faces keep their area, top / bottom faces tilt
    coord, zcorn = T.cartesian_cornerpoint(nx, ny, nz, dx, dy, dz, top=100.0, dip=0.1)
    g = T.cornerpoint_faces(nx, ny, nz, coord, zcorn)
    f = g["faces"]
    assert len(f["cell1"]) == 3 * nx * ny * nz - nx * ny - ny * nz - nx * nz
    np.testing.assert_allclose(g["volume"], dx * dy * dz, rtol=1e-13)
    for q in range(len(f["cell1"])):
        exp = {T.XP: [dy * dz, 0.0, 0.0], T.YP: [0.0, dx * dz, 0.0], T.ZP: [-0.1 * dx * dy, 0.0, dx * dy]}[int(f["face1"][q])]
        np.testing.assert_allclose(f["area_normal"][q], exp, rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(g["depth"][:nx], 100.0 + 0.5 * dz + 0.1 * dx * (np.arange(nx) + 0.5), rtol=1e-14)


def test_cornerpoint_nonplanar_face_and_degenerate_pillar():
    """one corner of a cell pulled down: the top face is no longer planar - its vector area is half the cross product of the
    diagonals; a pillar given as a single point (top = bottom) keeps its x, y"""
    coord, zcorn = T.cartesian_cornerpoint(1, 1, 2, 10.0, 10.0, 2.0, top=0.0)
    coord = coord.copy(); coord[0, 3:] = coord[0, :3]          # degenerate pillar 0
    z = zcorn.reshape(2, 2, 1, 2, 1, 2).copy()
    z[0, 1, 0, 1, 0, 1] += 0.6; z[1, 0, 0, 1, 0, 1] += 0.6     # the shared corner (jj = 1, ii = 1) of the interface
    g = T.cornerpoint_faces(1, 1, 2, coord, z.reshape(-1))
    f = g["faces"]
    assert len(f["cell1"]) == 1
    p = np.array([[0, 0, 2.0], [10, 0, 2.0], [10, 10, 2.6], [0, 10, 2.0]], float)
    np.testing.assert_allclose(f["area_normal"][0], 0.5 * np.cross(p[2] - p[0], p[3] - p[1]), rtol=1e-13)
    np.testing.assert_allclose(g["volume"].sum(), 10.0 * 10.0 * 4.0, rtol=1e-13)     # what one cell lost the other gained
    c = T.cornerpoint_corners(1, 1, 2, coord, z.reshape(-1))
    assert np.all(c[:, :, 0, 0, 0] == 0.0) and np.all(c[:, :, 0, 0, 1] == 0.0)


def test_cornerpoint_pinch_connections():
    """PINCH: layers 2 and 3 of a column pinched out (thin and inactive) - the cells above and below are connected when the
    gap is within the threshold; a thick inactive cell elsewhere stays a barrier; MULTZ of the pinched cells acts through
    the option ALL"""
    nx, ny, nz = 2, 1, 6
    coord, zcorn = T.cartesian_cornerpoint(nx, ny, nz, 10.0, 10.0, 2.0, top=0.0)
    z = zcorn.reshape(nz, 2, ny, 2, nx, 2).copy()
    # column 0: squeeze layers 2 and 3 to 0.05 m each (everything below moves up)
    col = z[:, :, 0, :, 0, :]
    th = np.array([2.0, 2.0, 0.05, 0.05, 2.0, 2.0])
    tops = np.concatenate([[0.0], np.cumsum(th)[:-1]])
    col[:, 0] = tops[:, None, None]; col[:, 1] = (tops + th)[:, None, None]
    cell = lambda i, k: i + nx * k
    act = np.ones(nx * nz, int)
    act[[cell(0, 2), cell(0, 3), cell(1, 3)]] = 0          # column 1: one full-thickness inactive cell
    g0 = T.cornerpoint_faces(nx, ny, nz, coord, z.reshape(-1), actnum=act)
    g = T.cornerpoint_faces(nx, ny, nz, coord, z.reshape(-1), actnum=act, pinch=0.2)
    comp = {int(cc): q for q, cc in enumerate(g["cart"])}
    p0, p = _by_pair(g0), _by_pair(g)
    new = set(p) - set(p0)
    assert new == {(comp[cell(0, 1)], comp[cell(0, 4)])}     # the 2 m gap of column 1 exceeds the threshold
    q = p[(comp[cell(0, 1)], comp[cell(0, 4)])]
    f = g["faces"]
    assert f["face1"][q] == T.ZP and f["face2"][q] == T.ZM
    np.testing.assert_allclose(f["area_normal"][q], [0.0, 0.0, 100.0])
    np.testing.assert_allclose(f["center1"][q][2], 4.0); np.testing.assert_allclose(f["center2"][q][2], 4.1)
    assert set(p0) <= set(p) and T.cornerpoint_faces(nx, ny, nz, coord, z.reshape(-1), actnum=act, pinch=0.05)["faces"]["cell1"].size == f["cell1"].size - 1
    # option ALL: the smallest MULTZ on the way down, pinched cells included
    perm = np.full((g["n"], 3), 1e-13)
    mz = np.ones(nx * nz); mz[cell(0, 3)] = 0.25; mz[cell(0, 1)] = 0.5
    base = T.face_transmissibilities(f, g["centroid"], perm)
    t = T.face_transmissibilities(f, g["centroid"], perm, mult={"Z+": mz[g["cart"]]}, multz_all=dict(cart=g["cart"], nxny=nx * ny, multz=mz))
    assert t[q] == base[q] * 0.25


def test_thpresft_overrides_the_region_table():
    """THPRESFT (eclgenericthresholdpressure.cc:77-100): a row of 6 cells, regions 0 0 0 1 1 1, fault A names cells 1 and 2,
    fault B names cell 4"""
    H = pkg.thpres
    n = 6
    rowptr = np.concatenate([[0], np.cumsum([2, 3, 3, 3, 3, 2])])
    col = np.array([0, 1, 0, 1, 2, 1, 2, 3, 2, 3, 4, 3, 4, 5, 4, 5])
    eql = np.array([0, 0, 0, 1, 1, 1])
    mat = np.array([[0.0, 7.0e5], [7.0e5, 0.0]])
    plain = H.per_entry(rowptr, col, eql, mat)
    entry = lambda a, b: [q for q in range(rowptr[a], rowptr[a + 1]) if col[q] == b][0]
    assert plain[entry(2, 3)] == plain[entry(3, 2)] == 7.0e5 and np.count_nonzero(plain) == 2
    fault = np.array([-1, 0, 0, -1, 1, -1])
    e = H.per_entry(rowptr, col, eql, mat, fault_of_cell=fault, thpresft=[3.0e5, 9.0e5])
    assert e[entry(0, 1)] == e[entry(1, 0)] == 3.0e5          # fault A against no fault
    assert e[entry(1, 2)] == 0.0                               # inside fault A
    assert e[entry(2, 3)] == e[entry(3, 2)] == 3.0e5          # the fault value replaces the region pair's 7e5
    assert e[entry(3, 4)] == e[entry(4, 5)] == 9.0e5
    assert all(e[entry(a, a)] == 0.0 for a in range(n))
    # two cells on no fault keep the region table; an empty THPRESFT changes nothing
    fault2 = np.array([-1, 0, -1, -1, -1, -1])
    assert H.per_entry(rowptr, col, eql, mat, fault_of_cell=fault2, thpresft=[3.0e5])[entry(2, 3)] == 7.0e5
    assert np.array_equal(H.per_entry(rowptr, col, eql, mat, fault_of_cell=fault, thpresft=[]), plain)
