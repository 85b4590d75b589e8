"""Threshold pressures between equilibration regions (THPRES): the per-entry array `opmhip_set_static` takes.

Host-side restatement of EclThresholdPressure (ebos/eclthresholdpressure.hh:86-167, ebos/eclgenericthresholdpressure.cc:72-203,
SURVEY.md §8 row f4).  The flux kernel applies the value of an entry (I,J) exactly as ebos/eclfluxmodule.hh:323-338 does;
here the values are made:

  thresholdPressure(i, j)     0 inside a region, else thpres[region_i][region_j]                          (.cc:72-108)
  explicit values             a THPRES record (r1, r2, value) sets both orientations; only pairs that meet at a face of
                              the grid are set (applyExplicitThresholdPressures_, .cc:149-203)
  defaulted values            a record without a value takes the largest initial phase potential difference over the
                              faces between the two regions, phases whose upstream cell holds them mobile only, faces
                              with |area * transmissibility| < 1e-18 skipped (computeDefaultThresholdPressures_, .hh:96-163)

The initial intensive quantities (phase pressures, densities, mobilities) come from the device: HipModel.iq() after
set_state - the numbers the flux kernel itself will see.  THPRESFT (fault threshold pressures) is an "experimental"
switch of the reference (.cc:80-98) and not restated.
"""
import numpy as np

F_P, F_MOB, F_RHO = 3, 9, 12   # fields of the intensive-quantity record (csrc/assemble.hip)
GRAVITY = 9.80665


def default_threshold_pressures(eqlnum, nreg, cell1, cell2, trans, area, iq, depth, gravity=GRAVITY):
    """-> (nreg, nreg) matrix of the defaults.  eqlnum 0-based per cell; connections cell1/cell2 with trans and area;
    iq (n, fields, 4): value in [..., 0]; depth (n)."""
    eqlnum = np.asarray(eqlnum, np.int64)
    c1, c2 = np.asarray(cell1, np.int64), np.asarray(cell2, np.int64)
    r1, r2 = eqlnum[c1], eqlnum[c2]
    use = (r1 != r2) & ~(np.abs(np.asarray(area, float) * np.asarray(trans, float)) < 1e-18)
    out = np.zeros((nreg, nreg))
    if not use.any():
        return out
    c1, c2, r1, r2 = c1[use], c2[use], r1[use], r2[use]
    v = np.asarray(iq, float)[..., 0]
    depth = np.asarray(depth, float)
    pth = np.zeros(len(c1))
    for inside, outside in ((c1, c2), (c2, c1)):        # every face is seen from both of its (interior) elements
        for ph in range(3):
            # calculateGradients_ (ebos/eclfluxmodule.hh:250-300): pEx + rhoAvg g (zIn - zEx) - pIn, upstream by its sign
            rho_avg = (v[inside, F_RHO + ph] + v[outside, F_RHO + ph]) / 2.0
            dp = v[outside, F_P + ph] + rho_avg * ((depth[inside] - depth[outside]) * gravity) - v[inside, F_P + ph]
            up = np.where(dp > 0.0, outside, inside)    # dp == 0: either way |dp| = 0 adds nothing
            mobile = v[up, F_MOB + ph] > 0.0
            pth = np.where(mobile, np.maximum(pth, np.abs(dp)), pth)
    np.maximum.at(out, (r1, r2), pth)
    np.maximum.at(out, (r2, r1), pth)
    return out


def threshold_pressure_matrix(nreg, records, eqlnum, cell1, cell2, defaults=None):
    """records: (region1, region2, value | None), regions 1-based as in the deck.  A record makes a barrier; without a value
    the default of the pair applies.  Only pairs that share a face get a value (the reference walks the intersections).
    -> (nreg, nreg)"""
    eqlnum = np.asarray(eqlnum, np.int64)
    touching = np.zeros((nreg, nreg), bool)
    r1, r2 = eqlnum[np.asarray(cell1, np.int64)], eqlnum[np.asarray(cell2, np.int64)]
    touching[r1, r2] = True
    touching[r2, r1] = True
    out = np.zeros((nreg, nreg))
    for a, b, val in records:
        a, b = a - 1, b - 1
        if not (0 <= a < nreg and 0 <= b < nreg):
            raise ValueError("THPRES: region out of range")
        if a == b or not touching[a, b]:
            continue
        if val is None:
            if defaults is None:
                raise ValueError("THPRES record %d %d is defaulted: default_threshold_pressures() needed" % (a + 1, b + 1))
            val = defaults[a, b]
        out[a, b] = out[b, a] = val
    return out


def per_entry(rowptr, col, eqlnum, matrix, fault_of_cell=None, thpresft=None):
    """thresholdPressure(I, J) for every block-CSR entry (0 on the diagonal and inside a region).
    THPRESFT (eclgenericthresholdpressure.cc:77-100, 217-243; behind --enable-experiments in the reference): fault_of_cell
    (per cell: index of the fault whose face list names the cell, the LAST such THPRESFT record winning, -1 = none) and
    thpresft (value per fault, Pa).  Two cells of one fault: 0, "even across EQUIL regions"; cells of different faults, or
    one of them on no fault: the larger of the two faults' values (0 for no fault) - the region table is then not consulted."""
    rowptr, col = np.asarray(rowptr, np.int64), np.asarray(col, np.int64)
    eqlnum = np.asarray(eqlnum, np.int64)
    row = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr))
    ri, rj = eqlnum[row], eqlnum[col]
    out = np.where(ri == rj, 0.0, np.asarray(matrix, float)[ri, rj])
    if thpresft is not None and len(thpresft) > 0:
        fc, val = np.asarray(fault_of_cell, np.int64), np.asarray(thpresft, float)
        fi, fj = fc[row], fc[col]
        vi = np.where(fi >= 0, val[np.maximum(fi, 0)], 0.0)
        vj = np.where(fj >= 0, val[np.maximum(fj, 0)], 0.0)
        out = np.where(fi != fj, np.maximum(vi, vj), np.where(fi >= 0, 0.0, out))
        out = np.where(row == col, 0.0, out)
    return out
