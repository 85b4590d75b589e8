// gfx950 kernels of the assembly half of the hot path: black-oil intensive quantities, TPFA flux and storage with
// dense forward AD (value + 3 derivatives), element-centred linearisation straight into block-CSR, convergence
// reductions and the Newton update with primary-variable switching.
//
// Reference behaviour restated (file:line under the reference tree; opm-models / opm-material routines are NOT in that
// tree and are restated from their public 2021.10 sources):
//   ebos/eclfluxmodule.hh:212-357 (TPFA flux), ebos/eclproblem.hh:1430-1486, 1711-1765, 1823-1845 (problem hooks),
//   flow/BlackoilModelEbos.hpp:549-563 (updateSolution), :628-904 (convergence),
//   BlackOilIntensiveQuantities::update, BlackOilLocalResidual::{computeStorage,computeFlux}, FvBaseLocalResidual::eval,
//   FvBaseAdLocalLinearizer, BlackOilNewtonMethod::update_, BlackOilPrimaryVariables::adaptPrimaryVariables.
//
// Mapping: ONE LANE PER BLOCK-CSR ENTRY.  A 256-thread workgroup owns a tile of whole rows with at most
// 256 entries (36 rows of a 7-point stencil: 252/256 lanes busy).  The lane of an off-diagonal entry (I,J) evaluates the
// face flux twice, once with focus I (goes into R_I and the diagonal block) and once with focus J (its derivative is
// the off-diagonal block (I,J), exactly the number the reference's column-wise linearisation writes there); the lane of
// the diagonal entry evaluates the storage term and then sums its row's fluxes out of LDS in ascending column order.
// All 3x3 blocks of the tile are staged in LDS and leave as one contiguous, 16-byte-per-lane coalesced store stream.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "fluid_tables.hpp"
#include "internal.hpp"

namespace opmhip {

// ============================== dense AD scalar (DenseAd::Evaluation<double,3>) ==============================
struct Ad {
    double v, d0, d1, d2;
};
__device__ __forceinline__ Ad ad_const(double c) { return Ad{c, 0.0, 0.0, 0.0}; }
__device__ __forceinline__ Ad ad_var(double x, int idx) { return Ad{x, idx == 0 ? 1.0 : 0.0, idx == 1 ? 1.0 : 0.0, idx == 2 ? 1.0 : 0.0}; }
__device__ __forceinline__ Ad operator+(const Ad& a, const Ad& b) { return Ad{a.v + b.v, a.d0 + b.d0, a.d1 + b.d1, a.d2 + b.d2}; }
__device__ __forceinline__ Ad operator-(const Ad& a, const Ad& b) { return Ad{a.v - b.v, a.d0 - b.d0, a.d1 - b.d1, a.d2 - b.d2}; }
__device__ __forceinline__ Ad operator+(const Ad& a, double b) { return Ad{a.v + b, a.d0, a.d1, a.d2}; }
__device__ __forceinline__ Ad operator+(double a, const Ad& b) { return Ad{a + b.v, b.d0, b.d1, b.d2}; }
__device__ __forceinline__ Ad operator-(const Ad& a, double b) { return Ad{a.v - b, a.d0, a.d1, a.d2}; }
__device__ __forceinline__ Ad operator-(double a, const Ad& b) { return Ad{a - b.v, -b.d0, -b.d1, -b.d2}; }
__device__ __forceinline__ Ad operator-(const Ad& a) { return Ad{-a.v, -a.d0, -a.d1, -a.d2}; }
__device__ __forceinline__ Ad operator*(const Ad& a, const Ad& b) {
    const double u = a.v, w = b.v;
    return Ad{u * w, a.d0 * w + b.d0 * u, a.d1 * w + b.d1 * u, a.d2 * w + b.d2 * u};
}
__device__ __forceinline__ Ad operator*(const Ad& a, double b) { return Ad{a.v * b, a.d0 * b, a.d1 * b, a.d2 * b}; }
__device__ __forceinline__ Ad operator*(double a, const Ad& b) { return b * a; }
__device__ __forceinline__ Ad operator/(const Ad& a, const Ad& b) {
    const double u = a.v, w = b.v;
    return Ad{u / w, (w * a.d0 - b.d0 * u) / (w * w), (w * a.d1 - b.d1 * u) / (w * w), (w * a.d2 - b.d2 * u) / (w * w)};
}
__device__ __forceinline__ Ad operator/(const Ad& a, double b) { return Ad{a.v / b, a.d0 / b, a.d1 / b, a.d2 / b}; }
__device__ __forceinline__ Ad operator/(double a, const Ad& b) {
    const double t = -a / (b.v * b.v);
    return Ad{a / b.v, t * b.d0, t * b.d1, t * b.d2};
}
// Opm::pow(Evaluation, Scalar) as oracle/eval.hpp restates it.  The device's pow() and the host's std::pow() agree to an ulp,
// not to the bit: VAPPARS is the one place where device and oracle are compared with a tolerance (tests/test_gpu_vappars.py)
__device__ __forceinline__ Ad epow(const Ad& a, double e) {
    const double px = pow(a.v, e);
    const double df = (a.v == 0.0) ? 0.0 : px / a.v * e;
    return Ad{px, df * a.d0, df * a.d1, df * a.d2};
}
__device__ __forceinline__ double epow(double a, double e) { return pow(a, e); }
__device__ __forceinline__ Ad ad_max(const Ad& a, const Ad& b) { return (a.v > b.v) ? a : b; }
__device__ __forceinline__ Ad ad_min(const Ad& a, const Ad& b) { return (a.v < b.v) ? a : b; }
__device__ __forceinline__ double val(const Ad& a) { return a.v; }
__device__ __forceinline__ double val(double a) { return a; }
template <class E> __device__ __forceinline__ E mk(double x, int idx);
template <> __device__ __forceinline__ Ad mk<Ad>(double x, int idx) { return ad_var(x, idx); }
template <> __device__ __forceinline__ double mk<double>(double x, int) { return x; }
template <class E> __device__ __forceinline__ E cst(double x);
template <> __device__ __forceinline__ Ad cst<Ad>(double x) { return ad_const(x); }
template <> __device__ __forceinline__ double cst<double>(double x) { return x; }
__device__ __forceinline__ double emin(double a, double b) { return (a < b) ? a : b; }
__device__ __forceinline__ Ad emin(const Ad& a, const Ad& b) { return ad_min(a, b); }
__device__ __forceinline__ double emax(double a, double b) { return (a > b) ? a : b; }
__device__ __forceinline__ Ad emax(const Ad& a, const Ad& b) { return ad_max(a, b); }

// ============================== property tables ===============================================================
typedef const double* GlobalTab;                                       // table blob where it lies in HBM (L1/K$-resident)
typedef const __attribute__((address_space(3))) double* LdsTab;        // ... or copied into LDS by the workgroup
template <class DP>
struct TablesT {
    DP dbl;
    const int* idx;
    double rock_pref, rock_cr;
    int ndbl, nidx;   // lengths of the blobs (for the LDS copy of the per-cell kernels)
    int rock_desc, num_rock;   // RockTabDesc array inside the int blob (ROCKTAB), 0 tables = none
    __device__ __forceinline__ const RockTabDesc& rock(int t) const { return reinterpret_cast<const RockTabDesc*>(idx + rock_desc)[t]; }
    __device__ __forceinline__ const PvtRegionDesc& pvt(int r) const { return reinterpret_cast<const PvtRegionDesc*>(idx + 2)[r]; }
    __device__ __forceinline__ const SatRegionDesc& sat(int s) const {
        return reinterpret_cast<const SatRegionDesc*>(idx + 2 + idx[0] * (int)(sizeof(PvtRegionDesc) / sizeof(int)))[s];
    }
};
typedef TablesT<GlobalTab> Tables;
// segment of x in an ascending array with clamping (Tabulated1DFunction / UniformXTabulated2DFunction, extrapolate = true):
// a value that sits exactly on an interior node belongs to the segment on its right
template <class DP> __device__ __forceinline__ int seg_right(DP x, int n, double xv) {
    if (xv <= x[0]) return 0;
    if (xv >= x[n - 1]) return n - 2;
    int lo = 0, hi = n - 1;
    while (lo + 1 < hi) {
        const int mid = (lo + hi) >> 1;
        if (x[mid] <= xv) lo = mid; else hi = mid;
    }
    return lo;
}
template <class E, class DP> __device__ __forceinline__ E tab1(DP x, DP y, int n, const E& xv) {
    const int s = seg_right(x, n, val(xv));
    const double x0 = x[s], x1 = x[s + 1], y0 = y[s], y1 = y[s + 1];
    return y0 + (y1 - y0) * (xv - x0) / (x1 - x0);
}
// plain bilinear interpolation in (x, y) with per-column y grids (UniformXTabulated2DFunction; the policy the Norne PVT
// points select): oil (x = Rs, y = p_o)
struct Tab2Desc { int nx, xs, yoff, ys; };
template <class E, class DP> __device__ __forceinline__ E tab2g(const TablesT<DP>& T, const Tab2Desc& G, int voff, const E& xv, const E& yv) {
    const DP xs = T.dbl + G.xs;
    const int* yo = T.idx + G.yoff;
    const int i = seg_right(xs, G.nx, val(xv));
    const E alpha = (xv - xs[i]) / (xs[i + 1] - xs[i]);
    const DP y1 = T.dbl + G.ys + yo[i];
    const DP y2 = T.dbl + G.ys + yo[i + 1];
    const DP v1 = T.dbl + voff + yo[i];
    const DP v2 = T.dbl + voff + yo[i + 1];
    const int j1 = seg_right(y1, yo[i + 1] - yo[i], val(yv)), j2 = seg_right(y2, yo[i + 2] - yo[i + 1], val(yv));
    const E beta1 = (yv - y1[j1]) / (y1[j1 + 1] - y1[j1]);
    const E beta2 = (yv - y2[j2]) / (y2[j2 + 1] - y2[j2]);
    const E s1 = v1[j1] * (1.0 - beta1) + v1[j1 + 1] * beta1;
    const E s2 = v2[j2] * (1.0 - beta2) + v2[j2 + 1] * beta2;
    return s1 * (1.0 - alpha) + s2 * alpha;
}
template <class E, class DP> __device__ __forceinline__ E tab2(const TablesT<DP>& T, const PvtRegionDesc& D, int voff, const E& xv, const E& yv) {
    return tab2g<E, DP>(T, Tab2Desc{D.o_nx, D.o_xs, D.o_yoff, D.o_ys}, voff, xv, yv);
}
// wet gas (x = p_g, y = Rv): interpolation guided by the columns' LAST samples (the saturated Rv), the shift fading linearly
// to 0 at Rv = 0 - the policy the 16-digit saturations of tests/test_equil.cc DeckWithRSVDAndRVVD / DeckWithPBVDAndPDVD
// select (oracle/fluid.hpp Tab2D, guide = 2); same statements, same order
template <class E, class DP> __device__ __forceinline__ E tab2wg(const TablesT<DP>& T, const PvtRegionDesc& D, int voff, const E& xv, const E& yv) {
    const DP xs = T.dbl + D.wg_xs;
    const int* yo = T.idx + D.wg_yoff;
    const int i = seg_right(xs, D.wg_n, val(xv));
    const E alpha = (xv - xs[i]) / (xs[i + 1] - xs[i]);
    const DP y1 = T.dbl + D.wg_ys + yo[i];
    const DP y2 = T.dbl + D.wg_ys + yo[i + 1];
    const DP v1 = T.dbl + voff + yo[i];
    const DP v2 = T.dbl + voff + yo[i + 1];
    const int n1 = yo[i + 1] - yo[i], n2 = yo[i + 2] - yo[i + 1];
    E yLower = yv, yUpper = yv;
    const double e0 = y1[n1 - 1], e1 = y2[n2 - 1];
    const E yEnd = e0 * (1.0 - alpha) + e1 * alpha;
    if (val(yEnd) > 0.0) {
        const E shift = (e1 - e0) * yv / yEnd;
        yLower = yv - alpha * shift;
        yUpper = yv + shift - alpha * shift;
    }
    const int j1 = seg_right(y1, n1, val(yLower)), j2 = seg_right(y2, n2, val(yUpper));
    const E beta1 = (yLower - y1[j1]) / (y1[j1 + 1] - y1[j1]);
    const E beta2 = (yUpper - y2[j2]) / (y2[j2 + 1] - y2[j2]);
    const E s1 = v1[j1] * (1.0 - beta1) + v1[j1 + 1] * beta1;
    const E s2 = v2[j2] * (1.0 - beta2) + v2[j2 + 1] * beta2;
    return s1 * (1.0 - alpha) + s2 * alpha;
}
// PiecewiseLinearTwoPhaseMaterial: constant outside the table; a value on a node belongs to the segment on its left
// water-induced compaction tables (rockCompPoroMultWc_ / rockCompTransMultWc_: UniformXTabulated2DFunction with the same S_w
// nodes under every pressure node, vertical interpolation, extrapolating) - oracle/fluid.hpp Tab2D::eval with guide 0, same
// statements, same order
template <class E> __device__ __forceinline__ E wc_eval(const int* __restrict__ d, const double* __restrict__ data, int voff, const E& xv, const E& yv) {
    const int ns = d[1];
    const double* xs = data + d[2];
    const double* ys = data + d[3];
    const int i = seg_right(xs, d[0], val(xv));
    const E alpha = (xv - xs[i]) / (xs[i + 1] - xs[i]);
    const double* v1 = data + voff + (size_t)i * ns;
    const double* v2 = v1 + ns;
    const int j1 = seg_right(ys, ns, val(yv)), j2 = j1;
    const E beta1 = (yv - ys[j1]) / (ys[j1 + 1] - ys[j1]);
    const E beta2 = (yv - ys[j2]) / (ys[j2 + 1] - ys[j2]);
    const E s1 = v1[j1] * (1.0 - beta1) + v1[j1 + 1] * beta1;
    const E s2 = v2[j2] * (1.0 - beta2) + v2[j2 + 1] * beta2;
    return s1 * (1.0 - alpha) + s2 * alpha;
}
template <class E, class DP> __device__ __forceinline__ E pwlin(DP x, DP y, int n, const E& xv) {
    const double s = val(xv);
    if (s <= x[0]) return cst<E>(y[0]);
    if (s >= x[n - 1]) return cst<E>(y[n - 1]);
    int lo = 0, hi = n - 1;
    while (lo + 1 < hi) {
        const int mid = (lo + hi) >> 1;
        if (x[mid] < s) lo = mid; else hi = mid;
    }
    const double x0 = x[lo], x1 = x[lo + 1], y0 = y[lo], y1 = y[lo + 1];
    const double m = (y1 - y0) / (x1 - x0);
    return y0 + (xv - x0) * m;
}
template <class DP> __device__ __forceinline__ double rs_sat_value(const TablesT<DP>& T, int pr, double po) {
    const PvtRegionDesc& D = T.pvt(pr);
    return tab1<double, DP>(T.dbl + D.sat_p, T.dbl + D.sat_rs, D.sat_n, po);
}
template <class DP> __device__ __forceinline__ double rv_sat_value(const TablesT<DP>& T, int pr, double pg) {
    const PvtRegionDesc& D = T.pvt(pr);
    return tab1<double, DP>(T.dbl + D.wg_xs, T.dbl + D.wgs_rv, D.wg_n, pg);
}
// EclDefaultMaterial capillary pressures: pC[water] = -pcow(Sw), pC[oil] = 0, pC[gas] = pcgo(1 - Swco - Sg)
template <class E, class DP> __device__ __forceinline__ void cap_pressures(const TablesT<DP>& T, int sr, const E& Sw, const E& Sg, E pC[3]) {
    const SatRegionDesc& Sd = T.sat(sr);
    const DP B = T.dbl;
    const double Swco = B[Sd.swco];
    pC[0] = -pwlin<E, DP>(B + Sd.sw_x, B + Sd.pcow, Sd.nw, Sw);
    pC[1] = cst<E>(0.0);
    pC[2] = pwlin<E, DP>(B + Sd.so_x, B + Sd.pcgo, Sd.ng, 1.0 - Swco - Sg);
}

// ============================== intensive quantities ========================================================
// Record of the intensive-quantity cache: fields x (value + 3 derivatives).  Two layouts, chosen per context when the
// fluid is set: the BASE one (live oil + dry gas + water: 17 fields) and the EXTENDED one (19 fields) for decks with wet
// gas (PVTG: Rv) and / or rock compaction tables (ROCKTAB: transmissibility multiplier).  Kernels are templated on it, so
// that the base instantiation computes bit for bit what it did.
// The cache is stored FIELD-MAJOR: field f of cell c = the four doubles at iq[(f * ncell + c) * 4].  Lanes that walk
// consecutive cells (the per-cell kernels) write and read contiguous memory, and the neighbours a tile of k_assemble
// gathers - runs of consecutive cells of a few z-lines - share cache lines: 3.6x fewer L1 accesses than with per-cell
// records, whose 544-byte stride gave every lane of a gather its own lines.
__device__ __forceinline__ const double* iq_at(const double* iq, int ncell, int f, int c) { return iq + ((size_t)f * ncell + c) * 4; }
__device__ __forceinline__ double* iq_at(double* iq, int ncell, int f, int c) { return iq + ((size_t)f * ncell + c) * 4; }
enum { F_S = 0, F_P = 3, F_B = 6, F_MOB = 9, F_RHO = 12, F_RS = 15 };
template <bool EXT> struct Lay {
    static constexpr int F_RV = 16, F_TMULT = 17;                    // EXT only
    static constexpr int F_PORO = EXT ? 18 : 16;
    static constexpr int IQF = EXT ? 19 : 17;                        // fields per cell
    static constexpr int IQS = IQF * 4;                              // doubles per cell
    static constexpr int RQ_F0 = F_P;                                // neighbour fields of the flux: p, 1/B, mobility, density, Rs [, Rv, tmult]
    static constexpr int RQ_NF = (EXT ? F_TMULT : F_RS) - F_P + 1;
};
enum { WATER = 0, OIL = 1, GAS = 2 };
enum { EQ_OIL = 0, EQ_WATER = 1, EQ_GAS = 2 };
constexpr double GRAVITY = 9.80665;

template <class E>
struct Iq {
    E S[3], p[3], invB[3], mob[3], rho[3], Rs, poro;
    E Rv, tmult;   // extended layout only
};
struct CellStatic {
    const double *poro, *volume, *depth, *rsmax;
    const int *pvtnum, *satnum;
    const double *rvmax, *overburden;   // extended layout: DRVDT cap, overburden pressure (may be NULL)
    const int* rocknum;                 // rock-table index per cell (NULL = table 0)
    const double* pcw;                  // extended layout: scaled maximum of pcow per cell (PCW / SWATINIT; NULL = the tables' own)
    const double* minpo;                // extended layout: minimum oil pressure so far (ROCKCOMP IRREVERS; NULL = reversible compaction)
    const double* maxso;                // extended layout: largest oil saturation seen at the start of a time step (VAPPARS; NULL = not in force)
    double vap1, vap2;                  // VAPPARS: exponent on RvSat / on RsSat
    // extended layout: water-induced compaction (ROCK2D / ROCK2DTR / ROCKWNOD; NULL = off): largest S_w seen at the start of a
    // time step, initial S_w, per table {np, nsw, pressure nodes at, S_w nodes at, pore-volume multipliers at, transmissibility
    // multipliers at (-1: none)} and the tables' doubles (global memory: a handful of nodes, used by few decks)
    const double *maxsw, *sw0;
    const int* wcdesc;
    const double* wcdata;
    const double* eps;                  // extended layout: scaled end points per cell, field-major [EPS_COUNT][ncell] (NULL = no end-point scaling)
    int epscfg;                         // EclEpsConfig: bit 0 saturation scaling, 1 three-point, 2-3 krw, 4-5 kro, 6-7 krg mode, 8 pcw, 9 pcg
    // extended layout: relative-permeability hysteresis (SATOPTS HYSTER; NULL = not in force): per cell the turning point and the
    // imbibition curve's shift of the oil-water and of the gas-oil system, field-major [4][ncell] (krnSwMdc_ow, deltaSwImbKrn_ow,
    // krnSwMdc_go, deltaSwImbKrn_go), the imbibition saturation region (IMBNUM) and, with end-point scaling in force, the
    // scaled end points of the imbibition curves [EPS_COUNT][ncell] (NULL: the imbibition tables' own); hystModel = EHYSTR item 2
    const double* hyst;
    const int* imbnum;
    const double* epsImb;
    int hystModel;
    double* invb;                       // packed 1/b_w, 1/b_o, 1/b_g per cell, written beside the record (convergence check)
    int ncell;                          // cells of the intensive-quantity cache (owned + ghost): the stride between its fields
};

// ---- saturation end-point scaling (EclEpsTwoPhaseLaw; oracle/fluid.hpp eps_*: same statements, same order) ----------------
struct EpsTriple { double s[3]; };
__device__ __forceinline__ EpsTriple eps_pc_ow(const double* e) { return {{e[EPS_SWL], e[EPS_SWU], e[EPS_SWU]}}; }
__device__ __forceinline__ EpsTriple eps_krw_ow(const double* e) { return {{e[EPS_SWCR], 1.0 - e[EPS_SOWCR] - e[EPS_SGL], e[EPS_SWU]}}; }
__device__ __forceinline__ EpsTriple eps_krn_ow(const double* e) { return {{e[EPS_SWL] + e[EPS_SGL], e[EPS_SWCR] + e[EPS_SGL], 1.0 - e[EPS_SOWCR]}}; }
__device__ __forceinline__ EpsTriple eps_pc_go(const double* e) { return {{1.0 - e[EPS_SWL] - e[EPS_SGU], 1.0 - e[EPS_SWL] - e[EPS_SGL], 1.0 - e[EPS_SWL] - e[EPS_SGL]}}; }
__device__ __forceinline__ EpsTriple eps_krw_go(const double* e) { return {{e[EPS_SOGCR], 1.0 - e[EPS_SGCR] - e[EPS_SWL], 1.0 - e[EPS_SWL] - e[EPS_SGL]}}; }
__device__ __forceinline__ EpsTriple eps_krn_go(const double* e) { return {{1.0 - e[EPS_SWL] - e[EPS_SGU], e[EPS_SOGCR], 1.0 - e[EPS_SWL] - e[EPS_SGCR]}}; }
template <class E> __device__ __forceinline__ E eps_sat_two_point(const E& S, const EpsTriple& u, const EpsTriple& sc) {
    return u.s[0] + (S - sc.s[0]) * ((u.s[2] - u.s[0]) / (sc.s[2] - sc.s[0]));
}
template <class E> __device__ __forceinline__ E eps_sat_three_point(const E& S, const EpsTriple& u, const EpsTriple& sc) {
    if (val(S) <= sc.s[0]) return cst<E>(u.s[0]);
    if (val(S) <= sc.s[1]) return u.s[0] + (S - sc.s[0]) * ((u.s[1] - u.s[0]) / (sc.s[1] - sc.s[0]));
    if (u.s[1] == u.s[2]) return cst<E>(u.s[1]);
    if (val(S) <= sc.s[2]) return u.s[1] + (S - sc.s[1]) * ((u.s[2] - u.s[1]) / (sc.s[2] - sc.s[1]));
    return cst<E>(u.s[2]);
}
template <class E> __device__ __forceinline__ E eps_to_unscaled(int cfg, const E& S, const EpsTriple& u, const EpsTriple& sc) {
    if (!(cfg & 1)) return S;
    return (cfg & 2) ? eps_sat_three_point(S, u, sc) : eps_sat_two_point(S, u, sc);
}
template <class E> __device__ __forceinline__ E eps_vertical_krw(int mode, const E& S, const E& kr, const EpsTriple& sc, double fdisp, double fmax, double fr, double fm) {
    if (mode == 0) return kr;
    if (mode == 1) return kr * (fm / fmax);
    const double sm = sc.s[2], sr = emin(sc.s[1], sm);
    if (!(val(S) > sr)) return kr * (fr / fdisp);
    if (fmax > fdisp) { const E t = (kr - fdisp) / (fmax - fdisp); return fr + t * (fm - fr); }
    if (sr < sm) { const E t = (S - sr) / (sm - sr); return fr + t * (fm - fr); }
    return cst<E>(fm);
}
template <class E> __device__ __forceinline__ E eps_vertical_krn(int mode, const E& S, const E& kr, const EpsTriple& sc, double fdisp, double fmax, double fr, double fm) {
    if (mode == 0) return kr;
    if (mode == 1) return kr * (fm / fmax);
    const double sl = sc.s[0], sr = emax(sc.s[1], sl);
    if (!(val(S) < sr)) return kr * (fr / fdisp);
    if (fmax > fdisp) { const E t = (kr - fdisp) / (fmax - fdisp); return fr + t * (fm - fr); }
    if (sr > sl) { const E t = (sr - S) / (sr - sl); return fr + t * (fm - fr); }
    return cst<E>(fm);
}
// the cell's scaled end points out of the field-major array, the region's unscaled ones out of the table blob
template <class DP> __device__ __forceinline__ void eps_load(const CellStatic& C, int c, DP B, const SatRegionDesc& Sd, double* u, double* s) {
#pragma unroll
    for (int f = 0; f < EPS_COUNT; ++f) { u[f] = B[Sd.eps + f]; s[f] = C.eps[(size_t)f * C.ncell + c]; }
}
// capillary pressures of a cell with scaled end points (SatFunc::capillaryPressuresEps)
template <class E, class DP> __device__ __forceinline__ void cap_pressures_eps(DP B, const SatRegionDesc& Sd, int cfg, const double* u, const double* s, const E& Sw, const E& Sg, E pC[3]) {
    const double Swco = B[Sd.swco];
    const double SwcoS = s[EPS_SWL];
    const E SoP = 1.0 - SwcoS - Sg;
    E swU = Sw, soU = SoP;
    if (cfg & 1) {
        swU = eps_sat_two_point(Sw, eps_pc_ow(u), eps_pc_ow(s));
        soU = eps_sat_two_point(SoP, eps_pc_go(u), eps_pc_go(s));
    } else soU = 1.0 - Swco - Sg;
    E pcw = pwlin<E, DP>(B + Sd.sw_x, B + Sd.pcow, Sd.nw, swU), pcg = pwlin<E, DP>(B + Sd.so_x, B + Sd.pcgo, Sd.ng, soU);
    if (cfg & 256) {
        const double sm = s[EPS_MAXPCOW], um = u[EPS_MAXPCOW];
        pcw = pcw * ((sm == um) ? 1.0 : sm / um);
    }
    if (cfg & 512) {
        const double sm = s[EPS_MAXPCGO], um = u[EPS_MAXPCGO];
        pcg = pcg * ((sm == um) ? 1.0 : sm / um);
    }
    pC[0] = -pcw;
    pC[1] = cst<E>(0.0);
    pC[2] = pcg;
}
// relative permeabilities of a cell with scaled end points (SatFunc::relativePermeabilitiesEps)
template <class E, class DP> __device__ __forceinline__ void rel_perms_eps(DP B, const SatRegionDesc& Sd, int cfg, const double* u, const double* s, const E& SwIn, const E& Sg, E kr[3]) {
    const double Swco = B[Sd.swco];
    const double SwcoS = (cfg & 1) ? s[EPS_SWL] : Swco;
    const int mKrw = (cfg >> 2) & 3, mKro = (cfg >> 4) & 3, mKrg = (cfg >> 6) & 3;
    const EpsTriple uKrwOw = eps_krw_ow(u), sKrwOw = eps_krw_ow(s), uKrnOw = eps_krn_ow(u), sKrnOw = eps_krn_ow(s);
    const EpsTriple uKrwGo = eps_krw_go(u), sKrwGo = eps_krw_go(s), uKrnGo = eps_krn_go(u), sKrnGo = eps_krn_go(s);
    kr[0] = eps_vertical_krw(mKrw, SwIn, pwlin<E, DP>(B + Sd.sw_x, B + Sd.krw, Sd.nw, eps_to_unscaled(cfg, SwIn, uKrwOw, sKrwOw)), sKrwOw, u[EPS_KRWR], u[EPS_MAXKRW], s[EPS_KRWR], s[EPS_MAXKRW]);
    const E SoP = 1.0 - SwcoS - Sg;
    kr[2] = eps_vertical_krn(mKrg, SoP, pwlin<E, DP>(B + Sd.so_x, B + Sd.krg, Sd.ng, eps_to_unscaled(cfg, SoP, uKrnGo, sKrnGo)), sKrnGo, u[EPS_KRGR], u[EPS_MAXKRG], s[EPS_KRGR], s[EPS_MAXKRG]);
    const E Sw = emax(cst<E>(SwcoS), SwIn);
    const E Sw_ow = Sg + Sw;
    const E So_go = 1.0 - Sw_ow;
    const E kro_ow = eps_vertical_krn(mKro, Sw_ow, pwlin<E, DP>(B + Sd.sw_x, B + Sd.krow, Sd.nw, eps_to_unscaled(cfg, Sw_ow, uKrnOw, sKrnOw)), sKrnOw, u[EPS_KRORW], u[EPS_MAXKROW], s[EPS_KRORW], s[EPS_MAXKROW]);
    const E kro_go = eps_vertical_krw(mKro, So_go, pwlin<E, DP>(B + Sd.so_x, B + Sd.krog, Sd.ng, eps_to_unscaled(cfg, So_go, uKrwGo, sKrwGo)), sKrwGo, u[EPS_KRORG], u[EPS_MAXKROG], s[EPS_KRORG], s[EPS_MAXKROG]);
    const double eps = 1e-5;
    if (val(Sw_ow) - SwcoS < eps) {
        const E kro2 = (kro_ow + kro_go) / 2.0;
        if (val(Sw_ow) - SwcoS > eps / 2.0) {
            const E kro1 = (Sg * kro_go + (Sw - SwcoS) * kro_ow) / (Sw_ow - SwcoS);
            const E alpha = (eps - (Sw_ow - SwcoS)) / (eps / 2.0);
            kr[1] = kro2 * alpha + kro1 * (1.0 - alpha);
        } else kr[1] = kro2;
    } else kr[1] = (Sg * kro_go + (Sw - SwcoS) * kro_ow) / (Sw_ow - SwcoS);
}
// ---- relative-permeability hysteresis (oracle/fluid.hpp: SatFunc::curve / curveInv / hystSee / relativePermeabilitiesHyst - same
//      statements, same order; EclHysteresisTwoPhaseLaw of opm-material restated, UNVERIFIED vs upstream) -------------------------
enum { KRW_OW = 0, KRN_OW = 1, KRW_GO = 2, KRN_GO = 3 };
// one relative-permeability curve of a saturation region at the (scaled) wetting saturation S of its two-phase system
template <class E, class DP> __device__ __forceinline__ E sat_curve(DP B, const SatRegionDesc& Sd, int kind, const E& S, bool scaled, int cfg, const double* u, const double* s) {
    const DP x = B + (kind <= KRN_OW ? Sd.sw_x : Sd.so_x);
    const DP y = B + (kind == KRW_OW ? Sd.krw : kind == KRN_OW ? Sd.krow : kind == KRW_GO ? Sd.krog : Sd.krg);
    const int n = kind <= KRN_OW ? Sd.nw : Sd.ng;
    if (!scaled) return pwlin<E, DP>(x, y, n, S);
    const EpsTriple ut = kind == KRW_OW ? eps_krw_ow(u) : kind == KRN_OW ? eps_krn_ow(u) : kind == KRW_GO ? eps_krw_go(u) : eps_krn_go(u);
    const EpsTriple st = kind == KRW_OW ? eps_krw_ow(s) : kind == KRN_OW ? eps_krn_ow(s) : kind == KRW_GO ? eps_krw_go(s) : eps_krn_go(s);
    const E k = pwlin<E, DP>(x, y, n, eps_to_unscaled(cfg, S, ut, st));
    if (kind == KRW_OW) return eps_vertical_krw((cfg >> 2) & 3, S, k, st, u[EPS_KRWR], u[EPS_MAXKRW], s[EPS_KRWR], s[EPS_MAXKRW]);
    if (kind == KRN_OW) return eps_vertical_krn((cfg >> 4) & 3, S, k, st, u[EPS_KRORW], u[EPS_MAXKROW], s[EPS_KRORW], s[EPS_MAXKROW]);
    if (kind == KRW_GO) return eps_vertical_krw((cfg >> 4) & 3, S, k, st, u[EPS_KRORG], u[EPS_MAXKROG], s[EPS_KRORG], s[EPS_MAXKROG]);
    return eps_vertical_krn((cfg >> 6) & 3, S, k, st, u[EPS_KRGR], u[EPS_MAXKRG], s[EPS_KRGR], s[EPS_MAXKRG]);
}
// PwLin::inv: the abscissa at which a piecewise-linear curve takes the value yv
template <class DP> __device__ __forceinline__ double pwlin_inv(DP x, DP y, int n, double yv) {
    if (y[0] > y[n - 1]) {
        if (yv >= y[0]) return x[0];
        if (yv <= y[n - 1]) return x[n - 1];
        int lo = 0, hi = n - 1;
        while (lo + 1 < hi) {
            const int mid = (lo + hi) / 2;
            if (y[mid] >= yv) lo = mid; else hi = mid;
        }
        const double m = (x[lo + 1] - x[lo]) / (y[lo + 1] - y[lo]);
        return x[lo] + (yv - y[lo]) * m;
    }
    if (yv <= y[0]) return x[0];
    if (yv >= y[n - 1]) return x[n - 1];
    int lo = 0, hi = n - 1;
    while (lo + 1 < hi) {
        const int mid = (lo + hi) / 2;
        if (y[mid] <= yv) lo = mid; else hi = mid;
    }
    const double m = (x[lo + 1] - x[lo]) / (y[lo + 1] - y[lo]);
    return x[lo] + (yv - y[lo]) * m;
}
// SatFunc::curveInv: the (scaled) wetting saturation at which non-wetting curve `kind` takes the value k
template <class DP> __device__ __forceinline__ double sat_curve_inv(DP B, const SatRegionDesc& Sd, int kind, double k, bool scaled, int cfg, const double* u, const double* s) {
    const double Su = kind == KRN_OW ? pwlin_inv<DP>(B + Sd.sw_x, B + Sd.krow, Sd.nw, k) : pwlin_inv<DP>(B + Sd.so_x, B + Sd.krg, Sd.ng, k);
    if (!scaled || !(cfg & 1)) return Su;
    const EpsTriple ut = kind == KRN_OW ? eps_krn_ow(u) : eps_krn_go(u);
    const EpsTriple st = kind == KRN_OW ? eps_krn_ow(s) : eps_krn_go(s);
    if (cfg & 2) {   // eps_unscaled_to_scaled_three_point
        if (Su <= ut.s[0]) return st.s[0];
        if (Su < ut.s[1]) return st.s[0] + (Su - ut.s[0]) * ((st.s[1] - st.s[0]) / (ut.s[1] - ut.s[0]));
        if (Su < ut.s[2]) return st.s[1] + (Su - ut.s[1]) * ((st.s[2] - st.s[1]) / (ut.s[2] - ut.s[1]));
        return st.s[2];
    }
    return st.s[0] + (Su - ut.s[0]) * ((st.s[2] - st.s[0]) / (ut.s[2] - ut.s[0]));
}
// the imbibition region's end points of cell c: the tables' own and the scaled ones (IS* arrays, else the tables' own)
template <class DP> __device__ __forceinline__ void eps_load_imb(const CellStatic& C, int c, DP B, const SatRegionDesc& Si, double* u, double* s) {
#pragma unroll
    for (int f = 0; f < EPS_COUNT; ++f) { u[f] = B[Si.eps + f]; s[f] = C.epsImb ? C.epsImb[(size_t)f * C.ncell + c] : u[f]; }
}
// SatFunc::relativePermeabilitiesHyst
template <class E, class DP> __device__ __forceinline__ void rel_perms_hyst(const TablesT<DP>& T, const CellStatic& C, int c, const SatRegionDesc& Sd, bool scaled,
                                                                             const double* uD, const double* sD, const E& SwIn, const E& Sg, E kr[3]) {
    const DP B = T.dbl;
    const SatRegionDesc& Si = T.sat(C.imbnum[c]);
    const int cfg = C.epscfg;
    double uI[EPS_COUNT], sI[EPS_COUNT];
    if (scaled) eps_load_imb<DP>(C, c, B, Si, uI, sI);
    const double Swco = B[Sd.swco];
    const double SwcoS = (scaled && (cfg & 1)) ? sD[EPS_SWL] : Swco;
    const size_t N = C.ncell;
    const double mdcOw = C.hyst[c], dOw = C.hyst[N + c], mdcGo = C.hyst[2 * N + c], dGo = C.hyst[3 * N + c];
    const bool wetImb = C.hystModel == 1;
    kr[0] = wetImb ? sat_curve<E, DP>(B, Si, KRW_OW, SwIn, scaled, cfg, uI, sI) : sat_curve<E, DP>(B, Sd, KRW_OW, SwIn, scaled, cfg, uD, sD);
    const E SoP = 1.0 - SwcoS - Sg;
    if (val(SoP) <= mdcGo) kr[2] = sat_curve<E, DP>(B, Sd, KRN_GO, SoP, scaled, cfg, uD, sD);
    else kr[2] = sat_curve<E, DP>(B, Si, KRN_GO, SoP + dGo, scaled, cfg, uI, sI);
    const E Sw = emax(cst<E>(SwcoS), SwIn);
    const E Sw_ow = Sg + Sw;
    const E So_go = 1.0 - Sw_ow;
    E kro_ow;
    if (val(Sw_ow) <= mdcOw) kro_ow = sat_curve<E, DP>(B, Sd, KRN_OW, Sw_ow, scaled, cfg, uD, sD);
    else kro_ow = sat_curve<E, DP>(B, Si, KRN_OW, Sw_ow + dOw, scaled, cfg, uI, sI);
    const E kro_go = wetImb ? sat_curve<E, DP>(B, Si, KRW_GO, So_go, scaled, cfg, uI, sI) : sat_curve<E, DP>(B, Sd, KRW_GO, So_go, scaled, cfg, uD, sD);
    const double eps = 1e-5;
    if (val(Sw_ow) - SwcoS < eps) {
        const E kro2 = (kro_ow + kro_go) / 2.0;
        if (val(Sw_ow) - SwcoS > eps / 2.0) {
            const E kro1 = (Sg * kro_go + (Sw - SwcoS) * kro_ow) / (Sw_ow - SwcoS);
            const E alpha = (eps - (Sw_ow - SwcoS)) / (eps / 2.0);
            kr[1] = kro2 * alpha + kro1 * (1.0 - alpha);
        } else kr[1] = kro2;
    } else kr[1] = (Sg * kro_go + (Sw - SwcoS) * kro_ow) / (Sw_ow - SwcoS);
}
// the capillary pressures of cell c at (Sw, Sg), values only: what the primary-variable switches need
// (computeCapillaryPressures_ of BlackOilPrimaryVariables), with the cell's scaled end points where the deck has them
template <class DP, bool EXT> __device__ __forceinline__ void cell_cap_pressures(const TablesT<DP>& T, const CellStatic& C, int c, int sr, double Sw, double Sg, double pC[3]) {
    if (EXT && C.eps) {
        double u[EPS_COUNT], s[EPS_COUNT];
        eps_load<DP>(C, c, T.dbl, T.sat(sr), u, s);
        cap_pressures_eps<double, DP>(T.dbl, T.sat(sr), C.epscfg, u, s, Sw, Sg, pC);
    } else cap_pressures<double, DP>(T, sr, Sw, Sg, pC);
}

// VAPPARS factor on a saturated Rs / Rv (oracle/blackoil.hpp vappars_factor: same statements)
template <class E> __device__ __forceinline__ E vappars_factor(const E& So, const E& SoMaxIn, double vapPar) {
    const E maxOilSaturation = emin(SoMaxIn, cst<E>(1.0));
    if (vapPar > 0.0 && val(maxOilSaturation) > 0.01 && val(So) < val(maxOilSaturation)) {
        const E S = emax(So, cst<E>(0.001));
        return emax(cst<E>(1e-3), epow(S / maxOilSaturation, vapPar));
    }
    return cst<E>(1.0);
}

// BlackOilIntensiveQuantities::update: live oil + water + dry gas (base) or wet gas / rock compaction tables (EXT)
template <class E, class DP, bool EXT>
__device__ __forceinline__ void update_iq(const TablesT<DP>& T, const CellStatic& C, int c, const double* pv, int meaning, Iq<E>& q) {
    const int pr = C.pvtnum ? C.pvtnum[c] : 0, sr = C.satnum ? C.satnum[c] : 0;
    const double RsMax = C.rsmax ? C.rsmax[c] : DBL_MAX / 2.0;
    const double refPoro = C.poro[c];
    const PvtRegionDesc& D = T.pvt(pr);
    const SatRegionDesc& Sd = T.sat(sr);
    const DP B = T.dbl;
    const bool wet = EXT && D.wg_n > 0;   // FluidSystem::enableVaporizedOil()
    const double Swco = B[Sd.swco];
    const E Sw = mk<E>(pv[0], 0);
    E Sg = cst<E>(0.0);
    if (meaning == OPMHIP_SW_PO_SG) Sg = mk<E>(pv[2], 2);
    else if (EXT && meaning == OPMHIP_SW_PG_RV) Sg = 1.0 - Sw;   // the oil phase is absent
    const E So = 1.0 - Sw - Sg;
    q.S[WATER] = Sw; q.S[GAS] = Sg; q.S[OIL] = So;
    // capillary pressures (EclDefaultMaterial): pC[water] = -pcow(Sw), pC[oil] = 0, pC[gas] = pcgo(1 - Swco - Sg)
    E pC[3];
    const bool scaled = EXT && C.eps;
    double epsU[EXT ? EPS_COUNT : 1], epsS[EXT ? EPS_COUNT : 1];
    if (scaled) {
        eps_load<DP>(C, c, B, Sd, epsU, epsS);
        cap_pressures_eps<E, DP>(B, Sd, C.epscfg, epsU, epsS, Sw, Sg, pC);
    } else {
        pC[0] = -pwlin<E, DP>(B + Sd.sw_x, B + Sd.pcow, Sd.nw, Sw);
        pC[1] = cst<E>(0.0);
        pC[2] = pwlin<E, DP>(B + Sd.so_x, B + Sd.pcgo, Sd.ng, 1.0 - Swco - Sg);
    }
    if (EXT && C.pcw && !scaled) {
        // end-point scaling of the oil-water curve (EclEpsTwoPhaseLaw with enablePcScaling): table value x (scaled max / table max),
        // the table's maximum being its value at the connate saturation (first row of SWOF)
        const double scaledMax = C.pcw[c], tableMax = B[Sd.pcow];
        const double alpha = (scaledMax == tableMax) ? 1.0 : scaledMax / tableMax;
        pC[0] = pC[0] * alpha;
    }
    if (EXT && meaning == OPMHIP_SW_PG_RV) {   // the pressure primary variable is the GAS pressure
        const E pg = mk<E>(pv[1], 1);
        for (int ph = 0; ph < 3; ++ph) q.p[ph] = pg + (pC[ph] - pC[GAS]);
    } else {
        const E po = mk<E>(pv[1], 1);
        for (int ph = 0; ph < 3; ++ph) q.p[ph] = po + (pC[ph] - pC[OIL]);
    }
    // relative permeabilities (stored in mob, divided by viscosity below)
    if (EXT && C.hyst) rel_perms_hyst<E, DP>(T, C, c, Sd, scaled, epsU, epsS, Sw, Sg, q.mob);
    else if (scaled) rel_perms_eps<E, DP>(B, Sd, C.epscfg, epsU, epsS, Sw, Sg, q.mob);
    else {
        q.mob[WATER] = pwlin<E, DP>(B + Sd.sw_x, B + Sd.krw, Sd.nw, Sw);
        q.mob[GAS] = pwlin<E, DP>(B + Sd.so_x, B + Sd.krg, Sd.ng, 1.0 - Swco - Sg);
        const E Swm = emax(cst<E>(Swco), Sw);
        const E Sw_ow = Sg + Swm;
        const E So_go = 1.0 - Sw_ow;
        const E kro_ow = pwlin<E, DP>(B + Sd.sw_x, B + Sd.krow, Sd.nw, Sw_ow);
        const E kro_go = pwlin<E, DP>(B + Sd.so_x, B + Sd.krog, Sd.ng, So_go);
        const double eps = 1e-5;
        if (val(Sw_ow) - Swco < eps) {
            const E kro2 = (kro_ow + kro_go) / 2.0;
            if (val(Sw_ow) - Swco > eps / 2.0) {
                const E kro1 = (Sg * kro_go + (Swm - Swco) * kro_ow) / (Sw_ow - Swco);
                const E alpha = (eps - (Sw_ow - Swco)) / (eps / 2.0);
                q.mob[OIL] = kro2 * alpha + kro1 * (1.0 - alpha);
            } else q.mob[OIL] = kro2;
        } else q.mob[OIL] = (Sg * kro_go + (Swm - Swco) * kro_ow) / (Sw_ow - Swco);
    }
    // Rs / Rv by the meaning of the switching variable, capped by RsMax / RvMax (DRSDT / DRVDT, eclproblem.hh:1711-1754)
    if (EXT) q.Rv = cst<E>(0.0);
    const bool vap = EXT && C.maxso;   // VAPPARS: SoMax = max(So, problem.maxOilSaturation)
    E SoMax = So;
    if (vap) SoMax = emax(So, cst<E>(C.maxso[c]));
    if (meaning == OPMHIP_SW_PO_SG || (EXT && meaning == OPMHIP_SW_PG_RV)) {
        // (Sw_pg_Rv: the oil phase is not present, its "composition" is still needed for the gravity term)
        E RsSat = tab1<E, DP>(B + D.sat_p, B + D.sat_rs, D.sat_n, q.p[OIL]);
        if (vap) RsSat = RsSat * vappars_factor(So, SoMax, C.vap2);
        q.Rs = emin(cst<E>(RsMax), RsSat);
    } else {
        q.Rs = emin(cst<E>(RsMax), mk<E>(pv[2], 2));
    }
    if (EXT && wet) {
        const double RvMax = C.rvmax ? C.rvmax[c] : DBL_MAX / 2.0;
        if (meaning == OPMHIP_SW_PG_RV) q.Rv = emin(cst<E>(RvMax), mk<E>(pv[2], 2));
        else {
            E RvSat = tab1<E, DP>(B + D.wg_xs, B + D.wgs_rv, D.wg_n, q.p[GAS]);
            if (vap) RvSat = RvSat * vappars_factor(So, SoMax, C.vap1);
            q.Rv = emin(cst<E>(RvMax), RvSat);
        }
    }
    // 1/B and viscosity per phase, each at its own phase pressure (BlackOilFluidSystem)
    {
        const bool saturated = val(q.S[GAS]) > 0.0 && val(q.Rs) >= (1.0 - 1e-10) * rs_sat_value(T, pr, val(q.p[OIL]));
        const DP W = B + D.water;  // p_ref, Bw_ref, c_w, mu_ref, c_v
        const E X = W[2] * (q.p[WATER] - W[0]);
        q.invB[WATER] = (1.0 + X * (1.0 + X / 2.0)) / W[1];
        const E Y = (W[2] - W[4]) * (q.p[WATER] - W[0]);
        E mu = (W[3] * W[1]) * q.invB[WATER] / (1.0 + Y * (1.0 + Y / 2.0));
        q.mob[WATER] = q.mob[WATER] / mu;
        if (saturated) {
            q.invB[OIL] = tab1<E, DP>(B + D.sat_p, B + D.sat_invB, D.sat_n, q.p[OIL]);
            mu = tab1<E, DP>(B + D.sat_p, B + D.sat_invB, D.sat_n, q.p[OIL]) / tab1<E, DP>(B + D.sat_p, B + D.sat_invBMu, D.sat_n, q.p[OIL]);
        } else {
            q.invB[OIL] = tab2<E, DP>(T, D, D.o_invB, q.Rs, q.p[OIL]);
            mu = tab2<E, DP>(T, D, D.o_invB, q.Rs, q.p[OIL]) / tab2<E, DP>(T, D, D.o_invBMu, q.Rs, q.p[OIL]);
        }
        q.mob[OIL] = q.mob[OIL] / mu;
        if (EXT && wet) {
            const bool gasSaturated = val(q.S[OIL]) > 0.0 && val(q.Rv) >= (1.0 - 1e-10) * rv_sat_value(T, pr, val(q.p[GAS]));
            if (gasSaturated) {
                q.invB[GAS] = tab1<E, DP>(B + D.wg_xs, B + D.wgs_invB, D.wg_n, q.p[GAS]);
                mu = tab1<E, DP>(B + D.wg_xs, B + D.wgs_invB, D.wg_n, q.p[GAS]) / tab1<E, DP>(B + D.wg_xs, B + D.wgs_invBMu, D.wg_n, q.p[GAS]);
            } else {
                q.invB[GAS] = tab2wg<E, DP>(T, D, D.wg_invB, q.p[GAS], q.Rv);
                mu = tab2wg<E, DP>(T, D, D.wg_invB, q.p[GAS], q.Rv) / tab2wg<E, DP>(T, D, D.wg_invBMu, q.p[GAS], q.Rv);
            }
        } else {
            q.invB[GAS] = tab1<E, DP>(B + D.gas_p, B + D.gas_invB, D.gas_n, q.p[GAS]);
            mu = tab1<E, DP>(B + D.gas_p, B + D.gas_invB, D.gas_n, q.p[GAS]) / tab1<E, DP>(B + D.gas_p, B + D.gas_invBMu, D.gas_n, q.p[GAS]);
        }
        q.mob[GAS] = q.mob[GAS] / mu;
    }
    const DP rr = B + D.density;  // oil, water, gas
    q.rho[WATER] = q.invB[WATER] * rr[1];
    q.rho[GAS] = q.invB[GAS] * rr[2];
    if (EXT && wet) q.rho[GAS] = q.rho[GAS] + q.invB[GAS] * q.Rv * rr[0];   // vaporised oil
    q.rho[OIL] = q.invB[OIL] * rr[0];
    q.rho[OIL] = q.rho[OIL] + q.invB[OIL] * q.Rs * rr[2];
    q.poro = cst<E>(refPoro);
    if (T.rock_cr > 0.0) {
        const E x = T.rock_cr * (q.p[OIL] - T.rock_pref);
        q.poro = q.poro * (1.0 + x + 0.5 * x * x);
    }
    if (EXT) {
        // rock compaction tables: rockCompPoroMultiplier / rockCompTransMultiplier (ebos/eclproblem.hh:1936-2007), reversible form
        q.tmult = cst<E>(1.0);
        if (T.num_rock > 0) {
            const RockTabDesc& R = T.rock(C.rocknum ? C.rocknum[c] : 0);
            E effectiveOilPressure = q.p[OIL];
            if (C.minpo) effectiveOilPressure = emin(q.p[OIL], cst<E>(C.minpo[c]));   // the pore space change is irreversible (eclproblem.hh:1948-1952)
            if (C.overburden) effectiveOilPressure = effectiveOilPressure - C.overburden[c];
            q.poro = q.poro * tab1<E, DP>(B + R.p, B + R.poroMult, R.n, effectiveOilPressure);
            q.tmult = tab1<E, DP>(B + R.p, B + R.transMult, R.n, effectiveOilPressure);
        } else if (C.wcdesc) {   // water compaction (eclproblem.hh:1962-1967, 2001-2005)
            const int* d = C.wcdesc + 6 * (C.rocknum ? C.rocknum[c] : 0);
            E effectiveOilPressure = q.p[OIL];
            if (C.minpo) effectiveOilPressure = emin(q.p[OIL], cst<E>(C.minpo[c]));
            if (C.overburden) effectiveOilPressure = effectiveOilPressure - C.overburden[c];
            const E SwMax = emax(q.S[WATER], cst<E>(C.maxsw[c]));
            const E SwDeltaMax = SwMax - C.sw0[c];
            q.poro = q.poro * wc_eval<E>(d, C.wcdata, d[4], effectiveOilPressure, SwDeltaMax);
            if (d[5] >= 0) q.tmult = wc_eval<E>(d, C.wcdata, d[5], effectiveOilPressure, SwDeltaMax);
        }
    }
}

__device__ __forceinline__ void store_ad(double* o, const Ad& a) {
    *reinterpret_cast<double2*>(o) = make_double2(a.v, a.d0);
    *reinterpret_cast<double2*>(o + 2) = make_double2(a.d1, a.d2);
}
__device__ __forceinline__ Ad load_ad(const double* o) {
    const double2 a = *reinterpret_cast<const double2*>(o), b = *reinterpret_cast<const double2*>(o + 2);
    return Ad{a.x, a.y, b.x, b.y};
}
// the convergence check needs 1/b of the three phases and nothing else of the record (flow/BlackoilModelEbos.hpp:650-664):
// a packed copy (24 B per cell) spares it a 544-byte-strided walk through the cache (PMC: 232 MB of traffic for 56 MB of data)
__device__ __forceinline__ void store_invb(double* invb, int c, const Iq<Ad>& q) {
    invb[(size_t)c * 3] = q.invB[0].v; invb[(size_t)c * 3 + 1] = q.invB[1].v; invb[(size_t)c * 3 + 2] = q.invB[2].v;
}
template <bool EXT>
__device__ __forceinline__ void store_iq(double* iq, int ncell, int c, const Iq<Ad>& q) {
    for (int k = 0; k < 3; ++k) {
        store_ad(iq_at(iq, ncell, F_S + k, c), q.S[k]); store_ad(iq_at(iq, ncell, F_P + k, c), q.p[k]); store_ad(iq_at(iq, ncell, F_B + k, c), q.invB[k]);
        store_ad(iq_at(iq, ncell, F_MOB + k, c), q.mob[k]); store_ad(iq_at(iq, ncell, F_RHO + k, c), q.rho[k]);
    }
    store_ad(iq_at(iq, ncell, F_RS, c), q.Rs);
    if (EXT) { store_ad(iq_at(iq, ncell, Lay<EXT>::F_RV, c), q.Rv); store_ad(iq_at(iq, ncell, Lay<EXT>::F_TMULT, c), q.tmult); }
    store_ad(iq_at(iq, ncell, Lay<EXT>::F_PORO, c), q.poro);
}

// invalidateAndUpdateIntensiveQuantities(0): one lane per cell
// The per-cell kernels walk the property tables with binary searches: ~130 dependent loads per cell, each an L1 round
// trip (PMC: 41 us per wavefront).  The tables are a few KiB, so every workgroup copies the double blob into LDS first
// and searches there with ds_read (LdsTab); tables too large for the LDS budget stay in global memory (GlobalTab).
constexpr int TAB_LDS_DBL = 3072;   // 24 KiB
template <class DP, bool EXT>
__device__ __forceinline__ void iq_update_cell(const TablesT<DP>& T, const CellStatic& C, int c, const double* __restrict__ pv,
                                               const unsigned char* __restrict__ meaning, double* __restrict__ iq) {
    const double x[3] = {pv[(size_t)c * 3], pv[(size_t)c * 3 + 1], pv[(size_t)c * 3 + 2]};
    Iq<Ad> q;
    update_iq<Ad, DP, EXT>(T, C, c, x, meaning[c], q);
    store_iq<EXT>(iq, C.ncell, c, q);
    store_invb(C.invb, c, q);
}
// copies the blob into `s_tab` (workgroup-wide, contains a barrier); true if the tables fit
__device__ __forceinline__ bool tables_to_lds(const Tables& T, double* s_tab) {
    if (T.ndbl > TAB_LDS_DBL) return false;
    for (int i = threadIdx.x; i < T.ndbl; i += blockDim.x) s_tab[i] = T.dbl[i];
    __syncthreads();
    return true;
}
__device__ __forceinline__ TablesT<LdsTab> lds_tables(const Tables& T, double* s_tab) {
    return TablesT<LdsTab>{(LdsTab)s_tab, T.idx, T.rock_pref, T.rock_cr, T.ndbl, T.nidx, T.rock_desc, T.num_rock};
}
// The base layout's two per-cell kernels are held to three wavefronts per SIMD (168 VGPRs; left alone they take 172-174 and
// run two: 0.1865 -> 0.1742 ms on the bench, four waves spill and lose).  The extended layout needs 202 registers and stays free.
#ifndef OPMHIP_IQ_WAVES
#define OPMHIP_IQ_WAVES 3
#endif
#if OPMHIP_IQ_WAVES > 0
#define IQ_OCC __attribute__((amdgpu_waves_per_eu(OPMHIP_IQ_WAVES, OPMHIP_IQ_WAVES)))
#else
#define IQ_OCC
#endif
template <bool EXT>
__device__ __forceinline__ void iq_update_body(int c0, int Nb, const Tables& T, const CellStatic& C, const double* __restrict__ pv,
                                               const unsigned char* __restrict__ meaning, double* __restrict__ iq) {
    __shared__ double s_tab[TAB_LDS_DBL];
    const bool inLds = tables_to_lds(T, s_tab);
    const int c = c0 + blockIdx.x * blockDim.x + threadIdx.x;  // cells [c0, Nb)
    if (c >= Nb) return;
    if (inLds) iq_update_cell<LdsTab, EXT>(lds_tables(T, s_tab), C, c, pv, meaning, iq);
    else iq_update_cell<GlobalTab, EXT>(T, C, c, pv, meaning, iq);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k_iq_update(int c0, int Nb, Tables T, CellStatic C, const double* __restrict__ pv,
                                                   const unsigned char* __restrict__ meaning, double* __restrict__ iq) {
    iq_update_body<EXT>(c0, Nb, T, C, pv, meaning, iq);
}
template <>
__global__ __launch_bounds__(256) IQ_OCC void k_iq_update<false>(int c0, int Nb, Tables T, CellStatic C, const double* __restrict__ pv,
                                                                 const unsigned char* __restrict__ meaning, double* __restrict__ iq) {
    iq_update_body<false>(c0, Nb, T, C, pv, meaning, iq);
}

// BlackOilNewtonMethod::update_ (chopped update) + BlackOilPrimaryVariables::adaptPrimaryVariables + IQ recompute
template <class DP, bool EXT>
__device__ __forceinline__ void newton_update_cell(const TablesT<DP>& T, const CellStatic& C, int c, const double* __restrict__ dx, double relax,
                                                   double* __restrict__ pv, unsigned char* __restrict__ meaning,
                                                   unsigned char* __restrict__ wasSwitched, double* __restrict__ iq,
                                                   int* __restrict__ nswitched) {
    const double dpMaxRel = 0.3, dsMax = 0.2, oscThreshold = 1e-5;
    double x[3] = {pv[(size_t)c * 3], pv[(size_t)c * 3 + 1], pv[(size_t)c * 3 + 2]};
    double u[3] = {dx[(size_t)c * 3], dx[(size_t)c * 3 + 1], dx[(size_t)c * 3 + 2]};
    if (relax != 1.0) { u[0] *= relax; u[1] *= relax; u[2] *= relax; }
    int mng = meaning[c];
    const double deltaSw = u[0];
    double deltaSo = -deltaSw, deltaSg = 0.0;
    if (mng == OPMHIP_SW_PO_SG) { deltaSg = u[2]; deltaSo -= deltaSg; }
    double maxSatDelta = fmax(fabs(deltaSg), fabs(deltaSo));
    maxSatDelta = fmax(maxSatDelta, fabs(deltaSw));
    double satAlpha = 1.0;
    if (maxSatDelta > dsMax) satAlpha = dsMax / maxSatDelta;
    double nx[3];
    {
        double delta = u[0] * satAlpha;
        nx[0] = x[0] - delta;
        delta = u[1];
        if (fabs(delta) > dpMaxRel * x[1]) delta = (delta < 0.0 ? -1.0 : 1.0) * dpMaxRel * x[1];
        nx[1] = x[1] - delta;
        delta = u[2];
        if (mng == OPMHIP_SW_PO_SG) delta *= satAlpha;
        else if (delta > x[2]) delta = x[2];   // Rs / Rv must not become negative
        nx[2] = x[2] - delta;
    }
    x[0] = nx[0]; x[1] = nx[1]; x[2] = nx[2];
    const int pr = C.pvtnum ? C.pvtnum[c] : 0, sr = C.satnum ? C.satnum[c] : 0;
    const double RsMax = C.rsmax ? C.rsmax[c] : DBL_MAX / 2.0;
    const bool wet = EXT && T.pvt(pr).wg_n > 0;
    const double eps = wasSwitched[c] ? oscThreshold : 0.0;
    bool sw = false;
    // VAPPARS in the switches (adaptPrimaryVariables: SoMax = max(So, problem.maxOilSaturation)); 1 where it is not in force
    const bool vap = EXT && C.maxso;
    auto vap_f = [&](double So_, double par) { return vap ? vappars_factor<double>(So_, emax(So_, C.maxso[c]), par) : 1.0; };
    if (x[0] >= 1.0) {   // cells with (almost) only water
        x[0] = 1.0; x[2] = 0.0;
        sw = mng != OPMHIP_SW_PO_SG;
        mng = OPMHIP_SW_PO_SG;
    } else if (mng == OPMHIP_SW_PO_SG) {
        const double So = 1.0 - x[0] - x[2];
        if (x[2] < -eps && So > 0.0) {   // the gas phase disappears
            mng = OPMHIP_SW_PO_RS;
            x[2] = emin(RsMax, rs_sat_value(T, pr, x[1]) * vap_f(So, C.vap2));
            sw = true;
        } else if (EXT && wet && So < -eps && x[2] > 0.0) {   // the oil phase disappears: { Sw, pg, Rv }
            double pC[3];
            cell_cap_pressures<DP, EXT>(T, C, c, sr, x[0], x[2], pC);
            const double pg = x[1] + (pC[GAS] - pC[OIL]);
            const double RvMax = C.rvmax ? C.rvmax[c] : DBL_MAX / 2.0;
            mng = OPMHIP_SW_PG_RV;
            x[1] = pg;
            x[2] = emin(RvMax, rv_sat_value(T, pr, pg) * vap_f(So, C.vap1));
            sw = true;
        }
    } else if (!EXT || mng == OPMHIP_SW_PO_RS) {
        const double RsSat = rs_sat_value(T, pr, x[1]) * vap_f(1.0 - x[0], C.vap2);   // no gas: So = 1 - Sw
        if (x[2] > emin(RsMax, RsSat * (1.0 + eps))) { mng = OPMHIP_SW_PO_SG; x[2] = 0.0; sw = true; }
    } else {   // Sw_pg_Rv: the oil phase appears once the gas holds more oil than saturated gas does
        const double RvMax = C.rvmax ? C.rvmax[c] : DBL_MAX / 2.0;
        const double RvSat = rv_sat_value(T, pr, x[1]) * vap_f(0.0, C.vap1);   // no oil phase: So = 0
        if (x[2] > emin(RvMax, RvSat * (1.0 + eps))) {
            double pC[3];
            cell_cap_pressures<DP, EXT>(T, C, c, sr, x[0], 1.0 - x[0], pC);
            mng = OPMHIP_SW_PO_SG;
            x[1] = x[1] + (pC[OIL] - pC[GAS]);
            x[2] = 1.0 - x[0];
            sw = true;
        }
    }
    wasSwitched[c] = sw ? 1 : 0;
    meaning[c] = (unsigned char)mng;
    pv[(size_t)c * 3] = x[0]; pv[(size_t)c * 3 + 1] = x[1]; pv[(size_t)c * 3 + 2] = x[2];
    if (sw) atomicAdd(nswitched, 1);
    Iq<Ad> q;
    update_iq<Ad, DP, EXT>(T, C, c, x, mng, q);
    store_iq<EXT>(iq, C.ncell, c, q);
    store_invb(C.invb, c, q);
}
template <bool EXT>
__device__ __forceinline__ void newton_update_body(int Nb, const Tables& T, const CellStatic& C, const double* __restrict__ dx, double relax,
                                                   double* __restrict__ pv, unsigned char* __restrict__ meaning,
                                                   unsigned char* __restrict__ wasSwitched, double* __restrict__ iq,
                                                   int* __restrict__ nswitched) {
    __shared__ double s_tab[TAB_LDS_DBL];
    const bool inLds = tables_to_lds(T, s_tab);
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Nb) return;
    if (inLds) newton_update_cell<LdsTab, EXT>(lds_tables(T, s_tab), C, c, dx, relax, pv, meaning, wasSwitched, iq, nswitched);
    else newton_update_cell<GlobalTab, EXT>(T, C, c, dx, relax, pv, meaning, wasSwitched, iq, nswitched);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k_newton_update(int Nb, Tables T, CellStatic C, const double* __restrict__ dx, double relax,
                                                       double* __restrict__ pv, unsigned char* __restrict__ meaning,
                                                       unsigned char* __restrict__ wasSwitched, double* __restrict__ iq,
                                                       int* __restrict__ nswitched) {
    newton_update_body<EXT>(Nb, T, C, dx, relax, pv, meaning, wasSwitched, iq, nswitched);
}
template <>
__global__ __launch_bounds__(256) IQ_OCC void k_newton_update<false>(int Nb, Tables T, CellStatic C, const double* __restrict__ dx, double relax,
                                                                     double* __restrict__ pv, unsigned char* __restrict__ meaning,
                                                                     unsigned char* __restrict__ wasSwitched, double* __restrict__ iq,
                                                                     int* __restrict__ nswitched) {
    newton_update_body<false>(Nb, T, C, dx, relax, pv, meaning, wasSwitched, iq, nswitched);
}

static Tables tables_of(const opmhip_ctx* c);
// opmhip_fluid_probe: the property functions of update_iq at given points (plain doubles)
__global__ __launch_bounds__(256) void k_fluid_probe(Tables T, int pr, int sr, int n, const double* __restrict__ p,
                                                     const double* __restrict__ rs, const double* __restrict__ sw,
                                                     const double* __restrict__ sg, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PvtRegionDesc& D = T.pvt(pr);
    const SatRegionDesc& Sd = T.sat(sr);
    const GlobalTab B = T.dbl;
    double* o = out + (size_t)i * 8;
    const double pi = p[i], rsi = rs[i];
    {
        const GlobalTab W = B + D.water;  // p_ref, Bw_ref, c_w, mu_ref, c_v
        const double X = W[2] * (pi - W[0]);
        o[0] = (1.0 + X * (1.0 + X / 2.0)) / W[1];
    }
    const bool wetg = D.wg_n > 0;   // PVTG: the saturated curve
    o[1] = wetg ? tab1<double, GlobalTab>(B + D.wg_xs, B + D.wgs_invB, D.wg_n, pi) : tab1<double, GlobalTab>(B + D.gas_p, B + D.gas_invB, D.gas_n, pi);
    const double RsSat = rs_sat_value(T, pr, pi);
    o[3] = RsSat;
    if (rsi >= RsSat) {
        o[2] = tab1<double, GlobalTab>(B + D.sat_p, B + D.sat_invB, D.sat_n, pi);
        o[6] = o[2] / tab1<double, GlobalTab>(B + D.sat_p, B + D.sat_invBMu, D.sat_n, pi);
    } else {
        o[2] = tab2<double, GlobalTab>(T, D, D.o_invB, rsi, pi);
        o[6] = o[2] / tab2<double, GlobalTab>(T, D, D.o_invBMu, rsi, pi);
    }
    const double Swco = B[Sd.swco];
    o[4] = pwlin<double, GlobalTab>(B + Sd.sw_x, B + Sd.pcow, Sd.nw, sw[i]);
    o[5] = pwlin<double, GlobalTab>(B + Sd.so_x, B + Sd.pcgo, Sd.ng, 1.0 - Swco - sg[i]);
    o[7] = o[1] / (wetg ? tab1<double, GlobalTab>(B + D.wg_xs, B + D.wgs_invBMu, D.wg_n, pi) : tab1<double, GlobalTab>(B + D.gas_p, B + D.gas_invBMu, D.gas_n, pi));
}
// opmhip_gas_probe: wet-gas functions at (p_g, Rv): 1/B_g and mu_g on the saturated curve where Rv >= RvSat(p_g), else the
// undersaturated 2-D tables; RvSat(p_g).  Dry gas: 1/B_g(p), mu_g(p), 0.
__global__ __launch_bounds__(256) void k_gas_probe(Tables T, int pr, int n, const double* __restrict__ p, const double* __restrict__ rv,
                                                   double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PvtRegionDesc& D = T.pvt(pr);
    const GlobalTab B = T.dbl;
    double* o = out + (size_t)i * 3;
    const double pi = p[i], rvi = rv[i];
    if (D.wg_n > 0) {
        const double RvSat = rv_sat_value(T, pr, pi);
        o[2] = RvSat;
        if (rvi >= RvSat) {
            o[0] = tab1<double, GlobalTab>(B + D.wg_xs, B + D.wgs_invB, D.wg_n, pi);
            o[1] = o[0] / tab1<double, GlobalTab>(B + D.wg_xs, B + D.wgs_invBMu, D.wg_n, pi);
        } else {
            o[0] = tab2wg<double, GlobalTab>(T, D, D.wg_invB, pi, rvi);
            o[1] = o[0] / tab2wg<double, GlobalTab>(T, D, D.wg_invBMu, pi, rvi);
        }
    } else {
        o[0] = tab1<double, GlobalTab>(B + D.gas_p, B + D.gas_invB, D.gas_n, pi);
        o[1] = o[0] / tab1<double, GlobalTab>(B + D.gas_p, B + D.gas_invBMu, D.gas_n, pi);
        o[2] = 0.0;
    }
}
int launch_gas_probe(opmhip_ctx* c, int pr, int n, const double* d_in, double* d_out) {
    hipLaunchKernelGGL(k_gas_probe, dim3((n + 255) / 256), dim3(256), 0, c->stream, tables_of(c), pr, n, d_in, d_in + n, d_out);
    return OPMHIP_SUCCESS;
}
// opmhip_sat_probe: the saturation functions with one set of scaled end points (eps[EPS_COUNT]; cfg = packed EclEpsConfig) or,
// cfg < 0, the tables' own: out[5 i ..] = krw, kro, krg, pcow, pcgo at (sw, sg) - the functions update_iq evaluates
__global__ __launch_bounds__(256) void k_sat_probe(Tables T, int sr, int cfg, const double* __restrict__ eps, int n, const double* __restrict__ sw,
                                                   const double* __restrict__ sg, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SatRegionDesc& Sd = T.sat(sr);
    const GlobalTab B = T.dbl;
    double u[EPS_COUNT], s[EPS_COUNT];
#pragma unroll
    for (int f = 0; f < EPS_COUNT; ++f) { u[f] = B[Sd.eps + f]; s[f] = cfg >= 0 ? eps[f] : u[f]; }
    double kr[3], pC[3];
    rel_perms_eps<double, GlobalTab>(B, Sd, cfg >= 0 ? cfg : 0, u, s, sw[i], sg[i], kr);
    cap_pressures_eps<double, GlobalTab>(B, Sd, cfg >= 0 ? cfg : 0, u, s, sw[i], sg[i], pC);
    double* o = out + (size_t)i * 5;
    o[0] = kr[0]; o[1] = kr[1]; o[2] = kr[2]; o[3] = -pC[0]; o[4] = pC[2];
}
int launch_sat_probe(opmhip_ctx* c, int sr, int cfg, const double* d_eps, int n, const double* d_in, double* d_out) {
    hipLaunchKernelGGL(k_sat_probe, dim3((n + 255) / 256), dim3(256), 0, c->stream, tables_of(c), sr, cfg, d_eps, n, d_in, d_in + n, d_out);
    return OPMHIP_SUCCESS;
}
int launch_fluid_probe(opmhip_ctx* c, int pr, int sr, int n, const double* d_in, double* d_out) {
    hipLaunchKernelGGL(k_fluid_probe, dim3((n + 255) / 256), dim3(256), 0, c->stream, tables_of(c), pr, sr, n, d_in, d_in + n, d_in + 2 * (size_t)n,
                       d_in + 3 * (size_t)n, d_out);
    return OPMHIP_SUCCESS;
}

// ============================== face flux ===================================================================
// calculateGradients_ + computeFlux for one face: `in` = focus (interior) cell with derivatives, `ex` = exterior cell,
// of which only values enter.  Result: flux[eq] * faceArea.  The two cells come through accessors: PtrQ reads a record
// of the IQ cache where it lies (here: the tile's own rows, staged in LDS), RegQ holds the flux-relevant fields of one
// record in registers (the neighbour cell, fetched with one batch of loads before any of the branchy arithmetic).
struct PtrQ {
    const double* q;
    __device__ __forceinline__ Ad ad(int f) const { return load_ad(q + f * 4); }
    __device__ __forceinline__ double v(int f) const { return q[f * 4]; }
};
template <bool EXT>
struct RegQ {   // fields F_P .. F_RS (.. tmult): p, 1/B, mobility, density (3 phases each), Rs [, Rv, transmissibility multiplier]
    static constexpr int F0 = Lay<EXT>::RQ_F0;
    double r[Lay<EXT>::RQ_NF * 4];
    __device__ __forceinline__ Ad ad(int f) const { return Ad{r[(f - F0) * 4], r[(f - F0) * 4 + 1], r[(f - F0) * 4 + 2], r[(f - F0) * 4 + 3]}; }
    __device__ __forceinline__ double v(int f) const { return r[(f - F0) * 4]; }
};
// keeps a batch of loads together: the values must exist here, so the compiler cannot sink each load into the branch
// that consumes it (where it would cost its own LDS round trip)
__device__ __forceinline__ void pin(Ad& a) { asm volatile("" : "+v"(a.v), "+v"(a.d0), "+v"(a.d1), "+v"(a.d2)); }
__device__ __forceinline__ void pin(double& a) { asm volatile("" : "+v"(a)); }
template <bool EXT, class IN, class EX>
__device__ __forceinline__ void face_flux(const IN& in, const EX& ex, bool wet, double trans, double faceArea,
                                          double thpres, double zIn, double zEx, double Vin, double Vex, bool inLower, Ad flux[3]) {
    flux[0] = flux[1] = flux[2] = ad_const(0.0);
    const double distZ = zIn - zEx;
    const int comp[3] = {EQ_WATER, EQ_OIL, EQ_GAS};
    Ad tmIn = ad_const(1.0);   // transMult = rockCompTransMultiplier of the upstream cell (eclfluxmodule.hh:340-355)
    double tmEx = 1.0;
    if (EXT) { tmIn = in.ad(Lay<EXT>::F_TMULT); tmEx = ex.v(Lay<EXT>::F_TMULT); }
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
        // everything this phase may need of the two cells in ONE round of (LDS) loads, before the branches
        Ad mobIn = in.ad(F_MOB + ph), rhoIn = in.ad(F_RHO + ph), pIn = in.ad(F_P + ph), bIn = in.ad(F_B + ph);
        double mobEx = ex.v(F_MOB + ph), rhoEx = ex.v(F_RHO + ph), pEx = ex.v(F_P + ph), bEx = ex.v(F_B + ph);
        Ad rIn = ad_const(0.0);   // Rs with the oil phase, Rv with the gas phase
        double rEx = 0.0;
        if (ph == OIL) { rIn = in.ad(F_RS); rEx = ex.v(F_RS); }
        if (EXT && ph == GAS) { rIn = in.ad(Lay<EXT>::F_RV); rEx = ex.v(Lay<EXT>::F_RV); }
        pin(mobIn); pin(rhoIn); pin(pIn); pin(bIn); pin(mobEx); pin(rhoEx); pin(pEx); pin(bEx);
        if (ph == OIL || (EXT && ph == GAS)) { pin(rIn); pin(rEx); }
        if (mobIn.v <= 0.0 && mobEx <= 0.0) continue;
        const Ad rhoAvg = (rhoIn + rhoEx) / 2.0;
        Ad pressureExterior = ad_const(pEx);
        pressureExterior = pressureExterior + rhoAvg * (distZ * GRAVITY);
        Ad dp = pressureExterior - pIn;
        bool upIn;
        if (dp.v > 0.0) upIn = false;
        else if (dp.v < 0.0) upIn = true;
        else if (Vin > Vex) upIn = true;
        else if (Vin < Vex) upIn = false;
        else upIn = inLower;  // equal pressures and volumes: the lower GLOBAL index is upstream (eclfluxmodule.hh:303-314)
        if (fabs(dp.v) > thpres) {
            if (dp.v < 0.0) dp = dp + thpres; else dp = dp - thpres;
        } else continue;
        Ad volumeFlux, surf;
        if (upIn) {
            volumeFlux = dp * mobIn * (EXT ? tmIn : ad_const(1.0)) * (-trans / faceArea);
            surf = bIn * volumeFlux;
        } else {
            volumeFlux = dp * (mobEx * (EXT ? tmEx : 1.0) * (-trans / faceArea));
            surf = bEx * volumeFlux;
        }
        flux[comp[ph]] = flux[comp[ph]] + surf;
        if (ph == OIL) {
            if (upIn) flux[EQ_GAS] = flux[EQ_GAS] + rIn * surf;
            else flux[EQ_GAS] = flux[EQ_GAS] + rEx * surf;
        } else if (EXT && ph == GAS && wet) {   // vaporised oil carried by the gas phase
            if (upIn) flux[EQ_OIL] = flux[EQ_OIL] + rIn * surf;
            else flux[EQ_OIL] = flux[EQ_OIL] + rEx * surf;
        }
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) flux[e] = flux[e] * faceArea;
}

// ============================== assembly ======================================================================
#ifndef OPMHIP_ASM_THREADS
#define OPMHIP_ASM_THREADS 64
#endif
constexpr int ASM_THREADS = OPMHIP_ASM_THREADS;  // 64: one wavefront per tile (9 rows of a 7-point grid); measured 0.80 ms against 0.84 (128) and 0.88 (256): no cross-wave barrier stalls
static_assert(ASM_THREADS == 64, "k_assemble: one lane per entry of a one-wavefront tile");
constexpr int ASM_MAX_ROWS = ASM_THREADS / 7 + 4;  // rows of one tile (their IQ records are staged in LDS); 36 of 40 on a 7-point grid
int asm_max_rows() { return ASM_MAX_ROWS; }
int asm_threads() { return ASM_THREADS; }
#ifndef OPMHIP_ASM_WAVES
#define OPMHIP_ASM_WAVES 3
#endif
#if OPMHIP_ASM_WAVES > 0
#define ASM_OCC __attribute__((amdgpu_waves_per_eu(OPMHIP_ASM_WAVES, OPMHIP_ASM_WAVES)))
#else
#define ASM_OCC
#endif
struct EntryStatic {
    const double *trans, *area, *thpres;  // per entry, internal order
};
constexpr int ASM_PRE = 10;  // what a diagonal lane fetches ahead for its cell: storageOld[3], source[3], drift[3], reference porosity
// FvBaseLinearizer::linearizeDomain.  One lane per block-CSR entry, one wavefront per tile of whole rows (<= ASM_THREADS
// entries, <= ASM_MAX_ROWS rows).  The kernel is bound by the latency of its dependent load rounds (PMC: waves wait 65 % of
// their life with two waves per SIMD and seven rounds), so there are two of them and room for three waves per SIMD:
//   1  the tile's schedule record (rows, entries) and - found from the workgroup index alone - every lane's entry record
//      (column, entry word: row inside the tile, tie-break flag, natural summation order; capi_asm.cpp)
//   2  own IQ records -> LDS (coalesced), depth / volume of the rows, per lane transmissibility, area, THPRES;
//      off-diagonal lanes: the neighbour's flux fields -> registers; diagonal lanes: storage of the old time level, source,
//      drift (into the same registers, parked in LDS before the arithmetic starts)
// The arithmetic then runs out of LDS / registers.  LDS: the own records are dead once every face is evaluated; the face
// fluxes and the tile's blocks take their place (11 KiB per workgroup instead of 18).
#ifdef OPMHIP_ASM_SOFT_SYNC
__device__ __forceinline__ void asm_wave_sync() { asm volatile("" ::: "memory"); }  // one wavefront per workgroup: its LDS accesses execute in program order
#else
__device__ __forceinline__ void asm_wave_sync() { __syncthreads(); }
#endif
template <bool EXT>
__global__ __launch_bounds__(ASM_THREADS) ASM_OCC void k_assemble(int nsched, const int4* __restrict__ sched, const int2* __restrict__ desc, int wet,
                                                          EntryStatic ES, CellStatic C, const double* __restrict__ iq,
                                                          double* __restrict__ storageOld, const double* __restrict__ source,
                                                          const double* __restrict__ dsource, const double* __restrict__ drift, double maxCompensation,
                                                          double dt, int iteration, double* __restrict__ A, double* __restrict__ resid) {
    constexpr int IQS = Lay<EXT>::IQS, RQ_F0 = Lay<EXT>::RQ_F0, RQ_NF = Lay<EXT>::RQ_NF, F_PORO = Lay<EXT>::F_PORO;
    constexpr int N_OWN = ASM_MAX_ROWS * IQS, N_FLUX = ASM_THREADS * 12, N_OUT = N_FLUX + (ASM_THREADS + 2) * BB;
    __shared__ __attribute__((aligned(16))) double sU[N_OWN > N_OUT ? N_OWN : N_OUT];
    double* const sI = sU;              // until the faces are evaluated: IQ records of the tile's rows
    double* const sflux = sU;           // then: face flux seen from the row's cell, 3 equations x (value, 3 derivatives) per entry
    double* const sblk = sU + N_FLUX;   //       and the tile's blocks, streamed out at the end
    __shared__ double sgeo[2 * ASM_MAX_ROWS];
    __shared__ double spre[ASM_MAX_ROWS * ASM_PRE];
    __shared__ unsigned char snat[ASM_THREADS];
    __shared__ short sfirst[ASM_MAX_ROWS + 1];
    static_assert(RQ_NF >= ASM_PRE, "the diagonal lane borrows the value slots of the neighbour record");
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nsched) return;
    // ---- round 1: the tile's schedule record and, independent of it, the lane's entry (column, entry word)
    const int4 S = sched[blockIdx.x];
    const int2 jm = desc[(size_t)blockIdx.x * ASM_THREADS + tid];
    const int r0 = S.x, k0 = S.z, nent = S.w - S.z, nrows = S.y - S.x;
    const int k0e = k0 & ~1;  // 16-byte aligned start of the output stream
    const bool act = tid < nent;
    const int k = act ? k0 + tid : k0;
    const int J = jm.x;
    const unsigned m = (unsigned)jm.y;
    {   // ---- round 2.  Own records: per field one contiguous run of nrows x 32 bytes of the cache, into per-row records in LDS
        const int n2r = nrows * 2;                              // 16-byte pieces per field
        const unsigned inv = n2r > 0 ? (65536u + n2r - 1) / n2r : 0u;  // i / n2r == (i * inv) >> 16 for the few hundred i of a tile
        const int n2 = n2r * Lay<EXT>::IQF;
        const double2* g2 = reinterpret_cast<const double2*>(iq);
        double2* s2 = reinterpret_cast<double2*>(sI);
        for (int i = tid; i < n2; i += ASM_THREADS) {
            const int fld = (int)(((unsigned)i * inv) >> 16), j = i - fld * n2r;
            s2[(j >> 1) * (IQS / 2) + fld * 2 + (j & 1)] = g2[((size_t)fld * C.ncell + r0) * 2 + j];
        }
    }
    if (tid < nrows) { sgeo[tid] = C.depth[r0 + tid]; sgeo[ASM_MAX_ROWS + tid] = C.volume[r0 + tid]; }
    const double trans = ES.trans[k], area = ES.area[k], thp = ES.thpres ? ES.thpres[k] : 0.0;
    const int lrow = m & 63;
    const bool lowI = ((m >> 6) & 1u) != 0;
    const int I = r0 + lrow;
    const bool isDiag = act && I == J, isOff = act && I != J;
    snat[tid] = (unsigned char)(m >> 8);
    {
        const int prevRow = __shfl_up(lrow, 1);
        if (act && (tid == 0 || prevRow != lrow)) sfirst[lrow] = (short)tid;
        if (tid == 0) sfirst[nrows] = (short)nent;
    }
    RegQ<EXT> qJ;
    double zJ = 0.0, VJ = 0.0;
    if (isOff) {
        const double2* g2 = reinterpret_cast<const double2*>(iq) + (size_t)J * 2;
#pragma unroll
        for (int i = 0; i < RQ_NF; ++i) {
            const double2 a = g2[(size_t)(RQ_F0 + i) * C.ncell * 2], b = g2[(size_t)(RQ_F0 + i) * C.ncell * 2 + 1];
            qJ.r[4 * i] = a.x; qJ.r[4 * i + 1] = a.y; qJ.r[4 * i + 2] = b.x; qJ.r[4 * i + 3] = b.y;
        }
        zJ = C.depth[J]; VJ = C.volume[J];
    } else if (isDiag) {
        const size_t o = (size_t)I * 3;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            qJ.r[4 * e] = (iteration == 0) ? 0.0 : storageOld[o + e];
            qJ.r[4 * (3 + e)] = source ? source[o + e] : 0.0;
            qJ.r[4 * (6 + e)] = drift ? drift[o + e] : 0.0;
        }
        qJ.r[4 * 9] = drift ? C.poro[I] : 1.0;   // referencePorosity
    }
    asm_wave_sync();
    const double zI = sgeo[lrow], VI = sgeo[ASM_MAX_ROWS + lrow];
    Ad f[3];         // off-diagonal lane: face flux seen from cell I; diagonal lane: storage term
    double blk[BB];  // off-diagonal lane: block (I,J)
    const PtrQ qI{sI + lrow * IQS};
    if (isOff) {
        face_flux<EXT>(qJ, qI, wet != 0, trans, area, thp, zJ, zI, VJ, VI, !lowI, f);  // focus J: residual[I] -= flux  ->  block (I,J)
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const Ad mm = ad_const(0.0) - f[e];
            blk[e * 3 + 0] = mm.d0; blk[e * 3 + 1] = mm.d1; blk[e * 3 + 2] = mm.d2;
        }
        face_flux<EXT>(qI, qJ, wet != 0, trans, area, thp, zI, zJ, VI, VJ, lowI, f);   // focus I: contribution to R_I
    } else if (isDiag) {
#pragma unroll
        for (int i = 0; i < ASM_PRE; ++i) spre[lrow * ASM_PRE + i] = qJ.r[4 * i];
        // computeStorage: surface volumes per bulk volume
        const Ad poro = qI.ad(F_PORO), Rs = qI.ad(F_RS);
        f[0] = f[1] = f[2] = ad_const(0.0);
        const int comp[3] = {EQ_WATER, EQ_OIL, EQ_GAS};
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const Ad surfaceVolume = qI.ad(F_S + ph) * qI.ad(F_B + ph) * poro;
            f[comp[ph]] = f[comp[ph]] + surfaceVolume;
            if (ph == OIL) f[EQ_GAS] = f[EQ_GAS] + Rs * surfaceVolume;
            if (EXT && ph == GAS && wet) f[EQ_OIL] = f[EQ_OIL] + qI.ad(Lay<EXT>::F_RV) * surfaceVolume;   // vaporised oil
        }
    }
    asm_wave_sync();   // every read of the own records is done: their LDS becomes sflux / sblk
    if (isOff) {
#pragma unroll
        for (int e = 0; e < 3; ++e) store_ad(&sflux[tid * 12 + e * 4], f[e]);
        double* b = &sblk[(k - k0e) * BB];
#pragma unroll
        for (int q = 0; q < BB; ++q) b[q] = blk[q];
    }
    asm_wave_sync();
    if (isDiag) {
        Ad R[3] = {ad_const(0.0), ad_const(0.0), ad_const(0.0)};
        // flux terms first (FvBaseLocalResidual::eval), faces in ascending NATURAL neighbour order whatever the
        // internal ordering is, so that the sum is the one the natural-order CPU path forms
        const int e0 = sfirst[lrow], e1 = sfirst[lrow + 1];
        // the next face's flux is read from LDS while the present one is added
        auto face_of = [&](int i) { const int q = e0 + snat[i < e1 ? i : e1 - 1]; return q; };
        Ad nx[3];
        int qn = face_of(e0);
        for (int e = 0; e < 3; ++e) nx[e] = load_ad(&sflux[qn * 12 + e * 4]);
        for (int i = e0; i < e1; ++i) {
            const int q = qn;
            Ad cur[3] = {nx[0], nx[1], nx[2]};
            qn = face_of(i + 1);
            for (int e = 0; e < 3; ++e) nx[e] = load_ad(&sflux[qn * 12 + e * 4]);
            if (q == tid) continue;   // the diagonal entry itself
            for (int e = 0; e < 3; ++e) R[e] = R[e] + cur[e];
        }
        const double V = VI;
        const double* pre = &spre[lrow * ASM_PRE];
        for (int e = 0; e < 3; ++e) {
            double old;
            if (iteration == 0) { old = f[e].v; storageOld[(size_t)I * 3 + e] = old; } else old = pre[e];
            Ad tt = f[e] - old;
            tt = tt * (V / dt);
            R[e] = R[e] + tt;
        }
        // drift compensation (ebos/eclproblem.hh:1847-1875): what the last accepted time step left unconverged in this
        // cell (residual * dt, opmhip_end_time_step) goes back in as a rate, capped at maxCompensation of the pore volume
        double dofDriftRate[3] = {0.0, 0.0, 0.0};
        if (drift) {
            const double poro = pre[9];   // referencePorosity
#pragma unroll
            for (int e = 0; e < 3; ++e) dofDriftRate[e] = pre[6 + e] / (dt * V);
            double totalDriftRate = 0.0;
#pragma unroll
            for (int e = 0; e < 3; ++e) totalDriftRate += fabs(dofDriftRate[e]) * dt * 1.0 / poro;   // eqWeight = 1 (UNVERIFIED, see oracle)
            if (totalDriftRate > maxCompensation) {
#pragma unroll
                for (int e = 0; e < 3; ++e) dofDriftRate[e] *= maxCompensation / totalDriftRate;
            }
        }
        for (int e = 0; e < 3; ++e) {
            Ad s = ad_const(pre[3 + e]);
            if (dsource) { s.d0 = dsource[(size_t)I * 9 + e * 3]; s.d1 = dsource[(size_t)I * 9 + e * 3 + 1]; s.d2 = dsource[(size_t)I * 9 + e * 3 + 2]; }
            s = s / V;
            if (drift) s = s - dofDriftRate[e];
            s = s * V;
            R[e] = R[e] - s;
        }
        double* b = &sblk[(k - k0e) * BB];
        for (int e = 0; e < 3; ++e) {
            resid[(size_t)I * 3 + e] = R[e].v;
            b[e * 3 + 0] = R[e].d0; b[e * 3 + 1] = R[e].d1; b[e * 3 + 2] = R[e].d2;
        }
    }
    asm_wave_sync();
    // stream the tile's blocks out: contiguous range [k0, k1) x 72 B
    {
        const int head = (k0 - k0e) * BB;      // doubles to skip at the front (0 or 9)
        const int n = nent * BB;
        double* dst = A + (size_t)k0e * BB;
        int b = head, e = head + n;
        // unaligned head / tail doubles go out one by one, the body as double2
        if ((b & 1) && tid == 0) dst[b] = sblk[b];
        if ((e & 1) && tid == 0) dst[e - 1] = sblk[e - 1];
        b = (b + 1) & ~1;
        e = e & ~1;
        const double2* s2 = reinterpret_cast<const double2*>(sblk);
        double2* d2 = reinterpret_cast<double2*>(dst);
        // nontemporal: the Jacobian leaves for good (500 MB), the intensive-quantity records the neighbouring tiles gather stay in L2
        // (0.607 -> 0.601 ms in alternation, profiles/r06_nt_operands_ab.txt)
#ifndef OPMHIP_ASM_PLAIN_STORES
        typedef double v2d_a __attribute__((ext_vector_type(2)));
        for (int i = (b >> 1) + tid; i < (e >> 1); i += ASM_THREADS) {
            const double2 t = s2[i];
            v2d_a v; v.x = t.x; v.y = t.y;
            __builtin_nontemporal_store(v, reinterpret_cast<v2d_a*>(&d2[i]));
        }
#else
        for (int i = (b >> 1) + tid; i < (e >> 1); i += ASM_THREADS) d2[i] = s2[i];
#endif
    }
}

// ============================== convergence ===================================================================
// pass 1 partials per block: R_sum[3], maxCoeff[3], sum(1/b)[3], pvSum ; pass 2: cnvErrorPv
__global__ __launch_bounds__(256) void k_conv_pass1(int Nb, CellStatic C, const double* __restrict__ resid, double* __restrict__ part) {
    __shared__ double sh[10][4];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    double a[10];
    for (int i = 0; i < 10; ++i) a[i] = (i >= 3 && i < 6) ? -DBL_MAX : 0.0;
    if (c < Nb) {
        const double pvValue = C.poro[c] * C.volume[c];
        const int comp[3] = {EQ_WATER, EQ_OIL, EQ_GAS};
        for (int ph = 0; ph < 3; ++ph) {
            const int e = comp[ph];
            a[6 + e] = 1.0 / C.invb[(size_t)c * 3 + ph];
            const double R2 = resid[(size_t)c * 3 + e];
            a[e] = R2;
            a[3 + e] = fabs(R2) / pvValue;
        }
        a[9] = pvValue;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = 0; i < 10; ++i) {
        double v = a[i];
        for (int o = 32; o > 0; o >>= 1) {
            const double w = __shfl_down(v, o, 64);
            v = (i >= 3 && i < 6) ? fmax(v, w) : v + w;
        }
        if (lane == 0) sh[i][wv] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const int i = threadIdx.x;
        double v = sh[i][0];
        for (int w = 1; w < 4; ++w) v = (i >= 3 && i < 6) ? fmax(v, sh[i][w]) : v + sh[i][w];
        part[(size_t)blockIdx.x * 10 + i] = v;
    }
}
// one workgroup: reduce the block partials in a fixed order; out[0..9] as above with B_avg divided by Nb
// decomposed runs: the local results are combined over the ranks (sum of R_sum, sum(1/b), pvSum; max of maxCoeff:
// convergenceReduction, flow/BlackoilModelEbos.hpp:572-625) before B_avg is formed with the GLOBAL cell count (:722-727)
__global__ void k_conv_pack(const double* __restrict__ out, double* __restrict__ red) {
    if (threadIdx.x < 3) { red[threadIdx.x] = out[threadIdx.x]; red[3 + threadIdx.x] = out[6 + threadIdx.x]; red[8 + threadIdx.x] = out[3 + threadIdx.x]; }
    if (threadIdx.x == 0) red[6] = out[9];
}
__global__ void k_conv_unpack(const double* __restrict__ red, double global_cells, double* __restrict__ out) {
    if (threadIdx.x < 3) { out[threadIdx.x] = red[threadIdx.x]; out[6 + threadIdx.x] = red[3 + threadIdx.x] / global_cells; out[3 + threadIdx.x] = red[8 + threadIdx.x]; }
    if (threadIdx.x == 0) out[9] = red[6];
}
__global__ __launch_bounds__(256) void k_conv_final1(int nblocks, int Nb, const double* __restrict__ part, double* __restrict__ out) {
    __shared__ double sh[10][256];
    double a[10];
    for (int i = 0; i < 10; ++i) a[i] = (i >= 3 && i < 6) ? -DBL_MAX : 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256)
        for (int i = 0; i < 10; ++i) {
            const double v = part[(size_t)b * 10 + i];
            a[i] = (i >= 3 && i < 6) ? fmax(a[i], v) : a[i] + v;
        }
    for (int i = 0; i < 10; ++i) sh[i][threadIdx.x] = a[i];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int i = 0; i < 10; ++i) {
                const double v = sh[i][threadIdx.x + o];
                sh[i][threadIdx.x] = (i >= 3 && i < 6) ? fmax(sh[i][threadIdx.x], v) : sh[i][threadIdx.x] + v;
            }
        __syncthreads();
    }
    if (threadIdx.x < 10) {
        double v = sh[threadIdx.x][0];
        if (threadIdx.x >= 6 && threadIdx.x < 9 && Nb > 0) v /= (double)Nb;  // Nb <= 0: keep the raw sum (decomposed run)
        out[threadIdx.x] = v;
    }
}
__global__ __launch_bounds__(256) void k_conv_pass2(int Nb, CellStatic C, const double* __restrict__ resid, const double* __restrict__ out1,
                                                    double dt, double tol_cnv, double* __restrict__ part) {
    __shared__ double sh[4];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    if (c < Nb) {
        const double pvValue = C.poro[c] * C.volume[c];
        bool violated = false;
        for (int e = 0; e < 3; ++e) {
            const double CNV = resid[(size_t)c * 3 + e] * dt * out1[6 + e] / pvValue;
            violated = violated || (fabs(CNV) > tol_cnv);
        }
        if (violated) v = pvValue;
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void k_conv_final2(int nblocks, const double* __restrict__ part, double* __restrict__ out) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) a += part[b];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[10] = sh[0];
}

// EclProblem::endTimeStep, drift part (ebos/eclproblem.hh:1126-1135): drift = residual * dt of the accepted step
__global__ void k_drift_update(int n, const double* __restrict__ resid, double dt, double* __restrict__ drift) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) drift[e] = resid[e] * dt;
}
// ---- EclProblem's per-cell bookkeeping between time steps ---------------------------------------------------------------
// updateCompositionChangeLimits_ (ebos/eclproblem.hh:2010-2107): lastRs = Rs where the DRSDT limit binds (all cells of the
// region, or those with free gas: Sg > freeGasMinSaturation_ = 1e-7, :2073, 2877), else infinity; lastRv = Rv
template <bool EXT>
__global__ __launch_bounds__(256) void k_last_rs_rv(int N, const int* __restrict__ pvtnum, const int* __restrict__ drsdt_all, const double* __restrict__ iq,
                                                    double* __restrict__ lastRs, double* __restrict__ lastRv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    if (lastRs) {
        const int pr = pvtnum ? pvtnum[c] : 0;
        const double Sg = iq_at(iq, N, F_S + GAS, c)[0];
        lastRs[c] = (drsdt_all[pr] || Sg > 1e-7) ? iq_at(iq, N, F_RS, c)[0] : INFINITY;
    }
    if (EXT && lastRv) lastRv[c] = iq_at(iq, N, Lay<EXT>::F_RV, c)[0];
}
// maxGasDissolutionFactor / maxOilVaporizationFactor (:1711-1754) of time level 0 (dt > 0: lastRs + DRSDT * dt,
// eclgenericproblem.cc: maxDRs_ = DRSDT * timeStepSize) or of time level 1 (dt = 0)
__global__ __launch_bounds__(256) void k_set_limits(int N, double dt, const int* __restrict__ pvtnum, const double* __restrict__ drsdt, const double* __restrict__ drvdt,
                                                    const double* __restrict__ lastRs, const double* __restrict__ lastRv, double* __restrict__ rsmax, double* __restrict__ rvmax) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const int pr = pvtnum ? pvtnum[c] : 0;
    if (drsdt) rsmax[c] = (drsdt[pr] < 0.0) ? DBL_MAX / 2.0 : lastRs[c] + drsdt[pr] * dt;
    if (drvdt) rvmax[c] = (drvdt[pr] < 0.0) ? DBL_MAX / 2.0 : lastRv[c] + drvdt[pr] * dt;
}
// updateMinPressure_ (:2172-2197); init: minOilPressure_ = min(1e99, p_o of the initial state) (eclgenericproblem.cc:165, :2293-2294)
__global__ __launch_bounds__(256) void k_min_pressure(int N, int init, const double* __restrict__ iq, double* __restrict__ minpo) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const double po = iq_at(iq, N, F_P + OIL, c)[0];
    const double old = init ? 1e99 : minpo[c];
    minpo[c] = (po < old) ? po : old;
}
// updateMaxOilSaturation_ (:2110-2141); init: maxOilSaturation_ = max(0, S_o of the initial state) (:2291-2292)
__global__ __launch_bounds__(256) void k_max_oil_saturation(int N, int init, const double* __restrict__ iq, double* __restrict__ maxso) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const double So = iq_at(iq, N, F_S + OIL, c)[0];
    const double old = init ? 0.0 : maxso[c];
    maxso[c] = (old > So) ? old : So;   // std::max(old, So)
}
// updateMaxWaterSaturation_ (:2144-2169); init: maxWaterSaturation_ = max(0, S_w of the initial state) (:2289-2290) and the
// initial saturation itself (initialFluidStates_).  The statement :2150 in front of the reference's loop (cell 1 takes over
// cell 0's stored maximum) is launch_max_water_saturation's copy
__global__ void k_copy_entry(double* __restrict__ a, int to, int from) { a[to] = a[from]; }
__global__ __launch_bounds__(256) void k_max_water_saturation(int N, int init, const double* __restrict__ iq, double* __restrict__ maxsw, double* __restrict__ sw0) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const double Sw = iq_at(iq, N, F_S + WATER, c)[0];
    const double old = init ? 0.0 : maxsw[c];
    maxsw[c] = (old > Sw) ? old : Sw;   // std::max(old, Sw)
    if (init) sw0[c] = Sw;
}
// updateHysteresis_ (:2603-2626) -> EclDefaultMaterial::updateHysteresis (its default "inconsistent" form): the oil-water system sees
// 1 - So, the gas-oil system 1 - Sg (Sg clamped to [0, 1]); a system whose wetting saturation falls below its turning point moves
// the turning point and recomputes the imbibition curve's shift (EclHysteresisTwoPhaseLawParams::update / updateDynamicParams_;
// oracle/fluid.hpp SatFunc::hystSee, same statements).  sw_ow / sw_go != NULL: restart (initHysteresisParams) - the turning
// points handed in (cell order of the device), starting from "nothing seen"
__global__ __launch_bounds__(256) void k_hyst_update(int N, Tables T, CellStatic C, const double* __restrict__ iq, double* __restrict__ hyst,
                                                     const double* __restrict__ sw_ow, const double* __restrict__ sw_go) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const GlobalTab B = T.dbl;
    const SatRegionDesc& Sd = T.sat(C.satnum ? C.satnum[c] : 0);
    const SatRegionDesc& Si = T.sat(C.imbnum[c]);
    const bool scaled = C.eps != nullptr;
    const int cfg = C.epscfg;
    double uD[EPS_COUNT], sD[EPS_COUNT], uI[EPS_COUNT], sI[EPS_COUNT];
    if (scaled) { eps_load<GlobalTab>(C, c, B, Sd, uD, sD); eps_load_imb<GlobalTab>(C, c, B, Si, uI, sI); }
    double mdcOw = 2.0, dOw = 0.0, mdcGo = 2.0, dGo = 0.0, sOw, sGo;
    if (sw_ow) { sOw = sw_ow[c]; sGo = sw_go[c]; }
    else {
        mdcOw = hyst[c]; dOw = hyst[(size_t)N + c]; mdcGo = hyst[(size_t)2 * N + c]; dGo = hyst[(size_t)3 * N + c];
        const double So = iq_at(iq, N, F_S + OIL, c)[0];
        double Sg = iq_at(iq, N, F_S + GAS, c)[0];
        Sg = emin(1.0, emax(0.0, Sg));
        sOw = 1.0 - So;
        sGo = 1.0 - Sg;
    }
    if (sOw < mdcOw) {
        mdcOw = sOw;
        const double krnMdcDrainage = sat_curve<double, GlobalTab>(B, Sd, KRN_OW, sOw, scaled, cfg, uD, sD);
        const double SwKrnMdcImbibition = sat_curve_inv<GlobalTab>(B, Si, KRN_OW, krnMdcDrainage, scaled, cfg, uI, sI);
        dOw = SwKrnMdcImbibition - sOw;
    }
    if (sGo < mdcGo) {
        mdcGo = sGo;
        const double krnMdcDrainage = sat_curve<double, GlobalTab>(B, Sd, KRN_GO, sGo, scaled, cfg, uD, sD);
        const double SwKrnMdcImbibition = sat_curve_inv<GlobalTab>(B, Si, KRN_GO, krnMdcDrainage, scaled, cfg, uI, sI);
        dGo = SwKrnMdcImbibition - sGo;
    }
    hyst[c] = mdcOw; hyst[(size_t)N + c] = dOw; hyst[(size_t)2 * N + c] = mdcGo; hyst[(size_t)3 * N + c] = dGo;
}
// the storage term of the cached intensive quantities, values only (computeStorage; the statements of k_assemble's diagonal
// lane): the old time level's storage where the first iteration's cannot be recycled (:1758-1765)
template <bool EXT>
__global__ __launch_bounds__(256) void k_storage_old(int Nb, int N, int wet, const double* __restrict__ iq, double* __restrict__ storageOld) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Nb) return;
    const double poro = iq_at(iq, N, Lay<EXT>::F_PORO, c)[0], Rs = iq_at(iq, N, F_RS, c)[0];
    double f[3] = {0.0, 0.0, 0.0};
    const int comp[3] = {EQ_WATER, EQ_OIL, EQ_GAS};
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
        const double surfaceVolume = iq_at(iq, N, F_S + ph, c)[0] * iq_at(iq, N, F_B + ph, c)[0] * poro;
        f[comp[ph]] = f[comp[ph]] + surfaceVolume;
        if (ph == OIL) f[EQ_GAS] = f[EQ_GAS] + Rs * surfaceVolume;
        if (EXT && ph == GAS && wet) f[EQ_OIL] = f[EQ_OIL] + iq_at(iq, N, Lay<EXT>::F_RV, c)[0] * surfaceVolume;
    }
    for (int e = 0; e < 3; ++e) storageOld[(size_t)c * 3 + e] = f[e];
}
void launch_drift_update(opmhip_ctx* c, double dt) {
    const int n = c->pat.Nb * 3;
    hipLaunchKernelGGL(k_drift_update, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, c->d_b, dt, c->asmb.d_drift);
}

// ============================== small permutation helpers =====================================================
__global__ void k_cellvec_to_internal_u8(int Nb, const int* __restrict__ fromOrder, const unsigned char* __restrict__ nat, unsigned char* __restrict__ internal) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < Nb) internal[p] = nat[fromOrder[p]];
}
__global__ void k_cellvec_to_natural_u8(int Nb, const int* __restrict__ toOrder, const unsigned char* __restrict__ internal, unsigned char* __restrict__ nat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Nb) nat[i] = internal[toOrder[i]];
}
__global__ void k_iq_to_natural(int Nb, int IQS, const int* __restrict__ toOrder, const double* __restrict__ internal, double* __restrict__ nat) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)Nb * IQS) return;
    const int i = (int)(e / IQS), q = (int)(e % IQS);
    nat[e] = internal[((size_t)(q >> 2) * Nb + toOrder[i]) * 4 + (q & 3)];   // field-major cache -> per-cell records
}
// records of n named cells out of the field-major cache (opmhip_get_iq_cells)
__global__ void k_iq_gather(int n, int IQS, int Nb, const int* __restrict__ pos, const double* __restrict__ internal, double* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n * IQS) return;
    const int i = (int)(e / IQS), q = (int)(e % IQS);
    out[e] = internal[((size_t)(q >> 2) * Nb + pos[i]) * 4 + (q & 3)];
}
// source terms of n DISTINCT cells into the (zeroed) per-cell arrays (opmhip_set_source_cells)
__global__ void k_source_scatter(int n, const int* __restrict__ pos, const double* __restrict__ src, const double* __restrict__ dsrc,
                                 double* __restrict__ source, double* __restrict__ dsource) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * 12) return;
    const int i = e / 12, q = e % 12;
    if (q < 3) source[(size_t)pos[i] * 3 + q] = src[(size_t)i * 3 + q];
    else if (dsrc) dsource[(size_t)pos[i] * 9 + (q - 3)] = dsrc[(size_t)i * 9 + (q - 3)];
}
__global__ void k_unpermute_blocks(int nnzb, const int* __restrict__ nnzMap, const double* __restrict__ internal, double* __restrict__ nat) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)nnzb * BB) return;
    const int k = (int)(e / BB), q = (int)(e % BB);
    nat[(size_t)nnzMap[k] * BB + q] = internal[e];
}

// ============================== launchers ====================================================================
static inline int cdiv(size_t a, size_t b) { return (int)((a + b - 1) / b); }
static Tables tables_of(const opmhip_ctx* c) {
    return Tables{c->asmb.d_tab_dbl, c->asmb.d_tab_idx, c->asmb.rock_pref, c->asmb.rock_cr, c->asmb.tab_ndbl, c->asmb.tab_nidx, c->asmb.rock_desc, c->asmb.num_rock};
}
static CellStatic cells_of(const opmhip_ctx* c) {
    return CellStatic{c->asmb.d_poro, c->asmb.d_volume, c->asmb.d_depth, c->asmb.d_rsmax, c->asmb.d_pvtnum, c->asmb.d_satnum,
                      c->asmb.d_rvmax, c->asmb.d_overburden, c->asmb.d_rocknum, c->asmb.d_pcw, c->asmb.d_minpo, c->asmb.d_maxso, c->asmb.vap1, c->asmb.vap2,
                      c->asmb.d_maxsw, c->asmb.d_sw0, c->asmb.d_wcdesc, c->asmb.d_wcdata, c->asmb.d_eps, c->asmb.epscfg,
                      c->asmb.d_hyst, c->asmb.d_imbnum, c->asmb.d_eps_imb, c->asmb.hyst_model, c->asmb.d_invb, c->pat.Nloc};
}
// sw_ow / sw_go (device, internal order, Nloc each) or NULL: from the intensive quantities of the state at hand
void launch_hyst_update(opmhip_ctx* c, const double* d_sw_ow, const double* d_sw_go) {
    const int N = c->pat.Nloc;   // all cells, the ghost ones too ("to avoid desynchronization of the processes", :2608-2609)
    hipLaunchKernelGGL(k_hyst_update, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, tables_of(c), cells_of(c), c->asmb.d_iq, c->asmb.d_hyst, d_sw_ow, d_sw_go);
}
// the context's record layout: extended when the fluid has PVTG or ROCKTAB tables
#define OPMHIP_LAYOUT(c, call_base, call_ext) do { if ((c)->asmb.ext) { call_ext; } else { call_base; } } while (0)

void launch_iq_update(opmhip_ctx* c) {
    const int Nb = c->pat.Nloc;  // ghost cells too: their intensive quantities feed the faces towards them
    const int ps = prof_begin(c, PROF_IQ_UPDATE);
    OPMHIP_LAYOUT(c,
        hipLaunchKernelGGL(k_iq_update<false>, dim3(cdiv(Nb, 256)), dim3(256), 0, c->stream, 0, Nb, tables_of(c), cells_of(c), c->asmb.d_pv, c->asmb.d_meaning, c->asmb.d_iq),
        hipLaunchKernelGGL(k_iq_update<true>, dim3(cdiv(Nb, 256)), dim3(256), 0, c->stream, 0, Nb, tables_of(c), cells_of(c), c->asmb.d_pv, c->asmb.d_meaning, c->asmb.d_iq));
    prof_end(c, ps);
}
// after a Newton update in a decomposed run: ghost primary variables come from their owners, then their IQs are redone
int launch_ghost_refresh(opmhip_ctx* c) {
    if (c->comm.nranks <= 1 || c->pat.Nghost == 0) return OPMHIP_SUCCESS;
    int rc;
    if ((rc = comm_halo_f64(c, c->asmb.d_pv, 3))) return rc;
    if ((rc = comm_halo_u8(c, c->asmb.d_meaning))) return rc;
    const int Nb = c->pat.Nb, Nloc = c->pat.Nloc;
    OPMHIP_LAYOUT(c,
        hipLaunchKernelGGL(k_iq_update<false>, dim3(cdiv(Nloc - Nb, 256)), dim3(256), 0, c->stream, Nb, Nloc, tables_of(c), cells_of(c), c->asmb.d_pv, c->asmb.d_meaning, c->asmb.d_iq),
        hipLaunchKernelGGL(k_iq_update<true>, dim3(cdiv(Nloc - Nb, 256)), dim3(256), 0, c->stream, Nb, Nloc, tables_of(c), cells_of(c), c->asmb.d_pv, c->asmb.d_meaning, c->asmb.d_iq));
    return OPMHIP_SUCCESS;
}
void launch_newton_update(opmhip_ctx* c, const double* d_dx, double relax) {
    const int Nb = c->pat.Nb;
    (void)hipMemsetAsync(c->asmb.d_nswitched, 0, sizeof(int), c->stream);
    const int ps = prof_begin(c, PROF_IQ_UPDATE);
    OPMHIP_LAYOUT(c,
        hipLaunchKernelGGL(k_newton_update<false>, dim3(cdiv(Nb, 256)), dim3(256), 0, c->stream, Nb, tables_of(c), cells_of(c), d_dx, relax, c->asmb.d_pv,
                           c->asmb.d_meaning, c->asmb.d_wasSwitched, c->asmb.d_iq, c->asmb.d_nswitched),
        hipLaunchKernelGGL(k_newton_update<true>, dim3(cdiv(Nb, 256)), dim3(256), 0, c->stream, Nb, tables_of(c), cells_of(c), d_dx, relax, c->asmb.d_pv,
                           c->asmb.d_meaning, c->asmb.d_wasSwitched, c->asmb.d_iq, c->asmb.d_nswitched));
    prof_end(c, ps);
}
void launch_assemble(opmhip_ctx* c, double dt, int iteration) {
    EntryStatic ES{c->asmb.d_trans, c->asmb.d_area, c->asmb.d_thpres};
    const int ps = prof_begin(c, PROF_ASSEMBLE);
    const double* drift = c->asmb.drift_enabled ? c->asmb.d_drift : (const double*)nullptr;
    const int grid = c->asmb.nsched;
    if (iteration == 0 && c->asmb.storage_frozen) iteration = 1;   // the old time level's storage was formed by opmhip_begin_time_step
    OPMHIP_LAYOUT(c,
        hipLaunchKernelGGL(k_assemble<false>, dim3(grid), dim3(ASM_THREADS), 0, c->stream, c->asmb.nsched, reinterpret_cast<const int4*>(c->asmb.d_asm_sched), reinterpret_cast<const int2*>(c->asmb.d_asm_desc), 0, ES,
                           cells_of(c), c->asmb.d_iq, c->asmb.d_storageOld, c->asmb.d_source, c->asmb.d_dsource, drift, c->asmb.max_compensation, dt, iteration, c->d_A, c->d_b),
        hipLaunchKernelGGL(k_assemble<true>, dim3(grid), dim3(ASM_THREADS), 0, c->stream, c->asmb.nsched, reinterpret_cast<const int4*>(c->asmb.d_asm_sched), reinterpret_cast<const int2*>(c->asmb.d_asm_desc), c->asmb.wet_gas ? 1 : 0, ES,
                           cells_of(c), c->asmb.d_iq, c->asmb.d_storageOld, c->asmb.d_source, c->asmb.d_dsource, drift, c->asmb.max_compensation, dt, iteration, c->d_A, c->d_b));
    prof_end(c, ps);
}
void launch_last_rs_rv(opmhip_ctx* c) {
    const AsmDev& A = c->asmb;
    const int N = c->pat.Nloc;
    OPMHIP_LAYOUT(c, hipLaunchKernelGGL(k_last_rs_rv<false>, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, A.d_pvtnum, A.d_drsdt_all, A.d_iq, A.drsdt_on ? A.d_lastRs : nullptr, (double*)nullptr),
                  hipLaunchKernelGGL(k_last_rs_rv<true>, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, A.d_pvtnum, A.d_drsdt_all, A.d_iq, A.drsdt_on ? A.d_lastRs : nullptr, A.drvdt_on ? A.d_lastRv : nullptr));
}
void launch_set_limits(opmhip_ctx* c, double dt) {
    const AsmDev& A = c->asmb;
    const int N = c->pat.Nloc;
    hipLaunchKernelGGL(k_set_limits, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, dt, A.d_pvtnum, A.drsdt_on ? A.d_drsdt : nullptr, A.drvdt_on ? A.d_drvdt : nullptr,
                       A.d_lastRs, A.d_lastRv, A.d_rsmax, A.d_rvmax);
}
void launch_min_pressure(opmhip_ctx* c, bool init) {
    const int N = c->pat.Nloc;
    hipLaunchKernelGGL(k_min_pressure, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, init ? 1 : 0, c->asmb.d_iq, c->asmb.d_minpo);
}
void launch_max_oil_saturation(opmhip_ctx* c, bool init) {
    const int N = c->pat.Nloc;
    hipLaunchKernelGGL(k_max_oil_saturation, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, init ? 1 : 0, c->asmb.d_iq, c->asmb.d_maxso);
}
void launch_max_water_saturation(opmhip_ctx* c, bool init) {
    const Pattern& P = c->pat;
    const int N = P.Nloc;
    // eclproblem.hh:2150: the cell with index 1 takes over the stored maximum of the cell with index 0 (natural numbering)
    if (!init && N > 1) hipLaunchKernelGGL(k_copy_entry, dim3(1), dim3(1), 0, c->stream, c->asmb.d_maxsw, P.toOrder[1], P.toOrder[0]);
    hipLaunchKernelGGL(k_max_water_saturation, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, init ? 1 : 0, c->asmb.d_iq, c->asmb.d_maxsw, c->asmb.d_sw0);
}
void launch_storage_old(opmhip_ctx* c) {
    const AsmDev& A = c->asmb;
    const int Nb = c->pat.Nb, N = c->pat.Nloc;
    OPMHIP_LAYOUT(c, hipLaunchKernelGGL(k_storage_old<false>, dim3((Nb + 255) / 256), dim3(256), 0, c->stream, Nb, N, 0, A.d_iq, A.d_storageOld),
                  hipLaunchKernelGGL(k_storage_old<true>, dim3((Nb + 255) / 256), dim3(256), 0, c->stream, Nb, N, A.wet_gas ? 1 : 0, A.d_iq, A.d_storageOld));
}

// BlackoilModelEbos::relativeChange (flow/BlackoilModelEbos.hpp:431-510): sum over the owned cells of (p_new - p_old)^2 and of
// the squared saturation changes, over the sum of p_new^2 and the squared new saturations - what the PID time-step control
// calls the error of a time step (timestepping/TimeStepControl.cpp:127-161).  solution(0) = the state, solution(1) = the
// time level opmhip_advance_time_level kept.  Per cell the terms are added in the reference's order (pressure, then water,
// oil, gas); the cells are summed by a fixed tree (a sequential sum over 10^6 cells is not what a GPU does), so the result
// agrees with the reference's sequential sum to rounding, not to the bit.
constexpr int RC_PARTS = 256;
__global__ __launch_bounds__(256) void k_relative_change_part(int Nb, const double* __restrict__ pv, const unsigned char* __restrict__ mg,
                                                              const double* __restrict__ pvOld, const unsigned char* __restrict__ mgOld,
                                                              double* __restrict__ part) {
    __shared__ double sd[256], sn[256];
    double delta = 0.0, denom = 0.0;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < Nb; c += RC_PARTS * 256) {
        const double* a = &pv[(size_t)c * 3];
        const double* b = &pvOld[(size_t)c * 3];
        double sN[3] = {0.0, 0.0, 0.0}, sO[3] = {0.0, 0.0, 0.0};
        double oilN = 1.0, oilO = 1.0;
        sN[WATER] = a[0]; oilN -= sN[WATER];
        if (mg[c] == OPMHIP_SW_PO_SG) { sN[GAS] = a[2]; oilN -= sN[GAS]; }
        sN[OIL] = oilN;
        const double tmp = a[1] - b[1];
        double d = tmp * tmp, q = a[1] * a[1];
        sO[WATER] = b[0]; oilO -= sO[WATER];
        if (mgOld[c] == OPMHIP_SW_PO_SG) { sO[GAS] = b[2]; oilO -= sO[GAS]; }
        sO[OIL] = oilO;
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const double t = sN[ph] - sO[ph];
            d += t * t;
            q += sN[ph] * sN[ph];
        }
        delta += d;
        denom += q;
    }
    sd[threadIdx.x] = delta; sn[threadIdx.x] = denom;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) { sd[threadIdx.x] += sd[threadIdx.x + h]; sn[threadIdx.x] += sn[threadIdx.x + h]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x] = sd[0]; part[RC_PARTS + blockIdx.x] = sn[0]; }
}
__global__ __launch_bounds__(64) void k_relative_change_final(const double* __restrict__ part, double* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double d = 0.0, q = 0.0;
    for (int i = 0; i < RC_PARTS; ++i) { d += part[i]; q += part[RC_PARTS + i]; }
    out[0] = d; out[1] = q;
}
// -> d_rc[2 * RC_PARTS .. + 2): (resultDelta, resultDenom), summed over the ranks of a decomposed run (gridView.comm().sum)
int launch_relative_change(opmhip_ctx* c) {
    const AsmDev& A = c->asmb;
    double* part = A.d_rc;
    double* out = A.d_rc + 2 * RC_PARTS;
    hipLaunchKernelGGL(k_relative_change_part, dim3(RC_PARTS), dim3(256), 0, c->stream, c->pat.Nb, A.d_pv, A.d_meaning, A.d_pv_prev, A.d_meaning_prev, part);
    hipLaunchKernelGGL(k_relative_change_final, dim3(1), dim3(64), 0, c->stream, part, out);
    if (c->comm.nranks > 1) return comm_allreduce(c, out, 2, 0);
    return OPMHIP_SUCCESS;
}

int launch_convergence(opmhip_ctx* c, double dt, double tol_cnv) {
    const int Nb = c->pat.Nb, nb = cdiv(Nb, 256);
    const int ps = prof_begin(c, PROF_CONVERGENCE);
    hipLaunchKernelGGL(k_conv_pass1, dim3(nb), dim3(256), 0, c->stream, Nb, cells_of(c), c->d_b, c->asmb.d_conv_part);
    const bool dd = c->comm.nranks > 1;
    hipLaunchKernelGGL(k_conv_final1, dim3(1), dim3(256), 0, c->stream, nb, dd ? 0 : Nb, c->asmb.d_conv_part, c->asmb.d_conv_out);
    if (dd) {
        hipLaunchKernelGGL(k_conv_pack, dim3(1), dim3(64), 0, c->stream, c->asmb.d_conv_out, c->comm.d_red);
        int rc;  // a failed reduction must not be read as "converged": report it
        if ((rc = comm_allreduce(c, c->comm.d_red, 7, 0))) return rc;
        if ((rc = comm_allreduce(c, c->comm.d_red + 8, 3, 1))) return rc;
        hipLaunchKernelGGL(k_conv_unpack, dim3(1), dim3(64), 0, c->stream, c->comm.d_red, (double)c->comm.global_cells, c->asmb.d_conv_out);
    }
    hipLaunchKernelGGL(k_conv_pass2, dim3(nb), dim3(256), 0, c->stream, Nb, cells_of(c), c->d_b, c->asmb.d_conv_out, dt, tol_cnv, c->asmb.d_conv_part);
    hipLaunchKernelGGL(k_conv_final2, dim3(1), dim3(256), 0, c->stream, nb, c->asmb.d_conv_part, c->asmb.d_conv_out);
    if (dd) {
        const int rc = comm_allreduce(c, c->asmb.d_conv_out + 10, 1, 0);
        if (rc) return rc;
    }
    prof_end(c, ps);
    return OPMHIP_SUCCESS;
}
void launch_u8_to_internal(opmhip_ctx* c, const unsigned char* nat, unsigned char* internal) {
    hipLaunchKernelGGL(k_cellvec_to_internal_u8, dim3(cdiv(c->pat.Nloc, 256)), dim3(256), 0, c->stream, c->pat.Nloc, c->pat.d_fromOrder, nat, internal);
}
void launch_u8_to_natural(opmhip_ctx* c, const unsigned char* internal, unsigned char* nat) {
    hipLaunchKernelGGL(k_cellvec_to_natural_u8, dim3(cdiv(c->pat.Nloc, 256)), dim3(256), 0, c->stream, c->pat.Nloc, c->pat.d_toOrder, internal, nat);
}
void launch_iq_to_natural(opmhip_ctx* c, double* d_nat) {
    const size_t n = (size_t)c->pat.Nloc * iq_doubles_per_cell(c);
    hipLaunchKernelGGL(k_iq_to_natural, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, c->pat.Nloc, iq_doubles_per_cell(c), c->pat.d_toOrder, c->asmb.d_iq, d_nat);
}
void launch_iq_gather(opmhip_ctx* c, int n, const int* d_pos, double* d_out) {
    const int IQS = iq_doubles_per_cell(c);
    hipLaunchKernelGGL(k_iq_gather, dim3(cdiv((size_t)n * IQS, 256)), dim3(256), 0, c->stream, n, IQS, c->pat.Nloc, d_pos, c->asmb.d_iq, d_out);
}
void launch_source_scatter(opmhip_ctx* c, int n, const int* d_pos, const double* d_src, const double* d_dsrc) {
    hipLaunchKernelGGL(k_source_scatter, dim3(cdiv((size_t)n * 12, 256)), dim3(256), 0, c->stream, n, d_pos, d_src, d_dsrc, c->asmb.d_source, c->asmb.d_dsource);
}
void launch_unpermute_blocks(opmhip_ctx* c, const double* internal, double* nat) {
    const size_t n = (size_t)c->pat.nnzb * BB;
    hipLaunchKernelGGL(k_unpermute_blocks, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, c->pat.nnzb, c->pat.d_nnzMap, internal, nat);
}
int iq_doubles_per_cell(const opmhip_ctx* c) { return c->asmb.ext ? Lay<true>::IQS : Lay<false>::IQS; }

}  // namespace opmhip
