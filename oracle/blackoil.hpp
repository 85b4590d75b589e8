// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp).
// CPU restatement of the per-Newton-iteration assembly path of Flow's three-phase black-oil model
// (live oil + dry gas + water, no extensions): intensive quantities, TPFA flux, storage, element-centred AD
// linearisation into block-CSR, convergence norms and the Newton update with primary-variable switching.
//
// In-tree sources followed line by line:
//   ebos/eclfluxmodule.hh:212-357                     (calculateGradients_: TPFA flux, upwinding, THPRES)
//   ebos/eclproblem.hh:1430-1486, 1682-1765, 1823-1845  (porosity, depth, rock compressibility, Rs caps, source)
//   opm/simulators/flow/BlackoilModelEbos.hpp:572-904 (convergence), :549-563 (updateSolution)
// Out-of-tree (opm-models), restated from the public 2021.10 sources as recalled,
// UNVERIFIED vs upstream: BlackOilIntensiveQuantities::update, BlackOilLocalResidual::{computeStorage,
// computeFlux}, FvBaseLocalResidual::eval, FvBaseAdLocalLinearizer, FvBaseLinearizer, BlackOilNewtonMethod::update_,
// BlackOilPrimaryVariables::adaptPrimaryVariables.
//
// PARITY UNPINNED for the assembly: the reference tree holds no numeric test of residual, Jacobian, intensive
// quantities or flux (SURVEY.md §4 "What is not tested").  Independent evidence kept in tests/: finite-difference
// check of the Jacobian, mass conservation of the fluxes, and the Norne PVT points for the oil PVT.
//
// Index conventions (BlackOilIndices / BlackOilFluidSystem, recalled): phases water 0, oil 1, gas 2;
// equations (components) oil 0, water 1, gas 2; primary variables Sw 0, p 1, {Sg | Rs} 2.
#pragma once
#include <cstdint>
#include <limits>

#include "fluid.hpp"
#include "linalg.hpp"

namespace orc {

enum { WATER = 0, OIL = 1, GAS = 2 };
enum { EQ_OIL = 0, EQ_WATER = 1, EQ_GAS = 2 };
enum { PV_SW = 0, PV_P = 1, PV_X = 2 };
enum Meaning : uint8_t { Sw_po_Sg = 0, Sw_po_Rs = 1, Sw_pg_Rv = 2 };  // PrimaryVariables::PrimaryVarsMeaning (Sw_pg_Rv: wet gas, oil phase absent)

constexpr double GRAVITY = 9.80665;

// What BlackOilIntensiveQuantities caches per cell (ebos/eclproblem.hh:456-458) and the flux/storage code reads.
struct IQ {
    Ev S[3], p[3], invB[3], mob[3], rho[3];
    Ev Rs, Rv, poro;
    double refPoro = 0.0;
};

struct Problem {
    Bcrs pat;  // pattern only (val unused)
    std::vector<int> rowOf, transIdx;  // row of each entry; transIdx[k] = entry (J,I) for entry k = (I,J)
    std::vector<double> trans, area, thpres;  // per entry (0 on the diagonal); thpres may be empty
    std::vector<double> poro, volume, depth;  // per cell
    std::vector<int> pvtnum, satnum;          // per cell, 0-based
    std::vector<double> rsMax;                // per cell; empty = no DRSDT limit (eclproblem.hh:1711-1732)
    std::vector<double> rvMax;                // per cell; empty = no DRVDT limit (maxOilVaporizationFactor, eclproblem.hh:1734-1754)
    std::vector<int> rockNum;                 // per cell rock-table index (rockTableIdx_, eclproblem.hh:1943-1945); empty = table 0
    std::vector<double> overburden;           // per cell overburden pressure (eclproblem.hh:1954-1955); empty = none
    std::vector<double> minOilPressure;       // per cell (ROCKCOMP IRREVERS, eclproblem.hh:1948-1952, 2172-2197); empty = reversible
    // VAPPARS (eclproblem.hh:1682-1688, 2110-2141): per cell the largest oil saturation seen at the start of a time step; empty =
    // keyword not in force (maxOilSaturation() = 0).  vapPar1 acts on RvSat, vapPar2 on RsSat (the PVT classes' 5-argument
    // saturatedGasDissolutionFactor / saturatedOilVaporizationFactor of opm-material, absent: restated, UNVERIFIED)
    std::vector<double> maxOilSaturation;
    double vapPar1 = 0.0, vapPar2 = 0.0;
    // water-induced rock compaction (ROCKCOMP + ROCK2D / ROCK2DTR / ROCKWNOD; tables built in ebos/eclgenericproblem.cc:186-240,
    // read in eclproblem.hh:1962-1967, 2001-2005): per rock region a 2-D table of pore-volume (and transmissibility) multipliers
    // over (effective oil pressure, SwMax - Sw_initial), SwMax = max(Sw, largest Sw seen at the start of a time step).  The
    // table class (UniformXTabulated2DFunction of opm-material, default = vertical interpolation, extrapolating) is absent from
    // the reference tree: Tab2D with guide 0, UNVERIFIED.  rock2dTrans may be empty (no ROCK2DTR: multiplier 1).
    std::vector<Tab2D> rock2dPoro, rock2dTrans;
    std::vector<double> maxWaterSaturation, initialSw;   // per cell; empty = no water-induced compaction
    // per cell scaled maximum of the oil-water capillary pressure (the deck's PCW, or what SWATINIT made of it:
    // ebos/equil/initstateequil.hh:1330-1343 -> EclMaterialLawManager::applySwatinit); empty = the tables' own.
    // UNVERIFIED against upstream sources: opm-material (EclEpsTwoPhaseLaw, EclEpsScalingPoints) is not in the reference tree;
    // restated from its published form - with enablePcScaling and no saturation scaling the scaled curve is
    // pcnw(Sw) = table(Sw) * alpha, alpha = scaledMaxPcnw / unscaledMaxPcnw (1 when the two are equal), unscaledMaxPcnw =
    // the SWOF pcow column's first entry.  Pinned by the reference's own numbers for equil_capillary_swatinit.DATA
    // (tests/test_equil.cc:1076-1091) through tests/test_equil.py::test_swatinit_deck.
    std::vector<double> pcw;
    // saturation end-point scaling (ENDSCALE family): per cell the scaled end points (EPS_COUNT doubles, oracle/fluid.hpp) handed
    // to EclEpsTwoPhaseLaw by the material-law manager (ebos/eclproblem.hh:1490-1498); empty = none.  UNVERIFIED (opm-material absent).
    std::vector<EpsPoints> eps;
    EpsConfig epsCfg;
    // relative-permeability hysteresis (SATOPTS HYSTER, EHYSTR item 2 = hystKrModel 0 | 1; oracle/fluid.hpp HystCell): per cell the
    // imbibition saturation region (IMBNUM), the turning points of its two two-phase systems and, with ENDSCALE, the scaled end
    // points of the imbibition curves (ISWL ... ; empty with ENDSCALE on: the imbibition tables' own).  hyst empty = not in force.
    int hystKrModel = -1;
    std::vector<int> imbnum;
    std::vector<HystCell> hyst;
    std::vector<EpsPoints> epsImb;
    Fluid fluid;
    void finish() {
        const int Nb = pat.Nb;
        rowOf.resize(pat.nnzb());
        for (int i = 0; i < Nb; ++i)
            for (int k = pat.rowptr[i]; k < pat.rowptr[i + 1]; ++k) rowOf[k] = i;
        transIdx.assign(pat.nnzb(), -1);
        for (int k = 0; k < pat.nnzb(); ++k) {
            const int i = rowOf[k], j = pat.col[k];
            for (int q = pat.rowptr[j]; q < pat.rowptr[j + 1]; ++q)
                if (pat.col[q] == i) transIdx[k] = q;
        }
    }
};

// LiveOilPvt::saturatedGasDissolutionFactor(region, T, p, So, maxSo) / WetGasPvt::saturatedOilVaporizationFactor: the factor
// VAPPARS puts on the saturated Rs (vapPar2) / Rv (vapPar1) where the oil saturation is below the largest one seen
template <class E> inline E vappars_factor(const E& So, const E& SoMaxIn, double vapPar) {
    const E maxOilSaturation = min(SoMaxIn, E(1.0));
    if (vapPar > 0.0 && value(maxOilSaturation) > 0.01 && value(So) < value(maxOilSaturation)) {
        const E S = max(So, E(0.001));
        return max(E(1e-3), pow(S / maxOilSaturation, vapPar));
    }
    return E(1.0);
}

// ---- BlackOilIntensiveQuantities::update ------------------------------------------------------
// E = Ev: the focus cell (derivatives w.r.t. its own primary variables); E = double: values only.
template <class E>
struct IQT {
    E S[3], p[3], invB[3], mob[3], rho[3], Rs, Rv, poro;
    E tmult;   // rockCompTransMultiplier (eclproblem.hh:1975-2007); 1 without ROCKTAB
    double refPoro;
};
template <class E> inline E mkvar(double x, int idx);
template <> inline Ev mkvar<Ev>(double x, int idx) { return Ev::variable(x, idx); }
template <> inline double mkvar<double>(double x, int) { return x; }

template <class E>
void update_iq(const Problem& P, int cell, const double* pv, uint8_t meaning, IQT<E>& q) {
    const Fluid& F = P.fluid;
    const int pr = P.pvtnum.empty() ? 0 : P.pvtnum[cell], sr = P.satnum.empty() ? 0 : P.satnum[cell];
    const double RsMax = P.rsMax.empty() ? std::numeric_limits<double>::max() / 2.0 : P.rsMax[cell];
    const double RvMax = P.rvMax.empty() ? std::numeric_limits<double>::max() / 2.0 : P.rvMax[cell];
    const bool wet = F.hasWetGas;   // FluidSystem::enableVaporizedOil()
    const E Sw = mkvar<E>(pv[PV_SW], PV_SW);
    E Sg = E(0.0);
    if (meaning == Sw_po_Sg) Sg = mkvar<E>(pv[PV_X], PV_X);
    else if (meaning == Sw_pg_Rv) Sg = 1.0 - Sw;   // the oil phase is absent
    const E So = 1.0 - Sw - Sg;
    q.S[WATER] = Sw; q.S[GAS] = Sg; q.S[OIL] = So;
    E pC[3];
    const bool scaled = !P.eps.empty();
    if (scaled) F.sat[sr].capillaryPressuresEps(pC, Sw, Sg, P.eps[cell], P.epsCfg);
    else F.sat[sr].capillaryPressures(pC, Sw, Sg);
    if (!scaled && !P.pcw.empty()) {
        const double scaledMax = P.pcw[cell], tableMax = F.sat[sr].pcow.y.front();
        const double alpha = (scaledMax == tableMax) ? 1.0 : scaledMax / tableMax;
        pC[0] = pC[0] * alpha;
    }
    if (meaning == Sw_pg_Rv) {   // the pressure primary variable is the GAS pressure
        const E pg = mkvar<E>(pv[PV_P], PV_P);
        for (int ph = 0; ph < 3; ++ph) q.p[ph] = pg + (pC[ph] - pC[GAS]);
    } else {
        const E po = mkvar<E>(pv[PV_P], PV_P);
        for (int ph = 0; ph < 3; ++ph) q.p[ph] = po + (pC[ph] - pC[OIL]);
    }
    if (!P.hyst.empty()) {
        const SatFunc& imb = F.sat[P.imbnum[cell]];
        EpsPoints own;
        const EpsPoints* scI = nullptr;
        if (scaled) { if (P.epsImb.empty()) { own = imb.unscaled; scI = &own; } else scI = &P.epsImb[cell]; }
        F.sat[sr].relativePermeabilitiesHyst(q.mob, Sw, Sg, P.hyst[cell], P.hystKrModel, imb, scaled ? &P.eps[cell] : nullptr, scI, P.epsCfg);
    } else if (scaled) F.sat[sr].relativePermeabilitiesEps(q.mob, Sw, Sg, P.eps[cell], P.epsCfg);
    else F.sat[sr].relativePermeabilities(q.mob, Sw, Sg);
    // SoMax = max(So, problem.maxOilSaturation) ; the latter is 0 without VAPPARS (eclproblem.hh:1682-1688), and without
    // VAPPARS the saturated Rs / Rv do not depend on it
    const bool vap = !P.maxOilSaturation.empty();
    const E SoMax = vap ? max(So, E(P.maxOilSaturation[cell])) : So;
    auto rs_sat = [&]() { E t = F.oil[pr].rsSat(q.p[OIL]); if (vap) t = t * vappars_factor(So, SoMax, P.vapPar2); return t; };
    auto rv_sat = [&]() { E t = F.wetGas[pr].rvSat(q.p[GAS]); if (vap) t = t * vappars_factor(So, SoMax, P.vapPar1); return t; };
    if (meaning == Sw_po_Sg) {
        const E RsSat = rs_sat();
        q.Rs = min(E(RsMax), RsSat);
        if (wet) { const E RvSat = rv_sat(); q.Rv = min(E(RvMax), RvSat); }
        else q.Rv = E(0.0);
    } else if (meaning == Sw_po_Rs) {
        const E Rs = mkvar<E>(pv[PV_X], PV_X);
        q.Rs = min(E(RsMax), Rs);
        // the gas phase is not present, but its "composition" is needed for the gravity correction term
        if (wet) { const E RvSat = rv_sat(); q.Rv = min(E(RvMax), RvSat); }
        else q.Rv = E(0.0);
    } else {
        const E Rv = mkvar<E>(pv[PV_X], PV_X);
        q.Rv = min(E(RvMax), Rv);
        const E RsSat = rs_sat();   // the oil phase is not present: same remark
        q.Rs = min(E(RsMax), RsSat);
    }
    // inverse formation volume factors and viscosities (BlackOilFluidSystem::inverseFormationVolumeFactor / viscosity)
    {
        const bool saturated = value(q.S[GAS]) > 0.0 && value(q.Rs) >= (1.0 - 1e-10) * F.oil[pr].rsSat(value(q.p[OIL]));
        E mu;
        q.invB[WATER] = F.water[pr].invB(q.p[WATER]);
        mu = F.water[pr].viscosity(q.p[WATER]);
        q.mob[WATER] = q.mob[WATER] / mu;
        if (saturated) { q.invB[OIL] = F.oil[pr].invBSat(q.p[OIL]); mu = F.oil[pr].viscositySat(q.p[OIL]); }
        else { q.invB[OIL] = F.oil[pr].invB(q.p[OIL], q.Rs); mu = F.oil[pr].viscosity(q.p[OIL], q.Rs); }
        q.mob[OIL] = q.mob[OIL] / mu;
        if (wet) {
            const bool gasSaturated = value(q.S[OIL]) > 0.0 && value(q.Rv) >= (1.0 - 1e-10) * F.wetGas[pr].rvSat(value(q.p[GAS]));
            if (gasSaturated) { q.invB[GAS] = F.wetGas[pr].invBSat(q.p[GAS]); mu = F.wetGas[pr].viscositySat(q.p[GAS]); }
            else { q.invB[GAS] = F.wetGas[pr].invB(q.p[GAS], q.Rv); mu = F.wetGas[pr].viscosity(q.p[GAS], q.Rv); }
        } else {
            q.invB[GAS] = F.gas[pr].invB(q.p[GAS]);
            mu = F.gas[pr].viscosity(q.p[GAS]);
        }
        q.mob[GAS] = q.mob[GAS] / mu;
    }
    const double* rr = &F.rhoRef[3 * pr];  // oil, water, gas
    q.rho[WATER] = q.invB[WATER] * rr[1];
    q.rho[GAS] = q.invB[GAS] * rr[2];
    if (wet) q.rho[GAS] = q.rho[GAS] + q.invB[GAS] * q.Rv * rr[0];   // vaporised oil
    q.rho[OIL] = q.invB[OIL] * rr[0];
    q.rho[OIL] = q.rho[OIL] + q.invB[OIL] * q.Rs * rr[2];
    // porosity with rock compressibility (BlackOilIntensiveQuantities; hooks ebos/eclproblem.hh:1454-1486)
    q.refPoro = P.poro[cell];
    q.poro = E(q.refPoro);
    if (F.rock_cr > 0.0) {
        const E x = F.rock_cr * (q.p[OIL] - F.rock_pref);
        q.poro = q.poro * (1.0 + x + 0.5 * x * x);
    }
    // rock compaction tables (ROCKTAB): rockCompPoroMultiplier / rockCompTransMultiplier, ebos/eclproblem.hh:1936-2007
    // (reversible form: no minOilPressure_; overburden pressure subtracted when given)
    q.tmult = E(1.0);
    if (!F.rockTab.empty()) {
        const RockTab& RT = F.rockTab[P.rockNum.empty() ? 0 : P.rockNum[cell]];
        E effectiveOilPressure = q.p[OIL];
        if (!P.minOilPressure.empty()) effectiveOilPressure = min(q.p[OIL], E(P.minOilPressure[cell]));   // the pore space change is irreversible
        if (!P.overburden.empty()) effectiveOilPressure = effectiveOilPressure - P.overburden[cell];
        q.poro = q.poro * RT.poroMult.eval(effectiveOilPressure);
        q.tmult = RT.transMult.eval(effectiveOilPressure);
    } else if (!P.rock2dPoro.empty()) {   // water compaction (eclproblem.hh:1962-1967, 2001-2005): taken when there is no ROCKTAB table
        const int tableIdx = P.rockNum.empty() ? 0 : P.rockNum[cell];
        E effectiveOilPressure = q.p[OIL];
        if (!P.minOilPressure.empty()) effectiveOilPressure = min(q.p[OIL], E(P.minOilPressure[cell]));
        if (!P.overburden.empty()) effectiveOilPressure = effectiveOilPressure - P.overburden[cell];
        const E SwMax = max(q.S[WATER], E(P.maxWaterSaturation[cell]));
        const E SwDeltaMax = SwMax - P.initialSw[cell];
        q.poro = q.poro * P.rock2dPoro[tableIdx].eval(effectiveOilPressure, SwDeltaMax);
        if (!P.rock2dTrans.empty()) q.tmult = P.rock2dTrans[tableIdx].eval(effectiveOilPressure, SwDeltaMax);
    }
}

// ---- BlackOilLocalResidual::computeStorage (App. B.4): surface volumes per pore volume ---------------------------
template <class E>
void compute_storage(const IQT<E>& q, E st[3], bool wet = false) {
    st[0] = st[1] = st[2] = E(0.0);
    static const int comp[3] = {EQ_WATER, EQ_OIL, EQ_GAS};
    for (int ph = 0; ph < 3; ++ph) {
        const E surfaceVolume = q.S[ph] * q.invB[ph] * q.poro;
        st[comp[ph]] = st[comp[ph]] + surfaceVolume;
        if (ph == OIL) st[EQ_GAS] = st[EQ_GAS] + q.Rs * surfaceVolume;
        if (ph == GAS && wet) st[EQ_OIL] = st[EQ_OIL] + q.Rv * surfaceVolume;   // vaporised oil
    }
}

// ---- EclTransExtensiveQuantities::calculateGradients_ + BlackOilLocalResidual::computeFlux ---------------------
// in: focus (interior) cell with derivatives, exterior cell values only.  out: flux[eq] leaving the interior cell
// through this face, already multiplied by the face area (FvBaseLocalResidual::evalFluxes).
inline void compute_face_flux(const IQT<Ev>& in, const IQT<double>& ex, double trans, double faceArea, double thpres,
                              double zIn, double zEx, double Vin, double Vex, int I, int J, Ev flux[3], bool wet = false) {
    flux[0] = flux[1] = flux[2] = Ev(0.0);
    const double distZ = zIn - zEx;
    static const int comp[3] = {EQ_WATER, EQ_OIL, EQ_GAS};
    for (int ph = 0; ph < 3; ++ph) {
        if (in.mob[ph].v <= 0.0 && ex.mob[ph] <= 0.0) continue;  // eclfluxmodule.hh:257-265
        const Ev rhoAvg = (in.rho[ph] + ex.rho[ph]) / 2.0;         // :269-271
        Ev pressureExterior = Ev(ex.p[ph]);
        pressureExterior += rhoAvg * (distZ * GRAVITY);           // :273-279
        Ev dp = pressureExterior - in.p[ph];                      // :281
        bool upIsInterior;                                        // :287-321
        if (dp.v > 0.0) upIsInterior = false;
        else if (dp.v < 0.0) upIsInterior = true;
        else if (Vin > Vex) upIsInterior = true;
        else if (Vin < Vex) upIsInterior = false;
        else upIsInterior = (I < J);
        if (std::fabs(dp.v) > thpres) {                           // :327-337
            if (dp.v < 0.0) dp = dp + thpres; else dp = dp - thpres;
        } else continue;
        Ev volumeFlux;                                            // :340-355, transMult = rockCompTransMultiplier of the upstream cell
        if (upIsInterior) volumeFlux = dp * in.mob[ph] * in.tmult * (-trans / faceArea);
        else volumeFlux = dp * (ex.mob[ph] * ex.tmult * (-trans / faceArea));
        // computeFlux / evalPhaseFluxes_: surface volume flux, plus dissolved gas carried by the oil phase
        Ev surf;
        if (upIsInterior) surf = in.invB[ph] * volumeFlux; else surf = ex.invB[ph] * volumeFlux;
        flux[comp[ph]] += surf;
        if (ph == OIL) {
            if (upIsInterior) flux[EQ_GAS] += in.Rs * surf; else flux[EQ_GAS] += ex.Rs * surf;
        } else if (ph == GAS && wet) {   // vaporised oil carried by the gas phase
            if (upIsInterior) flux[EQ_OIL] += in.Rv * surf; else flux[EQ_OIL] += ex.Rv * surf;
        }
    }
    for (int e = 0; e < 3; ++e) flux[e] *= faceArea;  // alpha = extrusionFactor (1) * face.area()
}

struct Model {
    Problem P;
    std::vector<double> pv;        // solution(0), Nb x 3
    std::vector<uint8_t> meaning;  // per cell
    std::vector<uint8_t> wasSwitched;
    std::vector<double> storageOld;  // cached storage of the old time level, Nb x 3 (eclproblem.hh:462-464)
    std::vector<double> source;      // per cell, 3 equations, surface m^3/s TOTAL for the cell (wells; default 0)
    std::vector<double> dsource;     // per cell 3x3: d(source)/d(primary variables of the cell)
    std::vector<IQT<Ev>> iqF;        // cached IQs with derivatives
    std::vector<IQT<double>> iqV;    // cached IQ values
    Bcrs J;
    std::vector<double> residual;
    // drift compensation (ebos/eclproblem.hh:1847-1875, on by default :496-498): what the last accepted time step left
    // unconverged, residual * dt per cell (:1126-1135, UseVolumetricResidual = false for Flow,
    // flow/BlackoilModelEbos.hpp:92-95), is subtracted from the source term of the following steps
    bool enableDriftCompensation = true;
    double maxCompensation = 10.0 * 1e-2;   // 10 * NewtonTolerance (eclproblem.hh:352-356, 1854)
    std::vector<double> drift;              // Nb x 3, zero until the first endTimeStep (:881-884)

    // DRSDT / DRVDT (eclproblem.hh:1711-1754, 2010-2107, eclgenericproblem.cc maxDRs_ = DRSDT * dt): rates per PVT region
    // [1/s], negative = no limit; drsdtAll: the OILVAP option (the limit binds all cells, not only those with free gas)
    std::vector<double> drsdt, drvdt;
    std::vector<int> drsdtAll;
    std::vector<double> lastRs, lastRv;
    bool storageFrozen = false;   // the old-time-level storage was formed by begin_time_step (recycleFirstIterationStorage() == false)
    bool limits_active() const { return !drsdt.empty() || !drvdt.empty(); }
    // updateCompositionChangeLimits_ (eclproblem.hh:2010-2107): from the state as it is (initial solution; end of a time step)
    void update_composition_change_limits() {
        const int Nb = P.pat.Nb;
        if (!drsdt.empty()) {
            lastRs.resize(Nb);
            for (int c = 0; c < Nb; ++c) {
                const int pr = P.pvtnum.empty() ? 0 : P.pvtnum[c];
                lastRs[c] = (drsdtAll[pr] || iqV[c].S[GAS] > 1e-7) ? iqV[c].Rs : std::numeric_limits<double>::infinity();
            }
        }
        if (!drvdt.empty()) {
            lastRv.resize(Nb);
            for (int c = 0; c < Nb; ++c) lastRv[c] = iqV[c].Rv;
        }
    }
    // maxGasDissolutionFactor / maxOilVaporizationFactor of time level 0 (maxD = rate * dt) or 1 (dt = 0)
    void set_limits_for(double dt) {
        const int Nb = P.pat.Nb;
        const double none = std::numeric_limits<double>::max() / 2.0;
        if (!drsdt.empty()) {
            P.rsMax.resize(Nb);
            for (int c = 0; c < Nb; ++c) {
                const int pr = P.pvtnum.empty() ? 0 : P.pvtnum[c];
                P.rsMax[c] = (drsdt[pr] < 0.0) ? none : lastRs[c] + drsdt[pr] * dt;
            }
        }
        if (!drvdt.empty()) {
            P.rvMax.resize(Nb);
            for (int c = 0; c < Nb; ++c) {
                const int pr = P.pvtnum.empty() ? 0 : P.pvtnum[c];
                P.rvMax[c] = (drvdt[pr] < 0.0) ? none : lastRv[c] + drvdt[pr] * dt;
            }
        }
    }
    // EclProblem::updateHysteresis_ (eclproblem.hh:2603-2626) -> EclMaterialLawManager::updateHysteresis -> EclDefaultMaterial::
    // updateHysteresis: every cell's two two-phase systems see the saturations of the state at hand
    bool update_hysteresis() {
        if (P.hyst.empty()) return false;
        const int Nb = P.pat.Nb;
        const bool scaled = !P.eps.empty();
        for (int c = 0; c < Nb; ++c) {
            const int sr = P.satnum.empty() ? 0 : P.satnum[c];
            const SatFunc& imb = P.fluid.sat[P.imbnum[c]];
            EpsPoints own;
            const EpsPoints* scI = nullptr;
            if (scaled) { if (P.epsImb.empty()) { own = imb.unscaled; scI = &own; } else scI = &P.epsImb[c]; }
            P.fluid.sat[sr].hystUpdate(P.hyst[c], iqV[c].S[OIL], iqV[c].S[GAS], imb, scaled ? &P.eps[c] : nullptr, scI, P.epsCfg);
        }
        return true;
    }
    // EclProblem::beginTimeStep, the per-cell part (eclproblem.hh:1042-1075): minimum oil pressure of irreversible
    // compaction, the DRSDT / DRVDT caps of a time step of size dt, intensive quantities; and, where the first iteration's
    // storage term cannot be recycled (:1758-1765), the old time level's storage with ITS caps (time index 1: no increment)
    void begin_time_step(double dt) {
        const int Nb = P.pat.Nb;
        if (!P.minOilPressure.empty())
            for (int c = 0; c < Nb; ++c) P.minOilPressure[c] = std::min(P.minOilPressure[c], iqV[c].p[OIL]);
        if (!P.maxOilSaturation.empty())   // updateMaxOilSaturation_ (eclproblem.hh:2110-2141)
            for (int c = 0; c < Nb; ++c) P.maxOilSaturation[c] = std::max(P.maxOilSaturation[c], iqV[c].S[OIL]);
        if (!P.maxWaterSaturation.empty()) {   // updateMaxWaterSaturation_ (eclproblem.hh:2144-2169)
            // :2150 reads `maxWaterSaturation_[/*timeIdx=*/1] = maxWaterSaturation_[/*timeIdx=*/0]` on the PER-CELL vector: cell 1
            // takes over cell 0's stored maximum before the loop.  Kept as the reference has it (same inputs, same results).
            if (Nb > 1) P.maxWaterSaturation[1] = P.maxWaterSaturation[0];
            for (int c = 0; c < Nb; ++c) P.maxWaterSaturation[c] = std::max(P.maxWaterSaturation[c], iqV[c].S[WATER]);
        }
        update_hysteresis();   // updateHysteresis_ (eclproblem.hh:1060, 2603-2626); the intensive quantities are redone below (:1064-1066)
        storageFrozen = false;
        if (limits_active()) {
            if (lastRs.empty() && lastRv.empty()) update_composition_change_limits();
            set_limits_for(0.0);
            update_all_iq();
            for (int I = 0; I < Nb; ++I) {
                Ev st[3];
                compute_storage(iqF[I], st, P.fluid.hasWetGas);
                for (int e = 0; e < 3; ++e) storageOld[(size_t)I * 3 + e] = st[e].v;
            }
            storageFrozen = true;
            set_limits_for(dt);
        }
        update_all_iq();
    }
    // EclProblem::endTimeStep (eclproblem.hh:1101-1135): DRSDT / DRVDT bookkeeping, then the drift part; call after an
    // ACCEPTED time step of size dt
    void end_time_step(double dt) {
        if (limits_active()) update_composition_change_limits();
        storageFrozen = false;
        if (!enableDriftCompensation) return;
        for (size_t i = 0; i < residual.size(); ++i) {
            drift[i] = residual[i];
            drift[i] *= dt;
        }
    }

    void init() {
        P.finish();
        const int Nb = P.pat.Nb;
        pv.assign((size_t)Nb * 3, 0.0);
        meaning.assign(Nb, Sw_po_Sg);
        wasSwitched.assign(Nb, 0);
        storageOld.assign((size_t)Nb * 3, 0.0);
        source.assign((size_t)Nb * 3, 0.0);
        dsource.assign((size_t)Nb * 9, 0.0);
        J = P.pat;
        J.val.assign((size_t)P.pat.nnzb() * BB, 0.0);
        residual.assign((size_t)Nb * 3, 0.0);
        drift.assign((size_t)Nb * 3, 0.0);
        iqF.resize(Nb);
        iqV.resize(Nb);
    }
    // invalidateAndUpdateIntensiveQuantities(0)  (BlackoilModelEbos.hpp:562)
    void update_all_iq() {
        const int Nb = P.pat.Nb;
#pragma omp parallel for schedule(static)
        for (int c = 0; c < Nb; ++c) {
            update_iq<Ev>(P, c, &pv[(size_t)c * 3], meaning[c], iqF[c]);
            update_iq<double>(P, c, &pv[(size_t)c * 3], meaning[c], iqV[c]);
        }
    }
    // FvBaseLinearizer::linearizeDomain with the AD local linearizer (App. B.2).  iteration == 0 with
    // recycleFirstIterationStorage() (eclproblem.hh:1758-1765): the old-time-level storage is the value of this
    // iteration's storage term.
    void assemble(double dt, int iteration) {
        const int Nb = P.pat.Nb;
        const Bcrs& A = P.pat;
#pragma omp parallel for schedule(static)
        for (int I = 0; I < Nb; ++I) {
            const IQT<Ev>& in = iqF[I];
            Ev R[3] = {Ev(0.0), Ev(0.0), Ev(0.0)};
            // flux terms first (FvBaseLocalResidual::eval), faces in ascending neighbour order
            for (int k = A.rowptr[I]; k < A.rowptr[I + 1]; ++k) {
                const int Jc = A.col[k];
                if (Jc == I) continue;
                Ev fl[3];
                compute_face_flux(in, iqV[Jc], P.trans[k], P.area[k], P.thpres.empty() ? 0.0 : P.thpres[k], P.depth[I],
                                  P.depth[Jc], P.volume[I], P.volume[Jc], I, Jc, fl, P.fluid.hasWetGas);
                for (int e = 0; e < 3; ++e) R[e] += fl[e];
                // residual[j] -= flux : block (J, I) = d(-flux)/d x_I
                double* blk = &J.val[(size_t)P.transIdx[k] * BB];
                for (int e = 0; e < 3; ++e)
                    for (int v = 0; v < 3; ++v) blk[e * 3 + v] = (Ev(0.0) - fl[e]).d[v];
            }
            // storage term, implicit Euler
            Ev st[3];
            compute_storage(in, st, P.fluid.hasWetGas);
            double* so = &storageOld[(size_t)I * 3];
            if (iteration == 0 && !storageFrozen)
                for (int e = 0; e < 3; ++e) so[e] = st[e].v;
            const double scvVolume = P.volume[I];
            for (int e = 0; e < 3; ++e) {
                Ev t = st[e] - so[e];
                t *= scvVolume / dt;
                R[e] += t;
            }
            // drift compensation of the source term (eclproblem.hh:1847-1875).  UNVERIFIED vs upstream: model.eqWeight()
            // (opm-models BlackOilModel::eqWeight) is taken as 1 for all three surface-volume equations; it only enters
            // the cap, which engages when a cell drifted by more than 10 % of its pore volume in one step
            double dofDriftRate[3] = {0.0, 0.0, 0.0};
            if (enableDriftCompensation) {
                const double poro = in.refPoro;
                for (int e = 0; e < 3; ++e) {
                    dofDriftRate[e] = drift[(size_t)I * 3 + e];
                    dofDriftRate[e] /= dt * scvVolume;
                }
                double totalDriftRate = 0.0;
                for (int e = 0; e < 3; ++e) totalDriftRate += std::fabs(dofDriftRate[e]) * dt * 1.0 / poro;
                if (totalDriftRate > maxCompensation)
                    for (int e = 0; e < 3; ++e) dofDriftRate[e] *= maxCompensation / totalDriftRate;
            }
            // source term: rate per volume (eclproblem.hh:1823-1845), then times volume again
            for (int e = 0; e < 3; ++e) {
                Ev s(source[(size_t)I * 3 + e]);
                for (int v = 0; v < 3; ++v) s.d[v] = dsource[(size_t)I * 9 + e * 3 + v];
                s /= scvVolume;
                if (enableDriftCompensation) s -= dofDriftRate[e];
                s *= scvVolume;
                R[e] -= s;
            }
            int kd = -1;
            for (int k = A.rowptr[I]; k < A.rowptr[I + 1]; ++k)
                if (A.col[k] == I) kd = k;
            double* blk = &J.val[(size_t)kd * BB];
            for (int e = 0; e < 3; ++e) {
                residual[(size_t)I * 3 + e] = R[e].v;
                for (int v = 0; v < 3; ++v) blk[e * 3 + v] = R[e].d[v];
            }
        }
    }

    // BlackoilModelEbos::localConvergenceData + computeCnvErrorPv + getReservoirConvergence (:628-904)
    struct Convergence {
        double R_sum[3], maxCoeff[3], B_avg[3], pvSum, cnvErrorPv, CNV[3], MB[3];
    };
    Convergence convergence(double dt, double tol_cnv) const {
        Convergence c{};
        const int Nb = P.pat.Nb;
        for (int e = 0; e < 3; ++e) { c.R_sum[e] = 0.0; c.maxCoeff[e] = std::numeric_limits<double>::lowest(); c.B_avg[e] = 0.0; }
        static const int comp[3] = {EQ_WATER, EQ_OIL, EQ_GAS};
        double pvSum = 0.0;
        for (int cell = 0; cell < Nb; ++cell) {
            const double pvValue = P.poro[cell] * P.volume[cell];
            pvSum += pvValue;
            for (int ph = 0; ph < 3; ++ph) {
                const int e = comp[ph];
                c.B_avg[e] += 1.0 / iqV[cell].invB[ph];
                const double R2 = residual[(size_t)cell * 3 + e];
                c.R_sum[e] += R2;
                c.maxCoeff[e] = std::max(c.maxCoeff[e], std::fabs(R2) / pvValue);
            }
        }
        for (int e = 0; e < 3; ++e) c.B_avg[e] /= (double)Nb;
        c.pvSum = pvSum;
        double errorPV = 0.0;
        for (int cell = 0; cell < Nb; ++cell) {
            const double pvValue = P.poro[cell] * P.volume[cell];
            bool violated = false;
            for (int e = 0; e < 3; ++e) {
                const double CNV = residual[(size_t)cell * 3 + e] * dt * c.B_avg[e] / pvValue;
                violated = violated || (std::fabs(CNV) > tol_cnv);
            }
            if (violated) errorPV += pvValue;
        }
        c.cnvErrorPv = errorPV;
        for (int e = 0; e < 3; ++e) {
            c.CNV[e] = c.B_avg[e] * dt * c.maxCoeff[e];
            c.MB[e] = std::fabs(c.B_avg[e] * c.R_sum[e]) * dt / pvSum;
        }
        return c;
    }

    // BlackOilNewtonMethod::update_ + adaptPrimaryVariables (App. B.7), then IQ recompute (updateSolution,
    // BlackoilModelEbos.hpp:549-563).  Returns the number of cells whose meaning switched.
    int update(const double* dx, double dpMaxRel = 0.3, double dsMax = 0.2, double oscThreshold = 1e-5) {
        const int Nb = P.pat.Nb;
        int nswitched = 0;
        for (int c = 0; c < Nb; ++c) {
            double* x = &pv[(size_t)c * 3];
            const double* u = &dx[(size_t)c * 3];
            const double deltaSw = u[PV_SW];
            double deltaSo = -deltaSw, deltaSg = 0.0;
            if (meaning[c] == Sw_po_Sg) { deltaSg = u[PV_X]; deltaSo -= deltaSg; }
            double maxSatDelta = std::max(std::fabs(deltaSg), std::fabs(deltaSo));
            maxSatDelta = std::max(maxSatDelta, std::fabs(deltaSw));
            double satAlpha = 1.0;
            if (maxSatDelta > dsMax) satAlpha = dsMax / maxSatDelta;
            double nx[3];
            for (int pvIdx = 0; pvIdx < 3; ++pvIdx) {
                double delta = u[pvIdx];
                if (pvIdx == PV_P) {
                    if (std::fabs(delta) > dpMaxRel * x[pvIdx]) delta = (delta < 0.0 ? -1.0 : 1.0) * dpMaxRel * x[pvIdx];
                } else if (pvIdx == PV_SW) delta *= satAlpha;
                else {
                    if (meaning[c] == Sw_po_Sg) delta *= satAlpha;
                    else if (delta > x[PV_X]) delta = x[PV_X];  // Rs must not become negative
                }
                nx[pvIdx] = x[pvIdx] - delta;
            }
            for (int k = 0; k < 3; ++k) x[k] = nx[k];
            const double eps = wasSwitched[c] ? oscThreshold : 0.0;
            wasSwitched[c] = adapt(c, eps) ? 1 : 0;
            nswitched += wasSwitched[c];
        }
        update_all_iq();
        return nswitched;
    }
    // BlackOilPrimaryVariables::adaptPrimaryVariables (App. B.7), three meanings.  UNVERIFIED vs upstream.
    bool adapt(int c, double eps) {
        const Fluid& F = P.fluid;
        const int pr = P.pvtnum.empty() ? 0 : P.pvtnum[c], sr = P.satnum.empty() ? 0 : P.satnum[c];
        const double RsMax = P.rsMax.empty() ? std::numeric_limits<double>::max() / 2.0 : P.rsMax[c];
        const double RvMax = P.rvMax.empty() ? std::numeric_limits<double>::max() / 2.0 : P.rvMax[c];
        double* x = &pv[(size_t)c * 3];
        const double Sw = x[PV_SW];
        // VAPPARS in the switches (BlackOilPrimaryVariables::adaptPrimaryVariables: SoMax = max(So, problem.maxOilSaturation))
        const bool vap = !P.maxOilSaturation.empty();
        auto vap_rs = [&](double So) { return vap ? vappars_factor<double>(So, std::max(So, P.maxOilSaturation[c]), P.vapPar2) : 1.0; };
        auto vap_rv = [&](double So) { return vap ? vappars_factor<double>(So, std::max(So, P.maxOilSaturation[c]), P.vapPar1) : 1.0; };
        const double thresholdWaterFilledCell = 1.0;  // static const 1.0 - eps of the first call (eps = 0)
        // special case: cells with (almost) only water
        if (Sw >= thresholdWaterFilledCell) {
            x[PV_SW] = 1.0;
            x[PV_X] = 0.0;
            const bool changed = meaning[c] != Sw_po_Sg;
            if (changed) meaning[c] = Sw_po_Sg;
            return changed;
        }
        if (meaning[c] == Sw_po_Sg) {
            const double Sg = x[PV_X];
            const double So = 1.0 - Sw - Sg;
            if (Sg < -eps && So > 0.0) {   // the gas phase disappears: { Sw, po, Rs }
                const double po = x[PV_P];
                const double RsSat = F.oil[pr].rsSat(po) * vap_rs(So);
                meaning[c] = Sw_po_Rs;
                x[PV_X] = std::min(RsMax, RsSat);
                return true;
            }
            if (So < -eps && Sg > 0.0 && F.hasWetGas) {   // the oil phase disappears: { Sw, pg, Rv }
                const double po = x[PV_P];
                double pC[3];
                if (!P.eps.empty()) F.sat[sr].capillaryPressuresEps(pC, Sw, Sg, P.eps[c], P.epsCfg);
                else F.sat[sr].capillaryPressures(pC, Sw, Sg);   // computeCapillaryPressures_(pC, So = 0, Sg, Sw)
                const double pg = po + (pC[GAS] - pC[OIL]);
                const double RvSat = F.wetGas[pr].rvSat(pg) * vap_rv(So);
                meaning[c] = Sw_pg_Rv;
                x[PV_P] = pg;
                x[PV_X] = std::min(RvMax, RvSat);
                return true;
            }
            return false;
        }
        if (meaning[c] == Sw_po_Rs) {
            const double po = x[PV_P];
            const double RsSat = F.oil[pr].rsSat(po) * vap_rs(1.0 - Sw);   // no gas: So = 1 - Sw
            const double Rs = x[PV_X];
            if (Rs > std::min(RsMax, RsSat * (1.0 + eps))) { meaning[c] = Sw_po_Sg; x[PV_X] = 0.0; return true; }
            return false;
        }
        // Sw_pg_Rv: the oil phase appears as soon as the gas holds more oil than saturated gas does
        const double pg = x[PV_P];
        const double RvSat = F.wetGas[pr].rvSat(pg) * vap_rv(0.0);   // no oil phase: So = 0
        const double Rv = x[PV_X];
        if (Rv > std::min(RvMax, RvSat * (1.0 + eps))) {
            meaning[c] = Sw_po_Sg;
            double pC[3];
            if (!P.eps.empty()) F.sat[sr].capillaryPressuresEps(pC, Sw, 1.0 - Sw, P.eps[c], P.epsCfg);
            else F.sat[sr].capillaryPressures(pC, Sw, 1.0 - Sw);
            const double po = pg + (pC[OIL] - pC[GAS]);
            x[PV_P] = po;
            x[PV_X] = 1.0 - Sw;   // hydrocarbon gas saturation
            return true;
        }
        return false;
    }
};

}  // namespace orc
