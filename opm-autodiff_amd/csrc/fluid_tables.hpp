// Device-resident black-oil property tables (product code).  Host side: build the interpolation tables from the
// deck-level input (opmhip_fluid) the way opm-material's LiveOilPvt / DryGasPvt / ConstantCompressibilityWaterPvt /
// EclDefaultMaterial do (those classes are not in the reference tree; behaviour restated from the public 2021.10
// sources; call sites ebos/eclproblem.hh:1490-1498, flow/BlackoilModelEbos.hpp:650-664).
// Device side: everything is flattened into ONE double blob plus ONE int blob so that a kernel needs two pointers.
#pragma once
#include <string>
#include <vector>

#include "../../include/opmhip.h"

namespace opmhip {

// Offsets into the blobs, per PVT region / saturation region.  POD, copied to the device as part of the int blob.
struct PvtRegionDesc {
    // 1-D tables: offsets of the x[] and y[] arrays (n entries each) inside the double blob
    int gas_n, gas_p, gas_invB, gas_invBMu;            // DryGasPvt: p -> 1/Bg, 1/(Bg mu_g)
    int sat_n, sat_p, sat_rs, sat_invB, sat_invBMu;    // saturated oil: p -> RsSat, 1/Bo, 1/(Bo mu_o)
    // 2-D table: nx Rs nodes at xs; per node i: yoff[i] (int blob) = start of its samples inside ys / invB / invBMu
    int o_nx, o_xs, o_yoff /*int blob, nx+1 entries*/, o_ys, o_invB, o_invBMu;
    int water;    // 5 doubles: p_ref, Bw_ref, c_w, mu_ref, c_v
    int density;  // 3 doubles: oil, water, gas
    // WetGasPvt (PVTG; wg_n == 0: dry gas): 2-D table over (p_g, Rv): wg_n pressure nodes at wg_xs, per node i its ascending
    // Rv samples wg_ys[yoff[i] .. yoff[i+1]) with 1/Bg and 1/(Bg mu_g); saturated 1-D tables on the same pressure nodes
    int wg_n, wg_xs, wg_yoff /*int blob, wg_n+1 entries*/, wg_ys, wg_invB, wg_invBMu;
    int wgs_rv, wgs_invB, wgs_invBMu;
};
struct RockTabDesc {  // ROCKTAB region: n rows, offsets of p, pore-volume multiplier, transmissibility multiplier
    int n, p, poroMult, transMult;
};
struct SatRegionDesc {
    int nw, sw_x, krw, krow, pcow;  // piecewise linear in Sw, x ascending
    int ng, so_x, krog, krg, pcgo;  // piecewise linear in So' = (1 - Swco) - Sg, x ascending
    int swco;                       // 1 double
    int eps;                        // EPS_COUNT doubles: the tables' own end points (saturation end-point scaling)
};
// end points of a saturation region's tables / of a cell (opm-material EclEpsScalingPointsInfo, absent from the reference
// tree - restated, see oracle/fluid.hpp): connate, critical and maximum saturations, maximum capillary pressures, maximum
// relative permeabilities and their values at the critical saturation of the displacing phase
enum EpsField { EPS_SWL = 0, EPS_SWCR, EPS_SWU, EPS_SOWCR, EPS_SGL, EPS_SGCR, EPS_SGU, EPS_SOGCR,
                EPS_MAXPCOW, EPS_MAXPCGO, EPS_MAXKRW, EPS_MAXKROW, EPS_MAXKRG, EPS_MAXKROG,
                EPS_KRWR, EPS_KRORW, EPS_KRGR, EPS_KRORG, EPS_COUNT };
// the end points of saturation region s of a table set built by build_fluid_tables (host copy of the blob)
void sat_end_points(const struct FluidTables& T, int s, double* out);

struct FluidTables {
    std::vector<double> dbl;
    std::vector<int> idx;  // [0] num_pvt, [1] num_sat, then PvtRegionDesc[num_pvt], SatRegionDesc[num_sat], then y offsets, then RockTabDesc[num_rock]
    double rock_pref = 1e5, rock_cr = 0.0;
    int num_pvt = 0, num_sat = 0, num_rock = 0;
    int rock_desc = 0;     // offset of the RockTabDesc array inside idx
    bool wet_gas = false;  // PVTG present: vaporised oil (Rv), third primary-variable meaning
    bool pc_scaling = false;  // per-cell end-point scaling of pcow (PCW / SWATINIT)
};

// returns "" on success, else an error text
std::string build_fluid_tables(const opmhip_fluid* f, FluidTables& out);

}  // namespace opmhip
