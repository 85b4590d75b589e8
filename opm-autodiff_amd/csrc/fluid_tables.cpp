// Host-side construction of the device property tables from deck-level input (see fluid_tables.hpp).
#include "fluid_tables.hpp"

#include <algorithm>
#include <cmath>
#include <string>

namespace opmhip {
namespace {

struct Node {
    double rs;
    std::vector<double> p, bo, mu;
};

int push(std::vector<double>& blob, const std::vector<double>& v) {
    const int o = (int)blob.size();
    blob.insert(blob.end(), v.begin(), v.end());
    return o;
}

}  // namespace

std::string build_fluid_tables(const opmhip_fluid* f, FluidTables& T) {
    if (!f) return "fluid == NULL";
    if (f->num_pvt < 1 || f->num_sat < 1) return "need at least one PVT and one saturation region";
    const bool wet = f->pvtg_node_ptr && f->pvtg_pg && f->pvtg_row_ptr && f->pvtg && f->pvtg_node_ptr[f->num_pvt] > 0;
    if (!f->pvtw || !f->density || (!wet && (!f->pvdg_ptr || !f->pvdg)) || !f->pvto_node_ptr || !f->pvto_rs || !f->pvto_row_ptr ||
        !f->pvto || !f->swof_ptr || !f->swof || !f->sgof_ptr || !f->sgof)
        return "fluid: null table pointer";
    if (f->num_rock < 0 || (f->num_rock > 0 && (!f->rocktab_ptr || !f->rocktab))) return "fluid: bad ROCKTAB input";
    T = FluidTables();
    T.wet_gas = wet;
    T.num_rock = f->num_rock;
    T.pc_scaling = f->pc_scaling != 0;
    T.num_pvt = f->num_pvt;
    T.num_sat = f->num_sat;
    T.rock_pref = f->rock_pref;
    T.rock_cr = f->rock_cr;
    std::vector<PvtRegionDesc> pd(f->num_pvt);
    std::vector<SatRegionDesc> sd(f->num_sat);
    std::vector<std::vector<int>> yoffs(f->num_pvt), gyoffs(f->num_pvt);
    std::vector<double>& B = T.dbl;

    for (int r = 0; r < f->num_pvt; ++r) {
        PvtRegionDesc& D = pd[r];
        // ---- ConstantCompressibilityWaterPvt / reference densities ----
        D.water = push(B, std::vector<double>(f->pvtw + 5 * r, f->pvtw + 5 * r + 5));
        D.density = push(B, std::vector<double>(f->density + 3 * r, f->density + 3 * r + 3));
        D.gas_n = D.gas_p = D.gas_invB = D.gas_invBMu = 0;
        D.wg_n = D.wg_xs = D.wg_yoff = D.wg_ys = D.wg_invB = D.wg_invBMu = D.wgs_rv = D.wgs_invB = D.wgs_invBMu = 0;
        // ---- WetGasPvt (PVTG): see oracle/fluid.hpp WetGasPvt for the construction being restated ----
        if (wet) {
            struct GNode { double pg; std::vector<double> rv, bg, mu; };
            std::vector<GNode> gn;
            for (int n = f->pvtg_node_ptr[r]; n < f->pvtg_node_ptr[r + 1]; ++n) {
                GNode g;
                g.pg = f->pvtg_pg[n];
                for (int q = f->pvtg_row_ptr[n]; q < f->pvtg_row_ptr[n + 1]; ++q) {
                    g.rv.push_back(f->pvtg[3 * q]); g.bg.push_back(f->pvtg[3 * q + 1]); g.mu.push_back(f->pvtg[3 * q + 2]);
                }
                if (g.rv.empty()) return "PVTG node without rows";
                if (!gn.empty() && g.pg <= gn.back().pg) return "PVTG pressure nodes must ascend";
                for (size_t q = 1; q < g.rv.size(); ++q)
                    if (g.rv[q] >= g.rv[q - 1]) return "PVTG rows of a node must start with the saturated one, Rv descending";
                gn.push_back(g);
            }
            const int gnn = (int)gn.size();
            if (gnn < 2) return "PVTG needs at least two pressure nodes";
            std::vector<double> xs, ysFlat, ibFlat, ibmFlat, sRv, sIb, sIbm;
            std::vector<int>& yo = gyoffs[r];
            for (int i = 0; i < gnn; ++i) {
                std::vector<double> Rv = gn[i].rv, Bg = gn[i].bg, Mu = gn[i].mu;
                if (Rv.size() < 2) {
                    int m = i + 1;
                    while (m < gnn && gn[m].rv.size() < 2) ++m;
                    if (m >= gnn) return "PVTG: the last pressure node must carry undersaturated data";
                    const GNode& M = gn[m];
                    for (size_t q = 1; q < M.rv.size(); ++q) {
                        const double diffRv = M.rv[q] - M.rv[q - 1];
                        const double newRv = Rv.back() + diffRv;
                        const double B1 = M.bg[q], B2 = M.bg[q - 1];
                        const double x = (B1 - B2) / ((B1 + B2) / 2.0);
                        const double newBg = Bg.back() * (1.0 + x / 2.0) / (1.0 - x / 2.0);
                        const double m1 = M.mu[q], m2 = M.mu[q - 1];
                        const double xMu = (m1 - m2) / ((m1 + m2) / 2.0);
                        const double newMu = Mu.back() * (1.0 + xMu / 2.0) / (1.0 - xMu / 2.0);
                        Rv.push_back(newRv); Bg.push_back(newBg); Mu.push_back(newMu);
                    }
                }
                xs.push_back(gn[i].pg);
                yo.push_back((int)ysFlat.size());
                for (int q = (int)Rv.size() - 1; q >= 0; --q) {   // ascending in Rv
                    ysFlat.push_back(Rv[q]);
                    ibFlat.push_back(1.0 / Bg[q]);
                    ibmFlat.push_back((1.0 / Bg[q]) / Mu[q]);
                }
                sRv.push_back(gn[i].rv[0]);
                sIb.push_back(ibFlat.back());     // the saturated sample is the last (largest Rv)
                sIbm.push_back(ibmFlat.back());
            }
            yo.push_back((int)ysFlat.size());
            D.wg_n = gnn;
            D.wg_xs = push(B, xs);
            D.wg_ys = push(B, ysFlat);
            D.wg_invB = push(B, ibFlat);
            D.wg_invBMu = push(B, ibmFlat);
            D.wgs_rv = push(B, sRv);
            D.wgs_invB = push(B, sIb);
            D.wgs_invBMu = push(B, sIbm);
        }
        // ---- DryGasPvt: p -> 1/Bg and 1/(Bg mu_g) on the deck's pressure samples ----
        if (!wet) {
            const int b = f->pvdg_ptr[r], e = f->pvdg_ptr[r + 1];
            if (e - b < 2) return "PVDG needs at least two rows";
            std::vector<double> p, ib, ibm;
            for (int q = b; q < e; ++q) {
                const double pp = f->pvdg[3 * q], Bg = f->pvdg[3 * q + 1], mu = f->pvdg[3 * q + 2];
                if (!p.empty() && pp <= p.back()) return "PVDG pressures must ascend";
                p.push_back(pp);
                ib.push_back(1.0 / Bg);
                ibm.push_back((1.0 / Bg) / mu);
            }
            D.gas_n = (int)p.size();
            D.gas_p = push(B, p);
            D.gas_invB = push(B, ib);
            D.gas_invBMu = push(B, ibm);
        }
        // ---- LiveOilPvt ----
        std::vector<Node> nodes;
        for (int n = f->pvto_node_ptr[r]; n < f->pvto_node_ptr[r + 1]; ++n) {
            Node nd;
            nd.rs = f->pvto_rs[n];
            for (int q = f->pvto_row_ptr[n]; q < f->pvto_row_ptr[n + 1]; ++q) {
                nd.p.push_back(f->pvto[3 * q]);
                nd.bo.push_back(f->pvto[3 * q + 1]);
                nd.mu.push_back(f->pvto[3 * q + 2]);
            }
            if (nd.p.empty()) return "PVTO node without rows";
            if (!nodes.empty() && nd.rs <= nodes.back().rs) return "PVTO Rs nodes must ascend";
            nodes.push_back(nd);
        }
        const int nn = (int)nodes.size();
        if (nn < 2) return "PVTO needs at least two Rs nodes";
        std::vector<std::vector<double>> ys(nn), ib(nn), mu(nn);
        for (int i = 0; i < nn; ++i) {
            ys[i] = nodes[i].p;
            mu[i] = nodes[i].mu;
            for (double b : nodes[i].bo) ib[i].push_back(1.0 / b);
        }
        // nodes with only the saturated sample inherit the next complete node's undersaturated branch
        // (LiveOilPvt::extendPvtoTable_: same relative compressibility / viscosibility step by step)
        for (int i = 0; i < nn; ++i) {
            if (nodes[i].p.size() > 1) continue;
            int m = i + 1;
            while (m < nn && nodes[m].p.size() <= 1) ++m;
            if (m >= nn) return "PVTO: the last Rs node must carry undersaturated data";
            const Node& M = nodes[m];
            double lastP = nodes[i].p.back(), lastBo = nodes[i].bo.back(), lastMu = nodes[i].mu.back();
            for (size_t q = 1; q < M.p.size(); ++q) {
                const double diffPo = M.p[q] - M.p[q - 1];
                const double newPo = lastP + diffPo;
                const double B1 = M.bo[q], B2 = M.bo[q - 1];
                const double x = (B1 - B2) / ((B1 + B2) / 2.0);
                const double newBo = lastBo * (1.0 + x / 2.0) / (1.0 - x / 2.0);
                const double mu1 = M.mu[q], mu2 = M.mu[q - 1];
                const double xMu = (mu1 - mu2) / ((mu1 + mu2) / 2.0);
                const double newMuo = lastMu * (1.0 + xMu / 2.0) / (1.0 - xMu / 2.0);
                ys[i].push_back(newPo);
                ib[i].push_back(1.0 / newBo);
                mu[i].push_back(newMuo);
                lastP = newPo; lastBo = newBo; lastMu = newMuo;
            }
        }
        std::vector<double> xs, ysFlat, ibFlat, ibmFlat, satP, satRs, satIb, satIbm;
        std::vector<int>& yo = yoffs[r];
        for (int i = 0; i < nn; ++i) {
            xs.push_back(nodes[i].rs);
            yo.push_back((int)ysFlat.size());
            for (size_t j = 0; j < ys[i].size(); ++j) {
                if (j > 0 && ys[i][j] <= ys[i][j - 1]) return "PVTO pressures of a node must ascend";
                ysFlat.push_back(ys[i][j]);
                ibFlat.push_back(ib[i][j]);
                ibmFlat.push_back(ib[i][j] / mu[i][j]);
            }
            satP.push_back(ys[i][0]);
            satRs.push_back(nodes[i].rs);
            satIb.push_back(ib[i][0]);
            satIbm.push_back(ib[i][0] / mu[i][0]);
            // equal bubble points of two nodes are let through, as the reference lets them (its own test deck
            // tests/SUMMARY_DECK_NON_CONSTANT_POROSITY.DATA has Rs = 0 and Rs = 1 both saturated at 1 bar): RsSat(p) has a jump there,
            // which only a state that sits exactly on it and asks for RsSat would see
            if (i > 0 && satP[i] < satP[i - 1]) return "PVTO bubble-point pressures must not descend";
        }
        yo.push_back((int)ysFlat.size());
        D.o_nx = nn;
        D.o_xs = push(B, xs);
        D.o_ys = push(B, ysFlat);
        D.o_invB = push(B, ibFlat);
        D.o_invBMu = push(B, ibmFlat);
        D.sat_n = nn;
        D.sat_p = push(B, satP);
        D.sat_rs = push(B, satRs);
        D.sat_invB = push(B, satIb);
        D.sat_invBMu = push(B, satIbm);
    }
    for (int s = 0; s < f->num_sat; ++s) {
        SatRegionDesc& D = sd[s];
        const int wb = f->swof_ptr[s], we = f->swof_ptr[s + 1], gb = f->sgof_ptr[s], ge = f->sgof_ptr[s + 1];
        if (we - wb < 2 || ge - gb < 2) return "SWOF/SGOF need at least two rows";
        std::vector<double> sw, krw, krow, pcow;
        for (int q = wb; q < we; ++q) {
            if (!sw.empty() && f->swof[4 * q] <= sw.back()) return "SWOF saturations must ascend";
            sw.push_back(f->swof[4 * q]); krw.push_back(f->swof[4 * q + 1]); krow.push_back(f->swof[4 * q + 2]); pcow.push_back(f->swof[4 * q + 3]);
        }
        const double swco = sw.front();
        // gas-oil system tabulated against So' = (1 - Swco) - Sg; reversed so that x ascends
        std::vector<double> so, krog, krg, pcgo;
        for (int q = ge - 1; q >= gb; --q) {
            const double x = (1.0 - swco) - f->sgof[4 * q];
            if (!so.empty() && x <= so.back()) return "SGOF saturations must ascend";
            so.push_back(x); krg.push_back(f->sgof[4 * q + 1]); krog.push_back(f->sgof[4 * q + 2]); pcgo.push_back(f->sgof[4 * q + 3]);
        }
        D.nw = (int)sw.size();
        D.sw_x = push(B, sw); D.krw = push(B, krw); D.krow = push(B, krow); D.pcow = push(B, pcow);
        D.ng = (int)so.size();
        D.so_x = push(B, so); D.krog = push(B, krog); D.krg = push(B, krg); D.pcgo = push(B, pcgo);
        D.swco = push(B, std::vector<double>{swco});
        // the tables' end points (opm-common satfunc, restated: connate = first row, maximum = last row, critical = the last
        // saturation at which the phase's relative permeability is still zero; same statements as oracle/fluid.hpp SatFunc::init)
        {
            const int nw = we - wb, ng = ge - gb;
            const double* W = f->swof + 4 * (size_t)wb;
            const double* G = f->sgof + 4 * (size_t)gb;
            std::vector<double> u(EPS_COUNT, 0.0);
            u[EPS_SWL] = sw.front(); u[EPS_SWU] = sw.back();
            u[EPS_SGL] = G[0]; u[EPS_SGU] = G[4 * (ng - 1)];
            u[EPS_SWCR] = sw.front();
            for (int i = 0; i < nw && W[4 * i + 1] <= 0.0; ++i) u[EPS_SWCR] = sw[i];
            u[EPS_SGCR] = G[0];
            for (int i = 0; i < ng && G[4 * i + 1] <= 0.0; ++i) u[EPS_SGCR] = G[4 * i];
            double swOilGone = sw.back();
            for (int i = nw - 1; i >= 0 && W[4 * i + 2] <= 0.0; --i) swOilGone = sw[i];
            u[EPS_SOWCR] = 1.0 - swOilGone - u[EPS_SGL];
            double sgOilGone = G[4 * (ng - 1)];
            for (int i = ng - 1; i >= 0 && G[4 * i + 2] <= 0.0; --i) sgOilGone = G[4 * i];
            u[EPS_SOGCR] = 1.0 - sgOilGone - u[EPS_SWL];
            u[EPS_MAXPCOW] = pcow.front(); u[EPS_MAXPCGO] = pcgo.front();   // pcgo is stored reversed: front = the last SGOF row
            u[EPS_MAXKRW] = krw.back(); u[EPS_MAXKROW] = krow.front(); u[EPS_MAXKRG] = krg.front(); u[EPS_MAXKROG] = krog.back();
            auto lin = [](const std::vector<double>& x, const std::vector<double>& y, double s) {   // PiecewiseLinearTwoPhaseMaterial
                if (s <= x.front()) return y.front();
                if (s >= x.back()) return y.back();
                size_t lo = 0, hi = x.size() - 1;
                while (lo + 1 < hi) { const size_t mid = (lo + hi) / 2; if (x[mid] < s) lo = mid; else hi = mid; }
                const double m = (y[lo + 1] - y[lo]) / (x[lo + 1] - x[lo]);
                return y[lo] + (s - x[lo]) * m;
            };
            u[EPS_KRWR] = lin(sw, krw, 1.0 - u[EPS_SOWCR] - u[EPS_SGL]);
            u[EPS_KRORW] = lin(sw, krow, u[EPS_SWCR] + u[EPS_SGL]);
            u[EPS_KRORG] = lin(so, krog, 1.0 - u[EPS_SGCR] - u[EPS_SWL]);
            u[EPS_KRGR] = lin(so, krg, u[EPS_SOGCR]);
            D.eps = push(B, u);
        }
    }
    std::vector<RockTabDesc> rd(f->num_rock);
    for (int t = 0; t < f->num_rock; ++t) {
        const int b = f->rocktab_ptr[t], e = f->rocktab_ptr[t + 1];
        if (e - b < 2) return "ROCKTAB needs at least two rows";
        std::vector<double> p, pm, tm;
        for (int q = b; q < e; ++q) {
            if (!p.empty() && f->rocktab[3 * q] <= p.back()) return "ROCKTAB pressures must ascend";
            p.push_back(f->rocktab[3 * q]); pm.push_back(f->rocktab[3 * q + 1]); tm.push_back(f->rocktab[3 * q + 2]);
        }
        rd[t].n = (int)p.size();
        rd[t].p = push(B, p); rd[t].poroMult = push(B, pm); rd[t].transMult = push(B, tm);
    }
    // int blob: header, descriptors, then the per-region y-offset arrays
    std::vector<int>& I = T.idx;
    I.clear();
    I.push_back(f->num_pvt);
    I.push_back(f->num_sat);
    const int pdInts = (int)(sizeof(PvtRegionDesc) / sizeof(int)), sdInts = (int)(sizeof(SatRegionDesc) / sizeof(int));
    const int base = 2 + f->num_pvt * pdInts + f->num_sat * sdInts;
    int cursor = base;
    for (int r = 0; r < f->num_pvt; ++r) {
        pd[r].o_yoff = cursor;
        cursor += (int)yoffs[r].size();
        pd[r].wg_yoff = cursor;
        cursor += (int)gyoffs[r].size();
    }
    T.rock_desc = cursor;
    for (int r = 0; r < f->num_pvt; ++r) {
        const int* q = reinterpret_cast<const int*>(&pd[r]);
        I.insert(I.end(), q, q + pdInts);
    }
    for (int s = 0; s < f->num_sat; ++s) {
        const int* q = reinterpret_cast<const int*>(&sd[s]);
        I.insert(I.end(), q, q + sdInts);
    }
    for (int r = 0; r < f->num_pvt; ++r) {
        I.insert(I.end(), yoffs[r].begin(), yoffs[r].end());
        I.insert(I.end(), gyoffs[r].begin(), gyoffs[r].end());
    }
    for (int t = 0; t < f->num_rock; ++t) {
        const int* q = reinterpret_cast<const int*>(&rd[t]);
        I.insert(I.end(), q, q + (int)(sizeof(RockTabDesc) / sizeof(int)));
    }
    return "";
}

void sat_end_points(const FluidTables& T, int s, double* out) {
    const int pdInts = (int)(sizeof(PvtRegionDesc) / sizeof(int));
    const SatRegionDesc* sd = reinterpret_cast<const SatRegionDesc*>(T.idx.data() + 2 + T.idx[0] * pdInts);
    for (int f = 0; f < EPS_COUNT; ++f) out[f] = T.dbl[sd[s].eps + f];
}

}  // namespace opmhip
