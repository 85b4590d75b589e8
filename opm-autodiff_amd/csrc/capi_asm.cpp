// C-ABI, assembly half (include/opmhip.h "assembly" section): fluid tables, static grid data, state, linearisation,
// convergence norms, Newton update.  Host arrays arrive in the NATURAL cell / entry order and are permuted on upload.
#include <climits>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "fluid_tables.hpp"
#include "internal.hpp"

using namespace opmhip;

namespace {

template <class F>
int guarded(opmhip_ctx* c, F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        return fail(c, OPMHIP_UNKNOWN_ERROR, "exception: %s", e.what());
    } catch (...) {
        return fail(c, OPMHIP_UNKNOWN_ERROR, "unknown exception");
    }
}

// host-side permutation of a per-cell array natural -> internal
template <class T>
std::vector<T> cells_to_internal(const Pattern& P, const T* nat, int width = 1) {
    std::vector<T> v((size_t)P.Nloc * width);
    for (int p = 0; p < P.Nloc; ++p)
        for (int q = 0; q < width; ++q) v[(size_t)p * width + q] = nat[(size_t)P.fromOrder[p] * width + q];
    return v;
}
template <class T>
int upload_cells(opmhip_ctx* c, T** dst, const T* nat, int width = 1) {
    const Pattern& P = c->pat;
    std::vector<T> v = cells_to_internal(P, nat, width);
    if (!*dst) {
        int rc = dev_alloc(c, dst, v.size());
        if (rc) return rc;
    }
    // the context's stream is non-blocking: an assembly enqueued earlier may still read this array
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    OPMHIP_HIP(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return OPMHIP_SUCCESS;
}
int upload_entries(opmhip_ctx* c, double** dst, const double* nat) {
    const Pattern& P = c->pat;
    std::vector<double> v(P.nnzb);
    for (int k = 0; k < P.nnzb; ++k) v[k] = nat[P.nnzMap[k]];
    if (!*dst) {
        int rc = dev_alloc(c, dst, v.size());
        if (rc) return rc;
    }
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    OPMHIP_HIP(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    return OPMHIP_SUCCESS;
}
}  // namespace

extern "C" {

int opmhip_set_fluid(opmhip_ctx* c, const opmhip_fluid* fluid) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        OPMHIP_HIP(c, hipSetDevice(c->device));
        AsmDev& A = c->asmb;
        // "setup, once": set_static sizes the intensive-quantity cache for THIS fluid's record layout and checks the region
        // arrays against ITS table counts - another fluid afterwards would let the 19-field kernels write past 17-field buffers
        if (A.static_set)
            return fail(c, OPMHIP_INVALID_ARGUMENT, "set_fluid: the static data of this context is already set (opmhip_set_static) - the fluid cannot be replaced any more");
        FluidTables T;
        const std::string msg = build_fluid_tables(fluid, T);
        if (!msg.empty()) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_fluid: %s", msg.c_str());
        int rc;
        if (A.fluid_set) {   // replaced before set_static: the old blobs go back (no kernel can be using them yet except a probe, which synchronises)
            OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
            dev_free(c, &A.d_tab_dbl);
            dev_free(c, &A.d_tab_idx);
        }
        if ((rc = dev_upload(c, &A.d_tab_dbl, T.dbl))) return rc;
        if ((rc = dev_upload(c, &A.d_tab_idx, T.idx))) return rc;
        A.tab_ndbl = (int)T.dbl.size();
        A.tab_nidx = (int)T.idx.size();
        A.rock_pref = T.rock_pref;
        A.rock_cr = T.rock_cr;
        A.num_pvt = T.num_pvt;
        A.num_sat = T.num_sat;
        A.num_rock = T.num_rock;
        A.rock_desc = T.rock_desc;
        A.wet_gas = T.wet_gas;
        A.pc_scaling = T.pc_scaling;
        A.ext = T.wet_gas || T.num_rock > 0 || T.pc_scaling;   // extended intensive-quantity record
        A.sat_eps.resize((size_t)T.num_sat * EPS_COUNT);
        for (int s = 0; s < T.num_sat; ++s) sat_end_points(T, s, &A.sat_eps[(size_t)s * EPS_COUNT]);
        A.fluid_set = true;
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_static(opmhip_ctx* c, const double* trans, const double* area, const double* thpres, const double* poro,
                      const double* volume, const double* depth, const int* pvtnum, const int* satnum, const double* rsmax) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->pattern_set) return fail(c, OPMHIP_NOT_READY, "set_static before set_pattern");
        if (!c->asmb.fluid_set) return fail(c, OPMHIP_NOT_READY, "set_static before set_fluid");
        if (!trans || !area || !poro || !volume || !depth) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_static: null array");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        const Pattern& P = c->pat;
        AsmDev& A = c->asmb;
        // symmetry of the per-connection data: entry (I,J) and (J,I) describe the same face
        {
            std::vector<int> tr(P.nnzb, -1);
            for (int i = 0; i < P.Nb; ++i)
                for (int k = P.nat_rowptr[i]; k < P.nat_rowptr[i + 1]; ++k) {
                    const int j = P.nat_col[k];
                    if (j >= P.Nb) { tr[k] = k; continue; }  // ghost column: its row lives on the neighbouring subdomain
                    const int* b = P.nat_col.data() + P.nat_rowptr[j];
                    const int* e = P.nat_col.data() + P.nat_rowptr[j + 1];   // (not &nat_col[...]: one past the end for the last row)
                    const int* q = std::lower_bound(b, e, i);
                    if (q == e || *q != i) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_static: pattern is not structurally symmetric at (%d,%d)", i, j);
                    tr[k] = (int)(q - P.nat_col.data());
                }
            for (int k = 0; k < P.nnzb; ++k)
                if (trans[k] != trans[tr[k]] || area[k] != area[tr[k]] || (thpres && thpres[k] != thpres[tr[k]]))
                    return fail(c, OPMHIP_INVALID_ARGUMENT, "set_static: trans/area/thpres differ between entry %d and its transpose", k);
        }
        for (int i = 0; i < P.Nloc; ++i) {
            if (!(volume[i] > 0.0) || !(poro[i] >= 0.0)) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_static: cell %d has non-positive volume or negative porosity", i);
            if (pvtnum && (pvtnum[i] < 0 || pvtnum[i] >= A.num_pvt)) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_static: pvtnum[%d] out of range", i);
            if (satnum && (satnum[i] < 0 || satnum[i] >= A.num_sat)) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_static: satnum[%d] out of range", i);
        }
        int rc;
        if ((rc = upload_entries(c, &A.d_trans, trans))) return rc;
        if ((rc = upload_entries(c, &A.d_area, area))) return rc;
        if (thpres) { if ((rc = upload_entries(c, &A.d_thpres, thpres))) return rc; } else A.d_thpres = nullptr;
        if ((rc = upload_cells(c, &A.d_poro, poro))) return rc;
        if ((rc = upload_cells(c, &A.d_volume, volume))) return rc;
        if ((rc = upload_cells(c, &A.d_depth, depth))) return rc;
        if (pvtnum) { if ((rc = upload_cells(c, &A.d_pvtnum, pvtnum))) return rc; } else A.d_pvtnum = nullptr;
        if (satnum) { if ((rc = upload_cells(c, &A.d_satnum, satnum))) return rc; } else A.d_satnum = nullptr;
        // DRSDT in force (opmhip_set_composition_change_limits): the cap array belongs to the library - lastRs + rate * dt, written by
        // begin_time_step - and must neither be overwritten nor withdrawn here
        if (A.drsdt_on) {
            if (rsmax) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_static: rsmax handed in while DRSDT is in force (opmhip_set_composition_change_limits owns the dissolution cap)");
        } else if (rsmax) { if ((rc = upload_cells(c, &A.d_rsmax, rsmax))) return rc; }
        else { OPMHIP_HIP(c, hipStreamSynchronize(c->stream)); dev_free(c, &A.d_rsmax); }
        if (!A.d_pv) {
            const size_t Nb = P.Nloc;  // per-cell state includes the ghost cells
            if ((rc = dev_alloc(c, &A.d_pv, Nb * 3))) return rc;
            if ((rc = dev_alloc(c, &A.d_iq, Nb * (size_t)iq_doubles_per_cell(c)))) return rc;
            if ((rc = dev_alloc(c, &A.d_storageOld, Nb * 3))) return rc;
            if ((rc = dev_alloc(c, &A.d_invb, Nb * 3))) return rc;
            if ((rc = dev_alloc(c, &A.d_drift, Nb * 3))) return rc;
            OPMHIP_HIP(c, hipMemsetAsync(A.d_drift, 0, Nb * 3 * sizeof(double), c->stream));
            if ((rc = dev_alloc(c, &A.d_source, Nb * 3))) return rc;
            if ((rc = dev_alloc(c, &A.d_dsource, Nb * 9))) return rc;
            if ((rc = dev_alloc(c, &A.d_meaning, Nb))) return rc;
            if ((rc = dev_alloc(c, &A.d_wasSwitched, Nb))) return rc;
            if ((rc = dev_alloc(c, &A.d_pv_prev, Nb * 3))) return rc;
            if ((rc = dev_alloc(c, &A.d_meaning_prev, Nb))) return rc;
            if ((rc = dev_alloc(c, &A.d_stage_u8, Nb))) return rc;
            if ((rc = dev_alloc(c, &A.d_nswitched, (size_t)1))) return rc;
            if ((rc = dev_alloc(c, &A.d_conv_part, ((Nb + 255) / 256) * 10))) return rc;
            if ((rc = dev_alloc(c, &A.d_conv_out, (size_t)16))) return rc;
            if ((rc = dev_alloc(c, &A.d_stage_cell, Nb * (size_t)iq_doubles_per_cell(c)))) return rc;
            OPMHIP_HIP(c, hipMemsetAsync(A.d_source, 0, Nb * 3 * sizeof(double), c->stream));
            OPMHIP_HIP(c, hipMemsetAsync(A.d_dsource, 0, Nb * 9 * sizeof(double), c->stream));
            OPMHIP_HIP(c, hipMemsetAsync(A.d_storageOld, 0, Nb * 3 * sizeof(double), c->stream));
            OPMHIP_HIP(c, hipMemsetAsync(A.d_wasSwitched, 0, Nb, c->stream));
            OPMHIP_HIP(c, hipStreamSynchronize(c->stream));   // (fills on the context's stream, complete before anything else can reach these arrays: capi.cpp, alloc_system)
            // assembly tiles: whole rows, at most 256 entries
            std::vector<int> row0;
            int r = 0;
            while (r < P.Nb) {
                row0.push_back(r);
                int e = r;
                while (e < P.Nb && P.rowptr[e + 1] - P.rowptr[r] <= asm_threads() && e - r < asm_max_rows()) ++e;
                if (e == r) return fail(c, OPMHIP_ANALYSIS_FAILED, "set_static: row %d has more than %d blocks", r, asm_threads());
                r = e;
            }
            row0.push_back(P.Nb);
            A.ntiles = (int)row0.size() - 1;
            // Schedule: the ILU ordering stores the colours one after the other, so a cell and its neighbours of another
            // colour sit at the same RELATIVE position of two far-apart regions.  Tiles are therefore launched by their
            // relative position inside their colour, colours interleaved: the intensive quantities a tile gathers from
            // the other colours were, or will shortly be, touched by the tiles running next to it and stay in L2.
            // One record per tile (first row, end row, first entry, end entry): the kernel's first load.
            std::vector<int> sched;
            {
                std::vector<int> colorOfTile(A.ntiles), first(P.numColors + 1, A.ntiles), cnt(P.numColors, 0);
                int cc = 0;
                for (int t = 0; t < A.ntiles; ++t) {
                    while (cc + 1 < P.numColors && row0[t] >= P.colorPrefix[cc + 1]) ++cc;
                    colorOfTile[t] = cc;
                    first[cc] = std::min(first[cc], t);
                    cnt[cc]++;
                }
                std::vector<int> order(A.ntiles);
                for (int t = 0; t < A.ntiles; ++t) order[t] = t;
                auto frac = [&](int t) { const int q = colorOfTile[t]; return (double)(t - first[q]) / (double)cnt[q]; };
                std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return frac(x) < frac(y); });
                // Workgroup b of the launch runs on XCD b % 8; every XCD gets one contiguous eighth of this order, so that the
                // neighbour records several nearby tiles gather are found in that XCD's own L2.  The permutation is folded
                // into the schedule (record b = what workgroup b, b + gridDim, ... work on); the last records may be empty.
                const int chunk = (A.ntiles + 7) / 8;
                A.nsched = 8 * chunk;
                sched.assign((size_t)4 * A.nsched, 0);
                for (int b = 0; b < A.nsched; ++b) {
                    const int pos = (b & 7) * chunk + (b >> 3);
                    if (pos >= A.ntiles) continue;
                    const int t = order[pos];
                    sched[(size_t)4 * b] = row0[t];
                    sched[(size_t)4 * b + 1] = row0[t + 1];
                    sched[(size_t)4 * b + 2] = P.rowptr[row0[t]];
                    sched[(size_t)4 * b + 3] = P.rowptr[row0[t + 1]];
                }
                if ((rc = dev_upload(c, &A.d_asm_sched, sched))) return rc;
            }
            // Per workgroup and lane = per entry (I,J) of the workgroup's tile: the column and an entry word - everything a
            // lane needs to know about its entry, addressed by the workgroup index alone (no dependent load):
            //   bits 0-5  row of the entry inside its tile
            //   bit  6    upwind tie-break of a face with equal pressures and volumes: "the DOF which exhibits the smaller
            //             global index" (ebos/eclfluxmodule.hh:303-314) - set if the global id of I (the index of the natural
            //             order; the global id in decomposed runs) is below that of J, never the position in the ILU ordering
            //   bits 8-15 the row's entries in ascending natural-column order (ascending GLOBAL neighbour id in decomposed
            //             runs): in-row position of the entry that comes i-th, stored with the row's i-th entry - the order
            //             in which the natural-order CPU path sums the face fluxes of a cell
            {
                const int T = asm_threads();
                std::vector<int> desc((size_t)2 * T * A.nsched, 0);
                std::vector<int> byNat;
                for (int b = 0; b < A.nsched; ++b) {
                    const int tr0 = sched[(size_t)4 * b], tr1 = sched[(size_t)4 * b + 1], tk0 = sched[(size_t)4 * b + 2];
                    for (int p = tr0; p < tr1; ++p) {
                        const int kb = P.rowptr[p], ke = P.rowptr[p + 1];
                        byNat.resize(ke - kb);
                        for (int k = kb; k < ke; ++k) byNat[k - kb] = k;
                        if (P.gids.empty())
                            std::sort(byNat.begin(), byNat.end(), [&](int x, int y) { return P.nnzMap[x] < P.nnzMap[y]; });
                        else  // natural local id of a column = fromOrder[internal col]
                            std::sort(byNat.begin(), byNat.end(),
                                      [&](int x, int y) { return P.gids[P.fromOrder[P.col[x]]] < P.gids[P.fromOrder[P.col[y]]]; });
                        const int in = P.fromOrder[p];
                        const long long gi = P.gids.empty() ? (long long)in : P.gids[in];
                        for (int k = kb; k < ke; ++k) {
                            const int jn = P.fromOrder[P.col[k]];
                            const long long gj = P.gids.empty() ? (long long)jn : P.gids[jn];
                            const size_t o = (size_t)2 * ((size_t)b * T + (k - tk0));
                            desc[o] = P.col[k];
                            desc[o + 1] = (p - tr0) | (gi < gj ? 64 : 0) | ((byNat[k - kb] - kb) << 8);
                        }
                    }
                }
                if ((rc = dev_upload(c, &A.d_asm_desc, desc))) return rc;
            }
        }
        A.static_set = true;
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_problem_extras(opmhip_ctx* c, const double* rvmax, const int* rocknum, const double* overburden) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.static_set) return fail(c, OPMHIP_NOT_READY, "set_problem_extras before set_static");
        if ((rvmax || rocknum || overburden) && !A.ext)
            return fail(c, OPMHIP_INVALID_ARGUMENT, "set_problem_extras: the fluid has neither PVTG nor ROCKTAB tables - nothing these arrays could act on");
        int rockMax = -1;
        if (rocknum) {
            // tables: ROCKTAB of the fluid, else those of opmhip_set_water_compaction - which may still be to come (it needs the
            // state): then the indices are checked when it arrives
            const int limit = A.num_rock > 0 ? A.num_rock : (A.num_wc > 0 ? A.num_wc : INT_MAX);
            for (int i = 0; i < c->pat.Nloc; ++i) {
                if (rocknum[i] < 0 || rocknum[i] >= limit) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_problem_extras: rocknum[%d] out of range", i);
                rockMax = std::max(rockMax, rocknum[i]);
            }
        }
        A.h_rocknum_max = rockMax;
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        // an array that is withdrawn goes back to the allocator (an assembly enqueued earlier may still read it: sync first)
        if (!rvmax || !rocknum || !overburden) OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        // DRVDT in force: the cap array is the library's (lastRv + rate * dt); re-sending rocknum / overburden must leave it alone
        if (A.drvdt_on) {
            if (rvmax) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_problem_extras: rvmax handed in while DRVDT is in force (opmhip_set_composition_change_limits owns the vaporisation cap)");
        } else if (rvmax) { if ((rc = upload_cells(c, &A.d_rvmax, rvmax))) return rc; } else dev_free(c, &A.d_rvmax);
        if (rocknum) { if ((rc = upload_cells(c, &A.d_rocknum, rocknum))) return rc; } else dev_free(c, &A.d_rocknum);
        if (overburden) { if ((rc = upload_cells(c, &A.d_overburden, overburden))) return rc; } else dev_free(c, &A.d_overburden);
        if (A.state_set) {   // the cached intensive quantities depend on these arrays
            launch_iq_update(c);
            OPMHIP_HIP(c, hipGetLastError());
            OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        }
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_pcw(opmhip_ctx* c, const double* pcw) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.static_set) return fail(c, OPMHIP_NOT_READY, "set_pcw before set_static");
        if (pcw && A.d_eps) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_pcw: opmhip_set_endpoint_scaling is in force - PCW is its points[OPMHIP_EPS_MAXPCOW]");
        if (pcw && !A.pc_scaling)
            return fail(c, OPMHIP_INVALID_ARGUMENT, "set_pcw: the fluid was set without pc_scaling - its capillary pressure curves have no per-cell end point");
        if (pcw)
            for (int i = 0; i < c->pat.Nloc; ++i)
                if (!std::isfinite(pcw[i])) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_pcw: pcw[%d] is not finite", i);
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        if (pcw) { if ((rc = upload_cells(c, &A.d_pcw, pcw))) return rc; }
        else if (A.d_pcw) {
            OPMHIP_HIP(c, hipStreamSynchronize(c->stream));   // an assembly enqueued earlier may still read it
            dev_free(c, &A.d_pcw);
        }
        if (A.state_set) {   // the cached intensive quantities depend on it
            launch_iq_update(c);
            OPMHIP_HIP(c, hipGetLastError());
            OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        }
        return OPMHIP_SUCCESS;
    });
}

int opmhip_sat_end_points(opmhip_ctx* c, int sat_region, double* out) {
    if (!c || !out) return OPMHIP_INVALID_ARGUMENT;
    const AsmDev& A = c->asmb;
    if (!A.fluid_set) return fail(c, OPMHIP_NOT_READY, "sat_end_points before set_fluid");
    if (sat_region < 0 || sat_region >= A.num_sat) return fail(c, OPMHIP_INVALID_ARGUMENT, "sat_end_points: region out of range");
    for (int f = 0; f < EPS_COUNT; ++f) out[f] = A.sat_eps[(size_t)sat_region * EPS_COUNT + f];
    return OPMHIP_SUCCESS;
}

int opmhip_set_endpoint_scaling(opmhip_ctx* c, const opmhip_endpoint_scaling* e) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    static_assert((int)OPMHIP_EPS_COUNT == (int)EPS_COUNT, "public and internal end-point indices");
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        const Pattern& P = c->pat;
        if (!A.static_set) return fail(c, OPMHIP_NOT_READY, "set_endpoint_scaling before set_static");
        if (e && !A.pc_scaling)
            return fail(c, OPMHIP_INVALID_ARGUMENT, "set_endpoint_scaling: the fluid was set without pc_scaling - its saturation functions have no per-cell end points");
        if (e && A.d_pcw) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_endpoint_scaling: opmhip_set_pcw is in force - hand PCW in as points[OPMHIP_EPS_MAXPCOW] with pcw = 1 instead");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        if (!e) {
            if (A.d_eps) { OPMHIP_HIP(c, hipStreamSynchronize(c->stream)); dev_free(c, &A.d_eps); }
            A.epscfg = 0;
        } else {
            if (e->krw < 0 || e->krw > 2 || e->kro < 0 || e->kro > 2 || e->krg < 0 || e->krg > 2)
                return fail(c, OPMHIP_INVALID_ARGUMENT, "set_endpoint_scaling: krw / kro / krg must be 0, 1 or 2");
            const int N = P.Nloc;
            std::vector<int> satnum(N, 0);   // internal order, as on the device
            if (A.d_satnum) {
                OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
                OPMHIP_HIP(c, hipMemcpy(satnum.data(), A.d_satnum, (size_t)N * sizeof(int), hipMemcpyDeviceToHost));
            }
            std::vector<double> v((size_t)EPS_COUNT * N);   // field-major, internal order
            for (int pos = 0; pos < N; ++pos) {
                const int i = P.fromOrder[pos];   // natural id (ghost cells keep their place behind the owned ones)
                double s[EPS_COUNT];
                for (int f = 0; f < EPS_COUNT; ++f) {
                    s[f] = e->points[f] ? e->points[f][i] : A.sat_eps[(size_t)satnum[pos] * EPS_COUNT + f];
                    if (!std::isfinite(s[f])) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_endpoint_scaling: end point %d of cell %d is not finite", f, i);
                    v[(size_t)f * N + pos] = s[f];
                }
                // the scaling maps [first, last] point of every curve onto the table's: the scaled intervals must not be empty
                const bool ok = s[EPS_SWL] < s[EPS_SWU] && s[EPS_SGL] < s[EPS_SGU] && s[EPS_SWCR] < s[EPS_SWU] && s[EPS_SGCR] < s[EPS_SGU] &&
                                s[EPS_SWL] + s[EPS_SGL] < 1.0 - s[EPS_SOWCR] && s[EPS_SOGCR] < 1.0 - s[EPS_SWL] - s[EPS_SGL];
                if (e->sat_scaling && !ok) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_endpoint_scaling: the end points of cell %d leave a curve no saturation interval", i);
            }
            int rc;
            if (!A.d_eps && (rc = dev_alloc(c, &A.d_eps, v.size()))) return rc;
            OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
            OPMHIP_HIP(c, hipMemcpy(A.d_eps, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
            A.epscfg = (e->sat_scaling ? 1 : 0) | (e->three_point_kr ? 2 : 0) | (e->krw << 2) | (e->kro << 4) | (e->krg << 6) | (e->pcw ? 256 : 0) | (e->pcg ? 512 : 0);
        }
        if (A.state_set) {   // the cached intensive quantities depend on it
            launch_iq_update(c);
            OPMHIP_HIP(c, hipGetLastError());
            OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        }
        return OPMHIP_SUCCESS;
    });
}

int opmhip_sat_probe(opmhip_ctx* c, int sat_region, const opmhip_endpoint_scaling* e, int n, const double* sw, const double* sg, double* out) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.fluid_set) return fail(c, OPMHIP_NOT_READY, "sat_probe before set_fluid");
        if (n < 0 || (n > 0 && (!sw || !sg || !out))) return fail(c, OPMHIP_INVALID_ARGUMENT, "sat_probe: null array");
        if (sat_region < 0 || sat_region >= A.num_sat) return fail(c, OPMHIP_INVALID_ARGUMENT, "sat_probe: region out of range");
        if (e && (e->krw < 0 || e->krw > 2 || e->kro < 0 || e->kro > 2 || e->krg < 0 || e->krg > 2))
            return fail(c, OPMHIP_INVALID_ARGUMENT, "sat_probe: krw / kro / krg must be 0, 1 or 2");
        if (n == 0) return OPMHIP_SUCCESS;
        OPMHIP_HIP(c, hipSetDevice(c->device));
        double pts[EPS_COUNT];
        int cfg = -1;
        if (e) {
            for (int f = 0; f < EPS_COUNT; ++f) pts[f] = e->points[f] ? e->points[f][0] : A.sat_eps[(size_t)sat_region * EPS_COUNT + f];
            cfg = (e->sat_scaling ? 1 : 0) | (e->three_point_kr ? 2 : 0) | (e->krw << 2) | (e->kro << 4) | (e->krg << 6) | (e->pcw ? 256 : 0) | (e->pcg ? 512 : 0);
        }
        struct Scratch { double *in = nullptr, *out = nullptr, *eps = nullptr; ~Scratch() { if (in) (void)hipFree(in); if (out) (void)hipFree(out); if (eps) (void)hipFree(eps); } } S;
        OPMHIP_HIP(c, hipMalloc((void**)&S.in, (size_t)2 * n * sizeof(double)));
        OPMHIP_HIP(c, hipMalloc((void**)&S.out, (size_t)5 * n * sizeof(double)));
        OPMHIP_HIP(c, hipMalloc((void**)&S.eps, (size_t)EPS_COUNT * sizeof(double)));
        OPMHIP_HIP(c, hipMemcpyAsync(S.in, sw, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        OPMHIP_HIP(c, hipMemcpyAsync(S.in + n, sg, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (e) OPMHIP_HIP(c, hipMemcpyAsync(S.eps, pts, sizeof pts, hipMemcpyHostToDevice, c->stream));
        launch_sat_probe(c, sat_region, cfg, S.eps, n, S.in, S.out);
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipMemcpyAsync(out, S.out, (size_t)5 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_fluid_probe(opmhip_ctx* c, int pvt_region, int sat_region, int n, const double* p, const double* rs, const double* sw,
                       const double* sg, double* out) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.fluid_set) return fail(c, OPMHIP_NOT_READY, "fluid_probe before set_fluid");
        if (n < 0 || (n > 0 && (!p || !rs || !sw || !sg || !out))) return fail(c, OPMHIP_INVALID_ARGUMENT, "fluid_probe: null array");
        if (pvt_region < 0 || pvt_region >= A.num_pvt || sat_region < 0 || sat_region >= A.num_sat)
            return fail(c, OPMHIP_INVALID_ARGUMENT, "fluid_probe: region out of range");
        if (n == 0) return OPMHIP_SUCCESS;
        OPMHIP_HIP(c, hipSetDevice(c->device));
        struct Scratch { double *in = nullptr, *out = nullptr; ~Scratch() { if (in) (void)hipFree(in); if (out) (void)hipFree(out); } } S;
        OPMHIP_HIP(c, hipMalloc((void**)&S.in, (size_t)4 * n * sizeof(double)));
        OPMHIP_HIP(c, hipMalloc((void**)&S.out, (size_t)8 * n * sizeof(double)));
        const double* src[4] = {p, rs, sw, sg};
        for (int k = 0; k < 4; ++k)
            OPMHIP_HIP(c, hipMemcpyAsync(S.in + (size_t)k * n, src[k], (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        launch_fluid_probe(c, pvt_region, sat_region, n, S.in, S.out);
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipMemcpyAsync(out, S.out, (size_t)8 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_gas_probe(opmhip_ctx* c, int pvt_region, int n, const double* p, const double* rv, double* out) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.fluid_set) return fail(c, OPMHIP_NOT_READY, "gas_probe before set_fluid");
        if (n < 0 || (n > 0 && (!p || !rv || !out))) return fail(c, OPMHIP_INVALID_ARGUMENT, "gas_probe: null array");
        if (pvt_region < 0 || pvt_region >= A.num_pvt) return fail(c, OPMHIP_INVALID_ARGUMENT, "gas_probe: region out of range");
        if (n == 0) return OPMHIP_SUCCESS;
        OPMHIP_HIP(c, hipSetDevice(c->device));
        struct Scratch { double *in = nullptr, *out = nullptr; ~Scratch() { if (in) (void)hipFree(in); if (out) (void)hipFree(out); } } S;
        OPMHIP_HIP(c, hipMalloc((void**)&S.in, (size_t)2 * n * sizeof(double)));
        OPMHIP_HIP(c, hipMalloc((void**)&S.out, (size_t)3 * n * sizeof(double)));
        OPMHIP_HIP(c, hipMemcpyAsync(S.in, p, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        OPMHIP_HIP(c, hipMemcpyAsync(S.in + n, rv, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        launch_gas_probe(c, pvt_region, n, S.in, S.out);
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipMemcpyAsync(out, S.out, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_state(opmhip_ctx* c, const double* pv, const unsigned char* meaning) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->asmb.static_set) return fail(c, OPMHIP_NOT_READY, "set_state before set_static");
        if (!pv || !meaning) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_state: null array");
        for (int i = 0; i < c->pat.Nloc; ++i)
            if (meaning[i] > (c->asmb.wet_gas ? OPMHIP_SW_PG_RV : OPMHIP_SW_PO_RS))
                return fail(c, OPMHIP_INVALID_ARGUMENT, "set_state: meaning[%d] = %d is not valid (Sw_po_Sg, Sw_po_Rs; Sw_pg_Rv only with a PVTG fluid)", i, (int)meaning[i]);
        OPMHIP_HIP(c, hipSetDevice(c->device));
        AsmDev& A = c->asmb;
        OPMHIP_HIP(c, hipMemcpyAsync(c->d_stageV, pv, (size_t)c->pat.Nloc * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        launch_vec_to_internal(c, c->d_stageV, A.d_pv, c->pat.Nloc);
        OPMHIP_HIP(c, hipMemcpyAsync(A.d_stage_u8, meaning, (size_t)c->pat.Nloc, hipMemcpyHostToDevice, c->stream));
        launch_u8_to_internal(c, A.d_stage_u8, A.d_meaning);
        OPMHIP_HIP(c, hipMemsetAsync(A.d_wasSwitched, 0, c->pat.Nloc, c->stream));
        launch_iq_update(c);
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        A.state_set = true;
        A.prev_set = false;
        return OPMHIP_SUCCESS;
    });
}

int opmhip_advance_time_level(opmhip_ctx* c) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.state_set) return fail(c, OPMHIP_NOT_READY, "advance_time_level before set_state");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        // ghost cells included: a later update_failed needs no communication
        OPMHIP_HIP(c, hipMemcpyAsync(A.d_pv_prev, A.d_pv, (size_t)c->pat.Nloc * 3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        OPMHIP_HIP(c, hipMemcpyAsync(A.d_meaning_prev, A.d_meaning, (size_t)c->pat.Nloc, hipMemcpyDeviceToDevice, c->stream));
        A.prev_set = true;
        return OPMHIP_SUCCESS;
    });
}

int opmhip_relative_change(opmhip_ctx* c, double* relative_change) {
    if (!c || !relative_change) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.prev_set) return fail(c, OPMHIP_NOT_READY, "relative_change before advance_time_level: there is no old time level to compare with");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        if (!A.d_rc && (rc = dev_alloc(c, &A.d_rc, (size_t)2 * 256 + 2))) return rc;
        if ((rc = launch_relative_change(c))) return rc;
        double h[2];
        OPMHIP_HIP(c, hipMemcpyAsync(h, A.d_rc + 2 * 256, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        *relative_change = h[1] > 0.0 ? h[0] / h[1] : 0.0;
        return OPMHIP_SUCCESS;
    });
}

int opmhip_update_failed(opmhip_ctx* c) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.prev_set) return fail(c, OPMHIP_NOT_READY, "update_failed before advance_time_level");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipMemcpyAsync(A.d_pv, A.d_pv_prev, (size_t)c->pat.Nloc * 3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        OPMHIP_HIP(c, hipMemcpyAsync(A.d_meaning, A.d_meaning_prev, (size_t)c->pat.Nloc, hipMemcpyDeviceToDevice, c->stream));
        OPMHIP_HIP(c, hipMemsetAsync(A.d_wasSwitched, 0, c->pat.Nloc, c->stream));
        launch_iq_update(c);
        OPMHIP_HIP(c, hipGetLastError());
        A.assembled = false;
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_composition_change_limits(opmhip_ctx* c, const double* drsdt, const int* drsdt_all, const double* drvdt) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.state_set) return fail(c, OPMHIP_NOT_READY, "set_composition_change_limits before set_state: lastRs / lastRv are taken from the initial solution");
        if (drvdt && !A.wet_gas) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_composition_change_limits: DRVDT needs a fluid with PVTG");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        const size_t N = c->pat.Nloc;
        int rc;
        auto put = [&](double** dst, const double* src) -> int {
            if (!*dst && (rc = dev_alloc(c, dst, (size_t)A.num_pvt))) return rc;
            OPMHIP_HIP(c, hipMemcpy(*dst, src, (size_t)A.num_pvt * sizeof(double), hipMemcpyHostToDevice));
            return OPMHIP_SUCCESS;
        };
        A.drsdt_on = drsdt != nullptr;
        A.drvdt_on = drvdt != nullptr;
        if (drsdt) {
            if ((rc = put(&A.d_drsdt, drsdt))) return rc;
            std::vector<int> all(A.num_pvt, 0);
            if (drsdt_all) all.assign(drsdt_all, drsdt_all + A.num_pvt);
            if (!A.d_drsdt_all && (rc = dev_alloc(c, &A.d_drsdt_all, (size_t)A.num_pvt))) return rc;
            OPMHIP_HIP(c, hipMemcpy(A.d_drsdt_all, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
            if (!A.d_lastRs && (rc = dev_alloc(c, &A.d_lastRs, N))) return rc;
            if (!A.d_rsmax && (rc = dev_alloc(c, &A.d_rsmax, N))) return rc;
        } else if (A.d_lastRs) {   // the keyword went out of force: no cap any more
            dev_free(c, &A.d_lastRs);
            dev_free(c, &A.d_rsmax);
        }
        if (drvdt) {
            if ((rc = put(&A.d_drvdt, drvdt))) return rc;
            if (!A.d_lastRv && (rc = dev_alloc(c, &A.d_lastRv, N))) return rc;
            if (!A.d_rvmax && (rc = dev_alloc(c, &A.d_rvmax, N))) return rc;
        } else if (A.d_lastRv) {
            dev_free(c, &A.d_lastRv);
            dev_free(c, &A.d_rvmax);
        }
        A.storage_frozen = false;
        if (A.drsdt_on || A.drvdt_on) {
            launch_last_rs_rv(c);            // updateCompositionChangeLimits_ on the initial solution (eclproblem.hh:1811)
            launch_set_limits(c, 0.0);       // until the first begin_time_step: the caps of the state itself
        }
        launch_iq_update(c);
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_irreversible_compaction(opmhip_ctx* c, int enable) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.state_set) return fail(c, OPMHIP_NOT_READY, "set_irreversible_compaction before set_state: the minimum pressure starts from the initial solution");
        if (enable && A.num_rock <= 0) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_irreversible_compaction: the fluid has no ROCKTAB tables");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        if (!enable) dev_free(c, &A.d_minpo);
        else {
            int rc;
            if (!A.d_minpo && (rc = dev_alloc(c, &A.d_minpo, (size_t)c->pat.Nloc))) return rc;
            launch_min_pressure(c, true);
        }
        launch_iq_update(c);
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_vappars(opmhip_ctx* c, int enable, double vap1, double vap2) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.state_set) return fail(c, OPMHIP_NOT_READY, "set_vappars before set_state: the maximum oil saturation starts from the initial solution");
        if (enable && !A.ext) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_vappars: needs a context with the extended record (a fluid with PVTG, ROCKTAB or pc_scaling)");
        if (enable && (!(vap1 >= 0.0) || !(vap2 >= 0.0))) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_vappars: the exponents must not be negative");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        if (!enable) dev_free(c, &A.d_maxso);
        else {
            int rc;
            if (!A.d_maxso && (rc = dev_alloc(c, &A.d_maxso, (size_t)c->pat.Nloc))) return rc;
            A.vap1 = vap1; A.vap2 = vap2;
            launch_max_oil_saturation(c, true);
        }
        launch_iq_update(c);
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_water_compaction(opmhip_ctx* c, int num_tables, const int* num_pressure, const int* num_sw, const double* pressure, const double* sw,
                                const double* pv_mult, const double* trans_mult) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.state_set) return fail(c, OPMHIP_NOT_READY, "set_water_compaction before set_state: the initial and the maximum water saturation start from the initial solution");
        if (num_tables < 0 || (num_tables > 0 && (!num_pressure || !num_sw || !pressure || !sw || !pv_mult)))
            return fail(c, OPMHIP_INVALID_ARGUMENT, "set_water_compaction: null table array");
        if (num_tables > 0 && !A.ext) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_water_compaction: needs a context with the extended record (a fluid with PVTG or pc_scaling)");
        if (num_tables > 0 && A.num_rock > 0) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_water_compaction: the fluid has ROCKTAB tables (the reference reads the one or the other)");
        std::vector<int> desc;
        std::vector<double> data;
        size_t op = 0, os = 0, ov = 0;
        for (int t = 0; t < num_tables; ++t) {
            const int np = num_pressure[t], ns = num_sw[t];
            if (np < 2 || ns < 2) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_water_compaction: table %d needs at least two pressure and two saturation nodes", t);
            for (int i = 1; i < np; ++i) if (!(pressure[op + i] > pressure[op + i - 1])) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_water_compaction: pressure nodes of table %d not ascending", t);
            for (int j = 1; j < ns; ++j) if (!(sw[os + j] > sw[os + j - 1])) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_water_compaction: saturation nodes of table %d not ascending", t);
            const int atP = (int)data.size();
            data.insert(data.end(), pressure + op, pressure + op + np);
            const int atS = (int)data.size();
            data.insert(data.end(), sw + os, sw + os + ns);
            const int atV = (int)data.size();
            data.insert(data.end(), pv_mult + ov, pv_mult + ov + (size_t)np * ns);
            int atT = -1;
            if (trans_mult) { atT = (int)data.size(); data.insert(data.end(), trans_mult + ov, trans_mult + ov + (size_t)np * ns); }
            const int d[6] = {np, ns, atP, atS, atV, atT};
            desc.insert(desc.end(), d, d + 6);
            op += np; os += ns; ov += (size_t)np * ns;
        }
        if (num_tables > 0 && A.h_rocknum_max >= num_tables) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_water_compaction: a cell's rock-table index is %d, %d tables given", A.h_rocknum_max, num_tables);
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        dev_free(c, &A.d_wcdesc); dev_free(c, &A.d_wcdata);
        if (num_tables == 0) { dev_free(c, &A.d_maxsw); dev_free(c, &A.d_sw0); }
        else {
            int rc;
            if ((rc = dev_upload(c, &A.d_wcdesc, desc)) || (rc = dev_upload(c, &A.d_wcdata, data))) return rc;
            if (!A.d_maxsw && (rc = dev_alloc(c, &A.d_maxsw, (size_t)c->pat.Nloc))) return rc;
            if (!A.d_sw0 && (rc = dev_alloc(c, &A.d_sw0, (size_t)c->pat.Nloc))) return rc;
            launch_max_water_saturation(c, true);
        }
        A.num_wc = num_tables;
        launch_iq_update(c);
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_get_max_water_saturation(opmhip_ctx* c, double* max_sw) {
    if (!c || !max_sw) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        const Pattern& P = c->pat;
        if (!A.static_set) return fail(c, OPMHIP_NOT_READY, "get_max_water_saturation before set_static");
        const int N = P.Nloc;
        if (!A.d_maxsw) { std::fill(max_sw, max_sw + N, 0.0); return OPMHIP_SUCCESS; }
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        std::vector<double> tmp(N);
        OPMHIP_HIP(c, hipMemcpy(tmp.data(), A.d_maxsw, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
        for (int pos = 0; pos < N; ++pos) max_sw[P.fromOrder[pos]] = tmp[pos];
        return OPMHIP_SUCCESS;
    });
}

int opmhip_begin_time_step(opmhip_ctx* c, double dt) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!(dt > 0.0)) return fail(c, OPMHIP_INVALID_ARGUMENT, "begin_time_step: dt must be positive");
        const bool limits = A.drsdt_on || A.drvdt_on;
        if ((A.drsdt_on && (!A.d_rsmax || !A.d_lastRs)) || (A.drvdt_on && (!A.d_rvmax || !A.d_lastRv)))
            return fail(c, OPMHIP_UNKNOWN_ERROR, "begin_time_step: DRSDT / DRVDT is in force but its cap array is gone (internal error)");
        if (!limits && !A.d_minpo && !A.d_maxso && !A.d_maxsw && !A.d_hyst) return OPMHIP_SUCCESS;
        if (!A.state_set) return fail(c, OPMHIP_NOT_READY, "begin_time_step before set_state");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        if (A.d_maxsw) launch_max_water_saturation(c, false);   // updateMaxWaterSaturation_ (eclproblem.hh:1056)
        if (A.d_minpo) launch_min_pressure(c, false);     // updateMinPressure_: from the intensive quantities of the state as it is
        if (A.d_hyst) launch_hyst_update(c);              // updateHysteresis_ (eclproblem.hh:1060, 2603-2626)
        if (A.d_maxso) launch_max_oil_saturation(c, false);   // updateMaxOilSaturation_
        A.storage_frozen = false;
        if (limits) {
            launch_set_limits(c, 0.0);                    // time index 1: lastRs / lastRv without the increment
            launch_iq_update(c);
            launch_storage_old(c);
            A.storage_frozen = true;
            launch_set_limits(c, dt);                     // time index 0: + DRSDT * dt
        }
        launch_iq_update(c);
        OPMHIP_HIP(c, hipGetLastError());
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_hysteresis(opmhip_ctx* c, int kr_model, const int* imbnum, const opmhip_endpoint_scaling* imb) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        const Pattern& P = c->pat;
        if (!A.static_set) return fail(c, OPMHIP_NOT_READY, "set_hysteresis before set_static");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        const int N = P.Nloc;
        if (kr_model < 0) {   // the keyword is not in force
            A.hyst_model = -1;
            dev_free(c, &A.d_hyst);
            dev_free(c, &A.d_imbnum);
            dev_free(c, &A.d_eps_imb);
        } else {
            if (kr_model > 1) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_hysteresis: EHYSTR item 2 = %d - only the Carlson models 0 and 1 are supported (as in the reference, PartiallySupportedFlowKeywords.cpp:301)", kr_model);
            if (!A.ext) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_hysteresis: needs a context with the extended record (a fluid with PVTG, ROCKTAB or pc_scaling)");
            if (!imbnum) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_hysteresis: imbnum is NULL");
            if (imb && !A.d_eps) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_hysteresis: scaled end points of the imbibition curves without opmhip_set_endpoint_scaling in force");
            for (int i = 0; i < N; ++i)
                if (imbnum[i] < 0 || imbnum[i] >= A.num_sat) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_hysteresis: imbnum[%d] out of range", i);
            int rc;
            if ((rc = upload_cells(c, &A.d_imbnum, imbnum))) return rc;
            if (imb) {
                std::vector<double> v((size_t)EPS_COUNT * N);   // field-major, internal order
                for (int pos = 0; pos < N; ++pos) {
                    const int i = P.fromOrder[pos];
                    for (int f = 0; f < EPS_COUNT; ++f) {
                        const double x = imb->points[f] ? imb->points[f][i] : A.sat_eps[(size_t)imbnum[i] * EPS_COUNT + f];
                        if (!std::isfinite(x)) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_hysteresis: imbibition end point %d of cell %d is not finite", f, i);
                        v[(size_t)f * N + pos] = x;
                    }
                }
                if (!A.d_eps_imb && (rc = dev_alloc(c, &A.d_eps_imb, v.size()))) return rc;
                OPMHIP_HIP(c, hipMemcpy(A.d_eps_imb, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
            } else dev_free(c, &A.d_eps_imb);
            // "nothing seen yet": turning points 2, shifts 0 - the first begin_time_step sets them (or opmhip_set_hysteresis_params)
            std::vector<double> h((size_t)4 * N, 0.0);
            for (int q = 0; q < N; ++q) h[q] = h[(size_t)2 * N + q] = 2.0;
            if (!A.d_hyst && (rc = dev_alloc(c, &A.d_hyst, h.size()))) return rc;
            OPMHIP_HIP(c, hipMemcpy(A.d_hyst, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
            A.hyst_model = kr_model;
        }
        if (A.state_set) {
            launch_iq_update(c);
            OPMHIP_HIP(c, hipGetLastError());
            OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        }
        return OPMHIP_SUCCESS;
    });
}

int opmhip_get_hysteresis(opmhip_ctx* c, double* sw_ow, double* delta_ow, double* sw_go, double* delta_go) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        const Pattern& P = c->pat;
        if (!A.d_hyst) return fail(c, OPMHIP_NOT_READY, "get_hysteresis: opmhip_set_hysteresis is not in force");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        const int N = P.Nloc;
        std::vector<double> h((size_t)4 * N);
        OPMHIP_HIP(c, hipMemcpy(h.data(), A.d_hyst, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        double* out[4] = {sw_ow, delta_ow, sw_go, delta_go};
        for (int f = 0; f < 4; ++f)
            if (out[f])
                for (int pos = 0; pos < N; ++pos) out[f][P.fromOrder[pos]] = h[(size_t)f * N + pos];
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_hysteresis_params(opmhip_ctx* c, const double* sw_ow, const double* sw_go) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.d_hyst) return fail(c, OPMHIP_NOT_READY, "set_hysteresis_params: opmhip_set_hysteresis is not in force");
        if (!sw_ow || !sw_go) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_hysteresis_params: null array");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));   // the staging area may still be read by an earlier download
        const size_t N = c->pat.Nloc;
        const std::vector<double> a = cells_to_internal(c->pat, sw_ow), b = cells_to_internal(c->pat, sw_go);
        OPMHIP_HIP(c, hipMemcpy(A.d_stage_cell, a.data(), N * sizeof(double), hipMemcpyHostToDevice));   // staged as [2][Nloc]
        OPMHIP_HIP(c, hipMemcpy(A.d_stage_cell + N, b.data(), N * sizeof(double), hipMemcpyHostToDevice));
        launch_hyst_update(c, A.d_stage_cell, A.d_stage_cell + N);
        if (A.state_set) launch_iq_update(c);
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_get_trackers(opmhip_ctx* c, double* last_rs, double* last_rv, double* min_po, double* max_so) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        const Pattern& P = c->pat;
        if (!A.static_set) return fail(c, OPMHIP_NOT_READY, "get_trackers before set_static");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        const int N = P.Nloc;
        std::vector<double> tmp(N);
        auto get = [&](double* out, const double* dev) -> int {
            if (!out) return OPMHIP_SUCCESS;
            if (!dev) { std::fill(out, out + N, 0.0); return OPMHIP_SUCCESS; }
            OPMHIP_HIP(c, hipMemcpy(tmp.data(), dev, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
            for (int pos = 0; pos < N; ++pos) out[P.fromOrder[pos]] = tmp[pos];
            return OPMHIP_SUCCESS;
        };
        int rc;
        if ((rc = get(last_rs, A.d_lastRs)) || (rc = get(last_rv, A.d_lastRv)) || (rc = get(min_po, A.d_minpo)) || (rc = get(max_so, A.d_maxso))) return rc;
        return OPMHIP_SUCCESS;
    });
}

int opmhip_end_time_step(opmhip_ctx* c, double dt) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        AsmDev& A = c->asmb;
        if (!A.assembled) return fail(c, OPMHIP_NOT_READY, "end_time_step before assemble: no residual of an accepted step on the device");
        if (!(dt > 0.0)) return fail(c, OPMHIP_INVALID_ARGUMENT, "end_time_step: dt must be positive");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        if (A.drsdt_on || A.drvdt_on) launch_last_rs_rv(c);   // updateCompositionChangeLimits_ (eclproblem.hh:1125)
        A.storage_frozen = false;
        if (!A.drift_enabled) return OPMHIP_SUCCESS;
        launch_drift_update(c, dt);
        OPMHIP_HIP(c, hipGetLastError());
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_drift_compensation(opmhip_ctx* c, int enable, double max_compensation) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (enable && !(max_compensation > 0.0)) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_drift_compensation: max_compensation must be positive");
        c->asmb.drift_enabled = enable != 0;
        if (enable) c->asmb.max_compensation = max_compensation;
        if (c->asmb.d_drift) {   // switching it on or off starts from a clean slate, like a fresh EclProblem
            OPMHIP_HIP(c, hipSetDevice(c->device));
            OPMHIP_HIP(c, hipMemsetAsync(c->asmb.d_drift, 0, (size_t)c->pat.Nloc * 3 * sizeof(double), c->stream));
        }
        return OPMHIP_SUCCESS;
    });
}

int opmhip_get_state(opmhip_ctx* c, double* pv, unsigned char* meaning) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->asmb.state_set) return fail(c, OPMHIP_NOT_READY, "get_state before set_state");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        AsmDev& A = c->asmb;
        if (pv) {
            launch_vec_to_natural(c, A.d_pv, c->d_stageV, c->pat.Nloc);
            OPMHIP_HIP(c, hipMemcpyAsync(pv, c->d_stageV, (size_t)c->pat.Nloc * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        if (meaning) {
            launch_u8_to_natural(c, A.d_meaning, A.d_stage_u8);
            OPMHIP_HIP(c, hipMemcpyAsync(meaning, A.d_stage_u8, (size_t)c->pat.Nloc, hipMemcpyDeviceToHost, c->stream));
        }
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_source(opmhip_ctx* c, const double* source, const double* dsource) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->asmb.static_set) return fail(c, OPMHIP_NOT_READY, "set_source before set_static");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        AsmDev& A = c->asmb;
        const size_t Nb = c->pat.Nloc;
        int rc;
        if (source) { if ((rc = upload_cells(c, &A.d_source, source, 3))) return rc; }
        else OPMHIP_HIP(c, hipMemsetAsync(A.d_source, 0, Nb * 3 * sizeof(double), c->stream));
        if (dsource) { if ((rc = upload_cells(c, &A.d_dsource, dsource, 9))) return rc; }
        else OPMHIP_HIP(c, hipMemsetAsync(A.d_dsource, 0, Nb * 9 * sizeof(double), c->stream));
        return OPMHIP_SUCCESS;
    });
}

// internal positions of n named cells on the device (d_cell_pos, grown on demand); the host copy stays valid until the caller's synchronise
static int stage_cell_positions(opmhip_ctx* c, const std::vector<int>& pos) {
    AsmDev& A = c->asmb;
    if (pos.size() > A.cell_pos_cap) {
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        dev_free(c, &A.d_cell_pos);
        A.cell_pos_cap = 0;
        const size_t cap = std::max<size_t>(256, 2 * pos.size());
        int rc = dev_alloc(c, &A.d_cell_pos, cap);
        if (rc) return rc;
        A.cell_pos_cap = cap;
    }
    OPMHIP_HIP(c, hipMemcpyAsync(A.d_cell_pos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    return OPMHIP_SUCCESS;
}

int opmhip_set_source_cells(opmhip_ctx* c, int n, const int* cells, const double* source, const double* dsource) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->asmb.static_set) return fail(c, OPMHIP_NOT_READY, "set_source_cells before set_static");
        if (n < 0 || (n > 0 && (!cells || !source))) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_source_cells: n = %d, cells / source missing", n);
        OPMHIP_HIP(c, hipSetDevice(c->device));
        AsmDev& A = c->asmb;
        const Pattern& P = c->pat;
        // distinct cells, each with the sum of what was named for it (two wells may perforate one cell), in the order of first mention
        std::vector<int> pos;
        std::vector<double> val;    // [distinct][3] then [distinct][9]
        std::vector<double> dval;
        {
            std::map<int, int> slot;
            for (int i = 0; i < n; ++i) {
                if (cells[i] < 0 || cells[i] >= P.Nloc) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_source_cells: cell %d of %d", cells[i], P.Nloc);
                auto it = slot.find(cells[i]);
                int sidx;
                if (it == slot.end()) {
                    sidx = (int)pos.size();
                    slot.emplace(cells[i], sidx);
                    pos.push_back(P.toOrder[cells[i]]);
                    val.insert(val.end(), 3, 0.0);
                    dval.insert(dval.end(), 9, 0.0);
                } else sidx = it->second;
                for (int q = 0; q < 3; ++q) {
                    if (!std::isfinite(source[(size_t)i * 3 + q])) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_source_cells: source of cell %d is not finite", cells[i]);
                    val[(size_t)sidx * 3 + q] += source[(size_t)i * 3 + q];
                }
                if (dsource)
                    for (int q = 0; q < 9; ++q) dval[(size_t)sidx * 9 + q] += dsource[(size_t)i * 9 + q];
            }
        }
        const size_t Nloc = P.Nloc, m = pos.size();
        OPMHIP_HIP(c, hipMemsetAsync(A.d_source, 0, Nloc * 3 * sizeof(double), c->stream));   // behind whatever assembly still reads them: same stream
        OPMHIP_HIP(c, hipMemsetAsync(A.d_dsource, 0, Nloc * 9 * sizeof(double), c->stream));
        if (m == 0) return OPMHIP_SUCCESS;
        const int rc = [&]() -> int {
            int r = stage_cell_positions(c, pos);
            if (r) return r;
            // values staged in d_stage_cell (Nloc * IQS doubles >= 12 per distinct cell)
            OPMHIP_HIP(c, hipMemcpyAsync(A.d_stage_cell, val.data(), m * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
            OPMHIP_HIP(c, hipMemcpyAsync(A.d_stage_cell + m * 3, dval.data(), m * 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
            launch_source_scatter(c, (int)m, A.d_cell_pos, A.d_stage_cell, A.d_stage_cell + m * 3);
            OPMHIP_HIP(c, hipGetLastError());
            return OPMHIP_SUCCESS;
        }();
        // the host vectors above end here - on the way out of a failure as well: no copy may still be reading them
        if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) return fail(c, OPMHIP_DEVICE_ERROR, "set_source_cells: hipStreamSynchronize failed");
        return rc;
    });
}

int opmhip_assemble(opmhip_ctx* c, double dt, int iteration, double* jac, double* residual) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->asmb.state_set) return fail(c, OPMHIP_NOT_READY, "assemble before set_state");
        if (!(dt > 0.0) || iteration < 0) return fail(c, OPMHIP_INVALID_ARGUMENT, "assemble: dt must be positive, iteration >= 0");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        c->asmb.last_dt = dt;
        c->asmb.last_iteration = iteration;
        launch_assemble(c, dt, iteration);
        OPMHIP_HIP(c, hipGetLastError());
        c->system_loaded = true;
        c->factored = false;
        c->asmb.assembled = true;
        if (jac) {
            launch_unpermute_blocks(c, c->d_A, c->d_stageA);
            OPMHIP_HIP(c, hipMemcpyAsync(jac, c->d_stageA, (size_t)c->pat.nnzb * BB * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        if (residual) {
            launch_vec_to_natural(c, c->d_b, c->d_stageV);
            OPMHIP_HIP(c, hipMemcpyAsync(residual, c->d_stageV, (size_t)c->pat.Nb * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        if (jac || residual) OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_iq_fields(opmhip_ctx* c) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return iq_doubles_per_cell(c) / 4;
}

int opmhip_get_iq(opmhip_ctx* c, double* out) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->asmb.state_set) return fail(c, OPMHIP_NOT_READY, "get_iq before set_state");
        if (!out) return fail(c, OPMHIP_INVALID_ARGUMENT, "get_iq: out == NULL");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        launch_iq_to_natural(c, c->asmb.d_stage_cell);
        OPMHIP_HIP(c, hipMemcpyAsync(out, c->asmb.d_stage_cell, (size_t)c->pat.Nloc * iq_doubles_per_cell(c) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_get_iq_cells(opmhip_ctx* c, int n, const int* cells, double* out) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->asmb.state_set) return fail(c, OPMHIP_NOT_READY, "get_iq_cells before set_state");
        if (n < 0 || (n > 0 && (!cells || !out))) return fail(c, OPMHIP_INVALID_ARGUMENT, "get_iq_cells: n = %d, cells / out missing", n);
        if (n == 0) return OPMHIP_SUCCESS;
        const Pattern& P = c->pat;
        if (n > P.Nloc) return fail(c, OPMHIP_INVALID_ARGUMENT, "get_iq_cells: %d cells asked for, the grid has %d (opmhip_get_iq copies them all)", n, P.Nloc);
        std::vector<int> pos(n);
        for (int i = 0; i < n; ++i) {
            if (cells[i] < 0 || cells[i] >= P.Nloc) return fail(c, OPMHIP_INVALID_ARGUMENT, "get_iq_cells: cell %d of %d", cells[i], P.Nloc);
            pos[i] = P.toOrder[cells[i]];
        }
        OPMHIP_HIP(c, hipSetDevice(c->device));
        const int rc = [&]() -> int {
            int r = stage_cell_positions(c, pos);
            if (r) return r;
            launch_iq_gather(c, n, c->asmb.d_cell_pos, c->asmb.d_stage_cell);
            OPMHIP_HIP(c, hipGetLastError());
            OPMHIP_HIP(c, hipMemcpyAsync(out, c->asmb.d_stage_cell, (size_t)n * iq_doubles_per_cell(c) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            return OPMHIP_SUCCESS;
        }();
        if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) return fail(c, OPMHIP_DEVICE_ERROR, "get_iq_cells: hipStreamSynchronize failed");   // `pos` ends here
        return rc;
    });
}

int opmhip_convergence(opmhip_ctx* c, double dt, double tol_cnv, double* out) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->asmb.assembled) return fail(c, OPMHIP_NOT_READY, "convergence before assemble");
        if (!out) return fail(c, OPMHIP_INVALID_ARGUMENT, "convergence: out == NULL");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        const int rcc = launch_convergence(c, dt, tol_cnv);
        if (rcc) return rcc;
        OPMHIP_HIP(c, hipGetLastError());
        double h[16];
        OPMHIP_HIP(c, hipMemcpyAsync(h, c->asmb.d_conv_out, 11 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < 11; ++i) out[i] = h[i];
        // CNV_c = B_avg dt maxCoeff ; MB_c = |B_avg R_sum| dt / pvSum   (flow/BlackoilModelEbos.hpp:797-801)
        for (int e = 0; e < 3; ++e) {
            out[11 + e] = h[6 + e] * dt * h[3 + e];
            out[14 + e] = std::fabs(h[6 + e] * h[e]) * dt / h[9];
        }
        return OPMHIP_SUCCESS;
    });
}

int opmhip_update(opmhip_ctx* c, const double* dx, double relax, int* num_switched) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->asmb.state_set) return fail(c, OPMHIP_NOT_READY, "update before set_state");
        if (!dx && !c->have_result) return fail(c, OPMHIP_NOT_READY, "update: dx == NULL but no solve result is resident");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        const double* d_dx = c->d_x;
        if (dx) {
            OPMHIP_HIP(c, hipMemcpyAsync(c->d_stageV, dx, (size_t)c->pat.Nb * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
            launch_vec_to_internal(c, c->d_stageV, c->d_t);
            d_dx = c->d_t;
        }
        launch_newton_update(c, d_dx, relax);
        int rcg = launch_ghost_refresh(c);
        if (rcg) return rcg;
        OPMHIP_HIP(c, hipGetLastError());
        if (num_switched) {
            OPMHIP_HIP(c, hipMemcpyAsync(num_switched, c->asmb.d_nswitched, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        }
        return OPMHIP_SUCCESS;
    });
}

}  // extern "C"
