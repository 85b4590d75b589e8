// Host-side logic of libopmhip under AddressSanitizer + UBSan + libstdc++'s container assertions (test infrastructure; built and run by
// tests/test_host_logic_sanitized.py with g++, no GPU): csrc/reorder.cpp (orderings, L/U split, tiles, launch schedules, stencil
// tables - index arithmetic a GPU test only sees through its results) and csrc/fluid_tables.cpp (table blobs).  The HIP runtime
// calls reorder.cpp makes (hipMalloc, hipMemcpy for the uploads at its end) are served from the host heap here, so that an upload that
// reads past a vector's end is seen as well.  Beside the sanitizers the harness checks what every ordering must satisfy: the
// permutations are inverse to each other, the reordered pattern is the natural one renamed, rows of one colour do not meet unless they
// belong to one chain (the property tests/test_graphcoloring.cpp:44-110 checks of the reference's colouring), the tiles cut [0, Nb) into
// pieces no longer than a wavefront can take, every launch schedule visits every tile once, every entry of the matrix has one place in
// L, U or on the diagonal.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../../opm-autodiff_amd/csrc/fluid_tables.hpp"
#include "../../opm-autodiff_amd/csrc/internal.hpp"

// ---- the HIP runtime entry points reorder.cpp links against, on the host heap --------------------------------------------------
extern "C" hipError_t hipMalloc(void** p, size_t n) {
    *p = std::malloc(n);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
extern "C" hipError_t hipFree(void* p) {
    std::free(p);
    return hipSuccess;
}
extern "C" hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    std::memcpy(d, s, n);
    return hipSuccess;
}
// (dev_alloc's debugging fill, internal.hpp: OPMHIP_POISON_ALLOC - not switched on here, but referenced)
extern "C" hipError_t hipMemset(void* d, int v, size_t n) {
    std::memset(d, v, n);
    return hipSuccess;
}
extern "C" hipError_t hipDeviceSynchronize() { return hipSuccess; }
extern "C" const char* hipGetErrorString(hipError_t) { return "host stand-in"; }

static int g_fail = 0;
#define CHECK(cond, ...)                                        \
    do {                                                        \
        if (!(cond)) {                                          \
            std::printf("FAILED %s:%d %s: ", __FILE__, __LINE__, #cond); \
            std::printf(__VA_ARGS__);                           \
            std::printf("\n");                                  \
            ++g_fail;                                           \
            return;                                             \
        }                                                       \
    } while (0)

struct Graph {
    int Nb = 0, Nghost = 0;
    std::vector<int> rowptr, col;
};

// 7-point grid in its natural order; owned part [0, nxo) of the x range when nxo < nx: the rest of the first layer behind the cut
// becomes ghost cells numbered behind the owned ones (a decomposed run's local pattern)
static Graph cartesian(int nx, int ny, int nz, int nxo = -1) {
    if (nxo < 0) nxo = nx;
    auto id = [&](int i, int j, int k) { return i + nxo * (j + ny * k); };
    Graph g;
    g.Nb = nxo * ny * nz;
    std::vector<int> ghostId((size_t)ny * nz, -1);
    std::vector<std::vector<int>> rows(g.Nb);
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nxo; ++i) {
                std::vector<int>& r = rows[id(i, j, k)];
                if (k > 0) r.push_back(id(i, j, k - 1));
                if (j > 0) r.push_back(id(i, j - 1, k));
                if (i > 0) r.push_back(id(i - 1, j, k));
                r.push_back(id(i, j, k));
                if (i + 1 < nxo) r.push_back(id(i + 1, j, k));
                else if (i + 1 < nx) {   // the neighbour behind the cut: a ghost cell
                    int& gidx = ghostId[j + (size_t)ny * k];
                    if (gidx < 0) gidx = g.Nb + g.Nghost++;
                    r.push_back(gidx);
                }
                if (j + 1 < ny) r.push_back(id(i, j + 1, k));
                if (k + 1 < nz) r.push_back(id(i, j, k + 1));
            }
    g.rowptr.push_back(0);
    for (auto& r : rows) {
        std::sort(r.begin(), r.end());
        g.col.insert(g.col.end(), r.begin(), r.end());
        g.rowptr.push_back((int)g.col.size());
    }
    return g;
}

// symmetric pattern with rows of 1 .. maxRow blocks: neighbours at small offsets plus a few long-range partners (the shape of a
// corner-point grid with faults and NNCs)
static Graph irregular(int Nb, int maxRow, unsigned seed) {
    std::mt19937 rng(seed);
    std::vector<std::set<int>> nb(Nb);
    for (int i = 0; i < Nb; ++i) nb[i].insert(i);
    const int offs[] = {1, 2, 3, 7, 19, 20, 131, 577};
    for (int i = 0; i < Nb; ++i) {
        const int want = 1 + (int)(rng() % (unsigned)maxRow);
        for (int o : offs) {
            if ((int)nb[i].size() >= want) break;
            const int j = i + o;
            if (j < Nb && (int)nb[j].size() < maxRow) { nb[i].insert(j); nb[j].insert(i); }
        }
        if (rng() % 50 == 0) {
            const int j = (int)(rng() % (unsigned)Nb);
            if (j != i && (int)nb[i].size() < maxRow && (int)nb[j].size() < maxRow) { nb[i].insert(j); nb[j].insert(i); }
        }
    }
    Graph g;
    g.Nb = Nb;
    g.rowptr.push_back(0);
    for (int i = 0; i < Nb; ++i) {
        g.col.insert(g.col.end(), nb[i].begin(), nb[i].end());
        g.rowptr.push_back((int)g.col.size());
    }
    return g;
}

// opmhip_default_config lives in capi.cpp beside the HIP calls; what reorder.cpp reads of the configuration is set here
static void default_config(opmhip_config* cfg) {
    std::memset(cfg, 0, sizeof *cfg);
    cfg->abi_version = OPMHIP_ABI_VERSION;
    cfg->reorder = OPMHIP_REORDER_AUTO;
}
static void release(opmhip_ctx& c) {
    for (void* p : c.allocs) std::free(p);
    c.allocs.clear();
}

static void check_pattern(const char* name, const Graph& g, int kind, int chain) {
    using namespace opmhip;
    opmhip_ctx c;
    default_config(&c.cfg);
    c.cfg.reorder = kind;
    c.cfg.chain_length = chain;
    const int nnzb = (int)g.col.size();
    const int rc = build_pattern(&c, g.Nb, g.Nghost, nnzb, g.rowptr.data(), g.col.data());
    struct Guard { opmhip_ctx& c; ~Guard() { release(c); } } guard{c};
    CHECK(rc == OPMHIP_SUCCESS, "%s kind %d chain %d: build_pattern -> %d (%s)", name, kind, chain, rc, c.err.c_str());
    const Pattern& P = c.pat;
    const int Nb = g.Nb, Nloc = g.Nb + g.Nghost;
    // permutations
    CHECK((int)P.toOrder.size() >= Nb && (int)P.fromOrder.size() >= Nb, "%s: permutation sizes", name);
    for (int i = 0; i < Nb; ++i) {
        CHECK(P.toOrder[i] >= 0 && P.toOrder[i] < Nb, "%s: toOrder[%d] = %d", name, i, P.toOrder[i]);
        CHECK(P.fromOrder[P.toOrder[i]] == i, "%s: fromOrder(toOrder(%d))", name, i);
    }
    // the reordered pattern is the natural one renamed, columns ascending, nnzMap a bijection
    CHECK((int)P.rowptr.size() == Nb + 1 && P.rowptr[0] == 0 && P.rowptr[Nb] == nnzb, "%s: rowptr ends", name);
    std::vector<char> seen(nnzb, 0);
    auto renamed = [&](int natCol) { return natCol < Nb ? P.toOrder[natCol] : ((int)P.toOrder.size() > natCol ? P.toOrder[natCol] : natCol); };
    for (int r = 0; r < Nb; ++r) {
        const int nat = P.fromOrder[r];
        CHECK(P.rowptr[r + 1] - P.rowptr[r] == g.rowptr[nat + 1] - g.rowptr[nat], "%s: row %d length", name, r);
        CHECK(P.col[P.diag[r]] == r && P.diag[r] >= P.rowptr[r] && P.diag[r] < P.rowptr[r + 1], "%s: diag of row %d", name, r);
        for (int k = P.rowptr[r]; k < P.rowptr[r + 1]; ++k) {
            CHECK(P.col[k] >= 0 && P.col[k] < Nloc, "%s: col[%d] = %d", name, k, P.col[k]);
            CHECK(k == P.rowptr[r] || P.col[k] > P.col[k - 1], "%s: row %d not ascending", name, r);
            const int kn = P.nnzMap[k];
            CHECK(kn >= g.rowptr[nat] && kn < g.rowptr[nat + 1] && !seen[kn], "%s: nnzMap[%d] = %d", name, k, kn);
            seen[kn] = 1;
            CHECK(renamed(g.col[kn]) == P.col[k], "%s: entry %d is not its natural entry renamed", name, k);
        }
    }
    // colours: a partition of the rows; rows of one colour meet only along a chain (chained orderings) or not at all
    const int ncol = P.numColors;
    CHECK(ncol >= 1 && (int)P.colorPrefix.size() == ncol + 1 && P.colorPrefix[0] == 0 && P.colorPrefix[ncol] == Nb, "%s: colour prefix", name);
    std::vector<int> colorOf(Nb);
    for (int cc = 0; cc < ncol; ++cc) {
        CHECK(P.colorPrefix[cc + 1] >= P.colorPrefix[cc], "%s: colour %d negative", name, cc);
        for (int r = P.colorPrefix[cc]; r < P.colorPrefix[cc + 1]; ++r) colorOf[r] = cc;
    }
    if (!P.chained)
        for (int r = 0; r < Nb; ++r)
            for (int k = P.rowptr[r]; k < P.rowptr[r + 1]; ++k)
                CHECK(P.col[k] >= Nb || P.col[k] == r || colorOf[P.col[k]] != colorOf[r], "%s kind %d: rows %d and %d share colour %d", name, kind, r, P.col[k], colorOf[r]);
    // L / U split: every entry has one place
    CHECK((int)P.fdest.size() == nnzb, "%s: fdest size", name);
    std::vector<char> inL(P.nl, 0), inU(P.nu, 0);
    CHECK((int)P.lrowptr.size() == Nb + 1 && (int)P.urowptr.size() == Nb + 1 && P.lrowptr[Nb] == P.nl && P.urowptr[Nb] == P.nu, "%s: L/U row pointers", name);
    for (int r = 0; r < Nb; ++r)
        for (int k = P.rowptr[r]; k < P.rowptr[r + 1]; ++k) {
            const int d = P.fdest[k];
            if (P.col[k] >= Nb || k == P.diag[r]) { CHECK(d == -1, "%s: entry %d (diagonal / ghost column) has fdest %d", name, k, d); continue; }
            if (P.col[k] < r) {
                CHECK(d >= P.lrowptr[r] && d < P.lrowptr[r + 1] && !inL[d] && P.lcol[d] == P.col[k], "%s: L place of entry %d", name, k);
                inL[d] = 1;
            } else {
                const int u = -2 - d;
                CHECK(d <= -2 && u >= P.urowptr[r] && u < P.urowptr[r + 1] && !inU[u] && P.ucol[u] == P.col[k], "%s: U place of entry %d", name, k);
                inU[u] = 1;
            }
        }
    CHECK(std::count(inL.begin(), inL.end(), 0) == 0 && std::count(inU.begin(), inU.end(), 0) == 0, "%s: L/U entries without a source", name);
    // tiles: consecutive pieces of [0, Nb), none across a colour boundary, none longer than a wavefront's tile
    const TileSet& T = P.tiles;
    CHECK(T.ntiles() >= 1 && T.row0.front() == 0 && T.row0.back() == Nb, "%s: tile ends", name);
    for (int t = 0; t < T.ntiles(); ++t) {
        const int r0 = T.row0[t], r1 = T.row0[t + 1];
        CHECK(r1 > r0 && r1 - r0 <= TILE_ROWS, "%s: tile %d has %d rows", name, t, r1 - r0);
        CHECK(colorOf[r0] == colorOf[r1 - 1], "%s: tile %d crosses a colour boundary", name, t);
        CHECK(r1 - r0 == 1 || P.rowptr[r1] - P.rowptr[r0] <= TILE_CAP_BLOCKS, "%s: tile %d has %d blocks", name, t, P.rowptr[r1] - P.rowptr[r0]);
    }
    // the product's schedule: every row in exactly one launch position; interior positions first and free of ghost columns
    CHECK((int)T.spmvSched.size() == 4 * T.nsched && T.nschedInt >= 0 && T.nschedInt <= T.nsched, "%s: schedule size", name);
    std::vector<char> rowSeen(Nb, 0);
    for (int p = 0; p < T.nsched; ++p) {
        const int r0 = T.spmvSched[4 * p], r1 = T.spmvSched[4 * p + 1];
        if (r1 <= r0) continue;   // padding
        CHECK(r0 >= 0 && r1 <= Nb && T.spmvSched[4 * p + 2] == P.rowptr[r0] && T.spmvSched[4 * p + 3] == P.rowptr[r1], "%s: schedule position %d", name, p);
        for (int r = r0; r < r1; ++r) {
            CHECK(!rowSeen[r], "%s: row %d in two launch positions", name, r);
            rowSeen[r] = 1;
            if (p < T.nschedInt && g.Nghost > 0)
                for (int k = P.rowptr[r]; k < P.rowptr[r + 1]; ++k) CHECK(P.col[k] < Nb, "%s: interior position %d reads ghost column %d", name, p, P.col[k]);
        }
    }
    CHECK(std::count(rowSeen.begin(), rowSeen.end(), 0) == 0, "%s: rows without a launch position", name);
    // the chain kernels' schedule: every chain-tile once per colour
    if (!T.ctSchedOff.empty()) {
        CHECK((int)T.ctSchedOff.size() == ncol + 1 && T.ctSchedOff[ncol] == (int)T.ctSched.size(), "%s: chain-tile schedule offsets", name);
        const int nct = (int)T.ctFirst.size() - 1;
        std::vector<char> ctSeen(std::max(nct, 0), 0);
        for (int v : T.ctSched)
            if (v >= 0) {
                CHECK(v < nct && !ctSeen[v], "%s: chain-tile %d scheduled twice or out of range", name, v);
                ctSeen[v] = 1;
            }
        CHECK(std::count(ctSeen.begin(), ctSeen.end(), 0) == 0, "%s: chain-tiles without a launch position", name);
    }
    // the rest of a row (what the half-product form streams): the entries that are NOT U entries, in the row's own order
    CHECK((int)P.rrowptr.size() == Nb + 1 && (int)P.rdest.size() == P.nnzb && P.nr == (int)P.rcol.size() && P.rrowptr[Nb] == P.nr, "%s: rest sizes", name);
    CHECK(P.nr + P.nu == P.nnzb, "%s: rest %d + U %d != %d blocks", name, P.nr, P.nu, P.nnzb);
    for (int r = 0; r < Nb; ++r) {
        int q = P.rrowptr[r];
        for (int k = P.rowptr[r]; k < P.rowptr[r + 1]; ++k) {
            if (P.fdest[k] <= -2) { CHECK(P.rdest[k] == -1, "%s: U entry %d has a rest place", name, k); continue; }
            CHECK(P.rdest[k] == q && q < P.rrowptr[r + 1] && P.rcol[q] == P.col[k], "%s: rest place of entry %d", name, k);
            ++q;
        }
        CHECK(q == P.rrowptr[r + 1], "%s: rest row %d has %d entries too many", name, r, P.rrowptr[r + 1] - q);
    }
    // U == upper(A): restated from the definition (an elimination step i <- i - L_ij U_j. touches a U entry of row i iff rows i and j share a column beyond i)
    {
        bool alias = true;
        std::vector<int> mark(P.Nloc > 0 ? P.Nloc : Nb, -1);
        for (int i = 0; i < Nb && alias; ++i) {
            for (int k = P.rowptr[i]; k < P.rowptr[i + 1]; ++k) if (P.col[k] < Nb && P.col[k] > i) mark[P.col[k]] = i;
            for (int k = P.rowptr[i]; k < P.rowptr[i + 1] && alias; ++k) {
                const int j = P.col[k];
                if (j >= i || j >= Nb) continue;
                for (int q = P.rowptr[j]; q < P.rowptr[j + 1]; ++q)
                    if (P.col[q] < Nb && P.col[q] > i && mark[P.col[q]] == i) { alias = false; break; }
            }
        }
        CHECK(alias == P.ualias, "%s: ualias %d, the definition says %d", name, (int)P.ualias, (int)alias);
    }
    // the rest product's schedule: rows once each over the part it covers, tables reproduce the columns
    const RestSched& R = P.rest;
    if (R.on) {
        CHECK(P.ualias && P.chained, "%s: rest schedule without its premises", name);
        CHECK((int)R.sched.size() == 4 * R.nsched && (int)R.table.size() >= 16 * R.nsched && (int)R.word.size() == Nb && (int)R.koff.size() == Nb, "%s: rest schedule sizes", name);
        CHECK(R.nschedInt >= 0 && R.nschedInt <= R.nsched && R.nsched % 8 == 0, "%s: rest schedule positions %d / %d", name, R.nschedInt, R.nsched);
        std::vector<char> seen(Nb, 0);
        for (int p = 0; p < R.nsched; ++p) {
            const int r0 = R.sched[4 * p], r1 = R.sched[4 * p + 1];
            if (r1 <= r0) continue;
            CHECK(r0 >= 0 && r1 <= Nb && r1 - r0 <= 64 && R.sched[4 * p + 2] == P.rrowptr[r0] && R.sched[4 * p + 3] == P.rrowptr[r1] && P.rrowptr[r1] - P.rrowptr[r0] <= TILE_CAP_BLOCKS,
                  "%s: rest position %d", name, p);
            for (int r = r0; r < r1; ++r) {
                CHECK(!seen[r], "%s: row %d in two rest positions", name, r);
                seen[r] = 1;
                const int len = P.rrowptr[r + 1] - P.rrowptr[r];
                CHECK(len <= 8 && (int)R.koff[r] == P.rrowptr[r] - P.rrowptr[r0], "%s: rest row %d: %d entries, first at %d", name, r, len, (int)R.koff[r]);
                for (int u = 0; u < 8; ++u) {
                    const unsigned idx = (R.word[r] >> (4 * u)) & 0xFu;
                    if (u >= len) { CHECK(idx == 15u, "%s: rest row %d slot %d not empty", name, r, u); continue; }
                    CHECK(idx < 15u && r + R.table[(size_t)16 * p + idx] == P.rcol[P.rrowptr[r] + u], "%s: rest row %d slot %d column", name, r, u);
                }
                if (g.Nghost > 0)
                    for (int k = P.rowptr[r]; k < P.rowptr[r + 1]; ++k) CHECK(P.col[k] < Nb, "%s: rest position %d (interior) reads ghost column %d", name, p, P.col[k]);
            }
        }
        // without ghosts every row is covered; with ghosts exactly the rows of the product's interior positions
        int want = 0;
        for (int p = 0; p < (g.Nghost > 0 ? T.nschedInt : T.nsched); ++p) want += std::max(0, T.spmvSched[4 * p + 1] - T.spmvSched[4 * p]);
        CHECK((int)std::count(seen.begin(), seen.end(), 1) == want, "%s: rest schedule covers %d rows of %d", name, (int)std::count(seen.begin(), seen.end(), 1), want);
    }
    std::printf("ok  %-34s Nb %7d ghosts %5d kind %d chain %2d -> kind in force %d, %d colours, %d tiles, stencil %d/%d, U==upper(A) %d, rest %d positions\n", name, Nb, g.Nghost,
                kind, chain, P.kindInForce, ncol, T.ntiles(), (int)T.stencil, (int)P.sweepStencil, (int)P.ualias, R.on ? R.nsched : 0);
}

static void check_refusals() {
    using namespace opmhip;
    auto run = [](const Graph& g, int expect, const char* what) {
        opmhip_ctx c;
        default_config(&c.cfg);
        const int rc = build_pattern(&c, g.Nb, g.Nghost, (int)g.col.size(), g.rowptr.data(), g.col.data());
        release(c);
        if (rc != expect) { std::printf("FAILED refusal '%s': %d instead of %d\n", what, rc, expect); ++g_fail; }
        else std::printf("ok  refused: %s (%s)\n", what, c.err.c_str());
    };
    Graph g = cartesian(3, 3, 2);
    Graph a = g; a.col[a.rowptr[4]] = 99;  // column out of range
    run(a, OPMHIP_INVALID_ARGUMENT, "column out of range");
    Graph b = g; std::swap(b.col[b.rowptr[4]], b.col[b.rowptr[4] + 1]);
    run(b, OPMHIP_INVALID_ARGUMENT, "columns not ascending");
    Graph d;   // a row without its diagonal block (linalg/ParallelOverlappingILU0.hpp:484-485)
    d.Nb = 2; d.rowptr = {0, 2, 3}; d.col = {0, 1, 0};
    run(d, OPMHIP_ANALYSIS_FAILED, "diagonal block missing");
    Graph e = g; e.rowptr[e.Nb] -= 1;
    run(e, OPMHIP_INVALID_ARGUMENT, "rows[] inconsistent with nnzb");
}

// ---- fluid tables ---------------------------------------------------------------------------------------------------------------
static void check_fluid() {
    using namespace opmhip;
    // two PVT regions and two saturation regions of SPE1-like shape (python/test_data/SPE1CASE1/SPE1CASE1.DATA:109-250, shortened), the
    // second with other lengths, plus PVTG and ROCKTAB: every branch of the blob builder
    const double pvtw[] = {277e5, 1.038, 4.67e-10, 0.318e-3, 0.0, 277e5, 1.02, 4.0e-10, 0.3e-3, 0.0};
    const double dens[] = {786.5, 1037.8, 0.97, 800.0, 1000.0, 1.0};
    const int pvdgPtr[] = {0, 4, 7};
    const double pvdg[] = {1e5, 0.9, 0.8e-5, 5e6, 0.02, 1.2e-5, 2e7, 0.005, 2e-5, 6e7, 0.002, 4e-5,
                           1e5, 1.0, 1e-5, 1e7, 0.01, 1.5e-5, 5e7, 0.003, 3e-5};
    const int nodePtr[] = {0, 3, 5};
    const double rs[] = {0.2, 60.0, 200.0, 1.0, 100.0};
    const int rowPtr[] = {0, 1, 2, 5, 6, 8};
    const double pvto[] = {1e5, 1.06, 1.0e-3, 1e7, 1.3, 0.8e-3, 3e7, 1.7, 0.5e-3, 4e7, 1.68, 0.52e-3, 6e7, 1.6, 0.6e-3,
                           1e5, 1.05, 1.1e-3, 2e7, 1.4, 0.7e-3, 5e7, 1.35, 0.8e-3};
    const int swofPtr[] = {0, 4, 7};
    const double swof[] = {0.12, 0, 1, 0, 0.3, 0.02, 0.6, 0, 0.7, 0.3, 0.05, 0, 1.0, 1.0, 0, 0,
                           0.2, 0, 1, 2e4, 0.6, 0.2, 0.2, 5e3, 1.0, 1.0, 0, 0};
    const int sgofPtr[] = {0, 4, 7};
    const double sgof[] = {0, 0, 1, 0, 0.1, 0.01, 0.7, 0, 0.5, 0.4, 0.1, 0, 0.88, 0.98, 0, 0,
                           0, 0, 1, 0, 0.4, 0.3, 0.2, 1e3, 0.8, 0.9, 0, 4e3};
    const int gNodePtr[] = {0, 2, 4};
    const double pg[] = {5e6, 3e7, 4e6, 2e7};
    const int gRowPtr[] = {0, 2, 4, 5, 7};
    const double pvtg[] = {2e-5, 0.02, 1.2e-5, 0.0, 0.021, 1.1e-5, 2e-4, 0.004, 2.5e-5, 0.0, 0.0042, 2.2e-5,
                           1e-5, 0.03, 1.0e-5, 1.5e-4, 0.006, 2.0e-5, 0.0, 0.0062, 1.9e-5};
    const int rockPtr[] = {0, 3, 5};
    const double rocktab[] = {1e5, 0.95, 0.9, 2e7, 1.0, 1.0, 5e7, 1.04, 1.1, 1e5, 0.98, 1.0, 4e7, 1.02, 1.0};
    opmhip_fluid f;
    std::memset(&f, 0, sizeof f);
    f.num_pvt = 2; f.num_sat = 2;
    f.pvtw = pvtw; f.density = dens; f.pvdg_ptr = pvdgPtr; f.pvdg = pvdg;
    f.pvto_node_ptr = nodePtr; f.pvto_rs = rs; f.pvto_row_ptr = rowPtr; f.pvto = pvto;
    f.swof_ptr = swofPtr; f.swof = swof; f.sgof_ptr = sgofPtr; f.sgof = sgof;
    f.rock_pref = 1e5; f.rock_cr = 4e-10;
    for (int variant = 0; variant < 4; ++variant) {
        opmhip_fluid v = f;
        if (variant & 1) { v.pvtg_node_ptr = gNodePtr; v.pvtg_pg = pg; v.pvtg_row_ptr = gRowPtr; v.pvtg = pvtg; }
        if (variant & 2) { v.num_rock = 2; v.rocktab_ptr = rockPtr; v.rocktab = rocktab; v.pc_scaling = 1; }
        FluidTables T;
        const std::string err = build_fluid_tables(&v, T);
        if (!err.empty()) { std::printf("FAILED fluid variant %d: %s\n", variant, err.c_str()); ++g_fail; continue; }
        // every offset of every descriptor points into its blob
        const PvtRegionDesc* pd = reinterpret_cast<const PvtRegionDesc*>(&T.idx[2]);
        bool ok = T.num_pvt == 2 && T.num_sat == 2 && T.wet_gas == (bool)(variant & 1);
        for (int r = 0; r < T.num_pvt && ok; ++r) {
            const PvtRegionDesc& d = pd[r];
            ok = d.gas_n >= 0 && d.sat_n >= 2 && d.o_nx >= 2 && d.water + 5 <= (int)T.dbl.size() && d.density + 3 <= (int)T.dbl.size() &&
                 d.sat_p + d.sat_n <= (int)T.dbl.size() && d.o_xs + d.o_nx <= (int)T.dbl.size() && d.o_yoff + d.o_nx + 1 <= (int)T.idx.size();
            if (ok) {
                const int last = T.idx[d.o_yoff + d.o_nx];
                ok = d.o_ys + last <= (int)T.dbl.size() && d.o_invB + last <= (int)T.dbl.size() && d.o_invBMu + last <= (int)T.dbl.size();
            }
        }
        for (int s = 0; s < T.num_sat && ok; ++s) {
            double e[EPS_COUNT];
            sat_end_points(T, s, e);
            ok = e[EPS_SWL] >= 0 && e[EPS_SWU] <= 1.0 && e[EPS_SWCR] >= e[EPS_SWL] && e[EPS_MAXKRW] > 0 && e[EPS_MAXKRG] > 0;
        }
        if (!ok) { std::printf("FAILED fluid variant %d: descriptor outside its blob\n", variant); ++g_fail; }
        else std::printf("ok  fluid tables variant %d: %zu doubles, %zu ints\n", variant, T.dbl.size(), T.idx.size());
    }
    // malformed input is refused with a text, not read past its end
    {
        opmhip_fluid v = f;
        const double badSwof[] = {0.3, 0, 1, 0, 0.2, 0.1, 0.5, 0, 1.0, 1.0, 0, 0, 0.2, 0, 1, 0, 0.6, 0.2, 0.2, 0, 1.0, 1.0, 0, 0};   // Sw descending
        const int p3[] = {0, 3, 6};
        v.swof = badSwof; v.swof_ptr = p3;
        FluidTables T;
        const std::string err = build_fluid_tables(&v, T);
        if (err.empty()) { std::printf("FAILED: descending SWOF accepted\n"); ++g_fail; }
        else std::printf("ok  refused: %s\n", err.c_str());
    }
    {
        opmhip_fluid v = f;
        v.pvdg = nullptr;   // dry gas needs PVDG
        FluidTables T;
        const std::string err = build_fluid_tables(&v, T);
        if (err.empty()) { std::printf("FAILED: missing PVDG accepted\n"); ++g_fail; }
        else std::printf("ok  refused: %s\n", err.c_str());
    }
}

int main() {
    const int kinds[] = {OPMHIP_REORDER_LEVEL_SCHEDULING, OPMHIP_REORDER_GRAPH_COLORING, OPMHIP_REORDER_GRAPH_COLORING_GREEDY, OPMHIP_REORDER_LINE_COLORING,
                         OPMHIP_REORDER_AUTO};
    struct Case { std::string name; Graph g; };
    std::vector<Case> cases;
    const int dims[][3] = {{1, 1, 1}, {2, 1, 1}, {1, 1, 7}, {1, 5, 1}, {3, 3, 3}, {10, 10, 3}, {24, 25, 15}, {17, 5, 9}, {33, 2, 40}, {40, 40, 40}};
    for (auto& d : dims) cases.push_back({"grid " + std::to_string(d[0]) + "x" + std::to_string(d[1]) + "x" + std::to_string(d[2]), cartesian(d[0], d[1], d[2])});
    cases.push_back({"subdomain 12(+ghosts)x10x9", cartesian(24, 10, 9, 12)});
    cases.push_back({"subdomain 1(+ghosts)x6x6", cartesian(2, 6, 6, 1)});       // every row touches a ghost: no interior tile
    cases.push_back({"subdomain 40(+ghosts)x40x30", cartesian(80, 40, 30, 40)});
    cases.push_back({"irregular 1 row", irregular(1, 4, 1)});
    cases.push_back({"irregular 37 rows", irregular(37, 12, 2)});
    cases.push_back({"irregular 3000 rows", irregular(3000, 12, 3)});
    cases.push_back({"irregular 44431 rows", irregular(44431, 12, 7)});
    cases.push_back({"irregular 20000 rows, rows to 20", irregular(20000, 20, 11)});   // rows longer than the stencil words and the chain kernels' chunk
    for (const Case& cs : cases)
        for (int kind : kinds) {
            check_pattern(cs.name.c_str(), cs.g, kind, 0);
            if (kind == OPMHIP_REORDER_LINE_COLORING)
                for (int chain : {1, 2, 5, 25, 64}) check_pattern(cs.name.c_str(), cs.g, kind, chain);
        }
    // the automatic choice at the sizes where it changes its mind (30 000 / 200 000 / 700 000 rows, include/opmhip.h)
    check_pattern("grid 31x31x32", cartesian(31, 31, 32), OPMHIP_REORDER_AUTO, 0);
    check_pattern("grid 60x60x60", cartesian(60, 60, 60), OPMHIP_REORDER_AUTO, 0);
    check_pattern("grid 90x90x90", cartesian(90, 90, 90), OPMHIP_REORDER_AUTO, 0);
    check_refusals();
    check_fluid();
    std::printf(g_fail ? "%d check(s) FAILED\n" : "all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
