// Host-side analysis done once per sparsity pattern (the reference's "analyse_matrix",
// bda/BILU0.cpp:51-158): ILU ordering, internal (reordered) block-CSR pattern, L/U split, device tiling.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <numeric>

#include "internal.hpp"

#ifndef OPMHIP_XCD_GROUP_DEFAULT
#define OPMHIP_XCD_GROUP_DEFAULT 8
#endif

namespace opmhip {
namespace {

void transpose_pattern(const Pattern& P, std::vector<int>& cptr, std::vector<int>& ridx);

// level(i) = 1 + max level over the rows j < i that row i is coupled to (Saad 11.6.3).  Couplings are taken
// from the row AND the column of i, as the reference's findLevelScheduling does with its CSR + CSC pair
// (bda/Reorder.cpp:266-318), so that both the forward and the backward sweep are race free on a
// structurally non-symmetric pattern.  For a symmetric pattern this is the plain row-dependency level.
void levels(const Pattern& P, std::vector<int>& color, int& ncol) {
    std::vector<int> cptr, ridx;
    transpose_pattern(P, cptr, ridx);
    color.assign(P.Nb, 0);
    ncol = 0;
    for (int i = 0; i < P.Nb; ++i) {
        int l = 0;
        for (int k = P.nat_rowptr[i]; k < P.nat_rowptr[i + 1]; ++k) {
            const int j = P.nat_col[k];
            if (j < i) l = std::max(l, color[j] + 1);
        }
        for (int k = cptr[i]; k < cptr[i + 1]; ++k) {
            const int j = ridx[k];
            if (j < i) l = std::max(l, color[j] + 1);
        }
        color[i] = l;
        ncol = std::max(ncol, l + 1);
    }
}

uint32_t mix(uint32_t i) {  // splitmix64 -> 31 bits: deterministic replacement for std::random_device weights
    uint64_t z = (uint64_t)i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z & 0x7fffffffu);
}

void transpose_pattern(const Pattern& P, std::vector<int>& cptr, std::vector<int>& ridx) {
    cptr.assign(P.Nb + 1, 0);
    ridx.resize(P.nnzb);
    for (int k = 0; k < P.nnzb; ++k) cptr[P.nat_col[k] + 1]++;
    std::partial_sum(cptr.begin(), cptr.end(), cptr.begin());
    std::vector<int> w(cptr.begin(), cptr.end() - 1);
    for (int i = 0; i < P.Nb; ++i)
        for (int k = P.nat_rowptr[i]; k < P.nat_rowptr[i + 1]; ++k) ridx[w[P.nat_col[k]]++] = i;
}

// Jones-Plassmann rounds (bda/Reorder.cpp:59-172): in round c an uncoloured node takes colour c when none
// of its neighbours (row or column) already has c and it carries the largest weight among its uncoloured
// neighbours.
void color_jp(const Pattern& P, std::vector<int>& color, int& ncol) {
    std::vector<int> cptr, ridx;
    transpose_pattern(P, cptr, ridx);
    color.assign(P.Nb, -1);
    std::vector<uint32_t> w(P.Nb);
    for (int i = 0; i < P.Nb; ++i) w[i] = mix((uint32_t)i);
    int left = P.Nb;
    ncol = 0;
    for (int c = 0; left > 0; ++c) {
        for (int i = 0; i < P.Nb; ++i) {
            if (color[i] != -1) continue;
            bool win = true;
            auto look = [&](const int* idx, int b, int e) {
                for (int k = b; k < e && win; ++k) {
                    const int j = idx[k];
                    if (j == i) continue;
                    const int jc = color[j];
                    if (jc == c) win = false;
                    else if (jc == -1 && (w[i] < w[j] || (w[i] == w[j] && i < j))) win = false;
                }
            };
            look(P.nat_col.data(), P.nat_rowptr[i], P.nat_rowptr[i + 1]);
            look(ridx.data(), cptr[i], cptr[i + 1]);
            if (win) {
                color[i] = c;
                --left;
            }
        }
        ncol = c + 1;
    }
}

void color_greedy(const Pattern& P, std::vector<int>& color, int& ncol) {
    std::vector<int> cptr, ridx;
    transpose_pattern(P, cptr, ridx);
    color.assign(P.Nb, -1);
    ncol = 0;
    std::vector<char> used;
    for (int i = 0; i < P.Nb; ++i) {
        used.assign(ncol + 1, 0);
        for (int k = P.nat_rowptr[i]; k < P.nat_rowptr[i + 1]; ++k)
            if (color[P.nat_col[k]] >= 0) used[color[P.nat_col[k]]] = 1;
        for (int k = cptr[i]; k < cptr[i + 1]; ++k)
            if (color[ridx[k]] >= 0) used[color[ridx[k]]] = 1;
        int c = 0;
        while (used[c]) ++c;
        color[i] = c;
        ncol = std::max(ncol, c + 1);
    }
}

// Sub-tiles (<= TILE_ROWS rows, <= TILE_CAP_BLOCKS blocks) grouped into chain-tiles: the sub-tiles of one chain-tile
// are the successive steps of up to TILE_ROWS chains and must be processed in order by one workgroup; different
// chain-tiles of a colour are independent.  Without chains every sub-tile is its own chain-tile.
void build_tiles(const std::vector<int>& rowptr, const std::vector<int>& colorPrefix, const std::vector<int>& stepPrefix,
                 TileSet& T) {
    // stepPrefix: row boundaries of (chain-tile, step) groups, ascending, containing every colour boundary; a value
    // of -1 marks the start of a new chain-tile (encoded as a parallel flag array to keep this simple)
    T.row0.clear();
    T.colorTile.clear();
    T.ctFirst.clear();
    T.colorCT.clear();
    (void)stepPrefix;
    const int ncol = (int)colorPrefix.size() - 1;
    for (int c = 0; c < ncol; ++c) {
        T.colorTile.push_back((int)T.row0.size());
        T.colorCT.push_back((int)T.ctFirst.size());
        int r = colorPrefix[c];
        const int rend = colorPrefix[c + 1];
        while (r < rend) {
            T.ctFirst.push_back((int)T.row0.size());
            T.row0.push_back(r);
            int e = r + 1;  // a tile always holds at least one row (an over-long row is read from HBM directly)
            while (e < rend && e - r < TILE_ROWS && rowptr[e + 1] - rowptr[r] <= TILE_CAP_BLOCKS) ++e;
            r = e;
        }
    }
    T.colorTile.push_back((int)T.row0.size());
    T.colorCT.push_back((int)T.ctFirst.size());
    T.ctFirst.push_back((int)T.row0.size());
    T.row0.push_back(colorPrefix[ncol]);
}

// Line colouring: chains of up to maxLen rows along each row's farthest neighbour (in the natural CpGrid order
// i + nx*(j + ny*k) that is the vertical neighbour, the strongest coupling of a thin-layered reservoir grid), chains
// coloured greedily so that chains of one colour share no coupling.  Inside a chain rows keep their natural order, so
// the factorisation is an exact ILU0 of the permuted matrix; it keeps most of the natural ordering's strength at
// two colours' worth of launches (tools/ordering_study.py: 12.5 BiCGStab iterations against 26.5 for red-black and
// 9.5 for the natural order on a 72^3 Jacobian).
struct Chains {
    std::vector<int> chainOf, posIn;          // per row
    std::vector<std::vector<int>> rows;       // per chain
};
void build_chains(const Pattern& P, int maxLen, Chains& C) {
    const int Nb = P.Nb;
    std::vector<int> succ(Nb, -1);
    for (int i = 0; i < Nb; ++i) {
        const int last = P.nat_col[P.nat_rowptr[i + 1] - 1];
        if (last > i && P.nat_col[P.nat_rowptr[last]] == i) succ[i] = last;  // mutual: i is the farthest lower neighbour of last
    }
    C.chainOf.assign(Nb, -1);
    C.posIn.assign(Nb, 0);
    C.rows.clear();
    for (int i = 0; i < Nb; ++i) {
        if (C.chainOf[i] >= 0) continue;
        const int id = (int)C.rows.size();
        C.rows.emplace_back();
        int cur = i;
        while (cur >= 0 && C.chainOf[cur] < 0 && (int)C.rows[id].size() < maxLen) {
            C.chainOf[cur] = id;
            C.posIn[cur] = (int)C.rows[id].size();
            C.rows[id].push_back(cur);
            cur = succ[cur];
        }
    }
}
void color_chains(const Pattern& P, const Chains& C, std::vector<int>& chainColor, int& ncol) {
    std::vector<int> cptr, ridx;
    transpose_pattern(P, cptr, ridx);
    const int nc = (int)C.rows.size();
    chainColor.assign(nc, -1);
    ncol = 0;
    std::vector<char> used;
    for (int id = 0; id < nc; ++id) {
        used.assign(ncol + 1, 0);
        for (int i : C.rows[id]) {
            for (int k = P.nat_rowptr[i]; k < P.nat_rowptr[i + 1]; ++k) {
                const int o = C.chainOf[P.nat_col[k]];
                if (o != id && chainColor[o] >= 0) used[chainColor[o]] = 1;
            }
            for (int k = cptr[i]; k < cptr[i + 1]; ++k) {
                const int o = C.chainOf[ridx[k]];
                if (o != id && chainColor[o] >= 0) used[chainColor[o]] = 1;
            }
        }
        int c = 0;
        while (used[c]) ++c;
        chainColor[id] = c;
        ncol = std::max(ncol, c + 1);
    }
}

// Launch schedules.  The hardware deals consecutive workgroups of a launch round-robin over the 8 XCDs, each with an L2 of
// its own.  With the identity map (workgroup b = tile b) a line of the gathered vector is wanted by ~7 tiles that land
// on ~5 different XCDs and is fetched from the fabric by each (PMC: the 24 MB input vector of the 100^3 SpMV costs
// 116 MB).  The schedule keeps the streaming shape (all 8 XCDs walk through memory side by side) but hands every XCD
// GROUPS of `G` consecutive chain-tiles of every colour at the same relative position of their colour - the same stretch
// of the grid - so that most neighbours of a tile are gathered through the L2 that already holds them.
// Results do not depend on the schedule (each tile's arithmetic is its own); only the order of the partial sums does.
// Launch schedule of the rest product (internal.hpp: RestSched).  The tiles follow the SpMV's launch order - position b and b + 8 of it are
// consecutive tiles of one stretch of the grid on one XCD - and consecutive tiles are merged while the rows stay contiguous, the tile holds
// at most 64 rows (one per lane) and TILE_CAP_BLOCKS blocks and its rows share at most 15 column offsets: a row of the rest has 2 - 6 blocks
// where the full row has 7, so a tile of the SpMV's 32 rows would stream 4.6 - 13.8 KB per step of the pipelined kernel instead of 16.
// Off (R.on = false) unless the pattern has the property the product rests on (Pattern::ualias), the ordering is line-coloured (the chain
// sweeps are the ones that emit the row sums) and every tile fits the stencil form.
void build_rest_schedule(Pattern& P, const std::vector<int>& order) {
    RestSched& R = P.rest;
    const TileSet& T = P.tiles;
    R.on = false;
    if (!P.ualias || !P.chained) return;
    struct Rt { int r0, r1; };
    auto offsets_of = [&](int r0, int r1, std::vector<int>& offs) {
        offs.clear();
        for (int r = r0; r < r1; ++r)
            for (int k = P.rrowptr[r]; k < P.rrowptr[r + 1]; ++k) offs.push_back(P.rcol[k] - r);
        std::sort(offs.begin(), offs.end());
        offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
    };
    std::vector<int> offs;
    static const int maxRows = [] { const char* e = tuning_env("OPMHIP_REST_ROWS"); const int v = e ? std::atoi(e) : 64; return v < 1 ? 1 : (v > 64 ? 64 : v); }();   // measurement switch: rows per tile at most
    static const int maxBlocks = [] { const char* e = tuning_env("OPMHIP_REST_BLOCKS"); const int v = e ? std::atoi(e) : TILE_CAP_BLOCKS; return v < 8 ? 8 : (v > TILE_CAP_BLOCKS ? TILE_CAP_BLOCKS : v); }();
    std::vector<Rt> out;   // launch positions of both parts, padding = {0, 0}
    int nInt = 0;
    // Decomposed runs (ghost columns): the INTERIOR tiles take this form; the boundary tiles - rows with ghost columns at arbitrary offsets,
    // which rarely fit a table of 15 - keep the whole product over the matrix itself (launch_spmv), so only part 0 is built
    const int nparts = P.Nghost > 0 ? 1 : 2;
    for (int part = 0; part < nparts; ++part) {
        const int p0 = part == 0 ? 0 : T.nschedInt, p1 = part == 0 ? T.nschedInt : T.nsched;
        std::vector<std::vector<Rt>> lists(8);
        for (int k = 0; k < 8; ++k) {
            Rt cur{0, 0};
            for (int b = p0 + k; b < p1; b += 8) {
                if (order[b] < 0) continue;
                const int r0 = T.row0[order[b]], r1 = T.row0[order[b] + 1];
                if (r1 <= r0) continue;
                if (cur.r1 > cur.r0 && r0 == cur.r1 && r1 - cur.r0 <= maxRows && P.rrowptr[r1] - P.rrowptr[cur.r0] <= maxBlocks) {
                    offsets_of(cur.r0, r1, offs);
                    if (offs.size() <= 15) { cur.r1 = r1; continue; }
                }
                if (cur.r1 > cur.r0) lists[k].push_back(cur);
                cur = Rt{r0, r1};
            }
            if (cur.r1 > cur.r0) lists[k].push_back(cur);
        }
        size_t L = 0;
        for (const auto& l : lists) L = std::max(L, l.size());
        for (size_t j = 0; j < L; ++j)
            for (int k = 0; k < 8; ++k) out.push_back(j < lists[k].size() ? lists[k][j] : Rt{0, 0});
        if (part == 0) nInt = (int)out.size();
    }
    R.nsched = (int)out.size();
    R.nschedInt = P.Nghost > 0 ? R.nsched : nInt;
    R.sched.assign((size_t)4 * R.nsched, 0);
    R.word.assign(P.Nb, 0xFFFFFFFFu);
    R.koff.assign(P.Nb, 0);
    R.table.assign((size_t)16 * std::max(1, R.nsched), 0);
    for (int b = 0; b < R.nsched; ++b) {
        const int r0 = out[b].r0, r1 = out[b].r1;
        if (r1 <= r0) continue;
        R.sched[4 * b] = r0; R.sched[4 * b + 1] = r1;
        R.sched[4 * b + 2] = P.rrowptr[r0]; R.sched[4 * b + 3] = P.rrowptr[r1];
        if (P.rrowptr[r1] - P.rrowptr[r0] > TILE_CAP_BLOCKS || r1 - r0 > 64) return;   // a single SpMV tile too large for the form: off
        offsets_of(r0, r1, offs);
        if (offs.size() > 15) return;
        for (size_t q = 0; q < offs.size(); ++q) R.table[(size_t)16 * b + q] = offs[q];
        for (int r = r0; r < r1; ++r) {
            const int len = P.rrowptr[r + 1] - P.rrowptr[r], ko = P.rrowptr[r] - P.rrowptr[r0];
            if (len > 8 || ko > 255) return;
            unsigned w = 0xFFFFFFFFu;
            for (int u = 0; u < len; ++u) {
                const int idx = (int)(std::lower_bound(offs.begin(), offs.end(), P.rcol[P.rrowptr[r] + u] - r) - offs.begin());
                w = (w & ~(0xFu << (4 * u))) | ((unsigned)idx << (4 * u));
            }
            R.word[r] = w;
            R.koff[r] = (unsigned char)ko;
        }
    }
    R.on = R.nsched > 0;
}

void build_schedules(Pattern& P, int G) {
    TileSet& T = P.tiles;
    const int ncol = P.numColors;
    const int nt = T.ntiles();
    auto deal = [](const std::vector<std::vector<int>>& lists, int pad, std::vector<int>& out) {
        size_t L = 0;
        for (const auto& l : lists) L = std::max(L, l.size());
        for (size_t j = 0; j < L; ++j)
            for (int k = 0; k < 8; ++k) out.push_back(j < lists[k].size() ? lists[k][j] : pad);
    };
    T.ctSched.clear();
    T.ctSchedOff.assign(ncol + 1, 0);
    std::vector<int> order;  // SpMV: tiles in launch order (-1 = padding)
    const bool grouped = P.chained && G > 0;
    if (!grouped) {
        for (int t = 0; t < nt; ++t) order.push_back(t);
        while (order.size() % 8) order.push_back(-1);
        for (int c = 0; c < ncol; ++c) {
            for (int q = T.colorCT[c]; q < T.colorCT[c + 1]; ++q) T.ctSched.push_back(q);
            while (T.ctSched.size() % 8) T.ctSched.push_back(-1);
            T.ctSchedOff[c + 1] = (int)T.ctSched.size();
        }
    } else {
        int maxct = 0;
        for (int c = 0; c < ncol; ++c) maxct = std::max(maxct, T.colorCT[c + 1] - T.colorCT[c]);
        const int NG = std::max(1, (maxct + G - 1) / G);
        // group of chain-tile q (0-based inside colour c): by relative position, so that the colours line up in space
        auto group_of = [&](int c, int q) { return (int)(((long long)q * NG) / std::max(1, T.colorCT[c + 1] - T.colorCT[c])); };
        std::vector<std::vector<int>> spmvLists(8);
        std::vector<std::vector<std::vector<int>>> byGroup(ncol, std::vector<std::vector<int>>(NG));
        for (int c = 0; c < ncol; ++c)
            for (int q = T.colorCT[c]; q < T.colorCT[c + 1]; ++q) byGroup[c][group_of(c, q - T.colorCT[c])].push_back(q);
        for (int g = 0; g < NG; ++g) {
            size_t m = 0;
            for (int c = 0; c < ncol; ++c) m = std::max(m, byGroup[c][g].size());
            for (size_t i = 0; i < m; ++i)   // colours interleaved chain-tile by chain-tile
                for (int c = 0; c < ncol; ++c)
                    if (i < byGroup[c][g].size())
                        for (int t = T.ctFirst[byGroup[c][g][i]]; t < T.ctFirst[byGroup[c][g][i] + 1]; ++t) spmvLists[g % 8].push_back(t);
        }
        deal(spmvLists, -1, order);
        for (int c = 0; c < ncol; ++c) {
            std::vector<std::vector<int>> lists(8);
            for (int g = 0; g < NG; ++g)
                for (int q : byGroup[c][g]) lists[g % 8].push_back(q);
            deal(lists, -1, T.ctSched);
            T.ctSchedOff[c + 1] = (int)T.ctSched.size();
        }
    }
    // descriptor records of the chain kernels' launch positions
    {
        int S = 1;
        for (size_t q = 0; q + 1 < T.ctFirst.size(); ++q) S = std::max(S, T.ctFirst[q + 1] - T.ctFirst[q]);
        T.descS1 = S + 1;
        T.descStride = 4 * ((4 + 3 * T.descS1 + 3) / 4);
        T.ctDesc.assign((size_t)T.ctSched.size() * T.descStride, 0);
        for (size_t pos = 0; pos < T.ctSched.size(); ++pos) {
            int* rec = &T.ctDesc[pos * T.descStride];
            const int ct = T.ctSched[pos];
            rec[1] = ct;
            if (ct < 0) continue;
            const int q0 = T.ctFirst[ct], n = T.ctFirst[ct + 1] - q0;
            rec[0] = n;
            rec[2] = q0;   // first tile of the chain-tile: where its steps' stencil tables start
            for (int i = 0; i <= n; ++i) {
                const int r = T.row0[q0 + i];
                rec[4 + i] = r;
                rec[4 + T.descS1 + i] = P.lrowptr[r];
                rec[4 + 2 * T.descS1 + i] = P.urowptr[r];
            }
        }
    }
    // decomposed runs: interior tiles first, boundary tiles (a row with a ghost column) behind them, each part keeping the
    // dealing over the 8 XCD columns (position % 8) it had
    T.nschedInt = (int)order.size();
    if (P.Nghost > 0) {
        auto boundary = [&](int t) {
            for (int k = P.rowptr[T.row0[t]]; k < P.rowptr[T.row0[t + 1]]; ++k)
                if (P.col[k] >= P.Nb) return true;
            return false;
        };
        std::vector<std::vector<int>> in8(8), bd8(8);
        for (size_t b = 0; b < order.size(); ++b)
            if (order[b] >= 0) (boundary(order[b]) ? bd8 : in8)[b % 8].push_back(order[b]);
        std::vector<int> o2;
        deal(in8, -1, o2);
        T.nschedInt = (int)o2.size();
        deal(bd8, -1, o2);
        order.swap(o2);
    }
    T.nsched = (int)order.size();
    T.spmvSched.assign((size_t)4 * T.nsched, 0);
    for (int b = 0; b < T.nsched; ++b)
        if (order[b] >= 0) {
            const int r0 = T.row0[order[b]], r1 = T.row0[order[b] + 1];
            T.spmvSched[4 * b] = r0; T.spmvSched[4 * b + 1] = r1;
            T.spmvSched[4 * b + 2] = P.rowptr[r0]; T.spmvSched[4 * b + 3] = P.rowptr[r1];
        }
    // stencil form of the index streams (internal.hpp: TileSet::stWord ...), judged per part of the schedule: the interior tiles of a
    // decomposed run are as regular as a single domain's, its boundary tiles (ghost columns at arbitrary offsets) usually are not
    T.stencilPart[0] = T.stencilPart[1] = true;
    T.stWord.assign(P.Nb, 0xFFFFFFFFu);
    T.stKoff.assign(P.Nb, 0);
    T.stTable.assign((size_t)16 * T.nsched, 0);
    for (int b = 0; b < T.nsched; ++b) {
        if (order[b] < 0) continue;
        const int part = b < T.nschedInt ? 0 : 1;
        if (!T.stencilPart[part]) continue;
        const int r0 = T.row0[order[b]], r1 = T.row0[order[b] + 1];
        std::vector<int> offs;
        for (int r = r0; r < r1; ++r)
            for (int k = P.rowptr[r]; k < P.rowptr[r + 1]; ++k) offs.push_back(P.col[k] - r);
        std::sort(offs.begin(), offs.end());
        offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
        if (offs.size() > 15) { T.stencilPart[part] = false; continue; }
        for (size_t q = 0; q < offs.size(); ++q) T.stTable[(size_t)16 * b + q] = offs[q];
        for (int r = r0; r < r1; ++r) {
            const int len = P.rowptr[r + 1] - P.rowptr[r], ko = P.rowptr[r] - P.rowptr[r0];
            if (len > 8 || ko > 255) { T.stencilPart[part] = false; break; }
            unsigned w = 0xFFFFFFFFu;
            for (int u = 0; u < len; ++u) {
                const int idx = (int)(std::lower_bound(offs.begin(), offs.end(), P.col[P.rowptr[r] + u] - r) - offs.begin());
                w = (w & ~(0xFu << (4 * u))) | ((unsigned)idx << (4 * u));
            }
            T.stWord[r] = w;
            T.stKoff[r] = (unsigned char)ko;
        }
    }
    T.stencil = T.stencilPart[0] || T.stencilPart[1];
    build_rest_schedule(P, order);
    // the same for the two factor parts the sweeps stream (chained orderings only: their kernels are the ones that read it)
    P.sweepStencil = P.chained;
    for (int part = 0; part < 2 && P.sweepStencil; ++part) {
        const std::vector<int>& prow = part == 0 ? P.lrowptr : P.urowptr;
        const std::vector<int>& pcol = part == 0 ? P.lcol : P.ucol;
        P.swWord[part].assign(P.Nb, 0xFFFFFFFFu);
        P.swKoff[part].assign(P.Nb, 0);
        P.swTable[part].assign((size_t)16 * nt, 0);
        for (int t = 0; t < nt && P.sweepStencil; ++t) {
            const int r0 = T.row0[t], r1 = T.row0[t + 1];
            std::vector<int> offs;
            for (int r = r0; r < r1; ++r)
                for (int k = prow[r]; k < prow[r + 1]; ++k) offs.push_back(pcol[k] - r);
            std::sort(offs.begin(), offs.end());
            offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
            if (offs.size() > 15) { P.sweepStencil = false; break; }
            for (size_t q = 0; q < offs.size(); ++q) P.swTable[part][(size_t)16 * t + q] = offs[q];
            for (int r = r0; r < r1; ++r) {
                const int len = prow[r + 1] - prow[r], ko = prow[r] - prow[r0];
                if (len > 8 || ko > 255) { P.sweepStencil = false; break; }
                unsigned w = 0xFFFFFFFFu;
                for (int u = 0; u < len; ++u) {
                    const int idx = (int)(std::lower_bound(offs.begin(), offs.end(), pcol[prow[r] + u] - r) - offs.begin());
                    w = (w & ~(0xFu << (4 * u))) | ((unsigned)idx << (4 * u));
                }
                P.swWord[part][r] = w;
                P.swKoff[part][r] = (unsigned char)ko;
            }
        }
    }
}

}  // namespace

int build_pattern(opmhip_ctx* c, int Nb, int Nghost, int nnzb, const int* rows, const int* cols) {
    // Nb = owned block rows (the rows of the matrix); columns may also point at Nghost ghost cells numbered
    // Nb .. Nb+Nghost-1 (owner-cells-first ordering, linalg/ISTLSolverEbos.hpp:171-178).  Ghost columns take part in
    // SpMV and assembly only; the ILU0 is the block-Jacobi one of ghost_last_bilu0_decomposition: it never sees them.
    Pattern& P = c->pat;
    if (Nb <= 0 || Nghost < 0 || nnzb <= 0 || !rows || !cols) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_pattern: bad arguments");
    if (rows[0] != 0 || rows[Nb] != nnzb) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_pattern: rows[] inconsistent with nnzb");
    const int Nloc = Nb + Nghost;
    P.Nb = Nb;
    P.Nghost = Nghost;
    P.Nloc = Nloc;
    P.nnzb = nnzb;
    P.maxRowBlocks = 0;
    for (int i = 0; i < Nb; ++i) P.maxRowBlocks = std::max(P.maxRowBlocks, rows[i + 1] - rows[i]);
    P.nat_rowptr.assign(rows, rows + Nb + 1);
    P.nat_col.assign(cols, cols + nnzb);
    for (int i = 0; i < Nb; ++i) {
        bool hasDiag = false;
        for (int k = rows[i]; k < rows[i + 1]; ++k) {
            if (cols[k] < 0 || cols[k] >= Nloc) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_pattern: column out of range in row %d", i);
            if (k > rows[i] && cols[k] <= cols[k - 1]) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_pattern: columns of row %d not ascending", i);
            hasDiag |= (cols[k] == i);
        }
        // "diagonal entry missing" (linalg/ParallelOverlappingILU0.hpp:484-485) -> analysis failure
        if (!hasDiag) return fail(c, OPMHIP_ANALYSIS_FAILED, "set_pattern: row %d has no diagonal block", i);
    }
    std::vector<int> color;
    int ncol = 0;
    Chains CH;
    std::vector<int> chainColor;
    // OPMHIP_REORDER_AUTO: the line colouring where it pays, the greedy colouring elsewhere.  Measured (tools/ordering_by_size.py,
    // tools/config_rates.py): the chain kernels walk a chain-tile's 8-10 steps one after the other, each step a few
    // dependent rounds of loads - on 10^6 rows eight workgroups per CU hide that, on a small system nothing does (44 777-cell corner-point
    // grid: 31 Newton its/s line-coloured, 259 greedy; regular grids: the greedy colouring wins below ~125 000 cells although it needs
    // twice the iterations - with chains of 10; shorter chains move that point down), and on an irregular pattern the sweeps lose their
    // stencil form as well.  Line colouring: at least AUTO_LINE_MIN_ROWS rows, rows of at most 8 blocks, at most 15 distinct column
    // offsets (col - row) in the natural order - a structured grid handed over in its natural order; the chain length grows with the size.
    int kind = c->cfg.reorder;
    int autoChain = 10;
    if (kind == OPMHIP_REORDER_AUTO) {
        // structured grids (tools/chain_by_size.py, Newton its/s): 32^3 greedy 568, chains of 4: 583; 40^3 423 / 487 (chains of 3-4); 50^3
        // 267 / 361 (4); 64^3 165 / 246 (8; 239 with 4, 208 with 10); 80^3 99 / 168 (8; 153 with 10); 100^3: 10 (section 5) - the shorter
        // the chains, the fewer dependent steps a launch walks through, the more iterations
        constexpr int AUTO_LINE_MIN_ROWS = 30000;
        autoChain = Nb < 200000 ? 4 : Nb < 700000 ? 8 : 10;
        bool regular = Nb >= AUTO_LINE_MIN_ROWS;
        std::vector<int> offs;
        for (int i = 0; i < Nb && regular; ++i) {
            if (rows[i + 1] - rows[i] > 8) regular = false;
            for (int k = rows[i]; k < rows[i + 1] && regular; ++k) {
                if (cols[k] >= Nb) continue;
                const int o = cols[k] - i;
                if (std::find(offs.begin(), offs.end(), o) == offs.end()) {
                    offs.push_back(o);
                    if (offs.size() > 15) regular = false;
                }
            }
        }
        kind = regular ? OPMHIP_REORDER_LINE_COLORING : OPMHIP_REORDER_GRAPH_COLORING_GREEDY;
        if (c->cfg.verbosity > 0) std::fprintf(stderr, "opmhip: reorder auto -> %s (%d rows)\n", regular ? "line_coloring" : "graph_coloring_greedy", Nb);
    }
    const bool chained = (kind == OPMHIP_REORDER_LINE_COLORING);
    int maxLen = 1;
    // the ordering is computed on the owned-owned couplings only
    Pattern Q;
    Q.Nb = Nb;
    Q.nat_rowptr.assign(Nb + 1, 0);
    for (int i = 0; i < Nb; ++i) {
        for (int k = rows[i]; k < rows[i + 1]; ++k)
            if (cols[k] < Nb) Q.nat_col.push_back(cols[k]);
        Q.nat_rowptr[i + 1] = (int)Q.nat_col.size();
    }
    Q.nnzb = (int)Q.nat_col.size();
    switch (kind) {
        case OPMHIP_REORDER_LEVEL_SCHEDULING: levels(Q, color, ncol); break;
        case OPMHIP_REORDER_GRAPH_COLORING: color_jp(Q, color, ncol); break;
        case OPMHIP_REORDER_GRAPH_COLORING_GREEDY: color_greedy(Q, color, ncol); break;
        case OPMHIP_REORDER_LINE_COLORING: {
            maxLen = c->cfg.chain_length > 0 ? c->cfg.chain_length : (c->cfg.reorder == OPMHIP_REORDER_AUTO ? autoChain : 8);
            build_chains(Q, maxLen, CH);
            color_chains(Q, CH, chainColor, ncol);
            color.resize(Nb);
            for (int i = 0; i < Nb; ++i) color[i] = chainColor[CH.chainOf[i]];
        } break;
        default: return fail(c, OPMHIP_INVALID_ARGUMENT, "unknown reorder kind %d", c->cfg.reorder);
    }
    P.numColors = ncol;
    P.chained = chained;
    P.kindInForce = kind;
    P.chainLen = chained ? maxLen : 0;
    P.colorPrefix.assign(ncol + 1, 0);
    for (int i = 0; i < Nb; ++i) P.colorPrefix[color[i] + 1]++;
    std::partial_sum(P.colorPrefix.begin(), P.colorPrefix.end(), P.colorPrefix.begin());
    P.toOrder.resize(Nloc);
    P.fromOrder.resize(Nloc);
    for (int g = Nb; g < Nloc; ++g) P.toOrder[g] = P.fromOrder[g] = g;  // ghosts stay last, in the given order
    // (chain-tile, step) row groups of the chained ordering: rows [groupRow[g], groupRow[g+1]) ; groupNewCT[g] = 1 if the
    // group opens a new chain-tile
    std::vector<int> groupRow, groupNewCT;
    if (!chained) {
        // rows keep their natural relative order inside a colour (colorsToReordering, bda/Reorder.cpp:212-226)
        std::vector<int> next(P.colorPrefix.begin(), P.colorPrefix.end() - 1);
        for (int i = 0; i < Nb; ++i) {
            const int p = next[color[i]]++;
            P.toOrder[i] = p;
            P.fromOrder[p] = i;
        }
    } else {
        // colour -> chain-tile of TILE_ROWS chains (longest chains first, so that the live chains of a step are a
        // prefix) -> step -> chain
        int p = 0;
        for (int cc = 0; cc < ncol; ++cc) {
            std::vector<int> ids;
            for (int id = 0; id < (int)CH.rows.size(); ++id)
                if (chainColor[id] == cc) ids.push_back(id);
            std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return CH.rows[a].size() > CH.rows[b].size(); });
            for (size_t b0 = 0; b0 < ids.size(); b0 += TILE_ROWS) {
                const size_t b1 = std::min(ids.size(), b0 + (size_t)TILE_ROWS);
                const int steps = (int)CH.rows[ids[b0]].size();
                for (int st = 0; st < steps; ++st) {
                    groupRow.push_back(p);
                    groupNewCT.push_back(st == 0 ? 1 : 0);
                    for (size_t b = b0; b < b1; ++b) {
                        if ((int)CH.rows[ids[b]].size() <= st) break;
                        const int i = CH.rows[ids[b]][st];
                        P.toOrder[i] = p;
                        P.fromOrder[p] = i;
                        ++p;
                    }
                }
            }
        }
        groupRow.push_back(p);
        if (p != Nb) return fail(c, OPMHIP_ANALYSIS_FAILED, "line colouring lost rows (%d of %d)", p, Nb);
    }
    // internal pattern: row p = natural row fromOrder[p], columns renamed and re-sorted
    // (reorderBlockedMatrixByPattern, bda/Reorder.cpp:179-207) - done once here for the pattern; values follow
    // on the device through nnzMap.
    P.rowptr.assign(Nb + 1, 0);
    P.col.resize(nnzb);
    P.nnzMap.resize(nnzb);
    P.diag.assign(Nb, -1);
    std::vector<std::pair<int, int>> tmp;
    for (int p = 0; p < Nb; ++p) {
        const int i = P.fromOrder[p];
        tmp.clear();
        for (int k = rows[i]; k < rows[i + 1]; ++k) tmp.emplace_back(P.toOrder[cols[k]], k);
        std::sort(tmp.begin(), tmp.end());
        int o = P.rowptr[p];
        for (auto& t : tmp) {
            P.col[o] = t.first;
            P.nnzMap[o] = t.second;
            if (t.first == p) P.diag[p] = o;
            ++o;
        }
        P.rowptr[p + 1] = o;
    }
    // every lower entry must point into an earlier colour, otherwise the colour-by-colour sweeps are wrong
    {
        std::vector<int> colorOf(Nb);
        for (int cc = 0; cc < ncol; ++cc)
            for (int p = P.colorPrefix[cc]; p < P.colorPrefix[cc + 1]; ++p) colorOf[p] = cc;
        for (int p = 0; p < Nb; ++p)
            for (int k = P.rowptr[p]; k < P.rowptr[p + 1]; ++k)
                if (P.col[k] != p && P.col[k] < Nb && colorOf[P.col[k]] == colorOf[p] &&
                    !(chained && CH.chainOf[P.fromOrder[P.col[k]]] == CH.chainOf[P.fromOrder[p]]))
                        return fail(c, OPMHIP_ANALYSIS_FAILED, "ordering is not a valid schedule at row %d", p);
    }
    // L / U split (what Dune's convertToCRS produces, linalg/ParallelOverlappingILU0.hpp:497-584; here both
    // parts keep ascending columns and the sweeps choose their own direction)
    P.lrowptr.assign(Nb + 1, 0);
    P.urowptr.assign(Nb + 1, 0);
    P.lcol.clear();
    P.ucol.clear();
    P.fdest.assign(P.nnzb, -1);
    for (int p = 0; p < Nb; ++p) {
        for (int k = P.rowptr[p]; k < P.rowptr[p + 1]; ++k) {
            if (P.col[k] < p) { P.fdest[k] = (int)P.lcol.size(); P.lcol.push_back(P.col[k]); }
            else if (P.col[k] > p && P.col[k] < Nb) { P.fdest[k] = -2 - (int)P.ucol.size(); P.ucol.push_back(P.col[k]); }  // ghost columns are not part of the ILU
        }
        P.lrowptr[p + 1] = (int)P.lcol.size();
        P.urowptr[p + 1] = (int)P.ucol.size();
    }
    P.nl = (int)P.lcol.size();
    P.nu = (int)P.ucol.size();
    // symbolic elimination: which entries of row p the step against row j updates (k_ilu_factor looks them up instead of merging two
    // column lists per L entry - three dependent rounds of loads became one)
    P.lmatch.assign(P.nnzb, -2);
    static const bool general = [] { const char* e = tuning_env("OPMHIP_FACTOR_GENERAL"); return e && e[0] == '1'; }();   // A/B switch: every step by the general search
    for (int p = 0; p < Nb && !general; ++p) {
        const int kb = P.rowptr[p], ke = P.rowptr[p + 1];
        for (int k = kb; k < ke; ++k) {
            const int j = P.col[k];
            if (j >= p) break;
            int count = 0, uidx = -1, target = -1;
            int ik = k + 1, jk = P.urowptr[j];
            const int jend = P.urowptr[j + 1];
            while (ik < ke && jk < jend) {
                if (P.col[ik] == P.ucol[jk]) { if (count++ == 0) { uidx = jk; target = ik - kb; } ++ik; ++jk; }
                else if (P.col[ik] < P.ucol[jk]) ++ik;
                else ++jk;
            }
            if (count == 0) P.lmatch[k] = -1;
            else if (count == 1 && target < 64 && uidx < (1 << 25)) P.lmatch[k] = uidx * 64 + target;
        }
    }
    // U == upper(A)?  (Pattern::ualias)  Every match of an elimination step must lie on or left of the diagonal of the row being eliminated.
    P.ualias = true;
    for (int p = 0; p < Nb && P.ualias; ++p) {
        const int kb = P.rowptr[p], ke = P.rowptr[p + 1];
        for (int k = kb; k < ke && P.ualias; ++k) {
            const int j = P.col[k];
            if (j >= p) break;
            int ik = k + 1, jk = P.urowptr[j];
            const int jend = P.urowptr[j + 1];
            while (ik < ke && jk < jend) {
                if (P.col[ik] == P.ucol[jk]) { if (P.col[ik] > p) { P.ualias = false; break; } ++ik; ++jk; }
                else if (P.col[ik] < P.ucol[jk]) ++ik;
                else ++jk;
            }
        }
    }
    // the rest of the matrix beside the U part, as a block-CSR of its own (ascending columns: the order of the full row without its U entries)
    P.rrowptr.assign(Nb + 1, 0);
    P.rcol.clear();
    P.rdest.assign(P.nnzb, -1);
    for (int p = 0; p < Nb; ++p) {
        for (int k = P.rowptr[p]; k < P.rowptr[p + 1]; ++k)
            if (P.fdest[k] > -2) { P.rdest[k] = (int)P.rcol.size(); P.rcol.push_back(P.col[k]); }
        P.rrowptr[p + 1] = (int)P.rcol.size();
    }
    P.nr = (int)P.rcol.size();
    P.lightL.assign(ncol, 0);
    P.lightU.assign(ncol, 0);
    if (chained) {
        for (int cc = 0; cc < ncol; ++cc) {
            bool lL = true, lU = true;
            for (int p = P.colorPrefix[cc]; p < P.colorPrefix[cc + 1] && (lL || lU); ++p) {
                const int i = P.fromOrder[p], id = CH.chainOf[i], pos = CH.posIn[i];
                const int nL = P.lrowptr[p + 1] - P.lrowptr[p], nU = P.urowptr[p + 1] - P.urowptr[p];
                if (nL > 1 || (nL == 1 && (pos == 0 || P.lcol[P.lrowptr[p]] != P.toOrder[CH.rows[id][pos - 1]]))) lL = false;
                if (nU > 1 || (nU == 1 && (pos + 1 >= (int)CH.rows[id].size() || P.ucol[P.urowptr[p]] != P.toOrder[CH.rows[id][pos + 1]]))) lU = false;
            }
            P.lightL[cc] = lL;
            P.lightU[cc] = lU;
        }
    }
    if (!chained) {
        build_tiles(P.rowptr, P.colorPrefix, groupRow, P.tiles);
    } else {
        // one sub-tile per (chain-tile, step) group: a group has <= TILE_ROWS rows by construction; over-long groups in
        // blocks are split (still processed in order inside the chain-tile)
        TileSet& T = P.tiles;
        T.row0.clear(); T.colorTile.clear(); T.ctFirst.clear(); T.colorCT.clear();
        int cnext = 0;
        for (size_t g = 0; g + 1 < groupRow.size(); ++g) {
            while (cnext <= ncol - 1 && groupRow[g] == P.colorPrefix[cnext]) {
                T.colorTile.push_back((int)T.row0.size());
                T.colorCT.push_back((int)T.ctFirst.size());
                ++cnext;
            }
            if (groupNewCT[g]) T.ctFirst.push_back((int)T.row0.size());
            int r = groupRow[g];
            const int rend = groupRow[g + 1];
            int pieces = 0;
            while (r < rend) {
                T.row0.push_back(r);
                int e = r + 1;
                while (e < rend && P.rowptr[e + 1] - P.rowptr[r] <= TILE_CAP_BLOCKS) ++e;
                r = e;
                ++pieces;
            }
            // a step cut into several sub-tiles no longer maps chain b to lane b: the lane-private light sweeps
            // (which keep the chain's previous row in registers without looking) are off for that colour
            if (pieces > 1 && cnext >= 1) P.lightL[cnext - 1] = P.lightU[cnext - 1] = 0;
        }
        T.colorTile.push_back((int)T.row0.size());
        T.colorCT.push_back((int)T.ctFirst.size());
        T.ctFirst.push_back((int)T.row0.size());
        T.row0.push_back(Nb);
        for (size_t q = 0; q + 1 < T.ctFirst.size(); ++q)
            if (T.ctFirst[q + 1] - T.ctFirst[q] > 128)
                return fail(c, OPMHIP_ANALYSIS_FAILED, "line colouring: a chain-tile has %d steps (limit 128); lower the chain length", T.ctFirst[q + 1] - T.ctFirst[q]);
    }

    {
        // OPMHIP_XCD_GROUP: chain-tiles per XCD group of the launch schedules (0 = identity map); tuning knob, see DESIGN.md
        const char* e = tuning_env("OPMHIP_XCD_GROUP");
        build_schedules(P, e ? std::atoi(e) : OPMHIP_XCD_GROUP_DEFAULT);
    }
    int rc;
    if ((rc = dev_upload(c, &P.tiles.d_spmvSched, P.tiles.spmvSched))) return rc;
    for (int part = 0; part < 2 && P.sweepStencil; ++part) {
        if ((rc = dev_upload(c, &P.d_swWord[part], P.swWord[part]))) return rc;
        if ((rc = dev_upload(c, &P.d_swKoff[part], P.swKoff[part]))) return rc;
        if ((rc = dev_upload(c, &P.d_swTable[part], P.swTable[part]))) return rc;
    }
    if (P.tiles.stencil) {
        if ((rc = dev_upload(c, &P.tiles.d_stWord, P.tiles.stWord))) return rc;
        if ((rc = dev_upload(c, &P.tiles.d_stKoff, P.tiles.stKoff))) return rc;
        if ((rc = dev_upload(c, &P.tiles.d_stTable, P.tiles.stTable))) return rc;
    }
    if ((rc = dev_upload(c, &P.tiles.d_ctSched, P.tiles.ctSched))) return rc;
    if ((rc = dev_upload(c, &P.tiles.d_ctDesc, P.tiles.ctDesc))) return rc;
    if ((rc = dev_upload(c, &P.d_rowptr, P.rowptr))) return rc;
    if ((rc = dev_upload(c, &P.d_col, P.col))) return rc;
    if ((rc = dev_upload(c, &P.d_diag, P.diag))) return rc;
    if ((rc = dev_upload(c, &P.d_nnzMap, P.nnzMap))) return rc;
    if ((rc = dev_upload(c, &P.d_toOrder, P.toOrder))) return rc;
    if ((rc = dev_upload(c, &P.d_fromOrder, P.fromOrder))) return rc;
    if ((rc = dev_upload(c, &P.d_lrowptr, P.lrowptr))) return rc;
    if ((rc = dev_upload(c, &P.d_lcol, P.lcol))) return rc;
    if ((rc = dev_upload(c, &P.d_urowptr, P.urowptr))) return rc;
    if ((rc = dev_upload(c, &P.d_ucol, P.ucol))) return rc;
    if ((rc = dev_upload(c, &P.d_fdest, P.fdest))) return rc;
    if ((rc = dev_upload(c, &P.d_lmatch, P.lmatch))) return rc;
    if (P.rest.on) {
        if ((rc = dev_upload(c, &P.d_rdest, P.rdest))) return rc;
        if ((rc = dev_upload(c, &P.d_rrowptr, P.rrowptr))) return rc;
        if ((rc = dev_upload(c, &P.rest.d_sched, P.rest.sched))) return rc;
        if ((rc = dev_upload(c, &P.rest.d_word, P.rest.word))) return rc;
        if ((rc = dev_upload(c, &P.rest.d_koff, P.rest.koff))) return rc;
        if ((rc = dev_upload(c, &P.rest.d_table, P.rest.table))) return rc;
    }
    if ((rc = dev_upload(c, &P.tiles.d_row0, P.tiles.row0))) return rc;
    if ((rc = dev_upload(c, &P.tiles.d_ctFirst, P.tiles.ctFirst))) return rc;
    return OPMHIP_SUCCESS;
}

}  // namespace opmhip
