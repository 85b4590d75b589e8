// libopmhip internals: context, device buffers, error plumbing.  Product code — never includes oracle/.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <climits>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/opmhip.h"

namespace opmhip {

constexpr int BS = 3;
constexpr int BB = 9;

// ---- tiling constants (see DESIGN.md "tile kernels") -------------------------------------------------
// One workgroup = one wavefront (64 lanes) = one tile of at most TILE_ROWS block rows, one row per lane.
// A tile's blocks are streamed into LDS with coalesced 16-byte loads, then each lane walks its own row in
// the CPU's sequential order.  TILE_CAP_BLOCKS bounds the LDS image (72 B per block + 4 B column).
#ifndef OPMHIP_TILE_ROWS
#define OPMHIP_TILE_ROWS 32
#endif
constexpr int TILE_ROWS = OPMHIP_TILE_ROWS;
constexpr int TILE_CAP_BLOCKS = 7 * TILE_ROWS;  // a full tile of a Cartesian 7-point stencil

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct TileSet {            // tiles over one block-CSR row-pointer array, never crossing a colour boundary
    std::vector<int> row0;  // [ntiles+1]
    std::vector<int> colorTile;  // [numColors+1] first tile of each colour
    std::vector<int> ctFirst;    // [nChainTiles+1] first sub-tile of each chain-tile (steps of <= TILE_ROWS chains)
    std::vector<int> colorCT;    // [numColors+1] first chain-tile of each colour
    // launch schedules (reorder.cpp: build_schedules): position -> work item, dealt so that position % 8 (= the XCD a
    // workgroup lands on) walks through one spatial stretch of the grid at a time
    std::vector<int> spmvSched;  // [4 * nsched] (r0, r1, rowptr[r0], rowptr[r1]) of every SpMV launch position; r1 <= r0 = padding
    int nsched = 0;
    // decomposed runs: positions [0, nschedInt) hold the INTERIOR tiles (no row of theirs has a ghost column: they can be
    // multiplied while the halo exchange is under way), [nschedInt, nsched) the boundary tiles; else nschedInt == nsched
    int nschedInt = 0;
    std::vector<int> ctSched;    // per colour, padded to a multiple of 8: chain-tile of every launch position or -1
    std::vector<int> ctSchedOff; // [numColors+1] offsets into ctSched
    int* d_row0 = nullptr;
    int* d_ctFirst = nullptr;
    // per schedule position of the chain kernels: [steps, chain-tile, -, -, first row of step 0..steps, first L entry of
    // those rows, first U entry of those rows], descS1 numbers each, record stride descStride (solver.hip: load_desc)
    std::vector<int> ctDesc;
    int descStride = 0, descS1 = 0;
    int* d_spmvSched = nullptr;
    // stencil form of the SpMV's index streams (reorder.cpp: build_schedules): the rows of a tile share their column offsets, so a launch
    // position carries a table of <= 15 offsets (col - row) and a row one word of eight 4-bit table indices (15 = no entry) and one byte
    // (first entry - the tile's first entry) - 5 bytes per row instead of 28 + 16 for column indices and row bounds.  stencil = false: a
    // tile needs more offsets or a row more than eight entries; the explicit index streams are used
    bool stencil = false;           // some part of the schedule has it
    bool stencilPart[2] = {false, false};   // [0] positions [0, nschedInt), [1] the boundary tiles of a decomposed run
    std::vector<unsigned> stWord;
    std::vector<unsigned char> stKoff;
    std::vector<int> stTable;   // [16 * nsched]
    unsigned* d_stWord = nullptr;
    unsigned char* d_stKoff = nullptr;
    int* d_stTable = nullptr;
    int* d_ctSched = nullptr;
    int* d_ctDesc = nullptr;
    int ntiles() const { return (int)row0.size() - 1; }
};

// Launch schedule of the "rest product" (Pattern::ualias): the tiles of y = R x + s u over the block-CSR of the matrix entries that are NOT
// in the factor's U part - lower entries, diagonal, ghost columns.  Same record formats as the SpMV's stencil form (TileSet::spmvSched /
// stWord / stKoff / stTable, read by k_spmv_pipe_st); tiles hold up to 64 rows (one per lane) and TILE_CAP_BLOCKS blocks.
struct RestSched {
    bool on = false;
    int nsched = 0, nschedInt = 0;            // launch positions; [0, nschedInt) interior tiles, the rest boundary tiles (decomposed runs)
    std::vector<int> sched;                   // [4 * nsched] (r0, r1, rrowptr[r0], rrowptr[r1]); r1 <= r0 = padding
    std::vector<unsigned> word;               // per row: eight 4-bit table indices (15 = no entry)
    std::vector<unsigned char> koff;          // per row: first entry - the tile's first entry
    std::vector<int> table;                   // [16 * nsched]
    int* d_sched = nullptr;
    unsigned* d_word = nullptr;
    unsigned char* d_koff = nullptr;
    int* d_table = nullptr;
};

struct Pattern {
    int Nb = 0, nnzb = 0, numColors = 0, nl = 0, nu = 0;  // Nb = owned block rows
    int Nghost = 0, Nloc = 0;                            // ghost cells numbered Nb..Nloc-1 (vectors have Nloc entries)
    int maxRowBlocks = 0;                                // longest row, in blocks
    bool chained = false;  // line colouring: rows of one colour may depend on earlier rows of their own chain
    int kindInForce = 0, chainLen = 0;  // the opmhip_reorder the pattern was ordered with (OPMHIP_REORDER_AUTO resolved) and its rows per chain (0: no chains)
    // per colour: every row's L (U) part is at most the row's own chain predecessor (successor) - the sweep is then a
    // lane-private recurrence and runs in the light kernel (no LDS staging, deep prefetch)
    std::vector<char> lightL, lightU;
    // natural order (as handed over)
    std::vector<int> nat_rowptr, nat_col;
    // ordering
    std::vector<int> toOrder, fromOrder, colorPrefix;  // colorPrefix[numColors+1] in rows
    // internal (reordered) pattern
    std::vector<int> rowptr, col, diag, nnzMap;  // nnzMap[k_internal] = k_natural
    std::vector<int> lrowptr, lcol, urowptr, ucol;
    std::vector<long long> gids;  // optional: global id of every local cell (decomposed runs), natural local order
    TileSet tiles;
    // device copies
    int *d_rowptr = nullptr, *d_col = nullptr, *d_diag = nullptr, *d_nnzMap = nullptr;
    int *d_toOrder = nullptr, *d_fromOrder = nullptr;
    int *d_lrowptr = nullptr, *d_lcol = nullptr, *d_urowptr = nullptr, *d_ucol = nullptr;
    std::vector<int> fdest;      // per matrix entry: where the factorisation puts it - L index (>= 0), -2 - U index, -1 (diagonal / dropped ghost column)
    int* d_fdest = nullptr;
    // stencil form of the sweeps' index streams (solver.hip: SweepStencil), per factor part: word / byte per row, table per tile;
    // sweepStencil = false: some tile needs more than 15 offsets or a row more than CGCH entries - explicit streams
    bool sweepStencil = false;
    std::vector<unsigned> swWord[2];
    std::vector<unsigned char> swKoff[2];
    std::vector<int> swTable[2];
    unsigned* d_swWord[2] = {nullptr, nullptr};
    unsigned char* d_swKoff[2] = {nullptr, nullptr};
    int* d_swTable[2] = {nullptr, nullptr};
    // per L entry (i, j) what its elimination step touches: the ONE entry of row i that meets the U part of row j (a 7-point grid has no
    // triangles: it is the diagonal) as U index * 64 + offset of the target in row i; -1: none; -2: several (the general search)
    std::vector<int> lmatch;
    int* d_lmatch = nullptr;
    // "U is upper(A)": no elimination step of the block ILU0 touches an entry right of the diagonal (linalg/ParallelOverlappingILU0.hpp:466-481
    // modifies A_ik only where (i,j), (j,k) and (i,k) all exist: on a pattern without triangles that is k == i alone), so the factor's U part
    // equals the matrix's own upper part bit for bit and the backward sweep's row sums u_i = sum_{j>i} U_ij x_j ARE the upper part of A x.
    // The product that follows an M^-1 application inside BiCGStab then needs only the REST of the matrix (lower entries, diagonal, ghost
    // columns): y_i = sum_rest A_ik x_k + u_i.  rrowptr / rcol: that rest as a block-CSR of its own, ascending columns; rdest: per matrix entry
    // its place there or -1; its values (d_R in the context) are written by the factorisation from the rows it has staged.
    bool ualias = false;
    int nr = 0;
    std::vector<int> rrowptr, rcol, rdest;
    int *d_rdest = nullptr, *d_rrowptr = nullptr;
    RestSched rest;
};

struct WellsDev {
    int num_wells = 0, nperf = 0;
    int *d_val_pointers = nullptr, *d_Ccols = nullptr, *d_Bcols = nullptr;
    double *d_C = nullptr, *d_D = nullptr, *d_B = nullptr;
    double *d_res = nullptr, *d_xw = nullptr;   // 4 doubles per well each (residual in, well solution out)
    // decomposed runs, wells whose perforations lie in several subdomains (opmhip_wells.distributed): every rank holds the list, B x is
    // summed over the ranks (wells/WellHelpers.hpp:68-123)
    bool distributed = false;
    double* d_bx = nullptr;   // 4 doubles per well: this rank's part of B x, then the sum
    size_t cap_wells = 0, cap_perf = 0;
    // what the device arrays hold, as handed over last time: a Newton iteration passes the same list three times (residual, solve, well
    // solution) and the perforated cells never change - an array that arrives unchanged is not copied again (upload_wells_local)
    std::vector<int> h_vp, h_cc, h_bc;
    std::vector<double> h_D, h_C, h_B;
    // multisegment wells: applied on the host between the product and the standard wells (opmhip_wells.ms_apply)
    int num_ms = 0;
    opmhip_ms_apply_fn ms_apply = nullptr;
    void* ms_user = nullptr;
    double *h_x = nullptr, *h_y = nullptr;   // pinned, Nb * 3 doubles each, natural order
    bool any() const { return num_wells > 0 || num_ms > 0; }
};

// assembly-side device state (all per-cell / per-entry arrays in the INTERNAL order)
struct AsmDev {
    bool fluid_set = false, static_set = false, state_set = false, assembled = false;
    double *d_tab_dbl = nullptr;
    int* d_tab_idx = nullptr;
    int tab_ndbl = 0, tab_nidx = 0;   // lengths of the two table blobs
    double rock_pref = 1e5, rock_cr = 0.0;
    int num_pvt = 0, num_sat = 0, num_rock = 0, rock_desc = 0;
    bool wet_gas = false;   // PVTG: vaporised oil, third primary-variable meaning
    bool ext = false;       // extended intensive-quantity record (wet gas and / or ROCKTAB): 19 fields instead of 17
    double *d_rvmax = nullptr, *d_overburden = nullptr;   // per cell: DRVDT cap, overburden pressure (optional)
    int* d_rocknum = nullptr;                             // per cell rock-table index (optional)
    bool pc_scaling = false;                              // the fluid allows a per-cell end point of pcow (PCW / SWATINIT)
    double* d_pcw = nullptr;                              // per cell: scaled maximum of the oil-water capillary pressure (optional)
    double* d_eps = nullptr;                              // per cell scaled saturation end points, field-major [EPS_COUNT][Nloc] (optional)
    int epscfg = 0;                                       // packed EclEpsConfig (assemble.hip CellStatic::epscfg)
    std::vector<double> sat_eps;                          // host copy: the tables' own end points, EPS_COUNT per saturation region
    // DRSDT / DRVDT bookkeeping (EclProblem::lastRs_ / lastRv_ / maxDRs_ / maxDRv_) and ROCKCOMP IRREVERS (minOilPressure_)
    bool drsdt_on = false, drvdt_on = false;
    double *d_drsdt = nullptr, *d_drvdt = nullptr;        // rates per PVT region [1/s], negative = none
    int* d_drsdt_all = nullptr;                           // per PVT region: the limit binds all cells (OILVAP option)
    double *d_lastRs = nullptr, *d_lastRv = nullptr;      // per cell
    double* d_minpo = nullptr;                            // per cell minimum oil pressure so far (irreversible compaction); NULL = reversible
    double* d_maxso = nullptr;                            // per cell largest oil saturation at the start of a time step (VAPPARS); NULL = not in force
    double vap1 = 0.0, vap2 = 0.0;                        // VAPPARS exponents: on RvSat, on RsSat
    double *d_maxsw = nullptr, *d_sw0 = nullptr;          // water-induced compaction: largest S_w at the start of a time step, initial S_w; NULL = off
    // relative-permeability hysteresis (SATOPTS HYSTER; EHYSTR item 2 = hyst_model 0 | 1, -1 = off): turning points and shifts
    // [4][Nloc], imbibition region per cell, scaled end points of the imbibition curves [EPS_COUNT][Nloc] (optional)
    int hyst_model = -1;
    double* d_hyst = nullptr;
    int* d_imbnum = nullptr;
    double* d_eps_imb = nullptr;
    double* d_rc = nullptr;                               // relativeChange: 2 x 256 partial sums + (delta, denominator)
    int num_wc = 0, h_rocknum_max = -1;                   // water-compaction tables; largest rock-table index handed in (-1: none)
    int* d_wcdesc = nullptr;                              // per table {np, nsw, pressure at, S_w at, pore-volume multipliers at, transmissibility multipliers at | -1}
    double* d_wcdata = nullptr;
    bool storage_frozen = false;                          // begin_time_step formed the old time level's storage: iteration 0 must not refill it
    double* d_invb = nullptr;                             // packed 1/b per cell and phase (Nloc x 3), for the convergence check
    double *d_trans = nullptr, *d_area = nullptr, *d_thpres = nullptr;                      // per entry
    double *d_poro = nullptr, *d_volume = nullptr, *d_depth = nullptr, *d_rsmax = nullptr;  // per cell
    int *d_pvtnum = nullptr, *d_satnum = nullptr;
    double *d_pv = nullptr, *d_iq = nullptr, *d_storageOld = nullptr, *d_source = nullptr, *d_dsource = nullptr;
    unsigned char *d_meaning = nullptr, *d_wasSwitched = nullptr, *d_stage_u8 = nullptr;
    int last_iteration = -1;                // Newton iteration index of the last opmhip_assemble (--cpr-reuse-setup=1)
    double last_dt = 0.0;                   // time step of the last opmhip_assemble (true-IMPES weights scale the storage term by V / dt)
    double* d_drift = nullptr;              // residual * dt of the last accepted time step (drift compensation), Nloc x 3
    bool drift_enabled = true;              // EclEnableDriftCompensation defaults to true (ebos/eclproblem.hh:496-498)
    double max_compensation = 0.1;          // 10 * NewtonTolerance (ebos/eclproblem.hh:352-356, 1854)
    double* d_pv_prev = nullptr;            // solution(1): primary variables at the start of the time step
    unsigned char* d_meaning_prev = nullptr;
    bool prev_set = false;
    int* d_nswitched = nullptr;
    int* d_asm_sched = nullptr;            // per workgroup of k_assemble: first row, end row, first entry, end entry of its tile
    int* d_asm_desc = nullptr;             // per workgroup and lane: column and entry word (row inside the tile, upwind tie-break flag, natural summation order)
    int ntiles = 0, nsched = 0;
    double *d_conv_part = nullptr, *d_conv_out = nullptr;
    double* d_stage_cell = nullptr;   // staging for per-cell doubles (natural order), Nb * max(9, IQS)
    int* d_cell_pos = nullptr;        // opmhip_get_iq_cells / opmhip_set_source_cells: internal positions of the cells named, grown on demand
    size_t cell_pos_cap = 0;
    double* d_stage_entry = nullptr;  // staging for per-entry doubles (natural order), nnzb
};

// domain decomposition: communicator + halo lists (comm.hip)
enum CommKind { COMM_NONE = 0, COMM_LOOPBACK = 1, COMM_RCCL = 2 };
struct CommDev {
    int kind = COMM_NONE, nranks = 1, rank = 0;
    void* nccl = nullptr;    // ncclComm_t
    void* group = nullptr;   // LoopGroup*
    long long global_cells = 0;
    bool halo_set = false;
    int nneigh = 0;
    std::vector<int> neigh, send_ptr, recv_ptr;  // per neighbour: rank, send range (into d_send_idx), ghost range
    int* d_send_idx = nullptr;                   // owned cells (internal positions) to send, grouped by neighbour
    double* d_sendbuf = nullptr;                 // 3 doubles per send cell
    unsigned char* d_sendbuf_u8 = nullptr;
    double* d_red = nullptr;                     // 16 doubles: all-reduce buffer
    // the halo exchange in front of an operator application runs on a stream of its own, beside the product of the interior
    // tiles: ev_x = "the input vector is complete" (main stream), ev_h = "ghost entries are in place" (halo stream)
    hipStream_t hstream = nullptr;
    hipEvent_t ev_x = nullptr, ev_h = nullptr;
    double* halo_vec = nullptr;                  // loopback: the vector whose exchange comm_halo_end still has to drive
    const double* ag_send = nullptr;             // loopback: this rank's contribution to the all-gather under way
    bool reduce_span_open = false;               // a caller of comm_allreduce times the reduction itself (its local sums included)
};

// CPR preconditioner (cpr.hip): pressure-AMG hierarchy, level 0 = the block pattern with scalar values
struct CprLevelDev {
    int n = 0, nnz = 0, nc = 0, W = 0;                              // W: longest row = width of the ELL image
    bool rm = false;                                                // row-major image [i * W + j] (coarse levels, lane groups per row) instead of [j * n + i]
    int *d_ecol = nullptr, *d_rlen = nullptr, *d_diag = nullptr;    // ELL columns [W x n], row lengths, ELL position of the diagonal
    int* d_cpos = nullptr;                                          // ELL position (next level) of every coarse entry
    double *d_val = nullptr, *d_dinv = nullptr, *d_x2 = nullptr;    // ELL values [W x n]
    int *d_agg = nullptr, *d_mptr = nullptr, *d_midx = nullptr;   // node -> aggregate, members of every aggregate
    int* d_mem4 = nullptr;                                         // members of every aggregate as four ints (-1: none), NULL if an aggregate has more: the coarse level then forms its right-hand side itself
    int *d_gptr = nullptr, *d_gidx = nullptr;                      // Galerkin gather lists for the next level's entries
    double *d_b = nullptr, *d_x = nullptr, *d_r = nullptr;         // level vectors
    // level 0 of a single domain on a regular pattern: the ELL columns in stencil form (cpr.hip: EllStencil) - per row a word of 4-bit indices
    // into the table of <= 15 column offsets its aligned group of 32 rows shares; NULL: the explicit column image d_ecol is read
    unsigned* d_sword = nullptr;
    int* d_stable = nullptr;
    // ILU0 smoothing (opmhip_config.cpr_amg_ilu_levels; cpr.hip: CprIluHost): scalar factors in the level's own image (strict lower = L,
    // diagonal = 1 / U_ii, strict upper = U), per row the words that say which slots are lower / upper entries in the level's elimination
    // order, the lower slots in the order the factorisation visits them, and a launch schedule: colour by colour, one thread per SEQUENCE
    // of rows that depend on each other inside the colour (level 0 of a line-coloured pattern: the chains; otherwise single rows)
    bool ilu = false;
    int iluMW = 0, iluWL = 0, iluWU = 0;                           // mask words per kind, lower / upper entries per row at most
    double *d_fval = nullptr, *d_t = nullptr;
    double *d_ilv = nullptr, *d_iuv = nullptr, *d_iud = nullptr;   // what the sweeps read: lower entries [iluWL][n], upper entries [iluWU][n], 1 / U_ii
    int *d_ilc = nullptr, *d_iuc = nullptr;                        // their columns (-1: none)
    bool iluSimple = false;                                        // no elimination step touches anything but a diagonal: k_cpr_ilu_factor_simple
    int* d_tpos = nullptr;                                         // simple levels: per lower entry the place of the transposed entry in the level's image
    std::vector<int> iluWl, iluWu;                                 // per colour: lower / upper entries per row at most
    unsigned* d_imask = nullptr;                                   // [2 * iluMW][n]: lower words, then upper words
    unsigned char* d_lorder = nullptr;                             // [iluWL][n], 255 = none
    int* d_rowAt = nullptr;                                        // per colour [steps][sequences]: row or -1
    std::vector<int> iluNseq, iluNsteps, iluOff;                   // per colour
    std::vector<char> iluFast;                                     // per colour: same-colour couplings are the sequence's neighbours only (register forwarding)
};
struct CprAsyncJob;   // cpr.hip: a structure being built on a host thread (--cpr-reuse-setup=2 with cpr_async_setup)
struct CprDev;
// Decomposed runs (opmhip_config.cpr_gather_rows): the LAST level of a rank's own hierarchy is not solved by the rank - its right-hand
// sides are gathered over the ranks, every rank runs the rest of the cycle on the joined system (`glob`: a hierarchy of its own whose
// level 0 is the joined level, the couplings between the subdomains included) and takes its slice of the result (cpr.hip: cpr_gather_*)
struct CprGatherDev {
    bool on = false;
    int nloc = 0, off = 0, NG = 0, maxn = 0;                       // my rows of the joined level, where they start, its size, the largest slice
    int nnzloc = 0, maxnnz = 0, nnzG = 0;                          // the same for its entries
    double *d_recv = nullptr, *d_vsend = nullptr, *d_vrecv = nullptr;   // gathered right-hand sides [nranks x maxn]; my entries' values, everybody's [nranks x maxnnz]
    int* d_cagg = nullptr;                                         // aggregate (on my last level) of every owned cell
    double* d_send = nullptr;                                      // my slice of the joined level's right-hand side [maxn]
    int *d_unpad = nullptr, *d_vunpad = nullptr, *d_vpos = nullptr;    // joined row -> place in d_recv; joined entry -> place in d_vrecv and in level 0's image of `glob`
    int *d_src = nullptr, *d_xptr = nullptr, *d_xidx = nullptr;    // my entries: place in my last level's image, or (-1 - q) the q-th coupling between subdomains = sum over d_apg[d_xidx[d_xptr[q] ..)]
    int nq = 0;                                                    // fine couplings to ghost cells
    int *d_qrow = nullptr, *d_qentry = nullptr;                    // their row and block-matrix entry
    double* d_apg = nullptr;                                       // their pressure values (k_cpr_ghost_pvals)
    std::shared_ptr<CprDev> glob;
};
struct CprDev {
    bool structured = false, coarse_direct = true;
    bool level0 = false;                                           // level 0's image (it belongs to the pattern) is on the device
    std::shared_ptr<CprAsyncJob> job;
    bool recreate = false;                                         // opmhip_cpr_recreate: the next cpr_update builds the structure anew
    std::vector<CprLevelDev> lv;
    double *d_w = nullptr, *d_lu = nullptr;
    double* d_pcol = nullptr;                                      // level 0: the pressure column of every block, ELL, component-major [3][W x Nb]
    bool w_given = false;                                          // d_w holds weights handed in (opmhip_set_cpr_weights): not recomputed
    double *d_r = nullptr, *d_y = nullptr, *d_z = nullptr;         // fine-level block vectors
    double omega = 2.0 / 3.0, damp = 1.6, beta = 0.25;             // Jacobi damping, prolongation damping, strength threshold
    CprGatherDev gather;
    int apply_rc = 0;                                              // first failure of a collective inside an application (the BiCGStab driver looks at it)
    bool pvals_fresh = false;                                      // the factorisation of this solve left weights and level 0's values behind (FactorRider): cpr_update skips its own pass
};
// What the block ILU0 factorisation does for the CPR on its way through the matrix (k_ilu_factor's rider): every row is in LDS, fixed up,
// before its elimination starts - the pressure-column image of its blocks and its entries of the pressure matrix
// a_p[k] = sum_r A_k[r][p] w_row[r] (PressureTransferPolicy::calculateCoarseEntries, PressureTransferPolicy.hpp:116-139) are written from
// there (k_cpr_pvals' statements on the same values: the same bits) instead of by a second pass over the Jacobian.
struct FactorRider {
    int mode = 0;                 // 0: none; 1: weights read from w (true-IMPES or handed in); 2: quasi-IMPES weights formed from the row's diagonal block (k_cpr_weights' statements) and stored to w
    double* w = nullptr;          // [Nb x 3]
    double* ap = nullptr;         // level 0's ELL values [W x Nb]
    double* pcol = nullptr;       // pressure columns, component-major [3][W x Nb]
    int W = 0;
    int ghostFrom = INT_MAX;      // columns >= this are left out (a subdomain's own pressure system): value 0
};

// per-kernel-class device timing with HIP events on the context's stream (opmhip_profile_*)
enum ProfClass { PROF_SPMV = 0, PROF_ILU_APPLY, PROF_ILU_FACTOR, PROF_VECTOR, PROF_ASSEMBLE, PROF_IQ_UPDATE, PROF_CONVERGENCE, PROF_CPR_AMG,
                 PROF_SPMV_BOUNDARY,   // decomposed runs: the second launch of a product (boundary tiles, after the halo exchange)
                 // communication spans of decomposed runs (prof_span_*): event pairs on the stream the step runs on, outside the chain of
                 // the kernel scopes above and overlapping them - the per-phase timers of the reference's back-ends
                 // (bda/cusparseSolverBackend.cu:303-308, 412-417; bda/openclSolverBackend.cpp:451-459) for what it has no kernels for
                 PROF_HALO,            // one halo exchange: pack -> send / receive -> the ghosts are in (copyOwnerToAll, ParallelOverlappingILU0.hpp:897), on the stream it runs on
                 PROF_ALLREDUCE,       // one global reduction: the local sums' kernel -> the all-reduce -> its result on the device (the sums behind BlackoilModelEbos.hpp:599-603 and the scalar products of BiCGStab)
                 PROF_CPR_GATHER,      // CPR across the ranks: the all-gather of the joined level's right-hand side and the cycle every rank runs on it
                 PROF_COUNT };
struct Profiler {
    bool enabled = false;
    int every = 1;                            // solver scopes are recorded in every `every`-th linear solve (1 = all)
    long solve_no = 0;
    bool suspended = false;                   // inside a linear solve that is not sampled
    std::vector<hipEvent_t> ev;               // event pool, allocated on demand
    size_t ev_used = 0;
    std::vector<int> cls, e0, e1;             // per scope: class (-1: voided - a speculative launch past the stopping
    size_t used = 0;                          //   point), start and end event
    int pending = -1;                         // scope whose end event is still to be recorded (shared with the next begin)
    bool lazy = false;                        // prof_end leaves the end to the next prof_begin (inside a linear solve)
    int handoff = -1;                         // stop event of a kernel launched with its own events: starts the next scope
    static constexpr size_t CAP = 1 << 15;
    double total_ms[PROF_COUNT] = {0};
    long count[PROF_COUNT] = {0};
};

enum Scal {  // device-resident BiCGStab scalars (double d_scal[SC_COUNT])
    SC_RHO = 0, SC_RHOP, SC_ALPHA, SC_OMEGA, SC_BETA, SC_TMP1, SC_TMP2, SC_NORM, SC_NORM0,
    SC_DONE,   // 1.0 once the stopping rule held: every later kernel of the solve returns at once (speculative launches)
    SC_ZERO,   // never written: the "not done" flag of launches outside a solve
    SC_RR,     // opmhip_config.fused_reductions: r.r carried from half iteration to half iteration
    SC_RHOH,   //   rw.r after the first half (rho - alpha v.rw)
    SC_DONEH,  //   number of the half iteration that met the stopping rule (its own update kernels still run), -1: none
    SC_COUNT = 16
};

}  // namespace opmhip

struct opmhip_ctx {
    opmhip_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    bool pattern_set = false, system_loaded = false, factored = false, have_result = false;
    opmhip::Pattern pat;
    // values, internal order
    double *d_A = nullptr, *d_L = nullptr, *d_U = nullptr, *d_invD = nullptr;
    double *d_R = nullptr, *d_usum = nullptr;   // Pattern::ualias: the rest of the matrix beside U (values), the backward sweeps' row sums (3 per row)
    bool half_product = false;                  // BiCGStab forms the product after an ILU0 application from d_R and d_usum (opmhip_config.half_product)
    // vectors, internal order, 3*Nb each
    double *d_b = nullptr, *d_x = nullptr, *d_r = nullptr, *d_rw = nullptr, *d_p = nullptr, *d_v = nullptr,
           *d_s = nullptr, *d_t = nullptr, *d_pw = nullptr, *d_vu = nullptr;
    // staging (natural order): matrix values and one vector
    double *d_stageA = nullptr, *d_stageV = nullptr;
    double* d_scal = nullptr;   // SC_COUNT doubles
    double* d_part = nullptr;   // partial sums: 2 x npart
    double minv_scale = 1.0;    // during a solve: the factor the preconditioned vectors (d_pw, d_s) are still to be multiplied by
    double* d_part2 = nullptr;  // second-level partials: 2 x RED1_BLOCKS
    int npart = 0;
    int last_solve_iterations = 0;   // BiCGStab iterations of the last opmhip_solve_system (--cpr-reuse-setup=2)
    int last_dot_count = 0;     // partial sums the last launch_spmv left in d_part (per list)
    double* h_pinned = nullptr;  // SC_COUNT doubles, pinned
    // read-back ring of the BiCGStab stopping rule: the finalize kernel writes (norm, norm_0, done) of half iteration h
    // straight into pinned host slot h % RB_SLOTS and then, system-scope release, the slot's sequence number, which the
    // host polls (no event, no copy, no stream bubble); the host runs one half iteration ahead of the device
    static constexpr int RB_SLOTS = 4, RB_DOUBLES = 4;   // slot: norm, norm_0, done, sequence number (written last)
    double* h_ring = nullptr;    // pinned, RB_SLOTS x RB_DOUBLES
    double* d_ring = nullptr;    // the same memory through the device's eyes
    double rb_seq = 0.0;         // last sequence number handed to a stopping-rule kernel (exact integers in a double)
    double rb_want[RB_SLOTS] = {0, 0, 0, 0};
    const double* d_done = nullptr;  // flag the tile kernels test first: &d_scal[SC_ZERO] outside a solve, &d_scal[SC_DONE] inside
    opmhip::WellsDev wells;
    opmhip::AsmDev asmb;
    opmhip::CommDev comm;
    opmhip::CprDev cpr;
    opmhip::Profiler prof;
    std::vector<void*> allocs;
    // opmhip_config.pin_host_arrays: host ranges of the caller registered for DMA (first address seen -> hipHostRegister; bytes == 0: refused, not tried again)
    struct HostRange { const void* p; size_t bytes; };
    std::vector<HostRange> pinned;
};

namespace opmhip {

// The library's measurement switches (tools/README.md: two forms that give the same bits, chosen by an environment variable so that a
// change can be judged by alternating runs inside one GPU session) are read only under the master switch OPMHIP_TUNING=1: a stray
// variable in a user's environment changes nothing, and every switch that is in force says so on stderr (once per call site).
inline const char* tuning_env(const char* name) {
    static const bool on = [] { const char* e = std::getenv("OPMHIP_TUNING"); return e && e[0] == '1'; }();
    const char* v = std::getenv(name);
    if (!v) return nullptr;
    if (!on) {
        std::fprintf(stderr, "opmhip: %s is set but OPMHIP_TUNING=1 is not - ignored\n", name);
        return nullptr;
    }
    std::fprintf(stderr, "opmhip: measurement switch %s=%s is in force\n", name, v);
    return v;
}

inline int fail(opmhip_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define OPMHIP_HIP(ctx, call)                                                                       \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return opmhip::fail(ctx, OPMHIP_DEVICE_ERROR, "%s failed: %s (%s:%d)", #call,           \
                                hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

// OPMHIP_POISON_ALLOC=1 (under OPMHIP_TUNING=1; a debugging aid): every floating-point array starts as NaNs, every other array as
// zeros - an entry that is read before it was written then shows in every run instead of in the runs in which hipMalloc hands
// back memory some earlier context left its numbers in
inline bool poison_allocations() {
    static const bool on = [] { const char* e = tuning_env("OPMHIP_POISON_ALLOC"); return e && e[0] == '1'; }();
    return on;
}
template <class T>
int dev_alloc(opmhip_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess)
        return fail(c, OPMHIP_DEVICE_ERROR, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    // (the contexts' streams do not wait for the null stream: the fill is complete before anybody can enqueue a kernel that writes the array)
    if (poison_allocations() && ((e = hipMemset(q, std::is_floating_point<T>::value ? 0xFF : 0x00, bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess)) {
        (void)hipFree(q);
        return fail(c, OPMHIP_DEVICE_ERROR, "hipMemset(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    c->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return OPMHIP_SUCCESS;
}
// gives one tracked allocation back (buffers that are re-allocated larger: the old one must not pile up until destroy)
template <class T>
void dev_free(opmhip_ctx* c, T** p) {
    if (!*p) return;
    for (size_t i = 0; i < c->allocs.size(); ++i)
        if (c->allocs[i] == (void*)*p) { c->allocs[i] = c->allocs.back(); c->allocs.pop_back(); break; }
    (void)hipFree((void*)*p);
    *p = nullptr;
}
template <class T>
int dev_upload(opmhip_ctx* c, T** p, const std::vector<T>& h) {
    int rc = dev_alloc(c, p, h.size());
    if (rc) return rc;
    if (!h.empty()) OPMHIP_HIP(c, hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return OPMHIP_SUCCESS;
}

// RAII-less profiling scope: prof_begin returns a slot (or -1), prof_end closes it.  An event record costs a few
// microseconds of bubble on the stream, so back-to-back scopes share one: prof_end(lazy) leaves the end open and the next
// prof_begin's event closes it (prof_flush closes it where nothing follows).
inline int prof_event(opmhip_ctx* c, hipStream_t s = nullptr) {
    Profiler& P = c->prof;
    if (P.ev.size() <= P.ev_used) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        P.ev.push_back(e);
    }
    const int i = (int)P.ev_used++;
    (void)hipEventRecord(P.ev[i], s ? s : c->stream);
    return i;
}
// an event slot without recording it (for hipExtLaunchKernelGGL, which stamps kernel begin / end itself)
inline int prof_event_slot(opmhip_ctx* c) {
    Profiler& P = c->prof;
    if (P.ev.size() <= P.ev_used) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        P.ev.push_back(e);
    }
    return (int)P.ev_used++;
}
inline void prof_flush(opmhip_ctx* c) {
    Profiler& P = c->prof;
    P.handoff = -1;
    if (P.pending < 0) return;
    const int e = prof_event(c);
    if (e >= 0) P.e1[P.pending] = e; else P.cls[P.pending] = -1;
    P.pending = -1;
}
inline int prof_begin(opmhip_ctx* c, int cls) {
    Profiler& P = c->prof;
    if (!P.enabled || P.suspended || P.used >= Profiler::CAP) { prof_flush(c); return -1; }
    int e;
    if (P.handoff >= 0) { e = P.handoff; P.handoff = -1; }   // the previous kernel's own stop event: no record needed
    else e = prof_event(c);
    if (e < 0) return -1;
    if (P.pending >= 0) { P.e1[P.pending] = e; P.pending = -1; }
    if (P.cls.size() <= P.used) { P.cls.push_back(cls); P.e0.push_back(e); P.e1.push_back(-1); }
    const int slot = (int)P.used++;
    P.cls[slot] = cls; P.e0[slot] = e; P.e1[slot] = -1;
    return slot;
}
// A single kernel timed by its own dispatch (hipExtLaunchKernelGGL start / stop events: no extra packets on the stream,
// kernel-exact like rocprofv3).  Returns false when profiling is off; else es / ee are the event slots to hand to the
// launch.  The start event also closes a pending scope; the stop event will open the next one.
inline bool prof_kernel_scope(opmhip_ctx* c, int cls, int* es, int* ee) {
    Profiler& P = c->prof;
    if (!P.enabled || P.suspended || P.used >= Profiler::CAP) return false;
    *es = prof_event_slot(c);
    *ee = prof_event_slot(c);
    if (*es < 0 || *ee < 0) return false;
    if (P.pending >= 0) { P.e1[P.pending] = *es; P.pending = -1; }
    if (P.cls.size() <= P.used) { P.cls.push_back(cls); P.e0.push_back(*es); P.e1.push_back(*ee); }
    const int slot = (int)P.used++;
    P.cls[slot] = cls; P.e0[slot] = *es; P.e1[slot] = *ee;
    P.handoff = P.lazy ? *ee : -1;
    return true;
}
inline void prof_end(opmhip_ctx* c, int slot) {
    if (slot < 0) return;
    c->prof.pending = slot;
    if (!c->prof.lazy) prof_flush(c);
}
// A span: two events of its own on stream s (default: the context's), independent of the kernel scopes' chain - it may enclose or
// overlap them and may run on the halo stream.  Costs two event records, only in the solves the profiler samples.
inline int prof_span_begin(opmhip_ctx* c, int cls, hipStream_t s = nullptr) {
    Profiler& P = c->prof;
    if (!P.enabled || P.suspended || P.used >= Profiler::CAP) return -1;
    const int e = prof_event(c, s);
    if (e < 0) return -1;
    if (P.cls.size() <= P.used) { P.cls.push_back(cls); P.e0.push_back(e); P.e1.push_back(-1); }
    const int slot = (int)P.used++;
    P.cls[slot] = cls; P.e0[slot] = e; P.e1[slot] = -1;
    return slot;
}
inline void prof_span_end(opmhip_ctx* c, int slot, hipStream_t s = nullptr) {
    if (slot < 0) return;
    const int e = prof_event(c, s);
    if (e >= 0) c->prof.e1[slot] = e; else c->prof.cls[slot] = -1;
}

// reorder.cpp (host): level scheduling / colouring, internal pattern, L/U split, tiles
int build_pattern(opmhip_ctx* c, int Nb, int Nghost, int nnzb, const int* rows, const int* cols);

// kernels.hip launchers (all on c->stream)
void launch_permute_blocks(opmhip_ctx* c, const double* nat, double* internal);
void launch_unpermute_blocks(opmhip_ctx* c, const double* internal, double* nat);  // for lu_out
void launch_vec_to_internal(opmhip_ctx* c, const double* nat, double* internal, int cells = -1);
void launch_vec_to_natural(opmhip_ctx* c, const double* internal, double* nat, int cells = -1);
void launch_zero_diag_fix(opmhip_ctx* c);
// y = A x (+ wells) with the partial sums of ndot scalar products.  exchange: x's ghost entries are brought up to date first
// (copyOwnerToAll) - on the halo stream, beside the product of the interior tiles; x is then written (its ghost part)
int launch_spmv(opmhip_ctx* c, double* x, double* y, int ndot, const double* w0, double xs = 1.0, bool exchange = false, const double* uadd = nullptr, const double* w1 = nullptr);
bool half_product_wanted(const opmhip_ctx* c);   // solver.hip: opmhip_config.half_product resolved for the pattern in hand (asked once, when the system's buffers are allocated)
void launch_wells_residual(opmhip_ctx* c, const double* d_resWell, double* r);
void launch_wells_add_to_matrix(opmhip_ctx* c, int w0, int nw, int serial, const int* d_pair_ptr, const int* d_entry);
int launch_wells_recover(opmhip_ctx* c, const double* d_resWell, const double* x, double* d_xw);   // distributed wells: one all-reduce inside
void launch_ilu_factor(opmhip_ctx* c, bool fix_zero_diagonal = false, const FactorRider* rider = nullptr);
int cpr_factor_rider(opmhip_ctx* c, FactorRider* r);   // cpr.hip: level 0 in place, this solve's weights where they do not come from the matrix; r->mode = 0: no rider this time
void launch_ilu_apply(opmhip_ctx* c, const double* d, double* v, double w_override = -1.0, double* unscaled = nullptr, const double* addp = nullptr, double* work = nullptr, double* usum = nullptr);
// cpr.hip
// solveBoundary: the --cpr-reuse-setup rules are looked at (a structure may be rebuilt, started or swapped in); false (opmhip_cpr_apply:
// a look at the preconditioner between two solves): values only, the structure stays as the last solve left it
int cpr_update(opmhip_ctx* c, bool solveBoundary = true);
void launch_cpr_apply(opmhip_ctx* c, const double* d, double* v);
int cpr_set_weights(opmhip_ctx* c, const double* w);
int cpr_level_sizes(const opmhip_ctx* c, int* n, int* nnz, int cap);
void cpr_shutdown(opmhip_ctx* c);   // joins a structure build in flight (before the context goes)
bool cpr_coarse_pivot_failed(opmhip_ctx* c);
int cpr_ilu_levels_in_force(const opmhip_ctx* c);   // opmhip_config.cpr_amg_ilu_levels with "< 0: the library's choice" resolved (0 without CPR)
inline bool use_cpr(const opmhip_ctx* c) { return c->cfg.preconditioner == OPMHIP_PRECOND_CPR_QUASIIMPES || c->cfg.preconditioner == OPMHIP_PRECOND_CPR_TRUEIMPES; }
int launch_wells_apply(opmhip_ctx* c, const double* x, double* y, double xs = 1.0);   // distributed wells: one all-reduce inside
void launch_lu_to_natural(opmhip_ctx* c, double* d_out_internal_layout);
int bicgstab(opmhip_ctx* c, opmhip_result* res);
// comm.hip
int comm_allreduce(opmhip_ctx* c, double* d_buf, int n, int op /*0 sum, 1 max*/);
int comm_allgather(opmhip_ctx* c, const double* d_send, double* d_recv, size_t count);   // count doubles per rank -> nranks * count on every rank
int comm_halo_f64(opmhip_ctx* c, double* vec, int w, hipStream_t s = nullptr);   // s: the stream it runs on (default: the context's)
int comm_halo_begin(opmhip_ctx* c, double* vec);   // 3 doubles per cell: the exchange on the halo stream, ordered behind the main stream's work so far
int comm_halo_end(opmhip_ctx* c);                  // the main stream waits for it
int comm_halo_u8(opmhip_ctx* c, unsigned char* vec);
void comm_halo_bystander(opmhip_ctx* c);           // a rank without neighbours, where its peers exchange (loopback: their barriers count every rank)
void comm_release(opmhip_ctx* c);
// assemble.hip launchers
void launch_iq_update(opmhip_ctx* c);
int launch_ghost_refresh(opmhip_ctx* c);
void launch_newton_update(opmhip_ctx* c, const double* d_dx_internal, double relax);
void launch_assemble(opmhip_ctx* c, double dt, int iteration);
void launch_drift_update(opmhip_ctx* c, double dt);
void launch_last_rs_rv(opmhip_ctx* c);
void launch_set_limits(opmhip_ctx* c, double dt);
void launch_min_pressure(opmhip_ctx* c, bool init);
void launch_storage_old(opmhip_ctx* c);
void launch_max_oil_saturation(opmhip_ctx* c, bool init);
void launch_max_water_saturation(opmhip_ctx* c, bool init);
void launch_hyst_update(opmhip_ctx* c, const double* d_sw_ow = nullptr, const double* d_sw_go = nullptr);
int launch_relative_change(opmhip_ctx* c);   // -> asmb.d_rc[512 .. 514)
int launch_convergence(opmhip_ctx* c, double dt, double tol_cnv);
void launch_u8_to_internal(opmhip_ctx* c, const unsigned char* nat, unsigned char* internal);
void launch_u8_to_natural(opmhip_ctx* c, const unsigned char* internal, unsigned char* nat);
void launch_iq_to_natural(opmhip_ctx* c, double* d_nat);
void launch_iq_gather(opmhip_ctx* c, int n, const int* d_pos, double* d_out);                                    // records of n cells (internal positions)
void launch_source_scatter(opmhip_ctx* c, int n, const int* d_pos, const double* d_src, const double* d_dsrc);   // d_dsrc nullable
int iq_doubles_per_cell(const opmhip_ctx* c);
int asm_max_rows();
int launch_fluid_probe(opmhip_ctx* c, int pr, int sr, int n, const double* d_in, double* d_out);
int launch_gas_probe(opmhip_ctx* c, int pr, int n, const double* d_in, double* d_out);
int launch_sat_probe(opmhip_ctx* c, int sr, int cfg, const double* d_eps, int n, const double* d_in, double* d_out);
int asm_threads();
void launch_vector_kernels_once(opmhip_ctx* c);
void launch_stream_read(opmhip_ctx* c);

}  // namespace opmhip
