/* opmhip.h — C-ABI of libopmhip.so: the MI355X (gfx950) implementation of OPM Flow's per-Newton-iteration
 * hot path (black-oil AD assembly + ILU0/BiCGStab solve on 3x3 block-CSR).
 *
 * This header is the drop-in boundary.  Every entry point names the reference interface it replaces
 * (paths relative to the reference tree, opm-simulators 2021.10-pre).  Conventions:
 *   - plain C types only; all pointers are caller-owned HOST memory unless the name says "dev";
 *   - every function returns OPMHIP_SUCCESS (0) or a negative opmhip_status; nothing throws across the
 *     boundary (the reference turns a non-success SolverStatus into a Dune fallback,
 *     linalg/ISTLSolverEbos.hpp:277-297, so a C++ shim maps these codes back to bda::SolverStatus);
 *   - one context per GPU, single-threaded per context (the reference calls its backend from the one
 *     Newton-loop thread of a rank, linalg/bda/BdaBridge.cpp:192-255);
 *   - block size is 3, values are IEEE double, indices 32-bit int (bda/BdaSolver.hpp:86-88);
 *   - block-CSR layout exactly as BdaBridge hands it over (bda/BdaBridge.cpp:231-232): rows[Nb+1] block row
 *     pointers, cols[nnzb] block columns ascending inside a row, vals[nnzb*9] row-major 3x3 blocks.
 */
#ifndef OPMHIP_H
#define OPMHIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define OPMHIP_ABI_VERSION 11 /* 11: opmhip_get_iq_cells, opmhip_set_source_cells (the well model's traffic is its perforated cells, not the grid);
                               * 10: opmhip_config gained half_product (the product after an ILU0 application from the sweep's row sums), pin_host_arrays,
                               *     fused_reductions; opmhip_get_product_form, opmhip_preconditioned_product;
                               * 9: opmhip_wells gained `distributed` (standard wells whose perforations lie in several subdomains of a decomposed run);
                               * 8: opmhip_default_config: reorder = OPMHIP_REORDER_AUTO, cpr_amg_ilu_levels = -1 (the measured configuration);
                               *    opmhip_get_ordering_info, opmhip_wells gained the multisegment-well leg (num_ms_wells, ms_apply);
                               * 7: opmhip_config names cpr_amg_ilu_levels, cpr_gather_rows (were reserved[0..1]);
                               * 6: opmhip_set_hysteresis, opmhip_get_hysteresis, opmhip_set_hysteresis_params (additive); opmhip_config names
                               *    cpr_async_setup (was reserved[0]);
                               * 5: opmhip_set_water_compaction, opmhip_get_max_water_saturation, opmhip_relative_change, opmhip_cpr_recreate (additive);
                               * 4: opmhip_config names chain_length, spmv_pipe_wgs, preconditioner (were reserved[0..2]);
                               *    opmhip_set_endpoint_scaling, opmhip_sat_end_points, opmhip_sat_probe, opmhip_synchronize, opmhip_comm_info,
                               *    opmhip_set_composition_change_limits, opmhip_set_irreversible_compaction, opmhip_set_vappars, opmhip_begin_time_step;
                               * 3: opmhip_fluid gained pc_scaling, opmhip_set_pcw */

typedef struct opmhip_ctx opmhip_ctx;

typedef enum opmhip_status {
    OPMHIP_SUCCESS = 0,
    /* the three below mirror bda::SolverStatus (bda/BdaSolver.hpp:32-37) */
    OPMHIP_ANALYSIS_FAILED = -1,
    OPMHIP_CREATE_PRECONDITIONER_FAILED = -2,
    OPMHIP_UNKNOWN_ERROR = -3,
    OPMHIP_INVALID_ARGUMENT = -4, /* bad pointer/size, dim != 3 (bda/BdaBridge.cpp:207-211) */
    OPMHIP_NOT_READY = -5,        /* call order violated (e.g. solve before set_pattern) */
    OPMHIP_DEVICE_ERROR = -6,     /* a HIP call failed; text in opmhip_last_error */
    OPMHIP_NO_DEVICE = -7         /* no gfx950 device visible: the library never falls back to the CPU */
} opmhip_status;

/* --opencl-ilu-reorder (linalg/FlowLinearSolverParameters.hpp:317-321, bda/ILUReorder.hpp) */
typedef enum opmhip_reorder {
    OPMHIP_REORDER_LEVEL_SCHEDULING = 1, /* same factors as the CPU's natural-order ILU0 (bda/Reorder.cpp:266-318) */
    OPMHIP_REORDER_GRAPH_COLORING = 2,   /* Jones-Plassmann rounds, deterministic weights (bda/Reorder.cpp:59-172) */
    OPMHIP_REORDER_GRAPH_COLORING_GREEDY = 3, /* first-fit colouring: red-black on Cartesian 7-point grids */
    OPMHIP_REORDER_LINE_COLORING = 4, /* chains of <= config.chain_length (default 8) rows along each row's farthest
                                        neighbour (the vertical one in CpGrid's natural order), chains coloured greedily:
                                        an exact ILU0 of that ordering, near the natural order's strength at colouring's
                                        parallelism (no counterpart in the reference) */
    OPMHIP_REORDER_AUTO = 5          /* line colouring where it pays - a structured grid in its natural order of at least 30 000 rows,
                                        chains of 4 below 200 000 rows, of 8 below 700 000, of 10 above (chain_length = 0; a non-zero
                                        chain_length is taken as given) - and the greedy colouring elsewhere: the chain sweeps walk their
                                        steps one after the other, which a small or irregular system cannot hide (a 44 777-cell
                                        corner-point grid: 31 Newton its/s with chains of 10, 259 greedy).  ABI 7 */
} opmhip_reorder;

/* --linear-solver-configuration (linalg/setupPropertyTree.cpp:62-76).  The CPR variants: pressure system solved by one AMG
 * V-cycle, ILU0 (relaxation 1) post-smoothing (setupPropertyTree.cpp:94-138); see csrc/cpr.hip for what of it is the
 * reference's and what is not */
typedef enum opmhip_preconditioner {
    OPMHIP_PRECOND_ILU0 = 0,           /* "ilu0" */
    OPMHIP_PRECOND_CPR_QUASIIMPES = 1, /* "cpr_quasiimpes": weights from the diagonal blocks (getQuasiImpesWeights.hpp:46-85) */
    OPMHIP_PRECOND_CPR_TRUEIMPES = 2   /* "cpr" = "cpr_trueimpes" (the reference's default CPR): weights from the storage term
                                          of the state on the device (contexts that assemble) or handed in with
                                          opmhip_set_cpr_weights (getQuasiImpesWeights.hpp:89-128) */
} opmhip_preconditioner;

/* how the ILU relaxation factor w enters M^-1 */
typedef enum opmhip_relax_mode {
    OPMHIP_RELAX_POST_SCALE = 0, /* v = w * U^-1 L^-1 d : the CPU path, linalg/ParallelOverlappingILU0.hpp:848-903 */
    OPMHIP_RELAX_IN_SWEEP = 1    /* x_i = w * D_i^-1 (y_i - sum U_ij x_j) : bda/openclKernels.cpp:301-383 */
} opmhip_relax_mode;

/* Solver configuration.  Defaults (opmhip_default_config) are Flow's: tol 1e-2, maxit 200, w 0.9
 * (linalg/FlowLinearSolverParameters.hpp:142-154) - and, where Flow has no say because the choice is this library's (the ILU ordering of the
 * accelerator path, the pressure AMG's smoother), the configuration bench.py measures.  ctor arguments of bda::BdaSolver (bda/BdaSolver.hpp:79). */
typedef struct opmhip_config {
    int abi_version;       /* OPMHIP_ABI_VERSION */
    int device_id;         /* --bda-device-id */
    int verbosity;         /* linear_solver_verbosity */
    int maxit;             /* --linear-solver-max-iter */
    double tolerance;      /* --linear-solver-reduction */
    double ilu_relaxation; /* --ilu-relaxation */
    int relax_mode;        /* opmhip_relax_mode */
    int reorder;           /* opmhip_reorder (opmhip_default_config: OPMHIP_REORDER_AUTO since ABI 8; opmhip_get_ordering_info says what it became) */
    int zero_diag_fix;     /* 1: exact 0.0 on a diagonal block's diagonal -> 1e-15 (bda/BdaBridge.cpp:125-161) */
    int chain_length;      /* OPMHIP_REORDER_LINE_COLORING: rows per chain (0 = 8) */
    int spmv_pipe_wgs;     /* pipelined SpMV: resident workgroups it is sized for (0 = 2048, the MI355X default; < 0 = one
                            * tile per workgroup instead); tuning only, results are the same bits either way */
    int preconditioner;    /* opmhip_preconditioner: --linear-solver-configuration */
    int cpr_reuse_setup;   /* --cpr-reuse-setup (linalg/FlowLinearSolverParameters.hpp:315, ISTLSolverEbos.hpp:401-426): when the
                            * STRUCTURE of the CPR hierarchy (aggregates, coarse patterns) is built anew from the matrix in hand:
                            * 0 for every linear solve, 1 at the first Newton iteration of every time step (contexts that
                            * assemble; others: never), 2 when the last solve took more than 10 iterations, 3 never after the
                            * first (Flow's default, and opmhip_default_config's).  The VALUES always follow the matrix. */
    int cpr_async_setup;   /* 1: with cpr_reuse_setup = 2 the new structure is built on a host thread BESIDE the solves and swapped in at
                            * the first solve boundary after it is ready (the solves in between keep the old one; one build at a time);
                            * which solve that is depends on the host's speed, so iteration counts are no longer reproducible run to
                            * run.  0 (default): the reference's rule to the letter - the solve that finds the rule met waits for the
                            * rebuild (0.28 s of host work at 10^6 cells).  (was reserved[0] until ABI 6) */
    int cpr_amg_ilu_levels; /* the pressure AMG's smoother: this many of its finest levels smooth with a scalar ILU0, relaxation 1 - the
                            * reference's AMG smoother (linalg/PreconditionerFactory.hpp:126-151, setupPropertyTree.cpp:116-137) - the others
                            * with damped Jacobi.  Level 0 eliminates in the ordering of the block ILU0 (opmhip_reorder), the levels below
                            * colour by colour of a greedy multi-colouring.  0: Jacobi on every level.  < 0 (opmhip_default_config's -1 since ABI 8): the library's choice -
                            * level 0 where the block ILU0's ordering has at most three colours (level 0's sweeps are one launch per colour),
                            * Jacobi otherwise.  (was reserved[0] until ABI 7) */
    int cpr_gather_rows;   /* decomposed runs (opmhip_comm_init_*): the pressure stage of the CPR spans the ranks, as the reference's does (Dune's
                            * parallel AMG behind linalg/OwningTwoLevelPreconditioner.hpp, PressureTransferPolicy.hpp:92-160).  Level 0 smooths
                            * with the pressure operator of the WHOLE system (ghost entries of the iterates exchanged: one double per boundary
                            * cell), its residual is summed over the aggregates down to the rank's first level of at most this many rows,
                            * those levels - their own entries plus the Galerkin sums of the couplings between aggregates of different
                            * subdomains - are joined into one system that every rank holds, coarsens further and cycles on (one all-gather
                            * of a right-hand side per application, one of matrix values per solve), and the post-smoothing residual
                            * d - A (0, x_p, 0) is the whole system's.  Aggregates never cross a rank boundary.  0: the default - over the loopback communicator ON with 100 000 rows (on a 10^6-cell subdomain its third level, aggregates of ~15 cells), over
                            * RCCL OFF until a multi-GPU run has covered its collectives (no round has had more than one GPU; ask for it with a positive value);
                            * < 0: off - one hierarchy per subdomain, no communication inside the preconditioner, iteration counts that
                            * grow with the number of ranks.  Ignored on a single rank; cpr_amg_ilu_levels and cpr_async_setup are ignored
                            * where it is in force.  (was reserved[1] until ABI 7) */
    int half_product;      /* ILU0-BiCGStab: the product that follows an M^-1 application is formed from the backward sweep's row sums.  On a
                            * pattern without triangles (every TPFA grid without well cliques or NNC triangles) no elimination step of the block
                            * ILU0 touches an entry right of the diagonal (linalg/ParallelOverlappingILU0.hpp:466-481 modifies A_ik only where
                            * (i,j), (j,k) and (i,k) all exist), so U == upper(A) bit for bit and u_i = sum_{j>i} U_ij z_j, which the backward
                            * sweep of z = U^-1 y forms anyway (:881-895), IS the upper part of (A z)_i.  The sweep stores u (24 bytes per row)
                            * and the product streams the matrix without its U part: y_i = (sum over lower entries, diagonal and ghost columns,
                            * ascending, of A_ik (w z_k)) + w u_i - 0.8 GB per preconditioned product instead of 1.0 (bda/cusparseSolverBackend.cu:
                            * 103-118 runs the sweep and then bsrmv over the whole matrix on the same vector).  Same products as the reference's
                            * A (w z), summed in another order: results differ from the plain form in the last bits, iteration counts
                            * agree (tests/test_gpu_half_product.py; the oracle states the same order).  0 (default): the library's choice - on
                            * where the pattern allows it, the ordering is line-coloured and the system is large enough for the pipelined
                            * kernels; > 0: wherever the pattern and the ordering allow it; < 0: never.  Ignored with a CPR preconditioner.
                            * Subdomains of a decomposed run: the tiles none of whose rows has a ghost column take the form (68 % of them on a
                            * 10^6-cell subdomain of a 2 x 2 x 2 decomposition), the boundary tiles keep the whole product - behind the halo
                            * exchange, as before; a subdomain without such tiles keeps the plain form.  opmhip_get_product_form says what
                            * is in force.  ABI 10 */
    int pin_host_arrays;   /* 1: the host arrays handed to opmhip_solve_system (vals, b) and opmhip_get_result (x) are registered with the
                            * driver (hipHostRegister) the first time each address is seen, so that every later copy is a DMA at the link's
                            * rate instead of a staged copy out of pageable memory (500 MB of values at 10^6 cells: 13.5 ms -> the PCIe
                            * figure).  For callers whose arrays keep their addresses for the life of the context - Flow's do:
                            * the matrix and the vectors are allocated once (bda/BdaBridge.cpp:199-232, linalg/ISTLSolverEbos.hpp:216-219),
                            * which is what the reference's CUDA back-end relies on when it copies from them every solve
                            * (bda/cusparseSolverBackend.cu:314-338); host/hipSolverBackend.hpp sets it.  The ranges are unregistered by
                            * opmhip_destroy; a refused registration falls back to the plain copy.  0 (default): plain copies - arrays that
                            * come and go (numpy temporaries) must not be pinned behind their owner's back.  ABI 10 */
    int fused_reductions;  /* 1: BiCGStab with ONE reduction per half iteration instead of two.  The reference forms alpha, |r|, omega, |r| and
                            * rho from five scalar products in four reductions per iteration (bda/cusparseSolverBackend.cu:92, 120-127, 151-161;
                            * in parallel runs each is an all-reduce, flow/BlackoilModelEbos.hpp:599-603 for the convergence sums alike).  Here
                            * the product's kernel leaves three sums - v.rw, v.v, v.r in the first half, t.r, t.t, t.rw in the second - and
                            *   |r - alpha v|^2 = r.r - 2 alpha v.r + alpha^2 v.v,   |r - omega t|^2 = r.r - 2 omega t.r + omega^2 t.t,
                            *   rho' = (rho - alpha v.rw) - omega t.rw
                            * give the norms and rho without a second pass: two reductions (of three doubles) and two scalar kernels per
                            * iteration instead of four, update kernels without partial sums.  The stopping rule then looks at RECURRED norms;
                            * the residual vector itself is still updated explicitly, its true norm is formed once at the end of the solve and
                            * reported in opmhip_result.reduction, and a solve whose true norm exceeds twice the tolerance although the recurred
                            * one met it comes back with converged = 0.  Same mathematics, other rounding: iteration counts may differ by a half
                            * step from the default form's; the oracle states the same arithmetic (oracle/linalg.hpp: bicgstab_fused_reductions).
                            * The recurred |r|^2 carries an absolute error of eps |r_0|^2, i.e. the norm a floor of 1.5e-8 |r_0|: opmhip_create refuses
                            * the mode with tolerance < 1e-6 (Flow's default is 1e-2).
                            * 0 (default): the reference's recurrence to the letter.  Meant for runs over many GPUs, where a half iteration's
                            * time is its all-reduces; measured on one GPU in DESIGN.md section 7.  ABI 10 */
} opmhip_config;

/* bda::BdaResult (bda/BdaResult.hpp:28-40) plus the reference's per-phase timers. */
typedef struct opmhip_result {
    int iterations;    /* min(it, maxit), it counted in half steps (bda/cusparseSolverBackend.cu:172) */
    int converged;     /* it != maxit + 0.5 (:176) */
    double reduction;  /* |r| / |r0| */
    double conv_rate;  /* reduction^(1/it) */
    double elapsed;    /* seconds in the whole call */
    double it;         /* raw half-iteration counter */
    double t_copy;     /* H2D of values/rhs (0 when the system is already device resident) */
    double t_factor;   /* ILU0 factorisation   = linear_solve_setup_time (timestepping/SimulatorReport.hpp:29-49) */
    double t_solve;    /* BiCGStab loop        = linear_solve_time */
    int num_colors;    /* colours / levels of the ILU ordering */
    int reserved[3];
} opmhip_result;

/* Standard-well contributions, the data bda::WellContributions carries across the boundary
 * (bda/WellContributions.hpp:60-214; fill order C, D, B per well, wells/StandardWellEval.cpp:1206-1250):
 * applied after each SpMV as y -= C^T (D^-1 (B x)) (bda/WellContributions.cu:36-126). dim = 3, dim_wells = 4.
 * One workgroup per well; where two wells perforate the same cell their updates of that cell are atomic adds (the
 * reference's kernel races there), so the result is then defined up to the order of two additions.
 *
 * Multisegment wells (ABI 8): the reference's WellContributions counts them in getNumWells() (bda/WellContributions.hpp:164-166) but
 * keeps their data in MultisegmentWellContribution objects whose D^-1 is a sparse LU on the HOST (UMFPack,
 * bda/MultisegmentWellContribution.cpp:35-62); its back-ends apply them through a host round trip after every product, in front of
 * the standard wells in the CUDA back-end (bda/WellContributions.cu:160-187), behind them in the OpenCL one (WellContributions.cpp:116-150).  The same here, in the CUDA back-end's order: with num_ms_wells > 0 the
 * library brings x and y (N doubles each, NATURAL order - the caller's objects know nothing of the ILU ordering, so there is no
 * setReordering to do) to pinned host memory after every product, calls ms_apply(ms_user, h_x, h_y) - which performs
 * y -= C^T (D^-1 (B x)) for every multisegment well, e.g. by looping MultisegmentWellContribution::apply(h_x, h_y) - and takes
 * h_y back.  num_wells counts the STANDARD wells only.  The callback runs on the calling thread, inside opmhip_solve_system.
 *
 * Decomposed runs (ABI 9).  Cell indices are the rank's local ones and must be OWNED cells (< Nb).  By default Flow's partitioner keeps
 * every well inside one subdomain (--allow-distributed-wells=false, ebos/eclbasevanguard.hh:148-151): distributed = 0, each rank hands
 * over its own wells, nothing is exchanged.  With distributed wells the reference sums B x over the ranks that share the well
 * (wellhelpers::ParallelStandardWellB::mv, wells/WellHelpers.hpp:68-123, called from StandardWell::apply, wells/StandardWell_impl.hpp:
 * 1254-1280) and has D and the well residual summed at assembly (sumDistributedWellEntries, StandardWell_impl.hpp:260-261), so that D^-1
 * and C^T are applied rank by rank without further exchange.  The same here with distributed = 1: EVERY rank hands over the SAME list
 * of wells in the same order - its own perforations of each (none: val_pointers[w] == val_pointers[w + 1]) and the complete D^-1 - and
 * the num_wells x 4 partial products travel through one all-reduce per operator application (all ranks, not a sub-communicator per
 * well: ranks without perforations add zeros).  Every call that takes such a list is then COLLECTIVE (all ranks, same order); a
 * different num_wells on some rank - including 0: an empty list with distributed = 1 still takes part in the comparison - is reported on
 * every rank (INVALID_ARGUMENT) instead of hanging in the reduction, and so is a rank-local failure to take the list (a null array, a
 * perforation outside the owned cells, an allocation): the local checks run first, their outcome travels in the same reduction and every
 * rank leaves with an error.  A rank that passes NULL or distributed = 0 where the others pass a shared list cannot be caught.
 * opmhip_wells_recover_solution forms resWell - sum_ranks(B x) as the reference's mmv does for a shared well (WellHelpers.hpp:126-142). */
typedef void (*opmhip_ms_apply_fn)(void* user, const double* h_x, double* h_y);
typedef struct opmhip_wells {
    int num_wells;           /* standard wells (num_std_wells of the reference, NOT getNumWells()) */
    const int* val_pointers; /* [num_wells+1] perforation ranges */
    const int* Ccols;        /* [nperf] cell (block row) of each perforation */
    const int* Bcols;        /* [nperf] */
    const double* Cnnzs;     /* [nperf*12] 4x3 row-major */
    const double* Dnnzs;     /* [num_wells*16] 4x4 row-major, already inverted (D^-1) */
    const double* Bnnzs;     /* [nperf*12] */
    int num_ms_wells;        /* multisegment wells behind ms_apply (0: none) */
    opmhip_ms_apply_fn ms_apply; /* must be non-NULL when num_ms_wells > 0 */
    void* ms_user;           /* handed back to ms_apply */
    int distributed;         /* decomposed runs only (ignored on one rank); see above.  0: every well of the list lies inside this rank's
                              * subdomain; 1: the same list of wells on every rank, each rank with the perforations in ITS cells */
} opmhip_wells;

/* ---- lifetime ------------------------------------------------------------------------------------------ */
void opmhip_default_config(opmhip_config* cfg);
/* replaces: BdaBridge ctor selecting a backend by string (bda/BdaBridge.cpp:55-121) */
int opmhip_create(const opmhip_config* cfg, opmhip_ctx** out);
void opmhip_destroy(opmhip_ctx* ctx);
/* last error text of this context (or of the failed create when ctx == NULL); never NULL */
const char* opmhip_last_error(const opmhip_ctx* ctx);
int opmhip_abi_version(void);
/* waits until everything this context has enqueued on its stream is done (callers that time a region from outside) */
int opmhip_synchronize(opmhip_ctx* ctx);

/* ---- linear solve (drop-in for bda::BdaSolver<3>) --------------------------------------------------- */
/* replaces: first-call branch of solve_system -> initialize + analyse_matrix
 * (bda/cusparseSolverBackend.cu:187-260, 348-422; bda/BILU0.cpp:51-158).  The pattern is fixed for the life of
 * the context (linalg/ISTLSolverEbos.hpp:216-219).  Computes the ILU ordering, the device tiling and uploads
 * the index arrays. */
int opmhip_set_pattern(opmhip_ctx* ctx, int Nb, int nnzb, const int* rows, const int* cols);
/* Domain-decomposed variant (one context per GPU, one subdomain per context): the matrix has Nb owned block rows; columns
 * may also reference Nghost ghost cells numbered Nb .. Nb+Nghost-1 - the owner-cells-first local ordering the reference
 * requires of parallel runs (linalg/ISTLSolverEbos.hpp:171-178).  Ghost columns take part in the operator (SpMV) and in
 * the assembly; the ILU0 is block-Jacobi per subdomain and ignores them, like detail::ghost_last_bilu0_decomposition
 * (linalg/ParallelOverlappingILU0.hpp:439-494, loop bounds :857-860).  Per-cell state/static arrays then have
 * Nb+Nghost entries, solver vectors (rhs, x, residual) Nb. */
int opmhip_set_pattern_dd(opmhip_ctx* ctx, int Nb, int Nghost, int nnzb, const int* rows, const int* cols);

/* replaces: bda::BdaSolver<3>::solve_system(N, nnz, dim, vals, rows, cols, b, wellContribs, res)
 * (bda/BdaSolver.hpp:86-88).  N = 3*Nb scalar rows, nnz = 9*nnzb scalars, dim = 3.  If the pattern was not set
 * yet this call sets it from rows/cols (the reference's "initialized == false" path); afterwards rows/cols may
 * be NULL.  vals == NULL and b == NULL mean "use the Jacobian and residual the context assembled on the
 * device".  vals may be modified when zero_diag_fix is on, as the reference does (bda/BdaBridge.hpp:68).
 * wells may be NULL.  A non-converged solve is NOT an error: it returns OPMHIP_SUCCESS with res->converged = 0
 * (the caller then falls back, linalg/ISTLSolverEbos.hpp:277-289). */
int opmhip_solve_system(opmhip_ctx* ctx, int N, int nnz, int dim, double* vals, const int* rows, const int* cols,
                        const double* b, const opmhip_wells* wells, opmhip_result* res);

/* replaces: bda::BdaSolver<3>::get_result(x) (bda/BdaSolver.hpp:90): N doubles to caller memory, natural order */
int opmhip_get_result(opmhip_ctx* ctx, double* x);

/* The two other places where the standard wells' Schur complement touches reservoir vectors, for runs that keep
 * residual and solution on the device.  res_well: num_wells x 4 well residuals (host).
 * replaces: BlackoilWellModel::apply(r) -> StandardWell::apply(BVector& r): r -= C^T (D^-1 resWell)
 *   (wells/BlackoilWellModel_impl.hpp:1031-1042, wells/StandardWell_impl.hpp:1283-1296), applied to the residual the
 *   last opmhip_assemble left on the device (before opmhip_solve_system); */
int opmhip_wells_apply_residual(opmhip_ctx* ctx, const opmhip_wells* wells, const double* res_well);
/* replaces: BlackoilWellModel::addWellContributions(mat) -> StandardWell::addWellContributions
 *   (wells/StandardWell_impl.hpp:1688-1712), the --matrix-add-well-contributions=true mode: A -= C^T D^-1 B written into
 *   the device-resident matrix (uploaded or assembled) instead of being applied after every SpMV; block (Ccols[c],
 *   Bcols[b]) of every perforation pair of a well must be in the pattern given to set_pattern (the "well cliques"
 *   ISTLSolverEbos adds to the sparsity pattern in that mode), else INVALID_ARGUMENT and the matrix is left untouched.
 *   Decomposed runs: only wells that lie inside this rank's subdomain (distributed = 0).  A shared list (distributed = 1) is refused with
 *   INVALID_ARGUMENT on every rank, before anything is exchanged: the blocks that couple perforations of different subdomains are in no
 *   rank's pattern - such wells go through the operator form (opmhip_solve_system's `wells`). */
int opmhip_add_well_contributions(opmhip_ctx* ctx, const opmhip_wells* wells);
/* the right-hand side / residual currently on the device (after opmhip_assemble, opmhip_upload_system or
 * opmhip_wells_apply_residual): N doubles, natural order - what linearizer().residual() holds on the host in Flow */
int opmhip_get_rhs(opmhip_ctx* ctx, double* b);
/* replaces: StandardWell::recoverSolutionWell: xw = D^-1 (resWell - B x) with x the solution of the last
 *   opmhip_solve_system, still on the device (wells/StandardWell_impl.hpp:1298-1311); xw: num_wells x 4 (host, out) */
int opmhip_wells_recover_solution(opmhip_ctx* ctx, const opmhip_wells* wells, const double* res_well, double* xw);

/* ---- pieces of the solve, exposed for parity tests and roofline measurement ------------------------- */
/* upload a system (natural order in, internal order on the device) without solving */
int opmhip_upload_system(opmhip_ctx* ctx, const double* vals, const double* b);
/* y = A x  (Dune::MatrixAdapter::apply, linalg/WellOperators.hpp:127-138); x, y host, natural order */
int opmhip_spmv(opmhip_ctx* ctx, const double* x, double* y);
/* block ILU0 of the uploaded matrix (ParallelOverlappingILU0::update, linalg/ParallelOverlappingILU0.hpp:923-1068);
 * lu_out (nullable, nnzb*9) receives the factors in the layout of Dune's in-place ILU (strict lower = L, strict
 * upper = U, diagonal = D^-1) of the REORDERED matrix, i.e. of the block-CSR pattern that
 * reorderBlockedMatrixByPattern (bda/Reorder.cpp:179-207) produces from opmhip_get_ordering's permutation */
int opmhip_ilu0_factor(opmhip_ctx* ctx, double* lu_out);
/* v = M^-1 d (ParallelOverlappingILU0::apply, :848-903); needs opmhip_ilu0_factor first */
int opmhip_ilu0_apply(opmhip_ctx* ctx, const double* d, double* v);
/* t = A (M^-1 d) formed the way ILU0-BiCGStab forms it inside a solve - the ILU0 application, then the product, in the form in force
 * (opmhip_get_product_form: the whole matrix, or the matrix beside its U part plus the backward sweep's row sums, opmhip_config.half_product)
 * - the pair bda/cusparseSolverBackend.cu:103-118 runs per half iteration.  z (nullable) receives M^-1 d.  Needs opmhip_ilu0_factor first;
 * for parity tests of the half-product form.  ABI 10 */
int opmhip_preconditioned_product(opmhip_ctx* ctx, const double* d, double* t, double* z);
/* replaces: the weights argument of the CPR preconditioner (linalg/ISTLSolverEbos.hpp:440-475: getTrueImpesWeights /
 * getQuasiImpesWeights handed to the preconditioner factory).  weights: 3 doubles per block row, natural order - what
 * Amg::getTrueImpesWeights (linalg/getQuasiImpesWeights.hpp:89-128) returns; they stay in force for every later solve.
 * NULL: back to the weights the library computes itself (quasi-IMPES from the matrix, or - OPMHIP_PRECOND_CPR_TRUEIMPES, contexts that
 * assemble - true-IMPES from the storage term of the present state with the dt of the last opmhip_assemble). */
int opmhip_set_cpr_weights(opmhip_ctx* ctx, const double* weights);
/* replaces: the decision ISTLSolverEbos::shouldCreateSolver takes (linalg/ISTLSolverEbos.hpp:401-426) where the host takes it:
 * the next solve builds the CPR hierarchy's structure anew from the matrix it is given, whatever opmhip_config.cpr_reuse_setup
 * says.  (Contexts that assemble decide modes 0 - 2 themselves; a host that only hands matrices over - the BdaSolver plug-in -
 * knows the Newton iteration number, the library does not.) */
int opmhip_cpr_recreate(opmhip_ctx* ctx);
/* the weights of the last CPR set-up (3 per block row, natural order): diagnosis and tests */
int opmhip_get_cpr_weights(opmhip_ctx* ctx, double* weights);

/* v = M_cpr^-1 d with the CPR preconditioner of the matrix now on the device (contexts created with a CPR preconditioner;
 * needs opmhip_ilu0_factor first: the fine smoother's factors) - for parity tests of the preconditioner alone.  It refreshes the
 * hierarchy's values but does not advance the --cpr-reuse-setup rules.  In decomposed runs with opmhip_config.cpr_gather_rows >= 0
 * it is a COLLECTIVE like opmhip_solve_system: every rank must call it (the joined level's values and right-hand side are
 * all-gathered); a rank that fails before it reaches a collective leaves its peers waiting, as with any MPI-style exchange. */
int opmhip_cpr_apply(opmhip_ctx* ctx, const double* d, double* v);
/* the ordering chosen at set_pattern: toOrder/fromOrder [Nb], rowsPerColor [num colours] (any may be NULL);
 * returns the number of colours/levels or a negative status */
int opmhip_get_ordering(opmhip_ctx* ctx, int* toOrder, int* fromOrder, int* rowsPerColor);
/* what opmhip_config's "the library chooses" settings resolved to at set_pattern: info[0] the opmhip_reorder in force (never
 * OPMHIP_REORDER_AUTO), info[1] rows per chain of the line colouring (0: no chains), info[2] colours / levels, info[3] the
 * cpr_amg_ilu_levels in force (0 without a CPR preconditioner).  The reference prints the same kind of line at set-up
 * (bda/openclSolverBackend.cpp:229-246, BILU0.cpp:106-108).  ABI 8 */
int opmhip_get_ordering_info(opmhip_ctx* ctx, int info[4]);
/* what opmhip_config.half_product resolved to at set_pattern: info[0] 1 if ILU0-BiCGStab forms the product after M^-1 from the backward
 * sweep's row sums, else 0; info[1] 1 if the pattern has the property it rests on (no elimination step touches an entry right of the
 * diagonal: U == upper(A)); info[2] the blocks that product streams: the matrix beside its U part (a subdomain with ghost columns: its interior tiles' rows beside
 * their U part plus its boundary tiles' whole rows); info[3] launch positions of that
 * product.  ABI 10 */
int opmhip_get_product_form(opmhip_ctx* ctx, int info[4]);
/* Device-timed repetitions of one kernel on the uploaded system, for bench.py's roofline object:
 * which = 0 SpMV, 1 ILU0 apply, 2 ILU0 factor, 3 one BiCGStab iteration's vector kernels, 4 a plain streaming read of
 * the Jacobian's value array (72 * nnzb bytes; the on-box HBM ceiling the roofline fractions are put beside), 5 / 6 the SpMV
 * with the partial sums of one / two scalar products (the forms BiCGStab launches); contexts with the half-product form in force
 * (opmhip_get_product_form): 7 the product over the matrix beside its U part, 8 the same with two scalar products, 9 the ILU0 application
 * with the backward sweeps' row sums stored.
 * Launches the kernel `reps` times back to back on the context's stream between two HIP events and returns the
 * average milliseconds per launch in *ms_per_launch. */
int opmhip_time_kernel(opmhip_ctx* ctx, int which, int reps, double* ms_per_launch);
/* CPR only: unknowns and entries of the pressure-AMG levels after the first solve (n, nnz: cap entries each); returns the
 * number of levels */
int opmhip_cpr_levels(opmhip_ctx* ctx, int* n, int* nnz, int cap);

/* ---- assembly ("linearization") half of the Newton iteration ------------------------------------------ */
/* Deck-level fluid and saturation-function tables, SI units, flat arrays: what PVTW, DENSITY, PVDG, PVTO, SWOF,
 * SGOF and ROCK provide (python/test_data/SPE1CASE1/SPE1CASE1.DATA:109-250).  Region r of the PVT tables is
 * PVTNUM r+1, of the saturation tables SATNUM r+1.  Live oil + dry gas + water (no PVTG). */
typedef struct opmhip_fluid {
    int num_pvt, num_sat;
    const double* pvtw;         /* [num_pvt*5]  p_ref, Bw_ref, c_w, mu_ref, c_v */
    const double* density;      /* [num_pvt*3]  oil, water, gas at surface conditions */
    const int* pvdg_ptr;        /* [num_pvt+1]  row ranges into pvdg */
    const double* pvdg;         /* rows (p, Bg, mu_g) */
    const int* pvto_node_ptr;   /* [num_pvt+1]  Rs-node ranges */
    const double* pvto_rs;      /* [nodes]      Rs of each node */
    const int* pvto_row_ptr;    /* [nodes+1]    row ranges into pvto; the first row of a node is the saturated point */
    const double* pvto;         /* rows (p, Bo, mu_o) */
    const int* swof_ptr;        /* [num_sat+1] */
    const double* swof;         /* rows (Sw, krw, krow, pcow) */
    const int* sgof_ptr;        /* [num_sat+1] */
    const double* sgof;         /* rows (Sg, krg, krog, pcog) */
    double rock_pref, rock_cr;  /* ROCK: reference pressure, compressibility (ebos/eclproblem.hh:1454-1486) */
    /* Wet gas (PVTG; NULL / no nodes = dry gas from PVDG, and then pvdg is mandatory): per PVT region its gas-pressure
     * nodes, per node the rows (Rv, Bg, mu_g) as the deck lists them - the saturated row first, Rv descending.  With PVTG
     * the oil component may vaporise into the gas phase: Rv, the third primary-variable meaning OPMHIP_SW_PG_RV, DRVDT cap
     * through opmhip_set_problem_extras (FluidSystem::enableVaporizedOil). */
    const int* pvtg_node_ptr;   /* [num_pvt+1]  pressure-node ranges */
    const double* pvtg_pg;      /* [nodes]      gas pressure of each node */
    const int* pvtg_row_ptr;    /* [nodes+1]    row ranges into pvtg */
    const double* pvtg;         /* rows (Rv, Bg, mu_g) */
    /* ROCKTAB: per rock region rows (p, pore-volume multiplier, transmissibility multiplier), evaluated with linear
     * extrapolation at the (effective) oil pressure: rockCompPoroMultiplier / rockCompTransMultiplier
     * (ebos/eclproblem.hh:1936-2007).  num_rock = 0: none. */
    int num_rock;
    const int* rocktab_ptr;     /* [num_rock+1] */
    const double* rocktab;
    /* Per-cell end points of the saturation functions (the deck has ENDSCALE, PCW or SWATINIT).  Non-zero: the context
     * keeps the extended intensive-quantity record and accepts opmhip_set_pcw / opmhip_set_endpoint_scaling. */
    int pc_scaling;
} opmhip_fluid;

/* primary-variable meaning per cell: BlackOilPrimaryVariables::PrimaryVarsMeaning */
#define OPMHIP_SW_PO_SG 0
#define OPMHIP_SW_PO_RS 1
#define OPMHIP_SW_PG_RV 2 /* wet gas only: oil phase absent, pressure variable = gas pressure, third variable = Rv */

/* replaces: FluidSystem / MaterialLawManager initialisation from the deck (setup, once) */
int opmhip_set_fluid(opmhip_ctx* ctx, const opmhip_fluid* fluid);

/* replaces: the per-cell / per-connection arrays EclProblem serves to the assembly
 * (ebos/eclproblem.hh:1332-1340 transmissibility, :1409 thresholdPressure, :1430-1447 porosity & depth,
 * dofTotalVolume, pvtRegionIndex/satnumRegionIndex, :1711-1732 maxGasDissolutionFactor).
 * trans, area, thpres: one value per block-CSR entry in the NATURAL pattern order (0 on diagonal entries); must be
 * symmetric ((I,J) == (J,I)); thpres may be NULL (= 0).  poro, volume, depth: per cell.  pvtnum/satnum: per cell,
 * 0-based, NULL = region 0.  rsmax: per cell cap on Rs (DRSDT), NULL = unlimited.  Needs set_pattern first. */
int opmhip_set_static(opmhip_ctx* ctx, const double* trans, const double* area, const double* thpres,
                      const double* poro, const double* volume, const double* depth, const int* pvtnum,
                      const int* satnum, const double* rsmax);

/* replaces: EclProblem::maxOilVaporizationFactor (DRVDT cap on Rv, ebos/eclproblem.hh:1734-1754), rockTableIdx_ (ROCKNUM,
 * :1943-1945) and overburdenPressure_ (:1954-1955).  Per cell, natural order, any may be NULL (no cap / table 0 / none);
 * only meaningful for a fluid with PVTG or ROCKTAB tables (else INVALID_ARGUMENT).  Needs set_static; recomputes the cached
 * intensive quantities if a state is set. */
int opmhip_set_problem_extras(opmhip_ctx* ctx, const double* rvmax, const int* rocknum, const double* overburden);

/* replaces: the DRSDT / DRVDT bookkeeping of EclProblem - maxDRs_ / maxDRv_ = rate x time-step size (ebos/eclgenericproblem.cc,
 * beginTimeStep_), lastRs_ / lastRv_ (updateCompositionChangeLimits_, ebos/eclproblem.hh:2010-2107, called when the initial
 * solution is applied, :1811, and from endTimeStep, :1125) and maxGasDissolutionFactor / maxOilVaporizationFactor
 * (:1711-1754).  drsdt / drvdt: per PVT region, in 1/s (Sm3/Sm3 per second); negative = no limit in that region; NULL = the
 * keyword is not in force.  drsdt_all_cells (per region, nullable = 0): the OILVAP option - non-zero: the limit binds every
 * cell, zero: only cells with free gas (Sg > 1e-7).  The caps then live on the device: lastRs / lastRv are taken from the
 * state now present (call it after opmhip_set_state, like initialSolutionApplied) and after every opmhip_end_time_step;
 * opmhip_begin_time_step(dt) turns them into this step's caps, replacing the rsmax / rvmax arrays of opmhip_set_static /
 * opmhip_set_problem_extras.  DRVDT needs a fluid with PVTG.  (The convective DRSDTCON variant is not built.) */
int opmhip_set_composition_change_limits(opmhip_ctx* ctx, const double* drsdt, const int* drsdt_all_cells, const double* drvdt);

/* replaces: ROCKCOMP's hysteresis mode IRREVERS - minOilPressure_ (ebos/eclgenericproblem.cc:155-170; initialised from the
 * initial state, ebos/eclproblem.hh:2293-2294; updateMinPressure_, :2172-2197; used by rockCompPoroMultiplier /
 * rockCompTransMultiplier, :1948-1952, 1988-1992): the rock tables are read at min(p_o, lowest p_o the cell has seen at the
 * start of a time step).  enable != 0: start tracking from the state now present; 0: reversible compaction again.  Needs a
 * fluid with ROCKTAB. */
int opmhip_set_irreversible_compaction(opmhip_ctx* ctx, int enable);

/* replaces: VAPPARS - EclProblem::maxOilSaturation (ebos/eclproblem.hh:1682-1688; initialised from the initial state,
 * :2291-2292; updateMaxOilSaturation_, :2110-2141) and what the PVT classes make of it: saturated Rs x max(1e-3, (S_o /
 * S_o,max)^vap2), saturated Rv x max(1e-3, (S_o / S_o,max)^vap1) where S_o is below the largest oil saturation the cell has
 * seen at the start of a time step (LiveOilPvt / WetGasPvt of opm-material, absent from the reference tree: restated).
 * enable != 0: start tracking from the state now present.  The power function makes this the one feature whose device
 * results agree with the CPU restatement to rounding (1e-12 of a quantity's magnitude, measured 1.1e-13), not to the bit.  Needs a context with the extended
 * record (a fluid with PVTG, ROCKTAB or pc_scaling). */
int opmhip_set_vappars(opmhip_ctx* ctx, int enable, double vap1, double vap2);

/* replaces: water-induced rock compaction (ROCKCOMP with ROCK2D / ROCK2DTR / ROCKWNOD) - the tables rockCompPoroMultWc_ /
 * rockCompTransMultWc_ (built in ebos/eclgenericproblem.cc:186-240), their use in rockCompPoroMultiplier /
 * rockCompTransMultiplier (ebos/eclproblem.hh:1962-1967, 2001-2005: multiplier(effective oil pressure, SwMax - Sw_initial),
 * SwMax = max(Sw, the largest Sw the cell has seen at the start of a time step), extrapolating) and the tracker
 * maxWaterSaturation_ (initialised from the initial state, :2289-2290; updateMaxWaterSaturation_, :2144-2169, run by
 * opmhip_begin_time_step - including its statement :2150, which hands cell 1 the stored maximum of cell 0 before the loop).
 * num_tables rock regions (0: feature off), selected per cell by the rocknum of opmhip_set_problem_extras; table t has
 * num_pressure[t] >= 2 pressure nodes [Pa] (ROCK2D records) and num_sw[t] >= 2 saturation nodes (ROCKWNOD), both ascending
 * and concatenated over the tables; pv_mult / trans_mult: num_pressure[t] x num_sw[t] values per table, row-major by pressure
 * node, concatenated; trans_mult NULL = no ROCK2DTR (multiplier 1).  The initial water saturation and the tracker start from
 * the state now present: call it after opmhip_set_state.  Not together with ROCKTAB tables in the fluid (the reference reads
 * the one or the other); needs a context with the extended record (a fluid with PVTG or pc_scaling).  The table class
 * (UniformXTabulated2DFunction of opm-material) is absent from the reference tree: restated, see oracle/blackoil.hpp. */
int opmhip_set_water_compaction(opmhip_ctx* ctx, int num_tables, const int* num_pressure, const int* num_sw, const double* pressure,
                                const double* sw, const double* pv_mult, const double* trans_mult);
/* the tracker of the above per cell (natural order, Nb + Nghost entries; zeros when the feature is off) */
int opmhip_get_max_water_saturation(opmhip_ctx* ctx, double* max_water_saturation);

/* replaces: the per-cell work of EclProblem::beginTimeStep (ebos/eclproblem.hh:1042-1075) for a time step of size dt [s]:
 * updateMaxWaterSaturation_, updateMinPressure_, updateMaxOilSaturation_, the DRSDT / DRVDT caps of this step, invalidateAndUpdateIntensiveQuantities(0); and, since
 * recycleFirstIterationStorage() is false with DRSDT / DRVDT (:1758-1765), the old time level's storage term formed with ITS
 * caps (time index 1: lastRs / lastRv without the increment) - opmhip_assemble(iteration 0) then leaves it alone.  Call it
 * after opmhip_advance_time_level, and again before every retry of a chopped step.  Also updateHysteresis_ (:1060) when
 * opmhip_set_hysteresis is in force.  A no-op (SUCCESS) when none of these features is in force. */
int opmhip_begin_time_step(opmhip_ctx* ctx, double dt);

/* the trackers, for restart files and tests: lastRs, lastRv, minimum oil pressure, maximum oil saturation per cell (natural
 * order, Nb + Nghost entries each; any may be NULL; an array that is not kept comes back as zeros) */
int opmhip_get_trackers(opmhip_ctx* ctx, double* last_rs, double* last_rv, double* min_oil_pressure, double* max_oil_saturation);

/* replaces: the per-cell scaled end point maxPcow of the oil-water capillary pressure - the PCW array of the deck, or
 * what SWATINIT made of it during equilibration (ebos/equil/initstateequil.hh:1330-1343 -> EclMaterialLawManager::
 * applySwatinit) - as EclEpsTwoPhaseLaw applies it with enablePcScaling: pcow(Sw) = table(Sw) * (pcw / table(Swl)),
 * the factor taken as 1 where pcw equals the table's own maximum.  (opm-material is absent from the reference tree:
 * restated from its published form, see oracle/blackoil.hpp.)  pcw: per cell, natural order, Pa; NULL = unscaled.
 * Needs a fluid with pc_scaling set, and set_static first. */
int opmhip_set_pcw(opmhip_ctx* ctx, const double* pcw);

/* Saturation end-point scaling (ENDSCALE, SCALECRS, SWL ... SOGCR, KRW / KRO / KRG, KRWR / KRORW / KRORG / KRGR, PCW / PCG).
 * replaces: the per-cell scaled end points the EclMaterialLawManager holds and hands to the material laws
 * (materialLawParams(elemIdx), ebos/eclproblem.hh:1490-1498; EclEpsScalingPointsInfo / EclEpsConfig / EclEpsTwoPhaseLaw of
 * opm-material, which is absent from the reference tree: restated from its published form, see oracle/fluid.hpp).
 * Indices of the end points of a cell / of a saturation region's tables: */
enum {
    OPMHIP_EPS_SWL = 0,  /* connate water */        OPMHIP_EPS_SWCR = 1,    /* critical water */
    OPMHIP_EPS_SWU = 2,  /* maximum water */        OPMHIP_EPS_SOWCR = 3,   /* critical oil in water */
    OPMHIP_EPS_SGL = 4,  /* connate gas */          OPMHIP_EPS_SGCR = 5,    /* critical gas */
    OPMHIP_EPS_SGU = 6,  /* maximum gas */          OPMHIP_EPS_SOGCR = 7,   /* critical oil in gas */
    OPMHIP_EPS_MAXPCOW = 8,  /* PCW */              OPMHIP_EPS_MAXPCGO = 9, /* PCG */
    OPMHIP_EPS_MAXKRW = 10,  /* KRW */              OPMHIP_EPS_MAXKROW = 11, /* KRO (oil-water) */
    OPMHIP_EPS_MAXKRG = 12,  /* KRG */              OPMHIP_EPS_MAXKROG = 13, /* KRO (gas-oil) */
    OPMHIP_EPS_KRWR = 14, OPMHIP_EPS_KRORW = 15, OPMHIP_EPS_KRGR = 16, OPMHIP_EPS_KRORG = 17,   /* values at the displacing phase's critical saturation */
    OPMHIP_EPS_COUNT = 18
};
typedef struct opmhip_endpoint_scaling {
    int sat_scaling;     /* ENDSCALE: two-point scaling of the saturation axis of every curve */
    int three_point_kr;  /* SCALECRS YES: the relative permeabilities keep a third point (the other phase's critical saturation) */
    int krw, kro, krg;   /* vertical scaling of krw / kro (krow and krog) / krg: 0 none, 1 at the maximum (KRW / KRO / KRG), 2 also
                          * at the displacing phase's critical saturation (KRWR / KRORW, KRORG / KRGR) */
    int pcw, pcg;        /* PCW / PCG: capillary pressures scaled to the cell's maximum */
    const double* points[OPMHIP_EPS_COUNT]; /* per cell (natural order, Nb + Nghost entries), SI; NULL = the end point of the
                                             * cell's SATNUM tables (opmhip_sat_end_points) */
} opmhip_endpoint_scaling;
/* Needs a fluid with pc_scaling set, and set_static.  eps == NULL: scaling off.  Recomputes the cached intensive quantities
 * if a state is set.  Not to be combined with opmhip_set_pcw (PCW then is points[OPMHIP_EPS_MAXPCOW] with pcw = 1). */
int opmhip_set_endpoint_scaling(opmhip_ctx* ctx, const opmhip_endpoint_scaling* eps);
/* the end points of the tables of saturation region sat_region (opm-common's satfunc end-point extraction, restated):
 * out[OPMHIP_EPS_COUNT].  Needs opmhip_set_fluid only. */
int opmhip_sat_end_points(opmhip_ctx* ctx, int sat_region, double* out);

/* replaces: relative-permeability hysteresis - SATOPTS HYSTER / EHYSTR / IMBNUM: the per-cell material-law parameters the
 * EclMaterialLawManager serves (ebos/eclproblem.hh:1490-1498 materialLawParams) with EclHysteresisTwoPhaseLaw as the two-phase
 * law, and their update at the start of a time step (updateHysteresis_, :1060, 2603-2626, done by opmhip_begin_time_step).
 * kr_model = EHYSTR item 2: 0 = Carlson's model for the non-wetting phases (oil in the oil-water system, gas in the gas-oil
 * system), wetting phases on their drainage curves; 1 = the same with the wetting phases on their IMBIBITION curves; other
 * values are refused as the reference refuses them (opm/simulators/utils/PartiallySupportedFlowKeywords.cpp:299-302: "only
 * Carlson Hysteresis Models supported (0 or 1)"; capillary pressures stay on the drainage curves, :502-507); negative = not in
 * force.  imbnum: per cell (natural order, Nb + Nghost) the saturation region (0-based, of opmhip_fluid's SWOF / SGOF tables)
 * that holds the imbibition curves; the drainage curves are those of set_static's satnum.  imb (nullable; only with
 * opmhip_set_endpoint_scaling in force): its points[] are the scaled end points of the IMBIBITION curves (ISWL, ISWCR, ...;
 * NULL entries = the imbibition tables' own), its flags are ignored (those of the drainage scaling apply).  The state per cell
 * and two-phase system - the smallest wetting saturation seen at the start of a time step (krnSwMdc; restart vectors KRNSW_OW /
 * KRNSW_GO, ebos/eclwriter.hh:285-288) and the imbibition curve's shift (deltaSwImbKrn) - starts at "nothing seen" (2, 0);
 * the first opmhip_begin_time_step sets it from the state then present, as the reference's first beginTimeStep does.  Needs a
 * context with the extended record (a fluid with PVTG, ROCKTAB or pc_scaling) and set_static.  The law itself lives in
 * opm-material, which is not in the reference tree: restated from its published form, UNVERIFIED (oracle/fluid.hpp). */
int opmhip_set_hysteresis(opmhip_ctx* ctx, int kr_model, const int* imbnum, const opmhip_endpoint_scaling* imb);
/* the hysteresis state per cell (natural order, Nb + Nghost; any NULL): turning point and shift of the oil-water system, of
 * the gas-oil system (what eclwriter writes as KRNSW_OW / KRNSW_GO - PCSWM_* are equal to them: the reference updates both with
 * the same saturation) */
int opmhip_get_hysteresis(opmhip_ctx* ctx, double* krn_sw_mdc_ow, double* delta_sw_imb_krn_ow, double* krn_sw_mdc_go, double* delta_sw_imb_krn_go);
/* replaces: initHysteresisParams of a restarted run (ebos/ecloutputblackoilmodule.hh:569-589 -> setOilWaterHysteresisParams /
 * setGasOilHysteresisParams): the turning points handed in, the shifts recomputed from them */
int opmhip_set_hysteresis_params(opmhip_ctx* ctx, const double* krn_sw_mdc_ow, const double* krn_sw_mdc_go);

/* Point evaluation of the fluid-system and saturation functions the assembly uses, ON THE DEVICE, for host-side setup
 * code (equilibration, ebos/equil/initstateequil.hh) and for tests that pin these functions against the reference's
 * EQUIL expectations.  For each of n points: out[8 i + 0..7] = 1/B_w(p), 1/B_g(p), 1/B_o(p, rs) (the saturated curve
 * where rs >= RsSat(p), as the equilibration's oil density does, initstateequil.hh:214-233), RsSat(p), pcow(sw),
 * pcgo(sg), mu_o(p, rs) [Pa s], mu_g(p).  Needs opmhip_set_fluid only.  All arrays host memory. */
int opmhip_fluid_probe(opmhip_ctx* ctx, int pvt_region, int sat_region, int n, const double* p, const double* rs,
                       const double* sw, const double* sg, double* out);

/* The saturation functions at (sw, sg) the same way, optionally with ONE set of scaled end points (eps: flags as in
 * opmhip_set_endpoint_scaling, points[f] pointing at ONE double each or NULL = the table's own; eps == NULL: unscaled):
 * out[5 i + 0..4] = krw, kro, krg, pcow, pcgo.  What an equilibration with ENDSCALE inverts per cell (equil.py). */
int opmhip_sat_probe(opmhip_ctx* ctx, int sat_region, const opmhip_endpoint_scaling* eps, int n, const double* sw, const double* sg,
                     double* out);

/* The gas-phase functions at (p_g, Rv) the same way: out[3 i + 0..2] = 1/B_g, mu_g (on the saturated curve where
 * Rv >= RvSat(p_g), as the equilibration's gas density does, initstateequil.hh:240-285), RvSat(p_g).  Dry-gas fluids:
 * 1/B_g(p), mu_g(p), 0. */
int opmhip_gas_probe(opmhip_ctx* ctx, int pvt_region, int n, const double* p, const double* rv, double* out);

/* replaces: model().solution(0) = ... ; model().invalidateAndUpdateIntensiveQuantities(0)
 * (flow/BlackoilModelEbos.hpp:552-562).  pv: Nb x 3 (Sw, p_o, Sg|Rs), meaning: Nb bytes. Natural order. */
int opmhip_set_state(opmhip_ctx* ctx, const double* pv, const unsigned char* meaning);
int opmhip_get_state(opmhip_ctx* ctx, double* pv, unsigned char* meaning);

/* replaces: BlackoilModelEbos::relativeChange (flow/BlackoilModelEbos.hpp:431-510), the error measure of the PID time-step
 * control (timestepping/TimeStepControl.cpp:127-161; Flow's default --time-step-control=pid+newtoniteration): the sum over
 * the owned cells of (p_new - p_old)^2 + sum_phases (S_new - S_old)^2 over the sum of p_new^2 + sum_phases S_new^2, new = the
 * state on the device, old = the time level opmhip_advance_time_level kept; summed over the ranks of a decomposed run.  Call
 * it after an accepted time step.  The cells are summed by a fixed tree: equal to the reference's sequential sum to rounding. */
int opmhip_relative_change(opmhip_ctx* ctx, double* relative_change);

/* The two time levels of the discretisation (opm-models FvBaseDiscretization: advanceTimeLevel() copies solution(0)
 * into solution(1) when a time step starts; updateFailed() copies it back and recomputes the intensive quantities
 * when the Newton method gave up - the path AdaptiveTimeSteppingEbos takes before it retries with a shorter step,
 * opm/simulators/timestepping/AdaptiveTimeSteppingEbos.hpp:355-441).  Device-to-device, ghost cells included,
 * asynchronous on the context's stream.  update_failed needs a preceding advance_time_level (else NOT_READY). */
int opmhip_advance_time_level(opmhip_ctx* ctx);
int opmhip_update_failed(opmhip_ctx* ctx);

/* replaces: the drift part of EclProblem::endTimeStep (ebos/eclproblem.hh:1126-1135) and of EclProblem::source
 * (:1847-1875).  Flow compensates systematic mass drift by default (EclEnableDriftCompensation = true, :496-498): after
 * every ACCEPTED time step of size dt the residual of its last linearisation, times dt, is remembered per cell and
 * subtracted as a rate from the source term of the following steps, capped at max_compensation (default 10 x
 * NewtonTolerance = 0.1, :352-356, :1854) of the cell's pore volume per step.  opmhip_end_time_step stores
 * residual * dt on the device (asynchronous; a rolled-back step never reaches it, so opmhip_update_failed leaves the
 * drift alone); opmhip_assemble applies it.  opmhip_set_drift_compensation(enable = 0) = --ecl-enable-drift-compensation=false;
 * switching it either way clears the stored drift. */
int opmhip_end_time_step(opmhip_ctx* ctx, double dt);
int opmhip_set_drift_compensation(opmhip_ctx* ctx, int enable, double max_compensation);

/* replaces: EclProblem::source (ebos/eclproblem.hh:1823-1845): total surface-volume rate per cell and equation
 * [m^3/s] (what BlackoilWellModel::computeTotalRatesForDof adds up, wells/BlackoilWellModel_impl.hpp:496-512) and
 * its 3x3 derivative w.r.t. the cell's primary variables.  Either may be NULL (= zero). */
int opmhip_set_source(opmhip_ctx* ctx, const double* source, const double* dsource);
/* The same for a well model: the rates of `n` perforated cells (natural local ids; a cell named twice receives the sum), every other
 * cell's source and derivative zero.  source[n*3], dsource[n*9] (nullable = zero).  What crosses PCIe is 100 bytes per perforation
 * instead of 96 bytes per cell of the grid (computeTotalRatesForDof visits the perforations, wells/BlackoilWellModel_impl.hpp:496-512).
 * n == 0: no sources at all.  ABI 11 */
int opmhip_set_source_cells(opmhip_ctx* ctx, int n, const int* cells, const double* source, const double* dsource);

/* replaces: model().linearizer().linearizeDomain() (flow/BlackoilModelEbos.hpp:424), then .jacobian() /
 * .residual() (:339-340, :526-527).  iteration == 0 also (re)fills the cached old-time-level storage term
 * (recycleFirstIterationStorage, ebos/eclproblem.hh:1758-1765).  Jacobian and residual stay on the device for
 * opmhip_solve_system(vals = NULL, b = NULL); jac (nnzb*9) / residual (Nb*3) are optional host copies in natural
 * order. */
int opmhip_assemble(opmhip_ctx* ctx, double dt, int iteration, double* jac, double* residual);

/* cached intensive quantities, for parity tests: per cell 17 fields x (value, d/dSw, d/dp, d/dX):
 * S_w S_o S_g | p_w p_o p_g | b_w b_o b_g | mob_w mob_o mob_g | rho_w rho_o rho_g | Rs | porosity
 * With a PVTG or ROCKTAB fluid the record has 19 fields: ... | Rs | Rv | transmissibility multiplier | porosity
 * (opmhip_iq_fields tells which). */
int opmhip_iq_fields(opmhip_ctx* ctx);
int opmhip_get_iq(opmhip_ctx* ctx, double* out);
/* the records of `n` cells (natural local ids, any order, repeats allowed), out[n * fields * 4]: what a well model reads per Newton
 * iteration - BlackoilWellModel::updatePerforationIntensiveQuantities evaluates the perforated cells only
 * (wells/BlackoilWellModel_impl.hpp:1606-1630) - gathered on the device, so that the copy is 544 bytes per perforation.  ABI 11 */
int opmhip_get_iq_cells(opmhip_ctx* ctx, int n, const int* cells, double* out);

/* replaces: BlackoilModelEbos::localConvergenceData + computeCnvErrorPv + the CNV/MB formulas of
 * getReservoirConvergence (flow/BlackoilModelEbos.hpp:628-904).  out[17]: R_sum[3], maxCoeff[3], B_avg[3], pvSum,
 * cnvErrorPv, CNV[3], MB[3]; component order oil, water, gas. */
int opmhip_convergence(opmhip_ctx* ctx, double dt, double tol_cnv, double* out);

/* replaces: BlackoilModelEbos::updateSolution (flow/BlackoilModelEbos.hpp:549-563) = BlackOilNewtonMethod::update_
 * (dp <= 0.3 p, dS <= 0.2, primary-variable switching) + invalidateAndUpdateIntensiveQuantities, preceded by
 * NonlinearSolverEbos::stabilizeNonlinearUpdate "dampen" (flow/NonlinearSolverEbos.hpp:307-353): dx *= relax.
 * dx == NULL uses the solution of the last opmhip_solve_system, which is still on the device.
 * num_switched (nullable) receives the number of cells whose meaning changed. */
int opmhip_update(opmhip_ctx* ctx, const double* dx, double relax, int* num_switched);

/* ---- domain decomposition over the GPUs of a node (restricted additive Schwarz) --------------------------------- */
/* The reference's parallel runs are MPI domain decomposition with block-Jacobi/overlapping ILU0 and halo updates
 * through Dune::OwnerOverlapCopyCommunication (linalg/ISTLSolverEbos.hpp:101-105, 171-178;
 * linalg/ParallelOverlappingILU0.hpp:857-860, 897; flow/BlackoilModelEbos.hpp:599-603); its accelerator back-ends are
 * switched off under MPI (linalg/ISTLSolverEbos.hpp:136-141), so this surface is new.  One process per GPU, one context
 * per process; set_pattern_dd describes the subdomain.  Communication runs on the context's stream with RCCL
 * (ncclSend/ncclRecv per neighbour for halos, ncclAllReduce for the fused scalars).
 *   rank 0: opmhip_comm_unique_id(id) -> broadcast the 128 bytes by any means (MPI_Bcast, torch.distributed, a file)
 *   every rank: opmhip_comm_init_rccl(ctx, nranks, rank, id) ; opmhip_set_halo(...) */
int opmhip_comm_unique_id(char* id128);
int opmhip_comm_init_rccl(opmhip_ctx* ctx, int nranks, int rank, const char* id128);
/* several contexts inside ONE process (each driven by its own host thread, all on one GPU), connected by device copies
 * and a host barrier: test vehicle for the decomposition logic on a single GPU */
int opmhip_comm_init_loopback(opmhip_ctx* ctx, int nranks, int rank, const char* group_name);
/* diagnostics: one RCCL all-reduce of {1 + rank, 2} on the context's stream; sum_out[2] */
int opmhip_comm_selftest(opmhip_ctx* ctx, double* sum_out);
/* what the communicator itself reports: info[0] = ncclCommCount, [1] = ncclCommUserRank, [2] = ncclCommCuDevice (the HIP
 * device RCCL bound this rank to), [3] = kind (0 none, 1 loopback, 2 RCCL).  Loopback / no communicator: the context's own
 * numbers.  bench.py prints them so that a multi-GPU record shows how many ranks RCCL actually connected. */
int opmhip_comm_info(opmhip_ctx* ctx, int* info4);
/* optional, before opmhip_set_static: the global id of every local cell (Nb + Nghost entries).  The assembly then adds a
 * row's face fluxes in ascending GLOBAL neighbour order, i.e. exactly the sum an undecomposed run forms, so that
 * residual and Jacobian do not depend on the decomposition down to the last bit. */
int opmhip_set_cell_global_ids(opmhip_ctx* ctx, const long long* gids);
/* halo description: for neighbour q (rank neigh_rank[q]) the owned cells send_cells[send_ptr[q] .. send_ptr[q+1])
 * (local natural ids, in the order the neighbour numbers its ghosts) are sent, and ghost cells
 * Nb + recv_ptr[q] .. Nb + recv_ptr[q+1] are received.  global_cells = number of cells of the whole grid (B_avg is a
 * mean over all cells, flow/BlackoilModelEbos.hpp:722-727). */
int opmhip_set_halo(opmhip_ctx* ctx, long long global_cells, int nneigh, const int* neigh_rank, const int* send_ptr,
                    const int* send_cells, const int* recv_ptr);

/* ---- measurement -------------------------------------------------------------------------------------------- */
/* Device-side timing per kernel class with HIP events recorded on the context's own stream, so that bench.py can
 * state the average launch duration of a kernel over its timed region (the reference prints the same split at
 * verbosity >= 3/4: bda/cusparseSolverBackend.cu:303-308, bda/openclSolverBackend.cpp:451-459).
 * classes: 0 SpMV, 1 ILU0 apply (all sweeps of one M^-1; with the line-coloured ordering the first sweep also carries
 * the BiCGStab p- / (r, x)-update), 2 ILU0 factorisation, 3 the remaining BiCGStab vector / reduction kernels (one
 * scope between two operator applications), 4 assembly kernel, 5 intensive-quantity update, 6 convergence, 7 the pressure
 * AMG cycle of CPR, 8 decomposed runs: the second launch of an operator application (the boundary tiles, multiplied after
 * the halo exchange that ran beside the interior tiles of class 0).
 * Communication spans of decomposed runs (ABI 8; they OVERLAP the kernel classes - a halo exchange runs beside class 0, an
 * all-reduce inside class 3 - and are what the reference's back-ends have no counterpart for, being single-process): 9 one halo
 * exchange, pack -> send / receive -> ghosts in place, on the stream it runs on (Dune's copyOwnerToAll in front of the operator,
 * linalg/ParallelOverlappingILU0.hpp:897, WellOperators.hpp:127-138); 10 one global reduction, local sums' kernel -> all-reduce ->
 * result on the device (BiCGStab's scalar products; the convergence sums of BlackoilModelEbos.hpp:599-603); 11 CPR across the ranks:
 * the all-gather of the joined level's right-hand side plus the cycle every rank runs on it. */
#define OPMHIP_PROF_CLASSES 12
/* on = 0: off; 1: every scope; k > 1: the linear-solver classes (0, 1, 3) are recorded in every k-th solve_system call
 * only - an event record costs a few microseconds of bubble on the stream and a BiCGStab iteration holds seven of
 * them.  Also resets the accumulated numbers. */
int opmhip_profile_enable(opmhip_ctx* ctx, int on);
/* waits for the stream, folds all pending event pairs in, returns launches and total milliseconds of one class */
int opmhip_profile_get(opmhip_ctx* ctx, int cls, long long* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* OPMHIP_H */
