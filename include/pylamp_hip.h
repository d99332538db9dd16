/* pylamp_hip.h — C ABI of libpylamp_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for PyLamp's per-time-step hot path.  The reference has no FFI layer:
 * its boundary is the Python module API of pylamp_stokes.py / pylamp_diff.py /
 * pylamp_trac.py called from the loop body of pylamp2.py.  the modules under pylamp_amd/ keep those
 * names and signatures and forwards to the entry points declared here through ctypes
 * (see INTEGRATION.md for the binding a PyLamp maintainer would add).
 *
 * Conventions
 *   - plain pointers and sizes only; all floating point data is IEEE double;
 *   - host arrays are C-contiguous, shape (nz, nx) indexed [i=z][j=x]
 *     (reference: pylamp_const.py:9-13, pylamp2.py:37,100-113);
 *   - Stokes vectors use the reference DOF order: (vz, vx, P) interleaved per node,
 *     nodes row-major, length 3*nz*nx (pylamp_stokes.py:22-35 gidx, 86-101 x2vp);
 *   - every function returns 0 on success, non-zero on error; pl_last_error() gives the
 *     message (the Python side raises Exception(msg), the reference's error convention);
 *   - one context per GPU / rank; calls on one context are not thread-safe;
 *   - the library never keeps a host pointer after a call returns;
 *   - there is no CPU fallback: without a usable HIP device pl_create fails.
 */
#ifndef PYLAMP_HIP_H
#define PYLAMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pl_ctx pl_ctx;

/* Boundary-condition codes (pylamp_stokes.py:17-20, pylamp_diff.py:12-13). */
enum { PL_BC_NOSLIP = 0, PL_BC_FREESLIP = 1, PL_BC_CYCLIC = 2, PL_BC_FLOWTHRU = 4 };
enum { PL_BC_FIXTEMP = 0, PL_BC_FIXFLOW = 1 };
/* Averaging schemes / interpolation methods (pylamp_trac.py:11-22). */
enum { PL_AVG_ARITHMETIC = 1, PL_AVG_GEOMETRIC = 2, PL_AVG_WEIGHTED = 4 };
enum { PL_INTERP_NEAREST = 8, PL_INTERP_LINEAR = 16, PL_INTERP_VELDIV = 32 };

typedef struct pl_solve_stats {
    int    iterations;      /* outer Krylov iterations used                       */
    int    converged;       /* 1 iff rel_residual <= rtol (the recomputed residual, never the recurrence) and, for Stokes,
                             * error_estimate <= 1.5 x its bound (below) */
    double rel_residual;    /* TRUE residual ||D(r0 - A dx)|| / ref, recomputed with the operator, where
                             * r0 = b - A x0 is evaluated once and x = x0 + dx (correction form);
                             * D = row scaling; ref = ||D b|| (heat) or the dynamic load
                             * ||D(b - A x_hydrostatic)|| (Stokes) */
    double solve_ms;        /* device time of the solve (HIP events)              */
    int    operator_applies;
    int    precond_applies;
    int    used_direct;     /* 1: the multigrid-preconditioned iteration did not converge and the banded-LU fallback for
                             * small single-GPU systems finished the solve (indefinite systems: the reference's
                             * free-surface stabilisation sign at the Courant step, pylamp2.py:387-405) */
    int    reserved_;       /* 0 */
    double error_estimate;  /* Stokes: estimate of the relative velocity error of the returned iterate from its true residual r
                             * (the residual norm alone does not bound it: error / residual is ~10 on large smooth problems and
                             * ~1e4 on coarse ones):
                             *     (n |r_cont| + |(M^-1 r)_vel|) / |x_vel|  +  |y.r_p / y.A w| |w_vel| / |x_vel|
                             * n = max(nz, nx, L_z / min dz, L_x / min dx) (graded grids count by their finest cell), M^-1 = one
                             * preconditioner application, the last term = the component of r along the deflated pressure-anchor
                             * mode w (it is amplified by |w| / |x|, not by n).  Between the exact evaluations (each costs a
                             * preconditioner application) the iteration uses (n |r_cont| + a |r_mom|) / |x_vel| on the
                             * recurrence residual, a = |(M^-1 s)_vel| / |s_mom| measured on the way.  A solve whose residual meets
                             * rtol keeps iterating until the estimate is <= 3e-8 (PYLAMP_STOKES_ETOL; 0 switches the test off).
                             * Viscosity contrast above 3e5 (PYLAMP_CONTRAST_GATE): the ROW-SCALED residual no longer bounds
                             * anything (reference model 5, contrast 1e10: scaled residual 7e-11 at a velocity error of 4e-3), so
                             * systems the banded LU can hold go there up front and every result is judged by its UNSCALED
                             * residual: error_estimate = max(nz, nx) |b - A x|_2 / |b - A x_hydrostatic|_2, converged iff that is
                             * <= 1e-6 as well.  0 when not evaluated (heat, or the solve ended otherwise) */
} pl_solve_stats;

/* ---- context --------------------------------------------------------------------- */
/* Global grid of nz x nx nodes with node coordinates zc[nz], xc[nx] (non-uniform
 * allowed for Stokes/heat; MIC requires a regular grid like the reference,
 * pylamp_trac.py:34,162).  The context owns all device memory and one HIP stream. */
int  pl_create(pl_ctx** out, int device, int nz, int nx, const double* zc, const double* xc);
void pl_destroy(pl_ctx* ctx);
const char* pl_last_error(const pl_ctx* ctx);   /* ctx may be NULL: error of a failed pl_create */
int  pl_sync(pl_ctx* ctx);                      /* hipStreamSynchronize on the context stream */
int  pl_device_info(pl_ctx* ctx, char* name, size_t name_len, int* cu_count, size_t* hbm_bytes);

/* ---- multi-GPU: one context per rank, 2-D block decomposition of the node grid ---------------------------------
 * The reference only strides tracers over MPI ranks and replicates every grid array
 * (pylamp2.py:30-32,445-455,550-555); here the grid itself is decomposed into Pz x Px blocks (1x2, 2x2, 2x4 ...).
 * Rank r = pz * Px + px owns node rows [pz*Cz, (pz+1)*Cz) and columns [px*Cx, (px+1)*Cx), Cz = (nz-1)/Pz,
 * Cx = (nx-1)/Px (the last block of an axis also owns the last node row / column); (nz-1) and (nx-1) must be
 * divisible by Pz resp. Px, with even quotients >= 8.  Every plane carries a halo ring filled from the (up to 8)
 * neighbour blocks, corners included (the momentum stencils reach (i-1,j+1) and (i+1,j-1)).
 * Transports: (a) direct RCCL calls on the context stream (opt-in, PYLAMP_RCCL=1, self-tested at start-up);
 * (b) this callback table into the host program (torch.distributed in pylamp_amd/parallel.py); (c) an in-process
 * group of "virtual ranks" sharing one GPU, one host thread per context (pl_local_group_*; tests and rehearsals).
 * Every callback returns 0 on success and must have completed when it returns.  Device pointers are plain HIP
 * allocations of this context's device. */
typedef struct pl_comm_ops {
    /* nmsg point-to-point messages: message k sends nsend[k] doubles at send[k] to rank peer[k] and receives nrecv[k]
     * doubles into recv[k] from it (either count may be 0).  Messages between the same pair of ranks are matched in
     * list order; both sides list them in the same order. */
    int (*sendrecv)(void* user, int nmsg, const int* peer, const double* const* send, const int64_t* nsend,
                    double* const* recv, const int64_t* nrecv);
    /* In-place all-reduce of n doubles in HOST memory; op: 0 sum, 1 min, 2 max. */
    int (*allreduce_host)(void* user, double* buf, int64_t n, int op);
    /* All-gather: every rank contributes count doubles at send; recv receives nranks*count (rank r at r*count). */
    int (*allgather)(void* user, const double* send, double* recv, int64_t count);
    void* user;
} pl_comm_ops;
/* Right after pl_create.  Pz * Px must equal nranks; pl_set_comm chooses the layout itself: PYLAMP_DECOMP="PzxPx",
 * otherwise the most square one with Px >= Pz (2 -> 1x2, 4 -> 2x2, 8 -> 2x4). */
int  pl_set_comm(pl_ctx* ctx, int rank, int nranks, const pl_comm_ops* ops);
int  pl_set_comm_2d(pl_ctx* ctx, int rank, int Pz, int Px, const pl_comm_ops* ops);
int  pl_local_rows(pl_ctx* ctx, int* first_row, int* n_rows);
int  pl_local_block(pl_ctx* ctx, int* first_row, int* n_rows, int* first_col, int* n_cols, int* Pz, int* Px);
/* In-process group of virtual ranks: create one group, then one context per rank (same or different devices), attach
 * each with pl_set_comm_local and drive every context from its OWN host thread (the calls are collective and block
 * until all ranks of the group have entered them). */
typedef struct pl_local_group pl_local_group;
int  pl_local_group_create(pl_local_group** out, int nranks);
void pl_local_group_destroy(pl_local_group* g);
void pl_local_group_abort(pl_local_group* g);    /* a rank's driver thread failed: every collective call of the group returns an error */
int  pl_set_comm_local(pl_ctx* ctx, pl_local_group* g, int rank, int Pz, int Px);
/* Bracket a rank thread's stretch of library calls.  No-ops unless PYLAMP_LOCAL_SERIAL=1 was set when the group was created: then the
 * group has ONE GPU token, held by the thread that runs and handed over (stream drained) while it waits inside a collective call, so
 * that the kernels of virtual ranks sharing a GPU never overlap -- a kernel trace then shows their true durations. */
void pl_local_group_enter(pl_local_group* g);
void pl_local_group_leave(pl_local_group* g, pl_ctx* ctx);
/* *native = 1 when the exchanges run as direct RCCL calls on the context stream (dlopen'ed librccl, self-tested at
 * pl_set_comm), 2 for the in-process group, 0 when they go through the callback table.
 * The native path is opt-in: PYLAMP_RCCL=1 (bench.py sets it for --transport native only; it has not run on a multi-GPU node yet). */
int  pl_comm_info(pl_ctx* ctx, int* rank, int* nranks, int* native);
/* Cumulative numbers of communication calls of this context: out[0] neighbour (halo) exchanges, [1] all-gathers,
 * [2] device all-reduces, [3] host all-reduces; reset != 0 clears the counters after reading. */
int  pl_comm_stats(pl_ctx* ctx, int64_t out[4], int reset);
/* Milliseconds spent in them since the last reset: out_ms[0..2] device time between HIP events recorded around every neighbour
 * exchange (pack -> messages -> unpack), all-gather and device all-reduce on the context stream; out_ms[3] host wall time of
 * the host all-reduces.  Synchronises the stream. */
int  pl_comm_times(pl_ctx* ctx, double out_ms[4], int reset);
/* raw copies between host and this context's device memory (used by the gloo fallback of the
 * communication layer, which stages through host buffers) */
int  pl_memcpy_d2h(pl_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int  pl_memcpy_h2d(pl_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int  pl_dev_add(pl_ctx* ctx, double* dst_dev, const double* src_dev, int64_t n);   /* dst += src */

/* HIP-event timing on the context's stream (bench.py measures kernels with these). */
int  pl_timer_start(pl_ctx* ctx);
int  pl_timer_stop_ms(pl_ctx* ctx, double* ms);

/* ---- Stokes: replaces makeStokesMatrix (pylamp_stokes.py:104-563) + the spsolve call
 *      site pylamp2.py:360 ------------------------------------------------------------ */
/* Upload viscosity/density and fix the operator.  bc = [z0, x0, zL, xL]
 * (index DIM*wall+dir as in pylamp_stokes.py:163,202,242,289).  Computes Kcont/Kbond
 * exactly as pylamp_stokes.py:116-122.  surfstab != 0 adds the stabilisation terms of
 * pylamp_stokes.py:422-426,483-487 with the given tstep and theta.  The terms are linear in
 * theta; the reference's sign makes them ANTI-stabilising (they reduce the momentum diagonal, see
 * DESIGN.md section 5), a negative theta gives the damping sign of Duretz et al. (2011). */
int  pl_stokes_set_coeffs(pl_ctx* ctx, const double* etas, const double* etan, const double* rho,
                          const int bc[4], int surfstab, double tstep, double theta);
int  pl_stokes_get_scaling(pl_ctx* ctx, double* kcont, double* kbond);
/* y = A x, matrix-free (what A@x would give for the lil_matrix the reference builds). */
int  pl_stokes_apply(pl_ctx* ctx, const double* x, double* y);
/* rhs vector of makeStokesMatrix (pylamp_stokes.py:429,490). */
int  pl_stokes_rhs(pl_ctx* ctx, double* rhs);
/* x = A^-1 rhs by preconditioned BiCGStab (replaces spsolve, pylamp2.py:360,394).
 * rhs == NULL uses the operator's own rhs.  x is output only (initial guess 0) unless
 * use_x0 != 0.
 * Stopping rule: the solve ends when BOTH the true relative residual is <= rtol AND the estimated relative velocity error
 * (pl_solve_stats.error_estimate) is <= 3e-8.  It is an estimate, not a bound: against direct solves it has been seen up to 6 x (10^3
 * block, cold start) and 12 x (sphere at contrast 1e5) below the true error, so 3e-8 leaves a factor of 30 to the 1e-6 the drop-in
 * promises against the reference's direct solve (measured true errors: 1e-9 ... 2e-7).  The Python layers pass rtol = 1e-7 by default; a smaller rtol is honoured as a residual bound. */
int  pl_stokes_solve(pl_ctx* ctx, const double* rhs, double* x, int use_x0, double rtol,
                     int maxit, pl_solve_stats* stats);
/* Times `reps` back-to-back applies of the device-resident operator on device-resident
 * vectors with HIP events (no host transfers); returns average ms per apply. */
int  pl_stokes_apply_bench(pl_ctx* ctx, int reps, double* avg_ms);
/* the same for the row-scaled operator y = D_r A x, the variant the Krylov solver launches */
int  pl_stokes_apply_scaled_bench(pl_ctx* ctx, int reps, double* avg_ms);
/* stream triad a = b + s c on three device arrays of n doubles (24 n bytes per launch): measured HBM rate of the box */
int  pl_stream_triad_bench(pl_ctx* ctx, int64_t n, int reps, double* avg_ms);

/* Diagnostics for component tests of the solver: z = M^-1 r for an UNSCALED residual r in
 * the reference DOF order (builds the multigrid hierarchy for the current coefficients; the
 * wall/slave/ghost velocity rows of r are ignored - they are identically zero in the solver), and
 * the hierarchy's level count / per-level Chebyshev lambda_max. */
int  pl_stokes_precond_apply(pl_ctx* ctx, const double* r, double* z);
int  pl_stokes_mg_info(pl_ctx* ctx, int* nlevels, double* lmax, int max_levels);
/* Average duration (HIP events) of one Chebyshev smoothing sweep on the finest multigrid level -- the kernel
 * with the largest share of a time step (80 B/node algorithmic).  Call after at least one solve. */
int  pl_stokes_sweep_bench(pl_ctx* ctx, int reps, double* avg_ms);
/* Precision of the velocity-block multigrid of the last solve: the first *nlevels_fp32 of the *nlevels levels store and
 * sweep in FP32 (the large, bandwidth-bound ones; off by default, PYLAMP_MG_FP32=1 or pl_stokes_set_mg_precision).  BiCGStab, the operator and
 * the stopping test are FP64 always: the V-cycle only approximates A_vv^-1.  pl_stokes_sweep_bench times the finest
 * level in the precision reported here (40 B/node algorithmic in FP32, 80 in FP64). */
int  pl_stokes_mg_precision(pl_ctx* ctx, int* nlevels, int* nlevels_fp32);
/* fp32 = 0: all levels FP64 (default); 1: the large levels in FP32.  min_nodes > 0: smallest level, in nodes of the rank's block, that
 * runs in FP32 (default 200000: smaller levels are launch-latency bound and gain nothing). */
int  pl_stokes_set_mg_precision(pl_ctx* ctx, int fp32, long long min_nodes);

/* ---- Heat: replaces makeDiffusionMatrix (pylamp_diff.py:85-183) + spsolve
 *      (pylamp2.py:419) --------------------------------------------------------------- */
int  pl_heat_set_coeffs(pl_ctx* ctx, const double* zmp, const double* xmp, const double* T,
                        const double* kz, const double* kx, const double* cp, const double* rho,
                        const double* H, const int bc[4], const double bcvalue[4], double tstep);
int  pl_heat_apply(pl_ctx* ctx, const double* x, double* y);
int  pl_heat_rhs(pl_ctx* ctx, double* rhs);
int  pl_heat_solve(pl_ctx* ctx, const double* rhs, double* x, double rtol, int maxit,
                   pl_solve_stats* stats);
int  pl_heat_apply_bench(pl_ctx* ctx, int reps, double* avg_ms);

/* ---- Marker-in-cell: replaces pylamp_trac.trac2grid / grid2trac / RK ----------------- */
/* Tracer -> grid (pylamp_trac.py:161-318, method ELEM).  tr_x is (n,2) [z,x]; tr_f is
 * (n,nf) with leading dimension ld_f; target node set given by its first coordinate and
 * spacing per axis (any of the 4 staggerings; regular grid); out[k] are nf host arrays
 * (nz,nx) written in place.  Nodes outside [0,nz)x[0,nx) are discarded, which is what
 * the reference's grid extension + crop (pylamp_trac.py:207-220,313-316) amounts to. */
int  pl_trac2grid(pl_ctx* ctx, int64_t n, const double* tr_x, const double* tr_f, int64_t ld_f,
                  int nf, const int* avgscheme, double z0, double hz, double x0, double hx,
                  double* const* out);
/* Rectilinear (non-uniform) grids, SURVEY 8 f4 -- beyond the reference, whose marker code is regular-grid
 * only (it picks the cell with the regular-grid formula, pylamp_trac.py:46-47,226-227).  pl_trac2grid_rect takes
 * the coordinates zc[nz], xc[nx] of the target node set and locates cells by per-axis search (outside the
 * axis the grid continues with the spacing of its end cell, as the reference's extension does).
 * pl_mic_set_search(ctx, 1) makes pl_grid2trac and pl_rk4 locate cells the same way in their gz/gx arrays;
 * 0 (default) restores the reference's formula.  On a uniform grid both give the same result. */
int  pl_trac2grid_rect(pl_ctx* ctx, int64_t n, const double* tr_x, const double* tr_f, int64_t ld_f,
                       int nf, const int* avgscheme, const double* zc, const double* xc, double* const* out);
int  pl_mic_set_search(pl_ctx* ctx, int on);
/* Grid -> tracer (pylamp_trac.py:30-158).  fields: nf host arrays (gnz,gnx) on the regular
 * grid gz[gnz], gx[gnx]; out is (n,nf) with leading dimension ld_out (may be a strided
 * view like pylamp2.py:445).  *n_outside returns how many tracers used defval; with
 * stop_on_error != 0 and any outside the call fails (pylamp_trac.py:53-54). */
int  pl_grid2trac(pl_ctx* ctx, int64_t n, const double* tr_x, int nf, const double* const* fields,
                  int gnz, int gnx, const double* gz, const double* gx, int method, double defval,
                  int stop_on_error, double* out, int64_t ld_out, int64_t* n_outside);
/* RK(order=4) (pylamp_trac.py:347-388): vels are (gnz,gnx) on gz,gx (the padded
 * cell-centre grid, pylamp2.py:491-545).  v_out, x_out are (n,2). */
int  pl_rk4(pl_ctx* ctx, int64_t n, const double* tr_x, int gnz, int gnx, const double* gz,
            const double* gx, const double* vz, const double* vx, double tstep, double* v_out,
            double* x_out);

/* ---- Device-resident time step (the build's counterpart of pylamp2.py:273-581) ------- */
typedef struct pl_step_config {
    int    do_heatdiff, do_subgrid_heatdiff, tdep_rho, tdep_eta;
    double etamin, etamax, tref;
    double tstep_adv_max, tstep_adv_min, tstep_dif_max, tstep_dif_min, tstep_modifier;
    int    bcstokes[4];
    int    bcheat[4];
    double bcheatvals[4];
    double stokes_rtol, heat_rtol;
    int    stokes_maxit, heat_maxit;
    double length[2];               /* domain size L[z], L[x] (pylamp2.py:38) */
    /* Tracer census + injection at the end of the step (pylamp2.py:588-633): cells holding fewer
     * than tracdens_min tracers are refilled to tracdens with tracers at (seeded) random positions
     * whose 12 fields are the plain mean of the tracers already in the cell.  tracdens_min <= 0
     * disables it.  IDs follow the reference (pylamp2.py:621-622): the first new ID of every refilled cell repeats
     * the last ID handed out; inject_unique_ids != 0 continues after the current maximum instead.  Positions come
     * from a counter-based generator seeded with inject_seed (the reference draws from numpy's global stream). */
    int    tracdens, tracdens_min;
    uint64_t inject_seed;
    /* Free-surface stabilisation (pylamp2.py:71-73,352-355,368-372,387-405): with
     * surfstab_tstep < 0 the Stokes system is re-assembled with the chosen time step and re-solved
     * until the step no longer shrinks (limiter 's' = the reference's "Ss").  surfstab_theta < 0
     * selects the corrected (damping) sign, see pl_stokes_set_coeffs. */
    int    surface_stabilization;
    double surfstab_theta, surfstab_tstep;
    int    inject_unique_ids;       /* see tracdens above */
    /* pylamp2.py:41,563-581: with the fence off a tracer at or beyond a wall is deleted (TR__ID = -1 -> np.delete)
     * instead of being put back EPS inside the wall.  0 = fence on (the reference's default). */
    int    tracs_fence_disabled;
} pl_step_config;

typedef struct pl_step_report {
    double tstep;                   /* chosen time step (pylamp2.py:374-385) */
    int    limiter;                 /* 'H', 'S' or 's' (= "Ss") */
    double tstep_heat, tstep_stokes;
    pl_solve_stats stokes, heat;
    double ms_props, ms_scatter, ms_stokes, ms_heat, ms_gather, ms_advect, ms_sort, ms_total;
    int64_t ntrac;                  /* tracers on this rank after the step (incl. injected) */
    int64_t ninjected;              /* tracers injected on this rank at the end of this step */
    int     stokes_resolves;        /* extra Stokes solves of the surface-stabilisation loop */
    int64_t nremoved;               /* tracers deleted on this rank at the end of this step (fence off) */
} pl_step_report;

/* Upload tracer state: tr_x (n,2), tr_f (n,13) AoS rows as in pylamp_const.py:29-42. */
int  pl_tracers_upload(pl_ctx* ctx, int64_t n, const double* tr_x, const double* tr_f);
int  pl_tracers_download(pl_ctx* ctx, int64_t n, double* tr_x, double* tr_f);
int  pl_tracers_count(pl_ctx* ctx, int64_t* n);
/* Census of the resident tracers (pylamp2.py:588-598: np.bincount of the cell index): counts[c] for this rank's
 * owned cells, row-major (n_cell_rows x (nx-1)), ncells = their number.  counts == NULL only queries the rows. */
int  pl_tracers_census(pl_ctx* ctx, int64_t ncells, int32_t* counts, int* first_cell_row, int* n_cell_rows);
/* Layout of the resident tracer columns (tests; no counterpart in the reference, whose tracers are plain NumPy rows).  The
 * end-of-step sort of the resident step moves only positions, temperature and a 4-byte slot per tracer:
 *   *epoch_age > 0: the ten columns no stage writes (everything but TR_RHO, TR_ETA, TR_TMP) are still in the order of the sort
 *                   that opened the epoch, *epoch_age sorts ago, and are read through the slot index;
 *   *lazy != 0:     TR_RHO, TR_ETA and the tracer velocities have not been moved by the last sort yet.
 * Downloads (pl_tracers_download, pl_get_tracer_velocity) bring every column into the current order first. */
int  pl_tracers_layout(pl_ctx* ctx, int* epoch_age, int* lazy);
/* One full time step on the device-resident state (pylamp2.py:284-583: properties, tracer -> grid, Stokes, time step, heat,
 * grid -> tracer with subgrid diffusion, RK4 advection, census + injection).  The iterative solves are warm-started from the
 * context's own history -- Stokes from the polynomial through its last three solutions evaluated at the new model time, heat
 * from the old nodal temperature plus the last increment scaled by the time-step ratio; pl_tracers_upload forgets the history.
 * The stopping rules are those of pl_stokes_solve / pl_heat_solve whatever the start, so the reference's results are reproduced
 * within the solver tolerances either way (PYLAMP_X0_EXTRAP=0 / PYLAMP_HEAT_X0=0 start from the last solution / from zero). */
int  pl_step(pl_ctx* ctx, const pl_step_config* cfg, int it, pl_step_report* rep);
/* The marker stages of pl_step ONE AT A TIME on the resident, cell-sorted tracers (one rank) -- the same functions and kernels
 * the timed step runs (fused k_scatter_cells, k_gather<true, 0|1|2>, k_rk4<true>), so that tests can pin them directly against
 * the reference's fixtures (pylamp_trac.py:161-318 trac2grid, :30-158 grid2trac, :321-388 RK) instead of through trajectories.
 * pl_resident_scatter: properties + the step's tracer -> grid calls (pylamp2.py:291-319); results by pl_get_field ("rho", "etas",
 *   "etan", "cp", "f_T", "H", "mat", "kz", "kx").
 * pl_resident_temp_to_tracers: pylamp2.py:436-480 with the new nodal temperature newtemp (nz,nx); the old nodal temperature is
 *   the plane "f_T" of the last scatter; first != 0 is the it == 1 branch; results in column TR_TMP (pl_tracers_download).
 * pl_resident_rk4: RK4 through vz_pad, vx_pad on the padded (nz+1, nx+1) centre grid (pylamp2.py:547-572, fence optional), then
 *   the end-of-step cell sort; positions by pl_tracers_download, velocities by pl_get_tracer_velocity. */
int  pl_resident_scatter(pl_ctx* ctx, const pl_step_config* cfg, int it);
int  pl_resident_temp_to_tracers(pl_ctx* ctx, const pl_step_config* cfg, int first, const double* newtemp, double tstep);
int  pl_resident_rk4(pl_ctx* ctx, const double* vz_pad, const double* vx_pad, double tstep, int fence, const double length[2]);
/* Copy a named grid field of the last step to host, shape (nz,nx): "velz","velx","pres",
 * "rho","etas","etan","temp","f_T","kz","kx","cp","H". */
int  pl_get_field(pl_ctx* ctx, const char* name, double* out);
/* Velocity of the last advection, (n,2) like trac_vel (pylamp2.py:547-555). */
int  pl_get_tracer_velocity(pl_ctx* ctx, int64_t n, double* out);

/* ---- 3-D staggered Stokes + heat (BASELINE config 5) ----------------------------------------------------------------
 * PARITY UNPINNED: the reference implements DIM = 2 only (pylamp_const.py:6; pylamp_stokes.py:30-35 prints "NOT
 * IMPLEMENTED" for dim != 2).  It fixes the intent -- axis order z, x, y (pylamp_const.py:9-13), arrays (nz, nx, ny),
 * IP = DIM, DOF order iz*nx*ny*4 + ix*ny*4 + iy*4 + ieq (pylamp_stokes.py:24) -- and these entry points extend the 2-D
 * rows (pylamp_stokes.py:376-518, pylamp_diff.py:99-179) dimension by dimension: a y-invariant extrusion reproduces the
 * 2-D operator and solution on every y-slice.  Host arrays are C-order (nz, nx, ny) doubles; Stokes vectors are
 * (nz, nx, ny, 4) with components (vz, vx, vy, P).  All walls free-slip; heat walls FIXTEMP / FIXFLOW in the order
 * [z0, x0, y0, zL, xL, yL].  The shear viscosity is given at the NODES and averaged onto the edges. */
typedef struct pl3_ctx pl3_ctx;
int  pl3_create(pl3_ctx** out, int device, int nz, int nx, int ny, const double* zc, const double* xc, const double* yc);
void pl3_destroy(pl3_ctx* ctx);
const char* pl3_last_error(const pl3_ctx* ctx);
/* Several ranks (BASELINE config 5: "1 -> 8 MI355X"; the reference has no 3-D code at all, pylamp_const.py:6, pylamp_stokes.py:24-35):
 * right after pl3_create -- with the GLOBAL grid on every rank -- this context becomes one block of a Pz x Px x Py decomposition of
 * the node grid, rank = (pz Px + px) Py + py, (n - 1) divisible by the block count along every axis, >= 4 cells per block.  `comm`
 * is a 2-D context (pl_create on any small grid) that carries the transport: its pl_set_comm / pl_set_comm_2d / pl_set_comm_local must
 * have been called with Pz * Px * Py ranks, and it must outlive this context (whose stream it shares from then on).  Host arrays stay
 * GLOBAL on every rank (the block and the one ring of halo nodes the stencils reach are cut out of them; results are assembled).
 * Every pl3_* call that follows is collective.  Halo: one node deep, exchanged axis by axis (z, x, y planes: 6 messages carry faces,
 * edges and corners); every multigrid level is distributed (blocks halved with the grid), dot products are all-reduced on the host. */
int  pl3_set_comm(pl3_ctx* ctx, pl_ctx* comm, int Pz, int Px, int Py);
int  pl3_local_block(pl3_ctx* ctx, int first[3], int count[3]);
/* out[0] halo exchanges, out[1] host all-reduces of this context so far; reset != 0 clears the counters */
int  pl3_comm_stats(pl3_ctx* ctx, int64_t out[2], int reset);
/* grav = G[3] (NULL: (9.81, 0, 0), pylamp_const.py:21); Kcont = 3 min(eta) / sum(avgd), Kbond = 9 min(eta) / sum(avgd)^2 */
int  pl3_stokes_set_coeffs(pl3_ctx* ctx, const double* etas, const double* etan, const double* rho, const double grav[3]);
/* slaved != 0 (default): the reference's wall rows extended to 3-D (outermost in-domain tangential velocities slaved to their
 * inner neighbours, pylamp_stokes.py:170-175 ...: first-order accurate at the walls); 0: natural mirror rows (second order) */
int  pl3_stokes_set_wall_rows(pl3_ctx* ctx, int slaved);
int  pl3_stokes_get_scaling(pl3_ctx* ctx, double* kcont, double* kbond);
int  pl3_stokes_apply(pl3_ctx* ctx, const double* x, double* y);
int  pl3_stokes_rhs(pl3_ctx* ctx, double* rhs);
/* x == NULL: device-resident solve -- nothing crosses PCIe, the solution stays in the context (pl3_get_solution), and use_x0 != 0
 * starts from the solution of the context's previous solve. */
int  pl3_stokes_solve(pl3_ctx* ctx, const double* rhs, double* x, int use_x0, double rtol, int maxit, pl_solve_stats* stats);
int  pl3_stokes_apply_bench(pl3_ctx* ctx, int scaled, int reps, double* avg_ms);     /* 80 B/node algorithmic */
int  pl3_stokes_mg_info(pl3_ctx* ctx, int* nlevels, double* lmax, int max_levels);
int  pl3_heat_set_coeffs(pl3_ctx* ctx, const double* zmp, const double* xmp, const double* ymp, const double* T, const double* kz,
                         const double* kx, const double* ky, const double* cp, const double* rho, const double* H, const int bc[6],
                         const double bcvalue[6], double tstep);
int  pl3_heat_apply(pl3_ctx* ctx, const double* x, double* y);
int  pl3_heat_rhs(pl3_ctx* ctx, double* rhs);
/* rhs must be NULL (the operator's own right-hand side, pl3_heat_rhs); x == NULL keeps the solution on the device */
int  pl3_heat_solve(pl3_ctx* ctx, const double* rhs, double* x, double rtol, int maxit, pl_solve_stats* stats);
/* solution of the last solve of this context: which = 0 Stokes (nz, nx, ny, 4), 1 heat (nz, nx, ny) */
int  pl3_get_solution(pl3_ctx* ctx, int which, double* out);

/* sizeof / offsetof of the structs above as compiled into the library: out = { sizeof(pl_solve_stats),
 * sizeof(pl_step_config), sizeof(pl_step_report), offsetof(config.length), offsetof(config.inject_seed),
 * offsetof(config.tracs_fence_disabled), offsetof(report.ntrac), offsetof(report.nremoved) } -- lets a binding
 * verify its mirror of the layout (tests/test_cabi.py). */
int  pl_abi_layout(size_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* PYLAMP_HIP_H */
