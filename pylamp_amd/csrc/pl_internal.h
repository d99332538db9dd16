// Internal declarations shared by the HIP translation units of libpylamp_hip.so.
// gfx950 only; wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include "../../include/pylamp_hip.h"

#define PL_PADL 16           // doubles of left padding: interior column 0 is 128-B aligned
#define PL_RING 6            // halo depth every plane carries: rows above/below and columns right of the local block
                             // (the left padding holds the halo columns there); deep enough for a smoothing
                             // sequence of a distributed multigrid level to run on ONE exchange
#define PL_WAVE 64
#define PL_TOFF 2            // 1-D tables are indexed by global node index + PL_TOFF (even: pairs are 16-B aligned)

// ---- device-side view of one grid block (fine grid or a multigrid level) -------------
// Local block of lnz x lnx nodes whose node (0,0) is global node (gi0,gj0); every 2-D
// plane carries a ring of PL_RING nodes (halo for neighbour ranks, unused at global walls).
struct PlGeom {
    int nz, nx;              // global node counts
    int lnz, lnx;            // local block
    int gi0, gj0;            // global index of local node (0,0)
    int pitch;               // doubles per row
    long long plane;         // doubles per plane = (lnz + 2 PL_RING) * pitch
    // 1-D tables indexed by GLOBAL node index + PL_TOFF, zero padded on both sides, 16-B aligned:
    const double* zc; const double* xc;
    const double* rdz; const double* rdx;   // rdz[i] = 1/(z[i+1]-z[i])
    const double* rDz; const double* rDx;   // rDz[i] = 1/(z[i+1]-z[i-1])
};

__host__ __device__ inline long long pl_idx(const PlGeom& g, int li, int lj) {
    return (long long)(li + PL_RING) * g.pitch + (lj + PL_PADL);
}

struct PlStokesOp {
    PlGeom g;
    const double* etas; const double* etan; const double* rho;
    double Kc, Kb, iKc;      // iKc = 1/Kc
    int bc_z0, bc_zL;        // z-wall BC for the tangential rows (x-walls are FREESLIP)
    int surfstab; double ss; // ss = theta * tstep
    int anchor_i, anchor_j;  // pressure anchor cell (3,2) (pylamp_stokes.py:536-551)
    double gz, gx;           // gravity components G[IZ], G[IX] (pylamp_const.py:21)
    int scaled;              // 1: y = D_r A x (Jacobi-like row scaling used by the Krylov solver)
    int wall_ps;             // 1: the pressure block of the preconditioner uses the wall stencils where cells are stretched (prec_p_value)
};

struct PlHeatOp {
    PlGeom g;
    const double* kz; const double* kx; const double* rhocp_inv_dt; // dt/(rho*Cp)
    const double* rdzb; const double* rdxb;  // 1/(zm[i]-zm[i-1]) tables (global idx + 1)
    int bc[4];
    double dt;
};

// ---- host-side context ----------------------------------------------------------------
struct PlGeomHost {
    PlGeom d;                               // device view
    std::vector<double> zc, xc;             // host copies of the node coordinates
    double* tables = nullptr;               // one device allocation holding the 6 tables
    bool uniform = true;
};

struct pl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    int nz = 0, nx = 0;
    PlGeomHost geom;
    // generic staging
    double* stage = nullptr; size_t stage_bytes = 0;     // device staging buffer
    // named device planes / vectors, allocated on demand
    std::map<std::string, double*> bufs;
    std::map<std::string, size_t> buf_bytes;
    // Stokes
    PlStokesOp sop{}; bool sop_ready = false;
    double visc_contrast = 1.0;  // max / min of the viscosity fields of the current operator (pl_stokes_solve_device: beyond 1e6 the
                                 // row-scaled residual alone no longer vouches for the solution, see there)
    // Heat
    PlHeatOp hop{}; bool hop_ready = false; double heat_bcvalue[4] = {0, 0, 0, 0};
    int mic_search = 0;          // host-API gathers locate cells by per-axis search (pl_mic_set_search)
    std::vector<double> zmp, xmp;
    // multi-GPU: Pz x Px blocks of the node grid; rank = pz * Px + px owns node rows [gi0, gi0 + lnz) and columns
    // [gj0, gj0 + lnx) of geom.d (the last block of an axis also owns the last node row / column)
    int rank = 0, nranks = 1, Pz = 1, Px = 1, pz = 0, px = 0;
    long long comm_calls[4] = {0, 0, 0, 0};     // neighbour exchanges, all-gathers, device all-reduces, host all-reduces (pl_comm_stats)
    // time spent in them (pl_comm_times): device time between HIP events recorded around every exchange / all-gather / device
    // all-reduce on the context stream (a pool of event pairs, resolved when the figures are read), host wall time of the host all-reduces
    double comm_ms[4] = {0, 0, 0, 0};
    std::vector<hipEvent_t> comm_ev;            // pairs (start, stop)
    std::vector<int> comm_ev_kind;
    size_t comm_ev_used = 0;
    pl_comm_ops comm{};
    void* nccl = nullptr;     // pl_comm.hip: native RCCL transport (optional)
    void* local = nullptr;    // pl_comm.hip: in-process group of virtual ranks (pl_local_group_*)
    // opaque extension slots owned by other translation units
    void* krylov = nullptr;   // pl_solver.hip
    void* mic = nullptr;      // pl_mic.hip
    void* step = nullptr;     // pl_step.hip
    void* direct = nullptr;   // pl_direct.hip: banded LU of the last direct fallback
};

// ---- node-kernel launch shape -------------------------------------------------------------------
// Blocks are 64 x 4 threads; a workgroup marches over `iters` groups of 4 rows (iters = 8 on large
// planes, 1 on small ones).  Vertically adjacent 64x4 blocks are dealt to DIFFERENT XCDs (no shared
// L2), so with iters = 1 every block re-fetches its halo rows from HBM: measured 1.6x read
// amplification on k_stokes_apply (profiles/r01_pmc_v1_summary.csv); 32-row blocks cut it to 2/32.
inline int pl_row_iters(const PlGeom& g) { (void)g; return 1; }   // 8 (32-row blocks) measured SLOWER: apply 83 -> 95 us; the re-reads hit the Infinity Cache
inline dim3 pl_grid_rows(const PlGeom& g) {
    const int it = pl_row_iters(g);
    return dim3((g.lnx + 63) / 64, (g.lnz + 4 * it - 1) / (4 * it));
}
// Reciprocal without the ~15-instruction IEEE division sequence.  Measured on gfx950 over 1e-30..1e30:
// v_rcp_f64 alone is good to 4.6e-8, one Newton step to 2.2e-15 (used here), two steps are exact.  (The bare
// instruction inside the smoother saved nothing measurable and moved BiCGStab's iteration counts.)
__device__ inline double pl_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}

// single precision (the FP32 multigrid levels): v_rcp_f32 is good to 1 ulp
__device__ inline float pl_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- two-columns-per-lane row loads (see k_stokes_apply_v2) ----
template <typename T> struct PlVec2;
template <> struct PlVec2<double> { typedef double2 type; static __host__ __device__ inline double2 make(double a, double b) { return make_double2(a, b); } };
template <> struct PlVec2<float> { typedef float2 type; static __host__ __device__ inline float2 make(float a, float b) { return make_float2(a, b); } };

template <typename T> struct Row2T { typename PlVec2<T>::type v; T w, e; };   // w = value at j-1 of .x ; e = value at j+1 of .y
typedef Row2T<double> Row2;

template <typename T>
__device__ inline Row2T<T> load_row2(const T* __restrict__ row, int lj0, bool active, bool need_w, bool need_e,
                                     int lane, bool has_right) {
    typedef typename PlVec2<T>::type V2;
    Row2T<T> r;
    r.v = active ? *reinterpret_cast<const V2*>(row + lj0) : PlVec2<T>::make(T(0), T(0));
    r.w = T(0); r.e = T(0);
    if (need_w) {
        const T up = __shfl_up(r.v.y, 1, 64);
        r.w = (lane == 0) ? (active ? row[lj0 - 1] : T(0)) : up;
    }
    if (need_e) {
        const T dn = __shfl_down(r.v.x, 1, 64);
        r.e = (lane == 63 || !has_right) ? (active ? row[lj0 + 2] : T(0)) : dn;
    }
    return r;
}

// the same for a row stored as TS, returned in the arithmetic type T (level 0 of the multigrid keeps three of its vectors in FP32)
template <typename T, typename TS>
__device__ inline Row2T<T> load_row2_as(const TS* __restrict__ row, int lj0, bool active, bool need_w, bool need_e,
                                        int lane, bool has_right) {
    const Row2T<TS> s = load_row2(row, lj0, active, need_w, need_e, lane, has_right);
    Row2T<T> r;
    r.v = PlVec2<T>::make((T)s.v.x, (T)s.v.y); r.w = (T)s.w; r.e = (T)s.e;
    return r;
}

#define PL_ROW_LOOP(g, iters)                                                           \
    const int lj = blockIdx.x * 64 + threadIdx.x;                                       \
    if (lj >= (g).lnx) return;                                                          \
    for (int it_ = 0, li = blockIdx.y * 4 * (iters) + threadIdx.y; it_ < (iters) && li < (g).lnz; it_++, li += 4)

// ---- error helpers ----------------------------------------------------------------------
extern thread_local std::string pl_tls_error;
int pl_fail(pl_ctx* ctx, const std::string& msg);
#define PL_HIP(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return pl_fail(ctx, std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)
#define PL_TRY(expr)                                                                        \
    do {                                                                                    \
        int rc_ = (expr);                                                                   \
        if (rc_) return rc_;                                                                \
    } while (0)

// ---- shared helpers (pl_ctx.hip) --------------------------------------------------------
int pl_buf(pl_ctx* ctx, const char* name, size_t bytes, double** out, bool zero = true);
int pl_stage(pl_ctx* ctx, size_t bytes);
int pl_geom_build(pl_ctx* ctx, PlGeomHost& gh, int nz, int nx, const double* zc, const double* xc);
// restrict the geometry to the block rows [gi0, gi0+lnz) x columns [gj0, gj0+lnx)
void pl_geom_set_block(PlGeomHost& gh, int gi0, int lnz, int gj0, int lnx);
inline bool pl_geom_is_dist(const PlGeom& g) { return g.lnz != g.nz || g.lnx != g.nx; }
// block of rank (pz, px) on a grid of n nodes along one axis split into P parts: first node, node count
inline void pl_block_1d(int n, int P, int p, int* first, int* count) {
    const int C = (n - 1) / P;
    *first = p * C; *count = (p == P - 1) ? C + 1 : C;
}
// Halo exchange with the (up to 8) neighbour blocks, `depth` <= PL_RING nodes deep, of nplanes planes (no-op on one
// rank).  add: reverse (accumulating) halo - the ring contributions are ADDED to the neighbours' owned boundary nodes.
int pl_halo(pl_ctx* ctx, const PlGeom& g, double* planes, int nplanes, long long plane_stride, int depth = 1, bool add = false);
int pl_halo(pl_ctx* ctx, const PlGeom& g, float* planes, int nplanes, long long plane_stride, int depth = 1, bool add = false);   // FP32 multigrid levels
// the same for any dense 2-D layout: `origin` points at the owned node (0,0) of plane 0
int pl_halo_generic(pl_ctx* ctx, int lnz, int lnx, double* origin, long long pitch, int nplanes, long long plane_stride,
                    int depth, bool add);
// Replicated planes (lnz = nz, lnx = nx on every rank) of which each rank has computed its own block: gather all blocks
int pl_gather_blocks(pl_ctx* ctx, const PlGeom& grepl, double* planes, int nplanes, long long plane_stride);
int pl_allreduce_host(pl_ctx* ctx, double* buf, long long n, int op);
// point-to-point messages between ranks (device buffers), matched per peer in list order; stream-ordered on the native
// RCCL transport, otherwise through the in-process group or the host callback table (pl_comm.hip)
struct PlMsg { int peer; const double* send; long long nsend; double* recv; long long nrecv; };
int pl_comm_sendrecv(pl_ctx* ctx, const PlMsg* msgs, int nmsg);
int pl_comm_allgather(pl_ctx* ctx, const double* send, double* recv, long long count);
int pl_comm_native_init(pl_ctx* ctx);
void pl_local_detach(pl_ctx* ctx);
void pl_comm_native_free(pl_ctx* ctx);
int pl_comm_native_enabled(pl_ctx* ctx);
int pl_comm_allreduce_dev(pl_ctx* ctx, double* dev, int n, int op = 0);       // op: 0 sum, 1 min, 2 max
// event pair around a communication call on the context stream (kind 0 exchange, 1 all-gather, 2 device all-reduce); -1 when the pool is full
int pl_comm_time_begin(pl_ctx* ctx, int kind);
void pl_comm_time_end(pl_ctx* ctx, int slot);
void pl_geom_free(PlGeomHost& gh);
// host (nz,nx) C-order  <->  device plane with ring/pitch
int pl_plane_upload(pl_ctx* ctx, const PlGeom& g, const double* host, double* dplane);
int pl_plane_download(pl_ctx* ctx, const PlGeom& g, const double* dplane, double* host);
// reference-interleaved host vector (3N) <-> 3 device planes
int pl_vec3_upload(pl_ctx* ctx, const PlGeom& g, const double* host, double* dvec);
int pl_vec3_download(pl_ctx* ctx, const PlGeom& g, const double* dvec, double* host);

// ---- kernels launched from several units -------------------------------------------------
void pl_launch_stokes_apply(pl_ctx* ctx, const PlStokesOp& op, const double* x, double* y, const double* add = nullptr,
                            const double* coef = nullptr);
void pl_launch_stokes_rhs(pl_ctx* ctx, const PlStokesOp& op, double* rhs);
void pl_launch_heat_apply(pl_ctx* ctx, const PlHeatOp& op, const double* x, double* y, bool scaled = false);

// ---- device-resident entry points shared between units -----------------------------------
void pl_launch_heat_rhs(pl_ctx* ctx, const PlHeatOp& op, const double* Told, const double* H, double* rhs);
void pl_launch_heat_coef(pl_ctx* ctx, const PlGeom& g, const double* rho, const double* cp, double dt, double* c);
int  pl_heat_tables(pl_ctx* ctx, const double* zmp, const double* xmp);
int  pl_heat_check_bc(pl_ctx* ctx, const int bc[4]);
int  pl_stokes_check_bc(pl_ctx* ctx, const int bc[4]);
void pl_stokes_scaling_host(const PlGeomHost& gh, double minetas, double minetan, double* Kc, double* Kb);
void pl_stokes_fill_op(pl_ctx* ctx, double* etas, double* etan, double* rho, const int bc[4], int surfstab,
                       double tstep, double theta, double Kc, double Kb);
void pl_stokes_deflation(pl_ctx* ctx, bool persistent);     // consecutive solves of one model: keep the deflation vector
int  pl_stokes_solve_device(pl_ctx* ctx, const double* b_dev, bool use_x0, double rtol, int maxit,
                            pl_solve_stats* st);
double* pl_stokes_solution_device(pl_ctx* ctx);
double* pl_stokes_rhs_buffer_device(pl_ctx* ctx);
int  pl_heat_solve_device(pl_ctx* ctx, const double* b_dev, double rtol, int maxit, pl_solve_stats* st,
                          double** x_out, const double* x0_dev = nullptr);

// direct fallback for small systems (pl_direct.hip)
bool pl_direct_possible(pl_ctx* ctx);
int  pl_direct_factor(pl_ctx* ctx, const PlStokesOp& op_scaled);
int  pl_direct_solve(pl_ctx* ctx, const double* in, double* out);
void pl_direct_free(pl_ctx* ctx);

// krylov / MIC / step teardown hooks
void pl_solver_free(pl_ctx* ctx);
void pl_mic_free(pl_ctx* ctx);
void pl_step_free(pl_ctx* ctx);
