// Matrix-free staggered-grid Stokes operator  y = A x  and its right-hand side.
//
// Replaces the lil_matrix assembly of pylamp_stokes.makeStokesMatrix
// (pylamp_stokes.py:104-563).  Rows are classified from the GLOBAL node index inside the
// kernel (ghost / wall / tangential-slave / corner / anchor / interior); nothing per-row
// is stored.  Unknowns live in three SoA planes (vz | vx | P) with a one-node ring.
//
// Bound: HBM.  Algorithmic traffic per node and apply: x 24 B + etas 8 + etan 8 + y 24
// = 64 B.
#include "pl_internal.h"
#include <cmath>
#include <cstdlib>
#include <limits>

#define TB(tab, k) (tab)[(k) + 1]

// One thread per node, 64 x 4 thread blocks: each wave owns 64 consecutive columns of ONE
// row, so every z-table value is wave-uniform (scalar loads) and all plane accesses are
// 512-B contiguous per wave-instruction; the +-1 column / +-pitch row neighbours are
// re-reads of lines the same or the adjacent wave just touched (L1/L2 hits).
__device__ inline void stokes_apply_node(const PlStokesOp& op, const double* __restrict__ x, double* __restrict__ y,
                                         int li, int lj) {
    const PlGeom& g = op.g;
    const int i = g.gi0 + li, j = g.gj0 + lj;
    const int nz = g.nz, nx = g.nx;
    const int p = g.pitch;
    // 32-bit element offsets from wave-uniform base pointers: lets the compiler use the
    // SGPR-base + VGPR-offset addressing mode instead of a 64-bit address per neighbour
    const int c = (li + 1) * p + (lj + PL_PADL);
    const double* __restrict__ vz = x;
    const double* __restrict__ vx = x + g.plane;
    const double* __restrict__ P = x + 2 * g.plane;
    const double* __restrict__ es = op.etas;
    const double* __restrict__ en = op.etan;
    const double Kc = op.Kc;
    const double vz_c = vz[c], vx_c = vx[c], p_c = P[c];

    // row scale factors D_r (used only when op.scaled): 1/(sum of the 4 own-component
    // coefficients) on interior momentum rows, 1/Kc on constraint rows, 1/(Kc (1/dx+1/dz)) on
    // continuity rows, 1/Kb on corner rows.  No extra memory traffic.
    const double iKc = 1.0 / Kc;
    double sz = iKc, sx = iKc, sp = iKc;

    // ---------------- vz row (z-momentum) ----------------
    double yz;
    if (j == nx - 1 || i == 0 || i == nz - 1) {
        yz = Kc * vz_c;                                   // ghost column / no flow through z-walls
    } else if (j == 0) {
        yz = Kc * (vz_c - vz[c + 1]);                     // free slip at x = 0  (pylamp_stokes.py:249-255)
    } else if (j == nx - 2) {
        yz = Kc * (vz_c - vz[c - 1]);                     // free slip at x = Lx (pylamp_stokes.py:296-301)
    } else {
        const double rdz_i = TB(g.rdz, i), rdz_m = TB(g.rdz, i - 1), rDz_i = TB(g.rDz, i);
        const double rdx_j = TB(g.rdx, j), rDx_j = TB(g.rDx, j), rDx_p = TB(g.rDx, j + 1);
        const double esC = es[c], esE = es[c + 1];
        const double cN = 4.0 * en[c] * rdz_i * rDz_i;
        const double cS = 4.0 * en[c - p] * rdz_m * rDz_i;
        const double cE = 2.0 * esE * rDx_p * rdx_j;
        const double cW = 2.0 * esC * rDx_j * rdx_j;
        const double xE = 2.0 * esE * rDz_i * rdx_j;
        const double xW = 2.0 * esC * rDz_i * rdx_j;
        yz = cN * (vz[c + p] - vz_c) - cS * (vz_c - vz[c - p]) + cE * (vz[c + 1] - vz_c) -
             cW * (vz_c - vz[c - 1]) + xE * (vx[c + 1] - vx[c - p + 1]) - xW * (vx_c - vx[c - p]) -
             2.0 * Kc * rDz_i * (p_c - P[c - p]);
        sz = 1.0 / (cN + cS + cE + cW);
        if (op.surfstab) {
            const double* __restrict__ r = op.rho;
            yz += op.ss * op.gz * 0.5 *
                  ((r[c + 1] + r[c + p + 1] - r[c - 1] - r[c + p - 1]) * rDx_j * vx_c +
                   (r[c + p] + r[c + p + 1] - r[c - p] - r[c - p + 1]) * rDz_i * vz_c);
        }
    }
    y[c] = op.scaled ? yz * sz : yz;            // each row is stored as soon as it is complete;
    __builtin_amdgcn_sched_barrier(0);          // the fence keeps the next row's loads from being hoisted
                                                // above it (110 -> fewer VGPRs, more waves per SIMD)

    // ---------------- vx row (x-momentum) ----------------
    double yx;
    if (i == nz - 1 || j == 0 || j == nx - 1) {
        yx = Kc * vx_c;                                   // ghost row / no flow through x-walls
    } else if (i == 0) {
        if (op.bc_z0 == PL_BC_FREESLIP) yx = Kc * (vx_c - vx[c + p]);
        else yx = Kc * ((-TB(g.rDz, 1) - TB(g.rdz, 0)) * vx_c + TB(g.rDz, 1) * vx[c + p]);
    } else if (i == nz - 2) {
        if (op.bc_zL == PL_BC_FREESLIP) yx = Kc * (vx_c - vx[c - p]);
        else yx = Kc * ((TB(g.rDz, nz - 2) + TB(g.rdz, nz - 2)) * vx_c - TB(g.rDz, nz - 2) * vx[c - p]);
    } else {
        const double rdx_j = TB(g.rdx, j), rdx_m = TB(g.rdx, j - 1), rDx_j = TB(g.rDx, j);
        const double rdz_i = TB(g.rdz, i), rDz_i = TB(g.rDz, i), rDz_p = TB(g.rDz, i + 1);
        const double esC = es[c], esN = es[c + p];
        const double cE = 4.0 * en[c] * rdx_j * rDx_j;
        const double cW = 4.0 * en[c - 1] * rdx_m * rDx_j;
        const double cN = 2.0 * esN * rDz_p * rdz_i;
        const double cS = 2.0 * esC * rDz_i * rdz_i;
        const double zN = 2.0 * esN * rDx_j * rdz_i;
        const double zS = 2.0 * esC * rDx_j * rdz_i;
        yx = cE * (vx[c + 1] - vx_c) - cW * (vx_c - vx[c - 1]) + cN * (vx[c + p] - vx_c) -
             cS * (vx_c - vx[c - p]) + zN * (vz[c + p] - vz[c + p - 1]) - zS * (vz_c - vz[c - 1]) -
             2.0 * Kc * rDx_j * (p_c - P[c - 1]);
        sx = 1.0 / (cE + cW + cN + cS);
        if (op.surfstab && op.gx != 0.0) {
            const double* __restrict__ r = op.rho;
            yx += op.ss * op.gx * 0.5 *
                  ((r[c + 1] + r[c + p + 1] - r[c - 1] - r[c + p - 1]) * rDx_j * vx_c +
                   (r[c + p] + r[c + p + 1] - r[c - p] - r[c - p + 1]) * rDz_i * vz_c);
        }
    }
    y[c + g.plane] = op.scaled ? yx * sx : yx;
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- P row (continuity) ----------------
    double yp;
    if (i == nz - 1 || j == nx - 1 || (i == op.anchor_i && j == op.anchor_j)) {
        yp = Kc * p_c;                                    // ghosts, pressure anchor
    } else if ((i == 0 || i == nz - 2) && j == 0) {
        yp = op.Kb * (P[c + 1] - p_c);                    // corner symmetry (pylamp_stokes.py:358-369)
        sp = 1.0 / op.Kb;
    } else if ((i == 0 || i == nz - 2) && j == nx - 2) {
        yp = op.Kb * (P[c - 1] - p_c);
        sp = 1.0 / op.Kb;
    } else {
        yp = Kc * ((vx[c + 1] - vx_c) * TB(g.rdx, j) + (vz[c + p] - vz_c) * TB(g.rdz, i));
        sp = 1.0 / (Kc * (TB(g.rdx, j) + TB(g.rdz, i)));
    }
    y[c + 2 * g.plane] = op.scaled ? yp * sp : yp;
}

#ifndef PL_APPLY_WAVES
#define PL_APPLY_WAVES 4
#endif
__global__ __launch_bounds__(256, PL_APPLY_WAVES) void k_stokes_apply(PlStokesOp op, const double* __restrict__ x,
                                                                      double* __restrict__ y, int iters) {
    PL_ROW_LOOP(op.g, iters) stokes_apply_node(op, x, y, li, lj);
}

// XCD-aware variant (1-D grid).  The dispatcher deals consecutive workgroup ids round-robin over
// the 8 XCDs (MI355X_MICROARCH.md, observed; used for speed only): id % 8 selects the XCD.  Block
// column bx is served by XCD bx % 8, and inside an XCD the blocks walk row-major, so vertically
// adjacent blocks - which share a halo row of all five input planes - share that XCD's L2.
__global__ __launch_bounds__(256, PL_APPLY_WAVES) void k_stokes_apply_xcd(PlStokesOp op, const double* __restrict__ x,
                                                                          double* __restrict__ y, int gx8) {
    const int id = blockIdx.x;
    const int xcd = id & 7, k = id >> 3;
    const int by = k / gx8, bx = xcd + 8 * (k % gx8);
    const int lj = bx * 64 + threadIdx.x, li = by * 4 + threadIdx.y;
    if (lj >= op.g.lnx || li >= op.g.lnz) return;
    stokes_apply_node(op, x, y, li, lj);
}

__global__ __launch_bounds__(256) void k_stokes_rhs(PlStokesOp op, double* __restrict__ rhs) {
    const PlGeom& g = op.g;
    const int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y;
    if (lj >= g.lnx || li >= g.lnz) return;
    const int i = g.gi0 + li, j = g.gj0 + lj, nz = g.nz, nx = g.nx, p = g.pitch;
    const long long c = pl_idx(g, li, lj);
    const double* __restrict__ r = op.rho;
    double bz = 0.0, bx = 0.0;
    if (i >= 1 && i <= nz - 2 && j >= 1 && j <= nx - 3) bz = -0.5 * (r[c] + r[c + 1]) * op.gz;
    if (i >= 1 && i <= nz - 3 && j >= 1 && j <= nx - 2) bx = -0.5 * (r[c] + r[c + p]) * op.gx;
    rhs[c] = bz; rhs[c + g.plane] = bx; rhs[c + 2 * g.plane] = 0.0;
}

static dim3 grid2d(const PlGeom& g) { return dim3((g.lnx + 63) / 64, (g.lnz + 3) / 4); }

void pl_launch_stokes_apply(pl_ctx* ctx, const PlStokesOp& op, const double* x, double* y) {
    static const int use_xcd = getenv("PYLAMP_APPLY_XCD") ? atoi(getenv("PYLAMP_APPLY_XCD")) : 1;
    if (use_xcd && (long long)op.g.lnz * op.g.lnx >= 8000000LL) {   // measured: +5% at 4097^2, neutral at 2049^2
        const int gx = (op.g.lnx + 63) / 64, gy = (op.g.lnz + 3) / 4, gx8 = (gx + 7) / 8;
        hipLaunchKernelGGL(k_stokes_apply_xcd, dim3(8 * gx8 * gy), dim3(64, 4), 0, ctx->stream, op, x, y, gx8);
        return;
    }
    hipLaunchKernelGGL(k_stokes_apply, pl_grid_rows(op.g), dim3(64, 4), 0, ctx->stream, op, x, y, pl_row_iters(op.g));
}

void pl_launch_stokes_rhs(pl_ctx* ctx, const PlStokesOp& op, double* rhs) {
    hipLaunchKernelGGL(k_stokes_rhs, grid2d(op.g), dim3(64, 4), 0, ctx->stream, op, rhs);
}

// np.min semantics: NaN if any NaN
static double np_min(const double* a, size_t n) {
    double m = std::numeric_limits<double>::infinity();
    bool nan = false;
    for (size_t k = 0; k < n; k++) { if (a[k] != a[k]) nan = true; else if (a[k] < m) m = a[k]; }
    return nan ? std::numeric_limits<double>::quiet_NaN() : m;
}

// pylamp_stokes.py:116-122; python's min(a, b) returns a unless b < a
void pl_stokes_scaling_host(const PlGeomHost& gh, double minetas, double minetan, double* Kc, double* Kb) {
    double mineta = (minetan < minetas) ? minetan : minetas;
    int nz = (int)gh.zc.size(), nx = (int)gh.xc.size();
    double avgdx = (gh.xc[nx - 1] - gh.xc[0]) / nx;
    double avgdz = (gh.zc[nz - 1] - gh.zc[0]) / nz;
    *Kc = 2.0 * mineta / (avgdx + avgdz);
    *Kb = 4.0 * mineta / ((avgdx + avgdz) * (avgdx + avgdz));
}

int pl_stokes_check_bc(pl_ctx* ctx, const int bc[4]) {
    // Only the wall types that give a non-singular system in the reference are offered
    // (SURVEY.md 8 a2.3): z-walls NOSLIP or FREESLIP, x-walls FREESLIP.
    for (int w = 0; w < 4; w += 2)
        if (bc[w] != PL_BC_NOSLIP && bc[w] != PL_BC_FREESLIP)
            return pl_fail(ctx, "stokes: z-wall boundary condition must be NOSLIP or FREESLIP");
    for (int w = 1; w < 4; w += 2)
        if (bc[w] != PL_BC_FREESLIP)
            return pl_fail(ctx, "stokes: x-wall boundary condition must be FREESLIP");
    return 0;
}

void pl_stokes_fill_op(pl_ctx* ctx, double* etas, double* etan, double* rho, const int bc[4], int surfstab,
                       double tstep, double theta, double Kc, double Kb) {
    PlStokesOp& op = ctx->sop;
    op.g = ctx->geom.d;
    op.etas = etas; op.etan = etan; op.rho = rho;
    op.Kc = Kc; op.Kb = Kb;
    op.bc_z0 = bc[0]; op.bc_zL = bc[2];
    op.surfstab = surfstab ? 1 : 0; op.ss = theta * tstep;
    op.anchor_i = 3; op.anchor_j = 2;
    op.gz = 9.81; op.gx = 0.0;                     // pylamp_const.py:21
    op.scaled = 0;
    ctx->sop_ready = true;
}

extern "C" int pl_stokes_set_coeffs(pl_ctx* ctx, const double* etas, const double* etan, const double* rho,
                                    const int bc[4], int surfstab, double tstep, double theta) {
    if (!etas || !etan || !rho || !bc) return pl_fail(ctx, "pl_stokes_set_coeffs: NULL argument");
    PL_TRY(pl_stokes_check_bc(ctx, bc));
    if (surfstab && !(tstep == tstep)) return pl_fail(ctx, "surface stabilization needs predetermined tstep");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    size_t pb = (size_t)g.plane * sizeof(double);
    double *d_es, *d_en, *d_rho;
    PL_TRY(pl_buf(ctx, "etas", pb, &d_es)); PL_TRY(pl_buf(ctx, "etan", pb, &d_en)); PL_TRY(pl_buf(ctx, "rho", pb, &d_rho));
    PL_TRY(pl_plane_upload(ctx, g, etas, d_es));
    PL_TRY(pl_plane_upload(ctx, g, etan, d_en));
    PL_TRY(pl_plane_upload(ctx, g, rho, d_rho));
    size_t n = (size_t)ctx->nz * ctx->nx;
    double Kc, Kb;
    pl_stokes_scaling_host(ctx->geom, np_min(etas, n), np_min(etan, n), &Kc, &Kb);
    pl_stokes_fill_op(ctx, d_es, d_en, d_rho, bc, surfstab, tstep, theta, Kc, Kb);
    return 0;
}

extern "C" int pl_stokes_get_scaling(pl_ctx* ctx, double* kcont, double* kbond) {
    if (!ctx->sop_ready) return pl_fail(ctx, "stokes operator not set");
    if (kcont) *kcont = ctx->sop.Kc;
    if (kbond) *kbond = ctx->sop.Kb;
    return 0;
}

extern "C" int pl_stokes_apply(pl_ctx* ctx, const double* x, double* y) {
    if (!ctx->sop_ready) return pl_fail(ctx, "stokes operator not set");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    size_t vb = (size_t)3 * g.plane * sizeof(double);
    double *dx, *dy;
    PL_TRY(pl_buf(ctx, "api_x", vb, &dx)); PL_TRY(pl_buf(ctx, "api_y", vb, &dy));
    PL_TRY(pl_vec3_upload(ctx, g, x, dx));
    PL_TRY(pl_halo_rows(ctx, g, dx, 3, g.plane));
    pl_launch_stokes_apply(ctx, ctx->sop, dx, dy);
    PL_HIP(ctx, hipGetLastError());
    PL_TRY(pl_vec3_download(ctx, g, dy, y));
    return 0;
}

extern "C" int pl_stokes_rhs(pl_ctx* ctx, double* rhs) {
    if (!ctx->sop_ready) return pl_fail(ctx, "stokes operator not set");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    double* dy;
    PL_TRY(pl_buf(ctx, "api_y", (size_t)3 * g.plane * sizeof(double), &dy));
    pl_launch_stokes_rhs(ctx, ctx->sop, dy);
    PL_HIP(ctx, hipGetLastError());
    PL_TRY(pl_vec3_download(ctx, g, dy, rhs));
    return 0;
}

__global__ void k_fill_pseudo(double* __restrict__ v, long long n, unsigned seed) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(k * 2654435761u) ^ seed;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        v[k] = (double)h * (2.0 / 4294967296.0) - 1.0;
    }
}

extern "C" int pl_stokes_apply_bench(pl_ctx* ctx, int reps, double* avg_ms) {
    if (!ctx->sop_ready) return pl_fail(ctx, "stokes operator not set");
    if (reps < 1) reps = 1;
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    size_t vb = (size_t)3 * g.plane * sizeof(double);
    double *dx, *dy;
    PL_TRY(pl_buf(ctx, "api_x", vb, &dx)); PL_TRY(pl_buf(ctx, "api_y", vb, &dy));
    hipLaunchKernelGGL(k_fill_pseudo, dim3(2048), dim3(256), 0, ctx->stream, dx, 3 * g.plane, 12345u);
    pl_launch_stokes_apply(ctx, ctx->sop, dx, dy);      // warm-up
    PL_TRY(pl_timer_start(ctx));
    for (int r = 0; r < reps; r++) pl_launch_stokes_apply(ctx, ctx->sop, dx, dy);
    double ms = 0;
    PL_TRY(pl_timer_stop_ms(ctx, &ms));
    PL_HIP(ctx, hipGetLastError());
    if (avg_ms) *avg_ms = ms / reps;
    return 0;
}
