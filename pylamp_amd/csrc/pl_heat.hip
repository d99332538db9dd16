// Matrix-free backward-Euler heat operator  y = A T  and right-hand side.
// Replaces pylamp_diff.makeDiffusionMatrix (pylamp_diff.py:85-183).
// Bound: HBM; algorithmic traffic 40 B/node/apply with dt/(rho*Cp) pre-multiplied
// (T 8 + kz 8 + kx 8 + c 8 + y 8), 48 B/node counted the reference way.
#include "pl_internal.h"
#include <algorithm>

#define TB(tab, k) (tab)[(k) + PL_TOFF]

// SCALED: y = D^-1 A T (the Jacobi-scaled operator the heat solver iterates on), diagonal from the values
// already loaded.
template <bool SCALED>
__global__ __launch_bounds__(256) void k_heat_apply(PlHeatOp op, const double* __restrict__ T,
                                                    double* __restrict__ y) {
    const PlGeom& g = op.g;
    const int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y;
    if (lj >= g.lnx || li >= g.lnz) return;
    const int i = g.gi0 + li, j = g.gj0 + lj, nz = g.nz, nx = g.nx, p = g.pitch;
    const long long c = pl_idx(g, li, lj);
    const double t = T[c];
    double r, dg = 1.0;
    if (i == 0) {                                   // z = 0 owns its corners (pylamp_diff.py:99-110)
        if (op.bc[0] == PL_BC_FIXTEMP) r = t;
        else { const double k = op.kz[c] * TB(g.rdz, 0); r = k * (T[c + p] - t); dg = -k; }
    } else if (i == nz - 1) {                       // z = Lz (pylamp_diff.py:112-124)
        if (op.bc[2] == PL_BC_FIXTEMP) r = t;
        else { const double k = op.kz[c - p] * TB(g.rdz, nz - 2); r = k * (t - T[c - p]); dg = k; }
    } else if (j == 0) {                            // x = 0 (pylamp_diff.py:126-138)
        if (op.bc[1] == PL_BC_FIXTEMP) r = t;
        else { const double k = op.kx[c] * TB(g.rdx, 0); r = k * (T[c + 1] - t); dg = -k; }
    } else if (j == nx - 1) {                       // x = Lx (pylamp_diff.py:140-152)
        if (op.bc[3] == PL_BC_FIXTEMP) r = t;
        else { const double k = op.kx[c - 1] * TB(g.rdx, nx - 2); r = k * (t - T[c - 1]); dg = k; }
    } else {                                        // interior (pylamp_diff.py:157-177)
        const double ke = op.kx[c] * TB(g.rdx, j), kw = op.kx[c - 1] * TB(g.rdx, j - 1);
        const double kn = op.kz[c] * TB(g.rdz, i), ks = op.kz[c - p] * TB(g.rdz, i - 1);
        const double fx = (ke * (T[c + 1] - t) - kw * (t - T[c - 1])) * TB(op.rdxb, j);
        const double fz = (kn * (T[c + p] - t) - ks * (t - T[c - p])) * TB(op.rdzb, i);
        const double cc = op.rhocp_inv_dt[c];
        r = cc * (fx + fz) - t;
        if (SCALED) dg = -cc * ((ke + kw) * TB(op.rdxb, j) + (kn + ks) * TB(op.rdzb, i)) - 1.0;
    }
    y[c] = SCALED ? r / dg : r;
}

__global__ __launch_bounds__(256) void k_heat_rhs(PlHeatOp op, const double* __restrict__ Told,
                                                  const double* __restrict__ H, double bz0, double bx0,
                                                  double bzL, double bxL, double* __restrict__ rhs) {
    const PlGeom& g = op.g;
    const int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y;
    if (lj >= g.lnx || li >= g.lnz) return;
    const int i = g.gi0 + li, j = g.gj0 + lj;
    const long long c = pl_idx(g, li, lj);
    double r;
    if (i == 0) r = bz0;
    else if (i == g.nz - 1) r = bzL;
    else if (j == 0) r = bx0;
    else if (j == g.nx - 1) r = bxL;
    else r = -Told[c] - op.rhocp_inv_dt[c] * H[c];   // pylamp_diff.py:179
    rhs[c] = r;
}

// c = dt / (rho * Cp)
// (computed on the block AND its halo ring, clipped at the domain boundary: rho and cp arrive with their halo filled, and the
//  Chebyshev heat solver of several ranks sweeps on the block extended into the ring)
__global__ __launch_bounds__(256) void k_heat_coef(PlGeom g, const double* __restrict__ rho,
                                                   const double* __restrict__ cp, double dt,
                                                   double* __restrict__ c_out, int r0, int r1, int c0, int c1) {
    const int lj = c0 + blockIdx.x * 64 + threadIdx.x, li = r0 + blockIdx.y * 4 + threadIdx.y;
    if (lj >= c1 || li >= r1) return;
    const long long c = pl_idx(g, li, lj);
    c_out[c] = dt / (rho[c] * cp[c]);
}

static dim3 grid2d(const PlGeom& g) { return dim3((g.lnx + 63) / 64, (g.lnz + 3) / 4); }

void pl_launch_heat_apply(pl_ctx* ctx, const PlHeatOp& op, const double* x, double* y, bool scaled) {
    if (scaled) hipLaunchKernelGGL(k_heat_apply<true>, grid2d(op.g), dim3(64, 4), 0, ctx->stream, op, x, y);
    else hipLaunchKernelGGL(k_heat_apply<false>, grid2d(op.g), dim3(64, 4), 0, ctx->stream, op, x, y);
}

void pl_launch_heat_rhs(pl_ctx* ctx, const PlHeatOp& op, const double* Told, const double* H, double* rhs) {
    hipLaunchKernelGGL(k_heat_rhs, grid2d(op.g), dim3(64, 4), 0, ctx->stream, op, Told, H, ctx->heat_bcvalue[0],
                       ctx->heat_bcvalue[1], ctx->heat_bcvalue[2], ctx->heat_bcvalue[3], rhs);
}

void pl_launch_heat_coef(pl_ctx* ctx, const PlGeom& g, const double* rho, const double* cp, double dt, double* c) {
    const int r0 = -std::min(PL_RING, g.gi0), r1 = g.lnz + std::min(PL_RING, g.nz - (g.gi0 + g.lnz));
    const int c0 = -std::min(PL_RING, g.gj0), c1 = g.lnx + std::min(PL_RING, g.nx - (g.gj0 + g.lnx));
    hipLaunchKernelGGL(k_heat_coef, dim3((c1 - c0 + 63) / 64, (r1 - r0 + 3) / 4), dim3(64, 4), 0, ctx->stream, g, rho, cp, dt, c, r0, r1, c0, c1);
}

// midpoint tables 1/(zm[i]-zm[i-1]) (pylamp_diff.py:167-170), indexed global + 1
int pl_heat_tables(pl_ctx* ctx, const double* zmp, const double* xmp) {
    int nz = ctx->nz, nx = ctx->nx;
    // the time-step loop passes the same midpoints every step: keep the uploaded tables
    if (ctx->hop.rdzb && (int)ctx->zmp.size() == nz && (int)ctx->xmp.size() == nx &&
        std::equal(zmp, zmp + nz, ctx->zmp.begin()) && std::equal(xmp, xmp + nx, ctx->xmp.begin())) {
        double* d0;
        PL_TRY(pl_buf(ctx, "heat_tables", ((size_t)nz + 2 * PL_TOFF + 2 + nx + 2 * PL_TOFF + 2) * sizeof(double), &d0));
        if (d0 == ctx->hop.rdzb) return 0;
    }
    ctx->zmp.assign(zmp, zmp + nz); ctx->xmp.assign(xmp, xmp + nx);
    const size_t lz = (size_t)nz + 2 * PL_TOFF + 2;
    std::vector<double> t(lz + nx + 2 * PL_TOFF + 2, 0.0);
    for (int i = 1; i < nz; i++) t[i + PL_TOFF] = 1.0 / (zmp[i] - zmp[i - 1]);
    for (int j = 1; j < nx; j++) t[lz + j + PL_TOFF] = 1.0 / (xmp[j] - xmp[j - 1]);
    double* d;
    PL_TRY(pl_buf(ctx, "heat_tables", t.size() * sizeof(double), &d));
    PL_HIP(ctx, hipMemcpyAsync(d, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->hop.rdzb = d; ctx->hop.rdxb = d + lz;
    return 0;
}

int pl_heat_check_bc(pl_ctx* ctx, const int bc[4]) {
    for (int w = 0; w < 4; w++)
        if (bc[w] != PL_BC_FIXTEMP && bc[w] != PL_BC_FIXFLOW)
            return pl_fail(ctx, "heat: boundary condition must be FIXTEMP or FIXFLOW");
    return 0;
}

extern "C" int pl_heat_set_coeffs(pl_ctx* ctx, const double* zmp, const double* xmp, const double* T,
                                  const double* kz, const double* kx, const double* cp, const double* rho,
                                  const double* H, const int bc[4], const double bcvalue[4], double tstep) {
    if (!zmp || !xmp || !T || !kz || !kx || !cp || !rho || !H || !bc || !bcvalue)
        return pl_fail(ctx, "pl_heat_set_coeffs: NULL argument");
    PL_TRY(pl_heat_check_bc(ctx, bc));
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    size_t pb = (size_t)g.plane * sizeof(double);
    double *d_T, *d_kz, *d_kx, *d_cp, *d_rho, *d_H, *d_c;
    PL_TRY(pl_buf(ctx, "f_T", pb, &d_T)); PL_TRY(pl_buf(ctx, "kz", pb, &d_kz)); PL_TRY(pl_buf(ctx, "kx", pb, &d_kx));
    PL_TRY(pl_buf(ctx, "cp", pb, &d_cp)); PL_TRY(pl_buf(ctx, "heat_rho", pb, &d_rho)); PL_TRY(pl_buf(ctx, "H", pb, &d_H));
    PL_TRY(pl_buf(ctx, "heat_c", pb, &d_c));
    PL_TRY(pl_plane_upload(ctx, g, T, d_T)); PL_TRY(pl_plane_upload(ctx, g, kz, d_kz));
    PL_TRY(pl_plane_upload(ctx, g, kx, d_kx)); PL_TRY(pl_plane_upload(ctx, g, cp, d_cp));
    PL_TRY(pl_plane_upload(ctx, g, rho, d_rho)); PL_TRY(pl_plane_upload(ctx, g, H, d_H));
    PL_TRY(pl_heat_tables(ctx, zmp, xmp));
    PlHeatOp& op = ctx->hop;
    op.g = g; op.kz = d_kz; op.kx = d_kx; op.rhocp_inv_dt = d_c; op.dt = tstep;
    for (int w = 0; w < 4; w++) { op.bc[w] = bc[w]; ctx->heat_bcvalue[w] = bcvalue[w]; }
    pl_launch_heat_coef(ctx, g, d_rho, d_cp, tstep, d_c);
    PL_HIP(ctx, hipGetLastError());
    ctx->hop_ready = true;
    return 0;
}

extern "C" int pl_heat_apply(pl_ctx* ctx, const double* x, double* y) {
    if (!ctx->hop_ready) return pl_fail(ctx, "heat operator not set");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    size_t pb = (size_t)g.plane * sizeof(double);
    double *dx, *dy;
    PL_TRY(pl_buf(ctx, "api_hx", pb, &dx)); PL_TRY(pl_buf(ctx, "api_hy", pb, &dy));
    PL_TRY(pl_plane_upload(ctx, g, x, dx));
    pl_launch_heat_apply(ctx, ctx->hop, dx, dy);
    PL_HIP(ctx, hipGetLastError());
    PL_TRY(pl_plane_download(ctx, g, dy, y));
    return 0;
}

extern "C" int pl_heat_rhs(pl_ctx* ctx, double* rhs) {
    if (!ctx->hop_ready) return pl_fail(ctx, "heat operator not set");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    double* dy;
    PL_TRY(pl_buf(ctx, "api_hy", (size_t)g.plane * sizeof(double), &dy));
    pl_launch_heat_rhs(ctx, ctx->hop, ctx->bufs["f_T"], ctx->bufs["H"], dy);
    PL_HIP(ctx, hipGetLastError());
    PL_TRY(pl_plane_download(ctx, g, dy, rhs));
    return 0;
}

extern "C" int pl_heat_apply_bench(pl_ctx* ctx, int reps, double* avg_ms) {
    if (!ctx->hop_ready) return pl_fail(ctx, "heat operator not set");
    if (reps < 1) reps = 1;
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    size_t pb = (size_t)g.plane * sizeof(double);
    double *dx, *dy;
    PL_TRY(pl_buf(ctx, "api_hx", pb, &dx)); PL_TRY(pl_buf(ctx, "api_hy", pb, &dy));
    pl_launch_heat_apply(ctx, ctx->hop, dx, dy);
    PL_TRY(pl_timer_start(ctx));
    for (int r = 0; r < reps; r++) pl_launch_heat_apply(ctx, ctx->hop, dx, dy);
    double ms = 0;
    PL_TRY(pl_timer_stop_ms(ctx, &ms));
    PL_HIP(ctx, hipGetLastError());
    if (avg_ms) *avg_ms = ms / reps;
    return 0;
}
