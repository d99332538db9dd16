// Direct fallback of the Stokes solve for SMALL systems on one GPU: banded LU with partial pivoting of the row-scaled
// operator, used as an (exact) preconditioner of the same BiCGStab when the multigrid-preconditioned iteration does not
// converge.  The case it exists for: the reference's free-surface stabilisation with the reference's own sign
// (pylamp_stokes.py:422-426,483-487) makes the velocity block indefinite at the Courant time step --
// no smoother-based preconditioner applies, while the reference itself solves with SuperLU (pylamp2.py:394).  Sizes are
// those the reference can run (41^2 ... 201 x 41: 5 043 ... 24 723 unknowns, band ~3 nx).
//
// The band is assembled from the matrix-free operator by 27-colour probing (every row reaches nodes within +-1 in i and j),
// in the reference's DOF order (node-major, (vz, vx, P) per node: bandwidth 3 (nx + 1) + 2).  Factorisation and the two
// triangular solves run in ONE workgroup each (column by column; the band is narrow, the work per column is a
// kl x (kl + ku) rank-1 update): tens of milliseconds at these sizes.
#include "pl_internal.h"
#include <algorithm>

struct PlDirect {
    int n = 0, kl = 0, ku = 0, ld = 0;
    double* ab = nullptr;        // LAPACK band storage with kl extra rows for the fill-in of pivoting: (i,j) at kl+ku+i-j + j*ld
    int* piv = nullptr;
    double* work = nullptr;      // n doubles
    double* probe = nullptr;     // 6 planes: indicator x, y = A x
    int* info = nullptr;
};

__global__ __launch_bounds__(256) void kd_indicator(PlGeom g, int ci, int cj, int q, double* __restrict__ x) {
    const int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y;
    if (lj >= g.lnx || li >= g.lnz) return;
    const long long c = pl_idx(g, li, lj);
    for (int p = 0; p < 3; p++) x[c + p * g.plane] = (p == q && li % 3 == ci && lj % 3 == cj) ? 1.0 : 0.0;
}
__global__ __launch_bounds__(256) void kd_scatter_band(PlGeom g, int ci, int cj, int q, const double* __restrict__ y, int kl, int ku, int ld,
                                                       double* __restrict__ ab) {
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (j >= g.lnx || i >= g.lnz) return;
    const long long c = pl_idx(g, i, j);
    const int pi = i + ((ci - i % 3 + 1 + 3) % 3 - 1), pj = j + ((cj - j % 3 + 1 + 3) % 3 - 1);      // the probed node within +-1
    if (pi < 0 || pi >= g.nz || pj < 0 || pj >= g.nx) return;
    const long long col = ((long long)pi * g.nx + pj) * 3 + q;
    for (int r = 0; r < 3; r++) {
        const double v = y[c + r * g.plane];
        if (v == 0.0) continue;
        const long long row = ((long long)i * g.nx + j) * 3 + r;
        ab[kl + ku + row - col + col * ld] = v;
    }
}
// unblocked banded LU with partial pivoting (the algorithm of LAPACK's dgbtf2), one workgroup
__global__ __launch_bounds__(1024) void kd_gbtrf(int n, int kl, int ku, int ld, double* __restrict__ ab, int* __restrict__ piv, int* __restrict__ info) {
    __shared__ double smax[1024]; __shared__ int sidx[1024]; __shared__ int sp; __shared__ double spiv;
    const int tid = threadIdx.x, kv = kl + ku;
    int ju = 0;                                            // last column touched by the row interchanges so far
    for (int k = 0; k < n; k++) {
        const int km = min(kl, n - 1 - k);
        // pivot search in column k, rows k .. k+km
        double best = -1.0; int bi = 0;
        for (int d = tid; d <= km; d += 1024) { const double a = fabs(ab[kv + d + (long long)k * ld]); if (a > best) { best = a; bi = d; } }
        smax[tid] = best; sidx[tid] = bi;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o && (smax[tid + o] > smax[tid] || (smax[tid + o] == smax[tid] && sidx[tid + o] < sidx[tid]))) { smax[tid] = smax[tid + o]; sidx[tid] = sidx[tid + o]; }
            __syncthreads();
        }
        if (tid == 0) { sp = sidx[0]; piv[k] = k + sidx[0]; if (!(smax[0] > 0.0) && *info == 0) *info = k + 1; }
        __syncthreads();
        const int p = sp;
        ju = max(ju, min(k + p + ku, n - 1));
        if (p != 0)                                        // swap rows k and k+p over columns k .. ju
            for (int j = k + tid; j <= ju; j += 1024) {
                const long long a = kv + k - j + (long long)j * ld, b = a + p;
                const double t = ab[a]; ab[a] = ab[b]; ab[b] = t;
            }
        __syncthreads();
        if (tid == 0) spiv = ab[kv + (long long)k * ld];
        __syncthreads();
        const double pv = spiv;
        if (pv != 0.0) {
            for (int d = 1 + tid; d <= km; d += 1024) ab[kv + d + (long long)k * ld] /= pv;       // multipliers
            __syncthreads();
            const int nc = ju - k;                         // trailing columns k+1 .. ju
            for (int t = tid; t < km * nc; t += 1024) {
                const int d = 1 + t % km, jj = k + 1 + t / km;
                ab[kv + k + d - jj + (long long)jj * ld] -= ab[kv + d + (long long)k * ld] * ab[kv + k - jj + (long long)jj * ld];
            }
        }
        __syncthreads();
    }
}
// x := U^-1 L^-1 P b (one right-hand side, in place), one workgroup
__global__ __launch_bounds__(1024) void kd_gbtrs(int n, int kl, int ku, int ld, const double* __restrict__ ab, const int* __restrict__ piv, double* __restrict__ b) {
    const int tid = threadIdx.x, kv = kl + ku;
    __shared__ double sb;
    for (int k = 0; k < n; k++) {                          // L y = P b
        if (tid == 0) { const int p = piv[k]; const double t = b[k]; b[k] = b[p]; b[p] = t; sb = b[k]; }
        __syncthreads();
        const double bk = sb;
        const int km = min(kl, n - 1 - k);
        for (int d = 1 + tid; d <= km; d += 1024) b[k + d] -= ab[kv + d + (long long)k * ld] * bk;
        __syncthreads();
    }
    for (int k = n - 1; k >= 0; k--) {                     // U x = y (U has kl + ku superdiagonals)
        if (tid == 0) { b[k] /= ab[kv + (long long)k * ld]; sb = b[k]; }
        __syncthreads();
        const double bk = sb;
        const int kmu = min(kv, k);
        for (int d = 1 + tid; d <= kmu; d += 1024) b[k - d] -= ab[kv - d + (long long)k * ld] * bk;
        __syncthreads();
    }
}
// 3 ring planes <-> interleaved node-major vector
__global__ __launch_bounds__(256) void kd_planes_to_vec(PlGeom g, const double* __restrict__ p, double* __restrict__ v) {
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (j >= g.lnx || i >= g.lnz) return;
    const long long c = pl_idx(g, i, j), r = ((long long)i * g.nx + j) * 3;
    v[r] = p[c]; v[r + 1] = p[c + g.plane]; v[r + 2] = p[c + 2 * g.plane];
}
__global__ __launch_bounds__(256) void kd_vec_to_planes(PlGeom g, const double* __restrict__ v, double* __restrict__ p) {
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (j >= g.lnx || i >= g.lnz) return;
    const long long c = pl_idx(g, i, j), r = ((long long)i * g.nx + j) * 3;
    p[c] = v[r]; p[c + g.plane] = v[r + 1]; p[c + 2 * g.plane] = v[r + 2];
}

static dim3 grid2d(const PlGeom& g) { return dim3((g.lnx + 63) / 64, (g.lnz + 3) / 4); }

void pl_direct_free(pl_ctx* ctx) {
    PlDirect* D = (PlDirect*)ctx->direct;
    if (!D) return;
    for (void* q : {(void*)D->ab, (void*)D->piv, (void*)D->work, (void*)D->probe, (void*)D->info}) if (q) (void)hipFree(q);
    delete D;
    ctx->direct = nullptr;
}

// can this context's Stokes system be factorised directly?  One rank, band storage below ~0.5 GB (up to ~130 x 130 nodes):
// factorisation and the triangular solves run column by column in ONE workgroup -- tens of milliseconds at 25 000 unknowns,
// about a second at the limit; beyond it the iteration's answer stands with converged = 0
bool pl_direct_possible(pl_ctx* ctx) {
    if (ctx->nranks != 1) return false;
    const long long n = 3LL * ctx->nz * ctx->nx, k = 3LL * (ctx->nx + 1) + 2;
    return n * (3 * k + 1) <= 60000000LL;
}

// factorise D_r A (the row-scaled operator the Krylov solver works on)
int pl_direct_factor(pl_ctx* ctx, const PlStokesOp& op_scaled) {
    const PlGeom& g = ctx->geom.d;
    PlDirect* D = (PlDirect*)ctx->direct;
    const int n = 3 * g.nz * g.nx, kl = 3 * (g.nx + 1) + 2, ku = kl, ld = 2 * kl + ku + 1;
    if (!D || D->n != n || D->kl != kl) {
        pl_direct_free(ctx);
        D = new PlDirect();
        ctx->direct = D;
        D->n = n; D->kl = kl; D->ku = ku; D->ld = ld;
        PL_HIP(ctx, hipMalloc((void**)&D->ab, (size_t)n * ld * sizeof(double)));
        PL_HIP(ctx, hipMalloc((void**)&D->piv, (size_t)n * sizeof(int)));
        PL_HIP(ctx, hipMalloc((void**)&D->work, (size_t)n * sizeof(double)));
        PL_HIP(ctx, hipMalloc((void**)&D->probe, (size_t)6 * g.plane * sizeof(double)));
        PL_HIP(ctx, hipMalloc((void**)&D->info, sizeof(int)));
    }
    PL_HIP(ctx, hipMemsetAsync(D->ab, 0, (size_t)n * ld * sizeof(double), ctx->stream));
    PL_HIP(ctx, hipMemsetAsync(D->probe, 0, (size_t)6 * g.plane * sizeof(double), ctx->stream));
    PL_HIP(ctx, hipMemsetAsync(D->info, 0, sizeof(int), ctx->stream));
    double* x = D->probe; double* y = D->probe + 3 * g.plane;
    for (int ci = 0; ci < 3; ci++)
        for (int cj = 0; cj < 3; cj++)
            for (int q = 0; q < 3; q++) {
                hipLaunchKernelGGL(kd_indicator, grid2d(g), dim3(64, 4), 0, ctx->stream, g, ci, cj, q, x);
                pl_launch_stokes_apply(ctx, op_scaled, x, y);
                hipLaunchKernelGGL(kd_scatter_band, grid2d(g), dim3(64, 4), 0, ctx->stream, g, ci, cj, q, (const double*)y, kl, ku, ld, D->ab);
            }
    hipLaunchKernelGGL(kd_gbtrf, dim3(1), dim3(1024), 0, ctx->stream, n, kl, ku, ld, D->ab, D->piv, D->info);
    int info = 0;
    PL_HIP(ctx, hipMemcpyAsync(&info, D->info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PL_HIP(ctx, hipGetLastError());
    if (info != 0) return pl_fail(ctx, "direct fallback: the matrix is exactly singular");
    return 0;
}

// out = (D_r A)^-1 in   (3 ring planes each)
int pl_direct_solve(pl_ctx* ctx, const double* in, double* out) {
    PlDirect* D = (PlDirect*)ctx->direct;
    if (!D) return pl_fail(ctx, "direct fallback: no factorisation");
    const PlGeom& g = ctx->geom.d;
    hipLaunchKernelGGL(kd_planes_to_vec, grid2d(g), dim3(64, 4), 0, ctx->stream, g, in, D->work);
    hipLaunchKernelGGL(kd_gbtrs, dim3(1), dim3(1024), 0, ctx->stream, D->n, D->kl, D->ku, D->ld, (const double*)D->ab, (const int*)D->piv, D->work);
    hipLaunchKernelGGL(kd_vec_to_planes, grid2d(g), dim3(64, 4), 0, ctx->stream, g, (const double*)D->work, out);
    PL_HIP(ctx, hipGetLastError());
    return 0;
}
