// Krylov solvers replacing the reference's scipy.sparse.linalg.spsolve call sites
// (pylamp2.py:360,394 Stokes; pylamp2.py:419 heat).
//
// Stokes:  right-preconditioned BiCGStab with a seeded random shadow residual on the
// ROW-SCALED system (D_r A) x = D_r b  (scaling makes ||r|| track the velocity error over
// 6 decades of viscosity; it does not change x).  Preconditioner: block upper triangular
//        M = [[A_vv, A_vp], [0, S^]],  S^ = diag(Kc^2/eta_n) on continuity rows,
// with A_vv^-1 ~ one geometric-multigrid V-cycle on the staggered velocity block:
// rediscretised coarse operators (arithmetic viscosity coarsening), Chebyshev-Jacobi
// smoothing, constraint rows (walls, slaved tangential rows, ghosts) closed inside every
// kernel.  The finest level uses the reference's rows (slaved outermost in-domain
// velocities, pylamp_stokes.py:170-175,209-214,249-255,296-301); coarse levels use natural
// mirror rows so the effective wall does not move by h_l per level.
// Heat: BiCGStab on the Jacobi-scaled operator.
//
// Algorithm prototype with the same structure: oracle/proto_stokes_solver.py (tests only).
#include "pl_internal.h"
#include <chrono>
#include <cmath>
#include <functional>
#include <limits>

#define TB(tab, k) (tab)[(k) + PL_TOFF]

void pl_launch_heat_rhs(pl_ctx* ctx, const PlHeatOp& op, const double* Told, const double* H, double* rhs);

// =========================================================================================
// Velocity block.  Everything here is a template over the storage / arithmetic type T of the multigrid level:
// double, or float for the large (bandwidth-bound) levels -- the preconditioner is only an approximation of
// A_vv^-1, BiCGStab itself stays FP64 ("FP32 multigrid levels").
// =========================================================================================
template <typename T>
struct PlVvOpT {
    PlGeom g;
    const T* etas; const T* etan;
    const T *rdz, *rDz, *rdx, *rDx;     // the tables of g in T (T = double: g's own)
    int slave_x;            // vz rows j=0 / j=nx-2 slaved to their inner neighbour (reference rows)
    int slave_z0, slave_zL; // vx rows i=0 / i=nz-2 slaved (reference rows, or NOSLIP on any level)
    T s0, sL;               // slave factor: v_slave = s * v_master
    // free-surface stabilisation terms of the z-momentum rows (pylamp_stokes.py:422-426), or NULL:
    // row_z += szz * vz[c] + szx * vx[c]   (the x rows carry G[IX] = 0 and are left alone)
    const T* szz; const T* szx;
};
typedef PlVvOpT<double> PlVvOp;
typedef PlVvOpT<float> PlVvOpF;

enum { VV_ZERO = 0, VV_INT = 1, VV_SLAVE = 2 };

template <typename T>
__device__ inline int vv_cls_z(const PlVvOpT<T>& op, int i, int j, int& moff, T& s) {
    const int nz = op.g.nz, nx = op.g.nx;
    moff = 0; s = T(1);
    if (j >= nx - 1 || i <= 0 || i >= nz - 1) return VV_ZERO;
    if (op.slave_x) {
        if (j == 0) { moff = 1; s = T(1); return VV_SLAVE; }
        if (j == nx - 2) { moff = -1; s = T(1); return VV_SLAVE; }
    }
    return VV_INT;
}

template <typename T>
__device__ inline int vv_cls_x(const PlVvOpT<T>& op, int i, int j, int& moff, T& s) {
    const int nz = op.g.nz, nx = op.g.nx;
    moff = 0; s = T(1);
    if (i >= nz - 1 || j <= 0 || j >= nx - 1) return VV_ZERO;
    if (i == 0 && op.slave_z0) { moff = op.g.pitch; s = op.s0; return VV_SLAVE; }
    if (i == nz - 2 && op.slave_zL) { moff = -op.g.pitch; s = op.sL; return VV_SLAVE; }
    return VV_INT;
}

// diagonal of a stabilised row is -dg + szz; the smoother keeps it at least half the viscous one
// (the reference's sign of the term REDUCES the diagonal)
template <typename T>
__device__ inline T vv_stab_diag(T dg, T szz) { const T d = dg - szz; return d > T(0.5) * dg ? d : T(0.5) * dg; }

// (A_vv v)_z and -diag at global node (i,j), plane offset c.  Zero-padded tables make the
// mirror terms of the natural rows vanish (rDx[0] = rDx[nx-1] = 0, same in z).
// (TV: storage type of the iterate; the arithmetic is T)
template <typename T, typename TV>
__device__ inline void vv_row_z(const PlVvOpT<T>& op, const TV* __restrict__ vz, const TV* __restrict__ vx,
                                int c, int i, int j, T& Av, T& dg) {
    const int p = op.g.pitch;
    const T rdz_i = TB(op.rdz, i), rdz_m = TB(op.rdz, i - 1), rDz_i = TB(op.rDz, i);
    const T rdx_j = TB(op.rdx, j), rDx_j = TB(op.rDx, j), rDx_p = TB(op.rDx, j + 1);
    const T esC = op.etas[c], esE = op.etas[c + 1];
    const T cN = T(4) * op.etan[c] * rdz_i * rDz_i, cS = T(4) * op.etan[c - p] * rdz_m * rDz_i;
    const T cE = T(2) * esE * rDx_p * rdx_j, cW = T(2) * esC * rDx_j * rdx_j;
    const T xE = T(2) * esE * rDz_i * rdx_j, xW = T(2) * esC * rDz_i * rdx_j;
    const T v0 = (T)vz[c];
    Av = cN * ((T)vz[c + p] - v0) - cS * (v0 - (T)vz[c - p]) + cE * ((T)vz[c + 1] - v0) - cW * (v0 - (T)vz[c - 1]) +
         xE * ((T)vx[c + 1] - (T)vx[c - p + 1]) - xW * ((T)vx[c] - (T)vx[c - p]);
    dg = cN + cS + cE + cW;
    if (op.szz) {                                   // wave-uniform
        const T sd = op.szz[c];
        Av += sd * v0 + op.szx[c] * (T)vx[c];
        dg = vv_stab_diag(dg, sd);
    }
}

template <typename T, typename TV>
__device__ inline void vv_row_x(const PlVvOpT<T>& op, const TV* __restrict__ vz, const TV* __restrict__ vx,
                                int c, int i, int j, T& Av, T& dg) {
    const int p = op.g.pitch;
    const T rdx_j = TB(op.rdx, j), rdx_m = TB(op.rdx, j - 1), rDx_j = TB(op.rDx, j);
    const T rdz_i = TB(op.rdz, i), rDz_i = TB(op.rDz, i), rDz_p = TB(op.rDz, i + 1);
    const T esC = op.etas[c], esN = op.etas[c + p];
    const T cE = T(4) * op.etan[c] * rdx_j * rDx_j, cW = T(4) * op.etan[c - 1] * rdx_m * rDx_j;
    const T cN = T(2) * esN * rDz_p * rdz_i, cS = T(2) * esC * rDz_i * rdz_i;
    const T zN = T(2) * esN * rDx_j * rdz_i, zS = T(2) * esC * rDx_j * rdz_i;
    const T v0 = (T)vx[c];
    Av = cE * ((T)vx[c + 1] - v0) - cW * (v0 - (T)vx[c - 1]) + cN * ((T)vx[c + p] - v0) - cS * (v0 - (T)vx[c - p]) +
         zN * ((T)vz[c + p] - (T)vz[c + p - 1]) - zS * ((T)vz[c] - (T)vz[c - 1]);
    dg = cE + cW + cN + cS;
}

#define PL_NODE_PROLOGUE(g)                                                         \
    const int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y; \
    if (lj >= (g).lnx || li >= (g).lnz) return;                                      \
    const int i = (g).gi0 + li, j = (g).gj0 + lj;                                    \
    const long long c = pl_idx((g), li, lj);

// One Chebyshev-Jacobi sweep in three-term form:
//     v_next = v_cur + c1 (v_cur - v_prev) + c2 D^-1 (f - A v_cur)
// with the constraint rows closed in the same pass (a slave thread evaluates its master's
// update; all inputs are read-only so there is no ordering hazard).  80 B/node/sweep (FP64).
// (32-bit element offsets from wave-uniform plane bases keep the address arithmetic in SGPR-base +
// VGPR-offset form: fewer VGPRs, more waves per SIMD)
template <typename T, typename TV, typename TF>
__device__ inline T cheb_val_z(const PlVvOpT<T>& op, const TV* __restrict__ vcur, const TV* __restrict__ vprev,
                               const TF* __restrict__ f, T c1, T c2, int i, int j, int c) {
    int moff = 0; T s = T(1);
    if (vv_cls_z(op, i, j, moff, s) == VV_ZERO) return T(0);
    const int cm = c + moff;
    T Av, dg;
    vv_row_z(op, vcur, vcur + op.g.plane, cm, i, j + moff, Av, dg);   // moff is +-1 for vz
    const T v0 = (T)vcur[cm];
    const T mom = (c1 != T(0)) ? c1 * (v0 - (vprev ? (T)vprev[cm] : T(0))) : T(0);   // nullptr: previous iterate is zero
    return s * (v0 + mom + (c2 * (Av - (T)f[cm])) * pl_rcp(dg));                     // D = -dg
}
template <typename T, typename TV, typename TF>
__device__ inline T cheb_val_x(const PlVvOpT<T>& op, const TV* __restrict__ vcur, const TV* __restrict__ vprev,
                               const TF* __restrict__ f, T c1, T c2, int i, int j, int c) {
    const long long P = op.g.plane;
    int moff = 0; T s = T(1);
    if (vv_cls_x(op, i, j, moff, s) == VV_ZERO) return T(0);
    const int cm = c + moff;
    const int im = i + (moff > 0 ? 1 : (moff < 0 ? -1 : 0));
    T Av, dg;
    vv_row_x(op, vcur, vcur + P, cm, im, j, Av, dg);
    const T v0 = (T)(vcur + P)[cm];
    const T mom = (c1 != T(0)) ? c1 * (v0 - (vprev ? (T)(vprev + P)[cm] : T(0))) : T(0);
    return s * (v0 + mom + (c2 * (Av - (T)(f + P)[cm])) * pl_rcp(dg));
}
// TO: type of the destination (the last sweep of an FP32 level 0 writes the FP64 Krylov vector, times oscale)
template <typename T, typename TO, typename TV, typename TF>
__device__ inline void cheb_node(const PlVvOpT<T>& op, const TV* __restrict__ vcur, const TV* __restrict__ vprev,
                                 const TF* __restrict__ f, TO* __restrict__ vnext, T c1, T c2, int i,
                                 int j, long long c64, TO oscale = TO(1)) {
    const int c = (int)c64;
    vnext[c] = (TO)cheb_val_z(op, vcur, vprev, f, c1, c2, i, j, c) * oscale;
    (vnext + op.g.plane)[c] = (TO)cheb_val_x(op, vcur, vprev, f, c1, c2, i, j, c) * oscale;
}

// diagonal sums only (no velocity reads)
template <typename T>
__device__ inline T vv_diag_z(const PlVvOpT<T>& op, int c, int i, int j) {
    const int p = op.g.pitch;
    const T rDz_i = TB(op.rDz, i), rdx_j = TB(op.rdx, j);
    const T dg = T(4) * op.etan[c] * TB(op.rdz, i) * rDz_i + T(4) * op.etan[c - p] * TB(op.rdz, i - 1) * rDz_i +
                 T(2) * op.etas[c + 1] * TB(op.rDx, j + 1) * rdx_j + T(2) * op.etas[c] * TB(op.rDx, j) * rdx_j;
    return op.szz ? vv_stab_diag(dg, op.szz[c]) : dg;
}
template <typename T>
__device__ inline T vv_diag_x(const PlVvOpT<T>& op, int c, int i, int j) {
    const int p = op.g.pitch;
    const T rDx_j = TB(op.rDx, j), rdz_i = TB(op.rdz, i);
    return T(4) * op.etan[c] * TB(op.rdx, j) * rDx_j + T(4) * op.etan[c - 1] * TB(op.rdx, j - 1) * rDx_j +
           T(2) * op.etas[c + p] * TB(op.rDz, i + 1) * rdz_i + T(2) * op.etas[c] * TB(op.rDz, i) * rdz_i;
}

// First sweep from a ZERO guess: A v = 0, so v1 = -c2 f / diag needs no stencil and no memset of the
// iterate (48 instead of 80 + 16 B/node).  Slaves copy their master's value as usual.
template <typename T, typename TF, typename TV>
__device__ inline void cheb_first_node(const PlVvOpT<T>& op, const TF* __restrict__ f, TV* __restrict__ vnext,
                                       T c2, int i, int j, long long c64) {
    const long long P = op.g.plane;
    const int c = (int)c64;
    const TF* __restrict__ fz = f; const TF* __restrict__ fx = f + P;
    int moff = 0; T s = T(1);
    int cls = vv_cls_z(op, i, j, moff, s);
    T out = T(0);
    if (cls != VV_ZERO) { const int cm = c + moff; out = (-s * c2 * (T)fz[cm]) * pl_rcp(vv_diag_z(op, cm, i, j + moff)); }
    vnext[c] = (TV)out;
    cls = vv_cls_x(op, i, j, moff, s);
    out = T(0);
    if (cls != VV_ZERO) {
        const int cm = c + moff;
        const int im = i + (moff > 0 ? 1 : (moff < 0 ? -1 : 0));
        out = (-s * c2 * (T)fx[cm]) * pl_rcp(vv_diag_x(op, cm, im, j));
    }
    (vnext + P)[c] = (TV)out;
}

template <typename T>
__global__ __launch_bounds__(256) void k_vv_cheb_first(PlVvOpT<T> op, const T* __restrict__ f, T* __restrict__ vnext,
                                                       T c2, int iters) {
    PL_ROW_LOOP(op.g, iters)
        cheb_first_node(op, f, vnext, c2, op.g.gi0 + li, op.g.gj0 + lj, pl_idx(op.g, li, lj));
}

// =========================================================================================
// LINE relaxation (stretched and graded grids); described for z-lines, k_vv_line<1> is the transposed twin.  Where the cells are much wider than high (dx >> dz) the
// z-couplings of a velocity row are ~ (dx/dz)^2 times its x-couplings: point Jacobi leaves the modes that are smooth along z untouched
// and full coarsening cannot carry them (513 x 129 nodes on a square: 41 iterations; a grid graded 30 x in z: 209).  Here the block
// of the Jacobi splitting is the TRIDIAGONAL matrix T of each component's own z-couplings, column by column:
//     v_next = v + c1 (v - v_prev) + c2 T^-1 (A v - f),        T v = dg v - cN v[i+1] - cS v[i-1]   on the interior rows
// (wall rows: identity; slaved rows take s x their master's value afterwards, as in the point sweeps), Chebyshev-accelerated on the
// spectrum of T^-1 A exactly like the point version (the power iteration runs through this kernel: mode 1).
// One workgroup solves LZ columns of ONE component by parallel cyclic reduction in LDS: ceil(log2 nz) steps, each thread keeps its
// rows' (a, b, c, d) in registers and reads the rows i -+ stride of the previous step from LDS.  32 nz LZ bytes of LDS: nz <= 4097.
// =========================================================================================
#define LZ_NT 1024
#define LZ_EPT 5            // rows per thread at most: nz <= 4097 + (LZ_EPT * LZ_NT >= nz * cols)
struct LineArgs {
    PlVvOp op; const double* v; const double* vprev; const double* f; double* out;
    double c1, c2, oscale; int mode /* 0 sweep, 1 out = T^-1 A v */; int cols;
};
// AX = 0: lines along z (cells wider than high); AX = 1: lines along x (cells higher than wide) -- the same kernel with the roles of the
// axes swapped: the couplings of T are the component's own couplings ALONG the line, a slaved line evaluates its master line, slaved
// points inside a line copy their neighbour behind the solve.
template <int AX>
__global__ __launch_bounds__(LZ_NT) void k_vv_line(LineArgs a) {
    extern __shared__ double lz_sh[];
    const PlGeom& g = a.op.g;
    const int n = AX == 0 ? g.nz : g.nx, nlines = AX == 0 ? g.nx : g.nz;     // points per line, lines of the level
    const int cols = a.cols, comp = blockIdx.y, l0 = blockIdx.x * cols, tot = n * cols;
    const int stride = AX == 0 ? cols : 1;                                    // LDS distance of neighbours along the line
    double* const SA = lz_sh; double* const SB = SA + tot; double* const SC = SB + tot; double* const SD = SC + tot;
    const long long P = g.plane;
    const double* vz = a.v; const double* vx = a.v ? a.v + P : nullptr;
    double A_[LZ_EPT], B_[LZ_EPT], C_[LZ_EPT], D_[LZ_EPT], base[LZ_EPT], sfac[LZ_EPT];
    int cout_[LZ_EPT];      // element offset of the node this entry is written to (-1: none)
    // interior (equation-carrying) rows of this component
    auto is_int = [&](int i, int j) {
        if (comp == 0) return i >= 1 && i <= g.nz - 2 && j >= 0 && j < g.nx - 1 && !(a.op.slave_x && (j == 0 || j == g.nx - 2));
        return j >= 1 && j <= g.nx - 2 && i >= (a.op.slave_z0 ? 1 : 0) && i <= (a.op.slave_zL ? g.nz - 3 : g.nz - 2);
    };
    // position along the line and line of entry e (AX = 0: the lines of a workgroup are adjacent columns, line index fastest;
    // AX = 1: adjacent rows, position fastest -- either way neighbouring threads read neighbouring columns)
    auto pos_of = [&](int e) { return AX == 0 ? e / cols : e % n; };
    auto line_of = [&](int e) { return AX == 0 ? e % cols : e / n; };
#pragma unroll
    for (int q = 0; q < LZ_EPT; q++) {
        const int e = threadIdx.x + q * LZ_NT;
        A_[q] = 0.0; B_[q] = 1.0; C_[q] = 0.0; D_[q] = 0.0; base[q] = 0.0; sfac[q] = 0.0; cout_[q] = -1;
        if (e >= tot) continue;
        const int ln = l0 + line_of(e);
        if (ln >= nlines) continue;
        const int i = AX == 0 ? pos_of(e) : ln, j = AX == 0 ? ln : pos_of(e);
        const int c = (int)pl_idx(g, i - g.gi0, j - g.gj0);
        cout_[q] = c;
        // the node whose row this entry evaluates: a point of a slaved LINE its master's
        int ie = i, je = j;
        if (AX == 0 && comp == 0 && a.op.slave_x && (j == 0 || j == g.nx - 2)) je = j == 0 ? 1 : g.nx - 3;
        if (AX == 1 && comp == 1 && ((i == 0 && a.op.slave_z0) || (i == g.nz - 2 && a.op.slave_zL)) && g.nz >= 4) ie = i == 0 ? 1 : g.nz - 3;
        if (!is_int(ie, je)) continue;               // wall row (0), or a slaved point inside the line (filled in behind the solve)
        const int p = g.pitch;
        const int cm = c + (je - j) + (ie - i) * p;
        double Av = 0.0, dg, cNext, cPrev;
        if (comp == 0) {
            if (AX == 0) {
                const double rdz_i = TB(a.op.rdz, ie), rdz_m = TB(a.op.rdz, ie - 1), rDz_i = TB(a.op.rDz, ie);
                cNext = 4.0 * a.op.etan[cm] * rdz_i * rDz_i; cPrev = 4.0 * a.op.etan[cm - p] * rdz_m * rDz_i;
            } else {
                const double rdx_j = TB(a.op.rdx, je), rDx_j = TB(a.op.rDx, je), rDx_p = TB(a.op.rDx, je + 1);
                cNext = 2.0 * a.op.etas[cm + 1] * rDx_p * rdx_j; cPrev = 2.0 * a.op.etas[cm] * rDx_j * rdx_j;
            }
            if (a.v) vv_row_z(a.op, vz, vx, cm, ie, je, Av, dg); else dg = vv_diag_z(a.op, cm, ie, je);
            if (a.v) base[q] = vz[cm] + ((a.c1 != 0.0 && a.mode == 0) ? a.c1 * (vz[cm] - (a.vprev ? a.vprev[cm] : 0.0)) : 0.0);
        } else {
            if (AX == 0) {
                const double rdz_i = TB(a.op.rdz, ie), rDz_i = TB(a.op.rDz, ie), rDz_p = TB(a.op.rDz, ie + 1);
                cNext = 2.0 * a.op.etas[cm + p] * rDz_p * rdz_i; cPrev = 2.0 * a.op.etas[cm] * rDz_i * rdz_i;
            } else {
                const double rdx_j = TB(a.op.rdx, je), rdx_m = TB(a.op.rdx, je - 1), rDx_j = TB(a.op.rDx, je);
                cNext = 4.0 * a.op.etan[cm] * rdx_j * rDx_j; cPrev = 4.0 * a.op.etan[cm - 1] * rdx_m * rDx_j;
            }
            if (a.v) vv_row_x(a.op, vz, vx, cm, ie, je, Av, dg); else dg = vv_diag_x(a.op, cm, ie, je);
            if (a.v) base[q] = vx[cm] + ((a.c1 != 0.0 && a.mode == 0) ? a.c1 * (vx[cm] - (a.vprev ? (a.vprev + P)[cm] : 0.0)) : 0.0);
        }
        // a slaved line takes s x its master line (vz columns: s = 1; vx rows: s0 / sL)
        sfac[q] = (AX == 1 && comp == 1 && ie != i) ? (i == 0 ? a.op.s0 : a.op.sL) : 1.0;
        B_[q] = dg;
        A_[q] = is_int(AX == 0 ? ie - 1 : ie, AX == 0 ? je : je - 1) ? -cPrev : 0.0;
        C_[q] = is_int(AX == 0 ? ie + 1 : ie, AX == 0 ? je : je + 1) ? -cNext : 0.0;
        D_[q] = Av - ((a.mode == 0 && a.f) ? (a.f + comp * P)[cm] : 0.0);
    }
    if (a.mode == 1) {
#pragma unroll
        for (int q = 0; q < LZ_EPT; q++) base[q] = 0.0;
    }
    // ---- parallel cyclic reduction: after the step of stride st every row couples to the rows pos -+ 2 st only
    for (int st = 1; st < n; st <<= 1) {
#pragma unroll
        for (int q = 0; q < LZ_EPT; q++) {
            const int e = threadIdx.x + q * LZ_NT;
            if (e < tot) { SA[e] = A_[q]; SB[e] = B_[q]; SC[e] = C_[q]; SD[e] = D_[q]; }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < LZ_EPT; q++) {
            const int e = threadIdx.x + q * LZ_NT;
            if (e >= tot) continue;
            const int ps = pos_of(e);
            double al = 0.0, ga = 0.0, am = 0.0, cm_ = 0.0, dm = 0.0, ap = 0.0, cp = 0.0, dp = 0.0;
            if (ps - st >= 0 && A_[q] != 0.0) { const int m = e - st * stride; al = -A_[q] / SB[m]; am = SA[m]; cm_ = SC[m]; dm = SD[m]; }
            if (ps + st < n && C_[q] != 0.0) { const int m = e + st * stride; ga = -C_[q] / SB[m]; ap = SA[m]; cp = SC[m]; dp = SD[m]; }
            B_[q] = B_[q] + al * cm_ + ga * ap;
            D_[q] = D_[q] + al * dm + ga * dp;
            A_[q] = al * am; C_[q] = ga * cp;
        }
        __syncthreads();
    }
    // ---- the update; slaved points inside a line copy s x their master's (the point next to them on the line)
#pragma unroll
    for (int q = 0; q < LZ_EPT; q++) {
        const int e = threadIdx.x + q * LZ_NT;
        if (e < tot) SD[e] = sfac[q] != 0.0 ? sfac[q] * (base[q] + (a.mode == 0 ? a.c2 : 1.0) * (D_[q] / B_[q])) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < LZ_EPT; q++) {
        const int e = threadIdx.x + q * LZ_NT;
        if (e >= tot || cout_[q] < 0) continue;
        const int ln = l0 + line_of(e);
        const int i = AX == 0 ? pos_of(e) : ln, j = AX == 0 ? ln : pos_of(e);
        double val = SD[e];
        if (AX == 0 && comp == 1 && j >= 1 && j <= g.nx - 2 && n >= 4) {
            if (i == 0 && a.op.slave_z0) val = a.op.s0 * SD[e + stride];
            else if (i == n - 2 && a.op.slave_zL) val = a.op.sL * SD[e - stride];
        }
        if (AX == 1 && comp == 0 && a.op.slave_x && i >= 1 && i <= g.nz - 2 && n >= 4) {
            if (j == 0) val = SD[e + 1];
            else if (j == n - 2) val = SD[e - 1];
        }
        (a.out + comp * P)[cout_[q]] = val * a.oscale;
    }
}
static bool line_z_possible(const pl_ctx* ctx, int npts) { return ctx->nranks == 1 && npts <= 4097 && npts >= 4; }
// ax = 0: z-lines, 1: x-lines
static int line_z_launch(pl_ctx* ctx, const PlVvOp& op, const double* v, const double* vprev, const double* f, double* out, double c1, double c2,
                         double oscale, int mode, int ax) {
    LineArgs a{}; a.op = op; a.v = v; a.vprev = vprev; a.f = f; a.out = out; a.c1 = c1; a.c2 = c2; a.oscale = oscale; a.mode = mode;
    const int n = ax == 0 ? op.g.nz : op.g.nx, nlines = ax == 0 ? op.g.nx : op.g.nz;
    int cols = (LZ_EPT * LZ_NT) / n; if (cols > 16) cols = 16; if (cols > nlines) cols = nlines; if (cols < 1) cols = 1;
    while (cols > 1 && (size_t)32 * n * cols > 150000) cols--;
    a.cols = cols;
    const size_t lds = (size_t)32 * n * cols;
    static bool lds_set = false;
    if (!lds_set) {
        PL_HIP(ctx, hipFuncSetAttribute((const void*)k_vv_line<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PL_HIP(ctx, hipFuncSetAttribute((const void*)k_vv_line<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_set = true;
    }
    if (ax == 0) hipLaunchKernelGGL(k_vv_line<0>, dim3((nlines + cols - 1) / cols, 2), dim3(LZ_NT), lds, ctx->stream, a);
    else hipLaunchKernelGGL(k_vv_line<1>, dim3((nlines + cols - 1) / cols, 2), dim3(LZ_NT), lds, ctx->stream, a);
    return 0;
}

#ifndef PL_CHEB_WAVES
#define PL_CHEB_WAVES 4
#endif
template <typename T, typename TO>
__global__ __launch_bounds__(256, PL_CHEB_WAVES) void k_vv_cheb(PlVvOpT<T> op, const T* __restrict__ vcur,
                                                 const T* __restrict__ vprev, const T* __restrict__ f,
                                                 TO* __restrict__ vnext, T c1, T c2, int iters, TO oscale) {
    PL_ROW_LOOP(op.g, iters)
        cheb_node(op, vcur, vprev, f, vnext, c1, c2, op.g.gi0 + li, op.g.gj0 + lj, pl_idx(op.g, li, lj), oscale);
}

// r = f - A v on interior rows, 0 elsewhere
// (f and r may alias: every thread reads only its own f entries before writing r)
template <typename T, typename TV, typename TF, typename TR>
__device__ inline void residual_node(const PlVvOpT<T>& op, const TV* __restrict__ v, const TF* f, TR* r, int i,
                                     int j, long long c64) {
    const long long P = op.g.plane;
    const int c = (int)c64;
    const TF* fx = f + P; TR* rx_ = r + P;
    int moff; T s, Av, dg;
    T rz = T(0), rx = T(0);
    if (vv_cls_z(op, i, j, moff, s) == VV_INT) { vv_row_z(op, v, v + P, c, i, j, Av, dg); rz = (T)f[c] - Av; }
    if (vv_cls_x(op, i, j, moff, s) == VV_INT) { vv_row_x(op, v, v + P, c, i, j, Av, dg); rx = (T)fx[c] - Av; }
    r[c] = (TR)rz; rx_[c] = (TR)rx;
}

template <typename T>
__global__ __launch_bounds__(256) void k_vv_residual(PlVvOpT<T> op, const T* __restrict__ v, const T* f,
                                                     T* r, int iters) {
    PL_ROW_LOOP(op.g, iters)
        residual_node(op, v, f, r, op.g.gi0 + li, op.g.gj0 + lj, pl_idx(op.g, li, lj));
}

// ---------------------------------------------------------------------------------------------------
// Two-columns-per-lane sweep (same reasoning as k_stokes_apply_v2: the scalar kernels keep the texture
// address unit busy 88 % of the time).  MODE 0: Chebyshev sweep, MODE 1: residual.  A wave covers 128
// columns of one row with 2-element vector loads; west/east neighbours come from the adjacent lanes.  The two
// wall-adjacent rows (vx slaved or zero) and the two slaved vz columns take the scalar path.
template <typename T>
struct VvVals {
    T vz_c, vz_w, vz_e, vz_n, vz_s, vz_nw;
    T vx_c, vx_w, vx_e, vx_n, vx_s, vx_se;
    T en_c, en_w, en_s, es_c, es_e, es_n;
    T rdx_j, rdx_m, rDx_j, rDx_p;
};
// Row-only products are shared by the lane's two columns, every viscosity is multiplied once (the arithmetic
// of these sweeps is not hidden behind the memory traffic: FP64 issues at half rate on CDNA4).
template <typename T> struct VvRowK { T Az, Azm, rDz_i, rDz_p, r2; };
template <bool NEED_D, typename T>
__device__ inline void vv_rows_vals(const VvVals<T>& q, const VvRowK<T>& k, T& Az, T& dz, T& Ax, T& dx) {
    const T k2 = T(2) * q.rdx_j;
    const T tE = q.es_e * k2, tW = q.es_c * k2;
    const T cN = q.en_c * k.Az, cS = q.en_s * k.Azm, cE = tE * q.rDx_p, cW = tW * q.rDx_j;
    Az = cN * (q.vz_n - q.vz_c) - cS * (q.vz_c - q.vz_s) + cE * (q.vz_e - q.vz_c) - cW * (q.vz_c - q.vz_w) +
         (tE * k.rDz_i) * (q.vx_e - q.vx_se) - (tW * k.rDz_i) * (q.vx_c - q.vx_s);
    if (NEED_D) dz = cN + cS + cE + cW;
    const T B4 = T(4) * q.rDx_j;
    const T uN = q.es_n * k.r2, uS = q.es_c * k.r2;
    const T dE = q.en_c * (B4 * q.rdx_j), dW = q.en_w * (B4 * q.rdx_m), dN = uN * k.rDz_p, dS = uS * k.rDz_i;
    Ax = dE * (q.vx_e - q.vx_c) - dW * (q.vx_c - q.vx_w) + dN * (q.vx_n - q.vx_c) - dS * (q.vx_c - q.vx_s) +
         (uN * q.rDx_j) * (q.vz_n - q.vz_nw) - (uS * q.rDx_j) * (q.vz_c - q.vz_w);
    if (NEED_D) dx = dE + dW + dN + dS;
}

// TV / TF: storage types of the iterate and of the right-hand side (the arithmetic is T; level 0 keeps f, its first iterate and its
// residual in FP32 -- PlSolver::l0_mixed)
template <int MODE, typename T, typename TO, typename TV = T, typename TF = T>
__global__ __launch_bounds__(256) void k_vv_sweep2(PlVvOpT<T> op, const TV* __restrict__ vcur, const TV* __restrict__ vprev,
                                                   const TF* f, TO* out, T c1, T c2, TO oscale) {
    typedef typename PlVec2<T>::type V2;
    typedef typename PlVec2<TO>::type VO2;
    typedef typename PlVec2<TV>::type VV2;
    typedef typename PlVec2<TF>::type VF2;
    const PlGeom& g = op.g;
    const int lane = threadIdx.x;
    const int lj0 = (blockIdx.x * 64 + lane) * 2;
    const int li = blockIdx.y * 4 + threadIdx.y;
    if (li >= g.lnz) return;                                // wave-uniform
    const bool active = lj0 < g.lnx;
    const bool has_right = (lj0 + 2) < g.lnx;
    const int p = g.pitch, nz = g.nz, nx = g.nx;
    const long long PLN = g.plane;
    const int c = (int)pl_idx(g, li, lj0);
    const int i = g.gi0 + li, j0 = g.gj0 + lj0;
    if (i <= 0 || i >= nz - 2) {                            // wall rows and the slaved vx rows (wave-uniform)
        if (!active) return;
        for (int q = 0; q < 2 && lj0 + q < g.lnx; q++) {
            if constexpr (MODE == 0) cheb_node(op, vcur, vprev, f, out, c1, c2, i, j0 + q, c + q, oscale);
            else residual_node(op, vcur, f, out, i, j0 + q, c + q);
        }
        return;
    }
    const TV* __restrict__ vz = vcur;
    const TV* __restrict__ vx = vcur + PLN;
#define ROW(ptr, dr, w, e) load_row2_as<T>((ptr) + (long long)c - lj0 + (long long)(dr) * p, lj0, active, w, e, lane, has_right)
    const Row2T<T> vz_s = ROW(vz, -1, false, false), vz_i = ROW(vz, 0, true, true), vz_n = ROW(vz, 1, true, false);
    const Row2T<T> vx_s = ROW(vx, -1, false, true), vx_i = ROW(vx, 0, true, true), vx_n = ROW(vx, 1, false, false);
    const Row2T<T> en_s = ROW(op.etan, -1, false, false), en_i = ROW(op.etan, 0, true, false);
    const Row2T<T> es_i = ROW(op.etas, 0, false, true), es_n = ROW(op.etas, 1, false, false);
#undef ROW
    const Row2T<T> t_rdx = load_row2(op.rdx + PL_TOFF + g.gj0, lj0, active, true, false, lane, has_right);
    const Row2T<T> t_rDx = load_row2(op.rDx + PL_TOFF + g.gj0, lj0, active, false, true, lane, has_right);
    if (!active) return;
    const bool colB = (lj0 + 1) < g.lnx;
    const VF2 fz_s = *reinterpret_cast<const VF2*>(f + c), fx_s = *reinterpret_cast<const VF2*>(f + PLN + c);
    const V2 fz = PlVec2<T>::make((T)fz_s.x, (T)fz_s.y), fx = PlVec2<T>::make((T)fx_s.x, (T)fx_s.y);
    V2 pz = PlVec2<T>::make(T(0), T(0)), px = pz;
    if (MODE == 0 && c1 != T(0) && vprev) {
        const VV2 a_ = *reinterpret_cast<const VV2*>(vprev + c), b_ = *reinterpret_cast<const VV2*>(vprev + PLN + c);
        pz = PlVec2<T>::make((T)a_.x, (T)a_.y); px = PlVec2<T>::make((T)b_.x, (T)b_.y);
    }
    VvRowK<T> rk;
    {
        const T rdz_i = TB(op.rdz, i), rdz_m = TB(op.rdz, i - 1), rDz_i = TB(op.rDz, i);
        rk.Az = T(4) * rdz_i * rDz_i; rk.Azm = T(4) * rdz_m * rDz_i; rk.rDz_i = rDz_i; rk.rDz_p = TB(op.rDz, i + 1); rk.r2 = T(2) * rdz_i;
    }
    VvVals<T> q;
    T Az[2], dz[2] = {T(1), T(1)}, Ax[2], dx[2] = {T(1), T(1)};
    q.vz_c = vz_i.v.x; q.vz_w = vz_i.w; q.vz_e = vz_i.v.y; q.vz_n = vz_n.v.x; q.vz_s = vz_s.v.x; q.vz_nw = vz_n.w;
    q.vx_c = vx_i.v.x; q.vx_w = vx_i.w; q.vx_e = vx_i.v.y; q.vx_n = vx_n.v.x; q.vx_s = vx_s.v.x; q.vx_se = vx_s.v.y;
    q.en_c = en_i.v.x; q.en_w = en_i.w; q.en_s = en_s.v.x;
    q.es_c = es_i.v.x; q.es_e = es_i.v.y; q.es_n = es_n.v.x;
    q.rdx_j = t_rdx.v.x; q.rdx_m = t_rdx.w; q.rDx_j = t_rDx.v.x; q.rDx_p = t_rDx.v.y;
    vv_rows_vals<MODE == 0>(q, rk, Az[0], dz[0], Ax[0], dx[0]);
    q.vz_c = vz_i.v.y; q.vz_w = vz_i.v.x; q.vz_e = vz_i.e; q.vz_n = vz_n.v.y; q.vz_s = vz_s.v.y; q.vz_nw = vz_n.v.x;
    q.vx_c = vx_i.v.y; q.vx_w = vx_i.v.x; q.vx_e = vx_i.e; q.vx_n = vx_n.v.y; q.vx_s = vx_s.v.y; q.vx_se = vx_s.e;
    q.en_c = en_i.v.y; q.en_w = en_i.v.x; q.en_s = en_s.v.y;
    q.es_c = es_i.v.y; q.es_e = es_i.e; q.es_n = es_n.v.y;
    q.rdx_j = t_rdx.v.y; q.rdx_m = t_rdx.v.x; q.rDx_j = t_rDx.v.y; q.rDx_p = t_rDx.e;
    vv_rows_vals<MODE == 0>(q, rk, Az[1], dz[1], Ax[1], dx[1]);
    const T v0z[2] = {vz_i.v.x, vz_i.v.y}, v0x[2] = {vx_i.v.x, vx_i.v.y};
    if (op.szz) {                                                   // wave-uniform
        const V2 sd = *reinterpret_cast<const V2*>(op.szz + c), sx = *reinterpret_cast<const V2*>(op.szx + c);
        Az[0] += sd.x * v0z[0] + sx.x * v0x[0]; dz[0] = vv_stab_diag(dz[0], sd.x);
        Az[1] += sd.y * v0z[1] + sx.y * v0x[1]; dz[1] = vv_stab_diag(dz[1], sd.y);
    }
    const T fzv[2] = {fz.x, fz.y}, fxv[2] = {fx.x, fx.y}, pzv[2] = {pz.x, pz.y}, pxv[2] = {px.x, px.y};
    T oz[2], ox[2];
    // wave-uniform: all 128 columns are interior in both components (no slaves, no zero rows): straight-line code
    const int jw = g.gj0 + blockIdx.x * 128;
    if (jw >= 1 && jw + 127 <= nx - 3) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (MODE == 0) {
                const T mz = c1 * (v0z[k] - pzv[k]), mx = c1 * (v0x[k] - pxv[k]);      // c1 = 0 on the first sweep
                oz[k] = v0z[k] + mz + (c2 * (Az[k] - fzv[k])) * pl_rcp(dz[k]);
                ox[k] = v0x[k] + mx + (c2 * (Ax[k] - fxv[k])) * pl_rcp(dx[k]);
            } else {
                oz[k] = fzv[k] - Az[k]; ox[k] = fxv[k] - Ax[k];
            }
        }
        *reinterpret_cast<VO2*>(out + c) = PlVec2<TO>::make((TO)oz[0] * oscale, (TO)oz[1] * oscale);
        *reinterpret_cast<VO2*>(out + PLN + c) = PlVec2<TO>::make((TO)ox[0] * oscale, (TO)ox[1] * oscale);
        return;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int j = j0 + k;
        const bool zslave = op.slave_x && (j == 0 || j == nx - 2);
        const bool zint = j < nx - 1 && !zslave, xint = j > 0 && j < nx - 1;
        if (MODE == 0) {
            const T mz = (c1 != T(0)) ? c1 * (v0z[k] - pzv[k]) : T(0), mx = (c1 != T(0)) ? c1 * (v0x[k] - pxv[k]) : T(0);
            oz[k] = zint ? v0z[k] + mz + (c2 * (Az[k] - fzv[k])) * pl_rcp(dz[k]) : T(0);
            ox[k] = xint ? v0x[k] + mx + (c2 * (Ax[k] - fxv[k])) * pl_rcp(dx[k]) : T(0);
            if (zslave && (k == 0 || colB)) oz[k] = cheb_val_z(op, vcur, vprev, f, c1, c2, i, j, c + k);
        } else {
            oz[k] = zint ? fzv[k] - Az[k] : T(0);
            ox[k] = xint ? fxv[k] - Ax[k] : T(0);
        }
    }
    if (colB) {
        *reinterpret_cast<VO2*>(out + c) = PlVec2<TO>::make((TO)oz[0] * oscale, (TO)oz[1] * oscale);
        *reinterpret_cast<VO2*>(out + PLN + c) = PlVec2<TO>::make((TO)ox[0] * oscale, (TO)ox[1] * oscale);
    } else {
        out[c] = (TO)oz[0] * oscale; out[PLN + c] = (TO)ox[0] * oscale;
    }
}

// wave-uniform: every node of the 128-column wave at row i is of the interior class in stage 1 (and in the first sweep)
// wall != 0 (PlStokesOp::wall_ps): the two cell rows / columns next to every wall take the per-node path as well -- the pressure
// block there is a stencil over its neighbours (prec_p_value), and the straight-line path evaluates S^-1 r_p of rows i, i - 1
__device__ inline bool stage1_fast_wave(const PlGeom& g, int i, int jw, int bx, int anchor_i, int anchor_j, int wall) {
    const bool anchor_near = (anchor_i == i || anchor_i == i - 1) && anchor_j >= jw - 1 && anchor_j <= jw + 127;
    const int m = wall ? 1 : 0;
    return !(i < 1 + 2 * m || i > g.nz - 3 - m || jw < 1 || jw + 127 > g.nx - 3 - m || anchor_near || bx * 128 + 127 >= g.lnx);
}
// First sweep from the zero guess, two columns per lane (see cheb_first_node): v1 = -c2 f / diag.
// only_slow != 0: stage 1 has already written the waves it treats as interior (stage1_fast_wave); do the others only
template <typename T, typename TF = T, typename TV = T>
__global__ __launch_bounds__(256) void k_vv_first2(PlVvOpT<T> op, const TF* __restrict__ f, TV* __restrict__ out, T c2,
                                                   int only_slow, int anchor_i, int anchor_j, int wall) {
    typedef typename PlVec2<TV>::type V2;
    typedef typename PlVec2<TF>::type VF2;
    const PlGeom& g = op.g;
    const int lane = threadIdx.x;
    // only_slow == 2: a 1-D launch over the FRAME of workgroups that can hold a slow wave (first / last two block rows,
    // first / last two block columns, the anchor's blocks) instead of the whole grid, most of which would exit at once
    int bx = blockIdx.x, by = blockIdx.y;
    if (only_slow == 2) {
        const int nbx = (g.lnx + 127) / 128, nby = (g.lnz + 3) / 4;
        int q = blockIdx.x;
        if (q < 3 * nbx) { bx = q % nbx; by = q / nbx == 0 ? 0 : nby - (q / nbx); }
        else if ((q -= 3 * nbx) < 3 * nby) { bx = q / nby == 0 ? 0 : nbx - (q / nby); by = q % nby; }     // column nx-2 can sit in the last but one
        else { q -= 3 * nby; bx = (anchor_j - g.gj0) / 128; by = (anchor_i + q - g.gi0) / 4; }
        if (bx < 0 || bx >= nbx || by < 0 || by >= nby) return;
    }
    const int lj0 = (bx * 64 + lane) * 2;
    const int li = by * 4 + threadIdx.y;
    if (li >= g.lnz) return;                                // wave-uniform
    const bool active = lj0 < g.lnx;
    const bool has_right = (lj0 + 2) < g.lnx;
    const int p = g.pitch, nz = g.nz, nx = g.nx;
    const long long PLN = g.plane;
    const int c = (int)pl_idx(g, li, lj0);
    const int i = g.gi0 + li, j0 = g.gj0 + lj0;
    const int jw = g.gj0 + bx * 128;
    if (only_slow && stage1_fast_wave(g, i, jw, bx, anchor_i, anchor_j, wall)) return;
    // walls, slaves, stabilised rows, or a wave that sticks out of the block (wave-uniform)
    if (only_slow || i <= 0 || i >= nz - 2 || jw < 1 || jw + 127 > nx - 3 || op.szz || bx * 128 + 127 >= g.lnx) {
        if (!active) return;
        for (int q = 0; q < 2 && lj0 + q < g.lnx; q++) cheb_first_node(op, f, out, c2, i, j0 + q, c + q);
        return;
    }
#define ROW(ptr, dr, w, e) load_row2((ptr) + (long long)c - lj0 + (long long)(dr) * p, lj0, true, w, e, lane, has_right)
    const Row2T<T> en_s = ROW(op.etan, -1, false, false), en_i = ROW(op.etan, 0, true, false);
    const Row2T<T> es_i = ROW(op.etas, 0, false, true), es_n = ROW(op.etas, 1, false, false);
#undef ROW
    const Row2T<T> t_rdx = load_row2(op.rdx + PL_TOFF + g.gj0, lj0, true, true, false, lane, has_right);
    const Row2T<T> t_rDx = load_row2(op.rDx + PL_TOFF + g.gj0, lj0, true, false, true, lane, has_right);
    const VF2 fz_s = *reinterpret_cast<const VF2*>(f + c), fx_s = *reinterpret_cast<const VF2*>(f + PLN + c);
    const typename PlVec2<T>::type fz = PlVec2<T>::make((T)fz_s.x, (T)fz_s.y), fx = PlVec2<T>::make((T)fx_s.x, (T)fx_s.y);
    const T rdz_i = TB(op.rdz, i), rdz_m = TB(op.rdz, i - 1), rDz_i = TB(op.rDz, i), rDz_p = TB(op.rDz, i + 1);
    const T Az = T(4) * rdz_i * rDz_i, Azm = T(4) * rdz_m * rDz_i, r2 = T(2) * rdz_i, nc2 = -c2;
    T oz[2], ox[2];
    {   // column A
        const T rdx_j = t_rdx.v.x, rdx_m = t_rdx.w, rDx_j = t_rDx.v.x, rDx_p = t_rDx.v.y, k2 = T(2) * rdx_j, B4 = T(4) * rDx_j;
        const T dz = en_i.v.x * Az + en_s.v.x * Azm + (es_i.v.y * k2) * rDx_p + (es_i.v.x * k2) * rDx_j;
        const T dx = en_i.v.x * (B4 * rdx_j) + en_i.w * (B4 * rdx_m) + (es_n.v.x * r2) * rDz_p + (es_i.v.x * r2) * rDz_i;
        oz[0] = (nc2 * fz.x) * pl_rcp(dz); ox[0] = (nc2 * fx.x) * pl_rcp(dx);
    }
    {   // column B
        const T rdx_j = t_rdx.v.y, rdx_m = t_rdx.v.x, rDx_j = t_rDx.v.y, rDx_p = t_rDx.e, k2 = T(2) * rdx_j, B4 = T(4) * rDx_j;
        const T dz = en_i.v.y * Az + en_s.v.y * Azm + (es_i.e * k2) * rDx_p + (es_i.v.y * k2) * rDx_j;
        const T dx = en_i.v.y * (B4 * rdx_j) + en_i.v.x * (B4 * rdx_m) + (es_n.v.y * r2) * rDz_p + (es_i.v.y * r2) * rDz_i;
        oz[1] = (nc2 * fz.y) * pl_rcp(dz); ox[1] = (nc2 * fx.y) * pl_rcp(dx);
    }
    *reinterpret_cast<V2*>(out + c) = PlVec2<TV>::make((TV)oz[0], (TV)oz[1]);
    *reinterpret_cast<V2*>(out + PLN + c) = PlVec2<TV>::make((TV)ox[0], (TV)ox[1]);
}

static inline dim3 pl_grid_rows2(const PlGeom& g) { return dim3((g.lnx + 127) / 128, (g.lnz + 3) / 4); }

// y = D^-1 A v with closure (power iteration for lambda_max)
__global__ __launch_bounds__(256) void k_vv_dinv_apply(PlVvOp op, const double* __restrict__ v,
                                                       double* __restrict__ y) {
    PL_NODE_PROLOGUE(op.g)
    const long long P = op.g.plane;
    int moff = 0; double s = 1.0, Av, dg;
    int cls = vv_cls_z(op, i, j, moff, s);
    double o = 0.0;
    if (cls != VV_ZERO) { vv_row_z(op, v, v + P, (int)(c + moff), i, j + moff, Av, dg); o = s * Av / dg; }
    y[c] = o;
    cls = vv_cls_x(op, i, j, moff, s);
    o = 0.0;
    if (cls != VV_ZERO) {
        const int im = i + (moff > 0 ? 1 : (moff < 0 ? -1 : 0));
        vv_row_x(op, v, v + P, (int)(c + moff), im, j, Av, dg); o = s * Av / dg;
    }
    y[c + P] = o;
}

// Full-weighting restriction of the velocity residual (uniform-grid weights):
// vz is vertex-centred in z [1/4,1/2,1/4] and cell-centred in x [1/8,3/8,3/8,1/8]; vx mirrored.
// TF / TC: types of the fine residual and of the coarse right-hand side (an FP32 level above an FP64 one converts here)
// cscale: 1, or sigma where an FP32 level (operator A/sigma) hands its residual to an FP64 one (operator A)
template <typename TF, typename TC>
__device__ inline void restrict_node(const PlGeom& gf, const PlVvOpT<TC>& opc, const TF* __restrict__ rf,
                                     TC* __restrict__ fc, int i, int j, long long c, TC cscale = TC(1)) {
    int moff; TC s;
    const int pf = gf.pitch;
    TF oz = TF(0), ox = TF(0);
    if (vv_cls_z(opc, i, j, moff, s) == VV_INT) {
        const long long b = pl_idx(gf, 2 * i - gf.gi0, 2 * j - gf.gj0);
        const TF wz[3] = {TF(0.25), TF(0.5), TF(0.25)}, wx[4] = {TF(0.125), TF(0.375), TF(0.375), TF(0.125)};
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int q = 0; q < 4; q++) oz += wz[a] * wx[q] * rf[b + (long long)(a - 1) * pf + (q - 1)];
    }
    if (vv_cls_x(opc, i, j, moff, s) == VV_INT) {
        const long long b = pl_idx(gf, 2 * i - gf.gi0, 2 * j - gf.gj0) + gf.plane;
        const TF wz[4] = {TF(0.125), TF(0.375), TF(0.375), TF(0.125)}, wx[3] = {TF(0.25), TF(0.5), TF(0.25)};
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int q = 0; q < 3; q++) ox += wz[a] * wx[q] * rf[b + (long long)(a - 1) * pf + (q - 1)];
    }
    fc[c] = (TC)oz * cscale; fc[c + opc.g.plane] = (TC)ox * cscale;
}

template <typename TF, typename TC>
__global__ __launch_bounds__(256) void k_vv_restrict(PlGeom gf, PlVvOpT<TC> opc, const TF* __restrict__ rf,
                                                     TC* __restrict__ fc, TC cscale) {
    PL_NODE_PROLOGUE(opc.g)
    restrict_node(gf, opc, rf, fc, i, j, c, cscale);
}

// (P e)(i,j) for vz / vx on the fine grid from the coarse correction e (bilinear; mirror clamp)
template <typename TC>
__device__ inline TC prolong_z_at(const PlGeom& gc, const TC* __restrict__ ez, int i, int j) {
    const int I0 = i >> 1, I1 = (i + 1) >> 1;
    int Jn = j >> 1, Jo = (j & 1) ? Jn + 1 : Jn - 1;
    const int jmax = gc.nx - 2;
    Jn = min(max(Jn, 0), jmax); Jo = min(max(Jo, 0), jmax);
    const int o0 = gc.gi0, o1 = gc.gj0;            // coarse indices are global; the block may be a slab
    const TC a = TC(0.5) * (ez[pl_idx(gc, I0 - o0, Jn - o1)] + ez[pl_idx(gc, I1 - o0, Jn - o1)]);
    const TC b = TC(0.5) * (ez[pl_idx(gc, I0 - o0, Jo - o1)] + ez[pl_idx(gc, I1 - o0, Jo - o1)]);
    return TC(0.75) * a + TC(0.25) * b;
}

template <typename TC>
__device__ inline TC prolong_x_at(const PlGeom& gc, const TC* __restrict__ ex, int i, int j) {
    const int J0 = j >> 1, J1 = (j + 1) >> 1;
    int In = i >> 1, Io = (i & 1) ? In + 1 : In - 1;
    const int imax = gc.nz - 2;
    In = min(max(In, 0), imax); Io = min(max(Io, 0), imax);
    const int o0 = gc.gi0, o1 = gc.gj0;
    const TC a = TC(0.5) * (ex[pl_idx(gc, In - o0, J0 - o1)] + ex[pl_idx(gc, In - o0, J1 - o1)]);
    const TC b = TC(0.5) * (ex[pl_idx(gc, Io - o0, J0 - o1)] + ex[pl_idx(gc, Io - o0, J1 - o1)]);
    return TC(0.75) * a + TC(0.25) * b;
}

template <typename TF, typename TC, typename TV>
__device__ inline void prolong_node(const PlVvOpT<TF>& opf, const PlGeom& gc, const TC* __restrict__ ec,
                                    const TV* __restrict__ vin, TF* __restrict__ vout, int i, int j, long long c) {
    const long long P = opf.g.plane;
    int moff = 0; TF s = TF(1);
    int cls = vv_cls_z(opf, i, j, moff, s);
    TF o = TF(0);
    if (cls != VV_ZERO) o = s * ((TF)vin[c + moff] + (TF)prolong_z_at(gc, ec, i, j + moff));
    vout[c] = o;
    cls = vv_cls_x(opf, i, j, moff, s);
    o = TF(0);
    if (cls != VV_ZERO) {
        const int im = i + (moff > 0 ? 1 : (moff < 0 ? -1 : 0));
        o = s * ((TF)vin[c + moff + P] + (TF)prolong_x_at(gc, ec + gc.plane, im, j));
    }
    vout[c + P] = o;
}

template <typename TF, typename TC, typename TV = TF>
__global__ __launch_bounds__(256) void k_vv_prolong_add(PlVvOpT<TF> opf, PlGeom gc, const TC* __restrict__ ec,
                                                        const TV* __restrict__ vin, TF* __restrict__ vout) {
    PL_NODE_PROLOGUE(opf.g)
    prolong_node(opf, gc, ec, vin, vout, i, j, c);
}

// ---- fused coarse tail: the whole V-cycle for the levels below PL_TAIL_MAX_NODES nodes runs in
// ONE workgroup (1024 threads), stages separated by __syncthreads().  These levels are pure launch
// latency as separate kernels (a few microseconds of work each, ~75 launches per cycle).
#define PL_TAIL_MAX_LEVELS 8
#define PL_TAIL_MAX_NODES (65 * 65)
struct TailLevel { PlVvOp op; double* v[3]; double* f; double* r; double lmax; };
struct TailArgs { int nlev; int nu_pre, nu_post, coarse_sweeps; double ratio; TailLevel L[PL_TAIL_MAX_LEVELS]; };

#define TAIL_FOR_NODES(g)                                                              \
    for (int idx_ = threadIdx.x; idx_ < (g).lnz * (g).lnx; idx_ += blockDim.x)

// zero_guess: buf[0] is taken as zero without being read (first sweep = -c2 f / diag, second without v_prev)
__device__ inline void tail_smooth(const TailLevel& L, double* buf[3], int nsweep, double ratio, bool zero_guess = false) {
    const double lmax = L.lmax, lmin = lmax / ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho_old = 1.0 / sigma;
    const PlGeom& g = L.op.g;
    for (int k = 0; k < nsweep; k++) {
        double c1, c2;
        if (k == 0) { c1 = 0.0; c2 = 1.0 / theta; }
        else { const double rho = 1.0 / (2.0 * sigma - rho_old); c1 = rho * rho_old; c2 = 2.0 * rho / delta; rho_old = rho; }
        TAIL_FOR_NODES(g) {
            const int li = idx_ / g.lnx, lj = idx_ % g.lnx;
            if (k == 0 && zero_guess) cheb_first_node(L.op, L.f, buf[2], c2, g.gi0 + li, g.gj0 + lj, pl_idx(g, li, lj));
            else cheb_node(L.op, buf[0], (k == 1 && zero_guess) ? (const double*)nullptr : buf[1], L.f, buf[2], c1, c2, g.gi0 + li,
                           g.gj0 + lj, pl_idx(g, li, lj));
        }
        __syncthreads();
        double* nxt = buf[2]; buf[2] = buf[1]; buf[1] = buf[0]; buf[0] = nxt;
    }
}

__global__ __launch_bounds__(1024) void k_mg_tail(TailArgs a) {
    double* cur[PL_TAIL_MAX_LEVELS][3];
    // ---- down sweep
    for (int l = 0; l < a.nlev; l++) {
        const TailLevel& L = a.L[l];
        const PlGeom& g = L.op.g;
        double* buf[3] = {L.v[0], L.v[1], L.v[2]};
        const bool coarsest = l == a.nlev - 1;
        if ((coarsest ? a.coarse_sweeps : a.nu_pre) < 1) {             // nothing will write the zero guess: do it here
            TAIL_FOR_NODES(g) { const long long c = pl_idx(g, idx_ / g.lnx, idx_ % g.lnx); buf[0][c] = 0.0; buf[0][c + g.plane] = 0.0; }
            __syncthreads();
        }
        if (coarsest) {
            double ratio = 0.4 * g.nz * g.nx; if (ratio < 30.0) ratio = 30.0;
            tail_smooth(L, buf, a.coarse_sweeps, ratio, true);
        } else {
            tail_smooth(L, buf, a.nu_pre, a.ratio, true);
            TAIL_FOR_NODES(g) { const int li = idx_ / g.lnx, lj = idx_ % g.lnx; residual_node(L.op, buf[0], L.f, L.r, g.gi0 + li, g.gj0 + lj, pl_idx(g, li, lj)); }
            __syncthreads();
            const TailLevel& C = a.L[l + 1];
            const PlGeom& gc = C.op.g;
            TAIL_FOR_NODES(gc) { const int li = idx_ / gc.lnx, lj = idx_ % gc.lnx; restrict_node(g, C.op, L.r, C.f, gc.gi0 + li, gc.gj0 + lj, pl_idx(gc, li, lj)); }
            __syncthreads();
        }
        cur[l][0] = buf[0]; cur[l][1] = buf[1]; cur[l][2] = buf[2];
    }
    // ---- up sweep
    for (int l = a.nlev - 2; l >= 0; l--) {
        const TailLevel& L = a.L[l];
        const PlGeom& g = L.op.g;
        const PlGeom& gc = a.L[l + 1].op.g;
        double* buf[3] = {cur[l][0], cur[l][1], cur[l][2]};
        const double* ec = cur[l + 1][0];
        TAIL_FOR_NODES(g) { const int li = idx_ / g.lnx, lj = idx_ % g.lnx; prolong_node(L.op, gc, ec, buf[0], buf[2], g.gi0 + li, g.gj0 + lj, pl_idx(g, li, lj)); }
        __syncthreads();
        { double* t = buf[0]; buf[0] = buf[2]; buf[2] = t; }
        tail_smooth(L, buf, a.nu_post, a.ratio);
        cur[l][0] = buf[0]; cur[l][1] = buf[1]; cur[l][2] = buf[2];
    }
    // result of the first tail level must be in L[0].v[0]
    if (cur[0][0] != a.L[0].v[0]) {
        const PlGeom& g = a.L[0].op.g;
        const double* src = cur[0][0]; double* dst = a.L[0].v[0];
        TAIL_FOR_NODES(g) { const long long c = pl_idx(g, idx_ / g.lnx, idx_ % g.lnx); dst[c] = src[c]; dst[c + g.plane] = src[c + g.plane]; }
    }
}

// node field (density) by the same [1 2 1]x[1 2 1]/16 stencil
__global__ __launch_bounds__(256) void k_coarsen_node(PlGeom gf, const double* __restrict__ ff, PlGeom gc,
                                                      double* __restrict__ fc) {
    PL_NODE_PROLOGUE(gc)
    double acc = 0.0;
#pragma unroll
    for (int a = -1; a <= 1; a++)
#pragma unroll
        for (int q = -1; q <= 1; q++) {
            const int fi = min(max(2 * i + a, 0), gf.nz - 1), fj = min(max(2 * j + q, 0), gf.nx - 1);
            acc += (a == 0 ? 2.0 : 1.0) * (q == 0 ? 2.0 : 1.0) * ff[pl_idx(gf, fi - gf.gi0, fj - gf.gj0)];
        }
    fc[c] = acc * (1.0 / 16.0);
}

// stabilisation coefficients of the interior z-momentum rows of one level (rediscretised on its own grid):
//   szz = coef * 1/2 (rho[i+1,j] + rho[i+1,j+1] - rho[i-1,j] - rho[i-1,j+1]) / (z[i+1]-z[i-1])
//   szx = coef * 1/2 (rho[i,j+1] + rho[i+1,j+1] - rho[i,j-1] - rho[i+1,j-1]) / (x[j+1]-x[j-1]),  coef = theta dt G[IZ]
__global__ __launch_bounds__(256) void k_stab_coeffs(PlVvOp op, const double* __restrict__ rho, double coef,
                                                     double* __restrict__ szz, double* __restrict__ szx) {
    PL_NODE_PROLOGUE(op.g)
    int moff; double s;
    double a = 0.0, b = 0.0;
    if (vv_cls_z(op, i, j, moff, s) == VV_INT) {
        const int p = op.g.pitch;
        const int jm = (j > 0) ? -1 : 0, jp = (j < op.g.nx - 1) ? 1 : 0;      // natural coarse rows reach the wall column
        const double hx = (jm && jp) ? TB(op.g.rDx, j) : TB(op.g.rdx, j);     // one-sided at the wall
        a = coef * 0.5 * (rho[c + p] + rho[c + p + jp] - rho[c - p] - rho[c - p + jp]) * TB(op.g.rDz, i);
        b = coef * 0.5 * (rho[c + jp] + rho[c + p + jp] - rho[c + jm] - rho[c + p + jm]) * hx;
    }
    szz[c] = a; szx[c] = b;
}

// arithmetic viscosity coarsening: nodes by a [1 2 1]x[1 2 1]/16 stencil (edge-clamped),
// cell centres by the mean of the 4 covered fine cells
// (Round 4, other means measured at 2049^2 -- Stokes iterations of the falling block (viscosity x 1e3) | the mantle model:
//  arithmetic 30 | 11, geometric 41 | 11, harmonic 89 | 12, quadratic 41 | 12, quartic 50 | 12: the arithmetic mean is the optimum of the family.)
__global__ __launch_bounds__(256) void k_coarsen_visc(PlGeom gf, const double* __restrict__ esf,
                                                      const double* __restrict__ enf, PlGeom gc,
                                                      double* __restrict__ esc, double* __restrict__ enc) {
    PL_NODE_PROLOGUE(gc)
    auto fw = [&](double v) { return v; };
    auto bw = [&](double v) { return v; };
    double acc = 0.0;
#pragma unroll
    for (int a = -1; a <= 1; a++)
#pragma unroll
        for (int q = -1; q <= 1; q++) {
            const int fi = min(max(2 * i + a, 0), gf.nz - 1), fj = min(max(2 * j + q, 0), gf.nx - 1);
            acc += (a == 0 ? 2.0 : 1.0) * (q == 0 ? 2.0 : 1.0) * fw(esf[pl_idx(gf, fi - gf.gi0, fj - gf.gj0)]);
        }
    esc[c] = bw(acc * (1.0 / 16.0));
    const int ci = min(i, gc.nz - 2), cj = min(j, gc.nx - 2);      // ghost row/col copies its neighbour
    const long long b = pl_idx(gf, 2 * ci - gf.gi0, 2 * cj - gf.gj0);
    enc[c] = bw(0.25 * (fw(enf[b]) + fw(enf[b + 1]) + fw(enf[b + gf.pitch]) + fw(enf[b + gf.pitch + 1])));
}

// =========================================================================================
// Vector kernels (nplanes planes; dots over interior nodes only)
// =========================================================================================
// Block partials of two dot products over the interior nodes of nplanes planes:
// part[2*b] = sum a.b, part[2*b+1] = sum c.d (either pair may be NULL).  Grid-stride over rows,
// no atomics; the host adds the (<= DOT_BLOCKS) partials in a fixed order, so the result is
// deterministic.
#define DOT_BLOCKS 1024
#define PL_SCAL_N 64            // device scalars in front of the dot partials (PlSolver::scal); 32..34: y.r, y.p, y.s of the lazy deflation;
                                // several ranks (lazy deflation riding in the two reductions): 35 r~.Aw, 36 this rank's share of the five-cell sum,
                                // 37 Aw.Aw, 38 the same over the continuity plane, 40..43 Aw.s, Aw.t and their continuity parts, 44..63 all-reduce staging
#define PL_PART_N 16            // doubles per block of the partial-sum area behind the scalars
template <bool HAS_A, bool HAS_C>
__global__ __launch_bounds__(256) void k_dot2(PlGeom g, int nplanes, const double* __restrict__ a,
                                              const double* __restrict__ b, const double* __restrict__ cc,
                                              const double* __restrict__ dd, double* __restrict__ part) {
    double s0 = 0.0, s1 = 0.0;
    const long long rows = (long long)g.lnz * nplanes;
    const int npair = g.lnx >> 1;                              // columns (2k, 2k+1) as one 16-byte load; rows start 16-B aligned
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int q = (int)(r / g.lnz), li = (int)(r % g.lnz);
        const long long base = pl_idx(g, li, 0) + q * g.plane;
#pragma unroll 4
        for (int k = threadIdx.x; k < npair; k += 256) {
            const long long o = base + 2 * k;
            if (HAS_A) {
                const double2 x = *reinterpret_cast<const double2*>(a + o), y = *reinterpret_cast<const double2*>(b + o);
                s0 += x.x * y.x + x.y * y.y;
            }
            if (HAS_C) {
                const double2 x = *reinterpret_cast<const double2*>(cc + o), y = *reinterpret_cast<const double2*>(dd + o);
                s1 += x.x * y.x + x.y * y.y;
            }
        }
        if ((g.lnx & 1) && threadIdx.x == 0) {
            const long long o = base + g.lnx - 1;
            if (HAS_A) s0 += a[o] * b[o];
            if (HAS_C) s1 += cc[o] * dd[o];
        }
    }
    __shared__ double sh[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o, 64); s1 += __shfl_down(s1, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        part[2 * blockIdx.x + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}

// The five sums BiCGStab needs at its second reduction point, in ONE pass over t, s and r~:
//   part[5b..] = (t.s, t.t, r~.s, r~.t, s.s)
// from which omega = ts/tt and -- algebraically, r = s - omega t -- rho' = r~.r = r~.s - omega r~.t and
// |r|^2 = s.s - 2 omega t.s + omega^2 t.t follow without another pass over r (and without a third all-reduce).
__global__ __launch_bounds__(256) void k_dot5(PlGeom g, int nplanes, int nsplit, const double* __restrict__ t, const double* __restrict__ sv,
                                              const double* __restrict__ rt, double* __restrict__ part) {
    // a5..a7: t.s, t.t, s.s over the planes >= nsplit only (the continuity block of the Stokes residual, see bicgstab)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
    const long long rows = (long long)g.lnz * nplanes;
    const int npair = g.lnx >> 1;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int q = (int)(r / g.lnz), li = (int)(r % g.lnz);
        const long long base = pl_idx(g, li, 0) + q * g.plane;
        double b0 = 0.0, b1 = 0.0, b4 = 0.0;                    // this row's t.s, t.t, s.s
#pragma unroll 2
        for (int k = threadIdx.x; k < npair; k += 256) {
            const long long o = base + 2 * k;
            const double2 T = *reinterpret_cast<const double2*>(t + o), Sv = *reinterpret_cast<const double2*>(sv + o),
                          Rt = *reinterpret_cast<const double2*>(rt + o);
            b0 += T.x * Sv.x + T.y * Sv.y; b1 += T.x * T.x + T.y * T.y; a2 += Rt.x * Sv.x + Rt.y * Sv.y;
            a3 += Rt.x * T.x + Rt.y * T.y; b4 += Sv.x * Sv.x + Sv.y * Sv.y;
        }
        if ((g.lnx & 1) && threadIdx.x == 0) {
            const long long o = base + g.lnx - 1;
            b0 += t[o] * sv[o]; b1 += t[o] * t[o]; a2 += rt[o] * sv[o]; a3 += rt[o] * t[o]; b4 += sv[o] * sv[o];
        }
        a0 += b0; a1 += b1; a4 += b4;
        if (q >= nsplit) { a5 += b0; a6 += b1; a7 += b4; }      // row-uniform
    }
    __shared__ double sh[8][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_down(a0, o, 64); a1 += __shfl_down(a1, o, 64); a2 += __shfl_down(a2, o, 64);
        a3 += __shfl_down(a3, o, 64); a4 += __shfl_down(a4, o, 64); a5 += __shfl_down(a5, o, 64);
        a6 += __shfl_down(a6, o, 64); a7 += __shfl_down(a7, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        sh[0][w] = a0; sh[1][w] = a1; sh[2][w] = a2; sh[3][w] = a3; sh[4][w] = a4; sh[5][w] = a5; sh[6][w] = a6; sh[7][w] = a7;
    }
    __syncthreads();
    if (threadIdx.x < 8) part[8 * blockIdx.x + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}
// sums of the k_dot5 partials -> out[8..15] (8..12: t.s, t.t, rt.s, rt.t, s.s; 13..15: t.s, t.t, s.s of the continuity planes); derive = 1: also omega -> out[3], rho' -> out[5], |r|^2 -> out[6], next beta -> out[7]
// (out[2] = alpha and out[4] = rho were left there by the first reduction point of the iteration)
__device__ inline void bicg_derive(double* __restrict__ out) {
    const double ts = out[8], tt = out[9], rts = out[10], rtt = out[11], ss = out[12];
    const double om = (tt > 0.0) ? ts / tt : 0.0;       // t = 0: s is already the residual
    out[3] = om; out[5] = rts - om * rtt;
    const double rr = ss - 2.0 * om * ts + om * om * tt;
    out[6] = rr > 0.0 ? rr : 0.0;
    out[7] = (out[5] / out[4]) * (out[2] / om);          // beta of the NEXT iteration: (rho' / rho) (alpha / omega)
    // lazy deflation (k_defl_coef_lazy): y.v = y.p and y.t = y.s hold by construction of the corrected preconditioner, hence
    // y.r' = y.s - omega y.t = (1 - omega) y.s  and  y.p' = y.r' + beta (y.p - omega y.v) = y.r' + beta (1 - omega) y.p
    out[32] = (1.0 - om) * out[34];
    out[33] = out[32] + out[7] * (1.0 - om) * out[33];
}
__global__ __launch_bounds__(256) void k_sum_partials5(int nb, const double* __restrict__ part, double* __restrict__ out, int derive) {
    __shared__ double sh[8][4];
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k = threadIdx.x; k < nb; k += 256)
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] += part[8 * k + q];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        for (int o = 32; o > 0; o >>= 1) a[q] += __shfl_down(a[q], o, 64);
        if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = a[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 0; q < 8; q++) out[8 + q] = sh[q][0] + sh[q][1] + sh[q][2] + sh[q][3];
        if (derive) bicg_derive(out);
    }
}
__global__ void k_bicg_derive(double* __restrict__ out) { bicg_derive(out); }

// ---- lazy deflation on SEVERAL ranks: the coefficients ride in the two reductions of the iteration --------------------------------
// One rank computes c = (y.in - y.(A x)) / y.(A w) right after the preconditioner (k_defl_coef_lazy) and lets the operator add c A w.
// On several ranks y.(A x) -- minus the area-weighted divergence of x in the anchor cell and the four corner cells -- lives on up to
// four ranks, and an all-reduce of its own per application (two per iteration) is what that would cost.  Instead the operator is
// applied to the UNCORRECTED vector and everything that follows is corrected algebraically, with the five-cell share added to the
// all-reduce that follows anyway:
//   first reduction:   r~.v_u and five(y_u)      ->  c_y = (y.p + five) / y.Aw,   r~.v = r~.v_u + c_y r~.Aw,   alpha = rho / r~.v
//                      s = r - alpha (v_u + c_y Aw);   v = v_u + c_y Aw is written in the same pass (k_s_update_defl)
//   second reduction:  the eight sums on t_u, four more with Aw, five(z_u)  ->  c_z = (y.s + five) / y.Aw and
//                      t.s = t_u.s + c_z Aw.s,  t.t = t_u.t_u + 2 c_z Aw.t_u + c_z^2 Aw.Aw,  r~.t = r~.t_u + c_z r~.Aw   (k_bicg_unpack_defl)
//                      r = s - omega (t_u + c_z Aw),  x += alpha y_u + omega z_u + (alpha c_y + omega c_z) w   (k_xrp_update_dev)
// r~.Aw, Aw.Aw (constant during a solve) are computed once per solve.  Two all-reduces per iteration, as without the deflation.
__global__ void k_defl_five_local(PlStokesOp op, const double* __restrict__ x, double* __restrict__ dst);
__global__ void k_set3(double* __restrict__ a, double va, double* __restrict__ b, double vb, double* __restrict__ c, double vc) { *a = va; *b = vb; *c = vc; }
__global__ void k_defl_alpha_ride(double* __restrict__ sc, double rho_new) {
    const double five = sc[1];                                   // summed over the ranks next to r~.v_u
    const double cy = (sc[24] != 0.0 && isfinite(sc[24])) ? (sc[33] + five) / sc[24] : 0.0;
    sc[30] = cy; sc[26] = sc[33];
    const double rtv = sc[0] + cy * sc[35];
    sc[0] = rtv; sc[2] = rho_new / rtv; sc[4] = rho_new;
}
__global__ void k_defl_pack_first(double* __restrict__ sc) { sc[1] = sc[36]; }
// s = r - alpha (v + c_y Aw), v += c_y Aw
__global__ void k_s_update_defl(long long n, double* __restrict__ s, const double* __restrict__ r, double* __restrict__ v, const double* __restrict__ aw,
                                const double* __restrict__ sc) {
    const double alpha = sc[2], cy = sc[30];
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) { const double vv = v[k] + cy * aw[k]; v[k] = vv; s[k] = r[k] - alpha * vv; }
}
// block partials of Aw.s, Aw.t over all planes and over the planes >= nsplit
__global__ __launch_bounds__(256) void k_dot4w(PlGeom g, int nplanes, int nsplit, const double* __restrict__ aw, const double* __restrict__ sv,
                                               const double* __restrict__ t, double* __restrict__ part) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const long long rows = (long long)g.lnz * nplanes;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int q = (int)(r / g.lnz), li = (int)(r % g.lnz);
        const long long base = pl_idx(g, li, 0) + q * g.plane;
        double b0 = 0.0, b1 = 0.0;
        for (int k = threadIdx.x; k < g.lnx; k += 256) { const double w = aw[base + k]; b0 += w * sv[base + k]; b1 += w * t[base + k]; }
        a0 += b0; a1 += b1;
        if (q >= nsplit) { a2 += b0; a3 += b1; }
    }
    __shared__ double sh[4][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_down(a0, o, 64); a1 += __shfl_down(a1, o, 64); a2 += __shfl_down(a2, o, 64); a3 += __shfl_down(a3, o, 64); }
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sh[0][w] = a0; sh[1][w] = a1; sh[2][w] = a2; sh[3][w] = a3; }
    __syncthreads();
    if (threadIdx.x < 4) part[4 * blockIdx.x + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}
__global__ __launch_bounds__(256) void k_sum_partials4(int nb, const double* __restrict__ part, double* __restrict__ out) {
    __shared__ double sh[4][4];
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = threadIdx.x; k < nb; k += 256)
#pragma unroll
        for (int q = 0; q < 4; q++) a[q] += part[4 * k + q];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        for (int o = 32; o > 0; o >>= 1) a[q] += __shfl_down(a[q], o, 64);
        if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = a[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) for (int q = 0; q < 4; q++) out[q] = sh[q][0] + sh[q][1] + sh[q][2] + sh[q][3];
}
// staging block of the second reduction: [0..7] the k_dot5 sums, [8..12] slots 16..20, [13] five(z_u) share, [14..17] the Aw sums
__global__ void k_bicg_pack_defl(double* __restrict__ sc) {
    const int k = threadIdx.x;
    if (k < 13) sc[44 + k] = sc[8 + k];
    else if (k == 13) sc[44 + 13] = sc[36];
    else if (k < 18) sc[44 + k] = sc[40 + (k - 14)];
}
__global__ void k_bicg_unpack_defl(double* __restrict__ sc) {
    for (int k = 0; k < 13; k++) sc[8 + k] = sc[44 + k];
    const double five = sc[44 + 13], aws = sc[44 + 14], awt = sc[44 + 15], aws_c = sc[44 + 16], awt_c = sc[44 + 17];
    const double ys = sc[32] - sc[2] * sc[33];                   // y.s = y.r - alpha y.v, y.v = y.p
    sc[34] = ys; sc[26] = ys;
    const double cz = (sc[24] != 0.0 && isfinite(sc[24])) ? (ys + five) / sc[24] : 0.0;
    sc[31] = cz;
    sc[8] += cz * aws;                                           // t.s
    sc[9] += 2.0 * cz * awt + cz * cz * sc[37];                  // t.t
    sc[11] += cz * sc[35];                                       // r~.t
    sc[13] += cz * aws_c;                                        // continuity parts
    sc[14] += 2.0 * cz * awt_c + cz * cz * sc[38];
    bicg_derive(sc);
}

// p = r + beta (p - omega v)
__global__ void k_p_update(long long n, double* __restrict__ p, const double* __restrict__ r,
                           const double* __restrict__ v, double beta, double omega) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) p[k] = r[k] + beta * (p[k] - omega * v[k]);
}
// y = a + alpha b
__global__ void k_axpy_out(long long n, double* __restrict__ y, const double* __restrict__ a,
                           const double* __restrict__ b, double alpha) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) y[k] = a[k] + alpha * b[k];
}
// x += alpha y + omega z ; r = s - omega t
__global__ void k_xr_update(long long n, double* __restrict__ x, const double* __restrict__ y,
                            const double* __restrict__ z, double* __restrict__ r, const double* __restrict__ s,
                            const double* __restrict__ t, double alpha, double omega) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) {
        x[k] += alpha * y[k] + omega * z[k];
        r[k] = s[k] - omega * t[k];
    }
}
// x += d
__global__ void k_add_inplace(long long n, double* __restrict__ x, const double* __restrict__ d) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) x[k] += d[k];
}
__global__ void k_scale(long long n, double* __restrict__ x, double a) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) x[k] *= a;
}
// seeded pseudo-random values in [-1,1) on interior nodes (shadow residual, power iteration)
__global__ __launch_bounds__(256) void k_random_interior(PlGeom g, int nplanes, double* __restrict__ v, unsigned seed) {
    PL_NODE_PROLOGUE(g)
    for (int q = 0; q < nplanes; q++) {
        unsigned h = (unsigned)(((unsigned)i * 73856093u) ^ ((unsigned)j * 19349663u) ^ ((unsigned)(q + 1) * 83492791u)) ^ seed;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        v[c + q * g.plane] = (double)h * (2.0 / 4294967296.0) - 1.0;
    }
}

static dim3 grid2d(const PlGeom& g) { return dim3((g.lnx + 63) / 64, (g.lnz + 3) / 4); }
static dim3 grid1d(long long n) { long long b = (n + 255) / 256; return dim3((unsigned)(b > 4096 ? 4096 : b)); }

// =========================================================================================
// Stokes row scaling and preconditioner glue kernels
// =========================================================================================
// Row scale factors D_r of the full Stokes system at node (i,j) (positive).
__device__ inline void stokes_row_scales(const PlStokesOp& op, int i, int j, long long c, double& sz, double& sx,
                                         double& sp) {
    const PlGeom& g = op.g;
    const int nz = g.nz, nx = g.nx, p = g.pitch;
    const double iKc = 1.0 / op.Kc;
    sz = iKc; sx = iKc; sp = iKc;
    if (i >= 1 && i <= nz - 2 && j >= 1 && j <= nx - 3) {
        const double rdz_i = TB(g.rdz, i), rdz_m = TB(g.rdz, i - 1), rDz_i = TB(g.rDz, i);
        const double rdx_j = TB(g.rdx, j), rDx_j = TB(g.rDx, j), rDx_p = TB(g.rDx, j + 1);
        sz = 1.0 / (4.0 * op.etan[c] * rdz_i * rDz_i + 4.0 * op.etan[c - p] * rdz_m * rDz_i +
                    2.0 * op.etas[c + 1] * rDx_p * rdx_j + 2.0 * op.etas[c] * rDx_j * rdx_j);
    }
    if (i >= 1 && i <= nz - 3 && j >= 1 && j <= nx - 2) {
        const double rdx_j = TB(g.rdx, j), rdx_m = TB(g.rdx, j - 1), rDx_j = TB(g.rDx, j);
        const double rdz_i = TB(g.rdz, i), rDz_i = TB(g.rDz, i), rDz_p = TB(g.rDz, i + 1);
        sx = 1.0 / (4.0 * op.etan[c] * rdx_j * rDx_j + 4.0 * op.etan[c - 1] * rdx_m * rDx_j +
                    2.0 * op.etas[c + p] * rDz_p * rdz_i + 2.0 * op.etas[c] * rDz_i * rdz_i);
    }
    if (i <= nz - 2 && j <= nx - 2 && !(i == op.anchor_i && j == op.anchor_j)) {
        if ((i == 0 || i == nz - 2) && (j == 0 || j == nx - 2)) sp = 1.0 / op.Kb;
        else sp = 1.0 / (op.Kc * (TB(g.rdx, j) + TB(g.rdz, i)));
    }
}

__global__ __launch_bounds__(256) void k_stokes_scale_rows(PlStokesOp op, double* __restrict__ v) {
    PL_NODE_PROLOGUE(op.g)
    double sz, sx, sp;
    stokes_row_scales(op, i, j, c, sz, sx, sp);
    v[c] *= sz; v[c + op.g.plane] *= sx; v[c + 2 * op.g.plane] *= sp;
}

// out_r = D_r^-1 (b - t), out_b = D_r^-1 b for scaled vectors b, t (out_b may alias t)
__global__ __launch_bounds__(256) void k_stokes_unscaled_pair(PlStokesOp op, const double* __restrict__ b, const double* __restrict__ t,
                                                              double* __restrict__ out_r, double* __restrict__ out_b) {
    PL_NODE_PROLOGUE(op.g)
    double sz, sx, sp;
    stokes_row_scales(op, i, j, c, sz, sx, sp);
    const long long P = op.g.plane;
    const double b0 = b[c], b1 = b[c + P], b2 = b[c + 2 * P], t0 = t[c], t1 = t[c + P], t2 = t[c + 2 * P];
    out_r[c] = (b0 - t0) / sz; out_r[c + P] = (b1 - t1) / sx; out_r[c + 2 * P] = (b2 - t2) / sp;
    out_b[c] = b0 / sz; out_b[c + P] = b1 / sx; out_b[c + 2 * P] = b2 / sp;
}

// S^ solve of one pressure node from the SCALED residual rs (continuity rows only)
__device__ inline double prec_p_cont(const PlStokesOp& op, const double* __restrict__ rs_p, int i, int j, long long c) {
    // unscale = Kc (1/dx + 1/dz);  S^-1 = eta_n / Kc^2   ->   rs * (1/dx + 1/dz) * eta_n / Kc
    return rs_p[c] * (TB(op.g.rdx, j) + TB(op.g.rdz, i)) * op.etan[c] * op.iKc;
}

// z_p = S^-1 r_p at one pressure node, from the SCALED residual (ghost/anchor rows: r/Kc = rs;
// corners: P_c = P_nb - r/Kb; continuity rows: r eta_n / Kc^2)
// Wall rows of the pressure Schur complement (round 4).  Isoviscous, S = A_pp - A_pv A_vv^-1 A_vp is EXACTLY
// Kc^2 / (2 eta) I in the bulk whatever the cell shape, but in the two cell columns next to an x-wall -- where the reference slaves the
// tangential velocity to its inner neighbour -- its rows are nonlocal along the wall and its eigenvalues go down to (dz/dx)^2 / 2 of
// that: on cells a = dx/dz times wider than high the diagonal S^ leaves O(wall length) eigenvalues of S^-1 S near 1/a^2 (isoviscous
// 65 x 17 nodes: 25 iterations instead of 11).  The INVERSE of the wall block, however, is local -- measured on dense Schur complements
// (tools/schur_spectrum.py ...), in units of eta / Kc^2 with r the unscaled continuity residual, wall column 0 and its neighbour 1:
//     z_0(i) = 1.45 r_0 - 0.225 r_1 - (a^2 / 4) [d(i-1) - 2 d(i) + d(i+1)],   d = r_0 - r_1;        z_1(i) = 0.25 r_0 + 0.775 r_1
// (the bulk value being 1 in these units: the library's S^-1 is half the exact inverse throughout).  The same with rows and columns
// swapped next to the z-walls on cells higher than wide.  Used where the local aspect ratio is at least 2 (op.wall_ps, one rank).
__device__ inline double prec_p_rhat(const PlStokesOp& op, const double* __restrict__ rs_p, int i, int j, long long c) {
    return rs_p[c] * (TB(op.g.rdx, j) + TB(op.g.rdz, i)) * op.iKc;              // prec_p_cont without the viscosity
}
__device__ inline bool prec_p_is_cont(const PlStokesOp& op, int i, int j) {         // a plain continuity row
    const int nz = op.g.nz, nx = op.g.nx;
    return i >= 0 && j >= 0 && i <= nz - 2 && j <= nx - 2 && !(i == op.anchor_i && j == op.anchor_j) && !((i == 0 || i == nz - 2) && (j == 0 || j == nx - 2));
}
__device__ inline double prec_p_value(const PlStokesOp& op, const double* __restrict__ rs_p, int i, int j, long long c) {
    const int nz = op.g.nz, nx = op.g.nx;
    if (i >= nz - 1 || j >= nx - 1 || (i == op.anchor_i && j == op.anchor_j)) return rs_p[c];
    if ((i == 0 || i == nz - 2) && j == 0) return prec_p_cont(op, rs_p, i, 1, c + 1) - rs_p[c];
    if ((i == 0 || i == nz - 2) && j == nx - 2) return prec_p_cont(op, rs_p, i, nx - 3, c - 1) - rs_p[c];
    if (op.wall_ps && nz >= 9 && nx >= 9) {
        const double a = TB(op.g.rdz, i) / TB(op.g.rdx, j);                  // dx / dz of this cell
        const int p = op.g.pitch;
        // x-walls: columns 0, 1 and nx-2, nx-3 (dj: towards the interior)
        if (a >= 2.0 && (j <= 1 || j >= nx - 3)) {
            const int dj = j <= 1 ? 1 : -1, jw = (j == 0 || j == nx - 2) ? j : j - dj, ji = jw + dj;
            const long long cw = c + (jw - j), ci = c + (ji - j);
            if (prec_p_is_cont(op, i, jw) && prec_p_is_cont(op, i, ji)) {
                const double r0 = prec_p_rhat(op, rs_p, i, jw, cw), r1 = prec_p_rhat(op, rs_p, i, ji, ci);
                if (j != jw) return op.etan[c] * (0.25 * r0 + 0.775 * r1);
                const double d0 = r0 - r1;
                double d2 = 0.0;
                if (prec_p_is_cont(op, i - 1, jw) && prec_p_is_cont(op, i - 1, ji))
                    d2 += prec_p_rhat(op, rs_p, i - 1, jw, cw - p) - prec_p_rhat(op, rs_p, i - 1, ji, ci - p) - d0;
                if (prec_p_is_cont(op, i + 1, jw) && prec_p_is_cont(op, i + 1, ji))
                    d2 += prec_p_rhat(op, rs_p, i + 1, jw, cw + p) - prec_p_rhat(op, rs_p, i + 1, ji, ci + p) - d0;
                return op.etan[c] * (1.45 * r0 - 0.225 * r1 - 0.25 * a * a * d2);
            }
        }
        // z-walls: rows 0, 1 and nz-2, nz-3.  Only up to 6:1: measured at 16:1 (65 x 1025 nodes, x-line smoothing) the iteration with
        // V(2,2) does not profit (isoviscous 30 -> 30, variable viscosity 62 -> 84; with V(3,3) 14 / 43, slower in time than without) --
        // unlike the x-wall case (1025 x 65: 84 -> 13 and 140 -> 34) -- for reasons that were not found
        if (a <= 0.5 && a > 1.0 / 6.0 && (i <= 1 || i >= nz - 3)) {
            const int di = i <= 1 ? 1 : -1, iw = (i == 0 || i == nz - 2) ? i : i - di, ii = iw + di;
            const long long cw = c + (long long)(iw - i) * p, ci = c + (long long)(ii - i) * p;
            if (prec_p_is_cont(op, iw, j) && prec_p_is_cont(op, ii, j)) {
                const double r0 = prec_p_rhat(op, rs_p, iw, j, cw), r1 = prec_p_rhat(op, rs_p, ii, j, ci);
                if (i != iw) return op.etan[c] * (0.25 * r0 + 0.775 * r1);
                const double d0 = r0 - r1;
                double d2 = 0.0;
                if (prec_p_is_cont(op, iw, j - 1) && prec_p_is_cont(op, ii, j - 1))
                    d2 += prec_p_rhat(op, rs_p, iw, j - 1, cw - 1) - prec_p_rhat(op, rs_p, ii, j - 1, ci - 1) - d0;
                if (prec_p_is_cont(op, iw, j + 1) && prec_p_is_cont(op, ii, j + 1))
                    d2 += prec_p_rhat(op, rs_p, iw, j + 1, cw + 1) - prec_p_rhat(op, rs_p, ii, j + 1, ci + 1) - d0;
                return op.etan[c] * (1.45 * r0 - 0.225 * r1 - 0.25 / (a * a) * d2);
            }
        }
    }
    return prec_p_cont(op, rs_p, i, j, c);
}

// First stage of z = M^-1 rs:  z_p = S^-1 r_p  and the velocity right-hand side
// f = r_v - A_vp z_p on the interior momentum rows (0 on constraint rows).
// Constraint-row residuals are NOT lifted: inside the Krylov iteration they are identically zero,
// because x0 is closed with k_close_constraints and every preconditioned direction satisfies the
// homogeneous wall/slave rows, so (A y)_constraint = 0 for all iterates.
// TF: type of the velocity right-hand side f (float on an FP32 level 0: f is written times fscale, see stokes_precond)
template <typename TF>
__device__ inline void stage1_node(const PlStokesOp& op, const PlVvOpT<TF>& vop, const double* __restrict__ rs,
                                   double* __restrict__ z, TF* __restrict__ f, int li, int lj, double fscale) {
    const int i = op.g.gi0 + li, j = op.g.gj0 + lj;
    const long long c = pl_idx(op.g, li, lj);
    const long long P = op.g.plane;
    const int p = op.g.pitch;
    const double* rs_p = rs + 2 * P;
    const double zp_c = prec_p_value(op, rs_p, i, j, c);
    int moff; TF s;
    double fz = 0.0, fx = 0.0;
    // un-scaling an interior momentum row = multiplying by the sum of its 4 own-component coefficients
    if (vv_cls_z(vop, i, j, moff, s) == VV_INT) {
        const PlGeom& g = op.g;
        const double rdz_i = TB(g.rdz, i), rdz_m = TB(g.rdz, i - 1), rDz_i = TB(g.rDz, i);
        const double rdx_j = TB(g.rdx, j), rDx_j = TB(g.rDx, j), rDx_p = TB(g.rDx, j + 1);
        const double sum = 4.0 * op.etan[c] * rdz_i * rDz_i + 4.0 * op.etan[c - p] * rdz_m * rDz_i +
                           2.0 * op.etas[c + 1] * rDx_p * rdx_j + 2.0 * op.etas[c] * rDx_j * rdx_j;
        fz = rs[c] * sum + 2.0 * op.Kc * rDz_i * (zp_c - prec_p_value(op, rs_p, i - 1, j, c - p));
    }
    if (vv_cls_x(vop, i, j, moff, s) == VV_INT) {
        const PlGeom& g = op.g;
        const double rdx_j = TB(g.rdx, j), rdx_m = TB(g.rdx, j - 1), rDx_j = TB(g.rDx, j);
        const double rdz_i = TB(g.rdz, i), rDz_i = TB(g.rDz, i), rDz_p = TB(g.rDz, i + 1);
        const double sum = 4.0 * op.etan[c] * rdx_j * rDx_j + 4.0 * op.etan[c - 1] * rdx_m * rDx_j +
                           2.0 * op.etas[c + p] * rDz_p * rdz_i + 2.0 * op.etas[c] * rDz_i * rdz_i;
        fx = rs[c + P] * sum + 2.0 * op.Kc * rDx_j * (zp_c - prec_p_value(op, rs_p, i, j - 1, c - 1));
    }
    f[c] = (TF)(fz * fscale); f[c + P] = (TF)(fx * fscale); z[c + 2 * P] = zp_c;
}

template <typename TF>
__global__ __launch_bounds__(256) void k_prec_stage1(PlStokesOp op, PlVvOpT<TF> vop, const double* __restrict__ rs,
                                                     double* __restrict__ z, TF* __restrict__ f, int iters, double fscale) {
    PL_ROW_LOOP(op.g, iters) stage1_node(op, vop, rs, z, f, li, lj, fscale);
}

// Two columns per lane; interior waves run straight-line code, the others the per-node function above.
// v1 != NULL: the first Chebyshev sweep of level 0 from the zero guess, v1 = -c2 f / diag, is written in the same pass (the
// diagonal sums are the row scales this kernel computes anyway): the separate k_vv_first2 pass over f and the viscosities
// is then only needed for the waves that take the per-node path here (k_vv_first2 with only_slow = 1).
// TF = float (FP32 level 0): f is written times fscale and v1 times v1scale (stokes_precond explains the two factors)
template <typename TF>
__global__ __launch_bounds__(256) void k_prec_stage1_v2(PlStokesOp op, PlVvOpT<TF> vop, const double* __restrict__ rs,
                                                        double* __restrict__ z, TF* __restrict__ f, TF* __restrict__ v1, double c2,
                                                        double fscale, double v1scale) {
    typedef typename PlVec2<TF>::type VF2;
    const PlGeom& g = op.g;
    const int lane = threadIdx.x;
    const int lj0 = (blockIdx.x * 64 + lane) * 2;
    const int li = blockIdx.y * 4 + threadIdx.y;
    if (li >= g.lnz) return;                                // wave-uniform
    const bool has_right = (lj0 + 2) < g.lnx;
    const int p = g.pitch, nz = g.nz, nx = g.nx;
    const long long PLN = g.plane;
    const int c = (int)pl_idx(g, li, lj0);
    const int i = g.gi0 + li;
    const int jw = g.gj0 + blockIdx.x * 128;
    (void)nz; (void)nx;
    if (!stage1_fast_wave(g, i, jw, blockIdx.x, op.anchor_i, op.anchor_j, op.wall_ps)) {     // walls, slaves, anchor, or the wave sticks out of the block
        for (int q = 0; q < 2 && lj0 + q < g.lnx; q++) stage1_node(op, vop, rs, z, f, li, lj0 + q, fscale);
        return;
    }
#define ROW(ptr, dr, w, e) load_row2((ptr) + (long long)c - lj0 + (long long)(dr) * p, lj0, true, w, e, lane, has_right)
    const double2 rz = *reinterpret_cast<const double2*>(rs + c), rx = *reinterpret_cast<const double2*>(rs + PLN + c);
    const Row2 rp_i = ROW(rs + 2 * PLN, 0, true, false), rp_s = ROW(rs + 2 * PLN, -1, false, false);
    const Row2 en_s = ROW(op.etan, -1, false, false), en_i = ROW(op.etan, 0, true, false);
    const Row2 es_i = ROW(op.etas, 0, false, true), es_n = ROW(op.etas, 1, false, false);
#undef ROW
    const Row2 t_rdx = load_row2(g.rdx + PL_TOFF + g.gj0, lj0, true, true, false, lane, has_right);
    const Row2 t_rDx = load_row2(g.rDx + PL_TOFF + g.gj0, lj0, true, false, true, lane, has_right);
    const double rdz_i = TB(g.rdz, i), rdz_m = TB(g.rdz, i - 1), rDz_i = TB(g.rDz, i), rDz_p = TB(g.rDz, i + 1);
    const double Az = 4.0 * rdz_i * rDz_i, Azm = 4.0 * rdz_m * rDz_i, r2 = 2.0 * rdz_i, twoKc = 2.0 * op.Kc, Pz = twoKc * rDz_i;
    const double iKc = op.iKc;
    double fz[2], fx[2], zp[2], dz[2], dx[2];
    {   // column A
        const double rdx_j = t_rdx.v.x, rdx_m = t_rdx.w, rDx_j = t_rDx.v.x, rDx_p = t_rDx.v.y, k2 = 2.0 * rdx_j, B4 = 4.0 * rDx_j;
        zp[0] = rp_i.v.x * (rdx_j + rdz_i) * en_i.v.x * iKc;
        const double zp_s = rp_s.v.x * (rdx_j + rdz_m) * en_s.v.x * iKc, zp_w = rp_i.w * (rdx_m + rdz_i) * en_i.w * iKc;
        const double sz = en_i.v.x * Az + en_s.v.x * Azm + (es_i.v.y * k2) * rDx_p + (es_i.v.x * k2) * rDx_j;
        const double sx = en_i.v.x * (B4 * rdx_j) + en_i.w * (B4 * rdx_m) + (es_n.v.x * r2) * rDz_p + (es_i.v.x * r2) * rDz_i;
        fz[0] = rz.x * sz + Pz * (zp[0] - zp_s);
        fx[0] = rx.x * sx + (twoKc * rDx_j) * (zp[0] - zp_w);
        dz[0] = sz; dx[0] = sx;
    }
    {   // column B (its west neighbour is column A)
        const double rdx_j = t_rdx.v.y, rdx_m = t_rdx.v.x, rDx_j = t_rDx.v.y, rDx_p = t_rDx.e, k2 = 2.0 * rdx_j, B4 = 4.0 * rDx_j;
        zp[1] = rp_i.v.y * (rdx_j + rdz_i) * en_i.v.y * iKc;
        const double zp_s = rp_s.v.y * (rdx_j + rdz_m) * en_s.v.y * iKc;
        const double sz = en_i.v.y * Az + en_s.v.y * Azm + (es_i.e * k2) * rDx_p + (es_i.v.y * k2) * rDx_j;
        const double sx = en_i.v.y * (B4 * rdx_j) + en_i.v.x * (B4 * rdx_m) + (es_n.v.y * r2) * rDz_p + (es_i.v.y * r2) * rDz_i;
        fz[1] = rz.y * sz + Pz * (zp[1] - zp_s);
        fx[1] = rx.y * sx + (twoKc * rDx_j) * (zp[1] - zp[0]);
        dz[1] = sz; dx[1] = sx;
    }
    *reinterpret_cast<VF2*>(f + c) = PlVec2<TF>::make((TF)(fz[0] * fscale), (TF)(fz[1] * fscale));
    *reinterpret_cast<VF2*>(f + PLN + c) = PlVec2<TF>::make((TF)(fx[0] * fscale), (TF)(fx[1] * fscale));
    *reinterpret_cast<double2*>(z + 2 * PLN + c) = make_double2(zp[0], zp[1]);
    if (v1) {                                               // wave-uniform; same expressions as k_vv_first2's interior path
        const double nc2 = -c2 * v1scale;
        *reinterpret_cast<VF2*>(v1 + c) = PlVec2<TF>::make((TF)((nc2 * fz[0]) * pl_rcp(dz[0])), (TF)((nc2 * fz[1]) * pl_rcp(dz[1])));
        *reinterpret_cast<VF2*>(v1 + PLN + c) = PlVec2<TF>::make((TF)((nc2 * fx[0]) * pl_rcp(dx[0])), (TF)((nc2 * fx[1]) * pl_rcp(dx[1])));
    }
}

// Make x satisfy the constraint rows of A x = b exactly (b given SCALED, bs = b / Kc on these rows):
// walls and ghosts x = bs; slaved rows x_s = s x_m + bs / a0.
__global__ __launch_bounds__(256) void k_close_constraints(PlStokesOp op, PlVvOp vop, const double* __restrict__ bs,
                                                           double* __restrict__ x) {
    PL_NODE_PROLOGUE(op.g)
    const long long P = op.g.plane;
    int moff; double s;
    int cls = vv_cls_z(vop, i, j, moff, s);
    if (cls == VV_ZERO) x[c] = bs[c];
    else if (cls == VV_SLAVE) x[c] = s * x[c + moff] + bs[c];
    cls = vv_cls_x(vop, i, j, moff, s);
    if (cls == VV_ZERO) x[c + P] = bs[c + P];
    else if (cls == VV_SLAVE) {
        double g0 = bs[c + P];
        if (i == 0 && op.bc_z0 != PL_BC_FREESLIP) g0 /= (-TB(op.g.rDz, 1) - TB(op.g.rdz, 0));
        if (i == op.g.nz - 2 && op.bc_zL != PL_BC_FREESLIP) g0 /= (TB(op.g.rDz, op.g.nz - 2) + TB(op.g.rdz, op.g.nz - 2));
        x[c + P] = s * x[c + moff + P] + g0;
    }
}

// Hydrostatic pressure guess: with v = 0 the interior z-momentum rows reduce to
//   -2 Kc rDz_i (P[i,j] - P[i-1,j]) = -1/2 (rho[i,j] + rho[i,j+1]) g
// which is integrated down every column (one thread per column), then shifted so that the
// anchor cell is 0.  b - A x_h is the DYNAMIC load the solver's tolerance is measured against.
// The column integral runs in chunks of PL_HYDRO_CHUNK rows (one thread per column and chunk; a single thread per
// column took ~1 ms of every solve at 2049 rows): local prefix, exclusive scan of the chunk totals, add back.
#define PL_HYDRO_CHUNK 64
__global__ __launch_bounds__(64) void k_hydro_chunk(PlStokesOp op, double* __restrict__ x, double* __restrict__ ctot) {
    const PlGeom& g = op.g;
    const int lj = blockIdx.x * 64 + threadIdx.x, ch = blockIdx.y;
    if (lj >= g.lnx) return;
    double* P = x + 2 * g.plane;
    const double* r = op.rho;
    const int l0 = ch * PL_HYDRO_CHUNK, l1 = min(l0 + PL_HYDRO_CHUNK, g.lnz);
    const int jn = (g.gj0 + lj + 1 < g.nx) ? 1 : 0;
    double acc = 0.0;
    for (int li = l0; li < l1; li++) {
        const int i = g.gi0 + li;
        const long long c = pl_idx(g, li, lj);
        if (i >= 1 && i <= g.nz - 2) acc += 0.5 * (r[c] + r[c + jn]) * op.gz / (2.0 * op.Kc * TB(g.rDz, i));
        P[c] = acc;                       // relative to the top of this chunk
    }
    ctot[(long long)ch * g.lnx + lj] = acc;
}
// per column: chunk totals -> exclusive prefix (in place), column total of the slab
__global__ __launch_bounds__(64) void k_hydro_chunk_scan(PlGeom g, int nch, double* __restrict__ ctot, double* __restrict__ coltot) {
    const int lj = blockIdx.x * 64 + threadIdx.x;
    if (lj >= g.lnx) return;
    double acc = 0.0;
    for (int ch = 0; ch < nch; ch++) { const double t = ctot[(long long)ch * g.lnx + lj]; ctot[(long long)ch * g.lnx + lj] = acc; acc += t; }
    coltot[lj] = acc;
}
__global__ __launch_bounds__(256) void k_hydro_add(PlGeom g, double* __restrict__ x, const double* __restrict__ cpre) {
    PL_NODE_PROLOGUE(g)
    (void)i; (void)j;
    x[2 * g.plane + c] += cpre[(long long)(li / PL_HYDRO_CHUNK) * g.lnx + lj];      // relative to the top of this slab
}

// P += prefix[j] (sum of the slabs above) - pa (anchor value); ghosts 0; velocities 0
__global__ __launch_bounds__(256) void k_hydrostatic_apply_shift(PlStokesOp op, double* __restrict__ x,
                                                                 const double* __restrict__ prefix, double pa) {
    PL_NODE_PROLOGUE(op.g)
    const PlGeom& g = op.g;
    double* P = x + 2 * g.plane;
    x[c] = 0.0; x[c + g.plane] = 0.0;
    P[c] = (i >= g.nz - 1 || j >= g.nx - 1) ? 0.0 : P[c] + prefix[lj] - pa;
}

// =========================================================================================
// Host side
// =========================================================================================
struct MgLevel {
    PlGeomHost gh;
    bool dist = false;          // rows decomposed over the ranks (halo exchanges needed)
    PlGeom win{};               // first tail level only: this rank's row window of the GLOBAL arrays
    long long win_shift = 0;     // element offset of the window's node (0,0) inside the replicated planes
    PlVvOp op{};
    double* etas = nullptr; double* etan = nullptr; bool own_visc = false;
    double *rho = nullptr, *szz = nullptr, *szx = nullptr;     // free-surface stabilisation (allocated on first use)
    double* eig = nullptr; bool eig_valid = false;             // dominant eigenvector of D^-1 A from the last solve
    int eig_confirm = 0, eig_skip = 0;                         // consecutive solves that confirmed lmax to 0.2 %; solves since the last refresh
    bool own_rho = false;
    double* v[3] = {nullptr, nullptr, nullptr};      // rotating Chebyshev buffers (2 planes each)
    double *f = nullptr, *r = nullptr;
    double* fe = nullptr;        // early coarse branch: R^l f of the finest level (levels 1 .. early_K-1)
    double lmax = 3.0;
    bool line_z = false;         // line relaxation instead of point Jacobi (k_vv_line): cells of this level at least 6 x wider than high (or
    int line_ax = 0;             // higher than wide) somewhere; line_ax 0: z-lines, 1: x-lines
    // FP32 twin (large levels only, see build_hierarchy): viscosity planes times 1/sigma, the 1-D tables, work planes
    bool f32 = false;
    PlVvOpF opf{};
    float *etas_f = nullptr, *etan_f = nullptr, *tab_f = nullptr;
    float* vf[3] = {nullptr, nullptr, nullptr};
    float *ff = nullptr, *rf = nullptr;
};
template <typename T> struct LevelT;
template <> struct LevelT<double> {
    static const PlVvOp& op(const MgLevel* L) { return L->op; }
    static double** v(MgLevel* L) { return L->v; }
    static double* f(MgLevel* L) { return L->f; }
    static double* r(MgLevel* L) { return L->r; }
};
template <> struct LevelT<float> {
    static const PlVvOpF& op(const MgLevel* L) { return L->opf; }
    static float** v(MgLevel* L) { return L->vf; }
    static float* f(MgLevel* L) { return L->ff; }
    static float* r(MgLevel* L) { return L->rf; }
};

struct PlSolver {
    std::vector<MgLevel*> levels;
    int bc_key[4] = {-1, -1, -1, -1};
    int repl_start = -1;        // multi-rank: first REPLICATED level (all-gathered, solved redundantly by every rank);
                                // the finer levels are distributed row slabs with halo exchanges
    long long repl_max_nodes = 300000;   // PYLAMP_MG_REPL_NODES: a level this small costs less to compute redundantly (two tile kernels
                                         // since round 3) than its halo exchanges per cycle (150 000 until round 4: the 513^2 level of the
                                         // 2049^2 grid was distributed: 4 more exchanges per BiCGStab iteration than one all-gather costs)
    // BiCGStab work vectors (3 planes each)
    double *r = nullptr, *rt = nullptr, *p = nullptr, *v = nullptr, *s = nullptr, *t = nullptr, *y = nullptr,
           *z = nullptr, *b = nullptr, *x = nullptr, *xb = nullptr, *dx = nullptr, *r0 = nullptr, *xh = nullptr;
    double* scal = nullptr;     // device scalars [0..PL_SCAL_N) + dot partials [PL_SCAL_N .. PL_SCAL_N + 8*DOT_BLOCKS)
    double* hpart = nullptr;    // pinned host copy of the dot partials
    int nu_pre = 2, nu_post = 2, coarse_sweeps = 12;
    int nu0_pre = -1, nu0_post = -1;                    // finest level only (PYLAMP_MG_NU0), -1: as the other levels
    bool aniso_auto = true, ratio_knob = false;        // PYLAMP_MG_ANISO=0 / an explicit PYLAMP_MG_RATIO disable the anisotropy rule
    int power_its_warm = 1;                             // PYLAMP_MG_POWER: power iterations when restarting from the last eigenvector, before the
                                                        // 1 % rule may stop them (so at least 2; 3 measured 0.8 ms slower per solve at 2049^2, same iterations)
    bool nu_auto = true;                                // no PYLAMP_MG_NU / PYLAMP_MG_NU0 given: chosen from the grid size
    bool use_tail = true;
    bool any_line = false;       // some level smooths with z-lines: no tile kernels / one-workgroup tail / FP32 storage / fused first sweep there
    long long tail_knob = 0;                            // PYLAMP_MG_TAIL_NODES (0: automatic)
    long long tail_max_nodes = 33 * 33;                 // levels up to this size run in the fused tail kernel
    int min_cells = 4;                                  // coarsest grid has >= min_cells cells per side 
    // halo policy inside the V-cycle on distributed levels: 2 = exchange before every sweep / residual /
    // transfer (identical numerics to one rank); 1 = once per level and direction; 0 = none (slab-local
    // smoothing with frozen zero halos; only the replicated coarse tail couples the slabs)
    int mg_halo = 2;
    // FP32 multigrid levels: OFF by default (PYLAMP_MG_FP32=1 / pl_stokes_set_mg_precision switch them on;
    // PYLAMP_MG_FP32_NODES: smallest level, in local nodes, that is worth it -- the smaller ones are launch-latency bound).
    // Measured at 2049^2: the FP32 sweep takes 40 instead of 75 us, an iteration 1.79 instead of 2.08 ms -- but storing the
    // smooth part of the iterate in FP32 puts high-frequency rounding noise of eps |v| into every preconditioned direction,
    // which A amplifies by (L/h)^2 relative to the signal: eps (L/h)^2 = 0.25 at 2049^2, and BiCGStab needs 41-49 instead of
    // 35-36 iterations (1025^2: 34-38 instead of 33).
    bool f32_enable = false; long long f32_min_nodes = 200000;
    // velocity-error estimate that a converged Stokes solve must meet (bicgstab); PYLAMP_STOKES_ETOL, 0: off.  The estimate is
    // within a factor ~3 of the true error where it was below it (and up to 60 x above it): 3e-8 keeps the error below 1e-7,
    // a tenth of the 1e-6 the drop-in promises, so that quantities derived from maxima (the time step) stay within 1e-6 too
    double etol = 3e-8;
    double kappa = 1.0, sigma = 1.0;     // scaling of the FP32 velocity solve (stokes_precond)
    // Deflation of the pressure-anchor mode (pl_stokes_solve_device): w = A^-1 u for the one residual-space vector u that the
    // block preconditioner cannot treat, kept across the solves of a time loop and refreshed every defl_refresh solves
    bool defl_enable = true;             // PYLAMP_DEFLATE=0 switches it off
    bool defl_persistent = false;        // (informational) set by the time-step driver: consecutive solves of one model
    bool defl_valid = false, defl_active = false;
    double *wdefl = nullptr, *udefl = nullptr;
    double defl_yAw = 0.0, defl_wvel2 = 0.0;     // host copies: y.(A w) and ||w_vel||^2 (the anchor-mode term of the error estimate)
    // Lazy correction (one rank, device scalars): M leaves z = M^-1 r uncorrected and only computes the coefficient c of w (device
    // slot 30 for the vector y, 31 for z); the operator application that follows adds c A w in its epilogue (Awdefl = A w under the
    // current operator) and the iterate update adds (alpha c_y + omega c_z) w -- instead of one 72 B/node axpy per application
    double* Awdefl = nullptr;
    bool defl_lazy = false;
    double schur_scale = 1.0;    // S^ = schur_scale * Kc^2 / eta_n (PYLAMP_SCHUR_SCALE)
    double ref_cached = 0.0, ref_kc = 0.0; int ref_age = 0;      // dynamic-load reference norm of the last solve that computed it (pl_stokes_solve_device)
    long long fused_max_nodes = 1100000;      // PYLAMP_MG_FUSED_MAX: largest level (nodes) that takes the tile kernels
    long long tile32_min_nodes = 1000000;            // PYLAMP_MG_TS32: levels from this many nodes use 32 x 32 tiles (2049^2: level 1; 36.6 against 37.4 ms per step)
    bool fused = true;           // PYLAMP_MG_FUSED=0: every multigrid stage as a kernel of its own (the path the tile kernels are checked against)
    bool fuse_first = true;      // false: the first sweep of level 0 as a pass of its own
    bool l0_mixed = true;        // PYLAMP_L0_MIXED=0: every vector of a staged level 0 in FP64 (stokes_precond_t)
    bool l0_mixed_now = true;    // ... and only for warm-started solves to rtol >= 1e-8 (pl_stokes_solve_device)
    bool deep = true;            // PYLAMP_MG_DEEP=0: distributed levels exchange before every sweep instead of once per smoothing sequence
    int tail_nu_pre = -1, tail_nu_post = -1;        // smoothing sweeps on the replicated tail levels (-1: same as nu)
    double cheb_ratio = 6.0, lmax_safety = 1.1;     // smoothing window [lmax/ratio, lmax]; lmax = safety * power-iteration estimate
    // heat work vectors (1 plane each)
    double* h[11] = {nullptr};
    double* hc3 = nullptr;                    // several ranks: the three Chebyshev iterates of the heat solve, contiguous (one halo exchange)
    double* hc[9] = {nullptr}; double* hc_part = nullptr; int hc_nb = 0, hc_last_its = 0;      // CG work planes (wall entries of d, r, z, p, q stay 0)
    int napply = 0, nprec = 0;
    // Early coarse branch (single rank): the levels >= early_K get R^K f -- the right-hand side itself, restricted K
    // times -- instead of the restricted residual of level K-1, which makes them independent of the pre-smoothing of the
    // fine levels: they run on a second stream, concurrently with it.  The coarse levels are pure launch latency
    // (~235 us of a 730 us preconditioner application at 2049^2 with the GPU idle), the fine ones pure bandwidth.
    // The NumPy prototype needs the same number of BiCGStab iterations either way (tools/early_coarse.py: mantle model,
    // 129^2: 33 / 33 / 32 for K = off / 2 / 3; 257^2: 36 / 37 / 35 / 36 for off / 2 / 3 / 4).
    // On the GPU (2049^2, K = 2 / 3 / 4) the branch costs 2-4 more iterations than the standard cycle (V(1,1) + V(3,3) smooth
    // harder than the prototype's V(2,2), so more of what the coarse levels are asked to remove is already gone), the
    // coarse kernels run 1.5-3 x slower while the bandwidth-bound fine kernels are in flight, and the 100 us tail kernel
    // stays on the critical path: 74-78 ms per solve against 73-74.  OFF by default.
    // PYLAMP_MG_EARLY=K switches it on (-1: automatic K, the first level of <= early_max_nodes nodes on large grids).
    int early_knob = 0; long long early_max_nodes = 300000;
    int early_K = 0;                         // decided by build_hierarchy
    hipStream_t stream2 = nullptr; hipEvent_t ev_f = nullptr, ev_b = nullptr;
};

static PlSolver* solver_of(pl_ctx* ctx) {
    if (!ctx->krylov) {
        PlSolver* S = new PlSolver();
        // tuning knobs (defaults chosen on MI355X at 2049^2): PYLAMP_MG_NU="pre,post", PYLAMP_MG_COARSE=sweeps
        if (const char* e = getenv("PYLAMP_MG_NU")) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 0 && b >= 0 && a + b > 0) { S->nu_pre = a; S->nu_post = b; S->nu_auto = false; } }
        if (const char* e = getenv("PYLAMP_MG_NU0")) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && b >= 1) { S->nu0_pre = a; S->nu0_post = b; S->nu_auto = false; } }
        if (const char* e = getenv("PYLAMP_MG_POWER")) { int a = atoi(e); if (a >= 1) S->power_its_warm = a; }
        if (const char* e = getenv("PYLAMP_MG_COARSE")) { int a = atoi(e); if (a > 0) S->coarse_sweeps = a; }
        if (const char* e = getenv("PYLAMP_MG_TAIL")) S->use_tail = atoi(e) != 0;
        if (const char* e = getenv("PYLAMP_MG_REPL_NODES")) { long long v = atoll(e); if (v >= 25) S->repl_max_nodes = v; }
        if (const char* e = getenv("PYLAMP_MG_TAIL_NODES")) { long long v = atoll(e); if (v >= 25 && v <= PL_TAIL_MAX_NODES) S->tail_knob = v; }
        if (const char* e = getenv("PYLAMP_MG_HALO")) S->mg_halo = atoi(e);
        if (const char* e = getenv("PYLAMP_MG_DEEP")) S->deep = atoi(e) != 0;
        if (const char* e = getenv("PYLAMP_MG_EARLY")) S->early_knob = atoi(e);
        if (const char* e = getenv("PYLAMP_DEFLATE")) S->defl_enable = atoi(e) != 0;
        if (const char* e = getenv("PYLAMP_STOKES_ETOL")) { const double v = atof(e); if (v >= 0.0) S->etol = v; }
        if (const char* e = getenv("PYLAMP_MG_FP32")) S->f32_enable = atoi(e) != 0;
        if (const char* e = getenv("PYLAMP_MG_FP32_NODES")) { long long v = atoll(e); if (v >= 1) S->f32_min_nodes = v; }
        if (const char* e = getenv("PYLAMP_L0_MIXED")) S->l0_mixed = atoi(e) != 0;
        if (const char* e = getenv("PYLAMP_MG_FUSED")) S->fused = atoi(e) != 0;
        if (const char* e = getenv("PYLAMP_MG_TS32")) { long long v = atoll(e); if (v > 0) S->tile32_min_nodes = v; }
        if (const char* e = getenv("PYLAMP_MG_FUSED_MAX")) { long long v = atoll(e); if (v > 0) S->fused_max_nodes = v; }
        if (const char* e = getenv("PYLAMP_SCHUR_SCALE")) { double v = atof(e); if (v > 0.0) S->schur_scale = v; }
        if (const char* e = getenv("PYLAMP_MG_TAIL_NU")) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2 && a + b > 0) { S->tail_nu_pre = a; S->tail_nu_post = b; } }
        if (const char* e = getenv("PYLAMP_MG_RATIO")) { double v = atof(e); if (v > 1.5) { S->cheb_ratio = v; S->ratio_knob = true; } }
        if (const char* e = getenv("PYLAMP_MG_ANISO")) S->aniso_auto = atoi(e) != 0;
        if (const char* e = getenv("PYLAMP_MG_SAFETY")) { double v = atof(e); if (v >= 1.0) S->lmax_safety = v; }
        ctx->krylov = S;
    }
    return (PlSolver*)ctx->krylov;
}

static void free_levels(PlSolver* S) {
    for (MgLevel* L : S->levels) {
        if (L->own_visc) { (void)hipFree(L->etas); (void)hipFree(L->etan); }
        if (L->own_rho && L->rho) (void)hipFree(L->rho);
        for (double* q : {L->szz, L->szx, L->eig}) if (q) (void)hipFree(q);
        for (double* q : {L->v[0], L->v[1], L->v[2], L->f, L->r, L->fe}) if (q) (void)hipFree(q);
        for (float* q : {L->etas_f, L->etan_f, L->tab_f, L->vf[0], L->vf[1], L->vf[2], L->ff, L->rf}) if (q) (void)hipFree(q);
        pl_geom_free(L->gh);
        delete L;
    }
    S->levels.clear();
}

void pl_solver_free(pl_ctx* ctx) {
    PlSolver* S = (PlSolver*)ctx->krylov;
    if (!S) return;
    free_levels(S);
    for (double* q : {S->r, S->rt, S->p, S->v, S->s, S->t, S->y, S->z, S->b, S->x, S->xb, S->dx, S->r0, S->xh, S->scal, S->wdefl, S->udefl, S->Awdefl})
        if (q) (void)hipFree(q);
    for (double* q : S->h) if (q) (void)hipFree(q);
    for (double* q : S->hc) if (q) (void)hipFree(q);
    if (S->hc_part) (void)hipFree(S->hc_part);
    if (S->hc3) (void)hipFree(S->hc3);
    if (S->hpart) (void)hipHostFree(S->hpart);
    if (S->stream2) (void)hipStreamDestroy(S->stream2);
    if (S->ev_f) (void)hipEventDestroy(S->ev_f);
    if (S->ev_b) (void)hipEventDestroy(S->ev_b);
    delete S;
    ctx->krylov = nullptr;
}

static int dmalloc0(pl_ctx* ctx, double** p, size_t bytes) {
    PL_HIP(ctx, hipMalloc((void**)p, bytes));
    PL_HIP(ctx, hipMemsetAsync(*p, 0, bytes, ctx->stream));
    return 0;
}

// ---- BiCGStab scalars kept on the device -------------------------------------------------------------
// alpha and omega are produced and consumed on the device (sc[0], sc[1] hold the two sums of the last
// dots_dev call; sc[2] = alpha, sc[3] = omega): two of the three host round trips per iteration (~30 us of
// idle GPU each) disappear; the third (rho, ||r||) stays because the host decides whether to go on.
__global__ void k_scalar_alpha(double* __restrict__ sc, double rho_new) { sc[2] = rho_new / sc[0]; sc[4] = rho_new; }
__global__ void k_scalar_omega(double* __restrict__ sc) { sc[3] = (sc[1] > 0.0) ? sc[0] / sc[1] : 0.0; }   // t = 0: s is already the residual
// y = a - alpha b
__global__ void k_s_update_dev(long long n, double* __restrict__ y, const double* __restrict__ a, const double* __restrict__ b,
                               const double* __restrict__ sc) {
    const double alpha = sc[2];
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) y[k] = a[k] - alpha * b[k];
}
// x += alpha y + omega z ; r = s - omega t
// x += alpha y + omega z ; r = s - omega t ; and the NEXT iteration's direction p = r + beta (p - omega v) in the same
// pass (beta from the fused reduction, sc[7]): r is not read back, 10 instead of 13 vector passes
__global__ void k_xrp_update_dev(long long n, double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                                 double* __restrict__ r, const double* __restrict__ s, const double* __restrict__ t,
                                 double* __restrict__ p, const double* __restrict__ v, const double* __restrict__ sc,
                                 const double* __restrict__ wd, const double* __restrict__ awd) {
    const double alpha = sc[2], omega = sc[3], beta = sc[7];
    const double cw = wd ? alpha * sc[30] + omega * sc[31] : 0.0;      // lazy deflation: y and z stand for y + c_y w and z + c_z w
    const double cz = awd ? sc[31] : 0.0;                              // ... and (several ranks) t for t + c_z A w
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) {
        double xn = x[k] + alpha * y[k] + omega * z[k];
        if (wd) xn += cw * wd[k];
        x[k] = xn;
        const double rn = s[k] - omega * (awd ? t[k] + cz * awd[k] : t[k]);
        r[k] = rn;
        p[k] = rn + beta * (p[k] - omega * v[k]);
    }
}
__global__ void k_xr_update_dev(long long n, double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                                double* __restrict__ r, const double* __restrict__ s, const double* __restrict__ t,
                                const double* __restrict__ sc) {
    const double alpha = sc[2], omega = sc[3];
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) { x[k] += alpha * y[k] + omega * z[k]; r[k] = s[k] - omega * t[k]; }
}

// sum the block partials on the device (one workgroup): out[0], out[1]
// mode 1 / 2 (single rank): also derive alpha = rho_new / out[0] -> out[2], resp. omega = out[0] / out[1] -> out[3]
__global__ __launch_bounds__(256) void k_sum_partials(int nb, const double* __restrict__ part, double* __restrict__ out,
                                                      int mode = 0, double rho_new = 0.0) {
    double s0 = 0.0, s1 = 0.0;
    for (int k = threadIdx.x; k < nb; k += 256) { s0 += part[2 * k]; s1 += part[2 * k + 1]; }
    __shared__ double sh[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o, 64); s1 += __shfl_down(s1, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a0 = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3], a1 = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        out[0] = a0; out[1] = a1;
        if (mode == 1) { out[2] = rho_new / a0; out[4] = rho_new; }
        if (mode == 2) out[3] = (a1 > 0.0) ? a0 / a1 : 0.0;
    }
}

static int dots(pl_ctx* ctx, PlSolver* S, const PlGeom& g, int np, const double* a, const double* b, const double* c,
                const double* d, double* out2) {
    long long rows = (long long)g.lnz * np;
    const int nb = (int)(rows < DOT_BLOCKS ? rows : DOT_BLOCKS);
    if (a && c) hipLaunchKernelGGL((k_dot2<true, true>), dim3(nb), dim3(256), 0, ctx->stream, g, np, a, b, c, d, S->scal + PL_SCAL_N);
    else if (a) hipLaunchKernelGGL((k_dot2<true, false>), dim3(nb), dim3(256), 0, ctx->stream, g, np, a, b, c, d, S->scal + PL_SCAL_N);
    else hipLaunchKernelGGL((k_dot2<false, true>), dim3(nb), dim3(256), 0, ctx->stream, g, np, a, b, c, d, S->scal + PL_SCAL_N);
    if (ctx->nranks > 1 && pl_geom_is_dist(g)) {
        // several ranks: reduce on the device, all-reduce 2 doubles (stream-ordered over xGMI on the native transport), one 16-byte copy back
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, ctx->stream, nb, S->scal + PL_SCAL_N, S->scal);
        PL_TRY(pl_comm_allreduce_dev(ctx, S->scal, 2));
        PL_HIP(ctx, hipMemcpyAsync(S->hpart, S->scal, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        out2[0] = S->hpart[0]; out2[1] = S->hpart[1];
        return 0;
    }
    PL_HIP(ctx, hipMemcpyAsync(S->hpart, S->scal + PL_SCAL_N, (size_t)2 * nb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < nb; k++) { s0 += S->hpart[2 * k]; s1 += S->hpart[2 * k + 1]; }
    out2[0] = s0; out2[1] = s1;
    return 0;
}

// the device-scalar path needs the global sums on the device: one rank, or the native (stream-ordered) all-reduce
static bool dots_on_device(pl_ctx* ctx, const PlGeom& g) {
    (void)ctx; (void)g;
    return true;       // pl_comm_allreduce_dev stages through the host on the non-native transports
}
// second reduction point of BiCGStab: omega -> scal[3], rho' -> scal[5], |r|^2 -> scal[6] (ONE all-reduce of 5 doubles)
static int dots5_dev(pl_ctx* ctx, PlSolver* S, const PlGeom& g, int np, int nsplit, const double* t, const double* sv, const double* rt,
                     const double* aw_ride = nullptr) {
    long long rows = (long long)g.lnz * np;
    const int nb = (int)(rows < DOT_BLOCKS ? rows : DOT_BLOCKS);
    hipLaunchKernelGGL(k_dot5, dim3(nb), dim3(256), 0, ctx->stream, g, np, nsplit, t, sv, rt, S->scal + PL_SCAL_N);
    const bool reduce = ctx->nranks > 1 && pl_geom_is_dist(g);
    hipLaunchKernelGGL(k_sum_partials5, dim3(1), dim3(256), 0, ctx->stream, nb, S->scal + PL_SCAL_N, S->scal, reduce ? 0 : 1);
    if (reduce && aw_ride) {          // lazy deflation on several ranks: its sums and the five-cell share ride in this ONE all-reduce (18 values)
        double* part2 = S->scal + PL_SCAL_N + 8 * DOT_BLOCKS;
        hipLaunchKernelGGL(k_dot4w, dim3(nb), dim3(256), 0, ctx->stream, g, np, nsplit, aw_ride, sv, t, part2);
        hipLaunchKernelGGL(k_sum_partials4, dim3(1), dim3(256), 0, ctx->stream, nb, (const double*)part2, S->scal + 40);
        hipLaunchKernelGGL(k_bicg_pack_defl, dim3(1), dim3(64), 0, ctx->stream, S->scal);
        PL_TRY(pl_comm_allreduce_dev(ctx, S->scal + 44, 18));
        hipLaunchKernelGGL(k_bicg_unpack_defl, dim3(1), dim3(1), 0, ctx->stream, S->scal);
        return 0;
    }
    if (reduce) {
        PL_TRY(pl_comm_allreduce_dev(ctx, S->scal + 8, 13));        // scal[16..20]: the local ||x_vel||^2, y.s, y.t, (unused) and ||z_vel||^2 ride along
        hipLaunchKernelGGL(k_bicg_derive, dim3(1), dim3(1), 0, ctx->stream, S->scal);
    }
    return 0;
}
// ||a + b||^2 over the interior nodes of the first nplanes planes -> S->scal[0] (b may be NULL)
__global__ __launch_bounds__(256) void k_norm2_sum(PlGeom g, int nplanes, const double* __restrict__ a, const double* __restrict__ b,
                                                   double* __restrict__ part) {
    double s0 = 0.0;
    const long long rows = (long long)g.lnz * nplanes;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int q = (int)(r / g.lnz), li = (int)(r % g.lnz);
        const long long base = pl_idx(g, li, 0) + q * g.plane;
        for (int k = threadIdx.x; k < g.lnx; k += 256) { const double v = a[base + k] + (b ? b[base + k] : 0.0); s0 += v * v; }
    }
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s0 += __shfl_down(s0, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s0;
    __syncthreads();
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3]; part[2 * blockIdx.x + 1] = 0.0; }
}
// -> S->scal[16] (this rank's share when reduce = false: the fused reduction of the iteration all-reduces it)
static int norm2_sum_dev(pl_ctx* ctx, PlSolver* S, const PlGeom& g, int np, const double* a, const double* b, bool reduce, int slot = 16) {
    long long rows = (long long)g.lnz * np;
    const int nb = (int)(rows < DOT_BLOCKS ? rows : DOT_BLOCKS);
    hipLaunchKernelGGL(k_norm2_sum, dim3(nb), dim3(256), 0, ctx->stream, g, np, a, b, S->scal + PL_SCAL_N);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, ctx->stream, nb, S->scal + PL_SCAL_N, S->scal + slot, 0, 0.0);
    if (reduce && ctx->nranks > 1 && pl_geom_is_dist(g)) PL_TRY(pl_comm_allreduce_dev(ctx, S->scal + slot, 2));
    return 0;
}
// sums of the two dot products into S->scal[0..1], no host synchronisation
// mode 1 / 2: alpha resp. omega are derived in the same pass (see k_sum_partials); with several ranks the sums
// are all-reduced first and a one-thread kernel derives the scalar
static int dots_dev(pl_ctx* ctx, PlSolver* S, const PlGeom& g, int np, const double* a, const double* b, const double* c,
                    const double* d, int mode, double rho_new, bool ride = false) {
    long long rows = (long long)g.lnz * np;
    const int nb = (int)(rows < DOT_BLOCKS ? rows : DOT_BLOCKS);
    if (a && c) hipLaunchKernelGGL((k_dot2<true, true>), dim3(nb), dim3(256), 0, ctx->stream, g, np, a, b, c, d, S->scal + PL_SCAL_N);
    else if (a) hipLaunchKernelGGL((k_dot2<true, false>), dim3(nb), dim3(256), 0, ctx->stream, g, np, a, b, c, d, S->scal + PL_SCAL_N);
    else hipLaunchKernelGGL((k_dot2<false, true>), dim3(nb), dim3(256), 0, ctx->stream, g, np, a, b, c, d, S->scal + PL_SCAL_N);
    const bool reduce = ctx->nranks > 1 && pl_geom_is_dist(g);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, ctx->stream, nb, S->scal + PL_SCAL_N, S->scal, reduce ? 0 : mode, rho_new);
    if (reduce && ride && mode == 1) {                  // lazy deflation on several ranks: the five-cell share of y_u travels with r~.v_u
        hipLaunchKernelGGL(k_defl_pack_first, dim3(1), dim3(1), 0, ctx->stream, S->scal);
        PL_TRY(pl_comm_allreduce_dev(ctx, S->scal, 2));
        hipLaunchKernelGGL(k_defl_alpha_ride, dim3(1), dim3(1), 0, ctx->stream, S->scal, rho_new);
        return 0;
    }
    if (reduce) {
        PL_TRY(pl_comm_allreduce_dev(ctx, S->scal, 2));
        if (mode == 1) hipLaunchKernelGGL(k_scalar_alpha, dim3(1), dim3(1), 0, ctx->stream, S->scal, rho_new);
        if (mode == 2) hipLaunchKernelGGL(k_scalar_omega, dim3(1), dim3(1), 0, ctx->stream, S->scal);
    }
    return 0;
}

// ---- hierarchy ------------------------------------------------------------------------------
static void level_flags(MgLevel* L, const PlStokesOp& sop, bool finest) {
    PlVvOp& o = L->op;
    o.g = L->gh.d; o.etas = L->etas; o.etan = L->etan;
    o.rdz = o.g.rdz; o.rDz = o.g.rDz; o.rdx = o.g.rdx; o.rDx = o.g.rDx;
    o.szz = nullptr; o.szx = nullptr;               // attached by build_hierarchy when the operator is stabilised
    o.slave_x = finest ? 1 : 0;
    const bool ns0 = sop.bc_z0 != PL_BC_FREESLIP, nsL = sop.bc_zL != PL_BC_FREESLIP;
    o.slave_z0 = (finest || ns0) ? 1 : 0;
    o.slave_zL = (finest || nsL) ? 1 : 0;
    const std::vector<double>& z = L->gh.zc;
    const int nz = (int)z.size();
    // NOSLIP extrapolation rows (pylamp_stokes.py:165-166,204-205): a0 v_s + a1 v_m = 0
    o.s0 = ns0 ? (1.0 / (z[2] - z[0])) / (1.0 / (z[2] - z[0]) + 1.0 / (z[1] - z[0])) : 1.0;
    o.sL = nsL ? (1.0 / (z[nz - 1] - z[nz - 3])) / (1.0 / (z[nz - 1] - z[nz - 3]) + 1.0 / (z[nz - 1] - z[nz - 2])) : 1.0;
}

// plane (or any array) to single precision, times a factor
__global__ void k_to_float(long long n, const double* __restrict__ a, float* __restrict__ o, double scale) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
        o[t] = (float)(a[t] * scale);
}
static int fmalloc0(pl_ctx* ctx, float** p, size_t bytes) {
    PL_HIP(ctx, hipMalloc((void**)p, bytes));
    PL_HIP(ctx, hipMemsetAsync(*p, 0, bytes, ctx->stream));
    return 0;
}
// FP32 twins of the large levels (a prefix of the hierarchy): the V-cycle is only an approximation of A_vv^-1 inside an
// FP64 BiCGStab, and on these levels the sweeps are bound by bytes and by the half-rate FP64 pipe.  The hierarchy itself
// (coarsening, eigenvalue estimates) is built in FP64 first; this converts the coefficients of the current solve.
static int setup_f32_levels(pl_ctx* ctx, PlSolver* S) {
    const PlStokesOp& sop = ctx->sop;
    const double hz = (ctx->geom.zc.back() - ctx->geom.zc.front()) / (ctx->nz - 1);
    const double hx = (ctx->geom.xc.back() - ctx->geom.xc.front()) / (ctx->nx - 1);
    S->sigma = sop.Kc / (0.5 * (hz + hx));                  // ~ eta_min / h^2
    if (!(S->sigma > 0.0) || !std::isfinite(S->sigma)) S->sigma = 1.0;
    bool prefix = true;
    for (size_t l = 0; l < S->levels.size(); l++) {
        MgLevel* L = S->levels[l];
        const PlGeom& g = L->gh.d;
        const bool tail = S->use_tail && !S->any_line && l > 0 && (long long)g.nz * g.nx <= S->tail_max_nodes && S->levels.size() - l <= PL_TAIL_MAX_LEVELS;
        const bool want = S->f32_enable && !S->any_line && prefix && !tail && l + 1 < S->levels.size() && (long long)g.lnz * g.lnx >= S->f32_min_nodes &&
                          !L->op.szz && (ctx->nranks == 1 || L->dist) && (g.plane % 2) == 0;
        L->f32 = want;
        prefix = prefix && want;
        if (!want) continue;
        const size_t pb = (size_t)g.plane * sizeof(float);
        if (!L->etas_f) {
            PL_TRY(fmalloc0(ctx, &L->etas_f, pb)); PL_TRY(fmalloc0(ctx, &L->etan_f, pb));
            for (int q = 0; q < 3; q++) PL_TRY(fmalloc0(ctx, &L->vf[q], 2 * pb));
            PL_TRY(fmalloc0(ctx, &L->ff, 2 * pb)); PL_TRY(fmalloc0(ctx, &L->rf, 2 * pb));
            // the four reciprocal-spacing tables of the level, same layout as PlGeom's (index + PL_TOFF, zero padded, even lengths)
            const std::vector<double>& zc = L->gh.zc; const std::vector<double>& xc = L->gh.xc;
            const int nz = (int)zc.size(), nx = (int)xc.size();
            const size_t lz = ((size_t)nz + 2 * PL_TOFF + 2 + 1) & ~(size_t)1, lx = ((size_t)nx + 2 * PL_TOFF + 2 + 1) & ~(size_t)1;
            std::vector<float> t(2 * lz + 2 * lx, 0.0f);
            float* rdz = t.data(); float* rDz = rdz + lz; float* rdx = rDz + lz; float* rDx = rdx + lx;
            for (int i = 0; i + 1 < nz; i++) rdz[i + PL_TOFF] = (float)(1.0 / (zc[i + 1] - zc[i]));
            for (int i = 1; i + 1 < nz; i++) rDz[i + PL_TOFF] = (float)(1.0 / (zc[i + 1] - zc[i - 1]));
            for (int j = 0; j + 1 < nx; j++) rdx[j + PL_TOFF] = (float)(1.0 / (xc[j + 1] - xc[j]));
            for (int j = 1; j + 1 < nx; j++) rDx[j + PL_TOFF] = (float)(1.0 / (xc[j + 1] - xc[j - 1]));
            PL_HIP(ctx, hipMalloc((void**)&L->tab_f, t.size() * sizeof(float)));
            PL_HIP(ctx, hipMemcpy(L->tab_f, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
            L->opf.rdz = L->tab_f; L->opf.rDz = L->tab_f + lz; L->opf.rdx = L->tab_f + 2 * lz; L->opf.rDx = L->tab_f + 2 * lz + lx;
        }
        // whole planes, halo ring included (level 0 arrives with its halo filled, the coarser ones were exchanged above)
        hipLaunchKernelGGL(k_to_float, grid1d(g.plane), dim3(256), 0, ctx->stream, g.plane, (const double*)L->etas, L->etas_f, 1.0 / S->sigma);
        hipLaunchKernelGGL(k_to_float, grid1d(g.plane), dim3(256), 0, ctx->stream, g.plane, (const double*)L->etan, L->etan_f, 1.0 / S->sigma);
        PlVvOpF& o = L->opf;
        o.g = L->op.g; o.etas = L->etas_f; o.etan = L->etan_f; o.szz = nullptr; o.szx = nullptr;
        o.slave_x = L->op.slave_x; o.slave_z0 = L->op.slave_z0; o.slave_zL = L->op.slave_zL; o.s0 = (float)L->op.s0; o.sL = (float)L->op.sL;
    }
    PL_HIP(ctx, hipGetLastError());
    return 0;
}

static int build_hierarchy(pl_ctx* ctx, PlSolver* S) {
    const PlStokesOp& sop = ctx->sop;
    const int R = ctx->nranks;
    bool rebuild = S->levels.empty();
    int key[4] = {sop.bc_z0, 0, sop.bc_zL, 0};
    if (!rebuild && (S->bc_key[0] != key[0] || S->bc_key[2] != key[2])) rebuild = true;
    if (rebuild) {
        free_levels(S);
        int nz = ctx->nz, nx = ctx->nx;
        std::vector<double> zc = ctx->geom.zc, xc = ctx->geom.xc;
        // one GPU: the 65^2 level is faster as ordinary kernels (64 workgroups) than inside the single-workgroup
        // tail (94 vs 99 ms per solve at 2049^2); several ranks: every distributed level costs halo exchanges, so
        // the replicated tail starts as early as it can
        // Smoothing sweeps.  On large grids the finest level is ~2/3 of a V-cycle's time: V(1,1) there (the first
        // sweep from the zero guess is nearly free) and V(3,3) below measured 80 vs 85 ms (mantle 2049^2),
        // 148 vs 196 ms (1e3 block 2049^2), 62 vs 76 ms (block 1025^2); small grids are launch-bound and keep
        // V(2,2) everywhere (257^2: 25.6 vs 27.7 ms).
        if (S->nu_auto) {
            const bool large = (long long)ctx->nz * ctx->nx >= 1000000LL;
            S->nu_pre = S->nu_post = large ? 3 : 2;
            S->nu0_pre = S->nu0_post = large ? 1 : -1;
        }
        // Anisotropic cells: with aspect ratio a the modes that oscillate only along the weakly coupled axis sit at
        // lmax / a^2 -- full coarsening cannot carry them and the default window [lmax/6, lmax] never touches
        // them.  Widen the window by a^2 and smooth ~a times longer (PYLAMP_MG_ANISO=0 switches this off).
        if (S->aniso_auto && S->nu_auto && !S->ratio_knob) {
            const double hz = (ctx->geom.zc.back() - ctx->geom.zc.front()) / (ctx->nz - 1);
            const double hx = (ctx->geom.xc.back() - ctx->geom.xc.front()) / (ctx->nx - 1);
            const double a = hz > hx ? hz / hx : hx / hz;
            const char* le = getenv("PYLAMP_MG_LINE");
            const bool lines = ((hx >= 6.0 * hz && line_z_possible(ctx, ctx->nz)) || (hz >= 6.0 * hx && line_z_possible(ctx, ctx->nx))) &&
                               !(le && atoi(le) == 0);                                    // (the line smoothers' cases)
            S->cheb_ratio = 6.0;
            if (a > 1.5 && !lines) {
                S->cheb_ratio = 6.0 * a * a;
                int nu = (int)std::ceil(2.0 * a); if (nu > 10) nu = 10;
                S->nu_pre = S->nu_post = nu; S->nu0_pre = S->nu0_post = -1;
            }
        }
        // (several ranks used to start the one-workgroup tail at 65^2 -- the levels above it cost halo exchanges then; since the
        //  replicated levels run the tile kernels it is the one-rank choice: 33^2, which is what the LDS-resident tail holds)
        S->tail_max_nodes = S->tail_knob ? S->tail_knob : 33LL * 33;
        S->repl_start = -1; S->any_line = false;
        for (int l = 0;; l++) {
            MgLevel* L = new MgLevel();
            if (pl_geom_build(ctx, L->gh, nz, nx, zc.data(), xc.data())) { delete L; return 1; }
            if (S->repl_start < 0 && l > 0 && (long long)nz * nx <= S->repl_max_nodes) S->repl_start = l;
            if (R > 1 && S->repl_start < 0) {                     // distributed level
                const int Cz = (nz - 1) / ctx->Pz, Cx = (nx - 1) / ctx->Px;
                if ((nz - 1) % ctx->Pz || (nx - 1) % ctx->Px || (ctx->Pz > 1 && Cz < 2) || (ctx->Px > 1 && Cx < 2)) {
                    delete L; return pl_fail(ctx, "multigrid: the level cannot be divided into the rank blocks (at least 2 node rows / columns each)");
                }
                L->dist = true;
                int i0, ni, j0, nj;
                pl_block_1d(nz, ctx->Pz, ctx->pz, &i0, &ni); pl_block_1d(nx, ctx->Px, ctx->px, &j0, &nj);
                pl_geom_set_block(L->gh, i0, ni, j0, nj);
            }
            if (R > 1 && l == S->repl_start) {                    // this rank's window of the replicated arrays
                if ((nz - 1) % ctx->Pz || (nx - 1) % ctx->Px) { delete L; return pl_fail(ctx, "multigrid: coarse grid smaller than the number of ranks"); }
                L->win = L->gh.d;
                pl_block_1d(nz, ctx->Pz, ctx->pz, &L->win.gi0, &L->win.lnz); pl_block_1d(nx, ctx->Px, ctx->px, &L->win.gj0, &L->win.lnx);
                L->win_shift = (long long)L->win.gi0 * L->gh.d.pitch + L->win.gj0;
            }
            size_t vb = (size_t)2 * L->gh.d.plane * sizeof(double), pb = (size_t)L->gh.d.plane * sizeof(double);
            if (l > 0) {
                L->own_visc = true;
                PL_TRY(dmalloc0(ctx, &L->etas, pb)); PL_TRY(dmalloc0(ctx, &L->etan, pb));
            }
            for (int q = 0; q < 3; q++) PL_TRY(dmalloc0(ctx, &L->v[q], vb));
            PL_TRY(dmalloc0(ctx, &L->f, vb)); PL_TRY(dmalloc0(ctx, &L->r, vb));
            {   // z-lines where some cell of the level is at least 6 x wider than high (PYLAMP_MG_LINE=0: never; =1: on every level).
                // Measured (tools/stretch_probe.py, viscosity contrast 1e3): cells 4:1 (513 x 129 nodes on a square) point sweeps with the
                // anisotropy rule 41 iterations / 72 ms, lines 45 / 93 -- the count is set by the pressure block there, not by the
                // smoother; z graded 30 x (cells up to 8.5:1): 209 / 101 ms against 91 / 217 ms; cells 16:1 (1025 x 65): the point
                // sweeps do NOT converge (residual 2e-4 after 109 iterations), lines reach 1e-8 in 164.
                double dzmin = 1e300, dxmax = 0.0, dxmin = 1e300, dzmax = 0.0;
                for (int i = 0; i + 1 < nz; i++) { dzmin = std::min(dzmin, zc[i + 1] - zc[i]); dzmax = std::max(dzmax, zc[i + 1] - zc[i]); }
                for (int j = 0; j + 1 < nx; j++) { dxmax = std::max(dxmax, xc[j + 1] - xc[j]); dxmin = std::min(dxmin, xc[j + 1] - xc[j]); }
                const char* e = getenv("PYLAMP_MG_LINE");
                const int knob = e ? atoi(e) : -1;                  // 0 never, 1 z-lines on every level, 2 x-lines on every level
                const double wide = dxmax / dzmin, high = dzmax / dxmin;
                L->line_z = false; L->line_ax = 0;
                if (knob == 1 || (knob < 0 && wide >= 6.0 && wide >= high)) { L->line_z = line_z_possible(ctx, nz); L->line_ax = 0; }
                else if (knob == 2 || (knob < 0 && high >= 6.0)) { L->line_z = line_z_possible(ctx, nx); L->line_ax = 1; }
                if (L->line_z) S->any_line = true;
            }
            S->levels.push_back(L);
            if ((nz - 1) % 2 || (nx - 1) % 2 || (nz - 1) / 2 < S->min_cells || (nx - 1) / 2 < S->min_cells) break;
            // blocks with an odd number of rows / columns cannot be halved rank by rank: this level stays the coarsest
            // one (a distributed coarsest level is smoothed with a halo exchange per sweep - slow, but correct)
            if (L->dist && ((ctx->Pz > 1 && (((nz - 1) / ctx->Pz) % 2)) || (ctx->Px > 1 && (((nx - 1) / ctx->Px) % 2)))) break;
            std::vector<double> z2, x2;
            for (int i = 0; i < nz; i += 2) z2.push_back(zc[i]);
            for (int j = 0; j < nx; j += 2) x2.push_back(xc[j]);
            zc.swap(z2); xc.swap(x2); nz = (nz - 1) / 2 + 1; nx = (nx - 1) / 2 + 1;
        }
        S->bc_key[0] = key[0]; S->bc_key[2] = key[2];
    }
    // (re)attach viscosities, coarsen, estimate lambda_max
    S->levels[0]->etas = (double*)sop.etas; S->levels[0]->etan = (double*)sop.etan;
    for (size_t l = 0; l < S->levels.size(); l++) {
        MgLevel* L = S->levels[l];
        if (l > 0) {
            MgLevel* F = S->levels[l - 1];
            if (F->dist && l > 1) {       // fine viscosity halos: the [1 2 1] stencil here, and the smoothing kernels that run on
                                          // the block extended into the halo (level 0 arrives with its halo filled: scatter / upload)
                const int dep = std::min(PL_RING, std::min(F->gh.d.lnz, F->gh.d.lnx));
                PL_TRY(pl_halo(ctx, F->gh.d, F->etas, 1, F->gh.d.plane, dep));
                PL_TRY(pl_halo(ctx, F->gh.d, F->etan, 1, F->gh.d.plane, dep));
            }
            if (R > 1 && (int)l == S->repl_start) {
                // my block of the replicated arrays, then gather everybody's
                const long long sh = L->win_shift;
                hipLaunchKernelGGL(k_coarsen_visc, grid2d(L->win), dim3(64, 4), 0, ctx->stream, F->gh.d, F->etas, F->etan,
                                   L->win, L->etas + sh, L->etan + sh);
                PL_TRY(pl_gather_blocks(ctx, L->gh.d, L->etas, 1, L->gh.d.plane));
                PL_TRY(pl_gather_blocks(ctx, L->gh.d, L->etan, 1, L->gh.d.plane));
            } else {
                hipLaunchKernelGGL(k_coarsen_visc, grid2d(L->gh.d), dim3(64, 4), 0, ctx->stream, F->gh.d, F->etas, F->etan,
                                   L->gh.d, L->etas, L->etan);
            }
        }
        level_flags(L, sop, l == 0);
    }
    {   // a distributed coarsest level never is the "fine" level of the loop above: its halo is filled here
        MgLevel* L = S->levels.back();
        if (L->dist && S->levels.size() > 1) {
            const int dep = std::min(PL_RING, std::min(L->gh.d.lnz, L->gh.d.lnx));
            PL_TRY(pl_halo(ctx, L->gh.d, L->etas, 1, L->gh.d.plane, dep));
            PL_TRY(pl_halo(ctx, L->gh.d, L->etan, 1, L->gh.d.plane, dep));
        }
    }
    const bool stab_in_mg = true;
    if (sop.surfstab && sop.ss != 0.0 && sop.gz != 0.0 && stab_in_mg) {
        // stabilisation-aware velocity block: density coarsened like the nodal viscosity, terms rediscretised per level
        const double coef = sop.ss * sop.gz;
        for (size_t l = 0; l < S->levels.size(); l++) {
            MgLevel* L = S->levels[l];
            const size_t pb = (size_t)L->gh.d.plane * sizeof(double);
            if (!L->szz) {
                PL_TRY(dmalloc0(ctx, &L->szz, pb)); PL_TRY(dmalloc0(ctx, &L->szx, pb));
                if (l > 0) { PL_TRY(dmalloc0(ctx, &L->rho, pb)); L->own_rho = true; }
            }
            if (l == 0) { L->rho = (double*)sop.rho; continue; }
            MgLevel* F = S->levels[l - 1];
            if (F->dist) PL_TRY(pl_halo(ctx, F->gh.d, F->rho, 1, F->gh.d.plane, 2));
            if (R > 1 && (int)l == S->repl_start) {
                const long long sh = L->win_shift;
                hipLaunchKernelGGL(k_coarsen_node, grid2d(L->win), dim3(64, 4), 0, ctx->stream, F->gh.d, F->rho, L->win, L->rho + sh);
                PL_TRY(pl_gather_blocks(ctx, L->gh.d, L->rho, 1, L->gh.d.plane));
            } else {
                hipLaunchKernelGGL(k_coarsen_node, grid2d(L->gh.d), dim3(64, 4), 0, ctx->stream, F->gh.d, F->rho, L->gh.d, L->rho);
            }
        }
        for (size_t l = 0; l < S->levels.size(); l++) {
            MgLevel* L = S->levels[l];
            if (L->dist) PL_TRY(pl_halo(ctx, L->gh.d, L->rho, 1, L->gh.d.plane, 2));
            hipLaunchKernelGGL(k_stab_coeffs, grid2d(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, L->rho, coef, L->szz, L->szx);
            if (L->dist) {
                const int dep = std::min(PL_RING, std::min(L->gh.d.lnz, L->gh.d.lnx));
                PL_TRY(pl_halo(ctx, L->gh.d, L->szz, 1, L->gh.d.plane, dep));
                PL_TRY(pl_halo(ctx, L->gh.d, L->szx, 1, L->gh.d.plane, dep));
            }
            L->op.szz = L->szz; L->op.szx = L->szx;
        }
    }
    PL_HIP(ctx, hipGetLastError());
    for (MgLevel* L : S->levels) {
        const PlGeom& g = L->gh.d;
        long long n2 = 2 * g.plane;
        // The viscosity changes little from one time step to the next: the power iteration restarts from the
        // eigenvector of the previous solve with 3 instead of 12 iterations (each is a kernel + a host round trip;
        // 12 x 9 levels were ~4 ms of every solve).
        if (!L->eig) { PL_TRY(dmalloc0(ctx, &L->eig, (size_t)n2 * sizeof(double))); L->eig_valid = false; }
        // In a time loop (the driver announces consecutive solves of one slowly changing model) an estimate that two solves in a row
        // have confirmed to 0.2 % is refreshed only every fourth solve: each refresh is two 2-plane copies, a kernel, an axpy and a
        // host round trip per level (0.5 ms per solve at 2049^2), and lmax carries a safety factor of 1.1
        if (S->defl_persistent && L->eig_valid && L->eig_confirm >= 2 && L->eig_skip < 3 && L->lmax > 0.0) { L->eig_skip++; continue; }
        L->eig_skip = 0;
        const double lam_before = (L->eig_valid && S->lmax_safety > 0.0) ? L->lmax / S->lmax_safety : 0.0;
        if (L->eig_valid) PL_HIP(ctx, hipMemcpyAsync(L->v[0], L->eig, (size_t)n2 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        else hipLaunchKernelGGL(k_random_interior, grid2d(g), dim3(64, 4), 0, ctx->stream, g, 2, L->v[0], 777u);
        const bool warm = L->eig_valid;
        // warm: the previous solve's estimate is the value the first iteration is compared with, so that ONE iteration suffices
        // when it confirms it to 1 % (the iteration then simply continues from solve to solve)
        double lam = (warm && L->lmax > 0.0 && S->lmax_safety > 0.0) ? L->lmax / S->lmax_safety : 2.5, lam_prev = 0.0, nn[2];
        for (int it = 0; it < 12; it++) {
            // warm restart: at least power_its_warm iterations, then stop once the estimate moves by < 1 %
            if (warm && it >= S->power_its_warm && std::fabs(lam - lam_prev) < 0.01 * lam) break;
            lam_prev = lam;
            if (L->dist) PL_TRY(pl_halo(ctx, g, L->v[0], 2, g.plane));
            if (L->line_z) PL_TRY(line_z_launch(ctx, L->op, L->v[0], nullptr, nullptr, L->v[1], 0.0, 1.0, 1.0, 1, L->line_ax));
            else hipLaunchKernelGGL(k_vv_dinv_apply, grid2d(g), dim3(64, 4), 0, ctx->stream, L->op, L->v[0], L->v[1]);
            PL_TRY(dots(ctx, S, g, 2, L->v[1], L->v[1], L->v[0], L->v[0], nn));
            if (!(nn[1] > 0.0) || !(nn[0] > 0.0)) break;
            lam = std::sqrt(nn[0] / nn[1]);
            hipLaunchKernelGGL(k_axpy_out, grid1d(n2), dim3(256), 0, ctx->stream, n2, L->v[0], L->v[1], L->v[1],
                               1.0 / std::sqrt(nn[0]) - 1.0);     // v0 = v1 / ||v1||
        }
        L->lmax = S->lmax_safety * lam;
        PL_HIP(ctx, hipMemcpyAsync(L->eig, L->v[0], (size_t)n2 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        L->eig_valid = std::isfinite(lam) && lam > 0.0;
        L->eig_confirm = (L->eig_valid && lam_before > 0.0 && std::fabs(lam - lam_before) < 0.002 * lam) ? L->eig_confirm + 1 : 0;
    }
    PL_TRY(setup_f32_levels(ctx, S));
    // early coarse branch: first level of at most early_max_nodes nodes, on one rank, when at least two levels lie above it
    S->early_K = 0;
    if (ctx->nranks == 1 && S->early_knob != 0 && !S->levels[0]->f32 && S->levels.size() > 2) {
        int K = S->early_knob > 0 ? S->early_knob : 0;
        if (K == 0 && (long long)ctx->nz * ctx->nx >= 1000000LL)
            for (size_t l = 2; l < S->levels.size(); l++)
                if ((long long)S->levels[l]->gh.d.nz * S->levels[l]->gh.d.nx <= S->early_max_nodes) { K = (int)l; break; }
        if (K >= 1 && K + 1 <= (int)S->levels.size() - 0 && K < (int)S->levels.size() && S->nu_pre >= 1) {
            S->early_K = K;
            if (!S->stream2) {
                PL_HIP(ctx, hipStreamCreateWithFlags(&S->stream2, hipStreamNonBlocking));
                PL_HIP(ctx, hipEventCreateWithFlags(&S->ev_f, hipEventDisableTiming));
                PL_HIP(ctx, hipEventCreateWithFlags(&S->ev_b, hipEventDisableTiming));
            }
            for (int l = 1; l < K; l++) {
                MgLevel* L = S->levels[l];
                if (!L->fe) PL_TRY(dmalloc0(ctx, &L->fe, (size_t)2 * L->gh.d.plane * sizeof(double)));
            }
        }
    }
    return 0;
}

// PYLAMP_VV_VEC=0 selects the scalar one-column-per-lane sweep kernels (kept as the cross-check)
static const bool g_vv_vec_env = [] { const char* e = getenv("PYLAMP_VV_VEC"); return !(e && e[0] == '0'); }();
// the two-columns-per-lane kernels load the x tables as aligned pairs: the block's first global column must be even
#define g_vv_vec (g_vv_vec_env && !(L->gh.d.gj0 & 1))

// ---- smoothing and V-cycle ----------------------------------------------------------------
// Deep halos (distributed levels): a kernel launched on the block EXTENDED by e nodes into the halo computes there
// exactly what the neighbour computes for its own nodes (same inputs, same arithmetic), so a sequence of nu sweeps
// needs ONE exchange -- of its right-hand side, nu+... nodes deep -- instead of one per sweep: sweep k runs on the
// block extended by e_k, e shrinking by one per sweep.  The view below is the level's operator on the extended
// block; every pointer handed to a kernel is shifted back by `sh` so that local node (0,0) of the view lands on
// global node (gi0-a, gj0-c).  The extension is clipped at the domain walls; towards -x it is rounded up to an even
// number (aligned double2 accesses of the two-columns-per-lane kernels) -- the extra column is computed from data
// one node beyond the valid depth, i.e. garbage that no needed node ever reads.
template <typename T> struct ExtViewT { PlVvOpT<T> op; long long sh; };
template <typename T>
static ExtViewT<T> ext_view(pl_ctx* ctx, const MgLevel* L, int e) {
    ExtViewT<T> v; v.op = LevelT<T>::op(L); v.sh = 0;
    if (e <= 0 || !L->dist) return v;
    const PlGeom& g = L->gh.d;
    const int a = ctx->pz > 0 ? e : 0, b = ctx->pz < ctx->Pz - 1 ? e : 0;
    const int c = ctx->px > 0 ? ((e + 1) & ~1) : 0, d = ctx->px < ctx->Px - 1 ? e : 0;
    v.op.g.gi0 -= a; v.op.g.lnz += a + b; v.op.g.gj0 -= c; v.op.g.lnx += c + d;
    v.sh = (long long)a * g.pitch + c;
    v.op.etas -= v.sh; v.op.etan -= v.sh;
    if (v.op.szz) { v.op.szz -= v.sh; v.op.szx -= v.sh; }
    return v;
}
// nsweep Chebyshev-Jacobi sweeps on level L; buf[0] is the current iterate on entry and on exit.
// ext_first >= 0 (deep mode): sweep k runs on the block extended by ext_first - k nodes, no exchanges in here;
// ext_first < 0: one halo exchange before every sweep that needs one (halo policy `halo`).
// final_out: the LAST sweep writes its result there -- always an FP64 Krylov vector -- times final_scale, and buf is
// left as it was before that sweep.  Returns true when it did.
template <typename T>
static bool smooth(pl_ctx* ctx, MgLevel* L, T* buf[3], const T* f, int nsweep, double ratio,
                   double* final_out = nullptr, double final_scale = 1.0, bool zero_guess = false, int halo = 2,
                   bool first_halo_valid = false, int ext_first = -1, const int* first_done_anchor = nullptr) {
    const double lmax = L->lmax, lmin = lmax / ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho_old = 1.0 / sigma;
    bool wrote_final = false;
    for (int k = 0; k < nsweep; k++) {
        double c1, c2;
        if (k == 0) { c1 = 0.0; c2 = 1.0 / theta; }
        else { const double rho = 1.0 / (2.0 * sigma - rho_old); c1 = rho * rho_old; c2 = 2.0 * rho / delta; rho_old = rho; }
        if (ext_first < 0 && L->dist && !(k == 0 && (zero_guess || first_halo_valid)) && (halo == 2 || (halo == 1 && k == 0)))
            (void)pl_halo(ctx, L->gh.d, buf[0], 2, L->gh.d.plane);
        const ExtViewT<T> V = ext_view<T>(ctx, L, ext_first < 0 ? 0 : std::max(ext_first - k, 0));
        const long long sh = V.sh;
        const bool to_final = final_out && k == nsweep - 1;
        T* dst = buf[2];
        if constexpr (std::is_same<T, double>::value) {
            if (L->line_z) {               // z-line relaxation: same three-term recurrence, T^-1 in place of D^-1 (one rank, FP64, no views)
                const double* cur = (k == 0 && zero_guess) ? nullptr : buf[0];
                const double* prv = (k == 0 || (k == 1 && zero_guess)) ? nullptr : buf[1];
                (void)line_z_launch(ctx, L->op, cur, prv, f, to_final ? final_out : dst, c1, c2, to_final ? final_scale : 1.0, 0, L->line_ax);
                if (to_final) { wrote_final = true; break; }
                { T* nxt = buf[2]; buf[2] = buf[1]; buf[1] = buf[0]; buf[0] = nxt; }
                continue;
            }
        }
        if (k == 0 && zero_guess) {        // buf[0] is NOT read (and need not be zeroed); never the final sweep of level 0 (a
                                           // V-cycle with a coarse level has a prolongation before its last sweep)
            if (g_vv_vec && first_done_anchor) {      // stage 1 wrote the interior waves of this sweep already (into dst = buf[2])
                const dim3 full = pl_grid_rows2(V.op.g);
                hipLaunchKernelGGL(k_vv_first2<T>, dim3(3 * full.x + 3 * full.y + 2), dim3(64, 4), 0, ctx->stream, V.op, f - sh, dst - sh, (T)c2, 2,
                                   first_done_anchor[0], first_done_anchor[1], ctx->sop.wall_ps);
            }
            else if (g_vv_vec) hipLaunchKernelGGL(k_vv_first2<T>, pl_grid_rows2(V.op.g), dim3(64, 4), 0, ctx->stream, V.op, f - sh, dst - sh, (T)c2, 0, -9, -9, 0);
            else hipLaunchKernelGGL(k_vv_cheb_first<T>, pl_grid_rows(V.op.g), dim3(64, 4), 0, ctx->stream, V.op, f - sh, dst - sh, (T)c2,
                                    pl_row_iters(V.op.g));
        } else {
            const T* prev = (k == 1 && zero_guess) ? (const T*)nullptr : buf[1] - sh;
            if (to_final) {
                if (g_vv_vec)
                    hipLaunchKernelGGL((k_vv_sweep2<0, T, double>), pl_grid_rows2(V.op.g), dim3(64, 4), 0, ctx->stream, V.op, (const T*)(buf[0] - sh), prev,
                                       f - sh, final_out - sh, (T)c1, (T)c2, final_scale);
                else
                    hipLaunchKernelGGL((k_vv_cheb<T, double>), pl_grid_rows(V.op.g), dim3(64, 4), 0, ctx->stream, V.op, (const T*)(buf[0] - sh), prev, f - sh,
                                       final_out - sh, (T)c1, (T)c2, pl_row_iters(V.op.g), final_scale);
                wrote_final = true;
                break;
            }
            if (g_vv_vec)
                hipLaunchKernelGGL((k_vv_sweep2<0, T, T>), pl_grid_rows2(V.op.g), dim3(64, 4), 0, ctx->stream, V.op, (const T*)(buf[0] - sh), prev,
                                   f - sh, dst - sh, (T)c1, (T)c2, T(1));
            else
                hipLaunchKernelGGL((k_vv_cheb<T, T>), pl_grid_rows(V.op.g), dim3(64, 4), 0, ctx->stream, V.op, (const T*)(buf[0] - sh), prev, f - sh,
                                   dst - sh, (T)c1, (T)c2, pl_row_iters(V.op.g), T(1));
        }
        { T* nxt = buf[2]; buf[2] = buf[1]; buf[1] = buf[0]; buf[0] = nxt; }    // (cur, prev, free)
    }
    return wrote_final;
}

// Chebyshev sweeps on the coarsest level: enough for the eigenvalue window [lmax/ratio, lmax] of a
// grid that could not be coarsened further (ratio ~ 0.4 nz nx; a 5x5 grid needs the default 12, a
// non-coarsenable 100x60 grid ~50)
static int coarsest_sweeps(const PlSolver* S, const PlGeom& g) {
    double ratio = 0.4 * g.nz * g.nx; if (ratio < 30.0) ratio = 30.0;
    int n = (int)std::sqrt(ratio);
    if (n < S->coarse_sweeps) n = S->coarse_sweeps;
    if (n > 150) n = 150;
    return n;
}

// smoothing counts of level l
static void level_nu(const PlSolver* S, size_t l, int& npre, int& npost) {
    npre = (l == 0 && S->nu0_pre > 0) ? S->nu0_pre : S->nu_pre;
    npost = (l == 0 && S->nu0_post > 0) ? S->nu0_post : S->nu_post;
}
// Deep-halo plan of a distributed level: the pre-smoothed iterate is needed e_last nodes into the halo (1 for the
// residual whose restriction reaches one fine node beyond the block, + the post-smoothing sequence that starts from
// it), hence the right-hand side e_last + npre - 1 nodes deep.  Returns false when the block is too small for that
// (or PYLAMP_MG_DEEP=0): the level then exchanges before every sweep.
static bool level_deep_plan(const PlSolver* S, const MgLevel* L, size_t l, int& e_last, int& f_depth) {
    int npre, npost;
    level_nu(S, l, npre, npost);
    e_last = std::max(2, npost);
    f_depth = e_last + std::max(npre, 1) - 1;
    if (!L->dist || !S->deep || l + 1 == S->levels.size() || npre < 1) return false;
    // kernels on the block extended by f_depth read coefficients one node further: f_depth + 1 <= PL_RING; the exchanges
    // (f_depth deep; f_depth + 1 for the residual vector of level 0) must fit into the block
    return f_depth <= PL_RING - 1 && f_depth + 1 <= std::min(L->gh.d.lnz, L->gh.d.lnx);
}

// residual of level L -> its r buffer, restriction into the coarse level C (type TC), recursive solve there,
// prolongation of the correction into buf[2] of L.  Separate function so that the FP32 / FP64 type of C is a template
// parameter: the FP32 levels are a prefix of the hierarchy, an FP32 level may sit above an FP64 one, never below.
template <typename T> static void vcycle(pl_ctx* ctx, PlSolver* S, size_t l, const T* f, T** out, bool* wrote_final, double* final_out,
                                         double final_scale, int f_valid_depth, const int* first_done_anchor);
static bool mg_fused_level_ok(pl_ctx* ctx, const PlSolver* S, size_t l);
static bool mg_tail_lds_fits(const PlSolver* S, size_t l);
__global__ void k_mg_tail_lds(TailArgs a);
static void vcycle_fused_level(pl_ctx* ctx, PlSolver* S, size_t l, const double* f, double** out, double* final_out, double final_scale,
                               const PlStokesOp* sop, const double* rs, double* z);

template <typename T, typename TC>
static void coarse_correction(pl_ctx* ctx, PlSolver* S, size_t l, MgLevel* L, MgLevel* C, T* buf[3], bool deep, int npost, int hp) {
    const PlGeom& g = L->gh.d;
    T* r = LevelT<T>::r(L);
    TC* cf = LevelT<TC>::f(C);
    // an FP32 level works with A / sigma (stokes_precond), an FP64 one with A: the residual changes units here
    const TC cscale = (std::is_same<T, float>::value && std::is_same<TC, double>::value) ? (TC)S->sigma : TC(1);
    if (L->dist && !C->dist) {
        // restrict my block of the replicated coarse rhs, then gather everybody's (replicated levels are FP64)
        PlVvOpT<TC> wop = LevelT<TC>::op(C); wop.g = C->win;
        const long long sh = C->win_shift;
        hipLaunchKernelGGL((k_vv_restrict<T, TC>), grid2d(C->win), dim3(64, 4), 0, ctx->stream, g, wop, (const T*)r, cf + sh, cscale);
        if constexpr (std::is_same<TC, double>::value) (void)pl_gather_blocks(ctx, C->gh.d, cf, 2, C->gh.d.plane);
    } else {
        hipLaunchKernelGGL((k_vv_restrict<T, TC>), grid2d(C->gh.d), dim3(64, 4), 0, ctx->stream, g, LevelT<TC>::op(C), (const T*)r, cf, cscale);
    }
    TC* ec = nullptr;
    bool wf = false;
    vcycle<TC>(ctx, S, l + 1, cf, &ec, &wf, nullptr, 1.0, 0, nullptr);
    // ---- way up.  The correction is prolonged into the halo as well -- `pe` nodes deep -- so that the post-smoothing
    // sequence needs no exchange of its own; the coarse correction must then be known (pe+1)/2 + 1 coarse nodes deep
    // (ONE exchange, none when the coarse level is replicated).
    const int pe = deep ? npost : ((L->dist && hp == 2) ? 1 : 0);
    if (C->dist && (deep || hp >= 1))
        (void)pl_halo(ctx, C->gh.d, ec, 2, C->gh.d.plane, std::min(std::max((pe + 1) / 2 + 1, 2), std::min(PL_RING, std::min(C->gh.d.lnz, C->gh.d.lnx))));
    ExtViewT<T> V = ext_view<T>(ctx, L, pe);
    if (!deep && pe == 1) {            // legacy extension by exactly one node (no rounding: the prolongation kernel is scalar)
        V.op = LevelT<T>::op(L);
        const int lo = ctx->pz > 0 ? 1 : 0, hi = ctx->pz < ctx->Pz - 1 ? 1 : 0, we = ctx->px > 0 ? 1 : 0, ea = ctx->px < ctx->Px - 1 ? 1 : 0;
        V.op.g.gi0 -= lo; V.op.g.lnz += lo + hi; V.op.g.gj0 -= we; V.op.g.lnx += we + ea;
        V.sh = (long long)lo * g.pitch + we;
    }
    hipLaunchKernelGGL((k_vv_prolong_add<T, TC>), grid2d(V.op.g), dim3(64, 4), 0, ctx->stream, V.op, C->gh.d, (const TC*)ec, (const T*)(buf[0] - V.sh),
                       buf[2] - V.sh);
}

// Early coarse branch (PlSolver::early_K), level l = K-1 of the fine branch: on the second stream, as soon as stage 1 has
// written the right-hand side of level 0, restrict IT down to level K and solve there; the fine branch (this stream) has
// meanwhile pre-smoothed levels 0 .. K-1 and now waits for the correction, which it prolongs as usual.
static void early_coarse_branch(pl_ctx* ctx, PlSolver* S, size_t l, MgLevel* L, MgLevel* C, double* buf[3]) {
    hipStream_t main_stream = ctx->stream;
    ctx->stream = S->stream2;                       // everything below is enqueued on the second stream
    (void)hipStreamWaitEvent(S->stream2, S->ev_f, 0);
    const double* src = S->levels[0]->f;
    for (size_t q = 1; q <= l + 1; q++) {
        MgLevel* F = S->levels[q - 1];
        MgLevel* Q = S->levels[q];
        double* dst = (q == l + 1) ? Q->f : Q->fe;
        hipLaunchKernelGGL((k_vv_restrict<double, double>), grid2d(Q->gh.d), dim3(64, 4), 0, ctx->stream, F->gh.d, Q->op, src, dst, 1.0);
        src = dst;
    }
    double* ec = nullptr;
    bool wf = false;
    vcycle<double>(ctx, S, l + 1, C->f, &ec, &wf, nullptr, 1.0, 0, nullptr);
    (void)hipEventRecord(S->ev_b, S->stream2);
    ctx->stream = main_stream;
    (void)hipStreamWaitEvent(main_stream, S->ev_b, 0);
    hipLaunchKernelGGL((k_vv_prolong_add<double, double>), grid2d(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, C->gh.d, (const double*)ec,
                       (const double*)buf[0], buf[2]);
}

// solves A_vv e = f(level l) approximately; result in *out (one of the level's v buffers)
// final_out (level 0 only): the last post-smoothing sweep writes its result there (zero-copy into z), times final_scale;
// *wrote_final tells whether that happened (otherwise the result is in *out)
// f_valid_depth: how deep into the halo the caller has already made f valid (level 0: stage 1 computes it there)
template <typename T>
static void vcycle(pl_ctx* ctx, PlSolver* S, size_t l, const T* f, T** out, bool* wrote_final, double* final_out,
                   double final_scale, int f_valid_depth, const int* first_done_anchor) {
    MgLevel* L = S->levels[l];
    const PlGeom& g = L->gh.d;
    *wrote_final = false;
    if constexpr (std::is_same<T, double>::value) {
        if (S->use_tail && !S->any_line && l > 0 && (long long)g.nz * g.nx <= S->tail_max_nodes &&
            S->levels.size() - l <= PL_TAIL_MAX_LEVELS && f == L->f) {
            TailArgs ta{};
            ta.nlev = (int)(S->levels.size() - l);
            ta.nu_pre = S->tail_nu_pre >= 0 ? S->tail_nu_pre : S->nu_pre; ta.nu_post = S->tail_nu_post >= 0 ? S->tail_nu_post : S->nu_post;
            ta.coarse_sweeps = coarsest_sweeps(S, S->levels.back()->gh.d); ta.ratio = S->cheb_ratio;
            for (int q = 0; q < ta.nlev; q++) {
                MgLevel* T_ = S->levels[l + q];
                ta.L[q].op = T_->op; ta.L[q].f = T_->f; ta.L[q].r = T_->r; ta.L[q].lmax = T_->lmax;
                for (int b = 0; b < 3; b++) ta.L[q].v[b] = T_->v[b];
            }
            // one rank: the LDS-resident twin (levels kept in LDS, the tile kernels' node functions) where its pool holds the levels
            if (S->fused && !L->dist && !L->op.szz && ta.nu_pre >= 1 && ta.nu_post >= 1 && ta.coarse_sweeps >= 1 && mg_tail_lds_fits(S, l))
                hipLaunchKernelGGL(k_mg_tail_lds, dim3(1), dim3(1024), 0, ctx->stream, ta);
            else
                hipLaunchKernelGGL(k_mg_tail, dim3(1), dim3(1024), 0, ctx->stream, ta);
            *out = L->v[0];
            return;
        }
    }
    if constexpr (std::is_same<T, double>::value) {
        if (l > 0 && !final_out && f == L->f && mg_fused_level_ok(ctx, S, l)) {       // the tile kernels (level 0: stokes_precond_t)
            vcycle_fused_level(ctx, S, l, f, out, nullptr, 1.0, nullptr, nullptr, nullptr);
            return;
        }
    }
    T** lv = LevelT<T>::v(L);
    T* buf[3] = {lv[0], lv[1], lv[2]};
    const bool coarsest = l + 1 == S->levels.size();
    if (!coarsest && S->nu_pre == 0)       // otherwise the zero guess is implicit (k_vv_cheb_first)
        (void)hipMemsetAsync(buf[0], 0, (size_t)2 * g.plane * sizeof(T), ctx->stream);
    if (coarsest) {
        double ratio = 0.4 * g.nz * g.nx; if (ratio < 30.0) ratio = 30.0;
        smooth<T>(ctx, L, buf, f, coarsest_sweeps(S, g), ratio, nullptr, 1.0, true, S->mg_halo);
        *out = buf[0];
        return;
    }
    const int hp = S->mg_halo;
    int npre, npost, e_last, f_depth;
    level_nu(S, l, npre, npost);
    const bool deep = level_deep_plan(S, L, l, e_last, f_depth);
    MgLevel* C = S->levels[l + 1];
    T* r = LevelT<T>::r(L);
    // early coarse branch: the coarse levels do not see this level's residual (see PlSolver::early_K)
    const bool early_here = std::is_same<T, double>::value && S->early_K > 0 && (int)l + 1 == S->early_K && !L->dist;
    if (early_here) {
        smooth<T>(ctx, L, buf, f, npre, S->cheb_ratio, nullptr, 1.0, true, hp, false, -1, first_done_anchor);
    } else if (deep) {
        // ---- ONE exchange on the way down: the right-hand side, deep enough for the whole pre-smoothing sequence,
        //      the residual and (later) the post-smoothing sequence
        if (f_valid_depth < f_depth) (void)pl_halo(ctx, g, (T*)f, 2, g.plane, f_depth);
        smooth<T>(ctx, L, buf, f, npre, S->cheb_ratio, nullptr, 1.0, true, hp, false, e_last + npre - 1, first_done_anchor);      // iterate valid e_last deep
        const ExtViewT<T> V = ext_view<T>(ctx, L, e_last - 1);
        if (g_vv_vec)
            hipLaunchKernelGGL((k_vv_sweep2<1, T, T>), pl_grid_rows2(V.op.g), dim3(64, 4), 0, ctx->stream, V.op, (const T*)(buf[0] - V.sh),
                               (const T*)nullptr, f - V.sh, r - V.sh, T(0), T(0), T(1));
        else
            hipLaunchKernelGGL(k_vv_residual<T>, pl_grid_rows(V.op.g), dim3(64, 4), 0, ctx->stream, V.op, (const T*)(buf[0] - V.sh), f - V.sh,
                               r - V.sh, pl_row_iters(V.op.g));
    } else {
        smooth<T>(ctx, L, buf, f, npre, S->cheb_ratio, nullptr, 1.0, true, hp, false, -1, first_done_anchor);
        if (L->dist && hp >= 1) (void)pl_halo(ctx, g, buf[0], 2, g.plane);
        if (g_vv_vec)
            hipLaunchKernelGGL((k_vv_sweep2<1, T, T>), pl_grid_rows2(g), dim3(64, 4), 0, ctx->stream, LevelT<T>::op(L), (const T*)buf[0], (const T*)nullptr, f,
                               r, T(0), T(0), T(1));
        else
            hipLaunchKernelGGL(k_vv_residual<T>, pl_grid_rows(g), dim3(64, 4), 0, ctx->stream, LevelT<T>::op(L), (const T*)buf[0], f, r, pl_row_iters(g));
        if (L->dist && hp >= 1) (void)pl_halo(ctx, g, r, 2, g.plane, 2);
    }
    if constexpr (std::is_same<T, float>::value) {
        if (C->f32) coarse_correction<float, float>(ctx, S, l, L, C, buf, deep, npost, hp);
        else coarse_correction<float, double>(ctx, S, l, L, C, buf, deep, npost, hp);
    } else {
        if (early_here) early_coarse_branch(ctx, S, l, L, C, buf);
        else coarse_correction<double, double>(ctx, S, l, L, C, buf, deep, npost, hp);
    }
    std::swap(buf[0], buf[2]);
    const int pe = deep ? npost : ((L->dist && hp == 2) ? 1 : 0);
    if (deep) *wrote_final = smooth<T>(ctx, L, buf, f, npost, S->cheb_ratio, final_out, final_scale, false, hp, true, npost - 1);
    else *wrote_final = smooth<T>(ctx, L, buf, f, npost, S->cheb_ratio, final_out, final_scale, false, hp, pe == 1);
    *out = buf[0];
}

// =========================================================================================
// Fused multigrid level kernels ("tile kernels"): one launch for everything a level does on the way down, one for the way up.
//
// A V-cycle visit of a level used to be 10 dependent launches (first sweep, sweeps, residual, restriction ... prolongation,
// sweeps), each at least the 4.5-6 us a dependent kernel costs on this GPU: at 2049^2 the levels 513^2 ... 65^2 were pure latency
// with the GPU idle (~27 % of a preconditioner application), and the two large levels moved every vector through HBM once per
// sweep.  Here a workgroup owns a tile of MGT_TS x MGT_TS nodes, loads the tile plus a halo into LDS once and runs the whole
// sequence there, each stage on a region one node smaller than the one before (temporal blocking: the halo is recomputed instead
// of exchanged):
//     k_mg_pre :  [level 0: stage 1 of the block preconditioner -- S^-1 r_p, f = r_v - A_vp z_p --] first sweep from the zero
//                 guess, NS - 1 Chebyshev sweeps, residual, full-weighting restriction  ->  iterate (this level), rhs (coarse level)
//     k_mg_post:  prolongation + correction, NS Chebyshev sweeps                        ->  iterate (this level, or z on level 0)
// Every thread owns the same node(s) of the region in every stage and keeps in REGISTERS what its rows need through all of them:
// the six coefficients, 1 / diagonal and right-hand side of the z- and the x-row it evaluates (its own, or its master's if the
// node is slaved) and that row's previous iterate for the Chebyshev momentum term.  LDS holds two iterates (22-72 KB per
// workgroup; the viscosity planes pass through one of them once, for the coefficients) and the spacing tables of the region, so
// rectilinear grids work.  Measured on the way (2049^2, level 1 = 1025^2, pre kernel): with viscosities, tables, right-hand side
// and three iterates in LDS and the coefficients recomputed in every stage 73 us; coefficients in registers 64; right-hand side
// requested with the first loads instead of after the first barrier 52 (32 x 32 tiles) -- of which the loads are 22, coefficients
// and first sweep 8, a sweep 6, residual + restriction 10.  Tile edge 32 from 10^6 nodes (halo recomputation 1.7x instead of
// 2.6x at three sweeps), 16 below (more workgroups than CUs down to 257^2).
// Row classes (wall, slaved, interior) are decided per node from the global index exactly as the one-kernel-per-stage path does,
// and a slaved node evaluates its master's update -- so both paths compute the same numbers
// (tests/test_hip_solve.py::test_fused_levels_match_the_staged_path).
// Requirements (vcycle_fused_ok): one rank, FP64 level, no stabilisation terms, 1 <= sweeps <= 3; otherwise the staged path runs.
// =========================================================================================
#define MGT_TS 16
#ifndef MGT_NT
#define MGT_NT 1024
#endif
struct MgTileArgs {
    PlVvOp op;                         // this (fine) level
    PlVvOp opc;                        // the coarse level: geometry and row classes
    const double* f; double* fout;     // right-hand side of this level: read (levels >= 1) | written (level 0: computed from rs)
    double* v;                         // pre: iterate after the pre-smoothing sequence (written); post: the same (read)
    double* fc;                        // pre: right-hand side of the coarse level (written)
    const double* ec;                  // post: coarse correction
    double* out; double oscale; long long out_plane;      // post: result, two planes out_plane doubles apart
    double c1[4], c2[4];               // Chebyshev coefficients of the sweeps of the sequence
    PlStokesOp sop; const double* rs; double* z;           // level 0 pre: the preconditioner's stage 1
    int tiles_x;
};

struct MgtTab { const double* es; const double* en; const double* rdz; const double* rDz; const double* rdx; const double* rDx; };

// row classes as vv_cls_z / vv_cls_x; a node outside the grid (tile halo beyond the walls) is a zero row
__device__ inline int mgt_cls_z(const PlVvOp& op, int i, int j, int& db, double& s) {
    db = 0; s = 1.0;
    if (i <= 0 || i >= op.g.nz - 1 || j < 0 || j >= op.g.nx - 1) return VV_ZERO;
    if (op.slave_x) {
        if (j == 0) { db = 1; return VV_SLAVE; }
        if (j == op.g.nx - 2) { db = -1; return VV_SLAVE; }
    }
    return VV_INT;
}
__device__ inline int mgt_cls_x(const PlVvOp& op, int i, int j, int& da, double& s) {
    da = 0; s = 1.0;
    if (j <= 0 || j >= op.g.nx - 1 || i < 0 || i >= op.g.nz - 1) return VV_ZERO;
    if (i == 0 && op.slave_z0) { da = 1; s = op.s0; return VV_SLAVE; }
    if (i == op.g.nz - 2 && op.slave_zL) { da = -1; s = op.sL; return VV_SLAVE; }
    return VV_INT;
}
// (A_vv v)_z, (A_vv v)_x and -diagonal at region node (a, b) = LDS index c (row pitch CR): vv_row_z / vv_row_x on LDS arrays.
// The tables are stored shifted by one: rdz[a + 1] belongs to region row a.
template <bool NEED_A>
__device__ inline void mgt_row_z(const MgtTab& t, const int CR, const double* vz, const double* vx, int c, int a, int b, double& Av, double& dg) {
    const double rdz_i = t.rdz[a + 1], rdz_m = t.rdz[a], rDz_i = t.rDz[a + 1];
    const double rdx_j = t.rdx[b + 1], rDx_j = t.rDx[b + 1], rDx_p = t.rDx[b + 2];
    const double esC = t.es[c], esE = t.es[c + 1];
    const double cN = 4.0 * t.en[c] * rdz_i * rDz_i, cS = 4.0 * t.en[c - CR] * rdz_m * rDz_i;
    const double cE = 2.0 * esE * rDx_p * rdx_j, cW = 2.0 * esC * rDx_j * rdx_j;
    dg = cN + cS + cE + cW;
    if (NEED_A) {
        const double xE = 2.0 * esE * rDz_i * rdx_j, xW = 2.0 * esC * rDz_i * rdx_j;
        const double v0 = vz[c];
        Av = cN * (vz[c + CR] - v0) - cS * (v0 - vz[c - CR]) + cE * (vz[c + 1] - v0) - cW * (v0 - vz[c - 1]) +
             xE * (vx[c + 1] - vx[c - CR + 1]) - xW * (vx[c] - vx[c - CR]);
    }
}
template <bool NEED_A>
__device__ inline void mgt_row_x(const MgtTab& t, const int CR, const double* vz, const double* vx, int c, int a, int b, double& Av, double& dg) {
    const double rdx_j = t.rdx[b + 1], rdx_m = t.rdx[b], rDx_j = t.rDx[b + 1];
    const double rdz_i = t.rdz[a + 1], rDz_i = t.rDz[a + 1], rDz_p = t.rDz[a + 2];
    const double esC = t.es[c], esN = t.es[c + CR];
    const double cE = 4.0 * t.en[c] * rdx_j * rDx_j, cW = 4.0 * t.en[c - 1] * rdx_m * rDx_j;
    const double cN = 2.0 * esN * rDz_p * rdz_i, cS = 2.0 * esC * rDz_i * rdz_i;
    dg = cE + cW + cN + cS;
    if (NEED_A) {
        const double zN = 2.0 * esN * rDx_j * rdz_i, zS = 2.0 * esC * rDx_j * rdz_i;
        const double v0 = vx[c];
        Av = cE * (vx[c + 1] - v0) - cW * (v0 - vx[c - 1]) + cN * (vx[c + CR] - v0) - cS * (v0 - vx[c - CR]) +
             zN * (vz[c + CR] - vz[c + CR - 1]) - zS * (vz[c] - vz[c - 1]);
    }
}

// everything the stages of one tile share
struct MgtTile {
    int ci0, cj0;                      // global node of region element (0, 0)
    int CR, NR;                        // row pitch (= columns) and rows of the region
    MgtTab t;
    // first sweep from the zero guess (cheb_first_node): v1 = -c2 f / diag, slaves copy s x their master's value
    __device__ inline void first(const PlVvOp& op, const double* fz, const double* fx, double* oz, double* ox, double c2, int a, int b) const {
        const int i = ci0 + a, j = cj0 + b, c = a * CR + b;
        int d; double s, Av, dg;
        double o = 0.0;
        int cls = mgt_cls_z(op, i, j, d, s);
        if (cls != VV_ZERO && b + d >= 1 && b + d <= CR - 2) { mgt_row_z<false>(t, CR, nullptr, nullptr, c + d, a, b + d, Av, dg); o = (-s * c2 * fz[c + d]) * pl_rcp(dg); }
        oz[c] = o;
        o = 0.0;
        cls = mgt_cls_x(op, i, j, d, s);
        if (cls != VV_ZERO && a + d >= 1 && a + d <= NR - 2) { mgt_row_x<false>(t, CR, nullptr, nullptr, c + d * CR, a + d, b, Av, dg); o = (-s * c2 * fx[c + d * CR]) * pl_rcp(dg); }
        ox[c] = o;
    }
    // one Chebyshev sweep (cheb_node); pz == nullptr: the previous iterate is zero
    __device__ inline void cheb(const PlVvOp& op, const double* vz, const double* vx, const double* pz, const double* px, const double* fz,
                                const double* fx, double c1, double c2, int a, int b, double& oz, double& ox) const {
        const int i = ci0 + a, j = cj0 + b, c = a * CR + b;
        int d; double s, Av, dg;
        oz = 0.0; ox = 0.0;
        int cls = mgt_cls_z(op, i, j, d, s);
        if (cls != VV_ZERO && b + d >= 1 && b + d <= CR - 2) {
            const int cm = c + d;
            mgt_row_z<true>(t, CR, vz, vx, cm, a, b + d, Av, dg);
            const double v0 = vz[cm], mom = (c1 != 0.0) ? c1 * (v0 - (pz ? pz[cm] : 0.0)) : 0.0;
            oz = s * (v0 + mom + (c2 * (Av - fz[cm])) * pl_rcp(dg));
        }
        cls = mgt_cls_x(op, i, j, d, s);
        if (cls != VV_ZERO && a + d >= 1 && a + d <= NR - 2) {
            const int cm = c + d * CR;
            mgt_row_x<true>(t, CR, vz, vx, cm, a + d, b, Av, dg);
            const double v0 = vx[cm], mom = (c1 != 0.0) ? c1 * (v0 - (px ? px[cm] : 0.0)) : 0.0;
            ox = s * (v0 + mom + (c2 * (Av - fx[cm])) * pl_rcp(dg));
        }
    }
};

// the tables of the region: value of table tab (indexed by global node + PL_TOFF, defined for -PL_TOFF .. n + PL_TOFF) at region rows / columns -1 .. CR + 1
__device__ inline void mgt_load_table(double* dst, const double* tab, int first_global, int count, int n) {
    for (int k = threadIdx.x; k < count; k += blockDim.x) {
        const int gk = first_global + k;
        dst[k] = (gk >= -PL_TOFF && gk < n + PL_TOFF) ? TB(tab, gk) : 0.0;
    }
}

// What a thread keeps in REGISTERS for one component of one node of its tile through all the stages: the coefficients of the row
// it evaluates (its own, or its master's if it is a slaved node), 1 / diagonal and the right-hand side there.  The stages were
// VALU-bound -- ~150 instructions per node and stage, two thirds of them recomputing these from the viscosity planes and spacing
// tables in LDS (the level-1 visit took as long as the nine separate launches it replaces) -- now a sweep is 18 LDS reads and ~45 flops.
// z: c0..c3 = cN cS cE cW, x0 x1 = xE xW;  x: c0..c3 = cE cW cN cS, x0 x1 = zN zS  (mgt_row_z / mgt_row_x: same expressions, same order).
struct MgtC { double c0, c1, c2, c3, x0, x1, rdg, s, f; int cm, d; bool on, interior; };
// first half (integers only, before anything is in LDS): which row the node evaluates -- so that the right-hand side there can be
// requested together with the region's viscosities
template <bool ZC>
__device__ inline MgtC mgt_classify(const PlVvOp& op, int CR, int NR, int ci0, int cj0, int a, int b) {
    MgtC k; k.c0 = k.c1 = k.c2 = k.c3 = k.x0 = k.x1 = k.rdg = k.f = 0.0; k.s = 1.0; k.cm = a * CR + b; k.d = 0; k.on = false; k.interior = false;
    if (a < 1 || a > NR - 2 || b < 1 || b > CR - 2) return k;
    int d; double s;
    const int cls = ZC ? mgt_cls_z(op, ci0 + a, cj0 + b, d, s) : mgt_cls_x(op, ci0 + a, cj0 + b, d, s);
    const int am = ZC ? a : a + d, bm = ZC ? b + d : b;
    if (cls == VV_ZERO || am < 1 || am > NR - 2 || bm < 1 || bm > CR - 2) return k;
    k.s = s; k.cm = am * CR + bm; k.d = d; k.on = true; k.interior = cls == VV_INT;
    return k;
}
// second half: the coefficients of that row from the viscosity planes and spacing tables in LDS
template <bool ZC>
__device__ inline void mgt_coefficients(MgtC& k, const MgtTab& t, int CR) {
    if (!k.on) return;
    const int c = k.cm, am = c / CR, bm = c - am * CR;
    if (ZC) {
        const double rdz_i = t.rdz[am + 1], rdz_m = t.rdz[am], rDz_i = t.rDz[am + 1];
        const double rdx_j = t.rdx[bm + 1], rDx_j = t.rDx[bm + 1], rDx_p = t.rDx[bm + 2];
        const double esC = t.es[c], esE = t.es[c + 1];
        k.c0 = 4.0 * t.en[c] * rdz_i * rDz_i; k.c1 = 4.0 * t.en[c - CR] * rdz_m * rDz_i;
        k.c2 = 2.0 * esE * rDx_p * rdx_j; k.c3 = 2.0 * esC * rDx_j * rdx_j;
        k.x0 = 2.0 * esE * rDz_i * rdx_j; k.x1 = 2.0 * esC * rDz_i * rdx_j;
    } else {
        const double rdx_j = t.rdx[bm + 1], rdx_m = t.rdx[bm], rDx_j = t.rDx[bm + 1];
        const double rdz_i = t.rdz[am + 1], rDz_i = t.rDz[am + 1], rDz_p = t.rDz[am + 2];
        const double esC = t.es[c], esN = t.es[c + CR];
        k.c0 = 4.0 * t.en[c] * rdx_j * rDx_j; k.c1 = 4.0 * t.en[c - 1] * rdx_m * rDx_j;
        k.c2 = 2.0 * esN * rDz_p * rdz_i; k.c3 = 2.0 * esC * rDz_i * rdz_i;
        k.x0 = 2.0 * esN * rDx_j * rdz_i; k.x1 = 2.0 * esC * rDx_j * rdz_i;
    }
    k.rdg = pl_rcp(k.c0 + k.c1 + k.c2 + k.c3);
}
__device__ inline double mgt_dg(const MgtC& k) { return k.c0 + k.c1 + k.c2 + k.c3; }
// (A_vv v) of the row, and v at its node
__device__ inline double mgt_av_z(const MgtC& k, int CR, const double* vz, const double* vx, double& v0) {
    const int c = k.cm; v0 = vz[c];
    return k.c0 * (vz[c + CR] - v0) - k.c1 * (v0 - vz[c - CR]) + k.c2 * (vz[c + 1] - v0) - k.c3 * (v0 - vz[c - 1]) +
           k.x0 * (vx[c + 1] - vx[c - CR + 1]) - k.x1 * (vx[c] - vx[c - CR]);
}
__device__ inline double mgt_av_x(const MgtC& k, int CR, const double* vz, const double* vx, double& v0) {
    const int c = k.cm; v0 = vx[c];
    return k.c0 * (vx[c + 1] - v0) - k.c1 * (v0 - vx[c - 1]) + k.c2 * (vx[c + CR] - v0) - k.c3 * (v0 - vx[c - CR]) +
           k.x0 * (vz[c + CR] - vz[c + CR - 1]) - k.x1 * (vz[c] - vz[c - 1]);
}
// one Chebyshev sweep at a node (cheb_node).  pvz / pvx: the row's previous iterate, carried in registers from sweep to sweep (every
// thread evaluates the same row in every stage, a slaved node its master's) -- so LDS holds two iterates, not three
__device__ inline void mgt_cheb(const MgtC& kz, const MgtC& kx, int CR, const double* vz, const double* vx, double& pvz, double& pvx,
                                double c1, double c2, double& oz, double& ox) {
    oz = 0.0; ox = 0.0;
    if (kz.on) {
        double v0; const double Av = mgt_av_z(kz, CR, vz, vx, v0);
        const double mom = (c1 != 0.0) ? c1 * (v0 - pvz) : 0.0;
        oz = kz.s * (v0 + mom + (c2 * (Av - kz.f)) * kz.rdg);
        pvz = v0;
    }
    if (kx.on) {
        double v0; const double Av = mgt_av_x(kx, CR, vz, vx, v0);
        const double mom = (c1 != 0.0) ? c1 * (v0 - pvx) : 0.0;
        ox = kx.s * (v0 + mom + (c2 * (Av - kx.f)) * kx.rdg);
        pvx = v0;
    }
}
// Node (ra, rb) of a region (global node i, j) belongs to the tile: its TS x TS nodes -- and the grid's LAST row / column where the tile
// ends right in front of it.  Those nodes are zero rows of both components on every level; a tile row and column of their own
// (33 x 33 tiles for 1025^2 nodes: 1089 workgroups, a fifth round of the 256 one-per-CU slots for 65 of them) cost level 1 a
// fifth of its time.
__device__ inline bool mgt_owns(int ra, int rb, int HC, int TS, int i, int j, const PlGeom& g) {
    return ra >= HC && (ra < HC + TS || (ra == HC + TS && i == g.nz - 1)) && rb >= HC && (rb < HC + TS || (rb == HC + TS && j == g.nx - 1));
}
// threads of a tile kernel: one node of the region per thread where the region has at most 1024, else the fewest passes
__host__ __device__ constexpr int mgt_npt(int nn) { return (nn + 1023) / 1024; }
__host__ __device__ constexpr int mgt_nt(int nn) { return ((nn + mgt_npt(nn) - 1) / mgt_npt(nn) + 63) / 64 * 64; }
__host__ __device__ constexpr int mgt_pre_nn(int NS, int TS) { return (TS + 2 * (NS + 2)) * (TS + 2 * (NS + 2)); }
__host__ __device__ constexpr int mgt_post_nn(int NS, int TS) { return (TS + 2 * (NS + 1)) * (TS + 2 * (NS + 1)); }

// NS: sweeps of the pre-smoothing sequence, the first from the zero guess included (1..3).  L0: level 0 of the Stokes
// preconditioner -- the right-hand side is computed from the scaled residual rs (stage1_node) instead of read.
// EXT = 1 (a level DISTRIBUTED over several ranks): the region is one node wider, so that the pre-smoothed iterate comes out valid NS
// nodes around the tile -- the tiles at the rim of the rank's block write it into the halo ring as well (the same numbers the
// neighbour computes for its own nodes), and k_mg_post finds there what it needs without an exchange of the iterate.
template <int NS, bool L0, int TS, int EXT = 0>
__global__ __launch_bounds__(mgt_nt(mgt_pre_nn(NS + EXT, TS))) void k_mg_pre(MgTileArgs a) {
    constexpr int H = NS + 1 + EXT, HC = H + 1, CR = TS + 2 * HC, NN = CR * CR, NPT = mgt_npt(NN), NT = mgt_nt(NN);
    __shared__ double V[2][2][NN];                  // two iterates; V[1] holds the viscosity planes until the rows are set up
    __shared__ double ZP[L0 ? NN : 1];
    __shared__ double TAB[4][CR + 4];
    double* const ES = V[1][0]; double* const EN = V[1][1];
    const PlGeom& g = a.op.g;
    const int tid = threadIdx.x;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x % a.tiles_x;
    const int ti0 = g.gi0 + ty * TS, tj0 = g.gj0 + tx * TS;          // tiles cover this rank's block (global node indices)
    const int ci0 = ti0 - HC, cj0 = tj0 - HC;
    MgtTab t; t.es = ES; t.en = EN; t.rdz = TAB[0]; t.rDz = TAB[1]; t.rdx = TAB[2]; t.rDx = TAB[3];
    mgt_load_table(TAB[0], a.op.rdz, ci0 - 1, CR + 3, g.nz); mgt_load_table(TAB[1], a.op.rDz, ci0 - 1, CR + 3, g.nz);
    mgt_load_table(TAB[2], a.op.rdx, cj0 - 1, CR + 3, g.nx); mgt_load_table(TAB[3], a.op.rDx, cj0 - 1, CR + 3, g.nx);
    const long long P = g.plane;
    // ---- load: viscosities (level 0: and S^-1 r_p) of the whole region, 0 beyond the planes' ring -- and in the same breath the
    //      right-hand side (level 0: the scaled residual) of the row each node evaluates: its own, a slaved node its master's
    MgtC kz[NPT], kx[NPT];
    int depth[NPT];                                 // distance of the node from the border of the region; -1: no node
#pragma unroll
    for (int q = 0; q < NPT; q++) {
        const int idx = tid + q * NT;
        const bool have = idx < NN;
        const int ra = have ? idx / CR : 0, rb = have ? idx % CR : 0, i = ci0 + ra, j = cj0 + rb;
        depth[q] = have ? min(min(ra, CR - 1 - ra), min(rb, CR - 1 - rb)) : -1;
        kz[q] = mgt_classify<true>(a.op, CR, CR, ci0, cj0, ra, rb);
        kx[q] = mgt_classify<false>(a.op, CR, CR, ci0, cj0, ra, rb);
        const bool inmem = i - g.gi0 >= -PL_RING && i - g.gi0 < g.lnz + PL_RING && j - g.gj0 >= -PL_RING && j - g.gj0 < g.lnx + PL_RING;
        const long long c = pl_idx(g, i - g.gi0, j - g.gj0);
        const double* const fsrc = L0 ? a.rs : a.f;
        if (kz[q].on) kz[q].f = fsrc[c + kz[q].d];
        if (kx[q].on) kx[q].f = fsrc[c + (long long)kx[q].d * g.pitch + P];
        if (have) {
            ES[idx] = inmem ? a.op.etas[c] : 0.0; EN[idx] = inmem ? a.op.etan[c] : 0.0;
            if (L0) {
                const bool indom = i >= 0 && i < g.nz && j >= 0 && j < g.nx;
                ZP[idx] = indom ? prec_p_value(a.sop, a.rs + 2 * P, i, j, c) : 0.0;
            }
        }
    }
    __syncthreads();
    // ---- the coefficients of those rows.  Level 0 turns the scaled residual into the right-hand side (stage1_node):
    //      f = r_v - A_vp z_p on the interior momentum rows; un-scaling a row = multiplying by the sum of its four own-component
    //      coefficients (= the diagonal the sweeps use).
#pragma unroll
    for (int q = 0; q < NPT; q++) {
        mgt_coefficients<true>(kz[q], t, CR); mgt_coefficients<false>(kx[q], t, CR);
        if (L0) {
            const int idx = tid + q * NT;
            const int ra = depth[q] >= 0 ? idx / CR : 0, rb = depth[q] >= 0 ? idx % CR : 0, i = ci0 + ra, j = cj0 + rb;
            if (kz[q].on) kz[q].f = kz[q].f * mgt_dg(kz[q]) + 2.0 * a.sop.Kc * TAB[1][ra + 1] * (ZP[kz[q].cm] - ZP[kz[q].cm - CR]);
            if (kx[q].on) kx[q].f = kx[q].f * mgt_dg(kx[q]) + 2.0 * a.sop.Kc * TAB[3][rb + 1] * (ZP[kx[q].cm] - ZP[kx[q].cm - 1]);
            if (depth[q] >= 0 && mgt_owns(ra, rb, HC, TS, i, j, g) && i < g.nz && j < g.nx && i - g.gi0 < g.lnz && j - g.gj0 < g.lnx) {      // the tile itself: keep f and z_p
                const long long c = pl_idx(g, i - g.gi0, j - g.gj0);
                a.fout[c] = kz[q].interior ? kz[q].f : 0.0; a.fout[c + P] = kx[q].interior ? kx[q].f : 0.0; a.z[c + 2 * P] = ZP[idx];
            }
        }
    }
    // ---- first sweep from the zero guess on the region +-H (cheb_first_node): v1 = -c2 f / diag, slaves s x their master's value
    int cur = 0;
#pragma unroll
    for (int q = 0; q < NPT; q++) {
        const int idx = tid + q * NT;
        if (depth[q] >= HC - H) {
            V[0][0][idx] = kz[q].on ? (-kz[q].s * a.c2[0] * kz[q].f) * kz[q].rdg : 0.0;
            V[0][1][idx] = kx[q].on ? (-kx[q].s * a.c2[0] * kx[q].f) * kx[q].rdg : 0.0;
        }
    }
    __syncthreads();
    // ---- Chebyshev sweeps, each on a region one node smaller
    double pvz[NPT], pvx[NPT];
#pragma unroll
    for (int q = 0; q < NPT; q++) { pvz[q] = 0.0; pvx[q] = 0.0; }
#pragma unroll
    for (int k = 1; k < NS; k++) {
        const int h = H - k;
#pragma unroll
        for (int q = 0; q < NPT; q++) {
            const int idx = tid + q * NT;
            if (depth[q] >= HC - h) {
                double oz, ox;
                mgt_cheb(kz[q], kx[q], CR, V[cur][0], V[cur][1], pvz[q], pvx[q], a.c1[k], a.c2[k], oz, ox);
                V[cur ^ 1][0][idx] = oz; V[cur ^ 1][1][idx] = ox;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    // ---- residual on the region +-1 (interior rows; 0 elsewhere) into the free buffer; the tile's iterate goes to memory
    const int nxt = cur ^ 1;
#pragma unroll
    for (int q = 0; q < NPT; q++) {
        const int idx = tid + q * NT;
        if (depth[q] >= HC - 1) {
            double v0, rz = 0.0, rx = 0.0;
            if (kz[q].interior) rz = kz[q].f - mgt_av_z(kz[q], CR, V[cur][0], V[cur][1], v0);
            if (kx[q].interior) rx = kx[q].f - mgt_av_x(kx[q], CR, V[cur][0], V[cur][1], v0);
            V[nxt][0][idx] = rz; V[nxt][1][idx] = rx;
        }
        // the iterate goes to memory: the tile's own nodes, and (EXT) the nodes of the halo ring NS deep around the rank's block
        if (depth[q] >= HC - (EXT ? NS : 1)) {
            const int ra = idx / CR, rb = idx % CR, i = ci0 + ra, j = cj0 + rb, li = i - g.gi0, lj = j - g.gj0;
            const bool owned = li >= 0 && li < g.lnz && lj >= 0 && lj < g.lnx;
            const bool ring = EXT && !owned && li >= -NS && li < g.lnz + NS && lj >= -NS && lj < g.lnx + NS;
            if (i >= 0 && i < g.nz && j >= 0 && j < g.nx && ((owned && mgt_owns(ra, rb, HC, TS, i, j, g)) || ring)) {
                const long long c = pl_idx(g, li, lj);
                a.v[c] = V[cur][0][idx]; a.v[c + P] = V[cur][1][idx];
            }
        }
    }
    __syncthreads();
    // ---- full-weighting restriction (restrict_node) onto the coarse nodes of the tile
    const PlGeom& gc = a.opc.g;
    for (int idx = tid; idx < (TS / 2 + 1) * (TS / 2 + 1); idx += NT) {          // (+ 1: the coarse grid's last row / column, as mgt_owns)
        const int u_ = idx / (TS / 2 + 1), w_ = idx % (TS / 2 + 1);
        const int I = (ti0 >> 1) + u_, J = (tj0 >> 1) + w_;
        if ((u_ == TS / 2 && I != gc.nz - 1) || (w_ == TS / 2 && J != gc.nx - 1)) continue;
        if (I >= gc.nz || J >= gc.nx || 2 * I - g.gi0 >= g.lnz || 2 * J - g.gj0 >= g.lnx) continue;      // (only the coarse nodes of this rank's block)
        const int b0 = (2 * I - ci0) * CR + (2 * J - cj0);
        int d; double s;
        double oz = 0.0, ox = 0.0;
        if (vv_cls_z(a.opc, I, J, d, s) == VV_INT) {
            const double wz[3] = {0.25, 0.5, 0.25}, wx[4] = {0.125, 0.375, 0.375, 0.125};
#pragma unroll
            for (int u = 0; u < 3; u++)
#pragma unroll
                for (int w = 0; w < 4; w++) oz += wz[u] * wx[w] * V[nxt][0][b0 + (u - 1) * CR + (w - 1)];
        }
        if (vv_cls_x(a.opc, I, J, d, s) == VV_INT) {
            const double wz[4] = {0.125, 0.375, 0.375, 0.125}, wx[3] = {0.25, 0.5, 0.25};
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int w = 0; w < 3; w++) ox += wz[u] * wx[w] * V[nxt][1][b0 + (u - 1) * CR + (w - 1)];
        }
        const long long cc = pl_idx(gc, I - gc.gi0, J - gc.gj0);
        a.fc[cc] = oz; a.fc[cc + gc.plane] = ox;
    }
}

// NS: sweeps of the post-smoothing sequence (1..3); the last one writes the tile to `out` (times oscale)
template <int NS, int TS>
__global__ __launch_bounds__(mgt_nt(mgt_post_nn(NS, TS))) void k_mg_post(MgTileArgs a) {
    constexpr int H = NS, HC = H + 1, CR = TS + 2 * HC, NN = CR * CR, ECR = TS / 2 + H + 6, ENN = ECR * ECR, NPT = mgt_npt(NN), NT = mgt_nt(NN);
    __shared__ double V[2][2][NN];                  // V[1]: the pre-smoothed iterate; V[0]: the viscosity planes until the rows are set up
    __shared__ double EC[2][ENN];
    __shared__ double TAB[4][CR + 4];
    double* const ES = V[0][0]; double* const EN = V[0][1];
    const PlGeom& g = a.op.g; const PlGeom& gc = a.opc.g;
    const int tid = threadIdx.x;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x % a.tiles_x;
    const int ti0 = g.gi0 + ty * TS, tj0 = g.gj0 + tx * TS;
    const int ci0 = ti0 - HC, cj0 = tj0 - HC;
    MgtTab t; t.es = ES; t.en = EN; t.rdz = TAB[0]; t.rDz = TAB[1]; t.rdx = TAB[2]; t.rDx = TAB[3];
    mgt_load_table(TAB[0], a.op.rdz, ci0 - 1, CR + 3, g.nz); mgt_load_table(TAB[1], a.op.rDz, ci0 - 1, CR + 3, g.nz);
    mgt_load_table(TAB[2], a.op.rdx, cj0 - 1, CR + 3, g.nx); mgt_load_table(TAB[3], a.op.rDx, cj0 - 1, CR + 3, g.nx);
    const long long P = g.plane;
    const int eI0 = ((ti0 - H) >> 1) - 2, eJ0 = ((tj0 - H) >> 1) - 2;      // coarse node of EC element (0, 0)
    MgtC kz[NPT], kx[NPT];
    int depth[NPT];
#pragma unroll
    for (int q = 0; q < NPT; q++) {
        const int idx = tid + q * NT;
        const bool have = idx < NN;
        const int ra = have ? idx / CR : 0, rb = have ? idx % CR : 0, i = ci0 + ra, j = cj0 + rb;
        depth[q] = have ? min(min(ra, CR - 1 - ra), min(rb, CR - 1 - rb)) : -1;
        kz[q] = mgt_classify<true>(a.op, CR, CR, ci0, cj0, ra, rb);
        kx[q] = mgt_classify<false>(a.op, CR, CR, ci0, cj0, ra, rb);
        const bool inmem = i - g.gi0 >= -PL_RING && i - g.gi0 < g.lnz + PL_RING && j - g.gj0 >= -PL_RING && j - g.gj0 < g.lnx + PL_RING;
        const long long c = pl_idx(g, i - g.gi0, j - g.gj0);
        if (kz[q].on) kz[q].f = a.f[c + kz[q].d];
        if (kx[q].on) kx[q].f = a.f[c + (long long)kx[q].d * g.pitch + P];
        if (have) {
            ES[idx] = inmem ? a.op.etas[c] : 0.0; EN[idx] = inmem ? a.op.etan[c] : 0.0;
            V[1][0][idx] = inmem ? a.v[c] : 0.0; V[1][1][idx] = inmem ? a.v[c + P] : 0.0;     // the pre-smoothed iterate
        }
    }
    for (int idx = tid; idx < ENN; idx += NT) {
        const int I = eI0 + idx / ECR, J = eJ0 + idx % ECR;
        const bool inmem = I - gc.gi0 >= -PL_RING && I - gc.gi0 < gc.lnz + PL_RING && J - gc.gj0 >= -PL_RING && J - gc.gj0 < gc.lnx + PL_RING;
        const long long cc = pl_idx(gc, I - gc.gi0, J - gc.gj0);
        EC[0][idx] = inmem ? a.ec[cc] : 0.0; EC[1][idx] = inmem ? a.ec[cc + gc.plane] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NPT; q++) { mgt_coefficients<true>(kz[q], t, CR); mgt_coefficients<false>(kx[q], t, CR); }
    __syncthreads();                                // the viscosity planes have been read: V[0] is free
    // ---- prolongation + correction (prolong_node) on the region +-H: V[0] = s (v[master] + P e at the master)
    auto ecz = [&](int I, int J) { return EC[0][(I - eI0) * ECR + (J - eJ0)]; };
    auto ecx = [&](int I, int J) { return EC[1][(I - eI0) * ECR + (J - eJ0)]; };
#pragma unroll
    for (int q = 0; q < NPT; q++) {
        const int idx = tid + q * NT;
        if (depth[q] >= HC - H) {
            const int ra = idx / CR, rb = idx % CR, i = ci0 + ra, j = cj0 + rb;
            int d; double s;
            double oz = 0.0, ox = 0.0;
            if (mgt_cls_z(a.op, i, j, d, s) != VV_ZERO) {
                const int jm = j + d;
                const int I0 = i >> 1, I1 = (i + 1) >> 1;
                int Jn = jm >> 1, Jo = (jm & 1) ? Jn + 1 : Jn - 1;
                Jn = min(max(Jn, 0), gc.nx - 2); Jo = min(max(Jo, 0), gc.nx - 2);
                const double pa = 0.5 * (ecz(I0, Jn) + ecz(I1, Jn)), pb = 0.5 * (ecz(I0, Jo) + ecz(I1, Jo));
                oz = s * (V[1][0][idx + d] + (0.75 * pa + 0.25 * pb));
            }
            if (mgt_cls_x(a.op, i, j, d, s) != VV_ZERO) {
                const int im = i + d;
                const int J0 = j >> 1, J1 = (j + 1) >> 1;
                int In = im >> 1, Io = (im & 1) ? In + 1 : In - 1;
                In = min(max(In, 0), gc.nz - 2); Io = min(max(Io, 0), gc.nz - 2);
                const double pa = 0.5 * (ecx(In, J0) + ecx(In, J1)), pb = 0.5 * (ecx(Io, J0) + ecx(Io, J1));
                ox = s * (V[1][1][idx + d * CR] + (0.75 * pa + 0.25 * pb));
            }
            V[0][0][idx] = oz; V[0][1][idx] = ox;
        }
    }
    __syncthreads();
    // ---- Chebyshev sweeps (the first with c1 = 0); the last one covers the tile only and goes to memory
    int cur = 0;
    double pvz[NPT], pvx[NPT];
#pragma unroll
    for (int q = 0; q < NPT; q++) { pvz[q] = 0.0; pvx[q] = 0.0; }
#pragma unroll
    for (int k = 0; k < NS; k++) {
        const int h = H - 1 - k;
#pragma unroll
        for (int q = 0; q < NPT; q++) {
            const int idx = tid + q * NT;
            if (depth[q] >= HC - h) {
                double oz, ox;
                mgt_cheb(kz[q], kx[q], CR, V[cur][0], V[cur][1], pvz[q], pvx[q], k == 0 ? 0.0 : a.c1[k], a.c2[k], oz, ox);
                if (k == NS - 1) {
                    const int ra = idx / CR, rb = idx % CR, i = ci0 + ra, j = cj0 + rb;
                    if (i < g.nz && j < g.nx && i - g.gi0 < g.lnz && j - g.gj0 < g.lnx) {
                        const long long c = pl_idx(g, i - g.gi0, j - g.gj0); a.out[c] = oz * a.oscale; a.out[c + a.out_plane] = ox * a.oscale;
                    }
                } else { V[cur ^ 1][0][idx] = oz; V[cur ^ 1][1][idx] = ox; }
            }
        }
        if (k < NS - 1) { __syncthreads(); cur ^= 1; }
    }
    // the grid's last row / column next to the tile (mgt_owns): zero rows
#pragma unroll
    for (int q = 0; q < NPT; q++) {
        const int idx = tid + q * NT;
        if (depth[q] == HC - 1) {
            const int ra = idx / CR, rb = idx % CR, i = ci0 + ra, j = cj0 + rb;
            if (mgt_owns(ra, rb, HC, TS, i, j, g) && i - g.gi0 < g.lnz && j - g.gj0 < g.lnx) {
                const long long c = pl_idx(g, i - g.gi0, j - g.gj0); a.out[c] = 0.0; a.out[c + a.out_plane] = 0.0;
            }
        }
    }
}

// ---- LDS-resident coarse tail -----------------------------------------------------------------------------------------
// The V-cycle of all levels <= 33^2 in ONE workgroup, like k_mg_tail -- but every level lives in LDS for the whole kernel (three
// rotating iterates, right-hand side, viscosities, spacing tables; 147 KB for 33^2 + 17^2 + 9^2 + 5^2) and the stages are the tile
// kernels' node functions on named __shared__ arrays.  k_mg_tail's 44 stages cost 2.3 us each whatever the level size (101 us per
// preconditioner application, 17 % of it): generic node functions through flat pointers into global memory.  Same arithmetic, same
// stage order: the result equals k_mg_tail's to rounding (test_fused_levels_match_the_staged_path runs both).
#define MGT_TAIL_POOL 18600          // doubles
// A level's arrays are addressed as pool + offset (never through stored pointers: those would be flat accesses, not ds_read).
struct TailLvl { int base, NR, CR, N; };
#define TL_V(D, buf, c) (pool + (D).base + (2 * (buf) + (c)) * (D).N)
#define TL_F(D, c) (pool + (D).base + (6 + (c)) * (D).N)
#define TL_ES(D) (pool + (D).base + 8 * (D).N)
#define TL_EN(D) (pool + (D).base + 9 * (D).N)
#define TL_TAB(D, k) (pool + (D).base + 10 * (D).N + ((k) < 2 ? (k) * ((D).NR + 3) : 2 * ((D).NR + 3) + ((k) - 2) * ((D).CR + 3)))
// nodes that carry an equation or a slave value: i <= nz - 2 and j <= nx - 2 (the last row and column are zero rows of both
// components and stay at the pool's zero) -- 32 x 32 = one pass of the 1024 threads on the 33^2 level
#define TL_FOR_NODES(g)                                                                                                     \
    for (int idx = threadIdx.x, nxm_ = (g).nx - 1, pw_ = (nxm_ & (nxm_ - 1)) == 0, sh_ = 31 - __clz(nxm_), i, j;            \
         idx < ((g).nz - 1) * nxm_ && (i = pw_ ? idx >> sh_ : idx / nxm_, j = idx - i * nxm_, true); idx += MGT_NT)
__device__ inline MgtTile mgt_tail_tile(double* pool, const TailLvl& D) {
    MgtTile T; T.ci0 = -1; T.cj0 = -1; T.CR = D.CR; T.NR = D.NR;
    T.t.es = TL_ES(D); T.t.en = TL_EN(D); T.t.rdz = TL_TAB(D, 0); T.t.rDz = TL_TAB(D, 1); T.t.rdx = TL_TAB(D, 2); T.t.rDx = TL_TAB(D, 3);
    return T;
}
__device__ inline void mgt_tail_smooth(double* pool, const TailLevel& L, const TailLvl& D, int& cur, int& prv, int& nxt, int nsweep, double ratio, bool zero_guess) {
    const double lmax = L.lmax, lmin = lmax / ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho_old = 1.0 / sigma;
    const MgtTile T = mgt_tail_tile(pool, D);
    for (int k = 0; k < nsweep; k++) {
        double c1, c2;
        if (k == 0) { c1 = 0.0; c2 = 1.0 / theta; }
        else { const double rho = 1.0 / (2.0 * sigma - rho_old); c1 = rho * rho_old; c2 = 2.0 * rho / delta; rho_old = rho; }
        double* const nz_ = TL_V(D, nxt, 0); double* const nx_ = TL_V(D, nxt, 1);
        const double* const cz = TL_V(D, cur, 0); const double* const cx = TL_V(D, cur, 1);
        const bool noprev = k == 1 && zero_guess;
        const double* const pz = noprev ? nullptr : TL_V(D, prv, 0); const double* const px = noprev ? nullptr : TL_V(D, prv, 1);
        TL_FOR_NODES(L.op.g) {
            const int ra = i + 1, rb = j + 1;
            if (k == 0 && zero_guess) T.first(L.op, TL_F(D, 0), TL_F(D, 1), nz_, nx_, c2, ra, rb);
            else {
                double oz, ox;
                T.cheb(L.op, cz, cx, pz, px, TL_F(D, 0), TL_F(D, 1), c1, c2, ra, rb, oz, ox);
                nz_[ra * D.CR + rb] = oz; nx_[ra * D.CR + rb] = ox;
            }
        }
        __syncthreads();
        const int o = prv; prv = cur; cur = nxt; nxt = o;
    }
}
__global__ __launch_bounds__(MGT_NT) void k_mg_tail_lds(TailArgs a) {
    __shared__ double pool[MGT_TAIL_POOL];
    __shared__ TailLvl lv[PL_TAIL_MAX_LEVELS];
    __shared__ int rot[PL_TAIL_MAX_LEVELS];          // which buffer holds a level's iterate after the way down
    const int tid = threadIdx.x;
    for (int k = tid; k < MGT_TAIL_POOL; k += MGT_NT) pool[k] = 0.0;          // rings, last row / column and unused corners read as zero
    if (tid == 0) {
        int q = 0;
        for (int l = 0; l < a.nlev; l++) {
            const PlGeom& g = a.L[l].op.g;
            TailLvl D; D.base = q; D.NR = g.nz + 2; D.CR = g.nx + 2; D.N = D.NR * D.CR;
            q += 10 * D.N + 2 * (D.NR + 3) + 2 * (D.CR + 3);
            lv[l] = D;
        }
    }
    __syncthreads();
    // ---- load: viscosities and tables of every level, right-hand side of the first
    for (int l = 0; l < a.nlev; l++) {
        const TailLevel& L = a.L[l]; const PlGeom& g = L.op.g; const TailLvl D = lv[l];
        for (int idx = tid; idx < g.nz * g.nx; idx += MGT_NT) {
            const int i = idx / g.nx, j = idx - i * g.nx;
            const long long c = pl_idx(g, i - g.gi0, j - g.gj0);
            const int o = (i + 1) * D.CR + (j + 1);
            TL_ES(D)[o] = L.op.etas[c]; TL_EN(D)[o] = L.op.etan[c];
            if (l == 0) { TL_F(D, 0)[o] = L.f[c]; TL_F(D, 1)[o] = L.f[c + g.plane]; }
        }
        mgt_load_table(TL_TAB(D, 0), L.op.rdz, -2, D.NR + 3, g.nz); mgt_load_table(TL_TAB(D, 1), L.op.rDz, -2, D.NR + 3, g.nz);
        mgt_load_table(TL_TAB(D, 2), L.op.rdx, -2, D.CR + 3, g.nx); mgt_load_table(TL_TAB(D, 3), L.op.rDx, -2, D.CR + 3, g.nx);
    }
    __syncthreads();
    // ---- down
    for (int l = 0; l < a.nlev; l++) {
        const TailLevel& L = a.L[l]; const PlGeom& g = L.op.g; const TailLvl D = lv[l];
        int cur = 0, prv = 1, nxt = 2;
        if (l == a.nlev - 1) {
            double ratio = 0.4 * g.nz * g.nx; if (ratio < 30.0) ratio = 30.0;
            mgt_tail_smooth(pool, L, D, cur, prv, nxt, a.coarse_sweeps, ratio, true);
            if (tid == 0) rot[l] = cur | (prv << 2) | (nxt << 4);
            break;
        }
        mgt_tail_smooth(pool, L, D, cur, prv, nxt, a.nu_pre, a.ratio, true);
        if (tid == 0) rot[l] = cur | (prv << 2) | (nxt << 4);
        // residual on the interior rows (0 elsewhere) into the free buffer
        const MgtTile T = mgt_tail_tile(pool, D);
        double* const rz = TL_V(D, nxt, 0); double* const rx = TL_V(D, nxt, 1);
        const double* const cz = TL_V(D, cur, 0); const double* const cx = TL_V(D, cur, 1);
        TL_FOR_NODES(g) {
            const int ra = i + 1, rb = j + 1, o = ra * D.CR + rb;
            int d; double s, Av, dg, vz = 0.0, vx = 0.0;
            if (mgt_cls_z(L.op, i, j, d, s) == VV_INT) { mgt_row_z<true>(T.t, D.CR, cz, cx, o, ra, rb, Av, dg); vz = TL_F(D, 0)[o] - Av; }
            if (mgt_cls_x(L.op, i, j, d, s) == VV_INT) { mgt_row_x<true>(T.t, D.CR, cz, cx, o, ra, rb, Av, dg); vx = TL_F(D, 1)[o] - Av; }
            rz[o] = vz; rx[o] = vx;
        }
        __syncthreads();
        // restriction onto the next level's right-hand side (restrict_node)
        const TailLevel& Cn = a.L[l + 1]; const TailLvl Dc = lv[l + 1];
        double* const fz = TL_F(Dc, 0); double* const fx = TL_F(Dc, 1);
        TL_FOR_NODES(Cn.op.g) {
            const int b0 = (2 * i + 1) * D.CR + (2 * j + 1);
            int d; double s, oz = 0.0, ox = 0.0;
            if (vv_cls_z(Cn.op, i, j, d, s) == VV_INT) {
                const double wz[3] = {0.25, 0.5, 0.25}, wx[4] = {0.125, 0.375, 0.375, 0.125};
#pragma unroll
                for (int u = 0; u < 3; u++)
#pragma unroll
                    for (int w = 0; w < 4; w++) oz += wz[u] * wx[w] * rz[b0 + (u - 1) * D.CR + (w - 1)];
            }
            if (vv_cls_x(Cn.op, i, j, d, s) == VV_INT) {
                const double wz[4] = {0.125, 0.375, 0.375, 0.125}, wx[3] = {0.25, 0.5, 0.25};
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int w = 0; w < 3; w++) ox += wz[u] * wx[w] * rx[b0 + (u - 1) * D.CR + (w - 1)];
            }
            const int oc = (i + 1) * Dc.CR + (j + 1);
            fz[oc] = oz; fx[oc] = ox;
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- up
    for (int l = a.nlev - 2; l >= 0; l--) {
        const TailLevel& L = a.L[l]; const PlGeom& g = L.op.g; const TailLvl D = lv[l];
        const PlGeom& gc = a.L[l + 1].op.g; const TailLvl Dc = lv[l + 1];
        int cur = rot[l] & 3, prv = (rot[l] >> 2) & 3, nxt = (rot[l] >> 4) & 3;
        const int curc = rot[l + 1] & 3;
        const double* const ez = TL_V(Dc, curc, 0); const double* const ex = TL_V(Dc, curc, 1);
        const double* const cz = TL_V(D, cur, 0); const double* const cx = TL_V(D, cur, 1);
        double* const wz_ = TL_V(D, nxt, 0); double* const wx_ = TL_V(D, nxt, 1);
        TL_FOR_NODES(g) {          // prolongation + correction (prolong_node) into the free buffer
            const int o = (i + 1) * D.CR + (j + 1);
            int d; double s, oz = 0.0, ox = 0.0;
            if (mgt_cls_z(L.op, i, j, d, s) != VV_ZERO) {
                const int jm = j + d, I0 = i >> 1, I1 = (i + 1) >> 1;
                int Jn = jm >> 1, Jo = (jm & 1) ? Jn + 1 : Jn - 1;
                Jn = min(max(Jn, 0), gc.nx - 2); Jo = min(max(Jo, 0), gc.nx - 2);
                const double pa = 0.5 * (ez[(I0 + 1) * Dc.CR + Jn + 1] + ez[(I1 + 1) * Dc.CR + Jn + 1]);
                const double pb = 0.5 * (ez[(I0 + 1) * Dc.CR + Jo + 1] + ez[(I1 + 1) * Dc.CR + Jo + 1]);
                oz = s * (cz[o + d] + (0.75 * pa + 0.25 * pb));
            }
            if (mgt_cls_x(L.op, i, j, d, s) != VV_ZERO) {
                const int im = i + d, J0 = j >> 1, J1 = (j + 1) >> 1;
                int In = im >> 1, Io = (im & 1) ? In + 1 : In - 1;
                In = min(max(In, 0), gc.nz - 2); Io = min(max(Io, 0), gc.nz - 2);
                const double pa = 0.5 * (ex[(In + 1) * Dc.CR + J0 + 1] + ex[(In + 1) * Dc.CR + J1 + 1]);
                const double pb = 0.5 * (ex[(Io + 1) * Dc.CR + J0 + 1] + ex[(Io + 1) * Dc.CR + J1 + 1]);
                ox = s * (cx[o + d * D.CR] + (0.75 * pa + 0.25 * pb));
            }
            wz_[o] = oz; wx_[o] = ox;
        }
        __syncthreads();
        { const int o = cur; cur = nxt; nxt = o; }
        mgt_tail_smooth(pool, L, D, cur, prv, nxt, a.nu_post, a.ratio, false);
        if (tid == 0) rot[l] = cur | (prv << 2) | (nxt << 4);
        __syncthreads();
    }
    // ---- the first tail level's iterate goes back to memory
    {
        const TailLevel& L = a.L[0]; const PlGeom& g = L.op.g; const TailLvl D = lv[0];
        const int cur = rot[0] & 3;
        const double* const cz = TL_V(D, cur, 0); const double* const cx = TL_V(D, cur, 1);
        for (int idx = tid; idx < g.nz * g.nx; idx += MGT_NT) {
            const int i = idx / g.nx, j = idx - i * g.nx, o = (i + 1) * D.CR + (j + 1);
            const long long c = pl_idx(g, i - g.gi0, j - g.gj0);
            L.v[0][c] = cz[o]; L.v[0][c + g.plane] = cx[o];
        }
    }
}
// does the tail starting at level l fit into k_mg_tail_lds' pool?
static bool mg_tail_lds_fits(const PlSolver* S, size_t l) {
    long long need = 0;
    for (size_t q = l; q < S->levels.size(); q++) {
        const PlGeom& g = S->levels[q]->gh.d;
        need += 10LL * (g.nz + 2) * (g.nx + 2) + 2LL * (g.nz + 5) + 2LL * (g.nx + 5);
    }
    return need <= MGT_TAIL_POOL;
}

// Chebyshev coefficients of a sequence of n sweeps on [lmax / ratio, lmax], as smooth() computes them
static void cheb_coeffs(double lmax, double ratio, int n, double* c1, double* c2) {
    const double lmin = lmax / ratio, theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho_old = 1.0 / sigma;
    for (int k = 0; k < n; k++) {
        if (k == 0) { c1[k] = 0.0; c2[k] = 1.0 / theta; }
        else { const double rho = 1.0 / (2.0 * sigma - rho_old); c1[k] = rho * rho_old; c2[k] = 2.0 * rho / delta; rho_old = rho; }
    }
}
static bool mg_fused_level_ok(pl_ctx* ctx, const PlSolver* S, size_t l) {
    if (!S->fused || l + 1 >= S->levels.size() || S->early_K > 0 || S->any_line) return false;
    const MgLevel* L = S->levels[l];
    // several ranks: the REPLICATED levels (every rank holds and computes the whole level) run the tile kernels like one rank does --
    // they are the latency-bound ones; a level whose own or whose coarse grid is distributed keeps the staged path with its exchanges
    if (L->f32 || L->op.szz) return false;
    const PlGeom& g = L->gh.d;
    int npre_, npost_;
    level_nu(S, l, npre_, npost_);
    if (L->dist) {
        // a DISTRIBUTED level (round 4): tiles over the rank's block, right-hand side exchanged npre + 2 deep, the iterate written into
        // the ring by the rim tiles (k_mg_pre<.., EXT = 1>), the coarse correction exchanged 4 deep unless its level is replicated.
        // Level 0 (stage 1 of the block preconditioner inside the tile kernel) stays with the staged path on several ranks.
        const bool dist_fused = !(getenv("PYLAMP_MG_FUSED_DIST") && atoi(getenv("PYLAMP_MG_FUSED_DIST")) == 0);      // (read per call: tests switch it)
        const MgLevel* C = S->levels[l + 1];
        if (!dist_fused || l == 0 || npre_ + 2 > std::min(g.lnz, g.lnx) || npre_ + 3 > PL_RING) return false;
        if (C->dist && std::min(C->gh.d.lnz, C->gh.d.lnx) < 4) return false;
        if ((g.gi0 & 1) || (g.gj0 & 1)) return false;
    } else if (S->levels[l + 1]->dist) return false;
    if ((long long)g.lnz * g.lnx > S->fused_max_nodes) return false;
    if (S->use_tail && l > 0 && (long long)g.nz * g.nx <= S->tail_max_nodes && S->levels.size() - l <= PL_TAIL_MAX_LEVELS) return false;   // the tail kernel's levels
    int npre, npost;
    level_nu(S, l, npre, npost);
    return npre >= 1 && npre <= 3 && npost >= 1 && npost <= 3;
}
// One visit of level l by the tile kernels: pre (-> iterate in L->v[0], coarse rhs in C->f), the coarse levels, post
// (-> final_out times final_scale if given, else L->v[2]; *out tells where).  rs != NULL (level 0 of the Stokes
// preconditioner): the right-hand side is computed from the scaled residual and kept in L->f, z_p goes to z.
static void vcycle_fused_level(pl_ctx* ctx, PlSolver* S, size_t l, const double* f, double** out, double* final_out, double final_scale,
                               const PlStokesOp* sop, const double* rs, double* z) {
    MgLevel* L = S->levels[l];
    MgLevel* C = S->levels[l + 1];
    const PlGeom& g = L->gh.d;
    int npre, npost;
    level_nu(S, l, npre, npost);
    MgTileArgs a{};
    a.op = L->op; a.opc = C->op;
    a.f = rs ? (const double*)L->f : f; a.fout = L->f; a.v = L->v[0]; a.fc = C->f;
    // tile edge: 32 on the large levels (halo recomputation 1.7x instead of 2.6x at three sweeps; one workgroup per CU), 16 below
    const int TS = (long long)g.lnz * g.lnx >= S->tile32_min_nodes ? 32 : MGT_TS;
    // (the grid's last row / column belong to the tiles in front of them: mgt_owns)
    const int own_z = g.lnz - (g.gi0 + g.lnz == g.nz ? 1 : 0), own_x = g.lnx - (g.gj0 + g.lnx == g.nx ? 1 : 0);
    a.tiles_x = (own_x + TS - 1) / TS;
    const dim3 grid((unsigned)(a.tiles_x * ((own_z + TS - 1) / TS)));
    cheb_coeffs(L->lmax, S->cheb_ratio, npre, a.c1, a.c2);
    if (rs) { a.sop = *sop; a.rs = rs; a.z = z; }
#define MGT_PRE(NS, L0) do { if (TS == 32) hipLaunchKernelGGL((k_mg_pre<NS, L0, 32>), grid, dim3(mgt_nt(mgt_pre_nn(NS, 32))), 0, ctx->stream, a); \
                             else hipLaunchKernelGGL((k_mg_pre<NS, L0, MGT_TS>), grid, dim3(mgt_nt(mgt_pre_nn(NS, MGT_TS))), 0, ctx->stream, a); } while (0)
#define MGT_PRE_X(NS) do { if (TS == 32) hipLaunchKernelGGL((k_mg_pre<NS, false, 32, 1>), grid, dim3(mgt_nt(mgt_pre_nn(NS + 1, 32))), 0, ctx->stream, a); \
                           else hipLaunchKernelGGL((k_mg_pre<NS, false, MGT_TS, 1>), grid, dim3(mgt_nt(mgt_pre_nn(NS + 1, MGT_TS))), 0, ctx->stream, a); } while (0)
    if (L->dist) {
        (void)pl_halo(ctx, g, (double*)a.f, 2, g.plane, npre + 2);          // ONE exchange on the way down: the right-hand side
        if (npre == 1) MGT_PRE_X(1); else if (npre == 2) MGT_PRE_X(2); else MGT_PRE_X(3);
        if (!C->dist) (void)pl_gather_blocks(ctx, C->gh.d, C->f, 2, C->gh.d.plane);      // replicated coarse level: everybody's restricted residual
    }
    else if (rs) { if (npre == 1) MGT_PRE(1, true); else if (npre == 2) MGT_PRE(2, true); else MGT_PRE(3, true); }
    else { if (npre == 1) MGT_PRE(1, false); else if (npre == 2) MGT_PRE(2, false); else MGT_PRE(3, false); }
#undef MGT_PRE
#undef MGT_PRE_X
    double* ec = nullptr;
    bool wf = false;
    vcycle<double>(ctx, S, l + 1, C->f, &ec, &wf, nullptr, 1.0, 0, nullptr);      // (takes the tile kernels itself where it can)
    if (L->dist && C->dist) (void)pl_halo(ctx, C->gh.d, ec, 2, C->gh.d.plane, 4);  // ONE exchange on the way up: the coarse correction
    a.ec = ec; a.out = final_out ? final_out : L->v[2]; a.oscale = final_out ? final_scale : 1.0; a.out_plane = g.plane;
    cheb_coeffs(L->lmax, S->cheb_ratio, npost, a.c1, a.c2);
#define MGT_POST(NS) do { if (TS == 32) hipLaunchKernelGGL((k_mg_post<NS, 32>), grid, dim3(mgt_nt(mgt_post_nn(NS, 32))), 0, ctx->stream, a); \
                          else hipLaunchKernelGGL((k_mg_post<NS, MGT_TS>), grid, dim3(mgt_nt(mgt_post_nn(NS, MGT_TS))), 0, ctx->stream, a); } while (0)
    if (npost == 1) MGT_POST(1); else if (npost == 2) MGT_POST(2); else MGT_POST(3);
#undef MGT_POST
    *out = final_out ? nullptr : L->v[2];
}

// copy 2 velocity planes into the FP64 Krylov vector (times oscale)
template <typename T>
__global__ __launch_bounds__(256) void k_copy_vel(PlGeom g, const T* __restrict__ e, double* __restrict__ z, double oscale) {
    PL_NODE_PROLOGUE(g)
    (void)i; (void)j;
    z[c] = (double)e[c] * oscale; z[c + g.plane] = (double)e[c + g.plane] * oscale;
}

// Constraint rows of a preconditioned direction, closed in FP64: walls 0, slaves s * master.  Stage 1 does not lift the
// residual of these rows because every direction satisfies them exactly (see stage1_node) -- true to 1e-16 when the
// last sweep runs in FP64, but a slave and its master evaluated in FP32 by two different code paths differ by 1e-7,
// which would leave a residual on those rows that nothing ever removes (BiCGStab then stalls near 1e-6).
// One thread per node of the frame: rows 0, nz-2, nz-1 and columns 0, nx-2, nx-1.
__global__ __launch_bounds__(256) void k_vv_close_frame(PlVvOp op, double* __restrict__ z) {
    const PlGeom& g = op.g;
    const int t = blockIdx.x * 256 + threadIdx.x;
    int i, j;
    if (t < 3 * g.lnx) { const int q = t / g.lnx; i = q == 0 ? 0 : g.nz - 3 + q; j = g.gj0 + t % g.lnx; }
    else if (t < 3 * g.lnx + 3 * g.lnz) { const int u = t - 3 * g.lnx, q = u / g.lnz; j = q == 0 ? 0 : g.nx - 3 + q; i = g.gi0 + u % g.lnz; }
    else return;
    const int li = i - g.gi0, lj = j - g.gj0;
    if (li < 0 || li >= g.lnz || lj < 0 || lj >= g.lnx) return;
    const long long c = pl_idx(g, li, lj), P = g.plane;
    int moff; double s;
    int cls = vv_cls_z(op, i, j, moff, s);
    if (cls == VV_ZERO) z[c] = 0.0; else if (cls == VV_SLAVE) z[c] = s * z[c + moff];
    cls = vv_cls_x(op, i, j, moff, s);
    if (cls == VV_ZERO) z[c + P] = 0.0; else if (cls == VV_SLAVE) z[c + P] = s * z[c + moff + P];
}

// z = M^-1 rs   (rs: scaled residual, 3 planes)
//
// FP32 level 0 (S->levels[0]->f32): the velocity block is solved in single precision as  A~ v~ = f~  with
//     A~ = A_vv / sigma  (viscosity planes stored times 1/sigma, sigma ~ eta_min / h^2: coefficients O(1) and above),
//     f~ = f kappa / sigma,   v~ = kappa v,   kappa = sqrt(n) / ||r||  (the current residual norm: v~ = O(1)),
// so that nothing leaves the FP32 range whatever the units of the model and however far the residual has dropped.
// stage 1 writes f~ (and the first sweep v~1), the last post-smoothing sweep writes v = v~ / kappa into z.
template <typename T>
static int stokes_precond_t(pl_ctx* ctx, PlSolver* S, const double* rs, double* z) {
    PlStokesOp op = ctx->sop;
    MgLevel* L0 = S->levels[0];
    MgLevel* L = L0;                                  // (g_vv_vec looks at L)
    const PlGeom& g = ctx->sop.g;
    const bool f32 = std::is_same<T, float>::value;
    const double kappa = f32 ? S->kappa : 1.0, fscale = f32 ? S->kappa / S->sigma : 1.0, oscale = f32 ? 1.0 / S->kappa : 1.0;
    int e_last = 0, f_depth = 0;
    const bool deep = level_deep_plan(S, L0, 0, e_last, f_depth);
    ExtViewT<T> V = ext_view<T>(ctx, L0, deep ? f_depth : 0);
    if (L0->dist) {
        // stage 1 reads the pressure residual one node up / left; in deep mode it is evaluated f_depth nodes into the halo,
        // which makes the velocity right-hand side of level 0 valid there without an exchange of its own
        if (deep) PL_TRY(pl_halo(ctx, g, (double*)rs, 3, g.plane, f_depth + 1));
        else PL_TRY(pl_halo(ctx, g, (double*)rs + 2 * g.plane, 1, g.plane));
    }
    op.g = V.op.g; op.etas -= V.sh; op.etan -= V.sh; if (op.rho) op.rho -= V.sh;
    op.iKc /= S->schur_scale;                     // only the S^-1 r_p evaluations of stage 1 use it
    if constexpr (std::is_same<T, double>::value) {
        if (mg_fused_level_ok(ctx, S, 0)) {           // stage 1, the pre-smoothing, residual and restriction in one launch; post likewise
            double* e2 = nullptr;
            vcycle_fused_level(ctx, S, 0, nullptr, &e2, z, 1.0, &op, rs, z);
            PL_HIP(ctx, hipGetLastError());
            S->nprec++;
            return 0;
        }
    }
    // the first pre-smoothing sweep of level 0 (from the zero guess: v1 = -c2 f / diag) is written by stage 1 itself
    int npre0, npost0;
    level_nu(S, 0, npre0, npost0);
    if constexpr (std::is_same<T, double>::value) {
        // Staged level 0 (the two bandwidth-bound levels do not take the tile kernels) with V(1,1): the right-hand side f, the first
        // iterate v1 = -c2 f / diag and the residual are STORED in FP32, everything is computed in FP64, and the iterate that carries
        // the coarse correction (v = v1 + P e) as well as the result stay FP64.  Rounding f, v1 and r to 24 bits perturbs z by 1e-7 of
        // |v1| << |z| -- unlike an FP32 iterate v, whose rough rounding error is amplified by (L/h)^2 in A v (why the all-FP32 level 0
        // of round 2 cost iterations).  64 of the level's 280 B/node less.  Values outside a safe FP32 range (exotic units): all FP64.
        // Measured (tools/l0_mixed_probe.py, 1025^2, cold start): 31 -> 33 iterations to 1e-7, 35 -> 40 to 1e-10; in the time loop at
        // 2049^2 (warm starts, 1e-7): 10.0 -> 10.05 iterations, Stokes 20.4 -> 19.8 ms.  The perturbation of the recurrences adds up with
        // the iteration count: hence only for solves to rtol >= 1e-8 that start from the caller's guess (the solves of a time loop).
        const double hmean = 0.5 * ((ctx->geom.zc.back() - ctx->geom.zc.front()) / (ctx->nz - 1) + (ctx->geom.xc.back() - ctx->geom.xc.front()) / (ctx->nx - 1));
        const double sig = ctx->sop.Kc / hmean, ik = 1.0 / S->kappa;            // ~ eta_min / h^2 and the typical entry of the scaled residual
        const bool range_ok = std::isfinite(sig) && sig > 0.0 && ik * std::min(1.0, sig) > 1e-30 && ik * std::max(1.0, sig) < 1e22;
        if (S->l0_mixed && S->l0_mixed_now && !S->any_line && range_ok && !L0->dist && !L0->op.szz && g_vv_vec && (g.plane % 2) == 0 && S->levels.size() > 1 && npre0 == 1 && npost0 == 1 &&
            S->fuse_first && S->early_K == 0 && ctx->nranks == 1 && !S->levels[1]->dist && !S->levels[1]->f32) {
            MgLevel* C = S->levels[1];
            // (planes of their own, zero-initialised: the ring around the block must read as zero, which a view of an FP64 buffer
            //  that an all-FP64 solve has used before does not)
            for (float** q : {&L0->ff, &L0->rf, &L0->vf[0]}) if (!*q) PL_TRY(fmalloc0(ctx, q, (size_t)2 * g.plane * sizeof(float)));
            float* ff = L0->ff; float* rf = L0->rf; float* v1f = L0->vf[0];
            double* vd = L0->v[0];
            PlVvOpT<float> cls{};                                 // classification only (stage 1 reads the FP64 viscosities of `op`)
            cls.g = L0->op.g; cls.slave_x = L0->op.slave_x; cls.slave_z0 = L0->op.slave_z0; cls.slave_zL = L0->op.slave_zL;
            cls.s0 = (float)L0->op.s0; cls.sL = (float)L0->op.sL;
            const double lmax = L0->lmax, lmin = lmax / S->cheb_ratio, c2 = 1.0 / (0.5 * (lmax + lmin));
            const dim3 rows2 = pl_grid_rows2(g), bl(64, 4);
            hipLaunchKernelGGL(k_prec_stage1_v2<float>, rows2, bl, 0, ctx->stream, op, cls, rs, z, ff, v1f, c2, 1.0, 1.0);
            hipLaunchKernelGGL((k_vv_first2<double, float, float>), dim3(3 * rows2.x + 3 * rows2.y + 2), bl, 0, ctx->stream, L0->op, (const float*)ff, v1f, c2, 2,
                               ctx->sop.anchor_i, ctx->sop.anchor_j, ctx->sop.wall_ps);
            static const bool r_double = getenv("PYLAMP_L0_MIXED") && atoi(getenv("PYLAMP_L0_MIXED")) == 2;      // experiment: residual kept in FP64
            if (r_double) {
                hipLaunchKernelGGL((k_vv_sweep2<1, double, double, float, float>), rows2, bl, 0, ctx->stream, L0->op, (const float*)v1f, (const float*)nullptr,
                                   (const float*)ff, L0->r, 0.0, 0.0, 1.0);
                hipLaunchKernelGGL((k_vv_restrict<double, double>), grid2d(C->gh.d), bl, 0, ctx->stream, g, C->op, (const double*)L0->r, C->f, 1.0);
            } else {
            hipLaunchKernelGGL((k_vv_sweep2<1, double, float, float, float>), rows2, bl, 0, ctx->stream, L0->op, (const float*)v1f, (const float*)nullptr,
                               (const float*)ff, rf, 0.0, 0.0, 1.0f);
            hipLaunchKernelGGL((k_vv_restrict<float, double>), grid2d(C->gh.d), bl, 0, ctx->stream, g, C->op, (const float*)rf, C->f, 1.0);
            }
            double* ec = nullptr; bool wf = false;
            vcycle<double>(ctx, S, 1, C->f, &ec, &wf, nullptr, 1.0, 0, nullptr);
            // (Round 3: prolongation + correction + post-sweep in ONE kernel -- v never written, P e evaluated in registers from the coarse
            //  rows m-1, m, m+1 with lane shuffles, 68 instead of 100 B/node; parity with this pair to 1e-11 -- took 115 us against
            //  33 + 70: the sweep is co-limited by its FP64 instruction issue, and the interpolation arithmetic moves into it.  Removed.)
            hipLaunchKernelGGL((k_vv_prolong_add<double, double, float>), grid2d(g), bl, 0, ctx->stream, L0->op, C->gh.d, (const double*)ec, (const float*)v1f, vd);
            hipLaunchKernelGGL((k_vv_sweep2<0, double, double, double, float>), rows2, bl, 0, ctx->stream, L0->op, (const double*)vd, (const double*)nullptr,
                               (const float*)ff, z, 0.0, c2, 1.0);
            PL_HIP(ctx, hipGetLastError());
            S->nprec++;
            return 0;
        }
    }
    const bool fuse_first = S->fuse_first && !S->any_line && g_vv_vec && (g.plane % 2) == 0 && S->levels.size() > 1 && npre0 >= 1 && !L0->op.szz;
    const int anchor[2] = {ctx->sop.anchor_i, ctx->sop.anchor_j};
    T* f0 = LevelT<T>::f(L0);
    if (g_vv_vec && (g.plane % 2) == 0) {
        const double lmax = L0->lmax, lmin = lmax / S->cheb_ratio, c2 = 1.0 / (0.5 * (lmax + lmin));     // as smooth() computes it
        hipLaunchKernelGGL(k_prec_stage1_v2<T>, pl_grid_rows2(op.g), dim3(64, 4), 0, ctx->stream, op, V.op, rs - V.sh, z - V.sh, f0 - V.sh,
                           fuse_first ? LevelT<T>::v(L0)[2] - V.sh : (T*)nullptr, c2, fscale, kappa);
    } else
        hipLaunchKernelGGL(k_prec_stage1<T>, pl_grid_rows(op.g), dim3(64, 4), 0, ctx->stream, op, V.op, rs - V.sh, z - V.sh, f0 - V.sh,
                           pl_row_iters(op.g), fscale);
    if (S->early_K > 0 && !f32) (void)hipEventRecord(S->ev_f, ctx->stream);        // level 0's right-hand side is complete
    T* e = nullptr;
    bool wrote = false;
    const bool direct = S->levels.size() > 1 && S->nu_post > 0;     // last sweep writes into z
    vcycle<T>(ctx, S, 0, f0, &e, &wrote, direct ? z : nullptr, oscale, deep ? f_depth : 0, fuse_first ? anchor : nullptr);
    if (!wrote) hipLaunchKernelGGL(k_copy_vel<T>, grid2d(g), dim3(64, 4), 0, ctx->stream, g, (const T*)e, z, oscale);
    if (f32) hipLaunchKernelGGL(k_vv_close_frame, dim3((3 * g.lnx + 3 * g.lnz + 255) / 256), dim3(256), 0, ctx->stream, L0->op, z);
    PL_HIP(ctx, hipGetLastError());
    S->nprec++;
    return 0;
}
static int stokes_precond(pl_ctx* ctx, PlSolver* S, const double* rs, double* z) {
    return S->levels[0]->f32 ? stokes_precond_t<float>(ctx, S, rs, z) : stokes_precond_t<double>(ctx, S, rs, z);
}

__global__ void k_defl_ysum(PlStokesOp op, const double* __restrict__ rp, double* __restrict__ part);      // defined with the deflation below
__global__ void k_defl_init(double* __restrict__ sc);

// =========================================================================================
// Generic right-preconditioned BiCGStab (host-driven scalars)
// =========================================================================================
typedef std::function<int(const double*, double*)> VecOp;

struct BicgVecs { double *r, *rt, *p, *v, *s, *t, *y, *z; double* xbest; double *dx, *r0; };

// Solves for the CORRECTION of the initial guess:  A dx = r0 := b - A x0,  dx from 0,  x = x0 + dx at the end.
// The Stokes solve starts from the hydrostatic state (or the previous step's solution), whose pressure is ~1e5 times
// the dynamic part: accumulating the iterate in x itself loses those digits in every update and puts the floor of the
// recomputed residual b - A x at ~1e-10 of the dynamic load -- the very tolerance asked for.  In correction form the
// floor is eps ||A|| ||dx||.  r0 is evaluated once in FP64; the residual that is reported and tested is
// r0 - A dx, recomputed with the operator (not the recurrence).
// The recurrence residual drifts away from the true one; whenever the recurrence meets the tolerance the true
// residual is recomputed, and if it does not meet it the iteration restarts from it (residual replacement, at most
// PL_MAX_RESTARTS times, and only while a restart still gains a factor 2).  converged = 1 only if the TRUE residual
// meets rtol.
// ref_norm > 0 replaces ||b|| as the reference of the stopping test and of rel_residual.
#define PL_MAX_RESTARTS 4
// etol > 0 (Stokes): the residual test alone does not bound the velocity error -- the ratio error / residual is ~10 on the
// 513^2 mantle problem and ~2e4 on the coarse 33 x 41 fixture of the reference trajectory.  What the norm misses is the
// CONTINUITY part of the residual: in the row-scaled system it is h div(v) / 2, a velocity, but it is compared with ||b||,
// which is dominated by the momentum rows of the low-viscosity regions and can be 1e4 x the velocity scale; a divergence
// error integrates over the domain, so the velocity error it causes is ~ (L/h) ||r_cont|| (measured on the fixture with
// the NumPy prototype: 1.8e-4 of the 1.8e-4 velocity error at rtol 1e-10 comes from the continuity rows, tools/ratio.py).
// The error of the first np_vel planes is therefore estimated as
//     ( n ||r_cont|| + ||(M^-1 r)_vel|| ) / ||x_vel||,   n = max(nz, nx)
// -- the second term is the velocity response to the momentum residual through one V-cycle -- (within a factor ~2 of the
// true error on every case checked against direct or much tighter solves: 33x41, 41^2, 129^2, 513^2, 1025^2, 2049^2 with the
// mantle and block models), and the solve ends when BOTH the residual meets rtol and the estimate
// meets etol.  Near the end the estimate is evaluated in every iteration from the recurrence (the fused reduction also
// returns the continuity block's share; one extra norm of the iterate); before the solve is declared converged it is
// evaluated from the true residual, and if that fails the same iteration continues -- no restart.
static int bicgstab(pl_ctx* ctx, PlSolver* S, const PlGeom& g, int np, const VecOp& A, const VecOp* M,
                    const double* b, double* x, bool use_x0, double rtol, int maxit, BicgVecs w,
                    pl_solve_stats* st, double ref_norm = 0.0, double etol = 0.0, int np_vel = 0) {
    const long long n = (long long)np * g.plane;
    const size_t bytes = (size_t)n * sizeof(double);
    double d2[2];
    PL_TRY(dots(ctx, S, g, np, b, b, nullptr, nullptr, d2));
    double bnorm = std::sqrt(d2[0]);
    if (ref_norm > 0.0 && bnorm > 0.0) bnorm = ref_norm;
    st->iterations = 0; st->converged = 0; st->rel_residual = 0.0;
    if (!(bnorm > 0.0)) {                       // b = 0 -> x = 0
        PL_HIP(ctx, hipMemsetAsync(x, 0, bytes, ctx->stream));
        st->converged = 1;
        return 0;
    }
    double* dx = x; const double* r0 = b;       // no initial guess: the correction IS the solution
    if (use_x0) {
        dx = w.dx;
        PL_TRY(A(x, w.v));
        hipLaunchKernelGGL(k_axpy_out, grid1d(n), dim3(256), 0, ctx->stream, n, w.r0, b, w.v, -1.0);
        r0 = w.r0;
    }
    PL_HIP(ctx, hipMemsetAsync(dx, 0, bytes, ctx->stream));
    PL_HIP(ctx, hipMemcpyAsync(w.r, r0, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    double xx_guess = 0.0;                        // ||x0_vel||^2 of a warm start (0: unknown / cold start from a state without velocities)
    if (use_x0 && etol > 0.0 && np_vel > 0 && np > np_vel) {
        double dxx[2];
        PL_TRY(dots(ctx, S, g, np_vel, x, x, nullptr, nullptr, dxx));
        if (std::isfinite(dxx[0]) && dxx[0] > 0.0) xx_guess = dxx[0];
    }
    // Shadow residual: the textbook choice r0 when a preconditioned solve starts from a guess (r0 = b - A x0 has components in all
    // rows), the seeded random vector otherwise -- with r0 = b the method breaks down (b lives on the vz rows only).  Against the
    // random vector in the time-step loop at 2049^2 the first iterations no longer stall on an arbitrary alpha (10.7 -> 9.7
    // iterations per solve, 8..11 instead of 8..14); the unpreconditioned heat solve gains nothing from it.  PYLAMP_SHADOW=0: random always.
    static const int shadow_mode = getenv("PYLAMP_SHADOW") ? atoi(getenv("PYLAMP_SHADOW")) : 1;
    if (shadow_mode == 1 && use_x0 && M) PL_HIP(ctx, hipMemcpyAsync(w.rt, r0, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    else {
        PL_HIP(ctx, hipMemsetAsync(w.rt, 0, bytes, ctx->stream));
        hipLaunchKernelGGL(k_random_interior, grid2d(g), dim3(64, 4), 0, ctx->stream, g, np, w.rt, 1234u);
    }
    const bool on_device = dots_on_device(ctx, g) && !getenv("PYLAMP_HOST_SCALARS");
    // lazy deflation on several ranks: its coefficients ride in the two reductions of an iteration (k_defl_alpha_ride ...)
    const bool ride = S->defl_lazy && S->defl_active && np == 3 && M && on_device && ctx->nranks > 1 && pl_geom_is_dist(g);
    if (ride) {                                          // r~.Aw, Aw.Aw and its continuity part: constant during this solve
        double c1[2], c2[2];
        PL_TRY(dots(ctx, S, g, 3, w.rt, S->Awdefl, S->Awdefl, S->Awdefl, c1));
        PL_TRY(dots(ctx, S, g, 1, S->Awdefl + 2 * g.plane, S->Awdefl + 2 * g.plane, nullptr, nullptr, c2));
        hipLaunchKernelGGL(k_set3, dim3(1), dim3(1), 0, ctx->stream, S->scal + 35, c1[0], S->scal + 37, c1[1], S->scal + 38, c2[0]);
    }
    static const bool trace = getenv("PYLAMP_SOLVER_TRACE") != nullptr;        // residual history on stderr
    int it = 0, restarts = 0;
    double true_norm = -1.0, last_true = -1.0;          // ||r0 - A dx|| of the current dx (< 0: not evaluated)
    // BiCGStab is not monotone and, past the attainable accuracy, drifts and can blow up: keep the best
    // iterate, stop after 60 iterations without a new best or when the residual explodes, return the best.
    double best = 0.0, best_kept = 0.0; int best_it = 0; bool have_best = false;      // best_kept: residual of the iterate in w.xbest
    double tol = rtol;                              // lowered when the velocity-error estimate asks for it
    int est_checks = 0;
    // velocity-error estimate (see above): from the recurrence residual in every iteration near the end, from the true
    // residual before the solve is declared converged
    const bool use_est = etol > 0.0 && np_vel > 0 && np > np_vel;
    // amplification of a divergence error: domain length over the SMALLEST cell along an axis (= max(nz, nx) - 1 on a uniform grid;
    // on a graded grid L / h_min can be far larger than the node count)
    double n_amp = (double)std::max(ctx->nz, ctx->nx);
    {
        const std::vector<double>& zc = ctx->geom.zc; const std::vector<double>& xc = ctx->geom.xc;
        double hzmin = zc.back() - zc.front(), hxmin = xc.back() - xc.front();
        for (size_t k = 0; k + 1 < zc.size(); k++) hzmin = std::min(hzmin, zc[k + 1] - zc[k]);
        for (size_t k = 0; k + 1 < xc.size(); k++) hxmin = std::min(hxmin, xc[k + 1] - xc[k]);
        if (hzmin > 0.0 && hxmin > 0.0) n_amp = std::max(n_amp, std::max((zc.back() - zc.front()) / hzmin, (xc.back() - xc.front()) / hxmin));
    }
    double est_rec = 0.0;                           // 0: unknown (far from convergence) -- only the residual test applies
    // ||A_vv^-1 r_mom|| / ||r_mom||: the amplification of the momentum residual.  Near the end it is measured in every iteration, for
    // free, on the pair (s, z = M^-1 s) the iteration computes anyway -- ||z_vel|| / ||s_mom|| -- and then serves the estimate of the
    // iterate's own residual r = s - omega t, which has the same character.  (Until round 3 it was measured by ONE MORE
    // preconditioner application on the true residual at every check: 0.8-1.6 ms of a 25 ms solve at 2049^2.  PYLAMP_EST_EXACT=1
    // brings that back.)
    double a_mom = 1.0; bool a_mom_measured = false; int amom_count = 0;
    const bool est_exact = getenv("PYLAMP_EST_EXACT") && atoi(getenv("PYLAMP_EST_EXACT")) != 0;      // (read per solve: tests compare both)
    // With the pressure-anchor deflation active the component of the residual along that mode needs its own term: its
    // amplification is ||w|| / ||u|| (1e4 and more), far beyond n -- r = gamma u + ..., gamma = y.r / y.u, and the error it stands
    // for is gamma w (A w = u).  Without the deflation BiCGStab has removed this component by the time it leaves its plateau;
    // with it nothing in the residual NORM shows it (found as a 1.9e-6 error behind an estimate of 2e-8 on the 33 x 41 fixture).
    const bool anchor_term = use_est && S->defl_active && S->defl_yAw != 0.0 && np == 3;
    auto estimate = [&](double rr_cont, double rr_total, double xx, double yr) {
        const double rc = rr_cont > 0.0 ? rr_cont : 0.0, rm = rr_total - rc > 0.0 ? rr_total - rc : 0.0;
        if (!(xx > 0.0)) return 0.0;
        double e = (n_amp * std::sqrt(rc) + a_mom * std::sqrt(rm)) / std::sqrt(xx);
        if (anchor_term) e += std::fabs(yr / S->defl_yAw) * std::sqrt(S->defl_wvel2 / xx);
        return e;
    };
    auto ysum_dev = [&](const double* rp, double* out_slot) {       // y . r over the continuity rows -> *out_slot (this rank's share)
        const int nb = g.lnz < DOT_BLOCKS ? g.lnz : DOT_BLOCKS;
        hipLaunchKernelGGL(k_defl_ysum, dim3(nb), dim3(256), 0, ctx->stream, ctx->sop, rp, S->scal + PL_SCAL_N);
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, ctx->stream, nb, S->scal + PL_SCAL_N, out_slot, 0, 0.0);
    };
    auto ysum_host = [&](const double* rp, double& yr) -> int {     // the same, reduced over the ranks, on the host
        ysum_dev(rp, S->scal + 28);
        if (ctx->nranks > 1 && pl_geom_is_dist(g)) PL_TRY(pl_comm_allreduce_dev(ctx, S->scal + 28, 1));
        PL_HIP(ctx, hipMemcpyAsync(S->hpart + PL_PART_N * DOT_BLOCKS, S->scal + 28, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        yr = S->hpart[PL_PART_N * DOT_BLOCKS];
        return 0;
    };
    bool resume = false;                            // continue the running iteration instead of restarting it
    double rho = 1.0, alpha = 1.0, omega = 1.0, rho_new = 0.0, rnorm = 0.0;
    bool p_fused = false, broke = false;            // p_fused: p of the coming iteration was already written by k_xrp_update_dev
    // scale of the FP32 velocity solve inside the preconditioner: v~ = kappa v = O(1) at the current residual level
    const double sqrt_n = std::sqrt((double)np * ctx->nz * ctx->nx);
    auto set_kappa = [&](double rn) { S->kappa = (std::isfinite(rn) && rn > 0.0) ? sqrt_n / rn : 1.0; };
    st->error_estimate = 0.0;
    for (;;) {
        if (!resume) {
            // ---- (re)start from the residual in w.r
            PL_HIP(ctx, hipMemsetAsync(w.p, 0, bytes, ctx->stream));
            PL_HIP(ctx, hipMemsetAsync(w.v, 0, bytes, ctx->stream));
            rho = alpha = omega = 1.0;
            p_fused = false;
            PL_TRY(dots(ctx, S, g, np, w.rt, w.r, w.r, w.r, d2));
            rho_new = d2[0]; rnorm = std::sqrt(d2[1]);
            if (restarts == 0) best = rnorm;
            set_kappa(rnorm);
            broke = false;
            if (S->defl_lazy && S->defl_active && np == 3 && M) {       // y.r of the (re)start residual; p = r
                ysum_dev(w.r + 2 * g.plane, S->scal + 32);
                if (ride) PL_TRY(pl_comm_allreduce_dev(ctx, S->scal + 32, 1));
                hipLaunchKernelGGL(k_defl_init, dim3(1), dim3(1), 0, ctx->stream, S->scal);
            }
        }
        resume = false;
        while (it < maxit && (rnorm > tol * bnorm || (use_est && est_rec > 0.7 * etol))) {
            it++;
            if (!(std::fabs(rho_new) > 0.0) || !std::isfinite(rho_new)) { broke = true; break; }
            const double beta = (rho_new / rho) * (alpha / omega);
            if (!p_fused) hipLaunchKernelGGL(k_p_update, grid1d(n), dim3(256), 0, ctx->stream, n, w.p, w.r, w.v, beta, omega);
            const double* yv = w.p;
            if (M) { PL_TRY((*M)(w.p, w.y)); yv = w.y; }
            PL_TRY(A(yv, w.v));
            const double* zv = w.s;
            if (on_device) {
                // alpha, omega stay on the device; a breakdown (rt.v = 0) shows up as a non-finite alpha / ||r|| below
                PL_TRY(dots_dev(ctx, S, g, np, w.rt, w.v, nullptr, nullptr, 1, rho_new, ride));
                if (ride) hipLaunchKernelGGL(k_s_update_defl, grid1d(n), dim3(256), 0, ctx->stream, n, w.s, (const double*)w.r, w.v, (const double*)S->Awdefl, (const double*)S->scal);
                else hipLaunchKernelGGL(k_s_update_dev, grid1d(n), dim3(256), 0, ctx->stream, n, w.s, w.r, w.v, S->scal);
                if (M) { PL_TRY((*M)(w.s, w.z)); zv = w.z; }
                PL_TRY(A(zv, w.t));
                // near the end: ||x_vel||^2 for the error estimate (-> scal[16]; of the iterate BEFORE this update, so that
                // it can ride along in the fused reduction -- the estimate needs it to ~10 %)
                const bool want_xx = use_est && rnorm <= 1e3 * tol * bnorm;
                // (warm start: ||x0_vel|| stands for ||x_vel|| in the per-iteration estimate -- they differ by the relative size of the
                //  correction, and the estimate needs the norm to ~10 %; the check on the true residual below uses the exact one)
                if (want_xx && !(xx_guess > 0.0)) PL_TRY(norm2_sum_dev(ctx, S, g, np_vel, dx, dx != x ? (const double*)x : (const double*)nullptr, false));
                // (the amplification moves slowly: measured in the first two late iterations, then in every third)
                const bool amom_now = want_xx && M && (amom_count < 2 || amom_count % 3 == 0);
                if (want_xx && M) amom_count++;
                if (amom_now) PL_TRY(norm2_sum_dev(ctx, S, g, np_vel, zv, nullptr, false, 20));      // ||z_vel||^2 of z = M^-1 s -> scal[20] (k_sum_partials also clears the slot behind its target: 19 belongs to y.t)
                const bool y_recur = S->defl_lazy && S->defl_active && np == 3 && M;       // y.r of the updated residual comes out of bicg_derive (sc[32])
                if (want_xx && anchor_term && !y_recur) {            // y.r of the updated residual = y.s - omega y.t: both ride in the reduction as well
                    ysum_dev(w.s + (long long)np_vel * g.plane, S->scal + 17);
                    ysum_dev(w.t + (long long)np_vel * g.plane, S->scal + 18);
                }
                PL_TRY(dots5_dev(ctx, S, g, np, use_est ? np_vel : np, w.t, w.s, w.rt, ride ? (const double*)S->Awdefl : (const double*)nullptr));       // omega, rho' and |r|^2 from ONE reduction
                // ... and the next direction in the same pass (the host's beta above is then only the breakdown test)
                if (yv != w.p) {
                    hipLaunchKernelGGL(k_xrp_update_dev, grid1d(n), dim3(256), 0, ctx->stream, n, dx, yv, zv, w.r, w.s, w.t, w.p, w.v, S->scal,
                                       (S->defl_lazy && S->defl_active && np == 3) ? (const double*)S->wdefl : (const double*)nullptr,
                                       ride ? (const double*)S->Awdefl : (const double*)nullptr);
                    p_fused = true;
                } else            // no preconditioner: y IS p
                    hipLaunchKernelGGL(k_xr_update_dev, grid1d(n), dim3(256), 0, ctx->stream, n, dx, yv, zv, w.r, w.s, w.t, S->scal);
                double* hs = S->hpart + PL_PART_N * DOT_BLOCKS;
                PL_HIP(ctx, hipMemcpyAsync(hs, S->scal, PL_SCAL_N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
                rho = rho_new;
                alpha = hs[2]; omega = hs[3];
                rho_new = hs[5]; rnorm = std::sqrt(hs[6]);
                if (!std::isfinite(alpha)) { broke = true; break; }
                if (amom_now) {
                    const double ss_mom = hs[12] - hs[15];
                    if (ss_mom > 0.0 && hs[20] > 0.0) { a_mom = std::min(std::max(std::sqrt(hs[20] / ss_mom), 1.0), n_amp * n_amp); a_mom_measured = true; }
                }
                if (want_xx) est_rec = estimate(hs[15] - 2.0 * omega * hs[13] + omega * omega * hs[14], hs[6], xx_guess > 0.0 ? xx_guess : hs[16],
                                                y_recur ? hs[32] : hs[17] - omega * hs[18]);
                else est_rec = 0.0;
            } else {
                PL_TRY(dots(ctx, S, g, np, w.rt, w.v, nullptr, nullptr, d2));
                if (!(std::fabs(d2[0]) > 0.0) || !std::isfinite(d2[0])) { broke = true; break; }
                alpha = rho_new / d2[0];
                hipLaunchKernelGGL(k_axpy_out, grid1d(n), dim3(256), 0, ctx->stream, n, w.s, w.r, w.v, -alpha);
                if (M) { PL_TRY((*M)(w.s, w.z)); zv = w.z; }
                PL_TRY(A(zv, w.t));
                PL_TRY(dots(ctx, S, g, np, w.t, w.s, w.t, w.t, d2));
                if (!(d2[1] > 0.0) || !std::isfinite(d2[1])) {        // s is already (numerically) zero
                    hipLaunchKernelGGL(k_xr_update, grid1d(n), dim3(256), 0, ctx->stream, n, dx, yv, zv, w.r, w.s, w.t, alpha, 0.0);
                    rnorm = 0.0;
                    break;
                }
                omega = d2[0] / d2[1];
                hipLaunchKernelGGL(k_xr_update, grid1d(n), dim3(256), 0, ctx->stream, n, dx, yv, zv, w.r, w.s, w.t, alpha, omega);
                rho = rho_new;
                PL_TRY(dots(ctx, S, g, np, w.rt, w.r, w.r, w.r, d2));
                rho_new = d2[0]; rnorm = std::sqrt(d2[1]);
                if (use_est && rnorm <= 1e3 * tol * bnorm) {        // the same estimate as the device-scalar path, with host round trips
                    double dc[2];
                    PL_TRY(dots(ctx, S, g, np - np_vel, w.r + (long long)np_vel * g.plane, w.r + (long long)np_vel * g.plane, nullptr, nullptr, dc));
                    PL_TRY(norm2_sum_dev(ctx, S, g, np_vel, dx, dx != x ? (const double*)x : (const double*)nullptr, true));
                    PL_HIP(ctx, hipMemcpyAsync(S->hpart + PL_PART_N * DOT_BLOCKS, S->scal + 16, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    const double xxh = S->hpart[PL_PART_N * DOT_BLOCKS];
                    double yr = 0.0;
                    if (anchor_term) PL_TRY(ysum_host(w.r + (long long)np_vel * g.plane, yr));
                    est_rec = estimate(dc[0], d2[1], xxh, yr);
                } else est_rec = 0.0;
            }
            if (!std::isfinite(rnorm) || !(std::fabs(omega) > 0.0)) { broke = true; break; }
            set_kappa(rnorm);
            if (trace) fprintf(stderr, "[pylamp bicgstab] it %3d  |r|/|b| %.3e  alpha %.3e omega %.3e\n", it, rnorm / bnorm, alpha, omega);
            // the best iterate is kept as a copy (what a stagnating or diverging iteration falls back to).  While the residual still
            // drops by a decade per iteration that copy -- 206 MB of traffic, 40 us at 2049^2, ten times per solve -- buys nothing: up to
            // iteration 12 only every fourth improvement is copied (best / best_it follow every one: the stagnation test is unchanged)
            if (rnorm < 0.9 * best && w.xbest) {
                best = rnorm; best_it = it;
                if (it > 12 || (it & 3) == 0) {
                    have_best = true; best_kept = rnorm;
                    PL_HIP(ctx, hipMemcpyAsync(w.xbest, dx, bytes, hipMemcpyDeviceToDevice, ctx->stream));
                }
            }
            if (w.xbest && (it - best_it > 60 || rnorm > 1e6 * best)) { broke = true; break; }       // stagnation / divergence
        }
        bool restored = false;                      // dx was replaced by the best iterate: the Krylov recurrence no longer belongs to it
        if (have_best && w.xbest && !(rnorm <= 1.5 * best_kept)) {
            PL_HIP(ctx, hipMemcpyAsync(dx, w.xbest, bytes, hipMemcpyDeviceToDevice, ctx->stream));
            restored = true;
        }
        // ---- true residual of the current dx
        PL_TRY(A(dx, w.t));
        hipLaunchKernelGGL(k_axpy_out, grid1d(n), dim3(256), 0, ctx->stream, n, w.s, r0, w.t, -1.0);
        PL_TRY(dots(ctx, S, g, np, w.s, w.s, nullptr, nullptr, d2));
        last_true = true_norm; true_norm = std::sqrt(d2[0]);
        if (trace) fprintf(stderr, "[pylamp bicgstab] it %3d  TRUE |r|/|b| %.3e (recurrence %.3e) restarts %d\n", it, true_norm / bnorm, rnorm / bnorm, restarts);
        if (true_norm <= tol * bnorm && !broke && it < maxit && use_est && M && est_checks < 6) {
            // the estimate of this iterate from the TRUE residual (which is in w.s): continuity part n ||r_cont||, momentum part
            // ||(M^-1 r)_vel|| -- one V-cycle is A_vv^-1 to ~10 %, so this is the velocity response to the momentum residual
            // (the block preconditioner's answer to the continuity residual is part of it too, but far too small: its Schur
            // complement is only an approximation; measured 37 x below the true error on the 33 x 41 fixture)
            double dc[2], dz[2];
            PL_TRY(dots(ctx, S, g, np - np_vel, w.s + (long long)np_vel * g.plane, w.s + (long long)np_vel * g.plane, nullptr, nullptr, dc));
            const bool exact = est_exact || !a_mom_measured;
            if (exact) {
                PL_TRY((*M)(w.s, w.z));
                PL_TRY(dots(ctx, S, g, np_vel, w.z, w.z, nullptr, nullptr, dz));
            }
            PL_TRY(norm2_sum_dev(ctx, S, g, np_vel, dx, dx != x ? (const double*)x : (const double*)nullptr, true));
            PL_HIP(ctx, hipMemcpyAsync(S->hpart + PL_PART_N * DOT_BLOCKS, S->scal + 16, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            est_checks++;
            const double xx = S->hpart[PL_PART_N * DOT_BLOCKS];
            if (xx_guess > 0.0 && xx > 0.0) xx_guess = xx;               // (the exact norm of this iterate from here on)
            const double rc = dc[0] > 0.0 ? dc[0] : 0.0, rm = true_norm * true_norm - rc;
            if (!exact) dz[0] = a_mom * a_mom * (rm > 0.0 ? rm : 0.0);            // the response the last iteration measured, scaled to this residual
            double est = xx > 0.0 ? (n_amp * std::sqrt(rc) + std::sqrt(dz[0] > 0.0 ? dz[0] : 0.0)) / std::sqrt(xx) : 0.0;
            double anchor_part = 0.0;
            if (anchor_term && xx > 0.0) {
                double yr = 0.0;
                PL_TRY(ysum_host(w.s + (long long)np_vel * g.plane, yr));
                anchor_part = std::fabs(yr / S->defl_yAw) * std::sqrt(S->defl_wvel2 / xx);
                est += anchor_part;
            }
            if (exact && rm > 0.0 && dz[0] > 0.0) { a_mom = std::min(std::max(std::sqrt(dz[0] / rm), 1.0), n_amp * n_amp); a_mom_measured = true; }
            st->error_estimate = est;
            if (trace) fprintf(stderr, "[pylamp bicgstab] it %3d  velocity-error estimate %.3e (etol %.1e): continuity %.3e momentum %.3e (amplification %.1f) anchor mode %.3e\n",
                               it, est, etol, xx > 0.0 ? n_amp * std::sqrt(rc / xx) : 0.0, xx > 0.0 ? std::sqrt(dz[0] / xx) : 0.0, a_mom, anchor_part);
            if (est > etol && tol > 1e-15) {
                tol = std::min(tol, true_norm / bnorm) * std::min(0.5, 0.7 * etol / est);
                est_rec = est;
                if (!restored) { resume = true; continue; }            // same Krylov iteration, smaller target
                // the iterate is the restored best one: w.r / w.p / rho belong to another dx -- restart from its true residual
                // (in w.s) instead of resuming
                restarts++;
                PL_HIP(ctx, hipMemcpyAsync(w.r, w.s, bytes, hipMemcpyDeviceToDevice, ctx->stream));
                best = true_norm; best_it = it; have_best = false;
                continue;
            }
        }
        if (true_norm <= tol * bnorm || broke || it >= maxit || restarts >= PL_MAX_RESTARTS) break;
        if (last_true >= 0.0 && !(true_norm < 0.5 * last_true)) break;      // a restart no longer pays: attainable accuracy
        // ---- residual replacement: continue from the TRUE residual
        restarts++;
        PL_HIP(ctx, hipMemcpyAsync(w.r, w.s, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        best = true_norm; best_it = it; have_best = false;
    }
    if (dx != x) hipLaunchKernelGGL(k_add_inplace, grid1d(n), dim3(256), 0, ctx->stream, n, x, dx);     // x = x0 + dx
    st->iterations = it;
    st->reserved_ = (it >= maxit && !broke) ? 1 : 0;            // ended by the caller's iteration limit (not stagnation / breakdown)
    st->rel_residual = true_norm / bnorm;
    st->converged = (st->rel_residual <= rtol && !(use_est && st->error_estimate > 1.5 * etol)) ? 1 : 0;
    return 0;
}

// =========================================================================================
// Stokes solve (device-resident core + C ABI)
// =========================================================================================
static int stokes_alloc(pl_ctx* ctx, PlSolver* S) {
    if (S->r) return 0;
    size_t vb = (size_t)3 * ctx->geom.d.plane * sizeof(double);
    for (double** q : {&S->r, &S->rt, &S->p, &S->v, &S->s, &S->t, &S->y, &S->z, &S->b, &S->x, &S->xb, &S->dx, &S->r0})
        PL_TRY(dmalloc0(ctx, q, vb));
    if (!S->scal) {
        PL_TRY(dmalloc0(ctx, &S->scal, (PL_SCAL_N + PL_PART_N * DOT_BLOCKS) * sizeof(double)));
        PL_HIP(ctx, hipHostMalloc((void**)&S->hpart, (PL_PART_N * DOT_BLOCKS + PL_SCAL_N) * sizeof(double)));
    }
    return 0;
}

// ---- deflation of the pressure-anchor mode ------------------------------------------------------------------------
// The reference pins the pressure by replacing the continuity row of ONE cell (the anchor, P[3,2] = 0, pylamp_stokes.py)
// -- so a net divergence defect of all other cells can only be absorbed by a sink flow towards that cell.  For the
// block-triangular preconditioner this is one eigenvalue of A M^-1 near 1/(number of cells) (7e-6 at 33 x 33 with a
// layered viscosity, everything else in [0.16, 1]; dense eigen-decomposition of the NumPy prototype, tools/spectrum.py):
// BiCGStab -- and GMRES alike -- sit on a plateau for ~15 iterations until the Krylov space has found it.
//   right eigenvector (residual space)  u: continuity residual 1 / (eta_n (rdz + rdx)), which M^-1 turns into a CONSTANT pressure
//   left eigenvector                    y: (hz + hx) on the continuity rows: y.(D_r A x) = sum of area x div(x) over these cells
//                                          = -(the same sum over the anchor and the four corner cells), by Gauss, for any x
//                                          whose wall-normal velocities vanish -- five cells, no operator application
// With w = A^-1 u (one extra solve to 1e-3; the vector is kept on the context, checked against the new operator before every
// solve -- ||u - A w|| / ||u||, one application -- and refreshed from itself when that exceeds 0.1) the
// preconditioner becomes  z = M^-1 r,  z += w y.(r - A z) / y.(A w):  the eigenvalue moves to 1, all others stay.
// NumPy prototype (tools/defl.py, mantle model 129^2, rtol 1e-10): 33 -> 21 iterations, no plateau.
__device__ inline bool defl_is_cont(const PlStokesOp& op, int i, int j) {
    const int nz = op.g.nz, nx = op.g.nx;
    if (i >= nz - 1 || j >= nx - 1 || (i == op.anchor_i && j == op.anchor_j)) return false;
    if ((i == 0 || i == nz - 2) && (j == 0 || j == nx - 2)) return false;
    return true;
}
__global__ __launch_bounds__(256) void k_defl_u(PlStokesOp op, double* __restrict__ u) {
    PL_NODE_PROLOGUE(op.g)
    const long long P = op.g.plane;
    u[c] = 0.0; u[c + P] = 0.0;
    u[c + 2 * P] = defl_is_cont(op, i, j) ? 1.0 / (op.etan[c] * (TB(op.g.rdz, i) + TB(op.g.rdx, j))) : 0.0;
}
// partial sums of y . r over the continuity rows (r: the pressure plane of a scaled residual)
__global__ __launch_bounds__(256) void k_defl_ysum(PlStokesOp op, const double* __restrict__ rp, double* __restrict__ part) {
    const PlGeom& g = op.g;
    double s0 = 0.0;
    for (int li = blockIdx.x; li < g.lnz; li += gridDim.x) {
        const int i = g.gi0 + li;
        const double hz = (i < g.nz - 1) ? 1.0 / TB(g.rdz, i) : 0.0;
        for (int lj = threadIdx.x; lj < g.lnx; lj += 256) {
            const int j = g.gj0 + lj;
            if (defl_is_cont(op, i, j)) s0 += (hz + 1.0 / TB(g.rdx, j)) * rp[pl_idx(g, li, lj)];
        }
    }
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s0 += __shfl_down(s0, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s0;
    __syncthreads();
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3]; part[2 * blockIdx.x + 1] = 0.0; }
}
// y . (D_r A x) from the five cells without a continuity row.  mode 1: sc[24] = y.(A x) (the denominator, x = w);
// mode 0: sc[25] = (sc[26] - y.(A x)) / sc[24]  (sc[26] = y . r, x = z); mode 2 + k_defl_divide: the same with an all-reduce in between
__device__ inline double defl_five_cells(const PlStokesOp& op, const double* __restrict__ x);
// lazy correction (one rank): sc[slot] = (y.in - y.(A x)) / sc[24] for x = M^-1 in.  y.in is NOT reduced: the corrected
// preconditioner makes y.(A M~^-1 r) = y.r an identity, so y.p and y.s follow from scalar recurrences (sc[32] = y.r, sc[33] = y.p,
// sc[34] = y.s; initialised by k_defl_init at every (re)start, advanced here and in bicg_derive) -- no pass over the vector
__global__ void k_defl_coef_lazy(PlStokesOp op, const double* __restrict__ x, double* __restrict__ sc, int slot, int for_s) {
    double yin = sc[33];                                        // in = p
    if (for_s) { yin = sc[32] - sc[2] * sc[33]; sc[34] = yin; }  // in = s = r - alpha v,  y.v = y.p
    const double yAx = -defl_five_cells(op, x);
    sc[26] = yin;
    sc[slot] = (sc[24] != 0.0 && isfinite(sc[24])) ? (yin - yAx) / sc[24] : 0.0;
}
__global__ void k_defl_init(double* __restrict__ sc) { sc[33] = sc[32]; sc[34] = sc[32]; }     // p = r at a (re)start
__global__ void k_defl_coef(PlStokesOp op, const double* __restrict__ x, double* __restrict__ sc, int mode) {
    const double yAx = -defl_five_cells(op, x);
    if (mode == 1) sc[24] = yAx;                                                                  // (this rank's share on several ranks)
    else if (mode == 2) sc[27] = sc[26] - yAx;                                                    // this rank's share of the numerator
    else sc[25] = (sc[24] != 0.0 && isfinite(sc[24])) ? (sc[26] - yAx) / sc[24] : 0.0;
}
__device__ inline double defl_five_cells(const PlStokesOp& op, const double* __restrict__ x) {
    const PlGeom& g = op.g;
    const int ci[5] = {op.anchor_i, 0, 0, g.nz - 2, g.nz - 2}, cj[5] = {op.anchor_j, 0, g.nx - 2, 0, g.nx - 2};
    const double* vz = x; const double* vx = x + g.plane;
    double five = 0.0;
    for (int k = 0; k < 5; k++) {
        const int i = ci[k], j = cj[k];
        if (i < 0 || j < 0 || i >= g.nz - 1 || j >= g.nx - 1) continue;
        if (k > 0 && i == op.anchor_i && j == op.anchor_j) continue;         // an anchor in a corner counts once
        if (i < g.gi0 || i >= g.gi0 + g.lnz || j < g.gj0 || j >= g.gj0 + g.lnx) continue;     // another rank's cell
        const long long c = pl_idx(g, i - g.gi0, j - g.gj0);
        five += (vz[c + g.pitch] - vz[c]) / TB(g.rdx, j) + (vx[c + 1] - vx[c]) / TB(g.rdz, i);      // area x div = hx dvz + hz dvx
    }
    return five;
}
__global__ void k_defl_five_local(PlStokesOp op, const double* __restrict__ x, double* __restrict__ dst) { *dst = defl_five_cells(op, x); }
__global__ void k_defl_divide(double* __restrict__ sc) { sc[25] = (sc[24] != 0.0 && isfinite(sc[24])) ? sc[27] / sc[24] : 0.0; }
__global__ void k_axpy_dev_scalar(long long n, double* __restrict__ z, const double* __restrict__ w, const double* __restrict__ sc) {
    const double a = sc[0];
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) z[k] += a * w[k];
}
// the time-step driver announces consecutive solves of one slowly changing model (deflation vector kept between them)
void pl_stokes_deflation(pl_ctx* ctx, bool persistent) { solver_of(ctx)->defl_persistent = persistent; }

// Solve A x = b for device vectors (3 planes, UNSCALED b).  x is S->x on return.
int pl_stokes_solve_device(pl_ctx* ctx, const double* b_dev, bool use_x0, double rtol, int maxit,
                           pl_solve_stats* st) {
    PlSolver* S = solver_of(ctx);
    PL_TRY(stokes_alloc(ctx, S));
    PL_TRY(pl_timer_start(ctx));
    static const bool trace_t = getenv("PYLAMP_SOLVER_TRACE") != nullptr;        // phase times (host clock, stream synchronised)
    auto now = [&]() { (void)hipStreamSynchronize(ctx->stream); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tph[5] = {0, 0, 0, 0, 0};
    if (trace_t) tph[0] = now();
    PL_TRY(build_hierarchy(ctx, S));
    if (trace_t) tph[1] = now();
    const PlGeom& g = ctx->geom.d;
    {   // wall stencils of the pressure block (prec_p_value): one rank, some cell at a wall at least 2 x longer along the wall than
        // across (PYLAMP_SCHUR_WALL=0: never, =1: wherever the local aspect ratio reaches 2, whatever the walls look like)
        const std::vector<double>& zc = ctx->geom.zc; const std::vector<double>& xc = ctx->geom.xc;
        const int nz = ctx->nz, nx = ctx->nx;
        double dzmin = 1e300, dzmax = 0.0, dxmin = 1e300, dxmax = 0.0;
        for (int i = 0; i + 1 < nz; i++) { dzmin = std::min(dzmin, zc[i + 1] - zc[i]); dzmax = std::max(dzmax, zc[i + 1] - zc[i]); }
        for (int j = 0; j + 1 < nx; j++) { dxmin = std::min(dxmin, xc[j + 1] - xc[j]); dxmax = std::max(dxmax, xc[j + 1] - xc[j]); }
        const double wx = std::max(xc[1] - xc[0], xc[nx - 1] - xc[nx - 2]) / dzmin;        // widest aspect of a cell in the x-wall columns
        const double wz = std::max(zc[1] - zc[0], zc[nz - 1] - zc[nz - 2]) / dxmin;        // ... of a cell in the z-wall rows
        const char* e = getenv("PYLAMP_SCHUR_WALL");
        const int knob = e ? atoi(e) : -1;
        ctx->sop.wall_ps = (ctx->nranks == 1 && nz >= 9 && nx >= 9 && knob != 0 && (knob == 1 || wx >= 2.0 || wz >= 2.0)) ? 1 : 0;
        (void)dzmax; (void)dxmax;
    }
    PlStokesOp sop = ctx->sop;
    PlStokesOp sop_scaled = sop; sop_scaled.scaled = 1;
    S->napply = 0; S->nprec = 0;
    S->l0_mixed_now = rtol >= 1e-8 && use_x0;                // warm-started solves of a time loop (see stokes_precond_t)
    // one rank with the scalars on the device: the deflation correction is applied lazily (PlSolver::Awdefl)
    S->defl_lazy = dots_on_device(ctx, g) && !getenv("PYLAMP_HOST_SCALARS") &&
                   !(getenv("PYLAMP_DEFL_LAZY") && atoi(getenv("PYLAMP_DEFL_LAZY")) == 0);
    VecOp A = [&](const double* in, double* out) -> int {
        PL_TRY(pl_halo(ctx, g, (double*)in, 3, g.plane));
        // y = D_r A x in one pass; for a vector the preconditioner has just produced: + c A w (its lazy deflation correction)
        if (S->defl_lazy && S->defl_active && ctx->nranks == 1 && (in == S->y || in == S->z))
            pl_launch_stokes_apply(ctx, sop_scaled, in, out, S->Awdefl, S->scal + (in == S->y ? 30 : 31));
        else pl_launch_stokes_apply(ctx, sop_scaled, in, out);
        S->napply++;
        return 0;
    };
    const long long n3v = 3 * g.plane;
    VecOp M = [&](const double* in, double* out) -> int {
        PL_TRY(stokes_precond(ctx, S, in, out));
        if (S->defl_active) {                               // z += w y.(r - A z) / y.(A w), all scalars on the device
            if (S->defl_lazy && ctx->nranks > 1 && (out == S->y || out == S->z)) {   // several ranks: this rank's share of the five-cell sum; it rides in
                hipLaunchKernelGGL(k_defl_five_local, dim3(1), dim3(1), 0, ctx->stream, sop, (const double*)out, S->scal + 36);      // the reduction that follows
                return 0;
            }
            if (S->defl_lazy && (out == S->y || out == S->z)) {       // the coefficient only: A and the iterate update do the rest
                hipLaunchKernelGGL(k_defl_coef_lazy, dim3(1), dim3(1), 0, ctx->stream, sop, (const double*)out, S->scal, out == S->y ? 30 : 31, out == S->z ? 1 : 0);
                return 0;
            }
            const int nb = g.lnz < DOT_BLOCKS ? g.lnz : DOT_BLOCKS;
            hipLaunchKernelGGL(k_defl_ysum, dim3(nb), dim3(256), 0, ctx->stream, sop, in + 2 * g.plane, S->scal + PL_SCAL_N);
            hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, ctx->stream, nb, S->scal + PL_SCAL_N, S->scal + 26, 0, 0.0);
            if (ctx->nranks > 1) {                          // one scalar per application: y.r and the five cells live on different ranks
                hipLaunchKernelGGL(k_defl_coef, dim3(1), dim3(1), 0, ctx->stream, sop, (const double*)out, S->scal, 2);
                PL_TRY(pl_comm_allreduce_dev(ctx, S->scal + 27, 1));
                hipLaunchKernelGGL(k_defl_divide, dim3(1), dim3(1), 0, ctx->stream, S->scal);
            } else
                hipLaunchKernelGGL(k_defl_coef, dim3(1), dim3(1), 0, ctx->stream, sop, (const double*)out, S->scal, 0);
            hipLaunchKernelGGL(k_axpy_dev_scalar, grid1d(n3v), dim3(256), 0, ctx->stream, n3v, out, (const double*)S->wdefl, (const double*)(S->scal + 25));
        }
        return 0;
    };
    BicgVecs w{S->r, S->rt, S->p, S->v, S->s, S->t, S->y, S->z, S->xb, S->dx, S->r0};
    S->defl_active = false;
    if (S->defl_enable && !sop.surfstab && S->levels.size() > 1) {
        if (!S->wdefl) { PL_TRY(dmalloc0(ctx, &S->wdefl, (size_t)n3v * sizeof(double))); PL_TRY(dmalloc0(ctx, &S->udefl, (size_t)n3v * sizeof(double))); S->defl_valid = false; }
        if (!S->Awdefl) PL_TRY(dmalloc0(ctx, &S->Awdefl, (size_t)n3v * sizeof(double)));
        hipLaunchKernelGGL(k_defl_u, grid2d(g), dim3(64, 4), 0, ctx->stream, sop, S->udefl);
        // How good is the vector kept from the previous solve on this context for THIS operator?  ||u - A w|| / ||u||
        // (one operator application).  <= 0.1: use it as it is (measured at 2049^2: 2-3e-2 after one time step, and the
        // iteration count does not notice); < 0.5 (a few time steps later, a similar model): refresh it from itself with the
        // old deflation active -- a few iterations; otherwise (first solve, another problem): from zero.
        double q = 1.0;
        if (S->defl_valid) {
            double dq[2];
            PL_TRY(A(S->wdefl, S->Awdefl));                 // kept: A w under this solve's operator (lazy correction)
            hipLaunchKernelGGL(k_axpy_out, grid1d(n3v), dim3(256), 0, ctx->stream, n3v, S->s, (const double*)S->udefl, (const double*)S->Awdefl, -1.0);
            PL_TRY(dots(ctx, S, g, 3, S->s, S->s, S->udefl, S->udefl, dq));
            q = (dq[1] > 0.0 && std::isfinite(dq[0])) ? std::sqrt(dq[0] / dq[1]) : 1.0;
        }
        static const bool trace_d = getenv("PYLAMP_SOLVER_TRACE") != nullptr;
        if (trace_d) fprintf(stderr, "[pylamp deflation] kept vector: ||u - A w|| / ||u|| = %.3e\n", q);
        static const double q_refresh = 0.3;     // (0.1 until round 3: 0.45 costs no iteration at 2049^2 and halves the refreshes)
        const bool reuse = S->defl_valid && q < std::max(0.5, q_refresh);
        if (reuse) {                                        // denominator y.(A w) of the old w under the new coefficients
            hipLaunchKernelGGL(k_defl_coef, dim3(1), dim3(1), 0, ctx->stream, sop, (const double*)S->wdefl, S->scal, 1);
            if (ctx->nranks > 1) PL_TRY(pl_comm_allreduce_dev(ctx, S->scal + 24, 1));
            S->defl_active = true;
        }
        if (!reuse || q > q_refresh) {
            if (!reuse) PL_HIP(ctx, hipMemsetAsync(S->wdefl, 0, (size_t)n3v * sizeof(double), ctx->stream));
            pl_solve_stats st2{};
            PL_TRY(bicgstab(ctx, S, g, 3, A, &M, S->udefl, S->wdefl, reuse, 1e-3, 80, w, &st2));
            S->defl_valid = st2.rel_residual < 0.05 && std::isfinite(st2.rel_residual);
            if (S->defl_valid) {
                hipLaunchKernelGGL(k_defl_coef, dim3(1), dim3(1), 0, ctx->stream, sop, (const double*)S->wdefl, S->scal, 1);
                if (ctx->nranks > 1) PL_TRY(pl_comm_allreduce_dev(ctx, S->scal + 24, 1));
                S->defl_active = false;                     // (A below must not take w for a preconditioned vector)
                PL_TRY(A(S->wdefl, S->Awdefl));
            }
            S->defl_active = S->defl_valid;
        }
    }
    if (S->defl_active) {                                   // host copies for the anchor-mode term of the error estimate
        double dw[2];
        PL_TRY(dots(ctx, S, g, 2, S->wdefl, S->wdefl, nullptr, nullptr, dw));
        S->defl_wvel2 = dw[0];
        PL_HIP(ctx, hipMemcpyAsync(S->hpart + PL_PART_N * DOT_BLOCKS, S->scal + 24, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        S->defl_yAw = S->hpart[PL_PART_N * DOT_BLOCKS];
    }
    if (trace_t) tph[2] = now();
    if (b_dev != S->b)
        PL_HIP(ctx, hipMemcpyAsync(S->b, b_dev, (size_t)3 * g.plane * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_stokes_scale_rows, grid2d(g), dim3(64, 4), 0, ctx->stream, sop, S->b);
    // hydrostatic pressure x_h (in S->y) and the dynamic-load reference norm ||D_r (b - A x_h)||
    double d2[2], ref = 0.0;
    // In a time loop (consecutive warm-started solves of one slowly changing model) the reference norm is recomputed every fourth solve
    // only: it is the SCALE of the stopping test -- the density field moves by a fraction of a cell per step, the norm by < 1e-3 of
    // itself -- and costs three kernels, two host round trips, an operator application and a reduction (0.3 ms of a 17 ms solve at
    // 2049^2).  Not where the hydrostatic state itself is needed: cold starts, and systems small enough for the direct fallback.
    static const bool ref_reuse = !(getenv("PYLAMP_REF_REUSE") && atoi(getenv("PYLAMP_REF_REUSE")) == 0);
    const bool reuse_ref = ref_reuse && S->defl_persistent && use_x0 && S->ref_cached > 0.0 && S->ref_age < 3 && !pl_direct_possible(ctx) &&
                           std::fabs(S->ref_kc / sop.Kc - 1.0) < 1e-2;
    if (reuse_ref) { ref = S->ref_cached; S->ref_age++; }
    else {
        const long long n3 = 3 * g.plane;
        double *coltot, *prefix;
        PL_TRY(pl_buf(ctx, "hydro_coltot", (size_t)g.lnx * sizeof(double), &coltot));
        PL_TRY(pl_buf(ctx, "hydro_prefix", (size_t)g.lnx * sizeof(double), &prefix));
        {
            const int nch = (g.lnz + PL_HYDRO_CHUNK - 1) / PL_HYDRO_CHUNK;
            double* ctot;
            PL_TRY(pl_buf(ctx, "hydro_chunks", (size_t)nch * g.lnx * sizeof(double), &ctot));
            hipLaunchKernelGGL(k_hydro_chunk, dim3((g.lnx + 63) / 64, nch), dim3(64), 0, ctx->stream, sop, S->y, ctot);
            hipLaunchKernelGGL(k_hydro_chunk_scan, dim3((g.lnx + 63) / 64), dim3(64), 0, ctx->stream, g, nch, ctot, coltot);
            hipLaunchKernelGGL(k_hydro_add, grid2d(g), dim3(64, 4), 0, ctx->stream, g, S->y, ctot);
        }
        // prefix over the blocks above (same block column) + anchor value: one table [Pz x NX column totals | anchor value] summed
        // over the ranks on the device, read back once
        const int Pz = ctx->Pz, NX = g.nx;
        std::vector<double> hb((size_t)Pz * NX + 1, 0.0);
        const bool own_anchor = sop.anchor_i >= g.gi0 && sop.anchor_i < g.gi0 + g.lnz && sop.anchor_j >= g.gj0 && sop.anchor_j < g.gj0 + g.lnx;
        if (ctx->nranks > 1) {
            double* tab;
            PL_TRY(pl_buf(ctx, "hydro_table", hb.size() * sizeof(double), &tab, false));
            PL_HIP(ctx, hipMemsetAsync(tab, 0, hb.size() * sizeof(double), ctx->stream));
            PL_HIP(ctx, hipMemcpyAsync(tab + (size_t)ctx->pz * NX + g.gj0, coltot, (size_t)g.lnx * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            if (own_anchor)
                PL_HIP(ctx, hipMemcpyAsync(tab + (size_t)Pz * NX, S->y + 2 * g.plane + pl_idx(g, sop.anchor_i - g.gi0, sop.anchor_j - g.gj0),
                                           sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            PL_TRY(pl_comm_allreduce_dev(ctx, tab, (int)hb.size(), 0));
            PL_HIP(ctx, hipMemcpyAsync(hb.data(), tab, hb.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        } else {
            PL_HIP(ctx, hipMemcpyAsync(hb.data() + g.gj0, coltot, (size_t)g.lnx * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            PL_HIP(ctx, hipMemcpyAsync(&hb[(size_t)Pz * NX], S->y + 2 * g.plane + pl_idx(g, sop.anchor_i - g.gi0, sop.anchor_j - g.gj0),
                                       sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        }
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<double> pre((size_t)g.lnx, 0.0);
        for (int q = 0; q < ctx->pz; q++) for (int jj = 0; jj < g.lnx; jj++) pre[jj] += hb[(size_t)q * NX + g.gj0 + jj];
        double pa = hb[(size_t)Pz * NX];
        if (ctx->nranks > 1) {      // the anchor value was taken before the prefix of its own block was added
            int owner = 0; while ((owner + 1) * ((g.nz - 1) / Pz) <= sop.anchor_i && owner + 1 < Pz) owner++;
            for (int q = 0; q < owner; q++) pa += hb[(size_t)q * NX + sop.anchor_j];
        }
        PL_HIP(ctx, hipMemcpyAsync(prefix, pre.data(), (size_t)g.lnx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_hydrostatic_apply_shift, grid2d(g), dim3(64, 4), 0, ctx->stream, sop, S->y, prefix, pa);
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PL_TRY(A(S->y, S->t));
        hipLaunchKernelGGL(k_axpy_out, grid1d(n3), dim3(256), 0, ctx->stream, n3, S->s, S->b, S->t, -1.0);
        PL_TRY(dots(ctx, S, g, 3, S->s, S->s, S->b, S->b, d2));
        ref = std::sqrt(d2[0]);
        // no dynamic load (density contrasts below 1e-9 of the hydrostatic load are marker-averaging round-off):
        // fall back to ||b||, against which the hydrostatic state already is the solution
        if (!(ref > 1e-9 * std::sqrt(d2[1]))) ref = 0.0;
        if (pl_direct_possible(ctx)) {                               // kept for the direct fallback (S->y is a work vector of the iteration)
            if (!S->xh) PL_TRY(dmalloc0(ctx, &S->xh, (size_t)n3 * sizeof(double)));
            PL_HIP(ctx, hipMemcpyAsync(S->xh, S->y, (size_t)n3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        }
        if (!use_x0) {                                               // cold start from the hydrostatic state
            PL_HIP(ctx, hipMemcpyAsync(S->x, S->y, (size_t)n3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            use_x0 = true;
        }
        S->ref_cached = ref; S->ref_age = 0; S->ref_kc = sop.Kc;
    }
    hipLaunchKernelGGL(k_close_constraints, grid2d(g), dim3(64, 4), 0, ctx->stream, sop, S->levels[0]->op, S->b, S->x);
    if (trace_t) tph[3] = now();
    // Viscosity contrast beyond the validated envelope (smooth variations over 1e3..1e6, jumps of 1e3): the row scaling divides a
    // momentum row by its viscosity, so a residual that is small in the scaled norm can still be a large FORCE inside a stiff
    // inclusion -- and what that force does to the velocities is decided by the WEAK fluid around it.  Measured on the reference's
    // stock model 5 (sphere 1e12 in 1e2, 201 x 41): scaled residual 7e-11, velocity error 4e-3, estimate 5e-9; the unscaled residual
    // ||b - A x|| / ||b|| is 1.4e-5 there against 1e-8 for the reference's direct solve (tools/model5_probe.py).  Hence, beyond
    // PYLAMP_CONTRAST_GATE (3e5; tools/contrast_probe.py: the iteration alone is 6e-8 from the accurate solution at 1e5 and 1.8e-6 at 1e6): where the banded LU fits it solves UP FRONT (error 4e-7 against the refined direct solution);
    // elsewhere the iteration runs and the solve counts as converged only if the UNSCALED residual meets the tolerance too.
    const double contrast_gate = getenv("PYLAMP_CONTRAST_GATE") ? atof(getenv("PYLAMP_CONTRAST_GATE")) : 3e5;
    const bool beyond = contrast_gate > 0.0 && ctx->visc_contrast > contrast_gate;
    // PYLAMP_FORCE_DIRECT=1 (experiments): skip the multigrid-preconditioned iteration where the banded LU fits
    const bool force_direct = ((getenv("PYLAMP_FORCE_DIRECT") && atoi(getenv("PYLAMP_FORCE_DIRECT")) != 0) || beyond) && pl_direct_possible(ctx) &&
                              !getenv("PYLAMP_NO_DIRECT");
    if (force_direct) { st->iterations = 0; st->converged = 0; st->reserved_ = 0; st->rel_residual = 1.0; st->error_estimate = 0.0; }
    else PL_TRY(bicgstab(ctx, S, g, 3, A, &M, S->b, S->x, use_x0, rtol, maxit, w, st, ref, S->etol, 2));
    // Beyond the gate the solve is judged by the UNSCALED residual ||b - A x|| / ||b|| of the iterate it returns: measured on the stock
    // model, velocity error / unscaled residual = 140 (LU path: 4.4e-7 / 3.2e-9) and 290 (iterative path: 4.1e-3 / 1.4e-5), i.e. the
    // domain's n = max(nz, nx) = 201 within a factor 1.5 -- the same amplification the continuity term of the estimate uses.  Hence
    // error_estimate = max(estimate, n x unscaled residual), and -- FP64 cannot do better than ~4e-7 on such a system (the refined
    // direct solve of the oracle needs an extended-precision residual) -- the bound it has to meet is the drop-in's own 1e-6.
    auto judge_unscaled = [&]() -> int {
        double du[2];
        PL_TRY(A(S->x, S->t));
        hipLaunchKernelGGL(k_stokes_unscaled_pair, grid2d(g), dim3(64, 4), 0, ctx->stream, sop, (const double*)S->b, (const double*)S->t, S->s, S->t);
        PL_TRY(dots(ctx, S, g, 3, S->s, S->s, S->t, S->t, du));
        const double rel_u = du[1] > 0.0 ? std::sqrt(du[0] / du[1]) : 0.0;
        const double est_u = (double)std::max(ctx->nz, ctx->nx) * rel_u;
        if (trace_t) fprintf(stderr, "[pylamp stokes] viscosity contrast %.1e beyond the gate: unscaled relative residual %.3e, estimate %.3e\n", ctx->visc_contrast, rel_u, est_u);
        st->error_estimate = std::max(st->error_estimate, est_u);
        st->converged = (st->rel_residual <= rtol && std::isfinite(est_u) && est_u <= 1e-6) ? 1 : 0;
        return 0;
    };
    if (beyond && !force_direct && st->converged) PL_TRY(judge_unscaled());
    if (trace_t) {
        tph[4] = now();
        fprintf(stderr, "[pylamp stokes] phases: hierarchy + eigenvalues %.2f ms, deflation vector %.2f ms, hydrostatic state + reference norm %.2f ms, "
                        "BiCGStab %.2f ms (%d iterations)\n", tph[1] - tph[0], tph[2] - tph[1], tph[3] - tph[2], tph[4] - tph[3], st->iterations);
    }
    st->used_direct = 0;
    // (not when the iteration merely ran into a small caller-chosen maxit: the fallback is for systems the multigrid-preconditioned
    // iteration cannot solve -- stagnation, breakdown, or the default budget exhausted)
    const bool limit_only = st->reserved_ == 1 && maxit < 200;
    st->reserved_ = 0;
    if (!st->converged && !limit_only && pl_direct_possible(ctx) && !getenv("PYLAMP_NO_DIRECT")) {
        // Small system the iteration could not solve (an indefinite velocity block: the reference's stabilisation sign at the
        // Courant step): factorise D_r A (banded LU, pl_direct.hip) and let it precondition the same iteration -- the LU is exact
        // up to rounding, BiCGStab then only refines.  Restart from the hydrostatic state: the failed iterate may be far off.
        PL_TRY(pl_direct_factor(ctx, sop_scaled));
        VecOp MD = [&](const double* in, double* out) -> int { S->nprec++; return pl_direct_solve(ctx, in, out); };
        const int it0 = st->iterations;
        PL_HIP(ctx, hipMemcpyAsync(S->x, S->xh, (size_t)3 * g.plane * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(k_close_constraints, grid2d(g), dim3(64, 4), 0, ctx->stream, sop, S->levels[0]->op, S->b, S->x);
        // MD never computes the lazy deflation coefficients (scal[30], scal[31]): with the deflation left active, A and the iterate
        // update would keep adding stale -- after a breakdown possibly non-finite -- multiples of w.  The LU needs no deflation.
        const bool defl_was = S->defl_active;
        S->defl_active = false;
        const int rc_direct = bicgstab(ctx, S, g, 3, A, &MD, S->b, S->x, true, rtol, 50, w, st, ref, S->etol, 2);
        S->defl_active = defl_was;
        PL_TRY(rc_direct);
        st->iterations += it0;
        st->used_direct = 1;
        if (beyond) PL_TRY(judge_unscaled());
    }
    double ms = 0;
    PL_TRY(pl_timer_stop_ms(ctx, &ms));
    st->solve_ms = ms; st->operator_applies = S->napply; st->precond_applies = S->nprec;
    return 0;
}

double* pl_stokes_solution_device(pl_ctx* ctx) { return solver_of(ctx)->x; }
double* pl_stokes_rhs_buffer_device(pl_ctx* ctx) {
    PlSolver* S = solver_of(ctx);
    if (stokes_alloc(ctx, S)) return nullptr;
    return S->b;
}

extern "C" int pl_stokes_solve(pl_ctx* ctx, const double* rhs, double* x, int use_x0, double rtol, int maxit,
                               pl_solve_stats* stats) {
    if (!ctx->sop_ready) return pl_fail(ctx, "stokes operator not set");
    if (!x) return pl_fail(ctx, "pl_stokes_solve: x is NULL");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PlSolver* S = solver_of(ctx);
    PL_TRY(stokes_alloc(ctx, S));
    const PlGeom& g = ctx->geom.d;
    if (rhs) PL_TRY(pl_vec3_upload(ctx, g, rhs, S->b));
    else pl_launch_stokes_rhs(ctx, ctx->sop, S->b);
    if (use_x0) PL_TRY(pl_vec3_upload(ctx, g, x, S->x));
    pl_solve_stats st{};
    if (rtol <= 0) rtol = 1e-10;
    if (maxit <= 0) maxit = 400;
    S->defl_persistent = false;                  // a solve of its own, not a step of a time loop (pl_step announces those)
    PL_TRY(pl_stokes_solve_device(ctx, S->b, use_x0 != 0, rtol, maxit, &st));
    PL_TRY(pl_vec3_download(ctx, g, S->x, x));
    if (stats) *stats = st;
    return 0;
}

extern "C" int pl_stokes_precond_apply(pl_ctx* ctx, const double* r, double* z) {
    if (!ctx->sop_ready) return pl_fail(ctx, "stokes operator not set");
    if (!r || !z) return pl_fail(ctx, "pl_stokes_precond_apply: NULL argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PlSolver* S = solver_of(ctx);
    PL_TRY(stokes_alloc(ctx, S));
    PL_TRY(build_hierarchy(ctx, S));
    const PlGeom& g = ctx->geom.d;
    PL_TRY(pl_vec3_upload(ctx, g, r, S->s));
    hipLaunchKernelGGL(k_stokes_scale_rows, grid2d(g), dim3(64, 4), 0, ctx->stream, ctx->sop, S->s);
    {   // scale of the FP32 velocity solve, as bicgstab sets it from the residual norm
        double d2[2];
        PL_TRY(dots(ctx, S, g, 3, S->s, S->s, nullptr, nullptr, d2));
        S->kappa = (std::isfinite(d2[0]) && d2[0] > 0.0) ? std::sqrt(3.0 * ctx->nz * ctx->nx / d2[0]) : 1.0;
    }
    PL_TRY(stokes_precond(ctx, S, S->s, S->z));
    PL_TRY(pl_vec3_download(ctx, g, S->z, z));
    return 0;
}

// fp32 = 0: every multigrid level in FP64; 1: FP32 on the large levels (the default, also PYLAMP_MG_FP32)
// min_nodes > 0: smallest level (local nodes) that runs in FP32 (default 200000, PYLAMP_MG_FP32_NODES)
extern "C" int pl_stokes_set_mg_precision(pl_ctx* ctx, int fp32, long long min_nodes) {
    PlSolver* S = solver_of(ctx);
    S->f32_enable = fp32 != 0;
    if (min_nodes > 0) S->f32_min_nodes = min_nodes;
    return 0;
}

extern "C" int pl_stokes_mg_info(pl_ctx* ctx, int* nlevels, double* lmax, int max_levels) {
    PlSolver* S = solver_of(ctx);
    if (nlevels) *nlevels = (int)S->levels.size();
    for (int l = 0; lmax && l < max_levels && l < (int)S->levels.size(); l++) lmax[l] = S->levels[l]->lmax;
    return 0;
}

// Shape of the multigrid hierarchy of the last solve: number of levels, how many of them (a prefix) run in FP32
extern "C" int pl_stokes_mg_precision(pl_ctx* ctx, int* nlevels, int* nlevels_fp32) {
    PlSolver* S = solver_of(ctx);
    int nf = 0;
    for (MgLevel* L : S->levels) nf += L->f32 ? 1 : 0;
    if (nlevels) *nlevels = (int)S->levels.size();
    if (nlevels_fp32) *nlevels_fp32 = nf;
    return 0;
}

// Average duration (HIP events on the context stream) of one Chebyshev sweep of the finest multigrid level,
// the kernel that takes the largest share of a time step.  Needs a built hierarchy (one solve).
extern "C" int pl_stokes_sweep_bench(pl_ctx* ctx, int reps, double* avg_ms) {
    if (!ctx->sop_ready) return pl_fail(ctx, "stokes operator not set");
    PlSolver* S = solver_of(ctx);
    if (S->levels.empty()) return pl_fail(ctx, "pl_stokes_sweep_bench: no multigrid hierarchy yet (solve once first)");
    if (reps < 1) reps = 1;
    PL_HIP(ctx, hipSetDevice(ctx->device));
    MgLevel* L = S->levels[0];
    const PlGeom& g = L->gh.d;
    const double lmax = L->lmax, lmin = lmax / S->cheb_ratio, theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
    const double sigma = theta / delta, rho1 = 1.0 / (2.0 * sigma - 1.0 / sigma);
    const double c1 = rho1 / sigma, c2 = 2.0 * rho1 / delta;           // coefficients of a second sweep
    auto launch = [&]() {      // in the precision the solver uses on this level (pl_stokes_mg_precision)
        if (L->f32)
            hipLaunchKernelGGL((k_vv_sweep2<0, float, float>), pl_grid_rows2(g), dim3(64, 4), 0, ctx->stream, L->opf, (const float*)L->vf[0],
                               (const float*)L->vf[1], (const float*)L->ff, L->vf[2], (float)c1, (float)c2, 1.0f);
        else
            hipLaunchKernelGGL((k_vv_sweep2<0, double, double>), pl_grid_rows2(g), dim3(64, 4), 0, ctx->stream, L->op, (const double*)L->v[0],
                               (const double*)L->v[1], (const double*)L->f, L->v[2], c1, c2, 1.0);
    };
    launch();
    PL_TRY(pl_timer_start(ctx));
    for (int r = 0; r < reps; r++) launch();
    double ms = 0;
    PL_TRY(pl_timer_stop_ms(ctx, &ms));
    PL_HIP(ctx, hipGetLastError());
    if (avg_ms) *avg_ms = ms / reps;
    return 0;
}

// =========================================================================================
// Heat solve: BiCGStab on the Jacobi-scaled operator D^-1 A
// =========================================================================================
__global__ __launch_bounds__(256) void k_heat_dinv(PlHeatOp op, double* __restrict__ v) {
    PL_NODE_PROLOGUE(op.g)
    const int nz = op.g.nz, nx = op.g.nx, p = op.g.pitch;
    double dg;
    if (i == 0) dg = (op.bc[0] == PL_BC_FIXTEMP) ? 1.0 : -op.kz[c] * TB(op.g.rdz, 0);
    else if (i == nz - 1) dg = (op.bc[2] == PL_BC_FIXTEMP) ? 1.0 : op.kz[c - p] * TB(op.g.rdz, nz - 2);
    else if (j == 0) dg = (op.bc[1] == PL_BC_FIXTEMP) ? 1.0 : -op.kx[c] * TB(op.g.rdx, 0);
    else if (j == nx - 1) dg = (op.bc[3] == PL_BC_FIXTEMP) ? 1.0 : op.kx[c - 1] * TB(op.g.rdx, nx - 2);
    else
        dg = -op.rhocp_inv_dt[c] * ((op.kx[c] * TB(op.g.rdx, j) + op.kx[c - 1] * TB(op.g.rdx, j - 1)) * TB(op.rdxb, j) +
                                    (op.kz[c] * TB(op.g.rdz, i) + op.kz[c - p] * TB(op.g.rdz, i - 1)) * TB(op.rdzb, i)) - 1.0;
    v[c] /= dg;
}

// -----------------------------------------------------------------------------------------------------------------
// Conjugate gradients on the SYMMETRISED heat system (one rank).
// The reference's backward-Euler rows (pylamp_diff.py:157-179) read  c [ (k_e dT_e - k_w dT_w) / dxb + (k_n dT_n - k_s dT_s) / dzb ] - T
// = -T_old - c H  with c = dt / (rho Cp); multiplied by  S = -dxb dzb / c  (cell volume x heat capacity / dt) they become
//     w_e (T - T_e) + w_w (T - T_w) + w_n (T - T_n) + w_s (T - T_s) + m T,     w_e = k_e dzb = w_w of the node to the east, ...
// a symmetric M-matrix with positive diagonal: SPD once the wall rows are eliminated -- FIXTEMP: the wall value is known
// (pylamp_diff.py:99-152), FIXFLOW: the wall value is its inner neighbour's plus a constant, so the coupling drops out of the
// matrix.  The iteration solves for the CORRECTION d of a guess x0 whose wall nodes satisfy their own equations exactly
// (k_heat_close_walls), so that d vanishes on FIXTEMP walls and copies its inner neighbour on FIXFLOW walls, and the reduced
// residual is S times the interior rows of the reference system's residual.  Preconditioner: the diagonal.
// One operator application per iteration (BiCGStab: two), two fused kernels per iteration:
//     k_heat_cg_apply :  p <- z + beta p (new direction, written once),  q = A_red p,  partial p.q
//     k_heat_cg_update:  d += alpha p,  r -= alpha q,  z = r / diag,  partial r.z, z.z
// Stopping test: || D^-1 r || <= rtol || D^-1 b || over the interior rows -- the quantity the BiCGStab path tests -- confirmed on
// the TRUE residual at the end.  pl_solve_stats.error_estimate: the Hestenes-Stiefel estimate of the energy-norm error,
// sqrt(sum of the last 4 alpha_j r_j.z_j), relative to the mass-weighted norm of the temperature.
struct HeatRed { double w[4]; double m; };          // E, W, N, S couplings kept in the matrix, and the mass term
__device__ inline HeatRed heat_red(const PlHeatOp& op, int i, int j, long long c) {
    const PlGeom& g = op.g;
    const int p = g.pitch;
    const double dzb = 1.0 / TB(op.rdzb, i), dxb = 1.0 / TB(op.rdxb, j);
    HeatRed h;
    h.w[0] = op.kx[c] * TB(g.rdx, j) * dzb; h.w[1] = op.kx[c - 1] * TB(g.rdx, j - 1) * dzb;
    h.w[2] = op.kz[c] * TB(g.rdz, i) * dxb; h.w[3] = op.kz[c - p] * TB(g.rdz, i - 1) * dxb;
    h.m = dxb * dzb / op.rhocp_inv_dt[c];
    // a FIXFLOW wall node follows its inner neighbour: no coupling (bc = [z0, x0, zL, xL])
    if (j + 1 == g.nx - 1 && op.bc[3] != PL_BC_FIXTEMP) h.w[0] = 0.0;
    if (j - 1 == 0 && op.bc[1] != PL_BC_FIXTEMP) h.w[1] = 0.0;
    if (i + 1 == g.nz - 1 && op.bc[2] != PL_BC_FIXTEMP) h.w[2] = 0.0;
    if (i - 1 == 0 && op.bc[0] != PL_BC_FIXTEMP) h.w[3] = 0.0;
    return h;
}
// wall nodes of x from their own equations and the current interior values; pass 0: x-walls (rows 1 .. nz-2), pass 1: z-walls
// (all columns: they own the corners and read the x-wall nodes of pass 0)
__global__ __launch_bounds__(256) void k_heat_close_walls(PlHeatOp op, double bz0, double bx0, double bzL, double bxL, double* __restrict__ x, int pass) {
    const PlGeom& g = op.g;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int p = g.pitch;
    // (every wall node is closed by the rank that owns it; its inner neighbour lies in the same block: blocks hold >= 2 rows / columns)
    if (pass == 0) {
        if (t >= 2 * (g.nz - 2)) return;
        const int i = 1 + t % (g.nz - 2), right = t / (g.nz - 2), j = right ? g.nx - 1 : 0;
        if (i < g.gi0 || i >= g.gi0 + g.lnz || j < g.gj0 || j >= g.gj0 + g.lnx) return;
        const long long c = pl_idx(g, i - g.gi0, j - g.gj0);
        if (!right) x[c] = (op.bc[1] == PL_BC_FIXTEMP) ? bx0 : x[c + 1] - bx0 / (op.kx[c] * TB(g.rdx, 0));
        else x[c] = (op.bc[3] == PL_BC_FIXTEMP) ? bxL : x[c - 1] + bxL / (op.kx[c - 1] * TB(g.rdx, g.nx - 2));
    } else {
        if (t >= 2 * g.nx) return;
        const int j = t % g.nx, bottom = t / g.nx, i = bottom ? g.nz - 1 : 0;
        if (i < g.gi0 || i >= g.gi0 + g.lnz || j < g.gj0 || j >= g.gj0 + g.lnx) return;
        const long long c = pl_idx(g, i - g.gi0, j - g.gj0);
        if (!bottom) x[c] = (op.bc[0] == PL_BC_FIXTEMP) ? bz0 : x[c + p] - bz0 / (op.kz[c] * TB(g.rdz, 0));
        else x[c] = (op.bc[2] == PL_BC_FIXTEMP) ? bzL : x[c - p] + bzL / (op.kz[c - p] * TB(g.rdz, g.nz - 2));
    }
}
// the (possibly extended) block a heat kernel runs on and the owned nodes whose contributions enter the sums
struct HeatOwn { int i0, i1, j0, j1; };                 // global node range [i0, i1) x [j0, j1) this rank owns
#define PL_HEAT_NODE(g)                                                                             \
    const int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y;                \
    const int i = (g).gi0 + li, j = (g).gj0 + lj;                                                    \
    const bool in_blk = lj < (g).lnx && li < (g).lnz;
__device__ inline void heat_block_sum3(double a0, double a1, double a2, double* __restrict__ part) {
    __shared__ double sh[3][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_down(a0, o, 64); a1 += __shfl_down(a1, o, 64); a2 += __shfl_down(a2, o, 64); }
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if ((tid & 63) == 0) { sh[0][tid >> 6] = a0; sh[1][tid >> 6] = a1; sh[2][tid >> 6] = a2; }
    __syncthreads();
    if (tid == 0) {
        const long long b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        part[3 * b] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]; part[3 * b + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        part[3 * b + 2] = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
    }
}
// mode 0 (start): r = S (b - A x) on the interior rows (x with closed walls), dinv = 1 / diag, z = r dinv, p = z, d = 0;
//                 partials: r.z, z.z, and (first call) the reference sum_int (dinv S b)^2 in slot 2
// mode 1 (check): r = S (b - A x) only; partials: (r dinv)^2 in slot 1
__global__ __launch_bounds__(256) void k_heat_cg_residual(PlHeatOp op, const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ r,
                                                          double* __restrict__ dinv, double* __restrict__ z, double* __restrict__ pdir, double* __restrict__ d,
                                                          int mode, double* __restrict__ part, HeatOwn own) {
    const PlGeom& g = op.g;
    PL_HEAT_NODE(g)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (in_blk && i >= 1 && i <= g.nz - 2 && j >= 1 && j <= g.nx - 2) {
        const long long c = pl_idx(g, li, lj);
        const int p = g.pitch;
        const double dzb = 1.0 / TB(op.rdzb, i), dxb = 1.0 / TB(op.rdxb, j), t = x[c];
        const double wE = op.kx[c] * TB(g.rdx, j) * dzb, wW = op.kx[c - 1] * TB(g.rdx, j - 1) * dzb;
        const double wN = op.kz[c] * TB(g.rdz, i) * dxb, wS = op.kz[c - p] * TB(g.rdz, i - 1) * dxb, m = dxb * dzb / op.rhocp_inv_dt[c];
        // S x (reference row) with the actual wall values of x: S b - [ w (t - T_nb) ... + m t ]
        const double Sb = -m * b[c];                       // S b = -dxb dzb / c * b
        const double res = Sb - (wE * (t - x[c + 1]) + wW * (t - x[c - 1]) + wN * (t - x[c + p]) + wS * (t - x[c - p]) + m * t);
        const HeatRed h = heat_red(op, i, j, c);
        const double di = 1.0 / (h.w[0] + h.w[1] + h.w[2] + h.w[3] + h.m);
        r[c] = res;
        const double zz = res * di;
        const bool mine = i >= own.i0 && i < own.i1 && j >= own.j0 && j < own.j1;
        if (mode == 0) { dinv[c] = di; z[c] = zz; pdir[c] = zz; d[c] = 0.0; if (mine) { s0 = res * zz; s1 = zz * zz; s2 = (Sb * di) * (Sb * di); } }
        else if (mine) { s0 = res * zz; s1 = zz * zz; }
    }
    heat_block_sum3(s0, s1, s2, part);
}
// p_new = z + beta p_old at the node and its four neighbours (beta = sc[3]; first iteration: use_beta = 0, p_old holds z), q = A_red p_new
__global__ __launch_bounds__(256) void k_heat_cg_apply(PlHeatOp op, const double* __restrict__ z, const double* __restrict__ pold, double* __restrict__ pnew,
                                                       double* __restrict__ q, const double* __restrict__ sc, int use_beta, double* __restrict__ part) {
    const PlGeom& g = op.g;
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    double s0 = 0.0;
    if (i >= 1 && i <= g.nz - 2 && j >= 1 && j <= g.nx - 2) {
        const long long c = pl_idx(g, i, j);
        const int p = g.pitch;
        const double beta = use_beta ? sc[3] : 0.0;
        auto pv = [&](long long k) { return use_beta ? z[k] + beta * pold[k] : pold[k]; };     // wall entries of z and p are 0 and stay 0
        const double pc = pv(c);
        const HeatRed h = heat_red(op, i, j, c);
        const double qq = h.w[0] * (pc - pv(c + 1)) + h.w[1] * (pc - pv(c - 1)) + h.w[2] * (pc - pv(c + p)) + h.w[3] * (pc - pv(c - p)) + h.m * pc;
        pnew[c] = pc; q[c] = qq;
        s0 = pc * qq;
    }
    heat_block_sum3(s0, 0.0, 0.0, part);
}
// alpha = sc[2]:  d += alpha p,  r -= alpha q,  z = r dinv;  partials r.z, z.z
__global__ __launch_bounds__(256) void k_heat_cg_update(PlGeom g, const double* __restrict__ pdir, const double* __restrict__ q, const double* __restrict__ dinv,
                                                        double* __restrict__ d, double* __restrict__ r, double* __restrict__ z, const double* __restrict__ sc,
                                                        double* __restrict__ part) {
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    double s0 = 0.0, s1 = 0.0;
    if (i >= 1 && i <= g.nz - 2 && j >= 1 && j <= g.nx - 2) {
        const long long c = pl_idx(g, i, j);
        const double alpha = sc[2];
        d[c] += alpha * pdir[c];
        const double rn = r[c] - alpha * q[c];
        r[c] = rn;
        const double zz = rn * dinv[c];
        z[c] = zz;
        s0 = rn * zz; s1 = zz * zz;
    }
    heat_block_sum3(s0, s1, 0.0, part);
}
// sums of the block partials and the CG scalars: sc[0] = r.z (current), sc[1] = z.z, sc[2] = alpha, sc[3] = beta, sc[4] = p.q, sc[5] = reference
// what 0: start (sc[0] = r.z, sc[1] = z.z, sc[5] = ref);  1: after apply (sc[4] = p.q, alpha = sc[0] / p.q);
//      2: after update (beta = r.z_new / sc[0], sc[6] = alpha * old r.z (Hestenes-Stiefel term), sc[0] = r.z_new, sc[1] = z.z);  3: check (sc[1] = z.z)
// (1024 threads: at 2049^2 there are 17 000 partials per sum, which one 256-thread block took 19 us to add -- twice per iteration)
//      4: sc[4] = sum of slot 0, sc[7] = MAXIMUM of slot 1 (k_heat_mass_norm)
__global__ __launch_bounds__(1024) void k_heat_cg_scalars(int nb, const double* __restrict__ part, double* __restrict__ sc, int what) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (what == 4) for (int k = threadIdx.x; k < nb; k += 1024) { a0 += part[3 * k]; a1 = fmax(a1, part[3 * k + 1]); }
    else for (int k = threadIdx.x; k < nb; k += 1024) { a0 += part[3 * k]; a1 += part[3 * k + 1]; a2 += part[3 * k + 2]; }
    __shared__ double sh[3][16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_down(a0, o, 64); a2 += __shfl_down(a2, o, 64);
        const double o1 = __shfl_down(a1, o, 64); a1 = what == 4 ? fmax(a1, o1) : a1 + o1;
    }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a0; sh[1][threadIdx.x >> 6] = a1; sh[2][threadIdx.x >> 6] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a0 = a1 = a2 = 0.0;
        for (int w = 0; w < 16; w++) { a0 += sh[0][w]; a1 = what == 4 ? fmax(a1, sh[1][w]) : a1 + sh[1][w]; a2 += sh[2][w]; }
        if (what == 4) { sc[4] = a0; sc[7] = a1; }
        else if (what == 0) { sc[0] = a0; sc[1] = a1; sc[5] = a2; }
        else if (what == 1) { sc[4] = a0; sc[2] = (a0 > 0.0) ? sc[0] / a0 : 0.0; }
        else if (what == 2) { sc[3] = (sc[0] > 0.0) ? a0 / sc[0] : 0.0; sc[6] = sc[2] * sc[0]; sc[0] = a0; sc[1] = a1; }
        else sc[1] = a1;
    }
}
// slot 0: mass-weighted norm of x; slot 1: the block's MAXIMUM of (sum of the off-diagonal couplings) / diagonal of the reduced
// operator -- the Gershgorin radius rho of D^-1 A_red, whose eigenvalues lie in [1 - rho, 1 + rho] (Chebyshev iteration below)
__global__ __launch_bounds__(256) void k_heat_mass_norm(PlHeatOp op, const double* __restrict__ x, double* __restrict__ part) {
    const PlGeom& g = op.g;
    PL_HEAT_NODE(g)
    double s0 = 0.0, rad = 0.0;
    if (in_blk && i >= 1 && i <= g.nz - 2 && j >= 1 && j <= g.nx - 2) {
        const long long c = pl_idx(g, li, lj);
        s0 = x[c] * x[c] / (TB(op.rdzb, i) * TB(op.rdxb, j) * op.rhocp_inv_dt[c]);
        const HeatRed h = heat_red(op, i, j, c);
        const double off = h.w[0] + h.w[1] + h.w[2] + h.w[3];
        rad = off / (off + h.m);
    }
    __shared__ double shm[4];
    for (int o = 32; o > 0; o >>= 1) rad = fmax(rad, __shfl_down(rad, o, 64));
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if ((tid & 63) == 0) shm[tid >> 6] = rad;
    heat_block_sum3(s0, 0.0, 0.0, part);                  // (has the barrier)
    if (tid == 0) part[3 * ((long long)blockIdx.y * gridDim.x + blockIdx.x) + 1] = fmax(fmax(shm[0], shm[1]), fmax(shm[2], shm[3]));
}
// One Chebyshev sweep on the correction system  A_red d = r0  with the diagonal as preconditioner (three-term form, as the multigrid smoother):
//     d_next = d + c1 (d - d_prev) + c2 dinv (r0 - A_red d)        -- no reduction, one launch per sweep
// Several ranks: launched on the block EXTENDED into the halo (op.g is the extended view, every pointer shifted accordingly), one
// node less per sweep, so that a round of sweeps needs ONE exchange (deep halo, as the multigrid smoother's)
__global__ __launch_bounds__(256) void k_heat_cheb(PlHeatOp op, const double* __restrict__ dcur, const double* __restrict__ dprev, const double* __restrict__ r0,
                                                   const double* __restrict__ dinv, double* __restrict__ dnext, double c1, double c2) {
    const PlGeom& g = op.g;
    PL_HEAT_NODE(g)
    if (!(in_blk && i >= 1 && i <= g.nz - 2 && j >= 1 && j <= g.nx - 2)) return;
    const long long c = pl_idx(g, li, lj);
    const int p = g.pitch;
    const HeatRed h = heat_red(op, i, j, c);
    const double dc = dcur[c];
    const double Ad = h.w[0] * (dc - dcur[c + 1]) + h.w[1] * (dc - dcur[c - 1]) + h.w[2] * (dc - dcur[c + p]) + h.w[3] * (dc - dcur[c - p]) + h.m * dc;
    (void)dinv;                                        // (the reciprocal diagonal is recomputed from the couplings at hand: 8 B/node less)
    const double di = pl_rcp(h.w[0] + h.w[1] + h.w[2] + h.w[3] + h.m);
    dnext[c] = dc + (c1 != 0.0 ? c1 * (dc - dprev[c]) : 0.0) + c2 * di * (r0[c] - Ad);      // wall entries stay 0
}

// extended view of the heat operator: the block grown by e nodes into the halo (clipped at the domain walls); sh = element offset
// of the view's local node (0,0) before the block's: every plane pointer handed to a kernel is shifted back by it
struct HeatExt { PlHeatOp op; long long sh; };
static HeatExt heat_ext(pl_ctx* ctx, const PlHeatOp& hop, int e) {
    HeatExt v; v.op = hop; v.sh = 0;
    if (e <= 0 || ctx->nranks <= 1) return v;
    const PlGeom& g = hop.g;
    const int a = std::min(e, g.gi0), b = std::min(e, g.nz - (g.gi0 + g.lnz)), c = std::min(e, g.gj0), d = std::min(e, g.nx - (g.gj0 + g.lnx));
    v.op.g.gi0 -= a; v.op.g.lnz += a + b; v.op.g.gj0 -= c; v.op.g.lnx += c + d;
    v.sh = (long long)a * g.pitch + c;
    v.op.kz -= v.sh; v.op.kx -= v.sh; v.op.rhocp_inv_dt -= v.sh;
    return v;
}

// CG / Chebyshev on the symmetrised heat system.  Several ranks (Chebyshev only): the sweeps need NO reduction; a round of up to
// PL_RING - 1 sweeps runs on the block extended into the halo (one node less per sweep) after ONE exchange of the iterates, so a
// solve costs ~5 halo exchanges and 4 small all-reduces (Gershgorin radius, mass norm, and the residual norms of its two passes)
// where the BiCGStab path took 17 exchanges and 16 all-reduces.
static int heat_solve_cg(pl_ctx* ctx, PlSolver* S, const double* b_dev, double rtol, int maxit, pl_solve_stats* st, double** x_out,
                         const double* x0_dev, bool* unsuited) {
    const PlGeom& g = ctx->geom.d;
    const size_t pb = (size_t)g.plane * sizeof(double);
    const bool multi = ctx->nranks > 1;
    for (int k = 0; k < 9; k++) if (!S->hc[k]) PL_TRY(dmalloc0(ctx, &S->hc[k], pb));
    if (multi && !S->hc3) PL_TRY(dmalloc0(ctx, &S->hc3, 3 * pb));            // the three Chebyshev iterates as ONE allocation: one exchange for all
    if (!S->scal) {
        PL_TRY(dmalloc0(ctx, &S->scal, (PL_SCAL_N + PL_PART_N * DOT_BLOCKS) * sizeof(double)));
        PL_HIP(ctx, hipHostMalloc((void**)&S->hpart, (PL_PART_N * DOT_BLOCKS + PL_SCAL_N) * sizeof(double)));
    }
    // deepest extension a kernel of this solve runs on: it reads the coefficient planes one node further (ring PL_RING deep)
    const int E = multi ? std::min(PL_RING - 1, std::min(g.lnz, g.lnx)) : 0;
    const HeatExt X = heat_ext(ctx, ctx->hop, E > 0 ? E - 1 : 0);             // residual / first sweep: extended by E - 1
    const dim3 grX = grid2d(X.op.g), gr = grid2d(g), bl(64, 4);
    const int nb = (int)(grX.x * grX.y);
    if (!S->hc_part || S->hc_nb < nb) {
        if (S->hc_part) (void)hipFree(S->hc_part);
        PL_TRY(dmalloc0(ctx, &S->hc_part, (size_t)3 * nb * sizeof(double)));
        S->hc_nb = nb;
    }
    PL_TRY(pl_timer_start(ctx));
    const PlHeatOp hop = ctx->hop;
    const HeatOwn own{g.gi0, g.gi0 + g.lnz, g.gj0, g.gj0 + g.lnx};
    double *x = S->hc[0], *d = S->hc[1], *r = S->hc[2], *z = S->hc[3], *q = S->hc[4], *dinv = S->hc[5], *pa = S->hc[6], *pbuf = S->hc[7];
    if (multi) { d = S->hc3; pa = S->hc3 + g.plane; pbuf = S->hc3 + 2 * g.plane; }
    double* sc = S->scal; double* hs = S->hpart;
    if (x0_dev) PL_HIP(ctx, hipMemcpyAsync(x, x0_dev, pb, hipMemcpyDeviceToDevice, ctx->stream));
    else PL_HIP(ctx, hipMemsetAsync(x, 0, pb, ctx->stream));
    const double* bv = ctx->heat_bcvalue;
    auto close_walls = [&](double* v) {
        hipLaunchKernelGGL(k_heat_close_walls, dim3((2 * (g.nz - 2) + 255) / 256), dim3(256), 0, ctx->stream, hop, bv[0], bv[1], bv[2], bv[3], v, 0);
        hipLaunchKernelGGL(k_heat_close_walls, dim3((2 * g.nx + 255) / 256), dim3(256), 0, ctx->stream, hop, bv[0], bv[1], bv[2], bv[3], v, 1);
    };
    auto fetch = [&]() -> int {
        PL_HIP(ctx, hipMemcpyAsync(hs, sc, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    };
    close_walls(x);
    hipLaunchKernelGGL(k_heat_mass_norm, gr, bl, 0, ctx->stream, hop, (const double*)x, S->hc_part);
    hipLaunchKernelGGL(k_heat_cg_scalars, dim3(1), dim3(1024), 0, ctx->stream, (int)(gr.x * gr.y), (const double*)S->hc_part, sc, 4);     // -> sc[4], sc[7]
    if (multi) { PL_TRY(pl_comm_allreduce_dev(ctx, sc + 4, 1, 0)); PL_TRY(pl_comm_allreduce_dev(ctx, sc + 7, 1, 2)); }
    PL_TRY(fetch());
    const double xmass = hs[4];
    // Chebyshev iteration instead of CG where its sweep count is known to be small: the eigenvalues of D^-1 A_red lie in
    // [1 - rho, 1 + rho] (Gershgorin; rho = 1 / (1 + m / sum w) < 1 for a backward-Euler step), so k sweeps reduce the residual by
    // 1 / T_k(1 / rho) -- with NO reduction and ONE launch per sweep (CG: 4 launches and 2 reductions per iteration, which is what the
    // 2.2 ms of a 12-iteration solve at 2049^2 were made of).  The true residual is confirmed afterwards as before.
    const double rho_g = hs[7];
    const int cheb_env = getenv("PYLAMP_HEAT_CHEB") ? atoi(getenv("PYLAMP_HEAT_CHEB")) : 1;         // (read per solve: tests switch it)
    const bool cheb_ok = cheb_env != 0 && rho_g > 0.0 && rho_g < 0.97 && std::isfinite(rho_g);
    if (unsuited) *unsuited = false;
    if (multi && !cheb_ok) {                              // (rho is the same on every rank: a collective decision) -- the caller runs BiCGStab
        if (unsuited) *unsuited = true;
        double ms0 = 0; PL_TRY(pl_timer_stop_ms(ctx, &ms0));
        return 0;
    }
    st->iterations = 0; st->converged = 0; st->rel_residual = 0.0; st->error_estimate = 0.0;
    S->napply = 0;
    double ref = 0.0, hist[4] = {0, 0, 0, 0};
    int it = 0;
    if (multi && E > 0) PL_TRY(pl_halo(ctx, g, (double*)b_dev, 1, g.plane, E));            // the right-hand side, once
    for (int pass = 0; pass < 5; pass++) {               // (re)start from the true residual of the current x = x0 + d (the last pass only checks)
        if (multi && E > 0) {
            PL_TRY(pl_halo(ctx, g, x, 1, g.plane, E));                                     // x with closed walls, E deep: residual on the block + (E - 1)
            PL_HIP(ctx, hipMemsetAsync(d, 0, pb, ctx->stream));                            // the zero correction, also beyond the region the start kernel writes
        }
        hipLaunchKernelGGL(k_heat_cg_residual, grX, bl, 0, ctx->stream, X.op, b_dev - X.sh, (const double*)(x - X.sh), r - X.sh, dinv - X.sh, z - X.sh, pa - X.sh,
                           d - X.sh, 0, S->hc_part, own);
        hipLaunchKernelGGL(k_heat_cg_scalars, dim3(1), dim3(1024), 0, ctx->stream, nb, (const double*)S->hc_part, sc, 0);
        if (multi) { PL_HIP(ctx, hipMemcpyAsync(sc + 2, sc + 5, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream)); PL_TRY(pl_comm_allreduce_dev(ctx, sc, 3, 0));
                     PL_HIP(ctx, hipMemcpyAsync(sc + 5, sc + 2, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream)); }
        S->napply++;
        PL_TRY(fetch());
        if (pass == 0) ref = std::sqrt(hs[5]);
        double znorm = std::sqrt(hs[1]);
        if (!(ref > 0.0)) { st->converged = 1; break; }      // b = 0 on the interior: x0 with closed walls is the answer
        st->rel_residual = znorm / ref;
        if (znorm <= rtol * ref || it >= maxit || pass == 4) { st->converged = znorm <= rtol * ref ? 1 : 0; break; }
        double* pold = pa; double* pnew = pbuf;
        bool first = true, ok = true;
        if (cheb_ok) {
            // sweeps needed for ||D^-1 r|| <= rtol ref with a margin of 4 (the bound holds in the D-weighted norm)
            const double sigma = 1.0 / rho_g, target = 0.25 * rtol * ref / znorm;
            int ks = 1; double t0 = 1.0, t1 = sigma;
            while (1.0 / t1 > target && ks < 200) { const double t2 = 2.0 * sigma * t1 - t0; t0 = t1; t1 = t2; ks++; }
            ks = std::min(ks, std::max(maxit - it, 1));
            const double theta = 1.0, delta = rho_g;
            double rho_old = 1.0 / sigma;
            double* bufs[3] = {d, pa, pbuf};                      // (cur, prev, free); d = 0 from the start kernel
            int ext = E - 1;                                      // extension of the NEXT sweep (several ranks); r0 and d = 0 are valid that far
            for (int k = 0; k < ks; k++) {
                double c1, c2;
                if (k == 0) { c1 = 0.0; c2 = 1.0 / theta; }
                else { const double rr = 1.0 / (2.0 * sigma - rho_old); c1 = rr * rho_old; c2 = 2.0 * rr / delta; rho_old = rr; }
                if (multi && ext < 0) {                           // the halo of the iterates is used up: ONE exchange of all three buffers
                    PL_TRY(pl_halo(ctx, g, S->hc3, 3, g.plane, E));
                    ext = E - 1;
                }
                const HeatExt V = heat_ext(ctx, hop, multi ? ext : 0);
                hipLaunchKernelGGL(k_heat_cheb, grid2d(V.op.g), bl, 0, ctx->stream, V.op, (const double*)(bufs[0] - V.sh), (const double*)(bufs[1] - V.sh),
                                   (const double*)(r - V.sh), (const double*)(dinv - V.sh), bufs[2] - V.sh, c1, c2);
                double* nx = bufs[2]; bufs[2] = bufs[1]; bufs[1] = bufs[0]; bufs[0] = nx;
                ext--;
                S->napply++;
            }
            it += ks;
            hipLaunchKernelGGL(k_add_inplace, grid1d(g.plane), dim3(256), 0, ctx->stream, (long long)g.plane, x, (const double*)bufs[0]);
            close_walls(x);
            continue;                                             // the next pass confirms the TRUE residual (and sweeps again if need be)
        }
        while (it < maxit) {
            it++;
            hipLaunchKernelGGL(k_heat_cg_apply, gr, bl, 0, ctx->stream, hop, (const double*)z, (const double*)pold, pnew, q, (const double*)sc, first ? 0 : 1, S->hc_part);
            hipLaunchKernelGGL(k_heat_cg_scalars, dim3(1), dim3(1024), 0, ctx->stream, nb, (const double*)S->hc_part, sc, 1);
            hipLaunchKernelGGL(k_heat_cg_update, gr, bl, 0, ctx->stream, g, (const double*)pnew, (const double*)q, (const double*)dinv, d, r, z, (const double*)sc, S->hc_part);
            hipLaunchKernelGGL(k_heat_cg_scalars, dim3(1), dim3(1024), 0, ctx->stream, nb, (const double*)S->hc_part, sc, 2);
            S->napply++;
            std::swap(pold, pnew); first = false;
            // the host only looks at the scalars (one stream synchronisation, ~35 us) where the iteration may end: from three iterations
            // before the previous solve's count on (consecutive time steps need the same number to within one or two)
            if (it + 3 < S->hc_last_its && it < maxit) continue;
            PL_TRY(fetch());
            hist[it & 3] = hs[6];
            znorm = std::sqrt(hs[1]);
            if (!std::isfinite(znorm) || !(hs[4] > 0.0)) { ok = false; break; }
            if (znorm <= rtol * ref) break;
        }
        // x = x0 + d with its wall nodes re-closed; the next pass confirms the TRUE residual (and restarts from it if need be)
        hipLaunchKernelGGL(k_add_inplace, grid1d(g.plane), dim3(256), 0, ctx->stream, (long long)g.plane, x, (const double*)d);
        close_walls(x);
        if (!ok) break;
    }
    PL_HIP(ctx, hipGetLastError());
    st->iterations = it;
    S->hc_last_its = st->converged ? it : 0;
    if (xmass > 0.0) st->error_estimate = std::sqrt(std::fabs(hist[0]) + std::fabs(hist[1]) + std::fabs(hist[2]) + std::fabs(hist[3])) / std::sqrt(xmass);
    // Chebyshev: the energy norm of the error is r.A^-1 r <= r.D^-1 r / lambda_min = (r.z) / (1 - rho), with r.z of the confirmed residual
    if (cheb_ok && xmass > 0.0 && hs[0] >= 0.0) st->error_estimate = std::sqrt(hs[0] / (1.0 - rho_g)) / std::sqrt(xmass);
    double ms = 0;
    PL_TRY(pl_timer_stop_ms(ctx, &ms));
    st->solve_ms = ms; st->operator_applies = S->napply; st->precond_applies = 0;
    *x_out = x;
    return 0;
}

int pl_heat_solve_device(pl_ctx* ctx, const double* b_dev, double rtol, int maxit, pl_solve_stats* st,
                         double** x_out, const double* x0_dev) {
    PlSolver* S = solver_of(ctx);
    // one rank: CG on the symmetrised system (PYLAMP_HEAT_CG=0: the Jacobi-scaled BiCGStab below, which several ranks still use)
    static const bool use_cg = !(getenv("PYLAMP_HEAT_CG") && atoi(getenv("PYLAMP_HEAT_CG")) == 0);
    if (use_cg && ctx->nz >= 3 && ctx->nx >= 3 && (ctx->nranks == 1 || std::min(ctx->geom.d.lnz, ctx->geom.d.lnx) >= 2)) {
        bool unsuited = false;
        PL_TRY(heat_solve_cg(ctx, S, b_dev, rtol, maxit, st, x_out, x0_dev, &unsuited));
        if (!unsuited) return 0;             // (several ranks and a Gershgorin radius >= 0.97: the Jacobi-scaled BiCGStab below)
    }
    const PlGeom& g = ctx->geom.d;
    size_t pb = (size_t)g.plane * sizeof(double);
    for (int k = 0; k < 11; k++) if (!S->h[k]) PL_TRY(dmalloc0(ctx, &S->h[k], pb));
    if (!S->scal) {
        PL_TRY(dmalloc0(ctx, &S->scal, (PL_SCAL_N + PL_PART_N * DOT_BLOCKS) * sizeof(double)));
        PL_HIP(ctx, hipHostMalloc((void**)&S->hpart, (PL_PART_N * DOT_BLOCKS + PL_SCAL_N) * sizeof(double)));
    }
    PL_TRY(pl_timer_start(ctx));
    PlHeatOp hop = ctx->hop;
    S->napply = 0;
    VecOp A = [&](const double* in, double* out) -> int {
        PL_TRY(pl_halo(ctx, g, (double*)in, 1, g.plane));
        pl_launch_heat_apply(ctx, hop, in, out, true);          // D^-1 A in one pass
        S->napply++;
        return 0;
    };
    double* b = S->h[8];
    PL_HIP(ctx, hipMemcpyAsync(b, b_dev, pb, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_heat_dinv, grid2d(g), dim3(64, 4), 0, ctx->stream, hop, b);
    BicgVecs w{S->h[0], S->h[1], S->h[2], S->h[3], S->h[4], S->h[5], nullptr, nullptr, S->h[6], S->h[9], S->h[10]};
    if (x0_dev) PL_HIP(ctx, hipMemcpyAsync(S->h[7], x0_dev, pb, hipMemcpyDeviceToDevice, ctx->stream));     // start from the caller's guess
    PL_TRY(bicgstab(ctx, S, g, 1, A, nullptr, b, S->h[7], x0_dev != nullptr, rtol, maxit, w, st));
    double ms = 0;
    PL_TRY(pl_timer_stop_ms(ctx, &ms));
    st->solve_ms = ms; st->operator_applies = S->napply; st->precond_applies = 0;
    *x_out = S->h[7];
    return 0;
}

extern "C" int pl_heat_solve(pl_ctx* ctx, const double* rhs, double* x, double rtol, int maxit,
                             pl_solve_stats* stats) {
    if (!ctx->hop_ready) return pl_fail(ctx, "heat operator not set");
    if (!x) return pl_fail(ctx, "pl_heat_solve: x is NULL");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    double* db;
    PL_TRY(pl_buf(ctx, "api_hx", (size_t)g.plane * sizeof(double), &db));
    if (rhs) PL_TRY(pl_plane_upload(ctx, g, rhs, db));
    else pl_launch_heat_rhs(ctx, ctx->hop, ctx->bufs["f_T"], ctx->bufs["H"], db);
    pl_solve_stats st{};
    if (rtol <= 0) rtol = 1e-12;
    if (maxit <= 0) maxit = 2000;
    double* xs = nullptr;
    PL_TRY(pl_heat_solve_device(ctx, db, rtol, maxit, &st, &xs));
    PL_TRY(pl_plane_download(ctx, g, xs, x));
    if (stats) *stats = st;
    return 0;
}
