// 3-D staggered Stokes + heat (BASELINE config 5).  PARITY UNPINNED: the reference implements 2-D only
// (pylamp_const.py:6 "DIM = 2  # 2 or 3, currently only 2 implemented"; pylamp_stokes.py:30-35 prints
// "NOT IMPLEMENTED" for dim != 2).  What it does fix is the intent: axis order z, x, y (pylamp_const.py:9-13), arrays
// (nz, nx, ny), pressure as equation IP = DIM = 3, DOF order iz*nx*ny*4 + ix*ny*4 + iy*4 + ieq (pylamp_stokes.py:24).
// This file extends the 2-D rows of pylamp_stokes.py:376-518 and pylamp_diff.py:157-179 dimension by dimension
//  so that a y-invariant extrusion of a 2-D problem reproduces the 2-D operator and solution on every
// y-slice (tests/test_hip_3d.py), and is otherwise validated by manufactured solutions and true residuals.
//
// Staggering: vz at (z_i, x_j+1/2, y_k+1/2), vx at (z_i+1/2, x_j, y_k+1/2), vy at (z_i+1/2, x_j+1/2, y_k), P and the
// normal viscosity eta_n at cell centres, the shear viscosity eta_s ONCE at the nodes (z_i, x_j, y_k) and averaged on
// the fly onto the edge a term needs (the 80 B/node layout of SURVEY 8d: x 32 + y 32 + eta_n 8 + eta_s 8).
// One velocity component's row is the same expression under a cyclic permutation of the axes: the kernels are
// templated on the component.  Solver: the 2-D design (pl_solver.hip) in 3-D -- row-scaled BiCGStab, block-triangular
// preconditioner [[A_vv, A_vp],[0, S^]] with one geometric-multigrid V-cycle (Chebyshev-Jacobi smoothing, rediscretised
// coarse operators with natural wall rows, arithmetic viscosity coarsening) for A_vv.  One GPU per problem.
#include "pl_internal.h"
#include <algorithm>
#include <cmath>
#include <functional>

#define P3_PAD 16                    // left padding of a y-line: node k = 0 is 128-B aligned
// Coefficient tables are read through the constant address space: with a wave-uniform index (the z and x axes: blockIdx.z and the
// readfirstlane'd row of the workgroup, K3_PROLOGUE) the load is a scalar one (s_load, scalar cache) instead of 64 identical lanes
// of a vector load.  The tables are written by the host before any kernel that reads them.
typedef const double __attribute__((address_space(4))) * pl_ctab;
#define TB(tab, k) (((pl_ctab)(tab))[(k) + PL_TOFF])

struct G3 {
    int n[3];                        // nodes of THIS RANK's block along z, x, y (the whole grid on one rank)
    int gn[3];                       // global node counts nz, nx, ny
    int o[3];                        // global index of local node (0, 0, 0)
    long long s[3];                  // element strides of z, x, y
    long long vol;                   // elements per scalar array (incl. one ring of nodes in z and x, padding in y)
    const double* rd[3];             // 1/(c[i+1]-c[i])      per axis, zero padded, GLOBAL index + PL_TOFF
    const double* rD[3];             // 1/(c[i+1]-c[i-1])
};
__host__ __device__ inline long long i3(const G3& g, int i, int j, int k) {
    return (long long)(i + 1) * g.s[0] + (long long)(j + 1) * g.s[1] + (k + P3_PAD);
}

struct Op3 {
    G3 g;
    const double* es; const double* en; const double* rho;
    double Kc, Kb, iKc;
    int slave;                       // 1: the reference's slaved outermost in-domain tangential rows (finest level); 0: natural rows
    double grav[3];
    int anchor[3];
};

enum { C3_ZERO = 0, C3_INT = 1, C3_SLAVE = 2 };

// class of the row of velocity component D at node idx; slaves get the offset to their (interior) master
template <int D> __device__ inline int cls3(const Op3& op, const int* idx, long long& moff) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    const int* n = op.g.gn;
    moff = 0;
    if (idx[D] <= 0 || idx[D] >= n[D] - 1 || idx[E] >= n[E] - 1 || idx[F] >= n[F] - 1) return C3_ZERO;
    if (op.slave) {
        if (idx[E] == 0) moff += op.g.s[E]; else if (idx[E] == n[E] - 2) moff -= op.g.s[E];
        if (idx[F] == 0) moff += op.g.s[F]; else if (idx[F] == n[F] - 2) moff -= op.g.s[F];
        if (moff != 0) return C3_SLAVE;
    }
    return C3_INT;
}

// (A_vv v)_D without the pressure term and the sum `dg` of the four..six own-component coefficients at element c.
// Zero-padded tables make the mirror terms of natural wall rows vanish (rD[0] = rD[n-1] = 0).
template <int D> __device__ inline void row3(const Op3& op, const double* const* __restrict__ v, long long c, const int* idx,
                                              double& Av, double& dg) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    const G3& g = op.g;
    const long long sd = g.s[D], se = g.s[E], sf = g.s[F];
    const double* __restrict__ u = v[D];
    const double* __restrict__ es = op.es;
    const double rd_d = TB(g.rd[D], idx[D]), rd_dm = TB(g.rd[D], idx[D] - 1), rD_d = TB(g.rD[D], idx[D]);
    const double u0 = u[c];
    const double cN = 4.0 * op.en[c] * rd_d * rD_d, cS = 4.0 * op.en[c - sd] * rd_dm * rD_d;
    double a = cN * (u[c + sd] - u0) - cS * (u0 - u[c - sd]);
    double d = cN + cS;
    {   // tangential axis E: the edge viscosities are node values averaged along F
        const double rd_e = TB(g.rd[E], idx[E]), rD_e = TB(g.rD[E], idx[E]), rD_ep = TB(g.rD[E], idx[E] + 1);
        const double ep = 0.5 * (es[c + se] + es[c + se + sf]), em = 0.5 * (es[c] + es[c + sf]);
        const double cE = 2.0 * ep * rD_ep * rd_e, cW = 2.0 * em * rD_e * rd_e;
        const double* __restrict__ w = v[E];
        a += cE * (u[c + se] - u0) - cW * (u0 - u[c - se]) + (2.0 * ep * rD_d * rd_e) * (w[c + se] - w[c + se - sd]) -
             (2.0 * em * rD_d * rd_e) * (w[c] - w[c - sd]);
        d += cE + cW;
    }
    {   // tangential axis F: averaged along E
        const double rd_f = TB(g.rd[F], idx[F]), rD_f = TB(g.rD[F], idx[F]), rD_fp = TB(g.rD[F], idx[F] + 1);
        const double ep = 0.5 * (es[c + sf] + es[c + sf + se]), em = 0.5 * (es[c] + es[c + se]);
        const double cE = 2.0 * ep * rD_fp * rd_f, cW = 2.0 * em * rD_f * rd_f;
        const double* __restrict__ w = v[F];
        a += cE * (u[c + sf] - u0) - cW * (u0 - u[c - sf]) + (2.0 * ep * rD_d * rd_f) * (w[c + sf] - w[c + sf - sd]) -
             (2.0 * em * rD_d * rd_f) * (w[c] - w[c - sd]);
        d += cE + cW;
    }
    Av = a; dg = d;
}
// diagonal sum only
template <int D> __device__ inline double diag3(const Op3& op, long long c, const int* idx) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    const G3& g = op.g;
    const long long sd = g.s[D], se = g.s[E], sf = g.s[F];
    const double* __restrict__ es = op.es;
    const double rD_d = TB(g.rD[D], idx[D]);
    double d = 4.0 * op.en[c] * TB(g.rd[D], idx[D]) * rD_d + 4.0 * op.en[c - sd] * TB(g.rd[D], idx[D] - 1) * rD_d;
    const double rd_e = TB(g.rd[E], idx[E]), rd_f = TB(g.rd[F], idx[F]);
    d += (es[c + se] + es[c + se + sf]) * TB(g.rD[E], idx[E] + 1) * rd_e + (es[c] + es[c + sf]) * TB(g.rD[E], idx[E]) * rd_e;
    d += (es[c + sf] + es[c + sf + se]) * TB(g.rD[F], idx[F] + 1) * rd_f + (es[c] + es[c + se]) * TB(g.rD[F], idx[F]) * rd_f;
    return d;
}

// li, lj, lk: node of this rank's block; i, j, k (= idx): its GLOBAL index -- row classes and spacing tables are global
#define K3_PROLOGUE(g)                                                                          \
    const int lk = blockIdx.x * 64 + threadIdx.x, lj = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y), li = blockIdx.z; \
    if (lk >= (g).n[2] || lj >= (g).n[1]) return;                                               \
    const long long c = i3((g), li, lj, lk);                                                    \
    const int k = lk + (g).o[2], j = __builtin_amdgcn_readfirstlane(lj + (g).o[1]), i = li + (g).o[0]; \
    const int idx[3] = {i, j, k};                                                               \
    (void)idx;
static dim3 grid3(const G3& g) { return dim3((g.n[2] + 63) / 64, (g.n[1] + 3) / 4, g.n[0]); }
// (Tried: z-marching -- a workgroup owns a 64 x 4 tile in (y, x) and walks 32 planes so that the planes i-1, i, i+1 it needs
// are the ones it has just touched.  Measured at 257^3 on MI355X: apply 1.44 ms against 0.93 ms with one plane per workgroup
// -- the kernel is bound by the latency of its ~70 dependent 8-byte loads per node, not by HBM traffic, and marching cuts the
// parallelism that hides it.  Removed.)

// ---- full Stokes operator --------------------------------------------------------------------------------------
struct V4 { const double* p[4]; };
struct W4 { double* p[4]; };
struct V3 { const double* p[3]; };
struct W3 { double* p[3]; };

// pressure row class: 0 ghost/anchor (Kc P), 1 continuity, 2 symmetry with the neighbour at offset moff (Kb (P_nb - P))
__device__ inline int cls3_p(const Op3& op, const int* idx, long long& moff) {
    const int* n = op.g.gn;
    moff = 0;
    if (idx[0] >= n[0] - 1 || idx[1] >= n[1] - 1 || idx[2] >= n[2] - 1) return 0;
    if (idx[0] == op.anchor[0] && idx[1] == op.anchor[1] && idx[2] == op.anchor[2]) return 0;
    // with natural wall rows every pressure cell appears in a momentum row: no symmetry rows
    if (!op.slave) return 1;
    const bool bz = idx[0] == 0 || idx[0] == n[0] - 2, bx = idx[1] == 0 || idx[1] == n[1] - 2, by = idx[2] == 0 || idx[2] == n[2] - 2;
    if (bz && bx) { moff = idx[1] == 0 ? op.g.s[1] : -op.g.s[1]; return 2; }            // the 2-D corner rule (pylamp_stokes.py:358-369)
    if (by && (bz || bx)) { moff = idx[2] == 0 ? op.g.s[2] : -op.g.s[2]; return 2; }     // remaining cube edges: inward along y
    return 1;
}

template <int D> __device__ inline double apply_vel3(const Op3& op, const double* const* v, const double* __restrict__ P, long long c,
                                                     const int* idx, bool scaled) {
    long long moff;
    const int cl = cls3<D>(op, idx, moff);
    const double u0 = v[D][c];
    if (cl == C3_ZERO) return scaled ? u0 : op.Kc * u0;
    if (cl == C3_SLAVE) {
        // the matrix row couples to the neighbour along the FIRST boundary axis (E before F), pylamp_stokes.py:170-175
        constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
        const int* n = op.g.gn;
        long long nb;
        if (idx[E] == 0) nb = op.g.s[E]; else if (idx[E] == n[E] - 2) nb = -op.g.s[E];
        else nb = (idx[F] == 0) ? op.g.s[F] : -op.g.s[F];
        const double y = u0 - v[D][c + nb];
        return scaled ? y : op.Kc * y;
    }
    double Av, dg;
    row3<D>(op, v, c, idx, Av, dg);
    Av -= 2.0 * op.Kc * TB(op.g.rD[D], idx[D]) * (P[c] - P[c - op.g.s[D]]);
    return scaled ? Av * pl_rcp(dg) : Av;
}

// rim_only != 0: only the nodes with a wall / slaved / symmetry row are evaluated (all four rows of such a node); the interior rows are
// k3_apply_m's
template <bool SCALED> __global__ __launch_bounds__(256) void k3_apply(Op3 op, V4 x, W4 y, int rim_only) {
    K3_PROLOGUE(op.g)
    if (rim_only) {
        long long m_;
        if (cls3<0>(op, idx, m_) == C3_INT && cls3<1>(op, idx, m_) == C3_INT && cls3<2>(op, idx, m_) == C3_INT && cls3_p(op, idx, m_) == 1) return;
    }
    const double* v[3] = {x.p[0], x.p[1], x.p[2]};
    const double* __restrict__ P = x.p[3];
    y.p[0][c] = apply_vel3<0>(op, v, P, c, idx, SCALED);
    y.p[1][c] = apply_vel3<1>(op, v, P, c, idx, SCALED);
    y.p[2][c] = apply_vel3<2>(op, v, P, c, idx, SCALED);
    long long moff;
    const int cl = cls3_p(op, idx, moff);
    double yp;
    if (cl == 0) yp = SCALED ? P[c] : op.Kc * P[c];
    else if (cl == 2) yp = SCALED ? (P[c + moff] - P[c]) : op.Kb * (P[c + moff] - P[c]);
    else {
        const G3& g = op.g;
        const double rz = TB(g.rd[0], i), rx = TB(g.rd[1], j), ry = TB(g.rd[2], k);
        const double div = (v[0][c + g.s[0]] - v[0][c]) * rz + (v[1][c + g.s[1]] - v[1][c]) * rx + (v[2][c + g.s[2]] - v[2][c]) * ry;
        yp = SCALED ? div * pl_rcp(rz + rx + ry) : op.Kc * div;
    }
    y.p[3][c] = yp;
}

// rhs (pylamp_stokes.py:429,490 extended: density averaged onto the face), optionally row-scaled like k3_apply<true>
template <int D> __device__ inline double rhs_vel3(const Op3& op, long long c, const int* idx, bool scaled) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    long long moff;
    if (cls3<D>(op, idx, moff) != C3_INT || op.grav[D] == 0.0) return 0.0;
    const double* r = op.rho;
    const long long se = op.g.s[E], sf = op.g.s[F];
    const double b = -0.25 * ((r[c] + r[c + se]) + (r[c + sf] + r[c + se + sf])) * op.grav[D];
    return scaled ? b / diag3<D>(op, c, idx) : b;
}
__global__ __launch_bounds__(256) void k3_rhs(Op3 op, W4 b, int scaled) {
    K3_PROLOGUE(op.g)
    b.p[0][c] = rhs_vel3<0>(op, c, idx, scaled);
    b.p[1][c] = rhs_vel3<1>(op, c, idx, scaled);
    b.p[2][c] = rhs_vel3<2>(op, c, idx, scaled);
    b.p[3][c] = 0.0;
}
// b *= D_r (a user-supplied right-hand side)
__global__ __launch_bounds__(256) void k3_scale_rows(Op3 op, W4 b) {
    K3_PROLOGUE(op.g)
    long long moff;
    b.p[0][c] *= (cls3<0>(op, idx, moff) == C3_INT) ? 1.0 / diag3<0>(op, c, idx) : op.iKc;
    b.p[1][c] *= (cls3<1>(op, idx, moff) == C3_INT) ? 1.0 / diag3<1>(op, c, idx) : op.iKc;
    b.p[2][c] *= (cls3<2>(op, idx, moff) == C3_INT) ? 1.0 / diag3<2>(op, c, idx) : op.iKc;
    const int cl = cls3_p(op, idx, moff);
    const G3& g = op.g;
    b.p[3][c] *= cl == 0 ? op.iKc : (cl == 2 ? 1.0 / op.Kb : op.iKc / (TB(g.rd[0], i) + TB(g.rd[1], j) + TB(g.rd[2], k)));
}

// ---- preconditioner pieces ---------------------------------------------------------------------------------------
// z_p = S^-1 r_p from the SCALED residual (continuity rows: r eta_n / Kc^2 -> rs (sum rd) eta_n / Kc; ghost / anchor: rs;
// symmetry rows: P = P_nb - rs)
__device__ inline double prec_p3(const Op3& op, const double* __restrict__ rp, long long c, const int* idx) {
    long long moff;
    const int cl = cls3_p(op, idx, moff);
    const G3& g = op.g;
    if (cl == 0) return rp[c];
    if (cl == 1) return rp[c] * (TB(g.rd[0], idx[0]) + TB(g.rd[1], idx[1]) + TB(g.rd[2], idx[2])) * op.en[c] * op.iKc;
    // symmetry: value of the neighbour (one or two steps of the chain end on a continuity cell) minus the own residual
    int id2[3] = {idx[0], idx[1], idx[2]};
    long long c2 = c; double acc = 0.0;
    for (int hop = 0; hop < 3; hop++) {
        long long m2;
        const int cl2 = cls3_p(op, id2, m2);
        if (cl2 != 2) break;
        acc -= rp[c2];
        const int ax = (m2 == op.g.s[1] || m2 == -op.g.s[1]) ? 1 : 2;
        id2[ax] += (m2 > 0) ? 1 : -1; c2 += m2;
    }
    long long m3;
    const int cl3 = cls3_p(op, id2, m3);
    const double base = cl3 == 1 ? rp[c2] * (TB(g.rd[0], id2[0]) + TB(g.rd[1], id2[1]) + TB(g.rd[2], id2[2])) * op.en[c2] * op.iKc : rp[c2];
    return base + acc;
}
template <int D> __device__ inline double stage1_vel3(const Op3& op, const double* __restrict__ rs, const double* __restrict__ rp, long long c,
                                                      const int* idx, double zp_c) {
    long long moff;
    if (cls3<D>(op, idx, moff) != C3_INT) return 0.0;
    int idm[3] = {idx[0], idx[1], idx[2]}; idm[D] -= 1;
    return rs[c] * diag3<D>(op, c, idx) + 2.0 * op.Kc * TB(op.g.rD[D], idx[D]) * (zp_c - prec_p3(op, rp, c - op.g.s[D], idm));
}
// z_p = S^-1 r_p and f = r_v - A_vp z_p on the interior momentum rows (unscaled), 0 elsewhere
__global__ __launch_bounds__(256) void k3_stage1(Op3 op, V4 rs, double* __restrict__ zp, W3 f) {
    K3_PROLOGUE(op.g)
    const double z0 = prec_p3(op, rs.p[3], c, idx);
    zp[c] = z0;
    f.p[0][c] = stage1_vel3<0>(op, rs.p[0], rs.p[3], c, idx, z0);
    f.p[1][c] = stage1_vel3<1>(op, rs.p[1], rs.p[3], c, idx, z0);
    f.p[2][c] = stage1_vel3<2>(op, rs.p[2], rs.p[3], c, idx, z0);
}

// one Chebyshev-Jacobi sweep  v_next = v + c1 (v - v_prev) + c2 D^-1 (f - A v)  with the constraint rows closed in the same
// pass (a slave evaluates its master's update).  zero != 0: v = 0 is implied and not read.
template <int D> __device__ inline double cheb3(const Op3& op, const double* const* v, const double* __restrict__ vprev,
                                                const double* __restrict__ f, double c1, double c2, long long c, const int* idx, int zero) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    long long moff;
    const int cl = cls3<D>(op, idx, moff);
    if (cl == C3_ZERO) return 0.0;
    const long long cm = c + moff;
    int im[3] = {idx[0], idx[1], idx[2]};
    if (cl == C3_SLAVE) {
        if (idx[E] == 0) im[E] = 1; else if (idx[E] == op.g.gn[E] - 2) im[E] = op.g.gn[E] - 3;
        if (idx[F] == 0) im[F] = 1; else if (idx[F] == op.g.gn[F] - 2) im[F] = op.g.gn[F] - 3;
    }
    if (zero) return (-c2 * f[cm]) * pl_rcp(diag3<D>(op, cm, im));
    double Av, dg;
    row3<D>(op, v, cm, im, Av, dg);
    const double v0 = v[D][cm];
    const double mom = (c1 != 0.0) ? c1 * (v0 - (vprev ? vprev[cm] : 0.0)) : 0.0;
    return v0 + mom + (c2 * (Av - f[cm])) * pl_rcp(dg);                                  // D = -dg
}
// rim_only != 0: only the nodes with a wall or slaved row (all three rows of such a node); the interior rows are k3_sweep_m's
__device__ inline bool k3_all_interior(const Op3& op, const int* idx) {
    long long m_;
    return cls3<0>(op, idx, m_) == C3_INT && cls3<1>(op, idx, m_) == C3_INT && cls3<2>(op, idx, m_) == C3_INT;
}
__global__ __launch_bounds__(256) void k3_cheb(Op3 op, V3 vcur, V3 vprev, V3 f, W3 vnext, double c1, double c2, int zero, int rim_only) {
    K3_PROLOGUE(op.g)
    if (rim_only && k3_all_interior(op, idx)) return;
    const double* v[3] = {vcur.p[0], vcur.p[1], vcur.p[2]};
    vnext.p[0][c] = cheb3<0>(op, v, vprev.p[0], f.p[0], c1, c2, c, idx, zero);
    vnext.p[1][c] = cheb3<1>(op, v, vprev.p[1], f.p[1], c1, c2, c, idx, zero);
    vnext.p[2][c] = cheb3<2>(op, v, vprev.p[2], f.p[2], c1, c2, c, idx, zero);
}
// 1 / diag of the three velocity rows (interior rows; 0 elsewhere), once per set of coefficients: the first sweep of every smoothing
// sequence starts from the zero guess, v1 = -c2 f / diag, and used to rebuild the diagonal from ~30 viscosity loads per node
__global__ __launch_bounds__(256) void k3_dinv(Op3 op, W3 dinv) {
    K3_PROLOGUE(op.g)
    long long moff;
    dinv.p[0][c] = cls3<0>(op, idx, moff) == C3_INT ? pl_rcp(diag3<0>(op, c, idx)) : 0.0;
    dinv.p[1][c] = cls3<1>(op, idx, moff) == C3_INT ? pl_rcp(diag3<1>(op, c, idx)) : 0.0;
    dinv.p[2][c] = cls3<2>(op, idx, moff) == C3_INT ? pl_rcp(diag3<2>(op, c, idx)) : 0.0;
}
// v1 = -c2 f / diag at the row's master (a slave evaluates its master's update: the same numbers as k3_cheb with zero != 0)
__global__ __launch_bounds__(256) void k3_cheb0(Op3 op, V3 f, V3 dinv, W3 vnext, double c2) {
    K3_PROLOGUE(op.g)
    long long moff;
    int cl = cls3<0>(op, idx, moff);
    vnext.p[0][c] = cl == C3_ZERO ? 0.0 : (-c2 * f.p[0][c + moff]) * dinv.p[0][c + moff];
    cl = cls3<1>(op, idx, moff);
    vnext.p[1][c] = cl == C3_ZERO ? 0.0 : (-c2 * f.p[1][c + moff]) * dinv.p[1][c + moff];
    cl = cls3<2>(op, idx, moff);
    vnext.p[2][c] = cl == C3_ZERO ? 0.0 : (-c2 * f.p[2][c + moff]) * dinv.p[2][c + moff];
}
// mode 0: r = f - A v on interior rows (0 elsewhere); mode 1: y = D^-1 A v with closure (power iteration)
template <int D> __device__ inline double resid3(const Op3& op, const double* const* v, const double* __restrict__ f, long long c,
                                                 const int* idx, int mode) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    long long moff;
    const int cl = cls3<D>(op, idx, moff);
    if (cl == C3_ZERO || (mode == 0 && cl != C3_INT)) return 0.0;
    const long long cm = c + moff;
    int im[3] = {idx[0], idx[1], idx[2]};
    if (cl == C3_SLAVE) {
        if (idx[E] == 0) im[E] = 1; else if (idx[E] == op.g.gn[E] - 2) im[E] = op.g.gn[E] - 3;
        if (idx[F] == 0) im[F] = 1; else if (idx[F] == op.g.gn[F] - 2) im[F] = op.g.gn[F] - 3;
    }
    double Av, dg;
    row3<D>(op, v, cm, im, Av, dg);
    return mode == 0 ? f[c] - Av : Av * pl_rcp(dg);
}
__global__ __launch_bounds__(256) void k3_resid(Op3 op, V3 vv, V3 f, W3 r, int mode, int rim_only) {
    K3_PROLOGUE(op.g)
    if (rim_only && k3_all_interior(op, idx)) return;
    const double* v[3] = {vv.p[0], vv.p[1], vv.p[2]};
    r.p[0][c] = resid3<0>(op, v, f.p[0], c, idx, mode);
    r.p[1][c] = resid3<1>(op, v, f.p[1], c, idx, mode);
    r.p[2][c] = resid3<2>(op, v, f.p[2], c, idx, mode);
}

// ---- LDS-tiled variants of the three heavy stencil kernels (round 4) ------------------------------------------------------------------------
// Counters (tools/pmc_passes.sh, profiles/r04_3d257_pmc.csv) say what binds the per-node kernels above: not HBM (the XCD-banded
// workgroup order cut k3_apply's traffic from 2.46x to 1.07x of the algorithmic bytes and made it 9 % SLOWER), not the branches (a
// branch-free rewrite with 114 registers: 29 % slower), but the NUMBER of 8-byte gather instructions -- 68 ... 129 per node, ~22 cycles
// of the CU's texture path each, whether they hit or not (time = 100 us + 11 us x loads per node over k3_cont_rows / k3_heat_apply /
// k3_apply).  Here a workgroup loads its 4 x 64 tile of the planes i-1, i, i+1 with a one-node rim ONCE, coalesced (23 ... 28 loads per
// thread instead of 88 ... 129), and evaluates the rows from LDS (ds_read: a quarter of the cycles of a gather).  Same thread <-> node
// mapping, same row functions (row3 / diag3 on an Op3 whose strides and viscosity pointers are the window's); the rare slaved rows (their
// master's stencil can leave the window) and symmetry rows take the global-memory path.  PYLAMP_3D_LDS=0 keeps the per-node kernels.
#define K3T_SR 68                        // row pitch of the window: 66 columns (k-1 .. k+64) padded
#define K3T_SP (6 * K3T_SR)              // plane pitch: 6 rows (j-1 .. j+4)
#define K3T_N (3 * K3T_SP)               // one array: 3 planes
// ---- z-marching variant: the workgroup keeps its tile and walks K3M_ZC planes --------------------------------------------------------------
// The window's three planes live in a ring of LDS slots; per step ONE new plane of every array is fetched (2 elements per thread and
// array, requested BEFORE the step's arithmetic and written behind it: the latency is hidden even at two workgroups per CU), i.e.
// ~12 loads per node instead of 28 (tile-at-a-time) or 88 (per-node kernel).  The rows are row3's expressions on (plane slot, in-plane
// index) pairs.
#define K3M_ZC 32
struct K3Ring { int o[3]; };                         // LDS offsets of the planes i-1, i, i+1
template <int D, int dD, int dE, int dF> __device__ inline double k3r_at(const double* __restrict__ A, const K3Ring& r, int q) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    constexpr int dz = (D == 0 ? dD : 0) + (E == 0 ? dE : 0) + (F == 0 ? dF : 0);
    constexpr int dx = (D == 1 ? dD : 0) + (E == 1 ? dE : 0) + (F == 1 ? dF : 0);
    constexpr int dy = (D == 2 ? dD : 0) + (E == 2 ? dE : 0) + (F == 2 ? dF : 0);
    return A[r.o[1 + dz] + q + dx * K3T_SR + dy];
}
// row3 on the ring (same expressions, same order)
// the spacing-table entries a node's rows use, per axis a: 1/(c[i+1]-c[i]), the same one node lower, 1/(c[i+1]-c[i-1]), the same one node
// higher.  The marching kernel loads the x and y entries ONCE per thread and the z entries once per plane (scalar loads share the LDS
// counter: one inside the row evaluation drains the LDS pipeline each time)
struct K3Tab { double rd[3], rdm[3], rD[3], rDp[3]; };
template <int D> __device__ inline void row3r(const K3Tab& tb, const double* const* __restrict__ V, const double* __restrict__ ES, const double* __restrict__ EN,
                                               const K3Ring& r, const K3Ring& rs, const K3Ring& rn, int q, double& Av, double& dg) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    const double* __restrict__ u = V[D];
    const double rd_d = tb.rd[D], rd_dm = tb.rdm[D], rD_d = tb.rD[D];
    const double u0 = k3r_at<D, 0, 0, 0>(u, r, q);
    const double cN = 4.0 * k3r_at<D, 0, 0, 0>(EN, rn, q) * rd_d * rD_d, cS = 4.0 * k3r_at<D, -1, 0, 0>(EN, rn, q) * rd_dm * rD_d;
    double a = cN * (k3r_at<D, 1, 0, 0>(u, r, q) - u0) - cS * (u0 - k3r_at<D, -1, 0, 0>(u, r, q));
    double d = cN + cS;
    {
        const double rd_e = tb.rd[E], rD_e = tb.rD[E], rD_ep = tb.rDp[E];
        const double ep = 0.5 * (k3r_at<D, 0, 1, 0>(ES, rs, q) + k3r_at<D, 0, 1, 1>(ES, rs, q)), em = 0.5 * (k3r_at<D, 0, 0, 0>(ES, rs, q) + k3r_at<D, 0, 0, 1>(ES, rs, q));
        const double cE = 2.0 * ep * rD_ep * rd_e, cW = 2.0 * em * rD_e * rd_e;
        const double* __restrict__ w = V[E];
        a += cE * (k3r_at<D, 0, 1, 0>(u, r, q) - u0) - cW * (u0 - k3r_at<D, 0, -1, 0>(u, r, q)) +
             (2.0 * ep * rD_d * rd_e) * (k3r_at<D, 0, 1, 0>(w, r, q) - k3r_at<D, -1, 1, 0>(w, r, q)) -
             (2.0 * em * rD_d * rd_e) * (k3r_at<D, 0, 0, 0>(w, r, q) - k3r_at<D, -1, 0, 0>(w, r, q));
        d += cE + cW;
    }
    {
        const double rd_f = tb.rd[F], rD_f = tb.rD[F], rD_fp = tb.rDp[F];
        const double ep = 0.5 * (k3r_at<D, 0, 0, 1>(ES, rs, q) + k3r_at<D, 0, 1, 1>(ES, rs, q)), em = 0.5 * (k3r_at<D, 0, 0, 0>(ES, rs, q) + k3r_at<D, 0, 1, 0>(ES, rs, q));
        const double cE = 2.0 * ep * rD_fp * rd_f, cW = 2.0 * em * rD_f * rd_f;
        const double* __restrict__ w = V[F];
        a += cE * (k3r_at<D, 0, 0, 1>(u, r, q) - u0) - cW * (u0 - k3r_at<D, 0, 0, -1>(u, r, q)) +
             (2.0 * ep * rD_d * rd_f) * (k3r_at<D, 0, 0, 1>(w, r, q) - k3r_at<D, -1, 0, 1>(w, r, q)) -
             (2.0 * em * rD_d * rd_f) * (k3r_at<D, 0, 0, 0>(w, r, q) - k3r_at<D, -1, 0, 0>(w, r, q));
        d += cE + cW;
    }
    Av = a; dg = d;
}
// this thread's (up to 2) elements of one plane slab (6 rows x 66 columns): global offset within the plane li = 0, LDS offset, validity
// (addresses beyond the block's ring are clamped into it: every thread loads unconditionally -- a predicated load per array and element
//  was a basic block of its own with a branch, 24 per plane -- and what lands in those window entries is never read: the rows the
//  marching kernels evaluate are strictly inside the domain.  has2: this thread also carries one of the elements 256..395.)
struct K3PSlots { long long go[2]; int lo[2]; bool has2; };
__device__ inline K3PSlots k3m_slots(const G3& g, int lj0, int lk0) {
    K3PSlots q;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    q.has2 = tid + 256 < 396;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int e = min(tid + 256 * u, 395), r = e / 66, c = e % 66;
        const int lj = min(lj0 + r - 1, g.n[1]), lk = min(lk0 + c - 1, g.n[2]);
        q.go[u] = i3(g, 0, lj, lk);
        q.lo[u] = r * K3T_SR + c;
    }
    return q;
}
// strictly inside the domain in the two in-plane axes / in z: all four rows of such a node are interior rows (cls3 == C3_INT, cls3_p == 1
// but for the anchor cell, which k3_rim rewrites behind the marching kernel)
__device__ inline bool k3m_inside(const G3& g, int ax, int l) { const int gi = l + g.o[ax]; return l < g.n[ax] && gi >= 1 && gi <= g.gn[ax] - 3; }
// ---- the rim of the marching kernels: every node with a wall, slaved, ghost or symmetry row lies on one of the nine planes
// index in {0, n - 2, n - 1} of an axis (global indices), plus the pressure anchor.  One thread per rim node (~600 000 of 17 M at 257^3;
// a full-grid pass that only classifies took 169 us per launch): ALL rows of the node by the per-node row functions.
struct RimArgs { Op3 op; V4 x; W4 y; V3 vprev; V3 f; double c1, c2; int what, scaled; };       // what: 0 apply, 1 Chebyshev sweep, 2 residual
__device__ inline long long k3_rim_count(const G3& g) { return 3LL * g.n[1] * g.n[2] + 3LL * g.n[0] * g.n[2] + 3LL * g.n[0] * g.n[1] + 1; }
// (the marching kernels run this in workgroups of their own BEHIND the tiles -- the rim fills the tail of the launch instead of
//  a 35 us launch of its own; the anchor cell, strictly inside the domain, is a pressure row only: the tiles leave it alone)
__device__ __forceinline__ void k3_rim_body(const RimArgs& a, long long t) {
    const G3& g = a.op.g;
    const long long N[3] = {3LL * g.n[1] * g.n[2], 3LL * g.n[0] * g.n[2], 3LL * g.n[0] * g.n[1]};
    bool anchor_only = false;
    int l[3];
    auto rimpos = [&](int ax, int which) { const int gi = which == 0 ? 0 : g.gn[ax] - 3 + which; return gi - g.o[ax]; };      // 0, n-2, n-1 -> local
    auto on_rim = [&](int ax, int li) { const int gi = li + g.o[ax]; return gi == 0 || gi >= g.gn[ax] - 2; };
    if (t < N[0]) {                                         // z slabs: (which, j, k)
        const int w = (int)(t / ((long long)g.n[1] * g.n[2])); const long long r = t % ((long long)g.n[1] * g.n[2]);
        l[0] = rimpos(0, w); l[1] = (int)(r / g.n[2]); l[2] = (int)(r % g.n[2]);
    } else if ((t -= N[0]) < N[1]) {                        // x slabs: (which, i, k), without the z slabs' nodes
        const int w = (int)(t / ((long long)g.n[0] * g.n[2])); const long long r = t % ((long long)g.n[0] * g.n[2]);
        l[1] = rimpos(1, w); l[0] = (int)(r / g.n[2]); l[2] = (int)(r % g.n[2]);
        if (on_rim(0, l[0])) return;
    } else if ((t -= N[1]) < N[2]) {                        // y slabs: (which, i, j), without the others'
        const int w = (int)(t / ((long long)g.n[0] * g.n[1])); const long long r = t % ((long long)g.n[0] * g.n[1]);
        l[2] = rimpos(2, w); l[0] = (int)(r / g.n[1]); l[1] = (int)(r % g.n[1]);
        if (on_rim(0, l[0]) || on_rim(1, l[1])) return;
    } else if (t == N[2] && a.what == 0) {                  // the pressure anchor
        for (int q = 0; q < 3; q++) l[q] = a.op.anchor[q] - g.o[q];
        if (on_rim(0, l[0]) || on_rim(1, l[1]) || on_rim(2, l[2])) return;
        anchor_only = true;
    } else return;
    for (int q = 0; q < 3; q++) if (l[q] < 0 || l[q] >= g.n[q]) return;
    const long long c = i3(g, l[0], l[1], l[2]);
    const int idx[3] = {l[0] + g.o[0], l[1] + g.o[1], l[2] + g.o[2]};
    const int i = idx[0], j = idx[1], k = idx[2];
    const double* v[3] = {a.x.p[0], a.x.p[1], a.x.p[2]};
    if (a.what == 0) {
        const double* __restrict__ P = a.x.p[3];
        const bool SC = a.scaled != 0;
        if (!anchor_only) { a.y.p[0][c] = apply_vel3<0>(a.op, v, P, c, idx, SC); a.y.p[1][c] = apply_vel3<1>(a.op, v, P, c, idx, SC); a.y.p[2][c] = apply_vel3<2>(a.op, v, P, c, idx, SC); }
        long long moff;
        const int cl = cls3_p(a.op, idx, moff);
        double yp;
        if (cl == 0) yp = SC ? P[c] : a.op.Kc * P[c];
        else if (cl == 2) yp = SC ? (P[c + moff] - P[c]) : a.op.Kb * (P[c + moff] - P[c]);
        else {
            const double rz = TB(g.rd[0], i), rx = TB(g.rd[1], j), ry = TB(g.rd[2], k);
            const double div = (v[0][c + g.s[0]] - v[0][c]) * rz + (v[1][c + g.s[1]] - v[1][c]) * rx + (v[2][c + g.s[2]] - v[2][c]) * ry;
            yp = SC ? div * pl_rcp(rz + rx + ry) : a.op.Kc * div;
        }
        a.y.p[3][c] = yp;
    } else if (a.what == 1) {
        a.y.p[0][c] = cheb3<0>(a.op, v, a.vprev.p[0], a.f.p[0], a.c1, a.c2, c, idx, 0);
        a.y.p[1][c] = cheb3<1>(a.op, v, a.vprev.p[1], a.f.p[1], a.c1, a.c2, c, idx, 0);
        a.y.p[2][c] = cheb3<2>(a.op, v, a.vprev.p[2], a.f.p[2], a.c1, a.c2, c, idx, 0);
    } else {
        a.y.p[0][c] = resid3<0>(a.op, v, a.f.p[0], c, idx, 0); a.y.p[1][c] = resid3<1>(a.op, v, a.f.p[1], c, idx, 0); a.y.p[2][c] = resid3<2>(a.op, v, a.f.p[2], c, idx, 0);
    }
}
__global__ __launch_bounds__(256) void k3_rim(RimArgs a) { k3_rim_body(a, (long long)blockIdx.x * 256 + threadIdx.x); }
// Tile of a workgroup.  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with an L2 of its own; with the
// plain (k, j, z-chunk) order the neighbours that share a tile's halo rows and the cache lines of its halo columns sit on 8 different
// L2s and the halo is fetched from the fabric every time (measured: 2.6x the algorithmic bytes).  band != 0: XCD x gets the x-th eighth
// of the tile list, so neighbours meet in one L2.
struct K3Tile { int bx, by, bz; long long rim; };          // rim >= 0: not a tile -- the rim.th workgroup of the rim
__device__ inline K3Tile k3m_tile(const G3& g, int band, int zc) {
    const unsigned nx = (g.n[2] + 63) / 64, ny = (g.n[1] + 3) / 4, nz = (g.n[0] + zc - 1) / zc;
    const unsigned total = band == 2 ? nx * 8 * ((ny * nz + 7) / 8) : nx * ny * nz, id = blockIdx.x;
    if (id >= total) { K3Tile t; t.bx = t.by = t.bz = 0; t.rim = id - total; return t; }
    unsigned lid = id;
    if (band == 2) {       // the nx tiles of a k-row on ONE XCD (they share the cache lines of their halo columns); rows round-robin as before
        const unsigned xcd = id % 8, m = id / 8, R = (m / nx) * 8 + xcd;
        K3Tile t; t.bx = m % nx; t.by = R % ny; t.bz = R / ny; t.rim = R < ny * nz ? -1 : (1LL << 40);      // (padding: a rim index beyond the rim)
        return t;
    }
    if (band == 1) { const unsigned fl = total / 8, rem = total % 8, xcd = id % 8; lid = xcd * fl + min(xcd, rem) + id / 8; }
    K3Tile t; t.bx = lid % nx; t.by = (lid / nx) % ny; t.bz = lid / (nx * ny); t.rim = -1;
    return t;
}
template <bool SCALED>
__global__ __launch_bounds__(256, 3) void k3_apply_m(Op3 op, V4 x, W4 y, int band, int zc) {
    // rings: the velocities need the planes i-1, i, i+1 (3 slots), the shear viscosity i, i+1, the normal viscosity and the pressure
    // i-1, i (2 slots each): 48 KB -> three workgroups per CU
    __shared__ double WV[3][3 * K3T_SP];
    __shared__ double WS[2 * K3T_SP], WN[2 * K3T_SP], WP[2 * K3T_SP];
    const G3& g = op.g;
    const K3Tile tl = k3m_tile(g, band, zc);
    if (tl.rim >= 0) {
        RimArgs ra; ra.op = op; ra.x = x; ra.y = y; ra.what = 0; ra.scaled = SCALED ? 1 : 0; ra.c1 = ra.c2 = 0.0;
        for (int q = 0; q < 3; q++) ra.vprev.p[q] = ra.f.p[q] = nullptr;
        k3_rim_body(ra, tl.rim * 256 + threadIdx.y * 64 + threadIdx.x);
        return;
    }
    const int lj0 = tl.by * 4, lk0 = tl.bx * 64;
    const int z0 = tl.bz * zc, z1 = min(z0 + zc, g.n[0]);
    const K3PSlots ps = k3m_slots(g, lj0, lk0);
    const double* vg[3] = {x.p[0], x.p[1], x.p[2]};
    // element u of plane `plane` of the six arrays (velocities, shear / normal viscosity, pressure)
    const double* src6[6] = {vg[0], vg[1], vg[2], op.es, op.en, x.p[3]};
    auto fetch = [&](int m, int plane, int u) { return src6[m][ps.go[u] + (long long)plane * g.s[0]]; };
    // prologue: plane p of the velocities in slot (p + 3) % 3 ... of the 2-slot rings in slot (p + 2) % 2
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (u == 1 && !ps.has2) break;
        double t0[9], t1[6];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int m = 0; m < 3; m++) t0[3 * a + m] = fetch(m, z0 + a - 1, u);
        t1[0] = fetch(3, z0, u); t1[1] = fetch(3, z0 + 1, u); t1[2] = fetch(4, z0 - 1, u); t1[3] = fetch(4, z0, u);
        t1[4] = fetch(5, z0 - 1, u); t1[5] = fetch(5, z0, u);
        const int lo = ps.lo[u];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int m = 0; m < 3; m++) WV[m][((z0 + a - 1 + 3) % 3) * K3T_SP + lo] = t0[3 * a + m];
        WS[(z0 & 1) * K3T_SP + lo] = t1[0]; WS[((z0 + 1) & 1) * K3T_SP + lo] = t1[1];
        WN[((z0 + 1) & 1) * K3T_SP + lo] = t1[2]; WN[(z0 & 1) * K3T_SP + lo] = t1[3];
        WP[((z0 + 1) & 1) * K3T_SP + lo] = t1[4]; WP[(z0 & 1) * K3T_SP + lo] = t1[5];
    }
    __syncthreads();
    const int lk = lk0 + threadIdx.x, lj = lj0 + threadIdx.y;
    const bool mine_jk = k3m_inside(g, 1, lj) && k3m_inside(g, 2, lk);
    const bool anchor_jk = lj + g.o[1] == op.anchor[1] && lk + g.o[2] == op.anchor[2];
    const int q = (threadIdx.y + 1) * K3T_SR + (threadIdx.x + 1);
    const double* V[3] = {WV[0], WV[1], WV[2]};
    K3Tab tb;
    {
        const int j = min(lj, g.n[1] - 1) + g.o[1], k = min(lk, g.n[2] - 1) + g.o[2];
        tb.rd[1] = TB(g.rd[1], j); tb.rdm[1] = TB(g.rd[1], j - 1); tb.rD[1] = TB(g.rD[1], j); tb.rDp[1] = TB(g.rD[1], j + 1);
        tb.rd[2] = TB(g.rd[2], k); tb.rdm[2] = TB(g.rd[2], k - 1); tb.rD[2] = TB(g.rD[2], k); tb.rDp[2] = TB(g.rD[2], k + 1);
    }
    for (int li = z0; li < z1; li++) {
        {
            const int i = li + g.o[0];
            tb.rd[0] = TB(g.rd[0], i); tb.rdm[0] = TB(g.rd[0], i - 1); tb.rD[0] = TB(g.rD[0], i); tb.rDp[0] = TB(g.rD[0], i + 1);
        }
        // requested now, written behind the arithmetic: the planes li + 2 (velocities, shear viscosity) and li + 1 (normal viscosity, pressure)
        const bool more = li + 1 < z1;
        double tq[6], tq2[6];
        if (more) {
#pragma unroll
            for (int m = 0; m < 6; m++) tq[m] = fetch(m, m < 4 ? li + 2 : li + 1, 0);
            if (ps.has2) {
#pragma unroll
                for (int m = 0; m < 6; m++) tq2[m] = fetch(m, m < 4 ? li + 2 : li + 1, 1);
            }
        }
        K3Ring rv, rs, rn;
        rv.o[0] = ((li + 2) % 3) * K3T_SP; rv.o[1] = (li % 3) * K3T_SP; rv.o[2] = ((li + 1) % 3) * K3T_SP;        // (li - 1 + 3) % 3 = (li + 2) % 3
        rs.o[0] = 0; rs.o[1] = (li & 1) * K3T_SP; rs.o[2] = ((li + 1) & 1) * K3T_SP;
        rn.o[0] = ((li + 1) & 1) * K3T_SP; rn.o[1] = (li & 1) * K3T_SP; rn.o[2] = 0;
        if (mine_jk && k3m_inside(g, 0, li)) {
            // (nodes strictly inside the domain only: the walls, slaved, ghost and symmetry rows -- a rim two nodes thick -- and the
            //  anchor cell are k3_rim's, launched behind)
            const long long c = i3(g, li, lj, lk);
#define K3M_COMP(D)                                                                                                   \
            {                                                                                                         \
                double Av, dg;                                                                                        \
                row3r<D>(tb, V, WS, WN, rv, rs, rn, q, Av, dg);                                                       \
                Av -= 2.0 * op.Kc * tb.rD[D] * (k3r_at<D, 0, 0, 0>(WP, rn, q) - k3r_at<D, -1, 0, 0>(WP, rn, q));        \
                y.p[D][c] = SCALED ? Av * pl_rcp(dg) : Av;                                                            \
            }
            K3M_COMP(0) K3M_COMP(1) K3M_COMP(2)
#undef K3M_COMP
            const double rz = tb.rd[0], rx = tb.rd[1], ry = tb.rd[2];
            const double div = (WV[0][rv.o[2] + q] - WV[0][rv.o[1] + q]) * rz + (WV[1][rv.o[1] + q + K3T_SR] - WV[1][rv.o[1] + q]) * rx +
                               (WV[2][rv.o[1] + q + 1] - WV[2][rv.o[1] + q]) * ry;
            if (!(anchor_jk && li + g.o[0] == op.anchor[0])) y.p[3][c] = SCALED ? div * pl_rcp(rz + rx + ry) : op.Kc * div;
        }
        __syncthreads();                              // everybody has read the oldest planes
        if (more) {
            const int sl[6] = {((li + 2) % 3) * K3T_SP, ((li + 2) % 3) * K3T_SP, ((li + 2) % 3) * K3T_SP, (li & 1) * K3T_SP, ((li + 1) & 1) * K3T_SP, ((li + 1) & 1) * K3T_SP};
            double* const dst6[6] = {WV[0], WV[1], WV[2], WS, WN, WP};
#pragma unroll
            for (int m = 0; m < 6; m++) dst6[m][sl[m] + ps.lo[0]] = tq[m];
            if (ps.has2) {
#pragma unroll
                for (int m = 0; m < 6; m++) dst6[m][sl[m] + ps.lo[1]] = tq2[m];
            }
        }
        __syncthreads();
    }
}
// the smoother / residual on the same rings (no pressure): MODE 0 one Chebyshev sweep from a non-zero iterate, 1 residual f - A v
template <int MODE>
__global__ __launch_bounds__(256, 3) void k3_sweep_m(Op3 op, V3 vcur, V3 vprev, V3 f, W3 out, double c1, double c2, int band, int zc) {
    __shared__ double WV[3][3 * K3T_SP];
    __shared__ double WS[2 * K3T_SP], WN[2 * K3T_SP];
    const G3& g = op.g;
    const K3Tile tl = k3m_tile(g, band, zc);
    if (tl.rim >= 0) {
        RimArgs ra; ra.op = op; ra.what = MODE == 0 ? 1 : 2; ra.scaled = 0; ra.c1 = c1; ra.c2 = c2; ra.vprev = vprev; ra.f = f;
        for (int q = 0; q < 3; q++) { ra.x.p[q] = vcur.p[q]; ra.y.p[q] = out.p[q]; }
        ra.x.p[3] = nullptr; ra.y.p[3] = nullptr;
        k3_rim_body(ra, tl.rim * 256 + threadIdx.y * 64 + threadIdx.x);
        return;
    }
    const int lj0 = tl.by * 4, lk0 = tl.bx * 64;
    const int z0 = tl.bz * zc, z1 = min(z0 + zc, g.n[0]);
    const K3PSlots ps = k3m_slots(g, lj0, lk0);
    const double* vg[3] = {vcur.p[0], vcur.p[1], vcur.p[2]};
    const double* src5[5] = {vg[0], vg[1], vg[2], op.es, op.en};
    auto fetch = [&](int m, int plane, int u) { return src5[m][ps.go[u] + (long long)plane * g.s[0]]; };
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (u == 1 && !ps.has2) break;
        double t0[9], t1[4];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int m = 0; m < 3; m++) t0[3 * a + m] = fetch(m, z0 + a - 1, u);
        t1[0] = fetch(3, z0, u); t1[1] = fetch(3, z0 + 1, u); t1[2] = fetch(4, z0 - 1, u); t1[3] = fetch(4, z0, u);
        const int lo = ps.lo[u];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int m = 0; m < 3; m++) WV[m][((z0 + a - 1 + 3) % 3) * K3T_SP + lo] = t0[3 * a + m];
        WS[(z0 & 1) * K3T_SP + lo] = t1[0]; WS[((z0 + 1) & 1) * K3T_SP + lo] = t1[1];
        WN[((z0 + 1) & 1) * K3T_SP + lo] = t1[2]; WN[(z0 & 1) * K3T_SP + lo] = t1[3];
    }
    __syncthreads();
    const int lk = lk0 + threadIdx.x, lj = lj0 + threadIdx.y;
    const bool mine_jk = k3m_inside(g, 1, lj) && k3m_inside(g, 2, lk);
    const int q = (threadIdx.y + 1) * K3T_SR + (threadIdx.x + 1);
    const double* V[3] = {WV[0], WV[1], WV[2]};
    K3Tab tb;
    {
        const int j = min(lj, g.n[1] - 1) + g.o[1], k = min(lk, g.n[2] - 1) + g.o[2];
        tb.rd[1] = TB(g.rd[1], j); tb.rdm[1] = TB(g.rd[1], j - 1); tb.rD[1] = TB(g.rD[1], j); tb.rDp[1] = TB(g.rD[1], j + 1);
        tb.rd[2] = TB(g.rd[2], k); tb.rdm[2] = TB(g.rd[2], k - 1); tb.rD[2] = TB(g.rD[2], k); tb.rDp[2] = TB(g.rD[2], k + 1);
    }
    for (int li = z0; li < z1; li++) {
        {
            const int i = li + g.o[0];
            tb.rd[0] = TB(g.rd[0], i); tb.rdm[0] = TB(g.rd[0], i - 1); tb.rD[0] = TB(g.rD[0], i); tb.rDp[0] = TB(g.rD[0], i + 1);
        }
        const bool more = li + 1 < z1;
        double tq[5], tq2[5];
        if (more) {
#pragma unroll
            for (int m = 0; m < 5; m++) tq[m] = fetch(m, m < 4 ? li + 2 : li + 1, 0);
            if (ps.has2) {
#pragma unroll
                for (int m = 0; m < 5; m++) tq2[m] = fetch(m, m < 4 ? li + 2 : li + 1, 1);
            }
        }
        K3Ring rv, rs, rn;
        rv.o[0] = ((li + 2) % 3) * K3T_SP; rv.o[1] = (li % 3) * K3T_SP; rv.o[2] = ((li + 1) % 3) * K3T_SP;
        rs.o[0] = 0; rs.o[1] = (li & 1) * K3T_SP; rs.o[2] = ((li + 1) & 1) * K3T_SP;
        rn.o[0] = ((li + 1) & 1) * K3T_SP; rn.o[1] = (li & 1) * K3T_SP; rn.o[2] = 0;
        if (mine_jk && k3m_inside(g, 0, li)) {
            const long long c = i3(g, li, lj, lk);
#define K3M_COMP(D)                                                                                                   \
            {                                                                                                         \
                double Av, dg;                                                                                        \
                row3r<D>(tb, V, WS, WN, rv, rs, rn, q, Av, dg);                                                       \
                if (MODE == 0) {                                                                                      \
                    const double v0 = WV[D][rv.o[1] + q];                                                             \
                    const double mom = (c1 != 0.0) ? c1 * (v0 - (vprev.p[D] ? vprev.p[D][c] : 0.0)) : 0.0;            \
                    out.p[D][c] = v0 + mom + (c2 * (Av - f.p[D][c])) * pl_rcp(dg);                                    \
                } else out.p[D][c] = f.p[D][c] - Av;                                                                  \
            }
            K3M_COMP(0) K3M_COMP(1) K3M_COMP(2)
#undef K3M_COMP
        }
        __syncthreads();
        if (more) {
            const int sl[5] = {((li + 2) % 3) * K3T_SP, ((li + 2) % 3) * K3T_SP, ((li + 2) % 3) * K3T_SP, (li & 1) * K3T_SP, ((li + 1) & 1) * K3T_SP};
            double* const dst5[5] = {WV[0], WV[1], WV[2], WS, WN};
#pragma unroll
            for (int m = 0; m < 5; m++) dst5[m][sl[m] + ps.lo[0]] = tq[m];
            if (ps.has2) {
#pragma unroll
                for (int m = 0; m < 5; m++) dst5[m][sl[m] + ps.lo[1]] = tq2[m];
            }
        }
        __syncthreads();
    }
}
// (measured at 257^3, profiles/r04_3d257_banded_pmc.csv: banding cuts k3_apply_m's fetch from 2091 to 1218 MB per launch and its time
//  goes UP from 455 to 517 us -- the kernel is not bound by fabric traffic; off unless PYLAMP_3D_BAND=1)
// PYLAMP_3D_BAND = 2 (the default): only the tiles of one k-row share an XCD -- they share the cache lines of their halo COLUMNS, the rows
// stay dealt round-robin: k3_apply_m 0.425 -> 0.410 ms (0.41 of the HBM roof on the algorithmic bytes); 0: plain order
static int k3_band() { static const int on = getenv("PYLAMP_3D_BAND") ? atoi(getenv("PYLAMP_3D_BAND")) : 2; return on; }
// planes a workgroup walks: K3M_ZC, shorter on the smaller levels until the launch has ~2 workgroups per slot of the chip (3 per CU)
static int k3m_zc(const G3& g) {
    const long long tiles = (long long)((g.n[2] + 63) / 64) * ((g.n[1] + 3) / 4);
    int zc = K3M_ZC;
    while (zc > 8 && tiles * ((g.n[0] + zc - 1) / zc) < 1536) zc >>= 1;
    return zc;
}
static dim3 grid3m(const G3& g) {          // the tiles, then the rim (one thread per rim node)
    const int zc = k3m_zc(g);
    const long long rim = 3LL * g.n[1] * g.n[2] + 3LL * g.n[0] * g.n[2] + 3LL * g.n[0] * g.n[1] + 1;
    const long long nx = (g.n[2] + 63) / 64, rows = (long long)((g.n[1] + 3) / 4) * ((g.n[0] + zc - 1) / zc);
    return dim3((unsigned)((k3_band() == 2 ? nx * 8 * ((rows + 7) / 8) : nx * rows) + (rim + 255) / 256));
}
static bool k3_use_lds(const G3& g) {
    static const bool on = !(getenv("PYLAMP_3D_LDS") && atoi(getenv("PYLAMP_3D_LDS")) == 0);
    return on && (long long)g.n[0] * g.n[1] * g.n[2] >= 200000;        // the small (latency-bound) levels keep the per-node kernels
}

// full-weighting restriction: vertex-centred [1/4 1/2 1/4] along the component's own axis, cell-centred [1/8 3/8 3/8 1/8]
// along the other two (uniform-grid weights)
template <int D> __device__ inline double restrict3(const G3& gf, const Op3& opc, const double* __restrict__ rf, const int* idx) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    long long moff;
    if (cls3<D>(opc, idx, moff) != C3_INT) return 0.0;
    const long long b = i3(gf, 2 * idx[0] - gf.o[0], 2 * idx[1] - gf.o[1], 2 * idx[2] - gf.o[2]);
    const double wv[3] = {0.25, 0.5, 0.25}, wc[4] = {0.125, 0.375, 0.375, 0.125};
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double s1 = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            double s2 = 0.0;
#pragma unroll
            for (int p = 0; p < 4; p++) s2 += wc[p] * rf[b + (a - 1) * gf.s[D] + (q - 1) * gf.s[E] + (p - 1) * gf.s[F]];
            s1 += wc[q] * s2;
        }
        acc += wv[a] * s1;
    }
    return acc;
}
__global__ __launch_bounds__(256) void k3_restrict(G3 gf, Op3 opc, V3 rf, W3 fc) {
    K3_PROLOGUE(opc.g)
    fc.p[0][c] = restrict3<0>(gf, opc, rf.p[0], idx);
    fc.p[1][c] = restrict3<1>(gf, opc, rf.p[1], idx);
    fc.p[2][c] = restrict3<2>(gf, opc, rf.p[2], idx);
}
// (P e)_D at a fine node: linear along the own axis, 3/4-1/4 (clamped) along the other two
template <int D> __device__ inline double prolong3_at(const G3& gc, const double* __restrict__ e, const int* idx) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    const int d0 = idx[D] >> 1, d1 = (idx[D] + 1) >> 1;
    int en_ = idx[E] >> 1, eo = (idx[E] & 1) ? en_ + 1 : en_ - 1;
    int fn = idx[F] >> 1, fo = (idx[F] & 1) ? fn + 1 : fn - 1;
    const int emax = gc.gn[E] - 2, fmax = gc.gn[F] - 2;
    en_ = min(max(en_, 0), emax); eo = min(max(eo, 0), emax); fn = min(max(fn, 0), fmax); fo = min(max(fo, 0), fmax);
    auto at = [&](int a, int b_, int c_) {
        int id[3]; id[D] = a; id[E] = b_; id[F] = c_;
        return e[i3(gc, id[0] - gc.o[0], id[1] - gc.o[1], id[2] - gc.o[2])];
    };
    auto lin = [&](int b_, int c_) { return 0.5 * (at(d0, b_, c_) + at(d1, b_, c_)); };
    return 0.75 * (0.75 * lin(en_, fn) + 0.25 * lin(en_, fo)) + 0.25 * (0.75 * lin(eo, fn) + 0.25 * lin(eo, fo));
}
template <int D> __device__ inline double prolong3(const Op3& opf, const G3& gc, const double* __restrict__ e, const double* __restrict__ vin,
                                                   long long c, const int* idx) {
    constexpr int E = (D + 1) % 3, F = (D + 2) % 3;
    long long moff;
    const int cl = cls3<D>(opf, idx, moff);
    if (cl == C3_ZERO) return 0.0;
    int im[3] = {idx[0], idx[1], idx[2]};
    if (cl == C3_SLAVE) {
        if (idx[E] == 0) im[E] = 1; else if (idx[E] == opf.g.gn[E] - 2) im[E] = opf.g.gn[E] - 3;
        if (idx[F] == 0) im[F] = 1; else if (idx[F] == opf.g.gn[F] - 2) im[F] = opf.g.gn[F] - 3;
    }
    return vin[c + moff] + prolong3_at<D>(gc, e, im);
}
__global__ __launch_bounds__(256) void k3_prolong_add(Op3 opf, G3 gc, V3 ec, V3 vin, W3 vout) {
    K3_PROLOGUE(opf.g)
    vout.p[0][c] = prolong3<0>(opf, gc, ec.p[0], vin.p[0], c, idx);
    vout.p[1][c] = prolong3<1>(opf, gc, ec.p[1], vin.p[1], c, idx);
    vout.p[2][c] = prolong3<2>(opf, gc, ec.p[2], vin.p[2], c, idx);
}
// arithmetic viscosity coarsening: nodes by [1 2 1]^3 / 64 (edge-clamped), centres by the mean of the 8 covered fine cells
__global__ __launch_bounds__(256) void k3_coarsen(G3 gf, const double* __restrict__ esf, const double* __restrict__ enf, G3 gc,
                                                  double* __restrict__ esc, double* __restrict__ enc) {
    K3_PROLOGUE(gc)
    double acc = 0.0;
    for (int a = -1; a <= 1; a++)
        for (int q = -1; q <= 1; q++)
            for (int p = -1; p <= 1; p++) {
                const int fi = min(max(2 * i + a, 0), gf.gn[0] - 1), fj = min(max(2 * j + q, 0), gf.gn[1] - 1), fk = min(max(2 * k + p, 0), gf.gn[2] - 1);
                acc += (a == 0 ? 2.0 : 1.0) * (q == 0 ? 2.0 : 1.0) * (p == 0 ? 2.0 : 1.0) * esf[i3(gf, fi - gf.o[0], fj - gf.o[1], fk - gf.o[2])];
            }
    esc[c] = acc * (1.0 / 64.0);
    const int ci = min(i, gc.gn[0] - 2), cj = min(j, gc.gn[1] - 2), ck = min(k, gc.gn[2] - 2);      // ghost entries copy their neighbour
    const long long b = i3(gf, 2 * ci - gf.o[0], 2 * cj - gf.o[1], 2 * ck - gf.o[2]);
    double s = 0.0;
    for (int a = 0; a < 2; a++) for (int q = 0; q < 2; q++) for (int p = 0; p < 2; p++) s += enf[b + a * gf.s[0] + q * gf.s[1] + p * gf.s[2]];
    enc[c] = 0.125 * s;
}
// make x satisfy the constraint rows of A x = b exactly (bs = scaled b): walls / ghosts x = bs, slaves x = x_master + bs
template <int D> __device__ inline void close3(const Op3& op, double* __restrict__ x, const double* __restrict__ bs, long long c, const int* idx) {
    long long moff;
    const int cl = cls3<D>(op, idx, moff);
    if (cl == C3_ZERO) x[c] = bs[c];
    else if (cl == C3_SLAVE) x[c] = x[c + moff] + bs[c];
}
__global__ __launch_bounds__(256) void k3_close(Op3 op, W3 x, V3 bs) {
    K3_PROLOGUE(op.g)
    close3<0>(op, x.p[0], bs.p[0], c, idx);
    close3<1>(op, x.p[1], bs.p[1], c, idx);
    close3<2>(op, x.p[2], bs.p[2], c, idx);
}

// ---- flat vector kernels (the ring / padding entries of every vector are zero and stay zero) ----------------------
__global__ void k3_axpby(long long n, double* __restrict__ y, double a, const double* __restrict__ x, double b, const double* __restrict__ z) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) y[t] = a * x[t] + b * z[t];
}
__global__ void k3_p_update(long long n, double* __restrict__ p, const double* __restrict__ r, const double* __restrict__ v, double beta, double omega) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) p[t] = r[t] + beta * (p[t] - omega * v[t]);
}
__global__ void k3_xr_update(long long n, double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z, double* __restrict__ r,
                             const double* __restrict__ s, const double* __restrict__ t_, double alpha, double omega) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        x[t] += alpha * y[t] + omega * z[t]; r[t] = s[t] - omega * t_[t];
    }
}
#define D3_BLOCKS 1024
// up to five dot products in one pass: part[5 b + q] = sum a_q[t] b_q[t]
struct Dot5 { const double* a[5]; const double* b[5]; int n; };
__global__ __launch_bounds__(256) void k3_dots(long long n, Dot5 d, double* __restrict__ part) {
    double acc[5] = {0, 0, 0, 0, 0};
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256)
        for (int q = 0; q < d.n; q++) acc[q] += d.a[q][t] * d.b[q][t];
    __shared__ double sh[5][4];
    for (int q = 0; q < 5; q++) {
        double a = acc[q];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = a;
    }
    __syncthreads();
    if (threadIdx.x < 5) part[5 * blockIdx.x + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}
// The second reduction point of BiCGStab with everything the velocity-error estimate needs, in ONE pass (one rank; vectors of 4 consecutive
// arrays of `vol` elements, the pressure array last): part[11 b + ..] = t.s, t.t, r~.s, r~.t, s.s | the same three of the pressure array
// (t.s, t.t, s.s: the continuity part of |s - omega t|^2 follows) | |(x + dx)_vel|^2 | yw.s_p, yw.t_p (the anchor-mode term).
// Until round 4 the estimate cost three more passes and three more host synchronisations per late iteration.
struct Fuse11 { const double* t; const double* s; const double* rt; const double* x; const double* dx; const double* yw; long long vol; };
__global__ __launch_bounds__(256) void k3_dots11(Fuse11 a, double* __restrict__ part) {
    double acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long n = 4 * a.vol, np = 3 * a.vol;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const double T = a.t[e], S = a.s[e], R = a.rt[e];
        acc[0] += T * S; acc[1] += T * T; acc[2] += R * S; acc[3] += R * T; acc[4] += S * S;
        if (e >= np) {
            acc[5] += T * S; acc[6] += T * T; acc[7] += S * S;
            if (a.yw) { const double Y = a.yw[e - np]; acc[9] += Y * S; acc[10] += Y * T; }
        } else if (a.x) { const double X = a.x[e] + (a.dx ? a.dx[e] : 0.0); acc[8] += X * X; }
    }
    __shared__ double sh[11][4];
    for (int q = 0; q < 11; q++) {
        double v = acc[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 11) part[11 * blockIdx.x + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}
// x += alpha y + omega z, r = s - omega t, and the NEXT direction p = r + beta (p - omega v) in the same pass
__global__ void k3_xrp_update(long long n, double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z, double* __restrict__ r,
                              const double* __restrict__ s, const double* __restrict__ t_, double* __restrict__ p, const double* __restrict__ v,
                              double alpha, double omega, double beta) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        x[t] += alpha * y[t] + omega * z[t];
        const double rn = s[t] - omega * t_[t];
        r[t] = rn;
        p[t] = rn + beta * (p[t] - omega * v[t]);
    }
}
__global__ __launch_bounds__(256) void k3_random(G3 g, double* __restrict__ v, unsigned seed) {
    K3_PROLOGUE(g)
    unsigned h = (unsigned)(((unsigned)i * 73856093u) ^ ((unsigned)j * 19349663u) ^ ((unsigned)k * 83492791u)) ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    v[c] = (i < g.gn[0] - 1 && j < g.gn[1] - 1 && k < g.gn[2] - 1) ? (double)h * (2.0 / 4294967296.0) - 1.0 : 0.0;
}

// ---- heat ------------------------------------------------------------------------------------------------------------
struct Heat3 {
    G3 g;
    const double* kk[3];             // conductivity on the faces normal to z, x, y: kz at (z_i+1/2, x_j, y_k) ...
    const double* cdt;               // dt / (rho Cp) at the nodes
    const double* rdb[3];            // 1 / (mid[i] - mid[i-1]) per axis
    int bc[6];                       // z0, x0, y0, zL, xL, yL  (pylamp_diff.py: FIXTEMP 0 / FIXFLOW 1)
};
// pylamp_diff.py:99-179 extended: z-walls own their edges, then x-walls, then y-walls
template <bool SCALED> __global__ __launch_bounds__(256) void k3_heat_apply(Heat3 op, const double* __restrict__ T, double* __restrict__ y) {
    K3_PROLOGUE(op.g)
    const G3& g = op.g;
    const double t = T[c];
    double r, dg = 1.0;
    int wall = -1, ax = 0, hi = 0;
    for (int a = 0; a < 3 && wall < 0; a++) {
        if (idx[a] == 0) { wall = a; ax = a; hi = 0; }
        else if (idx[a] == g.gn[a] - 1) { wall = a + 3; ax = a; hi = 1; }
    }
    if (wall >= 0) {
        if (op.bc[wall] == PL_BC_FIXTEMP) r = t;
        else if (!hi) { const double kq = op.kk[ax][c] * TB(g.rd[ax], 0); r = kq * (T[c + g.s[ax]] - t); dg = -kq; }
        else { const double kq = op.kk[ax][c - g.s[ax]] * TB(g.rd[ax], g.gn[ax] - 2); r = kq * (t - T[c - g.s[ax]]); dg = kq; }
    } else {
        double fl = 0.0, dsum = 0.0;
        for (int a = 0; a < 3; a++) {
            const double kp = op.kk[a][c] * TB(g.rd[a], idx[a]), km = op.kk[a][c - g.s[a]] * TB(g.rd[a], idx[a] - 1);
            const double rb = TB(op.rdb[a], idx[a]);
            fl += (kp * (T[c + g.s[a]] - t) - km * (t - T[c - g.s[a]])) * rb;
            dsum += (kp + km) * rb;
        }
        const double cc = op.cdt[c];
        r = cc * fl - t;
        if (SCALED) dg = -cc * dsum - 1.0;
    }
    y[c] = SCALED ? r / dg : r;
}
__global__ __launch_bounds__(256) void k3_heat_rhs(Heat3 op, const double* __restrict__ Told, const double* __restrict__ H, const double* bcv,
                                                   double* __restrict__ rhs, int scaled) {
    K3_PROLOGUE(op.g)
    const G3& g = op.g;
    int wall = -1, ax = 0, hi = 0;
    for (int a = 0; a < 3 && wall < 0; a++) {
        if (idx[a] == 0) { wall = a; ax = a; hi = 0; }
        else if (idx[a] == g.gn[a] - 1) { wall = a + 3; ax = a; hi = 1; }
    }
    double r, dg = 1.0;
    if (wall >= 0) {
        r = bcv[wall];
        if (op.bc[wall] != PL_BC_FIXTEMP) dg = hi ? op.kk[ax][c - g.s[ax]] * TB(g.rd[ax], g.gn[ax] - 2) : -op.kk[ax][c] * TB(g.rd[ax], 0);
    } else {
        r = -Told[c] - op.cdt[c] * H[c];
        if (scaled) {
            double dsum = 0.0;
            for (int a = 0; a < 3; a++)
                dsum += (op.kk[a][c] * TB(g.rd[a], idx[a]) + op.kk[a][c - g.s[a]] * TB(g.rd[a], idx[a] - 1)) * TB(op.rdb[a], idx[a]);
            dg = -op.cdt[c] * dsum - 1.0;
        }
    }
    rhs[c] = scaled ? r / dg : r;
}
__global__ __launch_bounds__(256) void k3_heat_coef(G3 g, const double* __restrict__ rho, const double* __restrict__ cp, double dt, double* __restrict__ out) {
    K3_PROLOGUE(g)
    out[c] = dt / (rho[c] * cp[c]);
}

// ---- host <-> device layout ------------------------------------------------------------------------------------------
// host arrays are C-order (nz, nx, ny) [optionally with ncomp interleaved components last]
// (src / dst: the GLOBAL array; from_host also fills the ring of nodes around the block from it -- coefficient arrays need no exchange)
__global__ __launch_bounds__(256) void k3_from_host(G3 g, const double* __restrict__ src, int ncomp, int comp, double* __restrict__ dst) {
    const int lk = (int)(blockIdx.x * 64 + threadIdx.x) - 1, lj = (int)(blockIdx.y * 4 + threadIdx.y) - 1, li = (int)blockIdx.z - 1;
    if (lk > g.n[2] || lj > g.n[1]) return;
    const int i = li + g.o[0], j = lj + g.o[1], k = lk + g.o[2];
    if (i < 0 || i >= g.gn[0] || j < 0 || j >= g.gn[1] || k < 0 || k >= g.gn[2]) return;
    dst[i3(g, li, lj, lk)] = src[(((long long)i * g.gn[1] + j) * g.gn[2] + k) * ncomp + comp];
}
__global__ __launch_bounds__(256) void k3_to_host(G3 g, const double* __restrict__ src, int ncomp, int comp, double* __restrict__ dst) {
    K3_PROLOGUE(g)
    dst[(((long long)i * g.gn[1] + j) * g.gn[2] + k) * ncomp + comp] = src[c];
}
// one plane of nodes normal to `axis` (local index `plane`, -1 .. n: the rings included) of na arrays <-> a dense buffer; the other
// two axes run over -1 .. n, so that exchanging z, then x, then y carries the edge and corner nodes along (halo3)
struct Slab3 { double* a[4]; int na; };
__global__ __launch_bounds__(256) void k3_slab(G3 g, int axis, int plane, Slab3 sl, double* __restrict__ buf, int to_buf) {
    const int A1 = axis == 0 ? 1 : 0, A2 = axis == 2 ? 1 : 2;       // the two in-plane axes, the faster one second
    const int n1 = g.n[A1] + 2, n2 = g.n[A2] + 2;
    const long long per = (long long)n1 * n2, total = per * sl.na;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int q = (int)(t / per); const long long e = t % per;
        int id[3]; id[axis] = plane; id[A1] = (int)(e / n2) - 1; id[A2] = (int)(e % n2) - 1;
        double* p = sl.a[q] + i3(g, id[0], id[1], id[2]);
        if (to_buf) buf[t] = *p; else *p = buf[t];
    }
}
// up to five dot products over the OWNED nodes of na consecutive arrays per vector (several ranks: the rings hold the neighbours' values)
__global__ __launch_bounds__(256) void k3_dots_own(G3 g, int na, Dot5 d, double* __restrict__ part) {
    double acc[5] = {0, 0, 0, 0, 0};
    const long long rows = (long long)na * g.n[0] * g.n[1];
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int q = (int)(r / ((long long)g.n[0] * g.n[1])); const int ij = (int)(r % ((long long)g.n[0] * g.n[1]));
        const long long base = (long long)q * g.vol + i3(g, ij / g.n[1], ij % g.n[1], 0);
        for (int kk = threadIdx.x; kk < g.n[2]; kk += 256)
            for (int u = 0; u < d.n; u++) acc[u] += d.a[u][base + kk] * d.b[u][base + kk];
    }
    __shared__ double sh[5][4];
    for (int u = 0; u < 5; u++) {
        double a = acc[u];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if ((threadIdx.x & 63) == 0) sh[u][threadIdx.x >> 6] = a;
    }
    __syncthreads();
    if (threadIdx.x < 5) part[5 * blockIdx.x + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

// =====================================================================================================================
// host side
// =====================================================================================================================
struct G3Host { G3 d; std::vector<double> c[3]; double* tables = nullptr; };
struct Lev3 {
    G3Host gh; Op3 op{};
    double *es = nullptr, *en = nullptr; bool own = false;
    double* v[3][3] = {{nullptr}}; double* f[3] = {nullptr}; double* r[3] = {nullptr};
    double* dinv[3] = {nullptr, nullptr, nullptr};      // 1 / diag of the velocity rows (k3_dinv)
    double lmax = 3.0;
    double* eig[3] = {nullptr, nullptr, nullptr}; bool eig_valid = false;      // eigenvector of the last power iteration (warm restart)
};
struct pl3_ctx {
    bool have_x = false, have_T = false;      // a device-resident Stokes / heat solution exists (pl3_get_solution)
    int device = 0; hipStream_t stream = nullptr; hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // several ranks (pl3_set_comm): Pz x Px x Py blocks of the node grid, rank = (pz Px + px) Py + py; the messages and host reductions
    // go through a 2-D context that carries the transport (native RCCL / callback table / in-process group, pl_comm.hip) and whose stream
    // this context then shares
    pl_ctx* comm = nullptr; bool own_stream = true;
    int P[3] = {1, 1, 1}, pc[3] = {0, 0, 0}, nranks = 1, rank = 0;
    int gn[3] = {0, 0, 0}; std::vector<double> gcoord[3];     // the global grid
    double* hbuf = nullptr; size_t hbuf_n = 0;                // halo exchange: send lo | send hi | recv lo | recv hi
    long long halo_calls = 0, reduce_calls = 0;
    bool halo_failed = false;       // an exchange inside the V-cycle failed (its callers are void): the preconditioner reports it
    std::string err;
    G3Host geom;
    double *es = nullptr, *en = nullptr, *rho = nullptr; Op3 op{}; bool op_ready = false;
    std::vector<Lev3*> levels;
    double* vec[14][4] = {{nullptr}};           // BiCGStab work vectors (4 arrays each)
    double* part = nullptr; double* hpart = nullptr;
    double* stage = nullptr; size_t stage_bytes = 0;
    // heat
    Heat3 hop{}; bool hop_ready = false; double* hk[3] = {nullptr}; double *hT = nullptr, *hH = nullptr, *hcdt = nullptr, *hrho = nullptr, *hcp = nullptr;
    double* hbcv = nullptr; double* htab = nullptr; double* hvec[12] = {nullptr}; double hbcv_host[6] = {0};
    int nu = 2, nu_fine = 0; bool nu_set = false; int coarse_sweeps = 12; double cheb_ratio = 6.0;      // nu_fine: sweeps on the finest level (0 = nu)
    // deflation of the pressure-anchor mode (see pl_solver.hip): the vector w = A^-1 u lives in vec[13], kept between solves
    double *dfl_y = nullptr, *dfl_t = nullptr; bool dfl_valid = false;
    bool dfl_active = false; double dfl_yAw = 0.0, dfl_wvel2 = 0.0;      // of the running solve: the anchor-mode term of the error estimate
    double etol = 3e-8;                     // bound on the velocity-error estimate of a converged Stokes solve (PYLAMP_STOKES_ETOL)
};
static thread_local std::string p3_tls_error;
static int p3_fail(pl3_ctx* ctx, const std::string& m) { if (ctx) ctx->err = m; p3_tls_error = m; return 1; }
#define P3_HIP(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return p3_fail(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define P3_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

// gn / c: the GLOBAL grid of the level; the block of this rank follows from ctx->P / ctx->pc: C = (gn - 1) / P cells per block, first
// node pc C, the last block of an axis also owns the last node
static int g3_build(pl3_ctx* ctx, G3Host& gh, const int gn[3], const double* const c[3]) {
    G3& g = gh.d;
    int n[3];
    for (int a = 0; a < 3; a++) {
        if ((gn[a] - 1) % ctx->P[a]) return p3_fail(ctx, "3-D grid: (nodes - 1) must be divisible by the number of blocks along every axis");
        const int C = (gn[a] - 1) / ctx->P[a];
        g.gn[a] = gn[a]; g.o[a] = ctx->pc[a] * C; g.n[a] = n[a] = C + (ctx->pc[a] == ctx->P[a] - 1 ? 1 : 0);
        gh.c[a].assign(c[a], c[a] + gn[a]);
    }
    const long long py = ((P3_PAD + n[2] + 1 + 15) / 16) * 16;
    g.s[2] = 1; g.s[1] = py; g.s[0] = (long long)(n[1] + 2) * py;
    g.vol = (long long)(n[0] + 2) * g.s[0];
    for (int a = 0; a < 3; a++) n[a] = gn[a];                 // the tables are global
    size_t len[3], tot = 0;
    for (int a = 0; a < 3; a++) { len[a] = ((size_t)n[a] + 2 * PL_TOFF + 3) & ~(size_t)1; tot += 2 * len[a]; }
    std::vector<double> t(tot, 0.0);
    size_t off = 0, offs[3][2];
    for (int a = 0; a < 3; a++) {
        offs[a][0] = off; off += len[a]; offs[a][1] = off; off += len[a];
        for (int i = 0; i + 1 < n[a]; i++) t[offs[a][0] + i + PL_TOFF] = 1.0 / (c[a][i + 1] - c[a][i]);
        for (int i = 1; i + 1 < n[a]; i++) t[offs[a][1] + i + PL_TOFF] = 1.0 / (c[a][i + 1] - c[a][i - 1]);
    }
    if (gh.tables) (void)hipFree(gh.tables);
    P3_HIP(ctx, hipMalloc((void**)&gh.tables, tot * sizeof(double)));
    P3_HIP(ctx, hipMemcpy(gh.tables, t.data(), tot * sizeof(double), hipMemcpyHostToDevice));
    for (int a = 0; a < 3; a++) { g.rd[a] = gh.tables + offs[a][0]; g.rD[a] = gh.tables + offs[a][1]; }
    return 0;
}
static int dmal(pl3_ctx* ctx, double** p, long long n) {
    P3_HIP(ctx, hipMalloc((void**)p, (size_t)n * sizeof(double)));
    P3_HIP(ctx, hipMemsetAsync(*p, 0, (size_t)n * sizeof(double), ctx->stream));
    return 0;
}
static dim3 g1(long long n) { long long b = (n + 255) / 256; return dim3((unsigned)(b > 8192 ? 8192 : b)); }

extern "C" const char* pl3_last_error(const pl3_ctx* ctx) { return ctx ? ctx->err.c_str() : p3_tls_error.c_str(); }

extern "C" int pl3_create(pl3_ctx** out, int device, int nz, int nx, int ny, const double* zc, const double* xc, const double* yc) {
    if (!out) return p3_fail(nullptr, "pl3_create: out is NULL");
    *out = nullptr;
    if (nz < 5 || nx < 5 || ny < 5 || !zc || !xc || !yc) return p3_fail(nullptr, "pl3_create: need at least 5 nodes per axis and the three coordinate arrays");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return p3_fail(nullptr, "pl3_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return p3_fail(nullptr, "pl3_create: bad device index");
    pl3_ctx* ctx = new pl3_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { delete ctx; return p3_fail(nullptr, "pl3_create: stream creation failed"); }
    const int n[3] = {nz, nx, ny}; const double* c[3] = {zc, xc, yc};
    for (int a = 0; a < 3; a++) { ctx->gn[a] = n[a]; ctx->gcoord[a].assign(c[a], c[a] + n[a]); }
    if (g3_build(ctx, ctx->geom, n, c)) { std::string m = ctx->err; delete ctx; return p3_fail(nullptr, m); }
    if (const char* e = getenv("PYLAMP_MG_NU3")) { int a = atoi(e); if (a >= 1 && a <= 6) { ctx->nu = a; ctx->nu_set = true; } }
    if (const char* e = getenv("PYLAMP_MG_NU3_FINE")) { int a = atoi(e); if (a >= 1 && a <= 6) ctx->nu_fine = a; }
    *out = ctx;
    return 0;
}
static void free_levels3(pl3_ctx* ctx) {
    for (Lev3* L : ctx->levels) {
        if (L->own) { (void)hipFree(L->es); (void)hipFree(L->en); }
        for (int b = 0; b < 3; b++) for (int q = 0; q < 3; q++) if (L->v[b][q]) (void)hipFree(L->v[b][q]);
        for (int q = 0; q < 3; q++) { if (L->f[q]) (void)hipFree(L->f[q]); if (L->r[q]) (void)hipFree(L->r[q]); if (L->eig[q]) (void)hipFree(L->eig[q]); if (L->dinv[q]) (void)hipFree(L->dinv[q]); }
        if (L->gh.tables) (void)hipFree(L->gh.tables);
        delete L;
    }
    ctx->levels.clear();
}
extern "C" void pl3_destroy(pl3_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_levels3(ctx);
    for (double* q : {ctx->es, ctx->en, ctx->rho, ctx->part, ctx->stage, ctx->hT, ctx->hH, ctx->hcdt, ctx->hrho, ctx->hcp, ctx->hbcv, ctx->htab, ctx->hbuf}) if (q) (void)hipFree(q);
    for (auto& v : ctx->vec) if (v[0]) (void)hipFree(v[0]);
    for (double* q : ctx->hk) if (q) (void)hipFree(q);
    for (double* q : ctx->hvec) if (q) (void)hipFree(q);
    if (ctx->hpart) (void)hipHostFree(ctx->hpart);
    if (ctx->geom.tables) (void)hipFree(ctx->geom.tables);
    (void)hipEventDestroy(ctx->ev0); (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
static int stage3(pl3_ctx* ctx, size_t bytes) {
    if (ctx->stage_bytes >= bytes) return 0;
    if (ctx->stage) { P3_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->stage); ctx->stage = nullptr; }
    P3_HIP(ctx, hipMalloc((void**)&ctx->stage, bytes));
    ctx->stage_bytes = bytes;
    return 0;
}
// host arrays are GLOBAL on every rank: the block and the ring of nodes around it are taken from them; a download assembles the global
// array from the blocks (own block + zeros, summed over the ranks)
static int upload3(pl3_ctx* ctx, const double* host, int ncomp, double* const* dst) {
    const G3& g = ctx->geom.d;
    const size_t bytes = (size_t)g.gn[0] * g.gn[1] * g.gn[2] * ncomp * sizeof(double);
    P3_TRY(stage3(ctx, bytes));
    P3_HIP(ctx, hipMemcpyAsync(ctx->stage, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    const dim3 gr((g.n[2] + 2 + 63) / 64, (g.n[1] + 2 + 3) / 4, g.n[0] + 2);
    for (int q = 0; q < ncomp; q++) hipLaunchKernelGGL(k3_from_host, gr, dim3(64, 4), 0, ctx->stream, g, (const double*)ctx->stage, ncomp, q, dst[q]);
    P3_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
static int download3(pl3_ctx* ctx, double* const* src, int ncomp, double* host) {
    const G3& g = ctx->geom.d;
    const size_t count = (size_t)g.gn[0] * g.gn[1] * g.gn[2] * ncomp, bytes = count * sizeof(double);
    P3_TRY(stage3(ctx, bytes));
    if (ctx->nranks > 1) P3_HIP(ctx, hipMemsetAsync(ctx->stage, 0, bytes, ctx->stream));
    for (int q = 0; q < ncomp; q++) hipLaunchKernelGGL(k3_to_host, grid3(g), dim3(64, 4), 0, ctx->stream, g, (const double*)src[q], ncomp, q, ctx->stage);
    P3_HIP(ctx, hipMemcpyAsync(host, ctx->stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
    P3_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->nranks > 1 && pl_allreduce_host(ctx->comm, host, (long long)count, 0)) return p3_fail(ctx, std::string("3-D download: ") + pl_last_error(ctx->comm));
    return 0;
}

// ---- several ranks: halo exchange and reductions ------------------------------------------------------------------------------
// One ring of nodes around the block is all the stencils reach (faces, the edge neighbours (i-1, j+1) ... of the momentum rows, and the
// corner nodes of the full-weighting restriction): exchanged axis by axis -- z planes, then x planes including the z ring just
// received, then y planes including both -- so that 6 messages carry all 26 neighbours' nodes.
static int halo3_impl(pl3_ctx* ctx, const G3& g, double* const* arr, int na);
static int halo3(pl3_ctx* ctx, const G3& g, double* const* arr, int na) {
    const int rc = halo3_impl(ctx, g, arr, na);
    if (rc) ctx->halo_failed = true;
    return rc;
}
static int halo3_impl(pl3_ctx* ctx, const G3& g, double* const* arr, int na) {
    if (ctx->nranks <= 1) return 0;
    ctx->halo_calls++;
    for (int a = 0; a < 3; a++) {
        if (ctx->P[a] == 1) continue;
        const int A1 = a == 0 ? 1 : 0, A2 = a == 2 ? 1 : 2;
        const long long cnt = (long long)na * (g.n[A1] + 2) * (g.n[A2] + 2);
        if (ctx->hbuf_n < (size_t)(4 * cnt)) {
            if (ctx->hbuf) { P3_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->hbuf); ctx->hbuf = nullptr; }
            P3_HIP(ctx, hipMalloc((void**)&ctx->hbuf, (size_t)(4 * cnt) * sizeof(double)));
            ctx->hbuf_n = (size_t)(4 * cnt);
        }
        Slab3 sl{}; sl.na = na; for (int q = 0; q < na; q++) sl.a[q] = arr[q];
        const bool lo = ctx->pc[a] > 0, hi = ctx->pc[a] < ctx->P[a] - 1;
        int step = 1; for (int b = a + 1; b < 3; b++) step *= ctx->P[b];          // rank distance of the neighbour along axis a
        const unsigned nb = (unsigned)std::min<long long>((cnt + 255) / 256, 2048);
        PlMsg m[2]; int nm = 0;
        if (lo) { hipLaunchKernelGGL(k3_slab, dim3(nb), dim3(256), 0, ctx->stream, g, a, 0, sl, ctx->hbuf, 1);
                  m[nm++] = PlMsg{ctx->rank - step, ctx->hbuf, cnt, ctx->hbuf + 2 * cnt, cnt}; }
        if (hi) { hipLaunchKernelGGL(k3_slab, dim3(nb), dim3(256), 0, ctx->stream, g, a, g.n[a] - 1, sl, ctx->hbuf + cnt, 1);
                  m[nm++] = PlMsg{ctx->rank + step, ctx->hbuf + cnt, cnt, ctx->hbuf + 3 * cnt, cnt}; }
        if (pl_comm_sendrecv(ctx->comm, m, nm)) return p3_fail(ctx, std::string("3-D halo exchange: ") + pl_last_error(ctx->comm));
        if (lo) hipLaunchKernelGGL(k3_slab, dim3(nb), dim3(256), 0, ctx->stream, g, a, -1, sl, ctx->hbuf + 2 * cnt, 0);
        if (hi) hipLaunchKernelGGL(k3_slab, dim3(nb), dim3(256), 0, ctx->stream, g, a, g.n[a], sl, ctx->hbuf + 3 * cnt, 0);
    }
    P3_HIP(ctx, hipGetLastError());
    return 0;
}
static int halo3_1(pl3_ctx* ctx, const G3& g, double* arr) { double* a[1] = {arr}; return halo3(ctx, g, a, 1); }
static int allreduce3(pl3_ctx* ctx, double* v, int n, int op) {
    if (ctx->nranks <= 1) return 0;
    ctx->reduce_calls++;
    if (pl_allreduce_host(ctx->comm, v, n, op)) return p3_fail(ctx, std::string("3-D all-reduce: ") + pl_last_error(ctx->comm));
    return 0;
}
static int need_vecs(pl3_ctx* ctx, int nvec) {
    const long long vol = ctx->geom.d.vol;
    // the four arrays of a work vector are ONE allocation (vec[v][q] = vec[v][0] + q vol): the flat vector kernels and the reductions
    // of BiCGStab then take a whole vector per launch (and per host synchronisation) instead of one array
    for (int v = 0; v < nvec; v++)
        if (!ctx->vec[v][0]) {
            P3_TRY(dmal(ctx, &ctx->vec[v][0], 4 * vol));
            for (int q = 1; q < 4; q++) ctx->vec[v][q] = ctx->vec[v][0] + (long long)q * vol;
        }
    if (!ctx->part) {
        P3_TRY(dmal(ctx, &ctx->part, 11 * D3_BLOCKS));
        P3_HIP(ctx, hipHostMalloc((void**)&ctx->hpart, 11 * D3_BLOCKS * sizeof(double)));
    }
    return 0;
}

// Several ranks (right after pl3_create, before any coefficients): this context becomes rank `comm`'s rank of a Pz x Px x Py block
// layout, rank = (pz Px + px) Py + py.  comm: a 2-D context (pl_create on any small grid) whose communicator has been set
// (pl_set_comm / pl_set_comm_2d / pl_set_comm_local with Pz * Px = the number of ranks); it must outlive this context.
extern "C" int pl3_set_comm(pl3_ctx* ctx, pl_ctx* comm, int Pz, int Px, int Py) {
    if (!ctx || !comm) return p3_fail(ctx, "pl3_set_comm: NULL argument");
    if (Pz < 1 || Px < 1 || Py < 1 || Pz * Px * Py != comm->nranks) return p3_fail(ctx, "pl3_set_comm: Pz * Px * Py must equal the number of ranks of the communicator");
    if (ctx->op_ready || ctx->hop_ready || !ctx->levels.empty()) return p3_fail(ctx, "pl3_set_comm: call it right after pl3_create");
    if (comm->device != ctx->device) return p3_fail(ctx, "pl3_set_comm: the communicator context lives on another device");
    P3_HIP(ctx, hipSetDevice(ctx->device));
    P3_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) { (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    ctx->stream = comm->stream;                      // messages are ordered with this context's kernels
    ctx->comm = comm; ctx->nranks = comm->nranks; ctx->rank = comm->rank;
    ctx->P[0] = Pz; ctx->P[1] = Px; ctx->P[2] = Py;
    ctx->pc[2] = ctx->rank % Py; ctx->pc[1] = (ctx->rank / Py) % Px; ctx->pc[0] = ctx->rank / (Py * Px);
    for (int a = 0; a < 3; a++)
        if (ctx->P[a] > 1 && (ctx->gn[a] - 1) / ctx->P[a] < 4) return p3_fail(ctx, "pl3_set_comm: at least 4 cells per block and axis");
    const double* c[3] = {ctx->gcoord[0].data(), ctx->gcoord[1].data(), ctx->gcoord[2].data()};
    return g3_build(ctx, ctx->geom, ctx->gn, c);
}
extern "C" int pl3_local_block(pl3_ctx* ctx, int first[3], int count[3]) {
    for (int a = 0; a < 3; a++) { if (first) first[a] = ctx->geom.d.o[a]; if (count) count[a] = ctx->geom.d.n[a]; }
    return 0;
}
// halo exchanges (3 phases each) and host all-reduces of this context so far; reset != 0 clears the counters
extern "C" int pl3_comm_stats(pl3_ctx* ctx, int64_t out[2], int reset) {
    if (out) { out[0] = ctx->halo_calls; out[1] = ctx->reduce_calls; }
    if (reset) { ctx->halo_calls = 0; ctx->reduce_calls = 0; }
    return 0;
}

// pylamp_stokes.py:116-122 extended dimension-wise: Kcont = DIM mineta / sum(avgd), Kbond = DIM^2 mineta / sum(avgd)^2
extern "C" int pl3_stokes_set_coeffs(pl3_ctx* ctx, const double* etas, const double* etan, const double* rho, const double grav[3]) {
    if (!etas || !etan || !rho) return p3_fail(ctx, "pl3_stokes_set_coeffs: NULL argument");
    P3_HIP(ctx, hipSetDevice(ctx->device));
    const G3& g = ctx->geom.d;
    for (double** q : {&ctx->es, &ctx->en, &ctx->rho}) if (!*q) P3_TRY(dmal(ctx, q, g.vol));
    double* d1[1];
    d1[0] = ctx->es; P3_TRY(upload3(ctx, etas, 1, d1));
    d1[0] = ctx->en; P3_TRY(upload3(ctx, etan, 1, d1));
    d1[0] = ctx->rho; P3_TRY(upload3(ctx, rho, 1, d1));
    const size_t N = (size_t)g.gn[0] * g.gn[1] * g.gn[2];
    double mes = INFINITY, men = INFINITY; bool nes = false, nen = false;
    for (size_t t = 0; t < N; t++) { if (etas[t] != etas[t]) nes = true; else mes = std::min(mes, etas[t]); if (etan[t] != etan[t]) nen = true; else men = std::min(men, etan[t]); }
    if (nes) mes = NAN; if (nen) men = NAN;
    const double mineta = (men < mes) ? men : mes;                  // python's min(a, b), pylamp_stokes.py:116-118
    double sum = 0.0;
    for (int a = 0; a < 3; a++) sum += (ctx->geom.c[a].back() - ctx->geom.c[a].front()) / g.gn[a];     // avgd = L / n (sic, :119-120)
    Op3& op = ctx->op;
    op.g = g; op.es = ctx->es; op.en = ctx->en; op.rho = ctx->rho;
    op.Kc = 3.0 * mineta / sum; op.Kb = 9.0 * mineta / (sum * sum); op.iKc = 1.0 / op.Kc;
    op.slave = 1;
    for (int a = 0; a < 3; a++) op.grav[a] = grav ? grav[a] : (a == 0 ? 9.81 : 0.0);
    op.anchor[0] = 3; op.anchor[1] = 2; op.anchor[2] = 2;
    ctx->op_ready = true;
    return 0;
}
// slaved != 0 (default): the reference's wall treatment extended to 3-D -- the outermost in-domain tangential velocities are
// slaved to their inner neighbours (pylamp_stokes.py:170-175,209-214,249-255,296-301), which imposes free slip half a cell
// inside the wall: first-order accurate.  0: natural (mirror) wall rows, second-order accurate.
extern "C" int pl3_stokes_set_wall_rows(pl3_ctx* ctx, int slaved) {
    if (!ctx->op_ready) return p3_fail(ctx, "stokes operator not set");
    ctx->op.slave = slaved ? 1 : 0;
    return 0;
}
extern "C" int pl3_stokes_get_scaling(pl3_ctx* ctx, double* kc, double* kb) {
    if (!ctx->op_ready) return p3_fail(ctx, "stokes operator not set");
    if (kc) *kc = ctx->op.Kc; if (kb) *kb = ctx->op.Kb;
    return 0;
}
static V4 cv4(double* const* p) { V4 v; for (int q = 0; q < 4; q++) v.p[q] = p[q]; return v; }
static W4 wv4(double* const* p) { W4 v; for (int q = 0; q < 4; q++) v.p[q] = p[q]; return v; }
static V3 cv3(double* const* p) { V3 v; for (int q = 0; q < 3; q++) v.p[q] = p[q]; return v; }
static W3 wv3(double* const* p) { W3 v; for (int q = 0; q < 3; q++) v.p[q] = p[q]; return v; }

// y = A x (scaled: D_r A x): the marching LDS kernel for the interior rows + the per-node kernel on the rim, or the per-node kernel alone
template <bool SCALED> static void launch_apply3(pl3_ctx* ctx, const Op3& op, double* const* in, double* const* out) {
    if (k3_use_lds(op.g)) {
        hipLaunchKernelGGL(k3_apply_m<SCALED>, grid3m(op.g), dim3(64, 4), 0, ctx->stream, op, cv4(in), wv4(out), k3_band(), k3m_zc(op.g));
    } else hipLaunchKernelGGL(k3_apply<SCALED>, grid3(op.g), dim3(64, 4), 0, ctx->stream, op, cv4(in), wv4(out), 0);
}
extern "C" int pl3_stokes_apply(pl3_ctx* ctx, const double* x, double* y) {
    if (!ctx->op_ready) return p3_fail(ctx, "stokes operator not set");
    P3_HIP(ctx, hipSetDevice(ctx->device));
    P3_TRY(need_vecs(ctx, 2));
    P3_TRY(upload3(ctx, x, 4, ctx->vec[0]));                 // (the upload fills the ring of nodes around the block as well)
    launch_apply3<false>(ctx, ctx->op, ctx->vec[0], ctx->vec[1]);
    P3_HIP(ctx, hipGetLastError());
    return download3(ctx, ctx->vec[1], 4, y);
}
extern "C" int pl3_stokes_rhs(pl3_ctx* ctx, double* rhs) {
    if (!ctx->op_ready) return p3_fail(ctx, "stokes operator not set");
    P3_HIP(ctx, hipSetDevice(ctx->device));
    P3_TRY(need_vecs(ctx, 2));
    hipLaunchKernelGGL(k3_rhs, grid3(ctx->op.g), dim3(64, 4), 0, ctx->stream, ctx->op, wv4(ctx->vec[1]), 0);
    P3_HIP(ctx, hipGetLastError());
    return download3(ctx, ctx->vec[1], 4, rhs);
}
extern "C" int pl3_stokes_apply_bench(pl3_ctx* ctx, int scaled, int reps, double* avg_ms) {
    if (!ctx->op_ready) return p3_fail(ctx, "stokes operator not set");
    P3_HIP(ctx, hipSetDevice(ctx->device));
    P3_TRY(need_vecs(ctx, 2));
    for (int q = 0; q < 4; q++) hipLaunchKernelGGL(k3_random, grid3(ctx->op.g), dim3(64, 4), 0, ctx->stream, ctx->op.g, ctx->vec[0][q], 99u + q);
    auto launch = [&]() {
        if (scaled) launch_apply3<true>(ctx, ctx->op, ctx->vec[0], ctx->vec[1]);
        else launch_apply3<false>(ctx, ctx->op, ctx->vec[0], ctx->vec[1]);
    };
    launch();
    P3_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; r++) launch();
    P3_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    P3_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0; P3_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (avg_ms) *avg_ms = ms / reps;
    return 0;
}

// ---- reductions -----------------------------------------------------------------------------------------------------------
static int dots3(pl3_ctx* ctx, long long n, int nd, const double* const* a, const double* const* b, double* out) {
    Dot5 d{}; d.n = nd;
    for (int q = 0; q < nd; q++) { d.a[q] = a[q]; d.b[q] = b[q]; }
    if (ctx->nranks > 1) {
        // several ranks: only the OWNED nodes count (the rings hold copies of the neighbours' values); n = m x (elements of one array
        // of the finest grid or of a multigrid level), m <= 4 consecutive arrays per vector
        const G3* g = nullptr;
        if (n % ctx->geom.d.vol == 0 && n / ctx->geom.d.vol <= 4) g = &ctx->geom.d;
        for (size_t l = 0; !g && l < ctx->levels.size(); l++) if (n % ctx->levels[l]->gh.d.vol == 0 && n / ctx->levels[l]->gh.d.vol <= 4) g = &ctx->levels[l]->gh.d;
        if (!g) return p3_fail(ctx, "dots3: vector length matches no grid level (internal error)");
        hipLaunchKernelGGL(k3_dots_own, dim3(D3_BLOCKS), dim3(256), 0, ctx->stream, *g, (int)(n / g->vol), d, ctx->part);
        P3_HIP(ctx, hipMemcpyAsync(ctx->hpart, ctx->part, 5 * D3_BLOCKS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        P3_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int q = 0; q < nd; q++) { double sum = 0.0; for (int k = 0; k < D3_BLOCKS; k++) sum += ctx->hpart[5 * k + q]; out[q] = sum; }
        return allreduce3(ctx, out, nd, 0);
    }
    hipLaunchKernelGGL(k3_dots, dim3(D3_BLOCKS), dim3(256), 0, ctx->stream, n, d, ctx->part);
    P3_HIP(ctx, hipMemcpyAsync(ctx->hpart, ctx->part, 5 * D3_BLOCKS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    P3_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < nd; q++) { double s = 0.0; for (int k = 0; k < D3_BLOCKS; k++) s += ctx->hpart[5 * k + q]; out[q] = s; }
    return 0;
}
// the na arrays of a vector follow each other in memory (need_vecs)
static bool contig3(double* const* a, int na, long long vol) {
    for (int c = 1; c < na; c++) if (a[c] != a[0] + (long long)c * vol) return false;
    return true;
}
// dot products of multi-array vectors (na arrays each)
static int vdots(pl3_ctx* ctx, long long vol, int na, int nd, double* const* const* a, double* const* const* b, double* out) {
    bool flat = na > 1;
    for (int q = 0; q < nd && flat; q++) flat = contig3(a[q], na, vol) && contig3(b[q], na, vol);
    if (flat) {                                    // one pass, one copy, one synchronisation for the whole vector
        const double* aa[5]; const double* bb[5];
        for (int q = 0; q < nd; q++) { aa[q] = a[q][0]; bb[q] = b[q][0]; }
        return dots3(ctx, (long long)na * vol, nd, aa, bb, out);
    }
    for (int q = 0; q < nd; q++) out[q] = 0.0;
    for (int c = 0; c < na; c++) {
        const double* aa[5]; const double* bb[5]; double o[5];
        for (int q = 0; q < nd; q++) { aa[q] = a[q][c]; bb[q] = b[q][c]; }
        P3_TRY(dots3(ctx, vol, nd, aa, bb, o));
        for (int q = 0; q < nd; q++) out[q] += o[q];
    }
    return 0;
}

// ---- multigrid -----------------------------------------------------------------------------------------------------------
static int build_levels3(pl3_ctx* ctx) {
    if (ctx->levels.empty()) {
        int n[3] = {ctx->gn[0], ctx->gn[1], ctx->gn[2]};                     // GLOBAL node counts of the level
        std::vector<double> c[3] = {ctx->gcoord[0], ctx->gcoord[1], ctx->gcoord[2]};
        for (int l = 0;; l++) {
            Lev3* L = new Lev3();
            const double* cc[3] = {c[0].data(), c[1].data(), c[2].data()};
            if (g3_build(ctx, L->gh, n, cc)) { delete L; return 1; }
            const long long vol = L->gh.d.vol;
            if (l > 0) { L->own = true; P3_TRY(dmal(ctx, &L->es, vol)); P3_TRY(dmal(ctx, &L->en, vol)); }
            for (int b = 0; b < 3; b++) for (int q = 0; q < 3; q++) P3_TRY(dmal(ctx, &L->v[b][q], vol));
            for (int q = 0; q < 3; q++) { P3_TRY(dmal(ctx, &L->f[q], vol)); P3_TRY(dmal(ctx, &L->r[q], vol)); P3_TRY(dmal(ctx, &L->dinv[q], vol)); }
            ctx->levels.push_back(L);
            bool stop = false;
            for (int a = 0; a < 3; a++) if ((n[a] - 1) % 2 || (n[a] - 1) / 2 < 4) stop = true;
            // several ranks: every block is halved with the grid -- a block with an odd number of cells (or one that would shrink below
            // two) ends the hierarchy here, and this level is smoothed as the coarsest one (more sweeps: coarse_sweeps by its size)
            for (int a = 0; a < 3; a++) if (ctx->P[a] > 1 && (((n[a] - 1) / ctx->P[a]) % 2 || (n[a] - 1) / ctx->P[a] / 2 < 2)) stop = true;
            if (stop) break;
            for (int a = 0; a < 3; a++) {
                std::vector<double> c2;
                for (int i = 0; i < n[a]; i += 2) c2.push_back(c[a][i]);
                c[a].swap(c2); n[a] = (n[a] - 1) / 2 + 1;
            }
        }
    }
    ctx->levels[0]->es = ctx->es; ctx->levels[0]->en = ctx->en;
    for (size_t l = 0; l < ctx->levels.size(); l++) {
        Lev3* L = ctx->levels[l];
        if (l > 0) {
            Lev3* F = ctx->levels[l - 1];
            hipLaunchKernelGGL(k3_coarsen, grid3(L->gh.d), dim3(64, 4), 0, ctx->stream, F->gh.d, (const double*)F->es, (const double*)F->en, L->gh.d, L->es, L->en);
            P3_TRY(halo3_1(ctx, L->gh.d, L->es)); P3_TRY(halo3_1(ctx, L->gh.d, L->en));      // the rows of this level and the next coarsening read the ring
        }
        L->op = ctx->op; L->op.g = L->gh.d; L->op.es = L->es; L->op.en = L->en; L->op.slave = (l == 0) ? ctx->op.slave : 0;
        hipLaunchKernelGGL(k3_dinv, grid3(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, wv3(L->dinv));
        P3_TRY(halo3(ctx, L->gh.d, L->dinv, 3));            // (a slaved row reads its master's entry, possibly in the neighbour block)
        // lambda_max of D^-1 A_vv by power iteration
        const long long vol = L->gh.d.vol;
        // Warm restart as in the 2-D solver (pl_solver.hip): consecutive solves on one context (a time loop, or the same operator
        // again) start from the eigenvector of the last one and stop as soon as an iteration confirms the last estimate to 1 % --
        // 12 cold iterations per level were 5 % of a 257^3 solve.
        const bool warm = L->eig_valid && L->eig[0];
        if (warm) for (int q = 0; q < 3; q++) P3_HIP(ctx, hipMemcpyAsync(L->v[0][q], L->eig[q], (size_t)vol * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        else for (int q = 0; q < 3; q++) hipLaunchKernelGGL(k3_random, grid3(L->gh.d), dim3(64, 4), 0, ctx->stream, L->gh.d, L->v[0][q], 777u + q);
        {   // the start vector must satisfy the constraints of THIS operator: close it
            for (int q = 0; q < 3; q++) P3_HIP(ctx, hipMemsetAsync(L->r[q], 0, (size_t)vol * sizeof(double), ctx->stream));
            P3_TRY(halo3(ctx, L->gh.d, L->v[0], 3));
            hipLaunchKernelGGL(k3_close, grid3(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, wv3(L->v[0]), cv3(L->r));
        }
        double lam = warm ? L->lmax / 1.1 : 2.5, lam_prev = 0.0;
        for (int it = 0; it < 12; it++) {
            if (warm && it >= 1 && std::fabs(lam - lam_prev) < 0.01 * lam) break;
            lam_prev = lam;
            P3_TRY(halo3(ctx, L->gh.d, L->v[0], 3));
            hipLaunchKernelGGL(k3_resid, grid3(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, cv3(L->v[0]), cv3(L->f), wv3(L->v[1]), 1, 0);
            double* const* aa[2] = {L->v[1], L->v[0]}; double* const* bb[2] = {L->v[1], L->v[0]};
            double nn[2];
            P3_TRY(vdots(ctx, vol, 3, 2, aa, bb, nn));
            if (!(nn[0] > 0.0) || !(nn[1] > 0.0)) break;
            lam = std::sqrt(nn[0] / nn[1]);
            for (int q = 0; q < 3; q++) hipLaunchKernelGGL(k3_axpby, g1(vol), dim3(256), 0, ctx->stream, vol, L->v[0][q], 1.0 / std::sqrt(nn[0]), (const double*)L->v[1][q], 0.0, (const double*)L->v[1][q]);
        }
        L->lmax = 1.1 * lam;
        if (!L->eig[0]) for (int q = 0; q < 3; q++) P3_TRY(dmal(ctx, &L->eig[q], vol));
        for (int q = 0; q < 3; q++) P3_HIP(ctx, hipMemcpyAsync(L->eig[q], L->v[0][q], (size_t)vol * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        L->eig_valid = std::isfinite(lam) && lam > 0.0;
    }
    P3_HIP(ctx, hipGetLastError());
    return 0;
}
// nsweep sweeps; cur = index of the buffer holding the iterate on entry (ignored if zero_guess) and on exit
static void smooth3(pl3_ctx* ctx, Lev3* L, double* const* f, int nsweep, double ratio, bool zero_guess, int& cur, const W3* final_out = nullptr) {
    const double lmax = L->lmax, lmin = lmax / ratio, theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho_old = 1.0 / sigma;
    int prev = (cur + 1) % 3;
    for (int k = 0; k < nsweep; k++) {
        double c1, c2;
        if (k == 0) { c1 = 0.0; c2 = 1.0 / theta; }
        else { const double rho = 1.0 / (2.0 * sigma - rho_old); c1 = rho * rho_old; c2 = 2.0 * rho / delta; rho_old = rho; }
        int nxt = 0; while (nxt == cur || nxt == prev) nxt++;
        V3 vp = cv3(L->v[prev]);
        if (k == 1 && zero_guess) for (int q = 0; q < 3; q++) vp.p[q] = nullptr;
        // the last sweep may write straight into the caller's arrays (the preconditioner's output) instead of a level buffer
        const W3 dst = (final_out && k == nsweep - 1) ? *final_out : wv3(L->v[nxt]);
        // several ranks: the sweep reads the iterate one node into the ring (the first sweep from the zero guess: only the right-hand
        // side, at a slave's master)
        if (k == 0 && zero_guess) (void)halo3(ctx, L->gh.d, (double* const*)f, 3); else (void)halo3(ctx, L->gh.d, L->v[cur], 3);
        if (!(k == 0 && zero_guess) && k3_use_lds(L->gh.d)) {        // interior rows: the marching LDS kernel; the rim: the per-node kernel
            hipLaunchKernelGGL(k3_sweep_m<0>, grid3m(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, cv3(L->v[cur]), vp, cv3(f), dst, c1, c2, k3_band(), k3m_zc(L->gh.d));
        } else if (k == 0 && zero_guess)
            hipLaunchKernelGGL(k3_cheb0, grid3(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, cv3(f), cv3(L->dinv), dst, c2);
        else
        hipLaunchKernelGGL(k3_cheb, grid3(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, cv3(L->v[cur]), vp, cv3(f), dst, c1, c2, 0, 0);
        prev = cur; cur = nxt;
    }
}
static void vcycle3(pl3_ctx* ctx, size_t l, double* const* f, int& out_buf, const W3* final_out = nullptr) {
    Lev3* L = ctx->levels[l];
    int cur = 0;
    if (l + 1 == ctx->levels.size()) {
        const G3& g = L->gh.d;
        double ratio = 0.4 * std::pow((double)g.gn[0] * g.gn[1] * g.gn[2], 2.0 / 3.0); if (ratio < 30.0) ratio = 30.0;
        int n = std::max((int)std::sqrt(ratio), ctx->coarse_sweeps); if (n > 150) n = 150;
        smooth3(ctx, L, f, n, ratio, true, cur);
        out_buf = cur;
        return;
    }
    // From 10^6 nodes up: V(1,1) on the finest level, V(3,3) below (the 2-D solver's choice, pl_solver.hip) -- 257^3: 494 ms per solve
    // pair with 24 iterations against 539 ms with 23 for V(2,2) throughout (1/4: 492, 2/3: 499, 1/2: 504; tools/run_nu3.sh).
    // PYLAMP_MG_NU3 / PYLAMP_MG_NU3_FINE override the two counts.
    const bool big = (long long)ctx->gn[0] * ctx->gn[1] * ctx->gn[2] >= 1000000;
    const int nu_fine = ctx->nu_fine > 0 ? ctx->nu_fine : (big && !ctx->nu_set ? 1 : ctx->nu);
    const int nu_rest = (big && !ctx->nu_set) ? 3 : ctx->nu;
    const int nu = l == 0 ? nu_fine : nu_rest;
    smooth3(ctx, L, f, nu, ctx->cheb_ratio, true, cur);
    (void)halo3(ctx, L->gh.d, L->v[cur], 3);
    if (k3_use_lds(L->gh.d)) {
        V3 none{};
        hipLaunchKernelGGL(k3_sweep_m<1>, grid3m(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, cv3(L->v[cur]), none, cv3(f), wv3(L->r), 0.0, 0.0, k3_band(), k3m_zc(L->gh.d));
    } else hipLaunchKernelGGL(k3_resid, grid3(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, cv3(L->v[cur]), cv3(f), wv3(L->r), 0, 0);
    Lev3* C = ctx->levels[l + 1];
    (void)halo3(ctx, L->gh.d, L->r, 3);
    hipLaunchKernelGGL(k3_restrict, grid3(C->gh.d), dim3(64, 4), 0, ctx->stream, L->gh.d, C->op, cv3(L->r), wv3(C->f));
    int cb = 0;
    vcycle3(ctx, l + 1, C->f, cb);
    const int nxt = (cur + 1) % 3;
    (void)halo3(ctx, C->gh.d, C->v[cb], 3);              // (the ring of L->v[cur] is still the one exchanged for the residual)
    hipLaunchKernelGGL(k3_prolong_add, grid3(L->gh.d), dim3(64, 4), 0, ctx->stream, L->op, C->gh.d, cv3(C->v[cb]), cv3(L->v[cur]), wv3(L->v[nxt]));
    cur = nxt;
    smooth3(ctx, L, f, nu, ctx->cheb_ratio, false, cur, final_out);
    out_buf = cur;
}

// ---- generic right-preconditioned BiCGStab on multi-array vectors (correction form, true-residual stopping) --------------------
typedef std::function<int(double* const*, double* const*)> Op3Fn;
// ---- deflation of the pressure-anchor mode and the velocity-error estimate: the 3-D counterparts of pl_solver.hip --------------
// u: the continuity residual that the block preconditioner turns into a constant pressure; yw: cell volume x (rdz + rdx + rdy) on the
// continuity rows, so that yw . (scaled continuity row of v) = sum of volume x div(v)
__global__ __launch_bounds__(256) void k3_defl_setup(Op3 op, W4 u, double* __restrict__ yw) {
    K3_PROLOGUE(op.g)
    long long moff;
    const bool cont = cls3_p(op, idx, moff) == 1;
    const G3& g = op.g;
    const double rs = TB(g.rd[0], i) + TB(g.rd[1], j) + TB(g.rd[2], k);
    u.p[0][c] = 0.0; u.p[1][c] = 0.0; u.p[2][c] = 0.0;
    u.p[3][c] = cont ? 1.0 / (op.en[c] * rs) : 0.0;
    yw[c] = cont ? rs / (TB(g.rd[0], i) * TB(g.rd[1], j) * TB(g.rd[2], k)) : 0.0;
}
// the scaled continuity rows of a velocity field (0 on the other pressure rows)
__global__ __launch_bounds__(256) void k3_cont_rows(Op3 op, V3 v, double* __restrict__ out) {
    K3_PROLOGUE(op.g)
    long long moff;
    double o = 0.0;
    if (cls3_p(op, idx, moff) == 1) {
        const G3& g = op.g;
        const double rz = TB(g.rd[0], i), rx = TB(g.rd[1], j), ry = TB(g.rd[2], k);
        o = ((v.p[0][c + g.s[0]] - v.p[0][c]) * rz + (v.p[1][c + g.s[1]] - v.p[1][c]) * rx + (v.p[2][c + g.s[2]] - v.p[2][c]) * ry) / (rz + rx + ry);
    }
    out[c] = o;
}

struct Stats3 { int iterations, converged; double rel_residual; int napply, nprec; double error_estimate; };
static int bicgstab3(pl3_ctx* ctx, long long vol, int na, const Op3Fn& A, const Op3Fn* M, double* const* b, double* const* x, bool use_x0,
                     double rtol, int maxit, double* const* const* w /* r rt p v s t y z dx r0 */, G3 g, Stats3* st, double ref_norm,
                     double etol = 0.0) {
    double* const* r = w[0]; double* const* rt = w[1]; double* const* p = w[2]; double* const* v = w[3]; double* const* s = w[4];
    double* const* t = w[5]; double* const* y = w[6]; double* const* z = w[7]; double* const* dxb = w[8]; double* const* r0b = w[9];
    auto flat3 = [&](double* const* a0, double* const* a1 = nullptr, double* const* a2 = nullptr) {
        return na > 1 && contig3(a0, na, vol) && (!a1 || contig3(a1, na, vol)) && (!a2 || contig3(a2, na, vol));
    };
    auto axpby = [&](double* const* yy, double a, double* const* xx, double bb, double* const* zz) {
        if (flat3(yy, xx, zz)) { hipLaunchKernelGGL(k3_axpby, g1(na * vol), dim3(256), 0, ctx->stream, na * vol, yy[0], a, (const double*)xx[0], bb, (const double*)zz[0]); return; }
        for (int c = 0; c < na; c++) hipLaunchKernelGGL(k3_axpby, g1(vol), dim3(256), 0, ctx->stream, vol, yy[c], a, (const double*)xx[c], bb, (const double*)zz[c]);
    };
    auto zero = [&](double* const* yy) {
        if (flat3(yy)) { (void)hipMemsetAsync(yy[0], 0, (size_t)na * vol * sizeof(double), ctx->stream); return; }
        for (int c = 0; c < na; c++) (void)hipMemsetAsync(yy[c], 0, (size_t)vol * sizeof(double), ctx->stream);
    };
    double d[5];
    { double* const* aa[1] = {b}; P3_TRY(vdots(ctx, vol, na, 1, aa, aa, d)); }
    double bnorm = std::sqrt(d[0]);
    if (ref_norm > 0.0 && bnorm > 0.0) bnorm = ref_norm;
    st->iterations = 0; st->converged = 0; st->rel_residual = 0.0;
    if (!(bnorm > 0.0)) { zero(x); st->converged = 1; return 0; }
    double* const* dx = x; double* const* r0 = b;
    if (use_x0) { dx = dxb; P3_TRY(A(x, v)); axpby(r0b, 1.0, b, -1.0, v); r0 = r0b; }
    zero(dx);
    axpby(r, 1.0, r0, 0.0, r0);
    for (int c = 0; c < na; c++) hipLaunchKernelGGL(k3_random, grid3(g), dim3(64, 4), 0, ctx->stream, g, rt[c], 1234u + c);
    int it = 0, restarts = 0; double true_norm = -1.0, last_true = -1.0;
    static const bool trace3 = getenv("PYLAMP_SOLVER_TRACE") != nullptr;
    // velocity-error estimate (n |r_cont| + |(M^-1 r)_vel|) / |x_vel| as in pl_solver.hip (na = 4: three velocity arrays + pressure)
    const bool use_est = etol > 0.0 && na == 4 && M;
    const double n_amp = (double)std::max(g.gn[0], std::max(g.gn[1], g.gn[2]));
    double tol = rtol, est_rec = 0.0, a_mom = 1.0;
    int est_checks = 0;
    bool resume = false, broke = false, p_fused = false;
    // one rank, Stokes vectors (4 consecutive arrays each): the fused reduction / update kernels
    const bool fused = ctx->nranks == 1 && na == 4 && M && flat3(t, s, rt) && flat3(dxb, x, p) && flat3(r, v, y) && flat3(z) &&
                       !(getenv("PYLAMP_3D_FUSED") && atoi(getenv("PYLAMP_3D_FUSED")) == 0);
    double rho = 1.0, alpha = 1.0, omega = 1.0, rho_new = 0.0, rnorm = 0.0, best = 0.0; int best_it = 0;
    st->error_estimate = 0.0;
    // the component of the residual along the deflated anchor mode has its own amplification |w| / |u| (see pl_solver.hip)
    const bool anchor_term = use_est && ctx->dfl_active && ctx->dfl_yAw != 0.0;
    auto anchor_part = [&](double* const* res, double xx, double& out) -> int {
        out = 0.0;
        if (!anchor_term || !(xx > 0.0)) return 0;
        const double* a1[1] = {ctx->dfl_y}; const double* b1[1] = {res[3]}; double o[1];
        P3_TRY(dots3(ctx, vol, 1, a1, b1, o));
        out = std::fabs(o[0] / ctx->dfl_yAw) * std::sqrt(ctx->dfl_wvel2 / xx);
        return 0;
    };
    auto vel_norm2 = [&](double& xx) -> int {        // |(x0 + dx)_vel|^2, through t as scratch
        double* const* src = dx;
        if (dx != x) { for (int c = 0; c < 3; c++) hipLaunchKernelGGL(k3_axpby, g1(vol), dim3(256), 0, ctx->stream, vol, t[c], 1.0, (const double*)x[c], 1.0, (const double*)dx[c]); src = t; }
        double* const* aa[1] = {src}; double o[1];
        P3_TRY(vdots(ctx, vol, 3, 1, aa, aa, o));
        xx = o[0];
        return 0;
    };
    for (;;) {
        if (!resume) {
            zero(p); zero(v);
            rho = alpha = omega = 1.0; p_fused = false;
            { double* const* aa[2] = {rt, r}; double* const* bb[2] = {r, r}; P3_TRY(vdots(ctx, vol, na, 2, aa, bb, d)); }
            rho_new = d[0]; rnorm = std::sqrt(d[1]);
            broke = false;
            best = rnorm; best_it = it;
        }
        resume = false;
        while (it < maxit && (rnorm > tol * bnorm || (use_est && est_rec > 0.7 * etol))) {
            it++;
            if (!(std::fabs(rho_new) > 0.0) || !std::isfinite(rho_new)) { broke = true; break; }
            const double beta = (rho_new / rho) * (alpha / omega);
            if (p_fused) p_fused = false;                   // (written by k3_xrp_update of the previous iteration)
            else if (flat3(p, r, v)) hipLaunchKernelGGL(k3_p_update, g1(na * vol), dim3(256), 0, ctx->stream, na * vol, p[0], (const double*)r[0], (const double*)v[0], beta, omega);
            else for (int c = 0; c < na; c++) hipLaunchKernelGGL(k3_p_update, g1(vol), dim3(256), 0, ctx->stream, vol, p[c], (const double*)r[c], (const double*)v[c], beta, omega);
            double* const* yv = p;
            if (M) { P3_TRY((*M)(p, y)); yv = y; }
            P3_TRY(A(yv, v));
            { double* const* aa[1] = {rt}; double* const* bb[1] = {v}; P3_TRY(vdots(ctx, vol, na, 1, aa, bb, d)); }
            if (!(std::fabs(d[0]) > 0.0) || !std::isfinite(d[0])) { broke = true; break; }
            alpha = rho_new / d[0];
            axpby(s, 1.0, r, -alpha, v);
            double* const* zv = s;
            if (M) { P3_TRY((*M)(s, z)); zv = z; }
            P3_TRY(A(zv, t));
            // one reduction: t.s, t.t, rt.s, rt.t, s.s  ->  omega, rho' = rt.s - omega rt.t, |r|^2 = s.s - 2 omega t.s + omega^2 t.t
            double f11[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const bool late = use_est && rnorm <= 1e3 * tol * bnorm;
            if (fused) {       // ... and, near the end, everything the error estimate of the updated iterate needs, in the same pass
                Fuse11 fa{}; fa.t = t[0]; fa.s = s[0]; fa.rt = rt[0]; fa.vol = vol;
                if (late) { fa.x = x[0]; fa.dx = (dx != x) ? dx[0] : nullptr; fa.yw = anchor_term ? ctx->dfl_y : nullptr; }
                hipLaunchKernelGGL(k3_dots11, dim3(D3_BLOCKS), dim3(256), 0, ctx->stream, fa, ctx->part);
                P3_HIP(ctx, hipMemcpyAsync(ctx->hpart, ctx->part, 11 * D3_BLOCKS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                P3_HIP(ctx, hipStreamSynchronize(ctx->stream));
                for (int q = 0; q < 11; q++) { double sm = 0.0; for (int kk = 0; kk < D3_BLOCKS; kk++) sm += ctx->hpart[11 * kk + q]; f11[q] = sm; }
                for (int q = 0; q < 5; q++) d[q] = f11[q];
            } else
            { double* const* aa[5] = {t, t, rt, rt, s}; double* const* bb[5] = {s, t, s, t, s}; P3_TRY(vdots(ctx, vol, na, 5, aa, bb, d)); }
            omega = (d[1] > 0.0) ? d[0] / d[1] : 0.0;
            if (fused) {
                const double rho_next = d[2] - omega * d[3];
                const double beta_next = (rho_next / rho_new) * (alpha / omega);
                hipLaunchKernelGGL(k3_xrp_update, g1(na * vol), dim3(256), 0, ctx->stream, na * vol, dx[0], (const double*)yv[0], (const double*)zv[0], r[0],
                                   (const double*)s[0], (const double*)t[0], p[0], (const double*)v[0], alpha, omega, beta_next);
                p_fused = std::isfinite(beta_next);
            }
            else if (flat3(dx, yv, zv) && flat3(r, s, t))
                hipLaunchKernelGGL(k3_xr_update, g1(na * vol), dim3(256), 0, ctx->stream, na * vol, dx[0], (const double*)yv[0], (const double*)zv[0], r[0],
                                   (const double*)s[0], (const double*)t[0], alpha, omega);
            else for (int c = 0; c < na; c++)
                hipLaunchKernelGGL(k3_xr_update, g1(vol), dim3(256), 0, ctx->stream, vol, dx[c], (const double*)yv[c], (const double*)zv[c], r[c],
                                   (const double*)s[c], (const double*)t[c], alpha, omega);
            rho = rho_new; rho_new = d[2] - omega * d[3];
            const double rr = d[4] - 2.0 * omega * d[0] + omega * omega * d[1];
            rnorm = rr > 0.0 ? std::sqrt(rr) : 0.0;
            if (!std::isfinite(rnorm) || !(std::fabs(omega) > 0.0)) { broke = true; break; }
            if (rnorm < best) { best = rnorm; best_it = it; }
            if (trace3) fprintf(stderr, "[pylamp3 bicgstab] it %3d  |r|/|b| %.3e\n", it, rnorm / bnorm);
            est_rec = 0.0;
            if (fused && late) {                                  // (sums of the fused reduction; |x_vel| of the iterate BEFORE this update: needed to ~10 %)
                const double rc0 = f11[7] - 2.0 * omega * f11[5] + omega * omega * f11[6], xx = f11[8];
                const double rcc = rc0 > 0.0 ? rc0 : 0.0, rm = rr - rcc > 0.0 ? rr - rcc : 0.0;
                const double ap = (anchor_term && xx > 0.0) ? std::fabs((f11[9] - omega * f11[10]) / ctx->dfl_yAw) * std::sqrt(ctx->dfl_wvel2 / xx) : 0.0;
                if (xx > 0.0) est_rec = (n_amp * std::sqrt(rcc) + a_mom * std::sqrt(rm)) / std::sqrt(xx) + ap;
            } else
            if (use_est && rnorm <= 1e3 * tol * bnorm) {          // near the end: the estimate from the recurrence residual
                double rc[1], xx = 0.0;
                { double* const* aa[1] = {r + 3}; P3_TRY(vdots(ctx, vol, 1, 1, aa, aa, rc)); }
                P3_TRY(vel_norm2(xx));
                const double rm = rr - rc[0] > 0.0 ? rr - rc[0] : 0.0;
                double ap = 0.0;
                P3_TRY(anchor_part(r, xx, ap));
                if (xx > 0.0) est_rec = (n_amp * std::sqrt(rc[0] > 0.0 ? rc[0] : 0.0) + a_mom * std::sqrt(rm)) / std::sqrt(xx) + ap;
            }
            if (it - best_it > 80 || rnorm > 1e8 * best) { broke = true; break; }
        }
        P3_TRY(A(dx, t));
        axpby(s, 1.0, r0, -1.0, t);
        { double* const* aa[1] = {s}; P3_TRY(vdots(ctx, vol, na, 1, aa, aa, d)); }
        last_true = true_norm; true_norm = std::sqrt(d[0]);
        if (true_norm <= tol * bnorm && !broke && it < maxit && use_est && est_checks < 6) {
            double rc[1], zz[1], xx = 0.0;
            { double* const* aa[1] = {s + 3}; P3_TRY(vdots(ctx, vol, 1, 1, aa, aa, rc)); }
            P3_TRY((*M)(s, z));
            { double* const* aa[1] = {z}; P3_TRY(vdots(ctx, vol, 3, 1, aa, aa, zz)); }
            P3_TRY(vel_norm2(xx));
            est_checks++;
            const double rcc = rc[0] > 0.0 ? rc[0] : 0.0, rm = true_norm * true_norm - rcc;
            double ap = 0.0;
            P3_TRY(anchor_part(s, xx, ap));
            const double est = (xx > 0.0 ? (n_amp * std::sqrt(rcc) + std::sqrt(zz[0] > 0.0 ? zz[0] : 0.0)) / std::sqrt(xx) : 0.0) + ap;
            if (rm > 0.0 && zz[0] > 0.0) a_mom = std::min(std::max(std::sqrt(zz[0] / rm), 1.0), n_amp * n_amp);
            st->error_estimate = est;
            if (trace3) fprintf(stderr, "[pylamp3 bicgstab] it %3d  velocity-error estimate %.3e (etol %.1e)\n", it, est, etol);
            if (est > etol && tol > 1e-15) {
                tol = std::min(tol, true_norm / bnorm) * std::min(0.5, 0.7 * etol / est);
                est_rec = est; resume = true;
                continue;
            }
        }
        if (true_norm <= tol * bnorm || broke || it >= maxit || restarts >= 4) break;
        if (last_true >= 0.0 && !(true_norm < 0.5 * last_true)) break;
        restarts++;
        axpby(r, 1.0, s, 0.0, s);
    }
    if (dx != x) axpby(x, 1.0, x, 1.0, dx);
    st->iterations = it; st->rel_residual = true_norm / bnorm;
    st->converged = (st->rel_residual <= rtol && !(use_est && st->error_estimate > 1.5 * etol)) ? 1 : 0;
    return 0;
}

// hydrostatic pressure guess: with v = 0 the interior z-momentum rows reduce to -2 Kc rDz_i (P[i] - P[i-1]) = b_z; integrate down
// every column, anchor cell to zero (same construction as pl_solver.hip)
// (several ranks: every block integrates from zero at its own top; coltot receives the block's column totals, and k3_hydro_add the sum
//  of the totals of the blocks above -- pl3_stokes_solve)
__global__ __launch_bounds__(256) void k3_hydro(Op3 op, double* __restrict__ P, double* __restrict__ coltot) {
    const int lk = blockIdx.x * 64 + threadIdx.x, lj = blockIdx.y * 4 + threadIdx.y;
    const G3& g = op.g;
    if (lk >= g.n[2] || lj >= g.n[1]) return;
    const int j = lj + g.o[1], k = lk + g.o[2];
    const double* r = op.rho;
    const int jn = (j + 1 < g.gn[1]) ? 1 : 0, kn = (k + 1 < g.gn[2]) ? 1 : 0;
    double acc = 0.0;
    for (int li = 0; li < g.n[0]; li++) {
        const int i = li + g.o[0];
        const long long c = i3(g, li, lj, lk);
        if (i >= 1 && i <= g.gn[0] - 2)
            acc += 0.25 * ((r[c] + r[c + jn * g.s[1]]) + (r[c + kn * g.s[2]] + r[c + jn * g.s[1] + kn * g.s[2]])) * op.grav[0] / (2.0 * op.Kc * TB(g.rD[0], i));
        P[c] = (i >= g.gn[0] - 1 || j >= g.gn[1] - 1 || k >= g.gn[2] - 1) ? 0.0 : acc;
    }
    if (coltot) coltot[(long long)lj * g.n[2] + lk] = acc;
}
__global__ __launch_bounds__(256) void k3_hydro_add(G3 g, double* __restrict__ P, const double* __restrict__ above) {
    K3_PROLOGUE(g)
    if (i < g.gn[0] - 1 && j < g.gn[1] - 1 && k < g.gn[2] - 1) P[c] += above[(long long)lj * g.n[2] + lk];
}
__global__ __launch_bounds__(256) void k3_shift(G3 g, double* __restrict__ P, const double* __restrict__ anchor_val) {
    K3_PROLOGUE(g)
    if (i < g.gn[0] - 1 && j < g.gn[1] - 1 && k < g.gn[2] - 1) P[c] -= *anchor_val;
}

extern "C" int pl3_stokes_solve(pl3_ctx* ctx, const double* rhs, double* x, int use_x0, double rtol, int maxit, pl_solve_stats* stats) {
    if (!ctx->op_ready) return p3_fail(ctx, "stokes operator not set");
    // x == NULL: device-resident solve -- the solution stays in the context (pl3_get_solution fetches it), use_x0 starts from the
    // solution the context holds from its previous solve
    if (!x && use_x0 && !ctx->have_x) return p3_fail(ctx, "pl3_stokes_solve: no resident solution to start from");
    P3_HIP(ctx, hipSetDevice(ctx->device));
    P3_TRY(need_vecs(ctx, 14));
    ctx->halo_failed = false;
    if (rtol <= 0) rtol = 1e-7;
    if (maxit <= 0) maxit = 600;
    const G3& g = ctx->geom.d;
    const long long vol = g.vol;
    P3_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    P3_TRY(build_levels3(ctx));
    double* const* B = ctx->vec[10]; double* const* X = ctx->vec[11]; double* const* XH = ctx->vec[12];
    if (rhs) { P3_TRY(upload3(ctx, rhs, 4, B)); hipLaunchKernelGGL(k3_scale_rows, grid3(g), dim3(64, 4), 0, ctx->stream, ctx->op, wv4(B)); }
    else hipLaunchKernelGGL(k3_rhs, grid3(g), dim3(64, 4), 0, ctx->stream, ctx->op, wv4(B), 1);
    int napply = 0, nprec = 0;
    bool defl_active = false; double yAw = 1.0;
    double* const* W = ctx->vec[13];
    Op3Fn A = [&](double* const* in, double* const* out) -> int {
        P3_TRY(halo3(ctx, g, in, 4));
        launch_apply3<true>(ctx, ctx->op, in, out);
        napply++; return 0;
    };
    Op3Fn M = [&](double* const* in, double* const* out) -> int {
        Lev3* L0 = ctx->levels[0];
        P3_TRY(halo3_1(ctx, g, in[3]));             // S^-1 r_p of the neighbour cell (A_vp z_p) reads the pressure residual one node beyond the block
        hipLaunchKernelGGL(k3_stage1, grid3(g), dim3(64, 4), 0, ctx->stream, ctx->op, cv4(in), out[3], wv3(L0->f));
        int ob = 0;
        if (ctx->levels.size() > 1) {           // the post-smoothing sweep of the finest level writes the velocities into out[0..2] itself
            const W3 fo = wv3(out);
            vcycle3(ctx, 0, L0->f, ob, &fo);
        } else {
            vcycle3(ctx, 0, L0->f, ob);
            for (int q = 0; q < 3; q++) P3_HIP(ctx, hipMemcpyAsync(out[q], L0->v[ob][q], (size_t)vol * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        }
        nprec++;
        if (ctx->halo_failed) return 1;      // (an exchange inside the V-cycle; ctx->err says which)
        if (defl_active) {                  // z += w yw.(r - A z) / yw.(A w); the continuity rows of A z come from z's velocities alone
            const double* a1[2] = {ctx->dfl_y, ctx->dfl_y}; const double* b1[2] = {in[3], ctx->dfl_t}; double o[2];
            P3_TRY(halo3(ctx, g, out, 3));
            hipLaunchKernelGGL(k3_cont_rows, grid3(g), dim3(64, 4), 0, ctx->stream, ctx->op, cv3(out), ctx->dfl_t);
            P3_TRY(dots3(ctx, vol, 2, a1, b1, o));
            const double coef = (o[0] - o[1]) / yAw;
            for (int c = 0; c < 4; c++) hipLaunchKernelGGL(k3_axpby, g1(vol), dim3(256), 0, ctx->stream, vol, out[c], 1.0, (const double*)out[c], coef, (const double*)W[c]);
        }
        return 0;
    };
    // hydrostatic start and the dynamic-load reference norm
    double ref = 0.0;
    {
        for (int q = 0; q < 3; q++) P3_HIP(ctx, hipMemsetAsync(XH[q], 0, (size_t)vol * sizeof(double), ctx->stream));
        if (ctx->nranks == 1) {
            hipLaunchKernelGGL(k3_hydro, dim3((g.n[2] + 63) / 64, (g.n[1] + 3) / 4), dim3(64, 4), 0, ctx->stream, ctx->op, XH[3], (double*)nullptr);
            P3_HIP(ctx, hipMemcpyAsync(ctx->part, XH[3] + i3(g, 3, 2, 2), sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));   // anchor value
        } else {
            // every block integrated from zero at its top: add the column totals of the blocks above (same block column), then the anchor
            // value -- one table [Pz][nx][ny] and one scalar summed over the ranks on the host (once per solve)
            const long long cols = (long long)g.n[1] * g.n[2], gcols = (long long)g.gn[1] * g.gn[2];
            double* coltot = ctx->vec[5][0];                    // scratch (a work vector of the iteration, free before it starts)
            hipLaunchKernelGGL(k3_hydro, dim3((g.n[2] + 63) / 64, (g.n[1] + 3) / 4), dim3(64, 4), 0, ctx->stream, ctx->op, XH[3], coltot);
            std::vector<double> mine((size_t)cols), tab((size_t)ctx->P[0] * gcols, 0.0);
            P3_HIP(ctx, hipMemcpyAsync(mine.data(), coltot, (size_t)cols * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            P3_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (int jj = 0; jj < g.n[1]; jj++) for (int kk = 0; kk < g.n[2]; kk++)
                tab[(size_t)ctx->pc[0] * gcols + (size_t)(g.o[1] + jj) * g.gn[2] + (g.o[2] + kk)] = mine[(size_t)jj * g.n[2] + kk];
            P3_TRY(allreduce3(ctx, tab.data(), (int)tab.size(), 0));
            for (int jj = 0; jj < g.n[1]; jj++) for (int kk = 0; kk < g.n[2]; kk++) {
                double a = 0.0;
                for (int q = 0; q < ctx->pc[0]; q++) a += tab[(size_t)q * gcols + (size_t)(g.o[1] + jj) * g.gn[2] + (g.o[2] + kk)];
                mine[(size_t)jj * g.n[2] + kk] = a;
            }
            P3_HIP(ctx, hipMemcpyAsync(coltot, mine.data(), (size_t)cols * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k3_hydro_add, grid3(g), dim3(64, 4), 0, ctx->stream, g, XH[3], (const double*)coltot);
            double av[1] = {0.0};
            const int ai = ctx->op.anchor[0] - g.o[0], aj = ctx->op.anchor[1] - g.o[1], ak = ctx->op.anchor[2] - g.o[2];
            if (ai >= 0 && ai < g.n[0] && aj >= 0 && aj < g.n[1] && ak >= 0 && ak < g.n[2])
                P3_HIP(ctx, hipMemcpyAsync(av, XH[3] + i3(g, ai, aj, ak), sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            P3_HIP(ctx, hipStreamSynchronize(ctx->stream));
            P3_TRY(allreduce3(ctx, av, 1, 0));
            P3_HIP(ctx, hipMemcpyAsync(ctx->part, av, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            P3_HIP(ctx, hipStreamSynchronize(ctx->stream));
            P3_HIP(ctx, hipMemsetAsync(coltot, 0, (size_t)cols * sizeof(double), ctx->stream));
        }
        hipLaunchKernelGGL(k3_shift, grid3(g), dim3(64, 4), 0, ctx->stream, g, XH[3], (const double*)ctx->part);
        P3_TRY(A(XH, ctx->vec[5]));
        for (int c = 0; c < 4; c++) hipLaunchKernelGGL(k3_axpby, g1(vol), dim3(256), 0, ctx->stream, vol, ctx->vec[4][c], 1.0, (const double*)B[c], -1.0, (const double*)ctx->vec[5][c]);
        double d[2];
        double* const* aa[2] = {ctx->vec[4], B}; P3_TRY(vdots(ctx, vol, 4, 2, aa, aa, d));
        ref = std::sqrt(d[0]);
        if (!(ref > 1e-9 * std::sqrt(d[1]))) ref = 0.0;
    }
    if (use_x0) { if (x) P3_TRY(upload3(ctx, x, 4, X)); }
    else for (int c = 0; c < 4; c++) P3_HIP(ctx, hipMemcpyAsync(X[c], XH[c], (size_t)vol * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    P3_TRY(halo3(ctx, g, X, 3));
    hipLaunchKernelGGL(k3_close, grid3(g), dim3(64, 4), 0, ctx->stream, ctx->op, wv3(X), cv3(B));
    double* const* w[10] = {ctx->vec[0], ctx->vec[1], ctx->vec[2], ctx->vec[3], ctx->vec[4], ctx->vec[5], ctx->vec[6], ctx->vec[7], ctx->vec[8], ctx->vec[9]};
    static const bool defl_on = !(getenv("PYLAMP_DEFLATE") && atoi(getenv("PYLAMP_DEFLATE")) == 0);
    if (const char* e = getenv("PYLAMP_STOKES_ETOL")) { const double v = atof(e); if (v >= 0.0) ctx->etol = v; }
    if (defl_on && ctx->levels.size() > 1) {
        // the deflation vector w = A^-1 u (pl_solver.hip, "deflation of the pressure-anchor mode"), kept in vec[13] between solves
        if (!ctx->dfl_y) { P3_TRY(dmal(ctx, &ctx->dfl_y, vol)); P3_TRY(dmal(ctx, &ctx->dfl_t, vol)); ctx->dfl_valid = false; }
        double* const* U = XH;               // the hydrostatic vector is not needed any more
        hipLaunchKernelGGL(k3_defl_setup, grid3(g), dim3(64, 4), 0, ctx->stream, ctx->op, wv4(U), ctx->dfl_y);
        auto denominator = [&]() -> int {
            const double* a1[1] = {ctx->dfl_y}; const double* b1[1] = {ctx->dfl_t}; double o[1];
            P3_TRY(halo3(ctx, g, W, 3));
            hipLaunchKernelGGL(k3_cont_rows, grid3(g), dim3(64, 4), 0, ctx->stream, ctx->op, cv3(W), ctx->dfl_t);
            P3_TRY(dots3(ctx, vol, 1, a1, b1, o));
            yAw = o[0];
            return 0;
        };
        double q = 1.0;
        if (ctx->dfl_valid) {               // quality of the kept vector for this operator: ||u - A w|| / ||u||
            double dq[2];
            P3_TRY(A(W, ctx->vec[5]));
            for (int c = 0; c < 4; c++) hipLaunchKernelGGL(k3_axpby, g1(vol), dim3(256), 0, ctx->stream, vol, ctx->vec[4][c], 1.0, (const double*)U[c], -1.0, (const double*)ctx->vec[5][c]);
            double* const* aa[2] = {ctx->vec[4], U}; P3_TRY(vdots(ctx, vol, 4, 2, aa, aa, dq));
            q = (dq[1] > 0.0 && std::isfinite(dq[0])) ? std::sqrt(dq[0] / dq[1]) : 1.0;
        }
        const bool reuse = ctx->dfl_valid && q < 0.5;
        if (reuse) { P3_TRY(denominator()); defl_active = yAw != 0.0 && std::isfinite(yAw); }
        if (!reuse || q > 0.1) {
            if (!reuse) for (int c = 0; c < 4; c++) P3_HIP(ctx, hipMemsetAsync(W[c], 0, (size_t)vol * sizeof(double), ctx->stream));
            Stats3 st2{};
            P3_TRY(bicgstab3(ctx, vol, 4, A, &M, U, W, reuse, 1e-3, 80, w, g, &st2, 0.0));
            ctx->dfl_valid = st2.rel_residual < 0.05 && std::isfinite(st2.rel_residual);
            defl_active = false;
            if (ctx->dfl_valid) { P3_TRY(denominator()); defl_active = yAw != 0.0 && std::isfinite(yAw); }
        }
    }
    ctx->dfl_active = defl_active; ctx->dfl_yAw = yAw; ctx->dfl_wvel2 = 0.0;
    if (defl_active) { double* const* aa[1] = {W}; double o[1]; P3_TRY(vdots(ctx, vol, 3, 1, aa, aa, o)); ctx->dfl_wvel2 = o[0]; }
    Stats3 st{};
    P3_TRY(bicgstab3(ctx, vol, 4, A, &M, B, X, true, rtol, maxit, w, g, &st, ref, ctx->etol));
    ctx->dfl_active = false;
    P3_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    P3_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0; P3_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->have_x = true;
    if (x) P3_TRY(download3(ctx, X, 4, x));
    if (stats) { stats->iterations = st.iterations; stats->converged = st.converged; stats->rel_residual = st.rel_residual; stats->solve_ms = ms;
                 stats->operator_applies = napply; stats->precond_applies = nprec; stats->used_direct = 0; stats->error_estimate = st.error_estimate; }
    return 0;
}
extern "C" int pl3_stokes_mg_info(pl3_ctx* ctx, int* nlevels, double* lmax, int max_levels) {
    if (nlevels) *nlevels = (int)ctx->levels.size();
    for (int l = 0; lmax && l < max_levels && l < (int)ctx->levels.size(); l++) lmax[l] = ctx->levels[l]->lmax;
    return 0;
}

// ---- heat ----------------------------------------------------------------------------------------------------------------------
extern "C" int pl3_heat_set_coeffs(pl3_ctx* ctx, const double* zmp, const double* xmp, const double* ymp, const double* T, const double* kz,
                                   const double* kx, const double* ky, const double* cp, const double* rho, const double* H, const int bc[6],
                                   const double bcvalue[6], double tstep) {
    if (!zmp || !xmp || !ymp || !T || !kz || !kx || !ky || !cp || !rho || !H || !bc || !bcvalue) return p3_fail(ctx, "pl3_heat_set_coeffs: NULL argument");
    for (int w = 0; w < 6; w++) if (bc[w] != PL_BC_FIXTEMP && bc[w] != PL_BC_FIXFLOW) return p3_fail(ctx, "heat: boundary condition must be FIXTEMP or FIXFLOW");
    P3_HIP(ctx, hipSetDevice(ctx->device));
    const G3& g = ctx->geom.d;
    for (double** q : {&ctx->hk[0], &ctx->hk[1], &ctx->hk[2], &ctx->hT, &ctx->hH, &ctx->hcdt, &ctx->hrho, &ctx->hcp}) if (!*q) P3_TRY(dmal(ctx, q, g.vol));
    const double* src[7] = {kz, kx, ky, T, H, rho, cp}; double* dst[7] = {ctx->hk[0], ctx->hk[1], ctx->hk[2], ctx->hT, ctx->hH, ctx->hrho, ctx->hcp};
    for (int q = 0; q < 7; q++) { double* d1[1] = {dst[q]}; P3_TRY(upload3(ctx, src[q], 1, d1)); }
    const double* mp[3] = {zmp, xmp, ymp};
    size_t len[3], tot = 0;
    for (int a = 0; a < 3; a++) { len[a] = (size_t)g.gn[a] + 2 * PL_TOFF + 2; tot += len[a]; }
    std::vector<double> t(tot, 0.0);
    size_t off = 0, offs[3];
    for (int a = 0; a < 3; a++) { offs[a] = off; for (int i = 1; i < g.gn[a]; i++) t[off + i + PL_TOFF] = 1.0 / (mp[a][i] - mp[a][i - 1]); off += len[a]; }
    if (!ctx->htab) P3_HIP(ctx, hipMalloc((void**)&ctx->htab, tot * sizeof(double)));
    P3_HIP(ctx, hipMemcpy(ctx->htab, t.data(), tot * sizeof(double), hipMemcpyHostToDevice));
    if (!ctx->hbcv) P3_HIP(ctx, hipMalloc((void**)&ctx->hbcv, 6 * sizeof(double)));
    P3_HIP(ctx, hipMemcpy(ctx->hbcv, bcvalue, 6 * sizeof(double), hipMemcpyHostToDevice));
    Heat3& op = ctx->hop;
    op.g = g; for (int a = 0; a < 3; a++) { op.kk[a] = ctx->hk[a]; op.rdb[a] = ctx->htab + offs[a]; }
    op.cdt = ctx->hcdt; for (int w = 0; w < 6; w++) op.bc[w] = bc[w];
    hipLaunchKernelGGL(k3_heat_coef, grid3(g), dim3(64, 4), 0, ctx->stream, g, (const double*)ctx->hrho, (const double*)ctx->hcp, tstep, ctx->hcdt);
    P3_HIP(ctx, hipGetLastError());
    ctx->hop_ready = true;
    return 0;
}
static int need_hvecs(pl3_ctx* ctx) {
    for (int v = 0; v < 12; v++) if (!ctx->hvec[v]) P3_TRY(dmal(ctx, &ctx->hvec[v], ctx->geom.d.vol));
    if (!ctx->part) { P3_TRY(dmal(ctx, &ctx->part, 11 * D3_BLOCKS)); P3_HIP(ctx, hipHostMalloc((void**)&ctx->hpart, 11 * D3_BLOCKS * sizeof(double))); }
    return 0;
}
extern "C" int pl3_heat_apply(pl3_ctx* ctx, const double* x, double* y) {
    if (!ctx->hop_ready) return p3_fail(ctx, "heat operator not set");
    P3_HIP(ctx, hipSetDevice(ctx->device));
    P3_TRY(need_hvecs(ctx));
    double* a[1] = {ctx->hvec[0]}; double* b[1] = {ctx->hvec[1]};
    P3_TRY(upload3(ctx, x, 1, a));
    hipLaunchKernelGGL(k3_heat_apply<false>, grid3(ctx->hop.g), dim3(64, 4), 0, ctx->stream, ctx->hop, (const double*)a[0], b[0]);
    P3_HIP(ctx, hipGetLastError());
    return download3(ctx, b, 1, y);
}
extern "C" int pl3_heat_rhs(pl3_ctx* ctx, double* rhs) {
    if (!ctx->hop_ready) return p3_fail(ctx, "heat operator not set");
    P3_HIP(ctx, hipSetDevice(ctx->device));
    P3_TRY(need_hvecs(ctx));
    double* b[1] = {ctx->hvec[1]};
    hipLaunchKernelGGL(k3_heat_rhs, grid3(ctx->hop.g), dim3(64, 4), 0, ctx->stream, ctx->hop, (const double*)ctx->hT, (const double*)ctx->hH, (const double*)ctx->hbcv, b[0], 0);
    P3_HIP(ctx, hipGetLastError());
    return download3(ctx, b, 1, rhs);
}
extern "C" int pl3_heat_solve(pl3_ctx* ctx, const double* rhs, double* x, double rtol, int maxit, pl_solve_stats* stats) {
    if (!ctx->hop_ready) return p3_fail(ctx, "heat operator not set");
    P3_HIP(ctx, hipSetDevice(ctx->device));       // x == NULL: the solution stays on the device (pl3_get_solution)
    P3_TRY(need_hvecs(ctx));
    if (rtol <= 0) rtol = 1e-12;
    if (maxit <= 0) maxit = 2000;
    const G3& g = ctx->geom.d;
    P3_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    double* B[1] = {ctx->hvec[10]}; double* X[1] = {ctx->hvec[11]};
    if (rhs) return p3_fail(ctx, "pl3_heat_solve: rhs must be NULL -- the system is solved for the operator's own right-hand side (pl3_heat_rhs)");
    hipLaunchKernelGGL(k3_heat_rhs, grid3(g), dim3(64, 4), 0, ctx->stream, ctx->hop, (const double*)ctx->hT, (const double*)ctx->hH, (const double*)ctx->hbcv, B[0], 1);
    int napply = 0;
    Op3Fn A = [&](double* const* in, double* const* out) -> int {
        P3_TRY(halo3(ctx, g, in, 1));
        hipLaunchKernelGGL(k3_heat_apply<true>, grid3(g), dim3(64, 4), 0, ctx->stream, ctx->hop, (const double*)in[0], out[0]);
        napply++; return 0;
    };
    double* wv[10][1]; double* const* w[10];
    for (int q = 0; q < 10; q++) { wv[q][0] = ctx->hvec[q]; w[q] = wv[q]; }
    Stats3 st{};
    P3_TRY(bicgstab3(ctx, g.vol, 1, A, nullptr, B, X, false, rtol, maxit, w, g, &st, 0.0));
    P3_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    P3_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0; P3_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->have_T = true;
    if (x) P3_TRY(download3(ctx, X, 1, x));
    if (stats) { stats->iterations = st.iterations; stats->converged = st.converged; stats->rel_residual = st.rel_residual; stats->solve_ms = ms;
                 stats->operator_applies = napply; stats->precond_applies = 0; }
    return 0;
}

// the solution of the last device-resident solve: which = 0 Stokes (nz, nx, ny, 4), 1 heat (nz, nx, ny)
extern "C" int pl3_get_solution(pl3_ctx* ctx, int which, double* out) {
    if (!out) return p3_fail(ctx, "pl3_get_solution: NULL argument");
    P3_HIP(ctx, hipSetDevice(ctx->device));
    if (which == 0) {
        if (!ctx->have_x) return p3_fail(ctx, "pl3_get_solution: no Stokes solution yet");
        return download3(ctx, ctx->vec[11], 4, out);
    }
    if (!ctx->have_T) return p3_fail(ctx, "pl3_get_solution: no heat solution yet");
    double* X[1] = {ctx->hvec[11]};
    return download3(ctx, X, 1, out);
}
