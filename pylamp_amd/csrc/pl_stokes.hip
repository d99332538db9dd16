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
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>

#define TB(tab, k) (tab)[(k) + PL_TOFF]

// ---------------------------------------------------------------------------------------------------
// Two-columns-per-lane variant: every plane row is read with 16-byte (double2) loads, the j-1 / j+1
// neighbours come from the adjacent lanes (ds_bpermute), only the two edge lanes issue an extra
// 8-byte load.  Why: with 8-byte loads the texture-address unit was busy 84 % of the kernel time
// (PMC TA_TA_BUSY, profiles/r01_pmc_stalls.csv): the L1 path serves 4 lanes per cycle whatever the
// width, so double2 halves the address cycles per byte and there are 14 instead of 34 loads per wave.
// A wave covers 128 columns of one row; rows/columns outside the block are never dereferenced past
// the ring (planes carry a ring row above/below and >= 1 pad column each side).
// all values one node needs, as scalars
struct StokesVals {
    double vz_c, vz_w, vz_e, vz_n, vz_s, vz_nw;      // n = row i+1, s = row i-1, nw = (i+1, j-1)
    double vx_c, vx_w, vx_e, vx_n, vx_s, vx_se;      // se = (i-1, j+1)
    double p_c, p_w, p_e, p_s;
    double en_c, en_w, en_s;
    double es_c, es_e, es_n;
    double rdx_j, rdx_m, rDx_j, rDx_p;
};

// The memory-only twin of this kernel runs in 58 us against 80 us (2049^2): the arithmetic is NOT free.  The
// row scales of the scaled operator are therefore a template switch (the plain operator performs no division
// at all) and use v_rcp_f64 + two Newton steps instead of the ~15-instruction IEEE division sequence.
template <bool SCALED>
__device__ inline void stokes_node_vals(const PlStokesOp& op, int i, int j, int c, const StokesVals& q,
                                        const double* __restrict__ x, double& oz, double& ox, double& op_) {
    const PlGeom& g = op.g;
    const int nz = g.nz, nx = g.nx, p = g.pitch;
    const double Kc = op.Kc;
    const double iKc = op.iKc;
    double sz = iKc, sx = iKc, sp = iKc;
    double yz;
    if (j == nx - 1 || i == 0 || i == nz - 1) yz = Kc * q.vz_c;
    else if (j == 0) yz = Kc * (q.vz_c - q.vz_e);
    else if (j == nx - 2) yz = Kc * (q.vz_c - q.vz_w);
    else {
        // factored so that the row-only products are shared by the lane's two columns and each viscosity is
        // multiplied once (the arithmetic of this kernel is ~1/3 of its run time: FP64 issues at half rate)
        const double rdz_i = TB(g.rdz, i), rdz_m = TB(g.rdz, i - 1), rDz_i = TB(g.rDz, i);
        const double Az = 4.0 * rdz_i * rDz_i, Azm = 4.0 * rdz_m * rDz_i, Pz = 2.0 * Kc * rDz_i;     // row constants
        const double k2 = 2.0 * q.rdx_j;                                                              // column constant
        const double tE = q.es_e * k2, tW = q.es_c * k2;
        const double cN = q.en_c * Az, cS = q.en_s * Azm, cE = tE * q.rDx_p, cW = tW * q.rDx_j;
        const double xE = tE * rDz_i, xW = tW * rDz_i;
        yz = cN * (q.vz_n - q.vz_c) - cS * (q.vz_c - q.vz_s) + cE * (q.vz_e - q.vz_c) - cW * (q.vz_c - q.vz_w) +
             xE * (q.vx_e - q.vx_se) - xW * (q.vx_c - q.vx_s) - Pz * (q.p_c - q.p_s);
        if (SCALED) sz = pl_rcp(cN + cS + cE + cW);
        if (op.surfstab) {
            const double* __restrict__ r = op.rho;
            yz += op.ss * op.gz * 0.5 * ((r[c + 1] + r[c + p + 1] - r[c - 1] - r[c + p - 1]) * q.rDx_j * q.vx_c +
                                         (r[c + p] + r[c + p + 1] - r[c - p] - r[c - p + 1]) * rDz_i * q.vz_c);
        }
    }
    double yx;
    if (i == nz - 1 || j == 0 || j == nx - 1) yx = Kc * q.vx_c;
    else if (i == 0) {
        if (op.bc_z0 == PL_BC_FREESLIP) yx = Kc * (q.vx_c - q.vx_n);
        else yx = Kc * ((-TB(g.rDz, 1) - TB(g.rdz, 0)) * q.vx_c + TB(g.rDz, 1) * q.vx_n);
    } else if (i == nz - 2) {
        if (op.bc_zL == PL_BC_FREESLIP) yx = Kc * (q.vx_c - q.vx_s);
        else yx = Kc * ((TB(g.rDz, nz - 2) + TB(g.rdz, nz - 2)) * q.vx_c - TB(g.rDz, nz - 2) * q.vx_s);
    } else {
        const double rdz_i = TB(g.rdz, i), rDz_i = TB(g.rDz, i), rDz_p = TB(g.rDz, i + 1);
        const double r2 = 2.0 * rdz_i, twoKc = 2.0 * Kc;                                             // row constants
        const double B4 = 4.0 * q.rDx_j;                                                              // column constants
        const double Bx = B4 * q.rdx_j, Bxm = B4 * q.rdx_m, Px = twoKc * q.rDx_j;
        const double uN = q.es_n * r2, uS = q.es_c * r2;
        const double cE = q.en_c * Bx, cW = q.en_w * Bxm, cN = uN * rDz_p, cS = uS * rDz_i;
        const double zN = uN * q.rDx_j, zS = uS * q.rDx_j;
        yx = cE * (q.vx_e - q.vx_c) - cW * (q.vx_c - q.vx_w) + cN * (q.vx_n - q.vx_c) - cS * (q.vx_c - q.vx_s) +
             zN * (q.vz_n - q.vz_nw) - zS * (q.vz_c - q.vz_w) - Px * (q.p_c - q.p_w);
        if (SCALED) sx = pl_rcp(cE + cW + cN + cS);
        if (op.surfstab && op.gx != 0.0) {
            const double* __restrict__ r = op.rho;
            yx += op.ss * op.gx * 0.5 * ((r[c + 1] + r[c + p + 1] - r[c - 1] - r[c + p - 1]) * q.rDx_j * q.vx_c +
                                         (r[c + p] + r[c + p + 1] - r[c - p] - r[c - p + 1]) * rDz_i * q.vz_c);
        }
    }
    double yp;
    if (i == nz - 1 || j == nx - 1 || (i == op.anchor_i && j == op.anchor_j)) yp = Kc * q.p_c;
    else if ((i == 0 || i == nz - 2) && j == 0) { yp = op.Kb * (q.p_e - q.p_c); if (SCALED) sp = pl_rcp(op.Kb); }
    else if ((i == 0 || i == nz - 2) && j == nx - 2) { yp = op.Kb * (q.p_w - q.p_c); if (SCALED) sp = pl_rcp(op.Kb); }
    else {
        yp = Kc * ((q.vx_e - q.vx_c) * q.rdx_j + (q.vz_n - q.vz_c) * TB(g.rdz, i));
        if (SCALED) sp = iKc * pl_rcp(q.rdx_j + TB(g.rdz, i));
    }
    (void)x;
    oz = SCALED ? yz * sz : yz; ox = SCALED ? yx * sx : yx; op_ = SCALED ? yp * sp : yp;
}

// Straight-line twin of stokes_node_vals for a node whose three rows are all of the interior class
// (1 <= i <= nz-3, 1 <= j <= nx-3, not the anchor, no stabilisation): no classification, no exec-mask regions.
struct StokesRowK { double Az, Azm, Pz, rDz_i, rDz_p, r2, twoKc, rdz_i, Kc, iKc; };
template <bool SCALED>
__device__ inline void stokes_interior_vals(const StokesRowK& k, const StokesVals& q, double& oz, double& ox, double& op_) {
    const double dzn = q.vz_n - q.vz_c, dxe = q.vx_e - q.vx_c;
    const double k2 = 2.0 * q.rdx_j;
    const double tE = q.es_e * k2, tW = q.es_c * k2;
    const double cN = q.en_c * k.Az, cS = q.en_s * k.Azm, cE = tE * q.rDx_p, cW = tW * q.rDx_j;
    const double yz = cN * dzn - cS * (q.vz_c - q.vz_s) + cE * (q.vz_e - q.vz_c) - cW * (q.vz_c - q.vz_w) +
                      (tE * k.rDz_i) * (q.vx_e - q.vx_se) - (tW * k.rDz_i) * (q.vx_c - q.vx_s) - k.Pz * (q.p_c - q.p_s);
    const double B4 = 4.0 * q.rDx_j;
    const double uN = q.es_n * k.r2, uS = q.es_c * k.r2;
    const double dE = q.en_c * (B4 * q.rdx_j), dW = q.en_w * (B4 * q.rdx_m), dN = uN * k.rDz_p, dS = uS * k.rDz_i;
    const double yx = dE * dxe - dW * (q.vx_c - q.vx_w) + dN * (q.vx_n - q.vx_c) - dS * (q.vx_c - q.vx_s) +
                      (uN * q.rDx_j) * (q.vz_n - q.vz_nw) - (uS * q.rDx_j) * (q.vz_c - q.vz_w) -
                      (k.twoKc * q.rDx_j) * (q.p_c - q.p_w);
    const double yp = k.Kc * (dxe * q.rdx_j + dzn * k.rdz_i);
    if (SCALED) {
        oz = yz * pl_rcp(cN + cS + cE + cW);
        ox = yx * pl_rcp(dE + dW + dN + dS);
        op_ = yp * (k.iKc * pl_rcp(q.rdx_j + k.rdz_i));
    } else { oz = yz; ox = yx; op_ = yp; }
}

// Both columns of one lane from the twelve loaded rows, then the (double2) stores.
struct StokesRows { Row2 vz_s, vz_i, vz_n, vx_s, vx_i, vx_n, p_s, p_i, en_s, en_i, es_i, es_n, t_rdx, t_rDx; };
template <bool SCALED>
__device__ inline void stokes_compute_store(const PlStokesOp& op, const StokesRows& R, int i, int jw, int lj0, int c,
                                            const double* __restrict__ x, double* __restrict__ y,
                                            const double* __restrict__ add, const double* __restrict__ coef) {
    const PlGeom& g = op.g;
    const Row2 &vz_s = R.vz_s, &vz_i = R.vz_i, &vz_n = R.vz_n, &vx_s = R.vx_s, &vx_i = R.vx_i, &vx_n = R.vx_n;
    const Row2 &p_s = R.p_s, &p_i = R.p_i, &en_s = R.en_s, &en_i = R.en_i, &es_i = R.es_i, &es_n = R.es_n;
    const double rdx_a = R.t_rdx.v.x, rdx_b = R.t_rdx.v.y, rdx_m = R.t_rdx.w;
    const double rDx_a = R.t_rDx.v.x, rDx_b = R.t_rDx.v.y, rDx_pp = R.t_rDx.e;
    const int j0 = g.gj0 + lj0;
    StokesVals q;
    double oz[2], ox[2], opv[2];
    const bool colB = (lj0 + 1) < g.lnx;
    // wave-uniform: every node of this wave is of the interior class in all three equations
    const bool fast = !op.surfstab && i >= 1 && i <= g.nz - 3 && jw >= 1 && jw + 127 <= g.nx - 3 &&
                      !(i == op.anchor_i && op.anchor_j >= jw && op.anchor_j < jw + 128);
    StokesRowK rk;
    if (fast) {
        const double rdz_i = TB(g.rdz, i), rdz_m = TB(g.rdz, i - 1), rDz_i = TB(g.rDz, i);
        rk.Az = 4.0 * rdz_i * rDz_i; rk.Azm = 4.0 * rdz_m * rDz_i; rk.Pz = 2.0 * op.Kc * rDz_i; rk.rDz_i = rDz_i;
        rk.rDz_p = TB(g.rDz, i + 1); rk.r2 = 2.0 * rdz_i; rk.twoKc = 2.0 * op.Kc; rk.rdz_i = rdz_i; rk.Kc = op.Kc; rk.iKc = op.iKc;
    }
    // column A (lj0)
    q.vz_c = vz_i.v.x; q.vz_w = vz_i.w; q.vz_e = vz_i.v.y; q.vz_n = vz_n.v.x; q.vz_s = vz_s.v.x; q.vz_nw = vz_n.w;
    q.vx_c = vx_i.v.x; q.vx_w = vx_i.w; q.vx_e = vx_i.v.y; q.vx_n = vx_n.v.x; q.vx_s = vx_s.v.x; q.vx_se = vx_s.v.y;
    q.p_c = p_i.v.x; q.p_w = p_i.w; q.p_e = p_i.v.y; q.p_s = p_s.v.x;
    q.en_c = en_i.v.x; q.en_w = en_i.w; q.en_s = en_s.v.x;
    q.es_c = es_i.v.x; q.es_e = es_i.v.y; q.es_n = es_n.v.x;
    q.rdx_j = rdx_a; q.rdx_m = rdx_m; q.rDx_j = rDx_a; q.rDx_p = rDx_b;
    if (fast) stokes_interior_vals<SCALED>(rk, q, oz[0], ox[0], opv[0]);
    else stokes_node_vals<SCALED>(op, i, j0, c, q, x, oz[0], ox[0], opv[0]);
    // column B (lj0 + 1)
    q.vz_c = vz_i.v.y; q.vz_w = vz_i.v.x; q.vz_e = vz_i.e; q.vz_n = vz_n.v.y; q.vz_s = vz_s.v.y; q.vz_nw = vz_n.v.x;
    q.vx_c = vx_i.v.y; q.vx_w = vx_i.v.x; q.vx_e = vx_i.e; q.vx_n = vx_n.v.y; q.vx_s = vx_s.v.y; q.vx_se = vx_s.e;
    q.p_c = p_i.v.y; q.p_w = p_i.v.x; q.p_e = p_i.e; q.p_s = p_s.v.y;
    q.en_c = en_i.v.y; q.en_w = en_i.v.x; q.en_s = en_s.v.y;
    q.es_c = es_i.v.y; q.es_e = es_i.e; q.es_n = es_n.v.y;
    q.rdx_j = rdx_b; q.rdx_m = rdx_a; q.rDx_j = rDx_b; q.rDx_p = rDx_pp;
    if (fast) stokes_interior_vals<SCALED>(rk, q, oz[1], ox[1], opv[1]);
    else if (colB) stokes_node_vals<SCALED>(op, i, j0 + 1, c + 1, q, x, oz[1], ox[1], opv[1]);
    if (add) {                                              // wave-uniform: y = A x + coef[0] * add (pl_solver.hip, deflation)
        const double cf = coef[0];
        const long long P = g.plane;
        if (colB) {
            const double2 az = *reinterpret_cast<const double2*>(add + c), ax = *reinterpret_cast<const double2*>(add + c + P),
                          ap = *reinterpret_cast<const double2*>(add + c + 2 * P);
            oz[0] += cf * az.x; oz[1] += cf * az.y; ox[0] += cf * ax.x; ox[1] += cf * ax.y; opv[0] += cf * ap.x; opv[1] += cf * ap.y;
        } else { oz[0] += cf * add[c]; ox[0] += cf * add[c + P]; opv[0] += cf * add[c + 2 * P]; }
    }
    if (colB) {
        *reinterpret_cast<double2*>(y + c) = make_double2(oz[0], oz[1]);
        *reinterpret_cast<double2*>(y + c + g.plane) = make_double2(ox[0], ox[1]);
        *reinterpret_cast<double2*>(y + c + 2 * g.plane) = make_double2(opv[0], opv[1]);
    } else {
        y[c] = oz[0]; y[c + g.plane] = ox[0]; y[c + 2 * g.plane] = opv[0];
    }
}

template <int ROWS, bool SCALED>
__global__ __launch_bounds__(64 * ROWS) void k_stokes_apply_v2(PlStokesOp op, const double* __restrict__ x,
                                                               double* __restrict__ y, const double* __restrict__ add,
                                                               const double* __restrict__ coef) {
    const PlGeom& g = op.g;
    const int lane = threadIdx.x;
    const int lj0 = (blockIdx.x * 64 + lane) * 2;          // this lane's two columns: lj0, lj0+1
    const int li = blockIdx.y * ROWS + threadIdx.y;
    if (li >= g.lnz) return;                                // wave-uniform
    const bool active = lj0 < g.lnx;                        // lanes beyond the row take part in shuffles only
    const bool has_right = (lj0 + 2) < g.lnx;               // the lane to the right holds real columns
    const int p = g.pitch;
    const int c = (li + PL_RING) * p + PL_PADL + lj0;             // element offset of column A (16-B aligned)
    const double* __restrict__ vz = x;
    const double* __restrict__ vx = x + g.plane;
    const double* __restrict__ P = x + 2 * g.plane;
    const double* __restrict__ es = op.etas;
    const double* __restrict__ en = op.etan;
#define ROW(ptr, dr, w, e) load_row2((ptr) + (long long)c - lj0 + (long long)(dr) * p, lj0, active, w, e, lane, has_right)
    const Row2 vz_s = ROW(vz, -1, false, false), vz_i = ROW(vz, 0, true, true), vz_n = ROW(vz, 1, true, false);
    const Row2 vx_s = ROW(vx, -1, false, true), vx_i = ROW(vx, 0, true, true), vx_n = ROW(vx, 1, false, false);
    const Row2 p_s = ROW(P, -1, false, false), p_i = ROW(P, 0, true, true);
    const Row2 en_s = ROW(en, -1, false, false), en_i = ROW(en, 0, true, false);
    const Row2 es_i = ROW(es, 0, false, true), es_n = ROW(es, 1, false, false);
#undef ROW
    // x tables: pair at (j, j+1) plus the two outer neighbours (tables are padded by PL_TOFF entries)
    const Row2 t_rdx = load_row2(g.rdx + PL_TOFF + g.gj0, lj0, active, true, false, lane, has_right);
    const Row2 t_rDx = load_row2(g.rDx + PL_TOFF + g.gj0, lj0, active, false, true, lane, has_right);
    if (!active) return;
    const int i = g.gi0 + li;
    StokesRows R{vz_s, vz_i, vz_n, vx_s, vx_i, vx_n, p_s, p_i, en_s, en_i, es_i, es_n, t_rdx, t_rDx};
    stokes_compute_store<SCALED>(op, R, i, g.gj0 + blockIdx.x * 128, lj0, c, x, y, add, coef);
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

// add != NULL: y = A x + coef[0] * add  (coef on the device; add: 3 planes like y)
void pl_launch_stokes_apply(pl_ctx* ctx, const PlStokesOp& op, const double* x, double* y, const double* add, const double* coef) {
    // Variants measured on MI355X at 2049^2: 8-byte loads 84 us; double2 81 us; + register row
    // marching (R = 8/16) 99/104 us; XCD-aware block remap, plane-stride padding: no effect; a memory-only twin
    // (same loads and stores, no arithmetic) 58 us -> the arithmetic was exposed: no divisions in the plain
    // operator 76 us, factored coefficients 73 us, branch-free interior path 64 us.  Sharing the rows of a
    // 4- or 8-row tile through LDS (5 instead of 12 global row loads per wave) is SLOWER (68 us): the L1 re-reads
    // were never the limit, the barrier is one.  Round 3: two output rows per wave (17 instead of 24 row loads per pair of rows, 162 VGPRs):
    // 69 vs 68 us at 2049^2, 298 vs 290 us at 4097^2 -- the re-reads are not what the kernel waits for.  Removed.
    // (The scalar one-column-per-lane kernel of round 1 is gone: the vectorised one is pinned directly against the reference's
    // explicit matrices on five grids, tests/test_hip_parity.py.)  double2 accesses need even plane strides and an even first
    // column: the pitch is a multiple of 16 and block columns start at even multiples (pl_set_comm_2d enforces even block widths).
    {
        const int gx = (op.g.lnx + 127) / 128;
#define PL_APPLY_LAUNCH(KERNEL, ROWS)                                                                                   \
        do {                                                                                                            \
            if (op.scaled) hipLaunchKernelGGL((KERNEL<ROWS, true>), dim3(gx, (op.g.lnz + ROWS - 1) / ROWS), dim3(64, ROWS), 0, ctx->stream, op, x, y, add, coef);  \
            else hipLaunchKernelGGL((KERNEL<ROWS, false>), dim3(gx, (op.g.lnz + ROWS - 1) / ROWS), dim3(64, ROWS), 0, ctx->stream, op, x, y, add, coef);           \
        } while (0)
        static const int rows_knob = [] { const char* e = getenv("PYLAMP_APPLY_ROWS"); return e ? atoi(e) : 0; }();
        const int rows = rows_knob ? rows_knob : ((long long)op.g.lnz * op.g.lnx >= 8000000LL ? 16 : 4);
        if (rows == 16) PL_APPLY_LAUNCH(k_stokes_apply_v2, 16);
        else if (rows == 8) PL_APPLY_LAUNCH(k_stokes_apply_v2, 8);
        else if (rows == 2) PL_APPLY_LAUNCH(k_stokes_apply_v2, 2);
        else PL_APPLY_LAUNCH(k_stokes_apply_v2, 4);
#undef PL_APPLY_LAUNCH
    }
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
    op.Kc = Kc; op.Kb = Kb; op.iKc = 1.0 / Kc;
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
    {   // viscosity contrast over the values the operator reads (etan: the physical cells)
        double lo = std::numeric_limits<double>::infinity(), hi = 0.0;
        for (int i = 0; i < ctx->nz; i++)
            for (int j = 0; j < ctx->nx; j++) {
                const double a = etas[(size_t)i * ctx->nx + j];
                if (a > 0.0) { lo = std::min(lo, a); hi = std::max(hi, a); }
                if (i < ctx->nz - 1 && j < ctx->nx - 1) { const double b = etan[(size_t)i * ctx->nx + j]; if (b > 0.0) { lo = std::min(lo, b); hi = std::max(hi, b); } }
            }
        ctx->visc_contrast = (hi > 0.0 && lo > 0.0 && std::isfinite(hi / lo)) ? hi / lo : 1.0;
    }
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
    PL_TRY(pl_halo(ctx, g, dx, 3, g.plane));
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

static int apply_bench(pl_ctx* ctx, int scaled, int reps, double* avg_ms) {
    if (!ctx->sop_ready) return pl_fail(ctx, "stokes operator not set");
    if (reps < 1) reps = 1;
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    size_t vb = (size_t)3 * g.plane * sizeof(double);
    double *dx, *dy;
    PL_TRY(pl_buf(ctx, "api_x", vb, &dx)); PL_TRY(pl_buf(ctx, "api_y", vb, &dy));
    hipLaunchKernelGGL(k_fill_pseudo, dim3(2048), dim3(256), 0, ctx->stream, dx, 3 * g.plane, 12345u);
    PlStokesOp op = ctx->sop; op.scaled = scaled ? 1 : 0;
    pl_launch_stokes_apply(ctx, op, dx, dy);      // warm-up
    PL_TRY(pl_timer_start(ctx));
    for (int r = 0; r < reps; r++) pl_launch_stokes_apply(ctx, op, dx, dy);
    double ms = 0;
    PL_TRY(pl_timer_stop_ms(ctx, &ms));
    PL_HIP(ctx, hipGetLastError());
    if (avg_ms) *avg_ms = ms / reps;
    return 0;
}
extern "C" int pl_stokes_apply_bench(pl_ctx* ctx, int reps, double* avg_ms) { return apply_bench(ctx, 0, reps, avg_ms); }
// the row-scaled variant y = D_r A x, the one the Krylov solver launches
extern "C" int pl_stokes_apply_scaled_bench(pl_ctx* ctx, int reps, double* avg_ms) { return apply_bench(ctx, 1, reps, avg_ms); }

// Stream triad a = b + s c on three arrays of n doubles (24 n bytes per launch, 16 B per lane): the measured HBM rate of
// this box, against which the roofline fractions are reported next to the datasheet peak.
__global__ __launch_bounds__(256) void k_triad(long long n2, double2* __restrict__ a, const double2* __restrict__ b,
                                               const double2* __restrict__ c, double s) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n2; k += (long long)gridDim.x * blockDim.x) {
        const double2 u = b[k], v = c[k];
        a[k] = make_double2(u.x + s * v.x, u.y + s * v.y);
    }
}
extern "C" int pl_stream_triad_bench(pl_ctx* ctx, int64_t n, int reps, double* avg_ms) {
    if (n < 1024 || reps < 1) return pl_fail(ctx, "pl_stream_triad_bench: bad argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    n &= ~(int64_t)1;
    double* buf;
    PL_TRY(pl_buf(ctx, "triad", (size_t)3 * n * sizeof(double), &buf, false));
    hipLaunchKernelGGL(k_fill_pseudo, dim3(2048), dim3(256), 0, ctx->stream, buf, 3 * n, 777u);
    // one double2 per thread measured fastest on MI355X (tools/triad_sweep.py: 6.07 TB/s at 3 x 1 GiB against 4.6 TB/s with 8192
    // grid-striding blocks)
    unsigned nblk = (unsigned)std::min<long long>((n / 2 + 255) / 256, 1 << 20);
    if (const char* e = getenv("PYLAMP_TRIAD_BLOCKS")) { long v = atol(e); if (v >= 64 && v <= (1 << 22)) nblk = (unsigned)v; }
    auto launch = [&]() {
        hipLaunchKernelGGL(k_triad, dim3(nblk), dim3(256), 0, ctx->stream, (long long)(n / 2), (double2*)buf, (const double2*)(buf + n),
                           (const double2*)(buf + 2 * n), 0.5);
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
