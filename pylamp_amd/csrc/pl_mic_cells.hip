// Fused tracer->grid scatter over CELL-SORTED tracers: all four staggered target sets of a time step in ONE pass.
// Replaces the four trac2grid calls of pylamp2.py:307-319 (pylamp_trac.py:161-318) in the resident step on regular grids.
//
// Layout of the work ("lane per cell"): a wave owns a strip of SC_W cell columns and marches down SC_ROWS cell rows.
// Each cell of the current row is served by Q lanes (lane = column * Q + q); the tracers of the strip row are contiguous
// in the sorted arrays, are staged once into LDS with coalesced loads, and lane (column, q) walks the tracers
// s + q, s + q + Q, ... of its own cell, accumulating every weighted sum of that cell in REGISTERS:
//     node set      2 x 2 corners          x (weight sum + NFN fields)
//     centre set    3 x 3 slots            x (weight sum | count, one field)      (target shifted by half a cell in z and x)
//     z-mid set     3 rows x 2 columns     x (weight sum, one field)
//     x-mid set     2 rows x 3 columns     x (weight sum, one field)
// No atomics and no cross-lane traffic per tracer (the previous kernel spent ~15 VALU instructions per value and corner
// on segmented DPP sums and one LDS atomic per run).  At the end of a row the Q partial sums of a cell are added (two
// butterfly steps), neighbouring columns are combined by lane shuffles, neighbouring rows through register carries, and a
// node that has received ALL its contributions inside the strip is written with one plain store.  A strip VISITS one more cell
// column each side and one more cell row above and below the cells it owns (lanes 0 and 15 are those virtual columns), so every
// node it owns receives all its contributions inside the strip: no global atomics, no zeroed accumulator planes, and the same
// bits from run to run (the rim cells are read by two strips; the round-2 kernel added the rim nodes with global atomics).
// A tracer that does not lie in the cell the sort put it in (1-ulp disagreement on a cell boundary) is appended to a
// list and scattered by the generic atomic kernel afterwards: the result never depends on the sort being exact.
#include "pl_internal.h"
#include "pl_mic.h"

#define SC_Q 4                       // lanes per cell: lane = q * 16 + column, so that the 16 columns of a strip are one DPP row
#define SC_NL 16                    // columns per wave; the first and the last are virtual (no cell of their own)
#define SC_W (SC_NL - 2)            // cells per strip row
#ifndef SC_ROWS
#define SC_ROWS 32
#endif
#ifndef SC_CAP
#define SC_CAP 320                  // tracers staged per window (a strip row with more is processed in several windows)
#endif
#define SC_CAPP (SC_CAP + SC_CAP / 16 + 2)
#ifndef SC_NB
#define SC_NB 2                     // staging batches per window
#endif

__device__ inline int sc_pidx(int o) { return o + (o >> 4); }       // one pad slot per 16 tracers: lanes of neighbouring cells hit different banks

// cross-lane traffic of the row epilogue, all on the VALU (no LDS round trips):
// value of the column to the left / right inside the 16-lane row (0 at the ends: those columns are virtual)
template <int CTRL> __device__ inline double sc_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double sc_left(double v) { return sc_dpp<0x111>(v); }        // row_shr:1
__device__ inline double sc_right(double v) { return sc_dpp<0x101>(v); }       // row_shl:1
// v + v(lane ^ 16), then + v(lane ^ 32): the sum over the four q-lanes of a column (gfx950 v_permlane{16,32}_swap)
__device__ inline double sc_qsum(double v) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
    rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}

template <int NFN, int CEN, bool MID>
struct ScAcc {
    double N[2][2][1 + NFN];                          // [di][dj][0 = weight sum, 1.. = fields]
    double C[CEN ? 3 : 1][CEN ? 3 : 1][2];            // [slot row][slot col][den, field]
    double Z[MID ? 3 : 1][MID ? 2 : 1][2];
    double X[MID ? 2 : 1][MID ? 3 : 1][2];
};

// Weights of the three slots (cells g-1, g, g+1 of the set shifted by half a cell) of a tracer at in-cell coordinate a of cell g:
// below the cell's middle it lies between the centres g-1 and g, else between g and g+1.  flat: unweighted scheme (every touched
// node counts 1).
template <bool FLAT>
__device__ inline void sc_slots3(double a, double w[3]) {
    const bool lo = a < 0.5;
    if (FLAT) { w[0] = lo ? 1.0 : 0.0; w[1] = 1.0; w[2] = lo ? 0.0 : 1.0; }
    else { w[0] = lo ? 0.5 - a : 0.0; w[1] = lo ? a + 0.5 : 1.5 - a; w[2] = lo ? 0.0 : a - 0.5; }
}

// IND (heat step only): epoch layout of pl_step.hip -- node fields 2, 4, 5 (cp, H, mat) and the mid-face field (conductivity) of
// tracer t are read at index a.ix[t]
#define SC_IND_MASK ((1u << 2) | (1u << 4) | (1u << 5))
template <int NFN, int CEN, bool MID, bool IND>
__global__ __launch_bounds__(64) void k_scatter_cells(PlScatterCellsArgs a, int strips_x) {
    constexpr int NST = 2 + NFN + (MID ? 1 : 0);
    constexpr unsigned IMASK = IND ? SC_IND_MASK : 0u;
    __shared__ double lds[NST][SC_CAPP];
    __shared__ int csl[SC_ROWS + 2][SC_NL + 1];                // cell_start of the rows / columns the strip visits
    const int lane = threadIdx.x, q = lane >> 4, L = lane & 15;
    const int sx = blockIdx.x % strips_x, sy = blockIdx.x / strips_x;
    // The strip OWNS the target columns of its cells c0 .. c0+SC_W-1 and rows r0 .. r1-1 (the first / last strips also the
    // ring column / row beyond the block) and visits one more cell column each side and one more cell row above and below,
    // so that every node it owns receives ALL its contributions here: plain stores, no atomics, no zeroing, and a result
    // that is the same bit for bit from run to run.
    const int c0 = sx * SC_W, r0 = sy * SC_ROWS, r1 = min(r0 + SC_ROWS, a.ncz);
    const int cc = c0 + L - 1;                                 // my cell column (block-local); lanes 0 and 15 only feed their neighbours
    const bool real = cc >= 0 && cc < a.ncx;
    const int gj = a.ccol0 + cc;                               // global cell column = global node / centre column of this lane
    const int nown = min(SC_W, a.ncx - c0);                    // cells the strip owns: lanes 1 .. nown
    const bool last_x = c0 + SC_W >= a.ncx, last_z = r1 >= a.ncz;
    // columns / rows this lane may write: its own cells', plus the ring column left of the block (first strip, lane 0) and
    // the closing column (last strip, lane nown + 1)
    const bool col_own = ((L >= 1 && L <= nown) || (sx == 0 && L == 0) || (last_x && L == nown + 1)) && gj >= 0 && gj < a.nx &&
                         gj >= a.col0 && gj < a.col0 + a.ncols;
    const int R0 = a.crow0 + (sy == 0 ? r0 - 1 : r0), R1 = a.crow0 + (last_z ? r1 + 1 : r1);       // owned global rows [R0, R1)
    const int i0 = max(r0 - 1, 0), i1 = min(r1 + 1, a.ncz);   // cell rows visited [i0, i1)
    const int cfirst = max(c0 - 1, 0), clast = min(c0 + SC_W + 1, a.ncx);                            // cell columns visited [cfirst, clast)
    for (int k = lane; k < (i1 - i0) * (SC_NL + 1); k += 64) {     // one pass instead of dependent loads per row
        const int r = k / (SC_NL + 1), c = k % (SC_NL + 1);
        const int col = min(max(c0 - 1 + c, cfirst), clast);
        csl[r][c] = a.cell_start[(long long)(i0 + r) * a.ncx + col];
    }
    const double xc = a.x0 + gj * a.hx;                        // origin of my cell column
    // carries between rows (see the header): node-like rows one, shifted rows two.  They live in LDS (the q = 0 lane of a
    // column owns its 17 slots): 34 VGPRs less across the tracer loop, where the 70 accumulators and the staging registers
    // already decide the occupancy
    __shared__ double car[NFN + 11][SC_NL];
    double* const cN = &car[0][L];                  // cN[k] -> car[k][L], stride SC_NL
    double* const cX = &car[NFN + 1][L]; double* const qZ = &car[NFN + 3][L]; double* const pZ = &car[NFN + 5][L];
    double* const qC = &car[NFN + 7][L]; double* const pC = &car[NFN + 9][L];
    if (q == 0)
        for (int k = 0; k < NFN + 11; k++) car[k][L] = 0.0;
    auto emit = [&](double* plane, int gi_row, double v) {
        if (!col_own || gi_row < R0 || gi_row >= R1 || gi_row < 0 || gi_row >= a.nz || gi_row < a.row0 || gi_row >= a.row0 + a.nrows) return;
        plane[(long long)(gi_row - a.row0) * a.ncols + (gj - a.col0)] = v;
    };
    __syncthreads();
    // Software pipeline: the first window of the NEXT row is loaded into registers while the tracers of the current row are
    // summed (one or two waves per SIMD cannot hide the load latency by themselves: without this the kernel took the SUM of its
    // memory time and its arithmetic time)
    constexpr int NU = SC_CAP / 64;
    double pf[NU][NST];
    // epoch layout: the constant fields of a tracer sit at its epoch index.  The indices travel one row AHEAD of the fields and reach
    // the loads that depend on them through LDS (ixs): prefetch(row) loads the fields of `row` with the indices staged a row ago, and
    // the indices of row + 1 next to them -- no load of a prefetch waits for another one (a dependent load through a register costs a
    // vmcnt wait that drains everything issued before it: 2.65 instead of 2.0 ms)
    __shared__ int ixs[IND ? SC_CAP : 1];
    int pix[NU];
    auto prefetch = [&](int row) {
        const int b0 = csl[row - i0][0], b1 = min(b0 + SC_CAP, csl[row - i0][SC_NL]);
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int t = b0 + lane + 64 * u;
            if (t < b1) {
                pf[u][0] = a.tz[t]; pf[u][1] = a.tx[t];
                const int e = IND ? ixs[lane + 64 * u] : t;
#pragma unroll
                for (int k = 0; k < NFN; k++) pf[u][2 + k] = a.fn[k][(IMASK >> k) & 1u ? e : t];
                if (MID) pf[u][2 + NFN] = a.fm[IND ? e : t];
            }
        }
        if (IND && row + 1 < i1) {
            const int c0n = csl[row + 1 - i0][0], c1n = min(c0n + SC_CAP, csl[row + 1 - i0][SC_NL]);
#pragma unroll
            for (int u = 0; u < NU; u++) { const int t = c0n + lane + 64 * u; if (t < c1n) pix[u] = a.ix[t]; }
        }
    };
    if (i0 < i1 && !(a.dbg & 8)) {
        if (IND) {
            const int b0 = csl[0][0], b1 = min(b0 + SC_CAP, csl[0][SC_NL]);
#pragma unroll
            for (int u = 0; u < NU; u++) { const int t = b0 + lane + 64 * u; if (t < b1) ixs[lane + 64 * u] = a.ix[t]; }
            __syncthreads();
        }
        prefetch(i0);
    }
    for (int i = i0; i < i1; i++) {
        const int gi = a.crow0 + i;
        const double zc = a.z0 + gi * a.hz;                    // origin of this cell row
        const int wrow0 = csl[i - i0][0], wrow1 = csl[i - i0][SC_NL];
        int s = 0, e = 0;
        if (real) { s = csl[i - i0][L]; e = csl[i - i0][L + 1]; }
        ScAcc<NFN, CEN, MID> A;
#pragma unroll
        for (int di = 0; di < 2; di++)
#pragma unroll
            for (int dj = 0; dj < 2; dj++)
#pragma unroll
                for (int k = 0; k <= NFN; k++) A.N[di][dj][k] = 0.0;
        if (CEN) {
#pragma unroll
            for (int u = 0; u < 3; u++)
#pragma unroll
                for (int v = 0; v < 3; v++) { A.C[u][v][0] = 0.0; A.C[u][v][1] = 0.0; }
        }
        if (MID) {
#pragma unroll
            for (int u = 0; u < 3; u++)
#pragma unroll
                for (int v = 0; v < 2; v++) { A.Z[u][v][0] = 0.0; A.Z[u][v][1] = 0.0; A.X[v][u][0] = 0.0; A.X[v][u][1] = 0.0; }
        }
        if (!(a.dbg & 8)) {                                    // first window: what was prefetched during the previous row
            const int we0 = min(wrow0 + SC_CAP, wrow1);
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int t = wrow0 + lane + 64 * u;
                if (t < we0) {
                    const int o = sc_pidx(t - wrow0);
#pragma unroll
                    for (int k = 0; k < NST; k++) lds[k][o] = pf[u][k];
                }
            }
            if (IND && i + 1 < i1) {                           // the indices of the next row (loaded with this row's fields)
                const int c0n = csl[i + 1 - i0][0], c1n = min(c0n + SC_CAP, csl[i + 1 - i0][SC_NL]);
#pragma unroll
                for (int u = 0; u < NU; u++) if (c0n + lane + 64 * u < c1n) ixs[lane + 64 * u] = pix[u];
            }
        }
        __syncthreads();
        if (i + 1 < i1 && !(a.dbg & 8)) prefetch(i + 1);
        for (int wb = wrow0; wb < wrow1; wb += SC_CAP) {
            const int we = min(wb + SC_CAP, wrow1);
            if (wb != wrow0) {                                 // a row with more tracers than a window holds (rare): stage in line
                __syncthreads();                               // the previous window has been consumed
                constexpr int UB = (NU + SC_NB - 1) / SC_NB;
#pragma unroll
                for (int b = 0; b < SC_NB; b++) {
                    double st[UB][NST];
#pragma unroll
                    for (int u = 0; u < UB; u++) {
                        const int t = wb + lane + 64 * (b * UB + u);
                        if (b * UB + u < NU && t < we) {
                            st[u][0] = a.tz[t]; st[u][1] = a.tx[t];
                            const int e = IND ? a.ix[t] : t;
#pragma unroll
                            for (int k = 0; k < NFN; k++) st[u][2 + k] = a.fn[k][(IMASK >> k) & 1u ? e : t];
                            if (MID) st[u][2 + NFN] = a.fm[IND ? e : t];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UB; u++) {
                        const int t = wb + lane + 64 * (b * UB + u);
                        if (b * UB + u < NU && t < we) {
                            const int o = sc_pidx(t - wb);
#pragma unroll
                            for (int k = 0; k < NST; k++) lds[k][o] = st[u][k];
                        }
                    }
                }
                __syncthreads();
            }
            int t = s + q;
            if (t < wb) t += ((wb - t + SC_Q - 1) / SC_Q) * SC_Q;
            const int tend = min(e, we);
            for (; t < tend && !(a.dbg & 1); t += SC_Q) {
                const int o = sc_pidx(t - wb);
                const double z = lds[0][o], x = lds[1][o];
                // in-cell coordinates from the cell the sort put the tracer in: the expressions of the generic lookup
                // (pylamp_trac.py:247) with the floor already known
                const double ca = (z - zc) * a.rhz, cb = (x - xc) * a.rhx;
                if (!(ca >= 0.0 && ca < 1.0 && cb >= 0.0 && cb < 1.0)) {      // not in that cell after all: generic kernel afterwards
                    // (only the strip that OWNS the cell lists it: the visiting neighbours skip it as well)
                    if (L >= 1 && L <= nown && i >= r0 && i < r1) {
                        const int k = atomicAdd(a.slow_count, 1);
                        if (k < a.slow_cap) a.slow_list[k] = t;
                    }
                    continue;
                }
                double f[NFN > 0 ? NFN : 1];
#pragma unroll
                for (int k = 0; k < NFN; k++) f[k] = lds[2 + k][o];
                const double wz2[2] = {1.0 - ca, ca}, wx2[2] = {1.0 - cb, cb};
#pragma unroll
                for (int di = 0; di < 2; di++)
#pragma unroll
                    for (int dj = 0; dj < 2; dj++) {
                        const double w = wx2[dj] * wz2[di];
                        A.N[di][dj][0] += w;
#pragma unroll
                        for (int k = 0; k < NFN; k++) A.N[di][dj][1 + k] += f[k] * w;
                    }
                double wz3[3], wx3[3];
                if (CEN || MID) { sc_slots3<CEN == 2>(ca, wz3); sc_slots3<CEN == 2>(cb, wx3); }
                if (CEN) {
                    const double fc = f[NFN > 1 ? 1 : 0];       // the centre set averages node field 1 (log viscosity)
#pragma unroll
                    for (int u = 0; u < 3; u++)
#pragma unroll
                        for (int v = 0; v < 3; v++) {
                            const double w = wx3[v] * wz3[u];
                            A.C[u][v][0] += w; A.C[u][v][1] += fc * w;
                        }
                }
                if (MID) {
                    const double fm = lds[2 + NFN][o];
#pragma unroll
                    for (int u = 0; u < 3; u++)
#pragma unroll
                        for (int v = 0; v < 2; v++) {
                            const double wzm = wx2[v] * wz3[u];              // z shifted, x on the nodes
                            A.Z[u][v][0] += wzm; A.Z[u][v][1] += fm * wzm;
                            const double wxm = wx3[u] * wz2[v];              // x shifted, z on the nodes
                            A.X[v][u][0] += wxm; A.X[v][u][1] += fm * wxm;
                        }
                }
            }
        }
        // ---- columns: this lane's column receives dj = 0 of its own cell and dj = 1 of the cell to the left (node-like
        //      sets), resp. slot 1 of its own, slot 2 of the left and slot 0 of the right cell (x-shifted sets); then the
        //      four q-lanes of the column are added
        double Hn[2][1 + NFN], Hz[3][2], Hx[2][2], Hc[3][2];
#pragma unroll
        for (int di = 0; di < 2; di++)
#pragma unroll
            for (int k = 0; k <= NFN; k++) Hn[di][k] = sc_qsum(A.N[di][0][k] + sc_left(A.N[di][1][k]));
        if (MID) {
#pragma unroll
            for (int u = 0; u < 3; u++)
#pragma unroll
                for (int k = 0; k < 2; k++) Hz[u][k] = sc_qsum(A.Z[u][0][k] + sc_left(A.Z[u][1][k]));
#pragma unroll
            for (int v = 0; v < 2; v++)
#pragma unroll
                for (int k = 0; k < 2; k++) Hx[v][k] = sc_qsum(A.X[v][1][k] + sc_left(A.X[v][2][k]) + sc_right(A.X[v][0][k]));
        }
        if (CEN) {
#pragma unroll
            for (int u = 0; u < 3; u++)
#pragma unroll
                for (int k = 0; k < 2; k++) Hc[u][k] = sc_qsum(A.C[u][1][k] + sc_left(A.C[u][2][k]) + sc_right(A.C[u][0][k]));
        }
        // ---- rows: a node row is complete once the cell rows above and below it have been visited; emit() keeps what this
        //      strip owns
        if (q == 0 && !(a.dbg & 4)) {
#pragma unroll
            for (int k = 0; k <= NFN; k++) {
                emit(a.accN + (long long)k * a.N, gi, Hn[0][k] + cN[k * SC_NL]);
                cN[k * SC_NL] = Hn[1][k];
            }
            if (MID) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    emit(a.accX + (long long)k * a.N, gi, Hx[0][k] + cX[k * SC_NL]);
                    cX[k * SC_NL] = Hx[1][k];
                    emit(a.accZ + (long long)k * a.N, gi - 1, qZ[k * SC_NL] + Hz[0][k]);
                    qZ[k * SC_NL] = Hz[1][k] + pZ[k * SC_NL]; pZ[k * SC_NL] = Hz[2][k];
                }
            }
            if (CEN) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    emit(a.accC + (long long)k * a.N, gi - 1, qC[k * SC_NL] + Hc[0][k]);
                    qC[k * SC_NL] = Hc[1][k] + pC[k * SC_NL]; pC[k * SC_NL] = Hc[2][k];
                }
            }
        }
        __syncthreads();                                       // LDS is restaged by the next row
    }
    if (q == 0 && i1 > i0 && !(a.dbg & 4)) {                   // below the last visited cell row (owned only where no cell row follows)
        const int gl = a.crow0 + i1;
#pragma unroll
        for (int k = 0; k <= NFN; k++) emit(a.accN + (long long)k * a.N, gl, cN[k * SC_NL]);
        if (MID) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                emit(a.accX + (long long)k * a.N, gl, cX[k * SC_NL]);
                emit(a.accZ + (long long)k * a.N, gl - 1, qZ[k * SC_NL]);
                emit(a.accZ + (long long)k * a.N, gl, pZ[k * SC_NL]);
            }
        }
        if (CEN) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                emit(a.accC + (long long)k * a.N, gl - 1, qC[k * SC_NL]);
                emit(a.accC + (long long)k * a.N, gl, pC[k * SC_NL]);
            }
        }
    }
}

// the tracers the fused kernel set aside: one thread each, global atomics into all sets
template <int NFN, int CEN, bool MID, bool IND>
__global__ __launch_bounds__(256) void k_scatter_cells_slow(PlScatterCellsArgs a) {
    constexpr unsigned IMASK = IND ? SC_IND_MASK : 0u;
    const int n = min(*a.slow_count, a.slow_cap);
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        const int t = a.slow_list[k];
        const int e = IND ? a.ix[t] : t;
        const double z = a.tz[t], x = a.tx[t];
        auto add = [&](double* plane, int ni, int nj, double v) {
            if (ni < 0 || ni >= a.nz || nj < 0 || nj >= a.nx || ni < a.row0 || ni >= a.row0 + a.nrows || nj < a.col0 || nj >= a.col0 + a.ncols) return;
            unsafeAtomicAdd(plane + (long long)(ni - a.row0) * a.ncols + (nj - a.col0), v);
        };
        for (int set = 0; set < 4; set++) {
            if ((set == 1 && !CEN) || (set >= 2 && !MID)) continue;
            const bool sz = set == 1 || set == 2, sx = set == 1 || set == 3;       // 0 nodes, 1 centres, 2 z-mid, 3 x-mid
            const double zo = a.z0 + (sz ? 0.5 * a.hz : 0.0), xo = a.x0 + (sx ? 0.5 * a.hx : 0.0);
            const double fz = floor((z - zo) * a.rhz), fx = floor((x - xo) * a.rhx);
            const int ie = (int)fz, je = (int)fx;
            const double ca = (z - (zo + fz * a.hz)) * a.rhz, cb = (x - (xo + fx * a.hx)) * a.rhx;
            double* base = set == 0 ? a.accN : set == 1 ? a.accC : set == 2 ? a.accZ : a.accX;
            const int nfs = set == 0 ? NFN : 1;
            for (int cnr = 0; cnr < 4; cnr++) {
                const int di = cnr & 1, dj = cnr >> 1;
                double w = (dj ? cb : 1.0 - cb) * (di ? ca : 1.0 - ca);
                const bool flat = set == 1 && CEN == 2;
                if (flat) w = 1.0;
                add(base, ie + di, je + dj, w);
                for (int f = 0; f < nfs; f++) {
                    const int fk = set == 0 ? f : (NFN > 1 ? 1 : 0);
                    const double v = set <= 1 ? a.fn[fk][(IMASK >> fk) & 1u ? e : t] : a.fm[IND ? e : t];
                    add(base + (long long)(1 + f) * a.N, ie + di, je + dj, v * w);
                }
            }
        }
    }
}

// out = g^-1(acc / den) for up to PL_SCF_MAX fields in one launch (owned nodes of the block)
__global__ __launch_bounds__(256) void k_scatter_finalize_multi(PlScatterFinalArgs a) {
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (j >= a.nx || i >= a.nz) return;
    const long long o = (long long)i * a.acc_pitch + j, oo = a.out_off + (long long)i * a.out_pitch + j;
    for (int k = 0; k < a.nf; k++) {
        double s = a.acc[k][o];
        const double d = a.den[k][o];
        double r;
        if (a.scheme[k] & PL_AVG_ARITHMETIC) r = s / d;
        else { if (isinf(s)) s = 0.0; r = exp(s / d); }        // pylamp_trac.py:301
        a.out[k][oo] = r;
    }
}

void pl_launch_scatter_finalize_multi(pl_ctx* ctx, const PlScatterFinalArgs& a) {
    if (a.nf <= 0) return;
    hipLaunchKernelGGL(k_scatter_finalize_multi, dim3((a.nx + 63) / 64, (a.nz + 3) / 4), dim3(64, 4), 0, ctx->stream, a);
}

template <int NFN, int CEN, bool MID, bool IND = false>
static void launch_cells(pl_ctx* ctx, const PlScatterCellsArgs& a) {
    const int strips_x = (a.ncx + SC_W - 1) / SC_W, strips_z = (a.ncz + SC_ROWS - 1) / SC_ROWS;
    hipLaunchKernelGGL((k_scatter_cells<NFN, CEN, MID, IND>), dim3((unsigned)(strips_x * strips_z)), dim3(64), 0, ctx->stream, a, strips_x);
    hipLaunchKernelGGL((k_scatter_cells_slow<NFN, CEN, MID, IND>), dim3(32), dim3(256), 0, ctx->stream, a);
}

// variant: 0 = heat step (6 node fields, weighted centres, both mid sets), 1 = heat off (2 node fields, unweighted centres),
// 2 = one field on the nodes (subgrid diffusion)
int pl_scatter_cells_device(pl_ctx* ctx, PlScatterCellsArgs& a, int variant) {
    if (a.ncz <= 0 || a.ncx <= 0) return 0;
    a.rhz = 1.0 / a.hz; a.rhx = 1.0 / a.hx;
    PL_HIP(ctx, hipMemsetAsync(a.slow_count, 0, sizeof(int), ctx->stream));
    if (a.ix && (variant != 0 || a.ind != (SC_IND_MASK | (1u << 31)))) return pl_fail(ctx, "pl_scatter_cells_device: unsupported set of indexed fields");
    if (variant == 0 && a.ix) launch_cells<6, 1, true, true>(ctx, a);
    else if (variant == 0) launch_cells<6, 1, true>(ctx, a);
    else if (variant == 1) launch_cells<2, 2, false>(ctx, a);
    else if (variant == 2) launch_cells<1, 0, false>(ctx, a);
    else return pl_fail(ctx, "pl_scatter_cells_device: unknown variant");
    PL_HIP(ctx, hipGetLastError());
    return 0;
}
