// Marker-in-cell kernels: tracer->grid scatter, grid->tracer gather, RK4 advection.
// Replaces pylamp_trac.trac2grid / grid2trac / RK (pylamp_trac.py:30-388).
// Tracers are held SoA on the device (tz[n], tx[n], f_k[n]); the AoS host layout of the
// reference (tr_x (n,2), tr_f (n,13)) is converted at the C-ABI boundary.
#include "pl_internal.h"
#include "pl_mic.h"
#include <cmath>
#include <algorithm>

// ---------------------------------------------------------------------------------------
// AoS <-> SoA
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_aos_to_soa(long long n, const double* __restrict__ src, long long ld,
                                                    int ncol, double* __restrict__ dst, long long dstride) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    for (int k = 0; k < ncol; k++) dst[k * dstride + t] = src[t * ld + k];
}

__global__ __launch_bounds__(256) void k_soa_to_aos(long long n, const double* __restrict__ src, long long sstride,
                                                    int ncol, double* __restrict__ dst, long long ld) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    for (int k = 0; k < ncol; k++) dst[t * ld + k] = src[k * sstride + t];
}

void pl_launch_aos_to_soa(pl_ctx* ctx, long long n, const double* src, long long ld, int ncol, double* dst,
                          long long dstride) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_aos_to_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, src, ld, ncol,
                       dst, dstride);
}

void pl_launch_soa_to_aos(pl_ctx* ctx, long long n, const double* src, long long sstride, int ncol, double* dst,
                          long long ld) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_soa_to_aos, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, src, sstride,
                       ncol, dst, ld);
}

// ---------------------------------------------------------------------------------------
// Scatter (trac2grid).  Clipped bilinear scatter, SURVEY.md B.3.
// ---------------------------------------------------------------------------------------
__device__ inline void mic_atomic_add(double* p, double v) {
    // hardware FP64 atomic add (global_atomic_add_f64); no CAS loop
    unsafeAtomicAdd(p, v);
}

__device__ inline void mic_scatter_locate(const PlScatterArgs& a, double z, double x, int& ie, int& je, double& ca, double& cb) {
    if (a.zc) {                                           // wave-uniform
        mic_axis_locate(a.zc, a.nz, z, ie, ca);
        mic_axis_locate(a.xc, a.nx, x, je, cb);
    } else {
        // reciprocal spacings: four FP64 divisions per marker were a visible share of this ALU-bound kernel
        const double rhz = a.rhz, rhx = a.rhx;
        const double fz = floor((z - a.z0) * rhz), fx = floor((x - a.x0) * rhx);
        ie = (int)fz; je = (int)fx;
        ca = (z - (a.z0 + fz * a.hz)) * rhz; cb = (x - (a.x0 + fx * a.hx)) * rhx;
    }
}

// Unsorted tracers: one thread per tracer, global FP64 atomics.
__global__ __launch_bounds__(256) void k_scatter_atomic(PlScatterArgs a) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n) return;
    const double z = a.tz[t], x = a.tx[t];
    int ie, je; double ca, cb;                            // cell and in-cell coordinates (a, b in pylamp_trac.py:247)
    mic_scatter_locate(a, z, x, ie, je, ca, cb);
    double w[4] = {(1 - cb) * (1 - ca), (1 - cb) * ca, cb * (1 - ca), cb * ca};
    double val[PL_MAX_SCATTER_FIELDS];
    for (int k = 0; k < a.nf; k++) {
        double v = a.f[k][t];
        val[k] = (a.scheme[k] & PL_AVG_GEOMETRIC) && !(a.scheme[k] & (PL_AVG_ARITHMETIC | PL_AVG_PRELOG)) ? log(v) : v;
    }
#pragma unroll
    for (int cnr = 0; cnr < 4; cnr++) {
        const int ni = ie + (cnr & 1), nj = je + (cnr >> 1);
        if (ni < 0 || ni >= a.nz || nj < 0 || nj >= a.nx) continue;
        if (ni < a.row0 || ni >= a.row0 + a.nrows || nj < a.col0 || nj >= a.col0 + a.ncols) continue;
        const long long o = (long long)(ni - a.row0) * a.ncols + (nj - a.col0);
        if (a.wsum) mic_atomic_add(a.wsum + o, w[cnr]);
        if (a.cnt) mic_atomic_add(a.cnt + o, 1.0);
        for (int k = 0; k < a.nf; k++)
            mic_atomic_add(a.acc[k] + o, (a.scheme[k] & PL_AVG_WEIGHTED) ? val[k] * w[cnr] : val[k]);
    }
}

// inclusive sum over the run of lanes [lane - reach, lane] inside a row of 16 lanes (DPP row_shr: pure VALU,
// lanes without a source read 0)
template <int D> __device__ inline double dpp_row_shr(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + D, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + D, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// PL_RUN_MAX: longest run summed in registers (runs are cut at multiples of it): 16 needs four DPP steps, 8 three -- at the
// price of two LDS atomics per 16-marker cell instead of one.  Measured at 68 M markers: the four scatters of a step take
// 5.8 ms with 8 against 6.4 ms with 16
#ifndef PL_RUN_MAX
#define PL_RUN_MAX 8
#endif
__device__ inline double run_sum(double v, int reach) {
    double o;
    o = dpp_row_shr<1>(v); v += (reach >= 1) ? o : 0.0;
    o = dpp_row_shr<2>(v); v += (reach >= 2) ? o : 0.0;
    o = dpp_row_shr<4>(v); v += (reach >= 4) ? o : 0.0;
    if (PL_RUN_MAX > 8) { o = dpp_row_shr<8>(v); v += (reach >= 8) ? o : 0.0; }
    return v;
}

// Cell-sorted tracers: one workgroup per tile of PL_TILE_R x PL_TILE_C sort cells.  The tile's
// tracers are PL_TILE_R contiguous runs; their contributions are accumulated with LDS atomics
// (ds_add_f64) into a (R+2) x (C+2) node window (one extra ring so that every staggering of the
// target node set fits), then flushed with one global atomic per touched node and accumulator.
// Global atomic traffic drops from 4*(nf+1) per TRACER to <= (nf+1) per NODE of the window.
// A contribution outside the window (a tracer that the sort had to clamp) falls back to a
// global atomic, so the result does not depend on the sort being exact.
// NF > 0: number of fields known at compile time (1, 2 and 6 are what the resident step uses; the loops over fields and
// accumulators unroll and val[] stays in registers); NF = 0: generic.  FAST: regular grid, every field weighted and already
// in the form that is summed (arithmetic, or geometric with the logarithm taken beforehand), no count accumulator -- the
// resident step's case; the per-field scheme tests, the count path and the coordinate-search branch drop out.
template <int NF, bool FAST>
__global__ __launch_bounds__(256) void k_scatter_binned(PlScatterArgs a, int tiles_x) {
    extern __shared__ double lds[];
    const int W = PL_TILE_C + 2, H = PL_TILE_R + 2, WH = W * H;
    const int nf = NF > 0 ? NF : a.nf;
    const int nacc = nf + 2;                                    // [0]=wsum, [1]=cnt, [2+k]=field k
    const int tid = threadIdx.x, lane = tid & 63;
    const int ti = blockIdx.x / tiles_x, tj = blockIdx.x % tiles_x;
    const int ci0 = ti * PL_TILE_R, cj0 = tj * PL_TILE_C;
    const int ni0 = a.crow0 + ci0 - 1;                          // global node row of window row 0
    for (int k = tid; k < nacc * WH; k += 256) lds[k] = 0.0;
    __syncthreads();
    const int cj1 = min(cj0 + PL_TILE_C, a.ncx);
    for (int r = 0; r < PL_TILE_R; r++) {
        const int ci = ci0 + r;
        if (ci >= a.ncz) break;
        const long long t0 = a.cell_start[(long long)ci * a.ncx + cj0], t1 = a.cell_start[(long long)ci * a.ncx + cj1];
        for (long long base = t0; base < t1; base += 256) {          // uniform trip count: every lane takes part in the DPP steps
            const long long t = base + tid;
            const bool valid = t < t1;
            double z = 0.0, x = 0.0;
            if (valid) { z = a.tz[t]; x = a.tx[t]; }
            int ie, je; double ca, cb;
            if (FAST) {
                const double fz = floor((z - a.z0) * a.rhz), fx = floor((x - a.x0) * a.rhx);
                ie = (int)fz; je = (int)fx;
                ca = (z - (a.z0 + fz * a.hz)) * a.rhz; cb = (x - (a.x0 + fx * a.hx)) * a.rhx;
            } else mic_scatter_locate(a, z, x, ie, je, ca, cb);
            if (!valid) ie = -0x40000000 - lane;                                  // invalid lanes never share a segment
            const double w[4] = {(1 - cb) * (1 - ca), (1 - cb) * ca, cb * (1 - ca), cb * ca};
            double val[NF > 0 ? NF : PL_MAX_SCATTER_FIELDS];
            for (int k = 0; k < nf; k++) {
                const double v = valid ? a.f[k][t] : 1.0;
                val[k] = FAST ? v : (a.scheme[k] & PL_AVG_GEOMETRIC) && !(a.scheme[k] & (PL_AVG_ARITHMETIC | PL_AVG_PRELOG)) ? log(v) : v;
            }
            // runs of consecutive lanes in the same target cell (the tracers are cell-sorted) are summed in registers
            // first, so that one lane per run issues the LDS atomics: 16 markers per cell made every ds_add_f64 a
            // 16-way same-address conflict (~150 cycles per wave instruction)
            const int ie_p = __builtin_amdgcn_update_dpp(0, ie, 0x111, 0xf, 0xf, false);
            const int je_p = __builtin_amdgcn_update_dpp(0, je, 0x111, 0xf, 0xf, false);
            const bool head = (lane & (PL_RUN_MAX - 1)) == 0 || ie_p != ie || je_p != je;
            const unsigned long long heads = __ballot(head);
            const int seg0 = 63 - __clzll(heads & (~0ull >> (63 - lane)));       // first lane of my run (same row of 16)
            const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
            const int reach = lane - seg0;                                        // lanes to my left in the run
#pragma unroll
            for (int cnr = 0; cnr < 4; cnr++) {
                const int ni = ie + (cnr & 1), nj = je + (cnr >> 1);
                const bool ok = valid && ni >= 0 && ni < a.nz && nj >= 0 && nj < a.nx && ni >= a.row0 && ni < a.row0 + a.nrows &&
                                nj >= a.col0 && nj < a.col0 + a.ncols;
                const int li = ni - ni0, lj = nj - (a.ccol0 + cj0 - 1);
                const bool in_win = li >= 0 && li < H && lj >= 0 && lj < W;
                const int o = li * W + lj;
                const long long go = (long long)(ni - a.row0) * a.ncols + (nj - a.col0);
                for (int q = 0; q < nacc; q++) {
                    if (FAST) { if (q == 1) continue; }
                    else {
                        if (q == 0 && !a.wsum) continue;                          // wave-uniform
                        if (q == 1 && !a.cnt) continue;
                    }
                    double v = (q == 0) ? w[cnr] : (q == 1) ? 1.0 : ((FAST || (a.scheme[q - 2] & PL_AVG_WEIGHTED)) ? val[q - 2] * w[cnr] : val[q - 2]);
                    if (!valid) v = 0.0;
                    v = run_sum(v, reach);
                    if (tail && ok) {
                        if (in_win) unsafeAtomicAdd(&lds[q * WH + o], v);
                        else mic_atomic_add((q == 0 ? a.wsum : q == 1 ? a.cnt : a.acc[q - 2]) + go, v);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int o = tid; o < WH; o += 256) {
        const int ni = ni0 + o / W, nj = a.ccol0 + cj0 - 1 + o % W;
        if (ni < 0 || ni >= a.nz || nj < 0 || nj >= a.nx) continue;
        if (ni < a.row0 || ni >= a.row0 + a.nrows || nj < a.col0 || nj >= a.col0 + a.ncols) continue;
        const long long go = (long long)(ni - a.row0) * a.ncols + (nj - a.col0);
        if (a.wsum) { const double v = lds[o]; if (v != 0.0) mic_atomic_add(a.wsum + go, v); }
        if (a.cnt) { const double v = lds[WH + o]; if (v != 0.0) mic_atomic_add(a.cnt + go, v); }
        for (int k = 0; k < nf; k++) { const double v = lds[(2 + k) * WH + o]; if (v != 0.0) mic_atomic_add(a.acc[k] + go, v); }
    }
}

// (Tried and removed: PL_PRIV_COPIES = 16 private copies of the window in LDS, lane l adding into copy l & 15 -- no same-
// address conflicts, no run sums, (nf+1) x 4 ds_add_f64 per marker from all 64 lanes.  9.6 ms against 6.4 ms for the
// four scatters of a step: ds_add_f64 retires about one LANE per cycle and CU, so the atomics themselves -- 52 per marker
// -- become the bound.  The run sums above cut them 16-fold; that is why they pay despite ~20 VALU instructions each.)

// (Also tried and removed: per-cell serial sums through LDS -- phase A one lane per marker writes its slot weights and
// values to LDS, phase B one lane per (sort cell, accumulator, slot row) adds up the cell's ~16 markers serially, no
// atomics and no cross-lane steps at all.  Correct (the GPU suite passed), but 15.3 ms against 6.3 ms for the four scatters
// of a step: ~84 dependent LDS reads per marker from lanes that all sit at different addresses, two barriers per 256
// markers and 58 KB of LDS per workgroup (8 waves per CU) cost more than the DPP steps they replace.)

// out = g^-1(acc / den), written into a ring/pitch plane or a dense (nz,nx) array
// acc / den point at the accumulator element of output node (0,0); acc_pitch = accumulator columns
__global__ __launch_bounds__(256) void k_scatter_finalize(int nz, int nx, const double* __restrict__ acc,
                                                          const double* __restrict__ den, long long acc_pitch, int scheme,
                                                          double* __restrict__ out, long long out_pitch,
                                                          long long out_off) {
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= nz) return;
    const long long o = (long long)i * acc_pitch + j;
    double s = acc[o];
    const double d = den[o];
    double r;
    if (scheme & PL_AVG_ARITHMETIC) r = s / d;
    else {
        if (isinf(s)) s = 0.0;                    // pylamp_trac.py:301
        r = exp(s / d);
    }
    out[out_off + (long long)i * out_pitch + j] = r;
}

int pl_scatter_device(pl_ctx* ctx, PlScatterArgs& a, double* const* out, long long out_pitch, long long out_off,
                      const PlGeom* slab) {
    if (a.nf < 1 || a.nf > PL_MAX_SCATTER_FIELDS) return pl_fail(ctx, "trac2grid: bad number of fields");
    bool has_w = false, has_c = false;
    for (int k = 0; k < a.nf; k++) {
        int s = a.scheme[k];
        if (!(s & (PL_AVG_ARITHMETIC | PL_AVG_GEOMETRIC))) return pl_fail(ctx, "!!! ERROR INVALID AVERAGING SCHEME");
        if (s & PL_AVG_WEIGHTED) has_w = true; else has_c = true;
    }
    a.rhz = 1.0 / a.hz; a.rhx = 1.0 / a.hx;
    if (slab) { a.row0 = slab->gi0 - 1; a.nrows = slab->lnz + 2; a.col0 = slab->gj0 - 1; a.ncols = slab->lnx + 2; }
    else { a.row0 = 0; a.nrows = a.nz; a.col0 = 0; a.ncols = a.nx; }
    size_t N = (size_t)a.nrows * a.ncols;
    double* accbuf;
    PL_TRY(pl_buf(ctx, "scatter_acc", (size_t)(a.nf + 2) * N * sizeof(double), &accbuf, false));
    PL_HIP(ctx, hipMemsetAsync(accbuf, 0, (size_t)(a.nf + 2) * N * sizeof(double), ctx->stream));
    a.wsum = has_w ? accbuf : nullptr;
    a.cnt = has_c ? accbuf + N : nullptr;
    for (int k = 0; k < a.nf; k++) a.acc[k] = accbuf + (size_t)(2 + k) * N;
    if (a.n > 0 && a.cell_start) {
        const int tiles_x = (a.ncx + PL_TILE_C - 1) / PL_TILE_C, tiles_z = (a.ncz + PL_TILE_R - 1) / PL_TILE_R;
        const size_t shm = (size_t)(a.nf + 2) * (PL_TILE_R + 2) * (PL_TILE_C + 2) * sizeof(double);
        const dim3 gt((unsigned)(tiles_x * tiles_z));
        bool fast = has_w && !has_c && !a.zc;
        for (int k = 0; k < a.nf; k++)
            fast = fast && (a.scheme[k] & PL_AVG_WEIGHTED) && ((a.scheme[k] & PL_AVG_ARITHMETIC) || (a.scheme[k] & PL_AVG_PRELOG));
        if (fast && a.nf == 1) hipLaunchKernelGGL((k_scatter_binned<1, true>), gt, dim3(256), shm, ctx->stream, a, tiles_x);
        else if (fast && a.nf == 2) hipLaunchKernelGGL((k_scatter_binned<2, true>), gt, dim3(256), shm, ctx->stream, a, tiles_x);
        else if (fast && a.nf == 6) hipLaunchKernelGGL((k_scatter_binned<6, true>), gt, dim3(256), shm, ctx->stream, a, tiles_x);
        else if (a.nf == 1) hipLaunchKernelGGL((k_scatter_binned<1, false>), gt, dim3(256), shm, ctx->stream, a, tiles_x);
        else if (a.nf == 2) hipLaunchKernelGGL((k_scatter_binned<2, false>), gt, dim3(256), shm, ctx->stream, a, tiles_x);
        else hipLaunchKernelGGL((k_scatter_binned<0, false>), gt, dim3(256), shm, ctx->stream, a, tiles_x);
        PL_HIP(ctx, hipGetLastError());
    } else if (a.n > 0) {
        hipLaunchKernelGGL(k_scatter_atomic, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, ctx->stream, a);
        PL_HIP(ctx, hipGetLastError());
    }
    int frows = a.nrows, fcols = a.ncols; const double* fbase = accbuf;
    if (slab) {
        // reverse halo: what I accumulated for nodes of the neighbour blocks is added to their accumulators
        if (ctx->nranks > 1)
            PL_TRY(pl_halo_generic(ctx, slab->lnz, slab->lnx, accbuf + a.ncols + 1, a.ncols, a.nf + 2, (long long)N, 1, true));
        frows = slab->lnz; fcols = slab->lnx; fbase = accbuf + a.ncols + 1;           // owned nodes only
    }
    dim3 g2((fcols + 63) / 64, (frows + 3) / 4);
    for (int k = 0; k < a.nf; k++) {
        const double* den = (a.scheme[k] & PL_AVG_WEIGHTED) ? fbase : fbase + N;
        hipLaunchKernelGGL(k_scatter_finalize, g2, dim3(64, 4), 0, ctx->stream, frows, fcols, fbase + (size_t)(2 + k) * N, den,
                           (long long)a.ncols, a.scheme[k], out[k], out_pitch, out_off);
    }
    PL_HIP(ctx, hipGetLastError());
    if (slab && ctx->nranks > 1)
        for (int k = 0; k < a.nf; k++)      // full depth: the solver's kernels also run on the block extended into the halo
            PL_TRY(pl_halo(ctx, *slab, out[k], 1, slab->plane, std::min(PL_RING, std::min(slab->lnz, slab->lnx))));
    return 0;
}

// ---------------------------------------------------------------------------------------
// Gather (grid2trac) and RK4
// ---------------------------------------------------------------------------------------
struct CellLoc { int ie, je; bool bad, oow; double a, b; };

// keep the cell inside the window of the field this rank holds (no-op for a whole grid: all bounds zero)
__device__ inline void mic_window(const PlGatherGrid& g, CellLoc& c) {
    c.oow = false;
    if (g.ie_hi | g.je_hi) {
        const int i2 = min(max(c.ie, g.ie_lo), g.ie_hi), j2 = min(max(c.je, g.je_lo), g.je_hi);
        c.oow = !c.bad && (i2 != c.ie || j2 != c.je);
        c.ie = i2; c.je = j2;
    }
}

// Cell lookup + normalised in-cell coordinates exactly as pylamp_trac.py:42-52,63-75,89-90.
// A cell index equal to n-1 passes the reference's range test but then indexes one past
// its coordinate array (IndexError); here it is treated as outside.
// FAST (compile time): regular grid with precomputed reciprocals (g.rect == 0, g.fast_uniform != 0) -- the resident step on a
// uniform grid; the coordinate-search path and every FP64 division drop out of the kernel
template <bool FAST = false>
__device__ inline CellLoc mic_locate(const PlGatherGrid& g, double z, double x) {
    CellLoc c;
    if (!FAST && g.rect) {                                         // wave-uniform; at or beyond the last coordinate = outside
        c.bad = !(z >= g.gz[0] && z < g.gz[g.nz - 1] && x >= g.gx[0] && x < g.gx[g.nx - 1]);
        c.ie = 0; c.je = 0; c.a = 0.0; c.b = 0.0;
        if (!c.bad) { mic_axis_locate(g.gz, g.nz, z, c.ie, c.a); mic_axis_locate(g.gx, g.nx, x, c.je, c.b); }
        else { c.a = (z - g.gz[0]) / (g.gz[1] - g.gz[0]); c.b = (x - g.gx[0]) / (g.gx[1] - g.gx[0]); }   // cell (0,0), as the strict path
        mic_window(g, c);
        return c;
    }
    const double fi = floor((z - g.zmin) * g.sz);          // the reference divides, (nz-1)(z-zmin)/Lz: same cell except within
    const double fj = floor((x - g.xmin) * g.sx);          // an ulp of a grid line, where both cells give the same value
    c.bad = !(fi >= 0.0 && fi <= (double)(g.nz - 2) && fj >= 0.0 && fj <= (double)(g.nx - 2));
    c.ie = c.bad ? 0 : (int)fi;
    c.je = c.bad ? 0 : (int)fj;
    mic_window(g, c);
    const double dz0 = z - g.gz[c.ie], dz1 = g.gz[c.ie + 1] - z;
    const double dx0 = x - g.gx[c.je], dx1 = g.gx[c.je + 1] - x;
    if (FAST || g.fast_uniform) {         // resident step on a regular grid: dz0 + dz1 = h, no FP64 divisions (8 per RK4 tracer otherwise)
        c.a = dz0 * g.sz; c.b = dx0 * g.sx;
    } else {
        c.a = dz0 / (dz0 + dz1);
        c.b = dx0 / (dx0 + dx1);
    }
    return c;
}

__device__ inline double mic_bilinear(const double* __restrict__ F, const PlGatherGrid& g, const CellLoc& c) {
    const long long o = g.off + (long long)c.ie * g.pitch + c.je;
    return (1 - c.b) * (1 - c.a) * F[o] + c.b * (1 - c.a) * F[o + 1] + (1 - c.b) * c.a * F[o + g.pitch] +
           c.b * c.a * F[o + g.pitch + 1];
}

struct __attribute__((packed, aligned(8))) MicPair { double a, b; };      // global_load_dwordx4 at 8-byte alignment

// divergence-conserving velocity interpolation (pylamp_trac.py:98-154)
template <bool FAST = false>
__device__ inline void mic_veldiv(const PlGatherGrid& g, const double* __restrict__ Vz,
                                  const double* __restrict__ Vx, double z, double x, double defval, double& uz,
                                  double& ux, bool& bad, bool& oow) {
    const CellLoc c = mic_locate<FAST>(g, z, x);
    const long long o = g.off + (long long)c.ie * g.pitch + c.je;
    const double hz = g.gz[c.ie + 1] - g.gz[c.ie], hx = g.gx[c.je + 1] - g.gx[c.je];
    const double rzx = (FAST || g.fast_uniform) ? g.hx_over_hz : hx / hz, rxz = (FAST || g.fast_uniform) ? g.hz_over_hx : hz / hx;
    double z00, z01, z10, z11, x00, x01, x10, x11;
    if (FAST) {      // the two nodes of a row are neighbours in memory: ONE 16-byte load (8-byte aligned) instead of two 8-byte ones --
                     // the kernel is bound by the number of gather instructions (16 -> 32 per tracer otherwise), not by their bytes
        const MicPair a = *reinterpret_cast<const MicPair*>(Vz + o), b = *reinterpret_cast<const MicPair*>(Vz + o + g.pitch);
        const MicPair c2 = *reinterpret_cast<const MicPair*>(Vx + o), d = *reinterpret_cast<const MicPair*>(Vx + o + g.pitch);
        z00 = a.a; z01 = a.b; z10 = b.a; z11 = b.b; x00 = c2.a; x01 = c2.b; x10 = d.a; x11 = d.b;
    } else {
        z00 = Vz[o]; z01 = Vz[o + 1]; z10 = Vz[o + g.pitch]; z11 = Vz[o + g.pitch + 1];
        x00 = Vx[o]; x01 = Vx[o + 1]; x10 = Vx[o + g.pitch]; x11 = Vx[o + g.pitch + 1];
    }
    const double w00 = (1 - c.b) * (1 - c.a), w01 = c.b * (1 - c.a), w10 = (1 - c.b) * c.a, w11 = c.b * c.a;
    const double C10 = (0.5 * rzx) * (z00 - z10 + z11 - z01);
    const double C20 = (0.5 * rxz) * (x00 - x01 + x11 - x10);
    ux = w00 * x00 + w01 * x01 + w10 * x10 + w11 * x11 + c.b * (1 - c.b) * C10;
    uz = w00 * z00 + w01 * z01 + w10 * z10 + w11 * z11 + c.a * (1 - c.a) * C20;
    // Reference quirk (pylamp_trac.py:83,156): only the LAST field (vx) receives defval for an
    // out-of-grid tracer; vz keeps the value extrapolated from cell (0,0).
    if (c.bad) ux = defval;
    bad = c.bad;
    oow = oow || c.oow;
}

// FAST: regular grid with reciprocal spacings, bilinear method (the resident step's temperature interpolation)
template <bool FAST, int EPI = 0>
__global__ __launch_bounds__(256) void k_gather(PlGatherArgs a) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n) return;
    const double z = a.tz[t], x = a.tx[t];
    if (EPI != 0) {                                     // FAST, one field, value consumed by the subgrid-diffusion formulas
        const CellLoc c = mic_locate<true>(a.g, z, x);
        const double v = c.bad ? a.defval : mic_bilinear(a.fields[0], a.g, c);
        if (c.bad) atomicAdd(a.n_outside, 1ull);
        if (EPI == 1) {
            const double Told = a.epi_T[t], Tnew = Told + v;
            if (!a.epi_subgrid) { a.epi_T[t] = Tnew; return; }
            const long long e = a.epi_ix ? (long long)a.epi_ix[t] : t;
            const double dt0 = a.epi_hcp[e] * a.epi_rho[t] / (a.epi_hcd[e] * a.epi_inv2);
            const double ts = Told - (Told - Tnew) * exp(-0.5 * a.epi_dt / dt0);
            a.epi_Tsub[t] = ts; a.epi_dTs[t] = ts - Tnew;
        } else a.epi_T[t] = a.epi_Tsub[t] - v;
        return;
    }
    if (!FAST && (a.method & PL_INTERP_NEAREST)) {
        const CellLoc c = mic_locate(a.g, z, x);
        const double dz0 = z - a.g.gz[c.ie], dz1 = a.g.gz[c.ie + 1] - z;
        const double dx0 = x - a.g.gx[c.je], dx1 = a.g.gx[c.je + 1] - x;
        const double d[4] = {dz0 * dz0 + dx0 * dx0, dz0 * dz0 + dx1 * dx1, dz1 * dz1 + dx0 * dx0, dz1 * dz1 + dx1 * dx1};
        int m = 0;
        for (int k = 1; k < 4; k++) if (d[k] < d[m]) m = k;     // first minimum, like np.argmin
        const long long o = a.g.off + (long long)(c.ie + (m >> 1)) * a.g.pitch + (c.je + (m & 1));
        for (int k = 0; k < a.nf; k++) a.out[k][t] = c.bad ? a.defval : a.fields[k][o];
        if (c.bad) atomicAdd(a.n_outside, 1ull);
    } else if (FAST || (a.method & PL_INTERP_LINEAR)) {
        const CellLoc c = mic_locate<FAST>(a.g, z, x);
        for (int k = 0; k < a.nf; k++) {
            const double v = mic_bilinear(a.fields[k], a.g, c);
            a.out[k][t] = c.bad ? a.defval : (a.accumulate ? a.out[k][t] + v : v);
        }
        if (c.bad) atomicAdd(a.n_outside, 1ull);
    } else {
        double uz, ux; bool bad, oow = false;
        mic_veldiv(a.g, a.fields[0], a.fields[1], z, x, a.defval, uz, ux, bad, oow);
        a.out[0][t] = uz; a.out[1][t] = ux;
        if (bad) atomicAdd(a.n_outside, 1ull);
    }
}

// RK4 with the reference's weights (1,1,1,1)/6 (pylamp_trac.py:385) and v = (x_new - x)/dt.
// 48 B/tracer: read (z,x), write (z',x') and (vz,vx); the velocity grid is read through L2.
template <bool FAST>
__global__ __launch_bounds__(256) void k_rk4(PlRk4Args a) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n) return;
    const double z = a.tz[t], x = a.tx[t], dt = a.dt;
    double k1z, k1x, k2z, k2x, k3z, k3x, k4z, k4x; bool bad, oow = false;
    mic_veldiv<FAST>(a.g, a.Vz, a.Vx, z, x, 0.0, k1z, k1x, bad, oow);
    mic_veldiv<FAST>(a.g, a.Vz, a.Vx, z + 0.5 * dt * k1z, x + 0.5 * dt * k1x, 0.0, k2z, k2x, bad, oow);
    mic_veldiv<FAST>(a.g, a.Vz, a.Vx, z + 0.5 * dt * k2z, x + 0.5 * dt * k2x, 0.0, k3z, k3x, bad, oow);
    mic_veldiv<FAST>(a.g, a.Vz, a.Vx, z + dt * k3z, x + dt * k3x, 0.0, k4z, k4x, bad, oow);
    if (oow && a.n_outside_window) atomicAdd(a.n_outside_window, 1ull);
    const double zn = z + (1.0 / 6.0) * dt * (k1z + k2z + k3z + k4z);
    const double xn = x + (1.0 / 6.0) * dt * (k1x + k2x + k3x + k4x);
    if (FAST || a.g.fast_uniform) { const double rdt = 1.0 / dt; a.vz_out[t] = (zn - z) * rdt; a.vx_out[t] = (xn - x) * rdt; }
    else { a.vz_out[t] = (zn - z) / dt; a.vx_out[t] = (xn - x) / dt; }
    double zf = zn, xf = xn;
    if (a.fence) {                                     // pylamp2.py:563-570
        if (zf <= 0.0) zf = a.eps; if (zf >= a.Lz) zf = a.Lz - a.eps;
        if (xf <= 0.0) xf = a.eps; if (xf >= a.Lx) xf = a.Lx - a.eps;
    }
    a.tz_out[t] = zf; a.tx_out[t] = xf;
    if (FAST && a.key.on) {                            // sort key of the new position + one counter atomic per run of equal keys
        const int c = mic_sort_key(a.key, zf, xf);
        a.key.cell[t] = c;
        int seg0, len;
        mic_wave_runs(c, threadIdx.x & 63, seg0, len);
        if ((int)(threadIdx.x & 63) == seg0) atomicAdd(&a.key.count[c], len);
    }
}

void pl_launch_gather(pl_ctx* ctx, const PlGatherArgs& a_in) {
    if (a_in.n <= 0) return;
    PlGatherArgs a = a_in;
    a.g.sz = (a.g.nz - 1) / a.g.Lz; a.g.sx = (a.g.nx - 1) / a.g.Lx;
    const bool fast = !a.g.rect && a.g.fast_uniform && (a.method & PL_INTERP_LINEAR) && !(a.method & PL_INTERP_NEAREST);
    const dim3 gr((unsigned)((a.n + 255) / 256));
    if (a.epi == 1 && fast && a.nf == 1) hipLaunchKernelGGL((k_gather<true, 1>), gr, dim3(256), 0, ctx->stream, a);
    else if (a.epi == 2 && fast && a.nf == 1) hipLaunchKernelGGL((k_gather<true, 2>), gr, dim3(256), 0, ctx->stream, a);
    else if (a.epi != 0) { (void)pl_fail(ctx, "pl_launch_gather: the fused epilogue needs the regular-grid bilinear path"); }
    else if (fast) hipLaunchKernelGGL((k_gather<true, 0>), gr, dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_gather<false, 0>), gr, dim3(256), 0, ctx->stream, a);
}

void pl_launch_rk4(pl_ctx* ctx, const PlRk4Args& a_in) {
    if (a_in.n <= 0) return;
    PlRk4Args a = a_in;
    a.g.sz = (a.g.nz - 1) / a.g.Lz; a.g.sx = (a.g.nx - 1) / a.g.Lx;
    // (Round 3: a variant that stages the velocity window of a workgroup's 256 cell-sorted tracers -- ~4 x 21 nodes -- and the node
    //  coordinates in LDS and gathers from there: 1.98 instead of 1.58 ms for the stage at 2049^2 / 68 M tracers.  The kernel is bound
    //  by its arithmetic and the dependent chain of its four stages, not by the gather path; the window bookkeeping adds to both.  Removed.)
    if (!a.g.rect && a.g.fast_uniform) hipLaunchKernelGGL(k_rk4<true>, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_rk4<false>, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, ctx->stream, a);
}

// ---------------------------------------------------------------------------------------
// C-ABI with host pointers
// ---------------------------------------------------------------------------------------
static int upload_grid(pl_ctx* ctx, const char* name, int gnz, int gnx, const double* gz, const double* gx,
                       PlGatherGrid& g) {
    double* d;
    PL_TRY(pl_buf(ctx, name, (size_t)(gnz + gnx) * sizeof(double), &d, false));
    PL_HIP(ctx, hipMemcpyAsync(d, gz, gnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PL_HIP(ctx, hipMemcpyAsync(d + gnz, gx, gnx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    g.nz = gnz; g.nx = gnx; g.gz = d; g.gx = d + gnz;
    g.zmin = gz[0]; g.xmin = gx[0]; g.Lz = gz[gnz - 1] - gz[0]; g.Lx = gx[gnx - 1] - gx[0];
    g.pitch = gnx; g.off = 0; g.rect = ctx->mic_search ? 1 : 0;
    return 0;
}

extern "C" int pl_mic_set_search(pl_ctx* ctx, int on) { ctx->mic_search = on ? 1 : 0; return 0; }

static int trac2grid_host(pl_ctx* ctx, int64_t n, const double* tr_x, const double* tr_f, int64_t ld_f, int nf,
                          const int* avgscheme, double z0, double hz, double x0, double hx, const double* zc,
                          const double* xc, double* const* out) {
    if (n < 0 || !tr_x || !tr_f || !avgscheme || !out) return pl_fail(ctx, "pl_trac2grid: bad argument");
    if (nf < 1 || nf > PL_MAX_SCATTER_FIELDS) return pl_fail(ctx, "pl_trac2grid: 1..8 fields per call");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const int nz = ctx->nz, nx = ctx->nx;
    size_t N = (size_t)nz * nx;
    double *d_aos, *d_soa, *d_out;
    size_t nn = (size_t)(n > 0 ? n : 1);
    PL_TRY(pl_buf(ctx, "mic_aos", nn * (size_t)(2 + ld_f) * sizeof(double), &d_aos, false));
    PL_TRY(pl_buf(ctx, "mic_soa", nn * (size_t)(2 + nf) * sizeof(double), &d_soa, false));
    PL_TRY(pl_buf(ctx, "mic_out", (size_t)nf * N * sizeof(double), &d_out, false));
    if (n > 0) {
        PL_HIP(ctx, hipMemcpyAsync(d_aos, tr_x, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        PL_HIP(ctx, hipMemcpyAsync(d_aos + 2 * n, tr_f, (size_t)n * ld_f * sizeof(double), hipMemcpyHostToDevice,
                                   ctx->stream));
        pl_launch_aos_to_soa(ctx, n, d_aos, 2, 2, d_soa, n);
        pl_launch_aos_to_soa(ctx, n, d_aos + 2 * n, ld_f, nf, d_soa + 2 * n, n);
    }
    PlScatterArgs a{};
    a.n = n; a.tz = d_soa; a.tx = d_soa + n; a.nf = nf;
    for (int k = 0; k < nf; k++) { a.f[k] = d_soa + (size_t)(2 + k) * n; a.scheme[k] = avgscheme[k]; }
    a.z0 = z0; a.hz = hz; a.x0 = x0; a.hx = hx; a.nz = nz; a.nx = nx;
    if (zc) {
        double* d_c;
        PL_TRY(pl_buf(ctx, "mic_tcoords", (size_t)(nz + nx) * sizeof(double), &d_c, false));
        PL_HIP(ctx, hipMemcpyAsync(d_c, zc, (size_t)nz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        PL_HIP(ctx, hipMemcpyAsync(d_c + nz, xc, (size_t)nx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        a.zc = d_c; a.xc = d_c + nz;
    }
    double* outs[PL_MAX_SCATTER_FIELDS];
    for (int k = 0; k < nf; k++) outs[k] = d_out + (size_t)k * N;
    if (ctx->nranks > 1) return pl_fail(ctx, "pl_trac2grid: host-array MIC calls are single-rank; use the resident step");
    PL_TRY(pl_scatter_device(ctx, a, outs, nx, 0));
    for (int k = 0; k < nf; k++)
        PL_HIP(ctx, hipMemcpyAsync(out[k], outs[k], N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int pl_trac2grid(pl_ctx* ctx, int64_t n, const double* tr_x, const double* tr_f, int64_t ld_f, int nf,
                            const int* avgscheme, double z0, double hz, double x0, double hx, double* const* out) {
    return trac2grid_host(ctx, n, tr_x, tr_f, ld_f, nf, avgscheme, z0, hz, x0, hx, nullptr, nullptr, out);
}

extern "C" int pl_trac2grid_rect(pl_ctx* ctx, int64_t n, const double* tr_x, const double* tr_f, int64_t ld_f, int nf,
                                 const int* avgscheme, const double* zc, const double* xc, double* const* out) {
    if (!zc || !xc) return pl_fail(ctx, "pl_trac2grid_rect: NULL coordinates");
    for (int i = 0; i + 1 < ctx->nz; i++) if (!(zc[i + 1] > zc[i])) return pl_fail(ctx, "pl_trac2grid_rect: coordinates must increase");
    for (int j = 0; j + 1 < ctx->nx; j++) if (!(xc[j + 1] > xc[j])) return pl_fail(ctx, "pl_trac2grid_rect: coordinates must increase");
    return trac2grid_host(ctx, n, tr_x, tr_f, ld_f, nf, avgscheme, 0.0, 1.0, 0.0, 1.0, zc, xc, out);
}

extern "C" int pl_grid2trac(pl_ctx* ctx, int64_t n, const double* tr_x, int nf, const double* const* fields, int gnz,
                            int gnx, const double* gz, const double* gx, int method, double defval,
                            int stop_on_error, double* out, int64_t ld_out, int64_t* n_outside) {
    if (n < 0 || !tr_x || !fields || !gz || !gx || !out) return pl_fail(ctx, "pl_grid2trac: bad argument");
    if (nf < 1 || nf > PL_MAX_GATHER_FIELDS) return pl_fail(ctx, "pl_grid2trac: 1..8 fields per call");
    if (!(method & (PL_INTERP_LINEAR | PL_INTERP_NEAREST | PL_INTERP_VELDIV)))
        return pl_fail(ctx, "pl_grid2trac: unknown interpolation method");
    if (!(method & (PL_INTERP_LINEAR | PL_INTERP_NEAREST)) && nf != 2)
        return pl_fail(ctx, "grid2trac(): method INTERP_METHOD_VELDIV only works in 2D and expects field to be (vz,vx)");
    if (gnz < 2 || gnx < 2) return pl_fail(ctx, "pl_grid2trac: grid too small");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    size_t GN = (size_t)gnz * gnx, nn = (size_t)(n > 0 ? n : 1);
    double *d_aos, *d_soa, *d_f, *d_o;
    PL_TRY(pl_buf(ctx, "mic_aos", nn * (size_t)(2 + nf) * sizeof(double), &d_aos, false));
    PL_TRY(pl_buf(ctx, "mic_soa", nn * (size_t)(2 + nf) * sizeof(double), &d_soa, false));
    PL_TRY(pl_buf(ctx, "mic_gfields", (size_t)nf * GN * sizeof(double), &d_f, false));
    PL_TRY(pl_buf(ctx, "mic_counter", 64, &d_o, false));
    PL_HIP(ctx, hipMemsetAsync(d_o, 0, 64, ctx->stream));
    PlGatherArgs a{};
    PL_TRY(upload_grid(ctx, "mic_ggrid", gnz, gnx, gz, gx, a.g));
    for (int k = 0; k < nf; k++) {
        PL_HIP(ctx, hipMemcpyAsync(d_f + k * GN, fields[k], GN * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        a.fields[k] = d_f + k * GN;
        a.out[k] = d_soa + (size_t)(2 + k) * nn;
    }
    if (n > 0) {
        PL_HIP(ctx, hipMemcpyAsync(d_aos, tr_x, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        pl_launch_aos_to_soa(ctx, n, d_aos, 2, 2, d_soa, nn);
    }
    a.n = n; a.tz = d_soa; a.tx = d_soa + nn; a.nf = nf; a.method = method; a.defval = defval; a.accumulate = 0;
    a.n_outside = (unsigned long long*)d_o;
    pl_launch_gather(ctx, a);
    PL_HIP(ctx, hipGetLastError());
    unsigned long long nout = 0;
    PL_HIP(ctx, hipMemcpyAsync(&nout, d_o, sizeof(nout), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_outside) *n_outside = (int64_t)nout;
    if (stop_on_error && nout > 0) return pl_fail(ctx, "stopOnError in grid2trac");
    if (n > 0) {
        pl_launch_soa_to_aos(ctx, n, d_soa + 2 * nn, nn, nf, d_aos, nf);
        PL_HIP(ctx, hipMemcpy2DAsync(out, (size_t)ld_out * sizeof(double), d_aos, (size_t)nf * sizeof(double),
                                     (size_t)nf * sizeof(double), (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

extern "C" int pl_rk4(pl_ctx* ctx, int64_t n, const double* tr_x, int gnz, int gnx, const double* gz,
                      const double* gx, const double* vz, const double* vx, double tstep, double* v_out,
                      double* x_out) {
    if (n < 0 || !tr_x || !gz || !gx || !vz || !vx || !v_out || !x_out) return pl_fail(ctx, "pl_rk4: bad argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    size_t GN = (size_t)gnz * gnx, nn = (size_t)(n > 0 ? n : 1);
    double *d_aos, *d_soa, *d_f;
    PL_TRY(pl_buf(ctx, "mic_aos", nn * 4 * sizeof(double), &d_aos, false));
    PL_TRY(pl_buf(ctx, "mic_soa", nn * 6 * sizeof(double), &d_soa, false));
    PL_TRY(pl_buf(ctx, "mic_gfields", 2 * GN * sizeof(double), &d_f, false));
    PlRk4Args a{};
    PL_TRY(upload_grid(ctx, "mic_ggrid", gnz, gnx, gz, gx, a.g));
    PL_HIP(ctx, hipMemcpyAsync(d_f, vz, GN * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PL_HIP(ctx, hipMemcpyAsync(d_f + GN, vx, GN * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (n > 0) {
        PL_HIP(ctx, hipMemcpyAsync(d_aos, tr_x, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        pl_launch_aos_to_soa(ctx, n, d_aos, 2, 2, d_soa, nn);
    }
    a.n = n; a.tz = d_soa; a.tx = d_soa + nn; a.Vz = d_f; a.Vx = d_f + GN; a.dt = tstep;
    a.tz_out = d_soa + 2 * nn; a.tx_out = d_soa + 3 * nn; a.vz_out = d_soa + 4 * nn; a.vx_out = d_soa + 5 * nn;
    a.fence = 0;
    pl_launch_rk4(ctx, a);
    PL_HIP(ctx, hipGetLastError());
    if (n > 0) {
        pl_launch_soa_to_aos(ctx, n, d_soa + 2 * nn, nn, 2, d_aos, 2);
        pl_launch_soa_to_aos(ctx, n, d_soa + 4 * nn, nn, 2, d_aos + 2 * nn, 2);
        PL_HIP(ctx, hipMemcpyAsync(x_out, d_aos, (size_t)n * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipMemcpyAsync(v_out, d_aos + 2 * nn, (size_t)n * 2 * sizeof(double), hipMemcpyDeviceToHost,
                                   ctx->stream));
    }
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

void pl_mic_free(pl_ctx* ctx) { (void)ctx; }
