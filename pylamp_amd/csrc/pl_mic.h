// Argument blocks of the marker-in-cell kernels (passed by value to the kernels).
#pragma once
#include "pl_internal.h"

#define PL_MAX_SCATTER_FIELDS 8
#define PL_MAX_GATHER_FIELDS 8

// internal scheme bit: the field values are logarithms already (the resident step takes log(eta) once per marker and step)
#define PL_AVG_PRELOG 64
struct PlScatterArgs {
    long long n;
    const double* tz; const double* tx;
    int nf;
    const double* f[PL_MAX_SCATTER_FIELDS];
    int scheme[PL_MAX_SCATTER_FIELDS];
    // regular target node set: node k at z0 + k*hz (k < nz), x0 + k*hx (k < nx)
    double z0, hz, x0, hx;
    double rhz, rhx;            // 1/hz, 1/hx (filled by pl_scatter_device)
    // rectilinear target node set: coordinates of the nz / nx nodes; NULL = regular (z0, hz, x0, hx)
    const double* zc; const double* xc;
    int nz, nx;
    // dense accumulators of nrows x ncols doubles; accumulator element (0,0) is GLOBAL node (row0, col0)
    // (host API: row0 = col0 = 0, nrows = nz, ncols = nx; resident block: row0 = gi0-1, nrows = lnz+2, col0 = gj0-1,
    // ncols = lnx+2 -- one ring of nodes that belong to the neighbour blocks)
    int row0, nrows, col0, ncols;
    double* wsum; double* cnt;
    double* acc[PL_MAX_SCATTER_FIELDS];
    // cell-sorted tracers (optional): tracers of sort cell (ci,cj) are [cell_start[ci*ncx+cj],
    // cell_start[ci*ncx+cj+1]); the sort grid has ncz x ncx cells.  NULL -> unsorted path.
    const int* cell_start; int ncz, ncx;
    int crow0, ccol0;           // global cell row / column of sort cell (0,0)
    // target node set shifted by half a cell against the sort cells along z / x (cell centres, mid-faces): the markers of a
    // sort cell then reach 3 instead of 2 node rows / columns (informational)
    int stag_z, stag_x;
};

// tile of sort cells handled by one workgroup of the LDS-binned scatter
#ifndef PL_TILE_R
#define PL_TILE_R 8
#endif
#define PL_TILE_C 32

struct PlGatherGrid {
    int nz, nx;                 // node counts of the field being interpolated
    const double* gz; const double* gx;
    double zmin, xmin, Lz, Lx;
    double sz, sx;              // (nz-1)/Lz, (nx-1)/Lx (filled by the launch wrappers)
    int rect;                   // 1: cells located by per-axis search in gz/gx (non-uniform grids), 0: the reference's regular formula
    long long pitch, off;       // field element (i,j) is F[off + i*pitch + j] (dense: pitch = nx, off = 0)
    // cells [ie_lo, ie_hi] x [je_lo, je_hi] are held in memory (a rank's window of the field; all zero: the whole
    // grid).  A lookup outside is clamped into the window and counted in *n_outside_window by the kernels that have it.
    int ie_lo, ie_hi, je_lo, je_hi;
    // resident step on a regular grid: in-cell coordinates by multiplication with sz, sx (= 1/h) instead of the reference's
    // division dz0 / (dz0 + dz1) (pylamp_trac.py:89-90) -- equal to within an ulp; the module API keeps the division
    int fast_uniform; double hx_over_hz, hz_over_hx;
};

struct PlGatherArgs {
    long long n;
    const double* tz; const double* tx;
    PlGatherGrid g;
    int nf;
    const double* fields[PL_MAX_GATHER_FIELDS];   // dense (nz,nx)
    double* out[PL_MAX_GATHER_FIELDS];
    int method;
    double defval;
    int accumulate;                                // LINEAR only: out += value
    unsigned long long* n_outside;
    // Optional epilogue of the resident step's temperature interpolation (regular grid, one field; pylamp2.py:455,473-480): the
    // interpolated value v is consumed in registers instead of being written for a second per-marker kernel to read.
    //   epi = 1: Told = T[t], Tnew = Told + v; without subgrid diffusion T[t] = Tnew, else dt0 = hcp rho / (hcd inv2),
    //            Tsub[t] = Told - (Told - Tnew) exp(-0.5 dt / dt0), dTs[t] = Tsub[t] - Tnew
    //   epi = 2: T[t] = Tsub[t] - v
    int epi; int epi_subgrid;
    double* epi_T; const double* epi_hcp; const double* epi_rho; const double* epi_hcd;
    double epi_inv2, epi_dt;
    double* epi_Tsub; double* epi_dTs;
    const int* epi_ix;                             // epoch layout (pl_step.hip): hcp / hcd of tracer t live at index epi_ix[t]; NULL: at t
};

// Sort key of a tracer for the end-of-step counting sort (pl_step.hip): the cell of this rank's block it lies in, one of the 8
// leaver buckets behind the cells (the neighbour block it has moved into) or the trash bucket (del_outside: at or beyond a wall).
// Regular grids; k_rk4 can produce the keys and the per-key counts in its epilogue (the positions are in registers there),
// which saves the separate pass over the positions.
struct PlSortKey {
    int on;                          // RK4: write keys and counts
    double z0, rhz, x0, rhx;         // node grid origin and reciprocal spacings
    int ncz, ncx, crow0, ccol0, gcz, gcx;
    int del_outside; double Lz, Lx;
    int* cell; int* count;
};
__device__ inline int mic_sort_key(const PlSortKey& k, double z, double x) {
    const int nc = k.ncz * k.ncx;
    if (k.del_outside && (z <= 0.0 || z >= k.Lz || x <= 0.0 || x >= k.Lx)) return nc + 8;
    int ci = (int)floor((z - k.z0) * k.rhz), cj = (int)floor((x - k.x0) * k.rhx);     // the lookup of the scatter kernels
    ci = min(max(ci, 0), k.gcz - 1) - k.crow0;                                          // global cell (clamped to the domain) -> block cell
    cj = min(max(cj, 0), k.gcx - 1) - k.ccol0;
    const int dz = ci < 0 ? -1 : (ci >= k.ncz ? 1 : 0), dx = cj < 0 ? -1 : (cj >= k.ncx ? 1 : 0);
    if (dz == 0 && dx == 0) return ci * k.ncx + cj;
    return nc + (dx == 0 ? (dz < 0 ? 0 : 1) : (dz == 0 ? (dx < 0 ? 2 : 3) : (dz < 0 ? (dx < 0 ? 4 : 5) : (dx < 0 ? 6 : 7))));
}
// runs of equal keys inside a wave (the tracers were sorted one step ago): first lane and length of my run
__device__ inline void mic_wave_runs(int c, int lane, int& seg0, int& len) {
    const int prev = __shfl_up(c, 1, 64);
    const bool head = lane == 0 || prev != c;
    const unsigned long long heads = __ballot(head);
    seg0 = 63 - __clzll(heads & (~0ull >> (63 - lane)));
    const unsigned long long above = (seg0 == 63) ? 0ull : (heads >> (seg0 + 1));
    const int nact = __popcll(__ballot(1));                  // the active lanes are a prefix of the wave (the tail of the array)
    len = above ? __ffsll((long long)above) : nact - seg0;
}

struct PlRk4Args {
    long long n;
    const double* tz; const double* tx;
    PlGatherGrid g;
    const double* Vz; const double* Vx;
    double dt;
    double* tz_out; double* tx_out; double* vz_out; double* vx_out;
    int fence; double eps, Lz, Lx;                 // optional fence of pylamp2.py:563-570
    unsigned long long* n_outside_window;          // stage positions whose cell lies outside the local window (or NULL)
    PlSortKey key;                                 // key.on: sort keys + counts of the advected positions (resident step)
};

// Fused scatter of a time step's four target sets over cell-sorted tracers (pl_mic_cells.hip)
struct PlScatterCellsArgs {
    const double* tz; const double* tx;
    const double* fn[6];            // node-set fields (all weighted; arithmetic, or geometric with the logarithm already taken)
    const double* fm;               // field of the two mid-face sets (heat conductivity)
    const int* ix; unsigned ind;    // epoch layout (pl_step.hip): bit k of ind set -> fn[k] of tracer t is fn[k][ix[t]] (bit 31: fm); ix NULL: none
    double z0, hz, rhz, x0, hx, rhx;   // node grid: node k at z0 + k hz; the shifted sets start half a cell later
    int nz, nx;                     // nodes per target set (every set is nz x nx)
    int row0, nrows, col0, ncols;   // accumulator window (as PlScatterArgs)
    const int* cell_start; int ncz, ncx, crow0, ccol0;
    double* accN; double* accC; double* accZ; double* accX;   // (1 + NFN) | 2 | 2 | 2 planes of N doubles: denominator first
    long long N;
    int* slow_count; int* slow_list; int slow_cap;            // tracers found outside their sort cell
    int dbg;                        // PYLAMP_SC_DBG (timing experiments only): 1 no tracer loop, 2 no row epilogue, 4 no emission, 8 no staging
};
#define PL_SCF_MAX 10
struct PlScatterFinalArgs {
    int nz, nx, nf;                 // owned nodes, fields
    const double* acc[PL_SCF_MAX]; const double* den[PL_SCF_MAX]; int scheme[PL_SCF_MAX]; double* out[PL_SCF_MAX];
    long long acc_pitch, out_pitch, out_off;
};
int pl_scatter_cells_device(pl_ctx* ctx, PlScatterCellsArgs& a, int variant);
void pl_launch_scatter_finalize_multi(pl_ctx* ctx, const PlScatterFinalArgs& a);

// Cell index and in-cell coordinate on a rectilinear axis c[0..n-1] (largest ie with c[ie] <= z; a marker exactly
// on the last coordinate belongs to the last cell with a = 1).  Outside the axis the grid continues with the
// spacing of its end cell, like the reference's auto-extension (pylamp_trac.py:207-220): ie < 0 or ie >= n-1.
__device__ inline void mic_axis_locate(const double* __restrict__ c, int n, double z, int& ie, double& a) {
    if (z < c[0]) {
        const double h = c[1] - c[0], f = floor((z - c[0]) / h);
        ie = (int)f; a = (z - (c[0] + f * h)) / h;
        return;
    }
    if (z > c[n - 1]) {
        const double h = c[n - 1] - c[n - 2], f = floor((z - c[n - 1]) / h);
        ie = n - 1 + (int)f; a = (z - (c[n - 1] + f * h)) / h;
        return;
    }
    int lo = 0, hi = n - 1;                      // c[lo] <= z <= c[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (c[mid] <= z) lo = mid; else hi = mid; }
    ie = lo; a = (z - c[lo]) / (c[lo + 1] - c[lo]);
}

// slab != NULL (the rank's block): accumulators carry one ring of nodes; they are summed across ranks
// (reverse halo with the 8 neighbour blocks), only the owned nodes are finalised into the ring planes `out`,
// whose halo is then filled by a forward exchange.
int pl_scatter_device(pl_ctx* ctx, PlScatterArgs& a, double* const* out, long long out_pitch, long long out_off,
                      const PlGeom* slab = nullptr);
void pl_launch_gather(pl_ctx* ctx, const PlGatherArgs& a);
void pl_launch_rk4(pl_ctx* ctx, const PlRk4Args& a);
void pl_launch_aos_to_soa(pl_ctx* ctx, long long n, const double* src, long long ld, int ncol, double* dst,
                          long long dstride);
void pl_launch_soa_to_aos(pl_ctx* ctx, long long n, const double* src, long long sstride, int ncol, double* dst,
                          long long ld);
