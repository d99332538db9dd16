// Device-resident time step: the build's counterpart of the loop body of pylamp2.py:273-581
// (property update, 4 scatters, time-step selection, Stokes solve, heat solve, temperature
// to tracers + subgrid diffusion, velocity re-centring + ghost fill, RK4 advection, fence).
// plus the end-of-step cell sort, slab migration and census / injection (pylamp2.py:588-633).
// Tracers and every grid field stay in HBM across steps; the host only sees scalars.
// Not included: tracer deletion (pylamp2.py:574-581) - it cannot trigger with the supported walls.
#include "pl_internal.h"
#include "pl_mic.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>

#define NFTRAC 13
enum { TR_RHO = 0, TR_ETA, TR_MRK, TR_TMP, TR_HCD, TR_HCP, TR_RH0, TR_ALP, TR_MAT, TR_ACE, TR_ET0, TR_IHT, TR__ID };
#define GASR 8.31446
#define PL_EPS (1.0 / 1024.0)      // pylamp_const.py:46

constexpr int X0_HIST = 5;                     // older Stokes solutions kept at most for the extrapolated initial guess
struct PlStepState {
    long long n = 0, cap = 0;
    double* tz = nullptr; double* tx = nullptr;      // positions (SoA)
    double* tz2 = nullptr; double* tx2 = nullptr;    // advected positions (swapped in)
    double* f[NFTRAC] = {nullptr};
    double* vtz = nullptr; double* vtx = nullptr;    // tracer velocities of the last advection
    double* tmp[3] = {nullptr, nullptr, nullptr};    // per-tracer scratch
    double* partial = nullptr;                       // reduction partials (device)
    std::vector<double> hpartial;
    double* gcoords = nullptr;                       // device copies of node / padded-centre coordinates
    bool have_newtemp = false, have_solution = false, have_dT = false;
    int n_prev = 0;                             // older solutions kept for the extrapolated initial guess (0..X0_HIST)
    double* x_hist[X0_HIST] = {};               // ... newest first (pl_buf storage, rotated by pointer)
    double dt_hist[X0_HIST + 1] = {};           // the last time steps taken (newest first): the model times of those solutions
    // cell sort
    double* f2[NFTRAC] = {nullptr};                  // permutation targets (swapped with f)
    int* cell = nullptr; int* dest = nullptr;        // per tracer: sort cell, destination slot
    int* orig = nullptr; int* orig2 = nullptr;       // caller's index of the tracer now stored at slot t
    int* cell_count = nullptr; int* cell_start = nullptr; int* block_sums = nullptr;
    int ncz = 0, ncx = 0;                            // sort grid = the cells of this rank's block
    int crow0 = 0, ccol0 = 0;                        // global cell row / column of sort cell (0,0)
    bool sorted = false;                             // tracers are cell-sorted and cell_start is valid
    int* need = nullptr; int* need_off = nullptr;    // injection: per-cell deficit and its exclusive scan
    int* need_flag = nullptr; int* need_rank = nullptr;   // 1 for a deficient cell, and its exclusive scan (reference ID rule)
    int* cell_res = nullptr;                         // residents per cell (census) while the sort makes room for new tracers
    double max_id = -1.0;                            // current maximum of TR__ID over all ranks
    bool max_id_valid = false;                       // false after a deletion: recomputed on the device when needed
    int sort_cells = 0;                              // cells of the current sort grid (the trash bucket follows them)
    // Epoch layout (one rank, regular grid; see sort_tracers): the ten columns no stage ever writes (tracer_const below) and `orig`
    // stay in the order of the sort that opened the epoch; slot[t] is the epoch index of the tracer now at sorted position t.
    int* slot = nullptr; int* slot2 = nullptr;
    bool epoch_on = false; int epoch_age = 0;
    long long epoch_len = 0;                         // entries of the epoch-ordered arrays in use (= n on one rank; several ranks: tracers that have
                                                     // left keep their entry until the next re-layout, arrivals and injected tracers append theirs)
    // Lazy columns: RHO, ETA (rewritten by the next property update) and the tracer velocities (rewritten by the next advection) of
    // the tracers [0, lazy_n) are still in PRE-sort order in f2[RHO], f2[ETA], tmp[0], tmp[1]; dest maps them (flush_lazy).
    bool lazy_pending = false, lazy_inject = false; long long lazy_n = 0;
    bool cols_undefined = false;      // a step dropped the lazy columns (RHO, ETA, tracer velocities) and failed before rewriting them
    std::vector<double> gmz, gmx;                    // midpoint grids (pylamp2.py:92-95)
    // several ranks: error counters of the marker stages of a step ([0] tracers outside the grid in grid2trac, [1] RK4 stages that
    // left the rank's velocity window) and the count tables of the sort, reduced over the ranks ON THE DEVICE (pl_comm_allreduce_dev)
    double* flags = nullptr; double* counts_dev = nullptr; int counts_cap = 0;
};

static PlStepState* state_of(pl_ctx* ctx) {
    if (!ctx->step) ctx->step = new PlStepState();
    return (PlStepState*)ctx->step;
}

void pl_step_free(pl_ctx* ctx) {
    PlStepState* s = (PlStepState*)ctx->step;
    if (!s) return;
    for (double* q : {s->tz, s->tx, s->tz2, s->tx2, s->vtz, s->vtx, s->tmp[0], s->tmp[1], s->tmp[2], s->partial, s->gcoords})
        if (q) (void)hipFree(q);
    for (double* q : s->f) if (q) (void)hipFree(q);
    for (double* q : s->f2) if (q) (void)hipFree(q);
    for (double* q : {s->flags, s->counts_dev}) if (q) (void)hipFree(q);
    for (int* q : {s->slot, s->slot2, s->cell, s->dest, s->orig, s->orig2, s->cell_count, s->cell_start, s->block_sums, s->need, s->need_off, s->need_flag, s->need_rank, s->cell_res}) if (q) (void)hipFree(q);
    delete s;
    ctx->step = nullptr;
}

static dim3 grid2d(const PlGeom& g) { return dim3((g.lnx + 63) / 64, (g.lnz + 3) / 4); }
static dim3 grid1d(long long n) { return dim3((unsigned)std::max<long long>((n + 255) / 256, 1)); }      // n = 0: one idle block

// ---- kernels --------------------------------------------------------------------------------
// AoS rows of the reference (tr_x (n,2), tr_f (n,13)) <-> SoA columns
__global__ __launch_bounds__(256) void k_soa_from_aos2(long long m, const double* __restrict__ src, double* __restrict__ a,
                                                       double* __restrict__ b) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) { a[t] = src[2 * t]; b[t] = src[2 * t + 1]; }
}
__global__ __launch_bounds__(256) void k_aos2_from_soa(long long m, const double* __restrict__ a, const double* __restrict__ b,
                                                       double* __restrict__ dst) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) { dst[2 * t] = a[t]; dst[2 * t + 1] = b[t]; }
}
__global__ __launch_bounds__(256) void k_col_from_aos(long long m, const double* __restrict__ src, int ld, int k,
                                                      double* __restrict__ dst) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) dst[t] = src[t * ld + k];
}
__global__ __launch_bounds__(256) void k_col_to_aos(long long m, const double* __restrict__ src, int ld, int k,
                                                    double* __restrict__ dst) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) dst[t * ld + k] = src[t];
}

// ---- counting sort of the tracers by node-grid cell --------------------------------------------
// The tracers were sorted one step ago and move less than a cell per step, so consecutive lanes mostly hold
// the same cell: one atomic per RUN of equal cells in a wave instead of one per tracer (16x fewer at 16
// markers per cell; the counters were atomic-bound: 2.7 + 3.7 ms for 67 M tracers).
// Sort key of a tracer: the cell of this rank's block it lies in, [0, ncz*ncx); or, behind the cells, one of 8 LEAVER
// buckets (the neighbour block it has moved into; order N S W E NW NE SW SE as in pl_comm.hip); or the trash bucket
// behind those -- del_outside: a tracer at or beyond a wall (pylamp2.py:563-572 with the fence off: TR__ID = -1).
// After the permutation the stayers are the first cell_start[nc] entries, then the leavers per neighbour.
__global__ __launch_bounds__(256) void k_cell_count(long long n, const double* __restrict__ tz, const double* __restrict__ tx,
                                                    double z0, double hz, double x0, double hx, const double* __restrict__ zc,
                                                    const double* __restrict__ xc, int ncz, int ncx, int crow0, int ccol0,
                                                    int gcz, int gcx, int* __restrict__ cell, int* __restrict__ count,
                                                    int del_outside, double Lz, double Lx) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int nc = ncz * ncx;
    int c = -1;                                          // lanes past the end form their own run and add nothing
    if (t < n && del_outside && (tz[t] <= 0.0 || tz[t] >= Lz || tx[t] <= 0.0 || tx[t] >= Lx)) {
        c = nc + 8;
        cell[t] = c;
    } else if (t < n) {
        int ci, cj;
        if (zc) { double a_; mic_axis_locate(zc, gcz + 1, tz[t], ci, a_); mic_axis_locate(xc, gcx + 1, tx[t], cj, a_); }   // rectilinear node grid
        else { ci = (int)floor((tz[t] - z0) * (1.0 / hz)); cj = (int)floor((tx[t] - x0) * (1.0 / hx)); }   // the lookup of the scatter kernels
        ci = min(max(ci, 0), gcz - 1) - crow0;              // global cell (clamped to the domain) -> block cell
        cj = min(max(cj, 0), gcx - 1) - ccol0;
        const int dz = ci < 0 ? -1 : (ci >= ncz ? 1 : 0), dx = cj < 0 ? -1 : (cj >= ncx ? 1 : 0);
        if (dz == 0 && dx == 0) c = ci * ncx + cj;
        else c = nc + (dx == 0 ? (dz < 0 ? 0 : 1) : (dz == 0 ? (dx < 0 ? 2 : 3) : (dz < 0 ? (dx < 0 ? 4 : 5) : (dx < 0 ? 6 : 7))));
        cell[t] = c;
    }
    int seg0, len;
    mic_wave_runs(c, lane, seg0, len);
    if (lane == seg0 && c >= 0) atomicAdd(&count[c], len);
}
// exclusive scan of m ints in three passes (1024 elements per block)
__global__ __launch_bounds__(256) void k_scan_block(int m, const int* __restrict__ in, int* __restrict__ out,
                                                    int* __restrict__ bsum) {
    __shared__ int sh[256];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    int v[4], s = 0;
    for (int k = 0; k < 4; k++) { v[k] = (base + k < m) ? in[base + k] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int add = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    int run = sh[threadIdx.x] - s;                 // exclusive prefix of this thread inside the block
    for (int k = 0; k < 4; k++) { if (base + k < m) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) bsum[blockIdx.x] = sh[255];
}
__global__ void k_scan_sums(int nb, int* __restrict__ bsum) {       // single thread block, serial over <= 64k sums
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 256) {
        __shared__ int sh[256];
        const int k = b0 + threadIdx.x;
        const int v = (k < nb) ? bsum[k] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            int add = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (k < nb) bsum[k] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_scan_add(int m, int* __restrict__ out, const int* __restrict__ bsum, int total_slot) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < m) out[k] += bsum[k / 1024];
    (void)total_slot;
}
__global__ __launch_bounds__(256) void k_cell_place(long long n, const int* __restrict__ cell, const int* __restrict__ start,
                                                    int* __restrict__ fill, int* __restrict__ dest) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int c = (t < n) ? cell[t] : -1;
    int seg0, len;
    mic_wave_runs(c, lane, seg0, len);
    int base = 0;
    if (lane == seg0 && c >= 0) base = atomicAdd(&fill[c], len);
    base = __shfl(base, seg0, 64);
    if (c >= 0) dest[t] = start[c] + base + (lane - seg0);
}
// out = a + w b over a whole ring plane
__global__ __launch_bounds__(256) void k_plane_axpy(long long n, double* __restrict__ out, const double* __restrict__ a,
                                                    const double* __restrict__ b, double w) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = a[t] + w * b[t];
}
// x <- l0 x + sum_k l[k] h[k] over nh older solutions; the old x replaces the oldest kept one, h[nk-1] (the host rotates the pointers)
struct ExtrapArgs { double* h[X0_HIST]; double l0, l[X0_HIST]; int nh, nk; };
__global__ __launch_bounds__(256) void k_extrap_x0(long long n, double* __restrict__ x, ExtrapArgs a) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double xo = x[t];
    double v = a.l0 * xo;
    for (int k = 0; k < a.nh; k++) v += a.l[k] * a.h[k][t];
    x[t] = v;
    a.h[a.nk - 1][t] = xo;
}
__global__ __launch_bounds__(256) void k_iota(long long n, int* __restrict__ v, int first = 0) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) v[t] = (int)t + first;
}
__global__ __launch_bounds__(256) void k_permute_int(long long n, const int* __restrict__ dest, const int* __restrict__ in,
                                                     int* __restrict__ out) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[dest[t]] = in[t];
}
// out_k[dest[t]] = in_k[t] for up to 5 arrays per launch
struct PermArgs { const double* in[5]; double* out[5]; int na; };
__global__ __launch_bounds__(256) void k_permute(long long n, const int* __restrict__ dest, PermArgs a) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int d = dest[t];
    for (int k = 0; k < a.na; k++) a.out[k][d] = a.in[k][t];
}

// k_cell_place and the permutation of the few columns the epoch layout moves, in one pass: the destination goes straight from
// the register into the stores (and into dest[] for the lazy columns); slot_in NULL: the sort opens an epoch (slot = old index)
__global__ __launch_bounds__(256) void k_place_permute(long long n, const int* __restrict__ cell, const int* __restrict__ start, int* __restrict__ fill,
                                                       int* __restrict__ dest, PermArgs a, const int* __restrict__ slot_in, int* __restrict__ slot_out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int c = (t < n) ? cell[t] : -1;
    int seg0, len;
    mic_wave_runs(c, lane, seg0, len);
    int base = 0;
    if (lane == seg0 && c >= 0) base = atomicAdd(&fill[c], len);
    base = __shfl(base, seg0, 64);
    if (c < 0) return;
    const int d = start[c] + base + (lane - seg0);
    dest[t] = d;
    for (int k = 0; k < a.na; k++) a.out[k][d] = a.in[k][t];
    slot_out[d] = slot_in ? slot_in[t] : (int)t;
}

// ---- census + injection (pylamp2.py:588-633) ---------------------------------------------------
__global__ __launch_bounds__(256) void k_deficit(int nc, int ncx, int row_lo, int row_hi, const int* __restrict__ start,
                                                 int dens, int dmin, int* __restrict__ need, int* __restrict__ flag) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c > nc) return;
    int v = 0;
    if (c < nc) {
        const int row = c / ncx;
        const int cnt = start[c + 1] - start[c];
        if (row >= row_lo && row < row_hi && cnt < dmin) v = dens - cnt;
    }
    need[c] = v;                                   // need[nc] = 0 so that the scan yields the total
    flag[c] = v > 0 ? 1 : 0;
}
// block partials of max(v[0..n))
__global__ __launch_bounds__(256) void k_max1d(long long n, const double* __restrict__ v, double* __restrict__ part) {
    double m = -INFINITY;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) m = fmax(m, v[t]);
    __shared__ double sh[4];
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
// survivors of a deletion keep the caller's relative order: orig -> rank among the survivors
__global__ __launch_bounds__(256) void k_mark_alive(long long n, const int* __restrict__ orig, int* __restrict__ alive) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) alive[orig[t]] = 1;
}
__global__ __launch_bounds__(256) void k_remap_orig(long long n, int* __restrict__ orig, const int* __restrict__ rank) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) orig[t] = rank[orig[t]];
}
__device__ inline double inj_uniform(unsigned long long seed, unsigned a, unsigned b, unsigned c) {
    unsigned long long h = seed ^ (0x9E3779B97F4A7C15ull * (a + 1)) ^ (0xC2B2AE3D27D4EB4Full * (b + 1)) ^ (0x165667B19E3779F9ull * (c + 1));
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);          // [0,1)
}
// pylamp2.py:291-303
__global__ __launch_bounds__(256) void k_property_update(long long n, const double* __restrict__ T,
                                                         const double* __restrict__ rh0, const double* __restrict__ alp,
                                                         const double* __restrict__ ace, const double* __restrict__ et0,
                                                         double* __restrict__ rho, double* __restrict__ eta, int tdep_rho,
                                                         int tdep_eta, double tref, double etamin, double etamax,
                                                         double* __restrict__ logeta, const int* __restrict__ ix) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const long long c = ix ? (long long)ix[t] : t;              // epoch layout: the constant columns sit at the tracer's epoch index
    const double Tt = T[t];
    rho[t] = tdep_rho ? 1.0 / ((alp[c] * (Tt - tref) + 1.0) / rh0[c]) : rh0[c];
    double e = et0[c];
    if (tdep_eta) {
        const double ac = ace[c];
        e = e * exp(ac / (GASR * Tt) - ac / (GASR * tref));
        if (e < etamin) e = etamin;
        if (e > etamax) e = etamax;
    }
    eta[t] = e;
    logeta[t] = log(e);         // for the two geometric-mean scatters of the viscosity (PL_AVG_PRELOG): one log per marker and step instead of two
}

// block partials of min / max / nan-flag of a ring plane over the lnz x lnx interior
__global__ __launch_bounds__(256) void k_minmax(PlGeom g, const double* __restrict__ a, const double* __restrict__ b,
                                                const double* __restrict__ c, int mode, double* __restrict__ part) {
    // mode 0: value = a ; mode 1: value = 2*a/(b*c)  (diffusivity, pylamp2.py:340-341)
    const int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y;
    double mn = INFINITY, mx = -INFINITY, nanf = 0.0;
    if (lj < g.lnx && li < g.lnz) {
        const long long o = pl_idx(g, li, lj);
        double v = a[o];
        if (mode == 1) v = 2.0 * v / (b[o] * c[o]);
        if (v != v) nanf = 1.0; else { mn = v; mx = v; }
    }
    __shared__ double sm[3][4];
    for (int o = 32; o > 0; o >>= 1) {
        mn = fmin(mn, __shfl_down(mn, o, 64)); mx = fmax(mx, __shfl_down(mx, o, 64)); nanf = fmax(nanf, __shfl_down(nanf, o, 64));
    }
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if ((tid & 63) == 0) { sm[0][tid >> 6] = mn; sm[1][tid >> 6] = mx; sm[2][tid >> 6] = nanf; }
    __syncthreads();
    if (tid == 0) {
        const long long b_ = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        part[3 * b_] = fmin(fmin(sm[0][0], sm[0][1]), fmin(sm[0][2], sm[0][3]));
        part[3 * b_ + 1] = fmax(fmax(sm[1][0], sm[1][1]), fmax(sm[1][2], sm[1][3]));
        part[3 * b_ + 2] = fmax(fmax(sm[2][0], sm[2][1]), fmax(sm[2][2], sm[2][3]));
    }
}

// second stage: the nb block partials -> out[0..2] (one workgroup; the host then reads 24 bytes instead of 400 KB of partials)
__global__ __launch_bounds__(1024) void k_minmax_final(int nb, const double* __restrict__ part, double* __restrict__ out) {
    double mn = INFINITY, mx = -INFINITY, nf = 0.0;
    for (int k = threadIdx.x; k < nb; k += 1024) { mn = fmin(mn, part[3 * k]); mx = fmax(mx, part[3 * k + 1]); nf = fmax(nf, part[3 * k + 2]); }
    __shared__ double sm[3][16];
    for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_down(mn, o, 64)); mx = fmax(mx, __shfl_down(mx, o, 64)); nf = fmax(nf, __shfl_down(nf, o, 64)); }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = mn; sm[1][threadIdx.x >> 6] = mx; sm[2][threadIdx.x >> 6] = nf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) { mn = fmin(mn, sm[0][w]); mx = fmax(mx, sm[1][w]); nf = fmax(nf, sm[2][w]); }
        out[0] = mn; out[1] = mx; out[2] = nf;
    }
}

// f_T boundary rows/cols from the previous solution (pylamp2.py:333-337)
__global__ __launch_bounds__(256) void k_copy_boundary(PlGeom g, const double* __restrict__ src, double* __restrict__ dst) {
    const int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y;
    if (lj >= g.lnx || li >= g.lnz) return;
    const int i = g.gi0 + li, j = g.gj0 + lj;
    if (i == 0 || i == g.nz - 1 || j == 0 || j == g.nx - 1) { const long long o = pl_idx(g, li, lj); dst[o] = src[o]; }
}

__global__ __launch_bounds__(256) void k_plane_sub(PlGeom g, const double* __restrict__ a, const double* __restrict__ b,
                                                   double* __restrict__ out) {
    const int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y;
    if (lj >= g.lnx || li >= g.lnz) return;
    const long long o = pl_idx(g, li, lj);
    out[o] = a[o] - b[o];
}

// Cell-centred advection velocities on the padded (nz+1, nx+1) grid with the ghost ring
// filled per wall BC in source order z0, x0, zL, xL (pylamp2.py:491-545).  Every rank computes the window
// [I0, I0+nr) x [J0, J0+nc) of it that the RK4 stages of its own tracers can reach, from its velocity planes and
// their halo (no all-gather).  Dense output of the window.
__global__ __launch_bounds__(256) void k_advection_velocity(PlGeom g, const double* __restrict__ vz,
                                                            const double* __restrict__ vx, int fs_z0, int fs_x0,
                                                            int fs_zL, int fs_xL, int I0, int J0, int nr, int nc,
                                                            double* __restrict__ Vz, double* __restrict__ Vx) {
    const int wj = blockIdx.x * 64 + threadIdx.x, wi = blockIdx.y * 4 + threadIdx.y;
    const int nz = g.nz, nx = g.nx;
    if (wj >= nc || wi >= nr) return;
    const int I = I0 + wi, J = J0 + wj;
    // value of the un-filled array at (I,J): interior = centre average, ring = 0
    auto raw = [&](int a, int b, double& oz, double& ox) {
        oz = 0.0; ox = 0.0;
        if (a >= 1 && a <= nz - 1 && b >= 1 && b <= nx - 1) {
            const long long o = pl_idx(g, a - 1 - g.gi0, b - 1 - g.gj0);
            oz = 0.5 * (vz[o + g.pitch] + vz[o]);
            ox = 0.5 * (vx[o + 1] + vx[o]);
        }
    };
    // apply the four fills in order; each reads the array state left by the previous ones.
    // Resolve by chasing the source index: a ring node copies from its inner neighbour, which may
    // itself be a ring node of an EARLIER fill.
    int a = I, b = J; double sz = 1.0, sx = 1.0;
    // fills executed last take precedence, so undo them in reverse order: xL, zL, x0, z0
    if (b == nx && fs_xL) { b = nx - 1; sx = -sx; }
    if (a == nz && fs_zL) { a = nz - 1; sz = -sz; }
    if (b == 0 && fs_x0) { b = 1; sx = -sx; }
    if (a == 0 && fs_z0) { a = 1; sz = -sz; }
    // after undoing z0 (executed first) a later-written column fill cannot apply again
    double oz, ox;
    raw(a, b, oz, ox);
    // zL fill of row nz reads row nz-1 INCLUDING its x0/xL ring columns as they stood after the
    // x0 fill but before the xL fill; the chase above handles every combination because each
    // step moves strictly inward and an inner node is never a ring node of an earlier fill.
    const long long o = (long long)wi * nc + wj;
    Vz[o] = sz * oz; Vx[o] = sx * ox;
}

// T_tr update with subgrid diffusion, part 1 (pylamp2.py:455,473-476):
//   Tnew = Told + dTinterp ; Tsub = Told - (Told - Tnew) exp(-0.5 dt/dt0) ; dTs = Tsub - Tnew
__global__ __launch_bounds__(256) void k_subgrid_part1(long long n, double* __restrict__ T, const double* __restrict__ dTi,
                                                       const double* __restrict__ hcp, const double* __restrict__ rho,
                                                       const double* __restrict__ hcd, double inv2, double dt,
                                                       int do_subgrid, double* __restrict__ Tsub, double* __restrict__ dTs,
                                                       const int* __restrict__ ix) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double Told = T[t], Tnew = Told + dTi[t];
    if (!do_subgrid) { T[t] = Tnew; return; }
    const long long c = ix ? (long long)ix[t] : t;
    const double dt0 = hcp[c] * rho[t] / (hcd[c] * inv2);
    const double ts = Told - (Told - Tnew) * exp(-0.5 * dt / dt0);
    Tsub[t] = ts; dTs[t] = ts - Tnew;
}
__global__ __launch_bounds__(256) void k_subgrid_part2(long long n, double* __restrict__ T, const double* __restrict__ Tsub,
                                                       const double* __restrict__ back) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    T[t] = Tsub[t] - back[t];
}

__global__ void k_flag_from_u64(const unsigned long long* __restrict__ src, double* __restrict__ dst) { *dst = (double)*src; }
// dst[first + k] = src[k + 1] - src[k] (sort buckets -> this rank's slots of a count table), or the ints themselves (diff = 0)
__global__ void k_counts_from_ints(int n, const int* __restrict__ src, int diff, double* __restrict__ dst, int first) {
    const int k = threadIdx.x;
    if (k < n) dst[first + k] = diff ? (double)(src[k + 1] - src[k]) : (double)src[k];
}
static int step_flags(pl_ctx* ctx, PlStepState* S) {
    if (!S->flags) { PL_HIP(ctx, hipMalloc((void**)&S->flags, 8 * sizeof(double))); PL_HIP(ctx, hipMemsetAsync(S->flags, 0, 8 * sizeof(double), ctx->stream)); }
    return 0;
}
// zeroed table of n doubles on the device for a sum over the ranks
static int step_counts(pl_ctx* ctx, PlStepState* S, int n) {
    if (S->counts_cap < n) {
        if (S->counts_dev) (void)hipFree(S->counts_dev);
        PL_HIP(ctx, hipMalloc((void**)&S->counts_dev, (size_t)n * sizeof(double)));
        S->counts_cap = n;
    }
    PL_HIP(ctx, hipMemsetAsync(S->counts_dev, 0, (size_t)n * sizeof(double), ctx->stream));
    return 0;
}

// ---- helpers ---------------------------------------------------------------------------------
// min / max / NaN flag of up to 4 planes in ONE host round trip and -- several ranks -- ONE device all-reduce (max) of 3 nq
// values (min as -max(-x)): the reference's global np.min / np.max (pylamp_stokes.py:116-118, pylamp2.py:340-341,364).
struct MinMaxSpec { const double* a; const double* b; const double* c; int mode; };
struct MinMaxOut { double mn, mx; bool has_nan; };
__global__ void k_minmax_negate(int nq, double* __restrict__ out) { if ((int)threadIdx.x < nq) out[3 * threadIdx.x] = -out[3 * threadIdx.x]; }
static int reduce_minmax_multi(pl_ctx* ctx, PlStepState* S, const PlGeom& g, int nq, const MinMaxSpec* spec, MinMaxOut* res) {
    dim3 gr = grid2d(g);
    const size_t nb = (size_t)gr.x * gr.y;
    if (nq < 1 || nq > 4) return pl_fail(ctx, "reduce_minmax_multi: 1..4 planes");
    if (!S->partial || S->hpartial.size() < 3 * (nb + 4)) {
        if (S->partial) (void)hipFree(S->partial);
        PL_HIP(ctx, hipMalloc((void**)&S->partial, 3 * (nb + 4) * sizeof(double)));
        S->hpartial.resize(3 * (nb + 4));
    }
    double* out = S->partial + 3 * nb;
    for (int q = 0; q < nq; q++) {
        hipLaunchKernelGGL(k_minmax, gr, dim3(64, 4), 0, ctx->stream, g, spec[q].a, spec[q].b, spec[q].c, spec[q].mode, S->partial);
        hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(1024), 0, ctx->stream, (int)nb, (const double*)S->partial, out + 3 * q);
    }
    if (ctx->nranks > 1) {
        hipLaunchKernelGGL(k_minmax_negate, dim3(1), dim3(64), 0, ctx->stream, nq, out);
        PL_TRY(pl_comm_allreduce_dev(ctx, out, 3 * nq, 2));
    }
    PL_HIP(ctx, hipMemcpyAsync(S->hpartial.data(), out, (size_t)3 * nq * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < nq; q++) {
        res[q].mn = ctx->nranks > 1 ? -S->hpartial[3 * q] : S->hpartial[3 * q];
        res[q].mx = S->hpartial[3 * q + 1]; res[q].has_nan = S->hpartial[3 * q + 2] > 0.0;
    }
    return 0;
}

// grow every tracer array to hold n tracers, keeping the first `keep` entries
static int grow_tracers(pl_ctx* ctx, PlStepState* S, long long n, long long keep) {
    if (n <= S->cap) return 0;
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const long long cap = n + n / 8 + 1024;
    auto regrow = [&](double** p, bool copy) -> int {
        double* q = nullptr;
        PL_HIP(ctx, hipMalloc((void**)&q, (size_t)cap * sizeof(double)));
        if (copy && *p && keep > 0) PL_HIP(ctx, hipMemcpy(q, *p, (size_t)keep * sizeof(double), hipMemcpyDeviceToDevice));
        if (*p) (void)hipFree(*p);
        *p = q;
        return 0;
    };
    PL_TRY(regrow(&S->tz, true)); PL_TRY(regrow(&S->tx, true));
    for (int k = 0; k < NFTRAC; k++) { PL_TRY(regrow(&S->f[k], true)); PL_TRY(regrow(&S->f2[k], false)); }
    PL_TRY(regrow(&S->vtz, true)); PL_TRY(regrow(&S->vtx, true));
    for (double** q : {&S->tz2, &S->tx2, &S->tmp[0], &S->tmp[1], &S->tmp[2]}) PL_TRY(regrow(q, false));
    for (int** q : {&S->dest, &S->orig2, &S->slot2}) {
        if (*q) (void)hipFree(*q);
        PL_HIP(ctx, hipMalloc((void**)q, (size_t)cap * sizeof(int)));
    }
    for (int** pq : {&S->orig, &S->cell, &S->slot}) {        // the sort keys of a sort in progress are kept as well
        int* q = nullptr; PL_HIP(ctx, hipMalloc((void**)&q, (size_t)cap * sizeof(int)));
        if (*pq && keep > 0) PL_HIP(ctx, hipMemcpy(q, *pq, (size_t)keep * sizeof(int), hipMemcpyDeviceToDevice));
        if (*pq) (void)hipFree(*pq);
        *pq = q;
    }
    S->cap = cap;
    return 0;
}

static int ensure_tracers(pl_ctx* ctx, PlStepState* S, long long n) {
    if (n <= S->cap) return 0;
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    long long cap = n + n / 8 + 1024;
    auto re = [&](double** p) -> int {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        PL_HIP(ctx, hipMalloc((void**)p, (size_t)cap * sizeof(double)));
        return 0;
    };
    for (double** q : {&S->tz, &S->tx, &S->tz2, &S->tx2, &S->vtz, &S->vtx, &S->tmp[0], &S->tmp[1], &S->tmp[2]}) PL_TRY(re(q));
    for (int k = 0; k < NFTRAC; k++) { PL_TRY(re(&S->f[k])); PL_TRY(re(&S->f2[k])); }
    for (int** q : {&S->cell, &S->dest, &S->orig, &S->orig2, &S->slot, &S->slot2}) {
        if (*q) (void)hipFree(*q);
        PL_HIP(ctx, hipMalloc((void**)q, (size_t)cap * sizeof(int)));
    }
    S->cap = cap;
    return 0;
}

static void unpermute(pl_ctx* ctx, PlStepState* S, int na, const double* const* in, double* const* out);
static bool scatter_cells_on() { const char* e = getenv("PYLAMP_SCATTER"); return !(e && atoi(e) == 0); }    // read per call: tests switch it
static int ensure_current(pl_ctx* ctx, PlStepState* S);
struct SortOpts {
    bool full = false;                // move every column (no epoch layout, no lazy columns): the first sort of uploaded tracers
    int del_outside = 0; double Lz = 0.0, Lx = 0.0; long long* removed = nullptr;     // fence off: delete leavers of the domain
    const pl_step_config* inject = nullptr; int it = 0; int64_t* ninjected = nullptr;  // census + refill fused into the sort
    bool keys_ready = false;          // S->cell and S->cell_count were filled by k_rk4's epilogue (prepare_sort_keys + stage_rk4)
    bool premigrated = false;         // several ranks: leavers have been sent and re-keyed as trash, arrivals appended with their keys (migrate_presort)
    bool drop_trash = false;          // the trash bucket behind the sorted tracers is dropped (S->n = everything before it)
};
static int sort_tracers(pl_ctx* ctx, PlStepState* S, double z0, double hz, double x0, double hx, const SortOpts& o = SortOpts());
// Host midpoint grids (pylamp2.py:92-95) and device copies of all coordinate arrays the marker kernels use:
// gcoords = [ node z (nz) | node x (nx) | padded centres z (nz+1) | padded centres x (nx+1) ]; the centre
// (midpoint) grids of the staggered targets are the padded ones without their first entry.
static int ensure_coords(pl_ctx* ctx, PlStepState* S) {
    if (S->gcoords) return 0;
    if (S->gmz.empty()) {
        for (int d = 0; d < 2; d++) {
            const std::vector<double>& c = d ? ctx->geom.xc : ctx->geom.zc;
            std::vector<double>& m = d ? S->gmx : S->gmz;
            for (size_t k = 0; k + 1 < c.size(); k++) m.push_back(0.5 * (c[k + 1] + c[k]));
            m.push_back(m.back() + (m.back() - m[m.size() - 2]));
        }
    }
    std::vector<double> h;
    h.insert(h.end(), ctx->geom.zc.begin(), ctx->geom.zc.end());
    h.insert(h.end(), ctx->geom.xc.begin(), ctx->geom.xc.end());
    h.push_back(S->gmz[0] - (S->gmz[1] - S->gmz[0])); h.insert(h.end(), S->gmz.begin(), S->gmz.end());   // padded centres
    h.push_back(S->gmx[0] - (S->gmx[1] - S->gmx[0])); h.insert(h.end(), S->gmx.begin(), S->gmx.end());
    PL_HIP(ctx, hipMalloc((void**)&S->gcoords, h.size() * sizeof(double)));
    PL_HIP(ctx, hipMemcpy(S->gcoords, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}
// device coordinates of a (possibly staggered) target node set; NULL on a regular grid (arithmetic lookup)
static const double* coords_z(pl_ctx* ctx, PlStepState* S, int stag) {
    if (ctx->geom.uniform) return nullptr;
    return stag ? S->gcoords + ctx->nz + ctx->nx + 1 : S->gcoords;
}
static const double* coords_x(pl_ctx* ctx, PlStepState* S, int stag) {
    if (ctx->geom.uniform) return nullptr;
    return stag ? S->gcoords + ctx->nz + ctx->nx + (ctx->nz + 1) + 1 : S->gcoords + ctx->nz;
}
static int migrate_tracers(pl_ctx* ctx, PlStepState* S, double z0, double hz, double x0, double hx, const SortOpts& o = SortOpts());

extern "C" int pl_tracers_upload(pl_ctx* ctx, int64_t n, const double* tr_x, const double* tr_f) {
    if (n < 0 || !tr_x || !tr_f) return pl_fail(ctx, "pl_tracers_upload: bad argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PlStepState* S = state_of(ctx);
    PL_TRY(ensure_tracers(ctx, S, n));
    S->epoch_on = false; S->epoch_age = 0; S->lazy_pending = false; S->cols_undefined = false;
    // stream the AoS rows through the staging buffer in chunks
    const long long chunk = 1 << 22;
    PL_TRY(pl_stage(ctx, (size_t)chunk * NFTRAC * sizeof(double)));
    for (long long t0 = 0; t0 < n; t0 += chunk) {
        long long m = std::min<long long>(chunk, n - t0);
        PL_HIP(ctx, hipMemcpyAsync(ctx->stage, tr_x + 2 * t0, (size_t)m * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_soa_from_aos2, grid1d(m), dim3(256), 0, ctx->stream, m, ctx->stage, S->tz + t0, S->tx + t0);
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PL_HIP(ctx, hipMemcpyAsync(ctx->stage, tr_f + NFTRAC * t0, (size_t)m * NFTRAC * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        for (int k = 0; k < NFTRAC; k++)
            hipLaunchKernelGGL(k_col_from_aos, grid1d(m), dim3(256), 0, ctx->stream, m, ctx->stage, NFTRAC, k, S->f[k] + t0);
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    PL_HIP(ctx, hipGetLastError());
    if (n > 0) hipLaunchKernelGGL(k_iota, grid1d(n), dim3(256), 0, ctx->stream, (long long)n, S->orig, 0);
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    S->n = n; S->have_newtemp = false; S->have_solution = false; S->have_dT = false; S->n_prev = 0; for (double& d : S->dt_hist) d = 0.0;
    for (double*& h : S->x_hist) h = nullptr;
    double idmax[1] = {-1.0};
    for (int64_t t = 0; t < n; t++) if (tr_f[NFTRAC * t + TR__ID] > idmax[0]) idmax[0] = tr_f[NFTRAC * t + TR__ID];
    PL_TRY(pl_allreduce_host(ctx, idmax, 1, 2));
    S->max_id = idmax[0]; S->max_id_valid = true;
    // cell-sort (and, on a slab, hand over anything that does not belong here)
    PL_TRY(ensure_coords(ctx, S));
    const int nz = ctx->nz, nx = ctx->nx;
    const double z0 = ctx->geom.zc[0], x0 = ctx->geom.xc[0];
    const double hz = (ctx->geom.zc[nz - 1] - z0) / (nz - 1), hx = (ctx->geom.xc[nx - 1] - x0) / (nx - 1);
    { SortOpts so; so.full = true; PL_TRY(sort_tracers(ctx, S, z0, hz, x0, hx, so)); }     // (an epoch opened on the caller's random order would scatter every read of the constant columns)
    PL_TRY(migrate_tracers(ctx, S, z0, hz, x0, hx));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    S->sorted = true;
    return 0;
}

extern "C" int pl_tracers_download(pl_ctx* ctx, int64_t n, double* tr_x, double* tr_f) {
    PlStepState* S = state_of(ctx);
    if (n != S->n) return pl_fail(ctx, "pl_tracers_download: n does not match the resident tracer count");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PL_TRY(ensure_current(ctx, S));
    const long long chunk = 1 << 22;
    PL_TRY(pl_stage(ctx, (size_t)chunk * NFTRAC * sizeof(double)));
    {   // caller's order
        const double* in[15]; double* out[15];
        in[0] = S->tz; out[0] = S->tz2; in[1] = S->tx; out[1] = S->tx2;
        for (int k = 0; k < NFTRAC; k++) { in[2 + k] = S->f[k]; out[2 + k] = S->f2[k]; }
        unpermute(ctx, S, 15, in, out);
    }
    double* const uz = S->tz2; double* const ux = S->tx2; double* const* uf = S->f2;
    for (long long t0 = 0; t0 < n; t0 += chunk) {
        long long m = std::min<long long>(chunk, n - t0);
        if (tr_x) {
            hipLaunchKernelGGL(k_aos2_from_soa, grid1d(m), dim3(256), 0, ctx->stream, m, uz + t0, ux + t0, ctx->stage);
            PL_HIP(ctx, hipMemcpyAsync(tr_x + 2 * t0, ctx->stage, (size_t)m * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        if (tr_f) {
            for (int k = 0; k < NFTRAC; k++)
                hipLaunchKernelGGL(k_col_to_aos, grid1d(m), dim3(256), 0, ctx->stream, m, uf[k] + t0, NFTRAC, k, ctx->stage);
            PL_HIP(ctx, hipMemcpyAsync(tr_f + NFTRAC * t0, ctx->stage, (size_t)m * NFTRAC * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    PL_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int pl_tracers_count(pl_ctx* ctx, int64_t* n) {
    if (n) *n = state_of(ctx)->n;
    return 0;
}

// Per-cell tracer counts of the resident (cell-sorted) state: the census of pylamp2.py:588-598 (np.bincount of the
// cell index), for the cells of this rank's block, row-major (n_cell_rows x n_cell_cols).
extern "C" int pl_tracers_census(pl_ctx* ctx, int64_t ncells, int32_t* counts, int* first_cell_row, int* n_cell_rows) {
    PlStepState* S = state_of(ctx);
    if (!S->sorted || !S->cell_start) return pl_fail(ctx, "pl_tracers_census: no cell-sorted tracers resident");
    if (first_cell_row) *first_cell_row = S->crow0;
    if (n_cell_rows) *n_cell_rows = S->ncz;
    if (!counts) return 0;
    if (ncells != (int64_t)S->ncz * S->ncx) return pl_fail(ctx, "pl_tracers_census: ncells does not match the owned cells");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int> st((size_t)ncells + 1);
    PL_HIP(ctx, hipMemcpyAsync(st.data(), S->cell_start, st.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t c = 0; c < ncells; c++) counts[c] = st[c + 1] - st[c];
    return 0;
}

extern "C" int pl_tracers_layout(pl_ctx* ctx, int* epoch_age, int* lazy) {
    PlStepState* S = state_of(ctx);
    if (epoch_age) *epoch_age = S->epoch_on ? S->epoch_age : 0;
    if (lazy) *lazy = S->lazy_pending ? 1 : 0;
    return 0;
}

extern "C" int pl_get_tracer_velocity(pl_ctx* ctx, int64_t n, double* out) {
    PlStepState* S = state_of(ctx);
    if (n != S->n || !out) return pl_fail(ctx, "pl_get_tracer_velocity: bad argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PL_TRY(ensure_current(ctx, S));
    const long long chunk = 1 << 22;
    PL_TRY(pl_stage(ctx, (size_t)chunk * 2 * sizeof(double)));
    {
        const double* in[2] = {S->vtz, S->vtx}; double* out[2] = {S->tz2, S->tx2};
        unpermute(ctx, S, 2, in, out);
    }
    for (long long t0 = 0; t0 < n; t0 += chunk) {
        long long m = std::min<long long>(chunk, n - t0);
        hipLaunchKernelGGL(k_aos2_from_soa, grid1d(m), dim3(256), 0, ctx->stream, m, S->tz2 + t0, S->tx2 + t0, ctx->stage);
        PL_HIP(ctx, hipMemcpyAsync(out + 2 * t0, ctx->stage, (size_t)m * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

// new tracers of the deficient cells, written straight into their sorted slots (behind the residents of the cell)
struct InjectSorted {
    int nc, ncx, crow0, ccol0;
    const int* start;            // cell_start of the sorted arrays INCLUDING the new tracers
    const int* count;            // residents per cell
    const int* need; const int* off; const int* rank;     // deficit, its exclusive scan, deficient cells before (or NULL)
    double* tz; double* tx; double* f[NFTRAC]; double* vtz; double* vtx; int* orig;
    double z0, hz, x0, hx; unsigned long long seed; unsigned step; double id0; int n_old;
    const double* zc; const double* xc;
    int* slot;                   // epoch layout: epoch index of the tracer at a sorted position (NULL: every column in sorted order)
    int lazy;                    // RHO / ETA of the residents are not in place yet: inject_fix_lazy writes the new tracers' when they are
};
__host__ __device__ inline bool tracer_const(int k) { return !(k == TR_RHO || k == TR_ETA || k == TR_TMP); }    // no stage writes these columns
__global__ __launch_bounds__(64) void k_inject_sorted(InjectSorted a) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= a.nc) return;
    const int m = a.need[c];
    if (m <= 0) return;
    const int t0 = a.start[c], t1 = t0 + a.count[c];
    const int ci = c / a.ncx + a.crow0, cj = c % a.ncx + a.ccol0;   // global cell
    double mean[NFTRAC];
    for (int k = 0; k < NFTRAC; k++) {
        double sum = 0.0;
        if (a.lazy && (k == TR_RHO || k == TR_ETA)) { mean[k] = 0.0; continue; }
        if (a.slot && tracer_const(k)) for (int t = t0; t < t1; t++) sum += a.f[k][a.slot[t]];
        else for (int t = t0; t < t1; t++) sum += a.f[k][t];
        mean[k] = sum / (double)(t1 - t0);                          // 0/0 = NaN for an empty cell, like the reference
    }
    for (int q = 0; q < m; q++) {
        const int d = t1 + q;
        const int e = a.slot ? a.n_old + a.off[c] + q : d;          // epoch layout: the constant columns of a new tracer go behind the epoch's
        const unsigned gc = (unsigned)(ci * 65536 + cj);            // the stream depends on the GLOBAL cell: same on any layout
        const double uz = inj_uniform(a.seed, gc, (unsigned)q, 2 * a.step), ux = inj_uniform(a.seed, gc, (unsigned)q, 2 * a.step + 1);
        a.tz[d] = a.zc ? a.zc[ci] + uz * (a.zc[ci + 1] - a.zc[ci]) : a.z0 + (ci + uz) * a.hz;
        a.tx[d] = a.xc ? a.xc[cj] + ux * (a.xc[cj + 1] - a.xc[cj]) : a.x0 + (cj + ux) * a.hx;
        for (int k = 0; k < NFTRAC; k++) {
            if (a.lazy && (k == TR_RHO || k == TR_ETA)) continue;
            a.f[k][tracer_const(k) ? e : d] = mean[k];
        }
        // reference rule (pylamp2.py:621-622): the first new ID of every cell repeats the last ID handed out
        a.f[TR__ID][e] = a.id0 + a.off[c] + q - (a.rank ? a.rank[c] : 0);
        a.vtz[d] = 0.0; a.vtx[d] = 0.0;                             // injected tracers have not been advected yet
        a.orig[e] = a.n_old + a.off[c] + q;                         // appended behind the residents in the caller's order
        if (a.slot) a.slot[d] = e;
    }
}
// RHO and ETA of the tracers the last sort injected, once the residents' values have been moved into place (flush_lazy)
__global__ __launch_bounds__(64) void k_inject_fix_lazy(int nc, const int* __restrict__ start, const int* __restrict__ count,
                                                        const int* __restrict__ need, double* __restrict__ rho, double* __restrict__ eta) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= nc) return;
    const int m = need[c];
    if (m <= 0) return;
    const int t0 = start[c], t1 = t0 + count[c];
    double sr = 0.0, se = 0.0;
    for (int t = t0; t < t1; t++) sr += rho[t];
    for (int t = t0; t < t1; t++) se += eta[t];
    sr /= (double)(t1 - t0); se /= (double)(t1 - t0);
    for (int q = 0; q < m; q++) { rho[t1 + q] = sr; eta[t1 + q] = se; }
}
// out_k[t] = in_k[idx[t]] for up to 5 arrays per launch (closing an epoch)
struct GatherColsArgs { const double* in[5]; double* out[5]; int na; };
__global__ __launch_bounds__(256) void k_gather_cols(long long n, const int* __restrict__ idx, GatherColsArgs a) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int s = idx[t];
    for (int k = 0; k < a.na; k++) a.out[k][t] = a.in[k][s];
}
__global__ __launch_bounds__(256) void k_gather_int(long long n, const int* __restrict__ idx, const int* __restrict__ in, int* __restrict__ out) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = in[idx[t]];
}
// total[c] = count[c] + need[c] for the cells, count[c] for the buckets behind them
__global__ __launch_bounds__(256) void k_add_need(int nc, int m, const int* __restrict__ count, const int* __restrict__ need, int* __restrict__ total) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < m) total[c] = count[c] + (c < nc ? need[c] : 0);
}

static int scan_ints(pl_ctx* ctx, PlStepState* S, int m, const int* in, int* out) {
    const int nb = (m + 1023) / 1024;
    hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(256), 0, ctx->stream, m, in, out, S->block_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, ctx->stream, nb, S->block_sums);
    hipLaunchKernelGGL(k_scan_add, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, m, out, S->block_sums, 0);
    return 0;
}

// ---- epoch layout and lazy columns (one rank, regular grid, no deletion) ------------------------------------------------------------
// The end-of-step sort used to move all 17 columns of every tracer (300 B per tracer and step, 4.4 ms for 68 M tracers: pure
// overhead the reference does not have).  But no stage of the time step ever writes ten of the thirteen fields (tracer_const), and
// four more columns are rewritten before anything reads them again:
//   * EPOCH: the constant columns and `orig` stay where they were when the epoch was opened; a 4-byte `slot` per tracer -- the only
//     thing the sort moves for them -- says where.  Tracers move less than a cell per step, so the tracers of a cell still find
//     their constants in a few neighbouring cache lines; after PYLAMP_EPOCH (default 64) sorts relayout() brings the columns into
//     the current order (one gather pass) and the next sort opens a new epoch.  The kernels of the step that read constants take
//     the index array (k_property_update, k_scatter_cells, the gather epilogue, k_inject_sorted).
//   * LAZY: RHO / ETA (next property update) and the tracer velocities (next advection) are not moved by the sort at all; whoever
//     wants them before they are rewritten (downloads, the resident test entry points) calls ensure_current(), which moves them then.
// Every path outside the resident step (downloads, migration, deletion, rectilinear grids, the one-set-per-pass scatter) works on
// the classic layout: ensure_current() first.  PYLAMP_EPOCH=0 switches both off (all columns move in every sort, as before).
static int epoch_length() { const char* e = getenv("PYLAMP_EPOCH"); return e ? atoi(e) : 64; }         // read per call: tests switch it
static int flush_lazy(pl_ctx* ctx, PlStepState* S) {
    if (!S->lazy_pending) return 0;
    S->lazy_pending = false;
    if (S->lazy_n > 0) {
        PermArgs pa{}; pa.na = 4;
        pa.in[0] = S->f2[TR_RHO]; pa.out[0] = S->f[TR_RHO]; pa.in[1] = S->f2[TR_ETA]; pa.out[1] = S->f[TR_ETA];
        pa.in[2] = S->tmp[0]; pa.out[2] = S->vtz; pa.in[3] = S->tmp[1]; pa.out[3] = S->vtx;
        hipLaunchKernelGGL(k_permute, grid1d(S->lazy_n), dim3(256), 0, ctx->stream, S->lazy_n, S->dest, pa);
    }
    if (S->lazy_inject)
        hipLaunchKernelGGL(k_inject_fix_lazy, dim3((S->sort_cells + 63) / 64), dim3(64), 0, ctx->stream, S->sort_cells, S->cell_start, S->cell_res,
                           S->need, S->f[TR_RHO], S->f[TR_ETA]);
    PL_HIP(ctx, hipGetLastError());
    return 0;
}
static int relayout(pl_ctx* ctx, PlStepState* S) {
    if (!S->epoch_on) return 0;
    S->epoch_on = false; S->epoch_age = 0;
    const long long n = S->n;
    if (n <= 0) return 0;
    int cols[NFTRAC], nc = 0;
    for (int k = 0; k < NFTRAC; k++) if (tracer_const(k)) cols[nc++] = k;
    for (int k0 = 0; k0 < nc; k0 += 5) {
        GatherColsArgs ga{}; ga.na = std::min(5, nc - k0);
        for (int k = 0; k < ga.na; k++) { ga.in[k] = S->f[cols[k0 + k]]; ga.out[k] = S->f2[cols[k0 + k]]; }
        hipLaunchKernelGGL(k_gather_cols, grid1d(n), dim3(256), 0, ctx->stream, n, S->slot, ga);
    }
    hipLaunchKernelGGL(k_gather_int, grid1d(n), dim3(256), 0, ctx->stream, n, S->slot, S->orig, S->orig2);
    for (int k = 0; k < nc; k++) std::swap(S->f[cols[k]], S->f2[cols[k]]);
    std::swap(S->orig, S->orig2);
    S->epoch_len = n;
    // several ranks: tracers come and go, a caller's order does not exist (downloads are matched by TR__ID): the identity
    if (ctx->nranks > 1) hipLaunchKernelGGL(k_iota, grid1d(n), dim3(256), 0, ctx->stream, n, S->orig, 0);
    PL_HIP(ctx, hipGetLastError());
    return 0;
}
// every column in the current (sorted) order, as the code outside the resident step expects
static int ensure_current(pl_ctx* ctx, PlStepState* S) {
    if (S->cols_undefined)
        return pl_fail(ctx, "tracer RHO / ETA / velocities are undefined: the last pl_step failed before it had rewritten them (upload the tracers again)");
    PL_TRY(flush_lazy(ctx, S));
    return relayout(ctx, S);
}

// Counting sort of all tracer arrays by the cells of this rank's block; leaves cell_start valid:
//   [0, nc] cells (cell_start[nc] = number of stayers), [nc, nc+8] the leaver buckets per neighbour block,
//   [nc+8] the trash bucket (deleted tracers), [nc+9] = all.
// o.del_outside: tracers at or beyond a wall are removed (fence off, pylamp2.py:563-581).
// o.inject: census + refill (pylamp2.py:588-633) in the SAME pass -- the deficits follow from the counts, the
// permutation leaves room for the new tracers behind the residents of their cell, and they are written straight
// into their sorted slots (no second sort: re-sorting 67 M tracers for a few thousand new ones took 4.4 ms).
static int sort_tracers(pl_ctx* ctx, PlStepState* S, double z0, double hz, double x0, double hx, const SortOpts& o) {
    const PlGeom& g = ctx->geom.d;
    const int ncz = (g.gi0 + g.lnz >= g.nz) ? g.lnz - 1 : g.lnz;        // cells of the block = its nodes (the last block owns one more node)
    const int ncx = (g.gj0 + g.lnx >= g.nx) ? g.lnx - 1 : g.lnx;
    const int nc = ncz * ncx, m = nc + 10;
    S->crow0 = g.gi0; S->ccol0 = g.gj0;
    long long n = S->n;
    if (n >= (1LL << 31)) return pl_fail(ctx, "sort_tracers: more than 2^31 tracers per GPU");
    // epoch layout + lazy columns (see above) where the resident step is the only reader until the next sort
    const bool epoch_ok = !o.full && epoch_length() > 0 && (ctx->nranks == 1 || o.premigrated) && !o.del_outside && ctx->geom.uniform && scatter_cells_on();
    if (!epoch_ok) PL_TRY(ensure_current(ctx, S));
    else {
        PL_TRY(flush_lazy(ctx, S));                     // (pl_step has dropped them already: rewritten before anything reads them)
        if (S->epoch_on && S->epoch_age >= epoch_length()) PL_TRY(relayout(ctx, S));
    }
    if (!S->cell_count || S->ncz != ncz || S->ncx != ncx) {
        if (o.keys_ready) return pl_fail(ctx, "sort_tracers: sort keys announced for another sort grid (internal error)");
        for (int** q : {&S->cell_count, &S->cell_start, &S->block_sums, &S->need, &S->need_off, &S->need_flag, &S->need_rank, &S->cell_res}) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
        const int nb = (m + 1023) / 1024;
        for (int** q : {&S->cell_count, &S->cell_start, &S->need, &S->need_off, &S->need_flag, &S->need_rank, &S->cell_res})
            PL_HIP(ctx, hipMalloc((void**)q, (size_t)m * sizeof(int)));
        PL_HIP(ctx, hipMalloc((void**)&S->block_sums, (size_t)(nb + 1) * sizeof(int)));
        S->ncz = ncz; S->ncx = ncx;
    }
    S->sort_cells = nc;
    if (!o.keys_ready) {
        PL_HIP(ctx, hipMemsetAsync(S->cell_count, 0, (size_t)m * sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_cell_count, grid1d(n), dim3(256), 0, ctx->stream, n, S->tz, S->tx, z0, hz, x0, hx, coords_z(ctx, S, 0),
                           coords_x(ctx, S, 0), ncz, ncx, S->crow0, S->ccol0, g.nz - 1, g.nx - 1, S->cell, S->cell_count, o.del_outside, o.Lz, o.Lx);
    }
    // ---- census + deficits (the counts ARE the census)
    long long ninj = 0; int ndef = 0;
    double id0 = 0.0; bool strict = false;
    const bool inj = o.inject && o.inject->tracdens_min > 0 && o.inject->tracdens > 0;
    if (inj) {
        const pl_step_config* cfg = o.inject;
        // resident counts of the cells as the deficit kernel expects them (start[c+1]-start[c]): scan the counts first
        scan_ints(ctx, S, m, S->cell_count, S->cell_res);
        hipLaunchKernelGGL(k_deficit, dim3((nc + 1 + 255) / 256), dim3(256), 0, ctx->stream, nc, ncx, 0, ncz, S->cell_res, cfg->tracdens,
                           cfg->tracdens_min, S->need, S->need_flag);
        scan_ints(ctx, S, nc + 1, S->need, S->need_off);
        scan_ints(ctx, S, nc + 1, S->need_flag, S->need_rank);
        int h2[2] = {0, 0};
        PL_HIP(ctx, hipMemcpyAsync(&h2[0], S->need_off + nc, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipMemcpyAsync(&h2[1], S->need_rank + nc, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ninj = h2[0]; ndef = h2[1];
        // new IDs in global (rank-major) cell order: rank r continues after the ranks before it
        const int R = ctx->nranks;
        std::vector<double> cnt((size_t)2 * R, 0.0);
        cnt[2 * ctx->rank] = (double)ninj; cnt[2 * ctx->rank + 1] = ndef;
        if (R > 1) {                                              // every rank's (new tracers, refilled cells): summed on the device
            PL_TRY(step_counts(ctx, S, 2 * R));
            hipLaunchKernelGGL(k_counts_from_ints, dim3(1), dim3(64), 0, ctx->stream, 1, (const int*)(S->need_off + nc), 0, S->counts_dev, 2 * ctx->rank);
            hipLaunchKernelGGL(k_counts_from_ints, dim3(1), dim3(64), 0, ctx->stream, 1, (const int*)(S->need_rank + nc), 0, S->counts_dev, 2 * ctx->rank + 1);
            PL_TRY(pl_comm_allreduce_dev(ctx, S->counts_dev, 2 * R, 0));
            PL_HIP(ctx, hipMemcpyAsync(cnt.data(), S->counts_dev, cnt.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        double before = 0.0, total = 0.0, def_before = 0.0, def_total = 0.0;
        for (int q = 0; q < R; q++) {
            if (q < ctx->rank) { before += cnt[2 * q]; def_before += cnt[2 * q + 1]; }
            total += cnt[2 * q]; def_total += cnt[2 * q + 1];
        }
        if (total > 0.0) {
            if (!S->max_id_valid) {                                // a deletion may have removed the holder of the maximum
                double mx[1] = {-1.0};
                if (n > 0) {
                    const int nbm = 256;
                    if (!S->partial || S->hpartial.size() < (size_t)3 * 4096) {
                        if (S->partial) (void)hipFree(S->partial);
                        PL_HIP(ctx, hipMalloc((void**)&S->partial, 3 * 4096 * sizeof(double)));
                        S->hpartial.resize(3 * 4096);
                    }
                    hipLaunchKernelGGL(k_max1d, dim3(nbm), dim3(256), 0, ctx->stream, n, S->f[TR__ID], S->partial);
                    PL_HIP(ctx, hipMemcpyAsync(S->hpartial.data(), S->partial, nbm * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    for (int k = 0; k < nbm; k++) mx[0] = std::fmax(mx[0], S->hpartial[k]);
                }
                PL_TRY(pl_allreduce_host(ctx, mx, 1, 2));
                S->max_id = mx[0]; S->max_id_valid = true;
            }
            strict = cfg->inject_unique_ids == 0;
            // strict: IDs start AT the current maximum and every refilled cell repeats one (pylamp2.py:621-622)
            id0 = strict ? S->max_id + before - def_before : S->max_id + 1.0 + before;
            S->max_id += strict ? total - def_total : total;
        }
        if (ninj > 0) {
            const long long used = std::max(n, (epoch_ok && S->epoch_on) ? S->epoch_len : n);
            PL_TRY(grow_tracers(ctx, S, used + ninj, used));
        }
    }
    // ---- offsets of the sorted arrays: residents (+ room for the new tracers of a cell right behind them)
    if (ninj > 0) {
        hipLaunchKernelGGL(k_add_need, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, nc, m, S->cell_count, S->need, S->cell_res);
        scan_ints(ctx, S, m, S->cell_res, S->cell_start);
        PL_HIP(ctx, hipMemcpyAsync(S->cell_res, S->cell_count, (size_t)m * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));   // residents per cell
    } else {
        scan_ints(ctx, S, m, S->cell_count, S->cell_start);
    }
    PL_HIP(ctx, hipMemsetAsync(S->cell_count, 0, (size_t)m * sizeof(int), ctx->stream));   // reused as fill counters
    if (n > 0 && epoch_ok) {                                // positions, temperature and slot: placement and permutation in ONE pass
        PermArgs pa{}; pa.na = 3;
        pa.in[0] = S->tz; pa.out[0] = S->tz2; pa.in[1] = S->tx; pa.out[1] = S->tx2; pa.in[2] = S->f[TR_TMP]; pa.out[2] = S->f2[TR_TMP];
        hipLaunchKernelGGL(k_place_permute, grid1d(n), dim3(256), 0, ctx->stream, n, S->cell, S->cell_start, S->cell_count, S->dest, pa,
                           S->epoch_on ? (const int*)S->slot : (const int*)nullptr, S->slot2);
        std::swap(S->tz, S->tz2); std::swap(S->tx, S->tx2);
        std::swap(S->vtz, S->tmp[0]); std::swap(S->vtx, S->tmp[1]);         // (lazy: the old values stay in tmp[0..1] / f2[RHO], f2[ETA])
        std::swap(S->slot, S->slot2);
        for (int k : {(int)TR_TMP, (int)TR_RHO, (int)TR_ETA}) std::swap(S->f[k], S->f2[k]);
    } else if (n > 0) {
        hipLaunchKernelGGL(k_cell_place, grid1d(n), dim3(256), 0, ctx->stream, n, S->cell, S->cell_start, S->cell_count, S->dest);
        // classic: positions, the 13 fields and the velocities of the last advection travel together
        const double* src[17]; double* dst[17];
        src[0] = S->tz; dst[0] = S->tz2; src[1] = S->tx; dst[1] = S->tx2;
        for (int k = 0; k < NFTRAC; k++) { src[2 + k] = S->f[k]; dst[2 + k] = S->f2[k]; }
        src[15] = S->vtz; dst[15] = S->tmp[0]; src[16] = S->vtx; dst[16] = S->tmp[1];
        for (int k0 = 0; k0 < 17; k0 += 5) {
            PermArgs pa{}; pa.na = std::min(5, 17 - k0);
            for (int k = 0; k < pa.na; k++) { pa.in[k] = src[k0 + k]; pa.out[k] = dst[k0 + k]; }
            hipLaunchKernelGGL(k_permute, grid1d(n), dim3(256), 0, ctx->stream, n, S->dest, pa);
        }
        hipLaunchKernelGGL(k_permute_int, grid1d(n), dim3(256), 0, ctx->stream, n, S->dest, S->orig, S->orig2);
        std::swap(S->tz, S->tz2); std::swap(S->tx, S->tx2); std::swap(S->orig, S->orig2);
        std::swap(S->vtz, S->tmp[0]); std::swap(S->vtx, S->tmp[1]);
        for (int k = 0; k < NFTRAC; k++) std::swap(S->f[k], S->f2[k]);
    }
    if (epoch_ok) {
        if (!S->epoch_on) S->epoch_len = n;                 // the sort has just opened the epoch: slot = the index before it
        S->epoch_age = S->epoch_on ? S->epoch_age + 1 : 1; S->epoch_on = true;
        S->lazy_pending = true; S->lazy_n = n; S->lazy_inject = ninj > 0;
    }
    if (ninj > 0) {
        InjectSorted a{};
        a.nc = nc; a.ncx = ncx; a.crow0 = S->crow0; a.ccol0 = S->ccol0; a.start = S->cell_start; a.count = S->cell_res;
        a.need = S->need; a.off = S->need_off; a.rank = strict ? S->need_rank : nullptr;
        a.tz = S->tz; a.tx = S->tx; for (int k = 0; k < NFTRAC; k++) a.f[k] = S->f[k];
        a.vtz = S->vtz; a.vtx = S->vtx; a.orig = S->orig;
        a.slot = epoch_ok ? S->slot : nullptr; a.lazy = epoch_ok ? 1 : 0;
        a.z0 = z0; a.hz = hz; a.x0 = x0; a.hx = hx; a.seed = o.inject->inject_seed; a.step = (unsigned)o.it; a.id0 = id0;
        a.n_old = (int)(epoch_ok ? S->epoch_len : n);        // epoch layout: the constants of new tracers go behind everything the epoch holds
        if (epoch_ok) S->epoch_len += ninj;
        a.zc = coords_z(ctx, S, 0); a.xc = coords_x(ctx, S, 0);
        hipLaunchKernelGGL(k_inject_sorted, dim3((nc + 63) / 64), dim3(64), 0, ctx->stream, a);
        S->n = n + ninj;
    }
    if (o.ninjected) *o.ninjected = ninj;
    PL_HIP(ctx, hipGetLastError());
    if (o.removed) *o.removed = 0;
    if (o.drop_trash) {                                   // (several ranks: the tracers that have left for a neighbour block)
        int keep = 0;
        PL_HIP(ctx, hipMemcpyAsync(&keep, S->cell_start + nc + 8, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        S->n = keep;
    }
    if (o.del_outside && n > 0) {
        int h[2] = {0, 0};
        PL_HIP(ctx, hipMemcpyAsync(h, S->cell_start + nc + 8, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int alive_n = h[0], all_n = h[1];
        if (alive_n < all_n) {
            // the survivors are the first alive_n entries; renumber `orig` so that downloads keep the caller's
            // relative order (np.delete keeps it, pylamp2.py:578-580).  cell / dest are free after the sort.
            PL_HIP(ctx, hipMemsetAsync(S->cell, 0, (size_t)all_n * sizeof(int), ctx->stream));
            if (alive_n > 0) hipLaunchKernelGGL(k_mark_alive, grid1d(alive_n), dim3(256), 0, ctx->stream, (long long)alive_n, S->orig, S->cell);
            const int nbb = (all_n + 1023) / 1024;
            int* bs = nullptr;
            PL_HIP(ctx, hipMalloc((void**)&bs, (size_t)(nbb + 1) * sizeof(int)));
            hipLaunchKernelGGL(k_scan_block, dim3(nbb), dim3(256), 0, ctx->stream, all_n, S->cell, S->dest, bs);
            hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, ctx->stream, nbb, bs);
            hipLaunchKernelGGL(k_scan_add, dim3((all_n + 255) / 256), dim3(256), 0, ctx->stream, all_n, S->dest, bs, 0);
            if (alive_n > 0) hipLaunchKernelGGL(k_remap_orig, grid1d(alive_n), dim3(256), 0, ctx->stream, (long long)alive_n, S->orig, S->dest);
            PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(bs);
            if (o.removed) *o.removed = all_n - alive_n;
            S->n = alive_n;
            S->max_id_valid = false;
        }
    }
    if (o.del_outside && ctx->nranks > 1) {        // collective: a deletion anywhere invalidates the cached maximum ID everywhere
        double v[1] = {S->max_id_valid ? 0.0 : 1.0};
        PL_TRY(pl_allreduce_host(ctx, v, 1, 2));
        if (v[0] > 0.0) S->max_id_valid = false;
    }
    return 0;
}

// Several ranks: send the tracers in the 8 leaver buckets to the neighbour blocks, append what arrives, then sort
// again.  (Tracers move less than a cell per step, so a neighbour block is always the destination; one that jumps
// further is simply forwarded again at the next step.)  The arrays grow when the arrivals do not fit.
// The census + refill of o.inject happens in the final sort, when every tracer is on its owner.
static int migrate_tracers(pl_ctx* ctx, PlStepState* S, double z0, double hz, double x0, double hx, const SortOpts& o) {
    if (ctx->nranks <= 1) return 0;
    static const int DZ[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, DX[8] = {0, 0, -1, 1, -1, 1, -1, 1}, OPP[8] = {1, 0, 3, 2, 7, 6, 5, 4};
    const int nc = S->sort_cells, R = ctx->nranks;
    int h[10];
    PL_HIP(ctx, hipMemcpyAsync(h, S->cell_start + nc, 10 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const long long stay = h[0];
    // who sends how many to whom: slot [r*8 + k] = tracers rank r sends towards direction k
    std::vector<double> cnt((size_t)8 * R, 0.0);
    PL_TRY(step_counts(ctx, S, 8 * R));
    hipLaunchKernelGGL(k_counts_from_ints, dim3(1), dim3(64), 0, ctx->stream, 8, (const int*)(S->cell_start + nc), 1, S->counts_dev, 8 * ctx->rank);
    PL_TRY(pl_comm_allreduce_dev(ctx, S->counts_dev, 8 * R, 0));
    PL_HIP(ctx, hipMemcpyAsync(cnt.data(), S->counts_dev, cnt.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int peer[8]; long long nsend[8], nrecv[8], roff[8]; long long incoming = 0;
    for (int k = 0; k < 8; k++) {
        const int qz = ctx->pz + DZ[k], qx = ctx->px + DX[k];
        peer[k] = (qz < 0 || qz >= ctx->Pz || qx < 0 || qx >= ctx->Px) ? -1 : qz * ctx->Px + qx;
        nsend[k] = h[k + 1] - h[k];
        nrecv[k] = peer[k] < 0 ? 0 : (long long)cnt[(size_t)8 * peer[k] + OPP[k]];
        if (peer[k] < 0 && nsend[k] > 0) return pl_fail(ctx, "migrate_tracers: a tracer left through a domain wall (internal error)");
        roff[k] = incoming; incoming += nrecv[k];
    }
    const long long n_all = S->n;
    PL_TRY(grow_tracers(ctx, S, std::max(stay + incoming, n_all), n_all));    // collective decision: every rank knows its own need
    const int NC = 17;
    double* cols[NC]; double* spare[NC];
    cols[0] = S->tz; cols[1] = S->tx; spare[0] = S->tz2; spare[1] = S->tx2;
    for (int k = 0; k < NFTRAC; k++) { cols[2 + k] = S->f[k]; spare[2 + k] = S->f2[k]; }
    cols[15] = S->vtz; spare[15] = S->tmp[0]; cols[16] = S->vtx; spare[16] = S->tmp[1];
    std::vector<PlMsg> msgs;
    for (int k = 0; k < 8; k++) {
        if (peer[k] < 0 || (nsend[k] == 0 && nrecv[k] == 0)) continue;
        for (int c = 0; c < NC; c++) msgs.push_back(PlMsg{peer[k], cols[c] + h[k], nsend[k], spare[c] + roff[k], nrecv[k]});
    }
    PL_TRY(pl_comm_sendrecv(ctx, msgs.data(), (int)msgs.size()));
    if (incoming > 0)
        for (int c = 0; c < NC; c++)
            PL_HIP(ctx, hipMemcpyAsync(cols[c] + stay, spare[c], (size_t)incoming * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    S->n = stay + incoming;
    hipLaunchKernelGGL(k_iota, grid1d(S->n), dim3(256), 0, ctx->stream, S->n, S->orig, 0);
    return sort_tracers(ctx, S, z0, hz, x0, hx, o);
}

// ---- several ranks, regular grid: migration BEFORE the sort ---------------------------------------------------------------------------
// k_rk4's epilogue has keyed every tracer (cell of this block, or one of the 8 leaver buckets) and counted the keys.  Instead of
// sort -> send the leaver buckets -> append -> sort again (two full sorts of 17 columns per step), the leavers are collected straight
// from the unsorted arrays into per-direction send buffers (a few thousand tracers: the rim of the block), re-keyed as trash, the
// arrivals are appended with their keys, and ONE sort -- the epoch-layout sort of one rank: positions, temperature and a 4-byte slot
// per tracer -- places everything and drops the trash.  Constants of arrivals are appended to the epoch-ordered arrays like those of
// injected tracers; the entries of tracers that have left stay behind unreferenced until the next re-layout.
struct LeaverArgs {
    long long n; int nc;
    int* cell; int* count; const int* slot;             // slot NULL: every column at the tracer's own index
    const double* col[17];                              // tz, tx, f[0..12], vtz, vtx
    int* fill;                                          // 8 counters
    long long off[8], cnt[8];                           // element offset (in tracers) and capacity of each direction's segment
    double* buf;
};
__global__ __launch_bounds__(256) void k_pack_leavers(LeaverArgs a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n) return;
    const int c = a.cell[t];
    if (c < a.nc || c >= a.nc + 8) return;
    const int k = c - a.nc;
    const int q = atomicAdd(&a.fill[k], 1);
    if (q >= a.cnt[k]) return;                          // (cannot happen: the counts come from the same keys)
    const long long e = a.slot ? (long long)a.slot[t] : t;
    double* seg = a.buf + 17 * a.off[k];
    for (int m = 0; m < 17; m++) {
        const bool at_epoch = m >= 2 && m < 15 && tracer_const(m - 2);
        seg[(long long)m * a.cnt[k] + q] = a.col[m][at_epoch ? e : t];
    }
    a.cell[t] = a.nc + 8;                               // trash: the sort drops it
    atomicAdd(&a.count[a.nc + 8], 1); atomicSub(&a.count[c], 1);
}
struct ArrivalArgs {
    long long first, e0; long long off[8], cnt[8]; int ndir;      // sorted-order index / epoch index of the first arrival
    double* col[17]; int* slot; int* orig; const double* buf;
};
__global__ __launch_bounds__(256) void k_unpack_arrivals(ArrivalArgs a, long long total) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= total) return;
    int k = 0;
    for (int d = 0; d < 8; d++) if (r >= a.off[d] && r < a.off[d] + a.cnt[d]) k = d;
    const long long q = r - a.off[k];
    const double* seg = a.buf + 17 * a.off[k];
    const long long t = a.first + r, e = a.e0 + r;
    for (int m = 0; m < 17; m++) {
        const bool at_epoch = a.slot && m >= 2 && m < 15 && tracer_const(m - 2);
        a.col[m][at_epoch ? e : t] = seg[(long long)m * a.cnt[k] + q];
    }
    if (a.slot) a.slot[t] = (int)e;
    a.orig[a.slot ? e : t] = (int)(a.slot ? e : t);
}
static int migrate_presort(pl_ctx* ctx, PlStepState* S, double z0, double hz, double x0, double hx) {
    static const int DZ[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, DX[8] = {0, 0, -1, 1, -1, 1, -1, 1}, OPP[8] = {1, 0, 3, 2, 7, 6, 5, 4};
    const PlGeom& g = ctx->geom.d;
    const int nc = S->sort_cells, R = ctx->nranks;
    const long long n_all = S->n;
    // who sends how many to whom (slot [r*8 + k] = tracers rank r sends towards direction k): summed over the ranks on the device
    std::vector<double> cnt((size_t)8 * R, 0.0);
    PL_TRY(step_counts(ctx, S, 8 * R));
    hipLaunchKernelGGL(k_counts_from_ints, dim3(1), dim3(64), 0, ctx->stream, 8, (const int*)(S->cell_count + nc), 0, S->counts_dev, 8 * ctx->rank);
    PL_TRY(pl_comm_allreduce_dev(ctx, S->counts_dev, 8 * R, 0));
    PL_HIP(ctx, hipMemcpyAsync(cnt.data(), S->counts_dev, cnt.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int peer[8]; long long nsend[8], nrecv[8], soff[8], roff[8], outgoing = 0, incoming = 0;
    for (int k = 0; k < 8; k++) {
        const int qz = ctx->pz + DZ[k], qx = ctx->px + DX[k];
        peer[k] = (qz < 0 || qz >= ctx->Pz || qx < 0 || qx >= ctx->Px) ? -1 : qz * ctx->Px + qx;
        nsend[k] = (long long)cnt[(size_t)8 * ctx->rank + k];
        nrecv[k] = peer[k] < 0 ? 0 : (long long)cnt[(size_t)8 * peer[k] + OPP[k]];
        if (peer[k] < 0 && nsend[k] > 0) return pl_fail(ctx, "migrate_presort: a tracer left through a domain wall (internal error)");
        soff[k] = outgoing; outgoing += nsend[k]; roff[k] = incoming; incoming += nrecv[k];
    }
    double *sb = nullptr, *rb = nullptr;
    PL_TRY(pl_buf(ctx, "mig_send", (size_t)std::max<long long>(17 * outgoing, 1) * sizeof(double), &sb, false));
    PL_TRY(pl_buf(ctx, "mig_recv", (size_t)std::max<long long>(17 * incoming, 1) * sizeof(double), &rb, false));
    const long long used = std::max(n_all, S->epoch_on ? S->epoch_len : n_all);
    PL_TRY(grow_tracers(ctx, S, used + incoming, used));         // (collective in effect: every rank knows its own need)
    double* cols[17];
    cols[0] = S->tz; cols[1] = S->tx; for (int k = 0; k < NFTRAC; k++) cols[2 + k] = S->f[k]; cols[15] = S->vtz; cols[16] = S->vtx;
    if (outgoing > 0) {
        double* fc;
        PL_TRY(pl_buf(ctx, "mig_fill", 64, &fc, false));
        PL_HIP(ctx, hipMemsetAsync(fc, 0, 64, ctx->stream));
        LeaverArgs a{};
        a.n = n_all; a.nc = nc; a.cell = S->cell; a.count = S->cell_count; a.slot = S->epoch_on ? S->slot : nullptr;
        for (int m = 0; m < 17; m++) a.col[m] = cols[m];
        a.fill = (int*)fc; a.buf = sb;
        for (int k = 0; k < 8; k++) { a.off[k] = soff[k]; a.cnt[k] = nsend[k]; }
        hipLaunchKernelGGL(k_pack_leavers, grid1d(n_all), dim3(256), 0, ctx->stream, a);
    }
    std::vector<PlMsg> msgs;
    for (int k = 0; k < 8; k++) {
        if (peer[k] < 0 || (nsend[k] == 0 && nrecv[k] == 0)) continue;
        msgs.push_back(PlMsg{peer[k], sb + 17 * soff[k], 17 * nsend[k], rb + 17 * roff[k], 17 * nrecv[k]});
    }
    PL_TRY(pl_comm_sendrecv(ctx, msgs.data(), (int)msgs.size()));
    if (incoming > 0) {
        ArrivalArgs a{};
        a.first = n_all; a.e0 = S->epoch_on ? S->epoch_len : n_all;
        for (int m = 0; m < 17; m++) a.col[m] = cols[m];
        a.slot = S->epoch_on ? S->slot : nullptr; a.orig = S->orig; a.buf = rb;
        for (int k = 0; k < 8; k++) { a.off[k] = roff[k]; a.cnt[k] = nrecv[k]; }
        hipLaunchKernelGGL(k_unpack_arrivals, grid1d(incoming), dim3(256), 0, ctx->stream, a, incoming);
        // their sort keys, counted into the same table (the arrivals may lie one block further already: forwarded at the next step)
        hipLaunchKernelGGL(k_cell_count, grid1d(incoming), dim3(256), 0, ctx->stream, incoming, (const double*)(S->tz + n_all), (const double*)(S->tx + n_all),
                           z0, hz, x0, hx, (const double*)nullptr, (const double*)nullptr, S->ncz, S->ncx, S->crow0, S->ccol0, g.nz - 1, g.nx - 1,
                           S->cell + n_all, S->cell_count, 0, 0.0, 0.0);
        if (S->epoch_on) S->epoch_len += incoming;
    }
    S->n = n_all + incoming;
    PL_HIP(ctx, hipGetLastError());
    return 0;
}

// un-permute resident arrays into the spare buffers so that downloads come out in the caller's order
static void unpermute(pl_ctx* ctx, PlStepState* S, int na, const double* const* in, double* const* out) {
    for (int k0 = 0; k0 < na; k0 += 5) {
        PermArgs pa{}; pa.na = std::min(5, na - k0);
        for (int k = 0; k < pa.na; k++) { pa.in[k] = in[k0 + k]; pa.out[k] = out[k0 + k]; }
        hipLaunchKernelGGL(k_permute, grid1d(S->n), dim3(256), 0, ctx->stream, S->n, S->orig, pa);
    }
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// scatter a set of tracer fields onto a staggered node set, writing ring planes
static int scatter_to_planes(pl_ctx* ctx, PlStepState* S, int nf, const int* fidx, const int* schemes, double z0, double hz,
                             double x0, double hx, double* const* planes, int stag_z = 0, int stag_x = 0) {
    const PlGeom& g = ctx->geom.d;
    PlScatterArgs a{};
    a.n = S->n; a.tz = S->tz; a.tx = S->tx; a.nf = nf;
    for (int k = 0; k < nf; k++) { a.f[k] = fidx[k] >= 0 ? S->f[fidx[k]] : S->tmp[-fidx[k] - 1]; a.scheme[k] = schemes[k]; }
    a.z0 = z0; a.hz = hz; a.x0 = x0; a.hx = hx; a.nz = g.nz; a.nx = g.nx;
    a.zc = coords_z(ctx, S, stag_z); a.xc = coords_x(ctx, S, stag_x);
    a.cell_start = S->cell_start; a.ncz = S->ncz; a.ncx = S->ncx; a.crow0 = S->crow0; a.ccol0 = S->ccol0;     // cell-sorted (sort_tracers)
    a.stag_z = stag_z; a.stag_x = stag_x;
    return pl_scatter_device(ctx, a, planes, g.pitch, pl_idx(g, 0, 0), &g);
}

// The scatters of a time step in ONE pass over the cell-sorted tracers (pl_mic_cells.hip; regular grids).
//   variant 0 (heat on):  nodes {rho AW, log eta GW, cp, T, H, mat AW} -> out[0..5], centres {log eta GW} -> out[6],
//                         z-mid {k AW} -> out[7], x-mid {k AW} -> out[8]                      (pylamp2.py:307-314)
//   variant 1 (heat off): nodes {rho AW, log eta GW} -> out[0..1], centres {log eta, UNWEIGHTED} -> out[2]   (pylamp2.py:316-319)
//   variant 2:            nodes {fn[0] AW} -> out[0]                                          (subgrid diffusion, pylamp2.py:477)
// PYLAMP_SCATTER=0 keeps the one-set-per-pass kernels (k_scatter_binned) -- the cross-check of the tests.
static int scatter_cells(pl_ctx* ctx, PlStepState* S, int variant, const double* const* fn, const double* fm, double z0, double hz,
                         double x0, double hx, double* const* out, const int* ix = nullptr, unsigned ind = 0u) {
    const PlGeom& g = ctx->geom.d;
    const int AW = PL_AVG_ARITHMETIC | PL_AVG_WEIGHTED, GW = PL_AVG_GEOMETRIC | PL_AVG_WEIGHTED, G0 = PL_AVG_GEOMETRIC;
    const int nfn = variant == 0 ? 6 : (variant == 1 ? 2 : 1);
    const bool cen = variant <= 1, mid = variant == 0;
    const int nplanes = (1 + nfn) + (cen ? 2 : 0) + (mid ? 4 : 0);
    PlScatterCellsArgs a{};
    a.tz = S->tz; a.tx = S->tx;
    for (int k = 0; k < nfn; k++) a.fn[k] = fn[k];
    a.fm = fm; a.ix = ind ? ix : nullptr; a.ind = ix ? ind : 0u;
    a.z0 = z0; a.hz = hz; a.x0 = x0; a.hx = hx; a.nz = g.nz; a.nx = g.nx;
    a.row0 = g.gi0 - 1; a.nrows = g.lnz + 2; a.col0 = g.gj0 - 1; a.ncols = g.lnx + 2;
    a.cell_start = S->cell_start; a.ncz = S->ncz; a.ncx = S->ncx; a.crow0 = S->crow0; a.ccol0 = S->ccol0;
    const size_t N = (size_t)a.nrows * a.ncols;
    a.N = (long long)N;
    double* accbuf;
    PL_TRY(pl_buf(ctx, "scatter_acc", (size_t)nplanes * N * sizeof(double), &accbuf, false));
    // every node of the block is written by exactly one wave (plain stores); only the ring that the reverse halo reads on
    // several ranks needs zeros where no tracer contributes
    if (ctx->nranks > 1) PL_HIP(ctx, hipMemsetAsync(accbuf, 0, (size_t)nplanes * N * sizeof(double), ctx->stream));
    a.accN = accbuf; a.accC = accbuf + (size_t)(1 + nfn) * N; a.accZ = a.accC + (cen ? 2 : 0) * N; a.accX = a.accZ + 2 * N;
    double* sc;
    PL_TRY(pl_buf(ctx, "scatter_slow_count", 64, &sc, false));
    a.slow_count = (int*)sc; a.slow_list = S->dest; a.slow_cap = (int)std::min<long long>(S->cap, 0x7fffffff);   // `dest` is free outside the sort
    a.dbg = 0;
#ifdef PL_SC_ABLATE          // (ablation builds only: the bits switch parts of the kernel OFF -- never in the shipped library)
    a.dbg = getenv("PYLAMP_SC_DBG") ? atoi(getenv("PYLAMP_SC_DBG")) : 0;
#endif
    PL_TRY(pl_scatter_cells_device(ctx, a, variant));
    if (ctx->nranks > 1)       // what I accumulated for nodes of the neighbour blocks is added to their accumulators
        PL_TRY(pl_halo_generic(ctx, g.lnz, g.lnx, accbuf + a.ncols + 1, a.ncols, nplanes, (long long)N, 1, true));
    PlScatterFinalArgs f{};
    f.nz = g.lnz; f.nx = g.lnx; f.acc_pitch = a.ncols; f.out_pitch = g.pitch; f.out_off = pl_idx(g, 0, 0);
    const double* base = accbuf + a.ncols + 1;                 // owned node (0,0)
    auto add = [&](const double* set, int k, int scheme, double* o) {
        f.acc[f.nf] = set + (size_t)(1 + k) * N; f.den[f.nf] = set; f.scheme[f.nf] = scheme; f.out[f.nf] = o; f.nf++;
    };
    if (variant == 0) {
        const int sch[6] = {AW, GW, AW, AW, AW, AW};
        for (int k = 0; k < 6; k++) add(base, k, sch[k], out[k]);
        add(base + 7 * N, 0, GW, out[6]); add(base + 9 * N, 0, AW, out[7]); add(base + 11 * N, 0, AW, out[8]);
    } else if (variant == 1) {
        add(base, 0, AW, out[0]); add(base, 1, GW, out[1]); add(base + 3 * N, 0, G0, out[2]);
    } else add(base, 0, AW, out[0]);
    const int nout = f.nf;
    pl_launch_scatter_finalize_multi(ctx, f);
    PL_HIP(ctx, hipGetLastError());
    if (ctx->nranks > 1)
        for (int k = 0; k < nout; k++)
            PL_TRY(pl_halo(ctx, g, out[k], 1, g.plane, std::min(PL_RING, std::min(g.lnz, g.lnx))));
    return 0;
}

// ---- the marker stages of a time step; pl_step runs them in order, the pl_resident_* entry points one at a time ------------
struct StepPlanes { double *rho, *etas, *etan, *cp, *T, *H, *mat, *kz, *kx, *newT, *c, *sgc, *dT; };
static int step_planes(pl_ctx* ctx, StepPlanes& P) {
    const size_t pb = (size_t)ctx->geom.d.plane * sizeof(double);
    PL_TRY(pl_buf(ctx, "rho", pb, &P.rho)); PL_TRY(pl_buf(ctx, "etas", pb, &P.etas)); PL_TRY(pl_buf(ctx, "etan", pb, &P.etan));
    PL_TRY(pl_buf(ctx, "cp", pb, &P.cp)); PL_TRY(pl_buf(ctx, "f_T", pb, &P.T)); PL_TRY(pl_buf(ctx, "H", pb, &P.H));
    PL_TRY(pl_buf(ctx, "mat", pb, &P.mat)); PL_TRY(pl_buf(ctx, "kz", pb, &P.kz)); PL_TRY(pl_buf(ctx, "kx", pb, &P.kx));
    PL_TRY(pl_buf(ctx, "temp", pb, &P.newT)); PL_TRY(pl_buf(ctx, "heat_c", pb, &P.c)); PL_TRY(pl_buf(ctx, "sgc", pb, &P.sgc));
    PL_TRY(pl_buf(ctx, "dT", pb, &P.dT));
    return 0;
}
struct StepGrid { double z0, hz, x0, hx; };
static StepGrid step_grid(pl_ctx* ctx) {
    const int nz = ctx->nz, nx = ctx->nx;
    StepGrid q; q.z0 = ctx->geom.zc[0]; q.x0 = ctx->geom.xc[0];
    q.hz = (ctx->geom.zc[nz - 1] - q.z0) / (nz - 1); q.hx = (ctx->geom.xc[nx - 1] - q.x0) / (nx - 1);
    return q;
}

// stage 1: tracer properties (pylamp2.py:291-303)
static int stage_props(pl_ctx* ctx, PlStepState* S, const pl_step_config* cfg) {
    const long long n = S->n;
    hipLaunchKernelGGL(k_property_update, grid1d(n), dim3(256), 0, ctx->stream, n, S->f[TR_TMP], S->f[TR_RH0], S->f[TR_ALP],
                       S->f[TR_ACE], S->f[TR_ET0], S->f[TR_RHO], S->f[TR_ETA], cfg->tdep_rho, cfg->tdep_eta, cfg->tref,
                       cfg->etamin, cfg->etamax, S->tmp[2], S->epoch_on ? (const int*)S->slot : (const int*)nullptr);
    PL_HIP(ctx, hipGetLastError());
    return 0;
}

// stage 2: tracer -> grid (pylamp2.py:307-319); S->tmp[2] holds log(eta) (stage 1)
static int stage_scatter(pl_ctx* ctx, PlStepState* S, const pl_step_config* cfg, int it, const StepPlanes& P) {
    const PlGeom& g = ctx->geom.d;
    const StepGrid q = step_grid(ctx);
    const double z0 = q.z0, hz = q.hz, x0 = q.x0, hx = q.hx;
    const int AW = PL_AVG_ARITHMETIC | PL_AVG_WEIGHTED, GW = PL_AVG_GEOMETRIC | PL_AVG_WEIGHTED | PL_AVG_PRELOG;
    const int ETA_LOG = -3;                     // S->tmp[2]: log(eta) written by k_property_update
    const bool cells = scatter_cells_on() && ctx->geom.uniform;
    if (!cells) PL_TRY(ensure_current(ctx, S));       // the one-set-per-pass kernels read every column at the tracer's own index
    if (cfg->do_heatdiff) {
        if (cells) {
            const double* fn[6] = {S->f[TR_RHO], S->tmp[2], S->f[TR_HCP], S->f[TR_TMP], S->f[TR_IHT], S->f[TR_MAT]};
            double* out[9] = {P.rho, P.etas, P.cp, P.T, P.H, P.mat, P.etan, P.kz, P.kx};
            // epoch layout: cp, H, mat (fields 2, 4, 5) and the conductivity of the mid-face sets are constant columns
            PL_TRY(scatter_cells(ctx, S, 0, fn, S->f[TR_HCD], z0, hz, x0, hx, out, S->epoch_on ? (const int*)S->slot : (const int*)nullptr,
                                 (1u << 2) | (1u << 4) | (1u << 5) | (1u << 31)));
        } else {
            const int fi[6] = {TR_RHO, ETA_LOG, TR_HCP, TR_TMP, TR_IHT, TR_MAT};
            const int sc[6] = {AW, GW, AW, AW, AW, AW};
            double* pl6[6] = {P.rho, P.etas, P.cp, P.T, P.H, P.mat};
            PL_TRY(scatter_to_planes(ctx, S, 6, fi, sc, z0, hz, x0, hx, pl6));
            const int f1[1] = {ETA_LOG}; const int s1[1] = {GW}; double* pn[1] = {P.etan};
            PL_TRY(scatter_to_planes(ctx, S, 1, f1, s1, z0 + 0.5 * hz, hz, x0 + 0.5 * hx, hx, pn, 1, 1));
            const int f2[1] = {TR_HCD}; const int s2[1] = {AW};
            double* pk[1] = {P.kz};
            PL_TRY(scatter_to_planes(ctx, S, 1, f2, s2, z0 + 0.5 * hz, hz, x0, hx, pk, 1, 0));
            pk[0] = P.kx;
            PL_TRY(scatter_to_planes(ctx, S, 1, f2, s2, z0, hz, x0 + 0.5 * hx, hx, pk, 0, 1));
        }
        if (it > 1 && S->have_newtemp)
            hipLaunchKernelGGL(k_copy_boundary, grid2d(g), dim3(64, 4), 0, ctx->stream, g, P.newT, P.T);
    } else if (cells) {
        const double* fn[2] = {S->f[TR_RHO], S->tmp[2]};
        double* out[3] = {P.rho, P.etas, P.etan};
        PL_TRY(scatter_cells(ctx, S, 1, fn, nullptr, z0, hz, x0, hx, out));
    } else {
        const int fi[2] = {TR_RHO, ETA_LOG}; const int sc[2] = {AW, GW};
        double* pl2[2] = {P.rho, P.etas};
        PL_TRY(scatter_to_planes(ctx, S, 2, fi, sc, z0, hz, x0, hx, pl2));
        const int f1[1] = {ETA_LOG}; const int s1[1] = {PL_AVG_GEOMETRIC | PL_AVG_PRELOG}; double* pn[1] = {P.etan};   // pylamp2.py:319 (unweighted)
        PL_TRY(scatter_to_planes(ctx, S, 1, f1, s1, z0 + 0.5 * hz, hz, x0 + 0.5 * hx, hx, pn, 1, 1));
    }
    PL_HIP(ctx, hipGetLastError());
    return 0;
}

// stage 5b: temperature to the tracers (+ subgrid diffusion), pylamp2.py:436-480.  P.newT: the new nodal temperature (halo
// filled), P.T: the nodal temperature the heat system was built from.  first: the it == 1 branch (plain interpolation).
static int stage_temp_to_tracers(pl_ctx* ctx, PlStepState* S, const pl_step_config* cfg, bool first, const StepPlanes& P, double tstep) {
    const PlGeom& g = ctx->geom.d;
    const int nz = g.nz, nx = g.nx;
    const long long n = S->n;
    const StepGrid q = step_grid(ctx);
    const double dz = cfg->length[0] / (nz - 1), dx = cfg->length[1] / (nx - 1);           // pylamp2.py:87
    const int AW = PL_AVG_ARITHMETIC | PL_AVG_WEIGHTED;
    const int* ix = S->epoch_on ? S->slot : nullptr;
    double* cnt;
    PL_TRY(pl_buf(ctx, "mic_counter", 64, &cnt, false));
    PL_HIP(ctx, hipMemsetAsync(cnt, 0, 64, ctx->stream));
    PlGatherArgs ga{};
    ga.n = n; ga.tz = S->tz; ga.tx = S->tx; ga.nf = 1; ga.method = PL_INTERP_LINEAR;
    ga.defval = std::numeric_limits<double>::quiet_NaN(); ga.accumulate = 0; ga.n_outside = (unsigned long long*)cnt;
    ga.g.nz = nz; ga.g.nx = nx; ga.g.gz = S->gcoords; ga.g.gx = S->gcoords + nz;
    ga.g.zmin = ctx->geom.zc[0]; ga.g.xmin = ctx->geom.xc[0];
    ga.g.Lz = ctx->geom.zc[nz - 1] - ctx->geom.zc[0]; ga.g.Lx = ctx->geom.xc[nx - 1] - ctx->geom.xc[0];
    ga.g.rect = ctx->geom.uniform ? 0 : 1;
    ga.g.fast_uniform = ctx->geom.uniform ? 1 : 0;            // regular grid: k_gather<true> (no coordinate search, no divisions)
    ga.g.pitch = g.pitch; ga.g.off = pl_idx(g, -g.gi0, -g.gj0);      // field indices are GLOBAL
    if (first) {
        ga.fields[0] = P.newT; ga.out[0] = S->f[TR_TMP];
        pl_launch_gather(ctx, ga);
    } else {
        hipLaunchKernelGGL(k_plane_sub, grid2d(g), dim3(64, 4), 0, ctx->stream, g, P.newT, P.T, P.dT);
        PL_TRY(pl_halo(ctx, g, P.dT, 1, g.plane));
        S->have_dT = true;
        const double inv2 = (2.0 / dx) * (2.0 / dx) + (2.0 / dz) * (2.0 / dz);
        // regular grid: the two per-marker subgrid kernels ride in the gathers (the interpolated value never goes to memory:
        // 152 -> 120 B per marker over the stage)
        const bool fuse = ga.g.fast_uniform && !ga.g.rect;
        ga.fields[0] = P.dT; ga.out[0] = S->tmp[0];
        if (fuse) {
            ga.epi = 1; ga.epi_subgrid = cfg->do_subgrid_heatdiff; ga.epi_T = S->f[TR_TMP]; ga.epi_hcp = S->f[TR_HCP];
            ga.epi_rho = S->f[TR_RHO]; ga.epi_hcd = S->f[TR_HCD]; ga.epi_inv2 = inv2; ga.epi_dt = tstep;
            ga.epi_Tsub = S->tmp[1]; ga.epi_dTs = S->tmp[2]; ga.epi_ix = ix;
        }
        pl_launch_gather(ctx, ga);
        if (!fuse)
            hipLaunchKernelGGL(k_subgrid_part1, grid1d(n), dim3(256), 0, ctx->stream, n, S->f[TR_TMP], S->tmp[0], S->f[TR_HCP],
                               S->f[TR_RHO], S->f[TR_HCD], inv2, tstep, cfg->do_subgrid_heatdiff, S->tmp[1], S->tmp[2], ix);
        if (cfg->do_subgrid_heatdiff) {
            // T currently holds Told for the subgrid branch; dTs = tmp[2] -> nodes -> back to tracers
            if (scatter_cells_on() && ctx->geom.uniform) {
                const double* fn[1] = {S->tmp[2]}; double* out[1] = {P.sgc};
                PL_TRY(scatter_cells(ctx, S, 2, fn, nullptr, q.z0, q.hz, q.x0, q.hx, out));
            } else {
                const int fs[1] = {-3}; const int ss[1] = {AW}; double* ps[1] = {P.sgc};
                PL_TRY(scatter_to_planes(ctx, S, 1, fs, ss, q.z0, q.hz, q.x0, q.hx, ps));
            }
            ga.fields[0] = P.sgc; ga.out[0] = S->tmp[0];
            if (fuse) ga.epi = 2;
            pl_launch_gather(ctx, ga);
            if (!fuse) hipLaunchKernelGGL(k_subgrid_part2, grid1d(n), dim3(256), 0, ctx->stream, n, S->f[TR_TMP], S->tmp[1], S->tmp[0]);
        }
        ga.epi = 0;
    }
    if (ctx->nranks > 1) {          // several ranks: the count joins the RK4 stage's in ONE reduction over the ranks (stage_rk4)
        PL_TRY(step_flags(ctx, S));
        hipLaunchKernelGGL(k_flag_from_u64, dim3(1), dim3(1), 0, ctx->stream, (const unsigned long long*)cnt, S->flags);
        return 0;
    }
    unsigned long long nout = 0;
    PL_HIP(ctx, hipMemcpyAsync(&nout, cnt, sizeof(nout), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nout > 0) return pl_fail(ctx, "stopOnError in grid2trac");        // pylamp2.py:445,453 stopOnError=True
    return 0;
}

// stage 6b: RK4 through the advection velocities V = (Vz | Vx), the window [I0, I1] x [J0, J1] of the padded centre grid held
// as dense (I1-I0+1) x (J1-J0+1) arrays; positions -> tz2/tx2 (swapped in), tracer velocities -> vtz/vtx (pylamp2.py:547-572)
static int stage_rk4(pl_ctx* ctx, PlStepState* S, const double* V, int I0, int I1, int J0, int J1, double tstep, int fence, double Lz, double Lx,
                     bool* keys_ready = nullptr) {
    const int nz = ctx->nz, nx = ctx->nx;
    const StepGrid q = step_grid(ctx);
    const int nVr = I1 - I0 + 1, nVc = J1 - J0 + 1;
    const size_t VN = (size_t)nVr * nVc;
    PlRk4Args ra{};
    ra.n = S->n; ra.tz = S->tz; ra.tx = S->tx;
    ra.g.nz = nz + 1; ra.g.nx = nx + 1; ra.g.gz = S->gcoords + nz + nx; ra.g.gx = S->gcoords + nz + nx + (nz + 1);
    {
        const double gz0 = S->gmz[0] - (S->gmz[1] - S->gmz[0]), gx0 = S->gmx[0] - (S->gmx[1] - S->gmx[0]);
        ra.g.zmin = gz0; ra.g.xmin = gx0; ra.g.Lz = S->gmz[nz - 1] - gz0; ra.g.Lx = S->gmx[nx - 1] - gx0;
    }
    ra.g.rect = ctx->geom.uniform ? 0 : 1;
    ra.g.fast_uniform = ctx->geom.uniform ? 1 : 0; ra.g.hx_over_hz = q.hx / q.hz; ra.g.hz_over_hx = q.hz / q.hx;
    ra.g.pitch = nVc; ra.g.off = -((long long)I0 * nVc + J0);      // cell indices are GLOBAL
    ra.g.ie_lo = I0; ra.g.ie_hi = I1 - 1; ra.g.je_lo = J0; ra.g.je_hi = J1 - 1;      // cells held locally
    ra.Vz = V; ra.Vx = V + VN; ra.dt = tstep;
    ra.tz_out = S->tz2; ra.tx_out = S->tx2; ra.vz_out = S->vtz; ra.vx_out = S->vtx;
    ra.fence = fence; ra.eps = PL_EPS; ra.Lz = Lz; ra.Lx = Lx;
    if (keys_ready) {
        // the end-of-step sort's keys and counts come out of this kernel (regular grid; the sort grid of this block is unchanged
        // since the last sort): no separate pass over the new positions
        *keys_ready = false;
        const PlGeom& g = ctx->geom.d;
        const int ncz = (g.gi0 + g.lnz >= g.nz) ? g.lnz - 1 : g.lnz, ncx = (g.gj0 + g.lnx >= g.nx) ? g.lnx - 1 : g.lnx;
        const bool fuse_keys = true;           // (the sort keys as a pass of their own cost 0.35 ms at 2049^2 / 68 M tracers)
        if (fuse_keys && ctx->geom.uniform && S->cell_count && S->ncz == ncz && S->ncx == ncx && S->cell && S->n < (1LL << 31)) {
            PlSortKey& k = ra.key;
            k.on = 1; k.z0 = q.z0; k.rhz = 1.0 / q.hz; k.x0 = q.x0; k.rhx = 1.0 / q.hx;
            k.ncz = ncz; k.ncx = ncx; k.crow0 = g.gi0; k.ccol0 = g.gj0; k.gcz = g.nz - 1; k.gcx = g.nx - 1;
            k.del_outside = fence ? 0 : 1; k.Lz = Lz; k.Lx = Lx; k.cell = S->cell; k.count = S->cell_count;
            PL_HIP(ctx, hipMemsetAsync(S->cell_count, 0, (size_t)(ncz * ncx + 10) * sizeof(int), ctx->stream));
            *keys_ready = true;
        }
    }
    double* oowc = nullptr;
    if (ctx->nranks > 1) {
        PL_TRY(pl_buf(ctx, "rk4_counter", 64, &oowc, false));
        PL_HIP(ctx, hipMemsetAsync(oowc, 0, 64, ctx->stream));
        ra.n_outside_window = (unsigned long long*)oowc;
    }
    pl_launch_rk4(ctx, ra);
    PL_HIP(ctx, hipGetLastError());
    std::swap(S->tz, S->tz2); std::swap(S->tx, S->tx2);
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->nranks > 1) {      // collective: a stage that left the local velocity window means the step was not CFL-limited
        PL_TRY(step_flags(ctx, S));
        hipLaunchKernelGGL(k_flag_from_u64, dim3(1), dim3(1), 0, ctx->stream, (const unsigned long long*)oowc, S->flags + 1);
        PL_TRY(pl_comm_allreduce_dev(ctx, S->flags, 2, 0));      // [0]: grid2trac's count of this step (stage_temp_to_tracers)
        double v[2] = {0.0, 0.0};
        PL_HIP(ctx, hipMemcpyAsync(v, S->flags, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipMemsetAsync(S->flags, 0, 2 * sizeof(double), ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (v[0] > 0) return pl_fail(ctx, "stopOnError in grid2trac");        // pylamp2.py:445,453 stopOnError=True
        if (v[1] > 0) return pl_fail(ctx, "pl_step: a tracer moved by more than one cell in an RK4 stage (time step not CFL-limited); the "
                                          "block decomposition holds the advection velocity one cell around each block only");
    }
    return 0;
}

extern "C" int pl_step(pl_ctx* ctx, const pl_step_config* cfg, int it, pl_step_report* rep) {
    if (!cfg || !rep) return pl_fail(ctx, "pl_step: NULL argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PlStepState* S = state_of(ctx);
    if (S->n <= 0 && ctx->nranks == 1) return pl_fail(ctx, "pl_step: no tracers resident (call pl_tracers_upload)");
    if (!S->cell_start) return pl_fail(ctx, "pl_step: no tracers uploaded on this rank (every rank calls pl_tracers_upload, with n = 0 if need be)");
    PL_TRY(ensure_coords(ctx, S));
    PL_TRY(pl_stokes_check_bc(ctx, cfg->bcstokes));
    if (cfg->do_heatdiff) PL_TRY(pl_heat_check_bc(ctx, cfg->bcheat));
    const PlGeom& g = ctx->geom.d;
    const int nz = g.nz, nx = g.nx;
    const size_t pb = (size_t)g.plane * sizeof(double);
    memset(rep, 0, sizeof(*rep));
    const double Lz = cfg->length[0], Lx = cfg->length[1];
    const double dz = Lz / (nz - 1), dx = Lx / (nx - 1);           // pylamp2.py:87
    const double z0 = ctx->geom.zc[0], x0 = ctx->geom.xc[0];
    const double hz = (ctx->geom.zc[nz - 1] - z0) / (nz - 1), hx = (ctx->geom.xc[nx - 1] - x0) / (nx - 1);
    double t_all = now_ms(), t0;
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));

    StepPlanes P;
    PL_TRY(step_planes(ctx, P));
    double* const p_rho = P.rho; double* const p_etas = P.etas; double* const p_etan = P.etan; double* const p_cp = P.cp;
    double* const p_T = P.T; double* const p_H = P.H; double* const p_kz = P.kz; double* const p_kx = P.kx;
    double* const p_newT = P.newT; double* const p_c = P.c; double* const p_dT = P.dT;

    // tracers arrive cell-sorted (pl_tracers_upload / end of the previous step)
    if (!S->sorted) return pl_fail(ctx, "pl_step: tracers are not sorted (internal error)");
    // the columns the last sort left behind (RHO, ETA, tracer velocities) are rewritten below before anything reads them -- unless
    // this configuration does not run the stage that rewrites them
    if (S->cols_undefined) return pl_fail(ctx, "pl_step: the previous step failed half-way; upload the tracers again");
    if (ctx->geom.uniform && scatter_cells_on()) { S->cols_undefined = S->lazy_pending; S->lazy_pending = false; }     // (defined again behind stage_rk4)
    else PL_TRY(ensure_current(ctx, S));

    // ---- 1. tracer properties --------------------------------------------------------------
    t0 = now_ms();
    PL_TRY(stage_props(ctx, S, cfg));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rep->ms_props = now_ms() - t0;

    // ---- 2. tracer -> grid (pylamp2.py:307-319) ------------------------------------------------
    t0 = now_ms();
    PL_TRY(stage_scatter(ctx, S, cfg, it, P));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rep->ms_scatter = now_ms() - t0;

    // ---- 3. heat time step (pylamp2.py:339-343) and the viscosity minima of the Stokes scaling (pylamp_stokes.py:116-118): ONE reduction
    double tstep_temp = 0.0, mx;
    double mes, men;
    {
        MinMaxSpec sp[3] = {{p_etas, nullptr, nullptr, 0}, {p_etan, nullptr, nullptr, 0}, {p_kz, p_rho, p_cp, 1}};
        MinMaxOut mm[3];
        PL_TRY(reduce_minmax_multi(ctx, S, g, cfg->do_heatdiff ? 3 : 2, sp, mm));
        mes = mm[0].has_nan ? std::numeric_limits<double>::quiet_NaN() : mm[0].mn;
        men = mm[1].has_nan ? std::numeric_limits<double>::quiet_NaN() : mm[1].mn;
        {
            const double lo = std::fmin(mm[0].mn, mm[1].mn), hi = std::fmax(mm[0].mx, mm[1].mx);
            ctx->visc_contrast = (lo > 0.0 && hi > 0.0 && std::isfinite(hi / lo)) ? hi / lo : 1.0;
        }
        if (cfg->do_heatdiff) {
            mx = mm[2].has_nan ? std::numeric_limits<double>::quiet_NaN() : mm[2].mx;
            const double mindx = std::fmin(dz, dx);
            tstep_temp = cfg->tstep_modifier * mindx * mindx / mx;
            tstep_temp = (cfg->tstep_dif_max < tstep_temp) ? cfg->tstep_dif_max : tstep_temp;   // python min/max order
            tstep_temp = (cfg->tstep_dif_min > tstep_temp) ? cfg->tstep_dif_min : tstep_temp;
        }
    }

    // ---- 4. Stokes (pylamp2.py:349-366) -----------------------------------------------------------
    t0 = now_ms();
    double Kc, Kb;
    pl_stokes_scaling_host(ctx->geom, mes, men, &Kc, &Kb);
    const bool ss = cfg->surface_stabilization != 0;
    if (!ss || cfg->surfstab_tstep < 0) pl_stokes_fill_op(ctx, p_etas, p_etan, p_rho, cfg->bcstokes, 0, 0.0, 0.5, Kc, Kb);
    else pl_stokes_fill_op(ctx, p_etas, p_etan, p_rho, cfg->bcstokes, 1, cfg->surfstab_tstep, cfg->surfstab_theta, Kc, Kb);
    double* b = pl_stokes_rhs_buffer_device(ctx);
    if (!b) return 1;
    pl_launch_stokes_rhs(ctx, ctx->sop, b);
    pl_stokes_deflation(ctx, true);
    {   // Initial guess extrapolated in model time from the last solutions (Lagrange polynomial through up to PYLAMP_X0_ORDER + 1
        // of them, default 2 = quadratic; PYLAMP_X0_EXTRAP scales the target time, 0 switches the extrapolation off).  With the
        // plateau of the pressure-anchor mode gone (pl_solver.hip) every factor 3 in the initial residual is an iteration; at 2049^2:
        // none 15.3 iterations / 38.6 ms per solve, linear 13.3 / 33.5 ms, quadratic 11.5 / 29.9 ms.  In round 1 the same
        // extrapolation changed nothing -- the plateau ate whatever the start gained.  Only the iteration count depends on it: the
        // stopping rule is the same whatever the start.
        static const double wx = getenv("PYLAMP_X0_EXTRAP") ? atof(getenv("PYLAMP_X0_EXTRAP")) : 1.0;
        if (wx != 0.0 && S->have_solution) {
            double* xs = pl_stokes_solution_device(ctx);
            const long long n3 = 3 * g.plane;
            static const int order = std::max(1, std::min(3, getenv("PYLAMP_X0_ORDER") ? atoi(getenv("PYLAMP_X0_ORDER")) : 2));
            static const int keep = std::max(order, std::min((int)X0_HIST, getenv("PYLAMP_X0_POINTS") ? atoi(getenv("PYLAMP_X0_POINTS")) : order));
            if (!S->x_hist[0]) {
                const char* nm[X0_HIST] = {"x_prev", "x_prev2", "x_prev3", "x_prev4", "x_prev5"};
                for (int k = 0; k < keep; k++) PL_TRY(pl_buf(ctx, nm[k], (size_t)n3 * sizeof(double), &S->x_hist[k]));
            }
            // Weights of the polynomial of degree `order` fitted (least squares when more than order + 1 solutions are kept, else
            // interpolating) to the solutions at model times 0, -dt1, -dt1-dt2, ... and evaluated at +dt0.  A time step that more
            // than doubled (or a missing history) falls back to the linear formula with its weight capped at 2.
            const double* d = S->dt_hist;
            int np = std::min(keep, S->n_prev);                // older solutions used
            for (int k = 0; k <= np; k++) if (!(d[k] > 0.0)) np = std::min(np, std::max(0, k - 1));
            if (np >= 2 && d[0] > 2.0 * d[1]) np = 1;
            ExtrapArgs ea{};
            ea.nk = keep; ea.nh = np; ea.l0 = 1.0;
            for (int k = 0; k < keep; k++) ea.h[k] = S->x_hist[k];
            if (np == 1) {
                const double w = std::min(2.0, wx * d[0] / d[1]);
                ea.l0 = 1.0 + w; ea.l[0] = -w;
            } else if (np >= 2) {
                const int deg = std::min(order, np), m = np + 1;
                double tau[X0_HIST + 1] = {}, G[4][5] = {};              // times in units of dt0; normal equations (V^T V) c = t
                for (int k = 1; k <= np; k++) tau[k] = tau[k - 1] - d[k] / d[0];
                const double T = wx;
                for (int a2 = 0; a2 <= deg; a2++) {
                    for (int b2 = 0; b2 <= deg; b2++) for (int i = 0; i < m; i++) G[a2][b2] += std::pow(tau[i], a2 + b2);
                    G[a2][deg + 1] = std::pow(T, a2);
                }
                bool ok = true;
                for (int c = 0; c <= deg && ok; c++) {                    // Gauss-Jordan with partial pivoting on the (deg+1) system
                    int pv = c;
                    for (int r = c + 1; r <= deg; r++) if (std::fabs(G[r][c]) > std::fabs(G[pv][c])) pv = r;
                    if (!(std::fabs(G[pv][c]) > 1e-300)) { ok = false; break; }
                    for (int k = 0; k <= deg + 1; k++) std::swap(G[c][k], G[pv][k]);
                    for (int r = 0; r <= deg; r++) if (r != c) {
                        const double f = G[r][c] / G[c][c];
                        for (int k = c; k <= deg + 1; k++) G[r][k] -= f * G[c][k];
                    }
                }
                if (ok) {
                    double L[X0_HIST + 1];
                    for (int i = 0; i < m; i++) {
                        L[i] = 0.0;
                        for (int a2 = 0; a2 <= deg; a2++) L[i] += std::pow(tau[i], a2) * G[a2][deg + 1] / G[a2][a2];
                    }
                    ea.l0 = L[0];
                    for (int k = 0; k < np; k++) ea.l[k] = L[k + 1];
                } else ea.nh = 0;
            }
            hipLaunchKernelGGL(k_extrap_x0, grid1d(n3), dim3(256), 0, ctx->stream, n3, xs, ea);
            {   // the slot that took the old x becomes the newest
                double* newest = S->x_hist[keep - 1];
                for (int k = keep - 1; k > 0; k--) S->x_hist[k] = S->x_hist[k - 1];
                S->x_hist[0] = newest;
            }
            S->n_prev = std::min(keep, S->n_prev + 1);
        }
    }
    PL_TRY(pl_stokes_solve_device(ctx, b, S->have_solution, cfg->stokes_rtol > 0 ? cfg->stokes_rtol : 1e-10,
                                  cfg->stokes_maxit > 0 ? cfg->stokes_maxit : 400, &rep->stokes));
    S->have_solution = true;
    double* xsol = pl_stokes_solution_device(ctx);
    double* p_vz = xsol; double* p_vx = xsol + g.plane;
    double vmax_z, vmax_x;
    auto velocity_max = [&]() -> int {
        MinMaxSpec sp[2] = {{p_vz, nullptr, nullptr, 0}, {p_vx, nullptr, nullptr, 0}};
        MinMaxOut mm[2];
        PL_TRY(reduce_minmax_multi(ctx, S, g, 2, sp, mm));
        vmax_z = mm[0].mx; vmax_x = mm[1].mx;
        return 0;
    };
    PL_TRY(velocity_max());
    const double vmax = std::fmax(vmax_z, vmax_x);                  // signed np.max over both arrays (pylamp2.py:364)
    double tstep_stokes = cfg->tstep_modifier * std::fmin(dz, dx) / vmax;
    tstep_stokes = (cfg->tstep_adv_max < tstep_stokes) ? cfg->tstep_adv_max : tstep_stokes;
    tstep_stokes = (cfg->tstep_adv_min > tstep_stokes) ? cfg->tstep_adv_min : tstep_stokes;
    if (cfg->surfstab_tstep > 0) tstep_stokes = cfg->surfstab_tstep;          // pylamp2.py:368-372
    double tstep; int limiter;
    if (cfg->do_heatdiff) {
        limiter = (tstep_temp < tstep_stokes) ? 'H' : 'S';
        tstep = (tstep_stokes < tstep_temp) ? tstep_stokes : tstep_temp;
    } else { tstep = tstep_stokes; limiter = 'S'; }
    if (ss && cfg->surfstab_tstep < 0) {
        // re-assemble with the chosen step and re-solve until the step stops shrinking (pylamp2.py:387-405)
        for (int guard = 0; guard < 50; guard++) {
            pl_stokes_fill_op(ctx, p_etas, p_etan, p_rho, cfg->bcstokes, 1, tstep, cfg->surfstab_theta, Kc, Kb);
            pl_launch_stokes_rhs(ctx, ctx->sop, b);
            pl_solve_stats st2{};
            PL_TRY(pl_stokes_solve_device(ctx, b, true, cfg->stokes_rtol > 0 ? cfg->stokes_rtol : 1e-10,
                                          cfg->stokes_maxit > 0 ? cfg->stokes_maxit : 400, &st2));
            rep->stokes_resolves++;
            rep->stokes.iterations += st2.iterations; rep->stokes.converged &= st2.converged;
            rep->stokes.rel_residual = st2.rel_residual; rep->stokes.error_estimate = st2.error_estimate; rep->stokes.solve_ms += st2.solve_ms;
            rep->stokes.operator_applies += st2.operator_applies; rep->stokes.precond_applies += st2.precond_applies;
            rep->stokes.used_direct |= st2.used_direct;
            PL_TRY(velocity_max());
            const double check = cfg->tstep_modifier * std::fmin(dz, dx) / std::fmax(vmax_z, vmax_x);
            if (check < tstep) { tstep = check; limiter = 's'; }
            else break;
        }
    }
    rep->ms_stokes = now_ms() - t0;
    rep->tstep = tstep; rep->limiter = limiter; rep->tstep_heat = tstep_temp; rep->tstep_stokes = tstep_stokes;
    for (int k = X0_HIST; k > 0; k--) S->dt_hist[k] = S->dt_hist[k - 1];
    S->dt_hist[0] = tstep;
    // the reference would carry a NaN time step on (NaN positions from the next advection); say what happened instead
    if (!std::isfinite(tstep))
        return pl_fail(ctx, "pl_step: the time step is not finite - a grid node without any marker in reach makes the interpolated "
                            "fields NaN, and so do tracers injected into a cell without residents (their fields are 0/0 as in "
                            "pylamp2.py:624-629): raise the marker density or enable injection before cells run empty");

    // ---- 5. heat (pylamp2.py:412-480) ----------------------------------------------------------------
    if (cfg->do_heatdiff) {
        t0 = now_ms();
        PL_TRY(pl_heat_tables(ctx, S->gmz.data(), S->gmx.data()));
        PlHeatOp& hop = ctx->hop;
        hop.g = g; hop.kz = p_kz; hop.kx = p_kx; hop.rhocp_inv_dt = p_c; hop.dt = tstep;
        for (int w = 0; w < 4; w++) { hop.bc[w] = cfg->bcheat[w]; ctx->heat_bcvalue[w] = cfg->bcheatvals[w]; }
        ctx->hop_ready = true;
        pl_launch_heat_coef(ctx, g, p_rho, p_cp, tstep, p_c);
        double* hb;
        PL_TRY(pl_buf(ctx, "api_hx", pb, &hb));
        pl_launch_heat_rhs(ctx, hop, p_T, p_H, hb);
        double* xs = nullptr;
        // Initial guess: the nodal temperature the right-hand side was built from, plus (from the third step on) the last step's
        // implicit increment scaled to this step's length -- instead of zero (PYLAMP_HEAT_X0 = 0: zero, 1: the old temperature only).
        static const int hx0 = getenv("PYLAMP_HEAT_X0") ? atoi(getenv("PYLAMP_HEAT_X0")) : 2;
        const double* guess = nullptr;
        if (hx0 >= 1) {
            guess = p_T;
            if (hx0 >= 2 && S->have_dT && S->dt_hist[1] > 0.0 && tstep <= 2.0 * S->dt_hist[1]) {
                double* hg;
                PL_TRY(pl_buf(ctx, "heat_x0", pb, &hg));
                const long long np1 = g.plane;
                hipLaunchKernelGGL(k_plane_axpy, grid1d(np1), dim3(256), 0, ctx->stream, np1, hg, p_T, p_dT, tstep / S->dt_hist[1]);
                guess = hg;
            }
        }
        PL_TRY(pl_heat_solve_device(ctx, hb, cfg->heat_rtol > 0 ? cfg->heat_rtol : 1e-12,
                                    cfg->heat_maxit > 0 ? cfg->heat_maxit : 2000, &rep->heat, &xs, guess));
        PL_HIP(ctx, hipMemcpyAsync(p_newT, xs, pb, hipMemcpyDeviceToDevice, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PL_TRY(pl_halo(ctx, g, p_newT, 1, g.plane));
        rep->ms_heat = now_ms() - t0;

        // temperature to tracers
        t0 = now_ms();
        PL_TRY(stage_temp_to_tracers(ctx, S, cfg, it == 1 || !S->have_newtemp, P, tstep));
        S->have_newtemp = true;
        rep->ms_gather = now_ms() - t0;
    }

    // ---- 6. advection (pylamp2.py:484-572) ------------------------------------------------------------
    t0 = now_ms();
    // window of the padded centre grid the RK4 stages of this rank's tracers can reach: a tracer moves less than a cell
    // per stage (the time step is CFL-limited, pylamp2.py:364), so cells [gi0-1, gi0+lnz+1] of the padded grid, i.e.
    // its nodes I in [gi0-1, gi0+lnz+2], computed from the velocity planes with a halo 3 nodes deep -- no all-gather.
    const int I0 = std::max(g.gi0 - 1, 0), I1 = std::min(g.gi0 + g.lnz + 2, nz), J0 = std::max(g.gj0 - 1, 0), J1 = std::min(g.gj0 + g.lnx + 2, nx);
    const int nVr = I1 - I0 + 1, nVc = J1 - J0 + 1;
    const size_t VN = (size_t)nVr * nVc;
    double* V;
    PL_TRY(pl_buf(ctx, "advect_vel", 2 * VN * sizeof(double), &V, false));
    PL_TRY(pl_halo(ctx, g, xsol, 2, g.plane, 3));
    {
        dim3 gr((nVc + 63) / 64, (nVr + 3) / 4);
        hipLaunchKernelGGL(k_advection_velocity, gr, dim3(64, 4), 0, ctx->stream, g, (const double*)p_vz, (const double*)p_vx,
                           (cfg->bcstokes[0] & PL_BC_FREESLIP) ? 1 : 0, (cfg->bcstokes[1] & PL_BC_FREESLIP) ? 1 : 0,
                           (cfg->bcstokes[2] & PL_BC_FREESLIP) ? 1 : 0, (cfg->bcstokes[3] & PL_BC_FREESLIP) ? 1 : 0, I0, J0, nVr, nVc, V, V + VN);
    }
    bool keys_ready = false;
    PL_TRY(stage_rk4(ctx, S, V, I0, I1, J0, J1, tstep, cfg->tracs_fence_disabled ? 0 : 1, Lz, Lx, &keys_ready));
    S->cols_undefined = false;        // RHO / ETA (stage 1) and the tracer velocities (here) have been rewritten
    rep->ms_advect = now_ms() - t0;

    // ---- 7. cell sort of the advected tracers, slab migration, census + injection --------------------
    t0 = now_ms();
    S->sorted = false;
    {
        long long removed = 0;
        SortOpts so; so.del_outside = cfg->tracs_fence_disabled ? 1 : 0; so.Lz = Lz; so.Lx = Lx; so.removed = &removed;
        so.keys_ready = keys_ready;
        SortOpts si; si.inject = cfg; si.it = it; si.ninjected = &rep->ninjected;
        static const bool presort_env = !(getenv("PYLAMP_MIGRATE_PRESORT") && atoi(getenv("PYLAMP_MIGRATE_PRESORT")) == 0);
        const bool presort = ctx->nranks > 1 && presort_env && keys_ready && !so.del_outside && ctx->geom.uniform && scatter_cells_on() && epoch_length() > 0;
        if (ctx->nranks == 1 || presort) { so.inject = si.inject; so.it = it; so.ninjected = si.ninjected; }     // one pass does it all
        if (presort) {                // several ranks: leavers out / arrivals in BEFORE the one sort (epoch layout, as on one rank)
            PL_TRY(migrate_presort(ctx, S, z0, hz, x0, hx));
            so.premigrated = true; so.drop_trash = true;
        }
        PL_TRY(sort_tracers(ctx, S, z0, hz, x0, hx, so));
        rep->nremoved = removed;
        if (!presort) PL_TRY(migrate_tracers(ctx, S, z0, hz, x0, hx, si));          // classic: sort, send the leaver buckets, append, sort again
    }
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    S->sorted = true;
    rep->ms_sort = now_ms() - t0;
    rep->ntrac = S->n;
    rep->ms_total = now_ms() - t_all;
    return 0;
}

// ---- the marker stages one at a time (tests pin the kernels the timed step launches against the reference's fixtures) -----
static int resident_ready(pl_ctx* ctx, PlStepState* S, const char* who) {
    if (ctx->nranks != 1) return pl_fail(ctx, std::string(who) + ": single-rank entry point (several ranks run pl_step)");
    if (!S->cell_start || !S->sorted) return pl_fail(ctx, std::string(who) + ": no cell-sorted tracers resident (call pl_tracers_upload)");
    return 0;
}

extern "C" int pl_resident_scatter(pl_ctx* ctx, const pl_step_config* cfg, int it) {
    if (!cfg) return pl_fail(ctx, "pl_resident_scatter: NULL argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PlStepState* S = state_of(ctx);
    PL_TRY(resident_ready(ctx, S, "pl_resident_scatter"));
    PL_TRY(flush_lazy(ctx, S));
    PL_TRY(ensure_coords(ctx, S));
    StepPlanes P;
    PL_TRY(step_planes(ctx, P));
    PL_TRY(stage_props(ctx, S, cfg));
    PL_TRY(stage_scatter(ctx, S, cfg, it, P));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int pl_resident_temp_to_tracers(pl_ctx* ctx, const pl_step_config* cfg, int first, const double* newtemp, double tstep) {
    if (!cfg || !newtemp) return pl_fail(ctx, "pl_resident_temp_to_tracers: NULL argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PlStepState* S = state_of(ctx);
    PL_TRY(resident_ready(ctx, S, "pl_resident_temp_to_tracers"));
    PL_TRY(flush_lazy(ctx, S));
    PL_TRY(ensure_coords(ctx, S));
    StepPlanes P;
    PL_TRY(step_planes(ctx, P));
    PL_TRY(pl_plane_upload(ctx, ctx->geom.d, newtemp, P.newT));
    PL_TRY(stage_temp_to_tracers(ctx, S, cfg, first != 0, P, tstep));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int pl_resident_rk4(pl_ctx* ctx, const double* vz_pad, const double* vx_pad, double tstep, int fence, const double length[2]) {
    if (!vz_pad || !vx_pad || !length) return pl_fail(ctx, "pl_resident_rk4: NULL argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PlStepState* S = state_of(ctx);
    PL_TRY(resident_ready(ctx, S, "pl_resident_rk4"));
    PL_TRY(flush_lazy(ctx, S));
    PL_TRY(ensure_coords(ctx, S));
    const int nz = ctx->nz, nx = ctx->nx;
    const size_t VN = (size_t)(nz + 1) * (nx + 1);
    double* V;
    PL_TRY(pl_buf(ctx, "advect_vel", 2 * VN * sizeof(double), &V, false));
    PL_HIP(ctx, hipMemcpyAsync(V, vz_pad, VN * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PL_HIP(ctx, hipMemcpyAsync(V + VN, vx_pad, VN * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PL_TRY(stage_rk4(ctx, S, V, 0, nz, 0, nx, tstep, fence ? 1 : 0, length[0], length[1]));
    S->sorted = false;
    const StepGrid q = step_grid(ctx);
    PL_TRY(sort_tracers(ctx, S, q.z0, q.hz, q.x0, q.hx));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    S->sorted = true;
    return 0;
}

extern "C" int pl_get_field(pl_ctx* ctx, const char* name, double* out) {
    if (!name || !out) return pl_fail(ctx, "pl_get_field: NULL argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const PlGeom& g = ctx->geom.d;
    std::string nm(name);
    const double* src = nullptr;
    if (nm == "velz" || nm == "velx" || nm == "pres") {
        double* x = pl_stokes_solution_device(ctx);
        if (!x) return pl_fail(ctx, "pl_get_field: no Stokes solution yet");
        src = x + (nm == "velz" ? 0 : nm == "velx" ? 1 : 2) * g.plane;
    } else {
        auto it = ctx->bufs.find(nm);
        if (it == ctx->bufs.end()) return pl_fail(ctx, "pl_get_field: unknown field '" + nm + "'");
        src = it->second;
    }
    return pl_plane_download(ctx, g, src, out);
}
