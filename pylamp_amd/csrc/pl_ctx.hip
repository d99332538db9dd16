// Context, memory, geometry tables and host<->device layout conversion.
#include "pl_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstddef>
#include <algorithm>

thread_local std::string pl_tls_error;

int pl_fail(pl_ctx* ctx, const std::string& msg) {
    if (ctx) ctx->err = msg;
    pl_tls_error = msg;
    return 1;
}

extern "C" const char* pl_last_error(const pl_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    return pl_tls_error.c_str();
}

int pl_buf(pl_ctx* ctx, const char* name, size_t bytes, double** out, bool zero) {
    auto it = ctx->bufs.find(name);
    if (it != ctx->bufs.end() && ctx->buf_bytes[name] >= bytes) {
        *out = it->second;
        return 0;
    }
    if (it != ctx->bufs.end()) {
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PL_HIP(ctx, hipFree(it->second));
        ctx->bufs.erase(it);
    }
    double* p = nullptr;
    PL_HIP(ctx, hipMalloc((void**)&p, bytes));
    if (zero) PL_HIP(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream));
    ctx->bufs[name] = p;
    ctx->buf_bytes[name] = bytes;
    *out = p;
    return 0;
}

int pl_stage(pl_ctx* ctx, size_t bytes) {
    if (ctx->stage_bytes >= bytes) return 0;
    if (ctx->stage) {
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PL_HIP(ctx, hipFree(ctx->stage));
        ctx->stage = nullptr; ctx->stage_bytes = 0;
    }
    PL_HIP(ctx, hipMalloc((void**)&ctx->stage, bytes));
    ctx->stage_bytes = bytes;
    return 0;
}

// ---- geometry -----------------------------------------------------------------------
int pl_geom_build(pl_ctx* ctx, PlGeomHost& gh, int nz, int nx, const double* zc, const double* xc) {
    gh.zc.assign(zc, zc + nz);
    gh.xc.assign(xc, xc + nx);
    PlGeom& g = gh.d;
    g.nz = nz; g.nx = nx; g.lnz = nz; g.lnx = nx; g.gi0 = 0; g.gj0 = 0;
    g.pitch = ((PL_PADL + nx + PL_RING + 15) / 16) * 16;
    g.plane = (long long)(nz + 2 * PL_RING) * g.pitch;
    // tables indexed by global index + 1, length n+3 each
    const int T0 = PL_TOFF;
    size_t lz = ((size_t)nz + 2 * T0 + 2 + 1) & ~(size_t)1, lx = ((size_t)nx + 2 * T0 + 2 + 1) & ~(size_t)1;   // even lengths
    std::vector<double> t(3 * lz + 3 * lx, 0.0);
    double* tz = t.data(); double* trdz = tz + lz; double* trDz = trdz + lz;
    double* tx = trDz + lz; double* trdx = tx + lx; double* trDx = trdx + lx;
    for (int i = 0; i < nz; i++) tz[i + T0] = zc[i];
    for (int j = 0; j < nx; j++) tx[j + T0] = xc[j];
    for (int i = 0; i + 1 < nz; i++) trdz[i + T0] = 1.0 / (zc[i + 1] - zc[i]);
    for (int i = 1; i + 1 < nz; i++) trDz[i + T0] = 1.0 / (zc[i + 1] - zc[i - 1]);
    for (int j = 0; j + 1 < nx; j++) trdx[j + T0] = 1.0 / (xc[j + 1] - xc[j]);
    for (int j = 1; j + 1 < nx; j++) trDx[j + T0] = 1.0 / (xc[j + 1] - xc[j - 1]);
    if (gh.tables) { (void)hipFree(gh.tables); gh.tables = nullptr; }
    PL_HIP(ctx, hipMalloc((void**)&gh.tables, t.size() * sizeof(double)));
    PL_HIP(ctx, hipMemcpy(gh.tables, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    g.zc = gh.tables; g.rdz = gh.tables + lz; g.rDz = gh.tables + 2 * lz;
    g.xc = gh.tables + 3 * lz; g.rdx = g.xc + lx; g.rDx = g.xc + 2 * lx;
    // uniformity (informational; kernels always use the tables)
    gh.uniform = true;
    double hz = (zc[nz - 1] - zc[0]) / (nz - 1), hx = (xc[nx - 1] - xc[0]) / (nx - 1);
    for (int i = 0; i + 1 < nz; i++) if (std::fabs((zc[i + 1] - zc[i]) - hz) > 1e-9 * hz) gh.uniform = false;
    for (int j = 0; j + 1 < nx; j++) if (std::fabs((xc[j + 1] - xc[j]) - hx) > 1e-9 * hx) gh.uniform = false;
    return 0;
}

void pl_geom_set_block(PlGeomHost& gh, int gi0, int lnz, int gj0, int lnx) {
    PlGeom& g = gh.d;
    g.gi0 = gi0; g.lnz = lnz; g.gj0 = gj0; g.lnx = lnx;
    g.pitch = ((PL_PADL + lnx + PL_RING + 15) / 16) * 16;
    g.plane = (long long)(lnz + 2 * PL_RING) * g.pitch;
}

// block of this rank; the checks every transport shares
static int set_layout(pl_ctx* ctx, int rank, int Pz, int Px) {
    const int nranks = Pz * Px;
    if (Pz < 1 || Px < 1 || rank < 0 || rank >= nranks) return pl_fail(ctx, "pl_set_comm: bad rank / layout");
    if (!ctx->bufs.empty() || ctx->krylov || ctx->step) return pl_fail(ctx, "pl_set_comm must be called right after pl_create");
    ctx->rank = rank; ctx->nranks = nranks; ctx->Pz = Pz; ctx->Px = Px; ctx->pz = rank / Px; ctx->px = rank % Px;
    if (nranks == 1) return 0;
    if ((ctx->nz - 1) % Pz || (ctx->nx - 1) % Px)
        return pl_fail(ctx, "pl_set_comm: (nz-1) and (nx-1) must be divisible by the number of blocks along z resp. x");
    const int Cz = (ctx->nz - 1) / Pz, Cx = (ctx->nx - 1) / Px;
    if ((Pz > 1 && (Cz < 8 || (Cz % 2))) || (Px > 1 && (Cx < 8 || (Cx % 2))))
        return pl_fail(ctx, "pl_set_comm: need an even number (>= 8) of node rows / columns per block");
    int i0, ni, j0, nj;
    pl_block_1d(ctx->nz, Pz, ctx->pz, &i0, &ni); pl_block_1d(ctx->nx, Px, ctx->px, &j0, &nj);
    pl_geom_set_block(ctx->geom, i0, ni, j0, nj);
    return 0;
}

static void choose_layout(int nranks, int* Pz, int* Px) {
    *Pz = 1; *Px = nranks;
    if (const char* e = getenv("PYLAMP_DECOMP")) {
        int a = 0, b = 0;
        if (sscanf(e, "%dx%d", &a, &b) == 2 && a >= 1 && b >= 1 && a * b == nranks) { *Pz = a; *Px = b; return; }
    }
    for (int a = 1; a * a <= nranks; a++) if (nranks % a == 0) { *Pz = a; *Px = nranks / a; }     // most square, Px >= Pz
}

extern "C" int pl_set_comm_2d(pl_ctx* ctx, int rank, int Pz, int Px, const pl_comm_ops* ops) {
    if (!ops) return pl_fail(ctx, "pl_set_comm: bad argument");
    PL_TRY(set_layout(ctx, rank, Pz, Px));
    if (ctx->nranks == 1) {
        if (getenv("PYLAMP_RCCL_SELFTEST")) return pl_comm_native_init(ctx);   // single-rank API check (tests)
        return 0;
    }
    if (!ops->sendrecv || !ops->allreduce_host || !ops->allgather) return pl_fail(ctx, "pl_set_comm: incomplete callback table");
    ctx->comm = *ops;
    return pl_comm_native_init(ctx);      // RCCL directly on the context stream when asked for and every rank can
}

extern "C" int pl_set_comm(pl_ctx* ctx, int rank, int nranks, const pl_comm_ops* ops) {
    if (nranks < 1) return pl_fail(ctx, "pl_set_comm: bad argument");
    int Pz, Px;
    choose_layout(nranks, &Pz, &Px);
    return pl_set_comm_2d(ctx, rank, Pz, Px, ops);
}

int pl_local_attach(pl_ctx* ctx, pl_local_group* g, int rank);       // pl_comm.hip
extern "C" int pl_set_comm_local(pl_ctx* ctx, pl_local_group* g, int rank, int Pz, int Px) {
    if (!g) return pl_fail(ctx, "pl_set_comm_local: no group");
    PL_TRY(set_layout(ctx, rank, Pz, Px));
    return pl_local_attach(ctx, g, rank);
}

// cumulative counts of the communication calls issued by this context (reset = 1 clears them afterwards)
extern "C" int pl_comm_stats(pl_ctx* ctx, int64_t out[4], int reset) {
    for (int k = 0; k < 4; k++) { if (out) out[k] = ctx->comm_calls[k]; if (reset) ctx->comm_calls[k] = 0; }
    return 0;
}

extern "C" int pl_local_rows(pl_ctx* ctx, int* first_row, int* n_rows) {
    if (first_row) *first_row = ctx->geom.d.gi0;
    if (n_rows) *n_rows = ctx->geom.d.lnz;
    return 0;
}
extern "C" int pl_local_block(pl_ctx* ctx, int* first_row, int* n_rows, int* first_col, int* n_cols, int* Pz, int* Px) {
    const PlGeom& g = ctx->geom.d;
    if (first_row) *first_row = g.gi0;
    if (n_rows) *n_rows = g.lnz;
    if (first_col) *first_col = g.gj0;
    if (n_cols) *n_cols = g.lnx;
    if (Pz) *Pz = ctx->Pz;
    if (Px) *Px = ctx->Px;
    return 0;
}

extern "C" int pl_memcpy_d2h(pl_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    PL_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" int pl_memcpy_h2d(pl_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    PL_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
__global__ void k_dev_add(long long n, double* __restrict__ d, const double* __restrict__ s) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) d[k] += s[k];
}
extern "C" int pl_dev_add(pl_ctx* ctx, double* dst_dev, const double* src_dev, int64_t n) {
    if (n <= 0) return 0;
    long long nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_dev_add, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (long long)n, dst_dev, src_dev);
    PL_HIP(ctx, hipGetLastError());
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

void pl_geom_free(PlGeomHost& gh) {
    if (gh.tables) (void)hipFree(gh.tables);
    gh.tables = nullptr;
}

// ---- context ---------------------------------------------------------------------------
extern "C" int pl_create(pl_ctx** out, int device, int nz, int nx, const double* zc, const double* xc) {
    if (!out) return pl_fail(nullptr, "pl_create: out is NULL");
    *out = nullptr;
    if (nz < 5 || nx < 5) return pl_fail(nullptr, "pl_create: grid must be at least 5x5 nodes");
    if (!zc || !xc) return pl_fail(nullptr, "pl_create: coordinate arrays are NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return pl_fail(nullptr, "pl_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return pl_fail(nullptr, "pl_create: bad device index");
    pl_ctx* ctx = new pl_ctx();
    ctx->device = device; ctx->nz = nz; ctx->nx = nx;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return pl_fail(nullptr, "pl_create: hipSetDevice failed"); }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx; return pl_fail(nullptr, "pl_create: stream/event creation failed");
    }
    if (pl_geom_build(ctx, ctx->geom, nz, nx, zc, xc)) {
        std::string m = ctx->err; pl_destroy(ctx); return pl_fail(nullptr, m);
    }
    *out = ctx;
    return 0;
}

extern "C" void pl_destroy(pl_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    pl_comm_native_free(ctx);
    pl_local_detach(ctx);
    pl_step_free(ctx);
    pl_mic_free(ctx);
    pl_solver_free(ctx);
    pl_direct_free(ctx);
    for (hipEvent_t e : ctx->comm_ev) if (e) (void)hipEventDestroy(e);
    for (auto& kv : ctx->bufs) (void)hipFree(kv.second);
    if (ctx->stage) (void)hipFree(ctx->stage);
    pl_geom_free(ctx->geom);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// sizes and a few field offsets of the ABI structs as THIS build sees them (the ctypes mirror is checked against it)
extern "C" int pl_abi_layout(size_t out[8]) {
    out[0] = sizeof(pl_solve_stats); out[1] = sizeof(pl_step_config); out[2] = sizeof(pl_step_report);
    out[3] = offsetof(pl_step_config, length); out[4] = offsetof(pl_step_config, inject_seed);
    out[5] = offsetof(pl_step_config, tracs_fence_disabled); out[6] = offsetof(pl_step_report, ntrac);
    out[7] = offsetof(pl_step_report, nremoved);
    return 0;
}

extern "C" int pl_sync(pl_ctx* ctx) {
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int pl_device_info(pl_ctx* ctx, char* name, size_t name_len, int* cu_count, size_t* hbm_bytes) {
    hipDeviceProp_t p;
    PL_HIP(ctx, hipGetDeviceProperties(&p, ctx->device));
    if (name && name_len) { std::snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName); }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return 0;
}

extern "C" int pl_timer_start(pl_ctx* ctx) {
    PL_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return 0;
}

extern "C" int pl_timer_stop_ms(pl_ctx* ctx, double* ms) {
    PL_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    PL_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float f = 0;
    PL_HIP(ctx, hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    if (ms) *ms = f;
    return 0;
}

// ---- plane / vector transfers ---------------------------------------------------------------
// host is the GLOBAL (nz,nx) array on every rank; the local block plus the neighbouring halo (PL_RING deep, clipped
// at the domain boundary) is taken from it, so coefficient planes need no halo exchange.
int pl_plane_upload(pl_ctx* ctx, const PlGeom& g, const double* host, double* dplane) {
    const int r0 = -std::min(PL_RING, g.gi0), r1 = g.lnz + std::min(PL_RING, g.nz - (g.gi0 + g.lnz));
    const int c0 = -std::min(PL_RING, g.gj0), c1 = g.lnx + std::min(PL_RING, g.nx - (g.gj0 + g.lnx));
    PL_HIP(ctx, hipMemcpy2DAsync(dplane + pl_idx(g, r0, c0), (size_t)g.pitch * sizeof(double),
                                 host + (size_t)(g.gi0 + r0) * g.nx + (g.gj0 + c0), (size_t)g.nx * sizeof(double),
                                 (size_t)(c1 - c0) * sizeof(double), r1 - r0, hipMemcpyHostToDevice, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host buffer may be reused by the caller
    return 0;
}

// every rank receives the complete GLOBAL (nz,nx) array (own block + zero elsewhere, summed over ranks)
int pl_plane_download(pl_ctx* ctx, const PlGeom& g, const double* dplane, double* host) {
    if (ctx->nranks > 1) memset(host, 0, (size_t)g.nz * g.nx * sizeof(double));
    PL_HIP(ctx, hipMemcpy2DAsync(host + (size_t)g.gi0 * g.nx + g.gj0, (size_t)g.nx * sizeof(double), dplane + pl_idx(g, 0, 0),
                                 (size_t)g.pitch * sizeof(double), (size_t)g.lnx * sizeof(double), g.lnz,
                                 hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return pl_allreduce_host(ctx, host, (long long)g.nz * g.nx, 0);
}

// interleaved (node-major, 3 per node) <-> 3 planes
__global__ __launch_bounds__(256) void k_deinterleave3(PlGeom g, const double* __restrict__ src,
                                                       double* __restrict__ dst) {
    int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y;
    if (lj >= g.lnx || li >= g.lnz) return;
    long long n = ((long long)li * g.lnx + lj) * 3, c = pl_idx(g, li, lj);
    dst[c] = src[n]; dst[c + g.plane] = src[n + 1]; dst[c + 2 * g.plane] = src[n + 2];
}

__global__ __launch_bounds__(256) void k_interleave3(PlGeom g, const double* __restrict__ src,
                                                     double* __restrict__ dst) {
    int lj = blockIdx.x * 64 + threadIdx.x, li = blockIdx.y * 4 + threadIdx.y;
    if (lj >= g.lnx || li >= g.lnz) return;
    long long n = ((long long)li * g.lnx + lj) * 3, c = pl_idx(g, li, lj);
    dst[n] = src[c]; dst[n + 1] = src[c + g.plane]; dst[n + 2] = src[c + 2 * g.plane];
}

static dim3 grid2d(const PlGeom& g) { return dim3((g.lnx + 63) / 64, (g.lnz + 3) / 4); }

int pl_vec3_upload(pl_ctx* ctx, const PlGeom& g, const double* host, double* dvec) {
    size_t bytes = (size_t)3 * g.lnz * g.lnx * sizeof(double);
    PL_TRY(pl_stage(ctx, bytes));
    PL_HIP(ctx, hipMemcpy2DAsync(ctx->stage, (size_t)3 * g.lnx * sizeof(double), host + ((size_t)g.gi0 * g.nx + g.gj0) * 3,
                                 (size_t)3 * g.nx * sizeof(double), (size_t)3 * g.lnx * sizeof(double), g.lnz,
                                 hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_deinterleave3, grid2d(g), dim3(64, 4), 0, ctx->stream, g, ctx->stage, dvec);
    PL_HIP(ctx, hipGetLastError());
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int pl_vec3_download(pl_ctx* ctx, const PlGeom& g, const double* dvec, double* host) {
    size_t bytes = (size_t)3 * g.lnz * g.lnx * sizeof(double);
    PL_TRY(pl_stage(ctx, bytes));
    hipLaunchKernelGGL(k_interleave3, grid2d(g), dim3(64, 4), 0, ctx->stream, g, dvec, ctx->stage);
    PL_HIP(ctx, hipGetLastError());
    if (ctx->nranks > 1) memset(host, 0, (size_t)3 * g.nz * g.nx * sizeof(double));
    PL_HIP(ctx, hipMemcpy2DAsync(host + ((size_t)g.gi0 * g.nx + g.gj0) * 3, (size_t)3 * g.nx * sizeof(double), ctx->stage,
                                 (size_t)3 * g.lnx * sizeof(double), (size_t)3 * g.lnx * sizeof(double), g.lnz,
                                 hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return pl_allreduce_host(ctx, host, (long long)3 * g.nz * g.nx, 0);
}
