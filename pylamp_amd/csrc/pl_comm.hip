// Communication dispatch for the row-slab decomposition.
//
// Fast path: RCCL called directly on the context's HIP stream (ncclSend/ncclRecv groups for the
// halo rows over the direct xGMI links, ncclAllGather for the replicated coarse level and the
// advection velocity).  Everything is stream-ordered: no host synchronisation per exchange.
// librccl is dlopen'ed at pl_set_comm time (soname librccl.so.1 - the copy torch already loaded
// when torch is in the process); the unique id travels through the host callback table.
// An init-time self-test (ring exchange + all-gather against known values) decides collectively
// whether the native path is used; otherwise every call falls back to the callback table
// (pylamp_amd/parallel.py: torch.distributed), which is the path the multi-rank tests exercise
// on a single GPU.
#include "pl_internal.h"
#include <dlfcn.h>
#include <chrono>
#include <cstdlib>
#include <thread>

typedef struct { char internal[128]; } pl_ncclUniqueId;
typedef void* pl_ncclComm_t;
enum { PL_NCCL_DOUBLE = 8, PL_NCCL_SUM = 0 };

struct PlNccl {
    void* lib = nullptr;
    pl_ncclComm_t comm = nullptr;
    bool ok = false;
    int (*GetUniqueId)(pl_ncclUniqueId*) = nullptr;
    int (*CommInitRank)(pl_ncclComm_t*, int, pl_ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(pl_ncclComm_t) = nullptr;
    int (*CommAbort)(pl_ncclComm_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, pl_ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, pl_ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, pl_ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, pl_ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    std::vector<double> selftest_host;      // lives as long as the context: a late copy of an abandoned self-test lands here
};

static PlNccl* nccl_of(pl_ctx* ctx) { return (PlNccl*)ctx->nccl; }

__global__ void k_comm_add(long long n, double* __restrict__ d, const double* __restrict__ s) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) d[k] += s[k];
}
__global__ void k_comm_fill(long long n, double* __restrict__ d, double v) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) d[k] = v;
}

void pl_comm_native_free(pl_ctx* ctx) {
    PlNccl* N = nccl_of(ctx);
    if (!N) return;
    if (N->comm && N->CommDestroy) (void)N->CommDestroy(N->comm);
    // the library handle is left open on purpose (torch may share it)
    delete N;
    ctx->nccl = nullptr;
}

// ---- native primitives ---------------------------------------------------------------------------
static int native_exchange(pl_ctx* ctx, const double* send_lo, double* recv_lo, const double* send_hi, double* recv_hi,
                           long long count, int nseg, long long stride, int add) {
    PlNccl* N = nccl_of(ctx);
    const int lo = ctx->rank - 1, hi = ctx->rank + 1;
    const bool has_lo = lo >= 0, has_hi = hi < ctx->nranks;
    double* tmp = nullptr;
    if (add) PL_TRY(pl_buf(ctx, "nccl_tmp", (size_t)2 * nseg * count * sizeof(double), &tmp, false));
    int rc = N->GroupStart();
    for (int k = 0; k < nseg && !rc; k++) {
        const long long o = (long long)k * stride;
        if (has_lo) {
            rc |= N->Send(send_lo + o, (size_t)count, PL_NCCL_DOUBLE, lo, N->comm, ctx->stream);
            rc |= N->Recv(add ? tmp + (long long)k * count : recv_lo + o, (size_t)count, PL_NCCL_DOUBLE, lo, N->comm, ctx->stream);
        }
        if (has_hi) {
            rc |= N->Send(send_hi + o, (size_t)count, PL_NCCL_DOUBLE, hi, N->comm, ctx->stream);
            rc |= N->Recv(add ? tmp + (long long)(nseg + k) * count : recv_hi + o, (size_t)count, PL_NCCL_DOUBLE, hi, N->comm, ctx->stream);
        }
    }
    rc |= N->GroupEnd();
    if (rc) return pl_fail(ctx, "RCCL neighbour exchange failed");
    if (add) {
        const unsigned nb = (unsigned)((count + 255) / 256 > 1024 ? 1024 : (count + 255) / 256);
        for (int k = 0; k < nseg; k++) {
            if (has_lo) hipLaunchKernelGGL(k_comm_add, dim3(nb), dim3(256), 0, ctx->stream, count, recv_lo + (long long)k * stride, tmp + (long long)k * count);
            if (has_hi) hipLaunchKernelGGL(k_comm_add, dim3(nb), dim3(256), 0, ctx->stream, count, recv_hi + (long long)k * stride, tmp + (long long)(nseg + k) * count);
        }
    }
    return 0;
}

static int native_allgather(pl_ctx* ctx, double* recv, long long count, int nseg, long long stride) {
    PlNccl* N = nccl_of(ctx);
    int rc = 0;
    for (int k = 0; k < nseg && !rc; k++) {
        double* base = recv + (long long)k * stride;
        rc = N->AllGather(base + (long long)ctx->rank * count, base, (size_t)count, PL_NCCL_DOUBLE, N->comm, ctx->stream);
    }
    if (rc) return pl_fail(ctx, "RCCL all-gather failed");
    return 0;
}

// ---- public dispatch -------------------------------------------------------------------------------
int pl_comm_exchange(pl_ctx* ctx, const double* send_lo, double* recv_lo, const double* send_hi, double* recv_hi,
                     long long count, int nseg, long long stride, int add) {
    if (ctx->nranks <= 1) return 0;
    ctx->comm_calls[0]++;
    PlNccl* N = nccl_of(ctx);
    if (N && N->ok) return native_exchange(ctx, send_lo, recv_lo, send_hi, recv_hi, count, nseg, stride, add);
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm.exchange(ctx->comm.user, send_lo, recv_lo, send_hi, recv_hi, count, nseg, stride, add))
        return pl_fail(ctx, "communication callback 'exchange' failed");
    return 0;
}

int pl_comm_allgather(pl_ctx* ctx, double* recv, long long count, int nseg, long long stride) {
    if (ctx->nranks <= 1) return 0;
    ctx->comm_calls[1]++;
    PlNccl* N = nccl_of(ctx);
    if (N && N->ok) return native_allgather(ctx, recv, count, nseg, stride);
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm.allgather(ctx->comm.user, recv, count, nseg, stride))
        return pl_fail(ctx, "communication callback 'allgather' failed");
    return 0;
}

// variable-size neighbour exchange of tracer columns; counts always travel through the host table
int pl_comm_exchange_var(pl_ctx* ctx, double* const* send_lo, long long n_lo, double* const* send_hi, long long n_hi,
                         double* const* recv, long long cap, int ncol, long long* got) {
    *got = 0;
    if (ctx->nranks <= 1) return 0;
    PlNccl* N = nccl_of(ctx);
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!(N && N->ok)) {
        int64_t g = 0;
        if (ctx->comm.exchange_var(ctx->comm.user, send_lo, n_lo, send_hi, n_hi, recv, cap, ncol, &g))
            return pl_fail(ctx, "communication callback 'exchange_var' failed (tracer migration)");
        *got = g;
        return 0;
    }
    // what do my neighbours send me?  slot r*2 = count rank r sends down (to r-1), r*2+1 = up (to r+1)
    std::vector<double> c((size_t)2 * ctx->nranks, 0.0);
    c[2 * ctx->rank] = (double)n_lo; c[2 * ctx->rank + 1] = (double)n_hi;
    PL_TRY(pl_allreduce_host(ctx, c.data(), (long long)c.size(), 0));
    const int lo = ctx->rank - 1, hi = ctx->rank + 1;
    const long long m_lo = lo >= 0 ? (long long)c[2 * lo + 1] : 0, m_hi = hi < ctx->nranks ? (long long)c[2 * hi] : 0;
    if (m_lo + m_hi > cap) return pl_fail(ctx, "tracer migration exceeds the receive capacity");
    int rc = N->GroupStart();
    for (int k = 0; k < ncol && !rc; k++) {
        if (lo >= 0 && n_lo) rc |= N->Send(send_lo[k], (size_t)n_lo, PL_NCCL_DOUBLE, lo, N->comm, ctx->stream);
        if (hi < ctx->nranks && n_hi) rc |= N->Send(send_hi[k], (size_t)n_hi, PL_NCCL_DOUBLE, hi, N->comm, ctx->stream);
        if (m_lo) rc |= N->Recv(recv[k], (size_t)m_lo, PL_NCCL_DOUBLE, lo, N->comm, ctx->stream);
        if (m_hi) rc |= N->Recv(recv[k] + m_lo, (size_t)m_hi, PL_NCCL_DOUBLE, hi, N->comm, ctx->stream);
    }
    rc |= N->GroupEnd();
    if (rc) return pl_fail(ctx, "RCCL tracer migration failed");
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *got = m_lo + m_hi;
    return 0;
}

// ---- set-up -------------------------------------------------------------------------------------------
#define PL_SYM(field, name) \
    N->field = (decltype(N->field))dlsym(N->lib, name); \
    if (!N->field) good = false;

// Called from pl_set_comm on every rank (collective).  Never fails hard: on any problem the native
// path is simply left disabled - but the decision is taken collectively so that all ranks agree.
int pl_comm_native_init(pl_ctx* ctx) {
    // Opt-in (PYLAMP_RCCL=1; bench.py sets it for the nccl backend): the native path is self-tested at start-up and
    // falls back to the callback table, but it has not yet been exercised on a multi-GPU node.
    const char* e = getenv("PYLAMP_RCCL");
    const bool want = (e && atoi(e) != 0) || getenv("PYLAMP_RCCL_SELFTEST");
    PlNccl* N = new PlNccl();
    ctx->nccl = N;
    bool good = want;
    if (good) {
        N->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!N->lib) N->lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        good = N->lib != nullptr;
    }
    if (good) {
        PL_SYM(GetUniqueId, "ncclGetUniqueId") PL_SYM(CommInitRank, "ncclCommInitRank") PL_SYM(CommDestroy, "ncclCommDestroy")
        PL_SYM(Send, "ncclSend") PL_SYM(Recv, "ncclRecv") PL_SYM(AllGather, "ncclAllGather") PL_SYM(AllReduce, "ncclAllReduce")
        PL_SYM(GroupStart, "ncclGroupStart") PL_SYM(GroupEnd, "ncclGroupEnd")
        N->CommAbort = (decltype(N->CommAbort))dlsym(N->lib, "ncclCommAbort");      // optional
    }
    // 1. does every rank have the library?
    double flag[1] = {good ? 1.0 : 0.0};
    PL_TRY(pl_allreduce_host(ctx, flag, 1, 1));
    if (flag[0] < 0.5) return 0;
    // 2. unique id from rank 0, carried as 128 doubles through the host all-reduce
    pl_ncclUniqueId id; memset(&id, 0, sizeof(id));
    double idbuf[129]; for (double& v : idbuf) v = 0.0;
    if (ctx->rank == 0) {
        if (N->GetUniqueId(&id) == 0) { for (int k = 0; k < 128; k++) idbuf[k] = (double)(unsigned char)id.internal[k]; idbuf[128] = 1.0; }
    }
    PL_TRY(pl_allreduce_host(ctx, idbuf, 129, 0));
    if (idbuf[128] < 0.5) return 0;
    for (int k = 0; k < 128; k++) id.internal[k] = (char)(unsigned char)idbuf[k];
    PL_HIP(ctx, hipSetDevice(ctx->device));
    int rc = N->CommInitRank(&N->comm, ctx->nranks, id, ctx->rank);
    flag[0] = (rc == 0) ? 1.0 : 0.0;
    PL_TRY(pl_allreduce_host(ctx, flag, 1, 1));
    if (flag[0] < 0.5) { N->comm = nullptr; return 0; }
    // 3. self-test: ring exchange (forward and accumulating) and all-gather against known values
    N->ok = true;
    const long long cnt = 64; const int R = ctx->nranks, r = ctx->rank;
    double* t = nullptr;
    bool pass = pl_buf(ctx, "nccl_selftest", (size_t)(6 * cnt + R * cnt + 2) * sizeof(double), &t, true) == 0;
    // The self-test runs on a stream of its own and is given 60 s: a transport that hangs must not take the solver
    // stream (and the whole job) with it - the communicator is aborted and the callback table is used instead.
    hipStream_t main_stream = ctx->stream, test_stream = nullptr;
    if (pass && hipStreamCreateWithFlags(&test_stream, hipStreamNonBlocking) != hipSuccess) pass = false;
    if (pass) { (void)hipStreamSynchronize(main_stream); ctx->stream = test_stream; }
    bool hung = false;
    std::vector<double>& h = N->selftest_host;
    h.assign((size_t)(6 * cnt + R * cnt + 2), 0.0);
    if (pass) {
        // layout: [recv_lo | own_first | own_last | recv_hi | acc_lo | acc_hi | gather(R*cnt) | all-reduce(2)]
        hipLaunchKernelGGL(k_comm_fill, dim3(1), dim3(64), 0, ctx->stream, 2, t + 6 * cnt + (long long)R * cnt, 1.0 + r);
        hipLaunchKernelGGL(k_comm_fill, dim3(1), dim3(64), 0, ctx->stream, cnt, t + cnt, 100.0 + r);
        hipLaunchKernelGGL(k_comm_fill, dim3(1), dim3(64), 0, ctx->stream, cnt, t + 2 * cnt, 200.0 + r);
        hipLaunchKernelGGL(k_comm_fill, dim3(1), dim3(64), 0, ctx->stream, 2 * cnt, t + 4 * cnt, 1.0);
        hipLaunchKernelGGL(k_comm_fill, dim3(1), dim3(64), 0, ctx->stream, cnt, t + 6 * cnt + (long long)r * cnt, 300.0 + r);
        pass = native_exchange(ctx, t + cnt, t, t + 2 * cnt, t + 3 * cnt, cnt, 1, 0, 0) == 0 &&
               native_exchange(ctx, t + cnt, t + 4 * cnt, t + 2 * cnt, t + 5 * cnt, cnt, 1, 0, 1) == 0 &&
               native_allgather(ctx, t + 6 * cnt, cnt, 1, 0) == 0 &&
               pl_comm_allreduce_dev(ctx, t + 6 * cnt + (long long)R * cnt, 2) == 0 &&
               hipMemcpyAsync(h.data(), t, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        if (pass) {
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t q;
            while ((q = hipStreamQuery(ctx->stream)) == hipErrorNotReady) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) { hung = true; break; }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (hung || q != hipSuccess) pass = false;
        }
    }
    ctx->stream = main_stream;
    if (hung && N->comm) {                 // abort the communicator BEFORE the stream is abandoned: its kernels must leave the device
        if (N->CommAbort) (void)N->CommAbort(N->comm);
        N->comm = nullptr;
    }
    if (test_stream && !hung) (void)hipStreamDestroy(test_stream);       // a hung stream is abandoned
    if (pass) {
        for (long long k = 0; k < cnt && pass; k++) {
            if (r > 0 && (h[k] != 200.0 + (r - 1) || h[4 * cnt + k] != 1.0 + 200.0 + (r - 1))) pass = false;
            if (r < R - 1 && (h[3 * cnt + k] != 100.0 + (r + 1) || h[5 * cnt + k] != 1.0 + 100.0 + (r + 1))) pass = false;
            for (int q = 0; q < R && pass; q++) if (h[6 * cnt + (long long)q * cnt + k] != 300.0 + q) pass = false;
        }
        const double want = 0.5 * R * (R + 1);                     // sum over ranks of (1 + r)
        if (h[6 * cnt + (long long)R * cnt] != want || h[6 * cnt + (long long)R * cnt + 1] != want) pass = false;
    }
    flag[0] = pass ? 1.0 : 0.0;
    PL_TRY(pl_allreduce_host(ctx, flag, 1, 1));
    N->ok = flag[0] > 0.5;
    if (!N->ok && N->comm) {               // some rank failed or hung: nobody uses the communicator
        if (N->CommAbort) (void)N->CommAbort(N->comm);
        N->comm = nullptr;
    }
    return 0;
}

// in-place sum all-reduce of n doubles in DEVICE memory on the context stream (native path only)
int pl_comm_allreduce_dev(pl_ctx* ctx, double* dev, int n) {
    PlNccl* N = nccl_of(ctx);
    if (!(N && N->ok)) return 1;
    ctx->comm_calls[2]++;
    if (N->AllReduce(dev, dev, (size_t)n, PL_NCCL_DOUBLE, PL_NCCL_SUM, N->comm, ctx->stream))
        return pl_fail(ctx, "RCCL all-reduce failed");
    return 0;
}

int pl_comm_native_enabled(pl_ctx* ctx) { PlNccl* N = nccl_of(ctx); return (N && N->ok) ? 1 : 0; }
