// Communication layer of the 2-D block decomposition (Pz x Px blocks of the node grid).
//
// Primitives everything else is built from:
//   pl_comm_sendrecv   point-to-point messages between neighbour ranks (device buffers)
//   pl_comm_allgather  equal-sized contributions of all ranks
//   pl_comm_allreduce_dev / pl_allreduce_host   Krylov dot products, time-step reductions
// and on top of them
//   pl_halo / pl_halo_generic   halo exchange with the (up to 8) neighbour blocks, corners included, any depth up to
//                               PL_RING: ONE pack kernel, one message per neighbour, ONE unpack kernel
//   pl_gather_blocks            a replicated multigrid level assembled from the blocks the ranks computed
//
// Three transports behind the primitives:
//   (a) RCCL called directly on the context's HIP stream (ncclSend/ncclRecv groups over the direct xGMI links,
//       ncclAllGather, ncclAllReduce): stream-ordered, no host synchronisation per exchange.  librccl is dlopen'ed at
//       pl_set_comm time; opt-in (PYLAMP_RCCL=1) and self-tested collectively before it is used;
//   (b) the host callback table (pylamp_amd/parallel.py: torch.distributed, gloo staging or nccl tensors);
//   (c) an in-process group of virtual ranks (pl_local_group_*): every rank is a context driven by its own host
//       thread, messages are device-to-device copies between the contexts' buffers.  This is how a 2 x 4 layout is
//       rehearsed on ONE GPU; the pack / unpack kernels and all neighbour logic are the ones (a) uses.
#include "pl_internal.h"
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>

typedef struct { char internal[128]; } pl_ncclUniqueId;
typedef void* pl_ncclComm_t;
enum { PL_NCCL_DOUBLE = 8, PL_NCCL_SUM = 0, PL_NCCL_MAX = 2, PL_NCCL_MIN = 3 };       // ncclDataType_t / ncclRedOp_t values

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

// ---- in-process group of virtual ranks ------------------------------------------------------------------------
struct pl_local_group {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long long gen = 0;
    std::vector<pl_ctx*> ctx;
    std::vector<const PlMsg*> msgs; std::vector<int> nmsg;
    std::vector<const void*> ptr;
    bool failed = false;         // a rank gave up (error on its side, or a barrier timed out): every later barrier fails at once
    // PYLAMP_LOCAL_SERIAL=1 (profiling rehearsals): one GPU token for the group -- a rank's host thread holds it while it runs and
    // hands it over (with its stream drained) while it waits in a collective call, so that the kernels of the virtual ranks never
    // overlap on the shared GPU and a kernel trace reports their true durations (tools/rehearse_trace.sh)
    bool serialize = false, token_busy = false;
    std::condition_variable token_cv;
    void token_acquire(std::unique_lock<std::mutex>& lk) { token_cv.wait(lk, [&] { return !token_busy || failed; }); token_busy = true; }
    void token_release() { token_busy = false; token_cv.notify_one(); }
    // false: the group has failed -- some rank did not arrive (it raised an error before the collective call)
    bool barrier(pl_ctx* who = nullptr) {
        if (serialize && who) (void)hipStreamSynchronize(who->stream);
        std::unique_lock<std::mutex> lk(m);
        if (failed) return false;
        if (serialize) token_release();
        const long long g = gen;
        bool ok = true;
        if (++arrived == n) { arrived = 0; gen++; cv.notify_all(); }
        else if (!cv.wait_for(lk, std::chrono::seconds(300), [&] { return gen != g || failed; })) { failed = true; cv.notify_all(); token_cv.notify_all(); }
        ok = !failed;
        if (serialize && ok) token_acquire(lk);
        return ok && !failed;
    }
};

extern "C" int pl_local_group_create(pl_local_group** out, int nranks) {
    if (!out || nranks < 1) return 1;
    pl_local_group* g = new pl_local_group();
    g->n = nranks; g->ctx.assign(nranks, nullptr); g->msgs.assign(nranks, nullptr); g->nmsg.assign(nranks, 0); g->ptr.assign(nranks, nullptr);
    g->serialize = getenv("PYLAMP_LOCAL_SERIAL") && atoi(getenv("PYLAMP_LOCAL_SERIAL")) != 0;
    *out = g;
    return 0;
}
// a rank's host thread starts / ends a stretch of library calls (the driver wraps every collective call sequence in them): with
// PYLAMP_LOCAL_SERIAL it takes / returns the group's GPU token; otherwise nothing
extern "C" void pl_local_group_enter(pl_local_group* g) {
    if (!g || !g->serialize) return;
    std::unique_lock<std::mutex> lk(g->m);
    g->token_acquire(lk);
}
extern "C" void pl_local_group_leave(pl_local_group* g, pl_ctx* ctx) {
    if (!g || !g->serialize) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    std::unique_lock<std::mutex> lk(g->m);
    g->token_release();
}
extern "C" void pl_local_group_destroy(pl_local_group* g) { delete g; }
// wake every rank waiting in a collective call with an error (a rank's driver thread has failed)
extern "C" void pl_local_group_abort(pl_local_group* g) {
    if (!g) return;
    std::unique_lock<std::mutex> lk(g->m);
    g->failed = true;
    g->cv.notify_all(); g->token_cv.notify_all();
}

int pl_local_attach(pl_ctx* ctx, pl_local_group* g, int rank) {
    if (rank < 0 || rank >= g->n || g->n != ctx->nranks) return pl_fail(ctx, "pl_set_comm_local: rank / group size mismatch");
    g->ctx[rank] = ctx;
    ctx->local = g;
    return 0;
}
void pl_local_detach(pl_ctx* ctx) { ctx->local = nullptr; }
static pl_local_group* local_of(pl_ctx* ctx) { return (pl_local_group*)ctx->local; }

static int local_sendrecv(pl_ctx* ctx, const PlMsg* msgs, int nmsg) {
    pl_local_group* G = local_of(ctx);
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));              // my send buffers are complete
    G->msgs[ctx->rank] = msgs; G->nmsg[ctx->rank] = nmsg;
    if (!G->barrier(ctx)) return pl_fail(ctx, "in-process transport: another rank failed or did not arrive");
    int rc = 0;
    for (int k = 0; k < nmsg && !rc; k++) {
        if (msgs[k].nrecv <= 0) continue;
        const int p = msgs[k].peer;
        int ord = 0;                                             // k is my ord-th message with this peer
        for (int q = 0; q < k; q++) if (msgs[q].peer == p) ord++;
        const PlMsg* pm = G->msgs[p]; const PlMsg* hit = nullptr;
        for (int q = 0, seen = 0; q < G->nmsg[p]; q++)
            if (pm[q].peer == ctx->rank) { if (seen == ord) { hit = &pm[q]; break; } seen++; }
        if (!hit || hit->nsend != msgs[k].nrecv) {
            char why[200];
            snprintf(why, sizeof why, "in-process transport: unmatched message (rank %d, message %d of %d: %lld doubles expected from rank %d, which sends %lld in %d messages)",
                     ctx->rank, k, nmsg, (long long)msgs[k].nrecv, p, hit ? (long long)hit->nsend : -1LL, G->nmsg[p]);
            rc = pl_fail(ctx, why); break;
        }
        if (hipMemcpyAsync(msgs[k].recv, hit->send, (size_t)msgs[k].nrecv * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            rc = pl_fail(ctx, "in-process transport: device copy failed");
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = pl_fail(ctx, "in-process transport: synchronisation failed");
    if (rc) pl_local_group_abort(G);
    if (!G->barrier(ctx) && !rc) rc = pl_fail(ctx, "in-process transport: another rank failed or did not arrive");   // everybody has read: send buffers may be reused
    return rc;
}

static int local_allreduce_host(pl_ctx* ctx, double* buf, long long n, int op) {
    pl_local_group* G = local_of(ctx);
    G->ptr[ctx->rank] = buf;
    if (!G->barrier(ctx)) return pl_fail(ctx, "in-process transport: another rank failed or did not arrive");
    std::vector<double> tmp((size_t)n);
    for (long long k = 0; k < n; k++) {
        double a = ((const double*)G->ptr[0])[k];
        for (int r = 1; r < G->n; r++) {                        // fixed rank order: every rank gets the same bits
            const double b = ((const double*)G->ptr[r])[k];
            a = (op == 0) ? a + b : (op == 1 ? std::min(a, b) : std::max(a, b));
        }
        tmp[(size_t)k] = a;
    }
    if (!G->barrier(ctx)) return pl_fail(ctx, "in-process transport: another rank failed or did not arrive");
    memcpy(buf, tmp.data(), (size_t)n * sizeof(double));
    return 0;
}

static int local_allgather(pl_ctx* ctx, const double* send, double* recv, long long count) {
    pl_local_group* G = local_of(ctx);
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    G->ptr[ctx->rank] = send;
    if (!G->barrier(ctx)) return pl_fail(ctx, "in-process transport: another rank failed or did not arrive");
    int rc = 0;
    for (int r = 0; r < G->n && !rc; r++)
        if (hipMemcpyAsync(recv + (long long)r * count, G->ptr[r], (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            rc = pl_fail(ctx, "in-process transport: device copy failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = pl_fail(ctx, "in-process transport: synchronisation failed");
    if (rc) pl_local_group_abort(G);
    if (!G->barrier(ctx) && !rc) rc = pl_fail(ctx, "in-process transport: another rank failed or did not arrive");
    return rc;
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
int pl_comm_sendrecv(pl_ctx* ctx, const PlMsg* msgs, int nmsg) {
    if (ctx->nranks <= 1) return 0;
    PlNccl* N = nccl_of(ctx);
    if (N && N->ok) {
        int rc = N->GroupStart();
        for (int k = 0; k < nmsg && !rc; k++) {
            if (msgs[k].nsend > 0) rc |= N->Send(msgs[k].send, (size_t)msgs[k].nsend, PL_NCCL_DOUBLE, msgs[k].peer, N->comm, ctx->stream);
            if (msgs[k].nrecv > 0) rc |= N->Recv(msgs[k].recv, (size_t)msgs[k].nrecv, PL_NCCL_DOUBLE, msgs[k].peer, N->comm, ctx->stream);
        }
        rc |= N->GroupEnd();
        if (rc) return pl_fail(ctx, "RCCL send/recv group failed");
        return 0;
    }
    if (ctx->local) return local_sendrecv(ctx, msgs, nmsg);
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> peer((size_t)nmsg); std::vector<const double*> sp((size_t)nmsg); std::vector<double*> rp((size_t)nmsg);
    std::vector<int64_t> ns((size_t)nmsg), nr((size_t)nmsg);
    for (int k = 0; k < nmsg; k++) { peer[k] = msgs[k].peer; sp[k] = msgs[k].send; rp[k] = msgs[k].recv; ns[k] = msgs[k].nsend; nr[k] = msgs[k].nrecv; }
    if (ctx->comm.sendrecv(ctx->comm.user, nmsg, peer.data(), sp.data(), ns.data(), rp.data(), nr.data()))
        return pl_fail(ctx, "communication callback 'sendrecv' failed");
    return 0;
}

// ---- time spent in communication (pl_comm_times) -----------------------------------------------------------------------
#define PL_COMM_EV_MAX 8192
int pl_comm_time_begin(pl_ctx* ctx, int kind) {
    if (ctx->comm_ev_used >= PL_COMM_EV_MAX) return -1;
    const size_t slot = ctx->comm_ev_used;
    if (ctx->comm_ev.size() < 2 * (slot + 1)) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        ctx->comm_ev.push_back(a); ctx->comm_ev.push_back(b); ctx->comm_ev_kind.push_back(kind);
    }
    ctx->comm_ev_kind[slot] = kind;
    (void)hipEventRecord(ctx->comm_ev[2 * slot], ctx->stream);
    ctx->comm_ev_used++;
    return (int)slot;
}
void pl_comm_time_end(pl_ctx* ctx, int slot) { if (slot >= 0) (void)hipEventRecord(ctx->comm_ev[2 * slot + 1], ctx->stream); }
struct PlCommTimer {
    pl_ctx* ctx; int slot;
    PlCommTimer(pl_ctx* c, int kind) : ctx(c), slot(pl_comm_time_begin(c, kind)) {}
    ~PlCommTimer() { pl_comm_time_end(ctx, slot); }
};
extern "C" int pl_comm_times(pl_ctx* ctx, double out_ms[4], int reset) {
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t k = 0; k < ctx->comm_ev_used; k++) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->comm_ev[2 * k], ctx->comm_ev[2 * k + 1]) == hipSuccess) ctx->comm_ms[ctx->comm_ev_kind[k]] += ms;
    }
    ctx->comm_ev_used = 0;
    for (int k = 0; k < 4; k++) { if (out_ms) out_ms[k] = ctx->comm_ms[k]; if (reset) ctx->comm_ms[k] = 0.0; }
    return 0;
}

int pl_comm_allgather(pl_ctx* ctx, const double* send, double* recv, long long count) {
    if (ctx->nranks <= 1) return 0;
    ctx->comm_calls[1]++;
    PlCommTimer tm_(ctx, 1);
    PlNccl* N = nccl_of(ctx);
    if (N && N->ok) {
        if (N->AllGather(send, recv, (size_t)count, PL_NCCL_DOUBLE, N->comm, ctx->stream)) return pl_fail(ctx, "RCCL all-gather failed");
        return 0;
    }
    if (ctx->local) return local_allgather(ctx, send, recv, count);
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm.allgather(ctx->comm.user, send, recv, count)) return pl_fail(ctx, "communication callback 'allgather' failed");
    return 0;
}

int pl_allreduce_host(pl_ctx* ctx, double* buf, long long n, int op) {
    if (ctx->nranks <= 1) return 0;
    ctx->comm_calls[3]++;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = 0;
    if (ctx->local) rc = local_allreduce_host(ctx, buf, n, op);
    else if (ctx->comm.allreduce_host(ctx->comm.user, buf, n, op)) rc = pl_fail(ctx, "communication callback 'allreduce_host' failed");
    ctx->comm_ms[3] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// in-place all-reduce of n doubles in DEVICE memory (op: 0 sum, 1 min, 2 max): stream-ordered on the native transport (the host
// does not wait), staged through the host otherwise.  Counted as ONE device all-reduce on every transport.
int pl_comm_allreduce_dev(pl_ctx* ctx, double* dev, int n, int op) {
    if (ctx->nranks <= 1 || n <= 0) return 0;
    ctx->comm_calls[2]++;
    PlCommTimer tm_(ctx, 2);
    PlNccl* N = nccl_of(ctx);
    if (N && N->ok) {
        const int nop = op == 0 ? PL_NCCL_SUM : (op == 1 ? PL_NCCL_MIN : PL_NCCL_MAX);
        if (N->AllReduce(dev, dev, (size_t)n, PL_NCCL_DOUBLE, nop, N->comm, ctx->stream)) return pl_fail(ctx, "RCCL all-reduce failed");
        return 0;
    }
    double small[16];
    std::vector<double> big;
    double* h = small;
    if (n > 16) { big.resize((size_t)n); h = big.data(); }
    PL_HIP(ctx, hipMemcpyAsync(h, dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->comm_calls[3]--;                                        // counted as ONE (device) all-reduce
    PL_TRY(pl_allreduce_host(ctx, h, n, op));
    PL_HIP(ctx, hipMemcpyAsync(dev, h, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int pl_comm_native_enabled(pl_ctx* ctx) { PlNccl* N = nccl_of(ctx); return (N && N->ok) ? 1 : 0; }

extern "C" int pl_comm_info(pl_ctx* ctx, int* rank, int* nranks, int* native) {
    if (rank) *rank = ctx->rank;
    if (nranks) *nranks = ctx->nranks;
    if (native) *native = pl_comm_native_enabled(ctx) ? 1 : (ctx->local ? 2 : 0);
    return 0;
}

// ---- halo exchange --------------------------------------------------------------------------------------------------
// Regions are rectangles in LOCAL node coordinates (owned node (0,0) = origin); a segment holds the region of every
// plane, plane after plane.
struct HaloDesc {
    int n, nplanes;
    int r0[8], nr[8], c0[8], nc[8];
    long long off[9];                 // element offset of segment k in the buffer; off[n] = total
};

__device__ inline void halo_locate(const HaloDesc& d, long long t, int& k, int& q, int& i, int& j) {
    k = 0;
#pragma unroll
    for (int s = 1; s < 8; s++) if (s < d.n && t >= d.off[s]) k = s;
    const long long local = t - d.off[k];
    const long long per = (long long)d.nr[k] * d.nc[k];
    q = (int)(local / per);
    if (q >= d.nplanes) { q = -1; i = j = 0; return; }     // segments of float planes are padded to an even element count
    const int e = (int)(local % per);
    i = d.r0[k] + e / d.nc[k]; j = d.c0[k] + e % d.nc[k];
}
template <typename T>
__global__ __launch_bounds__(256) void k_halo_pack(HaloDesc d, const T* __restrict__ origin, long long pitch,
                                                   long long stride, T* __restrict__ buf) {
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < d.off[d.n]; t += (long long)gridDim.x * 256) {
        int k, q, i, j;
        halo_locate(d, t, k, q, i, j);
        if (q < 0) continue;                               // padding element of a single-precision segment
        buf[t] = origin[q * stride + (long long)i * pitch + j];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_halo_unpack(HaloDesc d, T* __restrict__ origin, long long pitch,
                                                     long long stride, const T* __restrict__ buf, int add) {
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < d.off[d.n]; t += (long long)gridDim.x * 256) {
        int k, q, i, j;
        halo_locate(d, t, k, q, i, j);
        if (q < 0) continue;
        T* p = origin + q * stride + (long long)i * pitch + j;
        // reverse halo: the owned strips facing N, W and NW overlap in the corner node -> three segments add to it
        if (add) unsafeAtomicAdd(p, buf[t]); else *p = buf[t];
    }
}

static const int DZ[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, DX[8] = {0, 0, -1, 1, -1, 1, -1, 1};

// T = double, or float (the FP32 multigrid levels): messages are counted in doubles, so every segment of a float halo is
// padded to an even number of elements
template <typename T>
static int halo_generic_t(pl_ctx* ctx, int lnz, int lnx, T* origin, long long pitch, int nplanes, long long stride, int depth, bool add) {
    if (ctx->nranks <= 1) return 0;
    if (depth < 1 || depth > PL_RING || depth > lnz || depth > lnx) return pl_fail(ctx, "pl_halo: bad depth");
    ctx->comm_calls[0]++;
    PlCommTimer tm_(ctx, 0);               // pack -> messages -> unpack
    const int per_double = (int)(sizeof(double) / sizeof(T));
    HaloDesc S{}, R{};
    int peers[8];
    long long off = 0;
    int n = 0;
    for (int k = 0; k < 8; k++) {
        const int qz = ctx->pz + DZ[k], qx = ctx->px + DX[k];
        if (qz < 0 || qz >= ctx->Pz || qx < 0 || qx >= ctx->Px) continue;
        // owned boundary strip facing the neighbour, and the ring strip beyond it
        const int or0 = DZ[k] < 0 ? 0 : (DZ[k] > 0 ? lnz - depth : 0), onr = DZ[k] ? depth : lnz;
        const int oc0 = DX[k] < 0 ? 0 : (DX[k] > 0 ? lnx - depth : 0), onc = DX[k] ? depth : lnx;
        const int rr0 = DZ[k] < 0 ? -depth : (DZ[k] > 0 ? lnz : 0), rc0 = DX[k] < 0 ? -depth : (DX[k] > 0 ? lnx : 0);
        // forward: owned strip -> neighbour's ring; reverse (add): my ring -> neighbour's owned strip
        S.r0[n] = add ? rr0 : or0; S.c0[n] = add ? rc0 : oc0; S.nr[n] = onr; S.nc[n] = onc;
        R.r0[n] = add ? or0 : rr0; R.c0[n] = add ? oc0 : rc0; R.nr[n] = onr; R.nc[n] = onc;
        S.off[n] = R.off[n] = off;
        long long cnt = (long long)nplanes * onr * onc;
        cnt = (cnt + per_double - 1) / per_double * per_double;
        off += cnt;
        peers[n++] = qz * ctx->Px + qx;
    }
    S.n = R.n = n; S.nplanes = R.nplanes = nplanes; S.off[n] = R.off[n] = off;
    if (n == 0) return 0;
    double *sbd, *rbd;
    PL_TRY(pl_buf(ctx, "halo_send", (size_t)off * sizeof(T), &sbd, false));
    PL_TRY(pl_buf(ctx, "halo_recv", (size_t)off * sizeof(T), &rbd, false));
    T* sb = reinterpret_cast<T*>(sbd); T* rb = reinterpret_cast<T*>(rbd);
    const unsigned nb = (unsigned)std::min<long long>((off + 255) / 256, 1024);
    hipLaunchKernelGGL(k_halo_pack<T>, dim3(nb), dim3(256), 0, ctx->stream, S, (const T*)origin, pitch, stride, sb);
    PlMsg msgs[8];
    for (int k = 0; k < n; k++) {
        const long long cnt = (S.off[k + 1] - S.off[k]) / per_double;          // in doubles
        msgs[k] = PlMsg{peers[k], sbd + S.off[k] / per_double, cnt, rbd + R.off[k] / per_double, cnt};
    }
    PL_TRY(pl_comm_sendrecv(ctx, msgs, n));
    hipLaunchKernelGGL(k_halo_unpack<T>, dim3(nb), dim3(256), 0, ctx->stream, R, origin, pitch, stride, (const T*)rb, add ? 1 : 0);
    PL_HIP(ctx, hipGetLastError());
    return 0;
}

int pl_halo_generic(pl_ctx* ctx, int lnz, int lnx, double* origin, long long pitch, int nplanes, long long stride, int depth, bool add) {
    return halo_generic_t<double>(ctx, lnz, lnx, origin, pitch, nplanes, stride, depth, add);
}

int pl_halo(pl_ctx* ctx, const PlGeom& g, double* planes, int nplanes, long long plane_stride, int depth, bool add) {
    if (ctx->nranks <= 1) return 0;
    return pl_halo_generic(ctx, g.lnz, g.lnx, planes + pl_idx(g, 0, 0), g.pitch, nplanes, plane_stride, depth, add);
}
int pl_halo(pl_ctx* ctx, const PlGeom& g, float* planes, int nplanes, long long plane_stride, int depth, bool add) {
    if (ctx->nranks <= 1) return 0;
    return halo_generic_t<float>(ctx, g.lnz, g.lnx, planes + pl_idx(g, 0, 0), g.pitch, nplanes, plane_stride, depth, add);
}

// ---- replicated level from per-rank blocks ------------------------------------------------------------------------
struct GatherDesc { int Pz, Px, nz, nx, maxblk, nplanes; };
__device__ inline void gather_block(const GatherDesc& d, int r, int& i0, int& ni, int& j0, int& nj) {
    const int pz = r / d.Px, px = r % d.Px, Cz = (d.nz - 1) / d.Pz, Cx = (d.nx - 1) / d.Px;
    i0 = pz * Cz; ni = (pz == d.Pz - 1) ? Cz + 1 : Cz; j0 = px * Cx; nj = (px == d.Px - 1) ? Cx + 1 : Cx;
}
// buf layout: rank-major, then plane, then the block row-major (padded to maxblk)
__global__ __launch_bounds__(256) void k_blocks_copy(GatherDesc d, int rank_lo, int rank_hi, double* __restrict__ origin, long long pitch,
                                                     long long stride, double* __restrict__ buf, int to_buf) {
    const long long per = (long long)d.nplanes * d.maxblk, total = (long long)(rank_hi - rank_lo) * per;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int r = rank_lo + (int)(t / per);
        const long long l = t % per;
        const int q = (int)(l / d.maxblk), e = (int)(l % d.maxblk);
        int i0, ni, j0, nj;
        gather_block(d, r, i0, ni, j0, nj);
        if (e >= ni * nj) continue;
        double* p = origin + q * stride + (long long)(i0 + e / nj) * pitch + (j0 + e % nj);
        double* b = buf + (to_buf ? l : t);                      // the send buffer holds one rank's share only
        if (to_buf) *b = *p; else *p = *b;
    }
}

int pl_gather_blocks(pl_ctx* ctx, const PlGeom& g, double* planes, int nplanes, long long stride) {
    if (ctx->nranks <= 1) return 0;
    if ((g.nz - 1) % ctx->Pz || (g.nx - 1) % ctx->Px) return pl_fail(ctx, "pl_gather_blocks: level not divisible into blocks");
    GatherDesc d{ctx->Pz, ctx->Px, g.nz, g.nx, 0, nplanes};
    d.maxblk = ((g.nz - 1) / ctx->Pz + 1) * ((g.nx - 1) / ctx->Px + 1);
    const long long per = (long long)nplanes * d.maxblk;
    double *sb, *rb;
    PL_TRY(pl_buf(ctx, "gather_send", (size_t)per * sizeof(double), &sb, true));
    PL_TRY(pl_buf(ctx, "gather_recv", (size_t)per * ctx->nranks * sizeof(double), &rb, false));
    double* origin = planes + pl_idx(g, -g.gi0, -g.gj0);        // global node (0,0)
    const unsigned nb1 = (unsigned)std::min<long long>((per + 255) / 256, 1024), nbR = (unsigned)std::min<long long>((per * ctx->nranks + 255) / 256, 2048);
    hipLaunchKernelGGL(k_blocks_copy, dim3(nb1), dim3(256), 0, ctx->stream, d, ctx->rank, ctx->rank + 1, origin, (long long)g.pitch, stride, sb, 1);
    PL_TRY(pl_comm_allgather(ctx, sb, rb, per));
    hipLaunchKernelGGL(k_blocks_copy, dim3(nbR), dim3(256), 0, ctx->stream, d, 0, ctx->nranks, origin, (long long)g.pitch, stride, rb, 0);
    PL_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- native transport set-up ---------------------------------------------------------------------------------------
void pl_comm_native_free(pl_ctx* ctx) {
    PlNccl* N = nccl_of(ctx);
    if (!N) return;
    if (N->comm && N->CommDestroy) (void)N->CommDestroy(N->comm);
    // the library handle is left open on purpose (torch may share it)
    delete N;
    ctx->nccl = nullptr;
}

__global__ void k_comm_fill(long long n, double* __restrict__ d, double v) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k < n; k += (long long)gridDim.x * blockDim.x) d[k] = v;
}

#define PL_SYM(field, name) \
    N->field = (decltype(N->field))dlsym(N->lib, name); \
    if (!N->field) good = false;

// Called from pl_set_comm on every rank (collective).  Never fails hard: on any problem the native
// path is simply left disabled - but the decision is taken collectively so that all ranks agree.
int pl_comm_native_init(pl_ctx* ctx) {
    // Opt-in (PYLAMP_RCCL=1; bench.py sets it for --transport native only): the native path is self-tested at start-up and
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
    // 1. does every rank want it and have the library?
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
    // 3. self-test: ring send/recv in both directions, all-gather and all-reduce against known values
    N->ok = true;
    const long long cnt = 64; const int R = ctx->nranks, r = ctx->rank;
    const long long total = 4 * cnt + (long long)R * cnt + 2;
    double* t = nullptr;
    bool pass = pl_buf(ctx, "nccl_selftest", (size_t)total * sizeof(double), &t, true) == 0;
    // The self-test runs on a stream of its own and is given 60 s: a transport that hangs must not take the solver
    // stream (and the whole job) with it - the communicator is aborted and the callback table is used instead.
    hipStream_t main_stream = ctx->stream, test_stream = nullptr;
    if (pass && hipStreamCreateWithFlags(&test_stream, hipStreamNonBlocking) != hipSuccess) pass = false;
    if (pass) { (void)hipStreamSynchronize(main_stream); ctx->stream = test_stream; }
    bool hung = false;
    std::vector<double>& h = N->selftest_host;
    h.assign((size_t)total, 0.0);
    if (pass) {
        // layout: [recv_from_prev | send_to_prev | send_to_next | recv_from_next | gather(R*cnt) | all-reduce(2)]
        hipLaunchKernelGGL(k_comm_fill, dim3(1), dim3(64), 0, ctx->stream, cnt, t + cnt, 100.0 + r);
        hipLaunchKernelGGL(k_comm_fill, dim3(1), dim3(64), 0, ctx->stream, cnt, t + 2 * cnt, 200.0 + r);
        hipLaunchKernelGGL(k_comm_fill, dim3(1), dim3(64), 0, ctx->stream, 2, t + 4 * cnt + (long long)R * cnt, 1.0 + r);
        double* mine = nullptr;
        pass = pl_buf(ctx, "nccl_selftest_mine", (size_t)cnt * sizeof(double), &mine, false) == 0;
        if (pass) hipLaunchKernelGGL(k_comm_fill, dim3(1), dim3(64), 0, ctx->stream, cnt, mine, 300.0 + r);
        PlMsg m[2]; int nm = 0;
        if (r > 0) m[nm++] = PlMsg{r - 1, t + cnt, cnt, t, cnt};
        if (r < R - 1) m[nm++] = PlMsg{r + 1, t + 2 * cnt, cnt, t + 3 * cnt, cnt};
        pass = pass && (nm == 0 || pl_comm_sendrecv(ctx, m, nm) == 0) &&
               N->AllGather(mine, t + 4 * cnt, (size_t)cnt, PL_NCCL_DOUBLE, N->comm, ctx->stream) == 0 &&
               N->AllReduce(t + 4 * cnt + (long long)R * cnt, t + 4 * cnt + (long long)R * cnt, 2, PL_NCCL_DOUBLE, PL_NCCL_SUM, N->comm, ctx->stream) == 0;
        // The deadline is watched on an EVENT recorded behind the collectives: nothing on the host may wait on the stream
        // before it has fired.  (A device-to-host copy into pageable memory -- what this used to enqueue -- blocks the host
        // until the stream reaches it, so a hung send / receive never got to the polling loop: ADVICE r2.)
        hipEvent_t done = nullptr;
        pass = pass && hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess && hipEventRecord(done, ctx->stream) == hipSuccess;
        if (pass) {
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t q;
            while ((q = hipEventQuery(done)) == hipErrorNotReady) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) { hung = true; break; }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (hung || q != hipSuccess) pass = false;
        }
        // the stream is idle now: a plain (synchronous) copy cannot block on the transport any more
        if (pass && hipMemcpy(h.data(), t, h.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) pass = false;
        if (done && !hung) (void)hipEventDestroy(done);
    }
    ctx->stream = main_stream;
    if (hung && N->comm) {                 // abort the communicator BEFORE the stream is abandoned: its kernels must leave the device
        if (N->CommAbort) (void)N->CommAbort(N->comm);
        N->comm = nullptr;
    }
    if (test_stream && !hung) (void)hipStreamDestroy(test_stream);       // a hung stream is abandoned
    if (pass) {
        for (long long k = 0; k < cnt && pass; k++) {
            if (r > 0 && h[k] != 200.0 + (r - 1)) pass = false;
            if (r < R - 1 && h[3 * cnt + k] != 100.0 + (r + 1)) pass = false;
            for (int q = 0; q < R && pass; q++) if (h[4 * cnt + (long long)q * cnt + k] != 300.0 + q) pass = false;
        }
        const double want_sum = 0.5 * R * (R + 1);                 // sum over ranks of (1 + r)
        if (h[4 * cnt + (long long)R * cnt] != want_sum || h[4 * cnt + (long long)R * cnt + 1] != want_sum) pass = false;
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
