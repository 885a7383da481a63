// comm.cpp -- RCCL (over xGMI) exchange step of the row-sharded scan.
//
// The reference is single-process; this exchange exists only because the build shards a
// scan by image rows across the GPUs of one node.  One context = one rank = one GPU.  RCCL is loaded lazily with
// dlopen so single-GPU users never pay for it.  RCCL has no all-gatherv: equal shards laid out back to back take
// ncclAllGather, ragged ones a grouped ncclBroadcast (one per contributing rank).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "slgc_internal.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;            // the three below: RCCL's own view of a communicator (slgc_comm_info); optional
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            x.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (x.handle) break;
        }
        if (!x.handle) return x;
#define LOAD(sym) x.sym = reinterpret_cast<decltype(x.sym)>(dlsym(x.handle, "nccl" #sym))
        LOAD(GetUniqueId); LOAD(CommInitRank); LOAD(CommDestroy); LOAD(AllReduce); LOAD(AllGather);
        LOAD(Broadcast); LOAD(GroupStart); LOAD(GroupEnd); LOAD(GetErrorString);
        LOAD(CommCount); LOAD(CommUserRank); LOAD(CommCuDevice);
#undef LOAD
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.AllGather && x.Broadcast &&
               x.GroupStart && x.GroupEnd && x.GetErrorString;
        return x;
    }();
    return r;
}

#define NCCL_TRY(ctx, expr)                                                                                  \
    do {                                                                                                     \
        ncclResult_t r__ = (expr);                                                                           \
        if (r__ != ncclSuccess) return slgc_fail((ctx), SLGC_ECOMM, "%s: %s", #expr, rccl().GetErrorString(r__)); \
    } while (0)

int need_comm(slgc_ctx *ctx)
{
    if (!ctx) return SLGC_EINVAL;
    if (!ctx->comm) return slgc_fail(ctx, SLGC_ECOMM, "communicator not initialised (call slgc_comm_init)");
    return SLGC_OK;
}

// All collectives run on ctx->comm_stream.  Before one is enqueued the communication stream is made to wait for everything
// the caller has enqueued on the compute stream so far (its inputs), so collectives stay ordered after the kernels that
// produce their data while later kernels on the compute stream are free to overlap with them.
int comm_after_compute(slgc_ctx *ctx)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_compute, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_compute, 0));
    return SLGC_OK;
}

}  // namespace

extern "C" int slgc_comm_unique_id(void *id128)
{
    if (!id128) return SLGC_EINVAL;
    if (!rccl().ok) return SLGC_ECOMM;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return SLGC_ECOMM;
    static_assert(sizeof(id) == SLGC_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof id);
    return SLGC_OK;
}

extern "C" int slgc_comm_init(slgc_ctx *ctx, int rank, int nranks, const void *id128)
{
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return slgc_fail(ctx, SLGC_EINVAL, "bad rank/nranks");
    if (ctx->comm) return slgc_fail(ctx, SLGC_ESTATE, "communicator already initialised");
    if (!rccl().ok) return slgc_fail(ctx, SLGC_ECOMM, "librccl.so not loadable: %s", dlerror());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm;
    NCCL_TRY(ctx, rccl().CommInitRank(&comm, nranks, id, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    HIP_TRY(ctx, hipMalloc(&ctx->comm_scratch, SLGC_BUS_ID_BYTES * (size_t)(nranks + 1)));      // an int64 or a bus id per rank, + this rank's own
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_compute, hipEventDisableTiming));
    for (int i = 0; i < 4; ++i) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_comm_done[i], hipEventDisableTiming));
    return SLGC_OK;
}

extern "C" int slgc_comm_destroy(slgc_ctx *ctx)
{
    if (!ctx) return SLGC_EINVAL;
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
        rccl().CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    if (ctx->comm_stream) {
        (void)hipStreamDestroy(ctx->comm_stream);
        ctx->comm_stream = nullptr;
        (void)hipEventDestroy(ctx->ev_compute);
        for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ctx->ev_comm_done[i]);
    }
    if (ctx->comm_scratch) {
        (void)hipFree(ctx->comm_scratch);
        ctx->comm_scratch = nullptr;
    }
    ctx->nranks = 0;
    return SLGC_OK;
}

extern "C" int slgc_comm_allreduce_max_f64(slgc_ctx *ctx, double *value)
{
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (!value) return slgc_fail(ctx, SLGC_EINVAL, "null value");
    if ((rc = comm_after_compute(ctx))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->comm_scratch, value, 8, hipMemcpyHostToDevice, ctx->comm_stream));
    NCCL_TRY(ctx, rccl().AllReduce(ctx->comm_scratch, ctx->comm_scratch, 1, ncclFloat64, ncclMax, (ncclComm_t)ctx->comm, ctx->comm_stream));
    HIP_TRY(ctx, hipMemcpyAsync(value, ctx->comm_scratch, 8, hipMemcpyDeviceToHost, ctx->comm_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    return SLGC_OK;
}

extern "C" int slgc_comm_barrier(slgc_ctx *ctx)
{
    double one = 1.0;
    return slgc_comm_allreduce_max_f64(ctx, &one);
}

extern "C" int slgc_comm_allgather_i64(slgc_ctx *ctx, int64_t mine, int64_t *all)
{
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (!all) return slgc_fail(ctx, SLGC_EINVAL, "null output");
    int64_t *scratch = (int64_t *)ctx->comm_scratch;  // [0] = mine, [1..nranks] = gathered
    if ((rc = comm_after_compute(ctx))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(scratch, &mine, 8, hipMemcpyHostToDevice, ctx->comm_stream));
    NCCL_TRY(ctx, rccl().AllGather(scratch, scratch + 1, 1, ncclInt64, (ncclComm_t)ctx->comm, ctx->comm_stream));
    HIP_TRY(ctx, hipMemcpyAsync(all, scratch + 1, 8 * (size_t)ctx->nranks, hipMemcpyDeviceToHost, ctx->comm_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    return SLGC_OK;
}

// RCCL's own account of the communicator -- not what the caller was told: bench.py's N > 1 line quotes these.
extern "C" int slgc_comm_info(slgc_ctx *ctx, int *nranks, int *rank, int *device)
{
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (!rccl().CommCount || !rccl().CommUserRank || !rccl().CommCuDevice)
        return slgc_fail(ctx, SLGC_ECOMM, "this librccl does not export ncclCommCount / ncclCommUserRank / ncclCommCuDevice");
    int n = -1, r = -1, d = -1;
    NCCL_TRY(ctx, rccl().CommCount((ncclComm_t)ctx->comm, &n));
    NCCL_TRY(ctx, rccl().CommUserRank((ncclComm_t)ctx->comm, &r));
    NCCL_TRY(ctx, rccl().CommCuDevice((ncclComm_t)ctx->comm, &d));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (device) *device = d;
    return SLGC_OK;
}

// Every rank's PCI bus id (hipDeviceGetPCIBusId of the context's device), all-gathered through the communicator itself: SLGC_BUS_ID_BYTES
// NUL-padded bytes per rank, in rank order.  *distinct = how many different devices the ranks sit on (N ranks on N GPUs: N).
extern "C" int slgc_comm_allgather_bus_ids(slgc_ctx *ctx, char *ids, int *distinct)
{
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (!ids) return slgc_fail(ctx, SLGC_EINVAL, "null output");
    char mine[SLGC_BUS_ID_BYTES] = {0};
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceGetPCIBusId(mine, SLGC_BUS_ID_BYTES - 1, ctx->device));
    char *scratch = (char *)ctx->comm_scratch;      // [0] = mine, [1..nranks] = gathered
    if ((rc = comm_after_compute(ctx))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(scratch, mine, SLGC_BUS_ID_BYTES, hipMemcpyHostToDevice, ctx->comm_stream));
    NCCL_TRY(ctx, rccl().AllGather(scratch, scratch + SLGC_BUS_ID_BYTES, SLGC_BUS_ID_BYTES, ncclUint8, (ncclComm_t)ctx->comm, ctx->comm_stream));
    HIP_TRY(ctx, hipMemcpyAsync(ids, scratch + SLGC_BUS_ID_BYTES, SLGC_BUS_ID_BYTES * (size_t)ctx->nranks, hipMemcpyDeviceToHost, ctx->comm_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    if (distinct) {
        int n = 0;
        for (int a = 0; a < ctx->nranks; ++a) {
            bool seen = false;
            for (int b = 0; b < a && !seen; ++b) seen = memcmp(ids + (size_t)a * SLGC_BUS_ID_BYTES, ids + (size_t)b * SLGC_BUS_ID_BYTES, SLGC_BUS_ID_BYTES) == 0;
            n += !seen;
        }
        *distinct = n;
    }
    return SLGC_OK;
}

// One or two buffers with the same shard layout, one RCCL group (one launch): the map exchange sends h and v together.
static int allgatherv_enqueue(slgc_ctx *ctx, int nbuf, const void *const *d_send, void *const *d_recv, const int64_t *counts,
                              const int64_t *displs, int slot)
{
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (!counts || !displs || slot < 0 || slot > 3) return slgc_fail(ctx, SLGC_EINVAL, "null argument / slot outside 0..3");
    for (int b = 0; b < nbuf; ++b)
        if (!d_recv[b]) return slgc_fail(ctx, SLGC_EINVAL, "null receive buffer");
    for (int r = 0; r < ctx->nranks; ++r)
        if (counts[r] < 0 || displs[r] < 0) return slgc_fail(ctx, SLGC_EINVAL, "negative count/displacement");
    // Equal shards laid out back to back (the row-band plan when H % nranks == 0): RCCL's native all-gather, in place when the
    // caller's shard already sits in its slot.  Anything else: one broadcast per contributing rank.
    bool uniform = counts[0] > 0;
    for (int r = 0; r < ctx->nranks && uniform; ++r) uniform = counts[r] == counts[0] && displs[r] == (int64_t)r * counts[0];
    if (uniform)
        for (int b = 0; b < nbuf; ++b)
            if (!d_send[b]) return slgc_fail(ctx, SLGC_EINVAL, "null send buffer");
    if ((rc = comm_after_compute(ctx))) return rc;
    NCCL_TRY(ctx, rccl().GroupStart());
    ncclResult_t e = ncclSuccess;
    for (int b = 0; b < nbuf && e == ncclSuccess; ++b) {
        if (uniform) {
            e = rccl().AllGather(d_send[b], d_recv[b], (size_t)counts[0], ncclUint8, (ncclComm_t)ctx->comm, ctx->comm_stream);
            continue;
        }
        for (int r = 0; r < ctx->nranks && e == ncclSuccess; ++r) {
            if (counts[r] == 0) continue;  // same decision on every rank (counts are global)
            char *dst = (char *)d_recv[b] + displs[r];
            const void *src = (r == ctx->rank) ? d_send[b] : dst;
            e = rccl().Broadcast(src, dst, (size_t)counts[r], ncclUint8, r, (ncclComm_t)ctx->comm, ctx->comm_stream);
        }
    }
    if (e != ncclSuccess) {
        rccl().GroupEnd();
        return slgc_fail(ctx, SLGC_ECOMM, "all-gatherv enqueue: %s", rccl().GetErrorString(e));
    }
    NCCL_TRY(ctx, rccl().GroupEnd());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_comm_done[slot], ctx->comm_stream));
    return SLGC_OK;
}

extern "C" int slgc_comm_allgatherv_begin(slgc_ctx *ctx, const void *d_send, void *d_recv, const int64_t *counts, const int64_t *displs, int slot)
{
    return allgatherv_enqueue(ctx, 1, &d_send, &d_recv, counts, displs, slot);
}

extern "C" int slgc_comm_allgatherv_pair_begin(slgc_ctx *ctx, const void *d_send_a, void *d_recv_a, const void *d_send_b, void *d_recv_b,
                                               const int64_t *counts, const int64_t *displs, int slot)
{
    const void *send[2] = {d_send_a, d_send_b};
    void *recv[2] = {d_recv_a, d_recv_b};
    return allgatherv_enqueue(ctx, 2, send, recv, counts, displs, slot);
}

extern "C" int slgc_comm_wait(slgc_ctx *ctx, int slot)
{
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (slot < 0 || slot > 3) return slgc_fail(ctx, SLGC_EINVAL, "slot outside 0..3");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_comm_done[slot], 0));   // later compute-stream work sees the gathered data
    return SLGC_OK;
}

extern "C" int slgc_comm_allgatherv(slgc_ctx *ctx, const void *d_send, void *d_recv, const int64_t *counts, const int64_t *displs)
{
    int rc = slgc_comm_allgatherv_begin(ctx, d_send, d_recv, counts, displs, 3);
    if (rc) return rc;
    return slgc_comm_wait(ctx, 3);
}

extern "C" int slgc_shard_band(int H, int nranks, int rank, int *row0, int *rows)
{
    if (H < 0 || nranks < 1 || rank < 0 || rank >= nranks || !row0 || !rows) return SLGC_EINVAL;
    const int base = H / nranks, extra = H % nranks;
    *rows = base + (rank < extra ? 1 : 0);
    *row0 = rank * base + (rank < extra ? rank : extra);
    return SLGC_OK;
}

extern "C" int slgc_scan_sharded_dev(slgc_ctx *ctx, const uint8_t *d_band_stack, int n_runs, size_t run_stride, size_t plane_stride,
                                     int N, int H, int W, int proj_w, int proj_h, double eps, double m, int mode, int16_t *d_h_full,
                                     int16_t *d_v_full, float *d_xyz_full)
{
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (!d_h_full || !d_v_full || !d_xyz_full) return slgc_fail(ctx, SLGC_EINVAL, "null output");
    if (H < 0 || W < 0 || ctx->nranks > 1024) return slgc_fail(ctx, SLGC_EINVAL, "bad image size / nranks");
    int64_t counts[1024], displs[1024];
    int row0 = 0, rows = 0;
    for (int r = 0; r < ctx->nranks; ++r) {
        int b0, bn;
        slgc_shard_band(H, ctx->nranks, r, &b0, &bn);
        counts[r] = (int64_t)bn * W * 2;          // bytes of one int16 map band
        displs[r] = (int64_t)b0 * W * 2;
        if (r == ctx->rank) { row0 = b0; rows = bn; }
    }
    int16_t *my_h = d_h_full + (size_t)row0 * W, *my_v = d_v_full + (size_t)row0 * W;
    if (rows > 0 && (rc = slgc_decode_dev(ctx, d_band_stack, n_runs, run_stride, plane_stride, N, rows, W, eps, m, my_h, my_v, 0))) return rc;
    const int code_bits = (int)((double)(N - 2) / 4.0);
    if (ctx->tune_wire && code_bits <= SLGC_WIRE_MAX_CODE_BITS && (mode & 3) <= SLGC_TRI_ALGEBRAIC) {
        // 3-byte wire format (slgc_tune "wire" = 1): pack the band into its slot of a packed full-size buffer, ONE in-place all-gather of
        // 3 B/pixel, and the triangulation kernel unpacks while it loads (it also writes the int16 maps, which stay a product)
        void *wire;
        if ((rc = slgc_ws(ctx, 13, (size_t)H * W * 3 + 64, &wire))) return rc;
        for (int r = 0; r < ctx->nranks; ++r) {
            counts[r] = counts[r] / 2 * 3;
            displs[r] = displs[r] / 2 * 3;
        }
        uint8_t *my_w = (uint8_t *)wire + displs[ctx->rank];
        if (rows > 0 && (rc = slgc_pack_hv24_dev(ctx, my_h, my_v, (size_t)rows * W, code_bits, my_w))) return rc;
        if ((rc = slgc_comm_allgatherv_begin(ctx, my_w, wire, counts, displs, 3))) return rc;
        if ((rc = slgc_comm_wait(ctx, 3))) return rc;
        return slgc_triangulate_wire_dev(ctx, (const uint8_t *)wire, H, W, 0, proj_w, proj_h, mode & 1, d_h_full, d_v_full, d_xyz_full, nullptr);
    }
    if ((rc = slgc_comm_allgatherv_pair_begin(ctx, my_h, d_h_full, my_v, d_v_full, counts, displs, 3))) return rc;
    if ((rc = slgc_comm_wait(ctx, 3))) return rc;
    return slgc_triangulate_maps_dev(ctx, d_h_full, d_v_full, H, W, 0, proj_w, proj_h, mode & 3, d_xyz_full, nullptr);
}
