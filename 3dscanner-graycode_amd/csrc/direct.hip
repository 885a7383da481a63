// direct.hip -- the exchange step of the row-sharded scan written for this machine: every rank pushes its band straight into every
// peer's full-size buffer over xGMI, all G-1 links busy at once ("direct all-gather").
//
// The reference is single-process; this exists only because the build shards ONE scan by image rows over the GPUs of a node
// (BASELINE.json configs[3]).  RCCL's all-gather is a ring (per-link bound, G-1 steps) and its ragged form is a group of broadcasts;
// MI355X's xGMI is a full point-to-point mesh (7 links per GPU), so an all-gather of row bands is G-1 independent band copies per rank
// that can all be in flight together.  How it is built:
//   * one POSIX shared-memory segment per job (name = the job's key): IPC handles of the registered buffers, the flag words, a host barrier.
//     Every rank page-locks the segment (hipHostRegister): host memory is fine-grained, so flag words written by one GPU (or the host) are
//     seen by another GPU's system-scope loads -- device memory shared over IPC is coarse-grained and gives no such promise inside a kernel;
//   * slgc_direct_register (collective): each rank exports the hipIpcMemHandle of a full-size buffer, opens every peer's;
//   * slgc_direct_allgatherv_begin on up to 3 registered buffers with one band layout (h + v maps, or + XYZ), on the exchange stream:
//       gate kernel   waits until every peer has RELEASED the buffers (their previous contents were consumed there: slgc_direct_release),
//       push kernel   copies the rank's band from its own buffer into the same place of every peer's buffer (16-byte lanes, one workgroup
//                     per (peer, chunk): every link carries its band at the same time; system-scope write-through stores, acknowledged
//                     before the wave ends),
//       flag kernel   writes the exchange's sequence number into arrived[peer][rank][buffer] of every peer;
//   * slgc_direct_wait(slot), on the compute stream: a one-wave kernel polls arrived[rank][*][buffer] until every peer's band of that
//     sequence number is in.  The kernels that read the bands start AFTER it ends: the acquire at their start is what makes remotely
//     written coarse-grained memory visible (no kernel reads a byte a peer wrote while it runs).
// Polling kernels are one wave, poll with relaxed system-scope loads (one acquire fence after the loop), sleep between polls and give up:
// kDirectTimeoutS seconds after the PEER'S HOST has submitted the work that raises the flag (it says so in the segment when it enqueues it), or
// kDirectStartTimeoutS seconds if the peer never gets that far -- ranks whose hosts are merely out of step (a late capture, a first-call
// table build) wait for each other like RCCL would, a lost peer costs a failed scan, never a hung GPU.  A timeout is sticky (error word in the
// segment): every later kernel of the exchange skips its work, and slgc_synchronize / slgc_d2h / every slgc_direct_* call return SLGC_ECOMM --
// a scan that timed out never comes back as data.
// hipIpcOpenMemHandle also works between processes that share ONE GPU, so tests/test_gpu_rccl_multi.py runs this path bit for bit on the
// one-GPU test box; the links themselves are only exercised on a real node.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdlib>
#include <new>

#include "slgc_internal.h"

namespace {

constexpr int kMaxRanks = 16;
constexpr int kMaxBufs = 16;
constexpr uint32_t kMagic = 0x534c4744u;       // "SLGD"
constexpr double kDirectTimeoutS = 20.0;       // GPU-side polls, once the peer's host has submitted its side (s_memrealtime runs at 100 MHz)
constexpr double kDirectStartTimeoutS = 300.0; // GPU-side polls while the peer's host has not yet submitted its side (host skew between ranks)
constexpr double kHostTimeoutS = 120.0;        // host barrier (SLGC_DIRECT_HOST_TIMEOUT_S)

struct Segment {                               // lives in the shared-memory segment, identical view in every rank
    std::atomic<uint32_t> magic;               // set by rank 0 when the segment is initialised
    uint32_t nranks;
    std::atomic<uint32_t> attached;            // ranks that have mapped the segment
    std::atomic<uint32_t> bar_count, bar_sense;
    std::atomic<uint32_t> error;               // a GPU-side poll timed out (rank + 1 of the first one that did)
    uint32_t pad0[10];
    int64_t word[kMaxRanks];                   // host all-gather / all-reduce scratch
    double dword[kMaxRanks];
    hipIpcMemHandle_t handle[kMaxRanks][kMaxBufs];
    uint64_t bytes[kMaxRanks][kMaxBufs];
    // flag words, written by GPUs with system-scope stores: arrived[dst][src][buf] = sequence number of the last band of src that is complete
    // in dst's buffer; released[rank][buf] = sequence number up to which rank has consumed (and is about to overwrite) its buffer
    uint32_t arrived[kMaxRanks][kMaxRanks][kMaxBufs];
    uint32_t released[kMaxRanks][kMaxBufs];
    // written by the HOST of a rank when it enqueues the work that will raise the flags above: the sequence number of the last exchange it has
    // submitted on buffer b / the last release it has submitted.  A GPU-side poll starts its (short) deadline only once the peer's host is there.
    uint32_t submitted[kMaxRanks][kMaxBufs];
    uint32_t release_submitted[kMaxRanks][kMaxBufs];
};

struct Direct {
    int rank = 0, nranks = 0;
    char name[96] = {0};
    Segment *seg = nullptr;                    // host mapping
    Segment *dseg = nullptr;                   // the same memory as the device sees it
    bool registered_host = false;
    uint32_t local_sense = 0;
    int nbuf = 0;
    void *base[kMaxBufs] = {nullptr};          // this rank's buffers
    size_t bytes[kMaxBufs] = {0};
    void *peer[kMaxBufs][kMaxRanks] = {{nullptr}};   // every rank's buffer b as mapped here (own entry = base)
    uint32_t seq[kMaxBufs] = {0};              // exchanges started on buffer b (same on every rank: the calls are collective)
    uint32_t want[4][kMaxBufs] = {{0}};        // slot -> sequence number to wait for per buffer (0 = buffer not part of the slot)
    double timeout_s = kDirectTimeoutS;        // GPU-side polls give up this long after the peer's host submitted its side (SLGC_DIRECT_TIMEOUT_S, tests)
    double start_timeout_s = kDirectStartTimeoutS;   // ... or this long after they began when the peer's host never does (SLGC_DIRECT_START_TIMEOUT_S)
    double host_timeout_s = kHostTimeoutS;
    hipStream_t stream = nullptr;              // exchange stream
    hipEvent_t ev_compute = nullptr, ev_done[4] = {nullptr, nullptr, nullptr, nullptr};
};

Direct *state(slgc_ctx *ctx) { return (Direct *)ctx->direct; }

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int check_error(slgc_ctx *ctx)
{
    const uint32_t e = state(ctx)->seg->error.load(std::memory_order_acquire);
    if (e)
        return slgc_fail(ctx, SLGC_ECOMM, "direct exchange: a GPU-side wait on rank %u timed out (%.1f s after the peer submitted its side, or %.1f s without "
                         "it doing so: a peer is gone or stalled); the exchange is dead, results since then are incomplete", e - 1, state(ctx)->timeout_s, state(ctx)->start_timeout_s);
    return SLGC_OK;
}

int need_direct(slgc_ctx *ctx)
{
    if (!ctx) return SLGC_EINVAL;
    if (!ctx->direct) return slgc_fail(ctx, SLGC_ECOMM, "direct exchange not initialised (call slgc_direct_init)");
    return check_error(ctx);
}

// Host barrier over the segment (sense reversing).  Only around set-up / tear-down and the small host collectives: never per scan.
int host_barrier(slgc_ctx *ctx, Direct *d)
{
    Segment *s = d->seg;
    const uint32_t sense = d->local_sense ^= 1u;
    if (s->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)d->nranks) {
        s->bar_count.store(0, std::memory_order_relaxed);
        s->bar_sense.store(sense, std::memory_order_release);
        return SLGC_OK;
    }
    const double t0 = now_s();
    while (s->bar_sense.load(std::memory_order_acquire) != sense) {
        if (now_s() - t0 > d->host_timeout_s) return slgc_fail(ctx, SLGC_ECOMM, "direct exchange: host barrier timed out after %.0f s", d->host_timeout_s);
        usleep(50);
    }
    return SLGC_OK;
}

__device__ __forceinline__ uint32_t sys_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }      // polls: no cache action per load
__device__ __forceinline__ void sys_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

struct FlagList {
    uint32_t *p[kMaxRanks * 3];                // flag words (device view of the segment)
    uint32_t v[kMaxRanks * 3];                 // value to wait for (>=) / to store
    const uint32_t *begun[kMaxRanks * 3];      // waits only: the word in which the peer's HOST announces the work that will raise flag i ...
    uint32_t begun_v[kMaxRanks * 3];           // ... and the value it announces it with
    int n;
    uint32_t *error;
    uint32_t who;                              // rank + 1
    unsigned long long timeout_ticks, start_timeout_ticks;
};

// One wave: lane i polls flag i until it has reached its value (sequence numbers only grow; the comparison is wrap-safe).  The short deadline
// runs from the moment the peer's host has submitted its side; until then only the long one does.
__global__ void __launch_bounds__(64) k_flags_wait(const FlagList f)
{
    const int i = threadIdx.x;
    if (i < f.n) {
        const unsigned long long t_first = wall_clock64();
        unsigned long long t0 = t_first;
        bool begun = false;
        while ((int32_t)(sys_load(f.p[i]) - f.v[i]) < 0) {
            __builtin_amdgcn_s_sleep(32);
            if (sys_load(f.error)) break;           // some wait of the job has already given up: the polls queued behind it must not each sit out their own deadline
            const unsigned long long now = wall_clock64();
            if (!begun && (int32_t)(sys_load(f.begun[i]) - f.begun_v[i]) >= 0) {
                begun = true;
                t0 = now;
            }
            if (begun ? now - t0 > f.timeout_ticks : now - t_first > f.start_timeout_ticks) {
                sys_store(f.error, f.who);           // (a plain store: atomics on host memory would need PCIe atomics)
                break;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);         // one system-scope acquire for the whole wait (the flags say the bands are in)
}

__global__ void __launch_bounds__(64) k_flags_set(const FlagList f)
{
    const int i = threadIdx.x;
    if (i >= f.n) return;
    if (sys_load(f.error)) return;                   // the exchange has timed out somewhere: announce nothing (peers must not take half-filled buffers for complete ones)
    __threadfence_system();
    sys_store(f.p[i], f.v[i]);
}

struct PushArgs {
    const uint8_t *src[3];                     // the band in this rank's buffers
    uint8_t *dst[3][kMaxRanks];                // the same place in every peer's buffer (own rank: nullptr)
    uint64_t bytes[3];                         // band bytes per buffer
    int nbuf, npeers;
    int peer_rank[kMaxRanks];
    uint32_t chunks;                           // workgroups per (buffer, peer)
    const uint32_t *error;                     // the job's error word: set = the gate before this kernel (or any wait of the job) timed out
};

// grid = nbuf * npeers * chunks workgroups of 256 lanes; each moves a contiguous slice of one band to one peer, 16 bytes per lane and step.
// Loads are ordinary (the band was just written by this GPU: L2 / Infinity Cache hits).  Stores carry sc0 sc1 = system scope, write-through:
// a peer's memory is mapped non-coherently cacheable (MTYPE_NC) on this GPU, so an ordinary store may sit in this GPU's L2 until some later
// release -- written through, the bytes are at the peer when the store is acknowledged, and the wave waits for its acknowledgements before
// it ends.  The flag kernel that follows on the stream then publishes data that is already there (no per-wave L2 write-back fence).
typedef unsigned dv4u __attribute__((ext_vector_type(4)));
constexpr int kSysWriteThrough = 17;           // cache-policy bits of the raw buffer intrinsics on gfx942 / gfx950: bit 0 = sc0, bit 4 = sc1

__global__ void __launch_bounds__(256) k_push_bands(const PushArgs a)
{
    if (sys_load(a.error)) return;              // the gate timed out: the peers have NOT released their buffers -- writing now would clobber what a slow peer still reads
    uint32_t id = blockIdx.x;
    const uint32_t chunk = id % a.chunks;
    id /= a.chunks;
    const int pi = (int)(id % (uint32_t)a.npeers), b = (int)(id / (uint32_t)a.npeers);
    const uint8_t *src = a.src[b];
    uint8_t *dst = a.dst[b][a.peer_rank[pi]];
    const uint64_t n = a.bytes[b];                 // < 2^32 (checked by the launcher)
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, (uint32_t)n, 0x00020000);
    // slices in whole 16-byte units; the bytes before the first aligned unit and after the last one go with chunk 0 (bands of the sharded
    // scan are multiples of the row length, so src and dst share their alignment)
    const uint64_t head = (16u - ((uintptr_t)src & 15u)) & 15u;
    const uint64_t h = head < n ? head : n;
    const uint64_t units = (n - h) / 16u, tail = (n - h) % 16u;
    if (chunk == 0 && threadIdx.x < 32) {
        for (uint64_t i = threadIdx.x; i < h; i += 32) __builtin_amdgcn_raw_buffer_store_b8(src[i], rd, (uint32_t)i, 0, kSysWriteThrough);
        for (uint64_t i = threadIdx.x; i < tail; i += 32)
            __builtin_amdgcn_raw_buffer_store_b8(src[h + units * 16u + i], rd, (uint32_t)(h + units * 16u + i), 0, kSysWriteThrough);
    }
    const uint64_t per = (units + a.chunks - 1) / a.chunks;
    const uint64_t u0 = (uint64_t)chunk * per, u1 = u0 + per < units ? u0 + per : units;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src + h);
    if (((uintptr_t)(dst + h) & 15u) == 0) {
        for (uint64_t u = u0 + threadIdx.x; u < u1; u += 256) {
            const uint4 q = s4[u];
            __builtin_amdgcn_raw_buffer_store_b128(dv4u{q.x, q.y, q.z, q.w}, rd, (uint32_t)(h + u * 16u), 0, kSysWriteThrough);
        }
    } else {                                    // differently aligned destination (not produced by the sharded scan): bytes
        for (uint64_t u = u0 + threadIdx.x; u < u1; u += 256)
            for (uint32_t k = 0; k < 16; ++k) __builtin_amdgcn_raw_buffer_store_b8(src[h + u * 16u + k], rd, (uint32_t)(h + u * 16u + k), 0, kSysWriteThrough);
    }
    __builtin_amdgcn_s_waitcnt(0);              // every store of this wave has been acknowledged by the peer's memory before the wave ends
}

int find_buf(Direct *d, const void *base)
{
    for (int b = 0; b < d->nbuf; ++b)
        if (d->base[b] == base) return b;
    return -1;
}

int after_compute(slgc_ctx *ctx, Direct *d)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(d->ev_compute, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(d->stream, d->ev_compute, 0));
    return SLGC_OK;
}

void fill_common(FlagList &f, Direct *d)
{
    f.error = reinterpret_cast<uint32_t *>(&d->dseg->error);
    f.who = (uint32_t)d->rank + 1u;
    f.timeout_ticks = (unsigned long long)(d->timeout_s * 1e8);
    f.start_timeout_ticks = (unsigned long long)(d->start_timeout_s * 1e8);
}

}  // namespace

extern "C" int slgc_direct_init(slgc_ctx *ctx, int rank, int nranks, const char *key)
{
    if (!ctx || !key || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return slgc_fail(ctx, SLGC_EINVAL, "bad rank / nranks (at most %d) / key", kMaxRanks);
    if (ctx->direct) return slgc_fail(ctx, SLGC_ESTATE, "direct exchange already initialised");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Direct *d = new (std::nothrow) Direct();
    if (!d) return SLGC_ENOMEM;
    d->rank = rank;
    d->nranks = nranks;
    auto env_s = [](const char *name, double lo, double hi, double *out) {
        if (const char *t = getenv(name)) {
            const double v = atof(t);
            if (v >= lo && v <= hi) *out = v;
        }
    };
    env_s("SLGC_DIRECT_TIMEOUT_S", 0.05, 600.0, &d->timeout_s);
    d->start_timeout_s = d->timeout_s > kDirectStartTimeoutS ? d->timeout_s : kDirectStartTimeoutS;
    env_s("SLGC_DIRECT_START_TIMEOUT_S", 0.05, 3600.0, &d->start_timeout_s);
    env_s("SLGC_DIRECT_HOST_TIMEOUT_S", 1.0, 3600.0, &d->host_timeout_s);
    snprintf(d->name, sizeof d->name, "/slgc_direct_%.64s", key);
    for (char *c = d->name + 1; *c; ++c)
        if (*c == '/') *c = '_';
    const size_t seg_bytes = (sizeof(Segment) + 4095u) & ~(size_t)4095u;
    const double t0 = now_s();
    // Rank 0 creates the segment under the job's name (unlinking whatever an earlier job with the same key left behind) and takes the name away
    // again once every rank holds its mapping.  The other ranks open the name -- and may catch such a leftover before rank 0 has replaced it:
    // its magic and rank count look right, its rank 0 is dead.  So they keep comparing the inode the name points to with the one they mapped
    // while they wait for the job to assemble, and start over when the name has moved on to a new segment.
    void *m = MAP_FAILED;
    ino_t mapped_ino = 0;
    dev_t mapped_dev = 0;
    auto fail = [&](int status, const char *what) {
        if (m != MAP_FAILED) munmap(m, seg_bytes);
        if (rank == 0) shm_unlink(d->name);
        slgc_fail(ctx, status, "direct exchange: %s (%s, rank %d)", what, d->name, rank);
        delete d;
        return status;
    };
    auto name_moved_on = [&]() {                       // the name exists and is no longer the segment this rank mapped
        struct stat st;
        const int fd2 = shm_open(d->name, O_RDWR, 0600);
        if (fd2 < 0) return false;
        const bool other = fstat(fd2, &st) == 0 && (st.st_ino != mapped_ino || st.st_dev != mapped_dev);
        close(fd2);
        return other;
    };
    for (;;) {                                         // (ranks other than 0 may go round more than once)
        int fd = -1;
        if (rank == 0) {
            shm_unlink(d->name);                       // a stale segment of a crashed job with the same key
            fd = shm_open(d->name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd >= 0 && ftruncate(fd, (off_t)seg_bytes) != 0) {
                close(fd);
                fd = -1;
            }
        } else {
            while (fd < 0 && now_s() - t0 < d->host_timeout_s) {
                fd = shm_open(d->name, O_RDWR, 0600);
                struct stat st;
                if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < seg_bytes)) {      // rank 0 has not sized it yet
                    close(fd);
                    fd = -1;
                }
                if (fd < 0) usleep(200);
            }
        }
        if (fd < 0) return fail(SLGC_ECOMM, "shm_open failed");
        struct stat st;
        if (fstat(fd, &st) != 0) {
            close(fd);
            return fail(SLGC_ECOMM, "fstat of the shared segment failed");
        }
        mapped_ino = st.st_ino;
        mapped_dev = st.st_dev;
        m = mmap(nullptr, seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return fail(SLGC_ECOMM, "mmap of the shared segment failed");
        d->seg = (Segment *)m;
        if (rank == 0) {
            memset(m, 0, seg_bytes);
            d->seg->nranks = (uint32_t)nranks;
            d->seg->magic.store(kMagic, std::memory_order_release);
        } else {
            bool stale = false;
            while (d->seg->magic.load(std::memory_order_acquire) != kMagic && !(stale = name_moved_on())) {
                if (now_s() - t0 > d->host_timeout_s) return fail(SLGC_ECOMM, "rank 0 never initialised the shared segment");
                usleep(100);
            }
            if (!stale && d->seg->nranks != (uint32_t)nranks) {
                if (!name_moved_on()) return fail(SLGC_EINVAL, "ranks disagree on nranks");
                stale = true;
            }
            if (stale) {
                munmap(m, seg_bytes);
                m = MAP_FAILED;
                continue;
            }
        }
        // the job assembles: every rank counts itself in; rank 0 waits for all of them, the others for rank 0's go -- or for the name to move on
        d->seg->attached.fetch_add(1, std::memory_order_acq_rel);
        bool stale = false;
        if (rank == 0) {
            while (d->seg->attached.load(std::memory_order_acquire) < (uint32_t)nranks) {
                if (now_s() - t0 > d->host_timeout_s) return fail(SLGC_ECOMM, "not every rank attached to the shared segment");
                usleep(100);
            }
            shm_unlink(d->name);                       // every rank holds its mapping: the name can go (nothing is left behind if the job dies)
            d->seg->bar_sense.store(1u, std::memory_order_release);          // go (the host barrier's sense starts from here)
        } else {
            double last_check = now_s();
            while (d->seg->bar_sense.load(std::memory_order_acquire) != 1u) {
                const double t = now_s();
                if (t - t0 > d->host_timeout_s) return fail(SLGC_ECOMM, "rank 0 never released the job");
                if (t - last_check > 0.05) {
                    last_check = t;
                    if ((stale = name_moved_on())) break;
                }
                usleep(100);
            }
        }
        if (!stale) break;
        munmap(m, seg_bytes);
        m = MAP_FAILED;
    }
    d->local_sense = 1u;
    ctx->direct = d;
    hipError_t e = hipHostRegister(m, seg_bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e == hipSuccess) {
        d->registered_host = true;
        e = hipHostGetDevicePointer((void **)&d->dseg, m, 0);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_compute, hipEventDisableTiming);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&d->ev_done[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        slgc_fail(ctx, SLGC_EHIP, "direct exchange set-up: %s", hipGetErrorString(e));
        slgc_direct_destroy(ctx);
        return SLGC_EHIP;
    }
    return host_barrier(ctx, d);                       // every rank has its device view of the segment
}

extern "C" int slgc_direct_destroy(slgc_ctx *ctx)
{
    if (!ctx) return SLGC_EINVAL;
    Direct *d = state(ctx);
    if (!d) return SLGC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    for (int b = 0; b < d->nbuf; ++b)
        for (int r = 0; r < d->nranks; ++r)
            if (r != d->rank && d->peer[b][r]) (void)hipIpcCloseMemHandle(d->peer[b][r]);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    if (d->ev_compute) (void)hipEventDestroy(d->ev_compute);
    for (int i = 0; i < 4; ++i)
        if (d->ev_done[i]) (void)hipEventDestroy(d->ev_done[i]);
    if (d->seg) {
        if (d->registered_host) (void)hipHostUnregister(d->seg);
        munmap(d->seg, (sizeof(Segment) + 4095u) & ~(size_t)4095u);
    }
    delete d;
    ctx->direct = nullptr;
    return SLGC_OK;
}

// Collective, same order on every rank: the slot a buffer takes (the first free one) is then the same everywhere.
extern "C" int slgc_direct_register(slgc_ctx *ctx, void *d_base, size_t bytes)
{
    int rc = need_direct(ctx);
    if (rc) return rc;
    Direct *d = state(ctx);
    if (!d_base || !bytes) return slgc_fail(ctx, SLGC_EINVAL, "null buffer");
    if (find_buf(d, d_base) >= 0) return slgc_fail(ctx, SLGC_EINVAL, "buffer already registered");
    int b = 0;
    while (b < d->nbuf && d->base[b]) ++b;                            // a slot freed by slgc_direct_unregister, or a new one
    if (b >= kMaxBufs) return slgc_fail(ctx, SLGC_EINVAL, "at most %d buffers can be registered at a time (slgc_direct_unregister frees a slot)", kMaxBufs);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipIpcMemHandle_t mine;
    HIP_TRY(ctx, hipIpcGetMemHandle(&mine, d_base));              // d_base must be the start of a slgc_dev_alloc allocation
    d->seg->handle[d->rank][b] = mine;
    d->seg->bytes[d->rank][b] = bytes;
    if ((rc = host_barrier(ctx, d))) return rc;                     // every rank's handle of buffer b is in the segment
    int bad = 0;
    void *opened[kMaxRanks] = {nullptr};
    for (int r = 0; r < d->nranks && !bad; ++r) {
        if (r == d->rank) continue;
        if (d->seg->bytes[r][b] != bytes) {
            bad = 1;
            break;
        }
        const hipError_t e = hipIpcOpenMemHandle(&opened[r], d->seg->handle[r][b], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            slgc_fail(ctx, SLGC_EHIP, "hipIpcOpenMemHandle (rank %d's buffer %d): %s", r, b, hipGetErrorString(e));
            opened[r] = nullptr;
            bad = 2;
        }
    }
    if (!bad) {                                                      // nothing of this rank's state has changed before this point
        for (int r = 0; r < d->nranks; ++r) d->peer[b][r] = r == d->rank ? d_base : opened[r];
        d->base[b] = d_base;
        d->bytes[b] = bytes;
        d->seq[b] = 0;
        if (b == d->nbuf) d->nbuf = b + 1;
    } else {
        for (int r = 0; r < d->nranks; ++r)
            if (opened[r]) (void)hipIpcCloseMemHandle(opened[r]);
    }
    const int rc2 = host_barrier(ctx, d);                           // nobody re-uses the handle slots before everyone has opened them
    if (bad == 1) return slgc_fail(ctx, SLGC_EINVAL, "direct exchange: ranks registered buffers of different sizes");
    if (bad == 2) return SLGC_EHIP;
    return rc2;
}

// Collective: the buffer leaves the exchange on every rank (peer mappings closed, its flag words back to zero, its slot free for the next
// slgc_direct_register).  Both streams are drained first.  A registered buffer must be unregistered before it is freed: slgc_dev_free refuses it.
extern "C" int slgc_direct_unregister(slgc_ctx *ctx, void *d_base)
{
    int rc = need_direct(ctx);
    if (rc) return rc;
    Direct *d = state(ctx);
    const int b = find_buf(d, d_base);
    if (!d_base || b < 0) return slgc_fail(ctx, SLGC_EINVAL, "buffer is not registered");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(d->stream));
    if ((rc = host_barrier(ctx, d))) return rc;                     // no rank still pushes into (or polls for) this buffer
    for (int r = 0; r < d->nranks; ++r) {
        if (r != d->rank && d->peer[b][r]) (void)hipIpcCloseMemHandle(d->peer[b][r]);
        d->peer[b][r] = nullptr;
        d->seg->arrived[d->rank][r][b] = 0;
    }
    d->seg->released[d->rank][b] = 0;
    d->seg->submitted[d->rank][b] = 0;
    d->seg->release_submitted[d->rank][b] = 0;
    d->seg->bytes[d->rank][b] = 0;
    d->base[b] = nullptr;
    d->bytes[b] = 0;
    d->seq[b] = 0;
    for (int sl = 0; sl < 4; ++sl) d->want[sl][b] = 0;
    while (d->nbuf > 0 && !d->base[d->nbuf - 1]) --d->nbuf;
    return host_barrier(ctx, d);                                     // the slot's words are clean on every rank before anyone registers into it
}

// 1 if d_base is currently registered with the direct exchange of this context (slgc_dev_free asks)
int slgc_direct_is_registered(slgc_ctx *ctx, const void *d_base)
{
    return ctx && ctx->direct && d_base && find_buf(state(ctx), d_base) >= 0;
}

// The error word after the streams have drained: slgc_synchronize / slgc_d2h end with this, so a scan whose exchange timed out on the GPU
// comes back as SLGC_ECOMM, never as (half-filled) data.
int slgc_direct_check(slgc_ctx *ctx)
{
    if (!ctx || !ctx->direct) return SLGC_OK;
    return check_error(ctx);
}

extern "C" int slgc_direct_allgatherv_begin(slgc_ctx *ctx, int nbuf, void *const *d_bases, const int64_t *const *counts, const int64_t *const *displs, int slot)
{
    int rc = need_direct(ctx);
    if (rc) return rc;
    Direct *d = state(ctx);
    if (nbuf < 1 || nbuf > 3 || !d_bases || !counts || !displs || slot < 0 || slot > 3) return slgc_fail(ctx, SLGC_EINVAL, "1..3 buffers, slot 0..3");
    int ids[3];
    PushArgs pa{};
    FlagList gate{}, done{};
    fill_common(gate, d);
    fill_common(done, d);
    for (int i = 0; i < nbuf; ++i) {                                  // (everything is checked before any sequence number moves)
        ids[i] = find_buf(d, d_bases[i]);
        if (ids[i] < 0) return slgc_fail(ctx, SLGC_EINVAL, "buffer %d is not registered (slgc_direct_register)", i);
        if (!counts[i] || !displs[i]) return slgc_fail(ctx, SLGC_EINVAL, "null layout");
        const int64_t c = counts[i][d->rank], o = displs[i][d->rank];
        if (c < 0 || o < 0 || (uint64_t)(o + c) > d->bytes[ids[i]]) return slgc_fail(ctx, SLGC_EINVAL, "band outside the registered buffer");
        if ((uint64_t)c >= 0xfffffff0ull) return slgc_fail(ctx, SLGC_EINVAL, "a band of 4 GB or more");
        for (int j = 0; j < i; ++j)
            if (ids[j] == ids[i]) return slgc_fail(ctx, SLGC_EINVAL, "the same buffer twice in one exchange");
    }
    for (int i = 0; i < kMaxBufs; ++i) d->want[slot][i] = 0;
    for (int i = 0; i < nbuf; ++i) {
        const int64_t c = counts[i][d->rank], o = displs[i][d->rank];
        const uint32_t seq = ++d->seq[ids[i]];
        d->want[slot][ids[i]] = seq;
        pa.src[i] = (const uint8_t *)d->base[ids[i]] + o;
        pa.bytes[i] = (uint64_t)c;
        for (int r = 0; r < d->nranks; ++r) {
            if (r == d->rank) continue;
            pa.dst[i][r] = (uint8_t *)d->peer[ids[i]][r] + o;
            gate.p[gate.n] = &d->dseg->released[r][ids[i]];        // the peer has let go of what exchange seq - 1 left in its buffer
            gate.begun[gate.n] = &d->dseg->release_submitted[r][ids[i]];
            gate.begun_v[gate.n] = seq - 1u;
            gate.v[gate.n++] = seq - 1u;
            done.p[done.n] = &d->dseg->arrived[r][d->rank][ids[i]];
            done.v[done.n++] = seq;
        }
    }
    pa.nbuf = nbuf;
    pa.error = reinterpret_cast<const uint32_t *>(&d->dseg->error);
    for (int r = 0; r < d->nranks; ++r)
        if (r != d->rank) pa.peer_rank[pa.npeers++] = r;
    if ((rc = after_compute(ctx, d))) return rc;
    for (int i = 0; i < nbuf; ++i)                                    // this rank's host is here: peers polling for these bands start their short deadline
        __atomic_store_n(&d->seg->submitted[d->rank][ids[i]], d->seq[ids[i]], __ATOMIC_RELEASE);
    if (pa.npeers > 0) {
        uint64_t most = 0;
        for (int i = 0; i < nbuf; ++i) most = pa.bytes[i] > most ? pa.bytes[i] : most;
        // enough workgroups per link to keep it busy (64 KB slices, at most 64 per band and peer), never an empty grid
        uint32_t chunks = (uint32_t)((most + 65535u) / 65536u);
        pa.chunks = chunks < 1 ? 1 : chunks > 64 ? 64 : chunks;
        hipLaunchKernelGGL(k_flags_wait, dim3(1), dim3(64), 0, d->stream, gate);
        hipLaunchKernelGGL(k_push_bands, dim3((unsigned)(nbuf * pa.npeers) * pa.chunks), dim3(256), 0, d->stream, pa);
        hipLaunchKernelGGL(k_flags_set, dim3(1), dim3(64), 0, d->stream, done);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(d->ev_done[slot], d->stream));
    return SLGC_OK;
}

extern "C" int slgc_direct_wait(slgc_ctx *ctx, int slot)
{
    int rc = need_direct(ctx);
    if (rc) return rc;
    Direct *d = state(ctx);
    if (slot < 0 || slot > 3) return slgc_fail(ctx, SLGC_EINVAL, "slot outside 0..3");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, d->ev_done[slot], 0));     // this rank's own pushes are out (its band may be overwritten from here on)
    FlagList f{};
    fill_common(f, d);
    for (int b = 0; b < d->nbuf; ++b) {
        if (!d->want[slot][b]) continue;
        for (int r = 0; r < d->nranks; ++r) {
            if (r == d->rank) continue;
            f.p[f.n] = &d->dseg->arrived[d->rank][r][b];
            f.begun[f.n] = &d->dseg->submitted[r][b];
            f.begun_v[f.n] = d->want[slot][b];
            f.v[f.n++] = d->want[slot][b];
        }
    }
    if (f.n > 0) {
        hipLaunchKernelGGL(k_flags_wait, dim3(1), dim3(64), 0, ctx->stream, f);
        HIP_TRY(ctx, hipGetLastError());
    }
    return SLGC_OK;
}

// On the compute stream: "everything enqueued so far has finished with what the last exchange left in these buffers" -- peers may push the
// next bands into them.  The sharded scanner calls it right before it re-uses a buffer set itself.
extern "C" int slgc_direct_release(slgc_ctx *ctx, int nbuf, void *const *d_bases)
{
    int rc = need_direct(ctx);
    if (rc) return rc;
    Direct *d = state(ctx);
    if (nbuf < 1 || nbuf > 3 || !d_bases) return slgc_fail(ctx, SLGC_EINVAL, "1..3 buffers");
    FlagList f{};
    fill_common(f, d);
    for (int i = 0; i < nbuf; ++i) {
        const int b = find_buf(d, d_bases[i]);
        if (b < 0) return slgc_fail(ctx, SLGC_EINVAL, "buffer %d is not registered", i);
        f.p[f.n] = &d->dseg->released[d->rank][b];
        f.v[f.n++] = d->seq[b];
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_flags_set, dim3(1), dim3(64), 0, ctx->stream, f);
    HIP_TRY(ctx, hipGetLastError());
    for (int i = 0; i < nbuf; ++i) {                                  // this rank's host has submitted the release: peers gated on it start their short deadline
        const int b = find_buf(d, d_bases[i]);
        __atomic_store_n(&d->seg->release_submitted[d->rank][b], d->seq[b], __ATOMIC_RELEASE);
    }
    return SLGC_OK;
}

// Small host collectives over the segment (set-up, verification, timing): both streams are drained first.
extern "C" int slgc_direct_barrier(slgc_ctx *ctx)
{
    int rc = need_direct(ctx);
    if (rc) return rc;
    Direct *d = state(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(d->stream));
    if ((rc = need_direct(ctx))) return rc;                          // a GPU-side timeout shows here
    return host_barrier(ctx, d);
}

extern "C" int slgc_direct_allgather_i64(slgc_ctx *ctx, int64_t mine, int64_t *all)
{
    int rc = need_direct(ctx);
    if (rc) return rc;
    Direct *d = state(ctx);
    if (!all) return slgc_fail(ctx, SLGC_EINVAL, "null output");
    d->seg->word[d->rank] = mine;
    if ((rc = host_barrier(ctx, d))) return rc;
    for (int r = 0; r < d->nranks; ++r) all[r] = d->seg->word[r];
    return host_barrier(ctx, d);
}
