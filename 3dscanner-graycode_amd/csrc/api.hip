// api.hip -- the C-ABI of libslgc.so (declared in include/slgc.h): context, device memory, and the host-buffer
// entry points that mirror the reference's Python functions.  No CPU fallback anywhere: every entry point runs HIP
// kernels on the context's device or returns an error.
#include <sys/mman.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <new>
#include <thread>
#include <vector>

#include "host_util.h"
#include "slgc_internal.h"
#include "tri_math.h"

// ------------------------------------------------------------------------------------------ helpers
int slgc_fail(slgc_ctx *ctx, int status, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return status;
}

int slgc_ws(slgc_ctx *ctx, int slot, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 16;
    ++ctx->ws_gen[slot];
    if (ctx->ws_bytes[slot] < bytes) {
        if (ctx->ws[slot]) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipFree(ctx->ws[slot]));
            ctx->ws[slot] = nullptr;
            ctx->ws_bytes[slot] = 0;
        }
        const size_t want = bytes + bytes / 8;
        if (hipMalloc(&ctx->ws[slot], want) != hipSuccess) return slgc_fail(ctx, SLGC_ENOMEM, "hipMalloc(%zu) failed", want);
        ctx->ws_bytes[slot] = want;
    }
    *out = ctx->ws[slot];
    return SLGC_OK;
}

// Frame index tables, exactly as the reference computes them (decode_codes.py:109-111: float pattern_len, uint8
// truncation; :149 int pattern_len).  N = 44 -> hid = 19,17,15,40,38,36 (+2).
int slgc_make_geom(int N, int n_runs, DecodeGeom *g)
{
    if (N < 14 || N > 65 || n_runs < 1 || n_runs > SLGC_MAX_RUNS) return SLGC_EINVAL;
    const double plf = (double)(N - 2) / 4.0;
    const double h[6] = {2 * plf - 2, 2 * plf - 4, 2 * plf - 6, 4 * plf - 2, 4 * plf - 4, 4 * plf - 6};
    const double v[6] = {1, 3, 5, 2 * plf + 1, 2 * plf + 3, 2 * plf + 5};
    for (int k = 0; k < 6; ++k) {
        g->hid[k] = 2 + (int)(uint8_t)h[k];
        g->vid[k] = 2 + (int)(uint8_t)v[k];
        if (g->hid[k] >= N || g->vid[k] >= N) return SLGC_EINVAL;
    }
    g->N = N;
    g->L = (int)((double)(N - 2) / 4.0);
    g->n_runs = n_runs;
    return SLGC_OK;
}

namespace {

int check_ctx(slgc_ctx *ctx)
{
    if (!ctx) return SLGC_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return slgc_fail(ctx, SLGC_EHIP, "hipSetDevice(%d) failed", ctx->device);
    return SLGC_OK;
}

size_t esize(int dtype) { return dtype == SLGC_F64 ? 8 : 1; }

// narrow_f64_to_u8 and the staging-ring copy loop live in host_util.cpp (host only: unit-tested on the CPU under ASan / UBSan / TSan).
int host_staging(slgc_ctx *ctx, size_t bytes, void **out)
{
    if (ctx->stage_bytes < bytes) {
        if (ctx->stage) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipHostFree(ctx->stage);
            ctx->stage = nullptr;
            ctx->stage_bytes = 0;
        }
        if (hipHostMalloc(&ctx->stage, bytes + bytes / 8, hipHostMallocDefault) != hipSuccess)
            return slgc_fail(ctx, SLGC_ENOMEM, "hipHostMalloc(%zu) failed", bytes);
        ctx->stage_bytes = bytes + bytes / 8;
    }
    *out = ctx->stage;
    return SLGC_OK;
}

// ---- large results into memory nobody has touched yet ----
// The reference-shaped entry points return NumPy arrays.  A hipMemcpy into memory nobody has touched yet runs at the page-fault rate of one
// thread (measured on the MI355X host: 197 MB of int64 maps in 17.0 ms = 11.6 GB/s, the same as np.empty + fill; the link does 56 GB/s);
// into pages that exist already -- page-locked or not -- it runs at the rate of the link (tools/ubench/host_alloc.hip), which is why
// scanner/_native.py recycles its result buffers.  For the untouched case, results of 8 MB and more land in a pinned ring of 4 x 16 MB
// and host threads copy every landed chunk into its slice of the destination -- the first touches spread over the threads, the next chunk
// already on the link.  Synchronous: returns with dst complete.
constexpr size_t kDlChunk = 16u << 20;
constexpr int kDlSlots = 4;

struct DlRing {          // RingOps of download_par: chunk k lands in its pinned slot with a hipMemcpyAsync followed by the slot's event
    slgc_ctx *ctx;
    const char *d_src;
    hipError_t err;
};

int dl_fetch(void *u, size_t k, void *slot, size_t offset, size_t nbytes)
{
    DlRing *r = (DlRing *)u;
    r->err = hipMemcpyAsync(slot, r->d_src + offset, nbytes, hipMemcpyDeviceToHost, r->ctx->stream);
    if (r->err == hipSuccess) r->err = hipEventRecord(r->ctx->dl_ev[k % kDlSlots], r->ctx->stream);
    return r->err == hipSuccess ? 0 : 1;
}

int dl_wait(void *u, size_t k)
{
    DlRing *r = (DlRing *)u;
    r->err = hipEventSynchronize(r->ctx->dl_ev[k % kDlSlots]);
    return r->err == hipSuccess ? 0 : 1;
}

int download_par(slgc_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (bytes == 0 || !dst) return SLGC_OK;
    static const int par = xcd_env("SLGC_PAR_DOWNLOAD", 1);              // 0: plain hipMemcpy (A/B)
    unsigned hw = std::thread::hardware_concurrency();
    const int nthr = (int)(hw ? (hw > 16 ? 16 : hw) : 4);
    bool pinned = false;                                                   // page-locked destination (the caller's own hipHostMalloc / hipHostRegister): the DMA engine writes it directly
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, dst) == hipSuccess) pinned = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();                                      // ordinary memory is "invalid value" to the runtime: not an error here
    }
    // Pages that exist already (a recycled result buffer, an array the caller has used before) take the copy at the rate of the link, page-
    // locked or not; only memory nobody has touched is worth the ring.  mincore() on every 16th page tells the two apart in microseconds.
    bool resident = false;
    if (!pinned && par && bytes >= (8u << 20)) {
        const uintptr_t lo = (uintptr_t)dst & ~(uintptr_t)4095, hi = ((uintptr_t)dst + bytes + 4095) & ~(uintptr_t)4095;
        const size_t pages = (hi - lo) >> 12;
        unsigned char *vec = new (std::nothrow) unsigned char[pages];
        if (vec && mincore((void *)lo, hi - lo, vec) == 0) {
            size_t seen = 0, in = 0;
            for (size_t i = 0; i < pages; i += 16, ++seen) in += vec[i] & 1u;
            resident = in * 10 >= seen * 9;
        }
        delete[] vec;
    }
    if (pinned || resident || !par || bytes < (8u << 20) || nthr < 2) {
        HIP_TRY(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return SLGC_OK;
    }
    if (!ctx->dl_stage) {                                                  // the ring exists only with all of its events
        void *stage = nullptr;
        hipEvent_t ev[kDlSlots] = {};
        bool ok = hipHostMalloc(&stage, kDlSlots * kDlChunk, hipHostMallocDefault) == hipSuccess;
        for (int i = 0; i < kDlSlots && ok; ++i) ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            for (int i = 0; i < kDlSlots; ++i)
                if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (stage) (void)hipHostFree(stage);
            (void)hipGetLastError();
            return slgc_fail(ctx, SLGC_ENOMEM, "pinned download ring (%zu bytes + %d events) could not be created", (size_t)kDlSlots * kDlChunk, kDlSlots);
        }
        for (int i = 0; i < kDlSlots; ++i) ctx->dl_ev[i] = ev[i];
        ctx->dl_stage = stage;
    }
    {
        const uintptr_t lo = ((uintptr_t)dst + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1), hi = ((uintptr_t)dst + bytes) & ~(uintptr_t)((2u << 20) - 1);
        if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);  // fewer, larger first-touch faults; a no-op where THP is off (the memory is the caller's: advice only)
    }
    DlRing ring{ctx, (const char *)d_src, hipSuccess};
    const slgc_host::RingOps ops{&ring, dl_fetch, dl_wait};
    const int rc = slgc_host::ring_download(dst, bytes, ctx->dl_stage, kDlChunk, kDlSlots, 8, nthr, ops, par == 2 ? 0 : 1);      // 8 parts = 2 MB slices: one huge page each
    if (rc < 0) return slgc_fail(ctx, SLGC_ENOMEM, "download: host threads / memory for the staging ring");
    if (rc > 0) return slgc_fail(ctx, SLGC_EHIP, "download: %s", hipGetErrorString(ring.err));
    return SLGC_OK;
}

// Upload n_runs host stacks into workspace slot 0 back to back.  *dtype is in/out: a float64 stack that narrows to uint8 (above)
// is uploaded as uint8 (1/8 of the bytes, from pinned memory) and *dtype becomes SLGC_U8 for the kernels that follow.
int upload_runs(slgc_ctx *ctx, const void *const *stacks, int *dtype, int n_runs, int N, size_t npix, RunPtrs *out)
{
    for (int r = 0; r < n_runs; ++r)
        if (!stacks[r]) return slgc_fail(ctx, SLGC_EINVAL, "stack %d is null", r);
    const size_t elems = (size_t)N * npix;
    ctx->stack_tile_active = 0;               // host stacks are planar [N][H][W], whatever layout the context's device-resident scans use
    ctx->last_input_path = *dtype == SLGC_U8 ? 0 : 2;
    static const int pack = xcd_env("SLGC_F64_PACK", 1);          // 0: always ship float64 (A/B of the narrowing)
    bool plausible = *dtype == SLGC_F64 && pack && elems;
    if (plausible) {              // a look at 16 pieces spread over every run before any page is pinned: a stack of fractions (a normalised capture) fails at
                                  // the first piece, one whose late frames are not grey levels at one of the later ones -- not after a full narrowing pass
        uint8_t probe[256];
        constexpr size_t kPieces = 16;
        const size_t n = elems < sizeof probe ? elems : sizeof probe;
        for (int r = 0; r < n_runs && plausible; ++r)
            for (size_t k = 0; k < kPieces && plausible; ++k) {
                const size_t at = elems > n ? (k + 1 == kPieces ? elems - n : (elems - n) / (kPieces - 1) * k) : 0;      // the last piece ends with the stack
                const void *one = (const double *)stacks[r] + at;
                plausible = slgc_host::narrow_f64_to_u8(&one, 1, n, probe, 1) == 1;
            }
    }
    if (plausible) {
        void *st;
        int rc = host_staging(ctx, elems * n_runs, &st);
        if (rc) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // the staging buffer may still feed the previous call's copy
        const int narrowed = slgc_host::narrow_f64_to_u8(stacks, n_runs, elems, (uint8_t *)st);
        if (narrowed < 0) return slgc_fail(ctx, SLGC_ENOMEM, "host threads for the float64 -> uint8 narrowing");
        if (!narrowed) {          // a sample between the probes is not a grey level: this stack takes the float64 kernel
            if (ctx->stage_bytes > ((size_t)256 << 20)) {      // do not sit on a large pinned buffer; a small one stays for the next well-formed stack
                (void)hipHostFree(ctx->stage);
                ctx->stage = nullptr;
                ctx->stage_bytes = 0;
            }
        }
        if (narrowed) {
            void *d;
            if ((rc = slgc_ws(ctx, 0, elems * n_runs, &d))) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(d, st, elems * n_runs, hipMemcpyHostToDevice, ctx->stream));
            for (int r = 0; r < n_runs; ++r) out->p[r] = (char *)d + (size_t)r * elems;
            *dtype = SLGC_U8;
            ctx->last_input_path = 1;
            return SLGC_OK;
        }
    }
    const size_t run_bytes = elems * esize(*dtype);
    void *d;
    int rc = slgc_ws(ctx, 0, run_bytes * n_runs, &d);
    if (rc) return rc;
    for (int r = 0; r < n_runs; ++r) {
        out->p[r] = (char *)d + (size_t)r * run_bytes;
        if (run_bytes) HIP_TRY(ctx, hipMemcpyAsync((void *)out->p[r], stacks[r], run_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    return SLGC_OK;
}

int check_dims(slgc_ctx *ctx, int dtype, int N, int H, int W)
{
    if (dtype != SLGC_U8 && dtype != SLGC_F64) return slgc_fail(ctx, SLGC_EINVAL, "dtype must be SLGC_U8 or SLGC_F64");
    if (N < 14 || N > 65) return slgc_fail(ctx, SLGC_EINVAL, "N=%d outside [14, 65] (reference needs N >= 14; L <= 15)", N);
    if (H < 0 || W < 0) return slgc_fail(ctx, SLGC_EINVAL, "negative image size");
    return SLGC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------ library / context
extern "C" int slgc_version(void) { return SLGC_VERSION; }
extern "C" const char *slgc_backend(void) { return "hip:gfx950"; }
extern "C" const char *slgc_strerror(int s)
{
    switch (s) {
    case SLGC_OK: return "ok";
    case SLGC_EINVAL: return "invalid argument";
    case SLGC_ENODEV: return "no HIP device";
    case SLGC_EHIP: return "HIP runtime error";
    case SLGC_ENOMEM: return "out of device memory";
    case SLGC_ECOMM: return "RCCL error";
    case SLGC_ESTATE: return "call order error";
    default: return "unknown status";
    }
}

extern "C" int slgc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int slgc_create(int device, slgc_ctx **out)
{
    if (!out) return SLGC_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SLGC_ENODEV;
    if (device < 0 || device >= n) return SLGC_EINVAL;
    slgc_ctx *ctx = new (std::nothrow) slgc_ctx();
    if (!ctx) return SLGC_ENOMEM;
    memset(ctx, 0, sizeof *ctx);
    ctx->device = device;
    ctx->tune_fuse_tail = xcd_env("SLGC_FUSE_TAIL", 1);
    ctx->tune_proj_tile = xcd_env("SLGC_PROJ_TILE", 1);
    ctx->tune_fuse_nt = xcd_env("SLGC_FUSE_NT", -1);           // -1 = by image size (launch_scan_fused)
    ctx->tune_tri_nt = xcd_env("SLGC_TRI_NT", 1);
    ctx->tune_xcd = xcd_env("SLGC_XCD", 1);
    ctx->tune_park = xcd_env("SLGC_PARK", 1);
    ctx->tune_guard_list = xcd_env("SLGC_GUARD_LIST", 1);
    ctx->tune_fuse_xcd = xcd_env("SLGC_FUSE_XCD", 0);
    ctx->tune_prio = xcd_env("SLGC_PRIO", -1);           // -1 = by launch shape (launch_scan_fused)
    ctx->tune_cam_nodes = xcd_env("SLGC_CAM_NODES", 1);
    ctx->lut_nodes_err = -1.0f;
    ctx->tune_lists_lines = xcd_env("SLGC_LISTS_LINES", 1);
    ctx->tune_lists_order = xcd_env("SLGC_LISTS_ORDER", 4);     // column-major in runs of 4 tiles per XCD (whole-lines scatter: 218.8 -> 210.5 us; the tile-run scatter reads it as 1)
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return SLGC_EHIP;
    }
    for (int i = 0; i < SLGC_MAX_EVENTS; ++i)
        if (hipEventCreate(&ctx->events[i]) != hipSuccess) {
            delete ctx;
            return SLGC_EHIP;
        }
    *out = ctx;
    return SLGC_OK;
}

extern "C" int slgc_destroy(slgc_ctx *ctx)
{
    if (!ctx) return SLGC_EINVAL;
    (void)hipSetDevice(ctx->device);
    slgc_direct_destroy(ctx);
    slgc_comm_destroy(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < SLGC_WS_SLOTS; ++i)
        if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
    if (ctx->lut_cam) (void)hipFree(ctx->lut_cam);
    if (ctx->lut_nodes) (void)hipFree(ctx->lut_nodes);
    if (ctx->lut_check_word) (void)hipFree(ctx->lut_check_word);
    if (ctx->dl_stage) {
        (void)hipHostFree(ctx->dl_stage);
        for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ctx->dl_ev[i]);
    }
    if (ctx->lut_proj) (void)hipFree(ctx->lut_proj);
    if (ctx->count_slots) (void)hipFree(ctx->count_slots);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    for (int i = 0; i < SLGC_MAX_EVENTS; ++i)
        if (ctx->events[i]) (void)hipEventDestroy(ctx->events[i]);
    for (int i = 0; i < 2 * ctx->prof_cap; ++i) (void)hipEventDestroy(ctx->prof_ev[i]);
    delete[] ctx->prof_ev;
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SLGC_OK;
}

extern "C" const char *slgc_last_error(slgc_ctx *ctx) { return ctx ? ctx->err : "null context"; }

// Tuning knobs: every value gives the same results; they exist so that two settings can be timed in ONE process (interleaved A/B).
extern "C" int slgc_tune(slgc_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name) return SLGC_EINVAL;
    if (!strcmp(name, "fuse_tail")) ctx->tune_fuse_tail = value != 0;
    else if (!strcmp(name, "proj_tile")) ctx->tune_proj_tile = value != 0;      // the projector table is rebuilt on the next use
    else if (!strcmp(name, "fuse_nt")) ctx->tune_fuse_nt = value < 0 ? -1 : (value & 3);
    else if (!strcmp(name, "tri_nt")) ctx->tune_tri_nt = value & 1;
    else if (!strcmp(name, "xcd")) ctx->tune_xcd = value != 0;
    else if (!strcmp(name, "park")) ctx->tune_park = value != 0;
    else if (!strcmp(name, "guard_list")) ctx->tune_guard_list = value != 0;
    else if (!strcmp(name, "fuse_xcd")) ctx->tune_fuse_xcd = value < 0 ? 0 : value;
    else if (!strcmp(name, "prio")) ctx->tune_prio = value < 0 ? -1 : value;
    else if (!strcmp(name, "cam_nodes")) ctx->tune_cam_nodes = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(name, "lists_lines")) ctx->tune_lists_lines = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(name, "lists_order")) ctx->tune_lists_order = value < 0 ? 0 : (value > 64 ? 64 : value);
    else if (!strcmp(name, "image_rows")) ctx->tune_image_rows = value < 0 ? 0 : value;      // the ray tables are rebuilt on the next use
    else if (!strcmp(name, "stack_tile_log2")) {
        if (value != 0 && (value < 8 || value > 24)) return slgc_fail(ctx, SLGC_EINVAL, "stack_tile_log2 must be 0 (planar) or 8..24");
        ctx->tune_stack_tile = value;
    }
    else if (!strcmp(name, "wire")) ctx->tune_wire = value != 0;      // NOT result-neutral in bytes moved, result-neutral in maps / XYZ
#ifdef SLGC_DIAG
    else if (!strcmp(name, "fuse_abl")) ctx->tune_fuse_abl = value;
#endif
    else return slgc_fail(ctx, SLGC_EINVAL, "unknown tuning knob '%s'", name);
    return SLGC_OK;
}

extern "C" int slgc_last_input_path(slgc_ctx *ctx) { return ctx ? ctx->last_input_path : SLGC_EINVAL; }

extern "C" int slgc_last_scan_path(slgc_ctx *ctx, int *ns_frames, int *node_table, int *guard)
{
    if (!ctx) return SLGC_EINVAL;
    if (ns_frames) *ns_frames = ctx->last_ns;
    if (node_table) *node_table = ctx->last_nodes;
    if (guard) *guard = ctx->last_guard;
    return ctx->last_scan_path;
}

extern "C" int slgc_last_list_kernel(slgc_ctx *ctx) { return ctx ? ctx->last_list_kernel : SLGC_EINVAL; }

extern "C" int slgc_last_scan_ragged(slgc_ctx *ctx)
{
    if (!ctx) return SLGC_EINVAL;
    return (ctx->last_ragged ? 1 : 0) | (ctx->last_tri_ragged ? 2 : 0);
}

extern "C" int slgc_synchronize(slgc_ctx *ctx)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return slgc_direct_check(ctx);                 // a direct exchange that timed out on the GPU: the work is "done", its results are not (SLGC_ECOMM)
}

extern "C" int slgc_device_name(slgc_ctx *ctx, char *buf, int buflen)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!buf || buflen < 2) return slgc_fail(ctx, SLGC_EINVAL, "bad buffer");
    hipDeviceProp_t p;
    HIP_TRY(ctx, hipGetDeviceProperties(&p, ctx->device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name[0] ? p.name : "AMD GPU", p.gcnArchName, p.multiProcessorCount);   // some boxes report an empty marketing name
    return SLGC_OK;
}

extern "C" int slgc_device_pci_bus_id(slgc_ctx *ctx, char *buf, int buflen)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!buf || buflen < 16) return slgc_fail(ctx, SLGC_EINVAL, "bad buffer");
    HIP_TRY(ctx, hipDeviceGetPCIBusId(buf, buflen, ctx->device));
    return SLGC_OK;
}

// ------------------------------------------------------------------------------------------ decode, host buffers
extern "C" int slgc_direct_indirect(slgc_ctx *ctx, const void *stack, int dtype, int N, int H, int W, double *L_d, double *L_g)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if ((rc = check_dims(ctx, dtype, N, H, W))) return rc;
    if (!stack || !L_d || !L_g) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    DecodeGeom g;
    if (slgc_make_geom(N, 1, &g)) return slgc_fail(ctx, SLGC_EINVAL, "unsupported N=%d", N);
    const size_t npix = (size_t)H * W;
    RunPtrs runs{};
    if ((rc = upload_runs(ctx, &stack, &dtype, 1, N, npix, &runs))) return rc;
    void *d_out;
    if ((rc = slgc_ws(ctx, 1, npix * 16, &d_out))) return rc;
    double *d_ld = (double *)d_out, *d_lg = d_ld + npix;
    rc = launch_decode_generic(ctx, g, runs, dtype, npix, npix, 0.0, nullptr, nullptr, d_ld, d_lg, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr);
    if (rc) return rc;
    if (npix) {
        if ((rc = download_par(ctx, L_d, d_ld, npix * 8))) return rc;
        if ((rc = download_par(ctx, L_g, d_lg, npix * 8))) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

static int codes_common(slgc_ctx *ctx, const void *stack, int dtype, int N, int H, int W, const double *L_d, const double *L_g,
                        double eps, int8_t *h_codes, int8_t *v_codes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if ((rc = check_dims(ctx, dtype, N, H, W))) return rc;
    if (!stack || !h_codes || !v_codes) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (std::isnan(eps)) return slgc_fail(ctx, SLGC_EINVAL, "eps is NaN");
    DecodeGeom g;
    if (slgc_make_geom(N, 1, &g)) return slgc_fail(ctx, SLGC_EINVAL, "unsupported N=%d", N);
    const size_t npix = (size_t)H * W;
    RunPtrs runs{};
    if ((rc = upload_runs(ctx, &stack, &dtype, 1, N, npix, &runs))) return rc;
    double *d_ld = nullptr, *d_lg = nullptr;
    if (L_d) {
        void *d_in;
        if ((rc = slgc_ws(ctx, 1, npix * 16, &d_in))) return rc;
        d_ld = (double *)d_in;
        d_lg = d_ld + npix;
        if (npix) {
            HIP_TRY(ctx, hipMemcpyAsync(d_ld, L_d, npix * 8, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(d_lg, L_g, npix * 8, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    void *d_codes;
    if ((rc = slgc_ws(ctx, 2, 2 * (size_t)g.L * npix, &d_codes))) return rc;
    int8_t *d_hc = (int8_t *)d_codes, *d_vc = d_hc + (size_t)g.L * npix;
    rc = launch_decode_generic(ctx, g, runs, dtype, npix, npix, eps, d_ld, d_lg, nullptr, nullptr, d_hc, d_vc, nullptr, nullptr, nullptr,
                               nullptr);
    if (rc) return rc;
    if (npix) {
        if ((rc = download_par(ctx, h_codes, d_hc, (size_t)g.L * npix))) return rc;
        if ((rc = download_par(ctx, v_codes, d_vc, (size_t)g.L * npix))) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

extern "C" int slgc_is_lit(slgc_ctx *ctx, const void *stack, int dtype, int N, int H, int W, const double *L_d, const double *L_g,
                           double eps, double m, int8_t *h_codes, int8_t *v_codes)
{
    (void)m;  // rule 0 of get_is_lit (decode_codes.py:169-170) rewrites the initial -1: m has no effect
    if (!L_d || !L_g) return slgc_fail(ctx, SLGC_EINVAL, "L_d / L_g are required");
    return codes_common(ctx, stack, dtype, N, H, W, L_d, L_g, eps, h_codes, v_codes);
}

extern "C" int slgc_codes(slgc_ctx *ctx, const void *stack, int dtype, int N, int H, int W, double eps, double m, int8_t *h_codes,
                          int8_t *v_codes)
{
    (void)m;
    return codes_common(ctx, stack, dtype, N, H, W, nullptr, nullptr, eps, h_codes, v_codes);
}

extern "C" int slgc_codes_to_pixels(slgc_ctx *ctx, const int8_t *h_codes, const int8_t *v_codes, int n_runs, int L, int H, int W,
                                    int64_t *h_pixels, int64_t *v_pixels)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!h_codes || !v_codes || !h_pixels || !v_pixels) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (L < 1 || L > 15 || n_runs < 1 || H < 0 || W < 0) return slgc_fail(ctx, SLGC_EINVAL, "bad L / n_runs / size");
    const size_t npix = (size_t)H * W, cb = (size_t)n_runs * L * npix;
    void *d_codes, *d_maps;
    if ((rc = slgc_ws(ctx, 2, 2 * cb, &d_codes))) return rc;
    if ((rc = slgc_ws(ctx, 3, npix * 16, &d_maps))) return rc;
    int8_t *d_hc = (int8_t *)d_codes, *d_vc = d_hc + cb;
    int64_t *d_h = (int64_t *)d_maps, *d_v = d_h + npix;
    if (cb) {
        HIP_TRY(ctx, hipMemcpyAsync(d_hc, h_codes, cb, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_vc, v_codes, cb, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = launch_codes_to_pixels(ctx, d_hc, d_vc, n_runs, L, npix, d_h, d_v))) return rc;
    if (npix) {
        if ((rc = download_par(ctx, h_pixels, d_h, npix * 8))) return rc;
        if ((rc = download_par(ctx, v_pixels, d_v, npix * 8))) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

extern "C" int slgc_decode(slgc_ctx *ctx, const void *const *stacks, int dtype, int n_runs, int N, int H, int W, double eps, double m,
                           int64_t *h_pixels, int64_t *v_pixels)
{
    (void)m;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if ((rc = check_dims(ctx, dtype, N, H, W))) return rc;
    if (!stacks || !h_pixels || !v_pixels) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (std::isnan(eps)) return slgc_fail(ctx, SLGC_EINVAL, "eps is NaN");
    DecodeGeom g;
    if (slgc_make_geom(N, n_runs, &g)) return slgc_fail(ctx, SLGC_EINVAL, "unsupported N=%d / n_runs=%d (max %d)", N, n_runs, SLGC_MAX_RUNS);
    const size_t npix = (size_t)H * W;
    RunPtrs runs{};
    if ((rc = upload_runs(ctx, stacks, &dtype, n_runs, N, npix, &runs))) return rc;
    void *d_maps;
    if ((rc = slgc_ws(ctx, 3, npix * 16, &d_maps))) return rc;
    int64_t *d_h = (int64_t *)d_maps, *d_v = d_h + npix;
    int e;
    if (dtype == SLGC_U8 && decode_fast_eligible(eps, &e)) {
        // product kernel (integer thresholds, int16 maps), widened to the reference's int64 on the device
        void *d16;
        if ((rc = slgc_ws(ctx, 2, npix * 4 + 128, &d16))) return rc;
        int16_t *d_h16 = (int16_t *)d16, *d_v16 = d_h16 + ((npix + 31) & ~(size_t)31);
        rc = launch_decode_fast(ctx, g, runs, npix, H, W, e, d_h16, d_v16, 0);
        if (!rc) rc = launch_widen_maps(ctx, d_h16, d_v16, npix, d_h, d_v);
    } else {
        rc = launch_decode_generic(ctx, g, runs, dtype, npix, npix, eps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   nullptr, d_h, d_v);
    }
    if (rc) return rc;
    if (npix) {
        if ((rc = download_par(ctx, h_pixels, d_h, npix * 8))) return rc;
        if ((rc = download_par(ctx, v_pixels, d_v, npix * 8))) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

// ------------------------------------------------------------------------------------------ triangulation, host buffers
extern "C" int slgc_set_calibration(slgc_ctx *ctx, const double cam_K[9], const double *cam_dist, int n_cam_dist,
                                    const double proj_K[9], const double *proj_dist, int n_proj_dist, const double R[9],
                                    const double T[3])
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!cam_K || !proj_K || !R || !T) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (n_cam_dist < 0 || n_cam_dist > 14 || n_proj_dist < 0 || n_proj_dist > 14 || (n_cam_dist && !cam_dist) || (n_proj_dist && !proj_dist))
        return slgc_fail(ctx, SLGC_EINVAL, "distortion vectors must hold 0..14 coefficients");
    for (int j = 12; j < 14; ++j)
        if ((j < n_cam_dist && cam_dist[j] != 0.0) || (j < n_proj_dist && proj_dist[j] != 0.0))
            return slgc_fail(ctx, SLGC_EINVAL, "tilted-sensor coefficients (tauX, tauY) are not supported");
    Calib &c = ctx->calib;
    memset(&c, 0, sizeof c);
    c.cam_k[0] = cam_K[0]; c.cam_k[1] = cam_K[4]; c.cam_k[2] = cam_K[2]; c.cam_k[3] = cam_K[5];       // fx fy cx cy (skew ignored, as OpenCV)
    c.proj_k[0] = proj_K[0]; c.proj_k[1] = proj_K[4]; c.proj_k[2] = proj_K[2]; c.proj_k[3] = proj_K[5];
    for (int j = 0; j < n_cam_dist && j < 12; ++j) c.cam_d[j] = cam_dist[j];
    for (int j = 0; j < n_proj_dist && j < 12; ++j) c.proj_d[j] = proj_dist[j];
    memcpy(c.R, R, sizeof c.R);
    memcpy(c.T, T, sizeof c.T);
    c.t_len = sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);                                            // triangulate.py:89
    ctx->have_calib = true;
    ++ctx->calib_ver;
    return SLGC_OK;
}

extern "C" int slgc_cam_proj_pts_count(slgc_ctx *ctx, const int64_t *h_pixels, const int64_t *v_pixels, int cam_w, int cam_h,
                                       int proj_w, int proj_h, const uint8_t *white_rgb, int order, int64_t *M)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!h_pixels || !v_pixels || !M) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (cam_w < 0 || cam_h < 0 || (order != SLGC_ORDER_X && order != SLGC_ORDER_ROW)) return slgc_fail(ctx, SLGC_EINVAL, "bad size / order");
    const size_t npix = (size_t)cam_w * cam_h;
    ctx->pend_M = -1;
    void *d_maps, *d_white = nullptr, *d_out, *d_total;
    if ((rc = slgc_ws(ctx, 3, npix * 16, &d_maps))) return rc;
    if ((rc = slgc_ws(ctx, 6, npix * (8 + 8 + (white_rgb ? 24 : 0)) + 64, &d_out))) return rc;
    if ((rc = slgc_ws(ctx, 7, 64, &d_total))) return rc;
    int64_t *d_h = (int64_t *)d_maps, *d_v = d_h + npix;
    if (npix) {
        HIP_TRY(ctx, hipMemcpyAsync(d_h, h_pixels, npix * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_v, v_pixels, npix * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if (white_rgb) {
        if ((rc = slgc_ws(ctx, 1, npix * 3, &d_white))) return rc;
        if (npix) HIP_TRY(ctx, hipMemcpyAsync(d_white, white_rgb, npix * 3, hipMemcpyHostToDevice, ctx->stream));
    }
    double *d_colors = (double *)d_out;  // colours first (8-byte aligned), then cam, then proj
    float *d_cam = (float *)((char *)d_out + (white_rgb ? npix * 24 : 0)), *d_proj = d_cam + 2 * npix;
    rc = launch_correspond(ctx, d_h, d_v, cam_w, cam_h, proj_w, proj_h, (const uint8_t *)d_white, order, d_cam, d_proj, d_colors,
                           (unsigned long long *)d_total);
    if (rc) return rc;
    unsigned long long total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pend_M = (int64_t)total;
    ctx->pend_npix = npix;
    ctx->pend_colors = white_rgb != nullptr;
    ctx->pend_gen = ctx->ws_gen[6];
    *M = (int64_t)total;
    return SLGC_OK;
}

extern "C" int slgc_cam_proj_pts_fetch(slgc_ctx *ctx, float *cam_pts, float *proj_pts, double *colors)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (ctx->pend_M < 0) return slgc_fail(ctx, SLGC_ESTATE, "no pending correspondence list (call slgc_cam_proj_pts_count first)");
    if (ctx->pend_gen != ctx->ws_gen[6]) {
        ctx->pend_M = -1;
        return slgc_fail(ctx, SLGC_ESTATE, "the correspondence lists were overwritten by a later call on this context (count again)");
    }
    const size_t npix = ctx->pend_npix, M = (size_t)ctx->pend_M;
    char *d_out = (char *)ctx->ws[6];
    const double *d_colors = (const double *)d_out;
    const float *d_cam = (const float *)(d_out + (ctx->pend_colors ? npix * 24 : 0)), *d_proj = d_cam + 2 * npix;
    if (M) {
        if (cam_pts && (rc = download_par(ctx, cam_pts, d_cam, M * 8))) return rc;
        if (proj_pts && (rc = download_par(ctx, proj_pts, d_proj, M * 8))) return rc;
        if (colors && ctx->pend_colors && (rc = download_par(ctx, colors, d_colors, M * 24))) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

extern "C" int slgc_triangulate(slgc_ctx *ctx, const float *cam_pts, const float *proj_pts, int64_t M, int mode, double *xyz)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if (M < 0 || (M && (!cam_pts || !proj_pts || !xyz))) return slgc_fail(ctx, SLGC_EINVAL, "null pointer / negative M");
    if (mode != SLGC_TRI_EXACT && mode != SLGC_TRI_ALGEBRAIC) return slgc_fail(ctx, SLGC_EINVAL, "bad mode");
    if (M == 0) return SLGC_OK;
    void *d_in, *d_out;
    if ((rc = slgc_ws(ctx, 1, (size_t)M * 16, &d_in))) return rc;
    if ((rc = slgc_ws(ctx, 6, (size_t)M * 24, &d_out))) return rc;
    float *d_cam = (float *)d_in, *d_proj = d_cam + 2 * (size_t)M;
    HIP_TRY(ctx, hipMemcpyAsync(d_cam, cam_pts, (size_t)M * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_proj, proj_pts, (size_t)M * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_triangulate_list(ctx, d_cam, d_proj, M, mode, (double *)d_out))) return rc;
    if ((rc = download_par(ctx, xyz, d_out, (size_t)M * 24))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pend_M = -1;  // slot 6 was reused
    return SLGC_OK;
}

extern "C" int slgc_undistort_points(slgc_ctx *ctx, int which, const float *pts, int64_t M, float *out)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if ((which != 0 && which != 1) || M < 0 || (M && (!pts || !out))) return slgc_fail(ctx, SLGC_EINVAL, "which must be 0 (camera) or 1 (projector); null pointer / negative M");
    if (M == 0) return SLGC_OK;
    void *d_in, *d_out;
    if ((rc = slgc_ws(ctx, 1, (size_t)M * 8, &d_in))) return rc;
    if ((rc = slgc_ws(ctx, 6, (size_t)M * 8, &d_out))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_in, pts, (size_t)M * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_undistort_list(ctx, which, (const float *)d_in, M, (float *)d_out))) return rc;
    if ((rc = download_par(ctx, out, d_out, (size_t)M * 8))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

extern "C" int slgc_filter_count(slgc_ctx *ctx, const double *xyz, const double *colors, int64_t M, double threshold, int64_t *kept)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (M < 0 || !kept || (M && !xyz)) return slgc_fail(ctx, SLGC_EINVAL, "null pointer / negative M");
    ctx->filt_M = -1;
    void *d_in, *d_total;
    if ((rc = slgc_ws(ctx, 1, (size_t)M * (24 + (colors ? 24 : 0)), &d_in))) return rc;
    if ((rc = slgc_ws(ctx, 7, 64, &d_total))) return rc;
    double *d_xyz = (double *)d_in, *d_col = colors ? d_xyz + 3 * (size_t)M : nullptr;
    if (M) {
        HIP_TRY(ctx, hipMemcpyAsync(d_xyz, xyz, (size_t)M * 24, hipMemcpyHostToDevice, ctx->stream));
        if (colors) HIP_TRY(ctx, hipMemcpyAsync(d_col, colors, (size_t)M * 24, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = launch_filter(ctx, d_xyz, d_col, M, threshold, nullptr, nullptr, (unsigned long long *)d_total, 0))) return rc;
    unsigned long long total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    void *d_out;
    if ((rc = slgc_ws(ctx, 6, (size_t)total * (24 + (colors ? 24 : 0)), &d_out))) return rc;
    double *d_xo = (double *)d_out, *d_co = colors ? d_xo + 3 * (size_t)total : nullptr;
    if ((rc = launch_filter(ctx, d_xyz, d_col, M, threshold, d_xo, d_co, (unsigned long long *)d_total, 1))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->filt_M = (int64_t)total;
    ctx->filt_colors = colors != nullptr;
    ctx->filt_gen = ctx->ws_gen[6];
    ctx->pend_M = -1;
    *kept = (int64_t)total;
    return SLGC_OK;
}

extern "C" int slgc_filter_fetch(slgc_ctx *ctx, double *xyz_out, double *colors_out)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (ctx->filt_M < 0) return slgc_fail(ctx, SLGC_ESTATE, "no pending filter result (call slgc_filter_count first)");
    if (ctx->filt_gen != ctx->ws_gen[6]) {
        ctx->filt_M = -1;
        return slgc_fail(ctx, SLGC_ESTATE, "the filter result was overwritten by a later call on this context (count again)");
    }
    const size_t K = (size_t)ctx->filt_M;
    const double *d_xo = (const double *)ctx->ws[6], *d_co = d_xo + 3 * K;
    if (K) {
        if (xyz_out && (rc = download_par(ctx, xyz_out, d_xo, K * 24))) return rc;
        if (colors_out && ctx->filt_colors && (rc = download_par(ctx, colors_out, d_co, K * 24))) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

// ------------------------------------------------------------------------------------------ ingest ("next" rows)
extern "C" int slgc_to_gray(slgc_ctx *ctx, const uint8_t *bgr, int n_frames, int H, int W, int coeff_bits, uint8_t *gray)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!bgr || !gray || n_frames < 0 || H < 0 || W < 0) return slgc_fail(ctx, SLGC_EINVAL, "null pointer / negative size");
    if (coeff_bits != 14 && coeff_bits != 15) return slgc_fail(ctx, SLGC_EINVAL, "coeff_bits must be 15 (OpenCV 4.x) or 14");
    const size_t npix = (size_t)n_frames * H * W;
    void *d_in, *d_out;
    if ((rc = slgc_ws(ctx, 0, npix * 3, &d_in))) return rc;
    if ((rc = slgc_ws(ctx, 2, npix, &d_out))) return rc;
    if (npix) HIP_TRY(ctx, hipMemcpyAsync(d_in, bgr, npix * 3, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_bgr_to_gray(ctx, (const uint8_t *)d_in, (uint8_t *)d_out, npix, coeff_bits))) return rc;
    if (npix && (rc = download_par(ctx, gray, d_out, npix))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

extern "C" int slgc_to_gray_dev(slgc_ctx *ctx, const uint8_t *d_bgr, size_t npix, int coeff_bits, uint8_t *d_gray)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_bgr || !d_gray) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (coeff_bits != 14 && coeff_bits != 15) return slgc_fail(ctx, SLGC_EINVAL, "coeff_bits must be 15 (OpenCV 4.x) or 14");
    return launch_bgr_to_gray(ctx, d_bgr, d_gray, npix, coeff_bits);
}

// ---- the tile-interleaved stack layout (slgc_tune "stack_tile_log2"): the two ways a stack gets into it
extern "C" int slgc_tiled_stack_bytes(int N, size_t npix, int tile_log2, size_t *bytes)
{
    if (!bytes || N < 1 || tile_log2 < 8 || tile_log2 > 24) return SLGC_EINVAL;
    const size_t piece = (size_t)1 << tile_log2;
    *bytes = ((npix + piece - 1) >> tile_log2) * (size_t)N * piece;
    return SLGC_OK;
}

extern "C" int slgc_tile_stack_dev(slgc_ctx *ctx, const uint8_t *d_planar, size_t plane_stride, int N, size_t npix, int tile_log2, uint8_t *d_tiled)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (N < 0 || (N && npix && (!d_planar || !d_tiled)) || plane_stride < npix || tile_log2 < 8 || tile_log2 > 24)
        return slgc_fail(ctx, SLGC_EINVAL, "null pointer / plane_stride smaller than a frame / tile_log2 outside 8..24");
    return launch_tile_stack(ctx, d_planar, plane_stride, N, npix, tile_log2, d_tiled);
}

extern "C" int slgc_to_gray_tiled_dev(slgc_ctx *ctx, const uint8_t *d_bgr, int n_frames, size_t npix, int coeff_bits, int tile_log2, uint8_t *d_tiled)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (coeff_bits != 14 && coeff_bits != 15) return slgc_fail(ctx, SLGC_EINVAL, "coeff_bits must be 15 (OpenCV 4.x) or 14");
    if (n_frames < 0 || (n_frames && npix && (!d_bgr || !d_tiled)) || tile_log2 < 8 || tile_log2 > 24)
        return slgc_fail(ctx, SLGC_EINVAL, "null pointer / negative size / tile_log2 outside 8..24");
    return launch_bgr_to_gray_tiled(ctx, d_bgr, d_tiled, n_frames, npix, coeff_bits, tile_log2);
}

extern "C" int slgc_frame_diff_counts(slgc_ctx *ctx, const void *frames, int dtype, int n_frames, size_t elems_per_frame, double thresh,
                                      int64_t *counts)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (dtype != SLGC_U8 && dtype != SLGC_F64) return slgc_fail(ctx, SLGC_EINVAL, "dtype must be SLGC_U8 or SLGC_F64");
    if (n_frames < 0 || (n_frames > 1 && (!frames || !counts))) return slgc_fail(ctx, SLGC_EINVAL, "null pointer / negative size");
    if (n_frames < 2) return SLGC_OK;
    const size_t bytes = (size_t)n_frames * elems_per_frame * esize(dtype);
    void *d_in, *d_cnt;
    if ((rc = slgc_ws(ctx, 0, bytes, &d_in))) return rc;
    if ((rc = slgc_ws(ctx, 7, 8 * (size_t)n_frames + 64, &d_cnt))) return rc;
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(d_in, frames, bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_frame_diff_counts(ctx, d_in, dtype, n_frames, elems_per_frame, thresh, (unsigned long long *)d_cnt))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(counts, d_cnt, 8 * (size_t)(n_frames - 1), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

extern "C" int slgc_frame_diff_counts_dev(slgc_ctx *ctx, const void *d_frames, int dtype, int n_frames, size_t elems_per_frame, double thresh,
                                          unsigned long long *d_counts)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (dtype != SLGC_U8 && dtype != SLGC_F64) return slgc_fail(ctx, SLGC_EINVAL, "dtype must be SLGC_U8 or SLGC_F64");
    if (n_frames < 0 || (n_frames > 1 && (!d_frames || !d_counts))) return slgc_fail(ctx, SLGC_EINVAL, "null pointer / negative size");
    return launch_frame_diff_counts(ctx, d_frames, dtype, n_frames, elems_per_frame, thresh, d_counts);
}

// ------------------------------------------------------------------------------------------ whole pipeline, one upload
// Steps 2-4 of the device-resident chain shared by slgc_pipeline_count and slgc_compute_count: from int64 maps already in HBM to the lists of
// get_cam_proj_pts (slot 8), the float64 (3,M) points of triangulate (slot 9) and, with a threshold, the kept points / colours of filter_3d_pts
// (slot 10).  Leaves the pipe_* bookkeeping that slgc_pipeline_fetch reads; synchronises (the two list lengths are needed on the host).
static int lists_tri_filter(slgc_ctx *ctx, const int64_t *d_h, const int64_t *d_v, int cam_w, int cam_h, int proj_w, int proj_h,
                            const uint8_t *white_rgb, int order, int mode, double threshold)
{
    int rc;
    const size_t npix = (size_t)cam_w * cam_h;
    // 2. correspondences (slot 8: colours | cam | proj), colour source in slot 1
    void *d_white = nullptr, *d_corr, *d_total;
    if ((rc = slgc_ws(ctx, 8, npix * (8 + 8 + (white_rgb ? 24 : 0)) + 64, &d_corr))) return rc;
    if ((rc = slgc_ws(ctx, 7, 64, &d_total))) return rc;
    if (white_rgb) {
        if ((rc = slgc_ws(ctx, 1, npix * 3, &d_white))) return rc;
        if (npix) HIP_TRY(ctx, hipMemcpyAsync(d_white, white_rgb, npix * 3, hipMemcpyHostToDevice, ctx->stream));
    }
    double *d_colors = (double *)d_corr;
    float *d_cam = (float *)((char *)d_corr + (white_rgb ? npix * 24 : 0)), *d_proj = d_cam + 2 * npix;
    if ((rc = launch_correspond(ctx, d_h, d_v, cam_w, cam_h, proj_w, proj_h, (const uint8_t *)d_white, order, d_cam, d_proj, d_colors,
                                (unsigned long long *)d_total)))
        return rc;
    unsigned long long total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // 3. triangulate (slot 9: xyz (3,M) float64)
    void *d_xyz;
    if ((rc = slgc_ws(ctx, 9, (size_t)total * 24, &d_xyz))) return rc;
    if ((rc = launch_triangulate_list(ctx, d_cam, d_proj, (int64_t)total, mode, (double *)d_xyz))) return rc;
    ctx->pipe_M_raw = (int64_t)total;
    ctx->pipe_npix = npix;
    ctx->pipe_colors = white_rgb != nullptr;
    ctx->pipe_filtered = false;
    ctx->pipe_M = (int64_t)total;
    // 4. optional box filter (slot 10: xyz' | colours')
    if (!std::isnan(threshold)) {
        if ((rc = launch_filter(ctx, (const double *)d_xyz, white_rgb ? d_colors : nullptr, (int64_t)total, threshold, nullptr, nullptr,
                                (unsigned long long *)d_total, 0)))
            return rc;
        unsigned long long kept = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&kept, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        void *d_f;
        if ((rc = slgc_ws(ctx, 10, (size_t)kept * (24 + (white_rgb ? 24 : 0)), &d_f))) return rc;
        double *d_fx = (double *)d_f, *d_fc = white_rgb ? d_fx + 3 * (size_t)kept : nullptr;
        if ((rc = launch_filter(ctx, (const double *)d_xyz, white_rgb ? d_colors : nullptr, (int64_t)total, threshold, d_fx, d_fc,
                                (unsigned long long *)d_total, 1)))
            return rc;
        ctx->pipe_filtered = true;
        ctx->pipe_M = (int64_t)kept;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pend_M = -1;
    ctx->filt_M = -1;
    ctx->pipe_gen[0] = ctx->ws_gen[3]; ctx->pipe_gen[1] = ctx->ws_gen[8]; ctx->pipe_gen[2] = ctx->ws_gen[9]; ctx->pipe_gen[3] = ctx->ws_gen[10];
    return SLGC_OK;
}


// Driver glue of the reference in one device-resident pass: src/3-capture_decode.py:75-100 (get_codes per run, merge,
// gray_to_decimal) followed by src/4-triangulate.py:50-71 (get_cam_proj_pts, triangulate, filter_3d_pts).
extern "C" int slgc_pipeline_count(slgc_ctx *ctx, const void *const *stacks, int dtype, int n_runs, int N, int H, int W, double eps,
                                   double m, int proj_w, int proj_h, const uint8_t *white_rgb, int order, int mode, double threshold,
                                   int64_t *M)
{
    (void)m;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    ctx->pipe_M = -1;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if ((rc = check_dims(ctx, dtype, N, H, W))) return rc;
    if (!stacks || !M) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (std::isnan(eps)) return slgc_fail(ctx, SLGC_EINVAL, "eps is NaN");
    if (order != SLGC_ORDER_X && order != SLGC_ORDER_ROW) return slgc_fail(ctx, SLGC_EINVAL, "bad order");
    if (mode != SLGC_TRI_EXACT && mode != SLGC_TRI_ALGEBRAIC) return slgc_fail(ctx, SLGC_EINVAL, "bad mode");
    DecodeGeom g;
    if (slgc_make_geom(N, n_runs, &g)) return slgc_fail(ctx, SLGC_EINVAL, "unsupported N=%d / n_runs=%d (max %d)", N, n_runs, SLGC_MAX_RUNS);
    const size_t npix = (size_t)H * W;
    RunPtrs runs{};
    if ((rc = upload_runs(ctx, stacks, &dtype, n_runs, N, npix, &runs))) return rc;
    // 1. decode -> int64 maps (slot 3)
    void *d_maps;
    if ((rc = slgc_ws(ctx, 3, npix * 16, &d_maps))) return rc;
    int64_t *d_h = (int64_t *)d_maps, *d_v = d_h + npix;
    int e;
    if (dtype == SLGC_U8 && decode_fast_eligible(eps, &e)) {
        void *d16;
        if ((rc = slgc_ws(ctx, 2, npix * 4 + 128, &d16))) return rc;
        int16_t *d_h16 = (int16_t *)d16, *d_v16 = d_h16 + ((npix + 31) & ~(size_t)31);
        rc = launch_decode_fast(ctx, g, runs, npix, H, W, e, d_h16, d_v16, 0);
        if (!rc) rc = launch_widen_maps(ctx, d_h16, d_v16, npix, d_h, d_v);
    } else {
        rc = launch_decode_generic(ctx, g, runs, dtype, npix, npix, eps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   nullptr, d_h, d_v);
    }
    if (rc) return rc;
    if ((rc = lists_tri_filter(ctx, d_h, d_v, W, H, proj_w, proj_h, white_rgb, order, mode, threshold))) return rc;
    *M = ctx->pipe_M;
    return SLGC_OK;
}

// Triangulation.compute() of the drop-in class (the fused entry BASELINE.json's north star names): what src/4-triangulate.py:62-71 does with a
// Triangulate object -- get_cam_proj_pts (triangulate.py:39-71), triangulate (:73-97), filter_3d_pts (:99-122) -- as ONE upload, the same three
// kernels the three host entry points launch (bit-identical results by construction), and ONE download of what the script keeps.
extern "C" int slgc_compute_count(slgc_ctx *ctx, const int64_t *h_pixels, const int64_t *v_pixels, int cam_w, int cam_h, int proj_w, int proj_h,
                                  const uint8_t *white_rgb, int order, int mode, double threshold, int64_t *M, int64_t *M_unfiltered)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    ctx->pipe_M = -1;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if (!h_pixels || !v_pixels || !M) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (cam_w < 0 || cam_h < 0 || (order != SLGC_ORDER_X && order != SLGC_ORDER_ROW)) return slgc_fail(ctx, SLGC_EINVAL, "bad size / order");
    if (mode != SLGC_TRI_EXACT && mode != SLGC_TRI_ALGEBRAIC) return slgc_fail(ctx, SLGC_EINVAL, "bad mode");
    const size_t npix = (size_t)cam_w * cam_h;
    void *d_maps;
    if ((rc = slgc_ws(ctx, 3, npix * 16, &d_maps))) return rc;
    int64_t *d_h = (int64_t *)d_maps, *d_v = d_h + npix;
    ctx->last_input_path = 2;
    static const int pack = xcd_env("SLGC_I64_PACK", 1);          // 0: always ship int64 (A/B of the narrowing)
    bool narrowed = false;
    if (npix && pack) {          // the maps hold -1 or a projector coordinate: 2 of their 8 bytes cross the link, narrowed by host threads into pinned memory
        void *st;
        if ((rc = host_staging(ctx, npix * 4, &st))) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // the staging buffer may still feed the previous call's copy
        const int64_t *maps[2] = {h_pixels, v_pixels};
        const int ok = slgc_host::narrow_i64_to_i16(maps, 2, npix, (int16_t *)st);
        if (ok < 0) return slgc_fail(ctx, SLGC_ENOMEM, "host threads for the int64 -> int16 narrowing");
        if (ok) {
            void *d16;
            if ((rc = slgc_ws(ctx, 2, npix * 4, &d16))) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(d16, st, npix * 4, hipMemcpyHostToDevice, ctx->stream));
            if ((rc = launch_widen_maps(ctx, (const int16_t *)d16, (const int16_t *)d16 + npix, npix, d_h, d_v))) return rc;
            narrowed = true;
            ctx->last_input_path = 1;
        }
    }
    if (npix && !narrowed) {
        HIP_TRY(ctx, hipMemcpyAsync(d_h, h_pixels, npix * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_v, v_pixels, npix * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = lists_tri_filter(ctx, d_h, d_v, cam_w, cam_h, proj_w, proj_h, white_rgb, order, mode, threshold))) return rc;
    *M = ctx->pipe_M;
    if (M_unfiltered) *M_unfiltered = ctx->pipe_M_raw;
    return SLGC_OK;
}

extern "C" int slgc_pipeline_fetch(slgc_ctx *ctx, int64_t *h_pixels, int64_t *v_pixels, double *xyz, double *colors, int64_t *M_unfiltered,
                                   float *cam_pts, float *proj_pts);
extern "C" int slgc_compute_fetch(slgc_ctx *ctx, double *xyz, double *colors)
{
    return slgc_pipeline_fetch(ctx, nullptr, nullptr, xyz, colors, nullptr, nullptr, nullptr);
}

extern "C" int slgc_pipeline_fetch(slgc_ctx *ctx, int64_t *h_pixels, int64_t *v_pixels, double *xyz, double *colors, int64_t *M_unfiltered,
                                   float *cam_pts, float *proj_pts)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (ctx->pipe_M < 0) return slgc_fail(ctx, SLGC_ESTATE, "no pending pipeline result (call slgc_pipeline_count first)");
    if (ctx->pipe_gen[0] != ctx->ws_gen[3] || ctx->pipe_gen[1] != ctx->ws_gen[8] || ctx->pipe_gen[2] != ctx->ws_gen[9] ||
        ctx->pipe_gen[3] != ctx->ws_gen[10]) {
        ctx->pipe_M = -1;
        return slgc_fail(ctx, SLGC_ESTATE, "the pipeline result was overwritten by a later call on this context (count again)");
    }
    const size_t npix = ctx->pipe_npix, M = (size_t)ctx->pipe_M, Mr = (size_t)ctx->pipe_M_raw;
    const int64_t *d_h = (const int64_t *)ctx->ws[3];
    if (npix) {
        if (h_pixels && (rc = download_par(ctx, h_pixels, d_h, npix * 8))) return rc;
        if (v_pixels && (rc = download_par(ctx, v_pixels, d_h + npix, npix * 8))) return rc;
    }
    const char *d_corr = (const char *)ctx->ws[8];
    const double *d_colors = (const double *)d_corr;
    const float *d_cam = (const float *)(d_corr + (ctx->pipe_colors ? npix * 24 : 0)), *d_proj = d_cam + 2 * npix;
    if (Mr) {
        if (cam_pts && (rc = download_par(ctx, cam_pts, d_cam, Mr * 8))) return rc;      // unfiltered lists
        if (proj_pts && (rc = download_par(ctx, proj_pts, d_proj, Mr * 8))) return rc;
    }
    if (M) {
        if (ctx->pipe_filtered) {
            const double *d_fx = (const double *)ctx->ws[10];
            if (xyz && (rc = download_par(ctx, xyz, d_fx, M * 24))) return rc;
            if (colors && ctx->pipe_colors && (rc = download_par(ctx, colors, d_fx + 3 * M, M * 24))) return rc;
        } else {
            if (xyz && (rc = download_par(ctx, xyz, ctx->ws[9], M * 24))) return rc;
            if (colors && ctx->pipe_colors && (rc = download_par(ctx, colors, d_colors, M * 24))) return rc;
        }
    }
    if (M_unfiltered) *M_unfiltered = ctx->pipe_M_raw;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

// ------------------------------------------------------------------------------------------ device-resident path
extern "C" int slgc_dev_alloc(slgc_ctx *ctx, size_t bytes, void **dptr)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!dptr) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (hipMalloc(dptr, bytes ? bytes : 16) != hipSuccess) return slgc_fail(ctx, SLGC_ENOMEM, "hipMalloc(%zu) failed", bytes);
    return SLGC_OK;
}

extern "C" int slgc_dev_free(slgc_ctx *ctx, void *dptr)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (slgc_direct_is_registered(ctx, dptr))
        return slgc_fail(ctx, SLGC_ESTATE, "the buffer is registered with the direct exchange: peers hold mappings of it (slgc_direct_unregister first, collectively)");
    if (dptr) HIP_TRY(ctx, hipFree(dptr));
    return SLGC_OK;
}

extern "C" int slgc_h2d(slgc_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

extern "C" int slgc_d2h(slgc_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (bytes && (rc = download_par(ctx, dst_host, src_dev, bytes))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return slgc_direct_check(ctx);                 // bytes of a scan whose exchange timed out are not a result
}

extern "C" int slgc_dev_memset(slgc_ctx *ctx, void *dptr, int value, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (bytes) HIP_TRY(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return SLGC_OK;
}

static int prof_mark(slgc_ctx *ctx, int phase)
{
    if (!ctx->prof_on) return SLGC_OK;
    if (phase == 0) {
        ctx->prof_sampling = (ctx->prof_seen++ % ctx->prof_stride) == 0 && ctx->prof_n < ctx->prof_cap;
        ctx->prof_bound = false;
        ctx->prof_cur[0] = ctx->prof_sampling ? ctx->prof_ev[2 * ctx->prof_n] : nullptr;
        ctx->prof_cur[1] = ctx->prof_sampling ? ctx->prof_ev[2 * ctx->prof_n + 1] : nullptr;
        return SLGC_OK;       // the launcher binds the pair to the kernel's dispatch (SLGC_LAUNCH) ...
    }
    if (!ctx->prof_sampling) return SLGC_OK;      // not a sampled launch (stride) or ring full
    if (!ctx->prof_bound) {   // ... a launcher that does not know about it gets a record after the launch; its start is lost: drop the sample
        ctx->prof_cur[0] = ctx->prof_cur[1] = nullptr;
        return SLGC_OK;
    }
    ctx->prof_cur[0] = ctx->prof_cur[1] = nullptr;
    ++ctx->prof_n;
    return SLGC_OK;
}

static int decode_fast_timed(slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, int rows, int W, int e,
                             int16_t *d_h, int16_t *d_v, int variant)
{
    int rc = prof_mark(ctx, 0);
    if (rc) return rc;
    if ((rc = launch_decode_fast(ctx, g, runs, plane_stride, rows, W, e, d_h, d_v, variant))) return rc;
    return prof_mark(ctx, 1);
}

extern "C" int slgc_prof_begin(slgc_ctx *ctx, int max_launches, int stride)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (max_launches < 1 || max_launches > 65536 || stride < 1) return slgc_fail(ctx, SLGC_EINVAL, "max_launches / stride out of range");
    ctx->prof_stride = stride;
    ctx->prof_seen = 0;
    ctx->prof_sampling = false;
    if (ctx->prof_cap < max_launches) {
        hipEvent_t *ev = new (std::nothrow) hipEvent_t[2 * (size_t)max_launches];
        if (!ev) return slgc_fail(ctx, SLGC_ENOMEM, "event ring");
        for (int i = 0; i < 2 * ctx->prof_cap; ++i) ev[i] = ctx->prof_ev[i];
        for (int i = 2 * ctx->prof_cap; i < 2 * max_launches; ++i) HIP_TRY(ctx, hipEventCreate(&ev[i]));
        delete[] ctx->prof_ev;
        ctx->prof_ev = ev;
        ctx->prof_cap = max_launches;
    }
    ctx->prof_n = 0;
    ctx->prof_on = true;
    return SLGC_OK;
}

extern "C" int slgc_prof_end(slgc_ctx *ctx, double *total_ms, int *launches)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!total_ms || !launches) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    ctx->prof_on = false;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0;
    for (int i = 0; i < ctx->prof_n; ++i) {
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = ctx->prof_n;
    return SLGC_OK;
}

extern "C" int slgc_prof_samples(slgc_ctx *ctx, float *ms, int cap, int *n)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!n || cap < 0 || (cap && !ms)) return slgc_fail(ctx, SLGC_EINVAL, "null pointer / negative capacity");
    if (ctx->prof_on) return slgc_fail(ctx, SLGC_ESTATE, "slgc_prof_end has not been called");
    const int m = ctx->prof_n < cap ? ctx->prof_n : cap;
    for (int i = 0; i < m; ++i) HIP_TRY(ctx, hipEventElapsedTime(&ms[i], ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
    *n = ctx->prof_n;
    return SLGC_OK;
}

struct TileOff {           // the context's tile-interleaved setting suspended for the duration of a call (BGR entry points)
    slgc_ctx *c;
    int saved;
    explicit TileOff(slgc_ctx *ctx) : c(ctx), saved(ctx->tune_stack_tile) { c->tune_stack_tile = 0; c->stack_tile_active = 0; }
    ~TileOff() { c->tune_stack_tile = saved; c->stack_tile_active = 0; }
};

struct TileScope {         // what dev_geom switched on for THIS call is switched off when the call returns: no later launch can inherit a stale layout
    slgc_ctx *c;
    explicit TileScope(slgc_ctx *ctx) : c(ctx) {}
    ~TileScope() { c->stack_tile_active = 0; }
};

static int dev_geom(slgc_ctx *ctx, const uint8_t *d_stack, int n_runs, size_t run_stride, size_t plane_stride, int N, int rows, int W,
                    double eps, DecodeGeom *g, RunPtrs *runs, int *e)
{
    if (!d_stack) return slgc_fail(ctx, SLGC_EINVAL, "null stack");
    ctx->stack_tile_active = 0;
    if (ctx->tune_stack_tile) {              // tile-interleaved stack [tile][N][2^k]: the caller says so twice (slgc_tune + plane_stride = 2^k), d_stack = the band's first tile
        const size_t piece = (size_t)1 << ctx->tune_stack_tile;
        if (rows < 0 || W < 0 || plane_stride != piece)
            return slgc_fail(ctx, SLGC_EINVAL, "stack_tile_log2 = %d is set on this context: plane_stride must be the plane piece, %zu bytes", ctx->tune_stack_tile, piece);
        const uint64_t tiles = ((uint64_t)rows * W + piece - 1) >> ctx->tune_stack_tile;
        if (((size_t)rows * W) % 4 || tiles * (uint64_t)N * piece >= 0xfffffff0ull || (uintptr_t)d_stack % 16)
            return slgc_fail(ctx, SLGC_EINVAL, "a tile-interleaved stack needs a band of a multiple of 4 pixels, under 4 GB, 16-byte aligned");
        ctx->stack_tile_active = ctx->tune_stack_tile;
    } else
    if (rows < 0 || W < 0 || plane_stride < (size_t)rows * W) return slgc_fail(ctx, SLGC_EINVAL, "plane_stride smaller than the band");
    if (slgc_make_geom(N, n_runs, g)) return slgc_fail(ctx, SLGC_EINVAL, "unsupported N=%d / n_runs=%d", N, n_runs);
    if (!decode_fast_eligible(eps, e))
        return slgc_fail(ctx, SLGC_EINVAL, "device-resident decode needs an integer eps in [0,255] (got %g); use slgc_decode", eps);
    for (int r = 0; r < n_runs; ++r) runs->p[r] = d_stack + (size_t)r * run_stride;
    return SLGC_OK;
}

extern "C" int slgc_decode_dev(slgc_ctx *ctx, const uint8_t *d_stack, int n_runs, size_t run_stride, size_t plane_stride, int N,
                               int rows, int W, double eps, double m, int16_t *d_h, int16_t *d_v, int variant)
{
    (void)m;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_h || !d_v) return slgc_fail(ctx, SLGC_EINVAL, "null output");
    DecodeGeom g;
    RunPtrs runs{};
    int e;
    TileScope tile_scope(ctx);
    if ((rc = dev_geom(ctx, d_stack, n_runs, run_stride, plane_stride, N, rows, W, eps, &g, &runs, &e))) return rc;
    ctx->last_scan_path = SLGC_PATH_NONE;          // a decode alone is not a scan; slgc_triangulate_maps_dev after it completes the two-kernel path
    ctx->last_ragged = 0;
    ctx->last_tri_ragged = 0;                      // slgc_last_scan_ragged after a decode alone reports the decode's bit only
    ctx->decode_pending = 1;
    return decode_fast_timed(ctx, g, runs, plane_stride, rows, W, e, d_h, d_v, variant);
}

extern "C" int slgc_selftest_thresholds(slgc_ctx *ctx, int eps, int black_lo, int black_hi, unsigned long long *mismatches)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    const int skew = (eps & 0x100) ? 1 : 0;      // bit 8: negative control, the literal side is evaluated with eps + 1
    eps &= ~0x100;
    if (!mismatches || eps < 0 || eps > 255 || black_lo < 0 || black_hi > 256 || black_lo >= black_hi)
        return slgc_fail(ctx, SLGC_EINVAL, "eps in 0..255, 0 <= black_lo < black_hi <= 256");
    void *d_bad;
    if ((rc = slgc_ws(ctx, 7, 64, &d_bad))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    if ((rc = launch_selftest_thresholds(ctx, eps, black_lo, black_hi - black_lo, (unsigned long long *)d_bad, skew))) return rc;
    if ((rc = download_par(ctx, mismatches, d_bad, 8))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

extern "C" int slgc_selftest_classify(slgc_ctx *ctx, int negative_control, unsigned long long *mismatches)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!mismatches) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    void *d_bad;
    if ((rc = slgc_ws(ctx, 7, 64, &d_bad))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    if ((rc = launch_selftest_classify(ctx, (unsigned long long *)d_bad, negative_control ? 1 : 0))) return rc;
    if ((rc = download_par(ctx, mismatches, d_bad, 8))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

extern "C" int slgc_triangulate_maps_dev(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, int rows, int W, int row0,
                                         int proj_w, int proj_h, int mode, float *d_xyz, unsigned long long *d_count)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if (!d_h || !d_v || !d_xyz) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (mode < 0 || mode > 3) return slgc_fail(ctx, SLGC_EINVAL, "bad mode");
    if ((rc = launch_triangulate_maps(ctx, d_h, d_v, rows, W, row0, proj_w, proj_h, mode, d_xyz, d_count))) return rc;
    if (!ctx->decode_pending) ctx->last_ragged = 0;      // a triangulation on its own reports its own bit only (no decode of this context is part of it)
    ctx->last_scan_path = (ctx->last_tri_ragged || ctx->last_ragged) ? SLGC_PATH_SPLIT_RAGGED : SLGC_PATH_SPLIT;
    ctx->decode_pending = 0;
    return SLGC_OK;
}

extern "C" int slgc_build_ray_tables_dev(slgc_ctx *ctx, int rows, int W, int row0, int proj_w, int proj_h)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if (rows < 0 || W < 0 || proj_w < 1 || proj_h < 1 || (size_t)proj_w * proj_h >= (1u << 28)) return slgc_fail(ctx, SLGC_EINVAL, "bad band / projector size");
    return ensure_luts(ctx, rows, W, row0, proj_w, proj_h);
}

extern "C" int slgc_ray_table_info(slgc_ctx *ctx, int *in_use, double *max_err)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (in_use) {
        const int w = ctx->lut_cam_W;
        *in_use = SLGC_CAM_NODES_FOR(ctx, w, true).nodes ? 1 : 0;
    }
    if (max_err) *max_err = ctx->lut_cam ? (double)ctx->lut_nodes_err : -1.0;
    return SLGC_OK;
}

extern "C" int slgc_guard_count_dev(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, int rows, int W, int row0, int proj_w, int proj_h,
                                    unsigned long long *d_counts)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if (!d_h || !d_v || !d_counts) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (rows < 0 || W < 0 || proj_w < 1 || proj_h < 1 || (size_t)proj_w * proj_h >= (1u << 28)) return slgc_fail(ctx, SLGC_EINVAL, "bad band / projector size");
    return launch_guard_count(ctx, d_h, d_v, rows, W, row0, proj_w, proj_h, d_counts);
}

extern "C" int slgc_triangulate_wire_dev(slgc_ctx *ctx, const uint8_t *d_wire, int rows, int W, int row0, int proj_w, int proj_h, int mode,
                                         int16_t *d_h, int16_t *d_v, float *d_xyz, unsigned long long *d_count)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if (!d_wire || !d_h || !d_v || !d_xyz) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (mode < 0 || mode > 1) return slgc_fail(ctx, SLGC_EINVAL, "bad mode");
    if ((rc = launch_triangulate_maps(ctx, d_h, d_v, rows, W, row0, proj_w, proj_h, mode, d_xyz, d_count, d_wire))) return rc;
    ctx->last_ragged = 0;
    ctx->last_scan_path = ctx->last_tri_ragged ? SLGC_PATH_SPLIT_RAGGED : SLGC_PATH_SPLIT;      // the maps came over the wire: no decode of this context is part of it
    ctx->decode_pending = 0;
    return SLGC_OK;
}

extern "C" int slgc_scan_dev(slgc_ctx *ctx, const uint8_t *d_stack, int n_runs, size_t run_stride, size_t plane_stride, int N,
                             int rows, int W, int row0, int proj_w, int proj_h, double eps, double m, int mode, int16_t *d_h,
                             int16_t *d_v, float *d_xyz, unsigned long long *d_count)
{
    (void)m;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if (!d_xyz) return slgc_fail(ctx, SLGC_EINVAL, "null output");
    if (mode < 0 || mode > 7) return slgc_fail(ctx, SLGC_EINVAL, "bad mode");
    const bool want_split = (mode & SLGC_TRI_SPLIT) != 0;
    mode &= 3;
    DecodeGeom g;
    RunPtrs runs{};
    int e;
    TileScope tile_scope(ctx);
    if ((rc = dev_geom(ctx, d_stack, n_runs, run_stride, plane_stride, N, rows, W, eps, &g, &runs, &e))) return rc;
    if (!d_h || !d_v) d_h = d_v = nullptr;          // no map buffers: the fused kernel then stores XYZ only
    const size_t npix = (size_t)rows * W;
    if (mode == SLGC_TRI_ALGEBRAIC && !d_count && npix % 4 == 0 && scan_fused_eligible(g, runs, plane_stride, npix, d_h, d_v, d_xyz) &&
        proj_w >= 1 && proj_h >= 1 && (size_t)proj_w * proj_h < (1u << 28) && !want_split) {
        // one kernel: decode with the triangulation tail (maps never re-read from HBM)
        if ((rc = ensure_luts(ctx, rows, W, row0, proj_w, proj_h))) return rc;
        if ((rc = prof_mark(ctx, 0))) return rc;
        if ((rc = launch_scan_fused(ctx, g, runs, plane_stride, npix, e, d_h, d_v, ctx->lut_cam, ctx->lut_proj, d_xyz, proj_w, proj_h))) return rc;
        ctx->last_scan_path = SLGC_PATH_FUSED;
        ctx->decode_pending = 0;
        return prof_mark(ctx, 1);
    }
    ctx->last_scan_path = SLGC_PATH_NONE;
    ctx->last_ragged = 0;
    ctx->decode_pending = 0;
    if (!d_h) {                                     // the two-kernel path hands the maps over through HBM: scratch of the context
        void *maps;
        if ((rc = slgc_ws(ctx, 3, (size_t)rows * W * 4 + 64, &maps))) return rc;
        d_h = (int16_t *)maps;
        d_v = d_h + (((size_t)rows * W + 31) & ~(size_t)31);
    }
    if ((rc = decode_fast_timed(ctx, g, runs, plane_stride, rows, W, e, d_h, d_v, 0))) return rc;
    if ((rc = launch_triangulate_maps(ctx, d_h, d_v, rows, W, row0, proj_w, proj_h, mode, d_xyz, d_count))) return rc;
    ctx->last_scan_path = (ctx->last_ragged || ctx->last_tri_ragged) ? SLGC_PATH_SPLIT_RAGGED : SLGC_PATH_SPLIT;
    return SLGC_OK;
}

// The scan straight from the camera's BGR frames (src/3-capture_decode.py:66-70: cv2.cvtColor(BGR2GRAY) into the grey stack, :75 get_codes on it):
// d_bgr = [n_runs][N][rows][W][3] uint8, plane_stride = bytes between consecutive frames (>= 3 * rows * W).  One kernel when the shape allows
// (N = 42 / 44 / 46 / 50 / 54, 4-byte aligned planes, algebraic mode, no count): the luma is formed in registers inside the frame loads and the grey stack
// never exists in HBM -- 3 N + 12 bytes per pixel instead of 3 N + N (written) + N (read back) + 12.  Any other shape: slgc_to_gray_dev into
// scratch of the context, then slgc_scan_dev.  Maps and XYZ are bit-identical either way (same luma, same kernels behind it).
extern "C" int slgc_scan_bgr_dev(slgc_ctx *ctx, const uint8_t *d_bgr, int n_runs, size_t run_stride, size_t plane_stride, int N, int rows, int W, int row0,
                                 int proj_w, int proj_h, int coeff_bits, double eps, double m, int mode, int16_t *d_h, int16_t *d_v, float *d_xyz,
                                 unsigned long long *d_count)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    TileOff tile_off(ctx);                       // BGR frames are planar [N][rows][W][3]; so is the grey scratch of the fallback path
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if (!d_bgr || !d_xyz) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (coeff_bits != 14 && coeff_bits != 15) return slgc_fail(ctx, SLGC_EINVAL, "coeff_bits must be 15 (OpenCV 4.x) or 14");
    if (mode < 0 || mode > 7) return slgc_fail(ctx, SLGC_EINVAL, "bad mode");
    if (rows < 0 || W < 0 || plane_stride < 3 * (size_t)rows * W) return slgc_fail(ctx, SLGC_EINVAL, "plane_stride smaller than a BGR band");
    DecodeGeom g;
    RunPtrs runs{};
    int e;
    if (slgc_make_geom(N, n_runs, &g)) return slgc_fail(ctx, SLGC_EINVAL, "unsupported N=%d / n_runs=%d", N, n_runs);
    if (!decode_fast_eligible(eps, &e)) return slgc_fail(ctx, SLGC_EINVAL, "device-resident scans need an integer eps in [0,255] (got %g)", eps);
    for (int r = 0; r < n_runs; ++r) runs.p[r] = d_bgr + (size_t)r * run_stride;
    if (!d_h || !d_v) d_h = d_v = nullptr;
    const size_t npix = (size_t)rows * W;
    if ((mode & 3) == SLGC_TRI_ALGEBRAIC && !(mode & SLGC_TRI_SPLIT) && !d_count && proj_w >= 1 && proj_h >= 1 && (size_t)proj_w * proj_h < (1u << 28) &&
        scan_bgr_eligible(ctx, g, runs, plane_stride, npix, d_h, d_v, d_xyz)) {
        if ((rc = ensure_luts(ctx, rows, W, row0, proj_w, proj_h))) return rc;
        if ((rc = prof_mark(ctx, 0))) return rc;
        if ((rc = launch_scan_fused(ctx, g, runs, plane_stride, npix, e, d_h, d_v, ctx->lut_cam, ctx->lut_proj, d_xyz, proj_w, proj_h, 1, 0, coeff_bits))) return rc;
        ctx->last_scan_path = SLGC_PATH_FUSED_BGR;
        ctx->decode_pending = 0;
        return prof_mark(ctx, 1);
    }
    // grey stack in scratch, frames packed back to back, then the ordinary scan
    void *gray;
    const size_t gplane = (npix + 15) & ~(size_t)15;
    if ((rc = slgc_ws(ctx, 11, (size_t)n_runs * N * gplane + 64, &gray))) return rc;
    for (int r = 0; r < n_runs; ++r) {
        if (plane_stride == 3 * npix && gplane == npix) {          // frames back to back on both sides: the whole run in one launch
            if ((rc = launch_bgr_to_gray(ctx, (const uint8_t *)runs.p[r], (uint8_t *)gray + (size_t)r * N * gplane, (size_t)N * npix, coeff_bits))) return rc;
            continue;
        }
        for (int f = 0; f < N; ++f)
            if ((rc = launch_bgr_to_gray(ctx, (const uint8_t *)runs.p[r] + (size_t)f * plane_stride, (uint8_t *)gray + ((size_t)r * N + f) * gplane, npix, coeff_bits)))
                return rc;
    }
    return slgc_scan_dev(ctx, (const uint8_t *)gray, n_runs, (size_t)N * gplane, gplane, N, rows, W, row0, proj_w, proj_h, eps, m, mode, d_h, d_v, d_xyz, d_count);
}

// slgc_decode_dev straight from BGR frames (the maps alone: what slgc_cloud_lists_dev / slgc_triangulate_maps_dev take).  Same shapes, same fall-back as slgc_scan_bgr_dev.
extern "C" int slgc_decode_bgr_dev(slgc_ctx *ctx, const uint8_t *d_bgr, int n_runs, size_t run_stride, size_t plane_stride, int N, int rows, int W, int coeff_bits,
                                   double eps, double m, int16_t *d_h, int16_t *d_v)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    TileOff tile_off(ctx);                       // BGR frames are planar [N][rows][W][3]; so is the grey scratch of the fallback path
    if (!d_bgr || !d_h || !d_v) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (coeff_bits != 14 && coeff_bits != 15) return slgc_fail(ctx, SLGC_EINVAL, "coeff_bits must be 15 (OpenCV 4.x) or 14");
    if (rows < 0 || W < 0 || plane_stride < 3 * (size_t)rows * W) return slgc_fail(ctx, SLGC_EINVAL, "plane_stride smaller than a BGR band");
    DecodeGeom g;
    RunPtrs runs{};
    int e;
    if (slgc_make_geom(N, n_runs, &g)) return slgc_fail(ctx, SLGC_EINVAL, "unsupported N=%d / n_runs=%d", N, n_runs);
    if (!decode_fast_eligible(eps, &e)) return slgc_fail(ctx, SLGC_EINVAL, "device-resident decode needs an integer eps in [0,255] (got %g)", eps);
    for (int r = 0; r < n_runs; ++r) runs.p[r] = d_bgr + (size_t)r * run_stride;
    const size_t npix = (size_t)rows * W;
    ctx->last_scan_path = SLGC_PATH_NONE;
    ctx->last_ragged = 0;
    ctx->last_tri_ragged = 0;
    ctx->decode_pending = 1;
    if (scan_bgr_eligible(ctx, g, runs, plane_stride, npix, d_h, d_v, (const float *)nullptr)) {
        if ((rc = prof_mark(ctx, 0))) return rc;
        if ((rc = launch_decode_bgr(ctx, g, runs, plane_stride, npix, e, d_h, d_v, coeff_bits))) return rc;
        return prof_mark(ctx, 1);
    }
    void *gray;
    const size_t gplane = (npix + 15) & ~(size_t)15;
    if ((rc = slgc_ws(ctx, 11, (size_t)n_runs * N * gplane + 64, &gray))) return rc;
    for (int r = 0; r < n_runs; ++r) {
        if (plane_stride == 3 * npix && gplane == npix) {
            if ((rc = launch_bgr_to_gray(ctx, (const uint8_t *)runs.p[r], (uint8_t *)gray + (size_t)r * N * gplane, (size_t)N * npix, coeff_bits))) return rc;
            continue;
        }
        for (int f = 0; f < N; ++f)
            if ((rc = launch_bgr_to_gray(ctx, (const uint8_t *)runs.p[r] + (size_t)f * plane_stride, (uint8_t *)gray + ((size_t)r * N + f) * gplane, npix, coeff_bits)))
                return rc;
    }
    return slgc_decode_dev(ctx, (const uint8_t *)gray, n_runs, (size_t)N * gplane, gplane, N, rows, W, eps, m, d_h, d_v, 0);
}

// Throughput mode (BASELINE configs[4]): n_scans independent single-run scans of one geometry, their stacks scan_stride bytes apart, in ONE
// launch of the fused kernel when every scan is a whole number of its 512-pixel workgroups -- a 1920x1080 scan alone is one round of
// resident waves (all of them in the same phase of the kernel at the same time); sixteen of them overlap like a large image does.
extern "C" int slgc_scan_batch_dev(slgc_ctx *ctx, const uint8_t *d_stacks, int n_scans, size_t scan_stride, size_t plane_stride, int N, int rows,
                                   int W, int row0, int proj_w, int proj_h, double eps, double m, int mode, int16_t *d_h, int16_t *d_v, float *d_xyz)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    TileScope tile_scope(ctx);
    if (n_scans < 0 || !d_stacks || !d_xyz) return slgc_fail(ctx, SLGC_EINVAL, "null pointer / negative count");
    if (!d_h || !d_v) d_h = d_v = nullptr;          // XYZ only
    const size_t npix = (size_t)rows * W;
    DecodeGeom g;
    RunPtrs runs{};
    int e;
    const int tri = mode & 3;
    if (n_scans > 1 && tri == SLGC_TRI_ALGEBRAIC && !(mode & SLGC_TRI_SPLIT) && ctx->have_calib && npix % 512 == 0 && npix &&
        (uint64_t)(npix / 512) * (uint64_t)n_scans * (uint64_t)(npix / 512) < 0x100000000ull && proj_w >= 1 && proj_h >= 1 &&
        (size_t)proj_w * proj_h < (1u << 28) && scan_stride % 4 == 0 && !dev_geom(ctx, d_stacks, 1, 0, plane_stride, N, rows, W, eps, &g, &runs, &e) &&
        scan_fused_eligible(g, runs, plane_stride, npix, d_h, d_v, d_xyz)) {
        if ((rc = ensure_luts(ctx, rows, W, row0, proj_w, proj_h))) return rc;
        if ((rc = prof_mark(ctx, 0))) return rc;
        if ((rc = launch_scan_fused(ctx, g, runs, plane_stride, npix, e, d_h, d_v, ctx->lut_cam, ctx->lut_proj, d_xyz, proj_w, proj_h, n_scans, scan_stride)))
            return rc;
        ctx->last_scan_path = SLGC_PATH_BATCH_FUSED;
        return prof_mark(ctx, 1);
    }
    for (int s = 0; s < n_scans; ++s)                                     // any other shape / mode: scan after scan
        if ((rc = slgc_scan_dev(ctx, d_stacks + (size_t)s * scan_stride, 1, 0, plane_stride, N, rows, W, row0, proj_w, proj_h, eps, m, mode, d_h ? d_h + s * npix : nullptr,
                                d_v ? d_v + s * npix : nullptr, d_xyz + 3 * s * npix, nullptr)))
            return rc;
    return SLGC_OK;
}

extern "C" int slgc_pack_hv24_dev(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, size_t npix, int code_bits, uint8_t *d_wire)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_h || !d_v || !d_wire) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (code_bits < 1 || code_bits > SLGC_WIRE_MAX_CODE_BITS)
        return slgc_fail(ctx, SLGC_EINVAL, "the 3-byte wire format holds codes of at most %d bits (got %d)", SLGC_WIRE_MAX_CODE_BITS, code_bits);
    return launch_pack_hv24(ctx, d_h, d_v, npix, d_wire);
}

extern "C" int slgc_unpack_hv24_dev(slgc_ctx *ctx, const uint8_t *d_wire, size_t npix, int16_t *d_h, int16_t *d_v)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_h || !d_v || !d_wire) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    return launch_unpack_hv24(ctx, d_wire, npix, d_h, d_v);
}

extern "C" int slgc_cloud_lists_dev(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, const float *d_xyz, const uint8_t *d_white_rgb,
                                    int cam_w, int cam_h, int proj_w, int proj_h, float *d_cam_pts, float *d_proj_pts, double *d_pts,
                                    double *d_colors, unsigned long long *d_total)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_h || !d_v || !d_cam_pts || !d_proj_pts || !d_total) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if (d_xyz && !d_pts) return slgc_fail(ctx, SLGC_EINVAL, "d_xyz without d_pts");
    if (d_colors && !d_white_rgb) return slgc_fail(ctx, SLGC_EINVAL, "colours need the white image");
    if (cam_w < 0 || cam_h < 0 || proj_w < 1 || proj_h < 1) return slgc_fail(ctx, SLGC_EINVAL, "bad size");
    if (d_pts && !d_xyz) {          // points wanted, no dense XYZ given: triangulate every valid pixel inside the list build (slgc_cloud_dev's second half)
        if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
        if ((size_t)proj_w * proj_h >= (1u << 28)) return slgc_fail(ctx, SLGC_EINVAL, "bad projector size");
        if ((rc = ensure_luts(ctx, cam_h, cam_w, 0, proj_w, proj_h))) return rc;
        return launch_cloud_tri(ctx, d_h, d_v, d_colors ? d_white_rgb : nullptr, cam_w, cam_h, proj_w, proj_h, d_cam_pts, d_proj_pts, d_pts, d_colors, d_total);
    }
    return launch_cloud_lists(ctx, d_h, d_v, d_xyz, d_colors ? d_white_rgb : nullptr, cam_w, cam_h, proj_w, proj_h, d_cam_pts, d_proj_pts, d_pts,
                              d_colors, d_total);
}

static int cloud_dev_common(slgc_ctx *ctx, const uint8_t *d_stack, int n_runs, size_t run_stride, size_t plane_stride, int N, int cam_h, int cam_w, int proj_w, int proj_h,
                            double eps, const uint8_t *d_white_rgb, int16_t *d_h, int16_t *d_v, float *d_cam_pts, float *d_proj_pts, void *d_pts, void *d_colors,
                            unsigned long long *d_total, int f32)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if (!d_total) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    if ((d_cam_pts == nullptr) != (d_proj_pts == nullptr)) return slgc_fail(ctx, SLGC_EINVAL, "d_cam_pts and d_proj_pts go together");
    if (!d_cam_pts && !d_pts) return slgc_fail(ctx, SLGC_EINVAL, "nothing to produce: neither the correspondence lists nor the points are wanted");
    if (f32 && !d_pts) return slgc_fail(ctx, SLGC_EINVAL, "the float32 product is points (+ colours): d_pts32 is required");
    if (d_colors && !d_white_rgb) return slgc_fail(ctx, SLGC_EINVAL, "colours need the white image");
    if ((d_h == nullptr) != (d_v == nullptr)) return slgc_fail(ctx, SLGC_EINVAL, "d_h and d_v go together");
    if (proj_w < 1 || proj_h < 1 || (size_t)proj_w * proj_h >= (1u << 28)) return slgc_fail(ctx, SLGC_EINVAL, "bad projector size");
    DecodeGeom g;
    RunPtrs runs{};
    int e;
    TileScope tile_scope(ctx);
    if ((rc = dev_geom(ctx, d_stack, n_runs, run_stride, plane_stride, N, cam_h, cam_w, eps, &g, &runs, &e))) return rc;
    if (!d_h) {
        void *maps;
        if ((rc = slgc_ws(ctx, 3, (size_t)cam_h * cam_w * 4 + 64, &maps))) return rc;
        d_h = (int16_t *)maps;
        d_v = d_h + (((size_t)cam_h * cam_w + 31) & ~(size_t)31);
    }
    if ((rc = ensure_luts(ctx, cam_h, cam_w, 0, proj_w, proj_h))) return rc;
    ctx->last_ragged = 0;
    ctx->last_tri_ragged = 0;
    if ((rc = decode_fast_timed(ctx, g, runs, plane_stride, cam_h, cam_w, e, d_h, d_v, 0))) return rc;
    if ((rc = launch_cloud_tri(ctx, d_h, d_v, d_colors ? d_white_rgb : nullptr, cam_w, cam_h, proj_w, proj_h, d_cam_pts, d_proj_pts, (double *)d_pts, (double *)d_colors, d_total,
                               f32)))
        return rc;
    ctx->last_scan_path = SLGC_PATH_CLOUD;      // (slgc_last_scan_ragged tells whether the byte-wide decode kernel took part)
    ctx->decode_pending = 0;
    return SLGC_OK;
}

extern "C" int slgc_cloud_dev(slgc_ctx *ctx, const uint8_t *d_stack, int n_runs, size_t run_stride, size_t plane_stride, int N, int cam_h, int cam_w,
                              int proj_w, int proj_h, double eps, double m, const uint8_t *d_white_rgb, int16_t *d_h, int16_t *d_v, float *d_cam_pts,
                              float *d_proj_pts, double *d_pts, double *d_colors, unsigned long long *d_total)
{
    (void)m;
    return cloud_dev_common(ctx, d_stack, n_runs, run_stride, plane_stride, N, cam_h, cam_w, proj_w, proj_h, eps, d_white_rgb, d_h, d_v, d_cam_pts, d_proj_pts, d_pts,
                            d_colors, d_total, 0);
}

// NOT the reference's dtypes: the same scan with the points as float32 (3,M) and the colours as float32 [M][3] -- the values of slgc_cloud_dev rounded to
// float32 (the points ARE float32 before the reference's API widens them; a colour is b / 255 rounded once more) -- for callers that do not need
// float64: 40 instead of 64 bytes written per point with the correspondence lists, 24 instead of 48 without.
extern "C" int slgc_cloud32_dev(slgc_ctx *ctx, const uint8_t *d_stack, int n_runs, size_t run_stride, size_t plane_stride, int N, int cam_h, int cam_w,
                                int proj_w, int proj_h, double eps, double m, const uint8_t *d_white_rgb, int16_t *d_h, int16_t *d_v, float *d_cam_pts,
                                float *d_proj_pts, float *d_pts32, float *d_colors32, unsigned long long *d_total)
{
    (void)m;
    return cloud_dev_common(ctx, d_stack, n_runs, run_stride, plane_stride, N, cam_h, cam_w, proj_w, proj_h, eps, d_white_rgb, d_h, d_v, d_cam_pts, d_proj_pts, d_pts32,
                            d_colors32, d_total, 1);
}

extern "C" int slgc_compact_dev(slgc_ctx *ctx, const float *d_xyz, int rows, int W, int row0, float *d_points, uint32_t *d_keys,
                                unsigned long long *d_count)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_xyz || !d_points || !d_count) return slgc_fail(ctx, SLGC_EINVAL, "null pointer");
    return launch_compact_dense(ctx, d_xyz, rows, W, row0, d_points, d_keys, d_count);
}

extern "C" int slgc_compact_records_dev(slgc_ctx *ctx, const float *d_xyz, int rows, int W, int row0, void *d_records,
                                        unsigned long long *d_count)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_xyz || !d_records || !d_count || ((uintptr_t)d_records & 15)) return slgc_fail(ctx, SLGC_EINVAL, "null / misaligned pointer");
    return launch_compact_records(ctx, d_xyz, rows, W, row0, d_records, d_count);
}

extern "C" int slgc_move_only_dev(slgc_ctx *ctx, const uint8_t *d_stack, size_t plane_stride, int N, size_t npix, int16_t *d_h, int16_t *d_v, float *d_xyz)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_stack || (d_h == nullptr) != (d_v == nullptr)) return slgc_fail(ctx, SLGC_EINVAL, "null stack / d_h and d_v go together");
    if (N != 42 && N != 44 && N != 46) return slgc_fail(ctx, SLGC_EINVAL, "the movement yardstick is built for 42, 44 and 46 frames");
    if (npix % 256 || plane_stride % 4 || plane_stride < npix || ((uintptr_t)d_stack | (uintptr_t)d_h | (uintptr_t)d_v) % 8 || (uintptr_t)d_xyz % 16 ||
        npix >= ((size_t)1 << 32))
        return slgc_fail(ctx, SLGC_EINVAL, "npix must be a multiple of 256, planes 4-byte, maps 8-byte and XYZ 16-byte aligned");
    return launch_move_only(ctx, d_stack, plane_stride, N, npix, d_h, d_v, d_xyz);
}

extern "C" int slgc_synth_scene_dev(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows,
                                    uint32_t seed, int noise, int shadow)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_stack || N < 14 || N > 65 || rows < 0 || row0 < 0 || row0 + rows > H || plane_stride < (size_t)rows * W)
        return slgc_fail(ctx, SLGC_EINVAL, "bad synth arguments");
    return launch_synth(ctx, d_stack, plane_stride, N, H, W, row0, rows, seed, noise, shadow);
}

// The physical generator with its knobs exposed: the two surface gains of the 16-pixel checker (frames = 15 + gain * bit + noise), and
// r2_max, the radius (squared, normalised projector coordinates) up to which the projector's lens model is trusted to be monotonic.
extern "C" int slgc_synth_physical_ex_dev(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows, int proj_w,
                                          int proj_h, uint32_t seed, int noise, int gain_lo, int gain_hi, double r2_max, int16_t *d_h_true,
                                          int16_t *d_v_true, float *d_truth_xyz)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->have_calib) return slgc_fail(ctx, SLGC_ESTATE, "slgc_set_calibration has not been called");
    if (N < 14 || N > 65 || rows < 0 || W < 0 || row0 < 0 || row0 + rows > H || (d_stack && plane_stride < (size_t)rows * W) || proj_w < 1 || proj_h < 1 ||
        proj_w > 32767 || proj_h > 32767 || (d_h_true == nullptr) != (d_v_true == nullptr) || noise < 0 || gain_lo < 0 || gain_hi < 0 || gain_lo > 240 ||
        gain_hi > 240 || !(r2_max > 0.0))
        return slgc_fail(ctx, SLGC_EINVAL, "bad synth arguments");
    if (!d_h_true) {
        void *codes;
        const size_t npix = (size_t)rows * W;
        if ((rc = slgc_ws(ctx, 2, npix * 4 + 128, &codes))) return rc;
        d_h_true = (int16_t *)codes;
        d_v_true = d_h_true + ((npix + 31) & ~(size_t)31);
    }
    return launch_synth_physical(ctx, d_stack, plane_stride, N, H, W, row0, rows, proj_w, proj_h, seed, noise, gain_lo, gain_hi, r2_max, d_h_true, d_v_true,
                                 d_truth_xyz);
}

extern "C" int slgc_synth_physical_dev(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows, int proj_w,
                                       int proj_h, uint32_t seed, int noise, int16_t *d_h_true, int16_t *d_v_true, float *d_truth_xyz)
{
    return slgc_synth_physical_ex_dev(ctx, d_stack, plane_stride, N, H, W, row0, rows, proj_w, proj_h, seed, noise, 140, 180, 0.16, d_h_true, d_v_true,
                                      d_truth_xyz);
}

extern "C" int slgc_synth_bgr_dev(slgc_ctx *ctx, const uint8_t *d_gray, size_t gray_plane_stride, int N, int H, int W, int row0, int rows, uint8_t *d_bgr,
                                  size_t bgr_plane_stride)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_gray || !d_bgr || N < 1 || rows < 0 || W < 0 || row0 < 0 || row0 + rows > H || gray_plane_stride < (size_t)rows * W || bgr_plane_stride < 3 * (size_t)rows * W)
        return slgc_fail(ctx, SLGC_EINVAL, "bad synth arguments");
    return launch_synth_bgr(ctx, d_gray, gray_plane_stride, N, W, row0, rows, d_bgr, bgr_plane_stride);
}

extern "C" int slgc_synth_uniform_dev(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows, uint32_t seed)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_stack || N < 1 || rows < 0 || W < 0 || W % 4 || row0 < 0 || row0 + rows > H || plane_stride < (size_t)rows * W || plane_stride % 4 ||
        (uintptr_t)d_stack % 4 || (size_t)H * W / 4 >= ((size_t)1 << 32))
        return slgc_fail(ctx, SLGC_EINVAL, "bad synth arguments (W, plane_stride and the stack pointer must be multiples of 4)");
    return launch_synth_uniform(ctx, d_stack, plane_stride, N, W, row0, rows, seed);
}

extern "C" int slgc_event_record(slgc_ctx *ctx, int id)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (id < 0 || id >= SLGC_MAX_EVENTS) return slgc_fail(ctx, SLGC_EINVAL, "event id out of range");
    HIP_TRY(ctx, hipEventRecord(ctx->events[id], ctx->stream));
    return SLGC_OK;
}

extern "C" int slgc_event_elapsed_ms(slgc_ctx *ctx, int id_start, int id_stop, float *ms)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (id_start < 0 || id_start >= SLGC_MAX_EVENTS || id_stop < 0 || id_stop >= SLGC_MAX_EVENTS || !ms)
        return slgc_fail(ctx, SLGC_EINVAL, "bad event arguments");
    HIP_TRY(ctx, hipEventSynchronize(ctx->events[id_stop]));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->events[id_start], ctx->events[id_stop]));
    return SLGC_OK;
}
