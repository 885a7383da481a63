// Internal declarations shared by the HIP translation units of libslgc.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "slgc_bench.h"      // slgc.h (the reference-facing ABI) + the measurement / diagnostic exports

#define SLGC_MAX_RUNS 8
#define SLGC_MAX_EVENTS 16
#define SLGC_WS_SLOTS 14

// Frame bookkeeping of one decode call; the index arithmetic follows the reference exactly
// (decode_codes.py:109-111 float pattern_len + uint8 truncation; :149 int pattern_len).
struct DecodeGeom {
    int N;        // frames per run
    int L;        // code length int((N-2)/4)
    int hid[6];   // absolute frame indices (already +2) feeding L_max
    int vid[6];   // absolute frame indices feeding L_min
    int n_runs;
};

struct RunPtrs {
    const void *p[SLGC_MAX_RUNS];
};

// Calibration block in constant form for kernels (filled by slgc_set_calibration).
struct Calib {
    double cam_k[4];   // fx, fy, cx, cy
    double cam_d[12];  // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    double proj_k[4];
    double proj_d[12];
    double R[9];
    double T[3];
    double t_len;
};

struct slgc_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t events[SLGC_MAX_EVENTS];
    char err[512];
    // workspace (grown on demand, reused across calls)
    void *ws[SLGC_WS_SLOTS];
    size_t ws_bytes[SLGC_WS_SLOTS];
    unsigned ws_gen[SLGC_WS_SLOTS];   // bumped every time a slot is handed out (slgc_ws): a *_fetch checks that the slots holding its
                                      // pending result have not been handed to another call since its *_count (SLGC_ESTATE otherwise)
    // tuning knobs (slgc_tune): every setting gives the same results, they exist for same-process A/B timing
    int tune_fuse_tail;     // fused scan tail: 1 = wave-local LDS exchange (no workgroup barriers), 0 = workgroup-wide exchange
    int tune_proj_tile;     // projector ray table tiles: 0 = 8x8 pixels (512 B), 1 = 16x8 pixels (one 128-byte line per tile row)
    int tune_fuse_nt;       // fused scan: bit 0 XYZ, bit 1 maps leave with non-temporal stores; -1 (default) = 1 from 4 Mpixels per launch up, 3 below
    int tune_tri_nt;        // dense triangulation kernel: XYZ with non-temporal stores
    int tune_xcd;           // dense triangulation kernel: XCD-aware workgroup -> tile map
    int tune_fuse_xcd;      // the same map for the fused scan kernel
    int tune_prio;          // fused scan kernel: s_setprio per phase, head * 100 + body * 10 + tail (decode.hip: set_prio)
    int tune_lists_order;   // x-major scatter: workgroup -> tile order (correspond.hip): 0 row-major, 1 column-major, 2 column-major inside each XCD
    int tune_lists_lines;   // slgc_cloud_dev's scatter: 1 = k_xmajor_lines (whole 128-byte lines) for images of >= 2048 tiles (default), 2 = wherever the shape allows, 0 = k_xmajor_scatter<.., 2, ..>
    int tune_stack_tile;    // slgc_tune "stack_tile_log2": k > 0 = the _dev scan entry points read their stacks tile-interleaved, [tile][N][2^k bytes] (0 = planar [N][H][W], default)
    int stack_tile_active;  // the layout of the stack the decode launchers are about to read: set by every _dev entry (dev_geom), cleared by every host-buffer upload
    int tune_wire;          // slgc_scan_sharded_dev: 1 = exchange the maps in the 3-byte wire format, 0 = int16 (default)
    int tune_cam_nodes;     // scan kernels' camera rays: 0 per-pixel table, 1 node table when the per-pixel one would stream from HBM (default), 2 node table whenever accurate
    int tune_image_rows;    // height of the whole image a band belongs to (0 = the band IS the image): the node-table decision (size and accuracy) is taken
                            // for the whole image, so a pixel gets the same rays -- and bit-identical XYZ -- whether one GPU scans the image or N GPUs its bands
    // slgc_last_scan_path: what the last slgc_scan_dev / slgc_scan_batch_dev / slgc_decode_dev / slgc_triangulate_maps_dev call launched (written by the launchers)
    int last_scan_path;     // SLGC_PATH_*
    int last_ns;            // frames-per-run specialisation of the decode / fused kernel (42 / 44 / 46 / 50 / 54), 0 = generic kernel
    int last_nodes;         // 1 = the triangulation read the camera node table, 0 = the per-pixel table (or evaluated the rays per pixel)
    int last_guard;         // 1 = float32 fast form with the flat-triangle guard, 0 = exact (acos / sin) mode, -1 = unguarded (diagnostic build only)
    int last_list_kernel;   // SLGC_LISTS_*: which scatter the last x-major list build launched
    int last_ragged;        // 1 = a byte-wide fallback DECODE kernel took part in the last decode launch (misaligned buffers, ragged tails)
    int last_tri_ragged;    // 1 = the per-pixel fallback TRIANGULATION kernel took part in the last dense triangulation
    int decode_pending;     // 1 = the last scan-related call was slgc_decode_dev: the slgc_triangulate_maps_dev / _wire_dev that follows completes a two-kernel
                            // scan and inherits its raggedness; a triangulation on its own reports only its own
    int tune_guard_list;    // fused scan: 1 = flat triangles compacted over the wave and redone 64 per pass (default), 0 = redone lane by lane
    int tune_park;          // decode / fused kernels at N = 42, 44, 46, 50, 54: park the 12 threshold frames in LDS instead of fetching them twice
    int tune_fuse_abl;      // diagnostic build only: timing-only ablations of the fused kernel (wrong results)
    void *dl_stage;         // pinned ring the large device-to-host results land in (api.hip: download_par)
    hipEvent_t dl_ev[4];
    void *stage;            // pinned host staging (float64 stacks narrowed to uint8 before the upload)
    size_t stage_bytes;
    int last_input_path;    // slgc_last_input_path
    size_t stamp_waves;     // stamp build only (SLGC_STAMPS): waves of the last fused launch
    // calibration
    bool have_calib;
    Calib calib;
    unsigned calib_ver;
    // ray tables (triangulate.hip), rebuilt when the calibration or the geometry changes
    void *lut_cam, *lut_proj;
    void *lut_proj_th;      // second part of the projector table's allocation: tan(beta / 2) per projector pixel, float32 (tri_math.h fast form)
    void *lut_nodes;        // camera rays at every 4th column (tri_math.h CamNodes), nullptr when not built / not accurate enough
    float lut_nodes_err;    // error measure of k_check_cam_nodes over the band -- or, with tune_image_rows, over the whole image (-1: not measured)
    int lut_image_rows;     // tune_image_rows the tables were built under
    void *lut_check_word;   // 4-byte device word the node-table check reduces into
    float img_err;          // whole-image error measure, cached per (calibration, W, image_rows): every band of one image reuses it
    unsigned img_err_ver;
    int img_err_W, img_err_rows;
    void *count_slots;  // hashed valid-pixel counters (triangulate.hip)
    unsigned lut_cam_ver, lut_proj_ver;
    int lut_cam_W, lut_cam_row0, lut_cam_rows, lut_proj_w, lut_proj_h, lut_proj_tile;
    // results kept on the device between *_count and *_fetch
    int64_t pend_M;
    size_t pend_npix;
    bool pend_colors;
    unsigned pend_gen;      // ws_gen[6] when the correspondence lists were written
    int64_t filt_M;
    bool filt_colors;
    unsigned filt_gen;      // ws_gen[6] when the filter result was written
    // slgc_pipeline_* results kept on the device between _count and _fetch
    int64_t pipe_M, pipe_M_raw;
    size_t pipe_npix;
    bool pipe_colors, pipe_filtered;
    unsigned pipe_gen[4];   // ws_gen of slots 3, 8, 9, 10 when the pipeline result was written
    // per-launch HIP-event timing of the decode kernel (slgc_prof_*), recorded on the launch stream
    bool prof_on;
    hipEvent_t *prof_ev;   // pairs: [2i] before, [2i+1] after the decode launch
    int prof_cap, prof_n;
    int prof_stride, prof_seen;   // every prof_stride-th launch is bracketed (an event pair costs ~1.5 % of a 160 us kernel)
    bool prof_sampling;
    hipEvent_t prof_cur[2];       // events of the launch being sampled (null otherwise): bound to the kernel's own dispatch by SLGC_LAUNCH
    bool prof_bound;              // the launcher used them (hipExtLaunchKernelGGL) -> no hipEventRecord needed after the launch
    // communicator (comm.cpp)
    void *comm;
    int rank, nranks;
    void *comm_scratch;  // device scratch for small collectives
    hipStream_t comm_stream;          // every collective runs here, ordered against ctx->stream with events
    hipEvent_t ev_compute;            // "compute stream reached this point" (recorded before a collective is enqueued)
    hipEvent_t ev_comm_done[4];       // completion of the collective started in slot s (slgc_comm_allgatherv_begin / _wait)
    void *direct;                     // direct.hip: state of the all-links band exchange (slgc_direct_*), nullptr until slgc_direct_init
};

int slgc_fail(slgc_ctx *ctx, int status, const char *fmt, ...);
int slgc_ws(slgc_ctx *ctx, int slot, size_t bytes, void **out);
int slgc_make_geom(int N, int n_runs, DecodeGeom *g);

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess) return slgc_fail((ctx), SLGC_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// ---- launchers (each enqueues on ctx->stream; no sync) ----
// decode.hip
int launch_decode_fast(slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, int rows, int W, int e,
                       int16_t *d_h, int16_t *d_v, int variant);
int launch_decode_generic(slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, int dtype, size_t plane_stride_elems,
                          size_t npix, double eps, const double *d_Ld_in, const double *d_Lg_in, double *d_Ld_out,
                          double *d_Lg_out, int8_t *d_hc, int8_t *d_vc, int16_t *d_h16, int16_t *d_v16, int64_t *d_h64,
                          int64_t *d_v64);
int launch_codes_to_pixels(slgc_ctx *ctx, const int8_t *d_hc, const int8_t *d_vc, int n_runs, int L, size_t npix,
                           int64_t *d_h, int64_t *d_v);
bool decode_fast_eligible(double eps, int *e_out);
int launch_selftest_thresholds(slgc_ctx *ctx, int e, int black0, int n_black, unsigned long long *d_bad, int skew);
int launch_selftest_classify(slgc_ctx *ctx, unsigned long long *d_bad, int skew);
int launch_scan_fused(slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, size_t npix4, int e, int16_t *d_h,
                      int16_t *d_v, const void *cam_lut, const void *proj_lut, float *d_xyz, int proj_w, int proj_h, int n_batch = 1, size_t batch_stride = 0,
                      int bgr_bits = 0);
int launch_decode_bgr(slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, size_t npix4, int e, int16_t *d_h, int16_t *d_v, int bgr_bits);
bool scan_bgr_eligible(const slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, size_t npix, const int16_t *d_h, const int16_t *d_v,
                       const float *d_xyz);
inline int proj_tiles_x(const slgc_ctx *ctx, int proj_w) { return ctx->tune_proj_tile ? (proj_w + 15) / 16 : (proj_w + 7) / 8; }
bool scan_fused_eligible(const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, size_t npix, const int16_t *d_h, const int16_t *d_v,
                         const float *d_xyz);
int ensure_luts(slgc_ctx *ctx, int rows, int W, int row0, int proj_w, int proj_h);
int slgc_direct_check(slgc_ctx *ctx);                              // direct.hip: SLGC_ECOMM once a GPU-side wait of the direct exchange has timed out (OK without one)
int slgc_direct_is_registered(slgc_ctx *ctx, const void *d_base);  // direct.hip
int launch_widen_maps(slgc_ctx *ctx, const int16_t *d_h16, const int16_t *d_v16, size_t npix, int64_t *d_h, int64_t *d_v);
// correspond.hip
int launch_correspond(slgc_ctx *ctx, const int64_t *d_h, const int64_t *d_v, int cam_w, int cam_h, int proj_w, int proj_h,
                      const uint8_t *d_white, int order, float *d_cam, float *d_proj, double *d_colors,
                      unsigned long long *d_total);
int launch_cloud_lists(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, const float *d_xyz, const uint8_t *d_white, int cam_w, int cam_h,
                       int proj_w, int proj_h, float *d_cam, float *d_proj, double *d_pts, double *d_colors, unsigned long long *d_total);
int launch_cloud_tri(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, const uint8_t *d_white, int cam_w, int cam_h, int proj_w, int proj_h,
                     float *d_cam, float *d_proj, double *d_pts, double *d_colors, unsigned long long *d_total, int f32 = 0);
int launch_filter(slgc_ctx *ctx, const double *d_xyz, const double *d_colors, int64_t M, double thr, double *d_xyz_out,
                  double *d_colors_out, unsigned long long *d_total, int pass);
int launch_compact_dense(slgc_ctx *ctx, const float *d_xyz, int rows, int W, int row0, float *d_points, uint32_t *d_keys,
                         unsigned long long *d_count);
int launch_compact_records(slgc_ctx *ctx, const float *d_xyz, int rows, int W, int row0, void *d_records, unsigned long long *d_count);
// triangulate.hip
int launch_triangulate_list(slgc_ctx *ctx, const float *d_cam, const float *d_proj, int64_t M, int mode, double *d_xyz);
int launch_triangulate_maps(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, int rows, int W, int row0, int proj_w,
                            int proj_h, int mode, float *d_xyz, unsigned long long *d_count, const uint8_t *d_wire = nullptr);
int launch_undistort_list(slgc_ctx *ctx, int which, const float *d_pts, int64_t M, float *d_out);
int launch_guard_count(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, int rows, int W, int row0, int proj_w, int proj_h,
                       unsigned long long *d_counts);
// synth.hip
int launch_synth(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows, uint32_t seed,
                 int noise, int shadow);
int launch_move_only(slgc_ctx *ctx, const uint8_t *d_stack, size_t plane_stride, int N, size_t npix, int16_t *d_h, int16_t *d_v, float *d_xyz);
int launch_synth_physical(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows, int proj_w, int proj_h,
                          uint32_t seed, int noise, int gain_lo, int gain_hi, double r2_max, int16_t *d_h, int16_t *d_v, float *d_truth);
int launch_synth_bgr(slgc_ctx *ctx, const uint8_t *d_gray, size_t gray_stride, int N, int W, int row0, int rows, uint8_t *d_bgr, size_t bgr_stride);
int launch_synth_uniform(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int W, int row0, int rows, uint32_t seed);
// ingest.hip
int launch_bgr_to_gray(slgc_ctx *ctx, const uint8_t *d_bgr, uint8_t *d_gray, size_t npix, int coeff_bits);
int launch_bgr_to_gray_tiled(slgc_ctx *ctx, const uint8_t *d_bgr, uint8_t *d_tiled, int n_frames, size_t npix, int coeff_bits, int tile_log2);
int launch_tile_stack(slgc_ctx *ctx, const uint8_t *d_planar, size_t plane_stride, int n_frames, size_t npix, int tile_log2, uint8_t *d_tiled);
int launch_frame_diff_counts(slgc_ctx *ctx, const void *d_frames, int dtype, int n_frames, size_t elems, double thresh,
                             unsigned long long *d_counts);

// 3-byte wire format (wire.hip): three dwords = four pixels of (h: bits 0..11, v: bits 12..23, 0xFFF = -1) -> int16 pairs.
__device__ __forceinline__ void unpack_hv24_x4(uint32_t w0, uint32_t w1, uint32_t w2, uint2 &hw, uint2 &vw)
{
    const uint32_t p[4] = {w0 & 0xffffffu, (w0 >> 24) | ((w1 & 0xffffu) << 8), (w1 >> 16) | ((w2 & 0xffu) << 16), w2 >> 8};
    uint32_t hh[4], vv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t a = p[j] & 0xfffu, b = p[j] >> 12;
        hh[j] = a == 0xfffu ? 0xffffu : a;
        vv[j] = b == 0xfffu ? 0xffffu : b;
    }
    hw = make_uint2(hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16));
    vw = make_uint2(vv[0] | (vv[1] << 16), vv[2] | (vv[3] << 16));
}

// wire.hip
int launch_pack_hv24(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, size_t npix, uint8_t *d_out);
int launch_unpack_hv24(slgc_ctx *ctx, const uint8_t *d_in, size_t npix, int16_t *d_h, int16_t *d_v);

// XCD-aware workgroup -> tile map.  The dispatcher hands consecutive workgroup ids to the 8 XCDs round-robin, and each XCD has
// its own L2.  With the identity map every XCD sees every 8th tile of the image; with this map XCD x owns the contiguous
// tile range [x*chunk, (x+1)*chunk) (a band of whole rows), so the projector-ray lines it gathers are shared by neighbouring
// rows inside one L2.  Ids past 8*chunk (blocks % 8 of them) keep their position.  chunk = 0: identity.
__device__ __forceinline__ uint32_t xcd_block(uint32_t bid, uint32_t chunk)
{
    return (chunk != 0u && bid < chunk * 8u) ? (bid & 7u) * chunk + (bid >> 3) : bid;
}
// The same idea in small: groups of 8 * run workgroups, inside a group XCD x owns `run` consecutive tiles -- every XCD works on whole rows
// while all eight stay inside the same few dozen rows of the image (the banded map above sends them 1/8 of the image apart).
__device__ __forceinline__ uint32_t xcd_block_fine(uint32_t bid, uint32_t run, uint32_t nblocks)
{
    const uint32_t span = 8u * run, group = bid / span;
    if ((group + 1u) * span > nblocks) return bid;                      // the ragged last group keeps its order
    const uint32_t in = bid - group * span;
    return group * span + (in & 7u) * run + (in >> 3);
}
// Integer environment knob (A/B switches: SLGC_XCD, SLGC_TRI_NT, SLGC_FUSE_NT).
inline int xcd_env(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
struct CamNodes;
constexpr size_t kCamNodesMinTable = (size_t)12 << 20;      // bytes of per-pixel camera rays above which the scan kernels read the every-4th-column node table
// The node table of the band ensure_luts() last built (tri_math.h), or an empty one (kernels then read the per-pixel table).
// kCamNodesMinTable: round 4 re-measured the choice at 1920x1080 (a 16.6 MB per-pixel table) after the fast form: node table 24.0 vs 24.5 us on the physical
// capture, 25.1 vs 26.1 on the S-scene, 27.6 vs 28.5 on S-uniform (two boxes); at 1280x720 (7.4 MB) it still loses 2 %: the limit went from 64 MB to 12 MB.
// tune_cam_nodes 1 (default) = when it pays: above 12 MB of per-pixel rays (1920x1080 and larger; measured at 4096x3000: -1.3 % fused / -4.5 %
// two-kernel step) the node table wins, below it (1280x720, 7.4 MB: the table stays cache-resident between scans and the nodes only add
// arithmetic) it loses 2 %; 2 = whenever it is accurate enough (tests); 0 = never.  The node-table rays are not bit-identical with the per-pixel
// table's (within 1 ulp; XYZ within the 1e-4 bar): moving the limit from 64 MB to 12 MB in round 4 changed the low bits of XYZ at 1920x1080 --
// cam_nodes = 0 gives the per-pixel table's bits at every size.  The size that decides is the WHOLE image's (tune_image_rows
// when the context scans a band of a taller image): the choice must not depend on how many GPUs share the image.
#define SLGC_CAM_NODES_FOR(ctx, W, allow)                                                                                                   \
    (((allow) && (ctx)->lut_nodes && (ctx)->lut_cam_W == (W) &&                                                                            \
      ((ctx)->tune_cam_nodes == 2 ||                                                                                                       \
       ((ctx)->tune_cam_nodes == 1 && (size_t)((ctx)->lut_image_rows > 0 ? (ctx)->lut_image_rows : (ctx)->lut_cam_rows) * (size_t)(W) * 8u > kCamNodesMinTable))) \
         ? CamNodes{(const float2 *)(ctx)->lut_nodes, (uint32_t)((W) / 4), (uint32_t)((W) / 4 + 3), 1.0f / (float)((W) / 4)}              \
         : CamNodes{nullptr, 1u, 1u, 1.0f})
inline uint32_t xcd_chunk_for(const slgc_ctx *ctx, unsigned blocks)
{
    return (ctx->tune_xcd && blocks >= 64) ? blocks / 8 : 0u;
}

// Launch on the context's stream.  When the launch is being sampled (slgc_prof_begin .. _end), the event pair is attached to the
// kernel's own dispatch packet (hipExtLaunchKernelGGL): its elapsed time is the kernel's begin -> end, the interval rocprofv3's
// kernel trace reports, without the command-processor time a hipEventRecord pair around the launch adds.
#define SLGC_LAUNCH(ctx, kernel, grid, block, ...)                                                                        \
    do {                                                                                                                  \
        if ((ctx)->prof_cur[0]) {                                                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, (ctx)->prof_cur[0], (ctx)->prof_cur[1], 0, __VA_ARGS__); \
            (ctx)->prof_bound = true;                                                                                     \
            (ctx)->prof_cur[0] = (ctx)->prof_cur[1] = nullptr; /* one kernel per sample: a ragged-tail launch is not timed */ \
        } else {                                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, __VA_ARGS__);                                       \
        }                                                                                                                 \
    } while (0)
