/*
 * slgc.h -- C-ABI of libslgc.so: MI355X (gfx950) structured-light Gray-code decode + triangulation.
 *
 * This is the drop-in boundary for the hot path of guillaume-charron/3DScanner-GrayCode.  The
 * reference has no FFI layer (it is in-process Python on NumPy arrays), so each entry point below
 * is what a ctypes binding of the corresponding reference function binds; the Python package
 * `3dscanner-graycode_amd/scanner` is that binding.  Citations are
 * file:line under the reference checkout.
 *
 * Conventions
 *  - plain pointers and sizes only; every function returns an int status (0 = SLGC_OK, <0 = error)
 *    and never throws.  slgc_last_error(ctx) gives the text of the last failure on that context.
 *  - "host" entry points take caller-owned host buffers (NumPy arrays) and do H2D / kernels / D2H.
 *    "_dev" entry points take device pointers obtained from slgc_dev_alloc and only enqueue work on
 *    the context's HIP stream (no sync) -- they are what bench.py times.
 *  - a context = (device, HIP stream, workspace).  Calls on one context are serialised by the
 *    caller; separate contexts may be used from separate threads (ctypes releases the GIL).
 *  - image stacks are frame-major [N][H][W] (src/3-capture_decode.py:68-70), dtype SLGC_U8 or
 *    SLGC_F64 (the reference's own float64 stack; narrowed to uint8 on the host when every sample is a grey level,
 *    see slgc_last_input_path).  14 <= N <= 65 (code length L=int((N-2)/4) <= 15).
 *  - there is NO CPU fallback: without a HIP device slgc_create fails with SLGC_ENODEV.
 *  - this header is the reference-facing ABI only.  Measurement and diagnostic exports of the same library (synthetic captures, device
 *    self-tests, the movement-only yardstick, event / per-launch timing, the A/B knobs of slgc_tune) are declared in slgc_bench.h; nothing a
 *    binding of the reference's functions needs is in there.
 */
#ifndef SLGC_H
#define SLGC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLGC_VERSION 100 /* 0.1.0 */

enum {
    SLGC_OK = 0,
    SLGC_EINVAL = -1, /* bad argument (N out of range, null pointer, misaligned band, ...) */
    SLGC_ENODEV = -2, /* no usable HIP device */
    SLGC_EHIP = -3,   /* HIP runtime error (text in slgc_last_error) */
    SLGC_ENOMEM = -4,
    SLGC_ECOMM = -5,  /* RCCL error / communicator not initialised */
    SLGC_ESTATE = -6  /* call order error (e.g. triangulate before set_calibration) */
};

enum { SLGC_U8 = 0, SLGC_F64 = 1 };
enum { SLGC_ORDER_X = 0 /* reference scan order, triangulate.py:52-53 */, SLGC_ORDER_ROW = 1 };
enum {
    SLGC_TRI_EXACT = 0,     /* acos/sin as triangulate.py:91-94 */
    SLGC_TRI_ALGEBRAIC = 1, /* algebraically identical sqrt form */
    SLGC_TRI_DIRECT = 2,    /* flag for the dense (_dev) path: evaluate undistortPoints per pixel instead of the ray tables */
    SLGC_TRI_SPLIT = 4      /* flag for slgc_scan_dev: always run decode and triangulation as two kernels (no fusion) */
};

typedef struct slgc_ctx slgc_ctx;

/* ------------------------------------------------------------------ library / context */
int slgc_version(void);
const char *slgc_backend(void); /* "hip:gfx950" */
const char *slgc_strerror(int status);
int slgc_device_count(void);    /* number of HIP devices, 0 if none / runtime unusable */
int slgc_create(int device, slgc_ctx **out);
int slgc_destroy(slgc_ctx *ctx);
const char *slgc_last_error(slgc_ctx *ctx);
int slgc_synchronize(slgc_ctx *ctx);
/* Settings of a context by name.  Three of them belong to the product: "image_rows" H > 0 = this context scans row bands of an image of H
 * rows (the multi-GPU plan sets it): the camera ray-table choice is then taken for the WHOLE image, so a pixel's XYZ is bit-identical whether
 * one GPU scans the image or N GPUs scan its bands (0, the default: the band is the image); "wire" 1 = slgc_scan_sharded_dev exchanges the
 * maps in the 3-byte wire format / 0 = int16 (default); "cam_nodes" 1 (default) = above 12 MB of per-pixel camera rays (1920x1080 and up) the
 * scan kernels interpolate the rays from an every-4th-column table (rays within 2 float32 ulp, XYZ inside the 1e-4 tolerance, maps
 * untouched) / 0 = always the per-pixel table: the one setting that changes result bits.  Everything else slgc_tune accepts is an A/B
 * timing knob documented in slgc_bench.h.  Unknown names: SLGC_EINVAL. */
int slgc_tune(slgc_ctx *ctx, const char *name, int value);
/* How the last host-buffer decode call on this context took its stack in: 0 = uint8 as given; 1 = float64 whose samples were all
 * integers in [0,255] (what src/3-capture_decode.py:66-70 builds), narrowed to uint8 on host threads into pinned staging and
 * uploaded as 1 byte per sample; 2 = float64 shipped as it is (a fraction / negative / NaN was found) and decoded by the float64
 * kernel.  Results are identical on all three.  slgc_compute_count reports its int64 maps the same way: 1 = narrowed to int16 on host threads
 * (every value fits), 2 = shipped as int64. */
int slgc_last_input_path(slgc_ctx *ctx);

/* Which kernels the last slgc_scan_dev / slgc_scan_batch_dev / slgc_cloud_dev call (or slgc_decode_dev + slgc_triangulate_maps_dev pair) on this context launched (slgc_scan_dev silently takes the two-kernel
 * path when a buffer is misaligned, the band is ragged, a count is requested or the mode asks for it): returns one of SLGC_PATH_* (or a
 * negative status).  Optional outputs: *ns_frames = the frames-per-run specialisation the decode / fused kernel was compiled for (42, 44,
 * 46, 50, 54; 0 = the generic kernel); *node_table = 1 if the triangulation read the every-4th-column camera table; *guard = 1 float32 fast form
 * with the flat-triangle guard, 0 exact (acos / sin) mode.  bench.py reports its pipeline from this, not from its own arguments. */
enum {
    SLGC_PATH_NONE = 0,
    SLGC_PATH_FUSED = 1,        /* one kernel: decode with the triangulation tail */
    SLGC_PATH_SPLIT = 2,        /* decode kernel + dense triangulation kernel, both on their vector paths */
    SLGC_PATH_SPLIT_RAGGED = 3, /* two kernels and a byte-wide / per-pixel fallback kernel took part (misaligned or ragged band) */
    SLGC_PATH_BATCH_FUSED = 4,  /* slgc_scan_batch_dev: all scans in one launch of the fused kernel */
    SLGC_PATH_CLOUD = 5,        /* slgc_cloud_dev: decode kernel + x-major list build that triangulates in-kernel (no dense XYZ) */
    SLGC_PATH_FUSED_BGR = 6     /* slgc_scan_bgr_dev: the fused kernel reading the camera's BGR frames (luma formed inside the frame loads) */
};
int slgc_last_scan_path(slgc_ctx *ctx, int *ns_frames, int *node_table, int *guard);
/* Fallback kernels of the last scan-related call, whatever SLGC_PATH_* it reports (slgc_cloud_dev and the fused paths included): bit 0 = the
 * byte-wide decode kernel took part (misaligned buffers, a ragged tail), bit 1 = the per-pixel triangulation kernel did.  A
 * slgc_triangulate_maps_dev call on its own reports only its own bit; directly after slgc_decode_dev it completes a two-kernel scan and
 * SLGC_PATH_SPLIT_RAGGED then covers both. */
int slgc_last_scan_ragged(slgc_ctx *ctx);
/* Which scatter kernel the last x-major list build on this context (slgc_cloud_dev, slgc_cloud_lists_dev, slgc_correspond with
 * SLGC_ORDER_X) launched: SLGC_LISTS_TILE_RUNS = a tile writes its own run of every column (any shape, any map type);
 * SLGC_LISTS_WHOLE_LINES = slgc_cloud_dev's form that writes whole 16-record groups = aligned 128-byte lines (int16 maps, W % 4 == 0,
 * 4-byte aligned maps and white image, 2048 tiles ... 2^27 pixels; slgc_tune "lists_lines").  Same arrays either way. */
enum { SLGC_LISTS_NONE = 0, SLGC_LISTS_TILE_RUNS = 1, SLGC_LISTS_WHOLE_LINES = 2 };
int slgc_last_list_kernel(slgc_ctx *ctx);
int slgc_device_name(slgc_ctx *ctx, char *buf, int buflen);
/* ------------------------------------------------------------------ decode, host buffers */

/* get_direct_indirect(images) -- scanner/grayCode/decode_codes.py:90-122.  L_d, L_g: float64 [H][W]. */
int slgc_direct_indirect(slgc_ctx *ctx, const void *stack, int dtype, int N, int H, int W, double *L_d, double *L_g);

/* get_is_lit(images, L_d, L_g, eps, m) -- decode_codes.py:125-186.  codes: int8 [L][H][W] in {-1,0,1}. */
int slgc_is_lit(slgc_ctx *ctx, const void *stack, int dtype, int N, int H, int W, const double *L_d, const double *L_g,
                double eps, double m, int8_t *h_codes, int8_t *v_codes);

/* get_codes(images) -- decode_codes.py:231-248 (eps=1, m=10 are the reference defaults). */
int slgc_codes(slgc_ctx *ctx, const void *stack, int dtype, int N, int H, int W, double eps, double m, int8_t *h_codes,
               int8_t *v_codes);

/* Driver tail src/3-capture_decode.py:95-100: max-merge n_runs code stacks ([n_runs][L][H][W] int8), then
 * gray_to_decimal per pixel (decode_codes.py:209-229; v codes flipped).  Maps: int64 [H][W], -1 = undecodable. */
int slgc_codes_to_pixels(slgc_ctx *ctx, const int8_t *h_codes, const int8_t *v_codes, int n_runs, int L, int H, int W,
                         int64_t *h_pixels, int64_t *v_pixels);

/* Fused get_codes per run -> merge -> maps (src/3-capture_decode.py:75-100).  stacks: n_runs pointers. */
int slgc_decode(slgc_ctx *ctx, const void *const *stacks, int dtype, int n_runs, int N, int H, int W, double eps, double m,
                int64_t *h_pixels, int64_t *v_pixels);

/* ------------------------------------------------------------------ triangulation, host buffers */

/* Calibration as held by Triangulate.__init__ (scanner/triangulation/triangulate.py:5-37).  proj_K must already
 * carry the row scaling of :28-33 (the Python class applies it, mutating the caller's array like the reference).
 * dist: k1,k2,p1,p2[,k3[,k4,k5,k6[,s1..s4]]] (5x1 and 1x5 both occur in the reference's data files). */
int slgc_set_calibration(slgc_ctx *ctx, const double cam_K[9], const double *cam_dist, int n_cam_dist,
                         const double proj_K[9], const double *proj_dist, int n_proj_dist, const double R[9],
                         const double T[3]);

/* get_cam_proj_pts(img_white) -- triangulate.py:39-71.  Two calls: _count runs the kernels and leaves the
 * lists on the device, _fetch copies them out (cam/proj: float32 [M][2]; colors: float64 [M][3] = rgb/255,
 * may be NULL when white was NULL). */
int slgc_cam_proj_pts_count(slgc_ctx *ctx, const int64_t *h_pixels, const int64_t *v_pixels, int cam_w, int cam_h,
                            int proj_w, int proj_h, const uint8_t *white_rgb, int order, int64_t *M);
int slgc_cam_proj_pts_fetch(slgc_ctx *ctx, float *cam_pts, float *proj_pts, double *colors);

/* triangulate(cam_pts, proj_pts) -- triangulate.py:73-97.  xyz: float64 (3,M) row-major. */
int slgc_triangulate(slgc_ctx *ctx, const float *cam_pts, const float *proj_pts, int64_t M, int mode, double *xyz);

/* The two cv2.undistortPoints calls inside triangulate (triangulate.py:84-85) on their own: which = 0 the camera call
 * (cam_mtx, cam_dist, R = proj_R), which = 1 the projector call (proj_mtx, proj_dist).  pts / out: float32 [M][2].  OpenCV
 * (opencv-contrib-python 4.8.0.76) is third-party and absent from the build container: this is the published algorithm of
 * cvUndistortPointsInternal restated (5 fixed-point iterations, icdist < 0 bail-out, R after, float32 out) -- PARITY UNPINNED;
 * tools/pin_third_party.py closes it the day a cv2 wheel is at hand. */
int slgc_undistort_points(slgc_ctx *ctx, int which, const float *pts, int64_t M, float *out);

/* filter_3d_pts(Pts, colors, threshold) -- triangulate.py:99-122.  _count then _fetch (order preserved). */
int slgc_filter_count(slgc_ctx *ctx, const double *xyz, const double *colors, int64_t M, double threshold, int64_t *kept);
int slgc_filter_fetch(slgc_ctx *ctx, double *xyz_out, double *colors_out);

/* Triangulation.compute() -- the fused call BASELINE.json's north star names for the drop-in class: what src/4-triangulate.py:62-71 does with a
 * Triangulate object (get_cam_proj_pts :39-71 -> triangulate :73-97 -> filter_3d_pts :99-122) as one device-resident chain.  The int64 maps
 * cross the link ONCE, as int16 when every value fits (narrowed by host threads into pinned memory; otherwise as they are); the same three
 * kernels as the three entry points above run back to back in HBM (results bit-identical to calling them one after the other); only what the
 * script keeps comes back: _count returns M = points kept (and the unfiltered list length), _fetch copies out xyz float64 (3,M) and colours
 * float64 [M][3] (NULL when white_rgb was NULL).  threshold = NaN: no box filter.  Needs slgc_set_calibration. */
int slgc_compute_count(slgc_ctx *ctx, const int64_t *h_pixels, const int64_t *v_pixels, int cam_w, int cam_h, int proj_w, int proj_h,
                       const uint8_t *white_rgb, int order, int mode, double threshold, int64_t *M, int64_t *M_unfiltered);
int slgc_compute_fetch(slgc_ctx *ctx, double *xyz, double *colors);

/* ------------------------------------------------------------------ ingest (the frames either side of the path) */

/* to_gray(images) -- decode_codes.py:70-87 / src/3-capture_decode.py:66: cv2.cvtColor(BGR2GRAY) of n 8-bit BGR frames
 * [n][H][W][3] into the uint8 stack [n][H][W].  coeff_bits 15 = OpenCV 4.x fixed-point luma (9798, 19235, 3735), 14 = the
 * older (4899, 9617, 1868) set.  OpenCV is third-party and not installed in the build container: PARITY UNPINNED. */
int slgc_to_gray(slgc_ctx *ctx, const uint8_t *bgr, int n_frames, int H, int W, int coeff_bits, uint8_t *gray);
int slgc_to_gray_dev(slgc_ctx *ctx, const uint8_t *d_bgr, size_t npix, int coeff_bits, uint8_t *d_gray);

/* The arithmetic of remove_bad_images (decode_codes.py:34-68): counts[j] = number of elements with
 * |frames[j+1] - frames[j]| > thresh, j = 0 .. n_frames-2 (cv2.absdiff + np.argwhere + len).  frames: [n][elems]. */
int slgc_frame_diff_counts(slgc_ctx *ctx, const void *frames, int dtype, int n_frames, size_t elems_per_frame, double thresh,
                           int64_t *counts);
/* The same counts for frames that already sit in HBM (the stack a capture pipeline uploaded for slgc_decode_dev): asynchronous on the
 * context's stream, d_counts = n_frames - 1 uint64 in device memory. */
int slgc_frame_diff_counts_dev(slgc_ctx *ctx, const void *d_frames, int dtype, int n_frames, size_t elems_per_frame, double thresh,
                               unsigned long long *d_counts);

/* Point-cloud post-processing: mean distance of every point to its k nearest points, itself
 * included -- the arithmetic of Open3D's remove_statistical_outlier as called at scanner/utils/visualize.py:104 (exact k-NN on
 * a uniform grid; Open3D itself is absent from the build container, parity with it is UNPINNED).  pts float32 [M][3], 1<=k<=64. */
int slgc_knn_mean_distance(slgc_ctx *ctx, const float *pts, int64_t M, int k, double *mean);
/* The same on a cloud that already sits in HBM (d_pts float32 [M][3], d_mean float64 [M], both device memory of the caller).  Waits for the
 * work enqueued on the context's stream before it (the search grid is sized from statistics of the cloud); on return the last kernel is
 * enqueued, not finished (slgc_synchronize / slgc_d2h). */
int slgc_knn_mean_distance_dev(slgc_ctx *ctx, const float *d_pts, int64_t M, int k, double *d_mean);

/* ------------------------------------------------------------------ whole pipeline, one upload */

/* The reference's driver glue in one device-resident pass: src/3-capture_decode.py:75-100 (get_codes per run, max-merge,
 * gray_to_decimal) then src/4-triangulate.py:50-71 (get_cam_proj_pts, triangulate, filter_3d_pts when threshold is not NaN).
 * _count runs it and returns the number of points (after the filter); _fetch copies out whatever pointers are non-NULL:
 * maps int64 [H][W]; xyz float64 (3,M); colors float64 [M][3]; cam/proj float32 [M_unfiltered][2] (the lists before the filter). */
int slgc_pipeline_count(slgc_ctx *ctx, const void *const *stacks, int dtype, int n_runs, int N, int H, int W, double eps, double m,
                        int proj_w, int proj_h, const uint8_t *white_rgb, int order, int mode, double threshold, int64_t *M);
int slgc_pipeline_fetch(slgc_ctx *ctx, int64_t *h_pixels, int64_t *v_pixels, double *xyz, double *colors, int64_t *M_unfiltered,
                        float *cam_pts, float *proj_pts);

/* ------------------------------------------------------------------ device-resident path (what bench.py times) */
int slgc_dev_alloc(slgc_ctx *ctx, size_t bytes, void **dptr);
int slgc_dev_free(slgc_ctx *ctx, void *dptr);
int slgc_h2d(slgc_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int slgc_d2h(slgc_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int slgc_dev_memset(slgc_ctx *ctx, void *dptr, int value, size_t bytes);

/* Decode n_runs uint8 stacks resident in HBM into int16 maps [rows][W] (-1 = undecodable).
 * Run r, frame f starts at d_stack + r*run_stride + f*plane_stride (bytes); `rows` rows of W pixels are
 * decoded from each frame (a row band of a taller image when plane_stride > rows*W).  variant: 0 = auto. */
int slgc_decode_dev(slgc_ctx *ctx, const uint8_t *d_stack, int n_runs, size_t run_stride, size_t plane_stride, int N,
                    int rows, int W, double eps, double m, int16_t *d_h, int16_t *d_v, int variant);

/* Decode -> clamp -> triangulate on a row band: dense float32 XYZ [rows][W][3] (NaN where undecodable),
 * optional int16 maps (d_h = d_v = NULL: XYZ is the only product -- the one-kernel form then stores no maps at all, N + 12 bytes per
 * pixel of HBM traffic instead of N + 16; the two-kernel form keeps them in scratch of the context), and *d_count += number of valid pixels.  row0 = first row of the band in the
 * full image (camera y of local row 0).  With mode = SLGC_TRI_ALGEBRAIC, no count requested and 4-byte aligned buffers
 * this is ONE kernel (decode with the triangulation tail); otherwise (or with SLGC_TRI_SPLIT) two kernels. */
int slgc_scan_dev(slgc_ctx *ctx, const uint8_t *d_stack, int n_runs, size_t run_stride, size_t plane_stride, int N,
                  int rows, int W, int row0, int proj_w, int proj_h, double eps, double m, int mode, int16_t *d_h,
                  int16_t *d_v, float *d_xyz, unsigned long long *d_count);
/* slgc_scan_dev straight from the camera's BGR frames -- replaces the cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY) + grey-stack fill of
 * src/3-capture_decode.py:66-70 together with :75 (get_codes) and src/4-triangulate.py:50-64.  d_bgr: uint8 [n_runs][N][rows][W][3] (OpenCV's
 * pixel order), plane_stride = BYTES between consecutive frames (>= 3 * rows * W), run_stride = bytes between runs; coeff_bits as slgc_to_gray.
 * With N = 42 / 44 / 46 / 50 / 54 (what the reference's generator emits for projectors up to 1024 / 2048 / 2048 / 4096 / 8192 pixels), 4-byte aligned planes, SLGC_TRI_ALGEBRAIC and no count: ONE kernel, the luma formed in registers inside the frame
 * loads (3 N + 12 bytes per pixel; the grey stack never exists in HBM; slgc_last_scan_path = SLGC_PATH_FUSED_BGR).  Otherwise
 * slgc_to_gray_dev into scratch of the context + slgc_scan_dev.  Results bit-identical with that chain either way. */
int slgc_scan_bgr_dev(slgc_ctx *ctx, const uint8_t *d_bgr, int n_runs, size_t run_stride, size_t plane_stride, int N, int rows, int W, int row0,
                      int proj_w, int proj_h, int coeff_bits, double eps, double m, int mode, int16_t *d_h, int16_t *d_v, float *d_xyz,
                      unsigned long long *d_count);
/* slgc_decode_dev straight from BGR frames (src/3-capture_decode.py:66-70 + :75-100): the int16 maps alone, for callers that go on with slgc_cloud_lists_dev /
 * slgc_triangulate_maps_dev.  Arguments, shapes and fall-back as slgc_scan_bgr_dev; maps bit-identical with slgc_to_gray_dev + slgc_decode_dev. */
int slgc_decode_bgr_dev(slgc_ctx *ctx, const uint8_t *d_bgr, int n_runs, size_t run_stride, size_t plane_stride, int N, int rows, int W, int coeff_bits,
                        double eps, double m, int16_t *d_h, int16_t *d_v);
/* Throughput mode (BASELINE configs[4]): n_scans independent single-run scans of one geometry in one launch -- stacks scan_stride bytes
 * apart, d_h / d_v [n_scans][rows * W] int16 and d_xyz [n_scans][rows * W][3] float32 back to back.  Same results as n_scans calls of
 * slgc_scan_dev (which is what shapes that are not a whole number of 512-pixel workgroups, and the other modes, fall back to).
 * d_h = d_v = NULL: XYZ only, as there. */
int slgc_scan_batch_dev(slgc_ctx *ctx, const uint8_t *d_stacks, int n_scans, size_t scan_stride, size_t plane_stride, int N, int rows, int W,
                        int row0, int proj_w, int proj_h, double eps, double m, int mode, int16_t *d_h, int16_t *d_v, float *d_xyz);

/* Triangulate dense int16 maps (as written by slgc_decode_dev) into dense XYZ; same outputs as slgc_scan_dev. */
int slgc_triangulate_maps_dev(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, int rows, int W, int row0,
                              int proj_w, int proj_h, int mode, float *d_xyz, unsigned long long *d_count);

/* Per-calibration work of the dense path, hoisted out of the scans: both cv2.undistortPoints calls of triangulate.py:84-85 only
 * ever see integer pixel coordinates there, so their float32 results are evaluated once per (calibration, band, projector size)
 * into two ray tables.  slgc_scan_dev / slgc_triangulate_maps_dev build them on first use; this entry point builds them
 * explicitly (asynchronous, on the context's stream) so that a caller -- and bench.py -- can place and time that one-off cost. */
int slgc_build_ray_tables_dev(slgc_ctx *ctx, int rows, int W, int row0, int proj_w, int proj_h);
/* The camera table exists twice: per pixel (exact float32 rays, 8 B / pixel) and at every 4th column (2 B / pixel), from which the scan
 * kernels' fast form interpolates each row's rays with a cubic through four nodes -- kept only when, over every pixel of the band, the
 * interpolated ray stays within 2 float32 ulp (of a number in [1, 2)) of the exact one and every component of at least 1e-3 within 4e-6
 * of itself (flat triangles, lanes whose rays cross zero, and the exact mode always read the per-pixel table).
 * *in_use = 1 if the kernels will read the node table for the tables built last (0: W % 4 != 0, too rough a lens, an image of at
 * most 12 MB of rays under slgc_tune "cam_nodes" 1, or "cam_nodes" 0); *max_err = the measured error, in units where 2.4e-7 is the acceptance limit (-1 if no node table was built).
 * With slgc_tune "image_rows" both the size and the measured error are the WHOLE image's, whatever band the tables cover.
 * Building a node table reads its error back: that one call synchronises the context's stream (once per calibration / geometry). */
int slgc_ray_table_info(slgc_ctx *ctx, int *in_use, double *max_err);

/* The reference-shaped product without leaving HBM: from the int16 maps and the dense XYZ of a (full-image) scan, the x-major
 * correspondence lists of get_cam_proj_pts (triangulate.py:52-71: columns outer, rows inner, clamp to the projector, colour =
 * white[y][x][:] / 255.0 from a device-resident uint8 RGB image) and, gathered in the same pass, the float64 (3,M) point array
 * Triangulate.triangulate returns (:95).  Asynchronous; *d_total (device) receives M; d_pts rows start at d_pts, d_pts + M,
 * d_pts + 2M.  Capacity of every list: cam_w * cam_h entries.  d_white_rgb / d_colors may be NULL (pairwise); d_pts may be NULL (lists
 * only).  d_xyz = NULL with d_pts given: there is no dense XYZ -- every valid pixel is triangulated inside the list build from the maps and the
 * ray tables (needs slgc_set_calibration; the second half of slgc_cloud_dev). */
int slgc_cloud_lists_dev(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, const float *d_xyz, const uint8_t *d_white_rgb, int cam_w,
                         int cam_h, int proj_w, int proj_h, float *d_cam_pts, float *d_proj_pts, double *d_pts, double *d_colors,
                         unsigned long long *d_total);

/* The reference-shaped product of a whole scan in one call, nothing dense in between: decode the stack into the int16 maps (a product;
 * d_h / d_v may be NULL: workspace), then build the x-major lists of get_cam_proj_pts (triangulate.py:52-71) with every valid pixel
 * TRIANGULATED INSIDE the list build (triangulate.py:84-95; the fused scan kernel's arithmetic: same float32 XYZ, bit for bit, widened to the
 * reference's float64 (3,M)) and its colour gathered from the device-resident white image (:64, :69).  Replaces slgc_scan_dev +
 * slgc_cloud_lists_dev for callers that want the lists: the scan no longer writes 12 B/pixel of dense XYZ for the list build to read back.
 * Whole images only (x-major order needs every row): rows = cam_h, row0 = 0.  Outputs as slgc_cloud_lists_dev (capacity cam_w * cam_h
 * entries each; d_pts / d_colors may be NULL, d_colors needs d_white_rgb).  d_cam_pts / d_proj_pts may be NULL together: the two
 * correspondence lists are intermediates of src/4-triangulate.py:62-64 -- what that script keeps is pts_3d and colors (:67-68) -- and leaving
 * them out saves a quarter of the bytes the list build writes.  Asynchronous; *d_total (device) receives M. */
int slgc_cloud_dev(slgc_ctx *ctx, const uint8_t *d_stack, int n_runs, size_t run_stride, size_t plane_stride, int N, int cam_h, int cam_w,
                   int proj_w, int proj_h, double eps, double m, const uint8_t *d_white_rgb, int16_t *d_h, int16_t *d_v, float *d_cam_pts,
                   float *d_proj_pts, double *d_pts, double *d_colors, unsigned long long *d_total);
/* slgc_cloud_dev with the points as float32 (3,M) and the colours as float32 [M][3]: an explicitly NON-reference product (the reference returns float64,
 * triangulate.py:95, :69) for callers that do not need float64 -- the same values rounded to float32 (the points are float32 inside the kernels
 * already), 40 instead of 64 bytes written per point with the correspondence lists, 24 instead of 48 without them.  d_pts32 is required. */
int slgc_cloud32_dev(slgc_ctx *ctx, const uint8_t *d_stack, int n_runs, size_t run_stride, size_t plane_stride, int N, int cam_h, int cam_w,
                     int proj_w, int proj_h, double eps, double m, const uint8_t *d_white_rgb, int16_t *d_h, int16_t *d_v, float *d_cam_pts,
                     float *d_proj_pts, float *d_pts32, float *d_colors32, unsigned long long *d_total);

/* Row-major compaction of a dense band: keeps pixels with finite XYZ; writes float32 [M][3] points and uint32
 * linear pixel keys ((row0+y)*W + x); *d_count (device) receives M.  Capacity of outputs: rows*W records. */
int slgc_compact_dev(slgc_ctx *ctx, const float *d_xyz, int rows, int W, int row0, float *d_points, uint32_t *d_keys,
                     unsigned long long *d_count);

/* Same compaction into 16-byte exchange records {float32 x, y, z, uint32 key} (d_records 16-byte aligned). */
int slgc_compact_records_dev(slgc_ctx *ctx, const float *d_xyz, int rows, int W, int row0, void *d_records,
                             unsigned long long *d_count);

/* 3-byte wire format of the maps for the multi-GPU exchange (the exchange, not the kernels, bounds a sharded scan): per pixel
 * bits 0..11 = h, bits 12..23 = v, 0xFFF = -1.  Holds codes of at most SLGC_WIRE_MAX_CODE_BITS bits (N <= 49 frames; the
 * reference's captures use 10); _pack rejects more.  d_wire: 3 * npix bytes. */
#define SLGC_WIRE_MAX_CODE_BITS 11
int slgc_pack_hv24_dev(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, size_t npix, int code_bits, uint8_t *d_wire);
int slgc_unpack_hv24_dev(slgc_ctx *ctx, const uint8_t *d_wire, size_t npix, int16_t *d_h, int16_t *d_v);
/* slgc_triangulate_maps_dev on maps that arrive in the wire format: unpacks inside the triangulation kernel's map load and
 * writes the int16 maps d_h / d_v as well (the separate unpack pass disappears).  mode: SLGC_TRI_EXACT or SLGC_TRI_ALGEBRAIC. */
int slgc_triangulate_wire_dev(slgc_ctx *ctx, const uint8_t *d_wire, int rows, int W, int row0, int proj_w, int proj_h, int mode,
                              int16_t *d_h, int16_t *d_v, float *d_xyz, unsigned long long *d_count);

/* ------------------------------------------------------------------ multi-GPU (RCCL over xGMI) */
#define SLGC_UNIQUE_ID_BYTES 128
int slgc_comm_unique_id(void *id128);                                          /* rank 0 creates, host shares */
int slgc_comm_init(slgc_ctx *ctx, int rank, int nranks, const void *id128);    /* one context = one rank = one GPU */
int slgc_comm_destroy(slgc_ctx *ctx);
int slgc_comm_barrier(slgc_ctx *ctx);                                          /* all-reduce of one word + stream sync */
int slgc_comm_allreduce_max_f64(slgc_ctx *ctx, double *value);                 /* in place, host scalar */
int slgc_comm_allgather_i64(slgc_ctx *ctx, int64_t mine, int64_t *all);        /* all: host int64[nranks] */
/* What RCCL ITSELF reports for this communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) -- not what slgc_comm_init was told:
 * a job's report quotes these, so that "N ranks" is RCCL's statement.  Any output may be NULL. */
int slgc_comm_info(slgc_ctx *ctx, int *nranks, int *rank, int *device);
/* Every rank's PCI bus id ("0000:75:00.0"), all-gathered over the communicator: ids = nranks * SLGC_BUS_ID_BYTES NUL-padded bytes in rank
 * order; *distinct (may be NULL) = number of different devices.  N ranks of one node on N GPUs give distinct == N. */
#define SLGC_BUS_ID_BYTES 16
int slgc_comm_allgather_bus_ids(slgc_ctx *ctx, char *ids, int *distinct);
/* all-gatherv of byte records: rank r contributes counts[r] bytes from d_send; every rank receives all of them at
 * d_recv + displs[r].  Equal counts laid out back to back (displs[r] = r*count) run as ncclAllGather (in place when
 * d_send = d_recv + displs[rank]); anything else as one grouped ncclBroadcast per contributing rank (RCCL has no
 * native all-gatherv). */
int slgc_comm_allgatherv(slgc_ctx *ctx, const void *d_send, void *d_recv, const int64_t *counts, const int64_t *displs);
/* Split form for overlap: collectives run on the context's own communication stream, ordered after everything enqueued on
 * the compute stream at the time of the call.  _begin enqueues the all-gatherv and returns; kernels enqueued afterwards overlap
 * with it; slgc_comm_wait(slot) makes the compute stream wait for the exchange started in that slot (0..3). */
int slgc_comm_allgatherv_begin(slgc_ctx *ctx, const void *d_send, void *d_recv, const int64_t *counts, const int64_t *displs, int slot);
/* The same for two buffers that share one shard layout (the h and v maps), enqueued as one RCCL group. */
int slgc_comm_allgatherv_pair_begin(slgc_ctx *ctx, const void *d_send_a, void *d_recv_a, const void *d_send_b, void *d_recv_b,
                                    const int64_t *counts, const int64_t *displs, int slot);
int slgc_comm_wait(slgc_ctx *ctx, int slot);

/* ------------------------------------------------------------------ multi-GPU, direct exchange (all xGMI links at once)
 * The same exchange step without a ring: every rank pushes its band straight into every peer's full-size buffer (hipIpcMemHandle mappings),
 * G-1 band copies per rank in flight together -- xGMI on MI355X is a point-to-point mesh, a ring all-gather is bound by one link.  No RCCL
 * involved: set-up and flags go through one POSIX shared-memory segment named after `key` (the same string on every rank of the job, e.g.
 * the launcher's port + pid; at most 16 ranks, one node).  The reference has no counterpart (single process).
 *   slgc_direct_register       collective, same order on every rank: d_base = start of a slgc_dev_alloc buffer of `bytes` bytes (same size everywhere)
 *   slgc_direct_allgatherv_begin  nbuf = 1..3 registered buffers that each hold this rank's band at displs[i][rank] (counts[i][r] / displs[i][r]:
 *                              band layout of buffer i, int64 [nranks], bytes): enqueued on the exchange stream after the compute stream's work so
 *                              far; afterwards every peer holds the band at the same place of ITS buffer.  slot 0..3 names the exchange for _wait.
 *   slgc_direct_wait           compute stream waits until every peer's band of that exchange has arrived here
 *   slgc_direct_release        compute stream: work enqueued so far is done with what the last exchange left in these buffers; peers may overwrite.
 *                              Call it before re-using a buffer set (the first exchange on a buffer needs none).
 *   slgc_direct_barrier / _allgather_i64   small host-side collectives over the segment (both streams drained first).  Call slgc_direct_barrier before
 *                              slgc_direct_destroy (or slgc_destroy) and before freeing a registered buffer: a peer may still be pushing into it.
 *   slgc_direct_unregister     collective: the buffer leaves the exchange on every rank (mappings closed, slot free again; at most 16 buffers are
 *                              registered at a time).  slgc_dev_free refuses a buffer that is still registered.
 * Deadlines: a GPU-side poll gives up 20 s (SLGC_DIRECT_TIMEOUT_S) after the peer's HOST has submitted the work that will raise the flag, or
 * 300 s (SLGC_DIRECT_START_TIMEOUT_S) if the peer never gets that far -- ranks whose hosts are out of step wait for each other, like RCCL; host
 * barriers give up after 120 s (SLGC_DIRECT_HOST_TIMEOUT_S).  A timeout is sticky: the exchange's later kernels skip their work (nothing is pushed
 * into buffers that were not released, no half-filled buffer is announced as complete), and slgc_synchronize, slgc_d2h and every slgc_direct_*
 * call return SLGC_ECOMM from then on -- a scan whose exchange timed out never comes back as data.  A lost peer costs a failed call, never a hung GPU. */
int slgc_direct_init(slgc_ctx *ctx, int rank, int nranks, const char *key);
int slgc_direct_destroy(slgc_ctx *ctx);
int slgc_direct_register(slgc_ctx *ctx, void *d_base, size_t bytes);
int slgc_direct_unregister(slgc_ctx *ctx, void *d_base);      /* collective; before slgc_dev_free of a registered buffer (which refuses otherwise) */
int slgc_direct_allgatherv_begin(slgc_ctx *ctx, int nbuf, void *const *d_bases, const int64_t *const *counts, const int64_t *const *displs, int slot);
int slgc_direct_wait(slgc_ctx *ctx, int slot);
int slgc_direct_release(slgc_ctx *ctx, int nbuf, void *const *d_bases);
int slgc_direct_barrier(slgc_ctx *ctx);
int slgc_direct_allgather_i64(slgc_ctx *ctx, int64_t mine, int64_t *all);

/* Row-band plan of the sharded scan: contiguous bands, the first H % nranks ranks get one extra
 * row.  Pure arithmetic, no context needed. */
int slgc_shard_band(int H, int nranks, int rank, int *row0, int *rows);

/* One row-sharded scan in one call, nothing synchronises with the host ("maps" strategy): decode this rank's band
 * (d_band_stack = its first row of frame 0, band rows from slgc_shard_band with the context's rank / nranks) into its slot of
 * the full-size int16 maps, all-gatherv both maps in place over RCCL, triangulate the full maps.  Afterwards every rank
 * holds d_h_full / d_v_full [H][W] and dense float32 d_xyz_full [H][W][3] (NaN = undecodable) -- the reassembled cloud of
 * BASELINE.json configs[3].  Replaces, for N GPUs, src/3-capture_decode.py:75-100 + src/4-triangulate.py:50-64.  With slgc_tune("wire", 1)
 * and codes of <= SLGC_WIRE_MAX_CODE_BITS bits the bands travel packed to 3 B/pixel (one all-gather; unpacked inside the triangulation). */
int slgc_scan_sharded_dev(slgc_ctx *ctx, const uint8_t *d_band_stack, int n_runs, size_t run_stride, size_t plane_stride,
                          int N, int H, int W, int proj_w, int proj_h, double eps, double m, int mode, int16_t *d_h_full,
                          int16_t *d_v_full, float *d_xyz_full);

#ifdef __cplusplus
}
#endif
#endif /* SLGC_H */
