/*
 * slgc_bench.h -- measurement, diagnostic and A/B exports of libslgc.so.  NOT part of the reference-facing ABI (include/slgc.h): a binding of
 * the reference's functions needs nothing from this header.  bench.py, tools/ and the GPU tests use it: synthetic captures generated on the
 * device (each with a bit-identical NumPy twin in oracle/oracle_np.py), exhaustive device self-tests of the folded decode arithmetic, the
 * movement-only yardstick, HIP-event timing, per-launch kernel timing, and slgc_tune's knobs.
 */
#ifndef SLGC_BENCH_H
#define SLGC_BENCH_H

#include "slgc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ tile-interleaved stack layout: the layout question, kept reproducible
 * SURVEY.md D5 leaves the device layout of the frame stack to the build; the product reads the reference's planar [N][H][W]
 * (src/3-capture_decode.py:68-70).  Round 6 asked whether N plane streams a whole image apart cost the decode kernel its last 8-10 %.  Movement-only
 * kernels (tools/ubench/stream_rates.hip rows D / Dc / Dt / Dtk) move the same bytes 2-6 % faster from a TILE-INTERLEAVED stack
 * [tile][N][2^k bytes] -- pixel p of frame f at ((p >> k) * N + f) << k | (p & (2^k - 1)) -- at 4096x3000 and no faster at 1920x1080; the REAL
 * kernels, which can read that layout through the exports below, do not (tools/time_tiled.py, profiles/r06_tiled_layout.txt: decode and fused
 * scan within +-1 % of planar at k = 10..16, results bit-identical).  So planar stays the product's layout and this stays an A/B:
 * slgc_tune(ctx, "stack_tile_log2", k) (k = 8..24; 0 = planar, default) makes slgc_decode_dev, slgc_scan_dev, slgc_scan_batch_dev and
 * slgc_cloud_dev / slgc_cloud32_dev of that context read tile-interleaved stacks: pass plane_stride = 2^k and d_stack = the band's first tile
 * (run_stride / scan_stride = bytes between whole tiled stacks); bands hold a multiple of 4 pixels, start on a tile, stay under 4 GB.  Host-buffer
 * and BGR entry points are unaffected (their stacks are planar).  A stack gets into the layout from BGR frames at no extra pass
 * (slgc_to_gray_tiled_dev: cv2.cvtColor BGR2GRAY as slgc_to_gray_dev, d_bgr = n_frames frames of npix pixels back to back) or from a planar
 * grey stack with one copy pass (slgc_tile_stack_dev). */
int slgc_tiled_stack_bytes(int N, size_t npix, int tile_log2, size_t *bytes);   /* ceil(npix / 2^k) * N * 2^k: allocate this much */
int slgc_tile_stack_dev(slgc_ctx *ctx, const uint8_t *d_planar, size_t plane_stride, int N, size_t npix, int tile_log2, uint8_t *d_tiled);
int slgc_to_gray_tiled_dev(slgc_ctx *ctx, const uint8_t *d_bgr, int n_frames, size_t npix, int coeff_bits, int tile_log2, uint8_t *d_tiled);

/* ------------------------------------------------------------------ A/B knobs */
/* slgc_tune (declared in slgc.h) -- the knobs for same-process A/B timing; no setting but the last one named here changes any result.  "fuse_tail" 1 = wave-local LDS exchange in the fused
 * scan kernel's tail (default) / 0 = workgroup-wide; "proj_tile" 1 = 16x8-pixel projector-table tiles (default) / 0 = 8x8;
 * "park" 1 = at 42 / 44 / 46 / 50 / 54 frames the kernels park the 12 threshold frames in LDS instead of fetching them twice (default) /
 * 0 = generic kernels; "wire" 1 = slgc_scan_sharded_dev exchanges the maps in the 3-byte wire format / 0 = int16 (default;
 * experimental until measured on real xGMI); "fuse_nt" bit 0 XYZ, bit 1 maps non-temporal in the fused kernel (default -1: 1 from 4 Mpixels per launch up, 3 below);
 * "tri_nt" (1); "xcd" XCD-aware tile map of the dense triangulation kernel (1), "fuse_xcd" the same for the fused scan kernel (0 = off,
 * the default: 1 = one band of rows per XCD takes the kernel's traffic from 823 to 773 MB and its time from 127 to 137 us at
 * 4096x3000x44; n >= 2 = n consecutive tiles per XCD inside groups of 8 n: 782 MB and +2 % at n = 512).  The one knob that is NOT bit-neutral: "cam_nodes"
 * 1 = the scan kernels' fast form interpolates the camera rays from the every-4th-column table when the per-pixel table is too large
 * to pay as a per-pixel stream (> 12 MB: 1920x1080 and up; default, see slgc_ray_table_info: rays within 2 float32 ulp of the exact
 * ones, XYZ inside the 1e-4 tolerance, maps untouched) / 2 = whenever that table is accurate enough / 0 = reads the per-pixel table.
 * "lists_order" = workgroup -> tile order of the x-major list build, 0 .. 64, default 4: 0 row-major; 1 column-major (a column's run continues in the tile
 * below, so the seams are written close together in time; 217.6 -> 205.6 us at 4096x3000); 2 column-major inside each XCD (no better); n >= 3 (the
 * whole-lines scatter k_xmajor_lines only) column-major in runs of n consecutive tiles per XCD, so that a tile and the tile below it -- whose rows it also
 * reads -- share an L2 (n = 4: 218.8 -> 210.5 us).  The two scatter kernels read the value differently: the tile-run scatter k_xmajor_scatter treats
 * every value other than 0 and 2 as plain column-major (1).
 * "prio" = head * 100 + body * 10 + tail: s_setprio (0..3) of a wave of the fused scan kernel while it fetches its threshold frames and computes its thresholds /
 * walks the bit loop / runs the triangulation tail; -1 (default) = by launch shape: 210 for launches that fill more than half of the chip's 8 192 wave slots in one
 * round (1920x1080: -3.5 %), 0 otherwise (notes/r05.md).  The setting assumes the launch has the GPU to itself: contexts whose scans overlap on
 * the device (several streams) should set 0 -- two streams of 1920x1080 scans lose 12 % with "tail last".
 * "guard_list" 1 (default) = the fused scan kernel compacts its flat triangles over the wave and redoes 64 of them per float64 pass / 0 = redoes them lane by
 * lane (bit-identical XYZ; scattered wrong codes make the lane-by-lane form walk the float64 path in nearly every wave).
 * "lists_lines" 1 (default) = slgc_cloud_dev's list build writes whole aligned 128-byte lines (k_xmajor_lines; see slgc_last_list_kernel) for images of at
 * least 2048 tiles of 64 x 32 pixels (below that the tile-run kernel is the faster one) / 2 = wherever the shape allows / 0 = tile runs (A/B).
 * "image_rows" H > 0 = this context scans row bands of an image of H rows (the multi-GPU plan): the "cam_nodes" decision -- table size
 * and accuracy check -- is then taken for the WHOLE image, so a pixel's XYZ is bit-identical whether one GPU scans the image or N GPUs
 * scan its bands (0, the default: the band is the image).
 * Defaults can also be set with the environment (SLGC_FUSE_TAIL, SLGC_PROJ_TILE, SLGC_PARK, SLGC_FUSE_NT, SLGC_TRI_NT, SLGC_XCD,
 * SLGC_FUSE_XCD, SLGC_PRIO, SLGC_CAM_NODES, SLGC_LISTS_LINES, SLGC_LISTS_ORDER, SLGC_GUARD_LIST), read when the context is created.  Other environment
 * switches (read once per process, A/B only): SLGC_FD_SEGS (segment count of the frame-difference kernel), SLGC_PAR_DOWNLOAD, SLGC_F64_PACK. */


/* PCI address of the context's device, "0000:c1:00.0" (how bench.py finds the device's clock / busy nodes under /sys/bus/pci/devices). */
int slgc_device_pci_bus_id(slgc_ctx *ctx, char *buf, int buflen);


/* ------------------------------------------------------------------ device self-tests of the folded arithmetic */
/* Diagnostic.  The device-resident decode kernels fold the per-pixel fp64 quantities of decode_codes.py:113-120 into integer
 * thresholds; this checks, for every (black, white, L_max, L_min) with black in [black_lo, black_hi) and the other three over
 * 0..255, that the folded tests agree with the literal fp64 comparisons of :172-182 for every grey level 0..255.
 * *mismatches receives the number of disagreements (0 expected).  black_lo = 0, black_hi = 256 is the whole uint8 domain.
 * Negative control: eps | 0x100 evaluates the literal side with eps + 1 (the count must then be non-zero). */
int slgc_selftest_thresholds(slgc_ctx *ctx, int eps, int black_lo, int black_hi, unsigned long long *mismatches);

/* Diagnostic.  The packed 16-bit evaluation of the rule table (two pixels per register, decode_codes.py:162-182 "last match
 * wins") against the scalar rules for every threshold triple (tnd, tg in 0..256, cA in 1..256 or "not direct") and every
 * (normal, inverse) grey-level pair: 1.1e12 classifications; *mismatches = 0 expected.  negative_control != 0 shifts one
 * threshold of the scalar side by one (the count must then be non-zero). */
int slgc_selftest_classify(slgc_ctx *ctx, int negative_control, unsigned long long *mismatches);

/* Diagnostic: d_counts[0] += decodable pixels of the band, d_counts[1] += those among them that the dense triangulation redoes
 * on the reference's float32 intermediates because the triangle is flat (tri_is_flat, csrc/tri_math.h) -- the guarded slow path. */
int slgc_guard_count_dev(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, int rows, int W, int row0, int proj_w, int proj_h,
                         unsigned long long *d_counts);


/* ------------------------------------------------------------------ synthetic captures, yardstick, timing */
/* Synthetic capture written straight into HBM (SURVEY.md section 8(d) "S-scene", counter-based noise). */
int slgc_synth_scene_dev(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows,
                         uint32_t seed, int noise, int shadow);

/* Measurement yardstick, not a product: moves the bytes of one single-run scan of N = 42 / 44 / 46 frames and does nothing else -- N planes
 * read 4 bytes per lane and plane like the scan kernels read them, the two int16 maps (d_h / d_v, both or neither) and 12 bytes per pixel
 * at d_xyz (may be NULL) written in the scan kernels' store shapes; what lands there is meaningless.  bench.py times it beside the kernel it
 * grades: on MI355X the scan kernels take what this takes (tools/ubench/stream_rates.hip is the longer study).  npix % 256 == 0. */
int slgc_move_only_dev(slgc_ctx *ctx, const uint8_t *d_stack, size_t plane_stride, int N, size_t npix, int16_t *d_h, int16_t *d_v, float *d_xyz);

/* Physically consistent synthetic capture (needs slgc_set_calibration): every camera pixel's ray is cast into a fixed scene (a tilted plane
 * with a sphere in front, 0.4 - 0.65 m away), the hit point goes through the stereo pose and the projector's forward lens model to the
 * projector pixel that lights it, and the frames encode THAT pixel -- one surface seen by camera and projector, which is what
 * src/4-triangulate.py:50-64 assumes of its inputs.  Pixels the projector does not reach stay at ambient level in every frame.
 * Optional outputs (device, may be NULL): d_h_true / d_v_true int16 [rows][W] = the encoded projector pixel (-1 = unlit);
 * d_truth_xyz float32 [rows][W][3] = the true surface point in the frame Triangulate.triangulate reports (NaN = unlit).
 * d_stack may be NULL (codes / truth only).  Bit-identical NumPy twin: oracle/oracle_np.py synth_physical. */
int slgc_synth_physical_dev(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows, int proj_w,
                            int proj_h, uint32_t seed, int noise, int16_t *d_h_true, int16_t *d_v_true, float *d_truth_xyz);

/* The same generator with its knobs: gain_lo / gain_hi = the two surface gains of its 16-pixel checker (frames = 15 + gain * bit + noise;
 * slgc_synth_physical_dev uses 140 / 180), r2_max = the squared radius in normalised projector coordinates up to which the projector's lens
 * model is taken to be monotonic (0.16 for the reference's `proj` calibration, which folds over at ~0.22; a mild lens can take 1.5).
 * Twin: oracle_np.synth_physical(..., gains=, r2_max=). */
int slgc_synth_physical_ex_dev(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows, int proj_w,
                               int proj_h, uint32_t seed, int noise, int gain_lo, int gain_hi, double r2_max, int16_t *d_h_true,
                               int16_t *d_v_true, float *d_truth_xyz);

/* SURVEY.md section 8(d) "S-uniform": every byte of every frame uniform in 0..255 (counter hash keyed by frame, dword of the whole image and
 * seed: a band holds the bytes of the same rows of the whole image).  W, plane_stride, d_stack multiples of 4.  Twin: oracle_np.synth_uniform. */
int slgc_synth_uniform_dev(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows, uint32_t seed);

/* A synthetic BGR capture of a grey stack (what the camera hands src/3-capture_decode.py:66): B = clip(g + ((7 x + 3 y) mod 11) - 5), G = g,
 * R = clip(g - (((5 x + 11 y) mod 9) - 4)) with y counted in the whole image.  d_bgr: [N][rows][W][3], bgr_plane_stride bytes between frames.
 * Twin: oracle_np.gray_to_bgr_capture. */
int slgc_synth_bgr_dev(slgc_ctx *ctx, const uint8_t *d_gray, size_t gray_plane_stride, int N, int H, int W, int row0, int rows, uint8_t *d_bgr,
                       size_t bgr_plane_stride);

/* HIP-event timing on the context's stream: id in [0,16). */
int slgc_event_record(slgc_ctx *ctx, int id);
int slgc_event_elapsed_ms(slgc_ctx *ctx, int id_start, int id_stop, float *ms);

/* Per-launch timing of the decode kernel: between _begin and _end every stride-th decode launch made through slgc_decode_dev /
 * slgc_scan_dev is bracketed by a HIP event pair on the context's stream (up to max_launches pairs); _end synchronises and
 * returns the summed kernel time and the number of launches sampled. */
int slgc_prof_begin(slgc_ctx *ctx, int max_launches, int stride);
int slgc_prof_end(slgc_ctx *ctx, double *total_ms, int *launches);
/* After _end: the individual kernel durations (ms) of the sampled launches, in launch order; *n = number sampled (may exceed cap). */
int slgc_prof_samples(slgc_ctx *ctx, float *ms, int cap, int *n);


#ifdef __cplusplus
}
#endif
#endif /* SLGC_BENCH_H */
