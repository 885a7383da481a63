/*
 * slgc_oracle.c -- plain-C CPU restatement of the reference's Gray-code decode +
 * triangulation path (guillaume-charron/3DScanner-GrayCode).
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/libslgc_oracle.so and
 * loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- as the checker /
 * the timed CPU baseline, never as the thing shipped.  The product library (libslgc.so) does
 * not link, load or call anything in this file.
 *
 * Pinning: every function here is checked against the .npz files under tests/golden, which hold outputs of the
 * reference itself (tests/golden/make_golden.py).  Exception: orc_undistort restates OpenCV 4.8's
 * cvUndistortPointsInternal (third-party, opencv-contrib-python==4.8.0.76, requirements.txt:4,
 * not installed here) -- PARITY UNPINNED for that one function.
 *
 * Literal fp64, no contraction (compile with -ffp-contract=off): the reference's own rounding
 * sequence is part of its behaviour.  file:line cites are relative to
 * /root/reference.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EINVAL (-1)

/* Per-pixel loops are independent; orc_set_threads(n > 1) runs them on n host threads (OpenMP) for bench.py's all-cores CPU
 * baseline.  Default 1: the tests use the scalar path.  Results do not depend on the thread count. */
static int orc_threads = 1;
int orc_set_threads(int n) { orc_threads = n < 1 ? 1 : n; return orc_threads; }
#define ORC_PAR _Pragma("omp parallel for schedule(static) num_threads(orc_threads) if(orc_threads > 1)")

static inline double px(const void *stack, int is_f64, size_t idx)
{
    return is_f64 ? ((const double *)stack)[idx] : (double)((const uint8_t *)stack)[idx];
}

/* scanner/grayCode/decode_codes.py:109-111 -- float pattern_len, uint8 truncation. */
int orc_frame_ids(int n_frames, int hid[6], int vid[6])
{
    if (n_frames < 14 || n_frames > 65) return ORC_EINVAL;
    double plf = (double)(n_frames - 2) / 4.0;
    double h[6] = {2 * plf - 2, 2 * plf - 4, 2 * plf - 6, 4 * plf - 2, 4 * plf - 4, 4 * plf - 6};
    double v[6] = {1, 3, 5, 2 * plf + 1, 2 * plf + 3, 2 * plf + 5};
    for (int k = 0; k < 6; ++k) {
        hid[k] = (int)(uint8_t)h[k];
        vid[k] = (int)(uint8_t)v[k];
    }
    return ORC_OK;
}

/* decode_codes.py:149 */
int orc_code_len(int n_frames) { return (int)((double)(n_frames - 2) / 4.0); }

/* decode_codes.py:90-122 (get_direct_indirect). */
int orc_direct_indirect(const void *stack, int is_f64, int N, int H, int W, double *Ld, double *Lg)
{
    int hid[6], vid[6];
    if (orc_frame_ids(N, hid, vid)) return ORC_EINVAL;
    size_t plane = (size_t)H * W;
    ORC_PAR
    for (size_t p = 0; p < plane; ++p) {
        double black = px(stack, is_f64, p), white = px(stack, is_f64, plane + p);
        double b_inv = white / (white + black);                                   /* :113 */
        double lmax = px(stack, is_f64, (size_t)(2 + hid[0]) * plane + p);
        double lmin = px(stack, is_f64, (size_t)(2 + vid[0]) * plane + p);
        for (int k = 1; k < 6; ++k) {
            /* np.max / np.min propagate NaN; irrelevant for uint8 data, kept for f64 stacks */
            double a = px(stack, is_f64, (size_t)(2 + hid[k]) * plane + p);
            double b = px(stack, is_f64, (size_t)(2 + vid[k]) * plane + p);
            lmax = (isnan(a) || isnan(lmax)) ? NAN : (a > lmax ? a : lmax);       /* :116 */
            lmin = (isnan(b) || isnan(lmin)) ? NAN : (b < lmin ? b : lmin);       /* :117 */
        }
        double ld = (lmax - lmin) * b_inv;                                        /* :119 */
        double t = 2.0 * (lmax - ld);                                             /* :120 */
        Ld[p] = ld;
        Lg[p] = t * b_inv;
    }
    return ORC_OK;
}

/* One (normal, inverse) pair through the rule table of decode_codes.py:162-182. */
static inline int8_t classify(double d, double g, double n, double i, double eps)
{
    int8_t c = -1;                                                /* :162-163 (rule 0 :169-170 is a no-op) */
    if (d > (g + eps) && n > (i + eps)) c = 1;                    /* :172-173 */
    if (d > (g + eps) && (n + eps) < i) c = 0;                    /* :175-176 */
    if ((n + eps) < d && i > (g + eps)) c = 0;                    /* :178-179 */
    if (n > (g + eps) && (i + eps) < d) c = 1;                    /* :181-182 */
    return c;
}

/* decode_codes.py:125-186 (get_is_lit); hc/vc are int8 [L][H][W]. */
int orc_is_lit(const void *stack, int is_f64, int N, int H, int W, const double *Ld, const double *Lg,
               double eps, double m, int8_t *hc, int8_t *vc)
{
    (void)m;
    if (N < 14 || N > 65) return ORC_EINVAL;
    int L = orc_code_len(N);
    size_t plane = (size_t)H * W;
    for (int k = 0; k < L; ++k) {
        size_t hn = (size_t)(2 + 2 * k) * plane, hi = (size_t)(2 + 2 * L + 2 * k) * plane;         /* :154,157,159 */
        size_t vn = (size_t)(2 + 2 * k + 1) * plane, vi = (size_t)(2 + 2 * L + 2 * k + 1) * plane; /* :155,158,160 */
        ORC_PAR
        for (size_t p = 0; p < plane; ++p) {
            hc[(size_t)k * plane + p] = classify(Ld[p], Lg[p], px(stack, is_f64, hn + p), px(stack, is_f64, hi + p), eps);
            vc[(size_t)k * plane + p] = classify(Ld[p], Lg[p], px(stack, is_f64, vn + p), px(stack, is_f64, vi + p), eps);
        }
    }
    return ORC_OK;
}

/* decode_codes.py:189-207 */
int64_t orc_gray_decode(int64_t n)
{
    int64_t s = n >> 1;
    while (s) {
        n ^= s;
        s >>= 1;
    }
    return n;
}

/* src/3-capture_decode.py:95-100 + decode_codes.py:209-229: max-merge R runs of codes
 * ([R][L][H][W] int8), then per pixel: any -1 -> -1, else MSB-first bits, Gray -> binary.
 * h uses codes as stored; v uses them flipped (stored LSB first). */
int orc_codes_to_pixels(const int8_t *hc, const int8_t *vc, int R, int L, int H, int W, int64_t *hp, int64_t *vp)
{
    if (R < 1 || L < 1 || L > 62) return ORC_EINVAL;
    size_t plane = (size_t)H * W, run = (size_t)L * plane;
    ORC_PAR
    for (size_t p = 0; p < plane; ++p) {
        int64_t gh = 0, gv = 0;
        int bad_h = 0, bad_v = 0;
        for (int k = 0; k < L; ++k) {
            int8_t a = hc[(size_t)k * plane + p], b = vc[(size_t)k * plane + p];
            for (int r = 1; r < R; ++r) {
                int8_t a2 = hc[r * run + (size_t)k * plane + p], b2 = vc[r * run + (size_t)k * plane + p];
                if (a2 > a) a = a2;
                if (b2 > b) b = b2;
            }
            if (a < 0) bad_h = 1;
            if (b < 0) bad_v = 1;
            gh |= (int64_t)(a & 1) << (L - 1 - k);
            gv |= (int64_t)(b & 1) << k;
        }
        hp[p] = bad_h ? -1 : orc_gray_decode(gh);
        vp[p] = bad_v ? -1 : orc_gray_decode(gv);
    }
    return ORC_OK;
}

/* get_codes per run -> merge -> maps.  stack holds R runs back to back ([R][N][H][W]). */
int orc_decode(const void *stack, int is_f64, int R, int N, int H, int W, double eps, double m, int64_t *hp, int64_t *vp)
{
    if (N < 14 || N > 65 || R < 1) return ORC_EINVAL;
    int L = orc_code_len(N);
    size_t plane = (size_t)H * W, esz = is_f64 ? 8 : 1;
    double *Ld = malloc(plane * 8), *Lg = malloc(plane * 8);
    int8_t *hc = malloc((size_t)R * L * plane), *vc = malloc((size_t)R * L * plane);
    int rc = (Ld && Lg && hc && vc) ? ORC_OK : ORC_EINVAL;
    for (int r = 0; r < R && !rc; ++r) {
        const char *st = (const char *)stack + (size_t)r * N * plane * esz;
        rc = orc_direct_indirect(st, is_f64, N, H, W, Ld, Lg);
        if (!rc) rc = orc_is_lit(st, is_f64, N, H, W, Ld, Lg, eps, m, hc + (size_t)r * L * plane, vc + (size_t)r * L * plane);
    }
    if (!rc) rc = orc_codes_to_pixels(hc, vc, R, L, H, W, hp, vp);
    free(Ld); free(Lg); free(hc); free(vc);
    return rc;
}

/* scanner/triangulation/triangulate.py:39-71 (get_cam_proj_pts).  order 0 = x-major (the
 * reference's scan, :52-53), 1 = row-major.  h/v are [cam_h][cam_w] int64.  white may be NULL.
 * cam/proj: float32 [M][2]; colors: float64 [M][3] (= u8 / 255.0, :69).  Returns M (or <0). */
int64_t orc_cam_proj_pts(const int64_t *h, const int64_t *v, int cam_w, int cam_h, int proj_w, int proj_h,
                         const uint8_t *white, int order, float *cam, float *proj, double *colors)
{
    int64_t M = 0;
    int outer = order == 0 ? cam_w : cam_h, inner = order == 0 ? cam_h : cam_w;
    for (int a = 0; a < outer; ++a)
        for (int b = 0; b < inner; ++b) {
            int x = order == 0 ? a : b, y = order == 0 ? b : a;
            int64_t hv = h[(size_t)y * cam_w + x], vv = v[(size_t)y * cam_w + x];
            if (hv == -1 || vv == -1) continue;                                   /* :56 */
            if (cam) { cam[2 * M] = (float)x; cam[2 * M + 1] = (float)y; }        /* :59 */
            if (proj) {
                proj[2 * M] = (float)(hv < proj_w - 1 ? hv : proj_w - 1);         /* :60 */
                proj[2 * M + 1] = (float)(vv < proj_h - 1 ? vv : proj_h - 1);     /* :61 */
            }
            if (white && colors)
                for (int c = 0; c < 3; ++c) colors[3 * M + c] = (double)white[((size_t)y * cam_w + x) * 3 + c] / 255.0;
            ++M;
        }
    return M;
}

/* OpenCV 4.8 cv::undistortPoints(src, K, dist, R) with the default TermCriteria(MAX_ITER,5,.01)
 * -- calib3d/src/undistort.dispatch.cpp, cvUndistortPointsInternal.  PARITY UNPINNED (see header).
 * pts/out float32 [M][2]; dist has ndist (<=14) coefficients (k1,k2,p1,p2,k3,k4,k5,k6,s1..s4,tx,ty;
 * tilt terms must be 0); R may be NULL. */
int orc_undistort(const float *pts, int64_t M, const double K[9], const double *dist, int ndist, const double *R, float *out)
{
    double k[14] = {0};
    if (ndist < 0 || ndist > 14) return ORC_EINVAL;
    for (int j = 0; j < ndist; ++j) k[j] = dist[j];
    if (k[12] != 0 || k[13] != 0) return ORC_EINVAL;
    double RR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (R) memcpy(RR, R, sizeof RR);
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5], ifx = 1. / fx, ify = 1. / fy;
    ORC_PAR
    for (int64_t q = 0; q < M; ++q) {
        double u = pts[2 * q], v = pts[2 * q + 1];
        double x = (u - cx) * ifx, y = (v - cy) * ify, x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) {
                x = (u - cx) * ifx;
                y = (v - cy) * ify;
                break;
            }
            double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            double dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        double xx = RR[0] * x + RR[1] * y + RR[2];
        double yy = RR[3] * x + RR[4] * y + RR[5];
        double ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
        out[2 * q] = (float)(xx * ww);
        out[2 * q + 1] = (float)(yy * ww);
    }
    return ORC_OK;
}

/* triangulate.py:73-97.  cam_n/proj_n are the float32 undistorted points ([M][2], homogeneous 1
 * appended here, :84-85).  xyz is float64 (3,M) row-major (X row, Y row, Z row), as the reference
 * returns.  float32 where NumPy stays in float32 (:90 and the norm inside :92). */
int orc_law_of_sines(const float *cam_n, const float *proj_n, int64_t M, const double T[3], double *xyz)
{
    double t_len = sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);                /* :89 */
    ORC_PAR
    for (int64_t q = 0; q < M; ++q) {
        float cx = cam_n[2 * q], cy = cam_n[2 * q + 1], cz = 1.0f;
        float cn = sqrtf((cx * cx + cy * cy) + cz * cz);
        float rx = cx / cn, ry = cy / cn, rz = cz / cn;                           /* :90 (float32) */
        float qx = proj_n[2 * q], qy = proj_n[2 * q + 1], qz = 1.0f;
        float qn = sqrtf((qx * qx + qy * qy) + qz * qz);
        double alpha = acos(((-T[0]) * rx + (-T[1]) * ry + (-T[2]) * rz) / t_len);               /* :91 */
        double beta = acos((T[0] * qx + T[1] * qy + T[2] * qz) / (t_len * (double)qn));          /* :92 */
        double gamma = M_PI - alpha - beta;                                                       /* :93 */
        double len = t_len * sin(beta) / sin(gamma);                                              /* :94 */
        xyz[q] = (double)rx * len;                                                                /* :95 */
        xyz[M + q] = (double)ry * len;
        xyz[2 * M + q] = (double)rz * len;
    }
    return ORC_OK;
}

int orc_triangulate(const float *cam_pts, const float *proj_pts, int64_t M, const double camK[9], const double *cam_dist,
                    int n_cam_dist, const double projK[9], const double *proj_dist, int n_proj_dist, const double R[9],
                    const double T[3], double *xyz)
{
    float *a = malloc((size_t)(M ? M : 1) * 2 * sizeof(float)), *b = malloc((size_t)(M ? M : 1) * 2 * sizeof(float));
    int rc = (a && b) ? ORC_OK : ORC_EINVAL;
    if (!rc) rc = orc_undistort(cam_pts, M, camK, cam_dist, n_cam_dist, R, a);     /* :84 (R = proj_R) */
    if (!rc) rc = orc_undistort(proj_pts, M, projK, proj_dist, n_proj_dist, NULL, b); /* :85 */
    if (!rc) rc = orc_law_of_sines(a, b, M, T, xyz);
    free(a); free(b);
    return rc;
}

/* triangulate.py:99-122: strict box filter; writes kept columns compacted, returns count.
 * colors ([M][3] float64) may be NULL. */
int64_t orc_filter(const double *xyz, const double *colors, int64_t M, double thr, double *xyz_out, double *colors_out)
{
    int64_t kept = 0;
    for (int64_t q = 0; q < M; ++q) {
        double X = xyz[q], Y = xyz[M + q], Z = xyz[2 * M + q];
        if (Z < thr && Z > -thr && Y < thr && Y > -thr && X < thr && X > -thr) ++kept;            /* :119 */
    }
    int64_t w = 0;
    for (int64_t q = 0; q < M; ++q) {
        double X = xyz[q], Y = xyz[M + q], Z = xyz[2 * M + q];
        if (!(Z < thr && Z > -thr && Y < thr && Y > -thr && X < thr && X > -thr)) continue;
        if (xyz_out) { xyz_out[w] = X; xyz_out[kept + w] = Y; xyz_out[2 * kept + w] = Z; }
        if (colors && colors_out) memcpy(colors_out + 3 * w, colors + 3 * q, 3 * sizeof(double));
        ++w;
    }
    return kept;
}

/* Whole path on dense maps, for checking the fused device kernel: decode -> clamp ->
 * triangulate every pixel; invalid pixels get NaN.  xyz is [3][H][W] float64. */
int orc_scan_dense(const void *stack, int is_f64, int R, int N, int H, int W, double eps, double m, int proj_w,
                   int proj_h, const double camK[9], const double *cam_dist, int n_cam_dist, const double projK[9],
                   const double *proj_dist, int n_proj_dist, const double Rm[9], const double T[3], int64_t *hp,
                   int64_t *vp, double *xyz)
{
    int rc = orc_decode(stack, is_f64, R, N, H, W, eps, m, hp, vp);
    if (rc) return rc;
    size_t plane = (size_t)H * W;
    float *cam = malloc(plane * 2 * sizeof(float)), *proj = malloc(plane * 2 * sizeof(float));
    if (!cam || !proj) { free(cam); free(proj); return ORC_EINVAL; }
    ORC_PAR
    for (size_t p = 0; p < plane; ++p) {
        int64_t hv = hp[p], vv = vp[p];
        cam[2 * p] = (float)(p % W);
        cam[2 * p + 1] = (float)(p / W);
        proj[2 * p] = (float)(hv < 0 ? 0 : (hv < proj_w - 1 ? hv : proj_w - 1));
        proj[2 * p + 1] = (float)(vv < 0 ? 0 : (vv < proj_h - 1 ? vv : proj_h - 1));
    }
    rc = orc_triangulate(cam, proj, (int64_t)plane, camK, cam_dist, n_cam_dist, projK, proj_dist, n_proj_dist, Rm, T, xyz);
    if (!rc) {
        ORC_PAR
        for (size_t p = 0; p < plane; ++p)
            if (hp[p] == -1 || vp[p] == -1) xyz[p] = xyz[plane + p] = xyz[2 * plane + p] = NAN;
    }
    free(cam); free(proj);
    return rc;
}
