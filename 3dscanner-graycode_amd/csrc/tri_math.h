// tri_math.h -- device helpers shared by triangulate.hip and the fused scan kernel in decode.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct Ray2 {
    float x, y;
};


struct Xyz {
    double x, y, z;
};


// fp64 reciprocal / square root from the hardware seeds plus ONE Newton step (~1e-13 relative): the results feed a
// float32 XYZ, so the IEEE divide / sqrt expansions (and a second Newton step) would buy nothing.
__device__ __forceinline__ double fast_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}

__device__ __forceinline__ double fast_sqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    const double s = fma(g, fma(-h, g, 0.5), g);
    return x > 0.0 ? s : 0.0;
}

// Algebraic form of triangulate.py:86-95 with the normalisations cancelled (c = camera ray [cx, cy, 1], p = projector ray):
//   A = -T.c, B = T.p, Sa = sqrt(|T|^2 |c|^2 - A^2) = |T||c| sin(alpha), Sb likewise for beta,
//   sin(gamma) = sin(alpha + beta)   =>   Pts = c * |T|^2 * Sb / (Sa*B + A*Sb).
// The dot products, the two differences under the roots and the denominator (where cancellation can occur) are fp64; the
// final scale is rounded to float32 once and applied in float32 (triangulate4 below).  Agrees with the acos/sin form to ~2e-7
// relative (float32 output resolution) away from degenerate geometry; near it, see law_of_sines_mirror.
struct Xyzf {
    float x, y, z;
};

// Same cancelled form, but on the reference's own float32 intermediates (triangulate.py:90 NormedL and the float32 norm inside
// :92 are float32 in NumPy): cos(alpha) and cos(beta) are then the reference's values bit for bit, and only arccos/sin are
// replaced by sin = sqrt(1 - cos^2), sin(gamma) = sin(alpha + beta).  Tracks the reference to ~1e-9 even where its float32
// rounding dominates the answer (rays nearly parallel to each other or to the baseline).
__device__ __forceinline__ Xyzf law_of_sines_mirror(Ray2 cam, Ray2 prj, const double (&T)[3], double t_len)
{
    // plain operators: correctly rounded under -fno-fast-math -ffp-contract=off (the __f*_rn intrinsics map to native approximations)
    const float cn = sqrtf((cam.x * cam.x + cam.y * cam.y) + 1.0f);       // np.linalg.norm of float32 (:90)
    const float rx = cam.x / cn, ry = cam.y / cn, rz = 1.0f / cn;         // NormedL (float32)
    const float qn = sqrtf((prj.x * prj.x + prj.y * prj.y) + 1.0f);       // norm inside :92 (float32)
    const double it = fast_rcp(t_len);
    const double cos_a = (((-T[0]) * (double)rx + (-T[1]) * (double)ry) + (-T[2]) * (double)rz) * it;
    const double cos_b = ((T[0] * (double)prj.x + T[1] * (double)prj.y) + T[2]) * fast_rcp(t_len * (double)qn);
    const double sin_a = fast_sqrt(fma(-cos_a, cos_a, 1.0)), sin_b = fast_sqrt(fma(-cos_b, cos_b, 1.0));
    const double len = t_len * sin_b * fast_rcp(fma(sin_a, cos_b, cos_a * sin_b));
    return Xyzf{(float)((double)rx * len), (float)((double)ry * len), (float)((double)rz * len)};
}

// Shipped evaluation for a lane's four pixels: the fast form, except where the triangle is so flat (rays nearly parallel to
// each other or to the baseline) that the reference's float32 NormedL / norm roundings, amplified by the cotangents of its
// angles, move ITS answer by an amount that matters against the 1e-4 tolerance (criterion below).  Those pixels are
// redone with the mirrored form -- in ONE not-unrolled loop behind a lane-level "any of my four" test, so the common path pays
// for the two comparisons only and the instruction stream holds a single copy of the float32 divide / sqrt expansions.
// valid: bit j set = pixel j decodable; out = x0 y0 z0 x1 ... (NaN where not decodable).  GUARD = false: fast form everywhere.
// cam4 / proj_lut + idx: where the lane's rays came from -- the rare path reads them again instead of keeping 16 registers alive.
// ---- camera rays without the 8 B/pixel table --------------------------------------------------------------------------------
// The camera call of triangulate.py:84 is cv2.undistortPoints(pixel, cam_mtx, cam_dist, R = proj_R): undistort (5 fixed-point steps,
// float64), rotate by R, divide by the third component, round to float32.  The undistorted point U(x, y) BEFORE the rotation is a
// smooth, nearly linear function of the pixel, so it is kept as one bicubic (10 terms per component) per TS x TS pixel tile (TS = 16
// or 8), least-squares fitted at table-build time to the exact float64 U of every pixel of the tile; the rotation and the perspective
// divide -- the strongly curved part -- are evaluated exactly, in float64, per pixel.  96 bytes per tile (0.375 B/pixel at TS = 16)
// replace the 8 B/pixel stream.  The build measures the fit error max |ray_poly - ray_exact| against the exact ray BEFORE its
// float32 rounding, over the band, and the table is only used when that is <= kCamPolyTol; otherwise the exact table stays in use.
// A polynomial ray is therefore the reference's ray without its final float32 rounding (+ <= 1e-8): one more half-ulp perturbation
// of the camera ray, which the guard's error budget counts (tri_is_flat).
struct CamPolyTile {          // constant terms in float64; the nine higher terms of each component in float32, in the order
    double c0x, c0y;          //   dy, dy^2, dy^3,  dx, dx dy, dx dy^2,  dx^2, dx^2 dy,  dx^3        (dx, dy from the tile centre)
    float cx[9];
    float cy[9];
    float pad[2];
};
static_assert(sizeof(CamPolyTile) == 96, "six 16-byte words per tile");
constexpr double kCamPolyTol = 1.0e-8;
constexpr int kCamPolyTerms = 10;
// term k of the fit basis: (power of dx, power of dy); k = 0 is the constant, k = 1..9 follow the order of CamPolyTile::cx
__host__ __device__ constexpr int cam_poly_pow_x(int k) { return k < 4 ? 0 : k < 7 ? 1 : k < 9 ? 2 : 3; }
__host__ __device__ constexpr int cam_poly_pow_y(int k) { return k < 4 ? k : k < 7 ? k - 4 : k < 9 ? k - 7 : 0; }

struct CamPolyRef {           // what a kernel needs to evaluate camera rays from the tile table (nullptr tiles = use the exact table)
    const CamPolyTile *tiles;
    int tiles_x, shift;       // tile = (x >> shift, y_local >> shift), TS = 1 << shift
    int W;                    // image width (pixels per row of the band)
    double R[9];
};

// Along the row dy of a tile U(dx) = a[0] + dx * (a[1] + dx * (a[2] + dx * a[3])).
__device__ __forceinline__ void cam_poly_row(const double c0, const float (&c)[9], double dy, double (&a)[4])
{
    a[0] = fma(dy, fma(dy, fma(dy, (double)c[2], (double)c[1]), (double)c[0]), c0);
    a[1] = fma(dy, fma(dy, (double)c[5], (double)c[4]), (double)c[3]);
    a[2] = fma(dy, (double)c[7], (double)c[6]);
    a[3] = (double)c[8];
}

// Undistorted pre-rotation point at dx along the row, then R and the perspective divide: the float64 ray.
__device__ __forceinline__ void cam_poly_ray(const double (&ax)[4], const double (&ay)[4], double dx, const double (&R)[9], double &rx, double &ry)
{
    const double xu = fma(dx, fma(dx, fma(dx, ax[3], ax[2]), ax[1]), ax[0]), yu = fma(dx, fma(dx, fma(dx, ay[3], ay[2]), ay[1]), ay[0]);
    const double nx = fma(R[0], xu, fma(R[1], yu, R[2])), ny = fma(R[3], xu, fma(R[4], yu, R[5]));
    const double iw = fast_rcp(fma(R[6], xu, fma(R[7], yu, R[8])));
    rx = nx * iw;
    ry = ny * iw;
}

// The four consecutive pixels of a lane (first pixel `pix` of the band, a multiple of 4; W % 4 == 0 so they share a row and, being
// aligned to 4, a tile).
__device__ __forceinline__ void cam_rays4_poly(const CamPolyRef &cp, uint32_t pix, double (&cx)[4], double (&cy)[4])
{
    const uint32_t y = pix / (uint32_t)cp.W, x = pix - y * (uint32_t)cp.W;
    const int half = 1 << (cp.shift - 1), mask = (1 << cp.shift) - 1;
    const CamPolyTile t = cp.tiles[(size_t)(y >> cp.shift) * cp.tiles_x + (x >> cp.shift)];
    double ax[4], ay[4];
    const double dy = (double)((int)(y & mask) - half);
    cam_poly_row(t.c0x, t.cx, dy, ax);
    cam_poly_row(t.c0y, t.cy, dy, ay);
    const int dx0 = (int)(x & mask) - half;
#pragma unroll
    for (int j = 0; j < 4; ++j) cam_poly_ray(ax, ay, (double)(dx0 + j), cp.R, cx[j], cy[j]);
}

// Terms of the cancelled form for one pixel (shared by triangulate4 and the guard-count diagnostic so both apply the same test).
struct TriTerms {
    double ta, tb, ra, rb, Sb, D;
};

__device__ __forceinline__ TriTerms tri_terms(double dcx, double dcy, float pxf, float pyf, const double (&T)[3], double tl2)
{
    const double dpx = pxf, dpy = pyf;
    const double A = -fma(T[0], dcx, fma(T[1], dcy, T[2]));
    const double B = fma(T[0], dpx, fma(T[1], dpy, T[2]));
    TriTerms t;
    t.ta = tl2 * fma(dcx, dcx, fma(dcy, dcy, 1.0));
    t.tb = tl2 * fma(dpx, dpx, fma(dpy, dpy, 1.0));
    t.ra = fma(-A, A, t.ta);
    t.rb = fma(-B, B, t.tb);
    const double Sa = fast_sqrt(t.ra);
    t.Sb = fast_sqrt(t.rb);
    t.D = fma(Sa, B, A * t.Sb);
    return t;
}

// |d len / len| <= eps * [1 / sin^2(beta) + (1 / sin(alpha) + 1 / sin(beta)) / sin(gamma)], where eps bounds the error of
// cos(alpha), cos(beta) caused by float32 steps the fast form does not reproduce: in the reference, sqrt and three divisions for
// NormedL and one sqrt for the projector norm (3 * 2^-24); here, when the camera ray comes from the tile polynomials, the ray's own
// float32 rounding that the reference has and the polynomial does not (sqrt(2) * 2^-24) plus the fit error (<= 1e-8): eps <= 2.7e-7.
// A pixel is "flat" (redone on the reference's float32 intermediates, exact table rays) when either term can pass kGuardAmp = 170: the
// fast form is then never further than 2 * 170 * 2.7e-7 = 9.2e-5 from the reference, inside the 1e-4 tolerance even if every
// rounding aligns.
//   1 / sin^2(beta) > A                                  <=>  rb < tb / A
//   (1/sin a + 1/sin b) / sin(gamma) can exceed A         <=   min(sin a, sin b) * sin(gamma) < 2 / A
//                                                         <=>  D^2 * min(ra*tb, rb*ta) < (2 / A)^2 * (ta*tb)^2
constexpr double kGuardAmp = 170.0;
__device__ __forceinline__ bool tri_is_flat(const TriTerms &t)
{
    constexpr double k1 = 1.0 / kGuardAmp, k2 = (2.0 / kGuardAmp) * (2.0 / kGuardAmp);
    const double tatb = t.ta * t.tb;
    return (t.rb < k1 * t.tb) | ((t.D * t.D) * fmin(t.ra * t.tb, t.rb * t.ta) < (k2 * tatb) * tatb);     // no short-circuit: no branches
}

// cx / cy: camera rays in float64 (the exact table's float32 values widened, or the tile-polynomial rays).
template <bool GUARD>
__device__ __forceinline__ void triangulate4(const double (&cx)[4], const double (&cy)[4], const float (&px)[4], const float (&py)[4],
                                             uint32_t valid, const double (&T)[3], double t_len, float (&out)[12],
                                             const float2 *__restrict__ cam4, const float2 *__restrict__ proj_lut, const uint32_t (&idx)[4])
{
    const double tl2 = t_len * t_len;
    uint32_t ill = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const TriTerms t = tri_terms(cx[j], cy[j], px[j], py[j], T, tl2);
        const float s = (float)(tl2 * t.Sb * fast_rcp(t.D));
        if (GUARD) ill |= tri_is_flat(t) ? (1u << j) : 0u;
        const bool ok = (valid >> j) & 1u;
        out[3 * j] = ok ? (float)cx[j] * s : __builtin_nanf("");
        out[3 * j + 1] = ok ? (float)cy[j] * s : __builtin_nanf("");
        out[3 * j + 2] = ok ? s : __builtin_nanf("");
    }
    if (GUARD) {
        ill &= valid;
        if (ill) {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                if (!((ill >> j) & 1u)) continue;
                const float2 cr = cam4[j], pr = proj_lut[j == 0 ? idx[0] : j == 1 ? idx[1] : j == 2 ? idx[2] : idx[3]];
                const Xyzf r = law_of_sines_mirror(Ray2{cr.x, cr.y}, Ray2{pr.x, pr.y}, T, t_len);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    out[3 * k] = (k == j) ? r.x : out[3 * k];
                    out[3 * k + 1] = (k == j) ? r.y : out[3 * k + 1];
                    out[3 * k + 2] = (k == j) ? r.z : out[3 * k + 2];
                }
            }
        }
    }
}

// ---- float32 fast path ---------------------------------------------------------------------------------------------------------
// The same cancelled form with the sines taken from CROSS products, |T x c| = |T||c| sin(alpha), instead of sqrt(|T|^2|c|^2 - (T.c)^2):
// no subtraction of nearly equal numbers is left in the sines, so float32 carries them to a few 2^-24 of |T||c| however small the
// angle, and the only cancellation that remains is the one the geometry itself has, in D = Sa*B + A*Sb ~ sin(gamma).  Each of
// cos(alpha), sin(alpha), cos(beta), sin(beta) then carries an ABSOLUTE error <= ~4 * 2^-24 = 2.4e-7 from the float32 arithmetic (and
// 6e-8 from rounding T to float32), on top of the 2.7e-7 the reference's own float32 steps differ by (tri_is_flat above), and
//   |d len / len| <= eps_ref * [1/sin^2 b + (1/sin a + 1/sin b)/sin g]  +  eps_f32 * [1/sin b + 4/sin g].
// With the guard at kGuardAmpF32 = 60 (sin^2 b >= 1/60, min(sin a, sin b) * sin g >= 2/60 hence sin g >= 1/30):
//   2 * 60 * 2.7e-7 + 3.0e-7 * (7.8 + 120) = 3.2e-5 + 3.8e-5 = 7.0e-5 < 1e-4.
// Pixels past the guard are redone by law_of_sines_mirror (float64 on the reference's float32 intermediates), like before.
constexpr float kGuardAmpF32 = 60.0f;

struct TriF32 {               // per-launch constants of the float32 path
    float t0, t1, t2, tl2;
};

struct TriF32Terms {
    float A, B, Sa2, Sb2, Sa, Sb, D;
};

__device__ __forceinline__ TriF32Terms tri_f32_terms(float cx, float cy, float px, float py, const TriF32 &k)
{
    TriF32Terms t;
    t.A = -fmaf(k.t0, cx, fmaf(k.t1, cy, k.t2));
    t.B = fmaf(k.t0, px, fmaf(k.t1, py, k.t2));
    // T x c and T x p with c = (cx, cy, 1), p = (px, py, 1)
    const float ux = fmaf(-k.t2, cy, k.t1), uy = fmaf(k.t2, cx, -k.t0), uz = fmaf(k.t0, cy, -k.t1 * cx);
    const float vx = fmaf(-k.t2, py, k.t1), vy = fmaf(k.t2, px, -k.t0), vz = fmaf(k.t0, py, -k.t1 * px);
    t.Sa2 = fmaf(ux, ux, fmaf(uy, uy, uz * uz));
    t.Sb2 = fmaf(vx, vx, fmaf(vy, vy, vz * vz));
    t.Sa = __builtin_amdgcn_sqrtf(t.Sa2);      // v_sqrt_f32 / v_rcp_f32: 1 ulp, inside eps_f32
    t.Sb = __builtin_amdgcn_sqrtf(t.Sb2);
    t.D = fmaf(t.Sa, t.B, t.A * t.Sb);
    return t;
}

// sin^2(beta) < 1/A_g   <=>  Sb2 * A_g < tb ;   min(sin a, sin b) * sin g < 2/A_g  <=>  D^2 * min(Sa2*tb, Sb2*ta) < (2/A_g)^2 * (ta*tb)^2
__device__ __forceinline__ bool tri_f32_flat(const TriF32Terms &t, float cx, float cy, float px, float py, const TriF32 &k)
{
    const float ta = k.tl2 * fmaf(cx, cx, fmaf(cy, cy, 1.0f)), tb = k.tl2 * fmaf(px, px, fmaf(py, py, 1.0f));
    const float tatb = ta * tb;
    constexpr float k2 = (2.0f / kGuardAmpF32) * (2.0f / kGuardAmpF32);
    return (t.Sb2 * kGuardAmpF32 < tb) | ((t.D * t.D) * fminf(t.Sa2 * tb, t.Sb2 * ta) < (k2 * tatb) * tatb) | !(t.D == t.D);
}

__device__ __forceinline__ bool tri_f32_is_flat(float cx, float cy, float px, float py, const TriF32 &k)
{
    return tri_f32_flat(tri_f32_terms(cx, cy, px, py, k), cx, cy, px, py, k);
}

template <bool GUARD>
__device__ __forceinline__ void triangulate4_f32(const float (&cx)[4], const float (&cy)[4], const float (&px)[4], const float (&py)[4],
                                                 uint32_t valid, const TriF32 &k, const double (&T)[3], double t_len, float (&out)[12],
                                                 const float2 *__restrict__ cam4, const float2 *__restrict__ proj_lut, const uint32_t (&idx)[4])
{
    uint32_t ill = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const TriF32Terms t = tri_f32_terms(cx[j], cy[j], px[j], py[j], k);
        const float s = (k.tl2 * t.Sb) * __builtin_amdgcn_rcpf(t.D);
        if (GUARD) ill |= tri_f32_flat(t, cx[j], cy[j], px[j], py[j], k) ? (1u << j) : 0u;
        const bool ok = (valid >> j) & 1u;
        out[3 * j] = ok ? cx[j] * s : __builtin_nanf("");
        out[3 * j + 1] = ok ? cy[j] * s : __builtin_nanf("");
        out[3 * j + 2] = ok ? s : __builtin_nanf("");
    }
    if (GUARD) {
        ill &= valid;
        if (ill) {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                if (!((ill >> j) & 1u)) continue;
                const float2 cr = cam4[j], pr = proj_lut[j == 0 ? idx[0] : j == 1 ? idx[1] : j == 2 ? idx[2] : idx[3]];
                const Xyzf r = law_of_sines_mirror(Ray2{cr.x, cr.y}, Ray2{pr.x, pr.y}, T, t_len);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    out[3 * q] = (q == j) ? r.x : out[3 * q];
                    out[3 * q + 1] = (q == j) ? r.y : out[3 * q + 1];
                    out[3 * q + 2] = (q == j) ? r.z : out[3 * q + 2];
                }
            }
        }
    }
}

// The projector table is stored in tiles of 8 rows so that a wave's gather stays within a few cache lines whichever way the
// decoded projector coordinates drift along a camera row.  wide = 0: 8x8-pixel tiles (512 B, a tile row is half a 128-byte line);
// wide = 1: 16x8-pixel tiles (1 KB, a tile row is exactly one 128-byte line).  tiles_x counts tiles of the chosen width.
__device__ __forceinline__ uint32_t proj_lut_index(int pu, int pv, int tiles_x, int wide = 0)
{
    const uint32_t row = (uint32_t)(pv >> 3) * (uint32_t)tiles_x, in_y = (uint32_t)(pv & 7);
    return wide ? (((row + (uint32_t)(pu >> 4)) << 7) | (in_y << 4) | (uint32_t)(pu & 15))
                : (((row + (uint32_t)(pu >> 3)) << 6) | (in_y << 3) | (uint32_t)(pu & 7));
}

// Ordering point for data exchanged through LDS between the lanes of ONE wave: LDS operations of a wave execute in program
// order, so no s_barrier is needed -- only the compiler must not move the reads above the writes.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
