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

struct Xyzf {
    float x, y, z;
};

// The cancelled form on the reference's own float32 intermediates (triangulate.py:90 NormedL and the float32 norm inside
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

// ---- the dense / fused fast form -------------------------------------------------------------------------------------------
// triangulate.py:86-95 with the normalisations cancelled (c = camera ray [cx, cy, 1], p = projector ray [px, py, 1]):
//   A = -T.c, B = T.p, Sa = |T x c| = |T||c| sin(alpha), Sb = |T x p| = |T||p| sin(beta), sin(gamma) = sin(alpha + beta)
//   =>   Pts = c * |T|^2 * Sb / (Sa*B + A*Sb).
// The sines come from CROSS products, not from sqrt(|T|^2|c|^2 - (T.c)^2): no subtraction of nearly equal numbers is left in them,
// so float32 carries them to a few 2^-24 of |T||c| however small the angle, and the only cancellation that remains is the one the
// geometry itself has, in D = Sa*B + A*Sb ~ sin(gamma).  That is what lets the whole form run in float32 -- on gfx950 float64 vector
// instructions issue at half rate, v_rcp_f64 / v_rsq_f64 at an eighth (tools/ubench/valu_rates.hip), and a float64-heavy tail also
// pulls the clock down: the float64 version of this tail (sqrt of differences, round 1) measured 141-147 us per 4096x3000x44
// fused scan against 130-131 us for this one on the same box (gpurun_out/r2m/ab.log).
// Error budget.  Each of cos(alpha), sin(alpha), cos(beta), sin(beta) carries an ABSOLUTE error <= ~4 * 2^-24 = 2.4e-7 from the
// float32 arithmetic (+ 6e-8 from rounding T to float32): eps_f32 <= 3.0e-7; the reference's own float32 steps (NormedL: a sqrt and
// three divisions, the projector norm: a sqrt -- triangulate.py:90,92) move ITS cosines by eps_ref <= 3 * 2^-24 = 1.8e-7, and
//   |d len / len| <= eps_ref * [1/sin^2 b + (1/sin a + 1/sin b)/sin g]  +  eps_f32 * [1/sin b + 4/sin g].
// A pixel is "flat" when a term can pass kGuardAmp = 60 (sin^2 b < 1/60, or min(sin a, sin b) * sin g < 2/60 -- which also gives
// sin g >= 1/30 for the pixels that pass):  2 * 60 * 1.8e-7 + 3.0e-7 * (7.8 + 120) = 2.2e-5 + 3.8e-5 = 6.0e-5 < 1e-4 even if every
// rounding aligns.  Flat pixels are redone by law_of_sines_mirror (float64 on the reference's float32 intermediates) -- in ONE
// not-unrolled loop behind a lane-level "any of my four" test, so the common path pays for the comparisons only.
constexpr float kGuardAmp = 60.0f;

struct TriF32 {               // per-launch constants of the fast form
    float t0, t1, t2, tl2;
};

struct TriFastTerms {
    float A, B, Sa2, Sb2, Sa, Sb, D;
};

__device__ __forceinline__ TriFastTerms tri_fast_terms(float cx, float cy, float px, float py, const TriF32 &k)
{
    TriFastTerms t;
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
__device__ __forceinline__ bool tri_flat(const TriFastTerms &t, float cx, float cy, float px, float py, const TriF32 &k)
{
    const float ta = k.tl2 * fmaf(cx, cx, fmaf(cy, cy, 1.0f)), tb = k.tl2 * fmaf(px, px, fmaf(py, py, 1.0f));
    const float tatb = ta * tb;
    constexpr float k2 = (2.0f / kGuardAmp) * (2.0f / kGuardAmp);
    return (t.Sb2 * kGuardAmp < tb) | ((t.D * t.D) * fminf(t.Sa2 * tb, t.Sb2 * ta) < (k2 * tatb) * tatb) | !(t.D == t.D);
}

__device__ __forceinline__ bool tri_is_flat(float cx, float cy, float px, float py, const TriF32 &k)
{
    return tri_flat(tri_fast_terms(cx, cy, px, py, k), cx, cy, px, py, k);
}

// valid: bit j set = pixel j decodable; out = x0 y0 z0 x1 ... (NaN where not decodable).  GUARD = false: fast form everywhere (A/B).
// cam4 / proj_lut + idx: where the lane's rays came from -- the rare path reads them again instead of keeping 16 registers alive.
template <bool GUARD>
__device__ __forceinline__ void triangulate4(const float (&cx)[4], const float (&cy)[4], const float (&px)[4], const float (&py)[4],
                                                 uint32_t valid, const TriF32 &k, const double (&T)[3], double t_len, float (&out)[12],
                                                 const float2 *__restrict__ cam4, const float2 *__restrict__ proj_lut, const uint32_t (&idx)[4])
{
    uint32_t ill = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const TriFastTerms t = tri_fast_terms(cx[j], cy[j], px[j], py[j], k);
        const float s = (k.tl2 * t.Sb) * __builtin_amdgcn_rcpf(t.D);
        if (GUARD) ill |= tri_flat(t, cx[j], cy[j], px[j], py[j], k) ? (1u << j) : 0u;
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

// The fast pass alone: XYZ of the lane's four pixels by the float32 form, and which of them are flat (bit j of the result; decodable pixels
// only).  The caller redoes the flagged pixels through law_of_sines_mirror -- the fused scan kernel compacts them over the WAVE first
// (decode.hip), so a wave pays one pass per 64 flagged pixels instead of one pass per flagged position of its lanes.
__device__ __forceinline__ uint32_t triangulate4_flag(const float (&cx)[4], const float (&cy)[4], const float (&px)[4], const float (&py)[4],
                                                      uint32_t valid, const TriF32 &k, float (&out)[12])
{
    uint32_t ill = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const TriFastTerms t = tri_fast_terms(cx[j], cy[j], px[j], py[j], k);
        const float s = (k.tl2 * t.Sb) * __builtin_amdgcn_rcpf(t.D);
        ill |= tri_flat(t, cx[j], cy[j], px[j], py[j], k) ? (1u << j) : 0u;
        const bool ok = (valid >> j) & 1u;
        out[3 * j] = ok ? cx[j] * s : __builtin_nanf("");
        out[3 * j + 1] = ok ? cy[j] * s : __builtin_nanf("");
        out[3 * j + 2] = ok ? s : __builtin_nanf("");
    }
    return ill & valid;
}

// One pixel through the same arithmetic as triangulate4 (bit-identical results): the x-major scatter of slgc_cloud_dev triangulates a pixel where
// it writes it.  cam_exact = this pixel's entry of the exact per-pixel camera table (read only on the guarded path).
template <bool GUARD>
__device__ __forceinline__ Xyzf triangulate1(float cx, float cy, float px, float py, const TriF32 &k, const double (&T)[3], double t_len,
                                             const float2 *__restrict__ cam_exact)
{
    const TriFastTerms t = tri_fast_terms(cx, cy, px, py, k);
    const float s = (k.tl2 * t.Sb) * __builtin_amdgcn_rcpf(t.D);
    Xyzf r{cx * s, cy * s, s};
    if (GUARD && tri_flat(t, cx, cy, px, py, k)) {
        const float2 cr = *cam_exact;
        r = law_of_sines_mirror(Ray2{cr.x, cr.y}, Ray2{px, py}, T, t_len);
    }
    return r;
}

// ---- camera rays from a table kept at every 4th column (CamNodes) ----
// The camera ray table is the largest non-algorithmic stream of the scan kernels (8 B / pixel).  Along a row the ray is a smooth function of
// x, so the kernels can read NODES at x = -4, 0, 4, ..., W + 4 of every row (2 B / pixel) and put the cubic through the four nodes around
// a lane's 4-pixel group: pixel 4g + j, j = 1..3, = e0 + (Lm[j] dm + L1[j] d1 + L2[j] d2) with the differences dm = e(-1) - e0, d1 = e1 - e0,
// d2 = e2 - e0 and the Lagrange weights at t = j / 4 (exact binary fractions); pixel 4g is the node itself.  Forming the small correction
// first and adding it to e0 last keeps the float32 result within 1 ulp of the exactly evaluated float32 ray (ensure_luts measures the
// difference over every pixel of the calibration and falls back to the full table when it is too large); the truncation error of the
// cubic is < 2e-9 for the reference's lenses.  Flat pixels (triangulate4's rare path) and lanes whose rays cross zero
// (cam_rays_exact_where_tiny) still read the exact per-pixel table.
struct CamNodes {
    const float2 *nodes;      // [rows][ne], ne = W / 4 + 3: node n of a row sits at x = 4 (n - 1);  nullptr = use the per-pixel table
    uint32_t w4, ne;          // 4-pixel groups per row, nodes per row
    float inv_w4;
};

__device__ __forceinline__ const float2 *cam_node_ptr(const CamNodes &cn, uint32_t g)      // g = linear 4-pixel group index in the band
{
    uint32_t row = (uint32_t)((float)g * cn.inv_w4);                                        // g < 2^24: exact in float32; fix the quotient
    int32_t gx = (int32_t)(g - row * cn.w4);
    if (gx < 0) {
        --row;
        gx += (int32_t)cn.w4;
    } else if ((uint32_t)gx >= cn.w4) {
        ++row;
        gx -= (int32_t)cn.w4;
    }
    return cn.nodes + (size_t)row * cn.ne + (uint32_t)gx;
}

typedef float cam_v4f __attribute__((ext_vector_type(4), aligned(8)));                    // two nodes; a node pair starts on any 8-byte boundary

// n01 = nodes (-1, 0), n23 = nodes (1, 2) around the group, as loaded from cam_node_ptr()[0..3]
__device__ __forceinline__ void cam_rays_from_nodes(const cam_v4f &n01, const cam_v4f &n23, float (&cx)[4], float (&cy)[4])
{
    constexpr float Lm[3] = {-0.0546875f, -0.0625f, -0.0390625f}, L1[3] = {0.2734375f, 0.5625f, 0.8203125f}, L2[3] = {-0.0390625f, -0.0625f, -0.0546875f};
    const float ex = n01.z, ey = n01.w;
    const float dmx = n01.x - ex, dmy = n01.y - ey, d1x = n23.x - ex, d1y = n23.y - ey, d2x = n23.z - ex, d2y = n23.w - ey;
    cx[0] = ex;
    cy[0] = ey;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        cx[j] = ex + fmaf(L2[j - 1], d2x, fmaf(L1[j - 1], d1x, Lm[j - 1] * dmx));
        cy[j] = ey + fmaf(L2[j - 1], d2y, fmaf(L1[j - 1], d1y, Lm[j - 1] * dmy));
    }
}

// X = cx * s and Y = cy * s inherit the RELATIVE error of the ray component, and an interpolated component is only good to an absolute
// ~1e-9 where the ray crosses zero (a handful of columns / rows of the image): a lane with a component below kCamNodeTiny takes its four
// rays from the exact per-pixel table instead (cam4 = the lane's first pixel there).  ensure_luts checks both error measures.
constexpr float kCamNodeTiny = 1e-3f;

__device__ __forceinline__ void cam_rays_exact_where_tiny(float (&cx)[4], float (&cy)[4], const float2 *__restrict__ cam4)
{
    float m = fminf(fabsf(cx[0]), fabsf(cy[0]));
#pragma unroll
    for (int j = 1; j < 4; ++j) m = fminf(m, fminf(fabsf(cx[j]), fabsf(cy[j])));
    if (m < kCamNodeTiny) {
        const float4 c01 = reinterpret_cast<const float4 *>(cam4)[0], c23 = reinterpret_cast<const float4 *>(cam4)[1];
        cx[0] = c01.x, cy[0] = c01.y, cx[1] = c01.z, cy[1] = c01.w;
        cx[2] = c23.x, cy[2] = c23.y, cx[3] = c23.z, cy[3] = c23.w;
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
