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
// triangulate.py:86-95 with everything that depends on ONE ray only moved out of the per-pixel arithmetic.  With e = c / |c| the unit camera
// ray (c = [cx, cy, 1]), t = T / |T|:
//   cos(alpha) = -t.e      sin(alpha) = sqrt(1 - cos^2(alpha))      (alpha in (0, pi): the non-negative root, like arccos -> sin)
//   beta: per PROJECTOR PIXEL, evaluated once per calibration in float64 from the reference's float32 ray and float32 norm (triangulate.py:92)
//         and kept as ONE float32 per projector pixel, th = tan(beta / 2), in the table the scan kernels gather from (proj_tan_half below):
//             cos(beta) = (1 - th^2) / (1 + th^2)        sin(beta) = 2 th / (1 + th^2)            (beta in (0, pi): th in (0, inf))
//   sin(gamma) = sin(alpha + beta) = sa cb + ca sb =: Dn             Pts = e * |T| sb / Dn                           (:93-95)
// The common factor 1 / (1 + th^2) cancels in sb / Dn: with om = 1 - th^2, op = 1 + th^2, nb = 2 th the kernels form Dn' = sa om + ca nb (= Dn op)
// and Pts = e * |T| nb / Dn' -- no division by op anywhere, the flat test compares against op and op^2 instead.  Per pixel: one rsq, one sqrt,
// one rcp and ~25 multiply-adds.  (Rounds 1-3 formed the same quantities un-normalised from cross products of T with both rays -- twice the
// arithmetic; round 4 kept (cos(beta), sin(beta)) as 8 bytes per projector pixel: a gathered table entry drags a 128-byte line over the fabric,
// and with 4-byte entries a line -- and the L2 -- holds twice as many projector pixels.  Counter traffic: NOTES.md round 5.)
// Float32 suffices because no subtraction of nearly equal numbers is left except the one the geometry itself has, in Dn ~ sin(gamma):
// sa comes from 1 - ca^2 with |ca| <= 0.6 for any camera ray a scanner uses (and a ray nearly parallel to the baseline is flagged flat).
// On gfx950 float64 vector instructions issue at half rate, v_rcp_f64 / v_rsq_f64 at an eighth (tools/ubench/valu_rates.hip).
// Error budget.  cos(alpha), sin(alpha) carry an ABSOLUTE error <= ~4 * 2^-24 = 2.4e-7 from the float32 arithmetic (+ 6e-8 from rounding T to
// float32): eps_f32 <= 3.0e-7.  On the projector side the table's rounding of th (relative 2^-24) IS a perturbation of beta itself,
// d beta = sin(beta) 2^-24, i.e. |d cos(beta)|, |d sin(beta)| <= 6e-8; th^2, 1 - th^2 and 2 th add at most 2 * 2^-24 relative to op = 1 + th^2
// (1.2e-7 on the normalised cos(beta)); the division by op never happens.  So the projector side stays inside the same eps_f32 as before.  The
// reference's own float32 steps (NormedL: a sqrt and three divisions -- the projector norm's rounding IS in the table) move ITS cos(alpha) by
// eps_ref <= 3 * 2^-24 = 1.8e-7, and
//   |d len / len| <= eps_ref * [1/sin^2 b + (1/sin a + 1/sin b)/sin g]  +  eps_f32 * [1/sin b + 4/sin g].
// A pixel is "flat" when a term can pass kGuardAmp = 60 (sin^2 b < 1/60, or min(sin a, sin b) * |sin g| < 2/60 -- which also gives
// |sin g| >= 1/30 for the pixels that pass):  2 * 60 * 1.8e-7 + 3.0e-7 * (7.8 + 120) = 2.2e-5 + 3.8e-5 = 6.0e-5 < 1e-4 even if every
// rounding aligns.  Flat pixels are redone by law_of_sines_mirror (float64 on the reference's float32 intermediates).  th = inf or NaN
// (beta = pi: the ray points back along the baseline) makes Dn' NaN: flagged flat like any other NaN.
constexpr float kGuardAmp = 60.0f;

struct TriF32 {               // per-launch constants of the fast form: T / |T| and |T|
    float t0, t1, t2, tl;
};

__host__ __device__ inline TriF32 make_tri_f32(const double (&T)[3], double t_len)
{
    return TriF32{(float)(T[0] / t_len), (float)(T[1] / t_len), (float)(T[2] / t_len), (float)t_len};
}

// What the projector table holds per pixel for the fast form: tan(beta / 2) of the pixel's ray against the baseline.
__host__ __device__ inline float proj_tan_half(float px, float py, const double (&T)[3], double t_len)
{
    const float qn = sqrtf((px * px + py * py) + 1.0f);                                                    // norm inside triangulate.py:92 (float32)
    const double c = ((T[0] * (double)px + T[1] * (double)py) + T[2]) / (t_len * (double)qn);             // :92
    const double s = sqrt(fmax(0.0, 1.0 - c * c));
    const double th = c >= 0.0 ? s / (1.0 + c) : (1.0 - c) / s;                                           // the well-conditioned form on each side
    return (float)fmin(th, 1e9);      // beta -> pi (a ray back along the baseline): th -> inf; capped so that (1 + th^2)^2 stays finite in float32 and the
                                      // flat test sends the pixel to the exact path (4 th^2 * 60 < (1 + th^2)^2 from th = 15.4 on); a NaN ray takes the cap too (fmin) and gets its NaN from the exact path
}

struct TriFastTerms {
    float ex, ey, ez;         // unit camera ray
    float ca, sa;             // cos(alpha), sin(alpha)
    float nb, op, Dn;         // 2 th (= sin(beta) op), 1 + th^2, sin(gamma) op
};

__device__ __forceinline__ TriFastTerms tri_fast_terms(float cx, float cy, float th, const TriF32 &k)
{
    TriFastTerms t;
    const float r = __builtin_amdgcn_rsqf(fmaf(cx, cx, fmaf(cy, cy, 1.0f)));      // v_rsq_f32 / v_sqrt_f32 / v_rcp_f32: 1 ulp, inside eps_f32
    t.ex = cx * r;
    t.ey = cy * r;
    t.ez = r;
    t.ca = -fmaf(k.t0, t.ex, fmaf(k.t1, t.ey, k.t2 * t.ez));
    t.sa = __builtin_amdgcn_sqrtf(fmaf(-t.ca, t.ca, 1.0f));
    t.nb = th + th;
    t.op = fmaf(th, th, 1.0f);
    t.Dn = fmaf(t.sa, fmaf(-th, th, 1.0f), t.ca * t.nb);
    return t;
}

// sin^2(beta) < 1/A_g,  or  min(sin a, sin b) * |sin g| < 2/A_g,  or a NaN anywhere -- on the un-normalised quantities (sb = nb / op, Dn = Dn' / op)
__device__ __forceinline__ bool tri_flat(const TriFastTerms &t)
{
    const float op2 = t.op * t.op;
    return ((t.nb * t.nb) * kGuardAmp < op2) | (fminf(t.sa * t.op, t.nb) * fabsf(t.Dn) < (2.0f / kGuardAmp) * op2) | !(t.Dn == t.Dn);
}

__device__ __forceinline__ bool tri_is_flat(float cx, float cy, float th, const TriF32 &k)
{
    return tri_flat(tri_fast_terms(cx, cy, th, k));
}

// valid: bit j set = pixel j decodable; out = x0 y0 z0 x1 ... (NaN where not decodable).  GUARD = false: fast form everywhere (A/B).
// th: the pixels' entries of the projector table's tan(beta / 2) half.  cam4 / proj_rays + idx: where the lane's exact float32 rays are --
// the rare path reads them instead of keeping 16 registers alive.
template <bool GUARD>
__device__ __forceinline__ void triangulate4(const float (&cx)[4], const float (&cy)[4], const float (&th)[4],
                                                 uint32_t valid, const TriF32 &k, const double (&T)[3], double t_len, float (&out)[12],
                                                 const float2 *__restrict__ cam4, const float2 *__restrict__ proj_rays, const uint32_t (&idx)[4])
{
    uint32_t ill = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const TriFastTerms t = tri_fast_terms(cx[j], cy[j], th[j], k);
        const bool ok = (valid >> j) & 1u;
        const float s = ok ? (k.tl * t.nb) * __builtin_amdgcn_rcpf(t.Dn) : __builtin_nanf("");
        if (GUARD) ill |= tri_flat(t) ? (1u << j) : 0u;
        out[3 * j] = t.ex * s;
        out[3 * j + 1] = t.ey * s;
        out[3 * j + 2] = t.ez * s;
    }
    if (GUARD) {
        ill &= valid;
        if (ill) {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                if (!((ill >> j) & 1u)) continue;
                const float2 cr = cam4[j], pr = proj_rays[j == 0 ? idx[0] : j == 1 ? idx[1] : j == 2 ? idx[2] : idx[3]];
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
__device__ __forceinline__ uint32_t triangulate4_flag(const float (&cx)[4], const float (&cy)[4], const float (&th)[4],
                                                      uint32_t valid, const TriF32 &k, float (&out)[12])
{
    uint32_t ill = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const TriFastTerms t = tri_fast_terms(cx[j], cy[j], th[j], k);
        const bool ok = (valid >> j) & 1u;
        const float s = ok ? (k.tl * t.nb) * __builtin_amdgcn_rcpf(t.Dn) : __builtin_nanf("");
        ill |= tri_flat(t) ? (1u << j) : 0u;
        out[3 * j] = t.ex * s;
        out[3 * j + 1] = t.ey * s;
        out[3 * j + 2] = t.ez * s;
    }
    return ill & valid;
}

// One pixel through the same arithmetic as triangulate4 (bit-identical results): the x-major scatter of slgc_cloud_dev triangulates a pixel where
// it writes it.  cam_exact / proj_ray = this pixel's entries of the exact per-pixel camera table and of the projector RAY table (read only on the guarded path).
template <bool GUARD>
__device__ __forceinline__ Xyzf triangulate1(float cx, float cy, float th, const TriF32 &k, const double (&T)[3], double t_len,
                                             const float2 *__restrict__ cam_exact, const float2 *__restrict__ proj_ray)
{
    const TriFastTerms t = tri_fast_terms(cx, cy, th, k);
    const float s = (k.tl * t.nb) * __builtin_amdgcn_rcpf(t.Dn);
    Xyzf r{t.ex * s, t.ey * s, t.ez * s};
    if (GUARD && tri_flat(t)) {
        const float2 cr = *cam_exact, pr = *proj_ray;
        r = law_of_sines_mirror(Ray2{cr.x, cr.y}, Ray2{pr.x, pr.y}, T, t_len);
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

// A lane's 12 floats of XYZ into the LDS staging block as three WHOLE 16-byte stores.  Left to itself the compiler writes the 48 bytes as one
// ds_write_b96 and four or five ds_write2_b32 (whatever pairs become ready together); a dword store is banked over 32 lanes and 32 banks, and with
// lanes 12 dwords apart only 8 distinct banks come up -- 4-way conflicts, 54 of the 166 LDS cycles a wave of the fused kernel spent (round 5:
// SQ_LDS_BANK_CONFLICT 2 592 000 per launch at 4096x3000, the same number in the dense kernel).  ds_write_b128 is banked over 8 contiguous lanes:
// 12 i mod 32 is distinct for i = 0..7 -- conflict-free.  One vector store each through an LDS-address-space pointer keeps them whole.
typedef float lds_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) lds_v4f *lds_v4f_ptr;
__device__ __forceinline__ void stage_xyz12(float4 *slot, const float (&out)[12])      // slot: 16-byte aligned, in LDS
{
    lds_v4f_ptr q = (lds_v4f_ptr)slot;
    q[0] = lds_v4f{out[0], out[1], out[2], out[3]};
    q[1] = lds_v4f{out[4], out[5], out[6], out[7]};
    q[2] = lds_v4f{out[8], out[9], out[10], out[11]};
}

// Ordering point for data exchanged through LDS between the lanes of ONE wave: LDS operations of a wave execute in program
// order, so no s_barrier is needed -- only the compiler must not move the reads above the writes.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
