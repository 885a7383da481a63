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
