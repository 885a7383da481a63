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
// final scale is rounded to float32 once and applied in float32.  Agrees with the acos/sin form to ~2e-7 relative (float32
// output resolution) away from degenerate geometry, far inside the 1e-4 tolerance of the build's north star.
struct Xyzf {
    float x, y, z;
};

__device__ __forceinline__ Xyzf law_of_sines_fast(Ray2 cam, Ray2 prj, const double (&T)[3], double t_len)
{
    const double cx = cam.x, cy = cam.y, px = prj.x, py = prj.y;
    const double tl2 = t_len * t_len;
    const double A = -fma(T[0], cx, fma(T[1], cy, T[2]));
    const double B = fma(T[0], px, fma(T[1], py, T[2]));
    const double cn2 = fma(cx, cx, fma(cy, cy, 1.0)), pn2 = fma(px, px, fma(py, py, 1.0));
    const double Sa = fast_sqrt(fma(tl2, cn2, -A * A)), Sb = fast_sqrt(fma(tl2, pn2, -B * B));
    const float s = (float)(tl2 * Sb * fast_rcp(fma(Sa, B, A * Sb)));
    return Xyzf{cam.x * s, cam.y * s, s};
}

// The projector table is stored in 8x8-pixel tiles (512 B) so that a wave's gather stays within a few cache lines
// whichever way the decoded projector coordinates drift along a camera row.
__device__ __forceinline__ uint32_t proj_lut_index(int pu, int pv, int tiles_x)
{
    return (((uint32_t)(pv >> 3) * (uint32_t)tiles_x + (uint32_t)(pu >> 3)) << 6) | (uint32_t)((pv & 7) << 3) | (uint32_t)(pu & 7);
}

