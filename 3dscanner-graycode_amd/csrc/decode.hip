// decode.hip -- Gray-code decode kernels for gfx950 (MI355X).
//
// K1a  k_decode_fast     uint8 stacks, integer eps (the reference default eps=1): per-pixel fp64
//                        direct/global separation folded into three integer thresholds, then pure
//                        integer classification of every (normal, inverse) byte pair.  HBM-bound:
//                        N bytes in + 4 bytes out per pixel, every byte read from HBM once.
// K1b  k_decode_generic  literal fp64 restatement (uint8 or float64 stacks, any eps, caller-supplied
//                        L_d/L_g, int8 code planes, L_d/L_g output) -- the API-parity kernel.
// K1c  k_codes_to_pixels int8 code planes (n_runs) -> max-merge -> Gray->binary maps.
//
// Reference semantics (file:line under the reference checkout):
//   get_direct_indirect  scanner/grayCode/decode_codes.py:90-122
//   get_is_lit           scanner/grayCode/decode_codes.py:125-186   (last matching rule wins)
//   gray_to_decimal      scanner/grayCode/decode_codes.py:209-229
//   run merge + loops    src/3-capture_decode.py:95-100
//
// Compiled with -ffp-contract=off: the reference's fp64 rounding sequence is part of its
// behaviour (L_max - D*b_inv must not become an FMA).
#include "slgc_internal.h"

namespace {

// ------------------------------------------------------------------------------------------
// K1a: integer-threshold fast path
// ------------------------------------------------------------------------------------------
//
// For uint8 data and integer eps = e >= 0 every comparison of get_is_lit has an exact integer
// form.  With d = L_d, g = L_g (fp64, computed exactly as the reference does) and integer x:
//     x > fl(g + e)      <=>  x >= tg ,  tg  = floor(fl(g+e)) + 1   (0 if fl(g+e) < 0, 256 if >= 255 or NaN)
//     fl(x + e) < d      <=>  x <  tnd,  tnd = ceil(d) - e          (x + e is exact; 0 if d is NaN)
//     n > fl(i + e)      <=>  n - i >= e + 1
//     fl(n + e) < i      <=>  i - n >= e + 1
// "d > g + e" is per pixel and is folded into the difference threshold cA (= e+1, or unreachable).
// Rules :172-182 become (last wins):  r1 -> 1, r2 -> 0, r3 -> 0, r4 -> 1, so
//     bit = r4 | (r1 & ~r3)        valid = r1 | r2 | r3 | r4        (r1, r2 exclusive because e+1 >= 1)
// Runs are max-merged per code bit (1 > 0 > -1):  bit |= bit_r, valid |= valid_r.

constexpr int kUnreachable = 0x4000;

struct FastArgs {
    const uint8_t *run[SLGC_MAX_RUNS];
    size_t plane_stride;  // bytes between consecutive frames of a run
    size_t ngroups;       // groups of PX consecutive pixels
    int16_t *h;
    int16_t *v;
    DecodeGeom g;
    int e;                // integer eps
};

template <int PX>
struct Words {
    static constexpr int n = (PX + 3) / 4;
    uint32_t w[n];
};

template <int PX>
__device__ __forceinline__ Words<PX> load_px(const uint8_t *p)
{
    Words<PX> r;
    if constexpr (PX == 16) {
        uint4 t = *reinterpret_cast<const uint4 *>(p);
        r.w[0] = t.x; r.w[1] = t.y; r.w[2] = t.z; r.w[3] = t.w;
    } else if constexpr (PX == 8) {
        uint2 t = *reinterpret_cast<const uint2 *>(p);
        r.w[0] = t.x; r.w[1] = t.y;
    } else if constexpr (PX == 4) {
        r.w[0] = *reinterpret_cast<const uint32_t *>(p);
    } else {
        r.w[0] = *p;
    }
    return r;
}

template <int PX>
__device__ __forceinline__ int byte_of(const Words<PX> &r, int j)
{
    return (int)((r.w[j >> 2] >> (8 * (j & 3))) & 0xffu);
}

// Per-pixel thresholds from black/white and the 12 L_max/L_min frames.
__device__ __forceinline__ void pixel_thresholds(int black, int white, int lmax, int lmin, int e, int &tt, int &cA)
{
    const double w = (double)white, b = (double)black;
    const double b_inv = w / (w + b);                        // decode_codes.py:113 (0/0 -> NaN)
    const double ld = (double)(lmax - lmin) * b_inv;         // :119  (L_max - L_min is exact)
    const double t2 = 2.0 * ((double)lmax - ld);             // :120  2.0*(L_max - L_d) ...
    const double lg = t2 * b_inv;                            //       ... * b_inv
    const double ge = lg + (double)e;                        // L_g + eps (:172-182)
    const bool direct = ld > ge;
    int tg;
    if (!(ge < 255.0)) tg = 256;                             // also NaN
    else if (ge < 0.0) tg = 0;
    else tg = (int)ge + 1;
    int tnd;
    if (!(ld > 0.0)) tnd = 0;                                // also NaN
    else if (ld > 1024.0) tnd = 256;
    else {
        tnd = (int)ceil(ld) - e;
        tnd = tnd < 0 ? 0 : (tnd > 256 ? 256 : tnd);
    }
    tt = tnd | (tg << 16);
    cA = direct ? (e + 1) : kUnreachable;
}

template <int PX>
__device__ __forceinline__ void classify_pair(const Words<PX> &nw, const Words<PX> &iw, const int (&tt)[PX],
                                              const int (&cA)[PX], uint32_t mask, uint32_t (&accB)[PX], uint32_t (&accV)[PX])
{
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int n = byte_of<PX>(nw, j), i = byte_of<PX>(iw, j);
        const int tnd = tt[j] & 0xffff, tg = (int)((uint32_t)tt[j] >> 16);
        const int diff = n - i;
        const bool r1 = diff >= cA[j];
        const bool r2 = (i - n) >= cA[j];
        const bool r3 = (n < tnd) & (i >= tg);
        const bool r4 = (n >= tg) & (i < tnd);
        const bool bit = r4 | (r1 & !r3);
        const bool valid = r1 | r2 | r3 | r4;
        accB[j] |= bit ? mask : 0u;
        accV[j] |= valid ? mask : 0u;
    }
}

// Gray -> binary on both 16-bit halves at once (prefix XOR from the MSB, decode_codes.py:203-207).
__device__ __forceinline__ uint32_t gray_to_binary_2x16(uint32_t x)
{
    x ^= (x >> 1) & 0x7fff7fffu;
    x ^= (x >> 2) & 0x3fff3fffu;
    x ^= (x >> 4) & 0x0fff0fffu;
    x ^= (x >> 8) & 0x00ff00ffu;
    return x;
}

template <int PX>
__device__ __forceinline__ void store_i16(int16_t *dst, const int (&val)[PX])
{
    if constexpr (PX == 1) {
        dst[0] = (int16_t)val[0];
    } else {
        uint32_t w[PX / 2];
#pragma unroll
        for (int j = 0; j < PX / 2; ++j) w[j] = ((uint32_t)val[2 * j] & 0xffffu) | ((uint32_t)val[2 * j + 1] << 16);
        if constexpr (PX == 16) {
            reinterpret_cast<uint4 *>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
            reinterpret_cast<uint4 *>(dst)[1] = make_uint4(w[4], w[5], w[6], w[7]);
        } else if constexpr (PX == 8) {
            reinterpret_cast<uint4 *>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            reinterpret_cast<uint2 *>(dst)[0] = make_uint2(w[0], w[1]);
        }
    }
}

template <int PX, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_decode_fast(const FastArgs a)
{
    const size_t grp = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (grp >= a.ngroups) return;
    const size_t off = grp * PX;
    const int L = a.g.L;
    const size_t ps = a.plane_stride;

    uint32_t accB[PX], accV[PX];  // (column-code bits << 16) | row-code bits ; same for "classified"
#pragma unroll
    for (int j = 0; j < PX; ++j) accB[j] = accV[j] = 0u;

    for (int r = 0; r < a.g.n_runs; ++r) {
        const uint8_t *base = a.run[r] + off;
        int tt[PX], cA[PX];
        {
            const Words<PX> bl = load_px<PX>(base), wh = load_px<PX>(base + ps);
            Words<PX> hm[6], vm[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                hm[k] = load_px<PX>(base + (size_t)a.g.hid[k] * ps);
                vm[k] = load_px<PX>(base + (size_t)a.g.vid[k] * ps);
            }
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                int lmax = byte_of<PX>(hm[0], j), lmin = byte_of<PX>(vm[0], j);
#pragma unroll
                for (int k = 1; k < 6; ++k) {
                    lmax = max(lmax, byte_of<PX>(hm[k], j));   // :116 column-code frames only
                    lmin = min(lmin, byte_of<PX>(vm[k], j));   // :117 row-code frames only
                }
                pixel_thresholds(byte_of<PX>(bl, j), byte_of<PX>(wh, j), lmax, lmin, a.e, tt[j], cA[j]);
            }
        }
        // frames 2+2k (column code bit k, MSB first) / 3+2k (row code bit k, LSB first); inverses 2L later
        const uint8_t *pn = base + 2 * ps;
        const uint8_t *pi = base + (size_t)(2 + 2 * L) * ps;
#pragma unroll 2
        for (int k = 0; k < L; ++k) {
            const Words<PX> hn = load_px<PX>(pn), vn = load_px<PX>(pn + ps);
            const Words<PX> hi = load_px<PX>(pi), vi = load_px<PX>(pi + ps);
            pn += 2 * ps;
            pi += 2 * ps;
            classify_pair<PX>(hn, hi, tt, cA, 0x10000u << (L - 1 - k), accB, accV);
            classify_pair<PX>(vn, vi, tt, cA, 1u << k, accB, accV);
        }
    }

    const uint32_t full = (1u << L) - 1u;
    int hv[PX], vv[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const uint32_t bin = gray_to_binary_2x16(accB[j]);
        hv[j] = ((accV[j] >> 16) == full) ? (int)(bin >> 16) : -1;          // any -1 code -> -1 (:225-226)
        vv[j] = ((accV[j] & 0xffffu) == full) ? (int)(bin & 0xffffu) : -1;
    }
    store_i16<PX>(a.h + off, hv);
    store_i16<PX>(a.v + off, vv);
}

// ------------------------------------------------------------------------------------------
// K1b: literal fp64 kernel (API parity)
// ------------------------------------------------------------------------------------------

struct GenArgs {
    const void *run[SLGC_MAX_RUNS];
    size_t plane_stride;  // elements between frames
    size_t npix;
    DecodeGeom g;
    double eps;
    const double *Ld_in, *Lg_in;  // optional caller-supplied L_d/L_g (get_is_lit)
    double *Ld_out, *Lg_out;      // optional (get_direct_indirect)
    int8_t *hc, *vc;              // optional int8 [L][npix] (single run)
    int16_t *h16, *v16;           // optional maps
    int64_t *h64, *v64;
};

__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (b > a ? b : a); }
__device__ __forceinline__ double nanmin(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (b < a ? b : a); }

__device__ __forceinline__ int classify_literal(double d, double g, double n, double i, double eps)
{
    int c = -1;                                         // :162-163; rule 0 (:169-170) rewrites -1: no-op
    if ((d > (g + eps)) & (n > (i + eps))) c = 1;       // :172-173
    if ((d > (g + eps)) & ((n + eps) < i)) c = 0;       // :175-176
    if (((n + eps) < d) & (i > (g + eps))) c = 0;       // :178-179
    if ((n > (g + eps)) & ((i + eps) < d)) c = 1;       // :181-182
    return c;
}

template <typename T>
__global__ void __launch_bounds__(256) k_decode_generic(const GenArgs a)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.npix) return;
    const int L = a.g.L;
    const size_t ps = a.plane_stride;
    uint32_t bits_h = 0, bits_v = 0, ok_h = 0, ok_v = 0;
    for (int r = 0; r < a.g.n_runs; ++r) {
        const T *base = reinterpret_cast<const T *>(a.run[r]) + p;
        double d, g;
        if (a.Ld_in) {
            d = a.Ld_in[p];
            g = a.Lg_in[p];
        } else {
            const double black = (double)base[0], white = (double)base[ps];
            const double b_inv = white / (white + black);                        // :113
            double lmax = (double)base[(size_t)a.g.hid[0] * ps], lmin = (double)base[(size_t)a.g.vid[0] * ps];
            for (int k = 1; k < 6; ++k) {
                lmax = nanmax(lmax, (double)base[(size_t)a.g.hid[k] * ps]);      // :116
                lmin = nanmin(lmin, (double)base[(size_t)a.g.vid[k] * ps]);      // :117
            }
            d = (lmax - lmin) * b_inv;                                           // :119
            const double t2 = 2.0 * (lmax - d);                                  // :120
            g = t2 * b_inv;
        }
        if (a.Ld_out) {
            a.Ld_out[p] = d;
            a.Lg_out[p] = g;
        }
        if (!(a.hc || a.h16 || a.h64)) continue;
        for (int k = 0; k < L; ++k) {
            const double hn = (double)base[(size_t)(2 + 2 * k) * ps], hi = (double)base[(size_t)(2 + 2 * L + 2 * k) * ps];
            const double vn = (double)base[(size_t)(3 + 2 * k) * ps], vi = (double)base[(size_t)(3 + 2 * L + 2 * k) * ps];
            const int ch = classify_literal(d, g, hn, hi, a.eps);
            const int cv = classify_literal(d, g, vn, vi, a.eps);
            if (a.hc) {
                a.hc[(size_t)k * a.npix + p] = (int8_t)ch;
                a.vc[(size_t)k * a.npix + p] = (int8_t)cv;
            }
            if (ch >= 0) ok_h |= 1u << k;
            if (cv >= 0) ok_v |= 1u << k;
            if (ch > 0) bits_h |= 1u << (L - 1 - k);     // h: first code is the MSB (:223-228)
            if (cv > 0) bits_v |= 1u << k;               // v: flipped, stored LSB first (src/3-capture_decode.py:100)
        }
    }
    if (a.h16 || a.h64) {
        const uint32_t full = (1u << L) - 1u;
        const uint32_t bin = gray_to_binary_2x16((bits_h << 16) | bits_v);
        const int hv = ok_h == full ? (int)(bin >> 16) : -1;
        const int vv = ok_v == full ? (int)(bin & 0xffffu) : -1;
        if (a.h16) {
            a.h16[p] = (int16_t)hv;
            a.v16[p] = (int16_t)vv;
        }
        if (a.h64) {
            a.h64[p] = hv;
            a.v64[p] = vv;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1c: int8 code planes -> maps
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_codes_to_pixels(const int8_t *__restrict__ hc, const int8_t *__restrict__ vc, int n_runs,
                                                         int L, size_t npix, int64_t *__restrict__ h, int64_t *__restrict__ v)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    uint32_t bits_h = 0, bits_v = 0;
    bool bad_h = false, bad_v = false;
    for (int k = 0; k < L; ++k) {
        int a = hc[(size_t)k * npix + p], b = vc[(size_t)k * npix + p];
        for (int r = 1; r < n_runs; ++r) {                                   // np.max over runs (:95-96)
            a = max(a, (int)hc[((size_t)r * L + k) * npix + p]);
            b = max(b, (int)vc[((size_t)r * L + k) * npix + p]);
        }
        bad_h |= a < 0;
        bad_v |= b < 0;
        bits_h |= (uint32_t)(a & 1) << (L - 1 - k);
        bits_v |= (uint32_t)(b & 1) << k;
    }
    const uint32_t bin = gray_to_binary_2x16((bits_h << 16) | bits_v);
    h[p] = bad_h ? -1 : (int64_t)(bin >> 16);
    v[p] = bad_v ? -1 : (int64_t)(bin & 0xffffu);
}

__global__ void __launch_bounds__(256) k_widen_maps(const int16_t *__restrict__ h16, const int16_t *__restrict__ v16, size_t npix,
                                                    int64_t *__restrict__ h, int64_t *__restrict__ v)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    h[p] = h16[p];
    v[p] = v16[p];
}

template <int PX, int BLOCK>
int launch_fast_t(slgc_ctx *ctx, FastArgs &a, size_t npix_main)
{
    a.ngroups = npix_main / PX;
    if (a.ngroups == 0) return SLGC_OK;
    const size_t blocks = (a.ngroups + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL((k_decode_fast<PX, BLOCK>), dim3((unsigned)blocks), dim3(BLOCK), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

}  // namespace

bool decode_fast_eligible(double eps, int *e_out)
{
    if (!(eps >= 0.0 && eps <= 255.0)) return false;
    const int e = (int)eps;
    if ((double)e != eps) return false;
    *e_out = e;
    return true;
}

// variant: 0 = auto; otherwise PX*1000 + BLOCK (e.g. 16256, 8064) for sweeps.
int launch_decode_fast(slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, int rows, int W, int e,
                       int16_t *d_h, int16_t *d_v, int variant)
{
    FastArgs a{};
    a.g = g;
    a.e = e;
    a.plane_stride = plane_stride;
    a.h = d_h;
    a.v = d_v;
    const size_t npix = (size_t)rows * W;
    uintptr_t align_or = (uintptr_t)plane_stride | ((uintptr_t)d_h >> 1) | ((uintptr_t)d_v >> 1);
    for (int r = 0; r < g.n_runs; ++r) {
        a.run[r] = (const uint8_t *)runs.p[r];
        align_or |= (uintptr_t)runs.p[r];
    }
    int px = (align_or % 16 == 0) ? 16 : (align_or % 8 == 0) ? 8 : (align_or % 4 == 0) ? 4 : 1;
    int block = 256;
    if (variant > 0) {
        const int want_px = variant / 1000;
        if (want_px != 16 && want_px != 8 && want_px != 4 && want_px != 1) return slgc_fail(ctx, SLGC_EINVAL, "bad variant %d", variant);
        if (want_px > px) return slgc_fail(ctx, SLGC_EINVAL, "variant %d needs %d-byte alignment", variant, want_px);
        px = want_px;
        block = variant % 1000;
        if (block != 64 && block != 128 && block != 256) return slgc_fail(ctx, SLGC_EINVAL, "bad variant %d", variant);
    } else if (px == 16) {
        px = 8;  // measured default, see DESIGN.md (sweep in profiles/)
    }
    const size_t main_pix = npix / px * px;
    int rc;
#define SLGC_CASE(P, B) \
    if (px == P && block == B) { rc = launch_fast_t<P, B>(ctx, a, main_pix); } else
    SLGC_CASE(16, 256) SLGC_CASE(16, 128) SLGC_CASE(16, 64) SLGC_CASE(8, 256) SLGC_CASE(8, 128) SLGC_CASE(8, 64)
    SLGC_CASE(4, 256) SLGC_CASE(4, 128) SLGC_CASE(4, 64) SLGC_CASE(1, 256) { rc = launch_fast_t<1, 256>(ctx, a, main_pix); }
#undef SLGC_CASE
    if (rc) return rc;
    if (main_pix < npix) {  // ragged tail (< px pixels): byte-wide groups
        FastArgs t = a;
        for (int r = 0; r < g.n_runs; ++r) t.run[r] = a.run[r] + main_pix;
        t.h = d_h + main_pix;
        t.v = d_v + main_pix;
        rc = launch_fast_t<1, 256>(ctx, t, npix - main_pix);
    }
    return rc;
}

int launch_decode_generic(slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, int dtype, size_t plane_stride_elems,
                          size_t npix, double eps, const double *d_Ld_in, const double *d_Lg_in, double *d_Ld_out,
                          double *d_Lg_out, int8_t *d_hc, int8_t *d_vc, int16_t *d_h16, int16_t *d_v16, int64_t *d_h64,
                          int64_t *d_v64)
{
    if (npix == 0) return SLGC_OK;
    GenArgs a{};
    for (int r = 0; r < g.n_runs; ++r) a.run[r] = runs.p[r];
    a.plane_stride = plane_stride_elems;
    a.npix = npix;
    a.g = g;
    a.eps = eps;
    a.Ld_in = d_Ld_in; a.Lg_in = d_Lg_in; a.Ld_out = d_Ld_out; a.Lg_out = d_Lg_out;
    a.hc = d_hc; a.vc = d_vc; a.h16 = d_h16; a.v16 = d_v16; a.h64 = d_h64; a.v64 = d_v64;
    const size_t blocks = (npix + 255) / 256;
    if (dtype == SLGC_U8)
        hipLaunchKernelGGL(k_decode_generic<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(k_decode_generic<double>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_codes_to_pixels(slgc_ctx *ctx, const int8_t *d_hc, const int8_t *d_vc, int n_runs, int L, size_t npix,
                           int64_t *d_h, int64_t *d_v)
{
    if (npix == 0) return SLGC_OK;
    const size_t blocks = (npix + 255) / 256;
    hipLaunchKernelGGL(k_codes_to_pixels, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_hc, d_vc, n_runs, L, npix, d_h, d_v);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_widen_maps(slgc_ctx *ctx, const int16_t *d_h16, const int16_t *d_v16, size_t npix, int64_t *d_h, int64_t *d_v)
{
    if (npix == 0) return SLGC_OK;
    hipLaunchKernelGGL(k_widen_maps, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, d_h16, d_v16, npix, d_h, d_v);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}
