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
#include <cstdlib>

#include "slgc_internal.h"
#include "tri_math.h"

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

// Per-pixel thresholds from black/white and the 12 L_max/L_min frames: the reference's float64 sequence (decode_codes.py:113-120) with every
// rounding kept and everything around the roundings done more cheaply -- float64 issues at half rate on gfx950 and this block was a fifth of
// the decode kernel's vector-ALU time (round 3; exhaustively equal to the literal predicates over the whole uint8 domain: slgc_selftest_thresholds):
//  * b_inv = white / (white + black): the operands are integers 0 .. 510, so the quotient needs none of the scaling / fix-up steps of the
//    general IEEE division (v_div_scale x2, v_div_fmas, v_div_fixup): reciprocal seed, two Newton steps, quotient, one correction = the same
//    correctly rounded result (0 / 0: -0 * inf = NaN propagates, as the reference's NaN does);
//  * white + black is added as integers (one conversion instead of two and an add);
//  * L_g + eps = fl(fl(2 a) * b_inv) + eps with a = L_max - L_d: doubling is exact, so fl(2 a * b_inv) = 2 fl(a * b_inv) and the sum is ONE
//    fma(a * b_inv, 2, eps) -- the same single rounding of the same real number;
//  * the clamps run on integers after the conversion: L_g + eps >= 0 (a >= 0 because L_d <= L_max - L_min <= L_max), so truncation is floor,
//    and NaN (white = black = 0, the only way to get one) is an integer test on white + black instead of float64 compares / min / max.
__device__ __forceinline__ void pixel_thresholds(int black, int white, int lmax, int lmin, int e, int &tt, int &cA)
{
    const int si = white + black;
    const double w = (double)white, s = (double)si;
    double y = __builtin_amdgcn_rcp(s);
    y = __builtin_fma(__builtin_fma(-s, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-s, y, 1.0), y, y);
    const double q0 = w * y;
    const double b_inv = __builtin_fma(__builtin_fma(-s, q0, w), y, q0);      // decode_codes.py:113 (0/0 -> NaN)
    const double ld = (double)(lmax - lmin) * b_inv;                         // :119  (L_max - L_min is exact)
    const double a = (double)lmax - ld;                                       // :120  2.0*(L_max - L_d) ...
    const double ge = __builtin_fma(a * b_inv, 2.0, (double)e);               //       ... * b_inv, + eps (:172-182)
    const bool direct = ld > ge;                                              // false for NaN
    const bool nan = si == 0;
    const int tg = nan ? 256 : min((int)ge, 255) + 1;                         // floor(ge) + 1 in [1, 256]; NaN: no x satisfies x > NaN
    const int tnd = nan ? 0 : max((int)__builtin_ceil(ld) - e, 0);            // ceil(ld) - e in [0, 255]; NaN: no x satisfies x + e < NaN
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
// K1a-pk: the same integer-threshold algorithm, two pixels per 32-bit register (packed 16-bit lanes), branch-free,
// all boolean work on the VALU (no lane-mask logic on the scalar unit).
// ------------------------------------------------------------------------------------------
//
// A dword of a frame holds pixels p0..p3.  It is split into the pairs E = [p0, p2] and O = [p1, p3] (one v_perm /
// v_and each), so every later instruction handles two pixels.  For a pair register X (values 0..255 per 16-bit half):
//     X + (0x8000 - t)             has bit 15 of a half set  <=>  x >= t         (threshold tests, t in 0..256)
//     D = N - I (one 32-bit subtraction over both halves; the borrow a negative low half takes is given back by the next addition)
//     D + (0x8000 - c)             has bit 15 set  <=>  n - i >= c  (r1) ;    D + (0x8000 - 1 + c)  has bit 15 clear  <=>  i - n >= c  (r2)
// Every one of these sums stays inside its 16-bit half (no carry, no borrow: ranges in pack_pair_consts), so they are plain 32-bit
// v_add_u32 / v_sub_u32 -- full rate on gfx950, where the packed v_pk_add_u16 / v_pk_sub_i16 / v_pk_lshrrev_b16 / v_and_or_b32 this
// kernel first used issue at HALF rate (tools/ubench/valu_rates.hip: 2.5 vs 4.5 cycles per wave instruction).  The rule table is
// evaluated on bit 15 / bit 31 with three-input boolean ops (v_bitop3_b32, full rate):
//     bit   = (Nb & ~Ia) | (R1 & (Na | ~Ib))                  (r4 | (r1 & ~r3))
//     valid = (R1 | (~Na & Ib)) | (~R2 | (Nb & ~Ia))          (r1 | r3 | r2 | r4)
// Step t deposits its bit at position t of each half (acc |= (bit >> (15 - t)) & (0x00010001 << t): a 32-bit shift and one bitop3; the
// bits a 32-bit shift drags across the halves are masked away), so the column code walks k = L-1 .. 0 and the row code k = 0 .. L-1.
// Frames are fetched with buffer loads: per-lane 32-bit offset in a VGPR, per-frame offset in an SGPR, hardware bounds check instead of
// a tail branch.
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
typedef short v2ss __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct FuseArgs {            // triangulation appended to the decode kernel (slgc_scan_dev)
    const float2 *cam_lut;    // [npix] camera rays of this band
    CamNodes cn;              // the same rays at every 4th column (cn.nodes == nullptr: read cam_lut)
    const float2 *proj_lut;   // tiled projector rays (the guarded redo reads them)
    const float *proj_th;     // same index: tan(beta / 2) per projector pixel, 4 bytes -- what the fast form gathers
    float *xyz;               // [npix][3]
    int proj_w, proj_h, tiles_x, wide;   // projector table geometry (proj_lut_index)
    int nt_store;             // bit 0: XYZ, bit 1: maps leave with non-temporal stores (products nothing re-reads); bit 2: the maps are not stored at all
    int wave_tail;            // 1: wave-local LDS exchange in the tail (no workgroup barriers)
    int prio_head, prio_body, prio_tail;   // s_setprio of a wave while it fetches its threshold frames and computes its thresholds / walks the bit loop / runs the tail
    uint32_t xcd_chunk;       // XCD-aware workgroup -> tile map (slgc_internal.h: xcd_block), 0 = identity
    uint32_t xcd_run;         // ... or its fine-grained form (xcd_block_fine): tiles per XCD inside a group of 8 * xcd_run, 0 = off
    uint32_t batch_bps, batch_magic;   // slgc_scan_batch_dev: workgroups per scan (0 = one scan) and ceil(2^32 / batch_bps) for the division
    uint64_t batch_stride;    // bytes between the stacks of consecutive scans (maps and XYZ of consecutive scans are npix apart)
    TriF32 kf;                // T / |T| and |T| in float32 for the fast form
    double T[3], t_len;
#ifdef SLGC_STAMPS            // stamp build (make variant NAME=stamps EXTRA=-DSLGC_STAMPS; tools/time_stamps.py): 5 x s_memrealtime per wave, into memory nothing else reads
    unsigned long long *stamps;
#endif
};

// Issue priority of the wave among the waves of its SIMD (s_setprio takes an immediate: a wave-uniform switch selects it).  Which phase of
// k_decode_pk goes first decides how full the memory pipeline stays (round 5, notes/r05.md): a wave that still has to ask for its threshold
// frames goes first at every size; in launches of a single round of resident waves the tail goes last.
__device__ __forceinline__ void set_prio(int p)
{
    switch (p) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: __builtin_amdgcn_s_setprio(0); break;
    }
}

// Everything in k_decode_pk that is NOT the product, named in one place.  Diag<0> is the shipped kernel: every flag below is false and every
// `if constexpr (D::...)` folds away; the timing-only ablations (diagnostic build `make diag`, results wrong on purpose unless noted) and the
// stamp build (make variant NAME=stamps EXTRA=-DSLGC_STAMPS; tools/time_stamps.py) are the other instantiations.
template <int ABL>
struct Diag {
    static constexpr bool product = ABL == 0;
    static constexpr bool skip_threshold_frames = ABL == 1 || ABL == 2;   // the 14 threshold-frame loads and the thresholds are replaced by constants
    static constexpr bool loads_only = ABL == 2;                          // frames are fetched, nothing is classified
    static constexpr bool cheap_thresholds = ABL == 3;                    // integer stand-in for the float64 threshold arithmetic
    static constexpr bool stores_never = ABL == 5;                        // XYZ stores behind a condition that is never true
    static constexpr bool no_gather = ABL == 6;                           // projector-table gathers replaced by a constant
    static constexpr bool cam_rays_cached = ABL == 7;                     // every lane reads the same 2 KB of the camera table
    static constexpr bool unguarded = ABL == 8;                           // no flat-triangle redo
    static constexpr bool branchy_gather = ABL == 9;                      // round 1's gather form (correct results): a branch around every gather
};

// stamp i of the wave (stamp build only; compiles to nothing otherwise): 0 wave started, 1 thresholds computed, 2 every frame consumed,
// 3 tail done and stores issued, 4 stores acknowledged
template <int FUSE, int BLOCK, typename Args>
__device__ __forceinline__ void stamp(const Args &a, int i)
{
#ifdef SLGC_STAMPS
    if (FUSE != 0 && a.f.stamps && (threadIdx.x & 63) == 0)
        a.f.stamps[((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 5 + i] = __builtin_amdgcn_s_memrealtime();
#else
    (void)a; (void)i;
#endif
}

// cv2.cvtColor(BGR2GRAY) on 8-bit pixels inside the frame load (src/3-capture_decode.py:66): Y = (B*BY + G*GY + R*RY + rnd) >> shift.
// A lane's 4 pixels arrive as 12 bytes B0 G0 R0 B1 G1 R1 B2 G2 R2 B3 G3 R3 in three dwords; pixel 0 IS dword 0 with a zero coefficient for
// its 4th byte, pixels 1 and 2 are two v_alignbyte_b32 away, pixel 3 is dword 2 with a zero coefficient for its 1st byte.  The 14 / 15-bit
// coefficients are split into high and low bytes so that each pixel is two v_dot4_u32_u8 and one v_lshl_add_u32:
//     Y * 2^shift + remainder = (dot(px, hi) << 8) + dot(px, lo) + rnd          (< 2^23)
// -> the grey dword [Y0 Y1 Y2 Y3] the rest of the kernel works on (21 full-rate instructions per frame and lane).
struct Luma {
    uint32_t a_hi, a_lo;     // coefficient bytes [BY, GY, RY, 0], high / low byte of each
    uint32_t z_hi, z_lo;     // [0, BY, GY, RY]
    uint32_t rnd, shift;
};

__device__ __forceinline__ uint32_t bgr4_to_gray(uint32_t w0, uint32_t w1, uint32_t w2, const Luma &c)
{
    const uint32_t p1 = __builtin_amdgcn_alignbyte(w1, w0, 3), p2 = __builtin_amdgcn_alignbyte(w2, w1, 2);
    const uint32_t y0 = (__builtin_amdgcn_udot4(w0, c.a_hi, 0u, false) << 8) + __builtin_amdgcn_udot4(w0, c.a_lo, c.rnd, false);
    const uint32_t y1 = (__builtin_amdgcn_udot4(p1, c.a_hi, 0u, false) << 8) + __builtin_amdgcn_udot4(p1, c.a_lo, c.rnd, false);
    const uint32_t y2 = (__builtin_amdgcn_udot4(p2, c.a_hi, 0u, false) << 8) + __builtin_amdgcn_udot4(p2, c.a_lo, c.rnd, false);
    const uint32_t y3 = (__builtin_amdgcn_udot4(w2, c.z_hi, 0u, false) << 8) + __builtin_amdgcn_udot4(w2, c.z_lo, c.rnd, false);
    const uint32_t e = (y0 >> c.shift) | ((y2 >> c.shift) << 16), o = (y1 >> c.shift) | ((y3 >> c.shift) << 16);
    return e | (o << 8);
}

struct PkArgs {
    FuseArgs f;
    const uint8_t *run[SLGC_MAX_RUNS];
    uint32_t plane_stride;   // bytes between frames (< 2^32 / N)
    uint32_t npix;           // pixels in the band (multiple of PX handled here; ragged tail by the byte-wide kernel)
    uint32_t run_bytes;      // (N-1)*plane_stride + npix : bounds of the read descriptor
    uint32_t tile_log2;      // 0 = planar stack [N][npix]; k = tile-interleaved [tile][N][2^k bytes]: plane_stride = 2^k, a tile's N pieces are contiguous
    uint32_t tile_bytes;     // N << tile_log2
    int16_t *h;
    int16_t *v;
    DecodeGeom g;
    int e;
    Luma lum;                // BGR kernels only (slgc_scan_bgr_dev): cv2.cvtColor's fixed-point luma
};

__device__ __forceinline__ v2us as_us(uint32_t x) { return __builtin_bit_cast(v2us, x); }
__device__ __forceinline__ v2ss as_ss(uint32_t x) { return __builtin_bit_cast(v2ss, x); }
__device__ __forceinline__ uint32_t as_u(v2us x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t as_u(v2ss x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u(as_us(a) + as_us(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u(as_us(a) - as_us(b)); }
__device__ __forceinline__ uint32_t pk_shr1(uint32_t a) { return as_u(as_us(a) >> (unsigned short)1); }
__device__ __forceinline__ uint32_t even_pair(uint32_t w) { return w & 0x00ff00ffu; }                                  // [p0, p2]
__device__ __forceinline__ uint32_t odd_pair(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c030c01u); }        // [p1, p3]

template <int NW, int NT>
struct Frame {
    uint32_t w[NW];
};

template <int NW, int NT>
__device__ __forceinline__ Frame<NW, NT> load_frame(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff)
{
    Frame<NW, NT> f;
    constexpr int aux = NT ? 2 : 0;      // nt (scope bits on top of it measured within noise)
    if constexpr (NW == 1) {
        f.w[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, aux);
    } else if constexpr (NW == 2) {
        const v2u t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, aux);
        f.w[0] = t.x; f.w[1] = t.y;
    } else if constexpr (NW == 3) {           // four BGR pixels: 12 bytes, 4-byte aligned
        const v3u t = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, soff, aux);
        f.w[0] = t.x; f.w[1] = t.y; f.w[2] = t.z;
    } else {
        const v4u t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, aux);
        f.w[0] = t.x; f.w[1] = t.y; f.w[2] = t.z; f.w[3] = t.w;
    }
    return f;
}

// Thresholds of the two pixels of a pair register -> the four packed constants classify_pk adds
// (tt = tnd | tg << 16 and cc = cA from pixel_thresholds; lo / hi = the register's low / high half).  Ranges, per half, with
// n, i in 0..255, tnd, tg in 0..256, cA in 1..256 or kUnreachable = 0x4000:
//   x + KA,  x + KB   in 0x7f00 .. 0x80ff          d = n - i in -255 .. 255
//   d + K1 (K1 = 0x8000 - cA     in 0x4000 .. 0x7fff)   in 0x3f01 .. 0x80fe
//   d + K2 (K2 = 0x8000 - 1 + cA in 0x8000 .. 0xbfff)   in 0x7f01 .. 0xc0fe          -- all inside 16 bits: no carry between the halves.
// d is formed over both halves at once as the 32-bit N - I: a negative low half borrows from the high half, and the addition of K1 / K2
// gives the borrow back -- the sum of the per-half values d + K (each inside its 16 bits) is the same integer either way.
__device__ __forceinline__ void pack_pair_consts(int tt_lo, int cc_lo, int tt_hi, int cc_hi, uint32_t &KA, uint32_t &KB, uint32_t &K1, uint32_t &K2)
{
    const uint32_t a_lo = (uint32_t)tt_lo & 0xffffu, a_hi = (uint32_t)tt_hi & 0xffffu;
    const uint32_t b_lo = (uint32_t)tt_lo >> 16, b_hi = (uint32_t)tt_hi >> 16;
    KA = (0x8000u - a_lo) | ((0x8000u - a_hi) << 16);
    KB = (0x8000u - b_lo) | ((0x8000u - b_hi) << 16);
    K1 = (0x8000u - (uint32_t)cc_lo) | ((0x8000u - (uint32_t)cc_hi) << 16);
    K2 = (0x8000u - 1u + (uint32_t)cc_lo) | ((0x8000u - 1u + (uint32_t)cc_hi) << 16);
}

// (a & b) | c as ONE full-rate v_bitop3_b32 (the compiler's own choice for this shape, v_and_or_b32, issues at half rate on gfx950)
__device__ __forceinline__ uint32_t and_or_full_rate(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA);
}

// One (normal, inverse) pair register through the rule table and into the accumulators: bit 15 / bit 31 of every intermediate = the
// pixel's answer; step t deposits its code bit at bit t of each half (shift = 15 - t, mask = 0x00010001 << t).
// v_bitop3_b32 throughout (truth-table index = a << 2 | b << 1 | c), spelled out: left to itself the compiler closes these expressions
// with v_and_or_b32 / v_or3_b32, which issue at half rate.  Single run (MULTI = false): "classified" is folded straight into the running
// AND -- accV & (u1 | u2) is one three-input op, so the pair costs 7 additions + 5 boolean ops + 2 for the deposit (round 4: 15).
template <bool MULTI>
__device__ __forceinline__ void classify_pk(uint32_t N, uint32_t I, uint32_t KA, uint32_t KB, uint32_t K1, uint32_t K2,
                                            uint32_t &accB, uint32_t &accV, uint32_t shift, uint32_t mask)
{
    const uint32_t Na = N + KA, Nb = N + KB, Ia = I + KA, Ib = I + KB;
    const uint32_t D = N - I;
    const uint32_t R1 = D + K1, R2 = D + K2;
    const uint32_t t1 = __builtin_amdgcn_bitop3_b32(R1, Na, Ib, 0xD0);         // R1 & (Na | ~Ib)              = r1 & ~r3
    const uint32_t bit = __builtin_amdgcn_bitop3_b32(Nb, Ia, t1, 0xBA);        // (Nb & ~Ia) | t1              = r4 | (r1 & ~r3)
    const uint32_t u1 = __builtin_amdgcn_bitop3_b32(Na, Ib, R2, 0x5D);         // (~Na & Ib) | ~R2             = r3 | r2
    const uint32_t u2 = __builtin_amdgcn_bitop3_b32(Nb, Ia, R1, 0xBA);         // (Nb & ~Ia) | R1              = r4 | r1
    accB = and_or_full_rate(bit >> shift, mask, accB);
    if constexpr (MULTI) {
        const uint32_t ok = u1 | u2;                                           // r1 | r2 | r3 | r4
        accV = and_or_full_rate(ok >> shift, mask, accV);
    } else {
        accV = __builtin_amdgcn_bitop3_b32(accV, u1, u2, 0xE0);                // accV & (u1 | u2)
    }
}

// Compile-time twin of slgc_make_geom (decode_codes.py:109-111, :149) for the frame counts the generator and BASELINE.json use:
// which 12 frames feed L_max / L_min, and where each of them is parked in LDS (slot 0..11) for its second use in the bit loop.
template <int NS>
struct FrameSpec {
    static constexpr int L = (NS - 2) / 4;
    static constexpr int thr(int k)          // absolute frame index of threshold frame k (0..5 -> L_max, 6..11 -> L_min)
    {
        const double plf = (double)(NS - 2) / 4.0;
        const double id[12] = {2 * plf - 2, 2 * plf - 4, 2 * plf - 6, 4 * plf - 2, 4 * plf - 4, 4 * plf - 6, 1, 3, 5, 2 * plf + 1, 2 * plf + 3, 2 * plf + 5};
        return 2 + (int)(unsigned char)id[k];
    }
    struct Table {
        signed char slot[72];
    };
    static constexpr Table make()
    {
        Table t{};
        for (int f = 0; f < 72; ++f) t.slot[f] = -1;
        for (int k = 11; k >= 0; --k) t.slot[thr(k)] = (signed char)k;     // (a frame used by both lists keeps its lowest slot)
        return t;
    }
    static constexpr Table table = make();
};

#ifndef SLGC_PARK_DEPTH
#define SLGC_PARK_DEPTH 2      // steps of frame loads in flight ahead of the step being classified (specialised kernels)
#endif
#ifndef SLGC_MIN_WAVES
#define SLGC_MIN_WAVES 8       // waves per SIMD the 4-pixels-per-lane kernels are held to (<= 64 VGPRs); variant builds relax it for deeper prefetch (A/B)
#endif
constexpr int kWaveLdsBytes = 4096;     // LDS block of one wave: fused tail [indices 1 KB | rays / XYZ 3 KB], aliased by the 12 parked frames (3 KB)
constexpr int kWaveListBytes = 256;     // FUSE == 3: the wave's list of flat pixels (one byte each), behind the blocks of all waves -- inside the same
                                        // 1280-byte LDS allocation granule of gfx950 as the two 4 KB blocks (8704 -> 8960 B, as 8192 is): no wave less per CU

// ABL: 0 = the product; anything else selects a timing-only ablation, see Diag<ABL> above (diagnostic build only).
// NS > 0: specialised for NS frames per run (compile-time frame indices, bit loop fully unrolled) -- the 12 frames that feed the
// per-pixel thresholds are parked in lane-private LDS words after their first use, so the bit loop reads them from LDS instead of
// fetching them a second time (the re-reads are L2 hits, but each still costs a vector-memory instruction and its L2 -> CU trip:
// 11 of the 54 loads per lane at N = 44).  The parked words are read back by the lane that wrote them: no synchronisation.
// BGR = 1 (slgc_scan_bgr_dev): the planes hold the camera's BGR frames, 3 bytes per pixel; a lane loads its four pixels as one dwordx3 and forms
// OpenCV's 8-bit luma in registers -- the grey stack of src/3-capture_decode.py:66-70 never exists in HBM.  Raw frames in flight are three
// registers each: 4 waves per SIMD (<= 128 VGPRs), every wave with three times the bytes in flight.
template <int PX, int BLOCK, int NT, bool MULTI, int ABL = 0, int FUSE = 0, int NS = 0, int BGR = 0>
__global__ void __launch_bounds__(BLOCK, (PX == 4 && (ABL == 0 || (ABL == 3 && NS != 0))) ? (BGR ? 4 : SLGC_MIN_WAVES) : 1) k_decode_pk(const PkArgs a)      // 4 px / lane: 8 waves per SIMD (<= 64 VGPRs)
{
    constexpr int NW = PX / 4;      // dwords per lane per frame
    constexpr int NR = BGR ? 3 : NW;      // ... as loaded
    static_assert(!BGR || (NS > 0 && NW == 1), "the BGR kernels are the specialised ones");
    constexpr int NP = PX / 2;      // pixel-pair registers per lane
    constexpr bool SPEC = NS > 0;
    using D = Diag<ABL>;
    using FS = FrameSpec<SPEC ? NS : 14>;
    static_assert(!SPEC || (NW == 1 && FUSE != 1 && (ABL == 0 || ABL == 3)), "the specialised kernel is 4 pixels per lane, wave-local tail, no ablations but the cheap thresholds");
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[(FUSE != 0 || SPEC) ? (BLOCK / 64) * (kWaveLdsBytes + (FUSE == 3 ? kWaveListBytes : 0)) : 16];
    uint32_t *const park = reinterpret_cast<uint32_t *>(s_raw + (threadIdx.x >> 6) * kWaveLdsBytes) + (threadIdx.x & 63);   // + slot * 64
    uint32_t bid = FUSE != 0 ? (a.f.xcd_run ? xcd_block_fine(blockIdx.x, a.f.xcd_run, gridDim.x) : xcd_block(blockIdx.x, a.f.xcd_chunk)) : blockIdx.x;   // fused scan: optional XCD-aware tile maps (A/B)
    uint32_t scan = 0u;                                                                      // batched launch: which of the independent scans this workgroup belongs to
    if constexpr (FUSE != 0) {
        if (a.f.batch_bps) {
            scan = __umulhi(bid, a.f.batch_magic);                                           // bid / batch_bps, exact for bid * batch_bps < 2^32 (checked by the launcher)
            bid -= scan * a.f.batch_bps;
        }
    }
    stamp<FUSE, BLOCK>(a, 0);
    set_prio(a.f.prio_head);
    const uint32_t off = (bid * BLOCK + threadIdx.x) * PX;
    uint32_t voff = BGR ? 3u * off : off;            // byte offset of the lane's pixels inside a plane
    if (!BGR && a.tile_log2) voff = (off >> a.tile_log2) * a.tile_bytes + (off & ((1u << a.tile_log2) - 1u));      // ... inside plane 0's piece of the lane's tile
    const uint32_t ps = a.plane_stride;
    const int L = SPEC ? FS::L : a.g.L;
    uint32_t mB_h[NP], mB_v[NP], mV_h[NP], mV_v[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { mB_h[p] = mB_v[p] = 0u; mV_h[p] = mV_v[p] = MULTI ? 0u : 0xffffffffu; }

    for (int r = 0; r < a.g.n_runs; ++r) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.run[r] + (FUSE != 0 ? scan * a.f.batch_stride : 0ull)), 0, a.run_bytes, 0x00020000);
        uint32_t KA[NP], KB[NP], C1[NP], C2[NP];      // C1 / C2 hold K1 / K2 of pack_pair_consts
        // specialised kernels: DEPTH steps of frame loads stay in flight ahead of the step being classified
        constexpr int DEPTH = SLGC_PARK_DEPTH;
        Frame<NR, NT> ring[SPEC ? DEPTH + 1 : 1][4];
        auto fetch = [&](int f) {                   // f is a constant once the loop is unrolled: the choice below folds away
            Frame<NR, NT> fr;
            const int slot = FS::table.slot[f];
            if (slot >= 0) fr.w[0] = park[slot * 64];
            else fr = load_frame<NR, NT>(rs, voff, (uint32_t)f * ps);
            return fr;
        };
        auto gray_of = [&](const Frame<NR, NT> &fr, int f) -> Frame<NW, NT> {      // the grey dword(s) of a fetched frame (parked frames were converted before they were parked)
            if constexpr (BGR) {
                Frame<NW, NT> g;
                g.w[0] = FS::table.slot[f] >= 0 ? fr.w[0] : bgr4_to_gray(fr.w[0], fr.w[1], fr.w[2], a.lum);
                return g;
            } else {
                return fr;
            }
        };
        auto fetch_step = [&](int t, Frame<NR, NT> (&fr)[4]) {
            const int f_hn = 2 + 2 * (FS::L - 1 - t), f_vn = 3 + 2 * t;
            fr[0] = fetch(f_hn); fr[1] = fetch(f_hn + 2 * FS::L); fr[2] = fetch(f_vn); fr[3] = fetch(f_vn + 2 * FS::L);
        };
        if constexpr (D::skip_threshold_frames) {
#pragma unroll
            for (int p = 0; p < NP; ++p) { KA[p] = 0x7fb07fb0u + off; KB[p] = 0x7fd07fd0u; C1[p] = 0x00020002u; C2[p] = 0x00010001u; }
        } else {
            constexpr int TNT = SPEC ? NT : 0;      // parked frames are fetched once: streaming policy; otherwise cacheable (re-read from L2)
            Frame<NW, TNT> bl, wh, hm[6], vm[6];
            {
                const Frame<NR, TNT> r_bl = load_frame<NR, TNT>(rs, voff, 0), r_wh = load_frame<NR, TNT>(rs, voff, ps);
                Frame<NR, TNT> r_hm[6], r_vm[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    r_hm[k] = load_frame<NR, TNT>(rs, voff, (uint32_t)(SPEC ? FS::thr(k) : a.g.hid[k]) * ps);
                    r_vm[k] = load_frame<NR, TNT>(rs, voff, (uint32_t)(SPEC ? FS::thr(6 + k) : a.g.vid[k]) * ps);
                }
                if constexpr (BGR) {
                    bl.w[0] = bgr4_to_gray(r_bl.w[0], r_bl.w[1], r_bl.w[2], a.lum);
                    wh.w[0] = bgr4_to_gray(r_wh.w[0], r_wh.w[1], r_wh.w[2], a.lum);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        hm[k].w[0] = bgr4_to_gray(r_hm[k].w[0], r_hm[k].w[1], r_hm[k].w[2], a.lum);
                        vm[k].w[0] = bgr4_to_gray(r_vm[k].w[0], r_vm[k].w[1], r_vm[k].w[2], a.lum);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NW; ++q) {
                        bl.w[q] = r_bl.w[q]; wh.w[q] = r_wh.w[q];
#pragma unroll
                        for (int k = 0; k < 6; ++k) { hm[k].w[q] = r_hm[k].w[q]; vm[k].w[q] = r_vm[k].w[q]; }
                    }
                }
            }
            if constexpr (SPEC) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    park[k * 64] = hm[k].w[0];
                    park[(6 + k) * 64] = vm[k].w[0];
                }
                // every parked word is read back by the lane that wrote it; with the loop unrolled the compiler would forward the
                // stores to those loads, i.e. keep all 12 frames in registers (the variant that loses two waves per SIMD): an opaque
                // point between the stores and the loads stops that
                asm volatile("" ::: "memory");
                // the first DEPTH steps of the bit loop are requested NOW, before the float64 threshold arithmetic (~600 cycles per wave):
                // their latency runs under it instead of after it
#pragma unroll
                for (int d = 0; d < DEPTH; ++d)
                    if (d < FS::L) fetch_step(d, ring[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                uint32_t mx = hm[0].w[q], mn = vm[0].w[q];
                uint32_t mxe = even_pair(mx), mxo = odd_pair(mx), mne = even_pair(mn), mno = odd_pair(mn);
#pragma unroll
                for (int k = 1; k < 6; ++k) {
                    const uint32_t hw = hm[k].w[q], vw = vm[k].w[q];
                    mxe = as_u(__builtin_elementwise_max(as_us(mxe), as_us(even_pair(hw))));      // :116
                    mxo = as_u(__builtin_elementwise_max(as_us(mxo), as_us(odd_pair(hw))));
                    mne = as_u(__builtin_elementwise_min(as_us(mne), as_us(even_pair(vw))));      // :117
                    mno = as_u(__builtin_elementwise_min(as_us(mno), as_us(odd_pair(vw))));
                }
                int tt[4], cc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t mxp = (j & 1) ? mxo : mxe, mnp = (j & 1) ? mno : mne;
                    const int lmax = (int)((j & 2) ? (mxp >> 16) : (mxp & 0xffffu));
                    const int lmin = (int)((j & 2) ? (mnp >> 16) : (mnp & 0xffffu));
                    const int black = (int)((bl.w[q] >> (8 * j)) & 0xffu), white = (int)((wh.w[q] >> (8 * j)) & 0xffu);
                    if constexpr (D::cheap_thresholds) { tt[j] = ((lmax - lmin) & 0xff) | ((black + white) << 15 & 0xff0000); cc[j] = a.e + 1; }
                    else pixel_thresholds(black, white, lmax, lmin, a.e, tt[j], cc[j]);
                }
                // pair registers: even = pixels (0, 2), odd = pixels (1, 3) of this dword
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    const int lo = par, hi = par + 2, p = 2 * q + par;
                    pack_pair_consts(tt[lo], cc[lo], tt[hi], cc[hi], KA[p], KB[p], C1[p], C2[p]);
                }
            }
        }
        uint32_t aB_h[NP], aB_v[NP], aV_h[NP], aV_v[NP];
#ifdef SLGC_STAMPS
        asm volatile("" : "+v"(KA[0]), "+v"(KB[0]), "+v"(C1[0]), "+v"(C2[0]));      // the thresholds exist
#endif
        stamp<FUSE, BLOCK>(a, 1);
        set_prio(a.f.prio_body);
#pragma unroll
        for (int p = 0; p < NP; ++p) { aB_h[p] = aB_v[p] = 0u; aV_h[p] = aV_v[p] = MULTI ? 0u : 0xffffffffu; }
        // step t: column code bit k = L-1-t (its weight 2^t), row code bit k = t (its weight 2^t)
        auto step = [&](const Frame<NW, NT> &hn, const Frame<NW, NT> &hi, const Frame<NW, NT> &vn, const Frame<NW, NT> &vi, uint32_t t) {
            const uint32_t sh = 15u - t, mk = 0x00010001u << t;
            if constexpr (D::loads_only) {
#pragma unroll
                for (int q = 0; q < NW; ++q) { aB_h[2 * q] ^= hn.w[q] + hi.w[q]; aB_v[2 * q] ^= vn.w[q] + vi.w[q]; }
            } else {
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    classify_pk<MULTI>(even_pair(hn.w[q]), even_pair(hi.w[q]), KA[2 * q], KB[2 * q], C1[2 * q], C2[2 * q], aB_h[2 * q], aV_h[2 * q], sh, mk);
                    classify_pk<MULTI>(odd_pair(hn.w[q]), odd_pair(hi.w[q]), KA[2 * q + 1], KB[2 * q + 1], C1[2 * q + 1], C2[2 * q + 1], aB_h[2 * q + 1], aV_h[2 * q + 1], sh, mk);
                    classify_pk<MULTI>(even_pair(vn.w[q]), even_pair(vi.w[q]), KA[2 * q], KB[2 * q], C1[2 * q], C2[2 * q], aB_v[2 * q], aV_v[2 * q], sh, mk);
                    classify_pk<MULTI>(odd_pair(vn.w[q]), odd_pair(vi.w[q]), KA[2 * q + 1], KB[2 * q + 1], C1[2 * q + 1], C2[2 * q + 1], aB_v[2 * q + 1], aV_v[2 * q + 1], sh, mk);
                }
            }
        };
        if constexpr (SPEC) {
            // the scheduling barrier keeps the unrolled steps from being hoisted on top of each other (unbounded, that costs 270 registers and
            // all but one wave per SIMD)
#pragma unroll
            for (int t = 0; t < FS::L; ++t) {
                if (t + DEPTH < FS::L) fetch_step(t + DEPTH, ring[(t + DEPTH) % (DEPTH + 1)]);
                Frame<NR, NT> (&cur)[4] = ring[t % (DEPTH + 1)];
                const int f_hn = 2 + 2 * (FS::L - 1 - t), f_vn = 3 + 2 * t;
                step(gray_of(cur[0], f_hn), gray_of(cur[1], f_hn + 2 * FS::L), gray_of(cur[2], f_vn), gray_of(cur[3], f_vn + 2 * FS::L), (uint32_t)t);
                // pin the accumulators here: left alone, the optimiser defers the whole "classified?" chain of every unrolled step to the
                // end of the loop and keeps each step's intermediates alive until then (230+ registers)
                asm volatile("" : "+v"(aB_h[0]), "+v"(aB_h[1]), "+v"(aB_v[0]), "+v"(aB_v[1]), "+v"(aV_h[0]), "+v"(aV_h[1]), "+v"(aV_v[0]), "+v"(aV_v[1]));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            uint32_t s_hn = (uint32_t)(2 + 2 * (L - 1)) * ps, s_hi = (uint32_t)(2 + 2 * L + 2 * (L - 1)) * ps;
            uint32_t s_vn = 3u * ps, s_vi = (uint32_t)(3 + 2 * L) * ps;
#pragma unroll 2
            for (int t = 0; t < L; ++t) {
                const Frame<NW, NT> hn = load_frame<NW, NT>(rs, voff, s_hn), hi = load_frame<NW, NT>(rs, voff, s_hi);
                const Frame<NW, NT> vn = load_frame<NW, NT>(rs, voff, s_vn), vi = load_frame<NW, NT>(rs, voff, s_vi);
                s_hn -= 2 * ps; s_hi -= 2 * ps; s_vn += 2 * ps; s_vi += 2 * ps;
                step(hn, hi, vn, vi, (uint32_t)t);
            }
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            mB_h[p] |= aB_h[p]; mB_v[p] |= aB_v[p];
            if constexpr (MULTI) { mV_h[p] |= aV_h[p]; mV_v[p] |= aV_v[p]; }
            else { mV_h[p] = aV_h[p]; mV_v[p] = aV_v[p]; }
        }
    }

#ifdef SLGC_STAMPS
    asm volatile("" : "+v"(mB_h[0]), "+v"(mB_v[0]), "+v"(mV_h[0]), "+v"(mV_v[0]));
#endif
    stamp<FUSE, BLOCK>(a, 2);
    // accumulators hold the L code bits in bits 0 .. L-1 of each half
    const uint32_t full = ((1u << L) - 1u) * 0x00010001u;
    uint32_t oh[NP], ov[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const uint32_t gh = mB_h[p], gv = mB_v[p];
        uint32_t okh, okv;  // 0xffff per half where every code was classified
        if constexpr (MULTI) {
            okh = as_u(as_ss(pk_add(mV_h[p] ^ full, 0x7fff7fffu)) >> (short)15);   // half != 0 -> bit15 set -> 0xffff = NOT ok
            okv = as_u(as_ss(pk_add(mV_v[p] ^ full, 0x7fff7fffu)) >> (short)15);
            okh = ~okh; okv = ~okv;
        } else {
            okh = as_u(as_ss(mV_h[p]) >> (short)15);
            okv = as_u(as_ss(mV_v[p]) >> (short)15);
        }
        oh[p] = gray_to_binary_2x16(gh) | ~okh;     // undecodable -> 0xffff = -1 (:225-226)
        ov[p] = gray_to_binary_2x16(gv) | ~okv;
    }
    // pairs back to pixel order: E = [p0, p2], O = [p1, p3]  ->  dwords [p0, p1], [p2, p3]
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void *)(a.h + (size_t)scan * a.npix), 0, a.npix * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(a.v + (size_t)scan * a.npix), 0, a.npix * 2u, 0x00020000);
    uint32_t wh_[2 * NW], wv_[2 * NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        wh_[2 * q] = __builtin_amdgcn_perm(oh[2 * q + 1], oh[2 * q], 0x05040100u);
        wh_[2 * q + 1] = __builtin_amdgcn_perm(oh[2 * q + 1], oh[2 * q], 0x07060302u);
        wv_[2 * q] = __builtin_amdgcn_perm(ov[2 * q + 1], ov[2 * q], 0x05040100u);
        wv_[2 * q + 1] = __builtin_amdgcn_perm(ov[2 * q + 1], ov[2 * q], 0x07060302u);
    }
    if constexpr (NW == 1) {
        if (FUSE != 0 && (a.f.nt_store & 4)) {          // fused scan, caller passed no map buffers: XYZ is the only product (wave-uniform branch)
        } else if (a.f.nt_store & 2) {   // fused scan only: the maps are a product there, nothing re-reads them (the two-kernel path keeps them cacheable for K3)
            __builtin_amdgcn_raw_buffer_store_b64(v2u{wh_[0], wh_[1]}, rh, off * 2u, 0, 2);
            __builtin_amdgcn_raw_buffer_store_b64(v2u{wv_[0], wv_[1]}, rv, off * 2u, 0, 2);
        } else {
            __builtin_amdgcn_raw_buffer_store_b64(v2u{wh_[0], wh_[1]}, rh, off * 2u, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(v2u{wv_[0], wv_[1]}, rv, off * 2u, 0, 0);
        }
    } else {
#pragma unroll
        for (int q = 0; q < NW; q += 2) {
            __builtin_amdgcn_raw_buffer_store_b128(v4u{wh_[2 * q], wh_[2 * q + 1], wh_[2 * q + 2], wh_[2 * q + 3]}, rh, off * 2u + 8u * q, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(v4u{wv_[2 * q], wv_[2 * q + 1], wv_[2 * q + 2], wv_[2 * q + 3]}, rv, off * 2u + 8u * q, 0, 0);
        }
    }
    if constexpr (FUSE != 0) {
        set_prio(a.f.prio_tail);
        // K3 appended: the maps never leave registers before they are triangulated (triangulate.py:56-61, 86-95 in the
        // cancelled algebraic form; rays from the per-calibration tables).  Same LDS exchange as k_triangulate_maps_lds:
        // indices -> pixel-per-lane gathers -> per-lane triangulation -> wave-contiguous XYZ stores.
        // FUSE >= 2: the exchange is WAVE-local -- a wave's 256 pixels (64 lanes x 4) are self-contained, every LDS hand-over
        // stays inside the wave (LDS operations of one wave execute in order), so the tail has no workgroup barrier at all and
        // the two waves of a workgroup drift apart freely.  FUSE == 1: the same exchange across the whole workgroup (A/B).
        // FUSE == 3 (default): flat triangles are compacted over the wave before they are redone (below); FUSE == 2: redone lane by lane (A/B).
        static_assert(FUSE == 0 || NW == 1, "fused tail is written for 4 pixels per lane");
        constexpr bool WAVE = FUSE >= 2;
        constexpr bool GLIST = FUSE == 3 && D::product;
        constexpr int SPAN = WAVE ? 64 : BLOCK;                     // lanes that exchange with each other
        const int tid = threadIdx.x;
        const int t = WAVE ? (tid & 63) : tid;                      // index inside the exchanging group
        const int grp = WAVE ? (tid >> 6) : 0;
        // a group's block: [SPAN x uint4 indices | 3 * SPAN x float4 rays / XYZ]; with the wave-local tail it is the wave's own 4 KB
        // block, which the wave's parked frames (dead by now) aliased
        unsigned char *blk = s_raw + grp * (SPAN * 64);
        uint4 *s_idx = reinterpret_cast<uint4 *>(blk);
        float4 *s_buf = reinterpret_cast<float4 *>(blk + SPAN * 16);
        auto sync = [] {
            if constexpr (WAVE) wave_lds_sync();
            else __syncthreads();
        };
        const bool live = off < a.npix;
        uint32_t idx[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        float4 c01 = make_float4(0.f, 0.f, 0.f, 0.f), c23 = c01;
        if (live) {
            if (a.f.cn.nodes) {                        // the four nodes around the lane's pixels: requested now, land during the exchange
                const float2 *e = cam_node_ptr(a.f.cn, off >> 2);
                const cam_v4f n01 = *reinterpret_cast<const cam_v4f *>(e), n23 = *reinterpret_cast<const cam_v4f *>(e + 2);
                c01 = make_float4(n01.x, n01.y, n01.z, n01.w);
                c23 = make_float4(n23.x, n23.y, n23.z, n23.w);
            } else {
                const float4 *cl = reinterpret_cast<const float4 *>(a.f.cam_lut + (D::cam_rays_cached ? (off & 255u) : off));
                c01 = cl[0];
                c23 = cl[1];
            }
            const uint32_t hw2[2] = {wh_[0], wh_[1]}, vw2[2] = {wv_[0], wv_[1]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hv = (int)(short)(hw2[j >> 1] >> (16 * (j & 1))), vv = (int)(short)(vw2[j >> 1] >> (16 * (j & 1)));
                if (!(hv == -1 || vv == -1)) idx[j] = proj_lut_index(min(a.f.proj_w - 1, hv), min(a.f.proj_h - 1, vv), a.f.tiles_x, a.f.wide);
            }
        }
        s_idx[t] = make_uint4(idx[0], idx[1], idx[2], idx[3]);
        sync();
        float *s_th = reinterpret_cast<float *>(s_buf);
        const uint32_t *s_idx1 = reinterpret_cast<const uint32_t *>(s_idx);
        // four independent gathers in flight per lane: the loads are unconditional (an undecodable pixel reads entry 0, its ray is
        // never used), so no branch separates them and none waits for the one before
        if constexpr (D::branchy_gather) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const uint32_t i = s_idx1[it * SPAN + t];
                s_th[it * SPAN + t] = (i != 0xffffffffu) ? a.f.proj_th[i] : 0.5f;
            }
        } else {
            float gr[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const uint32_t i = s_idx1[it * SPAN + t];
                gr[it] = D::no_gather ? 0.5f : a.f.proj_th[i != 0xffffffffu ? i : 0u];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) s_th[it * SPAN + t] = gr[it];
        }
        sync();
        const float4 t4 = s_buf[t];
        sync();
        const float pth[4] = {t4.x, t4.y, t4.z, t4.w};
        float out[12];
        const uint32_t valid = (idx[0] != 0xffffffffu ? 1u : 0u) | (idx[1] != 0xffffffffu ? 2u : 0u) | (idx[2] != 0xffffffffu ? 4u : 0u) |
                               (idx[3] != 0xffffffffu ? 8u : 0u);
        float fx[4] = {c01.x, c01.z, c23.x, c23.z}, fy[4] = {c01.y, c01.w, c23.y, c23.w};
        if (a.f.cn.nodes) {
            cam_rays_from_nodes(cam_v4f{c01.x, c01.y, c01.z, c01.w}, cam_v4f{c23.x, c23.y, c23.z, c23.w}, fx, fy);
            if (live) cam_rays_exact_where_tiny(fx, fy, a.f.cam_lut + off);
        }
        uint32_t ill = 0;
        if constexpr (GLIST) ill = triangulate4_flag(fx, fy, pth, valid, a.f.kf, out);      // pth holds the gathered tan(beta / 2)
        else triangulate4<!D::unguarded>(fx, fy, pth, valid, a.f.kf, a.f.T, a.f.t_len, out, a.f.cam_lut + off, a.f.proj_lut, idx);
        stage_xyz12(s_buf + 3 * t, out);
        if constexpr (GLIST) {
            // Flat triangles (tri_math.h: error amplification > kGuardAmp) are redone on the reference's float32 intermediates in float64.  A lane-level
            // loop makes the whole wave walk that path once per flagged POSITION (up to 4 times, each for a handful of lanes -- and scattered
            // mis-decodes put a flagged pixel into nearly every wave).  Here the wave compacts its flagged pixels into a byte list (ballot + mbcnt
            // ranks) and redoes 64 of them per pass, lane r taking list entry r: its rays come from the exact tables (camera: per-pixel table,
            // projector: the index still in the wave's index block), its XYZ replaces the fast one in the staging block before the stores.
            if (__builtin_amdgcn_ballot_w64(ill != 0u)) {                       // wave-uniform
                unsigned char *s_list = s_raw + (BLOCK / 64) * kWaveLdsBytes + grp * kWaveListBytes;
                uint32_t n = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool f = (ill >> j) & 1u;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(f);
                    if (f) s_list[n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (unsigned char)(4 * t + j);
                    n += (uint32_t)__builtin_popcountll(m);
                }
                wave_lds_sync();
                float *s_xyz = reinterpret_cast<float *>(s_buf);
                const float2 *cam_wave = a.f.cam_lut + (off - 4u * (uint32_t)t);         // the wave's first pixel in the exact per-pixel table
                // up to 4 passes (256 pixels): the exact rays of ALL of a lane's entries are requested before the first pass computes -- one memory
                // round trip per wave, not one per pass (a wave inside a flat region has all 256 pixels on its list)
                const uint32_t npass = (n + 63u) >> 6;                                    // wave-uniform
                uint32_t pq[4];
                float2 crq[4], prq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if ((uint32_t)q < npass) {
                        pq[q] = s_list[min(64u * q + (uint32_t)t, n - 1u)];
                        crq[q] = cam_wave[pq[q]];
                        prq[q] = a.f.proj_lut[s_idx1[pq[q]]];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if ((uint32_t)q < npass && 64u * q + (uint32_t)t < n) {
                        const Xyzf r = law_of_sines_mirror(Ray2{crq[q].x, crq[q].y}, Ray2{prq[q].x, prq[q].y}, a.f.T, a.f.t_len);
                        s_xyz[3 * pq[q]] = r.x;
                        s_xyz[3 * pq[q] + 1] = r.y;
                        s_xyz[3 * pq[q] + 2] = r.z;
                    }
                }
            }
        }
        sync();
        const uint32_t first = bid * BLOCK + grp * SPAN, ngroups = a.npix / 4;                  // in 4-pixel groups
        const uint32_t nvec = first < ngroups ? ((ngroups - first < (uint32_t)SPAN ? ngroups - first : (uint32_t)SPAN) * 3u) : 0u;
        float4 *dst = reinterpret_cast<float4 *>(a.f.xyz + (size_t)scan * a.npix * 3) + (size_t)first * 3;
#pragma unroll
        for (int it = 0; it < 3; ++it)
            if ((uint32_t)(it * SPAN + t) < nvec && (!D::stores_never || s_buf[it * SPAN + t].x == 12345.678f)) {
                if (a.f.nt_store & 1) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const float4 q = s_buf[it * SPAN + t];
                    __builtin_nontemporal_store(v4f{q.x, q.y, q.z, q.w}, reinterpret_cast<v4f *>(dst) + it * SPAN + t);
                } else {
                    dst[it * SPAN + t] = s_buf[it * SPAN + t];
                }
            }
        stamp<FUSE, BLOCK>(a, 3);
#ifdef SLGC_STAMPS
        __builtin_amdgcn_s_waitcnt(0);                   // (stamp 4 = the stores are acknowledged)
#endif
        stamp<FUSE, BLOCK>(a, 4);
    }
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
    SLGC_LAUNCH(ctx, (k_decode_fast<PX, BLOCK>), dim3((unsigned)blocks), dim3(BLOCK), a);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Diagnostic: exhaustive equivalence of the integer-threshold folding (pixel_thresholds) with the literal fp64 predicates
// of decode_codes.py:172-182 over the whole uint8 domain.  One thread per (black, white, L_max, L_min) tuple; each checks
//   n + eps < L_d  <=>  n < tnd,    n > L_g + eps  <=>  n >= tg   for n = 0..255,   and   L_d > L_g + eps  <=>  cA reachable.
// (The remaining predicates n > i + eps, n + eps < i do not depend on the tuple: n - i >= e + 1, i - n >= e + 1 on integers.)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_selftest_thresholds(int e, int black0, unsigned long long *bad, int skew)
{
    const int lmin = threadIdx.x, lmax = blockIdx.x, white = blockIdx.y, black = black0 + (int)blockIdx.z;
    int tt, cA;
    pixel_thresholds(black, white, lmax, lmin, e, tt, cA);
    const int tnd = tt & 0xffff, tg = (int)((uint32_t)tt >> 16);
    const double w = (double)white, b = (double)black, eps = (double)(e + skew);      // skew != 0: negative control
    const double b_inv = w / (w + b);
    const double ld = ((double)lmax - (double)lmin) * b_inv;
    const double lg = (2.0 * ((double)lmax - ld)) * b_inv;
    unsigned n_bad = ((ld > (lg + eps)) != (cA != kUnreachable)) ? 1u : 0u;
    for (int n = 0; n < 256; ++n) {
        const double x = (double)n;
        n_bad += (((x + eps) < ld) != (n < tnd)) ? 1u : 0u;
        n_bad += ((x > (lg + eps)) != (n >= tg)) ? 1u : 0u;
    }
    if (n_bad) atomicAdd(bad, (unsigned long long)n_bad);
}

// Diagnostic: the packed-16 rule evaluation (classify_pk) against the scalar rule table for EVERY threshold triple
// (tnd, tg in 0..256; cA = 1..256 or unreachable) and EVERY (n, i) in 0..255^2 -- 1.1e12 classifications.  The register's other half
// carries a different triple and different pixel values, so leaks between the halves would show.
__global__ void __launch_bounds__(256) k_selftest_classify(unsigned long long *bad, int skew)
{
    const int tnd = blockIdx.x, tg = blockIdx.y, ci = blockIdx.z;                  // 257 x 257 x 257
    const int cA = ci < 256 ? ci + 1 : kUnreachable;
    const int tnd2 = 256 - tnd, tg2 = (tg * 7 + 3) % 257, cA2 = ci % 3 == 0 ? kUnreachable : (ci * 5) % 256 + 1;
    uint32_t KA, KB, C1, C2;
    pack_pair_consts(tnd | (tg << 16), cA, tnd2 | (tg2 << 16), cA2, KA, KB, C1, C2);
    const int n = threadIdx.x;
    unsigned n_bad = 0;
    for (int i = 0; i < 256; ++i) {
        const int n2 = 255 - n, i2 = (i * 37 + 11) & 255;
        // through the accumulating form the kernels use, at a step t that varies (deposit position t of each half), single-run and multi-run
        const uint32_t t = (uint32_t)((n + i) % 15), sh = 15u - t, mk = 0x00010001u << t;
        const uint32_t junk = 0x5a5a5a5au & ~mk;                      // bits of other steps already in the accumulator must survive
        uint32_t accB = junk, accV = 0xffffffffu, accVm = junk;
        const uint32_t Np = (uint32_t)n | ((uint32_t)n2 << 16), Ip = (uint32_t)i | ((uint32_t)i2 << 16);
        classify_pk<false>(Np, Ip, KA, KB, C1, C2, accB, accV, sh, mk);
        uint32_t accB2 = junk;
        classify_pk<true>(Np, Ip, KA, KB, C1, C2, accB2, accVm, sh, mk);
        auto expect = [](int n_, int i_, int tnd_, int tg_, int cA_, bool &bit, bool &valid) {
            const bool r1 = (n_ - i_) >= cA_, r2 = (i_ - n_) >= cA_;
            const bool r3 = (n_ < tnd_) & (i_ >= tg_), r4 = (n_ >= tg_) & (i_ < tnd_);
            bit = r4 | (r1 & !r3);
            valid = r1 | r2 | r3 | r4;
        };
        bool b_lo, v_lo, b_hi, v_hi;
        expect(n, i, tnd, tg + skew, cA, b_lo, v_lo);          // skew != 0: negative control
        expect(n2, i2, tnd2, tg2, cA2, b_hi, v_hi);
        // an unclassified pair leaves the code bit unspecified (the pixel becomes -1): compare it only where valid
        n_bad += (((accV >> 15) & 1u) != (unsigned)v_lo) + (((accV >> 31) & 1u) != (unsigned)v_hi);
        n_bad += (((accVm >> t) & 1u) != (unsigned)v_lo) + (((accVm >> (16 + t)) & 1u) != (unsigned)v_hi);
        n_bad += (v_lo && (((accB >> t) & 1u) != (unsigned)b_lo)) + (v_hi && (((accB >> (16 + t)) & 1u) != (unsigned)b_hi));
        n_bad += ((accB & ~mk) != junk) + ((accVm & ~mk) != junk) + (accB2 != accB);
    }
    if (n_bad) atomicAdd(bad, (unsigned long long)n_bad);
}

int launch_selftest_classify(slgc_ctx *ctx, unsigned long long *d_bad, int skew)
{
    hipLaunchKernelGGL(k_selftest_classify, dim3(257, 257, 257), dim3(256), 0, ctx->stream, d_bad, skew);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_selftest_thresholds(slgc_ctx *ctx, int e, int black0, int n_black, unsigned long long *d_bad, int skew)
{
    hipLaunchKernelGGL(k_selftest_thresholds, dim3(256, 256, (unsigned)n_black), dim3(256), 0, ctx->stream, e, black0, d_bad, skew);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

bool decode_fast_eligible(double eps, int *e_out)
{
    if (!(eps >= 0.0 && eps <= 255.0)) return false;
    const int e = (int)eps;
    if ((double)e != eps) return false;
    *e_out = e;
    return true;
}

// The specialised kernels bake the frame indices in: use one only when the run-time geometry (computed on the host exactly like the
// reference does) agrees with the compile-time table.
template <int NS>
static bool spec_matches(const DecodeGeom &g)
{
    if (g.N != NS || g.L != FrameSpec<NS>::L) return false;
    for (int k = 0; k < 6; ++k)
        if (g.hid[k] != FrameSpec<NS>::thr(k) || g.vid[k] != FrameSpec<NS>::thr(6 + k)) return false;
    return true;
}

static int spec_frames(const slgc_ctx *ctx, const DecodeGeom &g)
{
    if (!ctx->tune_park) return 0;
    if (spec_matches<44>(g)) return 44;
    if (spec_matches<46>(g)) return 46;
    if (spec_matches<42>(g)) return 42;
    if (spec_matches<50>(g)) return 50;      // 4K projectors: 4 * ceil(log2(3840)) + 2 frames (generate_codes.py:22-25,53), L = 12 code bits
    if (spec_matches<54>(g)) return 54;      // 8K: L = 13
    return 0;
}

template <int PX, int BLOCK, int NT>
static int launch_pk_t(slgc_ctx *ctx, const PkArgs &a, int abl = 0)
{
    const uint32_t groups = a.npix / PX;
    if (groups == 0) return SLGC_OK;
    const unsigned blocks = (groups + BLOCK - 1) / BLOCK;
    const int ns = (abl == 0 && PX == 4 && BLOCK == 128 && NT == 1) ? spec_frames(ctx, a.g) : 0;
    ctx->last_ns = ns;
#ifdef SLGC_DIAG      // timing-only ablation builds (wrong results on purpose): only in lib/libslgc_diag.so (make diag)
    if (abl == 1)
        SLGC_LAUNCH(ctx, (k_decode_pk<PX, BLOCK, NT, false, 1>), dim3(blocks), dim3(BLOCK), a);
    else if (abl == 2)
        SLGC_LAUNCH(ctx, (k_decode_pk<PX, BLOCK, NT, false, 2>), dim3(blocks), dim3(BLOCK), a);
    else if (abl == 3) {
        // (the NS = 44 specialisation with the integer stand-in, when the launch would have taken it: the fair partner of the product kernel)
        if (PX == 4 && BLOCK == 128 && NT == 1 && spec_frames(ctx, a.g) == 44) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 3, 0, 44>), dim3(blocks), dim3(128), a);
        else SLGC_LAUNCH(ctx, (k_decode_pk<PX, BLOCK, NT, false, 3>), dim3(blocks), dim3(BLOCK), a);
    }
    else
#else
    if (abl != 0) return slgc_fail(ctx, SLGC_EINVAL, "ablation variants exist only in the diagnostic build (make -C 3dscanner-graycode_amd diag)");
#endif
    {
    if constexpr (PX == 4 && BLOCK == 128 && NT == 1) {
#define SLGC_SPEC(NSV)                                                                                             \
        if (ns == NSV) {                                                                                           \
            if (a.g.n_runs > 1) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, true, 0, 0, NSV>), dim3(blocks), dim3(128), a);  \
            else SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 0, 0, NSV>), dim3(blocks), dim3(128), a);         \
        }
        SLGC_SPEC(44) SLGC_SPEC(46) SLGC_SPEC(42) SLGC_SPEC(50) SLGC_SPEC(54)
#undef SLGC_SPEC
    }
    if (ns == 0) {
        if (a.g.n_runs > 1)
            SLGC_LAUNCH(ctx, (k_decode_pk<PX, BLOCK, NT, true>), dim3(blocks), dim3(BLOCK), a);
        else
            SLGC_LAUNCH(ctx, (k_decode_pk<PX, BLOCK, NT, false>), dim3(blocks), dim3(BLOCK), a);
    }
    }
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

// Fused decode + triangulate (K1a-pk + K3 tail), 4 px/lane, 128-thread workgroups, NT loads.  Preconditions are checked by the caller.
int launch_scan_fused(slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, size_t npix4, int e, int16_t *d_h,
                      int16_t *d_v, const void *cam_lut, const void *proj_lut, float *d_xyz, int proj_w, int proj_h, int n_batch, size_t batch_stride,
                      int bgr_bits)
{
    PkArgs b{};
    for (int r = 0; r < g.n_runs; ++r) b.run[r] = (const uint8_t *)runs.p[r];
    b.plane_stride = (uint32_t)plane_stride;
    b.npix = (uint32_t)npix4;
    b.run_bytes = (uint32_t)((uint64_t)(g.N - 1) * plane_stride + (bgr_bits ? 3 : 1) * npix4);
    if (ctx->stack_tile_active) {              // tile-interleaved stack (the caller has checked: not BGR, plane_stride = 2^k, whole launch inside 32 bits)
        b.tile_log2 = (uint32_t)ctx->stack_tile_active;
        b.tile_bytes = (uint32_t)g.N << b.tile_log2;
        b.run_bytes = (uint32_t)(((npix4 + plane_stride - 1) >> b.tile_log2) * b.tile_bytes);
    }
    if (bgr_bits) {              // OpenCV's fixed-point BGR2GRAY coefficients (ingest.hip has the same two sets), split into bytes for v_dot4_u32_u8
        const uint32_t ry = bgr_bits == 14 ? 4899 : 9798, gy = bgr_bits == 14 ? 9617 : 19235, by = bgr_bits == 14 ? 1868 : 3735;
        b.lum.a_hi = (by >> 8) | ((gy >> 8) << 8) | ((ry >> 8) << 16);
        b.lum.a_lo = (by & 255u) | ((gy & 255u) << 8) | ((ry & 255u) << 16);
        b.lum.z_hi = b.lum.a_hi << 8;
        b.lum.z_lo = b.lum.a_lo << 8;
        b.lum.rnd = 1u << (bgr_bits - 1);
        b.lum.shift = (uint32_t)bgr_bits;
    }
    b.h = d_h; b.v = d_v; b.g = g; b.e = e;
    b.f.cam_lut = (const float2 *)cam_lut; b.f.proj_lut = (const float2 *)proj_lut; b.f.proj_th = (const float *)ctx->lut_proj_th; b.f.xyz = d_xyz;
    const int lut_w = ctx->lut_cam_W;
    b.f.cn = SLGC_CAM_NODES_FOR(ctx, lut_w, cam_lut == ctx->lut_cam && npix4 / 4 < (1u << 24));
    b.f.proj_w = proj_w; b.f.proj_h = proj_h; b.f.tiles_x = proj_tiles_x(ctx, proj_w); b.f.wide = ctx->tune_proj_tile;
    // store policy of the products: XYZ always non-temporal; the maps non-temporal on small images (1920x1080: 23.7 vs 24.1 us; 4096x750: no
    // difference), cacheable from 4 Mpixels up (4096x1500: 64.0 vs 65.4 us; 4096x3000: 119.6 vs 124.8 us physical scene, 122.8 vs 128.1 us S-scene, 127.9 vs 133.6 us at 46 frames -- interleaved A/B)
    const int nt_policy = ctx->tune_fuse_nt >= 0 ? (ctx->tune_fuse_nt & 3) : (npix4 * (size_t)(n_batch > 1 ? n_batch : 1) >= (4u << 20) ? 1 : 3);
    b.f.nt_store = nt_policy | ((d_h == nullptr || d_v == nullptr) ? 4 : 0);
    b.f.wave_tail = ctx->tune_fuse_tail;
    {
        // Issue priority per phase (slgc_tune "prio" = head * 100 + body * 10 + tail, each 0..3; -1 = by launch shape, the default).  A launch
        // that fills most of the chip's 8 192 wave slots in ONE round has every wave in the same phase at the same time: with the head (threshold
        // frames) first, the bit loop second and the tail last the memory pipeline stays full while the first waves finish -- 1920x1080x44
        // 23.3-23.8 -> 22.4-22.9 us on five boxes, 4096x375 (6 000 waves) 18.6 -> 18.2; below half of the slots (1280x720: 3 600 waves) it is a
        // wash, and a launch of more than one round needs its tails to finish to make room: 210 costs 4096x750 +6 %, 4096x3000 +8 %.  Raising only
        // the head (200 / 300) gave -4 to -8 % at 4096x3000 on two slow boxes (127-134 us) and +0.5 to +1.7 % on eleven fast ones (117-119): left at 0.
        int pr = ctx->tune_prio;
        if (pr < 0) {
            const uint64_t waves = (uint64_t)((b.npix / 4 + 127) / 128) * 2u * (uint64_t)(n_batch > 1 ? n_batch : 1);
            pr = (!bgr_bits && waves > 4096 && waves <= 8192) ? 210 : 0;
        }
        b.f.prio_head = (pr / 100) % 10; b.f.prio_body = (pr / 10) % 10; b.f.prio_tail = pr % 10;
    }
    b.f.kf = make_tri_f32(ctx->calib.T, ctx->calib.t_len);
    memcpy(b.f.T, ctx->calib.T, sizeof b.f.T);
    b.f.t_len = ctx->calib.t_len;
    const uint32_t groups = b.npix / 4;
    if (groups == 0) return SLGC_OK;
    unsigned blocks = (groups + 127) / 128;
    b.f.xcd_chunk = (ctx->tune_fuse_xcd == 1 && n_batch <= 1) ? xcd_chunk_for(ctx, blocks) : 0u;
    b.f.xcd_run = (ctx->tune_fuse_xcd >= 2 && n_batch <= 1 && blocks >= 64) ? (uint32_t)ctx->tune_fuse_xcd : 0u;       // fuse_xcd = n >= 2: fine map, n tiles per XCD
    if (n_batch > 1) {          // the caller has checked: every scan is a whole number of workgroups, blocks * n_batch * blocks < 2^32
        b.f.batch_bps = blocks;
        b.f.batch_magic = (uint32_t)((0x100000000ull + blocks - 1) / blocks);
        b.f.batch_stride = batch_stride;
        blocks *= (unsigned)n_batch;
    }
#ifdef SLGC_STAMPS
    {
        void *st = nullptr;
        const size_t nw = (size_t)blocks * 2;
        if (slgc_ws(ctx, 12, nw * 5 * 8 + 64, &st)) return SLGC_ENOMEM;      // (a slot of its own: 11 holds the grey stack of the BGR fall-back)
        b.f.stamps = (unsigned long long *)st;
        ctx->stamp_waves = nw;
    }
#endif
    const bool wave = b.f.wave_tail != 0;
#ifdef SLGC_DIAG      // timing-only ablation builds (wrong results on purpose): only in lib/libslgc_diag.so (make diag)
    const int fabl = ctx->tune_fuse_abl;
    if (fabl == 3 && wave && spec_frames(ctx, g) == 44) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 3, 3, 44>), dim3(blocks), dim3(128), b);
    else if (fabl == 5) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 5, 2>), dim3(blocks), dim3(128), b);
    else if (fabl == 6) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 6, 2>), dim3(blocks), dim3(128), b);
    else if (fabl == 7) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 7, 2>), dim3(blocks), dim3(128), b);
    else if (fabl == 8) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 8, 2>), dim3(blocks), dim3(128), b);
    else if (fabl == 9) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 9, 2>), dim3(blocks), dim3(128), b);
    else
#endif
    {
        const int ns = wave ? spec_frames(ctx, g) : 0;
        ctx->last_ns = ns;
        ctx->last_nodes = b.f.cn.nodes ? 1 : 0;
        ctx->last_guard = 1;
        ctx->last_ragged = 0;
        ctx->last_tri_ragged = 0;
        if (bgr_bits) {          // the caller has checked scan_bgr_eligible: a specialised frame count, wave-local tail
#define SLGC_BGR(NSV)                                                                                              \
            if (ns == NSV) {                                                                                       \
                if (g.n_runs > 1) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, true, 0, 3, NSV, 1>), dim3(blocks), dim3(128), b);    \
                else SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 0, 3, NSV, 1>), dim3(blocks), dim3(128), b);                \
            }
            SLGC_BGR(44) SLGC_BGR(46) SLGC_BGR(42) SLGC_BGR(50) SLGC_BGR(54)
#undef SLGC_BGR
            if (ns == 0) return slgc_fail(ctx, SLGC_EINVAL, "internal: BGR scan launched without a specialised frame count");
            HIP_TRY(ctx, hipGetLastError());
            return SLGC_OK;
        }
        const bool glist = ctx->tune_guard_list != 0;       // flat triangles compacted over the wave (default) or redone lane by lane (A/B)
#define SLGC_SPEC(NSV)                                                                                             \
        if (ns == NSV) {                                                                                           \
            if (g.n_runs > 1) {                                                                                    \
                if (glist) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, true, 0, 3, NSV>), dim3(blocks), dim3(128), b);       \
                else SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, true, 0, 2, NSV>), dim3(blocks), dim3(128), b);             \
            } else {                                                                                               \
                if (glist) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 0, 3, NSV>), dim3(blocks), dim3(128), b);      \
                else SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 0, 2, NSV>), dim3(blocks), dim3(128), b);            \
            }                                                                                                      \
        }
        SLGC_SPEC(44) SLGC_SPEC(46) SLGC_SPEC(42) SLGC_SPEC(50) SLGC_SPEC(54)
#undef SLGC_SPEC
        if (ns == 0) {
            if (g.n_runs > 1) {
                if (wave && glist) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, true, 0, 3>), dim3(blocks), dim3(128), b);
                else if (wave) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, true, 0, 2>), dim3(blocks), dim3(128), b);
                else SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, true, 0, 1>), dim3(blocks), dim3(128), b);
            } else {
                if (wave && glist) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 0, 3>), dim3(blocks), dim3(128), b);
                else if (wave) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 0, 2>), dim3(blocks), dim3(128), b);
                else SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 0, 1>), dim3(blocks), dim3(128), b);
            }
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

#ifdef SLGC_STAMPS
// stamp build only: the 5 stamps per wave of the last fused launch (100 MHz ticks of s_memrealtime), wave-major
extern "C" int slgc_diag_stamps(slgc_ctx *ctx, unsigned long long *host, size_t cap_waves, size_t *n_waves)
{
    if (!ctx || !host || !n_waves) return SLGC_EINVAL;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t n = ctx->stamp_waves < cap_waves ? ctx->stamp_waves : cap_waves;
    if (n) HIP_TRY(ctx, hipMemcpy(host, ctx->ws[12], n * 5 * 8, hipMemcpyDeviceToHost));      // (slot 12: launch_scan_fused)
    *n_waves = ctx->stamp_waves;
    return SLGC_OK;
}
#endif

bool scan_fused_eligible(const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, size_t npix, const int16_t *d_h, const int16_t *d_v,
                         const float *d_xyz)
{
    uintptr_t align_or = (uintptr_t)plane_stride | ((uintptr_t)d_h >> 1) | ((uintptr_t)d_v >> 1);
    for (int r = 0; r < g.n_runs; ++r) align_or |= (uintptr_t)runs.p[r];
    return align_or % 4 == 0 && (uintptr_t)d_xyz % 16 == 0 && npix >= 4 && (uint64_t)g.N * plane_stride + npix < 0xfffffff0ull &&
           npix < 0x7fffffffull;
}

// The decode kernel alone on BGR planes (slgc_decode_bgr_dev): the caller has checked scan_bgr_eligible (d_xyz = an aligned dummy).
int launch_decode_bgr(slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, size_t npix4, int e, int16_t *d_h, int16_t *d_v, int bgr_bits)
{
    PkArgs b{};
    for (int r = 0; r < g.n_runs; ++r) b.run[r] = (const uint8_t *)runs.p[r];
    b.plane_stride = (uint32_t)plane_stride;
    b.npix = (uint32_t)npix4;
    b.run_bytes = (uint32_t)((uint64_t)(g.N - 1) * plane_stride + 3 * npix4);
    b.h = d_h; b.v = d_v; b.g = g; b.e = e;
    const uint32_t ry = bgr_bits == 14 ? 4899 : 9798, gy = bgr_bits == 14 ? 9617 : 19235, by = bgr_bits == 14 ? 1868 : 3735;
    b.lum.a_hi = (by >> 8) | ((gy >> 8) << 8) | ((ry >> 8) << 16);
    b.lum.a_lo = (by & 255u) | ((gy & 255u) << 8) | ((ry & 255u) << 16);
    b.lum.z_hi = b.lum.a_hi << 8;
    b.lum.z_lo = b.lum.a_lo << 8;
    b.lum.rnd = 1u << (bgr_bits - 1);
    b.lum.shift = (uint32_t)bgr_bits;
    const uint32_t groups = b.npix / 4;
    if (groups == 0) return SLGC_OK;
    const unsigned blocks = (groups + 127) / 128;
    const int ns = spec_frames(ctx, g);
    ctx->last_ns = ns;
    ctx->last_ragged = 0;
#define SLGC_BGR(NSV)                                                                                              \
    if (ns == NSV) {                                                                                               \
        if (g.n_runs > 1) SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, true, 0, 0, NSV, 1>), dim3(blocks), dim3(128), b);    \
        else SLGC_LAUNCH(ctx, (k_decode_pk<4, 128, 1, false, 0, 0, NSV, 1>), dim3(blocks), dim3(128), b);                \
    }
    SLGC_BGR(44) SLGC_BGR(46) SLGC_BGR(42) SLGC_BGR(50) SLGC_BGR(54)
#undef SLGC_BGR
    if (ns == 0) return slgc_fail(ctx, SLGC_EINVAL, "internal: BGR decode launched without a specialised frame count");
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

// The BGR form of the fused scan (slgc_scan_bgr_dev): planes of 3 bytes per pixel, 4-byte aligned, 32-bit offsets, one of the specialised frame counts.
// d_xyz == nullptr: the decode-only BGR kernel (no fused tail: the fuse_tail knob does not apply).  The BGR kernels always compact flat
// triangles over the wave (FUSE = 3): the guard_list A/B knob is not honoured there.
bool scan_bgr_eligible(const slgc_ctx *ctx, const DecodeGeom &g, const RunPtrs &runs, size_t plane_stride, size_t npix, const int16_t *d_h, const int16_t *d_v,
                       const float *d_xyz)
{
    uintptr_t align_or = (uintptr_t)plane_stride | ((uintptr_t)d_h >> 1) | ((uintptr_t)d_v >> 1);
    for (int r = 0; r < g.n_runs; ++r) align_or |= (uintptr_t)runs.p[r];
    return align_or % 4 == 0 && (uintptr_t)d_xyz % 16 == 0 && npix >= 4 && npix % 4 == 0 && (uint64_t)g.N * plane_stride + 3 * (uint64_t)npix + 4096u < 0xfffffff0ull &&      // (+ the lanes of the last, partial workgroup)
           npix < 0x7fffffffull && (ctx->tune_fuse_tail || !d_xyz) && spec_frames(ctx, g) != 0;
}

// variant: 0 = library default; otherwise ABL*10000000 + NT*1000000 + ALG*100000 + PX*1000 + BLOCK, e.g. 104128 = packed-16 kernel,
// 4 pixels per lane, 128-thread workgroups; ALG 0 = lane-mask (v_cmp) kernel, 1 = packed-16 kernel; NT = non-temporal loads.
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
    const int max_px = (align_or % 16 == 0) ? 16 : (align_or % 8 == 0) ? 8 : (align_or % 4 == 0) ? 4 : 1;
    const bool fits32 = (uint64_t)g.N * plane_stride + npix < 0xfffffff0ull && npix < 0x7fffffffull;
    if (variant == 0) {
        if (max_px >= 4 && fits32) variant = 1104128;   // packed-16 kernel, NT loads, 4 px/lane, 128-thread WGs: measured best
        else variant = max_px * 1000 + 256;
    }
    const int abl = variant / 10000000;
#ifndef SLGC_DIAG
    if (abl) return slgc_fail(ctx, SLGC_EINVAL, "variant %d: the timing-only ablation kernels exist only in the diagnostic build (make -C 3dscanner-graycode_amd diag)", variant);
#endif
    variant %= 10000000;
    const int nt = variant / 1000000, alg = (variant / 100000) % 10, px = (variant / 1000) % 100, block = variant % 1000;
    if ((px != 16 && px != 8 && px != 4 && px != 1) || (block != 64 && block != 128 && block != 256) || alg > 1 || nt > 1)
        return slgc_fail(ctx, SLGC_EINVAL, "bad variant %d", variant);
    if (px > max_px) return slgc_fail(ctx, SLGC_EINVAL, "variant %d needs %d-byte alignment", variant, px);
    if (alg >= 1 && (!fits32 || px == 1)) return slgc_fail(ctx, SLGC_EINVAL, "variant %d: band too large for 32-bit offsets or px=1", variant);
    const size_t main_pix = npix / px * px;
    int rc = SLGC_EINVAL;
    if (alg >= 1) {
        PkArgs b{};
        for (int r = 0; r < g.n_runs; ++r) b.run[r] = a.run[r];
        b.plane_stride = (uint32_t)plane_stride;
        b.npix = (uint32_t)main_pix;
        b.run_bytes = (uint32_t)((uint64_t)(g.N - 1) * plane_stride + main_pix);
        if (ctx->stack_tile_active) {          // tile-interleaved stack: the packed kernel on whole 4-pixel groups only (checked by slgc_stack_tile_ok)
            if (main_pix != npix) return slgc_fail(ctx, SLGC_EINVAL, "a tile-interleaved stack needs a band of a multiple of %d pixels", px);
            b.tile_log2 = (uint32_t)ctx->stack_tile_active;
            b.tile_bytes = (uint32_t)g.N << b.tile_log2;
            b.run_bytes = (uint32_t)(((main_pix + plane_stride - 1) >> b.tile_log2) * b.tile_bytes);
        }
        b.h = d_h; b.v = d_v; b.g = g; b.e = e;
        {
            // issue priority (see launch_scan_fused): the decode kernel has no tail to put last, but a wave that still has to ask for its threshold
            // frames goes first here at every size from 4 096 waves up -- 1920x1080 18.5-18.8 -> 18.3 us, one band of 8 ranks (4096x375) 15.4 -> 14.8,
            // 4096x3000 99.1-101.6 -> 97.9-100.1 (three boxes); below that (1280x720) it costs 0.5-1 %
            int pr = ctx->tune_prio;
            if (pr < 0) pr = ((main_pix / 4 + 127) / 128) * 2 > 4096 ? 210 : 0;
            b.f.prio_head = (pr / 100) % 10; b.f.prio_body = (pr / 10) % 10;
        }
#define SLGC_PK(P, B, T) if (alg == 1 && px == P && block == B && nt == T) rc = launch_pk_t<P, B, T>(ctx, b, abl);
        SLGC_PK(4, 64, 0) SLGC_PK(4, 128, 0) SLGC_PK(4, 256, 0) SLGC_PK(8, 64, 0) SLGC_PK(8, 128, 0) SLGC_PK(8, 256, 0)
        SLGC_PK(16, 64, 0) SLGC_PK(16, 128, 0) SLGC_PK(16, 256, 0)
        SLGC_PK(4, 64, 1) SLGC_PK(4, 128, 1) SLGC_PK(4, 256, 1) SLGC_PK(8, 64, 1) SLGC_PK(8, 128, 1) SLGC_PK(8, 256, 1)
        SLGC_PK(16, 64, 1) SLGC_PK(16, 128, 1) SLGC_PK(16, 256, 1)
#undef SLGC_PK
    } else {
        if (nt) return slgc_fail(ctx, SLGC_EINVAL, "variant %d: NT only with the packed kernel", variant);
        if (ctx->stack_tile_active) return slgc_fail(ctx, SLGC_EINVAL, "a tile-interleaved stack needs 4-byte aligned buffers and a band inside 32-bit offsets (the packed kernel)");
        ctx->last_ns = 0;
        ctx->last_ragged = 1;             // the lane-mask kernel: what misaligned bands fall back to
#define SLGC_CASE(P, B) if (px == P && block == B) rc = launch_fast_t<P, B>(ctx, a, main_pix);
        SLGC_CASE(16, 256) SLGC_CASE(16, 128) SLGC_CASE(16, 64) SLGC_CASE(8, 256) SLGC_CASE(8, 128) SLGC_CASE(8, 64)
        SLGC_CASE(4, 256) SLGC_CASE(4, 128) SLGC_CASE(4, 64) SLGC_CASE(1, 256)
#undef SLGC_CASE
    }
    if (rc) return rc == SLGC_EINVAL ? slgc_fail(ctx, SLGC_EINVAL, "variant %d not built", variant) : rc;
    if (main_pix < npix) {  // ragged tail (< px pixels): byte-wide groups
        ctx->last_ragged = 1;
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
