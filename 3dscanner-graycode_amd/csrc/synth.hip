// synth.hip -- synthetic structured-light capture written straight into HBM (bench / test input, no PCIe).
//
// Follows the frame-order contract of scanner/grayCode/generate_codes.py:53-79 (black, white, then column-code
// bit k MSB-first at frame 2+2k, row-code bit k LSB-first at 3+2k, inverses 2L frames later) on a warped scene:
//     xs = ((29*x) >> 5) + tri(y),  ys = ((29*y) >> 5) + tri(x),  tri(t) = |((t >> 5) % 10) - 5|      (mod 2^L)
// an all-integer stand-in for SURVEY.md 8(d)'s 0.9*x + 5*sin(y/50) so the NumPy twin
// (oracle/oracle_np.py: synth_scene_int) is bit-identical.  Ambient 15, gain 180, hash noise in [-noise, noise],
// one shadow rectangle (all frames = ambient) to exercise the validity mask.
#include "slgc_internal.h"

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ int tri(int t) { return abs(((t >> 5) % 10) - 5); }

struct SynthArgs {
    uint8_t *stack;
    size_t plane_stride;
    int N, L, H, W, row0, rows;
    uint32_t seed;
    int noise;
    int sy0, sy1, sx0, sx1;  // shadow rectangle (empty when sy0 >= sy1)
    int gain_lo, gain_hi;    // k_synth_render: the two surface gains of the 16-pixel checker
};

// one thread = 4 consecutive pixels of one frame (dword store); grid.y = frame
__global__ void __launch_bounds__(256) k_synth(const SynthArgs a)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;  // dword index within the band
    const size_t nq = ((size_t)a.rows * a.W + 3) / 4;
    if (q >= nq) return;
    const int f = blockIdx.y;
    uint32_t word = 0;
    for (int j = 0; j < 4; ++j) {
        const size_t lp = q * 4 + j;  // pixel index within the band
        if (lp >= (size_t)a.rows * a.W) break;
        const int x = (int)(lp % a.W), y = a.row0 + (int)(lp / a.W);
        int val = 15;
        const bool shadow = (y >= a.sy0) & (y < a.sy1) & (x >= a.sx0) & (x < a.sx1);
        if (!shadow) {
            if (f == 1) val = 195;
            else if (f >= 2 && f < 2 + 4 * a.L) {
                const int idx = f - 2, inv = idx >= 2 * a.L, k2 = idx - (inv ? 2 * a.L : 0), k = k2 >> 1;
                const uint32_t msk = (1u << a.L) - 1u;
                int bit;
                if ((k2 & 1) == 0) {
                    const uint32_t xs = (uint32_t)(((29 * x) >> 5) + tri(y)) & msk, g = xs ^ (xs >> 1);
                    bit = (g >> (a.L - 1 - k)) & 1;
                } else {
                    const uint32_t ys = (uint32_t)(((29 * y) >> 5) + tri(x)) & msk, g = ys ^ (ys >> 1);
                    bit = (g >> k) & 1;
                }
                val = 15 + 180 * (inv ? 1 - bit : bit);
            }
        }
        if (a.noise > 0) {
            const uint32_t gp = (uint32_t)((size_t)y * a.W + x);
            const uint32_t r = mix32(gp * 0x9E3779B1u + (uint32_t)f * 0x85EBCA77u + a.seed);
            val += (int)(r % (uint32_t)(2 * a.noise + 1)) - a.noise;
        }
        val = val < 0 ? 0 : (val > 255 ? 255 : val);
        word |= (uint32_t)val << (8 * j);
    }
    uint8_t *dst = a.stack + (size_t)f * a.plane_stride + q * 4;
    const size_t left = (size_t)a.rows * a.W - q * 4;
    if (left >= 4 && ((uintptr_t)dst & 3) == 0) *reinterpret_cast<uint32_t *>(dst) = word;
    else for (size_t j = 0; j < (left < 4 ? left : 4); ++j) dst[j] = (uint8_t)(word >> (8 * j));
}

// ------------------------------------------------------------------------------------------------------------------------------
// Physically consistent capture: one surface seen by the camera AND lit by the projector (what src/4-triangulate.py:50-64 of the
// reference assumes about its inputs).  Pass 1 casts every camera pixel's ray into the scene (tilted back plane + sphere), carries the hit
// point through the stereo pose and the projector's forward lens model to the projector pixel that lights it, and writes that pixel
// (-1 = unlit) plus the true surface point; pass 2 renders the Gray-code frames of those codes.  float64, + - * / sqrt only, in the one
// operation order of the NumPy twin (oracle/oracle_np.py: synth_physical_codes / render_codes) -- bit-identical (-ffp-contract=off).
constexpr double kPlaneP[3] = {0.0, 0.0, 0.56}, kPlaneN[3] = {0.18, -0.10, -1.0};
constexpr double kSphereC[3] = {0.045, 0.015, 0.46}, kSphereR = 0.06;
constexpr int kPhysUndistortIters = 20;

__global__ void __launch_bounds__(256) k_synth_physical_codes(const Calib c, int W, int row0, size_t npix, int pw, int ph, int code_bits, double r2_max,
                                                              int16_t *__restrict__ h, int16_t *__restrict__ v, float *__restrict__ truth)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const double u = (double)(int)(p % (size_t)W), vv = (double)(row0 + (int)(p / (size_t)W));
    const double *k = c.cam_d, *q = c.proj_d, *R = c.R, *T = c.T;
    const double x0 = (u - c.cam_k[2]) / c.cam_k[0], y0 = (vv - c.cam_k[3]) / c.cam_k[1];
    double x = x0, y = y0;
#pragma unroll 1
    for (int it = 0; it < kPhysUndistortIters; ++it) {
        const double r2 = x * x + y * y;
        const double icd = (1.0 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1.0 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        const double dx = ((2.0 * k[2]) * x * y + k[3] * (r2 + (2.0 * x) * x)) + (k[8] * r2 + (k[9] * r2) * r2);
        const double dy = (k[2] * (r2 + (2.0 * y) * y) + (2.0 * k[3]) * x * y) + (k[10] * r2 + (k[11] * r2) * r2);
        x = (x0 - dx) * icd;
        y = (y0 - dy) * icd;
    }
    const double cx = kSphereC[0], cy = kSphereC[1], cz = kSphereC[2];
    const double a = (x * x + y * y) + 1.0;
    const double b = (x * cx + y * cy) + cz;
    const double cc = ((cx * cx + cy * cy) + cz * cz) - kSphereR * kSphereR;
    const double disc = b * b - a * cc;
    const bool on_sphere = disc > 0.0;
    const double ts = (b - sqrt(on_sphere ? disc : 0.0)) / a;
    const double tp = ((kPlaneN[0] * kPlaneP[0] + kPlaneN[1] * kPlaneP[1]) + kPlaneN[2] * kPlaneP[2]) / ((kPlaneN[0] * x + kPlaneN[1] * y) + kPlaneN[2]);
    const double t = on_sphere ? ts : tp;
    const double X = t * x, Y = t * y, Z = t;
    const double Xp = ((R[0] * X + R[1] * Y) + R[2] * Z) + T[0];
    const double Yp = ((R[3] * X + R[4] * Y) + R[5] * Z) + T[1];
    const double Zp = ((R[6] * X + R[7] * Y) + R[8] * Z) + T[2];
    const double Cx = -((R[0] * T[0] + R[3] * T[1]) + R[6] * T[2]);
    const double Cy = -((R[1] * T[0] + R[4] * T[1]) + R[7] * T[2]);
    const double Cz = -((R[2] * T[0] + R[5] * T[1]) + R[8] * T[2]);
    const double ex = Cx - X, ey = Cy - Y, ez = Cz - Z;
    const double fx = X - cx, fy = Y - cy, fz = Z - cz;
    const double ea = (ex * ex + ey * ey) + ez * ez;
    const double eb = (ex * fx + ey * fy) + ez * fz;
    const double ec = ((fx * fx + fy * fy) + fz * fz) - kSphereR * kSphereR;
    const bool shadow = !on_sphere && ((eb * eb - ea * ec) > 0.0) && (eb < 0.0);
    const double xn = Xp / Zp, yn = Yp / Zp;
    const double r2 = xn * xn + yn * yn;
    const double rad = (1.0 + ((q[4] * r2 + q[1]) * r2 + q[0]) * r2) / (1.0 + ((q[7] * r2 + q[6]) * r2 + q[5]) * r2);
    const double xd = (xn * rad + ((2.0 * q[2]) * xn * yn + q[3] * (r2 + (2.0 * xn) * xn))) + (q[8] * r2 + (q[9] * r2) * r2);
    const double yd = (yn * rad + (q[2] * (r2 + (2.0 * yn) * yn) + (2.0 * q[3]) * xn * yn)) + (q[10] * r2 + (q[11] * r2) * r2);
    const double pu = floor((c.proj_k[0] * xd + c.proj_k[2]) + 0.5);
    const double pv = floor((c.proj_k[1] * yd + c.proj_k[3]) + 0.5);
    const double top = (double)(((1 << code_bits) < 32767 ? (1 << code_bits) : 32767) - 1);
    const bool lit = (Zp > 0.0) && (r2 <= r2_max) && (pu >= 0.0) && (pu <= (double)pw - 1.0) && (pv >= 0.0) && (pv <= (double)ph - 1.0) && !shadow &&
                     (t > 0.0) && (pu <= top) && (pv <= top);
    h[p] = lit ? (int16_t)pu : (int16_t)-1;
    v[p] = lit ? (int16_t)pv : (int16_t)-1;
    if (truth) {
        const float nanf_ = __builtin_nanf("");
        truth[3 * p] = lit ? (float)((R[0] * X + R[1] * Y) + R[2] * Z) : nanf_;
        truth[3 * p + 1] = lit ? (float)((R[3] * X + R[4] * Y) + R[5] * Z) : nanf_;
        truth[3 * p + 2] = lit ? (float)((R[6] * X + R[7] * Y) + R[8] * Z) : nanf_;
    }
}

// frames of a capture whose pixel p is lit by projector pixel (h[p], v[p]): one thread = 4 consecutive pixels of one frame; grid.y = frame
__global__ void __launch_bounds__(256) k_synth_render(const SynthArgs a, const int16_t *__restrict__ h, const int16_t *__restrict__ v)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t npix = (size_t)a.rows * a.W, nq = (npix + 3) / 4;
    if (q >= nq) return;
    const int f = blockIdx.y;
    uint32_t word = 0;
    for (int j = 0; j < 4; ++j) {
        const size_t lp = q * 4 + j;
        if (lp >= npix) break;
        const int x = (int)(lp % a.W), y = a.row0 + (int)(lp / a.W);
        const int hv = h[lp], vv = v[lp];
        int val = 15;
        if (hv != -1 && vv != -1) {
            const int gain = ((((x >> 4) ^ (y >> 4)) & 1) == 1) ? a.gain_hi : a.gain_lo;
            if (f == 1) val = 15 + gain;
            else if (f >= 2 && f < 2 + 4 * a.L) {
                const int idx = f - 2, inv = idx >= 2 * a.L, k2 = idx - (inv ? 2 * a.L : 0), k = k2 >> 1;
                const uint32_t msk = (1u << a.L) - 1u;
                int bit;
                if ((k2 & 1) == 0) {
                    const uint32_t xs = (uint32_t)hv & msk, g = xs ^ (xs >> 1);
                    bit = (g >> (a.L - 1 - k)) & 1;
                } else {
                    const uint32_t ys = (uint32_t)vv & msk, g = ys ^ (ys >> 1);
                    bit = (g >> k) & 1;
                }
                val = 15 + gain * (inv ? 1 - bit : bit);
            }
        }
        if (a.noise > 0) {
            const uint32_t gp = (uint32_t)((size_t)y * a.W + x);
            const uint32_t r = mix32(gp * 0x9E3779B1u + (uint32_t)f * 0x85EBCA77u + a.seed);
            val += (int)(r % (uint32_t)(2 * a.noise + 1)) - a.noise;
        }
        val = val < 0 ? 0 : (val > 255 ? 255 : val);
        word |= (uint32_t)val << (8 * j);
    }
    uint8_t *dst = a.stack + (size_t)f * a.plane_stride + q * 4;
    const size_t left = npix - q * 4;
    if (left >= 4 && ((uintptr_t)dst & 3) == 0) *reinterpret_cast<uint32_t *>(dst) = word;
    else for (size_t j = 0; j < (left < 4 ? left : 4); ++j) dst[j] = (uint8_t)(word >> (8 * j));
}

// SURVEY.md 8(d) "S-uniform": every byte of every frame uniform in 0..255 (the worst case for the classification: ~23 % of the pixels decode,
// to arbitrary codes).  The survey draws it from NumPy's PCG64; on the device it is a counter hash -- one mix32 per dword, keyed by (frame,
// dword of the WHOLE image, seed), so a band holds the same bytes as the same rows of the whole image.  W % 4 == 0.  Twin: oracle_np.synth_uniform.
__global__ void __launch_bounds__(256) k_synth_uniform(uint8_t *__restrict__ stack, size_t plane_stride, uint32_t q0, uint32_t nq, uint32_t seed)
{
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= nq) return;
    const uint32_t f = blockIdx.y;
    const uint32_t r = mix32((q0 + q) * 0x9E3779B1u + f * 0x85EBCA77u + seed);
    *reinterpret_cast<uint32_t *>(stack + (size_t)f * plane_stride + (size_t)q * 4) = r;
}

// A BGR capture of a grey stack (the camera frames src/3-capture_decode.py:66 converts): per-pixel channel offsets so that the luma conversion
// has something to do -- B = clip(g + ((7 x + 3 y) mod 11) - 5), G = g, R = clip(g - (((5 x + 11 y) mod 9) - 4)).  One thread = one pixel of one
// frame; grid.y = frame.  Twin: oracle_np.gray_to_bgr_capture.
__global__ void __launch_bounds__(256) k_synth_bgr(const uint8_t *__restrict__ gray, size_t gray_stride, int W, int row0, size_t npix, uint8_t *__restrict__ bgr,
                                                   size_t bgr_stride)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const int f = blockIdx.y;
    const int x = (int)(p % (size_t)W), y = row0 + (int)(p / (size_t)W);
    const int g = gray[(size_t)f * gray_stride + p];
    const int b = g + ((7 * x + 3 * y) % 11) - 5, r = g - (((5 * x + 11 * y) % 9) - 4);
    uint8_t *dst = bgr + (size_t)f * bgr_stride + 3 * p;
    dst[0] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
    dst[1] = (uint8_t)g;
    dst[2] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

// ---- the yardstick of the roofline fractions: a kernel that ONLY moves a scan's bytes -------------------------------------------------
// N frame planes read 4 bytes per lane and plane (the scan kernels' loads), folded with one rotate-xor each so that no load can be dropped;
// written: the two int16 maps 8 bytes per lane each (if asked for) and 48 bytes of "XYZ" per 4 pixels laid out wave-contiguously, 16 bytes
// per lane and store, as the fused kernel's LDS transpose leaves them (if asked for).  bench.py times it beside the kernel it grades.
template <int NP>
__global__ void __launch_bounds__(128) k_move_only(const uint32_t *__restrict__ stack, uint32_t plane_stride4, uint32_t npix4, uint32_t *__restrict__ h,
                                                   uint32_t *__restrict__ v, uint32_t *__restrict__ xyz)
{
    typedef uint32_t v2 __attribute__((ext_vector_type(2)));
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const uint32_t g = blockIdx.x * 128u + threadIdx.x;
    if (g >= npix4) return;
    uint32_t w[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) w[p] = __builtin_nontemporal_load(stack + (size_t)p * plane_stride4 + g);
    uint32_t x = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) x = ((x << 1) | (x >> 31)) ^ w[p];
    if (h) {
        __builtin_nontemporal_store(v2{x, x + 1u}, reinterpret_cast<v2 *>(h) + g);
        __builtin_nontemporal_store(v2{x + 2u, x + 3u}, reinterpret_cast<v2 *>(v) + g);
    }
    if (xyz) {
        const uint32_t wave0 = g & ~63u, lane = g & 63u;
        v4 *dst = reinterpret_cast<v4 *>(xyz) + 3 * (size_t)wave0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_nontemporal_store(v4{x, x + (uint32_t)k, x + 5u, x + 7u}, dst + 64 * k + lane);
    }
}

}  // namespace

// Moves the bytes of one single-run scan of N frames and nothing else (results are meaningless).  d_h / d_v (both or neither) and d_xyz may be
// NULL; npix a multiple of 256 (whole waves: the XYZ of a wave is written as one block), 4-byte aligned planes.
int launch_move_only(slgc_ctx *ctx, const uint8_t *d_stack, size_t plane_stride, int N, size_t npix, int16_t *d_h, int16_t *d_v, float *d_xyz)
{
    const uint32_t npix4 = (uint32_t)(npix / 4), blocks = (npix4 + 127u) / 128u;
    if (!blocks) return SLGC_OK;
#define SLGC_MOVE(NPV)                                                                                                                        \
    if (N == NPV)                                                                                                                             \
        hipLaunchKernelGGL((k_move_only<NPV>), dim3(blocks), dim3(128), 0, ctx->stream, (const uint32_t *)d_stack, (uint32_t)(plane_stride / 4), npix4,  \
                           (uint32_t *)d_h, (uint32_t *)d_v, (uint32_t *)d_xyz);
    SLGC_MOVE(42) SLGC_MOVE(44) SLGC_MOVE(46)
#undef SLGC_MOVE
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_synth_physical(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows, int proj_w, int proj_h,
                          uint32_t seed, int noise, int gain_lo, int gain_hi, double r2_max, int16_t *d_h, int16_t *d_v, float *d_truth)
{
    const size_t npix = (size_t)rows * W;
    if (npix == 0) return SLGC_OK;
    SynthArgs a{};
    a.stack = d_stack; a.plane_stride = plane_stride; a.N = N; a.L = (N - 2) / 4; a.H = H; a.W = W; a.row0 = row0; a.rows = rows;
    a.seed = seed; a.noise = noise; a.gain_lo = gain_lo; a.gain_hi = gain_hi;
    hipLaunchKernelGGL(k_synth_physical_codes, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, ctx->calib, W, row0, npix, proj_w, proj_h,
                       a.L, r2_max, d_h, d_v, d_truth);
    if (d_stack) hipLaunchKernelGGL(k_synth_render, dim3((unsigned)(((npix + 3) / 4 + 255) / 256), N), dim3(256), 0, ctx->stream, a, d_h, d_v);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_synth_bgr(slgc_ctx *ctx, const uint8_t *d_gray, size_t gray_stride, int N, int W, int row0, int rows, uint8_t *d_bgr, size_t bgr_stride)
{
    const size_t npix = (size_t)rows * W;
    if (npix == 0) return SLGC_OK;
    hipLaunchKernelGGL(k_synth_bgr, dim3((unsigned)((npix + 255) / 256), N), dim3(256), 0, ctx->stream, d_gray, gray_stride, W, row0, npix, d_bgr, bgr_stride);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_synth_uniform(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int W, int row0, int rows, uint32_t seed)
{
    const size_t nq = (size_t)rows * W / 4;
    if (nq == 0) return SLGC_OK;
    hipLaunchKernelGGL(k_synth_uniform, dim3((unsigned)((nq + 255) / 256), N), dim3(256), 0, ctx->stream, d_stack, plane_stride, (uint32_t)((size_t)row0 * W / 4),
                       (uint32_t)nq, seed);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_synth(slgc_ctx *ctx, uint8_t *d_stack, size_t plane_stride, int N, int H, int W, int row0, int rows, uint32_t seed,
                 int noise, int shadow)
{
    SynthArgs a{};
    a.stack = d_stack; a.plane_stride = plane_stride; a.N = N; a.L = (N - 2) / 4; a.H = H; a.W = W; a.row0 = row0; a.rows = rows;
    a.seed = seed; a.noise = noise;
    if (shadow) {
        a.sy0 = (int)(0.30 * H); a.sy1 = (int)(0.30 * H + 0.387 * H);
        a.sx0 = (int)(0.55 * W); a.sx1 = (int)(0.55 * W + 0.387 * W);
    }
    const size_t nq = ((size_t)rows * W + 3) / 4;
    if (nq == 0) return SLGC_OK;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((nq + 255) / 256), N), dim3(256), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}
