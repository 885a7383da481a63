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

}  // namespace

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
