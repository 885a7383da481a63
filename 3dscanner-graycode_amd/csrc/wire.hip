// wire.hip -- 3-byte wire format of the decoded maps for the multi-GPU exchange.
//
// The row-sharded scan reassembles the cloud by all-gathering the decoded (h, v) maps and triangulating them on every rank
// (comm.cpp, scanner/sharded.py); that exchange, not the kernels, bounds a sharded scan on xGMI's point-to-point links.  The
// maps are int16 with values -1 .. 2^L - 1; for L <= 11 (N <= 49 frames; the reference's captures use L = 10) both fit in 12
// bits each, so a pixel travels as 3 bytes instead of 4:   bits 0..11 = h, bits 12..23 = v, 0xFFF = -1 (undecodable).
//
// K7  k_pack_hv24     int16 h, v  ->  3 bytes / pixel   (a lane takes 4 pixels: two 8-byte loads, one 12-byte store)
// K8  k_unpack_hv24   3 bytes / pixel -> int16 h, v     (one 12-byte load, two 8-byte stores)
// Both are streaming kernels with byte-wise paths for misaligned buffers and for the last < 4 pixels.
#include "slgc_internal.h"

namespace {

__device__ __forceinline__ uint32_t enc12(int x) { return (uint32_t)x & 0xfffu; }                  // -1 -> 0xFFF
__device__ __forceinline__ int16_t dec12(uint32_t x) { return x == 0xfffu ? (int16_t)-1 : (int16_t)x; }

__global__ void __launch_bounds__(256) k_pack_hv24(const int16_t *__restrict__ h, const int16_t *__restrict__ v, size_t npix,
                                                   uint8_t *__restrict__ out, int vec_ok)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= npix) return;
    if (vec_ok && q * 4 + 4 <= npix) {
        const uint2 hw = reinterpret_cast<const uint2 *>(h)[q], vw = reinterpret_cast<const uint2 *>(v)[q];
        uint32_t p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t hh = ((j < 2 ? hw.x : hw.y) >> (16 * (j & 1))) & 0xffffu, vv = ((j < 2 ? vw.x : vw.y) >> (16 * (j & 1))) & 0xffffu;
            p[j] = (hh & 0xfffu) | ((vv & 0xfffu) << 12);         // int16 -1 = 0xFFFF -> 0xFFF
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(out) + q * 3;
        dst[0] = p[0] | (p[1] << 24);
        dst[1] = (p[1] >> 8) | (p[2] << 16);
        dst[2] = (p[2] >> 16) | (p[3] << 8);
    } else {
        for (size_t i = q * 4; i < npix && i < q * 4 + 4; ++i) {
            const uint32_t p = enc12(h[i]) | (enc12(v[i]) << 12);
            out[3 * i] = (uint8_t)p; out[3 * i + 1] = (uint8_t)(p >> 8); out[3 * i + 2] = (uint8_t)(p >> 16);
        }
    }
}

__global__ void __launch_bounds__(256) k_unpack_hv24(const uint8_t *__restrict__ in, size_t npix, int16_t *__restrict__ h,
                                                     int16_t *__restrict__ v, int vec_ok)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= npix) return;
    if (vec_ok && q * 4 + 4 <= npix) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(in) + q * 3;
        uint2 hw, vw;
        unpack_hv24_x4(src[0], src[1], src[2], hw, vw);
        reinterpret_cast<uint2 *>(h)[q] = hw;
        reinterpret_cast<uint2 *>(v)[q] = vw;
    } else {
        for (size_t i = q * 4; i < npix && i < q * 4 + 4; ++i) {
            const uint32_t p = (uint32_t)in[3 * i] | ((uint32_t)in[3 * i + 1] << 8) | ((uint32_t)in[3 * i + 2] << 16);
            h[i] = dec12(p & 0xfffu);
            v[i] = dec12(p >> 12);
        }
    }
}

}  // namespace

int launch_pack_hv24(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, size_t npix, uint8_t *d_out)
{
    if (npix == 0) return SLGC_OK;
    const int vec_ok = (((uintptr_t)d_h | (uintptr_t)d_v) % 8 == 0) && ((uintptr_t)d_out % 4 == 0);
    hipLaunchKernelGGL(k_pack_hv24, dim3((unsigned)(((npix + 3) / 4 + 255) / 256)), dim3(256), 0, ctx->stream, d_h, d_v, npix, d_out, vec_ok);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_unpack_hv24(slgc_ctx *ctx, const uint8_t *d_in, size_t npix, int16_t *d_h, int16_t *d_v)
{
    if (npix == 0) return SLGC_OK;
    const int vec_ok = (((uintptr_t)d_h | (uintptr_t)d_v) % 8 == 0) && ((uintptr_t)d_in % 4 == 0);
    hipLaunchKernelGGL(k_unpack_hv24, dim3((unsigned)(((npix + 3) / 4 + 255) / 256)), dim3(256), 0, ctx->stream, d_in, npix, d_h, d_v, vec_ok);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}
