// ingest.hip -- the frames either side of the decode path (SURVEY.md section 8(f) "next" rows 1 and 3).
//
// K5  k_bgr_to_gray      cv2.cvtColor(BGR2GRAY) on 8-bit frames (scanner/grayCode/decode_codes.py:70-87,
//                        src/3-capture_decode.py:66): fixed-point luma, written straight into the [N][H][W] uint8 stack
//                        the decode kernels read.  OpenCV's 8-bit path (third-party, 4.8.0.76 not installed -> PARITY
//                        UNPINNED): Y = (B*BY + G*GY + R*RY + (1 << (shift-1))) >> shift with shift = 15,
//                        (RY, GY, BY) = (9798, 19235, 3735) in 4.x; the older 14-bit set (4899, 9617, 1868) is selectable.
// K6  k_frame_diff_count count(|frame[j+1] - frame[j]| > thresh) for consecutive frames -- the arithmetic of
//                        remove_bad_images (decode_codes.py:34-68: cv2.absdiff + np.argwhere + len); wave64 reduce
//                        (__shfl_down) then one atomic per workgroup.  The keep/drop state machine (:56-66) is host logic.
#include "slgc_internal.h"

namespace {

__global__ void __launch_bounds__(256) k_bgr_to_gray(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray, size_t npix, int ry, int gy,
                                                     int by, int shift)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;   // 4 pixels = 12 input bytes = 3 dwords per lane
    if (q * 4 >= npix) return;
    const int rnd = 1 << (shift - 1);
    if (q * 4 + 4 <= npix && ((uintptr_t)bgr & 3) == 0 && ((uintptr_t)gray & 3) == 0) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(bgr) + q * 3;
        const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
        const uint32_t b[4] = {w0 & 0xff, w0 >> 24, (w1 >> 16) & 0xff, (w2 >> 8) & 0xff};
        const uint32_t g[4] = {(w0 >> 8) & 0xff, w1 & 0xff, w1 >> 24, (w2 >> 16) & 0xff};
        const uint32_t r[4] = {(w0 >> 16) & 0xff, (w1 >> 8) & 0xff, w2 & 0xff, w2 >> 24};
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) out |= (uint32_t)((int)(b[j] * by + g[j] * gy + r[j] * ry + rnd) >> shift) << (8 * j);
        reinterpret_cast<uint32_t *>(gray)[q] = out;
    } else {
        for (size_t p = q * 4; p < npix && p < q * 4 + 4; ++p)
            gray[p] = (uint8_t)(((int)bgr[3 * p] * by + (int)bgr[3 * p + 1] * gy + (int)bgr[3 * p + 2] * ry + rnd) >> shift);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_frame_diff_count(const T *__restrict__ frames, size_t elems, double thresh,
                                                          unsigned long long *__restrict__ counts)
{
    const int pair = blockIdx.y;
    const T *a = frames + (size_t)pair * elems, *b = a + elems;
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < elems; i += (size_t)gridDim.x * 256) {
        const double d = fabs((double)b[i] - (double)a[i]);   // cv2.absdiff; NaN compares false like NumPy's '>'
        c += d > thresh ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    __shared__ unsigned w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = w[0] + w[1] + w[2] + w[3];
        if (t) atomicAdd(counts + pair, (unsigned long long)t);
    }
}

}  // namespace

int launch_bgr_to_gray(slgc_ctx *ctx, const uint8_t *d_bgr, uint8_t *d_gray, size_t npix, int coeff_bits)
{
    if (npix == 0) return SLGC_OK;
    const int ry = coeff_bits == 14 ? 4899 : 9798, gy = coeff_bits == 14 ? 9617 : 19235, by = coeff_bits == 14 ? 1868 : 3735;
    hipLaunchKernelGGL(k_bgr_to_gray, dim3((unsigned)(((npix + 3) / 4 + 255) / 256)), dim3(256), 0, ctx->stream, d_bgr, d_gray, npix, ry, gy, by,
                       coeff_bits);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_frame_diff_counts(slgc_ctx *ctx, const void *d_frames, int dtype, int n_frames, size_t elems, double thresh,
                             unsigned long long *d_counts)
{
    if (n_frames < 2) return SLGC_OK;
    HIP_TRY(ctx, hipMemsetAsync(d_counts, 0, 8 * (size_t)(n_frames - 1), ctx->stream));
    if (elems == 0) return SLGC_OK;
    const unsigned bx = (unsigned)((elems + 256 * 8 - 1) / (256 * 8) < 1024 ? (elems + 256 * 8 - 1) / (256 * 8) : 1024);
    const dim3 grid(bx ? bx : 1, n_frames - 1);
    if (dtype == SLGC_U8)
        hipLaunchKernelGGL(k_frame_diff_count<uint8_t>, grid, dim3(256), 0, ctx->stream, (const uint8_t *)d_frames, elems, thresh, d_counts);
    else
        hipLaunchKernelGGL(k_frame_diff_count<double>, grid, dim3(256), 0, ctx->stream, (const double *)d_frames, elems, thresh, d_counts);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}
