// ingest.hip -- the frames either side of the decode path (SURVEY.md section 8(f) "next" rows 1 and 3).
//
// K5  k_bgr_to_gray      cv2.cvtColor(BGR2GRAY) on 8-bit frames (scanner/grayCode/decode_codes.py:70-87,
//                        src/3-capture_decode.py:66): fixed-point luma, written straight into the [N][H][W] uint8 stack
//                        the decode kernels read.  OpenCV's 8-bit path (third-party, 4.8.0.76 not installed -> PARITY
//                        UNPINNED): Y = (B*BY + G*GY + R*RY + (1 << (shift-1))) >> shift with shift = 15,
//                        (RY, GY, BY) = (9798, 19235, 3735) in 4.x; the older 14-bit set (4899, 9617, 1868) is selectable.
// K6  k_frame_diff_*     count(|frame[j+1] - frame[j]| > thresh) for consecutive frames -- the arithmetic of
//                        remove_bad_images (decode_codes.py:34-68: cv2.absdiff + np.argwhere + len).  uint8 frames: one lane
//                        carries 16 pixels through ALL frames (every frame byte is read once: N B / pixel), two pixels per
//                        32-bit operation; float64 frames and ragged sizes take the generic kernel.  The keep/drop state
//                        machine (:56-66) is host logic.
#include <cmath>

#include "slgc_internal.h"

namespace {

// 16 pixels per lane = 48 input bytes, one 16-byte store.  The 48 bytes of a lane are three 16-byte pieces 48 bytes apart from the next
// lane's: read that way a wave's load instruction touches three times the cache lines it uses.  The workgroup reads its 12 KB with
// consecutive lanes on consecutive 16-byte pieces (three fully coalesced 1 KB-per-wave loads, non-temporal: every byte is read once), parks
// them in LDS and every lane picks its own 48 bytes up from there.
// TILED (slgc_to_gray_tiled_dev): blockIdx.y = frame f of n_frames, npix = pixels of ONE frame, and pixel p of frame f goes to
// gray[((p >> k) * n_frames + f) << k | (p & (2^k - 1))] -- the tile-interleaved stack [tile][N][2^k] the decode kernels read when slgc_tune
// "stack_tile_log2" = k: the ingest pass rewrites every byte anyway, so the other layout costs it nothing.
template <bool TILED>
__global__ void __launch_bounds__(256) k_bgr_to_gray(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray, size_t npix, int ry, int gy,
                                                     int by, int shift, int n_frames, int tile_log2)
{
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    __shared__ uint4 s_in[768];
    if (TILED) bgr += (size_t)blockIdx.y * 3 * npix;
    auto out_at = [&](size_t p) -> uint8_t * {      // where pixel p (of this frame) lives
        if (!TILED) return gray + p;
        return gray + ((((p >> tile_log2) * (size_t)n_frames + blockIdx.y) << tile_log2) | (p & (((size_t)1 << tile_log2) - 1)));
    };
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int rnd = 1 << (shift - 1);
    const size_t wg_px0 = (size_t)blockIdx.x * 4096;
    const bool whole = wg_px0 + 4096 <= npix && ((uintptr_t)bgr & 15) == 0 && ((uintptr_t)gray & 15) == 0;      // workgroup-uniform
    if (whole) {
        const v4u_ *src = reinterpret_cast<const v4u_ *>(bgr) + (size_t)blockIdx.x * 768;
        v4u_ t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = __builtin_nontemporal_load(src + i * 256 + threadIdx.x);
#pragma unroll
        for (int i = 0; i < 3; ++i) s_in[i * 256 + threadIdx.x] = make_uint4(t[i].x, t[i].y, t[i].z, t[i].w);
        __syncthreads();
        const uint4 a = s_in[3 * threadIdx.x], b = s_in[3 * threadIdx.x + 1], c = s_in[3 * threadIdx.x + 2];
        const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
        uint32_t out[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t B = (w[(3 * j) >> 2] >> (8 * ((3 * j) & 3))) & 0xffu, G = (w[(3 * j + 1) >> 2] >> (8 * ((3 * j + 1) & 3))) & 0xffu,
                           R = (w[(3 * j + 2) >> 2] >> (8 * ((3 * j + 2) & 3))) & 0xffu;
            out[j >> 2] |= (uint32_t)((int)(B * by + G * gy + R * ry + rnd) >> shift) << (8 * (j & 3));
        }
        *reinterpret_cast<uint4 *>(out_at(q * 16)) = make_uint4(out[0], out[1], out[2], out[3]);      // (16 pixels never straddle a plane piece: 2^k >= 256)
    } else {
        for (size_t p = q * 16; p < npix && p < q * 16 + 16; ++p)
            *out_at(p) = (uint8_t)(((int)bgr[3 * p] * by + (int)bgr[3 * p + 1] * gy + (int)bgr[3 * p + 2] * ry + rnd) >> shift);
    }
}

// planar [N][plane_stride] -> tile-interleaved [tile][N][2^k] (slgc_tile_stack_dev): for stacks that arrive grey and planar.  16 bytes per lane
// where the frame allows it; blockIdx.y = frame.
__global__ void __launch_bounds__(256) k_tile_stack(const uint8_t *__restrict__ planar, size_t plane_stride, size_t npix, int n_frames, int tile_log2,
                                                    uint8_t *__restrict__ tiled)
{
    const size_t p = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (p >= npix) return;
    const uint8_t *src = planar + (size_t)blockIdx.y * plane_stride + p;
    uint8_t *dst = tiled + ((((p >> tile_log2) * (size_t)n_frames + blockIdx.y) << tile_log2) | (p & (((size_t)1 << tile_log2) - 1)));
    if (p + 16 <= npix && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
        *reinterpret_cast<v4u_ *>(dst) = __builtin_nontemporal_load(reinterpret_cast<const v4u_ *>(src));
    } else {
        for (size_t i = 0; i < 16 && p + i < npix; ++i) dst[i] = src[i];
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
        if constexpr (sizeof(T) == 1) {
            const int d = (int)b[i] - (int)a[i];                  // cv2.absdiff on 8-bit frames
            c += (double)(d < 0 ? -d : d) > thresh ? 1u : 0u;
        } else {
            const double d = fabs((double)b[i] - (double)a[i]);   // NaN compares false like NumPy's '>'
            c += d > thresh ? 1u : 0u;
        }
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

// uint8 frames of a multiple of 16 pixels on a 16-byte boundary.  A lane keeps the previous frame's 16 pixels unpacked into 16-bit fields
// (even bytes / odd bytes of each dword) and meets the next frame with two pixels per operation: with D = prev - next over both fields
// at once (borrows between the fields cancel in the sums below) and C = (0x7fff - t) in both fields,
//     C + D  has bit 15 / 31 set  <=>  prev - next >= t + 1          C - D  has it set  <=>  next - prev >= t + 1
// for an integer threshold 0 <= t < 255 (|d| > thresh for integer d <=> |d| >= floor(thresh) + 1; the host deals with thresh < 0, >= 255
// and NaN).  Every field stays inside its 16 bits: 0x7fff - t +- 255.
constexpr unsigned kFieldMask = 0x00ff00ffu, kSignBits = 0x80008000u;

// Sum of c over the 64 lanes of a fully active wave, as a wave-uniform value: a DPP butterfly inside each row of 16 lanes (vector ALU,
// no LDS round trips), then the four row sums through v_readlane.
__device__ __forceinline__ unsigned wave_sum(unsigned c)
{
    c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0xB1, 0xf, 0xf, true);    // quad_perm:[1,0,3,2]
    c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x4E, 0xf, 0xf, true);    // quad_perm:[2,3,0,1]
    c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x141, 0xf, 0xf, true);   // row_half_mirror
    c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x140, 0xf, 0xf, true);   // row_mirror
    return (unsigned)(__builtin_amdgcn_readlane((int)c, 0) + __builtin_amdgcn_readlane((int)c, 16) + __builtin_amdgcn_readlane((int)c, 32) +
                      __builtin_amdgcn_readlane((int)c, 48));
}

// blockIdx.y = a segment of consecutive frame pairs [pair0, pair1): the launcher cuts the frames into segments when the workgroups of one
// pass would fill the chip's resident slots unevenly (4096x3000: 3 000 workgroups on 2 048 slots = two rounds for 1.46 rounds of work; two
// segments = 6 000 shorter workgroups = 2.93 rounds, at the price of reading one frame per lane twice).  Loads run two frames ahead
// (deeper: no faster), non-temporal: every frame byte is read once.
// counts: kFdSlots partial rows of n_frames - 1 counters, a workgroup adds into row blockIdx.x % kFdSlots (a single row takes ~3 000 atomic adds
// per counter on three cache lines); k_frame_diff_fold sums the rows.
constexpr int kFdSlots = 64;
__global__ void __launch_bounds__(256) k_frame_diff_u8x16(const uint4 *__restrict__ frames, size_t chunks, int n_frames, int pairs_per_seg, unsigned c_fields,
                                                          unsigned long long *__restrict__ counts, int row_stride)
{
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    extern __shared__ unsigned s_cnt[];                                  // pairs_per_seg counters of this workgroup
    for (int p = threadIdx.x; p < pairs_per_seg; p += 256) s_cnt[p] = 0u;
    __syncthreads();
    const int pair0 = (int)blockIdx.y * pairs_per_seg, pair1 = min(n_frames - 1, pair0 + pairs_per_seg);      // frames pair0 .. pair1
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = q < chunks;
    const size_t qc = active ? q : chunks - 1;
    const v4u_ *fr = reinterpret_cast<const v4u_ *>(frames) + qc;
    auto fetch = [&](int f) { return __builtin_nontemporal_load(fr + (size_t)f * chunks); };
    unsigned lo[4], hi[4];
    {
        const v4u_ a = fetch(pair0);
        const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = w[k] & kFieldMask;
            hi[k] = (w[k] >> 8) & kFieldMask;
        }
    }
    v4u_ n1 = fetch(min(pair0 + 1, pair1)), n2 = fetch(min(pair0 + 2, pair1));   // (the segment's last frame is re-read past its end)
    for (int f = pair0 + 1; f <= pair1; ++f) {
        const v4u_ b = n1;
        n1 = n2;
        n2 = fetch(min(f + 2, pair1));
        const unsigned w[4] = {b.x, b.y, b.z, b.w};
        unsigned c = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned lb = w[k] & kFieldMask, hb = (w[k] >> 8) & kFieldMask;
            const unsigned dl = lo[k] - lb, dh = hi[k] - hb;
            c += __popc(((c_fields + dl) | (c_fields - dl)) & kSignBits) + __popc(((c_fields + dh) | (c_fields - dh)) & kSignBits);
            lo[k] = lb;
            hi[k] = hb;
        }
        if (!active) c = 0;
        const unsigned total = wave_sum(c);
        if ((threadIdx.x & 63) == 0 && total) atomicAdd(&s_cnt[f - 1 - pair0], total);
    }
    __syncthreads();
    unsigned long long *row = counts + (size_t)(blockIdx.x % kFdSlots) * row_stride;
    for (int p = threadIdx.x; p < pair1 - pair0; p += 256)
        if (s_cnt[p]) atomicAdd(row + pair0 + p, (unsigned long long)s_cnt[p]);
}

__global__ void __launch_bounds__(64) k_frame_diff_fold(const unsigned long long *__restrict__ rows, int row_stride, int n, unsigned long long *__restrict__ counts)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n) return;
    unsigned long long t = 0;
    for (int r = 0; r < kFdSlots; ++r) t += rows[(size_t)r * row_stride + p];
    counts[p] = t;
}

__global__ void k_fill_u64(unsigned long long *__restrict__ out, int n, unsigned long long v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

}  // namespace

int launch_bgr_to_gray(slgc_ctx *ctx, const uint8_t *d_bgr, uint8_t *d_gray, size_t npix, int coeff_bits)
{
    if (npix == 0) return SLGC_OK;
    const int ry = coeff_bits == 14 ? 4899 : 9798, gy = coeff_bits == 14 ? 9617 : 19235, by = coeff_bits == 14 ? 1868 : 3735;
    hipLaunchKernelGGL(k_bgr_to_gray<false>, dim3((unsigned)(((npix + 15) / 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_bgr, d_gray, npix, ry, gy, by,
                       coeff_bits, 1, 0);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_bgr_to_gray_tiled(slgc_ctx *ctx, const uint8_t *d_bgr, uint8_t *d_tiled, int n_frames, size_t npix, int coeff_bits, int tile_log2)
{
    if (npix == 0 || n_frames == 0) return SLGC_OK;
    const int ry = coeff_bits == 14 ? 4899 : 9798, gy = coeff_bits == 14 ? 9617 : 19235, by = coeff_bits == 14 ? 1868 : 3735;
    hipLaunchKernelGGL(k_bgr_to_gray<true>, dim3((unsigned)(((npix + 15) / 16 + 255) / 256), (unsigned)n_frames), dim3(256), 0, ctx->stream, d_bgr, d_tiled, npix,
                       ry, gy, by, coeff_bits, n_frames, tile_log2);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_tile_stack(slgc_ctx *ctx, const uint8_t *d_planar, size_t plane_stride, int n_frames, size_t npix, int tile_log2, uint8_t *d_tiled)
{
    if (npix == 0 || n_frames == 0) return SLGC_OK;
    hipLaunchKernelGGL(k_tile_stack, dim3((unsigned)(((npix + 15) / 16 + 255) / 256), (unsigned)n_frames), dim3(256), 0, ctx->stream, d_planar, plane_stride, npix,
                       n_frames, tile_log2, d_tiled);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_frame_diff_counts(slgc_ctx *ctx, const void *d_frames, int dtype, int n_frames, size_t elems, double thresh,
                             unsigned long long *d_counts)
{
    if (n_frames < 2) return SLGC_OK;
    HIP_TRY(ctx, hipMemsetAsync(d_counts, 0, 8 * (size_t)(n_frames - 1), ctx->stream));
    if (elems == 0) return SLGC_OK;
    if (dtype == SLGC_U8 && thresh == thresh) {                      // integer differences: |d| > thresh <=> |d| >= floor(thresh) + 1
        if (thresh < 0.0) {                                              // every pixel counts
            hipLaunchKernelGGL(k_fill_u64, dim3((n_frames + 254) / 256), dim3(256), 0, ctx->stream, d_counts, n_frames - 1, (unsigned long long)elems);
            HIP_TRY(ctx, hipGetLastError());
            return SLGC_OK;
        }
        if (thresh >= 255.0) return SLGC_OK;                             // none can
        const unsigned t = (unsigned)thresh;                             // floor, 0..254
        if (elems % 16 == 0 && ((uintptr_t)d_frames & 15) == 0 && (size_t)(n_frames - 1) * 4 <= 48 * 1024) {
            const size_t chunks = elems / 16;
            // frame segments: the number (1 .. 4) whose workgroups fill whole rounds of the 256 CUs x 8 resident workgroups best, counting the
            // frame each extra segment reads twice
            const double wgs = (double)((chunks + 255) / 256), slots = 2048.0;
            static const int force = xcd_env("SLGC_FD_SEGS", 0);        // A/B
            int segs = 1;
            double best = 1e30;
            for (int sgs = 1; sgs <= 4 && sgs <= n_frames - 1; ++sgs) {
                const double rounds = wgs * sgs / slots, cost = std::ceil(rounds) / rounds * (1.0 + (double)(sgs - 1) / (double)(n_frames - 1));
                if (cost < best - 1e-9) { best = cost; segs = sgs; }
            }
            if (force >= 1 && force <= n_frames - 1) segs = force;
            const int per = (n_frames - 1 + segs - 1) / segs;
            const dim3 grid((unsigned)((chunks + 255) / 256), (unsigned)((n_frames - 1 + per - 1) / per));
            const int row_stride = (n_frames - 1 + 15) & ~15;             // rows start on their own 128-byte lines
            void *rows;
            int rc = slgc_ws(ctx, 4, (size_t)kFdSlots * row_stride * 8, &rows);      // (slot 4: the list build's counts; slot 7 may hold the caller's d_counts)
            if (rc) return rc;
            HIP_TRY(ctx, hipMemsetAsync(rows, 0, (size_t)kFdSlots * row_stride * 8, ctx->stream));
            hipLaunchKernelGGL(k_frame_diff_u8x16, grid, dim3(256), (size_t)per * 4, ctx->stream, (const uint4 *)d_frames, chunks, n_frames, per,
                               (0x7fffu - t) * 0x00010001u, (unsigned long long *)rows, row_stride);
            hipLaunchKernelGGL(k_frame_diff_fold, dim3((unsigned)((n_frames - 1 + 63) / 64)), dim3(64), 0, ctx->stream, (const unsigned long long *)rows, row_stride,
                               n_frames - 1, d_counts);
            HIP_TRY(ctx, hipGetLastError());
            return SLGC_OK;
        }
    }
    const unsigned bx = (unsigned)((elems + 256 * 8 - 1) / (256 * 8) < 1024 ? (elems + 256 * 8 - 1) / (256 * 8) : 1024);
    const dim3 grid(bx ? bx : 1, n_frames - 1);
    if (dtype == SLGC_U8)
        hipLaunchKernelGGL(k_frame_diff_count<uint8_t>, grid, dim3(256), 0, ctx->stream, (const uint8_t *)d_frames, elems, thresh, d_counts);
    else
        hipLaunchKernelGGL(k_frame_diff_count<double>, grid, dim3(256), 0, ctx->stream, (const double *)d_frames, elems, thresh, d_counts);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}
