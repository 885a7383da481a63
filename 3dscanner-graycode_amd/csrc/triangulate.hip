// triangulate.hip -- ray triangulation kernels (gfx950).
//
// K3  Triangulate.triangulate (scanner/triangulation/triangulate.py:73-97):
//       cam ray   = [undistortPoints(cam_pts, cam_mtx, cam_dist, R=proj_R); 1]   (:84)  float32
//       proj ray  = [undistortPoints(proj_pts, proj_mtx, proj_dist); 1]          (:85)  float32
//       law of sines on the camera ray against the baseline T                    (:86-95) float64
//     undistortPoints is OpenCV 4.8's cvUndistortPointsInternal (third-party; published algorithm restated:
//     5 fixed-point iterations, icdist < 0 bail-out, R applied afterwards, result rounded to float32).
//
// Two front ends share the per-point device function:
//   k_triangulate_list  float32 [M][2] point lists -> float64 (3,M)           (API parity with the reference)
//   k_triangulate_maps  dense int16 maps (decode output) -> dense float32 XYZ  (device-resident scan path)
//
// Compiled with -ffp-contract=off; float32 divide/sqrt are correctly rounded (hipcc default), so the float32 steps
// NumPy performs (:90 and the norm in :92) are reproduced bit for bit.
#include <cstdlib>

#include "slgc_internal.h"
#include "tri_math.h"

int launch_triangulate_maps_direct(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, size_t npix, size_t first, int W, int row0,
                                   int proj_w, int proj_h, int mode, float *d_xyz, unsigned long long *d_count);
static int launch_triangulate_maps_body(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, size_t npix, int W, int row0, int rows,
                                        int proj_w, int proj_h, int mode, bool direct, float *d_xyz, unsigned long long *d_count,
                                        const uint8_t *d_wire);

namespace {

// cv::undistortPoints(src, K, dist, R) for one point, default TermCriteria(MAX_ITER, 5, 0.01): the undistorted point before R.
__device__ __forceinline__ void undistort_xy(float uf, float vf, const double (&kk)[4], const double (&k)[12], double &xo, double &yo)
{
    const double fx = kk[0], fy = kk[1], cx = kk[2], cy = kk[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double u = (double)uf, v = (double)vf;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) {  // regression_14583 branch
            x = (u - cx) * ifx;
            y = (v - cy) * ify;
            break;
        }
        const double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    xo = x;
    yo = y;
}

__device__ __forceinline__ Ray2 undistort_point(float uf, float vf, const double (&kk)[4], const double (&k)[12], const double *R)
{
    double x, y;
    undistort_xy(uf, vf, kk, k, x, y);
    Ray2 o;
    if (R) {
        const double xx = R[0] * x + R[1] * y + R[2];
        const double yy = R[3] * x + R[4] * y + R[5];
        const double ww = 1. / (R[6] * x + R[7] * y + R[8]);
        o.x = (float)(xx * ww);
        o.y = (float)(yy * ww);
    } else {
        // identity R: xx = x, yy = y, ww = 1/(0*x + 0*y + 1) = 1 exactly
        o.x = (float)x;
        o.y = (float)y;
    }
    return o;
}

// Valid-pixel counting: a single counter word serialises at the memory side (~11 ns per atomic, 48k waves = 0.5 ms),
// so each workgroup adds once into one of kCountSlots words on separate 128-byte lines; k_count_finish folds them.
constexpr int kCountSlots = 256;
constexpr int kCountStride = 16;  // in 8-byte words

__device__ __forceinline__ void block_count_add(unsigned long long *slots, unsigned n)
{
    __shared__ unsigned wsum[4];
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t) atomicAdd(slots + (size_t)(blockIdx.x % kCountSlots) * kCountStride, (unsigned long long)t);
    }
}

__global__ void __launch_bounds__(256) k_count_finish(unsigned long long *__restrict__ slots, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long part[4];
    unsigned long long v = slots[(size_t)threadIdx.x * kCountStride];
    slots[(size_t)threadIdx.x * kCountStride] = 0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) *total += part[0] + part[1] + part[2] + part[3];
}

// triangulate.py:86-95 for one correspondence (cam/proj are the float32 normalised points).
template <int MODE>
__device__ __forceinline__ Xyz law_of_sines(Ray2 cam, Ray2 prj, const double (&T)[3], double t_len)
{
    const float cn = sqrtf((cam.x * cam.x + cam.y * cam.y) + 1.0f);       // np.linalg.norm of float32 (:90)
    const float rx = cam.x / cn, ry = cam.y / cn, rz = 1.0f / cn;         // NormedL (float32)
    const float qn = sqrtf((prj.x * prj.x + prj.y * prj.y) + 1.0f);       // norm inside :92 (float32)
    const double cos_a = (((-T[0]) * (double)rx + (-T[1]) * (double)ry) + (-T[2]) * (double)rz) / t_len;     // :91
    const double cos_b = ((T[0] * (double)prj.x + T[1] * (double)prj.y) + T[2]) / (t_len * (double)qn);      // :92
    double len;
    if constexpr (MODE == SLGC_TRI_EXACT) {
        const double alpha = acos(cos_a), beta = acos(cos_b);
        const double gamma = 3.141592653589793 - alpha - beta;                                               // :93
        len = t_len * sin(beta) / sin(gamma);                                                                // :94
    } else {
        // sin(gamma) = sin(alpha + beta); alpha, beta in [0, pi] so both sines are the non-negative roots.
        const double sin_a = sqrt(fmax(0.0, 1.0 - cos_a * cos_a)), sin_b = sqrt(fmax(0.0, 1.0 - cos_b * cos_b));
        len = t_len * sin_b / (sin_a * cos_b + cos_a * sin_b);
    }
    return Xyz{(double)rx * len, (double)ry * len, (double)rz * len};                                        // :95
}

template <int MODE>
__global__ void __launch_bounds__(256) k_triangulate_list(const Calib c_calib, const float *__restrict__ cam,
                                                          const float *__restrict__ proj, size_t M, double *__restrict__ xyz)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= M) return;
    const float2 c = reinterpret_cast<const float2 *>(cam)[q], p = reinterpret_cast<const float2 *>(proj)[q];
    const Ray2 a = undistort_point(c.x, c.y, c_calib.cam_k, c_calib.cam_d, c_calib.R);
    const Ray2 b = undistort_point(p.x, p.y, c_calib.proj_k, c_calib.proj_d, nullptr);
    const Xyz r = law_of_sines<MODE>(a, b, c_calib.T, c_calib.t_len);
    xyz[q] = r.x;
    xyz[M + q] = r.y;
    xyz[2 * M + q] = r.z;
}

// One pixel per lane, lanes along x.  XYZ is staged through LDS so the 12-byte records leave as whole dwords.
template <int MODE>
__global__ void __launch_bounds__(256) k_triangulate_maps(const Calib c_calib, const int16_t *__restrict__ h,
                                                          const int16_t *__restrict__ v, size_t npix, size_t first,
                                                          int W, int row0, int proj_w, int proj_h, float *__restrict__ xyz,
                                                          unsigned long long *__restrict__ count)
{
    __shared__ float stage[256 * 3];
    const size_t base = (size_t)blockIdx.x * 256;
    const size_t p = base + threadIdx.x;
    float X = __builtin_nanf(""), Y = X, Z = X;
    bool ok = false;
    if (p < npix) {
        const int hv = h[p], vv = v[p];
        ok = !(hv == -1 || vv == -1);                                         // triangulate.py:56
        if (ok) {
            const int x = (int)((first + p) % (size_t)W), y = row0 + (int)((first + p) / (size_t)W);
            const float pu = (float)min(proj_w - 1, hv), pv = (float)min(proj_h - 1, vv);   // :60-61
            const Ray2 a = undistort_point((float)x, (float)y, c_calib.cam_k, c_calib.cam_d, c_calib.R);
            const Ray2 b = undistort_point(pu, pv, c_calib.proj_k, c_calib.proj_d, nullptr);
            const Xyz r = law_of_sines<MODE>(a, b, c_calib.T, c_calib.t_len);
            X = (float)r.x; Y = (float)r.y; Z = (float)r.z;
        }
    }
    stage[3 * threadIdx.x] = X;
    stage[3 * threadIdx.x + 1] = Y;
    stage[3 * threadIdx.x + 2] = Z;
    if (count) block_count_add(count, ok ? 1u : 0u);
    __syncthreads();
    const size_t nfl = min((size_t)768, (npix - base) * 3);
    for (size_t i = threadIdx.x; i < nfl; i += 256) xyz[base * 3 + i] = stage[i];
}


// ---------------------------------------------------------------------------------------------------------------
// Ray tables.  Both undistortPoints calls of triangulate.py:84-85 see only integer pixel coordinates on the dense
// path (camera pixel (x, y); projector pixel (min(w-1,h), min(h-1,v))), and their float32 results depend only on the
// calibration -- not on the scan.  They are therefore evaluated once per calibration with the exact fp64 routine above
// and kept in HBM: cam table [rows][W] float2 (8 B/pixel, streamed), projector table [proj_h][proj_w] float2 (gathered,
// L2 / Infinity-Cache resident).  The per-scan kernel is then bandwidth-shaped instead of fp64-divide-shaped, and the
// float32 ray components are bit-identical to computing them in place.
__global__ void __launch_bounds__(256) k_build_cam_lut(const Calib c, float2 *__restrict__ lut, int W, int row0, size_t npix)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const Ray2 a = undistort_point((float)(int)(p % (size_t)W), (float)(row0 + (int)(p / (size_t)W)), c.cam_k, c.cam_d, c.R);
    lut[p] = make_float2(a.x, a.y);
}

// CamNodes table (tri_math.h): the exact rays at x = 4 (n - 1), n = 0 .. W / 4 + 2, of every row of the band
__global__ void __launch_bounds__(256) k_build_cam_nodes(const Calib c, float2 *__restrict__ nodes, int ne, int row0, size_t total)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int n = (int)(p % (size_t)ne), row = (int)(p / (size_t)ne);
    const Ray2 a = undistort_point((float)(4 * (n - 1)), (float)(row0 + row), c.cam_k, c.cam_d, c.R);
    nodes[p] = make_float2(a.x, a.y);
}

// ... and how far the rays the kernels will interpolate from it are from the per-pixel table: max over the band of the error measure
// below, as float bits through atomicMax (non-negative floats order like unsigned integers)
__global__ void __launch_bounds__(256) k_check_cam_nodes(const float2 *__restrict__ lut, CamNodes cn, size_t ngroups, unsigned *__restrict__ worst)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    float err = 0.0f;
    if (g < ngroups) {
        const float2 *e = cam_node_ptr(cn, (uint32_t)g);
        const cam_v4f n01 = *reinterpret_cast<const cam_v4f *>(e), n23 = *reinterpret_cast<const cam_v4f *>(e + 2);
        float cx[4], cy[4];
        cam_rays_from_nodes(n01, n23, cx, cy);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 t = lut[4 * g + j];
            // two measures in one number (accepted when <= 2.4e-7): the direction error against 2 ulp of a float32 in [1, 2), and the
            // relative error of a component the kernels will use (>= kCamNodeTiny, see cam_rays_exact_where_tiny) against 4e-6
            const float ax = fabsf(cx[j] - t.x), ay = fabsf(cy[j] - t.y);
            const float ex = fmaxf(ax / fmaxf(1.0f, fabsf(t.x)), (2.4e-7f / 4e-6f) * ax / fmaxf(kCamNodeTiny, fabsf(t.x)));
            const float ey = fmaxf(ay / fmaxf(1.0f, fabsf(t.y)), (2.4e-7f / 4e-6f) * ay / fmaxf(kCamNodeTiny, fabsf(t.y)));
            err = fmaxf(err, !(ex == ex) || !(ey == ey) ? __builtin_huge_valf() : fmaxf(ex, ey));
        }
    }
    for (int o = 32; o > 0; o >>= 1) err = fmaxf(err, __shfl_down(err, o, 64));
    // one atomic per wave at most, and none once the running maximum (a plain read: it only ever grows) already covers this wave
    if ((threadIdx.x & 63) == 0 && __float_as_uint(err) > *reinterpret_cast<volatile unsigned *>(worst)) atomicMax(worst, __float_as_uint(err));
}

// The same measure over a WHOLE image without its tables (slgc_tune "image_rows": the context holds one band of a taller image, the
// decision must be the one a single GPU scanning the whole image takes): nodes and exact rays are evaluated on the spot with the routine
// that fills the tables, so the number is the one k_check_cam_nodes reports on the whole image's tables.  g = linear 4-pixel group of the image.
__global__ void __launch_bounds__(256) k_check_cam_nodes_direct(const Calib c, int w4, size_t ngroups, unsigned *__restrict__ worst)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    float err = 0.0f;
    if (g < ngroups) {
        const int row = (int)(g / (size_t)w4), gx = (int)(g % (size_t)w4);
        float2 n[4];
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const Ray2 a = undistort_point((float)(4 * (gx + k - 1)), (float)row, c.cam_k, c.cam_d, c.R);
            n[k] = make_float2(a.x, a.y);
        }
        float cx[4], cy[4];
        cam_rays_from_nodes(cam_v4f{n[0].x, n[0].y, n[1].x, n[1].y}, cam_v4f{n[2].x, n[2].y, n[3].x, n[3].y}, cx, cy);
#pragma unroll 1
        for (int j = 1; j < 4; ++j) {                    // pixel 4 gx is node 1 itself
            const Ray2 a = undistort_point((float)(4 * gx + j), (float)row, c.cam_k, c.cam_d, c.R);
            const float ix = j == 1 ? cx[1] : j == 2 ? cx[2] : cx[3], iy = j == 1 ? cy[1] : j == 2 ? cy[2] : cy[3];
            const float ax = fabsf(ix - a.x), ay = fabsf(iy - a.y);
            const float ex = fmaxf(ax / fmaxf(1.0f, fabsf(a.x)), (2.4e-7f / 4e-6f) * ax / fmaxf(kCamNodeTiny, fabsf(a.x)));
            const float ey = fmaxf(ay / fmaxf(1.0f, fabsf(a.y)), (2.4e-7f / 4e-6f) * ay / fmaxf(kCamNodeTiny, fabsf(a.y)));
            err = fmaxf(err, !(ex == ex) || !(ey == ey) ? __builtin_huge_valf() : fmaxf(ex, ey));
        }
        if (!(n[1].x == n[1].x) || !(n[1].y == n[1].y)) err = __builtin_huge_valf();      // k_check_cam_nodes sees a NaN node through pixel 4 gx
    }
    for (int o = 32; o > 0; o >>= 1) err = fmaxf(err, __shfl_down(err, o, 64));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(err) > *reinterpret_cast<volatile unsigned *>(worst)) atomicMax(worst, __float_as_uint(err));
}

// Both parts of the projector table, same tiled index: lut = the float32 rays cv2.undistortPoints returns (exact kernels, the guarded redo),
// lut_th = tan(beta / 2) of each ray against the baseline, ONE float32 per projector pixel (tri_math.h: what the fast form gathers).
__global__ void __launch_bounds__(256) k_build_proj_lut(const Calib c, float2 *__restrict__ lut, float *__restrict__ lut_th, int proj_w, int proj_h, int tiles_x,
                                                        size_t nslots, int wide)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= nslots) return;
    const int sh = wide ? 7 : 6, tw = wide ? 16 : 8;                        // slots per tile = 8 rows x tw pixels (proj_lut_index)
    const int tile = (int)(p >> sh), in = (int)(p & ((1u << sh) - 1u));
    const int pu = (tile % tiles_x) * tw + (in & (tw - 1)), pv = (tile / tiles_x) * 8 + (in / tw);
    float2 o = make_float2(0.f, 0.f);
    float th = 1.0f;
    if (pu < proj_w && pv < proj_h) {
        const Ray2 b = undistort_point((float)pu, (float)pv, c.proj_k, c.proj_d, nullptr);
        o = make_float2(b.x, b.y);
        th = proj_tan_half(b.x, b.y, c.T, c.t_len);
    }
    lut[p] = o;
    lut_th[p] = th;
}

struct TriConst {
    double T[3];
    double t_len;
    TriF32 kf;               // T / |T| and |T| in float32 for the fast form (tri_math.h)
};

// Dense kernel.  Four pixels per lane (8-byte map loads, 32-byte ray loads).  The per-lane "4 consecutive pixels" layout is ideal for the streamed loads but makes
// each projector-table gather instruction touch ~50 cache lines and each XYZ store instruction a 48-byte-strided
// scatter.  Here the workgroup (1024 pixels) exchanges through LDS instead:
//   1. every lane turns its 4 decoded pixels into table indices            -> s_idx  (4 KB)
//   2. gathers run pixel-per-lane (64 neighbouring pixels per instruction) -> s_ray  (8 KB)
//   3. every lane triangulates its own 4 pixels                            -> s_xyz  (12 KB, aliases s_ray)
//   4. the 12 KB of XYZ leave as wave-contiguous 1 KB stores.
// wire != nullptr: the maps arrive in the 3-byte wire format of the multi-GPU exchange (wire.hip); the kernel unpacks them and
// also writes the int16 maps (h, v are outputs then), so the exchange needs no separate unpack pass.
template <int MODE>
__global__ void __launch_bounds__(256) k_triangulate_maps_lds(const TriConst tc, int16_t *__restrict__ h, int16_t *__restrict__ v,
                                                              const uint32_t *__restrict__ wire, const float2 *__restrict__ cam_lut,
                                                              const float2 *__restrict__ proj_lut, const float *__restrict__ proj_th, size_t ngroups, int proj_w,
                                                              int proj_h, int tiles_x, float *__restrict__ xyz,
                                                              unsigned long long *__restrict__ count, uint32_t xcd_chunk, int nt_store, int wide,
                                                              const CamNodes cn)
{
    __shared__ uint4 s_idx[256];
    __shared__ float4 s_buf[768];
    const int tid = threadIdx.x;
    const uint32_t bid = xcd_block(blockIdx.x, xcd_chunk);
    const size_t g = (size_t)bid * 256 + tid;
    const bool live = g < ngroups;
    uint32_t idx[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    float4 c01 = make_float4(0.f, 0.f, 0.f, 0.f), c23 = c01;
    if (live) {
        uint2 hw, vw;
        if (wire) {
            unpack_hv24_x4(wire[3 * g], wire[3 * g + 1], wire[3 * g + 2], hw, vw);
            reinterpret_cast<uint2 *>(h)[g] = hw;
            reinterpret_cast<uint2 *>(v)[g] = vw;
        } else {
            hw = reinterpret_cast<const uint2 *>(h)[g];
            vw = reinterpret_cast<const uint2 *>(v)[g];
        }
        if (cn.nodes) {                                    // four nodes around the group instead of the group's four rays
            const float2 *e = cam_node_ptr(cn, (uint32_t)g);
            const cam_v4f n01 = *reinterpret_cast<const cam_v4f *>(e), n23 = *reinterpret_cast<const cam_v4f *>(e + 2);
            c01 = make_float4(n01.x, n01.y, n01.z, n01.w);
            c23 = make_float4(n23.x, n23.y, n23.z, n23.w);
        } else {
            c01 = reinterpret_cast<const float4 *>(cam_lut)[2 * g];
            c23 = reinterpret_cast<const float4 *>(cam_lut)[2 * g + 1];
        }
        const unsigned hq[2] = {hw.x, hw.y}, vq[2] = {vw.x, vw.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hv = (int)(short)(hq[j >> 1] >> (16 * (j & 1))), vv = (int)(short)(vq[j >> 1] >> (16 * (j & 1)));
            if (!(hv == -1 || vv == -1))                                                              // triangulate.py:56
                idx[j] = proj_lut_index(min(proj_w - 1, hv), min(proj_h - 1, vv), tiles_x, wide);    // :60-61
        }
    }
    s_idx[tid] = make_uint4(idx[0], idx[1], idx[2], idx[3]);
    __syncthreads();
    const uint32_t *s_idx1 = reinterpret_cast<const uint32_t *>(s_idx);
    // four independent, unconditional gathers in flight per lane (an undecodable pixel reads entry 0, unused).  Exact mode: the float32 rays
    // (8 bytes each); fast form: tan(beta / 2) of each ray against the baseline (4 bytes each)
    float px[4], py[4];
    if constexpr (MODE == SLGC_TRI_EXACT) {
        float2 *s_ray = reinterpret_cast<float2 *>(s_buf);
        float2 gr[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t i = s_idx1[it * 256 + tid];
            gr[it] = proj_lut[i != 0xffffffffu ? i : 0u];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) s_ray[it * 256 + tid] = gr[it];
        __syncthreads();
        const float4 r01 = s_buf[2 * tid], r23 = s_buf[2 * tid + 1];
        px[0] = r01.x, px[1] = r01.z, px[2] = r23.x, px[3] = r23.z;
        py[0] = r01.y, py[1] = r01.w, py[2] = r23.y, py[3] = r23.w;
    } else {
        float *s_th = reinterpret_cast<float *>(s_buf);
        float gt[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t i = s_idx1[it * 256 + tid];
            gt[it] = proj_th[i != 0xffffffffu ? i : 0u];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) s_th[it * 256 + tid] = gt[it];
        __syncthreads();
        const float4 t4 = s_buf[tid];
        px[0] = t4.x, px[1] = t4.y, px[2] = t4.z, px[3] = t4.w;
        py[0] = py[1] = py[2] = py[3] = 0.f;
    }
    __syncthreads();
    float out[12];
    const uint32_t valid = (idx[0] != 0xffffffffu ? 1u : 0u) | (idx[1] != 0xffffffffu ? 2u : 0u) | (idx[2] != 0xffffffffu ? 4u : 0u) |
                           (idx[3] != 0xffffffffu ? 8u : 0u);
    const unsigned nvalid = __builtin_popcount(valid);
    float cx[4] = {c01.x, c01.z, c23.x, c23.z}, cy[4] = {c01.y, c01.w, c23.y, c23.w};
    if (cn.nodes) {
        cam_rays_from_nodes(cam_v4f{c01.x, c01.y, c01.z, c01.w}, cam_v4f{c23.x, c23.y, c23.z, c23.w}, cx, cy);
        if (live) cam_rays_exact_where_tiny(cx, cy, cam_lut + 4 * g);
    }
    if constexpr (MODE == SLGC_TRI_EXACT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float X = __builtin_nanf(""), Y = X, Z = X;
            if ((valid >> j) & 1u) {
                const Xyz r = law_of_sines<SLGC_TRI_EXACT>(Ray2{cx[j], cy[j]}, Ray2{px[j], py[j]}, tc.T, tc.t_len);
                X = (float)r.x; Y = (float)r.y; Z = (float)r.z;
            }
            out[3 * j] = X; out[3 * j + 1] = Y; out[3 * j + 2] = Z;
        }
    }
    uint32_t ill = 0;
    if constexpr (MODE != SLGC_TRI_EXACT) {
        ill = triangulate4_flag(cx, cy, px, valid, tc.kf, out);          // px holds the gathered tan(beta / 2)
        if (MODE == 2) ill = 0;                                          // MODE 2: unguarded (A/B, diagnostic build)
    }
    stage_xyz12(s_buf + 3 * tid, out);
    if constexpr (MODE != SLGC_TRI_EXACT) {
        // Flat triangles (tri_math.h) are redone in float64 on the reference's float32 intermediates.  Like the fused scan kernel this one compacts
        // them first -- here over the whole WORKGROUP, which exchanges through LDS anyway: ballots + per-wave counts rank the flagged pixels of the
        // 1024 into one list, and thread k redoes list entry k (a lane-level loop walked the float64 path once per flagged POSITION of each wave:
        // scattered wrong codes -- S-uniform -- cost the dense kernel 80 us instead of 45 at 4096x3000).
        __shared__ unsigned short s_list[1024];
        __shared__ unsigned s_cnt[4];
        const int wave = tid >> 6, lane = tid & 63;
        unsigned long long m[4];
        unsigned nw = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = __builtin_amdgcn_ballot_w64((ill >> j) & 1u);
            nw += (unsigned)__builtin_popcountll(m[j]);
        }
        if (lane == 0) s_cnt[wave] = nw;
        __syncthreads();
        const unsigned c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], total = c0 + c1 + c2 + s_cnt[3];
        if (total) {                                                     // workgroup-uniform
            unsigned at = wave == 0 ? 0u : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((ill >> j) & 1u) s_list[at + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[j], 0u))] = (unsigned short)(4 * tid + j);
                at += (unsigned)__builtin_popcountll(m[j]);
            }
            __syncthreads();
            float *s_xyz = reinterpret_cast<float *>(s_buf);
            const float2 *cam_wg = cam_lut + 4 * ((size_t)bid * 256);    // the workgroup's first pixel in the exact per-pixel table
            // up to 4 entries per thread (1 024 pixels): the exact rays of all of them are requested before the first one computes (one memory round
            // trip per workgroup, not one per pass)
            const unsigned npass = (total + 255u) >> 8;                  // workgroup-uniform
            unsigned pq[4];
            float2 crq[4], prq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if ((unsigned)q < npass) {
                    pq[q] = s_list[min(256u * q + (unsigned)tid, total - 1u)];
                    crq[q] = cam_wg[pq[q]];
                    prq[q] = proj_lut[s_idx1[pq[q]]];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if ((unsigned)q < npass && 256u * q + (unsigned)tid < total) {
                    const Xyzf r = law_of_sines_mirror(Ray2{crq[q].x, crq[q].y}, Ray2{prq[q].x, prq[q].y}, tc.T, tc.t_len);
                    s_xyz[3 * pq[q]] = r.x;
                    s_xyz[3 * pq[q] + 1] = r.y;
                    s_xyz[3 * pq[q] + 2] = r.z;
                }
            }
        }
    }
    __syncthreads();
    const size_t first = (size_t)bid * 256;
    const size_t nvec = (ngroups - first < 256 ? ngroups - first : 256) * 3;      // float4s this workgroup owns
    float4 *dst = reinterpret_cast<float4 *>(xyz) + first * 3;
#pragma unroll
    for (int it = 0; it < 3; ++it)
        if ((size_t)(it * 256 + tid) < nvec) {
            if (nt_store & 1) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const float4 q = s_buf[it * 256 + tid];
                __builtin_nontemporal_store(v4f{q.x, q.y, q.z, q.w}, reinterpret_cast<v4f *>(dst) + it * 256 + tid);
            } else {
                dst[it * 256 + tid] = s_buf[it * 256 + tid];
            }
        }
    if (count) block_count_add(count, nvalid);
}

// cv2.undistortPoints for a point list, as the two call sites of triangulate.py use it: which = 0 the camera call (:84, R = proj_R),
// which = 1 the projector call (:85, no R).  float32 [M][2] in, float32 [M][2] out (OpenCV returns CV_32FC2 for float32 input).
__global__ void __launch_bounds__(256) k_undistort_list(const Calib c, int which, const float *__restrict__ pts, size_t M, float *__restrict__ out)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= M) return;
    const float2 p = reinterpret_cast<const float2 *>(pts)[q];
    const Ray2 r = which == 0 ? undistort_point(p.x, p.y, c.cam_k, c.cam_d, c.R) : undistort_point(p.x, p.y, c.proj_k, c.proj_d, nullptr);
    reinterpret_cast<float2 *>(out)[q] = make_float2(r.x, r.y);
}

// Diagnostic: how many decodable pixels of a band take the guarded (float32-mirror) path of triangulate4 -- the same
// tri_is_flat test on the same table rays.  counts[0] += decodable pixels, counts[1] += flagged pixels.
__global__ void __launch_bounds__(256) k_guard_count(const TriConst tc, const int16_t *__restrict__ h, const int16_t *__restrict__ v,
                                                     const float2 *__restrict__ cam_lut, const float *__restrict__ proj_th, size_t npix,
                                                     int proj_w, int proj_h, int tiles_x, int wide, unsigned long long *__restrict__ counts)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned ok = 0, flat = 0;
    if (p < npix) {
        const int hv = h[p], vv = v[p];
        if (!(hv == -1 || vv == -1)) {
            ok = 1;
            const float2 c = cam_lut[p];
            const float q = proj_th[proj_lut_index(min(proj_w - 1, hv), min(proj_h - 1, vv), tiles_x, wide)];
            flat = tri_is_flat(c.x, c.y, q, tc.kf) ? 1u : 0u;
        }
    }
    unsigned packed = ok | (flat << 16);                       // 64 lanes: both sums fit 16 bits
    for (int o = 32; o > 0; o >>= 1) packed += __shfl_down(packed, o, 64);
    if ((threadIdx.x & 63) == 0 && packed) {
        atomicAdd(counts, (unsigned long long)(packed & 0xffffu));
        if (packed >> 16) atomicAdd(counts + 1, (unsigned long long)(packed >> 16));
    }
}

}  // namespace

int launch_undistort_list(slgc_ctx *ctx, int which, const float *d_pts, int64_t M, float *d_out)
{
    if (M == 0) return SLGC_OK;
    hipLaunchKernelGGL(k_undistort_list, dim3((unsigned)(((size_t)M + 255) / 256)), dim3(256), 0, ctx->stream, ctx->calib, which, d_pts, (size_t)M, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_guard_count(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, int rows, int W, int row0, int proj_w, int proj_h,
                       unsigned long long *d_counts)
{
    const size_t npix = (size_t)rows * W;
    if (npix == 0) return SLGC_OK;
    int rc = ensure_luts(ctx, rows, W, row0, proj_w, proj_h);
    if (rc) return rc;
    TriConst tc;
    memcpy(tc.T, ctx->calib.T, sizeof tc.T);
    tc.t_len = ctx->calib.t_len;
    tc.kf = make_tri_f32(ctx->calib.T, ctx->calib.t_len);
    hipLaunchKernelGGL(k_guard_count, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, tc, d_h, d_v, (const float2 *)ctx->lut_cam,
                       (const float *)ctx->lut_proj_th, npix, proj_w, proj_h, proj_tiles_x(ctx, proj_w), ctx->tune_proj_tile, d_counts);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

int launch_triangulate_list(slgc_ctx *ctx, const float *d_cam, const float *d_proj, int64_t M, int mode, double *d_xyz)
{
    if (M == 0) return SLGC_OK;
    const unsigned blocks = (unsigned)(((size_t)M + 255) / 256);
    if (mode == SLGC_TRI_EXACT)
        hipLaunchKernelGGL(k_triangulate_list<SLGC_TRI_EXACT>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->calib, d_cam, d_proj, (size_t)M, d_xyz);
    else
        hipLaunchKernelGGL(k_triangulate_list<SLGC_TRI_ALGEBRAIC>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->calib, d_cam, d_proj, (size_t)M, d_xyz);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

// Build (or reuse) the ray tables for this calibration / geometry.
// Asynchronous on the context's stream EXCEPT when a node table is (re)built: its accuracy check reads one word back (a stream
// synchronisation, once per calibration / geometry).  A node table that cannot be allocated or does not pass the check simply is not
// there (the kernels read the per-pixel table); the state describing a table is written only after its build has been enqueued.
int ensure_luts(slgc_ctx *ctx, int rows, int W, int row0, int proj_w, int proj_h)
{
    const int wide = ctx->tune_proj_tile ? 1 : 0;
    const int tiles_x = proj_tiles_x(ctx, proj_w), tiles_y = (proj_h + 7) / 8;
    const size_t npix = (size_t)rows * W, nproj = (size_t)tiles_x * tiles_y * (wide ? 128 : 64);
    // the whole image this band belongs to (slgc_tune "image_rows"); a band that is the image itself needs no separate treatment
    // (a band that does not fit into that image is not a band of it: the setting does not apply)
    const int img_rows = (ctx->tune_image_rows > 0 && !(row0 == 0 && rows == ctx->tune_image_rows) && row0 >= 0 && row0 + rows <= ctx->tune_image_rows)
                             ? ctx->tune_image_rows : 0;
    if (!(ctx->lut_cam && ctx->lut_cam_ver == ctx->calib_ver && ctx->lut_cam_W == W && ctx->lut_cam_row0 == row0 && ctx->lut_cam_rows == rows &&
          ctx->lut_image_rows == img_rows)) {
        if (ctx->lut_cam || ctx->lut_nodes) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->lut_cam) (void)hipFree(ctx->lut_cam);
            if (ctx->lut_nodes) (void)hipFree(ctx->lut_nodes);
            ctx->lut_cam = ctx->lut_nodes = nullptr;
        }
        ctx->lut_nodes_err = -1.0f;
        void *cam = nullptr;
        if (hipMalloc(&cam, npix * sizeof(float2) + 64) != hipSuccess) return slgc_fail(ctx, SLGC_ENOMEM, "camera ray table");
        if (npix) hipLaunchKernelGGL(k_build_cam_lut, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, ctx->calib, (float2 *)cam, W, row0, npix);
        if (hipGetLastError() != hipSuccess) {
            (void)hipFree(cam);
            return slgc_fail(ctx, SLGC_EHIP, "camera ray table build failed to launch");
        }
        ctx->lut_cam = cam;
        ctx->lut_cam_ver = ctx->calib_ver; ctx->lut_cam_W = W; ctx->lut_cam_row0 = row0; ctx->lut_cam_rows = rows; ctx->lut_image_rows = img_rows;
        // the every-4th-column table of the same band, kept only if the rays interpolated from it stay within 2 ulp of the exact ones --
        // over the band, or (image_rows) over the whole image the band belongs to, so that every band of one image decides alike
        const size_t img_groups = (size_t)(img_rows ? img_rows : rows) * (size_t)(W / 4);
        if (W % 4 == 0 && W >= 4 && npix >= 4 && img_groups < (1u << 24)) {
            const int ne = W / 4 + 3;
            const size_t total = (size_t)rows * ne;
            void *nodes = nullptr;
            if (!ctx->lut_check_word && hipMalloc(&ctx->lut_check_word, 64) != hipSuccess) ctx->lut_check_word = nullptr;
            if (ctx->lut_check_word && hipMalloc(&nodes, total * sizeof(float2) + 64) == hipSuccess) {
                unsigned *worst = (unsigned *)ctx->lut_check_word;
                hipLaunchKernelGGL(k_build_cam_nodes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->calib, (float2 *)nodes, ne, row0, total);
                float err = __builtin_huge_valf();
                bool measured = false;
                if (img_rows && ctx->img_err_ver == ctx->calib_ver && ctx->img_err_W == W && ctx->img_err_rows == img_rows) {
                    err = ctx->img_err;                  // another band of the same image measured it already
                    measured = true;
                } else {
                    bool ok = hipMemsetAsync(worst, 0, 4, ctx->stream) == hipSuccess;
                    if (ok && img_rows) {
                        hipLaunchKernelGGL(k_check_cam_nodes_direct, dim3((unsigned)((img_groups + 255) / 256)), dim3(256), 0, ctx->stream, ctx->calib, W / 4, img_groups, worst);
                    } else if (ok) {
                        const CamNodes cn{(const float2 *)nodes, (uint32_t)(W / 4), (uint32_t)ne, 1.0f / (float)(W / 4)};
                        hipLaunchKernelGGL(k_check_cam_nodes, dim3((unsigned)((npix / 4 + 255) / 256)), dim3(256), 0, ctx->stream, (const float2 *)ctx->lut_cam, cn, npix / 4, worst);
                    }
                    ok = ok && hipGetLastError() == hipSuccess && hipMemcpyAsync(&err, worst, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                         hipStreamSynchronize(ctx->stream) == hipSuccess;
                    if (!ok) {
                        (void)hipGetLastError();
                        err = __builtin_huge_valf();
                    } else {
                        measured = true;
                        if (img_rows) { ctx->img_err = err; ctx->img_err_ver = ctx->calib_ver; ctx->img_err_W = W; ctx->img_err_rows = img_rows; }
                    }
                }
                if (measured) ctx->lut_nodes_err = err;
                if (err <= 2.4e-7f) {                     // 2 ulp of a float32 in [1, 2); otherwise this lens is not smooth enough at 4-pixel nodes
                    ctx->lut_nodes = nodes;
                } else {
                    (void)hipStreamSynchronize(ctx->stream);
                    (void)hipFree(nodes);
                }
            } else {
                (void)hipGetLastError();                  // no memory for the node table: the kernels read the per-pixel one
            }
        }
    }
    if (!(ctx->lut_proj && ctx->lut_proj_ver == ctx->calib_ver && ctx->lut_proj_w == proj_w && ctx->lut_proj_h == proj_h && ctx->lut_proj_tile == wide)) {
        if (ctx->lut_proj) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->lut_proj);
            ctx->lut_proj = nullptr;
        }
        void *proj = nullptr;                                  // one allocation: [rays (float2) | tan(beta / 2) (float)], nproj entries each (nproj is a multiple of 64)
        if (hipMalloc(&proj, nproj * (sizeof(float2) + sizeof(float)) + 64) != hipSuccess) return slgc_fail(ctx, SLGC_ENOMEM, "projector ray table");
        hipLaunchKernelGGL(k_build_proj_lut, dim3((unsigned)((nproj + 255) / 256)), dim3(256), 0, ctx->stream, ctx->calib, (float2 *)proj, (float *)((float2 *)proj + nproj), proj_w,
                           proj_h, tiles_x, nproj, wide);
        if (hipGetLastError() != hipSuccess) {
            (void)hipFree(proj);
            return slgc_fail(ctx, SLGC_EHIP, "projector ray table build failed to launch");
        }
        ctx->lut_proj = proj;
        ctx->lut_proj_th = (float2 *)proj + nproj;
        ctx->lut_proj_ver = ctx->calib_ver; ctx->lut_proj_w = proj_w; ctx->lut_proj_h = proj_h; ctx->lut_proj_tile = wide;
    }
    return SLGC_OK;
}

int launch_triangulate_maps(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, int rows, int W, int row0, int proj_w,
                            int proj_h, int mode, float *d_xyz, unsigned long long *d_count, const uint8_t *d_wire)
{
    const size_t npix = (size_t)rows * W;
    ctx->last_tri_ragged = 0;
    if (npix == 0) return SLGC_OK;
    if (proj_w < 1 || proj_h < 1) return slgc_fail(ctx, SLGC_EINVAL, "bad projector size");
    unsigned long long *slots = nullptr;
    if (d_count) {
        if (!ctx->count_slots) {
            HIP_TRY(ctx, hipMalloc(&ctx->count_slots, (size_t)kCountSlots * kCountStride * 8));
            HIP_TRY(ctx, hipMemsetAsync(ctx->count_slots, 0, (size_t)kCountSlots * kCountStride * 8, ctx->stream));
        }
        slots = (unsigned long long *)ctx->count_slots;
    }
    const bool direct = (mode & SLGC_TRI_DIRECT) != 0;
    mode &= 1;
    int rc = launch_triangulate_maps_body(ctx, d_h, d_v, npix, W, row0, rows, proj_w, proj_h, mode, direct, d_xyz, slots, d_wire);
    if (rc) return rc;
    if (d_count) {
        hipLaunchKernelGGL(k_count_finish, dim3(1), dim3(256), 0, ctx->stream, slots, d_count);
        HIP_TRY(ctx, hipGetLastError());
    }
    return SLGC_OK;
}

static int launch_triangulate_maps_body(slgc_ctx *ctx, const int16_t *d_h_in, const int16_t *d_v_in, size_t npix, int W, int row0, int rows,
                                        int proj_w, int proj_h, int mode, bool direct, float *d_xyz, unsigned long long *d_count,
                                        const uint8_t *d_wire)
{
    int16_t *d_h = const_cast<int16_t *>(d_h_in), *d_v = const_cast<int16_t *>(d_v_in);      // written only when d_wire is given
    const bool vec_ok = !direct && (((uintptr_t)d_h | (uintptr_t)d_v) % 8 == 0) && ((uintptr_t)d_xyz % 16 == 0) && (size_t)proj_w * proj_h < (1u << 28) &&
                        (uintptr_t)d_wire % 4 == 0;
    if (d_wire && !(vec_ok && npix >= 4 && npix % 4 == 0)) {       // odd shapes: unpack first, then the ordinary path
        int rc = launch_unpack_hv24(ctx, d_wire, npix, d_h, d_v);
        if (rc) return rc;
        d_wire = nullptr;
    }
    const uint32_t *d_wire32 = reinterpret_cast<const uint32_t *>(d_wire);
    if (vec_ok && npix >= 4) {
        int rc = ensure_luts(ctx, rows, W, row0, proj_w, proj_h);
        if (rc) return rc;
        TriConst tc;
        memcpy(tc.T, ctx->calib.T, sizeof tc.T);
        tc.t_len = ctx->calib.t_len;
        tc.kf = make_tri_f32(ctx->calib.T, ctx->calib.t_len);
        const size_t groups = npix / 4;
        const unsigned blocks = (unsigned)((groups + 255) / 256);
        const int tri_nt = ctx->tune_tri_nt;                     // XYZ leaves with non-temporal stores (A/B: slgc_tune "tri_nt")
        // the exact float64 mode keeps the exact per-pixel rays; the fast form may interpolate them from the node table
        const CamNodes cn = SLGC_CAM_NODES_FOR(ctx, W, mode != SLGC_TRI_EXACT);
        ctx->last_nodes = cn.nodes ? 1 : 0;
        ctx->last_guard = mode == SLGC_TRI_EXACT ? 0 : 1;
        if (mode == SLGC_TRI_EXACT)
            hipLaunchKernelGGL(k_triangulate_maps_lds<SLGC_TRI_EXACT>, dim3(blocks), dim3(256), 0, ctx->stream, tc, d_h, d_v, d_wire32,
                               (const float2 *)ctx->lut_cam, (const float2 *)ctx->lut_proj, (const float *)ctx->lut_proj_th, groups, proj_w, proj_h, proj_tiles_x(ctx, proj_w), d_xyz, d_count, xcd_chunk_for(ctx, blocks), tri_nt, ctx->tune_proj_tile, cn);
#ifdef SLGC_DIAG      // A/B of the guard's cost: only in the diagnostic build
        else if (xcd_env("SLGC_TRI_UNGUARDED", 0))
            hipLaunchKernelGGL(k_triangulate_maps_lds<2>, dim3(blocks), dim3(256), 0, ctx->stream, tc, d_h, d_v, d_wire32,
                               (const float2 *)ctx->lut_cam, (const float2 *)ctx->lut_proj, (const float *)ctx->lut_proj_th, groups, proj_w, proj_h, proj_tiles_x(ctx, proj_w), d_xyz, d_count, xcd_chunk_for(ctx, blocks), tri_nt, ctx->tune_proj_tile, cn);
#endif
        else
            hipLaunchKernelGGL(k_triangulate_maps_lds<SLGC_TRI_ALGEBRAIC>, dim3(blocks), dim3(256), 0, ctx->stream, tc, d_h, d_v, d_wire32,
                               (const float2 *)ctx->lut_cam, (const float2 *)ctx->lut_proj, (const float *)ctx->lut_proj_th, groups, proj_w, proj_h, proj_tiles_x(ctx, proj_w), d_xyz, d_count, xcd_chunk_for(ctx, blocks), tri_nt, ctx->tune_proj_tile, cn);
        HIP_TRY(ctx, hipGetLastError());
        const size_t done = groups * 4;
        if (done == npix) return SLGC_OK;
        ctx->last_tri_ragged = 1;
        // ragged tail (< 4 pixels): direct-evaluation kernel on the remainder
        const int x0 = (int)(done % (size_t)W);
        (void)x0;
        return launch_triangulate_maps_direct(ctx, d_h + done, d_v + done, npix - done, done, W, row0, proj_w, proj_h, mode, d_xyz + 3 * done, d_count);
    }
    ctx->last_nodes = 0;
    ctx->last_guard = mode == SLGC_TRI_EXACT ? 0 : 1;
    ctx->last_tri_ragged = 1;
    return launch_triangulate_maps_direct(ctx, d_h, d_v, npix, 0, W, row0, proj_w, proj_h, mode, d_xyz, d_count);
}

// Direct evaluation (no tables): used for misaligned buffers and ragged tails.  first = linear index of d_h[0] in the band.
int launch_triangulate_maps_direct(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, size_t npix, size_t first, int W, int row0,
                                   int proj_w, int proj_h, int mode, float *d_xyz, unsigned long long *d_count)
{
    if (npix == 0) return SLGC_OK;
    const unsigned blocks = (unsigned)((npix + 255) / 256);
    if (mode == SLGC_TRI_EXACT)
        hipLaunchKernelGGL(k_triangulate_maps<SLGC_TRI_EXACT>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->calib, d_h, d_v, npix, first, W, row0,
                           proj_w, proj_h, d_xyz, d_count);
    else
        hipLaunchKernelGGL(k_triangulate_maps<SLGC_TRI_ALGEBRAIC>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->calib, d_h, d_v, npix, first, W,
                           row0, proj_w, proj_h, d_xyz, d_count);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}
