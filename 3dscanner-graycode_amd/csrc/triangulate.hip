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
#include "slgc_internal.h"

namespace {

struct Ray2 {
    float x, y;
};

// cv::undistortPoints(src, K, dist, R) for one point, default TermCriteria(MAX_ITER, 5, 0.01).
__device__ __forceinline__ Ray2 undistort_point(float uf, float vf, const double (&kk)[4], const double (&k)[12], const double *R)
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

struct Xyz {
    double x, y, z;
};

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
                                                          const int16_t *__restrict__ v, size_t npix,
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
            const int x = (int)(p % (size_t)W), y = row0 + (int)(p / (size_t)W);
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
    const unsigned long long m = __ballot(ok);
    if (count && (threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
    __syncthreads();
    const size_t nfl = min((size_t)768, (npix - base) * 3);
    for (size_t i = threadIdx.x; i < nfl; i += 256) xyz[base * 3 + i] = stage[i];
}

}  // namespace

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

int launch_triangulate_maps(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, int rows, int W, int row0, int proj_w,
                            int proj_h, int mode, float *d_xyz, unsigned long long *d_count)
{
    const size_t npix = (size_t)rows * W;
    if (npix == 0) return SLGC_OK;
    const unsigned blocks = (unsigned)((npix + 255) / 256);
    if (mode == SLGC_TRI_EXACT)
        hipLaunchKernelGGL(k_triangulate_maps<SLGC_TRI_EXACT>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->calib, d_h, d_v, npix, W, row0, proj_w,
                           proj_h, d_xyz, d_count);
    else
        hipLaunchKernelGGL(k_triangulate_maps<SLGC_TRI_ALGEBRAIC>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->calib, d_h, d_v, npix, W, row0,
                           proj_w, proj_h, d_xyz, d_count);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}
