// list_linear.hip -- premise check for an x-major list build that writes LINEARLY (maps / white image / camera rays transposed first, then a plain
// stream compaction in x-major scan order): how fast do the reference-shaped products leave when record k goes to position k of each stream?
//   hipcc --offload-arch=gfx950 -O3 -o list_linear list_linear.hip && ./list_linear
// streams per record: cam_pts float2, proj_pts float2, Pts double x3 (three rows M apart), colors double[3]  = 64 bytes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one record per thread, every stream written at index k (what a compaction with ~100 % valid pixels does)
__global__ void __launch_bounds__(256) k_records(size_t M, float2 *cam, float2 *proj, double *pts, double *colors)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= M) return;
    const float f = (float)(k & 1023);
    cam[k] = make_float2(f, f + 1.f);
    proj[k] = make_float2(f + 2.f, f + 3.f);
    pts[k] = f * 0.5;
    pts[M + k] = f * 0.25;
    pts[2 * M + k] = f * 0.125;
    colors[3 * k] = f;
    colors[3 * k + 1] = f + 1.0;
    colors[3 * k + 2] = f + 2.0;
}
// the colour triple staged so that a wave writes 1.5 KB contiguously in three 512-byte instructions (lane-contiguous doubles)
__global__ void __launch_bounds__(256) k_records_staged(size_t M, float2 *cam, float2 *proj, double *pts, double *colors)
{
    __shared__ double s_col[256 * 3];
    const size_t k0 = (size_t)blockIdx.x * 256, k = k0 + threadIdx.x;
    const float f = (float)(k & 1023);
    if (k < M) {
        cam[k] = make_float2(f, f + 1.f);
        proj[k] = make_float2(f + 2.f, f + 3.f);
        pts[k] = f * 0.5;
        pts[M + k] = f * 0.25;
        pts[2 * M + k] = f * 0.125;
    }
    s_col[3 * threadIdx.x] = f;
    s_col[3 * threadIdx.x + 1] = f + 1.0;
    s_col[3 * threadIdx.x + 2] = f + 2.0;
    __syncthreads();
    for (int q = 0; q < 3; ++q) {
        const size_t o = 3 * k0 + (size_t)q * 256 + threadIdx.x;
        if (o < 3 * M) colors[o] = s_col[q * 256 + threadIdx.x];
    }
}
// the linear records again, every record first READING 16 + 4 bytes of inputs (camera ray, maps, colour bytes: what the list build cannot avoid):
// the mix is what the memory system sees, not a pure fill
__global__ void __launch_bounds__(256) k_records_reading(size_t M, const float4 *in16, const unsigned *in4, float2 *cam, float2 *proj, double *pts, double *colors)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= M) return;
    const float4 a = in16[k];
    const unsigned b = in4[k];
    cam[k] = make_float2(a.x, a.y);
    proj[k] = make_float2(a.z, a.w);
    pts[k] = a.x * 0.5;
    pts[M + k] = a.y * 0.25;
    pts[2 * M + k] = a.z * 0.125;
    colors[3 * k] = (double)(b & 0xffu);
    colors[3 * k + 1] = (double)((b >> 8) & 0xffu);
    colors[3 * k + 2] = (double)(b >> 16);
}
// tiled transpose of a 2-byte map pair (h, v interleaved as 4 bytes per pixel): [H][W] -> [W][H]
__global__ void __launch_bounds__(256) k_transpose4(const unsigned *in, unsigned *out, int W, int H)
{
    __shared__ unsigned t[64][65];
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64, lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int r = ly; r < 64; r += 4)
        if (by + r < H && bx + lx < W) t[r][lx] = in[(size_t)(by + r) * W + bx + lx];
    __syncthreads();
    for (int r = ly; r < 64; r += 4)
        if (bx + r < W && by + lx < H) out[(size_t)(bx + r) * H + by + lx] = t[lx][r];
}

int main()
{
    const int W = 4096, H = 3000;
    const size_t M = 9924736;
    float2 *cam, *proj;
    double *pts, *colors;
    unsigned *a, *b;
    CK(hipMalloc(&cam, M * 8)); CK(hipMalloc(&proj, M * 8)); CK(hipMalloc(&pts, M * 24)); CK(hipMalloc(&colors, M * 24));
    CK(hipMalloc(&a, (size_t)W * H * 4)); CK(hipMalloc(&b, (size_t)W * H * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double bytes, auto &&fn) {
        for (int w = 0; w < 3; ++w) fn();
        CK(hipDeviceSynchronize());
        std::vector<float> ms;
        for (int r = 0; r < 9; ++r) {
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < 4; ++k) fn();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t / 4);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-78s %8.1f us  %5.2f TB/s\n", name, ms[4] * 1e3, bytes / (ms[4] * 1e-3) / 1e12);
    };
    timeit("records written linearly, 64 B each in 8 store instructions per thread", M * 64.0, [&] { hipLaunchKernelGGL(k_records, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0, M, cam, proj, pts, colors); });
    timeit("the same, colours staged through LDS (lane-contiguous doubles)", M * 64.0, [&] { hipLaunchKernelGGL(k_records_staged, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0, M, cam, proj, pts, colors); });
    float4 *in16;
    unsigned *in4;
    CK(hipMalloc(&in16, M * 16)); CK(hipMalloc(&in4, M * 4));
    CK(hipMemset(in16, 0, M * 16)); CK(hipMemset(in4, 0, M * 4));
    timeit("records written linearly, each reading 20 B of inputs first (84 B of traffic)", M * 84.0, [&] { hipLaunchKernelGGL(k_records_reading, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0, M, in16, in4, cam, proj, pts, colors); });
    timeit("transpose of 4 B/pixel (both int16 maps), 4096x3000", (double)W * H * 8.0, [&] { hipLaunchKernelGGL(k_transpose4, dim3((W + 63) / 64, (H + 63) / 64), dim3(256), 0, 0, a, b, W, H); });
    return 0;
}
