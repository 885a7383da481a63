// write_rates.hip -- why does hipMemsetAsync fill at 6.5 TB/s when every write-only kernel of stream_rates.hip reaches 3.9-4.35?
//   hipcc --offload-arch=gfx950 -O3 -o write_rates write_rates.hip && ./write_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// A: grid-stride, 16 B per thread per step (stream_rates' W')
__global__ void __launch_bounds__(256) k_stride(v4u *out, size_t n, unsigned val)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v4u{val, val, val, val};
}
// B: one 16-byte store per thread, no loop
__global__ void __launch_bounds__(256) k_one(v4u *out, size_t n, unsigned val)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v4u{val, val, val, val};
}
// C: a workgroup owns a contiguous chunk of CH KB and walks it front to back
template <int CHUNK_KB>
__global__ void __launch_bounds__(256) k_chunk(v4u *out, size_t n, unsigned val)
{
    const size_t per = (size_t)CHUNK_KB * 1024 / 16;
    const size_t b0 = (size_t)blockIdx.x * per, b1 = std::min(b0 + per, n);
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256) out[i] = v4u{val, val, val, val};
}
// D: a THREAD owns 64 contiguous bytes (4 stores), threads consecutive
__global__ void __launch_bounds__(256) k_thread64(v4u *out, size_t n, unsigned val)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[i + k] = v4u{val, val, val, val};
    }
}
// E: like B, data dependent on the index (not a constant fill)
__global__ void __launch_bounds__(256) k_one_data(v4u *out, size_t n, unsigned val)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v4u{(unsigned)i * 2654435761u, val ^ (unsigned)i, (unsigned)(i >> 3), 7u * (unsigned)i};
}
// F: non-temporal variants of B / E
__global__ void __launch_bounds__(256) k_one_nt(v4u *out, size_t n, unsigned val)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(v4u{(unsigned)i * 2654435761u, val ^ (unsigned)i, (unsigned)(i >> 3), 7u * (unsigned)i}, out + i);
}

int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    void *buf;
    CK(hipMalloc(&buf, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto &&fn) {
        for (int w = 0; w < 3; ++w) fn();
        CK(hipDeviceSynchronize());
        std::vector<float> ms;
        for (int r = 0; r < 7; ++r) {
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < 4; ++k) fn();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t / 4);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-72s %8.1f us  %5.2f TB/s (best %5.2f)\n", name, ms[3] * 1e3, bytes / (ms[3] * 1e-3) / 1e12, bytes / (ms[0] * 1e-3) / 1e12);
    };
    timeit("hipMemsetAsync value 0", [&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); });
    timeit("hipMemsetAsync value 0x5a", [&] { CK(hipMemsetAsync(buf, 0x5a, bytes, 0)); });
    timeit("hipMemsetD32Async value 0x12345678", [&] { CK(hipMemsetD32Async((hipDeviceptr_t)buf, 0x12345678, bytes / 4, 0)); });
    for (unsigned g : {1024u, 4096u, 16384u, 65536u})
        for (unsigned val : {0u, 0x9e3779b9u}) {
            char nm[128];
            snprintf(nm, sizeof nm, "A grid-stride 16 B, %u workgroups, value %#x", g, val);
            timeit(nm, [&] { hipLaunchKernelGGL(k_stride, dim3(g), dim3(256), 0, 0, (v4u *)buf, n, val); });
        }
    for (unsigned val : {0u, 0x9e3779b9u}) {
        char nm[128];
        snprintf(nm, sizeof nm, "B one 16-byte store per thread, value %#x", val);
        timeit(nm, [&] { hipLaunchKernelGGL(k_one, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (v4u *)buf, n, val); });
    }
    timeit("C workgroup owns 64 KB", [&] { hipLaunchKernelGGL((k_chunk<64>), dim3((unsigned)(bytes / (64 * 1024))), dim3(256), 0, 0, (v4u *)buf, n, 0x9e3779b9u); });
    timeit("C workgroup owns 1 MB", [&] { hipLaunchKernelGGL((k_chunk<1024>), dim3((unsigned)(bytes / (1024 * 1024))), dim3(256), 0, 0, (v4u *)buf, n, 0x9e3779b9u); });
    timeit("D thread owns 64 B", [&] { hipLaunchKernelGGL(k_thread64, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (v4u *)buf, n, 0x9e3779b9u); });
    timeit("E one store per thread, data = f(index)", [&] { hipLaunchKernelGGL(k_one_data, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (v4u *)buf, n, 5u); });
    timeit("F the same, non-temporal", [&] { hipLaunchKernelGGL(k_one_nt, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (v4u *)buf, n, 5u); });
    return 0;
}
