// host_alloc.hip -- what page-locking 100 MB costs, by method (the result-array pool of scanner/_native.py pays this once per block)
//   hipcc --offload-arch=gfx950 -O2 -o host_alloc host_alloc.hip && ./host_alloc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t n = 100u << 20;
    void *d;
    hipMalloc(&d, n);
    hipStream_t s;
    hipStreamCreate(&s);
    struct { const char *name; unsigned flags; } kinds[] = {{"hipHostMalloc default", hipHostMallocDefault}, {"hipHostMalloc non-coherent", hipHostMallocNonCoherent},
                                                            {"hipHostMalloc portable|mapped", hipHostMallocPortable | hipHostMallocMapped}};
    for (auto &k : kinds)
        for (int rep = 0; rep < 2; ++rep) {
            void *p = nullptr;
            double t0 = now();
            if (hipHostMalloc(&p, n, k.flags) != hipSuccess) { printf("%s: failed\n", k.name); break; }
            double t1 = now();
            hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            double t2 = now();
            hipHostFree(p);
            double t3 = now();
            printf("%-32s alloc %6.1f ms  D2H %5.1f ms (%5.1f GB/s)  free %5.1f ms\n", k.name, t1 - t0, t2 - t1, n / (t2 - t1) / 1e6, t3 - t2);
        }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        double t1 = now();
        hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
        double t2 = now();
        hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        double t3 = now();
        hipHostUnregister(p);
        munmap(p, n);
        double t4 = now();
        printf("%-32s mmap %4.1f register %6.1f ms (%s)  D2H %5.1f ms (%5.1f GB/s)  undo %5.1f ms\n", "mmap + hipHostRegister", t1 - t0, t2 - t1, hipGetErrorString(e), t3 - t2,
               n / (t3 - t2) / 1e6, t4 - t3);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        double t1 = now();
        hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        double t2 = now();
        munmap(p, n);
        printf("%-32s first D2H into fresh pages %5.1f ms (%5.1f GB/s), second %5.1f ms (%5.1f GB/s)\n", "plain mmap", t1 - t0, n / (t1 - t0) / 1e6, t2 - t1, n / (t2 - t1) / 1e6);
    }
    return 0;
}
