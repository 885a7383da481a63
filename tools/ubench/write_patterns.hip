// write_patterns.hip -- how fast does an MI355X take stores of the shapes the x-major list stage produces?
//   hipcc --offload-arch=gfx950 -O3 -o write_patterns write_patterns.hip && ./write_patterns
// Every kernel writes the same number of bytes (TOTAL) with 64-lane waves:
//   A  16 B per lane, a wave writes 1 KB contiguous, waves contiguous                      (the scan kernels' stores)
//   B   8 B per lane, a wave writes 512 B contiguous, waves contiguous
//   C   8 B per lane, each HALF-wave writes a 256-byte run; runs of one workgroup-iteration land RUN_STRIDE bytes apart (aligned)
//   D   as C with every run shifted by 8 * (run index % 16) bytes (runs start on arbitrary 8-byte boundaries, neighbours abut)
//   E   as D but only 26 of the 32 lanes of a half-wave write (rank-compacted column segments at 81 % valid pixels)
//   D'  as D with the phases restricted to multiples of 16 / 32 / 64 bytes (which write granularity hurts?)
//   F   as C with 4 B per lane (128-byte runs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr size_t TOTAL = 640ull << 20;

__global__ void __launch_bounds__(256) k_a(float4 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void __launch_bounds__(256) k_b(double *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (double)i;
}
// run r (256 B = 32 doubles) of the buffer is written by half-wave (r mod 8) of workgroup-iteration r / 8 ... permuted: consecutive
// half-waves write runs `stride_runs` apart (like the columns of a tile), consecutive iterations of a workgroup the next run of each column
template <int MODE>
__global__ void __launch_bounds__(256) k_runs(double *out, size_t nruns, size_t cols, int lanes_on, int phase_mul)
{
    const int hw = threadIdx.x >> 5, r = threadIdx.x & 31;
    const size_t per_col = nruns / cols;                 // runs per "column"
    for (size_t it = blockIdx.x; it < nruns / 8; it += gridDim.x) {
        const size_t col = (it % (cols / 8)) * 8 + hw, seg = it / (cols / 8);
        size_t base = (col * per_col + seg) * 32;
        if (MODE >= 1) base = col * per_col * 32 + seg * lanes_on + (col % 16) * phase_mul % 16;      // abutting runs of lanes_on records, arbitrary 8-byte phase
        if (r < lanes_on) out[base + r] = (double)it;
    }
}
// G: the abutting unaligned runs of D, but neighbouring runs of a column are written close in time:
//   WHO = 0  by the same workgroup in consecutive iterations (it walks down its 8 columns)
//   WHO = 1  by workgroups b and b + 8 at the same iteration (same XCD when workgroups go round-robin over the 8 XCDs)
//   WHO = 2  by workgroups b and b + 1 at the same iteration (different XCDs)
template <int WHO>
__global__ void __launch_bounds__(256) k_walk(double *out, size_t nruns, size_t cols, int lanes_on)
{
    const int hw = threadIdx.x >> 5, r = threadIdx.x & 31;
    const size_t per_col = nruns / cols, groups = cols / 8;          // column groups of 8 (one per half-wave)
    // work item (group g, seg): WHO 0: a workgroup takes group g = b % groups and the segs [part * span, ...) in order
    const size_t nb = gridDim.x, b = blockIdx.x;
    if (WHO == 0) {
        const size_t parts = nb / groups, g = b % groups, part = b / groups, span = per_col / parts;
        if (part >= parts) return;
        for (size_t seg = part * span; seg < (part + 1) * span; ++seg) {
            const size_t col = g * 8 + hw;
            if (r < lanes_on) out[col * per_col * 32 + seg * lanes_on + (col % 16) + r] = (double)seg;
        }
    } else {
        const size_t step = WHO == 1 ? 8 : 1;
        // workgroups are grouped in bundles of `bundle` = 16 * step ... simpler: seg = (b / step) % 16 + 16 * k, lane-of-bundle picks the group
        const size_t sub = b % step, q = b / step;                   // q-th workgroup of its residue class
        const size_t segl = q % 16;                                  // 16 consecutive segs are in flight together
        const size_t slot = (q / 16) * step + sub, nslots = nb / 16;  // slot -> column group
        for (size_t g = slot; g < groups; g += nslots)
            for (size_t seg = segl; seg < per_col; seg += 16) {
                const size_t col = g * 8 + hw;
                if (r < lanes_on) out[col * per_col * 32 + seg * lanes_on + (col % 16) + r] = (double)seg;
            }
    }
}
// H: G0 with NS output streams (like cam / proj / 3 point planes / colours): a half-wave walks down its column and, per segment, writes the
// run of every stream (BURST = 1), or writes BURST consecutive segments of one stream before it turns to the next stream.
// ALIGNED: the runs start on multiples of their own length (phase 0: what a scatter that buffers whole lines per column would write)
template <int NS, int BURST, bool ALIGNED = false>
__global__ void __launch_bounds__(256) k_walk_streams(double *out, size_t nruns, size_t cols, int lanes_on)
{
    const int hw = threadIdx.x >> 5, r = threadIdx.x & 31;
    const size_t runs_per_stream = nruns / NS, per_col = runs_per_stream / cols, groups = cols / 8, stream_elems = runs_per_stream * 32;
    const size_t nb = gridDim.x, b = blockIdx.x;
    const size_t parts = nb / groups, g = b % groups, part = b / groups, span = per_col / parts;
    if (part >= parts) return;
    const size_t col = g * 8 + hw;
    for (size_t seg0 = part * span; seg0 + BURST <= (part + 1) * span; seg0 += BURST)
#pragma unroll
        for (int st = 0; st < NS; ++st)
#pragma unroll
            for (int k = 0; k < BURST; ++k)
                if (r < lanes_on) out[st * stream_elems + col * per_col * 32 + (seg0 + k) * lanes_on + (ALIGNED ? 0 : col % 16) + r] = (double)seg0;
}
__global__ void __launch_bounds__(256) k_runs4(float *out, size_t nruns, size_t cols)
{
    const int hw = threadIdx.x >> 5, r = threadIdx.x & 31;
    const size_t per_col = nruns / cols;
    for (size_t it = blockIdx.x; it < nruns / 8; it += gridDim.x) {
        const size_t col = (it % (cols / 8)) * 8 + hw, seg = it / (cols / 8);
        out[(col * per_col + seg) * 32 + r] = (float)it;
    }
}

template <class F>
static void timeit(const char *name, size_t bytes, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> t;
    for (int rep = 0; rep < 7; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / 5);
    }
    std::sort(t.begin(), t.end());
    printf("%-58s %8.1f us  %6.2f TB/s\n", name, t[3] * 1e3, bytes / (t[3] * 1e-3) / 1e12);
}

int main()
{
    void *buf;
    CK(hipMalloc(&buf, TOTAL + (1 << 20)));
    CK(hipMemset(buf, 0, TOTAL + (1 << 20)));
    const int grid = 256 * 12;
    timeit("A 16 B/lane, contiguous", TOTAL, [&] { hipLaunchKernelGGL(k_a, dim3(grid), dim3(256), 0, 0, (float4 *)buf, TOTAL / 16); });
    timeit("B  8 B/lane, contiguous", TOTAL, [&] { hipLaunchKernelGGL(k_b, dim3(grid), dim3(256), 0, 0, (double *)buf, TOTAL / 8); });
    const size_t nruns = TOTAL / 256;
    for (size_t cols : {4096ul, 65536ul}) {
        char name[128];
        snprintf(name, sizeof name, "C  8 B/lane, aligned 256-B runs, %zu columns", cols);
        timeit(name, TOTAL, [&] { hipLaunchKernelGGL(k_runs<0>, dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, cols, 32, 1); });
        snprintf(name, sizeof name, "D  8 B/lane, abutting 256-B runs on 8-B phases, %zu columns", cols);
        timeit(name, TOTAL, [&] { hipLaunchKernelGGL(k_runs<1>, dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, cols, 32, 1); });
        for (int pm : {2, 4, 8}) {
            snprintf(name, sizeof name, "D' as D with run phases multiples of %d B, %zu columns", 8 * pm, cols);
            timeit(name, TOTAL, [&] { hipLaunchKernelGGL(k_runs<1>, dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, cols, 32, pm); });
        }
        snprintf(name, sizeof name, "E  as D, 26 of 32 lanes (208-B runs), %zu columns", cols);
        timeit(name, TOTAL * 26 / 32, [&] { hipLaunchKernelGGL(k_runs<1>, dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, cols, 26, 1); });
    }
    timeit("G0 as D, a workgroup walks down its columns", TOTAL, [&] { hipLaunchKernelGGL(k_walk<0>, dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 32); });
    timeit("G1 as D, neighbours from workgroups b, b+8 together", TOTAL, [&] { hipLaunchKernelGGL(k_walk<1>, dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 32); });
    timeit("G2 as D, neighbours from workgroups b, b+1 together", TOTAL, [&] { hipLaunchKernelGGL(k_walk<2>, dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 32); });
    timeit("H  as G0 with 8 streams, segment by segment (26 lanes)", TOTAL * 26 / 32, [&] { hipLaunchKernelGGL((k_walk_streams<8, 1>), dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 26); });
    timeit("H  as G0 with 8 streams, bursts of 4 segments (26 lanes)", TOTAL * 26 / 32, [&] { hipLaunchKernelGGL((k_walk_streams<8, 4>), dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 26); });
    timeit("H  as G0 with 1 stream (26 lanes)", TOTAL * 26 / 32, [&] { hipLaunchKernelGGL((k_walk_streams<1, 1>), dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 26); });
    timeit("H  as G0 with 8 streams, segment by segment (32 lanes)", TOTAL, [&] { hipLaunchKernelGGL((k_walk_streams<8, 1>), dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 32); });
    timeit("I  as H, 8 streams, ALIGNED 256-B runs (32 lanes)", TOTAL, [&] { hipLaunchKernelGGL((k_walk_streams<8, 1, true>), dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 32); });
    timeit("I  as H, 8 streams, ALIGNED 128-B runs (16 lanes)", TOTAL / 2, [&] { hipLaunchKernelGGL((k_walk_streams<8, 1, true>), dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 16); });
    timeit("I  as H, 1 stream, ALIGNED 256-B runs (32 lanes)", TOTAL, [&] { hipLaunchKernelGGL((k_walk_streams<1, 1, true>), dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 32); });
    timeit("I  as H, 8 streams, ALIGNED 256-B runs, bursts of 4", TOTAL, [&] { hipLaunchKernelGGL((k_walk_streams<8, 4, true>), dim3(grid), dim3(256), 0, 0, (double *)buf, nruns, 4096ul, 32); });
    timeit("F  4 B/lane, aligned 128-B runs, 4096 columns", TOTAL / 2, [&] { hipLaunchKernelGGL(k_runs4, dim3(grid), dim3(256), 0, 0, (float *)buf, nruns, 4096ul); });
    CK(hipFree(buf));
    return 0;
}
