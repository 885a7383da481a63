// stream_rates.hip -- what one MI355X sustains for the byte mixes of the scan kernels when a kernel does NOTHING but move those bytes, measured
// on the box at hand: the bar the roofline fractions in DESIGN.md are held against.
//   hipcc --offload-arch=gfx950 -O3 -o stream_rates stream_rates.hip && ./stream_rates
//   R    read only  (16 B / lane, non-temporal)            W / W' / Wa   write only (non-temporal / ordinary / every sc0-nt-sc1 policy: 3.9-4.25 TB/s, no magic bit)            C   copy
//   D    the decode kernel's mix: 44 planes of 4 B / lane read, the two int16 maps written as 8 B / lane each       (N + 4  B/px)
//   F    the fused scan kernel's mix with the XYZ stored as a lane holds it (48 B per lane, 16 B at a time)           (N + 16 B/px)
//   Ft   ... with the XYZ stored wave-contiguously, as the fused kernel's LDS transpose leaves it; Ft' without the maps (N + 12 B/px)
//   D8 / D16 / Dp   D with 8 / 16 bytes per lane and plane, or lane pairs splitting two planes: the width of the loads does not matter
//   Dc / Dc1 / Dt / Dt4   the same bytes read contiguously or from a tile-interleaved stack (round 6: `./stream_rates L` runs only these)
//   S / Sp          D in the decode kernel's real schedule (14 threshold planes, then steps of 4 behind a ring of DEPTH steps, with and
//                   without dependent arithmetic), singly or in plane pairs: neither depth, arithmetic nor pairing moves it
// One box (round 3): R 6.9 TB/s, W 3.9 (4.2 with ordinary stores), C 5.3-5.6; D 101 us (the decode kernel: 93-103 us); Ft 123 us (the fused kernel, which also gathers its
// ray tables: 122-128 us); Ft' 113 us (fused kernel without map buffers: 118 us); F 163 us (what skipping the transpose would cost).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

// what a lane writes for its 4 pixels, in the shapes the scan kernels use: WD = 4: two 8-byte stores (int16 h map, int16 v map);
// WD = 16: those + three 16-byte stores (XYZ); WD = 12: the three 16-byte stores only
typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
template <int WD>
__device__ __forceinline__ void store_like_the_kernels(unsigned x, size_t g, size_t npix4, unsigned *out)
{
    if (WD == 4 || WD == 16) {
        __builtin_nontemporal_store(v2u_{x, x + 1}, reinterpret_cast<v2u_ *>(out) + g);
        __builtin_nontemporal_store(v2u_{x + 2, x + 3}, reinterpret_cast<v2u_ *>(out + 2 * npix4) + g);
    }
    if (WD == 12 || WD == 16) {       // 48 bytes of XYZ per lane, as the lane holds them: three 16-byte stores 48 bytes apart from lane to lane
        v4u_ *xyz = reinterpret_cast<v4u_ *>(out + 4 * npix4) + 3 * g;
#pragma unroll
        for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(v4u_{x, x + k, x + 5, x + 7}, xyz + k);
    }
    if (WD == 112 || WD == 116) {     // the same bytes after the fused kernel's LDS transpose: every store instruction writes 1 KB contiguous per wave
        if (WD == 116) {
            __builtin_nontemporal_store(v2u_{x, x + 1}, reinterpret_cast<v2u_ *>(out) + g);
            __builtin_nontemporal_store(v2u_{x + 2, x + 3}, reinterpret_cast<v2u_ *>(out + 2 * npix4) + g);
        }
        const size_t wave0 = g & ~(size_t)63, lane = g & 63;
        v4u_ *xyz = reinterpret_cast<v4u_ *>(out + 4 * npix4) + 3 * wave0;
#pragma unroll
        for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(v4u_{x, x + k, x + 5, x + 7}, xyz + 64 * k + lane);
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_read(const v4u *in, size_t n, unsigned *sink)
{
    v4u acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= __builtin_nontemporal_load(in + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
__global__ void __launch_bounds__(256) k_write(v4u *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store(v4u{(unsigned)i, 1u, 2u, 3u}, out + i);
}
__global__ void __launch_bounds__(256) k_write_plain(v4u *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v4u{(unsigned)i, 1u, 2u, 3u};
}
// write only through a buffer descriptor with the cache-policy bits spelled out (aux: 1 = sc0, 2 = nt, 16 = sc1 on gfx94x / gfx950)
template <int AUX>
__global__ void __launch_bounds__(256) k_write_aux(v4u *out, size_t n)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)(n * 16 > 0x7fffffffu ? 0x7fffffffu : n * 16), 0x00020000);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        __builtin_amdgcn_raw_buffer_store_b128(v4u{(unsigned)i, 1u, 2u, 3u}, rs, (unsigned)(i * 16), 0, AUX);
}
__global__ void __launch_bounds__(256) k_copy(const v4u *in, v4u *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
// planes of npix bytes each, a lane takes 4 consecutive pixels: NP dword loads (one per plane), then `wd` dwords of output per lane
template <int NP, int WD>
__global__ void __launch_bounds__(128) k_planes(const unsigned *in, size_t npix4, unsigned *out)
{
    const size_t g = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (g >= npix4) return;
    unsigned v[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) v[p] = __builtin_nontemporal_load(in + (size_t)p * npix4 + g);
    unsigned x = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) x = ((x << 1) | (x >> 31)) ^ v[p];            // a rotation: every plane stays live (a plain shift lets the compiler drop the first NP - 32 loads)
    store_like_the_kernels<WD>(x, g, npix4, out);
}

// D with wider lanes: LW dwords (4 LW pixels) per lane and plane, folded as they arrive -- does the 44-plane pattern want 8 or 16 bytes per lane?
template <int NP, int LW>
__global__ void __launch_bounds__(128) k_planes_wide(const unsigned *in, size_t npix4, unsigned *out)
{
    const size_t g = ((size_t)blockIdx.x * 128 + threadIdx.x) * LW;
    if (g >= npix4) return;
    typedef unsigned vw __attribute__((ext_vector_type(LW)));
    vw x = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const vw v = __builtin_nontemporal_load(reinterpret_cast<const vw *>(in + (size_t)p * npix4 + g));
        x = ((x << 1) | (x >> 31)) ^ v;
    }
#pragma unroll
    for (int q = 0; q < LW; ++q) store_like_the_kernels<4>(x[q], g + q, npix4, out);
}

// D8 again, but the two dwords of a lane come from two PLANES: even lanes read 8 bytes of plane p, odd lanes 8 bytes of plane p + 1, both for
// the pixel octet of the lane pair (a wave still touches 256 contiguous bytes per plane, in half as many load instructions)
template <int NP>
__global__ void __launch_bounds__(128) k_planes_pair(const unsigned *in, size_t npix4, unsigned *out)
{
    const size_t t = (size_t)blockIdx.x * 128 + threadIdx.x;             // lane pair t / 2 owns pixels 8 (t / 2) .. + 8
    const size_t g2 = (t >> 1) * 2;                                      // dword index of the octet inside a plane
    if (g2 >= npix4) return;
    typedef unsigned v2 __attribute__((ext_vector_type(2)));
    v2 x = 0;
#pragma unroll
    for (int p = 0; p < NP; p += 2) {
        const v2 v = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(in + (size_t)(p + (t & 1)) * npix4 + g2));
        x = ((x << 1) | (x >> 31)) ^ v;
    }
    store_like_the_kernels<4>(x.x ^ x.y, t, npix4, out);
}

// The decode kernel's REAL load schedule without its arithmetic: 14 planes up front (the threshold frames; waited for together), then the
// other 30 in steps of 4 with a ring of DEPTH steps in flight ahead of the step being consumed.  How much memory-level parallelism does
// the pattern need before the kernel stops being latency-bound?  (DEPTH = 8 is "everything requested at once".)
template <int DEPTH, int WD, int SPIN>
__global__ void __launch_bounds__(128) k_staged(const unsigned *in, size_t npix4, unsigned *out)
{
    const size_t g = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (g >= npix4) return;
    unsigned x = 0;
    unsigned t[14];
#pragma unroll
    for (int p = 0; p < 14; ++p) t[p] = __builtin_nontemporal_load(in + (size_t)p * npix4 + g);
    constexpr int STEPS = 8;                                     // 7.5 steps of 4 planes = 30
    unsigned ring[DEPTH + 1][4];
    auto fetch = [&](int st, unsigned (&r)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = 14 + 4 * st + k;
            r[k] = p < 44 ? __builtin_nontemporal_load(in + (size_t)p * npix4 + g) : 0u;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < STEPS) fetch(d, ring[d]);
#pragma unroll
    for (int p = 0; p < 14; ++p) x = ((x << 1) | (x >> 31)) ^ t[p];
#pragma unroll
    for (int i = 0; i < SPIN; ++i) x = x * 1664525u + 1013904223u;      // stand-in for the threshold arithmetic (dependent chain)
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        if (st + DEPTH < STEPS) fetch(st + DEPTH, ring[(st + DEPTH) % (DEPTH + 1)]);
        unsigned (&c)[4] = ring[st % (DEPTH + 1)];
        x = ((x << 1) | (x >> 31)) ^ c[0] ^ ((c[1] >> 1) | (c[1] << 31)) ^ ((c[2] << 2) | (c[2] >> 30)) ^ ((c[3] >> 3) | (c[3] << 29));
#pragma unroll
        for (int i = 0; i < SPIN / 8; ++i) x = x * 1664525u + 1013904223u;
        asm volatile("" : "+v"(x));
    }
    store_like_the_kernels<WD>(x, g, npix4, out);
}

// S with the planes fetched in PAIRS: even lanes read 8 bytes of plane A, odd lanes 8 bytes of plane B (the pixel octet of the lane pair), then
// the two lanes trade halves (2 DPP moves + 2 selects per pair) so that each ends up with its own 4 pixels of both planes -- half the load
// instructions of S for the same bytes, registers and per-lane work.
__device__ __forceinline__ void trade(unsigned x, unsigned y, bool odd, unsigned &a, unsigned &b)
{
    const unsigned send = odd ? x : y;
    const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xf, 0xf, false);      // quad_perm [1, 0, 3, 2]
    a = odd ? recv : x;
    b = odd ? y : recv;
}
template <int DEPTH, int WD, int SPIN>
__global__ void __launch_bounds__(128) k_staged_pair(const unsigned *in, size_t npix4, unsigned *out)
{
    typedef unsigned v2 __attribute__((ext_vector_type(2)));
    const size_t g = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (g >= npix4) return;
    const bool odd = threadIdx.x & 1;
    const size_t g2 = g & ~(size_t)1;
    const unsigned *lane_base = in + g2 + (odd ? npix4 : 0);      // plane p for even lanes, p + 1 for odd lanes
    unsigned x = 0;
    v2 t[7];
#pragma unroll
    for (int p = 0; p < 7; ++p) t[p] = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(lane_base + (size_t)(2 * p) * npix4));
    constexpr int STEPS = 8;
    v2 ring[DEPTH + 1][2];
    auto fetch = [&](int st, v2 (&r)[2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = 14 + 4 * st + 2 * k;
            r[k] = p < 44 ? __builtin_nontemporal_load(reinterpret_cast<const v2 *>(lane_base + (size_t)p * npix4)) : v2{0u, 0u};
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < STEPS) fetch(d, ring[d]);
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        unsigned a, b;
        trade(t[p].x, t[p].y, odd, a, b);
        x = ((x << 2) | (x >> 30)) ^ a ^ ((b << 1) | (b >> 31));
    }
#pragma unroll
    for (int i = 0; i < SPIN; ++i) x = x * 1664525u + 1013904223u;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        if (st + DEPTH < STEPS) fetch(st + DEPTH, ring[(st + DEPTH) % (DEPTH + 1)]);
        v2 (&c)[2] = ring[st % (DEPTH + 1)];
        unsigned a0, b0, a1, b1;
        trade(c[0].x, c[0].y, odd, a0, b0);
        trade(c[1].x, c[1].y, odd, a1, b1);
        x = ((x << 1) | (x >> 31)) ^ a0 ^ ((b0 >> 1) | (b0 << 31)) ^ ((a1 << 2) | (a1 >> 30)) ^ ((b1 >> 3) | (b1 << 29));
#pragma unroll
        for (int i = 0; i < SPIN / 8; ++i) x = x * 1664525u + 1013904223u;
        asm volatile("" : "+v"(x));
    }
    store_like_the_kernels<WD>(x, g, npix4, out);
}

// Dc: the read stream of R (16 B per lane, contiguous, grid-stride) with the decode kernel's 4 B/px written beside it: after every 11 loads
// (176 B = 4 pixels x 44 planes) a lane stores its two 8-byte map pieces.  "Contiguous read + 4 B/px write": is the 8-10 % between the
// additive bound and D the 44-stream planar layout, or the read/write mix itself?
__global__ void __launch_bounds__(256) k_contig_rw(const v4u *in, size_t npix4, unsigned *out)
{
    const size_t T = (size_t)gridDim.x * 256, t = (size_t)blockIdx.x * 256 + threadIdx.x;
    // chunk c of 11 x T vectors: lane reads in[(11 c + k) T + t], k = 0..10 (each of the 11 sweeps is one contiguous T x 16 B run), writes pixel group c T + t
    for (size_t c = 0; c * T < npix4; ++c) {
        const size_t g = c * T + t;
        if (g >= npix4) break;
        v4u acc = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 11; ++k) acc ^= __builtin_nontemporal_load(in + (11 * c + k) * T + t);
        store_like_the_kernels<4>(acc.x ^ acc.y ^ acc.z ^ acc.w, g, npix4, out);
    }
}
// Dc1: one pass, every lane owns 4 pixels: its 176 B are the 11 vectors in[11 blockbase + 256 k + tid] of its WORKGROUP's contiguous 44 KB
__global__ void __launch_bounds__(256) k_contig_block(const v4u *in, size_t npix4, unsigned *out)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= npix4) return;
    const v4u *base = in + (size_t)blockIdx.x * 256 * 11 + threadIdx.x;
    v4u acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 11; ++k) acc ^= __builtin_nontemporal_load(base + 256 * k);
    store_like_the_kernels<4>(acc.x ^ acc.y ^ acc.z ^ acc.w, g, npix4, out);
}
// Dt: a tile-interleaved stack [tile][NP][TILE lanes x 4 B]: a tile's NP plane pieces are contiguous (TILE = 64: one wave's 44 x 256 B = 11 KB),
// the lane still gets its own 4 pixels of one plane per dword load -- the decode kernel's register layout unchanged
template <int NP, int TILE>
__global__ void __launch_bounds__(128) k_tiled(const unsigned *in, size_t npix4, unsigned *out)
{
    const size_t g = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (g >= npix4) return;
    const unsigned *base = in + (g / TILE) * (size_t)(TILE * NP) + (g % TILE);
    unsigned v[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) v[p] = __builtin_nontemporal_load(base + p * TILE);
    unsigned x = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) x = ((x << 1) | (x >> 31)) ^ v[p];
    store_like_the_kernels<4>(x, g, npix4, out);
}
// Dt4: [tile][NP / 4][TILE lanes][4 planes x 4 px]: one 16-byte load hands a lane its 4 pixels of FOUR planes (11 loads instead of 44)
template <int NP, int TILE>
__global__ void __launch_bounds__(128) k_tiled4(const v4u *in, size_t npix4, unsigned *out)
{
    const size_t g = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (g >= npix4) return;
    const v4u *base = in + (g / TILE) * (size_t)(TILE * NP / 4) + (g % TILE);
    v4u v[NP / 4];
#pragma unroll
    for (int p = 0; p < NP / 4; ++p) v[p] = __builtin_nontemporal_load(base + p * TILE);
    unsigned x = 0;
#pragma unroll
    for (int p = 0; p < NP / 4; ++p) {
        x = ((x << 1) | (x >> 31)) ^ v[p].x; x = ((x << 1) | (x >> 31)) ^ v[p].y;
        x = ((x << 1) | (x >> 31)) ^ v[p].z; x = ((x << 1) | (x >> 31)) ^ v[p].w;
    }
    store_like_the_kernels<4>(x, g, npix4, out);
}

template <class F>
static void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> t;
    for (int rep = 0; rep < 9; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / 5);
    }
    std::sort(t.begin(), t.end());
    printf("%-78s %8.1f us  %6.2f TB/s  (best %6.2f)\n", name, t[4] * 1e3, bytes / (t[4] * 1e-3) / 1e12, bytes / (t[0] * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const bool layout_only = argc > 1 && argv[1][0] == 'L';
    const size_t npix = 4096ull * 3000, npix4 = npix / 4, big = 44 * npix;          // the 4096 x 3000 x 44 stack: 541 MB
    void *a, *b, *sink;
    CK(hipMalloc(&a, 2 * big + (256ull << 20)));                                                         // two stacks: rotated like bench.py (> Infinity Cache)
    CK(hipMalloc(&b, big));                                                             // outputs: 16 B / pixel at most (197 MB)
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, 2 * big + (256ull << 20)));
    CK(hipMemset(b, 0, big));
    int flip = 0;
    const int grid = 256 * 16;
    if (!layout_only) {
    timeit("R  read only, 541 MB", (double)big, [&] { flip ^= 1; hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const v4u *)((char *)a + flip * big), big / 16, (unsigned *)sink); });
    timeit("W  write only, 541 MB", (double)big, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, (v4u *)b, big / 16); });
    timeit("W'  write only, ordinary (cacheable) stores", (double)big, [&] { hipLaunchKernelGGL(k_write_plain, dim3(grid), dim3(256), 0, 0, (v4u *)b, big / 16); });
#define WAUX(A) timeit("Wa write only, buffer stores, aux = " #A, (double)big, [&] { hipLaunchKernelGGL((k_write_aux<A>), dim3(grid), dim3(256), 0, 0, (v4u *)b, big / 16); });
    WAUX(0) WAUX(1) WAUX(2) WAUX(3) WAUX(16) WAUX(17) WAUX(18) WAUX(19)
    timeit("C  copy, 541 MB read + 541 MB written", 2.0 * big, [&] { flip ^= 1; hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const v4u *)((char *)a + flip * big), (v4u *)b, big / 16); });
    const unsigned blocks = (unsigned)((npix4 + 127) / 128);
    timeit("D  decode mix: 44 planes read, 4 B/px written (N + 4 = 48 B/px: 590 MB)", 48.0 * npix, [&] { flip ^= 1; hipLaunchKernelGGL((k_planes<44, 4>), dim3(blocks), dim3(128), 0, 0, (const unsigned *)((char *)a + flip * big), npix4, (unsigned *)b); });
    timeit("F  fused mix: 44 planes read, 16 B/px written (N + 16 = 60 B/px: 737 MB)", 60.0 * npix, [&] { flip ^= 1; hipLaunchKernelGGL((k_planes<44, 16>), dim3(blocks), dim3(128), 0, 0, (const unsigned *)((char *)a + flip * big), npix4, (unsigned *)b); });
    timeit("Ft fused mix, XYZ stored wave-contiguously (what the LDS transpose buys): 737 MB", 60.0 * npix, [&] { flip ^= 1; hipLaunchKernelGGL((k_planes<44, 116>), dim3(blocks), dim3(128), 0, 0, (const unsigned *)((char *)a + flip * big), npix4, (unsigned *)b); });
    timeit("Ft' the same without the maps: 688 MB", 56.0 * npix, [&] { flip ^= 1; hipLaunchKernelGGL((k_planes<44, 112>), dim3(blocks), dim3(128), 0, 0, (const unsigned *)((char *)a + flip * big), npix4, (unsigned *)b); });
    timeit("F' fused mix without maps: 44 planes read, 12 B/px written (N + 12 = 56 B/px: 688 MB)", 56.0 * npix, [&] { flip ^= 1; hipLaunchKernelGGL((k_planes<44, 12>), dim3(blocks), dim3(128), 0, 0, (const unsigned *)((char *)a + flip * big), npix4, (unsigned *)b); });
    timeit("D8  decode mix, 8 B per lane and plane", 48.0 * npix, [&] { flip ^= 1; hipLaunchKernelGGL((k_planes_wide<44, 2>), dim3((unsigned)((npix4 / 2 + 127) / 128)), dim3(128), 0, 0, (const unsigned *)((char *)a + flip * big), npix4, (unsigned *)b); });
    timeit("D16 decode mix, 16 B per lane and plane", 48.0 * npix, [&] { flip ^= 1; hipLaunchKernelGGL((k_planes_wide<44, 4>), dim3((unsigned)((npix4 / 4 + 127) / 128)), dim3(128), 0, 0, (const unsigned *)((char *)a + flip * big), npix4, (unsigned *)b); });
    timeit("Dp  decode mix, 8 B per lane, lane pairs split the planes", 48.0 * npix, [&] { flip ^= 1; hipLaunchKernelGGL((k_planes_pair<44>), dim3(blocks), dim3(128), 0, 0, (const unsigned *)((char *)a + flip * big), npix4, (unsigned *)b); });
#define STAGED(D, SP) timeit("S  decode schedule, ring depth " #D ", " #SP " dependent multiply-adds per wave phase", 48.0 * npix, [&] { flip ^= 1; hipLaunchKernelGGL((k_staged<D, 4, SP>), dim3(blocks), dim3(128), 0, 0, (const unsigned *)((char *)a + flip * big), npix4, (unsigned *)b); });
    STAGED(1, 0) STAGED(2, 0) STAGED(3, 0) STAGED(5, 0) STAGED(8, 0)
    STAGED(2, 64) STAGED(2, 256) STAGED(8, 256) STAGED(2, 512) STAGED(8, 512)
#define STAGEDP(D, SP) timeit("Sp decode schedule in plane PAIRS, ring depth " #D ", " #SP " dependent multiply-adds", 48.0 * npix, [&] { flip ^= 1; hipLaunchKernelGGL((k_staged_pair<D, 4, SP>), dim3(blocks), dim3(128), 0, 0, (const unsigned *)((char *)a + flip * big), npix4, (unsigned *)b); });
    STAGEDP(1, 512) STAGEDP(2, 512) STAGEDP(3, 512) STAGEDP(4, 512) STAGEDP(5, 512) STAGEDP(6, 512) STAGEDP(8, 512) STAGEDP(8, 0)
    }
    // ---- layout question (round 6): planar D against contiguous / tile-interleaved reads of the same bytes, at 4096 x 3000 and 1920 x 1080
    for (int sz = 0; sz < 2; ++sz) {
        const size_t w = sz ? 1920 : 4096, h = sz ? 1080 : 3000, np = w * h, np4 = np / 4, bg = 44 * np;
        const int nrot = sz ? 8 : 2;                                                // rotate over > 256 MB of stacks so that nothing is served from the Infinity Cache
        int rot = 0;
        auto src = [&]() { rot = (rot + 1) % nrot; return (char *)a + (size_t)rot * bg; };
        const unsigned bl = (unsigned)((np4 + 127) / 128), bl256 = (unsigned)((np4 + 255) / 256);
        char nm[160];
        auto T = [&](const char *s) { snprintf(nm, sizeof nm, "%s  [%zux%zu]", s, w, h); return nm; };
        timeit(T("R   read only, the stack"), (double)bg, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const v4u *)src(), bg / 16, (unsigned *)sink); });
        timeit(T("D   planar decode mix (44 plane streams, 4 B/px written)"), 48.0 * np, [&] { hipLaunchKernelGGL((k_planes<44, 4>), dim3(bl), dim3(128), 0, 0, (const unsigned *)src(), np4, (unsigned *)b); });
        timeit(T("S   planar, the kernel's schedule (ring depth 2, arithmetic)"), 48.0 * np, [&] { hipLaunchKernelGGL((k_staged<2, 4, 512>), dim3(bl), dim3(128), 0, 0, (const unsigned *)src(), np4, (unsigned *)b); });
        timeit(T("Dc  contiguous grid-stride read (R's stream) + 4 B/px written, 4096 workgroups"), 48.0 * np, [&] { hipLaunchKernelGGL(k_contig_rw, dim3(grid), dim3(256), 0, 0, (const v4u *)src(), np4, (unsigned *)b); });
        timeit(T("Dc' the same, 1024 workgroups"), 48.0 * np, [&] { hipLaunchKernelGGL(k_contig_rw, dim3(1024), dim3(256), 0, 0, (const v4u *)src(), np4, (unsigned *)b); });
        timeit(T("Dc1 contiguous, one pass: a workgroup reads its own 44 KB, writes its 1 KB + 1 KB"), 48.0 * np, [&] { hipLaunchKernelGGL(k_contig_block, dim3(bl256), dim3(256), 0, 0, (const v4u *)src(), np4, (unsigned *)b); });
        timeit(T("Dt  tile-interleaved [tile][44][256 B] (one wave = 11 KB contiguous), dword loads"), 48.0 * np, [&] { hipLaunchKernelGGL((k_tiled<44, 64>), dim3(bl), dim3(128), 0, 0, (const unsigned *)src(), np4, (unsigned *)b); });
        timeit(T("Dt' tile-interleaved [tile][44][512 B] (one workgroup = 22 KB contiguous)"), 48.0 * np, [&] { hipLaunchKernelGGL((k_tiled<44, 128>), dim3(bl), dim3(128), 0, 0, (const unsigned *)src(), np4, (unsigned *)b); });
        timeit(T("Dtk tile-interleaved [tile][44][4 KB] (8 workgroups share 176 KB)"), 48.0 * np, [&] { hipLaunchKernelGGL((k_tiled<44, 1024>), dim3(bl), dim3(128), 0, 0, (const unsigned *)src(), np4, (unsigned *)b); });
#define DTK(TL, LABEL) timeit(T("Dtk tile-interleaved, plane piece = " LABEL), 48.0 * np, [&] { hipLaunchKernelGGL((k_tiled<44, TL>), dim3(bl), dim3(128), 0, 0, (const unsigned *)src(), np4, (unsigned *)b); });
        DTK(256, "1 KB (tile 44 KB)") DTK(2048, "8 KB (tile 352 KB)") DTK(4096, "16 KB (tile 704 KB)") DTK(16384, "64 KB (tile 2.8 MB)") DTK(65536, "256 KB (tile 11 MB)") DTK(262144, "1 MB (tile 46 MB)")
        timeit(T("Dt4 tile-interleaved [tile][11][64 lanes][4 planes x 4 px], 16-byte loads"), 48.0 * np, [&] { hipLaunchKernelGGL((k_tiled4<44, 64>), dim3(bl), dim3(128), 0, 0, (const v4u *)src(), np4, (unsigned *)b); });
        timeit(T("Dt4' the same, tile = workgroup (128 lanes)"), 48.0 * np, [&] { hipLaunchKernelGGL((k_tiled4<44, 128>), dim3(bl), dim3(128), 0, 0, (const v4u *)src(), np4, (unsigned *)b); });
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(sink));
    return 0;
}
