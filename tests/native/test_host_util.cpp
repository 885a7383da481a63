// CPU unit tests of csrc/host_util.cpp (the host-only helpers of the C-ABI), built and run under ASan + UBSan and under TSan by
// `make -C oracle san`.  The pinned-ring copy loop runs against a memcpy shim of the transport (a "device" that is a host buffer and
// lands its chunks from helper threads after a delay, like a DMA engine would), so ring wrap-around, sizes that are not a multiple of the
// chunk, slot reuse and the abort paths are exercised without a GPU.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "../../3dscanner-graycode_amd/csrc/host_util.h"

static int g_failed = 0;
static bool g_light = false;       // `light`: a reduced matrix (the TSan build runs 10-20x slower and spins on yields)
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_failed;                                                          \
        }                                                                        \
    } while (0)

static unsigned rnd(unsigned &s) { return s = s * 1664525u + 1013904223u; }

// ---------------------------------------------------------------- narrow_f64_to_u8
static void test_narrow()
{
    unsigned seed = 7;
    for (int n_runs = 1; n_runs <= 3; ++n_runs)
        for (size_t elems : {(size_t)0, (size_t)1, (size_t)63, (size_t)1000, (size_t)4099})
            for (size_t chunk : {(size_t)1, (size_t)7, (size_t)64, (size_t)1000, (size_t)(1u << 20)})
                for (int thr : {1, 2, 5, 16}) {
                    if (g_light && (n_runs == 2 || elems == 1 || elems == 63 || chunk == 1 || chunk == 64 || thr == 1 || thr == 16)) continue;
                    std::vector<std::vector<double>> runs((size_t)n_runs, std::vector<double>(elems));
                    std::vector<const void *> ptrs;
                    for (auto &r : runs) {
                        for (auto &x : r) x = (double)(rnd(seed) >> 24);
                        ptrs.push_back(r.data());
                    }
                    std::vector<uint8_t> dst(elems * (size_t)n_runs + 1, 0xAB);
                    CHECK(slgc_host::narrow_f64_to_u8(ptrs.data(), n_runs, elems, dst.data(), thr, chunk) == 1);
                    bool same = true;
                    for (int r = 0; r < n_runs; ++r)
                        for (size_t i = 0; i < elems; ++i) same &= dst[(size_t)r * elems + i] == (uint8_t)runs[(size_t)r][i];
                    CHECK(same);
                    CHECK(dst.back() == 0xAB);                                    // nothing written past the end
                    if (elems == 0) continue;
                    // one sample that is not a grey level, at the first / a middle / the very last position of some run
                    const double bads[] = {0.5, -1.0, 256.0, 255.0000001, std::nan(""), std::numeric_limits<double>::infinity(), -0.0 - 1e-300};
                    for (double bad : bads)
                        for (size_t pos : {(size_t)0, elems / 2, elems - 1}) {
                            const int r = (int)(rnd(seed) % (unsigned)n_runs);
                            const double keep = runs[(size_t)r][pos];
                            runs[(size_t)r][pos] = bad;
                            CHECK(slgc_host::narrow_f64_to_u8(ptrs.data(), n_runs, elems, dst.data(), thr, chunk) == 0);
                            runs[(size_t)r][pos] = keep;
                        }
                    runs[0][0] = -0.0;                                            // minus zero IS the grey level 0 (x >= 0 and (double)0 == -0.0)
                    CHECK(slgc_host::narrow_f64_to_u8(ptrs.data(), n_runs, elems, dst.data(), thr, chunk) == 1 && dst[0] == 0);
                }
}

// ---------------------------------------------------------------- narrow_i64_to_i16
static void test_narrow_maps()
{
    unsigned seed = 11;
    for (int n_maps = 1; n_maps <= 2; ++n_maps)
        for (size_t elems : {(size_t)0, (size_t)1, (size_t)63, (size_t)1000, (size_t)4099})
            for (size_t chunk : {(size_t)1, (size_t)7, (size_t)1000, (size_t)(1u << 20)})
                for (int thr : {1, 2, 5, 16}) {
                    if (g_light && (elems == 1 || elems == 63 || chunk == 1 || thr == 1 || thr == 16)) continue;
                    std::vector<std::vector<int64_t>> maps((size_t)n_maps, std::vector<int64_t>(elems));
                    std::vector<const int64_t *> ptrs;
                    for (auto &m : maps) {
                        for (auto &x : m) x = (int64_t)(rnd(seed) >> 16) - 32768;            // the whole int16 range, -1 included
                        ptrs.push_back(m.data());
                    }
                    if (elems) { maps[0][0] = -32768; maps[(size_t)n_maps - 1][elems - 1] = 32767; }
                    std::vector<int16_t> dst(elems * (size_t)n_maps + 1, (int16_t)0x5A5A);
                    CHECK(slgc_host::narrow_i64_to_i16(ptrs.data(), n_maps, elems, dst.data(), thr, chunk) == 1);
                    bool same = true;
                    for (int m = 0; m < n_maps; ++m)
                        for (size_t i = 0; i < elems; ++i) same &= (int64_t)dst[(size_t)m * elems + i] == maps[(size_t)m][i];
                    CHECK(same);
                    CHECK(dst.back() == (int16_t)0x5A5A);
                    if (elems == 0) continue;
                    const int64_t bads[] = {32768, -32769, (int64_t)1 << 16, (int64_t)1 << 40, std::numeric_limits<int64_t>::max(),
                                            std::numeric_limits<int64_t>::min(), -((int64_t)1 << 16) - 1, ((int64_t)1 << 32) - 1};
                    for (int64_t bad : bads)
                        for (size_t pos : {(size_t)0, elems / 2, elems - 1}) {
                            const int m = (int)(rnd(seed) % (unsigned)n_maps);
                            const int64_t keep = maps[(size_t)m][pos];
                            maps[(size_t)m][pos] = bad;
                            CHECK(slgc_host::narrow_i64_to_i16(ptrs.data(), n_maps, elems, dst.data(), thr, chunk) == 0);
                            maps[(size_t)m][pos] = keep;
                        }
                }
}

// ---------------------------------------------------------------- ring_download against a memcpy "device"
struct FakeDevice {
    const char *src;
    int slots;
    int delay_us;                       // chunks land from helper threads after this long (0: inside fetch)
    long fail_fetch_at, fail_wait_at;   // inject an error at this chunk (-1: never)
    std::vector<std::thread> dma;
    std::vector<std::atomic<int>> landed;
    std::mutex m;
    std::vector<long> slot_holds;       // which chunk each slot holds (checked: a slot is refilled only after it was waited for)
    std::atomic<int> protocol_errors{0};
    FakeDevice(const char *s, int slots_, size_t nchunks, int delay, long ff, long fw)
        : src(s), slots(slots_), delay_us(delay), fail_fetch_at(ff), fail_wait_at(fw), landed(nchunks), slot_holds((size_t)slots_, -1)
    {
        for (auto &l : landed) l.store(0);
    }
    ~FakeDevice()
    {
        for (auto &t : dma)
            if (t.joinable()) t.join();
    }
};

static int fake_fetch(void *u, size_t k, void *slot, size_t offset, size_t n)
{
    FakeDevice *d = (FakeDevice *)u;
    if ((long)k == d->fail_fetch_at) return 7;
    {
        std::lock_guard<std::mutex> g(d->m);
        long &held = d->slot_holds[k % (size_t)d->slots];
        if (held >= 0 && !d->landed[(size_t)held].load()) d->protocol_errors++;      // refilled a slot whose previous chunk had not even landed
        held = (long)k;
    }
    if (d->delay_us == 0) {
        std::memcpy(slot, d->src + offset, n);
        d->landed[k].store(1, std::memory_order_release);
    } else {
        d->dma.emplace_back([=] {
            std::this_thread::sleep_for(std::chrono::microseconds(d->delay_us));
            std::memcpy(slot, d->src + offset, n);
            d->landed[k].store(1, std::memory_order_release);
        });
    }
    return 0;
}

static int fake_wait(void *u, size_t k)
{
    FakeDevice *d = (FakeDevice *)u;
    if ((long)k == d->fail_wait_at) return 9;
    while (!d->landed[k].load(std::memory_order_acquire)) std::this_thread::yield();
    return 0;
}

static void test_ring()
{
    unsigned seed = 11;
    for (size_t chunk : {(size_t)64, (size_t)1000, (size_t)4096})
        for (int slots : {1, 2, 4})
            for (int parts : {1, 3, 8})
                for (int nthr : {1, 2, 7})
                    for (size_t bytes : {(size_t)1, chunk - 1, chunk, chunk + 1, 3 * chunk, 11 * chunk + 17, 40 * chunk + chunk / 2})     // wrap-around: up to 40 chunks through <= 4 slots
                        for (int delay : {0, 50}) {
                            if (delay && bytes > 12 * chunk) continue;            // (keeps the number of helper threads small)
                            if (g_light && (chunk != 1000 || slots == 2 || parts != 3 || nthr == 1)) continue;
                            std::vector<char> src(bytes), dst(bytes + 8, 0x55), stage((size_t)slots * chunk);
                            for (auto &c : src) c = (char)(rnd(seed) >> 24);
                            const size_t nchunks = (bytes + chunk - 1) / chunk;
                            FakeDevice dev(src.data(), slots, nchunks, delay, -1, -1);
                            const slgc_host::RingOps ops{&dev, fake_fetch, fake_wait};
                            CHECK(slgc_host::ring_download(dst.data(), bytes, stage.data(), chunk, slots, parts, nthr, ops) == 0);
                            CHECK(std::memcmp(dst.data(), src.data(), bytes) == 0);
                            CHECK(dst[bytes] == 0x55 && dst[bytes + 7] == 0x55);
                            CHECK(dev.protocol_errors.load() == 0);
                        }
    // failures of the transport: the status comes back, nothing hangs, the workers are gone when the call returns
    for (long at : {0L, 1L, 5L, 12L})
        for (int which = 0; which < 2; ++which) {
            const size_t chunk = 256, bytes = 13 * chunk + 5;
            std::vector<char> src(bytes, 1), dst(bytes), stage(4 * chunk);
            FakeDevice dev(src.data(), 4, 14, 20, which == 0 ? at : -1, which == 1 ? at : -1);
            const slgc_host::RingOps ops{&dev, fake_fetch, fake_wait};
            CHECK(slgc_host::ring_download(dst.data(), bytes, stage.data(), chunk, 4, 8, 5, ops) == (which == 0 ? 7 : 9));
        }
    // argument checks
    char b[8];
    const slgc_host::RingOps none{nullptr, nullptr, nullptr};
    CHECK(slgc_host::ring_download(b, 8, b, 4, 2, 1, 1, none) == -1);
    CHECK(slgc_host::ring_download(b, 0, nullptr, 0, 0, 0, 0, none) == 0);
}

int main(int argc, char **argv)
{
    g_light = argc > 1 && std::strcmp(argv[1], "light") == 0;
    test_narrow();
    test_narrow_maps();
    test_ring();
    if (g_failed) {
        std::fprintf(stderr, "%d check(s) failed\n", g_failed);
        return 1;
    }
    std::puts("host_util: all checks passed");
    return 0;
}
