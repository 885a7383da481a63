// host_util.cpp -- see host_util.h.  Nothing here may throw across the C-ABI: every body is wrapped, failures come back as -1.
#include "host_util.h"

#include <atomic>
#include <cstring>
#include <memory>
#include <new>
#include <thread>
#include <vector>

namespace slgc_host {

int narrow_f64_to_u8(const void *const *stacks, int n_runs, size_t elems, uint8_t *dst, int max_threads, size_t chunk_samples)
{
    try {
        const size_t total = elems * (size_t)n_runs;
        if (total == 0) return 1;
        if (chunk_samples == 0) chunk_samples = 1;
        unsigned hw = std::thread::hardware_concurrency();
        size_t nthr = hw ? hw : 4;
        if (max_threads < 1) max_threads = 1;
        if (nthr > (size_t)max_threads) nthr = (size_t)max_threads;
        const size_t nchunks = (total + chunk_samples - 1) / chunk_samples;
        if (nthr > nchunks) nthr = nchunks;
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        auto work = [&]() {
            for (;;) {
                const size_t c = next.fetch_add(1, std::memory_order_relaxed);
                if (c >= nchunks || bad.load(std::memory_order_relaxed)) return;
                const size_t lo = c * chunk_samples, hi = lo + chunk_samples < total ? lo + chunk_samples : total;
                size_t i = lo;
                while (i < hi) {
                    const size_t r = i / elems, off = i - r * elems;
                    const size_t n = (hi - i) < (elems - off) ? (hi - i) : (elems - off);     // stay inside run r
                    const double *src = (const double *)stacks[r] + off;
                    uint8_t *out = dst + i;
                    unsigned wrong = 0;
                    for (size_t k = 0; k < n; ++k) {
                        const double x = src[k];
                        const bool in_range = (x >= 0.0) & (x <= 255.0);                      // false for NaN
                        const int q = in_range ? (int)x : 0;
                        out[k] = (uint8_t)q;
                        wrong |= (unsigned)(!in_range) | (unsigned)((double)q != x);
                    }
                    if (wrong) {
                        bad.store(true, std::memory_order_relaxed);
                        return;
                    }
                    i += n;
                }
            }
        };
        std::vector<std::thread> pool;
        pool.reserve(nthr);
        struct Joiner {                                   // threads already started are joined on every way out (a failed emplace_back included)
            std::vector<std::thread> &p;
            ~Joiner()
            {
                for (auto &t : p)
                    if (t.joinable()) t.join();
            }
        } joiner{pool};
        for (size_t t = 1; t < nthr; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        return bad.load() ? 0 : 1;
    } catch (...) {
        return -1;
    }
}

int narrow_i64_to_i16(const int64_t *const *maps, int n_maps, size_t elems, int16_t *dst, int max_threads, size_t chunk_values)
{
    try {
        const size_t total = elems * (size_t)n_maps;
        if (total == 0) return 1;
        if (chunk_values == 0) chunk_values = 1;
        unsigned hw = std::thread::hardware_concurrency();
        size_t nthr = hw ? hw : 4;
        if (max_threads < 1) max_threads = 1;
        if (nthr > (size_t)max_threads) nthr = (size_t)max_threads;
        const size_t nchunks = (total + chunk_values - 1) / chunk_values;
        if (nthr > nchunks) nthr = nchunks;
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        auto work = [&]() {
            for (;;) {
                const size_t c = next.fetch_add(1, std::memory_order_relaxed);
                if (c >= nchunks || bad.load(std::memory_order_relaxed)) return;
                const size_t lo = c * chunk_values, hi = lo + chunk_values < total ? lo + chunk_values : total;
                size_t i = lo;
                while (i < hi) {
                    const size_t m = i / elems, off = i - m * elems;
                    const size_t n = (hi - i) < (elems - off) ? (hi - i) : (elems - off);     // stay inside map m
                    const int64_t *src = maps[m] + off;
                    int16_t *out = dst + i;
                    uint64_t wrong = 0;
                    for (size_t k = 0; k < n; ++k) {
                        const int64_t x = src[k];
                        out[k] = (int16_t)x;
                        wrong |= ((uint64_t)x + 32768u) >> 16;                                // non-zero iff x is outside [-32768, 32767]
                    }
                    if (wrong) {
                        bad.store(true, std::memory_order_relaxed);
                        return;
                    }
                    i += n;
                }
            }
        };
        std::vector<std::thread> pool;
        pool.reserve(nthr);
        struct Joiner {
            std::vector<std::thread> &p;
            ~Joiner()
            {
                for (auto &t : p)
                    if (t.joinable()) t.join();
            }
        } joiner{pool};
        for (size_t t = 1; t < nthr; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        return bad.load() ? 0 : 1;
    } catch (...) {
        return -1;
    }
}

int ring_download(void *dst, size_t bytes, void *stage_v, size_t chunk, int slots, int parts, int nthr, const RingOps &ops, int copy)
{
    if (bytes == 0) return 0;
    if (!dst || !stage_v || chunk == 0 || slots < 1 || parts < 1 || nthr < 1 || !ops.fetch || !ops.wait) return -1;
    try {
        const size_t nchunks = (bytes + chunk - 1) / chunk;
        std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[nchunks]);
        for (size_t i = 0; i < nchunks; ++i) done[i].store(0, std::memory_order_relaxed);
        std::atomic<size_t> ready{0}, next{0};             // chunks landed so far; next (chunk, part) task to hand out
        std::atomic<bool> failed{false};
        char *stage = (char *)stage_v;
        auto work = [&]() {
            for (;;) {
                const size_t task = next.fetch_add(1, std::memory_order_relaxed);
                const size_t c = task / (size_t)parts, part = task % (size_t)parts;
                if (c >= nchunks) return;
                while (ready.load(std::memory_order_acquire) <= c) {
                    if (failed.load(std::memory_order_relaxed)) return;
                    std::this_thread::yield();
                }
                const size_t n = c + 1 < nchunks ? chunk : bytes - c * chunk;
                const size_t per = (chunk + (size_t)parts - 1) / (size_t)parts;
                const size_t lo = part * per < n ? part * per : n, hi = lo + per < n ? lo + per : n;
                if (hi > lo && copy) memcpy((char *)dst + c * chunk + lo, stage + (c % (size_t)slots) * chunk + lo, hi - lo);
                done[c].fetch_add(1, std::memory_order_release);
            }
        };
        std::vector<std::thread> pool;
        pool.reserve((size_t)nthr);
        int status = 0;
        struct Joiner {                                   // whatever happens below, the workers are released and joined before the frame goes away
            std::vector<std::thread> &p;
            std::atomic<bool> &failed;
            int &status;
            ~Joiner()
            {
                if (status) failed.store(true);
                for (auto &t : p)
                    if (t.joinable()) t.join();
            }
        } joiner{pool, failed, status};
        status = -1;                                      // until the loop below has run through
        for (int t = 0; t < nthr; ++t) pool.emplace_back(work);
        int err = 0;
        size_t landed = 0;                                 // chunks 0 .. landed-1 have been waited for and announced to the workers
        auto land_upto = [&](size_t n) {                   // wait for chunks landed .. n-1, in order
            while (!err && landed < n) {
                err = ops.wait(ops.user, landed);
                if (!err) ready.store(++landed, std::memory_order_release);
            }
        };
        for (size_t k = 0; k < nchunks && !err; ++k) {
            if (k >= (size_t)slots) {                      // the slot's previous chunk must have landed AND left it
                land_upto(k - (size_t)slots + 1);
                while (!err && done[k - (size_t)slots].load(std::memory_order_acquire) < parts) std::this_thread::yield();
            }
            if (err) break;
            const size_t n = k + 1 < nchunks ? chunk : bytes - k * chunk;
            err = ops.fetch(ops.user, k, stage + (k % (size_t)slots) * chunk, k * chunk, n);
            if (!err) land_upto(k);                        // chunk k is on its way: hand chunk k-1 to the workers meanwhile
        }
        if (!err) land_upto(nchunks);
        status = err;
        if (err) failed.store(true);
        for (auto &t : pool) t.join();
        return err;
    } catch (...) {
        return -1;
    }
}

}  // namespace slgc_host
