// host_util.h -- the host-only helpers of the C-ABI's host-buffer entry points (no HIP in here: csrc/host_util.cpp builds with any C++17
// compiler, and tests/native/test_host_util.cpp runs it on the CPU under ASan / UBSan / TSan -- `make -C oracle san`).
#pragma once
#include <cstddef>
#include <cstdint>

namespace slgc_host {

// The reference's own caller hands over a float64 stack whose values are uint8 grey levels (src/3-capture_decode.py:66-70).  Checks on host
// threads that EVERY sample of n_runs stacks of `elems` float64 each is an integer in [0, 255] while writing it as one byte into dst
// (dst[r * elems + i]).  Returns 1 = dst holds all samples; 0 = some sample is not a grey level (a fraction, negative, > 255, NaN, inf: the
// first thread to see one stops the others, dst is garbage); -1 = a thread or an allocation could not be made (nothing thrown across the C-ABI).
int narrow_f64_to_u8(const void *const *stacks, int n_runs, size_t elems, uint8_t *dst, int max_threads = 16, size_t chunk_samples = 1u << 20);

// Triangulate's h_pixels / v_pixels are int64 arrays (decode_codes.py:134-146 builds them with np.zeros(..., dtype=int)) whose values are -1 or a
// projector coordinate.  Writes n_maps maps of `elems` int64 each as int16 into dst (dst[m * elems + i]) on host threads while checking that
// every value survives the narrowing (-32768 .. 32767).  Returns 1 = dst holds all values; 0 = some value does not fit (dst is garbage; the
// caller ships int64); -1 = thread / allocation failure.
int narrow_i64_to_i16(const int64_t *const *maps, int n_maps, size_t elems, int16_t *dst, int max_threads = 16, size_t chunk_values = 1u << 20);

// The transport behind the staging ring of a large device-to-host result: HIP in the library (hipMemcpyAsync into a pinned slot + an event),
// plain memcpy in the CPU tests.  Both return 0 on success.
struct RingOps {
    void *user;
    int (*fetch)(void *user, size_t chunk_index, void *slot, size_t offset, size_t nbytes);   // start bringing source bytes [offset, offset + nbytes) into `slot`
    int (*wait)(void *user, size_t chunk_index);                                              // block until that chunk has landed in its slot
};

// Results into memory nobody has touched yet: `bytes` bytes arrive chunk by chunk (`chunk` bytes each) in a ring of `slots` staging slots
// (stage = slots * chunk bytes) while `nthr` host threads copy every landed chunk, in `parts` slices, into its place in dst -- the first
// touches of dst's pages spread over the threads, the next chunk already on its way.  A slot is refilled only after all parts of the chunk it
// held have been copied out.  Returns 0, the first non-zero status of fetch / wait, or -1 (thread / allocation failure).  copy = 0: transfers
// only (timing).
int ring_download(void *dst, size_t bytes, void *stage, size_t chunk, int slots, int parts, int nthr, const RingOps &ops, int copy = 1);

}  // namespace slgc_host
