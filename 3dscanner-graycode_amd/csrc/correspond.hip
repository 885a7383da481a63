// correspond.hip -- validity masking + order-preserving compaction kernels (gfx950).
//
// K2a  x-major correspondence build  = Triangulate.get_cam_proj_pts (scanner/triangulation/triangulate.py:39-71):
//      scan columns outer / rows inner (:52-53), drop pixels with h == -1 or v == -1 (:56), clamp to the
//      projector (:60-61), gather colour / 255.0 (:64,:69).  Lanes run along x so every row read is coalesced;
//      the x-major rank of a pixel = (valid pixels in columns < x) + (valid pixels above it in column x), built
//      from per-(row-chunk, column) counts.
// K2b  row-major stream compaction (wave64 ballot + mbcnt prefix) used for the row-major correspondence order,
//      filter_3d_pts (triangulate.py:99-122) and the dense-XYZ -> point-list step of the multi-GPU path.
#include "slgc_internal.h"
#include "tri_math.h"

namespace {

constexpr int kChunkRows = 32;

// Diagnostic build only (make diag): SLGC_LISTS_ABL = bit mask of timing-only ablations of the x-major scatter kernel (wrong results):
// 1 no cam / proj stores, 2 no point stores, 4 no colour stores, 8 no XYZ loads, 16 no white loads, 32 records written tile after tile.
#ifdef SLGC_DIAG
#define LISTS_ABL(bit) ((abl & (bit)) != 0)
inline int lists_abl()
{
    const char *e = getenv("SLGC_LISTS_ABL");
    return e ? atoi(e) : 0;
}
#else
#define LISTS_ABL(bit) false
inline int lists_abl() { return 0; }
#endif

__device__ __forceinline__ bool decodable(int64_t h, int64_t v) { return !(h == -1 || v == -1); }  // :56

// ---- x-major: pass A, per (chunk, column) counts ----
// A workgroup = 64 columns x one chunk of 32 rows; its 4 waves take 8 rows each with every load issued before the first use (one
// memory round trip per workgroup), and meet in LDS.
template <typename MapT>
__global__ void __launch_bounds__(256) k_xmajor_count(const MapT *__restrict__ h, const MapT *__restrict__ v, int W, int H,
                                                      unsigned *__restrict__ counts)
{
    __shared__ unsigned part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane, chunk = blockIdx.y;
    const int y0 = chunk * kChunkRows + wave * (kChunkRows / 4);
    const int xc = min(x, W - 1);                      // every load unconditional on a clamped address, masked afterwards:
    MapT hv[kChunkRows / 4], vv[kChunkRows / 4];       // the 16 loads of a lane are in flight together
#pragma unroll
    for (int i = 0; i < kChunkRows / 4; ++i) {
        const size_t p = (size_t)min(y0 + i, H - 1) * W + xc;
        hv[i] = h[p];
        vv[i] = v[p];
    }
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < kChunkRows / 4; ++i) c += (x < W && y0 + i < H && decodable(hv[i], vv[i])) ? 1u : 0u;
    part[wave][lane] = c;
    __syncthreads();
    if (wave == 0 && x < W) counts[(size_t)chunk * W + x] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}

// ---- x-major: pass B1, prefix over the row chunks of each column + column totals ----
// A workgroup = 64 columns x 16 groups of consecutive chunks (lanes along x: coalesced).  Every thread sums its group, the groups meet in
// LDS, then every thread rewrites its group's counts as "valid pixels above this chunk in column x".
constexpr int kPrefixGroups = 16;
__global__ void __launch_bounds__(64 * kPrefixGroups) k_xmajor_colprefix(unsigned *__restrict__ counts, int W, int nchunks,
                                                                         unsigned long long *__restrict__ colstart)
{
    __shared__ unsigned part[kPrefixGroups][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int per = (nchunks + kPrefixGroups - 1) / kPrefixGroups;
    const int c0 = min(nchunks, g * per), c1 = min(nchunks, c0 + per);
    unsigned sum = 0;
    if (x < W) {
#pragma unroll 8
        for (int c = c0; c < c1; ++c) sum += counts[(size_t)c * W + x];
    }
    part[g][lane] = sum;
    __syncthreads();
    unsigned run = 0, all = 0;
#pragma unroll
    for (int i = 0; i < kPrefixGroups; ++i) {
        const unsigned n = part[i][lane];
        run += i < g ? n : 0u;
        all += n;
    }
    if (x >= W) return;
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
        const unsigned n = counts[(size_t)c * W + x];
        counts[(size_t)c * W + x] = run;  // valid pixels above this chunk in column x
        run += n;
    }
    if (g == 0) colstart[x] = all;  // column total for now
}

// ---- x-major: pass B2, one workgroup: exclusive scan of the column totals ----
__global__ void __launch_bounds__(1024) k_xmajor_colscan(int W, unsigned long long *__restrict__ colstart, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (W + 1023) / 1024;                 // each thread owns a contiguous slab of columns: the scan stays ordered
    const int x0 = min(W, t * per), x1 = min(W, x0 + per);
    unsigned long long mine = 0;
    for (int x = x0; x < x1; ++x) mine += colstart[x];
    unsigned long long inc = mine;                     // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned long long n = wsum[w];
        before += w < wave ? n : 0ull;
        all += n;
    }
    if (t == 0) *total = all;
    unsigned long long acc = before + inc - mine;
    for (int x = x0; x < x1; ++x) {
        const unsigned long long n = colstart[x];
        colstart[x] = acc;
        acc += n;
    }
}

// ---- x-major: pass C, scatter through an LDS transpose ----
// A workgroup (512 threads) owns a tile of TC columns x TR = 2048 / TC rows (64 x 32).
//   Phase 1 reads it with lanes along the row bytes of the tile -- 128 B of each int16 map, 768 B of XYZ, 192 B of the white image per row,
//   every load unconditional (clamped addresses) and in flight before the first LDS store: one memory round trip per workgroup -- and
//   parks the clamped projector coordinates, the dense XYZ and the colour bytes in LDS.
//   Phase 2 turns the tile: a half-wave takes one column; its 32 lanes are the 32 rows of a segment, a ballot gives every valid pixel its
//   rank inside the segment, and the segment leaves as contiguous runs of records (x-major order = column after column, rows ascending)
//   -- the 24-byte colour records compacted and re-spread over the lanes with ds_permute / ds_bpermute so that every store instruction
//   writes one contiguous run.
// What bounds it (tools/ubench/write_patterns.hip): 69 % of the bytes are stores, a segment is ~26 records (208 B of
// an 8-byte stream) starting on an arbitrary 8-byte boundary, and MI355X takes partially written 128-byte lines at 2.3 TB/s against 5.0
// TB/s for aligned runs; neighbours arriving back to back from one wave are merged on the way (4.8 TB/s with one output stream), but with
// the 8 interleaved streams of this product that drops to 3.6 TB/s -- which is where this kernel sits.  Taller tiles (TC = 32, 16, 8:
// one half-wave writes 2, 4, 8 abutting segments of its column one after the other) measured the same or slower: the narrower rows
// cost on the read side what the longer runs gain.
#ifndef SLGC_SCATTER_TC
#define SLGC_SCATTER_TC 64
#endif
constexpr int kScatterThreads = 512, kTilePixels = 2048, kSegRows = 32;
constexpr unsigned kInvalidHV = 0xffffffffu;           // (pu, pv) = (-1, -1): pu = min(h, proj_w - 1) with h != -1 is never -1
constexpr int kInvalid = (int)0x80000000;
static_assert(kSegRows == kChunkRows, "segment bases come from the per-chunk column prefixes");

// b / 255.0 exactly as the reference's float64 division rounds it (triangulate.py:64,:69), for b = 0..255: one refinement step on
// b * (1 / 255) lands on the correctly rounded quotient for all 256 bytes (tests/test_abi_cpu.py checks the arithmetic exhaustively).
__device__ __forceinline__ double unit_of_byte(unsigned b)
{
    constexpr double rcp = 1.0 / 255.0;
    const double x = (double)b, q = x * rcp;
    return __builtin_fma(__builtin_fma(-q, 255.0, x), rcp, q);
}

// SRC = 1 (device-resident product, slgc_cloud_lists_dev): the dense float32 XYZ of the scan rides through the same transpose and
// leaves as the reference's float64 (3,M) array (triangulate.py:95; M = *total, written by the column scan before this kernel runs).
// SRC = 2 (slgc_cloud_dev): there is no dense XYZ -- the tile's camera rays ride through the transpose (per-pixel table, or the 19 nodes per
// row the tile's 16 groups interpolate from) and every valid pixel is triangulated where it is written: projector ray gathered from its table,
// triangulate1 = the fused scan kernel's arithmetic, bit for bit.  Saves the scan 12 B/pixel of XYZ written and 12 B/pixel read back.
struct TriScatter {
    const float2 *cam_lut;     // [H][W] exact camera rays
    CamNodes cn;               // every-4th-column nodes (cn.nodes == nullptr: read cam_lut)
    const float2 *proj_lut;      // projector rays (guarded redo)
    const float *proj_th;        // tan(beta / 2) per projector pixel, same index (fast form)
    int ptiles_x, wide;
    int f32;                   // slgc_cloud32_dev: points and colours leave as float32 (12 instead of 24 bytes each per point) -- NOT the reference's dtypes
    TriF32 kf;
    double T[3], t_len;
};

template <typename MapT, int SRC, int TC>
__global__ void __launch_bounds__(kScatterThreads, SRC == 1 ? 6 : 4)
k_xmajor_scatter(const MapT *__restrict__ h, const MapT *__restrict__ v, int W, int H, int proj_w, int proj_h, const uint8_t *__restrict__ white,
                 const unsigned *__restrict__ counts, const unsigned long long *__restrict__ colstart, float *__restrict__ cam,
                 float *__restrict__ proj, double *__restrict__ colors, const float *__restrict__ xyz, double *__restrict__ pts,
                 const unsigned long long *__restrict__ total, int tiles_x, int abl, const TriScatter ts, int tiles_y, int order)
{
    constexpr bool XYZ = SRC == 1, TRI = SRC == 2;
    static_assert(!TRI || TC == 64, "the in-kernel triangulation is written for 64-column tiles (16 four-pixel groups + 3 nodes per row)");
    constexpr bool PACKED = sizeof(MapT) == 2;          // int16 maps: (pu, pv) share a dword; int64 maps (API parity) keep 32 bits each
    constexpr int TR = kTilePixels / TC, NSEG = TR / kSegRows;
    constexpr int CPH = TC >= 16 ? TC / 16 : 1;         // columns one half-wave writes
    constexpr int SPLIT = TC >= 16 ? 1 : 16 / TC;       // half-waves that share a column (each a run of consecutive segments)
    constexpr int SPH = NSEG / SPLIT;                   // segments of a column one half-wave writes
    constexpr int WD = 3 * TC / 4;                      // dwords of white bytes per tile row
    constexpr int NM = kTilePixels / kScatterThreads, NX = 3 * kTilePixels / kScatterThreads, NW = (TR * WD + kScatterThreads - 1) / kScatterThreads;
    __shared__ unsigned s_hv[TR][TC + 1];
    __shared__ int s_pv[PACKED ? 1 : TR][PACKED ? 1 : TC + 1];
    __shared__ unsigned s_white[TR][WD + 1];            // +1: rows land in different banks
    __shared__ float s_xyz[XYZ ? TR : 1][XYZ ? 3 * TC + 1 : 1];
    __shared__ float2 s_cam[TRI ? TR : 1][TRI ? TC + 1 : 1];      // the tile's camera rays -- or, with the node table, nodes 0..18 of each row in [row][0..18]
    __shared__ unsigned long long s_base[NSEG][TC];     // where each column segment's first record goes
    const int tid = threadIdx.x, lane = tid & 63;
    // Which tile this workgroup takes.  order 0: row-major.  The output is x-major -- a column's records are contiguous, so the run a tile
    // writes for column x continues in the tile BELOW it, and the 128-byte lines at the seams are completed by that other workgroup.  order 2
    // walks the tiles column-major inside each XCD (workgroup ids go round-robin over the 8 XCDs, each with its own L2): the two halves of a
    // seam line are written by consecutive workgroups of one XCD and meet in its L2 instead of leaving it as two partial lines.  order 1:
    // column-major without the XCD grouping (A/B).
    uint32_t tile = blockIdx.x;
    if (order == 2) tile = xcd_block(blockIdx.x, gridDim.x / 8u);
    const int x_tile = (order ? (int)(tile / (unsigned)tiles_y) : (int)(tile % (unsigned)tiles_x)) * TC;
    const int y_tile = (order ? (int)(tile % (unsigned)tiles_y) : (int)(tile / (unsigned)tiles_x)) * TR;
    const int cols = min(TC, W - x_tile);
    const size_t npix = (size_t)W * H;
    const bool white_dwords = colors && !LISTS_ABL(16) && (((uintptr_t)white | (unsigned)W) & 3u) == 0 && npix >= 2;   // row starts dword aligned
    const bool white_bytes = colors && !LISTS_ABL(16) && !white_dwords;
    unsigned long long M = 0;
    if constexpr (XYZ || TRI) M = *total;
    const bool nodes = TRI && ts.cn.nodes != nullptr;

    // Phase 1: unconditional loads on clamped addresses (rows past H re-read row H - 1, elements past the image its last one; phase 2
    // never looks at those slots), all of a lane's loads in flight together.
    MapT hq[NM], vq[NM];
    float xq[XYZ ? NX : 1];
    unsigned wq[NW];
    unsigned long long bq = 0;
#pragma unroll
    for (int q = 0; q < NM; ++q) {
        const int i = q * kScatterThreads + tid, row = i / TC, col = i % TC;
        const size_t p = min((size_t)min(y_tile + row, H - 1) * W + x_tile + col, npix - 1);
        hq[q] = h[p];
        vq[q] = v[p];
    }
    if constexpr (XYZ) {
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int i = q * kScatterThreads + tid, row = i / (3 * TC), k = i % (3 * TC);
            xq[q] = LISTS_ABL(8) ? 0.0f : xyz[min(3 * ((size_t)min(y_tile + row, H - 1) * W + x_tile) + k, 3 * npix - 1)];
        }
    }
    float2 cq[TRI ? NM : 1];
    if constexpr (TRI) {
        if (nodes) {                                        // 19 nodes per row: node n of the table row sits at x = 4 (n - 1); the tile's groups need x_tile / 4 .. + 18
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = min(q * kScatterThreads + tid, TR * 19 - 1), row = i / 19, n = i % 19;
                cq[q] = ts.cn.nodes[(size_t)min(y_tile + row, H - 1) * ts.cn.ne + min((uint32_t)(x_tile / 4 + n), ts.cn.ne - 1u)];
            }
        } else {
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                const int i = q * kScatterThreads + tid, row = i / TC, col = i % TC;
                cq[q] = ts.cam_lut[min((size_t)min(y_tile + row, H - 1) * W + x_tile + col, npix - 1)];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) wq[q] = 0u;
    if (white_dwords) {
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int i = min(q * kScatterThreads + tid, TR * WD - 1), row = i / WD, d = i % WD;
            wq[q] = *reinterpret_cast<const unsigned *>(white + min(3 * ((size_t)min(y_tile + row, H - 1) * W + x_tile) + 4 * d, 3 * npix - 4));
        }
    } else if (white_bytes) {
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int i = min(q * kScatterThreads + tid, TR * WD - 1), row = i / WD, d = i % WD;
            const size_t o = 3 * ((size_t)min(y_tile + row, H - 1) * W + x_tile) + 4 * d;
            unsigned b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) b[k] = white[min(o + k, 3 * npix - 1)];
            wq[q] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        }
    }
    if (tid < NSEG * TC) {
        const int seg = tid / TC, xb = min(x_tile + tid % TC, W - 1), chunk = min(y_tile / kChunkRows + seg, (H - 1) / kChunkRows);
        bq = colstart[xb] + counts[(size_t)chunk * W + xb];
    }
#pragma unroll
    for (int q = 0; q < NM; ++q) {
        const int i = q * kScatterThreads + tid, row = i / TC, col = i % TC;
        const int64_t hv = hq[q], vv = vq[q];
        const bool ok = col < cols && y_tile + row < H && decodable(hv, vv);
        const int pu = (int)(hv < proj_w - 1 ? hv : proj_w - 1);                    // triangulate.py:60 (maps from this library fit 32 bits)
        const int pv = (int)(vv < proj_h - 1 ? vv : proj_h - 1);                    // :61
        if constexpr (PACKED) {
            s_hv[row][col] = ok ? ((unsigned)pu & 0xffffu) | ((unsigned)pv << 16) : kInvalidHV;
        } else {
            s_hv[row][col] = ok ? (unsigned)(pu == kInvalid ? kInvalid + 1 : pu) : (unsigned)kInvalid;
            s_pv[row][col] = pv;
        }
    }
    if constexpr (XYZ) {
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int i = q * kScatterThreads + tid;
            s_xyz[i / (3 * TC)][i % (3 * TC)] = xq[q];
        }
    }
    if constexpr (TRI) {
        if (nodes) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = q * kScatterThreads + tid;
                if (i < TR * 19) s_cam[i / 19][i % 19] = cq[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                const int i = q * kScatterThreads + tid;
                s_cam[i / TC][i % TC] = cq[q];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const int i = q * kScatterThreads + tid;
        if (i < TR * WD) s_white[i / WD][i % WD] = wq[q];
    }
    if (tid < NSEG * TC) s_base[tid / TC][tid % TC] = bq;
    __syncthreads();

    // Phase 2
    const int half = lane >> 5, r = lane & 31, hw = tid >> 5;
    // colour doubles of a segment, re-spread: lane r of round k writes double j = 32 k + r = channel j % 3 of record j / 3
    int rec_lane[3], ch_shift[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int j = k * 32 + r;
        rec_lane[k] = (half * 32 + j / 3) * 4;
        ch_shift[k] = 8 * (j % 3);
    }
    // SRC == 2: the projector rays of this lane's CPH pixels are requested now, all in flight together (unconditional: an undecodable pixel
    // reads entry 0, unused) -- not one memory round trip per column inside the loop
    float prj[TRI ? CPH : 1];
    if constexpr (TRI) {
        static_assert(!TRI || (SPLIT == 1 && SPH == 1), "one segment per column and half-wave");
#pragma unroll
        for (int j = 0; j < CPH; ++j) {
            const unsigned hv = s_hv[r][hw + 16 * j];
            const bool ok = hv != kInvalidHV;
            prj[j] = ts.proj_th[ok ? proj_lut_index((int)(short)(hv & 0xffffu), (int)(short)(hv >> 16), ts.ptiles_x, ts.wide) : 0u];
        }
    }
#pragma unroll
    for (int j = 0; j < CPH; ++j) {
        const int c = SPLIT == 1 ? hw + 16 * j : hw % TC, x = x_tile + c;           // this half-wave's column
#pragma unroll
        for (int q = 0; q < SPH; ++q) {
            const int seg = SPLIT == 1 ? q : (hw / TC) * SPH + q, row = seg * kSegRows + r;
            const unsigned hv = s_hv[row][c];
            const bool ok = PACKED ? hv != kInvalidHV : hv != (unsigned)kInvalid;
            const unsigned long long m = __ballot(ok);
            const unsigned mh = (unsigned)(half ? (m >> 32) : m);
            const unsigned rank = (unsigned)__popc(mh & ((1u << r) - 1u)), n = (unsigned)__popc(mh);
            const unsigned long long o0 = LISTS_ABL(32) ? ((unsigned long long)blockIdx.x * TC + c) * TR + seg * kSegRows : s_base[seg][c];
            if (ok) {
                const unsigned long long o = o0 + rank;
                const int pu = PACKED ? (int)(short)(hv & 0xffffu) : (int)hv;
                const int pv = PACKED ? (int)(short)(hv >> 16) : s_pv[PACKED ? 0 : row][PACKED ? 0 : c];
                if (cam && !LISTS_ABL(1)) {              // (slgc_cloud_dev may leave the two correspondence lists out: the points and colours are the product)
                    reinterpret_cast<float2 *>(cam)[o] = make_float2((float)x, (float)(y_tile + row));          // :59 [i, j] = (x, y)
                    reinterpret_cast<float2 *>(proj)[o] = make_float2((float)pu, (float)pv);
                }
                if (XYZ && !LISTS_ABL(2)) {
                    pts[o] = (double)s_xyz[row][3 * c];                              // Pts (3,M) float64, :95
                    pts[M + o] = (double)s_xyz[row][3 * c + 1];
                    pts[2 * M + o] = (double)s_xyz[row][3 * c + 2];
                }
                if constexpr (TRI) {
                    // triangulate.py:84-95 for this correspondence, as the fused scan kernel evaluates it (tri_math.h): camera ray of pixel
                    // (x, y_tile + row), projector ray of the clamped (pu, pv) from its table
                    const size_t pix = (size_t)(y_tile + row) * W + x;
                    float cxr, cyr;
                    if (nodes) {
                        const int g = c >> 2;
                        const float2 n0 = s_cam[row][g], n1 = s_cam[row][g + 1], n2 = s_cam[row][g + 2], n3 = s_cam[row][g + 3];
                        float fx[4], fy[4];
                        cam_rays_from_nodes(cam_v4f{n0.x, n0.y, n1.x, n1.y}, cam_v4f{n2.x, n2.y, n3.x, n3.y}, fx, fy);
                        cam_rays_exact_where_tiny(fx, fy, ts.cam_lut + (pix - (size_t)(c & 3)));      // the group's decision, like the lane that owns it in the scan kernels
                        const int j = c & 3;
                        cxr = j == 0 ? fx[0] : j == 1 ? fx[1] : j == 2 ? fx[2] : fx[3];
                        cyr = j == 0 ? fy[0] : j == 1 ? fy[1] : j == 2 ? fy[2] : fy[3];
                    } else {
                        const float2 cr = s_cam[row][c];
                        cxr = cr.x;
                        cyr = cr.y;
                    }
                    const float pr = prj[j];
                    const Xyzf r3 = triangulate1<true>(cxr, cyr, pr, ts.kf, ts.T, ts.t_len, ts.cam_lut + pix, ts.proj_lut + proj_lut_index(pu, pv, ts.ptiles_x, ts.wide));
                    if (!LISTS_ABL(2)) {
                        if (ts.f32) {                                                // (3,M) float32 (slgc_cloud32_dev)
                            float *p32 = reinterpret_cast<float *>(pts);
                            p32[o] = r3.x;
                            p32[M + o] = r3.y;
                            p32[2 * M + o] = r3.z;
                        } else {
                            pts[o] = (double)r3.x;                                   // Pts (3,M) float64, :95
                            pts[M + o] = (double)r3.y;
                            pts[2 * M + o] = (double)r3.z;
                        }
                    }
                }
            }
            if (colors && !LISTS_ABL(4)) {
                // compact the segment's colours over the lanes of the half-wave: valid pixels to lane `rank`, the others behind them
                const uint8_t *wb = reinterpret_cast<const uint8_t *>(&s_white[row][0]) + 3 * c;
                const unsigned rgb = (unsigned)wb[0] | ((unsigned)wb[1] << 8) | ((unsigned)wb[2] << 16);
                const unsigned dst = ok ? rank : n + ((unsigned)r - rank);
                const unsigned packed = (unsigned)__builtin_amdgcn_ds_permute((int)((half * 32 + dst) * 4), (int)rgb);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const unsigned px = (unsigned)__builtin_amdgcn_ds_bpermute(rec_lane[k], (int)packed);
                    if ((unsigned)(k * 32 + r) < 3 * n) {
                        const double u = unit_of_byte((px >> ch_shift[k]) & 0xffu);                             // :64, :69
                        if (TRI && ts.f32) reinterpret_cast<float *>(colors)[3 * o0 + k * 32 + r] = (float)u;
                        else colors[3 * o0 + k * 32 + r] = u;
                    }
                }
            }
        }
    }
}

// ---- x-major: pass C', the scatter of slgc_cloud_dev that writes WHOLE 128-byte lines ----
// What bounds k_xmajor_scatter is not the bytes but their shape: a tile's run of a column starts and ends on arbitrary 8-byte boundaries, and
// MI355X takes lines that arrive in two pieces from two workgroups at 3.6 TB/s where whole aligned lines go at 5.2-5.7 TB/s, 8 interleaved
// streams included (tools/ubench/write_patterns.hip, patterns H / I).  Here the unit of output is the GROUP = 16 consecutive records of the
// x-major order, aligned on the ABSOLUTE record index: 128 bytes of cam_pts, of proj_pts and of each plane of Pts, 384 bytes of colours.
//   A workgroup loads 64 rows of its 64 columns: its 32 NOMINAL rows (the tile of k_xmajor_scatter) and the 32 rows below (the next tile's).
//   Per column it OWNS the groups whose first record (of that column) lies in its nominal rows and writes every record of them it can see
//   -- the group that straddles the seam is completed from the rows below instead of being left half-written.  The records at the head of
//   its nominal rows that belong to a group begun in the tile above are that tile's business (it sees them); only when the group began even
//   earlier (fewer than 16 valid pixels in 32 rows: sparse columns) does a tile write its own records of a foreign group, as a partial line
//   like before.  Every record is written exactly once, by a rule each tile evaluates from the same per-chunk prefix counts.
//   A half-wave takes one column: ballots rank the valid pixels of the 64 rows, their row numbers go into a 64-byte list in LDS, and lane r
//   of pass p handles record (first group start) + 32 p + r -- whichever row that is: the tile sits in LDS, any row is one read away.  So every
//   store instruction covers an aligned window of the output, full except at column ends and in sparse columns.
// Costs: the maps / white image / camera nodes are read twice (+ 0.16 GB at 4096x3000), the points are still triangulated once.
constexpr int kLinesRows = 64, kLinesNominal = 32;                   // rows loaded (nominal + halo), nominal rows
static_assert(kLinesRows == 2 * kLinesNominal, "the ownership rule needs a tile to see ALL nominal rows of the tile below (a shorter halo gave 4 workgroups "
                                               "per CU instead of 3 and the same time, 218 us, with wrong seams)");
static_assert(kLinesNominal == kChunkRows, "the nominal rows of a tile are one chunk of the prefix counts");

template <bool NODES>
__global__ void __launch_bounds__(kScatterThreads, 4)
k_xmajor_lines(const int16_t *__restrict__ h, const int16_t *__restrict__ v, int W, int H, int proj_w, int proj_h, const uint8_t *__restrict__ white,
               const unsigned *__restrict__ counts, const unsigned long long *__restrict__ colstart, float *__restrict__ cam, float *__restrict__ proj,
               double *__restrict__ colors, double *__restrict__ pts, const unsigned long long *__restrict__ total, int tiles_x, const TriScatter ts,
               int tiles_y, int order, int abl)
{
    constexpr int TC = 64, TR = kLinesRows, NR = kLinesNominal, WD = 3 * TC / 4, NN = 19;
    constexpr int NMP = (TR * TC / 2 + kScatterThreads - 1) / kScatterThreads;                  // column pairs of each map per thread (4)
    constexpr int NW = (TR * WD + kScatterThreads - 1) / kScatterThreads;                       // white dwords per thread (6)
    constexpr int NQ = (TR * NN + kScatterThreads - 1) / kScatterThreads;
    __shared__ unsigned s_hv[TR][TC + 1];
    __shared__ unsigned s_white[TR][WD + 1];
    __shared__ float2 s_cam[NODES ? TR : 1][NODES ? NN + 2 : 1];        // nodes x_tile / 4 - 1 .. + 17 of every row (21 float2 per row: rows 16 apart share banks)
    __shared__ unsigned s_b0[TC], s_bprev[TC], s_cs[TC];               // record indices fit 32 bits (the launcher checks the image size)
    __shared__ uint8_t s_list[TC][TR];                                  // per column: the rows of its valid pixels, in rank order
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t tile = blockIdx.x;
    if (order == 2) tile = xcd_block(blockIdx.x, gridDim.x / 8u);
    if (order >= 3) {                                    // column-major, runs of `order` consecutive tiles on one XCD (workgroup ids go round-robin over the 8 XCDs):
        const uint32_t R = (uint32_t)order, span = 8u * R, g = tile / span;             // a tile and the tile below it -- whose rows it also reads -- share an L2
        if ((g + 1u) * span <= gridDim.x) tile = g * span + (tile % 8u) * R + (tile / 8u) % R;
    }
    const int tx = order ? (int)(tile / (unsigned)tiles_y) : (int)(tile % (unsigned)tiles_x);
    const int ty = order ? (int)(tile % (unsigned)tiles_y) : (int)(tile / (unsigned)tiles_x);
    const int x_tile = tx * TC, y_tile = ty * NR;
    const int cols = min(TC, W - x_tile);
    const unsigned npix = (unsigned)W * (unsigned)H;
    const unsigned M = (unsigned)*total;
    // every address below = a workgroup-uniform base + a 32-bit byte offset (scalar base, one vector add per access)
    const unsigned tile0 = (unsigned)y_tile * (unsigned)W + (unsigned)x_tile;                       // first pixel of the tile
    const char *h_t = reinterpret_cast<const char *>(h + tile0), *v_t = reinterpret_cast<const char *>(v + tile0);
    const char *w_t = reinterpret_cast<const char *>(white) + 3 * (size_t)tile0;
    const int rows_in = H - 1 - y_tile;                                                              // last row of the image, relative to the tile

    // Phase 1: every load unconditional on a clamped address and in flight before the first LDS store (W % 4 == 0: column pairs and the
    // white bytes of a tile row are dword aligned)
    unsigned hq[NMP], vq[NMP], wq[NW];
    float2 cq[NODES ? NQ : 1];
#pragma unroll
    for (int q = 0; q < NMP; ++q) {
        const int i = min(q * kScatterThreads + tid, TR * TC / 2 - 1), row = i / (TC / 2), cp = i % (TC / 2);
        const unsigned e = min((unsigned)min(row, rows_in) * (unsigned)W + 2u * cp, npix - 2u - tile0);
        hq[q] = *reinterpret_cast<const unsigned *>(h_t + 2u * e);
        vq[q] = *reinterpret_cast<const unsigned *>(v_t + 2u * e);
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) wq[q] = 0u;
    if (colors) {
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int i = min(q * kScatterThreads + tid, TR * WD - 1), row = i / WD, d = i % WD;
            wq[q] = *reinterpret_cast<const unsigned *>(w_t + min(3u * (unsigned)min(row, rows_in) * (unsigned)W + 4u * d, 3u * (npix - tile0) - 4u));
        }
    }
    if constexpr (NODES) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = min(q * kScatterThreads + tid, TR * NN - 1), row = i / NN, n = i % NN;
            cq[q] = ts.cn.nodes[(unsigned)min(y_tile + row, H - 1) * ts.cn.ne + min((uint32_t)(x_tile / 4 + n), ts.cn.ne - 1u)];
        }
    }
    unsigned b0 = 0, bprev = 0, cs = 0;
    if (tid < TC) {
        const unsigned xb = (unsigned)min(x_tile + tid, W - 1);
        cs = (unsigned)colstart[xb];
        b0 = cs + counts[(unsigned)ty * (unsigned)W + xb];
        bprev = ty > 0 ? cs + counts[(unsigned)(ty - 1) * (unsigned)W + xb] : b0;
    }
#pragma unroll
    for (int q = 0; q < NMP; ++q) {
        const int i = q * kScatterThreads + tid, row = i / (TC / 2), cp = i % (TC / 2);
        if (i < TR * TC / 2)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int col = 2 * cp + k;
            const int hv = (int)(short)(hq[q] >> (16 * k)), vv = (int)(short)(vq[q] >> (16 * k));
            const bool ok = col < cols && y_tile + row < H && decodable(hv, vv);
            const int pu = min(hv, proj_w - 1), pv = min(vv, proj_h - 1);                // triangulate.py:60-61
            s_hv[row][col] = ok ? ((unsigned)pu & 0xffffu) | ((unsigned)pv << 16) : kInvalidHV;
        }
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const int i = q * kScatterThreads + tid;
        if (i < TR * WD) s_white[i / WD][i % WD] = wq[q];
    }
    if constexpr (NODES) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = q * kScatterThreads + tid;
            if (i < TR * NN) s_cam[i / NN][i % NN] = cq[q];
        }
    }
    if (tid < TC) {
        s_b0[tid] = b0;
        s_bprev[tid] = bprev;
        s_cs[tid] = cs;
    }
    __syncthreads();

    // Phase 2
    const int half = lane >> 5, r = lane & 31, hw = tid >> 5;
    constexpr int CPH = TC / 16;                                   // columns of a half-wave: c = hw + 16 j
    int rec_of[3], ch_of[3];                                       // colour double 32 q + r of a pass = channel ch_of[q] of its record rec_of[q]
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        rec_of[q] = (32 * q + r) / 3;
        ch_of[q] = (32 * q + r) % 3;
    }
    // (a) per column: rank the valid pixels of the 64 rows into the list, and which records [ka, kb) (as ranks inside that list; absolute
    //     index = B0 + rank) this tile writes -- the rule above; kA = the 16-aligned start of the window the passes walk (may be negative)
    int win[CPH];                                                  // (kA + 16) | ka << 8 | kb << 16
#pragma unroll
    for (int j = 0; j < CPH; ++j) {
        const int c = hw + 16 * j;
        const bool ok1 = s_hv[r][c] != kInvalidHV, ok2 = NR + r < TR && s_hv[min(NR + r, TR - 1)][c] != kInvalidHV;
        const unsigned long long m1 = __ballot(ok1), m2 = __ballot(ok2);
        const unsigned mh1 = (unsigned)(half ? (m1 >> 32) : m1), mh2 = (unsigned)(half ? (m2 >> 32) : m2), below = (1u << r) - 1u;
        const int n1 = __popc(mh1), n2 = __popc(mh2);
        if (ok1) s_list[c][__popc(mh1 & below)] = (uint8_t)r;
        if (ok2) s_list[c][n1 + __popc(mh2 & below)] = (uint8_t)(NR + r);
        const unsigned B0 = s_b0[c], Bp = s_bprev[c], S = max(B0 & ~15u, s_cs[c]);                  // S = first record (of this column) of the group B0 is in
        const int phase = (int)(B0 & 15u), to_next = 16 - phase;                                     // records left in that group
        const int head_end = min(to_next, n1);
        int ka = 0;
        if (S != B0 && ty > 0 && S >= Bp) ka = head_end;           // the tile above began that group and sees these rows: its records
        const bool owns = n1 > 0 && (S == B0 || to_next < n1);
        const int kb = c >= cols ? ka : owns ? min(((phase + n1 - 1) & ~15) + 16 - phase, n1 + n2) : head_end;
        const int kA = ((phase + ka) & ~15) - phase;
        win[j] = (kA + 16) | (ka << 8) | (kb << 16);
    }
    wave_lds_sync();                                               // every list is written and read by one wave only
    // (b) first pass of every column: whichever rows its records are, and their projector rays requested -- all CPH gathers in flight together
    int rowk[CPH];
    float prj[CPH];
#pragma unroll
    for (int j = 0; j < CPH; ++j) {
        const int c = hw + 16 * j, k = (win[j] & 0xff) - 16 + r;
        const bool act = k >= ((win[j] >> 8) & 0xff) && k < (win[j] >> 16);
        const int row = s_list[c][min(max(k, 0), TR - 1)];
        const unsigned hv = s_hv[row][c];
        rowk[j] = act ? row : -1;
        prj[j] = ts.proj_th[act ? proj_lut_index((int)(short)(hv & 0xffffu), (int)(short)(hv >> 16), ts.ptiles_x, ts.wide) : 0u];
    }
    // (c) one record per lane and pass
    char *const cam_b = reinterpret_cast<char *>(cam), *const proj_b = reinterpret_cast<char *>(proj), *const col_b = reinterpret_cast<char *>(colors);
    char *const p0_b = reinterpret_cast<char *>(pts), *const p1_b = reinterpret_cast<char *>(pts + M), *const p2_b = reinterpret_cast<char *>(pts + 2 * (size_t)M);
    auto emit = [&](int c, int kp, int ka, int kb, int row, float pr) {
        const int x = x_tile + c;
        const unsigned B0 = s_b0[c];
        if (row >= 0) {
            const unsigned o8 = (B0 + (unsigned)(kp + r)) * 8u;                                      // byte offset of the record in the 8-byte streams
            const unsigned hv = s_hv[row][c];
            const int pu = (int)(short)(hv & 0xffffu), pv = (int)(short)(hv >> 16);
            const unsigned pix = (unsigned)(y_tile + row) * (unsigned)W + (unsigned)x;
            float cxr, cyr;
            if constexpr (NODES) {
                const int g = c >> 2;
                const float2 n0 = s_cam[row][g], n1_ = s_cam[row][g + 1], n2_ = s_cam[row][g + 2], n3 = s_cam[row][g + 3];
                float fx[4], fy[4];
                cam_rays_from_nodes(cam_v4f{n0.x, n0.y, n1_.x, n1_.y}, cam_v4f{n2_.x, n2_.y, n3.x, n3.y}, fx, fy);
                cam_rays_exact_where_tiny(fx, fy, ts.cam_lut + (pix - (unsigned)(c & 3)));            // the group's decision, like the lane that owns it in the scan kernels
                const int jj = c & 3;
                cxr = jj == 0 ? fx[0] : jj == 1 ? fx[1] : jj == 2 ? fx[2] : fx[3];
                cyr = jj == 0 ? fy[0] : jj == 1 ? fy[1] : jj == 2 ? fy[2] : fy[3];
            } else {
                const float2 cr = ts.cam_lut[pix];
                cxr = cr.x;
                cyr = cr.y;
            }
            if (cam && !LISTS_ABL(1)) {
                *reinterpret_cast<float2 *>(cam_b + o8) = make_float2((float)x, (float)(y_tile + row));          // :59 [i, j] = (x, y)
                *reinterpret_cast<float2 *>(proj_b + o8) = make_float2((float)pu, (float)pv);
            }
            const Xyzf r3 = triangulate1<true>(cxr, cyr, pr, ts.kf, ts.T, ts.t_len, ts.cam_lut + pix, ts.proj_lut + proj_lut_index(pu, pv, ts.ptiles_x, ts.wide));
            if (!LISTS_ABL(2)) {
                if (ts.f32) {                                                                                   // (3,M) float32 (slgc_cloud32_dev): a group of 16 records = one 64-byte piece
                    const unsigned o4 = o8 >> 1;
                    *reinterpret_cast<float *>(p0_b + o4) = r3.x;
                    *reinterpret_cast<float *>(p0_b + (size_t)M * 4 + o4) = r3.y;
                    *reinterpret_cast<float *>(p0_b + (size_t)M * 8 + o4) = r3.z;
                } else {
                    *reinterpret_cast<double *>(p0_b + o8) = (double)r3.x;                                      // Pts (3,M) float64, :95
                    *reinterpret_cast<double *>(p1_b + o8) = (double)r3.y;
                    *reinterpret_cast<double *>(p2_b + o8) = (double)r3.z;
                }
            } else if (r3.x == 12345.678f) {
                *reinterpret_cast<double *>(p0_b + o8) = 0.0;
            }
        }
        if (colors && !LISTS_ABL(4)) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int kc = kp + rec_of[q];
                if (kc >= ka && kc < kb) {
                    const int rowc = s_list[c][min(max(kc, 0), TR - 1)];
                    const unsigned byte = reinterpret_cast<const uint8_t *>(&s_white[rowc][0])[3 * c + ch_of[q]];
                    if (ts.f32) *reinterpret_cast<float *>(col_b + ((B0 + (unsigned)kp) * 12u + (unsigned)(32 * q + r) * 4u)) = (float)unit_of_byte(byte);
                    else *reinterpret_cast<double *>(col_b + ((B0 + (unsigned)kp) * 24u + (unsigned)(32 * q + r) * 8u)) = unit_of_byte(byte);          // :64, :69
                }
            }
        }
    };
#pragma unroll
    for (int j = 0; j < CPH; ++j) emit(hw + 16 * j, (win[j] & 0xff) - 16, (win[j] >> 8) & 0xff, win[j] >> 16, rowk[j], prj[j]);
    // (d) the windows longer than 32 records (a column segment that owns three groups, or two behind a partial head): further passes
#pragma unroll 1
    for (int j = 0; j < CPH; ++j) {
        const int c = hw + 16 * j, ka = (win[j] >> 8) & 0xff, kb = win[j] >> 16;
#pragma unroll 1
        for (int kp = (win[j] & 0xff) + 16; __any(kp < kb); kp += 32) {
            const int k = kp + r;
            const bool act = k >= ka && k < kb;
            const int row = s_list[c][min(max(k, 0), TR - 1)];
            const unsigned hv = s_hv[row][c];
            const float pr = ts.proj_th[act ? proj_lut_index((int)(short)(hv & 0xffffu), (int)(short)(hv >> 16), ts.ptiles_x, ts.wide) : 0u];
            emit(c, kp, ka, kb, act ? row : -1, pr);
        }
    }
}

// ---- row-major stream compaction framework ----
// Tile = 1024 consecutive elements per 256-thread workgroup (4 per lane, lane-interleaved so loads coalesce).
constexpr int kTile = 1024;

struct PredCorr {  // decodable pixel, row-major order
    const int64_t *h, *v;
    __device__ bool operator()(size_t i) const { return decodable(h[i], v[i]); }
};
struct PredBox {  // filter_3d_pts, triangulate.py:119 (strict, NaN drops)
    const double *xyz;
    size_t M;
    double thr;
    __device__ bool operator()(size_t i) const
    {
        const double X = xyz[i], Y = xyz[M + i], Z = xyz[2 * M + i];
        return (Z < thr) & (Z > -thr) & (Y < thr) & (Y > -thr) & (X < thr) & (X > -thr);
    }
};
struct PredFinite {  // dense XYZ [n][3] float32: NaN marks undecodable pixels
    const float *xyz;
    __device__ bool operator()(size_t i) const { return xyz[3 * i] == xyz[3 * i]; }
};

template <class Pred>
__global__ void __launch_bounds__(256) k_tile_count(Pred pred, size_t n, unsigned *__restrict__ tile_counts)
{
    const size_t base = (size_t)blockIdx.x * kTile;
    unsigned c = 0;
#pragma unroll
    for (int s = 0; s < kTile / 256; ++s) {
        const size_t i = base + s * 256 + threadIdx.x;
        c += (i < n && pred(i)) ? 1u : 0u;
    }
    // wave reduce, then one LDS slot per wave
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    __shared__ unsigned w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// exclusive scan of tile counts by one workgroup (ntiles <= a few 10^4): every thread owns a contiguous slab (the scan stays ordered),
// the slab sums are scanned with wave shuffles + one LDS hop -- no serial loop over the 1024 partial sums
__global__ void __launch_bounds__(1024) k_tile_scan(const unsigned *__restrict__ tile_counts, size_t ntiles,
                                                    unsigned long long *__restrict__ tile_off, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t per = (ntiles + 1023) / 1024;
    const size_t i0 = min(ntiles, (size_t)t * per), i1 = min(ntiles, i0 + per);
    unsigned long long mine = 0;
    for (size_t i = i0; i < i1; ++i) mine += tile_counts[i];
    unsigned long long inc = mine;                     // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned long long n = wsum[w];
        before += w < wave ? n : 0ull;
        all += n;
    }
    if (t == 0) *total = all;
    unsigned long long acc = before + inc - mine;
    for (size_t i = i0; i < i1; ++i) {
        tile_off[i] = acc;
        acc += tile_counts[i];
    }
}

// Scatter: within a tile the order is element order.  Sub-step s covers elements base+s*256 .. +255, so ranks are
// (sub-steps before) + (waves before in this sub-step) + (lanes before in this wave: ballot + mbcnt).
template <class Pred, class Emit>
__global__ void __launch_bounds__(256) k_tile_scatter(Pred pred, Emit emit, size_t n, const unsigned long long *__restrict__ tile_off)
{
    __shared__ unsigned wcount[kTile / 256][4];
    const size_t base = (size_t)blockIdx.x * kTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool keep[kTile / 256];
    unsigned rank_in_wave[kTile / 256];
#pragma unroll
    for (int s = 0; s < kTile / 256; ++s) {
        const size_t i = base + s * 256 + threadIdx.x;
        keep[s] = (i < n) && pred(i);
        const unsigned long long m = __ballot(keep[s]);
        rank_in_wave[s] = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (lane == 0) wcount[s][wave] = (unsigned)__popcll(m);
    }
    __syncthreads();
    unsigned long long o = tile_off[blockIdx.x];
#pragma unroll
    for (int s = 0; s < kTile / 256; ++s) {
        unsigned before = 0;
        for (int w = 0; w < wave; ++w) before += wcount[s][w];
        if (keep[s]) emit(base + s * 256 + threadIdx.x, o + before + rank_in_wave[s]);
        o += wcount[s][0] + wcount[s][1] + wcount[s][2] + wcount[s][3];
    }
}

struct EmitCorr {
    const int64_t *h, *v;
    const uint8_t *white;
    int W, proj_w, proj_h;
    float *cam, *proj;
    double *colors;
    __device__ void operator()(size_t p, unsigned long long o) const
    {
        const int64_t hv = h[p], vv = v[p];
        cam[2 * o] = (float)(p % W);
        cam[2 * o + 1] = (float)(p / W);
        proj[2 * o] = (float)(hv < proj_w - 1 ? hv : proj_w - 1);
        proj[2 * o + 1] = (float)(vv < proj_h - 1 ? vv : proj_h - 1);
        if (colors) {
            colors[3 * o] = (double)white[3 * p] / 255.0;
            colors[3 * o + 1] = (double)white[3 * p + 1] / 255.0;
            colors[3 * o + 2] = (double)white[3 * p + 2] / 255.0;
        }
    }
};
struct EmitBox {
    const double *xyz, *colors;
    size_t M;
    const unsigned long long *kept;  // device total (written by the scan): row length of the output (3, kept)
    double *xyz_out, *colors_out;
    __device__ void operator()(size_t i, unsigned long long o) const
    {
        const unsigned long long K = *kept;
        xyz_out[o] = xyz[i];
        xyz_out[K + o] = xyz[M + i];
        xyz_out[2 * K + o] = xyz[2 * M + i];
        if (colors_out) {
            colors_out[3 * o] = colors[3 * i];
            colors_out[3 * o + 1] = colors[3 * i + 1];
            colors_out[3 * o + 2] = colors[3 * i + 2];
        }
    }
};
struct EmitPoints {
    const float *xyz;
    uint32_t key0;
    float *points;
    uint32_t *keys;
    __device__ void operator()(size_t i, unsigned long long o) const
    {
        points[3 * o] = xyz[3 * i];
        points[3 * o + 1] = xyz[3 * i + 1];
        points[3 * o + 2] = xyz[3 * i + 2];
        if (keys) keys[o] = key0 + (uint32_t)i;
    }
};

struct EmitRecords {   // 16-byte exchange records: float32 x, y, z + uint32 linear pixel key (one all-gatherv instead of two)
    const float *xyz;
    uint32_t key0;
    float4 *rec;
    __device__ void operator()(size_t i, unsigned long long o) const
    {
        rec[o] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], __uint_as_float(key0 + (uint32_t)i));
    }
};

template <class Pred, class Emit>
int compact(slgc_ctx *ctx, Pred pred, Emit emit, size_t n, unsigned long long *d_total)
{
    const size_t ntiles = (n + kTile - 1) / kTile;
    void *tc, *to;
    int rc = slgc_ws(ctx, 4, (ntiles + 1) * sizeof(unsigned), &tc);
    if (rc) return rc;
    rc = slgc_ws(ctx, 5, (ntiles + 1) * sizeof(unsigned long long), &to);
    if (rc) return rc;
    if (n) hipLaunchKernelGGL((k_tile_count<Pred>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, pred, n, (unsigned *)tc);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned *)tc, ntiles, (unsigned long long *)to, d_total);
    if (n)
        hipLaunchKernelGGL((k_tile_scatter<Pred, Emit>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, pred, emit, n,
                           (const unsigned long long *)to);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

// The four passes of the x-major list build on ctx->stream (nothing synchronises with the host).
template <typename MapT, int SRC>
int xmajor_lists(slgc_ctx *ctx, const MapT *d_h, const MapT *d_v, int cam_w, int cam_h, int proj_w, int proj_h, const uint8_t *d_white, float *d_cam,
                 float *d_proj, double *d_colors, const float *d_xyz, double *d_pts, unsigned long long *d_total, const TriScatter &ts = TriScatter{})
{
    constexpr int TC = SLGC_SCATTER_TC, TR = kTilePixels / TC;
    const size_t npix = (size_t)cam_w * cam_h;
    const int nchunks = (cam_h + kChunkRows - 1) / kChunkRows;
    void *counts, *colstart;
    int rc = slgc_ws(ctx, 4, ((size_t)nchunks * cam_w + 1) * sizeof(unsigned), &counts);
    if (rc) return rc;
    rc = slgc_ws(ctx, 5, ((size_t)cam_w + 1) * sizeof(unsigned long long), &colstart);
    if (rc) return rc;
    const int groups_x = (cam_w + 63) / 64;
    if (npix) hipLaunchKernelGGL(k_xmajor_count<MapT>, dim3(groups_x, nchunks), dim3(256), 0, ctx->stream, d_h, d_v, cam_w, cam_h, (unsigned *)counts);
    if (cam_w)
        hipLaunchKernelGGL(k_xmajor_colprefix, dim3(groups_x), dim3(64 * kPrefixGroups), 0, ctx->stream, (unsigned *)counts, cam_w, npix ? nchunks : 0,
                           (unsigned long long *)colstart);
    hipLaunchKernelGGL(k_xmajor_colscan, dim3(1), dim3(1024), 0, ctx->stream, cam_w, (unsigned long long *)colstart, d_total);
    if constexpr (SRC == 2 && sizeof(MapT) == 2 && TC == 64) {
        // slgc_cloud_dev's scatter in whole 128-byte lines (k_xmajor_lines) when the shape allows its dword loads; "lists_lines" 0 = the tile-run kernel (A/B)
        // ("lists_lines" 1 = where it pays: the tile-run kernel is the faster one while the image is only one or two rounds of resident workgroups --
        //  1280x720: 35.5 vs 41.0 us per scan, 1920x1080: 58.5 vs 61.8, 4096x3000: 383 vs 306; 2 = wherever the shape allows)
        const bool big = (size_t)((cam_w + 63) / 64) * (size_t)((cam_h + kLinesNominal - 1) / kLinesNominal) >= 2048;
        if (npix >= 4 && npix <= ((size_t)1 << 27) && (ctx->tune_lists_lines == 2 || (ctx->tune_lists_lines == 1 && big)) && cam_w % 4 == 0 && (!d_colors || (uintptr_t)d_white % 4 == 0) && ((uintptr_t)d_h | (uintptr_t)d_v) % 4 == 0) {
            const int tiles_x = (cam_w + 63) / 64, tiles_y = (cam_h + kLinesNominal - 1) / kLinesNominal;
            if (ts.cn.nodes)
                hipLaunchKernelGGL((k_xmajor_lines<true>), dim3((unsigned)tiles_x * (unsigned)tiles_y), dim3(kScatterThreads), 0, ctx->stream, d_h, d_v, cam_w, cam_h,
                                   proj_w, proj_h, d_white, (const unsigned *)counts, (const unsigned long long *)colstart, d_cam, d_proj, d_colors, d_pts,
                                   (const unsigned long long *)d_total, tiles_x, ts, tiles_y, ctx->tune_lists_order, lists_abl());
            else
                hipLaunchKernelGGL((k_xmajor_lines<false>), dim3((unsigned)tiles_x * (unsigned)tiles_y), dim3(kScatterThreads), 0, ctx->stream, d_h, d_v, cam_w, cam_h,
                                   proj_w, proj_h, d_white, (const unsigned *)counts, (const unsigned long long *)colstart, d_cam, d_proj, d_colors, d_pts,
                                   (const unsigned long long *)d_total, tiles_x, ts, tiles_y, ctx->tune_lists_order, lists_abl());
            HIP_TRY(ctx, hipGetLastError());
            ctx->last_list_kernel = SLGC_LISTS_WHOLE_LINES;
            return SLGC_OK;
        }
    }
    ctx->last_list_kernel = SLGC_LISTS_TILE_RUNS;
    if (npix) {
        const int tiles_x = (cam_w + TC - 1) / TC, tiles_y = (cam_h + TR - 1) / TR;
        hipLaunchKernelGGL((k_xmajor_scatter<MapT, SRC, TC>), dim3((unsigned)tiles_x * (unsigned)tiles_y), dim3(kScatterThreads), 0, ctx->stream, d_h, d_v, cam_w,
                           cam_h, proj_w, proj_h, d_white, (const unsigned *)counts, (const unsigned long long *)colstart, d_cam, d_proj, d_colors, d_xyz, d_pts,
                           (const unsigned long long *)d_total, tiles_x, lists_abl(), ts, tiles_y, ctx->tune_lists_order);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

}  // namespace

int launch_correspond(slgc_ctx *ctx, const int64_t *d_h, const int64_t *d_v, int cam_w, int cam_h, int proj_w, int proj_h,
                      const uint8_t *d_white, int order, float *d_cam, float *d_proj, double *d_colors,
                      unsigned long long *d_total)
{
    const size_t npix = (size_t)cam_w * cam_h;
    if (order == SLGC_ORDER_ROW) {
        PredCorr pred{d_h, d_v};
        EmitCorr emit{d_h, d_v, d_white, cam_w, proj_w, proj_h, d_cam, d_proj, d_white ? d_colors : nullptr};
        return compact(ctx, pred, emit, npix, d_total);
    }
    return xmajor_lists<int64_t, 0>(ctx, d_h, d_v, cam_w, cam_h, proj_w, proj_h, d_white, d_cam, d_proj, d_white ? d_colors : nullptr, nullptr, nullptr,
                                    d_total);
}

// Device-resident form of the reference-shaped product (slgc_cloud_lists_dev): int16 maps + dense float32 XYZ (+ white image) ->
// x-major cam_pts / proj_pts float32 [M][2], Pts float64 (3,M), colors float64 [M][3]; nothing synchronises with the host.
int launch_cloud_lists(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, const float *d_xyz, const uint8_t *d_white, int cam_w, int cam_h,
                       int proj_w, int proj_h, float *d_cam, float *d_proj, double *d_pts, double *d_colors, unsigned long long *d_total)
{
    if (d_xyz && d_pts)
        return xmajor_lists<int16_t, 1>(ctx, d_h, d_v, cam_w, cam_h, proj_w, proj_h, d_white, d_cam, d_proj, d_white ? d_colors : nullptr, d_xyz, d_pts,
                                        d_total);
    return xmajor_lists<int16_t, 0>(ctx, d_h, d_v, cam_w, cam_h, proj_w, proj_h, d_white, d_cam, d_proj, d_white ? d_colors : nullptr, nullptr, nullptr,
                                    d_total);
}

// slgc_cloud_dev: the same lists straight from the int16 maps -- every valid pixel triangulated inside the scatter (no dense XYZ anywhere).
// The ray tables of the whole image must be in place (ensure_luts(cam_h, cam_w, 0, ...), done by the caller).
int launch_cloud_tri(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, const uint8_t *d_white, int cam_w, int cam_h, int proj_w, int proj_h,
                     float *d_cam, float *d_proj, double *d_pts, double *d_colors, unsigned long long *d_total, int f32)
{
    if (!d_pts)                      // lists only: nothing to triangulate
        return xmajor_lists<int16_t, 0>(ctx, d_h, d_v, cam_w, cam_h, proj_w, proj_h, d_white, d_cam, d_proj, d_white ? d_colors : nullptr, nullptr, nullptr, d_total);
    TriScatter ts{};
    ts.cam_lut = (const float2 *)ctx->lut_cam;
    ts.cn = SLGC_CAM_NODES_FOR(ctx, cam_w, (size_t)cam_w * cam_h / 4 < (1u << 24));
    ts.proj_lut = (const float2 *)ctx->lut_proj;
    ts.proj_th = (const float *)ctx->lut_proj_th;
    ts.ptiles_x = proj_tiles_x(ctx, proj_w);
    ts.wide = ctx->tune_proj_tile;
    ts.f32 = f32;
    ts.kf = make_tri_f32(ctx->calib.T, ctx->calib.t_len);
    memcpy(ts.T, ctx->calib.T, sizeof ts.T);
    ts.t_len = ctx->calib.t_len;
    ctx->last_nodes = ts.cn.nodes ? 1 : 0;
    ctx->last_guard = 1;
    return xmajor_lists<int16_t, 2>(ctx, d_h, d_v, cam_w, cam_h, proj_w, proj_h, d_white, d_cam, d_proj, d_white ? d_colors : nullptr, nullptr, d_pts, d_total, ts);
}

// pass 0: count only (total -> *d_total); pass 1: count again + scatter (outputs sized from pass 0's total).
int launch_filter(slgc_ctx *ctx, const double *d_xyz, const double *d_colors, int64_t M, double thr, double *d_xyz_out,
                  double *d_colors_out, unsigned long long *d_total, int pass)
{
    PredBox pred{d_xyz, (size_t)M, thr};
    if (pass == 0) {
        const size_t ntiles = ((size_t)M + kTile - 1) / kTile;
        void *tc, *to;
        int rc = slgc_ws(ctx, 4, (ntiles + 1) * sizeof(unsigned), &tc);
        if (rc) return rc;
        rc = slgc_ws(ctx, 5, (ntiles + 1) * sizeof(unsigned long long), &to);
        if (rc) return rc;
        if (M) hipLaunchKernelGGL((k_tile_count<PredBox>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, pred, (size_t)M, (unsigned *)tc);
        hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned *)tc, ntiles, (unsigned long long *)to, d_total);
        HIP_TRY(ctx, hipGetLastError());
        return SLGC_OK;
    }
    EmitBox emit{d_xyz, d_colors, (size_t)M, d_total, d_xyz_out, d_colors ? d_colors_out : nullptr};
    return compact(ctx, pred, emit, (size_t)M, d_total);
}

int launch_compact_dense(slgc_ctx *ctx, const float *d_xyz, int rows, int W, int row0, float *d_points, uint32_t *d_keys,
                         unsigned long long *d_count)
{
    PredFinite pred{d_xyz};
    EmitPoints emit{d_xyz, (uint32_t)((size_t)row0 * W), d_points, d_keys};
    return compact(ctx, pred, emit, (size_t)rows * W, d_count);
}

int launch_compact_records(slgc_ctx *ctx, const float *d_xyz, int rows, int W, int row0, void *d_records, unsigned long long *d_count)
{
    PredFinite pred{d_xyz};
    EmitRecords emit{d_xyz, (uint32_t)((size_t)row0 * W), (float4 *)d_records};
    return compact(ctx, pred, emit, (size_t)rows * W, d_count);
}
