// cloud.hip -- point-cloud post-processing after the path:
// the arithmetic of Open3D's remove_statistical_outlier(nb_neighbors, std_ratio) as called at
// scanner/utils/visualize.py:104 -- for every point the mean distance to its nb_neighbors nearest points (the point itself
// included, as Open3D's KD-tree query returns it).  Open3D is third-party and not installed in the build container:
// parity with Open3D is UNPINNED; the kernel is exact k-NN and is checked against scipy's cKDTree.
//
// K7  uniform-grid exact k-NN: points are binned into cubic cells of edge s (counting sort: atomics + the exclusive scan below), every
//     query scans the 3x3x3 block around its cell keeping the K smallest squared distances in registers (static insertion
//     network, no dynamic register indexing).  A result is exact when its k-th distance <= s (nothing outside the block can
//     be closer); the others are retried with a larger cell edge until all are exact.  The first round takes its queries in cell
//     order (see k_knn_mean).  The grid covers a ROBUST box (0.5 % .. 99.5 % quantiles of a sample per axis; points outside are
//     clamped into its boundary cells, which keeps the "within s => inside the 3x3x3 block" argument intact), and the first cell edge
//     is tuned on the cloud itself: a few counting passes steer the mean population of the non-empty cells to ~9 points (a surface
//     sampled so that the disc holding k = 20 points has radius ~0.85 s).  A bounding-box estimate is off by an order of magnitude
//     as soon as a scan has a few far outliers -- which is what this routine exists to remove.
#include "slgc_internal.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

// ---- exclusive prefix sum of uint32 (cell counts -> cell starts) -------------------------------------------------------------
// Three-level reduce-then-scan: a 256-thread workgroup owns a tile of 2048 consecutive elements (8 per lane, two 16-byte loads);
// pass 1 writes every tile's sum, the sums are scanned recursively (537 M cells -> 262 145 tile sums -> 129 -> 1), pass 2 rescans
// each tile from its prefix.  Inside a tile: per-lane serial prefix, wave64 inclusive scan with __shfl_up, wave totals through LDS.
constexpr int kScanTile = 2048;

__device__ __forceinline__ unsigned block_exclusive_scan_256(unsigned v, unsigned &block_total)
{
    __shared__ unsigned wtot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    unsigned before = 0;
    for (int w = 0; w < wave; ++w) before += wtot[w];
    block_total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();                                   // wtot is reused by the caller's next call
    return before + inc - v;
}

__global__ void __launch_bounds__(256) k_scan_tile_sums(const unsigned *__restrict__ in, size_t n, unsigned *__restrict__ sums)
{
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * 8;
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (base + j < n) ? in[base + j] : 0u;
    unsigned total;
    (void)block_exclusive_scan_256(s, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// prefix == nullptr: single tile (top level).  total (optional) receives the grand total from the last tile.
__global__ void __launch_bounds__(256) k_scan_tiles(const unsigned *__restrict__ in, size_t n, const unsigned *__restrict__ prefix,
                                                    unsigned *__restrict__ out)
{
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * 8;
    unsigned v[8], s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = (base + j < n) ? in[base + j] : 0u;
        s += v[j];
    }
    unsigned total;
    unsigned run = block_exclusive_scan_256(s, total) + (prefix ? prefix[blockIdx.x] : 0u);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
}

// out[i] = in[0] + ... + in[i-1], i < n.  scratch: >= scan_scratch_elems(n) uint32.  in and out must not overlap.
size_t scan_scratch_elems(size_t n)
{
    size_t t = 0;
    while (n > (size_t)kScanTile) {
        n = (n + kScanTile - 1) / kScanTile;
        t += 2 * n;                                     // tile sums + their scan
    }
    return t + 16;
}

int exclusive_scan_u32(slgc_ctx *ctx, const unsigned *d_in, unsigned *d_out, size_t n, unsigned *d_scratch)
{
    if (n == 0) return SLGC_OK;
    const size_t tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles == 1) {
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(256), 0, ctx->stream, d_in, n, (const unsigned *)nullptr, d_out);
    } else {
        unsigned *sums = d_scratch, *sums_scanned = d_scratch + tiles;
        hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, d_in, n, sums);
        int rc = exclusive_scan_u32(ctx, sums, sums_scanned, tiles, d_scratch + 2 * tiles);
        if (rc) return rc;
        hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, d_in, n, (const unsigned *)sums_scanned, d_out);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

struct Grid {
    float ox, oy, oz, inv_s;
    int nx, ny, nz;
};

__device__ __forceinline__ int cell_coord(float v, float o, float inv_s, int n)
{
    const int c = (int)((v - o) * inv_s);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__global__ void __launch_bounds__(256) k_cell_count(const float *__restrict__ pts, size_t M, Grid g, unsigned *__restrict__ counts,
                                                    unsigned *__restrict__ cell_of)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int cx = cell_coord(pts[3 * i], g.ox, g.inv_s, g.nx), cy = cell_coord(pts[3 * i + 1], g.oy, g.inv_s, g.ny),
              cz = cell_coord(pts[3 * i + 2], g.oz, g.inv_s, g.nz);
    const unsigned c = ((unsigned)cz * g.ny + cy) * g.nx + cx;
    cell_of[i] = c;
    atomicAdd(counts + c, 1u);
}

__global__ void __launch_bounds__(256) k_cell_fill(const float *__restrict__ pts, size_t M, const unsigned *__restrict__ cell_of,
                                                   const unsigned *__restrict__ cell_start, unsigned *__restrict__ cursor,
                                                   float4 *__restrict__ sorted)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const unsigned c = cell_of[i];
    const unsigned slot = cell_start[c] + atomicAdd(cursor + c, 1u);
    sorted[slot] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __uint_as_float((unsigned)i));   // .w = where the point came from
}

// non-empty cells: one partial count per workgroup, folded on the host (16 K waves adding to ONE word took 0.38 ms of same-address atomics)
__global__ void __launch_bounds__(256) k_count_nonzero(const unsigned *__restrict__ v, size_t n, unsigned *__restrict__ partial)
{
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c += v[i] ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    __shared__ unsigned w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// bounding box + finiteness of the cloud, on the device the points are uploaded to anyway (the host loop over 15 M floats took longer than the
// first k-NN pass): every workgroup writes {lo[3], hi[3], non-finite count} of its strided share; the host folds the <= 1024 partial rows
__global__ void __launch_bounds__(256) k_bbox(const float *__restrict__ pts, size_t M, float *__restrict__ partial)
{
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (size_t)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = pts[3 * i + c];
            bad += (!(v == v) || v > 3e38f || v < -3e38f) ? 1u : 0u;
            lo[c] = fminf(lo[c], v);
            hi[c] = fmaxf(hi[c], v);
        }
    }
    __shared__ float s_lo[4][3], s_hi[4][3];
    __shared__ unsigned s_bad[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], __shfl_down(lo[c], o, 64));
            hi[c] = fmaxf(hi[c], __shfl_down(hi[c], o, 64));
        }
        bad += __shfl_down(bad, o, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int c = 0; c < 3; ++c) { s_lo[wave][c] = lo[c]; s_hi[wave][c] = hi[c]; }
        s_bad[wave] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *out = partial + 8 * (size_t)blockIdx.x;
        for (int c = 0; c < 3; ++c) {
            out[c] = fminf(fminf(s_lo[0][c], s_lo[1][c]), fminf(s_lo[2][c], s_lo[3][c]));
            out[3 + c] = fmaxf(fmaxf(s_hi[0][c], s_hi[1][c]), fmaxf(s_hi[2][c], s_hi[3][c]));
        }
        out[6] = __uint_as_float(s_bad[0] + s_bad[1] + s_bad[2] + s_bad[3]);
        out[7] = 0.f;
    }
}

// queries: indices of the points still to be resolved (nullptr = all points 0..nq-1)
template <int K>
__global__ void __launch_bounds__(128) k_knn_mean(const float *__restrict__ pts, const unsigned *__restrict__ queries, size_t nq, Grid g,
                                                  float s, const unsigned *__restrict__ cell_start, const unsigned *__restrict__ cell_end,
                                                  const float4 *__restrict__ sorted, int k, double *__restrict__ mean_dist,
                                                  unsigned *__restrict__ unresolved, unsigned *__restrict__ n_unresolved)
{
    const size_t t = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (t >= nq) return;
    // First round (queries == nullptr): the queries are the points in CELL order -- the lanes of a wave sit in the same or neighbouring
    // cells, walk the same candidate runs (coalesced / broadcast loads, loops of equal length) and insert at similar times.
    unsigned i;
    double qx, qy, qz;
    if (queries) {
        i = queries[t];
        qx = pts[3 * (size_t)i], qy = pts[3 * (size_t)i + 1], qz = pts[3 * (size_t)i + 2];
    } else {
        const float4 q = sorted[t];
        i = __float_as_uint(q.w);
        qx = q.x, qy = q.y, qz = q.z;
    }
    const int cx = cell_coord((float)qx, g.ox, g.inv_s, g.nx), cy = cell_coord((float)qy, g.oy, g.inv_s, g.ny),
              cz = cell_coord((float)qz, g.oz, g.inv_s, g.nz);
    double best[K];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = 1e300;
    // the nine x-runs of the block, nearest first (the query's own row of cells, then the four that share a face with it, then the corners):
    // the k-th best distance tightens early and fewer of the later candidates enter the insertion network
    for (int run = 0; run < 9; ++run) {
        const int dz = (0x28215 >> (2 * run) & 3) - 1, dy = (0x22161 >> (2 * run) & 3) - 1;      // (dz, dy) = (0,0) (0,-1) (0,1) (-1,0) (1,0) (-1,-1) (-1,1) (1,-1) (1,1)
        const int z = cz + dz;
        if (z < 0 || z >= g.nz) continue;
        {
            const int y = cy + dy;
            if (y < 0 || y >= g.ny) continue;
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
            const unsigned row = ((unsigned)z * g.ny + y) * g.nx;
            const unsigned b = cell_start[row + x0], e = cell_end[row + x1];     // the x-run of cells is contiguous in the sorted array
            for (unsigned p = b; p < e; ++p) {
                const float4 c = sorted[p];
                const double ddx = (double)c.x - qx, ddy = (double)c.y - qy, ddz = (double)c.z - qz;
                double d = ddx * ddx + ddy * ddy + ddz * ddz;
                if (d < best[K - 1]) {
#pragma unroll
                    for (int j = 0; j < K; ++j) {          // static insertion network: best[] stays sorted ascending
                        double lo, hi;                                         // one v_min_f64 + one v_max_f64 per step (fmin / fmax would add a quieting
                        asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(d), "v"(best[j]));       // v_max_f64 x, x per step; nothing here is NaN: the guard above)
                        asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(d), "v"(best[j]));
                        best[j] = lo;
                        d = hi;
                    }
                }
            }
        }
    }
    // exact iff the k-th neighbour lies within s of the query (the 3x3x3 block contains the ball of radius s)
    double kth = 0.0, sum = 0.0;
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < k) {
            sum += sqrt(best[j]);
            kth = best[j];
        }
    if (kth <= (double)s * (double)s) {
        mean_dist[i] = sum / (double)k;
    } else {
        unresolved[atomicAdd(n_unresolved, 1u)] = i;
    }
}

// The stragglers of the first round (k-th neighbour farther than one cell edge: sparse regions, the outliers this routine exists to find) are
// retried on the SAME grid with a wider block, (2R + 1)^3 cells, exact when the k-th distance <= R s.  They are few, scattered and their
// candidate lists long and uneven, so here G = 16 lanes share one query: each lane scans every 16th candidate of every x-run into its own
// sorted top-K, then the group merges the 16 lists by popping the smallest head k times (a 16-lane min per pop; the lane that held it shifts
// its list).  Four queries ride in a wave and share the merge's instruction stream.
template <int K, int G>
__global__ void __launch_bounds__(256) k_knn_mean_group(const float *__restrict__ pts, const unsigned *__restrict__ queries, size_t nq, Grid g, float reach, int R,
                                                        const unsigned *__restrict__ cell_start, const unsigned *__restrict__ cell_end,
                                                        const float4 *__restrict__ sorted, int k, double *__restrict__ mean_dist,
                                                        unsigned *__restrict__ unresolved, unsigned *__restrict__ n_unresolved)
{
    static_assert(G == 16, "the group is one DPP row of 16 lanes");
    const size_t t = ((size_t)blockIdx.x * 256 + threadIdx.x) / G;
    const int sub = (int)(threadIdx.x % G), lane = (int)(threadIdx.x & 63), group_base = lane & ~(G - 1);
    const bool live = t < nq;
    const unsigned i = live ? queries[t] : 0u;
    const double qx = pts[3 * (size_t)i], qy = pts[3 * (size_t)i + 1], qz = pts[3 * (size_t)i + 2];
    const int cx = cell_coord((float)qx, g.ox, g.inv_s, g.nx), cy = cell_coord((float)qy, g.oy, g.inv_s, g.ny),
              cz = cell_coord((float)qz, g.oz, g.inv_s, g.nz);
    double best[K];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = 1e300;
    if (live) {
        const int x0 = max(cx - R, 0), x1 = min(cx + R, g.nx - 1);
        for (int z = max(cz - R, 0); z <= min(cz + R, g.nz - 1); ++z)
            for (int y = max(cy - R, 0); y <= min(cy + R, g.ny - 1); ++y) {
                const unsigned row = ((unsigned)z * g.ny + y) * g.nx;
                const unsigned b = cell_start[row + x0], e = cell_end[row + x1];
                for (unsigned p = b + (unsigned)sub; p < e; p += G) {
                    const float4 c = sorted[p];
                    const double ddx = (double)c.x - qx, ddy = (double)c.y - qy, ddz = (double)c.z - qz;
                    double d = ddx * ddx + ddy * ddy + ddz * ddz;
                    if (d < best[K - 1]) {
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            double lo, hi;
                            asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(d), "v"(best[j]));
                            asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(d), "v"(best[j]));
                            best[j] = lo;
                            d = hi;
                        }
                    }
                }
            }
    }
    double sum = 0.0, kth = 0.0;
    for (int j = 0; j < k; ++j) {                       // k <= K pops
        double m = best[0];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
            const double other = __shfl_xor(m, o, G);
            m = other < m ? other : m;
        }
        const unsigned long long holders = __ballot(best[0] == m);
        const unsigned mine = (unsigned)(holders >> group_base) & ((1u << G) - 1u);
        if (sub == __builtin_ctz(mine | (1u << G))) {  // the first lane of the group that holds it gives it up (equal distances of distinct points pop one by one)
#pragma unroll
            for (int q = 0; q + 1 < K; ++q) best[q] = best[q + 1];
            best[K - 1] = 1e300;
        }
        sum += sqrt(m);
        kth = m;
    }
    if (live && sub == 0) {
        if (kth <= (double)reach * (double)reach) mean_dist[i] = sum / (double)k;
        else unresolved[atomicAdd(n_unresolved, 1u)] = i;
    }
}

template <int K>
void launch_knn_group(slgc_ctx *ctx, const float *d_pts, const unsigned *d_q, size_t nq, const Grid &g, float reach, int R, const unsigned *cs, const unsigned *ce,
                      const float4 *sorted, int k, double *d_mean, unsigned *d_unres, unsigned *d_nun)
{
    hipLaunchKernelGGL((k_knn_mean_group<K, 16>), dim3((unsigned)((nq * 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_pts, d_q, nq, g, reach, R, cs, ce, sorted, k,
                       d_mean, d_unres, d_nun);
}

template <int K>
void launch_knn(slgc_ctx *ctx, const float *d_pts, const unsigned *d_q, size_t nq, const Grid &g, float s, const unsigned *cs, const unsigned *ce,
                const float4 *sorted, int k, double *d_mean, unsigned *d_unres, unsigned *d_nun)
{
    hipLaunchKernelGGL((k_knn_mean<K>), dim3((unsigned)((nq + 127) / 128)), dim3(128), 0, ctx->stream, d_pts, d_q, nq, g, s, cs, ce, sorted, k, d_mean,
                       d_unres, d_nun);
}

}  // namespace

namespace {

// every stride-th point, packed (the robust box's sample when the cloud only exists in HBM)
__global__ void __launch_bounds__(256) k_sample_points(const float *__restrict__ pts, size_t M, size_t stride, size_t n, float *__restrict__ out)
{
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const size_t i = j * stride;
    out[3 * j] = pts[3 * i];
    out[3 * j + 1] = pts[3 * i + 1];
    out[3 * j + 2] = pts[3 * i + 2];
}

// d_pts: float32 [M][3] in HBM (already there, or uploaded by the caller on ctx->stream); pts: the same points on the host, or nullptr;
// d_mean: float64 [M] in HBM.  Synchronises the stream (the grid is sized from statistics of the cloud).
int knn_mean_core(slgc_ctx *ctx, const void *d_pts, const float *pts, int64_t M, int k, void *d_mean)
{
    int rc;
    float lo[3], hi[3];
    {
        void *d_part;
        const unsigned nb = (unsigned)std::min<int64_t>((M + 255) / 256, 1024);
        if ((rc = slgc_ws(ctx, 7, (size_t)nb * 32 + 64, &d_part))) return rc;
        hipLaunchKernelGGL(k_bbox, dim3(nb), dim3(256), 0, ctx->stream, (const float *)d_pts, (size_t)M, (float *)d_part);
        std::vector<float> part((size_t)nb * 8);
        HIP_TRY(ctx, hipMemcpyAsync(part.data(), d_part, (size_t)nb * 32, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        unsigned long long bad = 0;
        for (int c = 0; c < 3; ++c) { lo[c] = part[c]; hi[c] = part[3 + c]; }
        for (unsigned b = 0; b < nb; ++b) {
            for (int c = 0; c < 3; ++c) {
                lo[c] = std::min(lo[c], part[8 * (size_t)b + c]);
                hi[c] = std::max(hi[c], part[8 * (size_t)b + 3 + c]);
            }
            unsigned u;
            memcpy(&u, &part[8 * (size_t)b + 6], 4);
            bad += u;
        }
        if (bad && !pts) return slgc_fail(ctx, SLGC_EINVAL, "%llu non-finite coordinates in the cloud", bad);
        if (bad) {
            for (int64_t i = 0; i < M; ++i)             // name the first one (error path only)
                for (int c = 0; c < 3; ++c) {
                    const float v = pts[3 * i + c];
                    if (!(v == v) || v > 3e38f || v < -3e38f) return slgc_fail(ctx, SLGC_EINVAL, "non-finite coordinate at point %lld", (long long)i);
                }
            return slgc_fail(ctx, SLGC_EINVAL, "non-finite coordinates");
        }
    }
    // robust box: per-axis 0.5 % / 99.5 % quantiles of a sample of <= 65 536 points, widened by 5 %, inside the true box
    {
        const int64_t stride = M > 65536 ? M / 65536 : 1;
        const size_t ns = (size_t)((M + stride - 1) / stride);
        std::vector<float> sample;                      // the same points either way: every stride-th one
        if (!pts) {
            void *d_s;
            if ((rc = slgc_ws(ctx, 7, ns * 12 + 64, &d_s))) return rc;
            hipLaunchKernelGGL(k_sample_points, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream, (const float *)d_pts, (size_t)M, (size_t)stride, ns, (float *)d_s);
            sample.resize(ns * 3);
            HIP_TRY(ctx, hipMemcpyAsync(sample.data(), d_s, ns * 12, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        std::vector<float> col;
        col.reserve(ns + 1);
        for (int c = 0; c < 3; ++c) {
            col.clear();
            if (pts) for (int64_t i = 0; i < M; i += stride) col.push_back(pts[3 * i + c]);
            else for (size_t j = 0; j < ns; ++j) col.push_back(sample[3 * j + c]);
            const size_t n = col.size(), a = (size_t)(0.005 * (double)(n - 1)), b = (size_t)(0.995 * (double)(n - 1));
            std::nth_element(col.begin(), col.begin() + a, col.end());
            const float qa = col[a];
            std::nth_element(col.begin(), col.begin() + b, col.end());
            const float qb = col[b];
            const float pad = 0.05f * (qb - qa);
            lo[c] = std::max(lo[c], qa - pad);
            hi[c] = std::min(hi[c], qb + pad);
        }
    }
    const double ext[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
    double e_sorted[3] = {ext[0], ext[1], ext[2]};
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (e_sorted[b] > e_sorted[a]) { const double t = e_sorted[a]; e_sorted[a] = e_sorted[b]; e_sorted[b] = t; }
    // scanner clouds are surfaces: first guess of the cell edge from the 2-D density over the two largest extents (radius holding ~k
    // points, x1.25); tuned below on the cloud itself
    const double area = (e_sorted[0] > 0 ? e_sorted[0] : 1e-6) * (e_sorted[1] > 0 ? e_sorted[1] : 1e-6);
    double s = 1.25 * sqrt((double)k / (3.141592653589793 * ((double)M / area)));
    if (!(s > 0)) s = 1e-6;

    void *d_sorted, *d_cellof, *d_unres[2], *d_nun;
    if ((rc = slgc_ws(ctx, 2, (size_t)M * 16, &d_sorted))) return rc;
    if ((rc = slgc_ws(ctx, 3, (size_t)M * 4, &d_cellof))) return rc;
    if ((rc = slgc_ws(ctx, 8, (size_t)M * 4, &d_unres[0]))) return rc;
    if ((rc = slgc_ws(ctx, 9, (size_t)M * 4, &d_unres[1]))) return rc;
    if ((rc = slgc_ws(ctx, 10, 64, &d_nun))) return rc;

    size_t nq = (size_t)M;
    const unsigned *d_q = nullptr;
    int cur = 0;
    const double occ_target = 0.45 * (double)k;                          // points per non-empty cell (k = 20 -> 9)
    int tune_left = 3;
    for (int round = 0; round < 40 && nq; ++round) {
        Grid g;
        g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
        void *d_counts, *d_start, *d_tmp;
        size_t ncell;
        for (;;) {
            // keep the grid below 2^29 cells (2 x 2 GiB of counters, a 4 ms scan); growing s only makes the search more conservative
            for (;;) {
                const double nx = floor(ext[0] / s) + 1, ny = floor(ext[1] / s) + 1, nz = floor(ext[2] / s) + 1;
                if (nx * ny * nz <= 536870912.0) { g.nx = (int)nx; g.ny = (int)ny; g.nz = (int)nz; break; }
                s *= 1.26;
                tune_left = 0;
            }
            g.inv_s = (float)(1.0 / s);
            ncell = (size_t)g.nx * g.ny * g.nz;
#ifdef SLGC_DIAG
            fprintf(stderr, "slgc k-NN round %d: %zu queries, cell edge %.4g, grid %d x %d x %d, extents %.4g %.4g %.4g\n", round, nq, s, g.nx, g.ny, g.nz,
                    ext[0], ext[1], ext[2]);
#endif
            if ((rc = slgc_ws(ctx, 4, (ncell + 1) * 4, &d_counts))) return rc;
            if ((rc = slgc_ws(ctx, 5, (ncell + 1) * 4, &d_start))) return rc;
            HIP_TRY(ctx, hipMemsetAsync(d_counts, 0, (ncell + 1) * 4, ctx->stream));
            hipLaunchKernelGGL(k_cell_count, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream, (const float *)d_pts, (size_t)M, g,
                               (unsigned *)d_counts, (unsigned *)d_cellof);
            if (round != 0 || tune_left <= 0 || ncell == 1) break;
            // first round only: steer the population of the non-empty cells to the target (a surface: population ~ s^2)
            --tune_left;
            unsigned long long nonempty = 0;
            {
                const unsigned nb = (unsigned)std::min<size_t>((ncell + 2047) / 2048, 4096);
                void *d_nz;
                if ((rc = slgc_ws(ctx, 7, (size_t)nb * 4 + 64, &d_nz))) return rc;
                hipLaunchKernelGGL(k_count_nonzero, dim3(nb), dim3(256), 0, ctx->stream, (const unsigned *)d_counts, ncell, (unsigned *)d_nz);
                std::vector<unsigned> nz(nb);
                HIP_TRY(ctx, hipMemcpyAsync(nz.data(), d_nz, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                for (unsigned b = 0; b < nb; ++b) nonempty += nz[b];
            }
            const double occ = (double)M / (double)(nonempty ? nonempty : 1);
            if (occ > occ_target / 1.3 && occ < occ_target * 1.3) break;
            s *= std::min(8.0, std::max(0.125, sqrt(occ_target / occ)));
        }
        if ((rc = slgc_ws(ctx, 6, scan_scratch_elems(ncell + 1) * 4, &d_tmp))) return rc;
        if ((rc = exclusive_scan_u32(ctx, (const unsigned *)d_counts, (unsigned *)d_start, ncell + 1, (unsigned *)d_tmp))) return rc;
        HIP_TRY(ctx, hipMemsetAsync(d_counts, 0, (ncell + 1) * 4, ctx->stream));          // reused as the fill cursor
        hipLaunchKernelGGL(k_cell_fill, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream, (const float *)d_pts, (size_t)M,
                           (const unsigned *)d_cellof, (const unsigned *)d_start, (unsigned *)d_counts, (float4 *)d_sorted);
        HIP_TRY(ctx, hipMemsetAsync(d_nun, 0, 4, ctx->stream));
        const unsigned *cs = (const unsigned *)d_start, *ce = cs + 1;                     // cell_end[c] = cell_start[c + 1]
        unsigned *d_out = (unsigned *)d_unres[cur];
        // the radius the 3x3x3 block is guaranteed to cover, float rounding included; a single-cell grid holds every point, so the
        // scan is exhaustive there and every result is exact whatever its k-th distance (small clouds with k close to M)
        const bool one_cell = g.nx == 1 && g.ny == 1 && g.nz == 1;
        const float s_safe = one_cell ? __builtin_huge_valf() : (float)(0.999 / (double)g.inv_s);
        if (k <= 20) launch_knn<20>(ctx, (const float *)d_pts, d_q, nq, g, s_safe, cs, ce, (const float4 *)d_sorted, k, (double *)d_mean, d_out, (unsigned *)d_nun);
        else if (k <= 32) launch_knn<32>(ctx, (const float *)d_pts, d_q, nq, g, s_safe, cs, ce, (const float4 *)d_sorted, k, (double *)d_mean, d_out, (unsigned *)d_nun);
        else launch_knn<64>(ctx, (const float *)d_pts, d_q, nq, g, s_safe, cs, ce, (const float4 *)d_sorted, k, (double *)d_mean, d_out, (unsigned *)d_nun);
        HIP_TRY(ctx, hipGetLastError());
        unsigned nun = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&nun, d_nun, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        nq = nun;
        d_q = d_out;
        cur ^= 1;
        if (one_cell && nq) return slgc_fail(ctx, SLGC_EHIP, "k-NN: %zu queries unresolved on a single-cell grid (internal error)", nq);
        // the stragglers: same grid, blocks of (2R + 1)^3 cells, R = 2, 4, 8, ... (k_knn_mean_group); a block that spans the grid is exhaustive
        for (int R = 2; nq; R *= 2) {
            const bool whole = R >= g.nx && R >= g.ny && R >= g.nz;
            const float reach = whole ? __builtin_huge_valf() : (float)(0.999 * (double)R / (double)g.inv_s);
            unsigned *d_out2 = (unsigned *)d_unres[cur];
            HIP_TRY(ctx, hipMemsetAsync(d_nun, 0, 4, ctx->stream));
            if (k <= 20) launch_knn_group<20>(ctx, (const float *)d_pts, d_q, nq, g, reach, R, cs, ce, (const float4 *)d_sorted, k, (double *)d_mean, d_out2, (unsigned *)d_nun);
            else if (k <= 32) launch_knn_group<32>(ctx, (const float *)d_pts, d_q, nq, g, reach, R, cs, ce, (const float4 *)d_sorted, k, (double *)d_mean, d_out2, (unsigned *)d_nun);
            else launch_knn_group<64>(ctx, (const float *)d_pts, d_q, nq, g, reach, R, cs, ce, (const float4 *)d_sorted, k, (double *)d_mean, d_out2, (unsigned *)d_nun);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipMemcpyAsync(&nun, d_nun, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (whole && nun) return slgc_fail(ctx, SLGC_EHIP, "k-NN: %u queries unresolved by an exhaustive scan (internal error)", nun);
            nq = nun;
            d_q = d_out2;
            cur ^= 1;
        }
    }
    if (nq) return slgc_fail(ctx, SLGC_EHIP, "k-NN left %zu points unresolved", nq);
    ctx->pend_M = ctx->filt_M = ctx->pipe_M = -1;   // workspace slots were reused
    return SLGC_OK;
}

int knn_check(slgc_ctx *ctx, const void *pts, int64_t M, int k, const void *mean)
{
    if (!ctx) return SLGC_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return slgc_fail(ctx, SLGC_EHIP, "hipSetDevice failed");
    if (M < 0 || k < 1 || k > 64 || (M && (!pts || !mean))) return slgc_fail(ctx, SLGC_EINVAL, "bad arguments (1 <= k <= 64)");
    if (M > 0x7fffffffll) return slgc_fail(ctx, SLGC_EINVAL, "too many points");
    if (M && (int64_t)k > M) return slgc_fail(ctx, SLGC_EINVAL, "k = %d exceeds the number of points %lld", k, (long long)M);
    return SLGC_OK;
}

}  // namespace

// mean distance of every point to its k nearest points (itself included).  pts: host float32 [M][3]; mean: host float64 [M].
extern "C" int slgc_knn_mean_distance(slgc_ctx *ctx, const float *pts, int64_t M, int k, double *mean)
{
    int rc = knn_check(ctx, pts, M, k, mean);
    if (rc || M == 0) return rc;
    // the points go up first; bounding box and the finiteness check run on the device behind the copy
    void *d_pts, *d_mean;
    if ((rc = slgc_ws(ctx, 0, (size_t)M * 12, &d_pts))) return rc;
    if ((rc = slgc_ws(ctx, 1, (size_t)M * 8, &d_mean))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_pts, pts, (size_t)M * 12, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = knn_mean_core(ctx, d_pts, pts, M, k, d_mean))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(mean, d_mean, (size_t)M * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLGC_OK;
}

// The same on a cloud that already sits in HBM: d_pts float32 [M][3], d_mean float64 [M] (both device; not the context's workspace).
// Work enqueued on the context's stream before the call is waited for (the grid is sized from statistics of the cloud); on return the
// last kernel is enqueued, not finished.
extern "C" int slgc_knn_mean_distance_dev(slgc_ctx *ctx, const float *d_pts, int64_t M, int k, double *d_mean)
{
    const int rc = knn_check(ctx, d_pts, M, k, d_mean);
    if (rc || M == 0) return rc;
    return knn_mean_core(ctx, d_pts, nullptr, M, k, d_mean);
}
