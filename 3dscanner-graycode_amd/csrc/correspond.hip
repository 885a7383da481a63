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

namespace {

constexpr int kChunkRows = 32;

__device__ __forceinline__ bool decodable(int64_t h, int64_t v) { return !(h == -1 || v == -1); }  // :56

// ---- x-major: pass A, per (chunk, column) counts ----
template <typename MapT>
__global__ void __launch_bounds__(256) k_xmajor_count(const MapT *__restrict__ h, const MapT *__restrict__ v, int W, int H,
                                                      unsigned *__restrict__ counts)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y;
    if (x >= W) return;
    const int y0 = chunk * kChunkRows, y1 = min(H, y0 + kChunkRows);
    unsigned c = 0;
    for (int y = y0; y < y1; ++y) c += decodable(h[(size_t)y * W + x], v[(size_t)y * W + x]) ? 1u : 0u;
    counts[(size_t)chunk * W + x] = c;
}

// ---- x-major: pass B1, one thread per column: prefix over the row chunks (lanes along x: coalesced), column totals ----
__global__ void __launch_bounds__(256) k_xmajor_colprefix(unsigned *__restrict__ counts, int W, int nchunks,
                                                          unsigned long long *__restrict__ colstart)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    unsigned run = 0;
#pragma unroll 8
    for (int c = 0; c < nchunks; ++c) {
        const unsigned n = counts[(size_t)c * W + x];
        counts[(size_t)c * W + x] = run;  // valid pixels above this chunk in column x
        run += n;
    }
    colstart[x] = run;  // column total for now
}

// ---- x-major: pass B2, one workgroup: exclusive scan of the column totals ----
__global__ void __launch_bounds__(1024) k_xmajor_colscan(int W, unsigned long long *__restrict__ colstart, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const int per = (W + 1023) / 1024;                 // each thread owns a contiguous slab of columns: the scan stays ordered
    const int x0 = min(W, t * per), x1 = min(W, x0 + per);
    unsigned long long mine = 0;
    for (int x = x0; x < x1; ++x) mine += colstart[x];
    part[t] = mine;
    __syncthreads();
    if (t == 0) {
        unsigned long long acc = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned long long n = part[i];
            part[i] = acc;
            acc += n;
        }
        *total = acc;
    }
    __syncthreads();
    unsigned long long acc = part[t];
    for (int x = x0; x < x1; ++x) {
        const unsigned long long n = colstart[x];
        colstart[x] = acc;
        acc += n;
    }
}

// ---- x-major: pass C, scatter through an LDS transpose ----
// A workgroup owns a tile of kChunkRows (32) rows x 64 columns.  Phase 1 reads it with lanes along x (coalesced 512-byte rows
// of the int64 maps) and parks the clamped projector coordinates (and the packed colour) in LDS.  Phase 2 turns the tile: a
// half-wave takes one column, its 32 lanes are the 32 rows, a ballot gives every valid pixel its rank inside the column
// segment, and the segment leaves as one contiguous run of records (x-major order = column after column, rows ascending).
constexpr int kTileCols = 64;
constexpr int kInvalid = (int)0x80000000;

// XYZ = true (device-resident product, slgc_cloud_lists_dev): the dense float32 XYZ of the scan rides through the same transpose and
// leaves as the reference's float64 (3,M) array (triangulate.py:95; M = *total, written by the column scan before this kernel runs).
template <typename MapT, bool XYZ>
__global__ void __launch_bounds__(256) k_xmajor_scatter(const MapT *__restrict__ h, const MapT *__restrict__ v, int W, int H,
                                                        int proj_w, int proj_h, const uint8_t *__restrict__ white,
                                                        const unsigned *__restrict__ counts,
                                                        const unsigned long long *__restrict__ colstart, float *__restrict__ cam,
                                                        float *__restrict__ proj, double *__restrict__ colors,
                                                        const float *__restrict__ xyz, double *__restrict__ pts,
                                                        const unsigned long long *__restrict__ total)
{
    __shared__ int s_pu[kChunkRows][kTileCols + 1], s_pv[kChunkRows][kTileCols + 1];
    __shared__ unsigned s_rgb[kChunkRows][kTileCols + 1];
    __shared__ float s_xyz[XYZ ? 3 : 1][kChunkRows][kTileCols + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x_tile = blockIdx.x * kTileCols, chunk = blockIdx.y, y_tile = chunk * kChunkRows;
#pragma unroll
    for (int s = 0; s < kChunkRows / 4; ++s) {
        const int r = s * 4 + wave, x = x_tile + lane, y = y_tile + r;
        int pu = kInvalid, pv = 0;
        unsigned rgb = 0;
        if (x < W && y < H) {
            const size_t p = (size_t)y * W + x;
            const int64_t hv = h[p], vv = v[p];
            if (decodable(hv, vv)) {
                pu = (int)(hv < proj_w - 1 ? hv : proj_w - 1);                  // triangulate.py:60 (maps from this library fit 32 bits)
                pv = (int)(vv < proj_h - 1 ? vv : proj_h - 1);                  // :61
                if (pu == kInvalid) pu = kInvalid + 1;
                if (colors) rgb = (unsigned)white[3 * p] | ((unsigned)white[3 * p + 1] << 8) | ((unsigned)white[3 * p + 2] << 16);
                if constexpr (XYZ) {
                    s_xyz[0][r][lane] = xyz[3 * p];
                    s_xyz[1][r][lane] = xyz[3 * p + 1];
                    s_xyz[2][r][lane] = xyz[3 * p + 2];
                }
            }
        }
        s_pu[r][lane] = pu;
        s_pv[r][lane] = pv;
        s_rgb[r][lane] = rgb;
    }
    __syncthreads();
    const int half = lane >> 5, r = lane & 31;
#pragma unroll 2
    for (int s = 0; s < kTileCols / 8; ++s) {
        const int c = wave * (kTileCols / 4) + 2 * s + half, x = x_tile + c;    // this half-wave's column
        const int pu = s_pu[r][c];
        const bool ok = pu != kInvalid;
        const unsigned long long m = __ballot(ok);
        const unsigned mh = (unsigned)(half ? (m >> 32) : m);
        if (ok && x < W) {
            const unsigned long long o = colstart[x] + counts[(size_t)chunk * W + x] + (unsigned)__popc(mh & ((1u << r) - 1u));
            reinterpret_cast<float2 *>(cam)[o] = make_float2((float)x, (float)(y_tile + r));                    // :59 [i, j] = (x, y)
            reinterpret_cast<float2 *>(proj)[o] = make_float2((float)pu, (float)s_pv[r][c]);
            if (colors) {
                const unsigned rgb = s_rgb[r][c];
                colors[3 * o] = (double)(rgb & 0xffu) / 255.0;                   // :64, :69
                colors[3 * o + 1] = (double)((rgb >> 8) & 0xffu) / 255.0;
                colors[3 * o + 2] = (double)((rgb >> 16) & 0xffu) / 255.0;
            }
            if constexpr (XYZ) {
                const unsigned long long M = *total;
                pts[o] = (double)s_xyz[0][r][c];                                 // Pts (3,M) float64, :95
                pts[M + o] = (double)s_xyz[1][r][c];
                pts[2 * M + o] = (double)s_xyz[2][r][c];
            }
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

// exclusive scan of tile counts by one workgroup (ntiles <= a few 10^4)
__global__ void __launch_bounds__(1024) k_tile_scan(const unsigned *__restrict__ tile_counts, size_t ntiles,
                                                    unsigned long long *__restrict__ tile_off, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const size_t per = (ntiles + 1023) / 1024;
    const size_t i0 = min(ntiles, (size_t)t * per), i1 = min(ntiles, i0 + per);
    unsigned long long mine = 0;
    for (size_t i = i0; i < i1; ++i) mine += tile_counts[i];
    part[t] = mine;
    __syncthreads();
    if (t == 0) {
        unsigned long long acc = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned long long n = part[i];
            part[i] = acc;
            acc += n;
        }
        *total = acc;
    }
    __syncthreads();
    unsigned long long acc = part[t];
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
    const int nchunks = (cam_h + kChunkRows - 1) / kChunkRows;
    void *counts, *colstart;
    int rc = slgc_ws(ctx, 4, ((size_t)nchunks * cam_w + 1) * sizeof(unsigned), &counts);
    if (rc) return rc;
    rc = slgc_ws(ctx, 5, ((size_t)cam_w + 1) * sizeof(unsigned long long), &colstart);
    if (rc) return rc;
    const dim3 grid((cam_w + 255) / 256, nchunks);
    if (npix) hipLaunchKernelGGL(k_xmajor_count<int64_t>, grid, dim3(256), 0, ctx->stream, d_h, d_v, cam_w, cam_h, (unsigned *)counts);
    if (cam_w)
        hipLaunchKernelGGL(k_xmajor_colprefix, dim3((cam_w + 255) / 256), dim3(256), 0, ctx->stream, (unsigned *)counts, cam_w, npix ? nchunks : 0,
                           (unsigned long long *)colstart);
    hipLaunchKernelGGL(k_xmajor_colscan, dim3(1), dim3(1024), 0, ctx->stream, cam_w, (unsigned long long *)colstart, d_total);
    if (npix)
        hipLaunchKernelGGL((k_xmajor_scatter<int64_t, false>), dim3((cam_w + kTileCols - 1) / kTileCols, nchunks), dim3(256), 0, ctx->stream, d_h, d_v,
                           cam_w, cam_h, proj_w, proj_h, d_white, (const unsigned *)counts, (const unsigned long long *)colstart, d_cam, d_proj,
                           d_white ? d_colors : nullptr, (const float *)nullptr, (double *)nullptr, (const unsigned long long *)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
}

// Device-resident form of the reference-shaped product (slgc_cloud_lists_dev): int16 maps + dense float32 XYZ (+ white image) ->
// x-major cam_pts / proj_pts float32 [M][2], Pts float64 (3,M), colors float64 [M][3]; nothing synchronises with the host.
int launch_cloud_lists(slgc_ctx *ctx, const int16_t *d_h, const int16_t *d_v, const float *d_xyz, const uint8_t *d_white, int cam_w, int cam_h,
                       int proj_w, int proj_h, float *d_cam, float *d_proj, double *d_pts, double *d_colors, unsigned long long *d_total)
{
    const size_t npix = (size_t)cam_w * cam_h;
    const int nchunks = (cam_h + kChunkRows - 1) / kChunkRows;
    void *counts, *colstart;
    int rc = slgc_ws(ctx, 4, ((size_t)nchunks * cam_w + 1) * sizeof(unsigned), &counts);
    if (rc) return rc;
    rc = slgc_ws(ctx, 5, ((size_t)cam_w + 1) * sizeof(unsigned long long), &colstart);
    if (rc) return rc;
    const dim3 grid((cam_w + 255) / 256, nchunks);
    if (npix) hipLaunchKernelGGL(k_xmajor_count<int16_t>, grid, dim3(256), 0, ctx->stream, d_h, d_v, cam_w, cam_h, (unsigned *)counts);
    if (cam_w)
        hipLaunchKernelGGL(k_xmajor_colprefix, dim3((cam_w + 255) / 256), dim3(256), 0, ctx->stream, (unsigned *)counts, cam_w, npix ? nchunks : 0,
                           (unsigned long long *)colstart);
    hipLaunchKernelGGL(k_xmajor_colscan, dim3(1), dim3(1024), 0, ctx->stream, cam_w, (unsigned long long *)colstart, d_total);
    if (npix) {
        const dim3 sgrid((cam_w + kTileCols - 1) / kTileCols, nchunks);
        if (d_xyz && d_pts)
            hipLaunchKernelGGL((k_xmajor_scatter<int16_t, true>), sgrid, dim3(256), 0, ctx->stream, d_h, d_v, cam_w, cam_h, proj_w, proj_h, d_white,
                               (const unsigned *)counts, (const unsigned long long *)colstart, d_cam, d_proj, d_white ? d_colors : nullptr, d_xyz, d_pts,
                               (const unsigned long long *)d_total);
        else
            hipLaunchKernelGGL((k_xmajor_scatter<int16_t, false>), sgrid, dim3(256), 0, ctx->stream, d_h, d_v, cam_w, cam_h, proj_w, proj_h, d_white,
                               (const unsigned *)counts, (const unsigned long long *)colstart, d_cam, d_proj, d_white ? d_colors : nullptr,
                               (const float *)nullptr, (double *)nullptr, (const unsigned long long *)nullptr);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SLGC_OK;
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
