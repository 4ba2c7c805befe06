// Binning stage: duplicate every drawn Gaussian once per 16x16 tile its box overlaps, keyed by
// the tile id, and find each tile's [start,end) in the tile-sorted list.
//
// The reference has no binning (it splats each Gaussian over its box in one global far-to-near
// order, gaussian_renderer/__init__.py:151-159, :366).  Here that one global order is the
// per-Gaussian depth sort (`order`, descending depth, ties by ascending Gaussian index: the sort
// is stable); pairs are emitted in that order, so a STABLE sort on the tile id alone leaves every
// tile's list far-to-near.  Bit-exact against oracle/gs_oracle.c (gso_duplicate_with_keys +
// gso_sort_pairs on tile_id << 32 | depth_key give the same list).
//
// HBM-bound: 16 B (box, id, offset) read per Gaussian, all coalesced (the boxes were gathered into
// depth order by gs_scan_tiles), 8 B written per pair.
#include "gsplat_common.h"

namespace {

__global__ __launch_bounds__(256) void duplicate_with_keys_kernel(int N, int tiles_x, int row_b, int row_e,
                                                                   const uint32_t* __restrict__ order,
                                                                   const uint2* __restrict__ rect_in_order,
                                                                   const uint32_t* __restrict__ offsets,
                                                                   uint32_t* __restrict__ keys,
                                                                   uint32_t* __restrict__ vals, size_t capacity) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    // lane r emits the Gaussian of depth rank r at offsets[r]; all reads are coalesced
    const uint2 rect = rect_in_order[r];
    GsTileSpan t;
    if (gs_tile_span(rect.x, rect.y, row_b, row_e, t) == 0) return;
    const uint32_t i = order[r];
    size_t o = offsets[r];
    for (int ty = t.ty0; ty < t.ty1; ++ty)
        for (int tx = t.tx0; tx < t.tx1; ++tx) {
            if (o < capacity) {   // pairs beyond the caller's buffers are dropped (the caller checks D afterwards)
                keys[o] = (uint32_t)(ty * tiles_x + tx);
                vals[o] = i;
            }
            ++o;
        }
}

__global__ __launch_bounds__(256) void identify_tile_ranges_kernel(long long cap,
                                                                    const uint32_t* __restrict__ count_dev,
                                                                    const uint32_t* __restrict__ keys,
                                                                    uint32_t* __restrict__ ranges) {
    long long D = cap;
    if (count_dev) {
        const long long c = (long long)*count_dev;
        D = c < cap ? c : cap;
    }
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const uint32_t t = keys[i];
    if (i == 0 || keys[i - 1] != t) ranges[2 * (size_t)t] = (uint32_t)i;
    if (i == D - 1 || keys[i + 1] != t) ranges[2 * (size_t)t + 1] = (uint32_t)(i + 1);
}

}  // namespace

extern "C" int gs_duplicate_with_keys(int N, int W, const void* rect_in_order, const uint32_t* order,
                                      const uint32_t* offsets, uint32_t* keys, uint32_t* vals, long long capacity,
                                      int tile_row_begin, int tile_row_end, void* stream_) {
    GS_REQUIRE(N >= 0 && W > 0 && capacity >= 0, "bad arguments");
    if (N == 0) return GS_OK;
    GS_REQUIRE(rect_in_order && order && offsets && keys && vals, "null buffer");
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE;
    hipLaunchKernelGGL(duplicate_with_keys_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream_, N,
                       tiles_x, tile_row_begin, tile_row_end, order, (const uint2*)rect_in_order, offsets, keys, vals,
                       (size_t)capacity);
    GS_CHECK_LAUNCH("duplicate_with_keys");
    return GS_OK;
}

extern "C" int gs_identify_tile_ranges(long long D, const uint32_t* count_dev, int num_tiles, const uint32_t* keys,
                                       uint32_t* ranges, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GS_REQUIRE(D >= 0 && num_tiles > 0 && ranges, "bad arguments");
    GS_CHECK_HIP(hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)num_tiles, stream));
    if (D == 0) return GS_OK;
    GS_REQUIRE(keys, "null keys");
    hipLaunchKernelGGL(identify_tile_ranges_kernel, dim3(gs_div_up(D, 256)), dim3(256), 0, stream, D, count_dev, keys,
                       ranges);
    GS_CHECK_LAUNCH("identify_tile_ranges");
    return GS_OK;
}
