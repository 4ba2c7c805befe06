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

constexpr int TR_ITEMS = 8;            // consecutive keys per lane
constexpr int TR_BIG_GAP = 16;         // longer runs of empty tiles are cleared by the whole workgroup
constexpr int TR_MAX_BIG = 64;

// Every lane takes TR_ITEMS consecutive keys of the tile-sorted list (two 16-byte loads) and the key before them; a change
// of key between positions i - 1 and i ends tile keys[i - 1] at i, starts tile keys[i] at i -- and says that the tiles
// between the two are EMPTY: whoever sees the change writes their (0, 0) too, so the table needs no clearing pass
// (short runs by the lane itself, long ones -- and the tiles before the first and behind the last key -- by the whole
// workgroup, 256 tiles per step).
__global__ __launch_bounds__(256) void identify_tile_ranges_kernel(long long cap,
                                                                    const uint32_t* __restrict__ count_dev,
                                                                    const uint32_t* __restrict__ keys, int num_tiles,
                                                                    uint2* __restrict__ ranges) {
    __shared__ uint2 big[TR_MAX_BIG];
    __shared__ int nbig;
    long long D = cap;
    if (count_dev) {
        const long long c = (long long)*count_dev;
        D = c < cap ? c : cap;
    }
    const uint32_t nt = (uint32_t)num_tiles;
    const long long wg0 = (long long)blockIdx.x * (256 * TR_ITEMS);
    if (wg0 >= D && !(D == 0 && blockIdx.x == 0)) return;      // workgroup-uniform
    if (threadIdx.x == 0) nbig = 0;
    __syncthreads();
    // tiles (a, b) exclusive are empty
    auto clear_between = [&](long long a, long long b) {
        if (b - a - 1 <= TR_BIG_GAP) {
            for (long long t = a + 1; t < b; ++t) ranges[t] = make_uint2(0u, 0u);
        } else {
            const int slot = atomicAdd(&nbig, 1);
            if (slot < TR_MAX_BIG) big[slot] = make_uint2((uint32_t)(a + 1), (uint32_t)b);
            else for (long long t = a + 1; t < b; ++t) ranges[t] = make_uint2(0u, 0u);
        }
    };
    if (D == 0) {
        if (threadIdx.x == 0) clear_between(-1, num_tiles);
    } else {
        const long long i0 = wg0 + (long long)threadIdx.x * TR_ITEMS;
        if (i0 < D) {
            uint32_t k[TR_ITEMS + 1];
            if (i0 + TR_ITEMS <= D) {
                const uint4 lo = *reinterpret_cast<const uint4*>(keys + i0), hi = *reinterpret_cast<const uint4*>(keys + i0 + 4);
                k[1] = lo.x; k[2] = lo.y; k[3] = lo.z; k[4] = lo.w; k[5] = hi.x; k[6] = hi.y; k[7] = hi.z; k[8] = hi.w;
            } else {
#pragma unroll
                for (int j = 0; j < TR_ITEMS; ++j) k[j + 1] = i0 + j < D ? keys[i0 + j] : 0u;
            }
            k[0] = i0 > 0 ? keys[i0 - 1] : 0u;
            // A key that is no tile id (never emitted by gs_duplicate_with_keys; sorted, so only at the end of the list) owns no
            // entry, but the table is still written in full: such keys count as one tile BEHIND the table, so the last valid
            // tile is closed where they begin and every tile behind it is cleared.
#pragma unroll
            for (int j = 0; j <= TR_ITEMS; ++j) k[j] = k[j] < nt ? k[j] : nt;
            if (i0 == 0) {
                clear_between(-1, k[1]);
                if (k[1] < nt) ranges[k[1]].x = 0u;
            }
#pragma unroll
            for (int j = 0; j < TR_ITEMS; ++j) {
                const long long i = i0 + j;
                if (i < D && i > 0 && k[j] != k[j + 1] && k[j] < nt) {   // (sorted tile ids: k[j] < k[j + 1] <= nt)
                    ranges[k[j]].y = (uint32_t)i;
                    if (k[j + 1] < nt) ranges[k[j + 1]].x = (uint32_t)i;
                    clear_between(k[j], k[j + 1]);
                }
                if (i == D - 1 && k[j + 1] < nt) {
                    ranges[k[j + 1]].y = (uint32_t)D;
                    clear_between(k[j + 1], num_tiles);
                }
            }
        }
    }
    __syncthreads();
    const int nb = min(nbig, TR_MAX_BIG);
    for (int g = 0; g < nb; ++g)
        for (uint32_t t = big[g].x + threadIdx.x; t < big[g].y; t += 256) ranges[t] = make_uint2(0u, 0u);
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
    GS_REQUIRE(D == 0 || keys, "null keys");
    GS_REQUIRE(((uintptr_t)keys & 15) == 0 && ((uintptr_t)ranges & 7) == 0, "keys must be 16-byte, ranges 8-byte aligned");
    // keys above num_tiles - 1 own no entry (the kernel treats them as one tile behind the table); the caller's keys are tile
    // ids by construction
    const long long per_wg = 256 * TR_ITEMS;
    const long long nwg = D > 0 ? (D + per_wg - 1) / per_wg : 1;
    hipLaunchKernelGGL(identify_tile_ranges_kernel, dim3((unsigned)nwg), dim3(256), 0, stream, D, count_dev, keys, num_tiles,
                       (uint2*)ranges);
    GS_CHECK_LAUNCH("identify_tile_ranges");
    return GS_OK;
}
