// Pair offsets: device-wide exclusive prefix sum of the per-Gaussian tile counts, in depth order.
// No counterpart in the reference (its rasterizer has no binning stage).
//
// Two launches: per-block sums (which also gathers every Gaussian's box into depth order), then per-block scan + base,
// every workgroup adding up the block sums before it (beyond 16.8 M Gaussians: a one-workgroup scan of the sums between).
// HBM-bound: per Gaussian 8 B read + one gathered record line, 8 B written, 8 B read, 4 B written.
#include "gsplat_common.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;  // consecutive elements per lane
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
constexpr int SCAN_FUSE_MAX = 4096;   // workgroups (16.8 M Gaussians) up to which the final pass sums the block totals itself

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int) { return gs_wave_incl_scan(v); }

// exclusive scan of one value per lane across the workgroup; returns (exclusive prefix, block total)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds_wave, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) lds_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < nwaves; ++w) {
        const uint32_t s = lds_wave[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

// one workgroup: exclusive scan of block_sums in place; writes the grand total to *total_out
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t* __restrict__ block_sums, int nblocks,
                                                         uint32_t* __restrict__ total_out) {
    __shared__ uint32_t lds_wave[16];
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += 1024) {
        const int idx = base + threadIdx.x;
        const uint32_t v = idx < nblocks ? block_sums[idx] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, lds_wave, total);
        if (idx < nblocks) block_sums[idx] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// Element r is the Gaussian of depth rank r.  The reduce pass gathers its box from the compact
// per-Gaussian box array (the only random access of the binning stage: 8 bytes per Gaussian out of a
// 24 MB table at 3 M Gaussians) and leaves it in rect_out[r] for the final pass and for
// duplicate_with_keys, which then read everything coalesced.
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_rect_kernel(
    const uint2* __restrict__ rects, const uint32_t* __restrict__ order, const uint32_t* __restrict__ sorted_keys,
    uint2* __restrict__ rect_out, long long n, int row_b, int row_e, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t lds_wave[SCAN_THREADS / 64];
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const long long idx = base + (long long)j * SCAN_THREADS + threadIdx.x;  // coalesced
        if (idx < n) {
            // a rank whose depth key is 0xFFFFFFFF holds a Gaussian that is not drawn (they sort last): its box is all
            // zero, so it is not fetched -- one coalesced 4-byte read instead of a random line
            const bool culled = sorted_keys && sorted_keys[idx] == 0xFFFFFFFFu;
            const uint2 r = culled ? make_uint2(0u, 0u) : rects[order[idx]];
            rect_out[idx] = r;
            GsTileSpan t;
            s += gs_tile_span(r.x, r.y, row_b, row_e, t);
        }
    }
    uint32_t total;
    block_excl_scan(s, lds_wave, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Striped arrangement so that every load and store is coalesced: wave w of the workgroup owns 1024
// consecutive elements and walks them in 16 rounds of 64.
// SUMS_RAW: block_sums holds the totals as scan_reduce_rect left them -- every workgroup adds up the totals before
// it by itself (a few coalesced reads out of L2 per lane at up to SCAN_FUSE_MAX workgroups) and the last one writes the grand
// total: no single-workgroup launch between the two passes.
template <bool SUMS_RAW>
__global__ __launch_bounds__(SCAN_THREADS) void scan_final_rect_kernel(const uint2* __restrict__ rect, long long n,
                                                                        int row_b, int row_e,
                                                                        const uint32_t* __restrict__ block_sums,
                                                                        uint32_t* __restrict__ out,
                                                                        uint32_t* __restrict__ total_out) {
    __shared__ uint32_t wave_total[SCAN_THREADS / 64];
    __shared__ uint32_t wave_before[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t before = 0;
    if (SUMS_RAW) {
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_THREADS) before += block_sums[b];
    }
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)wave * (64 * SCAN_ITEMS) + lane;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const long long idx = base + j * 64;
        v[j] = 0u;
        if (idx < n) {
            const uint2 r = rect[idx];
            GsTileSpan t;
            v[j] = gs_tile_span(r.x, r.y, row_b, row_e, t);
        }
        s += v[j];
    }
    const uint32_t wsum = wave_incl_scan(s, lane);   // lane 63: this wave's total
    if (lane == 63) wave_total[wave] = wsum;
    if (SUMS_RAW) {
        before = wave_incl_scan(before, lane);
        if (lane == 63) wave_before[wave] = before;
    }
    __syncthreads();
    uint32_t run;
    if (SUMS_RAW) {
        run = 0;
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; ++w) run += wave_before[w];
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && total_out) {
            uint32_t tot = run;
#pragma unroll
            for (int w = 0; w < SCAN_THREADS / 64; ++w) tot += wave_total[w];
            *total_out = tot;
        }
    } else {
        run = block_sums[blockIdx.x];
    }
    for (int w = 0; w < wave; ++w) run += wave_total[w];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const long long idx = base + j * 64;
        const uint32_t incl = wave_incl_scan(v[j], lane);
        if (idx < n) out[idx] = run + incl - v[j];
        run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
}

}  // namespace


// C ABI: per depth rank r, the box of Gaussian order[r] (rects[order[r]]) -> rect_in_order[r], and the exclusive scan of
// the tile counts of those boxes -> offsets[r]; total pair count to *total_dev and, when host_total
// is given, synchronously to the host.
extern "C" size_t gs_scan_workspace_bytes(long long n) {
    const long long nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    return (size_t)(nblocks + 1) * sizeof(uint32_t);
}

extern "C" int gs_scan_tiles(long long N, const void* rects, const uint32_t* order, const uint32_t* sorted_depth_keys, void* rect_in_order,
                             uint32_t* offsets, uint32_t* total_dev, void* workspace, uint32_t* host_total,
                             int tile_row_begin, int tile_row_end, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GS_REQUIRE(N >= 0 && total_dev, "bad arguments");
    GS_REQUIRE(N == 0 || (rects && order && rect_in_order && offsets && workspace), "null buffer");
    GS_REQUIRE(tile_row_begin >= 0 && tile_row_end >= tile_row_begin, "bad tile row band [%d,%d)", tile_row_begin,
               tile_row_end);
    if (N == 0) {
        GS_CHECK_HIP(hipMemsetAsync(total_dev, 0, sizeof(uint32_t), stream));
    } else {
        GS_REQUIRE(((uintptr_t)rects & 7) == 0 && ((uintptr_t)rect_in_order & 7) == 0, "misaligned buffer");
        const int nblocks = (int)((N + SCAN_TILE - 1) / SCAN_TILE);
        uint32_t* sums = (uint32_t*)workspace;
        hipLaunchKernelGGL(scan_reduce_rect_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, stream, (const uint2*)rects,
                           order, sorted_depth_keys, (uint2*)rect_in_order, N, tile_row_begin, tile_row_end, sums);
        GS_CHECK_LAUNCH("scan_reduce_rect");
        if (nblocks <= SCAN_FUSE_MAX) {
            hipLaunchKernelGGL((scan_final_rect_kernel<true>), dim3(nblocks), dim3(SCAN_THREADS), 0, stream,
                               (const uint2*)rect_in_order, N, tile_row_begin, tile_row_end, sums, offsets, total_dev);
        } else {
            hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, nblocks, total_dev);
            GS_CHECK_LAUNCH("scan_sums");
            hipLaunchKernelGGL((scan_final_rect_kernel<false>), dim3(nblocks), dim3(SCAN_THREADS), 0, stream,
                               (const uint2*)rect_in_order, N, tile_row_begin, tile_row_end, sums, offsets, total_dev);
        }
        GS_CHECK_LAUNCH("scan_final_rect");
    }
    if (host_total) {
        GS_CHECK_HIP(hipMemcpyAsync(host_total, total_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        GS_CHECK_HIP(hipStreamSynchronize(stream));
    }
    return GS_OK;
}
