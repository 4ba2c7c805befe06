// Device-wide exclusive prefix sum of uint32 (tiles_touched -> pair offsets; radix digit tables).
// No counterpart in the reference (its rasterizer has no binning stage).
//
// Three launches: per-block sums, one workgroup scanning the block sums, per-block scan + base.
// HBM-bound: 4 B read twice + 4 B written per element.
#include "gsplat_common.h"
#include "scan.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;  // consecutive elements per lane
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

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

// with `gather`: sums in[gather[i]] and leaves the gathered values in gathered[i] for the final pass
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ in,
                                                                    const uint32_t* __restrict__ gather,
                                                                    uint32_t* __restrict__ gathered, long long n,
                                                                    uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t lds_wave[SCAN_THREADS / 64];
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const long long idx = base + (long long)j * SCAN_THREADS + threadIdx.x;  // coalesced
        if (idx < n) {
            uint32_t v;
            if (gather) {
                v = in[gather[idx]];
                gathered[idx] = v;
            } else {
                v = in[idx];
            }
            s += v;
        }
    }
    uint32_t total;
    block_excl_scan(s, lds_wave, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
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

// in == out is allowed (each lane reads its 16 elements before it writes them)
__global__ __launch_bounds__(SCAN_THREADS) void scan_final_kernel(const uint32_t* in, long long n,
                                                                   const uint32_t* __restrict__ block_sums,
                                                                   uint32_t* out) {
    __shared__ uint32_t lds_wave[SCAN_THREADS / 64];
    // blocked arrangement: lane t owns elements [t*ITEMS, (t+1)*ITEMS) of the tile
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
    if (base + SCAN_ITEMS <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS / 4; ++j) {
            const uint4 t = p[j];
            v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) v[j] = (base + j < n) ? in[base + j] : 0u;
    }
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) s += v[j];
    uint32_t total;
    uint32_t run = block_excl_scan(s, lds_wave, total) + block_sums[blockIdx.x];
    if (base + SCAN_ITEMS <= n) {
        uint4* q = reinterpret_cast<uint4*>(out + base);
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS / 4; ++j) {
            uint4 t;
            t.x = run; run += v[4 * j];
            t.y = run; run += v[4 * j + 1];
            t.z = run; run += v[4 * j + 2];
            t.w = run; run += v[4 * j + 3];
            q[j] = t;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) {
            if (base + j < n) out[base + j] = run;
            run += v[j];
        }
    }
}

}  // namespace

size_t gs_scan_ws_bytes(long long n) {
    const long long nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    return (size_t)(nblocks + 1) * sizeof(uint32_t);
}

int gs_exclusive_scan_u32(const uint32_t* in, uint32_t* out, long long n, uint32_t* total_out,
                          const uint32_t* gather, uint32_t* gathered, void* ws, hipStream_t stream) {
    if (n <= 0) {
        if (total_out) GS_CHECK_HIP(hipMemsetAsync(total_out, 0, sizeof(uint32_t), stream));
        return GS_OK;
    }
    GS_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, "scan buffers must be 16-byte aligned");
    GS_REQUIRE(!gather || (gathered && ((uintptr_t)gathered & 15) == 0), "gathered scan needs a 16-byte aligned buffer");
    const int nblocks = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    uint32_t* sums = (uint32_t*)ws;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, stream, in, gather, gathered, n, sums);
    GS_CHECK_LAUNCH("scan_reduce");
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, nblocks, total_out);
    GS_CHECK_LAUNCH("scan_sums");
    hipLaunchKernelGGL(scan_final_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, stream, gather ? gathered : in, n, sums,
                       out);
    GS_CHECK_LAUNCH("scan_final");
    return GS_OK;
}

// C ABI: exclusive scan of tiles_touched -> offsets; total pair count to *total (device) and,
// when host_total is given, synchronously to the host.
extern "C" size_t gs_scan_workspace_bytes(long long n) { return gs_scan_ws_bytes(n); }

extern "C" int gs_scan_tiles(long long N, const uint32_t* tiles_touched, const uint32_t* order,
                             uint32_t* tiles_in_order, uint32_t* offsets, uint32_t* total_dev, void* workspace,
                             uint32_t* host_total, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GS_REQUIRE(N >= 0 && total_dev, "bad arguments");
    GS_REQUIRE(N == 0 || (tiles_touched && offsets && workspace), "null buffer");
    GS_REQUIRE(N == 0 || !order || tiles_in_order, "tiles_in_order is required with order");
    int rc = gs_exclusive_scan_u32(tiles_touched, offsets, N, total_dev, order, tiles_in_order, workspace, stream);
    if (rc) return rc;
    if (host_total) {
        GS_CHECK_HIP(hipMemcpyAsync(host_total, total_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        GS_CHECK_HIP(hipStreamSynchronize(stream));
    }
    return GS_OK;
}
