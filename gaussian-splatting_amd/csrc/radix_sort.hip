// In-house stable LSD radix sort of (uint64 key, uint32 value) pairs for the binning stage.
// No counterpart in the reference (torch.argsort at gaussian_renderer/__init__.py:153 is the only
// ordering it has); the result is bit-exact against oracle/gs_oracle.c::gso_sort_pairs.
//
// 8 bits per pass over a caller-given bit range.  Two uses per frame: (1) 32-bit depth keys of the
// N Gaussians (4 passes over N items) and (2) the 64-bit tile|depth keys of the D pairs on the
// tile bits only (2 passes): pairs are emitted in depth order, so they are already sorted on the
// low 32 bits and a stable sort of the tile bits alone yields exactly the order a full 45-bit
// sort would (the full sort is still available: bit_begin = 0).
// Per pass:
//   1. histogram: each workgroup counts the digits of its 4096-key tile (wave64 ballot "match":
//      one LDS add per distinct digit per wave, so skewed tile digits cost the same as random ones);
//   2. exclusive scan of the digit-major [256][blocks] table (scan.hip) -> global base of every
//      (digit, block) run;
//   3. scatter: stable rank of every key inside the workgroup from wave64 ballots + per-wave LDS
//      counters, keys staged through LDS in digit order so each run is written contiguously.
// HBM-bound: per pass 8 B read (histogram) + 12 B read + 12 B written per pair.
#include "gsplat_common.h"
#include "scan.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 pairs per workgroup
constexpr int RS_RADIX = 256;

// lanes of this wave holding the same 8-bit digit (only among `valid` lanes)
__device__ __forceinline__ uint64_t match_digit(uint32_t digit, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return valid ? peers : 0ull;
}

__device__ __forceinline__ uint64_t lanemask_lt() {
    const int lane = threadIdx.x & 63;
    return (1ull << lane) - 1ull;
}

template <typename KeyT>
__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const KeyT* __restrict__ keys, long long D,
                                                                 int shift, int nblocks,
                                                                 uint32_t* __restrict__ table) {
    __shared__ uint32_t hist[RS_RADIX];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long base = (long long)blockIdx.x * RS_TILE + (long long)wave * 64 * RS_ITEMS;
#pragma unroll 4
    for (int it = 0; it < RS_ITEMS; ++it) {
        const long long idx = base + it * 64 + lane;
        const bool valid = idx < D;
        const uint32_t d = valid ? (uint32_t)(keys[idx] >> shift) & 0xFFu : 0u;
        const uint64_t peers = match_digit(d, valid);
        if (valid && (peers & lanemask_lt()) == 0) atomicAdd(&hist[d], (uint32_t)__popcll(peers));
    }
    __syncthreads();
    table[(size_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

template <typename KeyT>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(
    const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, KeyT* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, long long D, int shift, int nblocks, const uint32_t* __restrict__ table) {
    __shared__ KeyT skeys[RS_TILE];
    __shared__ uint32_t svals[RS_TILE];
    __shared__ uint32_t wave_cnt[RS_WAVES][RS_RADIX];
    __shared__ uint32_t digit_base[RS_RADIX];
    __shared__ uint32_t gofs[RS_RADIX];
    __shared__ uint32_t scan_tmp[RS_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) wave_cnt[w][tid] = 0;
    __syncthreads();

    const long long block_base = (long long)blockIdx.x * RS_TILE;
    const long long base = block_base + (long long)wave * 64 * RS_ITEMS;
    KeyT key[RS_ITEMS];
    uint32_t val[RS_ITEMS];
    uint32_t rank[RS_ITEMS];
    const uint64_t lt = lanemask_lt();
    volatile uint32_t* my_cnt = wave_cnt[wave];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const long long idx = base + it * 64 + lane;
        const bool valid = idx < D;
        key[it] = valid ? keys_in[idx] : (KeyT)~(KeyT)0;
        val[it] = valid ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const long long idx = base + it * 64 + lane;
        const bool valid = idx < D;
        const uint32_t d = (uint32_t)(key[it] >> shift) & 0xFFu;
        const uint64_t peers = match_digit(d, valid);
        uint32_t prior = 0;
        if (valid) prior = my_cnt[d];  // every peer reads the same counter ...
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt) == 0) my_cnt[d] = prior + (uint32_t)__popcll(peers);  // ... the first one bumps it
        __builtin_amdgcn_wave_barrier();
        rank[it] = prior + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    // one lane per digit: counts of earlier waves, then the workgroup-local start of every digit
    {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const uint32_t c = wave_cnt[w][tid];
            wave_cnt[w][tid] = run;
            run += c;
        }
        // exclusive scan of `run` over the 256 digits
        uint32_t incl = run;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const uint32_t t = __shfl_up(incl, dlt, 64);
            if (lane >= dlt) incl += t;
        }
        if (lane == 63) scan_tmp[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w)
            if (w < wave) wbase += scan_tmp[w];
        const uint32_t excl = wbase + incl - run;
        digit_base[tid] = excl;
        gofs[tid] = table[(size_t)tid * nblocks + blockIdx.x] - excl;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const long long idx = base + it * 64 + lane;
        if (idx < D) {
            const uint32_t d = (uint32_t)(key[it] >> shift) & 0xFFu;
            const uint32_t p = digit_base[d] + wave_cnt[wave][d] + rank[it];
            skeys[p] = key[it];
            svals[p] = val[it];
        }
    }
    __syncthreads();
    const long long remaining = D - block_base;
    const int count = remaining < RS_TILE ? (int)remaining : RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int p = j * RS_THREADS + tid;
        if (p < count) {
            const KeyT k = skeys[p];
            const uint32_t d = (uint32_t)(k >> shift) & 0xFFu;
            const uint32_t o = (uint32_t)p + gofs[d];
            keys_out[o] = k;
            vals_out[o] = svals[p];
        }
    }
}

}  // namespace

extern "C" size_t gs_sort_workspace_bytes(long long D);

namespace {

template <typename KeyT>
int sort_pairs_impl(long long D, int bit_begin, int bit_end, KeyT* keys_a, uint32_t* vals_a, KeyT* keys_b,
                    uint32_t* vals_b, void* workspace, size_t workspace_bytes, int* result_in_b,
                    hipStream_t stream) {
    GS_REQUIRE(D >= 0 && D < (1ll << 32) && result_in_b, "bad arguments");
    GS_REQUIRE(bit_begin >= 0 && bit_begin <= bit_end && bit_end <= (int)(8 * sizeof(KeyT)), "bad bit range [%d,%d)",
               bit_begin, bit_end);
    *result_in_b = 0;
    if (D == 0 || bit_begin == bit_end) return GS_OK;
    GS_REQUIRE(keys_a && vals_a && keys_b && vals_b && workspace, "null buffer");
    GS_REQUIRE(workspace_bytes >= gs_sort_workspace_bytes(D), "sort workspace too small");
    GS_REQUIRE(((uintptr_t)workspace & 15) == 0, "sort workspace must be 16-byte aligned");
    const int nblocks = (int)((D + RS_TILE - 1) / RS_TILE);
    const long long table_n = (long long)nblocks * RS_RADIX;
    uint32_t* table = (uint32_t*)workspace;
    size_t tb = ((size_t)table_n * sizeof(uint32_t) + 255) & ~(size_t)255;
    void* scan_ws = (char*)workspace + tb;
    KeyT *ks = keys_a, *kd = keys_b;
    uint32_t *vs = vals_a, *vd = vals_b;
    int passes = 0;
    for (int shift = bit_begin; shift < bit_end; shift += 8) {
        hipLaunchKernelGGL((radix_hist_kernel<KeyT>), dim3(nblocks), dim3(RS_THREADS), 0, stream, ks, D, shift,
                           nblocks, table);
        GS_CHECK_LAUNCH("radix_hist");
        int rc = gs_exclusive_scan_u32(table, table, table_n, nullptr, nullptr, scan_ws, stream);
        if (rc) return rc;
        hipLaunchKernelGGL((radix_scatter_kernel<KeyT>), dim3(nblocks), dim3(RS_THREADS), 0, stream, ks, vs, kd, vd,
                           D, shift, nblocks, table);
        GS_CHECK_LAUNCH("radix_scatter");
        KeyT* tk = ks; ks = kd; kd = tk;
        uint32_t* tv = vs; vs = vd; vd = tv;
        ++passes;
    }
    *result_in_b = passes & 1;
    return GS_OK;
}

}  // namespace

extern "C" size_t gs_sort_workspace_bytes(long long D) {
    const long long nblocks = (D + RS_TILE - 1) / RS_TILE;
    const long long table = nblocks * RS_RADIX;
    size_t bytes = (size_t)table * sizeof(uint32_t);
    bytes = (bytes + 255) & ~(size_t)255;
    return bytes + gs_scan_ws_bytes(table) + 256;
}

// Stable sort of (uint64 key, uint32 value) pairs on key bits [bit_begin, bit_end).  keys_a/vals_a hold
// the input; *result_in_b tells which buffer pair holds the output (b after an odd number of passes).
extern "C" int gs_sort_pairs(long long D, int bit_begin, int bit_end, uint64_t* keys_a, uint32_t* vals_a,
                             uint64_t* keys_b, uint32_t* vals_b, void* workspace, size_t workspace_bytes,
                             int* result_in_b, void* stream_) {
    return sort_pairs_impl<uint64_t>(D, bit_begin, bit_end, keys_a, vals_a, keys_b, vals_b, workspace,
                                     workspace_bytes, result_in_b, (hipStream_t)stream_);
}

// Same for uint32 keys (the per-Gaussian depth sort).
extern "C" int gs_sort_pairs32(long long D, int bit_begin, int bit_end, uint32_t* keys_a, uint32_t* vals_a,
                               uint32_t* keys_b, uint32_t* vals_b, void* workspace, size_t workspace_bytes,
                               int* result_in_b, void* stream_) {
    return sort_pairs_impl<uint32_t>(D, bit_begin, bit_end, keys_a, vals_a, keys_b, vals_b, workspace,
                                     workspace_bytes, result_in_b, (hipStream_t)stream_);
}
