// In-house stable LSD radix sort of (uint32 key, uint32 value) pairs for the binning stage.
// No counterpart in the reference (torch.argsort at gaussian_renderer/__init__.py:153 is the only
// ordering it has); the result is bit-exact against oracle/gs_oracle.c::gso_sort_pairs.
//
// 8 bits per pass over a caller-given bit range.  Two uses per frame: (1) the 32-bit depth keys of
// the N Gaussians (4 passes) -- the reference's one global far-to-near order -- and (2) the tile
// ids of the D (tile, Gaussian) pairs (2 passes for up to 65536 tiles): pairs are emitted in depth
// order, so a stable sort on the tile id alone leaves every tile's list far-to-near.
//
// Two launches per pass (no separate scan kernels):
//   1. histogram: each workgroup counts the digits of its 4096-key tile with LDS adds (one add per
//      wave when all its 64 keys share the digit), stores its 256 counts (blk table) and adds them
//      to the counts of its group of 32 workgroups (grp table, low-contention global atomics);
//   2. scatter: lane d of every workgroup sums the grp rows before its group, the blk rows before
//      it inside its group, and all grp rows (digit totals -> block scan) to get the global base of
//      its (digit, workgroup) run -- ~(nblocks/32 + 32) independent coalesced 1 KB reads from L2,
//      issued before the keys are loaded.  Then the stable rank of every key inside the workgroup
//      from wave64 ballots + per-wave LDS counters, keys staged through LDS in digit order so each
//      run is written contiguously.
// HBM-bound: per pass 4 B read (histogram) + 8 B read + 8 B written per pair.
#include "gsplat_common.h"
#include <mutex>

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 pairs per workgroup
constexpr int RS_RADIX = 256;
constexpr int RS_GROUP = 32;                    // workgroups per grp row
constexpr int RS_MAX_PASSES = 4;

// lanes of this wave holding the same NBITS-bit digit (only among `valid` lanes).  Per digit bit: x = the bit spread over
// a register (0 / ~0, one v_bfe_i32), m = the ballot of it (one v_cmp), and peers &= ~(m ^ x) -- lanes whose bit equals
// mine -- as one v_bitop3_b32 per half: four instructions per bit where the select-based form took nine.
template <int NBITS>
__device__ __forceinline__ uint64_t match_digit(uint32_t key, int shift, bool valid) {
    const uint64_t v = __builtin_amdgcn_ballot_w64(valid);
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll
    for (int b = 0; b < NBITS; ++b) {
        const uint32_t x = (uint32_t)__builtin_amdgcn_sbfe((int)key, (uint32_t)(shift + b), 1u);   // bit shift + b of the key
        const uint64_t m = __builtin_amdgcn_ballot_w64(x != 0u);
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)m, x, 0x90);           // lo & ~(m ^ x)
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(m >> 32), x, 0x90);
    }
    const uint64_t peers = ((uint64_t)hi << 32) | lo;
    return valid ? peers : 0ull;
}

__device__ __forceinline__ uint64_t lanemask_lt() {
    const int lane = threadIdx.x & 63;
    return (1ull << lane) - 1ull;
}

// count_dev != nullptr: the number of pairs lives on the device (no host read-back); `cap` sized the grid
__device__ __forceinline__ long long actual_count(long long cap, const uint32_t* __restrict__ count_dev) {
    if (!count_dev) return cap;
    const long long c = (long long)*count_dev;
    return c < cap ? c : cap;
}

__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys, long long cap,
                                                                 const uint32_t* __restrict__ count_dev,
                                                                 int shift, uint32_t mask,
                                                                 uint32_t* __restrict__ blk,
                                                                 uint32_t* __restrict__ grp) {
    __shared__ uint32_t hist[RS_RADIX];
    const long long D = actual_count(cap, count_dev);
    if ((long long)blockIdx.x * RS_TILE >= D) return;   // workgroup-uniform
    hist[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long base = (long long)blockIdx.x * RS_TILE + (long long)wave * 64 * RS_ITEMS;
    uint32_t key[RS_ITEMS];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {   // all loads in flight before the first LDS add
        const long long idx = base + it * 64 + lane;
        key[it] = idx < D ? keys[idx] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const long long idx = base + it * 64 + lane;
        const bool valid = idx < D;
        const uint32_t d = (key[it] >> shift) & mask;
        // LDS adds cost one cycle per lane on the same address: a wave whose 64 keys share the digit
        // (high bytes of depth keys, tile ids of a sorted list) adds once; anything else goes to
        // the LDS atomic unit directly (cheaper than ranking the lanes with ballots)
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
        if (__all(d == d0 || !valid)) {
            const uint64_t m = __ballot(valid);
            if (lane == 0 && m) atomicAdd(&hist[d0], (uint32_t)__popcll(m));
        } else if (valid) {
            atomicAdd(&hist[d], 1u);
        }
    }
    __syncthreads();
    const uint32_t c = hist[threadIdx.x];
    blk[(size_t)blockIdx.x * RS_RADIX + threadIdx.x] = c;
    if (c) atomicAdd(&grp[(size_t)(blockIdx.x / RS_GROUP) * RS_RADIX + threadIdx.x], c);
}

// NBITS: width of this pass's digit (the last pass of a bit range that is no multiple of 8 is narrower -- the tile sort's
// second pass has 5 bits at 4 346 tiles -- and ranks its keys with that many ballots instead of 8)
template <int NBITS>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, long long cap, const uint32_t* __restrict__ count_dev, int shift, uint32_t mask,
    const uint32_t* __restrict__ blk, const uint32_t* __restrict__ grp) {
    __shared__ uint32_t skeys[RS_TILE];
    __shared__ uint32_t svals[RS_TILE];
    __shared__ uint32_t wave_cnt[RS_WAVES][RS_RADIX];
    __shared__ uint32_t digit_base[RS_RADIX];
    __shared__ uint32_t gofs[RS_RADIX];
    __shared__ uint32_t scan_tmp[2][RS_WAVES];

    const long long D = actual_count(cap, count_dev);
    if ((long long)blockIdx.x * RS_TILE >= D) return;   // workgroup-uniform
    const int ngroups = (int)(((D + RS_TILE - 1) / RS_TILE + RS_GROUP - 1) / RS_GROUP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) wave_cnt[w][tid] = 0;

    // lane `tid` owns digit `tid`: keys of this digit in earlier workgroups, and in all workgroups
    uint32_t before = 0, total = 0;
    {
        const int g0 = blockIdx.x / RS_GROUP;
#pragma unroll 8
        for (int g = 0; g < ngroups; ++g) {
            const uint32_t c = grp[(size_t)g * RS_RADIX + tid];
            total += c;
            before += g < g0 ? c : 0u;
        }
#pragma unroll 8
        for (int b = g0 * RS_GROUP; b < (int)blockIdx.x; ++b) before += blk[(size_t)b * RS_RADIX + tid];
    }
    __syncthreads();

    const long long block_base = (long long)blockIdx.x * RS_TILE;
    const long long base = block_base + (long long)wave * 64 * RS_ITEMS;
    uint32_t key[RS_ITEMS];
    uint32_t val[RS_ITEMS];
    uint32_t rank[RS_ITEMS];
    const uint64_t lt = lanemask_lt();
    // volatile (every peer reads the counter before the first one bumps it) AND explicitly in the LDS address space: through
    // a generic volatile pointer the compiler emits flat loads / stores with system scope for these 32 accesses per lane
    typedef __attribute__((address_space(3))) volatile uint32_t lds_counter_t;
    lds_counter_t* my_cnt = (lds_counter_t*)wave_cnt[wave];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const long long idx = base + it * 64 + lane;
        const bool valid = idx < D;
        key[it] = valid ? keys_in[idx] : 0xFFFFFFFFu;
        val[it] = valid ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const long long idx = base + it * 64 + lane;
        const bool valid = idx < D;
        const uint32_t d = (key[it] >> shift) & mask;
        const uint64_t peers = match_digit<NBITS>(key[it], shift, valid);
        uint32_t prior = 0;
        if (valid) prior = my_cnt[d];  // every peer reads the same counter ...
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt) == 0) my_cnt[d] = prior + (uint32_t)__popcll(peers);  // ... the first one bumps it
        __builtin_amdgcn_wave_barrier();
        rank[it] = prior + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    // one lane per digit: counts of earlier waves, then the workgroup-local and the global start of
    // every digit (two exclusive scans over the 256 digits: local counts and global totals)
    {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const uint32_t c = wave_cnt[w][tid];
            wave_cnt[w][tid] = run;
            run += c;
        }
        const uint32_t incl = gs_wave_incl_scan(run), gincl = gs_wave_incl_scan(total);
        if (lane == 63) { scan_tmp[0][wave] = incl; scan_tmp[1][wave] = gincl; }
        __syncthreads();
        uint32_t wbase = 0, gbase = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w)
            if (w < wave) { wbase += scan_tmp[0][w]; gbase += scan_tmp[1][w]; }
        const uint32_t excl = wbase + incl - run;
        digit_base[tid] = excl;
        gofs[tid] = (gbase + gincl - total) + before - excl;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const long long idx = base + it * 64 + lane;
        if (idx < D) {
            const uint32_t d = (key[it] >> shift) & mask;
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
            const uint32_t k = skeys[p];
            const uint32_t d = (k >> shift) & mask;
            const uint32_t o = (uint32_t)p + gofs[d];
            keys_out[o] = k;
            vals_out[o] = svals[p];
        }
    }
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// workspace: blk table [nblocks][256] (rewritten every pass) + one grp table [ngroups][256] per pass
extern "C" size_t gs_sort_workspace_bytes(long long D) {
    const long long nblocks = (D + RS_TILE - 1) / RS_TILE;
    const long long ngroups = (nblocks + RS_GROUP - 1) / RS_GROUP;
    return align256((size_t)nblocks * RS_RADIX * sizeof(uint32_t)) +
           RS_MAX_PASSES * align256((size_t)ngroups * RS_RADIX * sizeof(uint32_t)) + 256;
}

// Stable sort of (uint32 key, uint32 value) pairs on key bits [bit_begin, bit_end).  keys_a/vals_a hold
// the input; *result_in_b tells which buffer pair holds the output (b after an odd number of passes).
// count_dev (device, may be NULL): sort min(*count_dev, D) pairs -- D is then only the capacity that
// sizes grids and workspace, so the caller needs no host read-back of the pair count.
static int sort_pairs_impl(long long D, const uint32_t* count_dev, int bit_begin, int bit_end, uint32_t* keys_a,
                           uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, void* workspace,
                           size_t workspace_bytes, int* result_in_b, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GS_REQUIRE(D >= 0 && D < (1ll << 32) && result_in_b, "bad arguments");
    GS_REQUIRE(bit_begin >= 0 && bit_begin <= bit_end && bit_end <= 32, "bad bit range [%d,%d)", bit_begin, bit_end);
    *result_in_b = 0;
    if (D == 0 || bit_begin == bit_end) return GS_OK;
    GS_REQUIRE(keys_a && vals_a && keys_b && vals_b && workspace, "null buffer");
    GS_REQUIRE(workspace_bytes >= gs_sort_workspace_bytes(D), "sort workspace too small");
    GS_REQUIRE(((uintptr_t)workspace & 15) == 0, "sort workspace must be 16-byte aligned");
    const int nblocks = (int)((D + RS_TILE - 1) / RS_TILE);
    const int ngroups = (nblocks + RS_GROUP - 1) / RS_GROUP;
    const int npasses = (bit_end - bit_begin + 7) / 8;
    uint32_t* blk = (uint32_t*)workspace;
    const size_t blk_bytes = align256((size_t)nblocks * RS_RADIX * sizeof(uint32_t));
    const size_t grp_bytes = align256((size_t)ngroups * RS_RADIX * sizeof(uint32_t));
    char* grp0 = (char*)workspace + blk_bytes;
    GS_CHECK_HIP(hipMemsetAsync(grp0, 0, grp_bytes * npasses, stream));
    uint32_t *ks = keys_a, *kd = keys_b, *vs = vals_a, *vd = vals_b;
    int pass = 0;
    for (int shift = bit_begin; shift < bit_end; shift += 8, ++pass) {
        uint32_t* grp = (uint32_t*)(grp0 + grp_bytes * pass);
        const uint32_t mask = bit_end - shift >= 8 ? 0xFFu : (1u << (bit_end - shift)) - 1u;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(RS_THREADS), 0, stream, ks, D, count_dev, shift, mask,
                           blk, grp);
        GS_CHECK_LAUNCH("radix_hist");
        const int nbits = bit_end - shift >= 8 ? 8 : bit_end - shift;
#define GS_SCATTER(NB)                                                                                              \
    hipLaunchKernelGGL((radix_scatter_kernel<NB>), dim3(nblocks), dim3(RS_THREADS), 0, stream, ks, vs, kd, vd, D, \
                       count_dev, shift, mask, blk, grp)
        switch (nbits) {
            case 1: GS_SCATTER(1); break;      // exact widths: the ranking reads the key's bits [shift, shift + NBITS) unmasked
            case 2: GS_SCATTER(2); break;
            case 3: GS_SCATTER(3); break;
            case 4: GS_SCATTER(4); break;
            case 5: GS_SCATTER(5); break;
            case 6: GS_SCATTER(6); break;
            case 7: GS_SCATTER(7); break;
            default: GS_SCATTER(8); break;
        }
#undef GS_SCATTER
        GS_CHECK_LAUNCH("radix_scatter");
        uint32_t* tk = ks; ks = kd; kd = tk;
        uint32_t* tv = vs; vs = vd; vd = tv;
    }
    *result_in_b = pass & 1;
    return GS_OK;
}

extern "C" int gs_sort_pairs(long long D, const uint32_t* count_dev, int bit_begin, int bit_end, uint32_t* keys_a,
                             uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, void* workspace,
                             size_t workspace_bytes, int* result_in_b, void* stream_) {
    return sort_pairs_impl(D, count_dev, bit_begin, bit_end, keys_a, vals_a, keys_b, vals_b, workspace, workspace_bytes,
                           result_in_b, stream_);
}

// ---------------------------------------------------------------------------------------------
// gs_sort_pairs_static: the same sort for callers that cannot allocate (HIP graph capture, serving
// loops, harnesses that forbid hipMalloc).  The histogram tables (blk + grp, what gs_sort_pairs takes
// as `workspace`) live in a file-scope device array instead; every GPU the library runs on holds its
// own copy of it until the process ends.  `scratch` below is only the HOST-side handle of that array:
// the per-device address comes from hipGetSymbolAddress, once per device, and only that pointer is
// ever handed to kernels and memsets.
// ---------------------------------------------------------------------------------------------
#define GS_SORT_STATIC_MAX_PAIRS (1ll << 24)
namespace {
constexpr size_t static_blk_bytes = (size_t)(GS_SORT_STATIC_MAX_PAIRS / RS_TILE) * RS_RADIX * sizeof(uint32_t);
constexpr size_t static_grp_bytes = (size_t)(GS_SORT_STATIC_MAX_PAIRS / RS_TILE / RS_GROUP) * RS_RADIX * sizeof(uint32_t);
constexpr size_t SCRATCH_BYTES = static_blk_bytes + RS_MAX_PASSES * static_grp_bytes + 256;   // 4.5 MiB + 256
constexpr int MAX_DEVICES = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
static void* scratch_ptr[MAX_DEVICES];          // per device: address of that device's copy of `scratch`
static std::mutex scratch_mutex;

static int scratch_for_current_device(void** out) {
    int dev = -1;
    GS_CHECK_HIP(hipGetDevice(&dev));
    GS_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device index %d outside 0..%d", dev, MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (!scratch_ptr[dev]) {
        void* p = nullptr;
        GS_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        GS_REQUIRE(p != nullptr, "hipGetSymbolAddress returned a null scratch address");
        scratch_ptr[dev] = p;
    }
    *out = scratch_ptr[dev];
    return GS_OK;
}

extern "C" long long gs_sort_pairs_static_max(void) { return GS_SORT_STATIC_MAX_PAIRS; }

extern "C" int gs_sort_pairs_static(long long D, const uint32_t* count_dev, int bit_begin, int bit_end,
                                    uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b,
                                    int* result_in_b, void* stream_) {
    if (D > GS_SORT_STATIC_MAX_PAIRS) {
        gs_set_error("gs_sort_pairs_static: %lld pairs exceed the static scratch (at most %lld); use gs_sort_pairs with a "
                     "caller workspace", D, (long long)GS_SORT_STATIC_MAX_PAIRS);
        return GS_ERR_SIZE;
    }
    GS_REQUIRE(D >= 0 && result_in_b, "bad arguments");
    if (gs_sort_workspace_bytes(D) > SCRATCH_BYTES) {      // cannot happen for D <= the maximum; guards a retuned tile size
        gs_set_error("gs_sort_pairs_static: workspace for %lld pairs larger than the static scratch", D);
        return GS_ERR_SIZE;
    }
    void* ws = nullptr;
    const int rc = scratch_for_current_device(&ws);
    if (rc != GS_OK) return rc;
    if (D == 0) {      // the documented set-up call (first call per device, outside any stream capture): nothing to sort
        *result_in_b = 0;
        return GS_OK;
    }
    // the tables are rewritten (blk) or cleared (grp) by every call: nothing of the previous call is read
    return sort_pairs_impl(D, count_dev, bit_begin, bit_end, keys_a, vals_a, keys_b, vals_b, ws, SCRATCH_BYTES,
                           result_in_b, stream_);
}
