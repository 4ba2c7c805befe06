// Per-tile alpha compositing, forward and backward.
//
// Arithmetic restated from reference gaussian_renderer/__init__.py::splat_gaussians:
//   power  = -0.5 (a dx^2 + (b+c) dx dy + d dy^2), weight = exp(power) * opacity      (:417-426)
//   alpha  = weight * (1 - A);  image += alpha * colour;  A += alpha;  depth += alpha * z (:429-440)
// over the Gaussian's hard integer box only (:396-413), far-to-near, image starting at bg (:361),
// no alpha clamp, no 1/255 skip.  The backward is the derivative of those lines (autograd in
// the reference), replayed in forward order:
//   u_i = c_i.g_C + z_i g_Z + g_A ;  dL/dw_i = T_i u_i - (Total - sum_{j<=i} u_j alpha_j) / (1 - w_i)
// with Total = g_C.(C - bg) + g_Z Z + g_A A taken from the forward's outputs.
//
// Structure (MI355X): one 256-lane workgroup per 16x16 tile; each of its 4 wave64s owns one 8x8
// sub-tile and each 16-lane DPP row of a wave one 4x4 pixel block of it (lane = pixel).  The four
// rows of a wave walk FOUR DIFFERENT hit lists in lockstep: per batch of the tile's sorted list the
// workgroup stages the records in LDS, and each wave ballots the boxes against its four 4x4 blocks
// and compacts one list per row.  A 4x4 block wastes far fewer lanes on a ~13 px box than an 8x8
// sub-tile does (63 % of the lane slots are inside the box instead of 43 %), and the loop length of
// a wave is the longest of its four lists: 0.68 of the 8x8 hit count on the benchmark scene.  The
// row's list entry (16-bit record byte offset, one hand-issued ds_read_u16 with four distinct addresses,
// one visit ahead) and the 48-byte record (three ds_read_b128) are per-lane LDS reads; shorter lists are
// padded with a record no lane is inside, so there is no per-visit bounds test, and the loops take two
// visits per trip.  (History: 8x8 per wave with
// the hit index through v_readlane; before that b96/b32/u8 record reads, 26 LDS cycles per
// Gaussian; scalar s_load of the records is slower: the scalar cache cannot stream misses.)
// A lane stops once its transmittance 1 - A drops below t_eps, a wave when all its lanes stopped
// (ballot), a workgroup when all its waves did.  t_eps = 0 reproduces the reference exactly.
//
// Per-visit work is trimmed at staging time (once per (tile, Gaussian) pair instead of once per
// visit and lane): the LDS copy of a record carries the conic pre-scaled for v_exp_f32
// (power * log2 e = A' dx^2 + B' dx dy + C' dy^2) and the box as two packed u16 pairs, so the
// per-lane box test is v_pk_max_u16 + v_pk_min_u16 + one compare, and a stopped lane is folded
// into the same test by setting its packed pixel coordinate to 0xFFFFFFFF.
//
// Backward gradient accumulation: a visit reduces ONE value per lane, pv = G dL/dw, to its six moments in the
// 4x4 block's own pixel coordinates (row_moments_and_totals: 8 DPP instructions) plus the totals of the four
// colour / depth products (8 DPP adds) -- 16 half-rate instructions instead of the 22 a reduce-scatter of the ten
// per-Gaussian sums needs, and the products with dx, dy are never formed.  Ten lanes of the row store one dword
// each into the 40-byte LDS slot of (4x4 block, list position) -- plain stores into a slot pool shared by the tile's 16
// blocks, so that a 64-pair batch fits the LDS of 6 workgroups per CU; after the batch the 16 lanes of a
// DPP row take one pair each, lane b reads block b's slot and moves its moments to the centre of the Gaussian's
// box (formed once per pair at staging), a 22-add reduce-scatter adds the 16 blocks up, and one row-shaped atomic
// per (tile, Gaussian) goes into that Gaussian's 64-byte gradient row (six geometric sums as fp64 atomics:
// gsplat_common.h).  A block holds sums for a pair iff the pair's list position was walked by the block's wave.
//
// Bound: VALU issue (~20 instructions per visit forward, 47 backward); HBM traffic is the record
// gather per (tile, Gaussian) pair plus 28 / 44 bytes per pixel.
#include "gsplat_common.h"

// -DGS_STATS (tools/build_variant.sh stats -DGS_STATS, read with tools/bwd_stats.py): utilisation counters of the backward
// kernel -- batches, list lengths, executed visits, active lanes, live rows, flush rounds.  Not in the product build.
#ifdef GS_STATS
__device__ unsigned long long gs_dbg[16];
extern "C" int gs_dbg_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gs_dbg), sizeof(gs_dbg)); }
extern "C" int gs_dbg_clear() { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(gs_dbg), z, sizeof(z)); }
#define GS_STAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&gs_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define GS_STAT(i, v) do { } while (0)
#endif

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

struct Rec { f4 q0, q1, q2; };

constexpr float GS_LOG2E = 1.4426950408889634f;
constexpr int GS_ROWS = 4;                 // 16-lane rows per wave, one 4x4 pixel block each
constexpr int GS_QUAD = 4;                 // edge of a row's pixel block
#ifndef GS_FWD_BATCH_DEF
#define GS_FWD_BATCH_DEF 192     // pairs per forward batch: 128 / 192 / 256 = 0.290 / 0.279 / 0.277 ms at config 3 with 16-bit lists
                                 // (15.5 KB of LDS at 192; config 5: 3.17 / 3.10 / 3.11 ms, config 2: 0.49 / 0.45 / 0.47 ms)
#endif
#ifndef GS_BWD_BATCH_DEF
#define GS_BWD_BATCH_DEF 64
#endif
#ifndef GS_BWD_POOL
#define GS_BWD_POOL 480     // slots of the backward's pool (0 = BATCH slots per block); measured 400..492: see render_bwd_kernel
#endif
constexpr int GS_BWD_BATCH = GS_BWD_BATCH_DEF;           // pairs per backward batch (list positions are 8-bit; more than 64 pairs is untested)

// LDS form of a splat record (written by stage_and_cull):
//   q0 = (mean_x, mean_y, A', B')   q1 = (C', opacity, x_min | y_min << 16, (x_max - 1) | (y_max - 1) << 16)
//   q2 = (red, green, blue, depth)
// with A' = -0.5 log2e conic_a, B' = -log2e conic_b, C' = -0.5 log2e conic_d.  The box sits in one
// 16-byte word (a visit is tested with q1 alone) and (red, green), (blue, depth) are aligned register
// pairs for the forward's two v_pk_fma_f32.
__device__ __forceinline__ void to_lds_form(f4& p0, f4& p1, f4& p2) {
    const uint32_t rx = __float_as_uint(p1.w), ry = __float_as_uint(p2.w);
    p0.z *= -0.5f * GS_LOG2E;
    p0.w *= -GS_LOG2E;
    p1.x *= -0.5f * GS_LOG2E;
    p2.w = p1.z;                                                         // depth
    p1.z = __uint_as_float((rx & 0xFFFFu) | (ry << 16));
    p1.w = __uint_as_float(((rx >> 16) - 1u) | (((ry >> 16) - 1u) << 16));
}

// per-lane record read: `off` (bytes) is the same for the 16 lanes of a row -> three ds_read_b128
// with at most four distinct addresses
__device__ __forceinline__ Rec load_rec(const f4* sq, uint32_t off) {
    const f4* p = (const f4*)((const char*)sq + off);
    Rec r;
    r.q0 = p[0];
    r.q1 = p[1];
    r.q2 = p[2];
    return r;
}

// log2 e * power = (A' dx + B' dy) dx + (C' dy) dy : mul, fma, mul, mul, fma
__device__ __forceinline__ float gauss_power2(const Rec& rec, float dx, float dy) {
    const float t = __builtin_fmaf(rec.q0.w, dy, rec.q0.z * dx);
    return __builtin_fmaf(rec.q1.x * dy, dy, t * dx);
}

// packed pixel inside the packed inclusive box?  (2 x v_pk_{max,min}_u16 + v_cmp)
__device__ __forceinline__ bool in_box(uint32_t P, uint32_t lo, uint32_t hi) {
    const us2 c = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(us2, P),
                                                                      __builtin_bit_cast(us2, lo)),
                                            __builtin_bit_cast(us2, hi));
    return __builtin_bit_cast(uint32_t, c) == P;
}

// list entries are record byte offsets (48 j <= 48 x 256) and list positions are < 64: 16 and 8 bits, so that the lists and
// the position maps leave the backward's LDS to its slot pool
typedef uint16_t wl_t;
typedef uint8_t pos_t;
constexpr pos_t POS_NONE = 0xFF;
constexpr uint32_t P_STOPPED = 0xFFFFFFFFu;   // never inside a box: x_max - 1 <= 65534
// Box of the padding record that fills the shorter row lists: x test passes only for x = 0xFFFF
// (stopped lanes), y test only for y = 0 (never a stopped lane) -> no lane is ever inside it.
constexpr uint32_t PAD_LO = 0xFFFFFFFFu, PAD_HI = 0x0000FFFFu;

struct TileCtx {
    int px, py;        // this lane's pixel
    int sx0, sy0;      // this wave's 8x8 sub-tile origin
    bool inside;
};

__device__ __forceinline__ int tile_of_block(int bid, int ntiles) {
    // XCD-aware order: consecutive workgroup ids are dealt round-robin to the 8 XCDs; give every
    // XCD one contiguous range of tiles so neighbouring tiles (shared Gaussians) share an L2.
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// wave -> 8x8 sub-tile of the 16x16 tile; 16-lane row -> 4x4 quadrant of it; lane -> pixel
__device__ __forceinline__ TileCtx make_ctx(int tile, int tiles_x, int H, int W) {
    TileCtx c;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 4, q = lane & 15;
    c.sx0 = tx * GS_TILE + (wave & 1) * GS_SUB;
    c.sy0 = ty * GS_TILE + (wave >> 1) * GS_SUB;
    c.px = c.sx0 + (row & 1) * GS_QUAD + (q & 3);
    c.py = c.sy0 + (row >> 1) * GS_QUAD + (q >> 2);
    c.inside = c.px < W && c.py < H;
    return c;
}

// Stage the records of `n` pairs of the tile's sorted list into LDS (sq, 3 x 16 B per pair), then
// build, per wave, FOUR compacted hit lists -- one per 16-lane row, i.e. per 4x4 pixel block -- of
// the byte offsets (48 j) of the records whose box overlaps that block, in list order.  Lists
// shorter than the longest are padded with the offset of a record no lane is ever inside
// (sq[BATCH]), so the per-visit loop needs no "is my list exhausted" test.  wl = this wave's lists,
// GS_ROWS x (BATCH + 1) entries (one spare for the loop's prefetch); posmap (optional, GS_ROWS x BATCH, u8) = position of pair j in each list,
// POS_NONE when it is not in it.  dead_rows: bit q set = every pixel of row q's block has stopped (or lies outside the image):
// its list stays empty, so it neither lengthens the wave's walk nor costs skipped visits.  cnt_out (optional) = the four
// list lengths; load = false: the records are staged already (a batch culled again with a smaller n).  Returns the length
// of the longest list.
template <int BATCH>
__device__ __forceinline__ int stage_and_cull(const f4* __restrict__ splat, const uint32_t* __restrict__ point_list,
                                              uint32_t base, int n, const TileCtx& c, f4* sq, uint2* sid,
                                              wl_t* wl, pos_t* posmap, uint32_t dead_rows,
                                              int* cnt_out = nullptr, bool load = true) {
    static_assert(BATCH % 16 == 0 && BATCH >= 16 && BATCH <= GS_BLOCK, "one staging lane per pair");
    const int tid = threadIdx.x;
    if (load && tid < n && tid < BATCH) {
        const uint32_t id = point_list[base + tid];
        const f4* p = splat + 3 * (size_t)id;
        f4 p0 = p[0], p1 = p[1], p2 = p[2];
        if (sid) {
            // the backward's flush wants the Gaussian's index and the centre of its box (the integer pixel its moments are
            // taken about): formed once per pair here instead of once per lane and flush round
            const uint32_t rx = __float_as_uint(p1.w), ry = __float_as_uint(p2.w);
            const uint32_t cx = ((rx & 0xFFFFu) + ((rx >> 16) - 1u)) >> 1, cy = ((ry & 0xFFFFu) + ((ry >> 16) - 1u)) >> 1;
            sid[tid] = make_uint2(id << 2, cx | (cy << 16));     // 4 x index: the flush turns it into a row address with one shift-add
        }
        to_lds_form(p0, p1, p2);
        sq[3 * tid] = p0;
        sq[3 * tid + 1] = p1;
        sq[3 * tid + 2] = p2;
    }
    __syncthreads();
    const int lane = tid & 63;
    const uint32_t* squ = (const uint32_t*)sq;
    int cnt[GS_ROWS] = {0, 0, 0, 0};
    if (BATCH <= 32) {
        // A batch of at most 32 pairs would leave half the wave idle in the loop below: here lane (j, h) tests pair j =
        // lane & 31 against the block rows h = lane >> 5 selects (h = 0: the sub-tile's upper two 4x4 blocks, h = 1: the
        // lower two), so one ballot carries two lists -- bits 0..31 the upper block's, bits 32..63 the lower block's --
        // and the four lists take two ballots, two list stores and two position stores instead of four of each.
        const int j = lane & 31, h = lane >> 5;
        bool hx0 = false, hx1 = false, hy = false;
        if (j < n) {
            const uint32_t lo = squ[12 * j + 6], hi = squ[12 * j + 7];
            const int xmin = lo & 0xFFFF, ymin = lo >> 16, xlast = hi & 0xFFFF, ylast = hi >> 16;
            const int y0 = c.sy0 + h * GS_QUAD;
            hx0 = xmin < c.sx0 + GS_QUAD && xlast >= c.sx0;
            hx1 = xmin < c.sx0 + 2 * GS_QUAD && xlast >= c.sx0 + GS_QUAD;
            hy = ymin < y0 + GS_QUAD && ylast >= y0;
        }
        const uint32_t lt32 = (1u << j) - 1u;
#pragma unroll
        for (int xq = 0; xq < 2; ++xq) {
            const int q = 2 * h + xq;                              // this lane's row: make_ctx's row = 2 (y half) + (x half)
            const bool hit = (xq ? hx1 : hx0) && hy && !((dead_rows >> q) & 1u);
            const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
            const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
            const int pos = __popc((h ? mhi : mlo) & lt32);
            if (hit) wl[q * (BATCH + 1) + pos] = (wl_t)(48u * (uint32_t)j);
            if (posmap && j < BATCH) posmap[q * BATCH + j] = hit ? (pos_t)pos : POS_NONE;
            cnt[xq] = __popc(mlo);
            cnt[2 + xq] = __popc(mhi);
        }
    } else {
#pragma unroll
    for (int r = 0; r < (BATCH + 63) / 64; ++r) {
        const int j = r * 64 + lane;
        bool hx0 = false, hx1 = false, hy0 = false, hy1 = false;
        if (j < n) {
            const uint32_t lo = squ[12 * j + 6], hi = squ[12 * j + 7];
            const int xmin = lo & 0xFFFF, ymin = lo >> 16, xlast = hi & 0xFFFF, ylast = hi >> 16;
            hx0 = xmin < c.sx0 + GS_QUAD && xlast >= c.sx0;
            hx1 = xmin < c.sx0 + 2 * GS_QUAD && xlast >= c.sx0 + GS_QUAD;
            hy0 = ymin < c.sy0 + GS_QUAD && ylast >= c.sy0;
            hy1 = ymin < c.sy0 + 2 * GS_QUAD && ylast >= c.sy0 + GS_QUAD;
        }
        const bool hit[GS_ROWS] = {hx0 && hy0 && !(dead_rows & 1u), hx1 && hy0 && !(dead_rows & 2u),
                                   hx0 && hy1 && !(dead_rows & 4u), hx1 && hy1 && !(dead_rows & 8u)};
#pragma unroll
        for (int q = 0; q < GS_ROWS; ++q) {
            const uint64_t m = __builtin_amdgcn_ballot_w64(hit[q]);     // (the lane mask itself; __ballot goes through a select and a compare)
            const int pos = cnt[q] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (hit[q]) wl[q * (BATCH + 1) + pos] = (wl_t)(48u * (uint32_t)j);
            if (posmap && j < BATCH) posmap[q * BATCH + j] = hit[q] ? (pos_t)pos : POS_NONE;
            cnt[q] += __popcll(m);
        }
    }
    }
    const int nmax = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
    if (cnt_out) {
#pragma unroll
        for (int q = 0; q < GS_ROWS; ++q) cnt_out[q] = cnt[q];
    }
#pragma unroll
    for (int q = 0; q < GS_ROWS; ++q)
        for (int p = cnt[q] + lane; p < ((nmax + 1) & ~1); p += 64) wl[q * (BATCH + 1) + p] = (wl_t)(48u * (uint32_t)BATCH);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return __builtin_amdgcn_readfirstlane(nmax);      // wave-uniform: keep the visit loop's bound in an SGPR
}

// List entries (u16 record offsets) are read with hand-issued ds_read_u16: through the compiler the loop-carried entry
// stays a 16-bit value and every visit pays a v_and_b32 0xffff before it can address the record, although the read has
// zero-extended it already, and the second visit of an unrolled trip pays an address add that the instruction's
// immediate offset covers.  The result of an asm is opaque to the compiler's wait-count bookkeeping, so the wait is
// written by hand too (list_wait, right before the first use); extra LDS reads in flight only make the compiler's own
// waits conservative (LDS returns in order).
template <int BYTE_OFF>
__device__ __forceinline__ uint32_t list_read(uint32_t lds_addr) {
    uint32_t v;
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(BYTE_OFF) : "memory");
    return v;
}
__device__ __forceinline__ void list_wait(uint32_t& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); }
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)p;
}
template <int V> struct IntC { static constexpr int value = V; };

// Barrier + "does pred hold in every lane of the workgroup?" (GS_BLOCK = 4 waves, all lanes active).  The library form,
// __syncthreads_and, is a DPP reduction per wave, an LDS exchange and TWO barriers; here a wave's answer is one ballot,
// lane 0 leaves it in flags[wave], and the one barrier the batch loop needs anyway publishes it.  The next call's
// writes come after every wave has passed a later barrier (stage_and_cull's), so the reads here cannot see them.
__device__ __forceinline__ bool block_all(bool pred, uint32_t* flags) {
    static_assert(GS_BLOCK == 256, "four wave flags read as one 16-byte word");
    const uint32_t wave_ok = __builtin_amdgcn_ballot_w64(!pred) == 0ull ? 1u : 0u;
    if ((threadIdx.x & 63) == 0) flags[threadIdx.x >> 6] = wave_ok;
    __syncthreads();
    const uint4 f = *reinterpret_cast<const uint4*>(flags);
    return (f.x & f.y & f.z & f.w) != 0u;
}

// bit q: all 16 lanes of DPP row q of this wave have stopped (wave-uniform)
__device__ __forceinline__ uint32_t dead_row_mask(uint32_t P) {
    const uint64_t st = __builtin_amdgcn_ballot_w64(P == P_STOPPED);
    uint32_t d = 0;
#pragma unroll
    for (int q = 0; q < GS_ROWS; ++q) d |= (((st >> (16 * q)) & 0xFFFFull) == 0xFFFFull) ? (1u << q) : 0u;
    return d;
}

__device__ __forceinline__ void write_pad_record(f4* sq, int BATCH) {
    if (threadIdx.x == 0) {
        const f4 z = {0.f, 0.f, 0.f, 0.f};
        sq[3 * BATCH] = z;
        sq[3 * BATCH + 1] = f4{0.f, 0.f, __uint_as_float(PAD_LO), __uint_as_float(PAD_HI)};
        sq[3 * BATCH + 2] = z;
    }
}

template <bool AUX>
__global__ __launch_bounds__(GS_BLOCK) void render_fwd_kernel(
    int H, int W, int tiles_x, int tile0, int ntiles, const float* __restrict__ bg, const float4* __restrict__ splat,
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, float* __restrict__ out_img,
    float* __restrict__ out_depth, float* __restrict__ out_alpha, float* __restrict__ aux_rgb, float t_eps,
    f4* __restrict__ zero_buf, size_t zero_n) {
    constexpr int BATCH = GS_FWD_BATCH_DEF;
    constexpr int NW = GS_BLOCK / 64;
    if (zero_buf) {
        // This kernel is VALU-bound and leaves HBM nearly idle: clearing the backward's gradient-row
        // buffer here (fire-and-forget 16-byte stores, one slice per workgroup) is free, a memset
        // launched on its own costs ~24 us per frame at 3 M Gaussians.
        const size_t chunk = (zero_n + gridDim.x - 1) / gridDim.x;
        const size_t z0 = (size_t)blockIdx.x * chunk;
        const f4 z = {0.f, 0.f, 0.f, 0.f};
        for (size_t j = threadIdx.x; j < chunk && z0 + j < zero_n; j += GS_BLOCK) zero_buf[z0 + j] = z;
    }
    __shared__ f4 sq[3 * (BATCH + 1)];
    __shared__ wl_t wl_all[NW][GS_ROWS * (BATCH + 1)];
    __shared__ __attribute__((aligned(16))) uint32_t wave_flags[NW];
    const int tile = tile0 + tile_of_block(blockIdx.x, ntiles);
    const TileCtx c = make_ctx(tile, tiles_x, H, W);
    const int lane = threadIdx.x & 63;
    wl_t* wl = wl_all[threadIdx.x >> 6];
    const wl_t* mylist = wl + (lane >> 4) * (BATCH + 1);
    const uint2 range = ranges[tile];
    const float fpx = (float)c.px, fpy = (float)c.py;
    float A = 0.f, T = 1.f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Z = 0.f;
    uint32_t P = c.inside ? ((uint32_t)c.px | (uint32_t)c.py << 16) : P_STOPPED;
    write_pad_record(sq, BATCH);
    for (uint32_t base = range.x; base < range.y; base += BATCH) {
        if (block_all(P == P_STOPPED, wave_flags)) break;
        const int n = min((int)(range.y - base), BATCH);
        const int nmax = stage_and_cull<BATCH>((const f4*)splat, point_list, base, n, c, sq, nullptr, wl, nullptr,
                                               dead_row_mask(P));
        // list entry one visit ahead (one LDS latency per visit, not two); two visits per trip: an odd nmax ends on the
        // padding record behind the lists (stage_and_cull pads to an even length), whose alpha is 0
        uint32_t lp = lds_addr_of(mylist);
        uint32_t off_next = list_read<0>(lp);
        auto visit = [&](auto next_off) {
            list_wait(off_next);
            const Rec rec = load_rec(sq, off_next);
            off_next = list_read<decltype(next_off)::value>(lp);       // entry nmax + 1 is never used
            const bool act = in_box(P, __float_as_uint(rec.q1.z), __float_as_uint(rec.q1.w));
            const float dx = fpx - rec.q0.x, dy = fpy - rec.q0.y;
            const float power2 = gauss_power2(rec, dx, dy);
            const float w = __builtin_amdgcn_exp2f(power2) * rec.q1.y;
            const float alpha = act ? w * T : 0.0f;
            Cr += alpha * rec.q2.x;
            Cg += alpha * rec.q2.y;
            Cb += alpha * rec.q2.z;
            Z += alpha * rec.q2.w;
            A += alpha;
            T = 1.0f - A;
            P = T <= t_eps ? P_STOPPED : P;   // t_eps = 0: stops at T == 0 exactly, where every later alpha is 0 anyway
        };
        for (int kk = 0; kk < nmax; kk += 2, lp += 2 * sizeof(wl_t)) {
            if ((kk & 7) == 0 && __all(P == P_STOPPED)) break;
            visit(IntC<sizeof(wl_t)>());
            visit(IntC<2 * sizeof(wl_t)>());
        }
    }
    if (c.inside) {
        const size_t HW = (size_t)H * W, p = (size_t)c.py * W + c.px;
        out_img[p] = bg[0] + Cr;
        out_img[HW + p] = bg[1] + Cg;
        out_img[2 * HW + p] = bg[2] + Cb;
        out_depth[p] = Z;
        out_alpha[p] = A;
        if (AUX) {
            aux_rgb[p] = Cr;
            aux_rgb[HW + p] = Cg;
            aux_rgb[2 * HW + p] = Cb;
        }
    }
}

// Reduce-scatter of 10 values over each 16-lane row (registers only, 22 DPP adds + 2 selects):
// every step adds the partner lane's copy AND halves the number of values a lane still carries.
//   step 1  lane^8  row_ror:8, bank_mask picks v[j] (banks 0,1) or v[j+5] (banks 2,3)   10 -> 5
//   step 2  lane^4  row_shl:4 / row_shr:4 per bank                                        5 -> 3
//   step 3  quad sum of the 3 remaining registers (quad_perm xor 1, xor 2)
// Result: lane (bank b = lane bits 3:2, t = lane & 3) of every row holds the row total of
//   t = 0: v[(b&1 ? 2 : 0) + 5 (b>>1)]   t = 1: v[(b&1 ? 3 : 1) + 5 (b>>1)]   t = 2: v[4 + 5 (b>>1)]
// (tools/dpp_test/t3.hip is the standalone check).  s_nop: a DPP operand needs 2 wait states after
// the VALU write of that register.
template <bool HAS9 = true>
__device__ __forceinline__ float row_reduce_scatter10(const float (&v)[10], int t) {
    float o0, o1, o2, o3, o4, p0, p1, p2;
    if (!HAS9) {
        // v[9] is not needed: one add fewer; the lanes that would end up with its total (banks 2,3, t = 2) hold
        // leftovers and must not be stored
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %0, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %1, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %2, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %3, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %5, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %6, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %5, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %6, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %7, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x1\n\t"
            "v_add_f32_dpp %7, %4, %4 row_shr:4 row_mask:0xf bank_mask:0x2\n\t"
            "s_nop 0\n\t"
            "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0x3\n\t"
            "s_nop 1"
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(p0), "=&v"(p1), "=&v"(p2)
            : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
              "v"(v[9]));
        return t == 0 ? p0 : (t == 1 ? p1 : p2);
    }
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %6, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %7, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(p0), "=&v"(p1), "=&v"(p2)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
          "v"(v[9]));
    return t == 0 ? p0 : (t == 1 ? p1 : p2);
}

// Per-visit reduction of a 4x4 block (one 16-lane DPP row, lane = 4 ly + lx) in LOCAL pixel coordinates:
// from ONE value per lane, pv = G dL/dw, the six moments  M_ab = sum pv lx^a ly^b  (a + b <= 2), and beside them
// the plain totals of up to four colour / depth products.  The moments about the Gaussian's mean that the gradient
// needs follow per (block, Gaussian) at flush time from the block origin (dx = (bx - mean_x) + lx, q = opacity pv),
// and M_00 is d/d opacity itself -- so a visit reduces 1 + 4 values instead of 10 products:
//   level 1  x pairs      t  = pv[lx|1] + pv                               (odd lanes: 2 pv)          1 DPP add
//   level 2  quad         u  = (M0, M1, M2) over lx in lanes lx = 0, 1, 2: constant 3x4 map          3 DPP mul/fmac
//   level 3  y pairs      u += u[ly + 1] in even banks (odd banks keep their own)                     1 DPP add
//   level 4  row          r  = (N0, N1, N2) over ly in banks 0, 1, 2: the same 3x4 map on banks      1 mul + 3 DPP fmac
// Lane (bank B, lx X) of r holds M_XB: (0,0) M00, (1,0) M10, (2,0) M20, (0,1) M01, (1,1) M11, (0,2) M02.
// Colours: two select-free halvings over the banks, then the quad sum -- every lane of bank j ends with the row
// total of c_j (8 DPP adds, 7 without c3).  16 half-rate instructions per visit instead of 22, and the five
// products q dx, q dy, q dx^2, q dx dy, q dy^2 are never formed.  k[] = this lane's seven map coefficients.
struct MomentCoef { float a, b, c, d0, d2; };
__device__ __forceinline__ MomentCoef moment_coef(int lane) {
    const int x = lane & 3, y = (lane >> 2) & 3;
    MomentCoef k;
    k.a = x == 0 ? 1.0f : (x == 3 ? 0.0f : 0.5f);            // t[0,1,1,-]
    k.b = x == 0 ? 1.0f : (x == 1 ? 2.0f : (x == 2 ? 4.0f : 0.0f));   // t[2]
    k.c = x == 1 ? 0.5f : (x == 2 ? 2.5f : 0.0f);             // t[3]
    k.d0 = y == 2 ? 4.0f : (y == 3 ? 0.0f : 1.0f);            // own bank
    k.d2 = y == 1 ? 2.0f : 5.0f;                              // bank + 1 (written in banks 1, 2 only); banks + 2 and - 1 enter with 1: plain DPP adds
    return k;
}

template <bool HAS3>
__device__ __forceinline__ void row_moments_and_totals(float pv, float c0, float c1, float c2, float c3,
                                                       const MomentCoef& k, float& r, float& f) {
    float t, u, e0, e1;
    if (HAS3) {
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %6, %6 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %2, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %2, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %3, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_mul_f32_dpp %1, %0, %11 quad_perm:[0,1,1,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %1, %0, %12 quad_perm:[2,2,2,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %1, %0, %13 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %5, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %5, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_mul_f32 %4, %14, %1\n\t"
            "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %4, %1, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_fmac_f32_dpp %4, %1, %15 row_shl:4 row_mask:0xf bank_mask:0x6\n\t"
            "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %4, %1, %4 row_shr:4 row_mask:0xf bank_mask:0x4\n\t"
            "s_nop 1"
            : "=&v"(t), "=&v"(u), "=&v"(e0), "=&v"(e1), "=&v"(r), "=&v"(f)
            : "v"(pv), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(k.a), "v"(k.b), "v"(k.c), "v"(k.d0), "v"(k.d2));
    } else {
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %6, %6 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %2, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %2, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_mul_f32_dpp %1, %0, %11 quad_perm:[0,1,1,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %1, %0, %12 quad_perm:[2,2,2,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %1, %0, %13 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %5, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %5, %3, %3 row_shr:4 row_mask:0xf bank_mask:0x2\n\t"
            "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_mul_f32 %4, %14, %1\n\t"
            "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0x7\n\t"
            "v_add_f32_dpp %4, %1, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_fmac_f32_dpp %4, %1, %15 row_shl:4 row_mask:0xf bank_mask:0x6\n\t"
            "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0x7\n\t"
            "v_add_f32_dpp %4, %1, %4 row_shr:4 row_mask:0xf bank_mask:0x4\n\t"
            "s_nop 1"
            : "=&v"(t), "=&v"(u), "=&v"(e0), "=&v"(e1), "=&v"(r), "=&v"(f)
            : "v"(pv), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(k.a), "v"(k.b), "v"(k.c), "v"(k.d0), "v"(k.d2));
    }
}

// HAS_ZA: an upstream gradient of the depth or alpha output exists (otherwise u has three terms and the depth sum is skipped)
// POOL > 0: the (block, list position) slots of a batch come out of one pool of POOL slots, handed out per row after the
// cull (row bases = exclusive scan of the 16 rows' needs: the longest list of the row's wave), instead of BATCH slots per block: the per-block arrays are
// three quarters empty on the benchmark scene, and they are what sets the kernel's LDS and with it the batch size.  A batch
// whose lists need more than the pool is culled again with half the pairs (49 of 70 k tile-batches on the benchmark scene
// at 480 slots; 8.6 % of them at 416, which is why the lists are 16-bit and the position maps 8-bit).  With the pool,
// 64-pair batches fit the LDS that 32-pair batches with per-block arrays took (26.3 KB, 6 workgroups per CU): half as many
// batches to stage, cull and synchronise, and four lists twice as long are walked in lockstep with less waste (live rows
// 0.76 -> 0.79): 0.783 -> 0.713 ms.  Pool sizes 400 / 416 / 448 / 480 / 492: 0.729 / 0.724 / 0.712 / 0.714 / 0.714 ms;
// 496 slots cross the LDS step to 5 workgroups per CU: 0.748 ms.
template <int BATCH, bool HAS_ZA, int POOL = 0>
__global__ __launch_bounds__(GS_BLOCK) void render_bwd_kernel(
    int H, int W, int tiles_x, int tile0, int ntiles, const float4* __restrict__ splat, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float* __restrict__ aux_rgb,
    const float* __restrict__ out_depth, const float* __restrict__ out_alpha, const float* __restrict__ g_img,
    const float* __restrict__ g_depth, const float* __restrict__ g_alpha, float* __restrict__ grad_rows,
    float t_eps) {
    constexpr int NW = GS_BLOCK / 64;
    constexpr int NBLK = NW * GS_ROWS;          // 16 4x4 blocks per tile
    constexpr int SLOT = 10;                    // floats per (block, list position) slot
    __shared__ f4 sq[3 * (BATCH + 1)];
    __shared__ uint2 sid[BATCH];                       // (4 x Gaussian index, centre of its box as x | y << 16)
    __shared__ wl_t wl_all[NW][GS_ROWS * (BATCH + 1)];
    __shared__ pos_t posmap_all[NW][GS_ROWS * BATCH];
#ifndef GS_SLOT_PAD
#define GS_SLOT_PAD 4      // measured: 0.8135 ms -> 0.7975 ms (rows of a wave start in different banks)
#endif
    constexpr int BSTR = BATCH * SLOT + GS_SLOT_PAD;   // floats between the slot arrays of two blocks (POOL == 0)
    constexpr int NSLOTF = POOL > 0 ? POOL * SLOT : NBLK * BSTR;
    static_assert(POOL == 0 || POOL >= NBLK, "one pair can touch all 16 blocks");
    __shared__ float slots[NSLOTF + SLOT];             // plain stores, one 40-byte slot per row visit (+ a zero slot)
    __shared__ uint32_t row_cnt[NBLK];                 // POOL: list length of every 4x4 block, this batch
    __shared__ uint32_t row_base[POOL > 0 ? NW : 1][NBLK];   // POOL: first slot of every block (a private copy per wave)
    __shared__ __attribute__((aligned(16))) uint32_t wave_flags[NW];
    __shared__ uint32_t walked_all[NW];              // list positions the wave walked this batch (all of them unless every pixel stopped)
    const int tile = tile0 + tile_of_block(blockIdx.x, ntiles);
    const TileCtx c = make_ctx(tile, tiles_x, H, W);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    wl_t* wl = wl_all[wave];
    const wl_t* mylist = wl + (lane >> 4) * (BATCH + 1);
    const uint2 range = ranges[tile];
    const float fpx = (float)c.px, fpy = (float)c.py;
    float gr = 0.f, gg = 0.f, gb = 0.f, gz = 0.f, ga = 0.f, total = 0.f;
    if (c.inside) {
        const size_t HW = (size_t)H * W, p = (size_t)c.py * W + c.px;
        gr = g_img[p];
        gg = g_img[HW + p];
        gb = g_img[2 * HW + p];
        if (g_depth) gz = g_depth[p];
        if (g_alpha) ga = g_alpha[p];
        total = (gr * aux_rgb[p] + gg * aux_rgb[HW + p] + gb * aux_rgb[2 * HW + p]) + gz * out_depth[p] +
                ga * out_alpha[p];
    }
    float A = 0.f, T = 1.f, rem = total;
    const f2 g_rg = {gr, gg}, g_bz = {gb, gz};
    uint32_t P = c.inside ? ((uint32_t)c.px | (uint32_t)c.py << 16) : P_STOPPED;
    // which of the 10 values this lane holds after row_reduce_scatter10 (-1: a duplicate)
    const int t = lane & 3, bank = (lane >> 2) & 3;
    int red_slot = -1;
    if (t == 0) red_slot = ((bank & 1) ? 2 : 0) + 5 * (bank >> 1);
    else if (t == 1) red_slot = ((bank & 1) ? 3 : 1) + 5 * (bank >> 1);
    else if (t == 2 && !(bank & 1) && (HAS_ZA || bank == 0)) red_slot = 4 + 5 * (bank >> 1);
    // gradient row (GS_GRAD_ROW = 16 floats = 64 bytes, see gsplat_common.h): sums 0..5 are doubles, 6..9 floats 12..15
    static_assert(GS_GRAD_ROW == 16, "row layout");
    char* const lane_row0 = reinterpret_cast<char*>(grad_rows) + (red_slot < 0 ? 0 : (red_slot < 6 ? 8 * red_slot : 4 * (6 + red_slot)));
    static_assert(BATCH <= 64 && BATCH % 16 == 0, "tested up to 64 pairs; the flush takes 16 pairs per round");
    // what this lane stores per visit (row_moments_and_totals): slot = [M00 M10 M20 M01 M11 M02 | c0 c1 c2 c3]
    int vis_slot = -1;
    if (t == 3) vis_slot = (bank < 3 || HAS_ZA) ? 6 + bank : -1;
    else if (bank == 0) vis_slot = t;
    else if (bank == 1 && t < 2) vis_slot = 3 + t;
    else if (bank == 2 && t == 0) vis_slot = 5;
    const MomentCoef mk = moment_coef(lane);
    // this lane's float inside slot 0 of its row's block
    float* my_slot = slots + (size_t)(wave * GS_ROWS + (lane >> 4)) * BSTR + (vis_slot < 0 ? 0 : vis_slot);
    write_pad_record(sq, BATCH);
    if (tid < SLOT) slots[NSLOTF + tid] = 0.0f;
    uint32_t flush_base = 0;                            // POOL: first slot of the block this lane flushes (tid & 15)
    int n = 0;
    for (uint32_t base = range.x; base < range.y; base += n) {
        if (block_all(P == P_STOPPED, wave_flags)) break;
        n = min((int)(range.y - base), BATCH);
        int nmax;
        if (POOL > 0) {
            const uint32_t dead = dead_row_mask(P);
            for (bool first = true;; first = false) {
#ifdef GS_STATS
                __shared__ int stat_cnt[NBLK];
                int cq[GS_ROWS];
                nmax = stage_and_cull<BATCH>((const f4*)splat, point_list, base, n, c, sq, sid, wl, posmap_all[wave], dead,
                                             cq, first);
                if (lane < GS_ROWS) stat_cnt[wave * GS_ROWS + lane] = cq[lane];
#else
                nmax = stage_and_cull<BATCH>((const f4*)splat, point_list, base, n, c, sq, sid, wl, posmap_all[wave], dead,
                                             nullptr, first);
#endif
                // every row of a wave gets the wave's longest list: a wave walks nmax positions with all four rows, and a
                // row whose own list is shorter still stores (zeros) at the positions behind it
                if (lane < GS_ROWS) row_cnt[wave * GS_ROWS + lane] = (uint32_t)nmax;
                __syncthreads();
                // inclusive scan of the 16 list lengths, in every 16-lane row alike (DPP row shifts: no LDS round trips)
                const uint32_t own = row_cnt[lane & 15];
                uint32_t incl = own;
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
                if (total <= (uint32_t)POOL || n == 1) {
#ifdef GS_STATS
                    if (tid == 0) {       // what a sorted block -> wave assignment would walk
                        int cs[NBLK], live = 0, geo = 0, srt = 0;
                        for (int i = 0; i < NBLK; ++i) { cs[i] = stat_cnt[i]; live += cs[i]; }
                        for (int w = 0; w < NW; ++w) geo += 4 * max(max(cs[4 * w], cs[4 * w + 1]), max(cs[4 * w + 2], cs[4 * w + 3]));
                        {   // blocks with a list packed, in index order, into as few waves as they need (no sorting by length)
                            int pk = 0, g = 0, k = 0;
                            for (int i = 0; i < NBLK; ++i) if (cs[i] > 0) { g = max(g, cs[i]); if (++k == 4) { pk += 4 * g; g = 0; k = 0; } }
                            pk += 4 * g;
                            atomicAdd(&gs_dbg[14], (unsigned long long)pk);
                        }
                        for (int i = 1; i < NBLK; ++i) { const int v = cs[i]; int j = i; while (j > 0 && cs[j - 1] < v) { cs[j] = cs[j - 1]; --j; } cs[j] = v; }
                        for (int w = 0; w < NW; ++w) srt += 4 * cs[4 * w];
                        atomicAdd(&gs_dbg[11], (unsigned long long)live);
                        atomicAdd(&gs_dbg[12], (unsigned long long)geo);
                        atomicAdd(&gs_dbg[13], (unsigned long long)srt);
                        int nz = 0; for (int w = 0; w < NW; ++w) nz += cs[4 * w] > 0;
                        atomicAdd(&gs_dbg[15], (unsigned long long)nz);
                    }
#endif
                    flush_base = incl - own;                                                // block tid & 15 = lane & 15
                    if (lane < NBLK) row_base[wave][lane] = flush_base;                     // own row's base: via LDS, this wave only
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    my_slot = slots + (size_t)row_base[wave][wave * GS_ROWS + (lane >> 4)] * SLOT + (vis_slot < 0 ? 0 : vis_slot);
                    break;
                }
                GS_STAT(10, 1);
                n = n > 1 ? n / 2 : 1;
                __syncthreads();        // everybody has read row_cnt and the lists before they are rebuilt
            }
        } else {
            nmax = stage_and_cull<BATCH>((const f4*)splat, point_list, base, n, c, sq, sid, wl, posmap_all[wave],
                                         dead_row_mask(P));
        }
        uint32_t lp = lds_addr_of(mylist);
        uint32_t off_next = list_read<0>(lp);
        GS_STAT(0, 1); GS_STAT(1, nmax); GS_STAT(8, n);
        // two visits per trip (list pointer and slot pointer advance once, the second visit uses immediate offsets); an odd
        // nmax ends on the padding record behind the lists (stage_and_cull pads to an even length), which is skipped
        auto visit = [&](const int kk, auto next_off) {
            GS_STAT(2, 1);
            list_wait(off_next);
            const Rec rec = load_rec(sq, off_next);
            off_next = list_read<decltype(next_off)::value>(lp);
            const bool act = in_box(P, __float_as_uint(rec.q1.z), __float_as_uint(rec.q1.w));
            if (!__any(act)) {
                // no lane of the wave is inside this box: the slot still gets its zeros (the flush takes every walked
                // position), except behind the lists -- an odd nmax ends on the padding record, which owns no slot
                if (kk < nmax && vis_slot >= 0) my_slot[kk * SLOT] = 0.0f;
                return;
            }
#ifdef GS_STATS
            {
                const uint64_t am = __ballot(act);
                GS_STAT(3, 1); GS_STAT(4, __popcll(am));
                GS_STAT(5, ((am & 0xFFFFull) != 0) + (((am >> 16) & 0xFFFFull) != 0) + (((am >> 32) & 0xFFFFull) != 0) + ((am >> 48) != 0));
            }
#endif
            const float dx = fpx - rec.q0.x, dy = fpy - rec.q0.y;
            const float power2 = gauss_power2(rec, dx, dy);
            const float G = act ? __builtin_amdgcn_exp2f(power2) : 0.0f;  // inactive lanes: every sum gets 0
            const float w = G * rec.q1.y;
            const float alpha = w * T;
            const float u = HAS_ZA ? __builtin_fmaf(rec.q2.x, g_rg.x, __builtin_fmaf(rec.q2.y, g_rg.y,
                                         __builtin_fmaf(rec.q2.z, g_bz.x, __builtin_fmaf(rec.q2.w, g_bz.y, ga))))
                                   : __builtin_fmaf(rec.q2.x, g_rg.x, __builtin_fmaf(rec.q2.y, g_rg.y, rec.q2.z * g_bz.x));
            A += alpha;
            const float Tn = 1.0f - A;
            const bool stop = Tn <= t_eps;
            rem = __builtin_fmaf(-u, alpha, rem);      // total - sum_{j<=i} u_j alpha_j
            // Behind a pixel that stops here nothing is composited (every later alpha is 0, exactly so for
            // t_eps = 0), so the tail term is 0 by definition -- not the rounding residue of total - prefix
            // (~1e-7 |total|), which would otherwise leak into the sums of every hidden Gaussian.  A pixel that
            // goes on has 1 - w > 0, so the reciprocal needs no clamp.
            const float inv = stop ? 0.0f : __builtin_amdgcn_rcpf(1.0f - w);
            const float dLdw = T * u - rem * inv;
            const float pv = G * dLdw;         // dL/dpower = opacity * pv: the opacity is applied at flush time
            const f2 c_rg = alpha * g_rg;
            const float c_b = alpha * g_bz.x, c_z = HAS_ZA ? alpha * g_bz.y : 0.0f;
            T = Tn;
            P = stop ? P_STOPPED : P;
            float mom, tot;
            row_moments_and_totals<HAS_ZA>(pv, c_rg.x, c_rg.y, c_b, c_z, mk, mom, tot);
            if (vis_slot >= 0) my_slot[kk * SLOT] = t == 3 ? tot : mom;   // one ds_write_b32 per visit, 10 lanes per row
        };
        int kk = 0;
        for (; kk < nmax; kk += 2, lp += 2 * sizeof(wl_t)) {
            if ((kk & 7) == 0 && __all(P == P_STOPPED)) break;
            visit(kk, IntC<sizeof(wl_t)>());
            visit(kk + 1, IntC<2 * sizeof(wl_t)>());
        }
        if (lane == 0) walked_all[wave] = (uint32_t)min(kk, nmax);
        __syncthreads();
        // flush: the 16 lanes of a DPP row take one pair, lane b the sums of 4x4 block b (moved from the block's
        // origin to the centre of the Gaussian's box); the 22-add row reduce-scatter adds the 16 blocks up and ten
        // lanes add one value each to the Gaussian's row: sum pv {dx, dy, dx^2, dx dy, dy^2, 1} and the four
        // colour / depth totals.  The conic and opacity factors are applied by gs_preprocess_bwd.
        const int blk = tid & 15;
        const uint32_t blk_walked = walked_all[blk >> 2];
        const pos_t* blk_posmap = posmap_all[blk >> 2] + (blk & 3) * BATCH;
        const float* blk_slots = POOL > 0 ? slots + (size_t)flush_base * SLOT : slots + (size_t)blk * BSTR;
        // origin of 4x4 block blk = 4 wave + row (make_ctx)
        const int tile_x0 = (tile % tiles_x) * GS_TILE, tile_y0 = (tile / tiles_x) * GS_TILE;
        const float blk_x = (float)(tile_x0 + ((blk >> 2) & 1) * GS_SUB + (blk & 1) * GS_QUAD);
        const float blk_y = (float)(tile_y0 + (blk >> 3) * GS_SUB + ((blk >> 1) & 1) * GS_QUAD);
#pragma unroll
        for (int r = 0; r < BATCH / 16; ++r) {
            const int j = r * 16 + (tid >> 4);
            const uint32_t pos = blk_posmap[j];
            // block blk holds sums for pair j if the pair is in its list at a position its wave walked (POS_NONE = 255 is
            // behind every list; pairs behind n are in no list)
            const bool valid = pos < blk_walked;
            const uint64_t vm = __builtin_amdgcn_ballot_w64(valid);
            GS_STAT(6, 1); GS_STAT(9, __popcll(vm));
            if (vm == 0) continue;                   // wave-uniform: none of this wave's 4 pairs is in a walked list
            GS_STAT(7, 1);
            const bool any = ((vm >> (lane & 48)) & 0xFFFFull) != 0;
            // lanes without a slot read the always-zero slot behind the last one
            const float* sl = valid ? blk_slots + pos * SLOT : slots + NSLOTF;
            float m[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) m[k] = sl[k];
            // block b's local moments -> moments of pv about the CENTRE OF THE GAUSSIAN'S BOX (an integer pixel that
            // every tile derives identically): dx_o = (block origin - centre) + lx with small exact integers, so
            // nothing is lost here however far away the mean is; gs_preprocess_bwd moves the sums to the mean, in
            // double, once per Gaussian
            const uint2 idc = sid[j];
            const float ex = blk_x - (float)(idc.y & 0xFFFFu);
            const float ey = blk_y - (float)(idc.y >> 16);
            const float exM = ex * m[0], eyM = ey * m[0];
            float v[10];
            v[0] = exM + m[1];
            v[1] = eyM + m[3];
            v[2] = __builtin_fmaf(ex, exM + 2.0f * m[1], m[2]);
            v[3] = __builtin_fmaf(ex, eyM + m[3], m[4]) + ey * m[1];
            v[4] = __builtin_fmaf(ey, eyM + 2.0f * m[3], m[5]);
            v[5] = m[0];
            v[6] = m[6];
            v[7] = m[7];
            v[8] = m[8];
            v[9] = m[9];
            const float sum = row_reduce_scatter10<HAS_ZA>(v, t);
            if (any && red_slot >= 0) {
                const float outv = sum;
                // idc.x = 4 x Gaussian index (staging): a row is 64 bytes = 16 x that (one v_lshl_add_u64); the lane's own byte
                // offset inside a row sits in lane_row0 already
                char* dst = lane_row0 + ((uint64_t)idc.x << 4);
                if (red_slot < 6) atomicAdd(reinterpret_cast<double*>(dst), (double)outv);
                else atomicAdd(reinterpret_cast<float*>(dst), outv);
            }
        }
    }
}


}  // namespace

extern "C" int gs_render_fwd(int H, int W, const float* bg, const void* splat, const uint32_t* ranges,
                             const uint32_t* point_list, float* out_image, float* out_depth, float* out_alpha,
                             float* aux_rgb, float t_eps, int tile_row_begin, int tile_row_end, void* zero_buf,
                             size_t zero_bytes, void* stream_) {
    GS_REQUIRE(H > 0 && W > 0 && H < 65536 && W < 65536, "image size %dx%d outside 1..65535", W, H);
    GS_REQUIRE(bg && splat && ranges && out_image && out_depth && out_alpha, "null argument");
    GS_REQUIRE(!zero_buf || ((((uintptr_t)zero_buf | zero_bytes) & 15) == 0), "zero_buf must be 16-byte aligned and sized");
    if (!zero_buf) zero_bytes = 0;
    GS_REQUIRE(t_eps >= 0.0f && t_eps < 1.0f, "t_eps must be in [0,1)");
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    GS_REQUIRE(tile_row_begin >= 0 && tile_row_begin <= tile_row_end, "bad tile row band [%d,%d)", tile_row_begin, tile_row_end);
    if (tile_row_end > tiles_y) tile_row_end = tiles_y;
    hipStream_t stream = (hipStream_t)stream_;
    if (tile_row_begin >= tile_row_end) {
        if (zero_bytes) GS_CHECK_HIP(hipMemsetAsync(zero_buf, 0, zero_bytes, stream));
        return GS_OK;
    }
    const int tile0 = tile_row_begin * tiles_x, ntiles = tiles_x * (tile_row_end - tile_row_begin);
    if (aux_rgb)
        hipLaunchKernelGGL((render_fwd_kernel<true>), dim3(ntiles), dim3(GS_BLOCK), 0, stream, H, W, tiles_x, tile0, ntiles,
                           bg, (const float4*)splat, (const uint2*)ranges, point_list, out_image, out_depth,
                           out_alpha, aux_rgb, t_eps, (f4*)zero_buf, zero_bytes / 16);
    else
        hipLaunchKernelGGL((render_fwd_kernel<false>), dim3(ntiles), dim3(GS_BLOCK), 0, stream, H, W, tiles_x, tile0, ntiles,
                           bg, (const float4*)splat, (const uint2*)ranges, point_list, out_image, out_depth,
                           out_alpha, aux_rgb, t_eps, (f4*)zero_buf, zero_bytes / 16);
    GS_CHECK_LAUNCH("render_fwd");
    return GS_OK;
}

// grad_rows (N x GS_GRAD_ROW floats) must be zero on entry; rows of Gaussians that are not drawn stay zero.
extern "C" int gs_render_bwd(int H, int W, const void* splat, const uint32_t* ranges,
                             const uint32_t* point_list, const float* aux_rgb, const float* out_depth,
                             const float* out_alpha, const float* g_image, const float* g_depth,
                             const float* g_alpha, float* grad_rows, float t_eps, int tile_row_begin,
                             int tile_row_end, void* stream_) {
    GS_REQUIRE(H > 0 && W > 0 && H < 65536 && W < 65536, "image size %dx%d outside 1..65535", W, H);
    GS_REQUIRE(splat && ranges && aux_rgb && out_depth && out_alpha && g_image && grad_rows, "null argument");
    GS_REQUIRE(t_eps >= 0.0f && t_eps < 1.0f, "t_eps must be in [0,1)");
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    GS_REQUIRE(tile_row_begin >= 0 && tile_row_begin <= tile_row_end, "bad tile row band [%d,%d)", tile_row_begin, tile_row_end);
    if (tile_row_end > tiles_y) tile_row_end = tiles_y;
    if (tile_row_begin >= tile_row_end) return GS_OK;
    const int tile0 = tile_row_begin * tiles_x, ntiles = tiles_x * (tile_row_end - tile_row_begin);
    GS_REQUIRE(((uintptr_t)grad_rows & 15) == 0, "grad_rows must be 16-byte aligned");
    if (g_depth || g_alpha)
        hipLaunchKernelGGL((render_bwd_kernel<GS_BWD_BATCH, true, GS_BWD_POOL>), dim3(ntiles), dim3(GS_BLOCK), 0, (hipStream_t)stream_, H, W, tiles_x, tile0, ntiles,
                           (const float4*)splat, (const uint2*)ranges, point_list, aux_rgb, out_depth, out_alpha,
                           g_image, g_depth, g_alpha, grad_rows, t_eps);
    else
        hipLaunchKernelGGL((render_bwd_kernel<GS_BWD_BATCH, false, GS_BWD_POOL>), dim3(ntiles), dim3(GS_BLOCK), 0, (hipStream_t)stream_, H, W, tiles_x, tile0, ntiles,
                           (const float4*)splat, (const uint2*)ranges, point_list, aux_rgb, out_depth, out_alpha,
                           g_image, g_depth, g_alpha, grad_rows, t_eps);
    GS_CHECK_LAUNCH("render_bwd");
    return GS_OK;
}
