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
// sub-tile (lane = pixel).  Per batch of the tile's sorted list the workgroup stages (id, box) of
// every pair in LDS; each wave ballots the boxes against its own sub-tile and keeps only the
// compacted hits, so lanes are not spent on Gaussians whose box misses the whole wave.  Inside
// the per-Gaussian loop the 48-byte record is read from LDS with exactly three broadcast
// ds_read_b128 (12 LDS cycles); the hit index comes from a register via v_readlane.  (The first
// version used b96/b32/u8 reads, 26 LDS cycles per Gaussian, plus LDS float atomics in the
// backward: rocprof showed the LDS array 70-80 % busy, profiles/r1a_*.  Scalar s_load of the
// records was tried and is slower: the scalar cache cannot stream misses.)  A lane stops once its
// transmittance 1 - A drops below t_eps, a wave when all its lanes stopped (ballot), a workgroup
// when all its waves did.  t_eps = 0 reproduces the reference exactly (no stop).
//
// Backward gradient accumulation: per Gaussian 10 partial sums are reduced over the wave with DPP
// row shifts + row broadcasts (registers only); lane 63 stores the row into the wave's own LDS
// slot (plain stores, no LDS atomics); after the batch the 4 waves' rows are summed and flushed
// once per (tile, Gaussian) as one 40-byte row-shaped float atomic into that Gaussian's 64-byte
// gradient row.
//
// Bound: VALU (exp + ~25 fp32 ops per lane-visit forward, ~100 backward); HBM traffic is the
// record gather per (tile, Gaussian) pair plus 28 / 44 bytes per pixel.
#include "gsplat_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

struct Rec { f4 q0, q1, q2; };

// j is wave-uniform (v_readlane result): three broadcast ds_read_b128
__device__ __forceinline__ Rec load_rec(const f4* sq, int j) {
    Rec r;
    r.q0 = sq[3 * j];
    r.q1 = sq[3 * j + 1];
    r.q2 = sq[3 * j + 2];
    return r;
}

struct TileCtx {
    int px, py;        // this lane's pixel
    int sx0, sy0;      // this wave's sub-tile origin
    bool inside;
};

__device__ __forceinline__ int tile_of_block(int bid, int ntiles) {
    // XCD-aware order: consecutive workgroup ids are dealt round-robin to the 8 XCDs; give every
    // XCD one contiguous range of tiles so neighbouring tiles (shared Gaussians) share an L2.
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__device__ __forceinline__ TileCtx make_ctx(int tile, int tiles_x, int H, int W) {
    TileCtx c;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    c.sx0 = tx * GS_TILE + (wave & 1) * GS_SUB;
    c.sy0 = ty * GS_TILE + (wave >> 1) * GS_SUB;
    c.px = c.sx0 + (lane & 7);
    c.py = c.sy0 + (lane >> 3);
    c.inside = c.px < W && c.py < H;
    return c;
}

// Stage the records of `n` pairs of the tile's sorted list into LDS (sq, 3 x 16 B per pair), then
// compact, per wave, the batch indices of the pairs whose box overlaps the wave's 8x8 sub-tile into
// wl[] (in list order).  When posmap is given, posmap[j] = position of pair j in the wave's
// compacted list, 0xFFFF if culled, and sid[j] = Gaussian id.  Returns the wave's hit count.
template <int BATCH>
__device__ __forceinline__ int stage_and_cull(const f4* __restrict__ splat, const uint32_t* __restrict__ point_list,
                                              uint32_t base, int n, const TileCtx& c, f4* sq, uint32_t* sid,
                                              uint32_t* wl, uint16_t* posmap) {
    const int tid = threadIdx.x;
    if (tid < n && tid < BATCH) {
        const uint32_t id = point_list[base + tid];
        const f4* p = splat + 3 * (size_t)id;
        sq[3 * tid] = p[0];
        sq[3 * tid + 1] = p[1];
        sq[3 * tid + 2] = p[2];
        if (sid) sid[tid] = id;
    }
    __syncthreads();
    const int lane = tid & 63;
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t* squ = (const uint32_t*)sq;
    int count = 0;
#pragma unroll
    for (int r = 0; r < BATCH / 64; ++r) {
        const int j = r * 64 + lane;
        bool hit = false;
        if (j < n) {
            const uint32_t rx = squ[12 * j + 7], ry = squ[12 * j + 11];
            const int xmin = rx & 0xFFFF, xmax = rx >> 16, ymin = ry & 0xFFFF, ymax = ry >> 16;
            hit = xmin < c.sx0 + GS_SUB && xmax > c.sx0 && ymin < c.sy0 + GS_SUB && ymax > c.sy0;
        }
        const uint64_t m = __ballot(hit);
        const int pos = count + __popcll(m & lt);
        if (hit) wl[pos] = (uint32_t)j;
        if (posmap) posmap[j] = hit ? (uint16_t)pos : (uint16_t)0xFFFF;
        count += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return count;
}

template <bool AUX>
__global__ __launch_bounds__(GS_BLOCK) void render_fwd_kernel(
    int H, int W, int tiles_x, int tile0, int ntiles, const float* __restrict__ bg, const float4* __restrict__ splat,
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, float* __restrict__ out_img,
    float* __restrict__ out_depth, float* __restrict__ out_alpha, float* __restrict__ aux_rgb, float t_eps) {
    constexpr int BATCH = 256;
    __shared__ f4 sq[3 * BATCH];
    __shared__ uint32_t wl_all[GS_BLOCK / 64][BATCH];
    const int tile = tile0 + tile_of_block(blockIdx.x, ntiles);
    const TileCtx c = make_ctx(tile, tiles_x, H, W);
    const int lane = threadIdx.x & 63;
    uint32_t* wl = wl_all[threadIdx.x >> 6];
    const uint2 range = ranges[tile];
    const float fpx = (float)c.px, fpy = (float)c.py;
    float A = 0.f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Z = 0.f;
    bool done = !c.inside;
    for (uint32_t base = range.x; base < range.y; base += BATCH) {
        if (__syncthreads_and(done)) break;
        const int n = min((int)(range.y - base), BATCH);
        const int count = stage_and_cull<BATCH>((const f4*)splat, point_list, base, n, c, sq, nullptr, wl, nullptr);
        bool stop = false;
        for (int c0 = 0; c0 < count && !stop; c0 += 64) {
            const int cn = min(64, count - c0);
            const uint32_t jv = wl[c0 + lane];
            for (int kk = 0; kk < cn; ++kk) {
                if ((kk & 7) == 0 && __all(done)) { stop = true; break; }
                const Rec rec = load_rec(sq, __builtin_amdgcn_readlane(jv, kk));
                const uint32_t rx = __float_as_uint(rec.q1.w), ry = __float_as_uint(rec.q2.w);
                const int xmin = rx & 0xFFFF, xmax = rx >> 16, ymin = ry & 0xFFFF, ymax = ry >> 16;
                const bool act = !done && c.px >= xmin && c.px < xmax && c.py >= ymin && c.py < ymax;
                const float dx = fpx - rec.q0.x, dy = fpy - rec.q0.y;
                const float power = -0.5f * ((rec.q0.z * dx * dx + (rec.q0.w + rec.q0.w) * dx * dy) + rec.q1.x * dy * dy);
                const float w = __expf(power) * rec.q1.y;
                const float alpha = act ? w * (1.0f - A) : 0.0f;
                Cr += alpha * rec.q2.x;
                Cg += alpha * rec.q2.y;
                Cb += alpha * rec.q2.z;
                Z += alpha * rec.q1.z;
                A += alpha;
                done = done || (1.0f - A) < t_eps;
            }
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

// Transposing wave reduction of 10 values (registers only, 29 VALU instead of 60-70 for ten
// separate butterfly reductions): every step adds the partner lane's copy AND halves the number
// of values a lane still carries (reduce-scatter), so the work shrinks 10 -> 5 -> 3 -> 2 -> 1.
//   step A  lane^1  quad_perm:[1,0,3,2]   keep v[j] (even lanes) or v[j+5] (odd lanes)
//   step B  lane^2  quad_perm:[2,3,0,1]   keep by lane bit 1
//   step C  lane^8  row_ror:8             selection by bank_mask (banks = groups of 4 lanes): free
//   step D  lane^4  row_shl:4 / row_shr:4 per bank: free
//   then v_permlane16_swap / v_permlane32_swap fold the four 16-lane rows lane-wise.
// Result: every lane with (lane & 15) == k, in particular lane 48 + k, holds the 64-lane total of value GS_RED_VALUE_OF_LANE[k]:
//   k: 0->v0 1->v5 2->v3 3->v8 4->v2 5->v7 8->v1 9->v6 10->v4 11->v9   (k = 6,7,12..15: duplicates)
// s_nop: a DPP operand needs 2 wait states after the VALU write of that register.
__device__ __forceinline__ float wave_reduce_scatter10(float (&v)[10], bool odd, bool bit1) {
    float t0, out;
    // step A: every register += its lane^1 copy
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %9, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
          "+v"(v[8]), "+v"(v[9]));
    float a0 = odd ? v[5] : v[0], a1 = odd ? v[6] : v[1], a2 = odd ? v[7] : v[2], a3 = odd ? v[8] : v[3],
          a4 = odd ? v[9] : v[4];
    // step B: += lane^2 copy
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4));
    float b0 = bit1 ? a3 : a0, b1 = bit1 ? a4 : a1;
    // step C (lane^8, selection by bank_mask) and step D (lane^4, per bank)
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %1, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "s_nop 1"
        : "=&v"(t0), "=&v"(out), "+v"(b0), "+v"(b1), "+v"(a2));
    // fold the four 16-lane rows lane-wise (row_bcast would mix lanes that carry different values)
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(out), __float_as_uint(out), false, false);
    const float s = __uint_as_float(r.x) + __uint_as_float(r.y);
    const u2 q = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(q.x) + __uint_as_float(q.y);
}

__global__ __launch_bounds__(GS_BLOCK) void render_bwd_kernel(
    int H, int W, int tiles_x, int tile0, int ntiles, const float4* __restrict__ splat, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float* __restrict__ aux_rgb,
    const float* __restrict__ out_depth, const float* __restrict__ out_alpha, const float* __restrict__ g_img,
    const float* __restrict__ g_depth, const float* __restrict__ g_alpha, float* __restrict__ grad_rows,
    float t_eps) {
    constexpr int BATCH = 128;
    constexpr int NW = GS_BLOCK / 64;
    __shared__ f4 sq[3 * BATCH];
    __shared__ uint32_t sid[BATCH];
    __shared__ uint32_t wl_all[NW][BATCH];
    __shared__ uint16_t posmap_all[NW][BATCH];
    __shared__ f4 acc_all[NW][BATCH][3];   // per wave, per compacted hit: 10 sums, [10] = "written" flag
    const int tile = tile0 + tile_of_block(blockIdx.x, ntiles);
    const TileCtx c = make_ctx(tile, tiles_x, H, W);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* wl = wl_all[wave];
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
    float A = 0.f, prefix = 0.f;
    bool done = !c.inside;
    // which float of the 12-float row this lane stores after wave_reduce_scatter10 (-1: none)
    int red_slot = -1;
    if (lane >= 48) {
        // k:    0 1 2 3 4 5 6  7 8 9 10 11 12..15   (one nibble per k, 0xF = none; 10 = flag slot)
        // slot: 0 5 3 8 2 7 10 - 1 6 4  9  -
        const int nib = (int)((0xFFFF9461FA728350ull >> (4 * (lane - 48))) & 0xF);
        red_slot = nib == 0xF ? -1 : nib;
    }
    for (uint32_t base = range.x; base < range.y; base += BATCH) {
        if (__syncthreads_and(done)) break;
        const int n = min((int)(range.y - base), BATCH);
        // clear the "written" flags of every slot (2 per lane: NW*BATCH = 512 slots)
#pragma unroll
        for (int s = tid; s < NW * BATCH; s += GS_BLOCK) ((float*)&acc_all[0][0][0])[s * 12 + 10] = 0.0f;
        const int count = stage_and_cull<BATCH>((const f4*)splat, point_list, base, n, c, sq, sid, wl,
                                                posmap_all[wave]);
        bool stop = false;
        for (int c0 = 0; c0 < count && !stop; c0 += 64) {
            const int cn = min(64, count - c0);
            const uint32_t jv = wl[c0 + lane];
            for (int kk = 0; kk < cn; ++kk) {
                if ((kk & 7) == 0 && __all(done)) { stop = true; break; }
                const Rec rec = load_rec(sq, __builtin_amdgcn_readlane(jv, kk));
                const uint32_t rx = __float_as_uint(rec.q1.w), ry = __float_as_uint(rec.q2.w);
                const int xmin = rx & 0xFFFF, xmax = rx >> 16, ymin = ry & 0xFFFF, ymax = ry >> 16;
                const bool act = !done && c.px >= xmin && c.px < xmax && c.py >= ymin && c.py < ymax;
                if (!__any(act)) continue;   // slot keeps flag 0
                const float ca = rec.q0.z, cb = rec.q0.w, cd = rec.q1.x, op = rec.q1.y, z = rec.q1.z;
                const float dx = fpx - rec.q0.x, dy = fpy - rec.q0.y;
                const float power = -0.5f * ((ca * dx * dx + (cb + cb) * dx * dy) + cd * dy * dy);
                const float G = act ? __expf(power) : 0.0f;  // inactive lanes must not leak inf/NaN into the sums
                const float w = G * op;
                const float T = 1.0f - A;
                const float alpha = w * T;
                const float u = (rec.q2.x * gr + rec.q2.y * gg + rec.q2.z * gb) + z * gz + ga;
                prefix += u * alpha;
                const float suffix = total - prefix;
                const float inv = __builtin_amdgcn_rcpf(fmaxf(1.0f - w, 1e-6f));
                const float dLdw = act ? (T * u - suffix * inv) : 0.0f;
                const float dLdp = w * dLdw;
                const float h = -0.5f * dLdp;
                float v[10];
                v[0] = dLdp * (ca * dx + cb * dy);
                v[1] = dLdp * (cd * dy + cb * dx);
                v[2] = h * dx * dx;
                v[3] = h * dx * dy;
                v[4] = h * dy * dy;
                v[5] = G * dLdw;
                v[6] = alpha * gr;
                v[7] = alpha * gg;
                v[8] = alpha * gb;
                v[9] = alpha * gz;
                A += alpha;
                done = done || (1.0f - A) < t_eps;
                const float red = wave_reduce_scatter10(v, (lane & 1) != 0, (lane & 2) != 0);
                // lanes 48.. hold the totals; lane 54 (a duplicate) carries the "written" flag instead
                if (red_slot >= 0) ((float*)&acc_all[wave][c0 + kk][0])[red_slot] = (lane == 54) ? 1.0f : red;
            }
        }
        __syncthreads();
        // flush: 16 lanes per pair, lanes 0..9 carry the row summed over the waves that hit it
        const int sub = tid & 15;
#pragma unroll 2
        for (int r = 0; r < BATCH / 16; ++r) {
            const int j = r * 16 + (tid >> 4);
            if (j < n && sub < 10) {
                float sum = 0.0f;
                bool any = false;
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) {
                    const uint32_t p = posmap_all[w2][j];
                    if (p != 0xFFFFu) {
                        const float* a = (const float*)&acc_all[w2][p][0];
                        if (a[10] != 0.0f) { sum += a[sub]; any = true; }
                    }
                }
                if (any) atomicAdd(grad_rows + (size_t)sid[j] * GS_GRAD_ROW + sub, sum);
            }
        }
    }
}

}  // namespace

extern "C" int gs_render_fwd(int H, int W, const float* bg, const void* splat, const uint32_t* ranges,
                             const uint32_t* point_list, float* out_image, float* out_depth, float* out_alpha,
                             float* aux_rgb, float t_eps, int tile_row_begin, int tile_row_end, void* stream_) {
    GS_REQUIRE(H > 0 && W > 0 && H < 65536 && W < 65536, "image size %dx%d outside 1..65535", W, H);
    GS_REQUIRE(bg && splat && ranges && out_image && out_depth && out_alpha, "null argument");
    GS_REQUIRE(t_eps >= 0.0f && t_eps < 1.0f, "t_eps must be in [0,1)");
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    GS_REQUIRE(tile_row_begin >= 0 && tile_row_begin <= tile_row_end, "bad tile row band [%d,%d)", tile_row_begin, tile_row_end);
    if (tile_row_end > tiles_y) tile_row_end = tiles_y;
    if (tile_row_begin >= tile_row_end) return GS_OK;
    const int tile0 = tile_row_begin * tiles_x, ntiles = tiles_x * (tile_row_end - tile_row_begin);
    hipStream_t stream = (hipStream_t)stream_;
    if (aux_rgb)
        hipLaunchKernelGGL((render_fwd_kernel<true>), dim3(ntiles), dim3(GS_BLOCK), 0, stream, H, W, tiles_x, tile0, ntiles,
                           bg, (const float4*)splat, (const uint2*)ranges, point_list, out_image, out_depth,
                           out_alpha, aux_rgb, t_eps);
    else
        hipLaunchKernelGGL((render_fwd_kernel<false>), dim3(ntiles), dim3(GS_BLOCK), 0, stream, H, W, tiles_x, tile0, ntiles,
                           bg, (const float4*)splat, (const uint2*)ranges, point_list, out_image, out_depth,
                           out_alpha, aux_rgb, t_eps);
    GS_CHECK_LAUNCH("render_fwd");
    return GS_OK;
}

// grad_rows (N x 16 floats) must be zero on entry; rows of Gaussians that are not drawn stay zero.
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
    hipLaunchKernelGGL(render_bwd_kernel, dim3(ntiles), dim3(GS_BLOCK), 0, (hipStream_t)stream_, H, W, tiles_x, tile0, ntiles,
                       (const float4*)splat, (const uint2*)ranges, point_list, aux_rgb, out_depth, out_alpha,
                       g_image, g_depth, g_alpha, grad_rows, t_eps);
    GS_CHECK_LAUNCH("render_bwd");
    return GS_OK;
}
