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
// Per-visit work is trimmed at staging time (once per (tile, Gaussian) pair instead of once per
// visit and lane): the LDS copy of a record carries the conic pre-scaled for v_exp_f32
// (power * log2 e = A' dx^2 + B' dx dy + C' dy^2) and the box as two packed u16 pairs, so the
// per-lane box test is v_pk_max_u16 + v_pk_min_u16 + one compare, and a stopped lane is folded
// into the same test by setting its packed pixel coordinate to 0xFFFFFFFF.
//
// Backward gradient accumulation: per Gaussian 10 partial sums (the mean/conic ones as raw
// moments sum q dx, q dy, q dx^2, q dx dy, q dy^2 with q = dL/dpower; the per-Gaussian factors are
// applied at flush time) are reduced over the wave by a select-free reduce-scatter (25 VALU,
// registers only); ten lanes store one dword each into the wave's own LDS slot (plain stores, no
// LDS atomics); after the batch the 4 waves' rows are summed and flushed once per (tile, Gaussian)
// as one 40-byte row-shaped float atomic into that Gaussian's 64-byte gradient row.
//
// Bound: VALU (exp + ~20 fp32 ops per lane-visit forward, ~80 backward); HBM traffic is the
// record gather per (tile, Gaussian) pair plus 28 / 44 bytes per pixel.
#include "gsplat_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct Rec { f4 q0, q1, q2; };

// j is wave-uniform (v_readlane result): three broadcast ds_read_b128
__device__ __forceinline__ Rec load_rec(const f4* sq, int j) {
    Rec r;
    r.q0 = sq[3 * j];
    r.q1 = sq[3 * j + 1];
    r.q2 = sq[3 * j + 2];
    return r;
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

constexpr float GS_LOG2E = 1.4426950408889634f;

// LDS form of a splat record (written by stage_and_cull):
//   q0 = (mean_x, mean_y, A', B')   q1 = (C', opacity, depth, x_min | y_min << 16)
//   q2 = (red, green, blue, (x_max - 1) | (y_max - 1) << 16)
// with A' = -0.5 log2e conic_a, B' = -log2e conic_b, C' = -0.5 log2e conic_d.
__device__ __forceinline__ void to_lds_form(f4& p0, f4& p1, f4& p2) {
    const uint32_t rx = __float_as_uint(p1.w), ry = __float_as_uint(p2.w);
    p0.z *= -0.5f * GS_LOG2E;
    p0.w *= -GS_LOG2E;
    p1.x *= -0.5f * GS_LOG2E;
    p1.w = __uint_as_float((rx & 0xFFFFu) | (ry << 16));
    p2.w = __uint_as_float(((rx >> 16) - 1u) | (((ry >> 16) - 1u) << 16));
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

constexpr uint32_t P_STOPPED = 0xFFFFFFFFu;   // never inside a box: x_max - 1 <= 65534

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
__device__ __forceinline__ int stage_and_cull(  // BATCH: multiple of 64, at most GS_BLOCK
const f4* __restrict__ splat, const uint32_t* __restrict__ point_list,
                                              uint32_t base, int n, const TileCtx& c, f4* sq, uint32_t* sid,
                                              uint32_t* wl, uint16_t* posmap) {
    static_assert(BATCH % 64 == 0 && BATCH >= 64 && BATCH <= GS_BLOCK, "one staging lane per pair, whole waves of cull lanes");
    const int tid = threadIdx.x;
    if (tid < n && tid < BATCH) {
        const uint32_t id = point_list[base + tid];
        const f4* p = splat + 3 * (size_t)id;
        f4 p0 = p[0], p1 = p[1], p2 = p[2];
        to_lds_form(p0, p1, p2);
        sq[3 * tid] = p0;
        sq[3 * tid + 1] = p1;
        sq[3 * tid + 2] = p2;
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
            const uint32_t lo = squ[12 * j + 7], hi = squ[12 * j + 11];
            const int xmin = lo & 0xFFFF, ymin = lo >> 16, xlast = hi & 0xFFFF, ylast = hi >> 16;
            hit = xmin < c.sx0 + GS_SUB && xlast >= c.sx0 && ymin < c.sy0 + GS_SUB && ylast >= c.sy0;
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
    float* __restrict__ out_depth, float* __restrict__ out_alpha, float* __restrict__ aux_rgb, float t_eps,
    f4* __restrict__ zero_buf, size_t zero_n) {
    constexpr int BATCH = 256;
    if (zero_buf) {
        // This kernel is VALU-bound and leaves HBM nearly idle: clearing the backward's gradient-row
        // buffer here (fire-and-forget 16-byte stores, one slice per workgroup) is free, a memset
        // launched on its own costs ~24 us per frame at 3 M Gaussians.
        const size_t chunk = (zero_n + gridDim.x - 1) / gridDim.x;
        const size_t z0 = (size_t)blockIdx.x * chunk;
        const f4 z = {0.f, 0.f, 0.f, 0.f};
        for (size_t j = threadIdx.x; j < chunk && z0 + j < zero_n; j += GS_BLOCK) zero_buf[z0 + j] = z;
    }
    __shared__ f4 sq[3 * BATCH];
    __shared__ uint32_t wl_all[GS_BLOCK / 64][BATCH];
    const int tile = tile0 + tile_of_block(blockIdx.x, ntiles);
    const TileCtx c = make_ctx(tile, tiles_x, H, W);
    const int lane = threadIdx.x & 63;
    uint32_t* wl = wl_all[threadIdx.x >> 6];
    const uint2 range = ranges[tile];
    const float fpx = (float)c.px, fpy = (float)c.py;
    float A = 0.f, T = 1.f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Z = 0.f;
    uint32_t P = c.inside ? ((uint32_t)c.px | (uint32_t)c.py << 16) : P_STOPPED;
    for (uint32_t base = range.x; base < range.y; base += BATCH) {
        if (__syncthreads_and(P == P_STOPPED)) break;
        const int n = min((int)(range.y - base), BATCH);
        const int count = stage_and_cull<BATCH>((const f4*)splat, point_list, base, n, c, sq, nullptr, wl, nullptr);
        bool stop = false;
        for (int c0 = 0; c0 < count && !stop; c0 += 64) {
            const int cn = min(64, count - c0);
            const uint32_t jv = wl[c0 + lane];
            for (int kk = 0; kk < cn; ++kk) {
                if ((kk & 7) == 0 && __all(P == P_STOPPED)) { stop = true; break; }
                const Rec rec = load_rec(sq, __builtin_amdgcn_readlane(jv, kk));
                const bool act = in_box(P, __float_as_uint(rec.q1.w), __float_as_uint(rec.q2.w));
                const float dx = fpx - rec.q0.x, dy = fpy - rec.q0.y;
                const float power2 = gauss_power2(rec, dx, dy);
                const float w = __builtin_amdgcn_exp2f(power2) * rec.q1.y;
                const float alpha = act ? w * T : 0.0f;
                Cr += alpha * rec.q2.x;
                Cg += alpha * rec.q2.y;
                Cb += alpha * rec.q2.z;
                Z += alpha * rec.q1.z;
                A += alpha;
                T = 1.0f - A;
                P = T < t_eps ? P_STOPPED : P;
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

// Select-free wave reduce-scatter of 10 values (registers only, 25 VALU instead of 60-70 for ten
// separate butterfly reductions): every step adds the partner lane's copy AND halves the number
// of values a lane still carries, so the work shrinks 10 -> 5 -> 3 -> 2 -> 1.  The steps whose
// "which half do I keep" is a DPP bank (4 lanes) or a row (16 lanes) come first, while there are
// many values, because there the selection is free (bank_mask / permlane swap); the lane^1 and
// lane^2 steps, which would need v_cndmask selects, come last on the single remaining value.
//   step 1  lane^8   row_ror:8, bank_mask picks v[j] (banks 0,1) or v[j+5] (banks 2,3)
//   step 2  lane^4   row_shl:4 / row_shr:4 per bank
//   step 3  lane^16  v_permlane16_swap + add      step 4  lane^32  v_permlane32_swap + add
//   step 5  quad sum quad_perm xor 1, xor 2
// Result: all four lanes of a quad hold the 64-lane total of one value; first lane of the quad:
//   v0:0  v1:16  v2:4  v3:20  v4:32  v5:8  v6:24  v7:12  v8:28  v9:40     (tools/dpp_test/t2.hip)
// s_nop: a DPP operand needs 2 wait states after the VALU write of that register.
__device__ __forceinline__ float wave_reduce_scatter10(const float (&v)[10]) {
    float o0, o1, o2, o3, o4, p0, p1, p2;
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
        "s_nop 1"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(p0), "=&v"(p1), "=&v"(p2)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
          "v"(v[9]));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
    const float q0 = __uint_as_float(a.x) + __uint_as_float(a.y);
    const u2 b = __builtin_amdgcn_permlane16_swap(__float_as_uint(p2), __float_as_uint(p2), false, false);
    const float q1 = __uint_as_float(b.x) + __uint_as_float(b.y);
    const u2 c = __builtin_amdgcn_permlane32_swap(__float_as_uint(q0), __float_as_uint(q1), false, false);
    float r = __uint_as_float(c.x) + __uint_as_float(c.y);
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(r));
    return r;
}

__global__ __launch_bounds__(GS_BLOCK) void render_bwd_kernel(
    int H, int W, int tiles_x, int tile0, int ntiles, const float4* __restrict__ splat, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float* __restrict__ aux_rgb,
    const float* __restrict__ out_depth, const float* __restrict__ out_alpha, const float* __restrict__ g_img,
    const float* __restrict__ g_depth, const float* __restrict__ g_alpha, float* __restrict__ grad_rows,
    float t_eps) {
    constexpr int BATCH = 64;
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
    float A = 0.f, T = 1.f, rem = total;
    const f2 g_rg = {gr, gg}, g_bz = {gb, gz};
    uint32_t P = c.inside ? ((uint32_t)c.px | (uint32_t)c.py << 16) : P_STOPPED;
    // which float of the 12-float slot this lane stores after wave_reduce_scatter10 (-1: none):
    // value v_k = row float k; lane 36 (a duplicate of v4) carries the "written" flag, float 10
    int red_slot = -1;
    if ((lane & 3) == 0) {
        // quad:  0 1 2 3 4 5 6 7 8 9  10 11..15   (one nibble per quad, 0xF = none)
        // float: 0 2 5 7 1 3 6 8 4 10 9  -
        const int nib = (int)((0xFFFFF9A486317520ull >> (4 * (lane >> 2))) & 0xF);
        red_slot = nib == 0xF ? -1 : nib;
    }
    for (uint32_t base = range.x; base < range.y; base += BATCH) {
        if (__syncthreads_and(P == P_STOPPED)) break;
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
                if ((kk & 7) == 0 && __all(P == P_STOPPED)) { stop = true; break; }
                const Rec rec = load_rec(sq, __builtin_amdgcn_readlane(jv, kk));
                const bool act = in_box(P, __float_as_uint(rec.q1.w), __float_as_uint(rec.q2.w));
                if (!__any(act)) continue;   // slot keeps flag 0
                const float dx = fpx - rec.q0.x, dy = fpy - rec.q0.y;
                const float power2 = gauss_power2(rec, dx, dy);
                const float G = act ? __builtin_amdgcn_exp2f(power2) : 0.0f;  // inactive lanes: every sum gets 0
                const float w = G * rec.q1.y;
                const float alpha = w * T;
                const float u = __builtin_fmaf(rec.q2.x, g_rg.x, __builtin_fmaf(rec.q2.y, g_rg.y,
                                __builtin_fmaf(rec.q2.z, g_bz.x, __builtin_fmaf(rec.q1.z, g_bz.y, ga))));
                rem = __builtin_fmaf(-u, alpha, rem);      // total - sum_{j<=i} u_j alpha_j
                const float inv = __builtin_amdgcn_rcpf(fmaxf(1.0f - w, 1e-6f));
                const float dLdw = T * u - rem * inv;
                const float q = w * dLdw;          // dL/dpower
                const f2 dxy = {dx, dy};
                const f2 qxy = q * dxy;            // v_pk_mul_f32
                const f2 qxd = qxy.x * dxy;
                const f2 c_rg = alpha * g_rg, c_bz = alpha * g_bz;
                float v[10];
                v[0] = qxy.x;
                v[1] = qxy.y;
                v[2] = qxd.x;
                v[3] = qxd.y;
                v[4] = qxy.y * dy;
                v[5] = G * dLdw;
                v[6] = c_rg.x;
                v[7] = c_rg.y;
                v[8] = c_bz.x;
                v[9] = c_bz.y;
                A += alpha;
                T = 1.0f - A;
                P = T < t_eps ? P_STOPPED : P;
                const float red = wave_reduce_scatter10(v);
                if (red_slot >= 0) ((float*)&acc_all[wave][c0 + kk][0])[red_slot] = (lane == 36) ? 1.0f : red;
            }
        }
        __syncthreads();
        // flush: 16 lanes per pair, lanes 0..9 carry the row summed over the waves that hit it.
        // Floats 0..4 are the moments (sum q dx, q dy, q dx^2, q dx dy, q dy^2); the gradient row is
        //   d mean = (a Sx + b Sy, d Sy + b Sx),  d conic = -0.5 (Sxx, Sxy, Syy).
        const int sub = tid & 15;
#pragma unroll 2
        for (int r = 0; r < BATCH / 16; ++r) {
            const int j = r * 16 + (tid >> 4);
            float sum = 0.0f;
            bool any = false;
            if (j < n && sub < 10) {
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) {
                    const uint32_t p = posmap_all[w2][j];
                    if (p != 0xFFFFu) {
                        const float* a = (const float*)&acc_all[w2][p][0];
                        if (a[10] != 0.0f) { sum += a[sub]; any = true; }
                    }
                }
            }
            const float other = __shfl_xor(sum, 1);
            if (any) {
                float outv = sum;
                if (sub < 2) {
                    const float* rq = (const float*)&sq[3 * j];
                    const float ca = rq[2] * (-2.0f / GS_LOG2E), cb = rq[3] * (-1.0f / GS_LOG2E),
                                cd = rq[4] * (-2.0f / GS_LOG2E);
                    outv = (sub == 0 ? ca : cd) * sum + cb * other;
                } else if (sub < 5) {
                    outv = -0.5f * sum;
                }
                atomicAdd(grad_rows + (size_t)sid[j] * GS_GRAD_ROW + sub, outv);
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
