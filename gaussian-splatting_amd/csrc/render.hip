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
// sub-tile (lane = pixel).  A batch of up to 256 splat records is staged in LDS; each wave then
// ballots the records' boxes against its own sub-tile and walks only the compacted hits, so
// lanes are not spent on Gaussians whose box misses the whole wave.  A lane stops once its
// transmittance 1 - A drops below t_eps, a wave when all its lanes stopped (ballot), a
// workgroup when all its waves did.  t_eps = 0 reproduces the reference exactly (no stop).
//
// Backward gradient accumulation: per Gaussian 10 partial sums are reduced over the wave with
// DPP row shifts / broadcasts (no LDS traffic), added by one lane into LDS accumulators shared
// by the 4 waves, and flushed once per (tile, Gaussian) as one 40-byte row-shaped float atomic
// into the 64-byte gradient row of that Gaussian.
//
// Bound: VALU (exp + ~25 fp32 ops per lane-visit forward, ~70 backward); HBM traffic is the
// 48-byte record gather per (tile, Gaussian) pair plus 28 / 44 bytes per pixel.
#include "gsplat_common.h"

namespace {

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

// Stage `n` records of the tile's sorted list into LDS, then compact, per wave, the indices of
// the records whose box overlaps the wave's 8x8 sub-tile.  Returns the wave's hit count.
__device__ __forceinline__ int stage_and_cull(const float4* __restrict__ splat,
                                              const uint32_t* __restrict__ point_list, uint32_t base, int n,
                                              const TileCtx& c, float4* sq0, float4* sq1, float4* sq2,
                                              uint32_t* sid, uint8_t* widx) {
    const int tid = threadIdx.x;
    if (tid < n) {
        const uint32_t id = point_list[base + tid];
        const float4* p = splat + 3 * (size_t)id;
        sq0[tid] = p[0];
        sq1[tid] = p[1];
        sq2[tid] = p[2];
        if (sid) sid[tid] = id;
    }
    __syncthreads();
    const int lane = tid & 63;
    const uint64_t lt = (1ull << lane) - 1ull;
    int count = 0;
#pragma unroll
    for (int r = 0; r < GS_BLOCK / 64; ++r) {
        const int j = r * 64 + lane;
        bool hit = false;
        if (j < n) {
            const uint32_t rx = __float_as_uint(sq1[j].w), ry = __float_as_uint(sq2[j].w);
            const int xmin = rx & 0xFFFF, xmax = rx >> 16, ymin = ry & 0xFFFF, ymax = ry >> 16;
            hit = xmin < c.sx0 + GS_SUB && xmax > c.sx0 && ymin < c.sy0 + GS_SUB && ymax > c.sy0;
        }
        const uint64_t m = __ballot(hit);
        if (hit) widx[count + __popcll(m & lt)] = (uint8_t)j;
        count += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return count;
}

template <bool AUX>
__global__ __launch_bounds__(GS_BLOCK) void render_fwd_kernel(
    int H, int W, int tiles_x, int ntiles, const float* __restrict__ bg, const float4* __restrict__ splat,
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, float* __restrict__ out_img,
    float* __restrict__ out_depth, float* __restrict__ out_alpha, float* __restrict__ aux_rgb, float t_eps) {
    __shared__ float4 sq0[GS_BLOCK], sq1[GS_BLOCK], sq2[GS_BLOCK];
    __shared__ uint8_t widx_all[GS_BLOCK / 64][GS_BLOCK];
    const int tile = tile_of_block(blockIdx.x, ntiles);
    const TileCtx c = make_ctx(tile, tiles_x, H, W);
    uint8_t* widx = widx_all[threadIdx.x >> 6];
    const uint2 range = ranges[tile];
    const float fpx = (float)c.px, fpy = (float)c.py;
    float A = 0.f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Z = 0.f;
    bool done = !c.inside;
    for (uint32_t base = range.x; base < range.y; base += GS_BLOCK) {
        if (__syncthreads_and(done)) break;
        const int n = min((int)(range.y - base), GS_BLOCK);
        const int count = stage_and_cull(splat, point_list, base, n, c, sq0, sq1, sq2, nullptr, widx);
        for (int k = 0; k < count; ++k) {
            if ((k & 7) == 0 && __all(done)) break;
            const int j = widx[k];
            const float4 q0 = sq0[j], q1 = sq1[j], q2 = sq2[j];
            const uint32_t rx = __float_as_uint(q1.w), ry = __float_as_uint(q2.w);
            const int xmin = rx & 0xFFFF, xmax = rx >> 16, ymin = ry & 0xFFFF, ymax = ry >> 16;
            const bool act = !done && c.px >= xmin && c.px < xmax && c.py >= ymin && c.py < ymax;
            const float dx = fpx - q0.x, dy = fpy - q0.y;
            const float power = -0.5f * ((q0.z * dx * dx + (q0.w + q0.w) * dx * dy) + q1.x * dy * dy);
            const float w = __expf(power) * q1.y;
            const float alpha = act ? w * (1.0f - A) : 0.0f;
            Cr += alpha * q2.x;
            Cg += alpha * q2.y;
            Cb += alpha * q2.z;
            Z += alpha * q1.z;
            A += alpha;
            done = done || (1.0f - A) < t_eps;
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

// sum over the 64 lanes; the total is valid in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
#define GS_DPP_ADD(ctrl, rmask)                                                                          \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, true))
    GS_DPP_ADD(0x111, 0xF);  // row_shr:1
    GS_DPP_ADD(0x112, 0xF);  // row_shr:2
    GS_DPP_ADD(0x114, 0xF);  // row_shr:4
    GS_DPP_ADD(0x118, 0xF);  // row_shr:8   -> lane 15 of each row holds the row sum
    GS_DPP_ADD(0x142, 0xA);  // row_bcast:15 into rows 1 and 3
    GS_DPP_ADD(0x143, 0xC);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef GS_DPP_ADD
    return v;
}

__global__ __launch_bounds__(GS_BLOCK) void render_bwd_kernel(
    int H, int W, int tiles_x, int ntiles, const float4* __restrict__ splat, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float* __restrict__ aux_rgb,
    const float* __restrict__ out_depth, const float* __restrict__ out_alpha, const float* __restrict__ g_img,
    const float* __restrict__ g_depth, const float* __restrict__ g_alpha, float* __restrict__ grad_rows,
    float t_eps) {
    __shared__ float4 sq0[GS_BLOCK], sq1[GS_BLOCK], sq2[GS_BLOCK];
    __shared__ uint32_t sid[GS_BLOCK];
    __shared__ uint8_t widx_all[GS_BLOCK / 64][GS_BLOCK];
    __shared__ float acc[GS_BLOCK][10];
    __shared__ uint32_t hit[GS_BLOCK];
    const int tile = tile_of_block(blockIdx.x, ntiles);
    const TileCtx c = make_ctx(tile, tiles_x, H, W);
    const int tid = threadIdx.x, lane = tid & 63;
    uint8_t* widx = widx_all[tid >> 6];
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
    for (uint32_t base = range.x; base < range.y; base += GS_BLOCK) {
        if (__syncthreads_and(done)) break;
        const int n = min((int)(range.y - base), GS_BLOCK);
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[tid][k] = 0.0f;
        hit[tid] = 0;
        const int count = stage_and_cull(splat, point_list, base, n, c, sq0, sq1, sq2, sid, widx);
        for (int k = 0; k < count; ++k) {
            if ((k & 7) == 0 && __all(done)) break;
            const int j = widx[k];
            const float4 q0 = sq0[j], q1 = sq1[j], q2 = sq2[j];
            const uint32_t rx = __float_as_uint(q1.w), ry = __float_as_uint(q2.w);
            const int xmin = rx & 0xFFFF, xmax = rx >> 16, ymin = ry & 0xFFFF, ymax = ry >> 16;
            const bool act = !done && c.px >= xmin && c.px < xmax && c.py >= ymin && c.py < ymax;
            if (!__any(act)) continue;
            const float ca = q0.z, cb = q0.w, cd = q1.x, op = q1.y, z = q1.z;
            const float dx = fpx - q0.x, dy = fpy - q0.y;
            const float power = -0.5f * ((ca * dx * dx + (cb + cb) * dx * dy) + cd * dy * dy);
            const float G = act ? __expf(power) : 0.0f;  // inactive lanes must not leak inf/NaN into the sums
            const float w = G * op;
            const float T = 1.0f - A;
            const float alpha = act ? w * T : 0.0f;
            const float u = (q2.x * gr + q2.y * gg + q2.z * gb) + z * gz + ga;
            prefix += u * alpha;
            const float suffix = total - prefix;
            const float inv = __frcp_rn(fmaxf(1.0f - w, 1e-6f));
            const float dLdw = act ? (T * u - suffix * inv) : 0.0f;
            const float dLdp = w * dLdw;
            float v0 = dLdp * (ca * dx + cb * dy);
            float v1 = dLdp * (cd * dy + cb * dx);
            const float h = -0.5f * dLdp;
            float v2 = h * dx * dx;
            float v3 = h * dx * dy;
            float v4 = h * dy * dy;
            float v5 = G * dLdw;
            float v6 = alpha * gr;
            float v7 = alpha * gg;
            float v8 = alpha * gb;
            float v9 = alpha * gz;
            A += alpha;
            done = done || (1.0f - A) < t_eps;
            v0 = wave_sum_to_lane63(v0);
            v1 = wave_sum_to_lane63(v1);
            v2 = wave_sum_to_lane63(v2);
            v3 = wave_sum_to_lane63(v3);
            v4 = wave_sum_to_lane63(v4);
            v5 = wave_sum_to_lane63(v5);
            v6 = wave_sum_to_lane63(v6);
            v7 = wave_sum_to_lane63(v7);
            v8 = wave_sum_to_lane63(v8);
            v9 = wave_sum_to_lane63(v9);
            if (lane == 63) {
                float* a = acc[j];
                atomicAdd(a + 0, v0);
                atomicAdd(a + 1, v1);
                atomicAdd(a + 2, v2);
                atomicAdd(a + 3, v3);
                atomicAdd(a + 4, v4);
                atomicAdd(a + 5, v5);
                atomicAdd(a + 6, v6);
                atomicAdd(a + 7, v7);
                atomicAdd(a + 8, v8);
                atomicAdd(a + 9, v9);
                hit[j] = 1;
            }
        }
        __syncthreads();
        // flush: 16 lanes per Gaussian, lanes 0..9 carry the row -> row-shaped float atomics
        const int sub = tid & 15;
#pragma unroll 4
        for (int r = 0; r < GS_BLOCK / 16; ++r) {
            const int gidx = r * 16 + (tid >> 4);
            if (gidx < n && sub < 10 && hit[gidx])
                atomicAdd(grad_rows + (size_t)sid[gidx] * GS_GRAD_ROW + sub, acc[gidx][sub]);
        }
    }
}

}  // namespace

extern "C" int gs_render_fwd(int H, int W, const float* bg, const void* splat, const uint32_t* ranges,
                             const uint32_t* point_list, float* out_image, float* out_depth, float* out_alpha,
                             float* aux_rgb, float t_eps, void* stream_) {
    GS_REQUIRE(H > 0 && W > 0 && H < 65536 && W < 65536, "image size %dx%d outside 1..65535", W, H);
    GS_REQUIRE(bg && splat && ranges && out_image && out_depth && out_alpha, "null argument");
    GS_REQUIRE(t_eps >= 0.0f && t_eps < 1.0f, "t_eps must be in [0,1)");
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int ntiles = tiles_x * tiles_y;
    hipStream_t stream = (hipStream_t)stream_;
    if (aux_rgb)
        hipLaunchKernelGGL((render_fwd_kernel<true>), dim3(ntiles), dim3(GS_BLOCK), 0, stream, H, W, tiles_x, ntiles,
                           bg, (const float4*)splat, (const uint2*)ranges, point_list, out_image, out_depth,
                           out_alpha, aux_rgb, t_eps);
    else
        hipLaunchKernelGGL((render_fwd_kernel<false>), dim3(ntiles), dim3(GS_BLOCK), 0, stream, H, W, tiles_x, ntiles,
                           bg, (const float4*)splat, (const uint2*)ranges, point_list, out_image, out_depth,
                           out_alpha, aux_rgb, t_eps);
    GS_CHECK_LAUNCH("render_fwd");
    return GS_OK;
}

// grad_rows (N x 16 floats) must be zero on entry; rows of Gaussians that are not drawn stay zero.
extern "C" int gs_render_bwd(int H, int W, const void* splat, const uint32_t* ranges,
                             const uint32_t* point_list, const float* aux_rgb, const float* out_depth,
                             const float* out_alpha, const float* g_image, const float* g_depth,
                             const float* g_alpha, float* grad_rows, float t_eps, void* stream_) {
    GS_REQUIRE(H > 0 && W > 0 && H < 65536 && W < 65536, "image size %dx%d outside 1..65535", W, H);
    GS_REQUIRE(splat && ranges && aux_rgb && out_depth && out_alpha && g_image && grad_rows, "null argument");
    GS_REQUIRE(t_eps >= 0.0f && t_eps < 1.0f, "t_eps must be in [0,1)");
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int ntiles = tiles_x * tiles_y;
    hipLaunchKernelGGL(render_bwd_kernel, dim3(ntiles), dim3(GS_BLOCK), 0, (hipStream_t)stream_, H, W, tiles_x, ntiles,
                       (const float4*)splat, (const uint2*)ranges, point_list, aux_rgb, out_depth, out_alpha,
                       g_image, g_depth, g_alpha, grad_rows, t_eps);
    GS_CHECK_LAUNCH("render_bwd");
    return GS_OK;
}
