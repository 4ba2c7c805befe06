// Shared definitions for the MI355X (gfx950) Gaussian rasterizer kernels.
//
// Data layout in HBM
// ------------------
// Inputs arrive exactly as the reference's rasterizer takes them
// (reference gaussian_renderer/__init__.py:76-86): means3D (N,3), scales (N,3),
// rotations (N,4), opacities (N,1), shs (N,K,3), colors_precomp (N,3),
// cov3D_precomp (N,6); all fp32, contiguous.
//
// The preprocess kernel packs everything the per-tile kernels need about one
// Gaussian into one 48-byte "splat record" (three float4, 16-byte aligned) so
// that the per-tile gather is three 16-byte loads from one or two cache lines:
//   q0 = (mean_x, mean_y, conic_a, conic_b)
//   q1 = (conic_d, opacity, depth, rect_x)     rect_x = x_min | x_max << 16
//   q2 = (red, green, blue, rect_y)            rect_y = y_min | y_max << 16
// conic = inverse of the 2x2 screen covariance, [[a,b],[b,d]]; rect is the
// reference's hard integer 3-sigma box (:393-399), half-open, clipped to the image.
//
// The backward per-Gaussian gradient rows are GS_GRAD_ROW = 16 floats = 64 bytes, one aligned 64-byte
// segment per Gaussian: six DOUBLES (bytes 0..47) and four floats (bytes 48..63).
//   d[0..4] = sum over the Gaussian's pixels of pv * {dx, dy, dx^2, dx dy, dy^2},  d[5] = sum pv = d loss / d opacity
// with pv = exp(power) * dL/dw and dx, dy measured from the CENTRE OF THE GAUSSIAN'S INTEGER BOX,
// ((x_min + x_max - 1) >> 1, (y_min + y_max - 1) >> 1) -- not from its mean: the box centre is within half a box of
// every pixel the Gaussian draws, so these sums stay small and exact-ish however far off screen the mean lies;
// gs_preprocess_bwd moves them to the mean (q = opacity * pv, dx_mean = dx + (centre - mean)) in double, once per
// Gaussian, and applies the conic factors.  fp64 atomics make the six sums independent of the order in which the
// tiles arrive (bit-reproducible geometry gradients): their chain rule through the 2x2 inverse cancels by orders of
// magnitude for large, nearly singular splats, where float sums in arrival order moved d means3D by ~1e-3 of its value
// from run to run.   f[12..14] = d loss / d rgb,  f[15] = d loss / d depth  (float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GS_TILE 16          // tile edge in pixels (keys, ranges)
#define GS_SUB 8            // one wave64 owns an 8x8 sub-tile
#define GS_BLOCK 256        // threads per tile workgroup = 4 waves
#define GS_SH_JAC_ROW 12     // floats per row of the forward's colour side table (layout: preprocess.hip, include/gsplat.h)
#ifndef GS_GRAD_ROW
#define GS_GRAD_ROW 16      // floats per gradient row (see above; the rows are read as float4s)
#endif

#define GS_OK 0
#define GS_ERR_ARG (-1)
#define GS_ERR_HIP (-2)
#define GS_ERR_SIZE (-3)

void gs_set_error(const char* fmt, ...);

#define GS_CHECK_HIP(expr)                                                          \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            gs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                         __FILE__, __LINE__);                                       \
            return GS_ERR_HIP;                                                      \
        }                                                                           \
    } while (0)

#define GS_CHECK_LAUNCH(name)                                                       \
    do {                                                                            \
        hipError_t _e = hipGetLastError();                                          \
        if (_e != hipSuccess) {                                                     \
            gs_set_error("launch of %s failed: %s", name, hipGetErrorString(_e));   \
            return GS_ERR_HIP;                                                      \
        }                                                                           \
    } while (0)

#define GS_REQUIRE(cond, ...)                                                       \
    do {                                                                            \
        if (!(cond)) {                                                              \
            gs_set_error(__VA_ARGS__);                                              \
            return GS_ERR_ARG;                                                      \
        }                                                                           \
    } while (0)

static inline int gs_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// depth -> 32-bit key whose ascending order is descending depth
__device__ __forceinline__ uint32_t gs_depth_key_desc(float z) {
    z = z + 0.0f;  // -0 -> +0
    uint32_t u = __float_as_uint(z);
    uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}

// Inclusive prefix sum over the 64 lanes of a wave in six DPP adds (shifts by 1, 2, 4, 8 inside every 16-lane row, then
// lane 15 of a row into the next row, lane 31 into the upper half): no LDS crossbar round trips (__shfl_up is a
// ds_bpermute per step -- six dependent ~100-cycle hops per scan in kernels that are latency-bound as it is).
__device__ __forceinline__ uint32_t gs_wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// Tiles overlapped by a packed box (rect_x = x_min | x_max << 16, rect_y likewise; half-open, pixels)
// inside the band of tile rows [row_b, row_e).  An all-zero box (Gaussian not drawn) has no tiles.
struct GsTileSpan { int tx0, tx1, ty0, ty1; };
__device__ __forceinline__ uint32_t gs_tile_span(uint32_t rect_x, uint32_t rect_y, int row_b, int row_e, GsTileSpan& t) {
    const int xmin = rect_x & 0xFFFF, xmax = rect_x >> 16, ymin = rect_y & 0xFFFF, ymax = rect_y >> 16;
    if (xmax <= xmin || ymax <= ymin) { t.tx0 = t.tx1 = t.ty0 = t.ty1 = 0; return 0u; }
    t.tx0 = xmin / GS_TILE; t.tx1 = (xmax - 1) / GS_TILE + 1;
    t.ty0 = ymin / GS_TILE; t.ty1 = (ymax - 1) / GS_TILE + 1;
    t.ty0 = t.ty0 > row_b ? t.ty0 : row_b;
    t.ty1 = t.ty1 < row_e ? t.ty1 : row_e;
    return t.ty1 > t.ty0 ? (uint32_t)((t.tx1 - t.tx0) * (t.ty1 - t.ty0)) : 0u;
}
