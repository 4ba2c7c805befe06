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
// segment per Gaussian.  The five geometric sums are accumulated as DOUBLES (bytes 0..39):
//   d[0] d mean_x  d[1] d mean_y  d[2] d conic_a  d[3] d conic_b (each off-diagonal entry)  d[4] d conic_d
// -- their chain rule through the 2x2 inverse cancels by orders of magnitude for large, nearly singular
// splats, so the order in which the tiles' float atomics arrive used to move d means3D by ~1e-3 of its
// value from run to run; fp64 atomics make the sums order-independent (bit-reproducible gradients) --
// and the rest as floats (bytes 40..59): f[10] d opacity  f[11..13] d rgb  f[14] d depth  f[15] unused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GS_TILE 16          // tile edge in pixels (keys, ranges)
#define GS_SUB 8            // one wave64 owns an 8x8 sub-tile
#define GS_BLOCK 256        // threads per tile workgroup = 4 waves
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
