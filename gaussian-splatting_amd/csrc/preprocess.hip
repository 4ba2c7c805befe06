// Per-Gaussian stage, forward and backward: one lane per Gaussian.
//
// Forward restates (arithmetic of) reference gaussian_renderer/__init__.py:
//   :116-133 homogeneous transforms + NDC -> pixel, :210-264 / :266-289 Sigma,
//   :291-334 J Sigma J^T + 0.3 I, :185-208 + utils/sh_utils.py:40-96 SH colour with the
//   two-sided clamp, and the per-Gaussian tests of splat_gaussians :368-402
//   (+-50 px guard, det < 1e-7, hard integer 3-sigma box).
// Backward is the reverse-mode derivative of those lines (autograd in the reference).
//
// The geometry part is compiled with fp contraction OFF and correctly rounded
// divide/sqrt (hipcc default), in the same operation order as oracle/gs_oracle.c, so the
// integer outputs (rect, radii, tiles_touched) are bit-exact against the oracle.
//
// HBM-bound: algorithmic bytes per Gaussian are 40 (mean, scale, rotation) + 4 (opacity)
// + 12 K (SH) read and 48 (record) + 8 (radii, tiles_touched) written.
#include "gsplat_common.h"
#include "../../include/gsplat.h"

namespace {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f;
constexpr float SH_C2_1 = -1.0925484305920792f;
constexpr float SH_C2_2 = 0.3153915652525205f;
constexpr float SH_C2_3 = -1.0925484305920792f;
constexpr float SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f;
constexpr float SH_C3_1 = 2.890611442640554f;
constexpr float SH_C3_2 = -0.4570457994644658f;
constexpr float SH_C3_3 = 0.3731763325901154f;
constexpr float SH_C3_4 = -0.4570457994644658f;
constexpr float SH_C3_5 = 1.445305721320277f;
constexpr float SH_C3_6 = -0.5900435899266435f;

struct Geom {
    float pos[3];        // the mean as loaded (project_mean)
    float scl[3], quat[4];   // scales and rotation as loaded (project_cov, scale/rotation path)
    float cam[3];
    float clip[4];
    float mx, my;
    float S[3][3];       // Sigma
    float R[3][3];       // rotation (scale/rotation path only)
    float j00, j02, j11, j12;
    float c00, c01, c11;
    float inv_det, ca, cb, cd;
    int xmin, ymin, xmax, ymax;
    int radius;
    uint32_t tiles;
    bool drawn;
};

// All of this must round exactly like oracle/gs_oracle.c::gso_preprocess.
// First half of the projection: camera-space and screen-space mean.  Enough to tell whether the mean lies within 50
// pixels of the frame (the reference's first visibility test), which is what most culled Gaussians fail.
__device__ __forceinline__ void project_mean(int i, int H, int W, const float* __restrict__ view,
                                             const float* __restrict__ proj, const float* __restrict__ means3D, Geom& g) {
#pragma clang fp contract(off)
    const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1], pz = means3D[3 * (size_t)i + 2];
    g.pos[0] = px; g.pos[1] = py; g.pos[2] = pz;
#pragma unroll
    for (int r = 0; r < 3; ++r)
        g.cam[r] = view[4 * r] * px + view[4 * r + 1] * py + view[4 * r + 2] * pz + view[4 * r + 3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        g.clip[r] = proj[4 * r] * px + proj[4 * r + 1] * py + proj[4 * r + 2] * pz + proj[4 * r + 3];
    const float fW = (float)W, fH = (float)H;
    const float wden = g.clip[3] + 1e-7f;
    const float ndcx = g.clip[0] / wden, ndcy = g.clip[1] / wden;
    g.mx = ((ndcx + 1.0f) * fW - 1.0f) * 0.5f;
    g.my = ((ndcy + 1.0f) * fH - 1.0f) * 0.5f;
}

__device__ __forceinline__ bool mean_near_frame(const Geom& g, int H, int W) {
    const float fW = (float)W, fH = (float)H;
    return !(g.mx < -50.0f || g.mx > fW + 50.0f || g.my < -50.0f || g.my > fH + 50.0f);
}

// Second half: 3D covariance, EWA projection, conic, box, radius, tile count.
__device__ __forceinline__ void project_cov(int i, int H, int W, int row_b, int row_e, const float* __restrict__ proj,
                                            const float* __restrict__ scales, const float* __restrict__ rots,
                                            const float* __restrict__ cov6, Geom& g) {
#pragma clang fp contract(off)
    const float fW = (float)W, fH = (float)H;
    const float focal_x = proj[0] * fW / 2.0f;
    const float focal_y = proj[5] * fH / 2.0f;
    if (cov6) {
        const float* c = cov6 + 6 * (size_t)i;
        g.S[0][0] = c[0]; g.S[0][1] = c[1]; g.S[0][2] = c[2];
        g.S[1][0] = c[1]; g.S[1][1] = c[3]; g.S[1][2] = c[4];
        g.S[2][0] = c[2]; g.S[2][1] = c[4]; g.S[2][2] = c[5];
    } else {
        const float4 q = *reinterpret_cast<const float4*>(rots + 4 * (size_t)i);
        g.quat[0] = q.x; g.quat[1] = q.y; g.quat[2] = q.z; g.quat[3] = q.w;
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        g.R[0][0] = 1.0f - 2.0f * (y * y + z * z);
        g.R[0][1] = 2.0f * (x * y - r * z);
        g.R[0][2] = 2.0f * (x * z + r * y);
        g.R[1][0] = 2.0f * (x * y + r * z);
        g.R[1][1] = 1.0f - 2.0f * (x * x + z * z);
        g.R[1][2] = 2.0f * (y * z - r * x);
        g.R[2][0] = 2.0f * (x * z - r * y);
        g.R[2][1] = 2.0f * (y * z + r * x);
        g.R[2][2] = 1.0f - 2.0f * (x * x + y * y);
        const float s0 = scales[3 * (size_t)i], s1 = scales[3 * (size_t)i + 1], s2 = scales[3 * (size_t)i + 2];
        g.scl[0] = s0; g.scl[1] = s1; g.scl[2] = s2;
        float M[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            M[a][0] = g.R[a][0] * s0;
            M[a][1] = g.R[a][1] * s1;
            M[a][2] = g.R[a][2] * s2;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                g.S[a][b] = M[a][0] * M[b][0] + M[a][1] * M[b][1] + M[a][2] * M[b][2];
    }
    const float x = g.cam[0], y = g.cam[1], z = g.cam[2];
    g.j00 = focal_x / (z + 1e-7f);
    g.j02 = -x * focal_x / (z * z + 1e-7f);
    g.j11 = focal_y / (z + 1e-7f);
    g.j12 = -y * focal_y / (z * z + 1e-7f);
    const float t00 = g.j00 * g.S[0][0] + g.j02 * g.S[2][0];
    const float t01 = g.j00 * g.S[0][1] + g.j02 * g.S[2][1];
    const float t02 = g.j00 * g.S[0][2] + g.j02 * g.S[2][2];
    const float t11 = g.j11 * g.S[1][1] + g.j12 * g.S[2][1];
    const float t12 = g.j11 * g.S[1][2] + g.j12 * g.S[2][2];
    g.c00 = (t00 * g.j00 + t02 * g.j02) + 0.3f;
    g.c01 = t01 * g.j11 + t02 * g.j12;
    g.c11 = (t11 * g.j11 + t12 * g.j12) + 0.3f;

    bool drawn = mean_near_frame(g, H, W);
    const float det = g.c00 * g.c11 - g.c01 * g.c01;
    if (det < 1e-7f) drawn = false;
    g.inv_det = 1.0f / det;
    g.ca = g.c11 * g.inv_det;
    g.cb = -g.c01 * g.inv_det;
    g.cd = g.c00 * g.inv_det;
    const float sx = sqrtf(g.c00) * 3.0f, sy = sqrtf(g.c11) * 3.0f;
    float lo, hi;
    lo = g.mx - sx;        lo = (lo > 0.0f) ? lo : 0.0f;
    hi = g.mx + sx + 1.0f; hi = (hi < fW) ? hi : fW;
    g.xmin = (int)lo; g.xmax = (int)hi;
    lo = g.my - sy;        lo = (lo > 0.0f) ? lo : 0.0f;
    hi = g.my + sy + 1.0f; hi = (hi < fH) ? hi : fH;
    g.ymin = (int)lo; g.ymax = (int)hi;
    if (g.xmin >= g.xmax || g.ymin >= g.ymax) drawn = false;
    g.drawn = drawn;
    if (!drawn) {
        g.xmin = g.ymin = g.xmax = g.ymax = 0;
        g.radius = 0;
        g.tiles = 0;
    } else {
        const float m = (sx > sy) ? sx : sy;
        g.radius = (m >= 1.0f && m < 32767.0f) ? (int)ceilf(m) : (m >= 32767.0f ? 32767 : 1);
        const int tx0 = g.xmin / GS_TILE, tx1 = (g.xmax - 1) / GS_TILE + 1;
        int ty0 = g.ymin / GS_TILE, ty1 = (g.ymax - 1) / GS_TILE + 1;
        ty0 = ty0 > row_b ? ty0 : row_b;      // band of tile rows owned by this call (whole frame: 0..tiles_y)
        ty1 = ty1 < row_e ? ty1 : row_e;
        g.tiles = ty1 > ty0 ? (uint32_t)((tx1 - tx0) * (ty1 - ty0)) : 0u;
    }
}

__device__ __forceinline__ void project_gaussian(int i, int H, int W, int row_b, int row_e, const float* __restrict__ view,
                                                 const float* __restrict__ proj,
                                                 const float* __restrict__ means3D,
                                                 const float* __restrict__ scales,
                                                 const float* __restrict__ rots,
                                                 const float* __restrict__ cov6, Geom& g) {
    project_mean(i, H, W, view, proj, means3D, g);
    project_cov(i, H, W, row_b, row_e, proj, scales, rots, cov6, g);
}

// Contraction is off here and where the colour is summed: whether a product feeds an fma depends on how many uses it has
// after inlining, and the instantiations (with / without the side table, forward / backward) must all produce the same
// colour bit for bit -- the image of a forward with and without a backward to follow, and the clamp mask.
template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* b) {
#pragma clang fp contract(off)
    b[0] = SH_C0;
    if (DEG > 0) {
        b[1] = -SH_C1 * y;
        b[2] = SH_C1 * z;
        b[3] = -SH_C1 * x;
    }
    if (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = SH_C2_0 * xy;
        b[5] = SH_C2_1 * yz;
        b[6] = SH_C2_2 * (2.0f * zz - xx - yy);
        b[7] = SH_C2_3 * xz;
        b[8] = SH_C2_4 * (xx - yy);
        if (DEG > 2) {
            b[9] = SH_C3_0 * y * (3.0f * xx - yy);
            b[10] = SH_C3_1 * xy * z;
            b[11] = SH_C3_2 * y * (4.0f * zz - xx - yy);
            b[12] = SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            b[13] = SH_C3_4 * x * (4.0f * zz - xx - yy);
            b[14] = SH_C3_5 * z * (xx - yy);
            b[15] = SH_C3_6 * x * (xx - 3.0f * yy);
        }
    }
}

// d basis_k / d(x,y,z)
template <int DEG>
__device__ __forceinline__ void sh_basis_grad(float x, float y, float z, float (*g)[3]) {
    g[0][0] = g[0][1] = g[0][2] = 0.0f;
    if (DEG > 0) {
        g[1][0] = 0; g[1][1] = -SH_C1; g[1][2] = 0;
        g[2][0] = 0; g[2][1] = 0; g[2][2] = SH_C1;
        g[3][0] = -SH_C1; g[3][1] = 0; g[3][2] = 0;
    }
    if (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        g[4][0] = SH_C2_0 * y; g[4][1] = SH_C2_0 * x; g[4][2] = 0;
        g[5][0] = 0; g[5][1] = SH_C2_1 * z; g[5][2] = SH_C2_1 * y;
        g[6][0] = SH_C2_2 * -2.0f * x; g[6][1] = SH_C2_2 * -2.0f * y; g[6][2] = SH_C2_2 * 4.0f * z;
        g[7][0] = SH_C2_3 * z; g[7][1] = 0; g[7][2] = SH_C2_3 * x;
        g[8][0] = SH_C2_4 * 2.0f * x; g[8][1] = SH_C2_4 * -2.0f * y; g[8][2] = 0;
        if (DEG > 2) {
            g[9][0] = SH_C3_0 * 6.0f * xy; g[9][1] = SH_C3_0 * (3.0f * xx - 3.0f * yy); g[9][2] = 0;
            g[10][0] = SH_C3_1 * yz; g[10][1] = SH_C3_1 * xz; g[10][2] = SH_C3_1 * xy;
            g[11][0] = SH_C3_2 * -2.0f * xy; g[11][1] = SH_C3_2 * (4.0f * zz - xx - 3.0f * yy);
            g[11][2] = SH_C3_2 * 8.0f * yz;
            g[12][0] = SH_C3_3 * -6.0f * xz; g[12][1] = SH_C3_3 * -6.0f * yz;
            g[12][2] = SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
            g[13][0] = SH_C3_4 * (4.0f * zz - 3.0f * xx - yy); g[13][1] = SH_C3_4 * -2.0f * xy;
            g[13][2] = SH_C3_4 * 8.0f * xz;
            g[14][0] = SH_C3_5 * 2.0f * xz; g[14][1] = SH_C3_5 * -2.0f * yz; g[14][2] = SH_C3_5 * (xx - yy);
            g[15][0] = SH_C3_6 * (3.0f * xx - 3.0f * yy); g[15][1] = SH_C3_6 * -6.0f * xy; g[15][2] = 0;
        }
    }
}

// W[a][c] = sum_k (d basis_k / d dir_a) sh_kc for the coefficient row sh (k-major triples): sh_basis_grad contracted with
// the coefficients, written out term by term so that the 15 structurally zero derivatives cost nothing and no 16 x 3
// table has to live in registers next to the 48 coefficients.  (Register note for this file: the build passes
// -fno-slp-vectorize -- packing these scalar chains into v_pk_fma_f32 gains nothing on gfx950, where the packed form
// issues at half rate, and its shuffles cost the forward kernel 138 VGPRs instead of 58.)
template <int DEG>
__device__ __forceinline__ void sh_dir_jacobian(float x, float y, float z, const float* sh, float (*Wm)[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) Wm[a][0] = Wm[a][1] = Wm[a][2] = 0.0f;
    auto add = [&](int a, int k, float g) {
        Wm[a][0] += g * sh[3 * k];
        Wm[a][1] += g * sh[3 * k + 1];
        Wm[a][2] += g * sh[3 * k + 2];
    };
    if (DEG > 0) {
        add(1, 1, -SH_C1);
        add(2, 2, SH_C1);
        add(0, 3, -SH_C1);
    }
    if (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        add(0, 4, SH_C2_0 * y); add(1, 4, SH_C2_0 * x);
        add(1, 5, SH_C2_1 * z); add(2, 5, SH_C2_1 * y);
        add(0, 6, SH_C2_2 * -2.0f * x); add(1, 6, SH_C2_2 * -2.0f * y); add(2, 6, SH_C2_2 * 4.0f * z);
        add(0, 7, SH_C2_3 * z); add(2, 7, SH_C2_3 * x);
        add(0, 8, SH_C2_4 * 2.0f * x); add(1, 8, SH_C2_4 * -2.0f * y);
        if (DEG > 2) {
            add(0, 9, SH_C3_0 * 6.0f * xy); add(1, 9, SH_C3_0 * (3.0f * xx - 3.0f * yy));
            add(0, 10, SH_C3_1 * yz); add(1, 10, SH_C3_1 * xz); add(2, 10, SH_C3_1 * xy);
            add(0, 11, SH_C3_2 * -2.0f * xy); add(1, 11, SH_C3_2 * (4.0f * zz - xx - 3.0f * yy)); add(2, 11, SH_C3_2 * 8.0f * yz);
            add(0, 12, SH_C3_3 * -6.0f * xz); add(1, 12, SH_C3_3 * -6.0f * yz);
            add(2, 12, SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy));
            add(0, 13, SH_C3_4 * (4.0f * zz - 3.0f * xx - yy)); add(1, 13, SH_C3_4 * -2.0f * xy); add(2, 13, SH_C3_4 * 8.0f * xz);
            add(0, 14, SH_C3_5 * 2.0f * xz); add(1, 14, SH_C3_5 * -2.0f * yz); add(2, 14, SH_C3_5 * (xx - yy));
            add(0, 15, SH_C3_6 * (3.0f * xx - 3.0f * yy)); add(1, 15, SH_C3_6 * -6.0f * xy);
        }
    }
}

// loads the first NB coefficient triples of Gaussian i; VEC4: rows are 16-byte aligned multiples of 16 B
template <int NB, bool VEC4>
__device__ __forceinline__ void load_sh(const float* __restrict__ shs, int i, int K, float* sh) {
    const float* p = shs + (size_t)i * K * 3;
    if (VEC4) {
        constexpr int NV = (NB * 3 + 3) / 4;
        const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 t = p4[v];
            if (4 * v + 0 < NB * 3) sh[4 * v + 0] = t.x;
            if (4 * v + 1 < NB * 3) sh[4 * v + 1] = t.y;
            if (4 * v + 2 < NB * 3) sh[4 * v + 2] = t.z;
            if (4 * v + 3 < NB * 3) sh[4 * v + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NB * 3; ++k) sh[k] = p[k];
    }
}

// Layout of the optional colour side table sh_jac: one 48-byte row (three float4) per Gaussian (zeros for Gaussians that
// are not drawn; only rows of drawn Gaussians are read):
//   floats 0..8  W[a][c] = sum_k (d basis_k / d dir_a) sh_kc at 3 a + c  (a = x, y, z of the unit view direction; c = r, g, b)
//   word   9     bit c set: channel c was inside the clamp [0, 1] (its gradient passes, gaussian_renderer/__init__.py:206)
//   words 10, 11 unused (row padded to whole 16-byte words)
// With it the backward needs no second pass over the N x K x 3 coefficients: d colour / d dir is W, d/d sh is basis x
// colour gradient.  Written by the forward in the order the backward's own pass would have summed it.
static_assert(GS_SH_JAC_ROW == 12, "nine Jacobian entries + the clamp mask, padded to three float4");

// One view of one Gaussian: everything preprocess_fwd_kernel does, shared with the multi-view kernel below.
// sh_h / opacity_i / requested: the coefficients and the opacity of this Gaussian and whether they have been requested
// already (a multi-view caller keeps them across its views: they are read once).  st: this wave's 3 KB of LDS.
template <int DEG, bool VEC4, bool JAC>
__device__ __forceinline__ void fwd_one_view(
    const int i, const int lane, int N, int K, int H, int W, int row_b, int row_e, const float* __restrict__ view,
    const float* __restrict__ proj, const float* __restrict__ campos, const float* __restrict__ means3D,
    const float* __restrict__ scales, const float* __restrict__ rots, const float* __restrict__ cov6,
    const float* __restrict__ opacities, const float* __restrict__ shs, const float* __restrict__ colors_precomp,
    float4* __restrict__ splat, int32_t* __restrict__ radii, uint32_t* __restrict__ tiles_touched,
    uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ ids, uint2* __restrict__ rects,
    float4* __restrict__ sh_jac, float4* st, float* sh_h, float& opacity_i, bool& requested) {
    const bool in_range = i < N;
    bool drawn = false;
    Geom g;
    // The SH coefficients (192 bytes per Gaussian at degree 3, half of what this kernel reads) are requested as soon as the
    // screen-space mean says the Gaussian can be visible, BEFORE the covariance chain, so that they travel while it is
    // computed (issued after it, the kernel paid two memory latencies in a row: 0.222 -> 0.203 ms).
    constexpr int NBH = (DEG + 1) * (DEG + 1);
    if (in_range) {
        project_mean(i, H, W, view, proj, means3D, g);
        if (!requested && mean_near_frame(g, H, W)) {
            if (!colors_precomp) load_sh<NBH, VEC4>(shs, i, K, sh_h);
            opacity_i = opacities[i];      // only copied into the record at the very end: loaded there, it was one more wait
            requested = true;
        }
        project_cov(i, H, W, row_b, row_e, proj, scales, rots, cov6, g);
        drawn = g.drawn;
        radii[i] = g.radius;
        tiles_touched[i] = g.tiles;
        if (depth_keys) {   // input of the per-Gaussian depth sort; culled Gaussians sort last
            depth_keys[i] = g.drawn ? gs_depth_key_desc(g.cam[2]) : 0xFFFFFFFFu;
            ids[i] = (uint32_t)i;
            // compact copy of the box (all zero when not drawn) for the binning stage's one gather
            rects[i] = make_uint2((uint32_t)g.xmin | ((uint32_t)g.xmax << 16), (uint32_t)g.ymin | ((uint32_t)g.ymax << 16));
        }
    }
    const int i0 = i - lane;                      // first Gaussian of this wave
    // Rows of culled Gaussians are written too, as zeros: skipping them leaves 128-byte lines partly written, and a
    // partial line costs the memory system far more than the 48 bytes saved (measured with 15 % of the rows culled:
    // 0.246 -> 0.215 ms).  float4 m of the wave's chunk belongs to Gaussian i0 + m / 3.
    const auto write_rows = [&](float4* __restrict__ dst) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int m = r * 64 + lane;
            if (i0 + m / 3 < N) dst[3 * (size_t)i0 + m] = st[m];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    float rgb[3] = {0.f, 0.f, 0.f};
    if (!drawn) st[3 * lane] = st[3 * lane + 1] = st[3 * lane + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (drawn) {
    if (colors_precomp) {
        rgb[0] = colors_precomp[3 * (size_t)i];
        rgb[1] = colors_precomp[3 * (size_t)i + 1];
        rgb[2] = colors_precomp[3 * (size_t)i + 2];
    } else {
        constexpr int NB = (DEG + 1) * (DEG + 1);
        float* sh = sh_h;       // loaded above (a drawn Gaussian has passed mean_near_frame)
        float b[NB];
        float dx = g.pos[0] - campos[0];     // the mean as project_mean loaded it (a second load here was a second wait)
        float dy = g.pos[1] - campos[1];
        float dz = g.pos[2] - campos[2];
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz) + 1e-7f;
        dx = dx / nrm; dy = dy / nrm; dz = dz / nrm;
        sh_basis<DEG>(dx, dy, dz, b);
        uint32_t pass = 0;
        float v[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < NB; ++k) {     // per channel the same chain as a loop over k inside a loop over c
#pragma clang fp contract(off)             // one rounding sequence in every instantiation: see sh_basis
            v[0] += b[k] * sh[3 * k];
            v[1] += b[k] * sh[3 * k + 1];
            v[2] += b[k] * sh[3 * k + 2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float vc = v[c] + 0.5f;
            rgb[c] = fminf(fmaxf(vc, 0.0f), 1.0f);
            if (vc >= 0.0f && vc <= 1.0f) pass |= 1u << c;
        }
        if (JAC) {
            float Wm[3][3];
            sh_dir_jacobian<DEG>(dx, dy, dz, sh, Wm);
            st[3 * lane] = make_float4(Wm[0][0], Wm[0][1], Wm[0][2], Wm[1][0]);
            st[3 * lane + 1] = make_float4(Wm[1][1], Wm[1][2], Wm[2][0], Wm[2][1]);
            st[3 * lane + 2] = make_float4(Wm[2][2], __uint_as_float(pass), 0.0f, 0.0f);
        }
    }
    }
    if (JAC) write_rows(sh_jac);
    if (drawn) {
        const uint32_t rx = (uint32_t)g.xmin | ((uint32_t)g.xmax << 16);
        const uint32_t ry = (uint32_t)g.ymin | ((uint32_t)g.ymax << 16);
        st[3 * lane] = make_float4(g.mx, g.my, g.ca, g.cb);
        st[3 * lane + 1] = make_float4(g.cd, opacity_i, g.cam[2], __uint_as_float(rx));
        st[3 * lane + 2] = make_float4(rgb[0], rgb[1], rgb[2], __uint_as_float(ry));
    }
    write_rows(splat);
}

template <int DEG, bool VEC4, bool JAC>
__global__ __launch_bounds__(256) void preprocess_fwd_kernel(
    int N, int K, int H, int W, int row_b, int row_e, const float* __restrict__ view, const float* __restrict__ proj,
    const float* __restrict__ campos, const float* __restrict__ means3D,
    const float* __restrict__ scales, const float* __restrict__ rots,
    const float* __restrict__ cov6, const float* __restrict__ opacities,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp,
    float4* __restrict__ splat, int32_t* __restrict__ radii, uint32_t* __restrict__ tiles_touched,
    uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ ids, uint2* __restrict__ rects,
    float4* __restrict__ sh_jac) {
    // Output rows (48-byte records, 48-byte side-table rows) leave through LDS: a wave's 64 rows are 3 KB of
    // consecutive memory, written as three 1 KB store instructions of whole lines.  One 16-byte store per lane and
    // row word instead (48-byte lane stride) leaves every 128-byte line in three partial pieces, which cost this
    // HBM-bound kernel 0.18 -> 0.30 ms once a second table was written that way.
    __shared__ float4 out_stage[4][64 * 3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float sh_h[(DEG + 1) * (DEG + 1) * 3 + 3];
    float opacity_i = 0.0f;
    bool requested = false;
    fwd_one_view<DEG, VEC4, JAC>(i, threadIdx.x & 63, N, K, H, W, row_b, row_e, view, proj, campos, means3D, scales, rots, cov6,
                                 opacities, shs, colors_precomp, splat, radii, tiles_touched, depth_keys, ids, rects, sh_jac,
                                 out_stage[threadIdx.x >> 6], sh_h, opacity_i, requested);
}

// k views of the same Gaussians in one launch (the forward of the multi-view training step): the
// 192 bytes of SH coefficients per Gaussian -- half of what the single-view kernel reads -- the mean, scale, rotation and
// opacity are read ONCE and kept in registers over the views; per view the lane writes what the single-view kernel writes.
struct FwdView {
    const float* view;
    const float* proj;
    const float* campos;
    float4* splat;
    int32_t* radii;
    uint32_t* tiles_touched;
    uint32_t* depth_keys;
    uint32_t* ids;
    uint2* rects;
    float4* sh_jac;
    int H, W, row_b, row_e;
};
struct FwdViews {
    FwdView v[GS_MAX_FWD_VIEWS];
};

template <int DEG, bool JAC>
__global__ __launch_bounds__(256) void preprocess_fwd_views_kernel(
    int N, int K, int nv, const FwdViews views, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rots, const float* __restrict__ cov6, const float* __restrict__ opacities,
    const float* __restrict__ shs) {
    __shared__ float4 out_stage[4][64 * 3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float sh_h[(DEG + 1) * (DEG + 1) * 3 + 3];
    float opacity_i = 0.0f;
    bool requested = false;
    for (int v = 0; v < nv; ++v) {
        const FwdView& w = views.v[v];
        fwd_one_view<DEG, true, JAC>(i, threadIdx.x & 63, N, K, w.H, w.W, w.row_b, w.row_e, w.view, w.proj, w.campos, means3D,
                                     scales, rots, cov6, opacities, shs, nullptr, w.splat, w.radii, w.tiles_touched,
                                     w.depth_keys, w.ids, w.rects, w.sh_jac, out_stage[threadIdx.x >> 6], sh_h, opacity_i,
                                     requested);
    }
}

// MODE 0: scalar SH access; 1: 16-byte SH loads/stores per lane (rows 16-byte aligned);
// 2: like 1 but staged through LDS so that HBM sees the wave's 64 consecutive SH rows as one
//    contiguous stream of full 16-byte-per-lane requests (rows of at most 48 floats).
constexpr int BWD_STAGE_MAX_ROW = 48;
constexpr int BWD_STAGE_STRIDE = BWD_STAGE_MAX_ROW + 1;   // odd stride: per-lane rows hit all banks

// What one view contributes to one drawn Gaussian's gradients (everything but the SH coefficients' own gradient, which
// goes to st_row / gsh_out).  Shared by the single-view kernel and the multi-view kernel below.
struct BwdGrads {
    float gm[3], gm2[2], gs[3], gq[4], gc6[6], gop, gcol[3];
};

// What bwd_one_view reads of one (Gaussian, view) besides the Gaussian itself; requested by bwd_request, apart from the
// chain rule, so that a caller can have the next view's values travel while this view's are used.
struct BwdInputs {
    float4 r0, r1, r2, r3;     // the gradient row: 6 doubles (moments of pv about the box centre) + floats 12..15 (rgb, depth): gsplat_common.h
    float4 j0, j1, j2;         // the colour side-table row (JAC)
    float opacity;
};

template <int PART, bool JAC>
__device__ __forceinline__ BwdInputs bwd_request(int i, const float* __restrict__ opacities,
                                                 const float4* __restrict__ grad_rows, const float4* __restrict__ sh_jac) {
    static_assert(GS_GRAD_ROW == 16, "row layout");
    const float4* row = grad_rows + (GS_GRAD_ROW / 4) * (size_t)i;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    BwdInputs in;
    in.r0 = PART != 2 ? row[0] : z4; in.r1 = PART != 2 ? row[1] : z4; in.r2 = PART != 2 ? row[2] : z4; in.r3 = row[3];
    in.opacity = PART != 2 ? opacities[i] : 0.0f;
    in.j0 = in.j1 = in.j2 = z4;
    if (JAC) {
        const float4* jp = sh_jac + 3 * (size_t)i;
        in.j0 = jp[0]; in.j1 = jp[1]; in.j2 = jp[2];
    }
    return in;
}

// STAGE_ADD (staged form only): st_row holds the sum over the views before this one; add to it instead of overwriting.
// Returns the number of SH gradient floats written for this Gaussian.
template <int DEG, int MODE, int PART, bool JAC, bool STAGE_ADD>
__device__ __forceinline__ int bwd_one_view(const BwdInputs& in, 
    int i, int K, int H, int W, const float* __restrict__ view, const float* __restrict__ proj,
    const float* __restrict__ campos, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rots, const float* __restrict__ cov6, const float* __restrict__ shs,
    const float* __restrict__ colors_precomp, float* st_row, float* gsh_out, bool acc, BwdGrads& o) {
    constexpr bool VEC4 = MODE >= 1;
    constexpr bool STAGE = MODE == 2;
    constexpr int NB = (DEG + 1) * (DEG + 1);
    float (&gm)[3] = o.gm;
    float (&gm2)[2] = o.gm2;
    float (&gs)[3] = o.gs;
    float (&gq)[4] = o.gq;
    float (&gc6)[6] = o.gc6;
    float& gop = o.gop;
    float (&gcol)[3] = o.gcol;
    int sh_written = 0;
    const float4 r0 = in.r0, r1 = in.r1, r2 = in.r2, r3 = in.r3, j0 = in.j0, j1 = in.j1, j2 = in.j2;
    const float opacity_i = in.opacity;
    Geom g;
    if (PART != 2) project_gaussian(i, H, W, 0, 1 << 20, view, proj, means3D, scales, rots, cov6, g);
    const auto dbl = [](float lo, float hi) {
        return __hiloint2double((int)__float_as_uint(hi), (int)__float_as_uint(lo));
    };
    float d_mx = 0.f, d_my = 0.f, d_a = 0.f, d_b = 0.f, d_d = 0.f, d_op = 0.f;
    const float d_r = r3.x, d_g = r3.y, d_bl = r3.z, d_z = r3.w;
    if (PART != 2) {
        // moments of q = opacity * pv about the mean, from the moments of pv about the box centre (double: the
        // shift can be thousands of pixels for a splat whose mean is far off screen)
        const double op = (double)opacity_i;
        const double M0 = dbl(r2.z, r2.w);
        const double ex = (double)((g.xmin + g.xmax - 1) >> 1) - (double)g.mx;
        const double ey = (double)((g.ymin + g.ymax - 1) >> 1) - (double)g.my;
        const double S1x = dbl(r0.x, r0.y), S1y = dbl(r0.z, r0.w);
        const double Sx = op * (ex * M0 + S1x), Sy = op * (ey * M0 + S1y);
        const double Sxx = op * (ex * (ex * M0 + 2.0 * S1x) + dbl(r1.x, r1.y));
        const double Sxy = op * (ex * (ey * M0 + S1y) + ey * S1x + dbl(r1.z, r1.w));
        const double Syy = op * (ey * (ey * M0 + 2.0 * S1y) + dbl(r2.x, r2.y));
        // power = -0.5 (a dx^2 + 2 b dx dy + d dy^2), dx = pixel - mean (reference :417-424)
        d_mx = (float)((double)g.ca * Sx + (double)g.cb * Sy);
        d_my = (float)((double)g.cd * Sy + (double)g.cb * Sx);
        d_a = (float)(-0.5 * Sxx);
        d_b = (float)(-0.5 * Sxy);
        d_d = (float)(-0.5 * Syy);
        d_op = (float)M0;
    }
    gop = d_op;
    gm2[0] = d_mx;
    gm2[1] = d_my;
    float px, py, pz;            // (the projection has loaded the mean already: a second load here was a second wait)
    if (PART != 2) { px = g.pos[0]; py = g.pos[1]; pz = g.pos[2]; }
    else { px = means3D[3 * (size_t)i]; py = means3D[3 * (size_t)i + 1]; pz = means3D[3 * (size_t)i + 2]; }
    // ---- colour
    if (colors_precomp) {
        gcol[0] = d_r; gcol[1] = d_g; gcol[2] = d_bl;
    } else if (shs && PART != 1) {
        float b[NB];
        const float vx = px - campos[0], vy = py - campos[1], vz = pz - campos[2];
        const float n = sqrtf(vx * vx + vy * vy + vz * vz);
        const float ne = n + 1e-7f;
        const float dx = vx / ne, dy = vy / ne, dz = vz / ne;
        sh_basis<DEG>(dx, dy, dz, b);
        const float dcol[3] = {d_r, d_g, d_bl};
        float gcl[3];
        float ddir[3] = {0.f, 0.f, 0.f};
        if (JAC) {
            // W = d colour / d dir and the clamp mask as the forward left them
            const float Wm[3][3] = {{j0.x, j0.y, j0.z}, {j0.w, j1.x, j1.y}, {j1.z, j1.w, j2.x}};
            const uint32_t pass = __float_as_uint(j2.y);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gcl[c] = ((pass >> c) & 1u) ? dcol[c] : 0.0f;
#pragma unroll
                for (int a2 = 0; a2 < 3; ++a2) ddir[a2] += Wm[a2][c] * gcl[c];
            }
        } else {
        float bg[NB][3];
        sh_basis_grad<DEG>(dx, dy, dz, bg);
        // one streaming pass over the coefficients (each loaded value is consumed at once, so the
        // 48 floats never sit in registers together): colour v[c] = sum_k b_k sh_kc and
        // Wm[a][c] = sum_k (d b_k / d dir_a) sh_kc; the clamp mask is applied afterwards
        float vcol[3] = {0.f, 0.f, 0.f};
        float Wm[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        {
            const float* p = shs + (size_t)i * K * 3;
            constexpr int NF = NB * 3;
            constexpr int NV = (VEC4 && !STAGE) ? (NF + 3) / 4 : 0;
            const float4* p4 = reinterpret_cast<const float4*>(p);
            float4 t4[NV > 0 ? NV : 1];
#pragma unroll
            for (int v = 0; v < NV; ++v) t4[v] = p4[v];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#pragma clang fp contract(off)     // the colour must round as in the forward (clamp mask)
                float val;
                if (STAGE) {
                    val = st_row[f];
                } else if (VEC4) {
                    const float4 t = t4[f / 4];
                    val = (f % 4 == 0) ? t.x : (f % 4 == 1) ? t.y : (f % 4 == 2) ? t.z : t.w;
                } else {
                    val = p[f];
                }
                const int k = f / 3, c = f % 3;
                vcol[c] += b[k] * val;
                Wm[0][c] += bg[k][0] * val;
                Wm[1][c] += bg[k][1] * val;
                Wm[2][c] += bg[k][2] * val;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = vcol[c] + 0.5f;
            gcl[c] = (v >= 0.0f && v <= 1.0f) ? dcol[c] : 0.0f;
#pragma unroll
            for (int a2 = 0; a2 < 3; ++a2) ddir[a2] += Wm[a2][c] * gcl[c];
        }
        }
        if (gsh_out && STAGE) {
#pragma unroll
            for (int f = 0; f < NB * 3; ++f) st_row[f] = b[f / 3] * gcl[f % 3] + (STAGE_ADD ? st_row[f] : 0.0f);
            sh_written = NB * 3;
        } else if (gsh_out) {
            if (VEC4) {   // 16-byte stores: a quarter of the partial-line write requests
                constexpr int NV = (NB * 3) / 4;      // whole float4s; the 0..3 floats left go scalar
                float4* o4 = reinterpret_cast<float4*>(gsh_out);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = b[(4 * v + e) / 3] * gcl[(4 * v + e) % 3];
                    if (acc) {
                        const float4 o = o4[v];
                        t[0] += o.x; t[1] += o.y; t[2] += o.z; t[3] += o.w;
                    }
                    o4[v] = make_float4(t[0], t[1], t[2], t[3]);
                }
#pragma unroll
                for (int f = 4 * NV; f < NB * 3; ++f) gsh_out[f] = b[f / 3] * gcl[f % 3] + (acc ? gsh_out[f] : 0.0f);
            } else {
#pragma unroll
                for (int f = 0; f < NB * 3; ++f) gsh_out[f] = b[f / 3] * gcl[f % 3] + (acc ? gsh_out[f] : 0.0f);
            }
            sh_written = NB * 3;
        }
        if (n > 0.0f) {
            const float dot = ddir[0] * vx + ddir[1] * vy + ddir[2] * vz;
            const float k2 = dot / (ne * ne * n);
            gm[0] += ddir[0] / ne - vx * k2;
            gm[1] += ddir[1] / ne - vy * k2;
            gm[2] += ddir[2] / ne - vz * k2;
        } else {
            gm[0] += ddir[0] / ne; gm[1] += ddir[1] / ne; gm[2] += ddir[2] / ne;
        }
    }
    if (PART != 2) {
    // ---- screen mean
    {
        const float wden = g.clip[3] + 1e-7f;
        const float dnx = 0.5f * (float)W * d_mx, dny = 0.5f * (float)H * d_my;
        const float dcx = dnx / wden, dcy = dny / wden;
        const float dcw = -(dnx * g.clip[0] + dny * g.clip[1]) / (wden * wden);
#pragma unroll
        for (int c = 0; c < 3; ++c) gm[c] += proj[c] * dcx + proj[4 + c] * dcy + proj[12 + c] * dcw;
    }
    // ---- conic -> cov2D
    const float d_inv_det = d_a * g.c11 - 2.0f * d_b * g.c01 + d_d * g.c00;
    const float d_det = -g.inv_det * g.inv_det * d_inv_det;
    float G2[2][2];
    G2[0][0] = d_d * g.inv_det + d_det * g.c11;
    G2[1][1] = d_a * g.inv_det + d_det * g.c00;
    G2[0][1] = -d_b * g.inv_det - d_det * g.c01;
    G2[1][0] = G2[0][1];
    // ---- cov2D = J Sigma J^T
    const float J[2][3] = {{g.j00, 0.0f, g.j02}, {0.0f, g.j11, g.j12}};
    float dS[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) v += J[r][a] * G2[r][c] * J[c][b2];
            dS[a][b2] = v;
        }
    float JS[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            JS[r][c] = J[r][0] * g.S[0][c] + J[r][1] * g.S[1][c] + J[r][2] * g.S[2][c];
    float dJ[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            dJ[r][c] = (G2[r][0] + G2[0][r]) * JS[0][c] + (G2[r][1] + G2[1][r]) * JS[1][c];
    const float focal_x = proj[0] * (float)W / 2.0f, focal_y = proj[5] * (float)H / 2.0f;
    const float x = g.cam[0], y = g.cam[1], z = g.cam[2];
    const float zi = z + 1e-7f, z2 = z * z + 1e-7f;
    float dcam[3];
    dcam[0] = dJ[0][2] * (-focal_x / z2);
    dcam[1] = dJ[1][2] * (-focal_y / z2);
    dcam[2] = dJ[0][0] * (-focal_x / (zi * zi)) + dJ[1][1] * (-focal_y / (zi * zi)) +
              dJ[0][2] * (x * focal_x * 2.0f * z / (z2 * z2)) +
              dJ[1][2] * (y * focal_y * 2.0f * z / (z2 * z2)) + d_z;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        gm[c] += view[c] * dcam[0] + view[4 + c] * dcam[1] + view[8 + c] * dcam[2];
    if (cov6) {
        gc6[0] = dS[0][0]; gc6[1] = dS[0][1] + dS[1][0]; gc6[2] = dS[0][2] + dS[2][0];
        gc6[3] = dS[1][1]; gc6[4] = dS[1][2] + dS[2][1]; gc6[5] = dS[2][2];
    } else {
        const float s[3] = {g.scl[0], g.scl[1], g.scl[2]};     // as the projection loaded them: no second load, no second wait
        float dR[3][3];
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) {
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) v += (dS[a][k] + dS[k][a]) * (g.R[k][b2] * s[b2]);
                acc += v * g.R[a][b2];
                dR[a][b2] = v * s[b2];
            }
            gs[b2] = acc;
        }
        const float r = g.quat[0], qx = g.quat[1], qy = g.quat[2], qz = g.quat[3];
        gq[0] = 2.0f * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
        gq[1] = 2.0f * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2.0f * qx * dR[1][1] - r * dR[1][2] + qz * dR[2][0] + r * dR[2][1] - 2.0f * qx * dR[2][2]);
        gq[2] = 2.0f * (-2.0f * qy * dR[0][0] + qx * dR[0][1] + r * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] - r * dR[2][0] + qz * dR[2][1] - 2.0f * qy * dR[2][2]);
        gq[3] = 2.0f * (-2.0f * qz * dR[0][0] - r * dR[0][1] + qx * dR[0][2] + r * dR[1][0] - 2.0f * qz * dR[1][1] + qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
    }
    }   // PART != 2
    return sh_written;
}

// PART 0: everything in one launch.  The SH degree-3 part needs ~180 VGPRs next to the geometry
// (2 waves per SIMD: latency-bound), so the staged mode runs as two launches instead:
// PART 1 = everything but the SH colours (126 VGPRs), PART 2 = the SH part alone, which adds its
// share of d/d means3D to what PART 1 wrote.
// JAC (with MODE 2, PART 0 only): the forward left d colour / d dir and the clamp mask in sh_jac (layout above preprocess_fwd_kernel), so
// the SH coefficients are not read again; the SH part is then light enough (16 basis values, no 48-float pass) to run in
// the geometry launch: one kernel, one read of every row, no read-modify-write of d/d means3D.
template <int DEG, int MODE, int PART, bool JAC = false>
__global__ __launch_bounds__(256) void preprocess_bwd_kernel(
    int N, int K, int H, int W, const float* __restrict__ view, const float* __restrict__ proj,
    const float* __restrict__ campos, const float* __restrict__ means3D,
    const float* __restrict__ scales, const float* __restrict__ rots,
    const float* __restrict__ cov6, const float* __restrict__ opacities, const float* __restrict__ shs,
    const float* __restrict__ colors_precomp, const int32_t* __restrict__ radii,
    const float4* __restrict__ grad_rows, float* __restrict__ g_means3D,
    float* __restrict__ g_means2D, float* __restrict__ g_scales, float* __restrict__ g_rots,
    float* __restrict__ g_cov6, float* __restrict__ g_opac, float* __restrict__ g_shs,
    float* __restrict__ g_colors, bool acc, const float4* __restrict__ sh_jac) {
    static_assert(!JAC || (MODE == 2 && PART == 0), "the side-table form is the staged single launch");
    // acc: add this view's gradients to what the outputs hold (drawn Gaussians only; g_means2D, a
    // per-view statistic, is always overwritten) instead of overwriting them
    constexpr bool VEC4 = MODE >= 1;
    constexpr bool STAGE = MODE == 2;
    __shared__ float stage_all[STAGE ? 4 * 64 * BWD_STAGE_STRIDE : 1];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = i < N;
    if (!STAGE && !in_range) return;
    const bool drawn = in_range && radii[i] > 0;
    BwdGrads o = {};
    float (&gm)[3] = o.gm;
    float (&gm2)[2] = o.gm2;
    float (&gs)[3] = o.gs;
    float (&gq)[4] = o.gq;
    float (&gc6)[6] = o.gc6;
    float& gop = o.gop;
    float (&gcol)[3] = o.gcol;
    float* gsh_out = (g_shs && in_range && PART != 1) ? g_shs + (size_t)i * K * 3 : nullptr;
    int sh_written = 0;
    // ---- staged SH rows of this wave's 64 Gaussians
    const int lane = threadIdx.x & 63;
    float* st_wave = stage_all + (STAGE ? (threadIdx.x >> 6) * 64 * BWD_STAGE_STRIDE : 0);
    float* st_row = st_wave + (STAGE ? lane * BWD_STAGE_STRIDE : 0);
    const int RL = 3 * K;
    const int i0 = i - lane;                      // first Gaussian of this wave
    const int nq = STAGE && i0 < N ? ((N - i0 < 64 ? N - i0 : 64) * RL) / 4 : 0;   // float4s in the chunk
    if (STAGE && shs && PART != 1 && !JAC) {
        const uint64_t dmask = __ballot(drawn);
        const float4* p4 = reinterpret_cast<const float4*>(shs + (size_t)i0 * RL);
        for (int m = lane; m < nq; m += 64) {
            const unsigned f = 4u * (unsigned)m, gi = f / (unsigned)RL, off = f - gi * (unsigned)RL;
            if ((dmask >> gi) & 1ull) {
                const float4 t = p4[m];
                float* d = st_wave + gi * BWD_STAGE_STRIDE + off;
                d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (drawn) {
        // The gradient row, the opacity and the side-table row are requested BEFORE the projection is recomputed, so that
        // they travel while it runs (requested after it, the kernel waited for memory three times in a row: radius,
        // geometry, rows).
        const BwdInputs in = bwd_request<PART, JAC>(i, opacities, grad_rows, sh_jac);
        sh_written = bwd_one_view<DEG, MODE, PART, JAC, false>(in, i, K, H, W, view, proj, campos, means3D, scales, rots, cov6, shs,
                                                               colors_precomp, st_row, gsh_out, acc, o);
    }
    if (PART == 2) {
        if (drawn && g_means3D) {   // add the SH part's share to what PART 1 wrote
            g_means3D[3 * (size_t)i] += gm[0]; g_means3D[3 * (size_t)i + 1] += gm[1]; g_means3D[3 * (size_t)i + 2] += gm[2];
        }
    } else if (in_range && acc) {
        if (g_means2D) {
            g_means2D[3 * (size_t)i] = gm2[0] * 0.5f * (float)W;
            g_means2D[3 * (size_t)i + 1] = gm2[1] * 0.5f * (float)H;
            g_means2D[3 * (size_t)i + 2] = 0.0f;
        }
        if (drawn) {
            if (g_means3D) { g_means3D[3 * (size_t)i] += gm[0]; g_means3D[3 * (size_t)i + 1] += gm[1]; g_means3D[3 * (size_t)i + 2] += gm[2]; }
            if (g_scales) { g_scales[3 * (size_t)i] += gs[0]; g_scales[3 * (size_t)i + 1] += gs[1]; g_scales[3 * (size_t)i + 2] += gs[2]; }
            if (g_rots) {
                float4* o = reinterpret_cast<float4*>(g_rots + 4 * (size_t)i);
                const float4 t = *o;
                *o = make_float4(t.x + gq[0], t.y + gq[1], t.z + gq[2], t.w + gq[3]);
            }
            if (g_cov6) {
#pragma unroll
                for (int k = 0; k < 6; ++k) g_cov6[6 * (size_t)i + k] += gc6[k];
            }
            if (g_opac) g_opac[i] += gop;
            if (g_colors) { g_colors[3 * (size_t)i] += gcol[0]; g_colors[3 * (size_t)i + 1] += gcol[1]; g_colors[3 * (size_t)i + 2] += gcol[2]; }
        }
    } else if (in_range) {
        if (g_means3D) { g_means3D[3 * (size_t)i] = gm[0]; g_means3D[3 * (size_t)i + 1] = gm[1]; g_means3D[3 * (size_t)i + 2] = gm[2]; }
        if (g_means2D) {
            // densification statistic: gradient w.r.t. the NDC-scaled screen mean (pixels * W/2, H/2)
            g_means2D[3 * (size_t)i] = gm2[0] * 0.5f * (float)W;
            g_means2D[3 * (size_t)i + 1] = gm2[1] * 0.5f * (float)H;
            g_means2D[3 * (size_t)i + 2] = 0.0f;
        }
        if (g_scales) { g_scales[3 * (size_t)i] = gs[0]; g_scales[3 * (size_t)i + 1] = gs[1]; g_scales[3 * (size_t)i + 2] = gs[2]; }
        if (g_rots) *reinterpret_cast<float4*>(g_rots + 4 * (size_t)i) = make_float4(gq[0], gq[1], gq[2], gq[3]);
        if (g_cov6) {
#pragma unroll
            for (int k = 0; k < 6; ++k) g_cov6[6 * (size_t)i + k] = gc6[k];
        }
        if (g_opac) g_opac[i] = gop;
        if (g_colors) { g_colors[3 * (size_t)i] = gcol[0]; g_colors[3 * (size_t)i + 1] = gcol[1]; g_colors[3 * (size_t)i + 2] = gcol[2]; }
    }
    if (STAGE) {
        if (g_shs && PART != 1) {   // wave-uniform
            if (in_range)
                for (int k = sh_written; k < RL; ++k) st_row[k] = 0.0f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float4* q4 = reinterpret_cast<float4*>(g_shs + (size_t)i0 * RL);
            const uint64_t dmask = __ballot(drawn);
            for (int m = lane; m < nq; m += 64) {
                const unsigned f = 4u * (unsigned)m, gi = f / (unsigned)RL, off = f - gi * (unsigned)RL;
                const float* d = st_wave + gi * BWD_STAGE_STRIDE + off;
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v o = {d[0], d[1], d[2], d[3]};
                if (acc) {    // rows of culled Gaussians stay as they are
                    if (!((dmask >> gi) & 1ull)) continue;
                    const float4 t = q4[m];
                    o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
                }
                __builtin_nontemporal_store(o, reinterpret_cast<f4v*>(q4) + m);   // written once, read by the optimizer later
            }
        }
    } else if (gsh_out && !acc) {
        int k = sh_written;
        if (VEC4 && (k & 3) == 0) {
            float4* o4 = reinterpret_cast<float4*>(gsh_out);
            for (; k + 4 <= K * 3; k += 4) o4[k >> 2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (; k < K * 3; ++k) gsh_out[k] = 0.0f;
    }
}

// k views of the SAME Gaussians in one launch (the training step of SURVEY.md 8e / config 4: a rank renders several views
// per step and only the sum of their gradients is wanted).  The side-table staged form only (shs with 3 K <= 48 floats,
// every view's sh_jac): a lane keeps the sum of the views' geometry gradients in registers and of the SH rows in its LDS
// row, and every output row is written ONCE -- the view-by-view form (gs_preprocess_bwd with `accumulate`) reads and
// rewrites all 59 floats per Gaussian and view.  The sums are taken in view order, like the view-by-view form takes them.
struct BwdView {
    const float* view;
    const float* proj;
    const float* campos;
    const int32_t* radii;
    const float4* rows;
    const float4* sh_jac;
    float* g_means2D;
    int H, W;
};
struct BwdViews {
    BwdView v[GS_MAX_BWD_VIEWS];
};

template <int DEG>
__global__ __launch_bounds__(256) void preprocess_bwd_views_kernel(
    int N, int K, int nv, const BwdViews views, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rots, const float* __restrict__ cov6, const float* __restrict__ opacities,
    const float* __restrict__ shs, float* __restrict__ g_means3D, float* __restrict__ g_scales,
    float* __restrict__ g_rots, float* __restrict__ g_cov6, float* __restrict__ g_opac, float* __restrict__ g_shs,
    bool acc) {
    __shared__ float stage_all[4 * 64 * BWD_STAGE_STRIDE];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = i < N;
    const int lane = threadIdx.x & 63;
    float* st_wave = stage_all + (threadIdx.x >> 6) * 64 * BWD_STAGE_STRIDE;
    float* st_row = st_wave + lane * BWD_STAGE_STRIDE;
    const int RL = 3 * K;
    const int i0 = i - lane;
    const int nq = i0 < N ? ((N - i0 < 64 ? N - i0 : 64) * RL) / 4 : 0;
    if (g_shs)
        for (int f = 0; f < RL; ++f) st_row[f] = 0.0f;
    BwdGrads tot = {};
    bool any = false;
    for (int v = 0; v < nv; ++v) {
        const BwdView& vw = views.v[v];
        const bool drawn = in_range && vw.radii[i] > 0;
        BwdGrads o = {};
        if (drawn) {
            const BwdInputs in = bwd_request<0, true>(i, opacities, vw.rows, vw.sh_jac);
            bwd_one_view<DEG, 2, 0, true, true>(in, i, K, vw.H, vw.W, vw.view, vw.proj, vw.campos, means3D, scales, rots, cov6,
                                                shs, nullptr, st_row, g_shs ? st_row : nullptr, false, o);
#pragma unroll
            for (int c = 0; c < 3; ++c) { tot.gm[c] += o.gm[c]; tot.gs[c] += o.gs[c]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) tot.gq[c] += o.gq[c];
#pragma unroll
            for (int c = 0; c < 6; ++c) tot.gc6[c] += o.gc6[c];
            tot.gop += o.gop;
            any = true;
        }
        if (in_range && vw.g_means2D) {
            vw.g_means2D[3 * (size_t)i] = o.gm2[0] * 0.5f * (float)vw.W;
            vw.g_means2D[3 * (size_t)i + 1] = o.gm2[1] * 0.5f * (float)vw.H;
            vw.g_means2D[3 * (size_t)i + 2] = 0.0f;
        }
    }
    if (in_range && (any || !acc)) {       // acc: rows of Gaussians no view draws stay as they are
        if (g_means3D) {
            float* q = g_means3D + 3 * (size_t)i;
            const float o0 = acc ? q[0] : 0.f, o1 = acc ? q[1] : 0.f, o2 = acc ? q[2] : 0.f;
            q[0] = o0 + tot.gm[0]; q[1] = o1 + tot.gm[1]; q[2] = o2 + tot.gm[2];
        }
        if (g_scales) {
            float* q = g_scales + 3 * (size_t)i;
            const float o0 = acc ? q[0] : 0.f, o1 = acc ? q[1] : 0.f, o2 = acc ? q[2] : 0.f;
            q[0] = o0 + tot.gs[0]; q[1] = o1 + tot.gs[1]; q[2] = o2 + tot.gs[2];
        }
        if (g_rots) {
            float4* q = reinterpret_cast<float4*>(g_rots + 4 * (size_t)i);
            const float4 t = acc ? *q : make_float4(0.f, 0.f, 0.f, 0.f);
            *q = make_float4(t.x + tot.gq[0], t.y + tot.gq[1], t.z + tot.gq[2], t.w + tot.gq[3]);
        }
        if (g_cov6) {
#pragma unroll
            for (int k = 0; k < 6; ++k) g_cov6[6 * (size_t)i + k] = (acc ? g_cov6[6 * (size_t)i + k] : 0.f) + tot.gc6[k];
        }
        if (g_opac) g_opac[i] = (acc ? g_opac[i] : 0.f) + tot.gop;
    }
    if (g_shs) {   // kernel-uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float4* q4 = reinterpret_cast<float4*>(g_shs + (size_t)i0 * RL);
        const uint64_t dmask = __ballot(any);
        for (int m = lane; m < nq; m += 64) {
            const unsigned f = 4u * (unsigned)m, gi = f / (unsigned)RL, off = f - gi * (unsigned)RL;
            const float* d = st_wave + gi * BWD_STAGE_STRIDE + off;
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v o = {d[0], d[1], d[2], d[3]};
            if (acc) {
                if (!((dmask >> gi) & 1ull)) continue;
                const float4 t = q4[m];
                o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
            }
            __builtin_nontemporal_store(o, reinterpret_cast<f4v*>(q4) + m);
        }
    }
}

template <int DEG>
int launch_fwd(bool vec4, int N, int K, int H, int W, int row_b, int row_e, const float* view, const float* proj,
               const float* campos, const float* means3D, const float* scales, const float* rots,
               const float* cov6, const float* opac, const float* shs, const float* colors,
               float4* splat, int32_t* radii, uint32_t* tiles, uint32_t* depth_keys, uint32_t* ids, uint2* rects,
               float4* sh_jac, hipStream_t stream) {
    const dim3 grid(gs_div_up(N, 256)), block(256);
#define GS_LAUNCH_FWD(VEC4, JAC)                                                                                      \
    hipLaunchKernelGGL((preprocess_fwd_kernel<DEG, VEC4, JAC>), grid, block, 0, stream, N, K, H, W, row_b, row_e, view, \
                       proj, campos, means3D, scales, rots, cov6, opac, shs, colors, splat, radii, tiles, depth_keys,  \
                       ids, rects, sh_jac)
    const bool jac = sh_jac != nullptr && shs != nullptr;
    if (vec4 && jac) GS_LAUNCH_FWD(true, true);
    else if (vec4) GS_LAUNCH_FWD(true, false);
    else if (jac) GS_LAUNCH_FWD(false, true);
    else GS_LAUNCH_FWD(false, false);
#undef GS_LAUNCH_FWD
    GS_CHECK_LAUNCH("preprocess_fwd");
    return GS_OK;
}

template <int DEG>
int launch_bwd(int mode, int N, int K, int H, int W, const float* view, const float* proj,
               const float* campos, const float* means3D, const float* scales, const float* rots,
               const float* cov6, const float* opac, const float* shs, const float* colors, const int32_t* radii,
               const float4* rows, float* g_means3D, float* g_means2D, float* g_scales, float* g_rots,
               float* g_cov6, float* g_opac, float* g_shs, float* g_colors, bool acc, const float4* sh_jac,
               hipStream_t stream) {
    const dim3 grid(gs_div_up(N, 256)), block(256);
#define GS_LAUNCH_BWD(D_, MODE, PART, ...)                                                                       \
    hipLaunchKernelGGL((preprocess_bwd_kernel<D_, MODE, PART, ##__VA_ARGS__>), grid, block, 0, stream, N, K, H, W, view, proj, \
                       campos, means3D, scales, rots, cov6, opac, shs, colors, radii, rows, g_means3D, g_means2D,  \
                       g_scales, g_rots, g_cov6, g_opac, g_shs, g_colors, acc, sh_jac)
    if (mode == 2 && sh_jac && shs) {
        GS_LAUNCH_BWD(DEG, 2, 0, true);
    } else if (mode == 2) {
        GS_LAUNCH_BWD(0, 0, 1);
        GS_CHECK_LAUNCH("preprocess_bwd (geometry)");
        GS_LAUNCH_BWD(DEG, 2, 2);
    } else if (mode == 1) GS_LAUNCH_BWD(DEG, 1, 0);
    else GS_LAUNCH_BWD(DEG, 0, 0);
#undef GS_LAUNCH_BWD
    GS_CHECK_LAUNCH("preprocess_bwd");
    return GS_OK;
}

int check_common(int N, int K, int deg, int H, int W, const void* means3D, const void* scales,
                 const void* rots, const void* cov6, const void* shs, const void* colors) {
    GS_REQUIRE(N >= 0 && N < (1 << 30), "N = %d outside 0..2^30-1 (the tile kernels keep 4 x the Gaussian index in 32 bits)", N);
    GS_REQUIRE(H > 0 && W > 0 && H < 65536 && W < 65536, "image size %dx%d outside 1..65535", W, H);
    GS_REQUIRE(deg >= 0 && deg <= 3, "sh_degree %d outside 0..3", deg);
    GS_REQUIRE(N == 0 || means3D, "means3D is null");
    GS_REQUIRE((shs != nullptr) != (colors != nullptr), "exactly one of shs / colors_precomp must be given");
    GS_REQUIRE(!shs || (deg + 1) * (deg + 1) <= K, "shs has K=%d coefficients, sh_degree %d needs %d", K, deg,
               (deg + 1) * (deg + 1));
    GS_REQUIRE((cov6 != nullptr) != (scales != nullptr && rots != nullptr),
               "exactly one of (scales, rotations) / cov3D_precomp must be given");
    return GS_OK;
}

}  // namespace

extern "C" int gs_preprocess_fwd(int N, int K, int sh_degree, int H, int W, const float* viewmatrix,
                                 const float* projmatrix, const float* campos, const float* means3D,
                                 const float* scales, const float* rotations, const float* cov3D_precomp,
                                 const float* opacities, const float* shs, const float* colors_precomp,
                                 void* splat, int32_t* radii, uint32_t* tiles_touched, uint32_t* depth_keys,
                                 uint32_t* ids, void* rects, int tile_row_begin, int tile_row_end, float* sh_jac,
                                 void* stream_) {
    if (N == 0) return GS_OK;
    int rc = check_common(N, K, sh_degree, H, W, means3D, scales, rotations, cov3D_precomp, shs, colors_precomp);
    if (rc) return rc;
    GS_REQUIRE(viewmatrix && projmatrix && campos && opacities && splat && radii && tiles_touched, "null argument");
    GS_REQUIRE(!rotations || ((uintptr_t)rotations & 15) == 0, "rotations must be 16-byte aligned");
    GS_REQUIRE(((uintptr_t)splat & 15) == 0, "splat buffer must be 16-byte aligned");
    GS_REQUIRE((depth_keys == nullptr) == (ids == nullptr) && (ids == nullptr) == (rects == nullptr),
               "depth_keys, ids and rects go together");
    GS_REQUIRE(((uintptr_t)rects & 7) == 0, "rects must be 8-byte aligned");
    GS_REQUIRE(tile_row_begin >= 0 && tile_row_end >= tile_row_begin, "bad tile row band [%d,%d)", tile_row_begin, tile_row_end);
    GS_REQUIRE(((uintptr_t)sh_jac & 15) == 0, "sh_jac must be 16-byte aligned");
    if (N == 0) return GS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const bool vec4 = shs && ((K * 3) % 4 == 0) && (((uintptr_t)shs & 15) == 0);
    const int deg = shs ? sh_degree : 0;
    float4* sp = (float4*)splat;
    switch (deg) {
        case 0: return launch_fwd<0>(vec4, N, K, H, W, tile_row_begin, tile_row_end, viewmatrix, projmatrix, campos, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, sp, radii, tiles_touched, depth_keys, ids, (uint2*)rects, (float4*)sh_jac, stream);
        case 1: return launch_fwd<1>(vec4, N, K, H, W, tile_row_begin, tile_row_end, viewmatrix, projmatrix, campos, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, sp, radii, tiles_touched, depth_keys, ids, (uint2*)rects, (float4*)sh_jac, stream);
        case 2: return launch_fwd<2>(vec4, N, K, H, W, tile_row_begin, tile_row_end, viewmatrix, projmatrix, campos, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, sp, radii, tiles_touched, depth_keys, ids, (uint2*)rects, (float4*)sh_jac, stream);
        default: return launch_fwd<3>(vec4, N, K, H, W, tile_row_begin, tile_row_end, viewmatrix, projmatrix, campos, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, sp, radii, tiles_touched, depth_keys, ids, (uint2*)rects, (float4*)sh_jac, stream);
    }
}

extern "C" int gs_preprocess_fwd_views(int n_views, const GsFwdView* views, int N, int K, int sh_degree,
                                       const float* means3D, const float* scales, const float* rotations,
                                       const float* cov3D_precomp, const float* opacities, const float* shs,
                                       const float* colors_precomp, void* stream_) {
    GS_REQUIRE(n_views >= 0 && (n_views == 0 || views), "n_views = %d with views = %p", n_views, (const void*)views);
    if (N == 0 || n_views == 0) return GS_OK;
    const bool vec4 = shs && ((K * 3) % 4 == 0) && (((uintptr_t)shs & 15) == 0);
    bool fused = vec4 && n_views > 1;
    for (int v = 1; v < n_views; ++v) fused = fused && (views[v].sh_jac != nullptr) == (views[0].sh_jac != nullptr);
    if (!fused) {     // the same work view by view
        for (int v = 0; v < n_views; ++v) {
            const GsFwdView& w = views[v];
            const int rc = gs_preprocess_fwd(N, K, sh_degree, w.H, w.W, w.viewmatrix, w.projmatrix, w.campos, means3D, scales,
                                             rotations, cov3D_precomp, opacities, shs, colors_precomp, w.splat, w.radii,
                                             w.tiles_touched, w.depth_keys, w.ids, w.rects, w.tile_row_begin,
                                             w.tile_row_end, w.sh_jac, stream_);
            if (rc) return rc;
        }
        return GS_OK;
    }
    GS_REQUIRE(opacities, "null argument");
    GS_REQUIRE(!rotations || ((uintptr_t)rotations & 15) == 0, "rotations must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid(gs_div_up(N, 256)), block(256);
    const bool jac = views[0].sh_jac != nullptr;
    for (int v0 = 0; v0 < n_views; v0 += GS_MAX_FWD_VIEWS) {
        const int nv = n_views - v0 < GS_MAX_FWD_VIEWS ? n_views - v0 : GS_MAX_FWD_VIEWS;
        FwdViews a = {};
        for (int v = 0; v < nv; ++v) {
            const GsFwdView& w = views[v0 + v];
            const int rc = check_common(N, K, sh_degree, w.H, w.W, means3D, scales, rotations, cov3D_precomp, shs, colors_precomp);
            if (rc) return rc;
            GS_REQUIRE(w.viewmatrix && w.projmatrix && w.campos && w.splat && w.radii && w.tiles_touched,
                       "view %d: null argument", v0 + v);
            GS_REQUIRE(((uintptr_t)w.splat & 15) == 0 && ((uintptr_t)w.sh_jac & 15) == 0,
                       "view %d: splat and sh_jac must be 16-byte aligned", v0 + v);
            GS_REQUIRE((w.depth_keys == nullptr) == (w.ids == nullptr) && (w.ids == nullptr) == (w.rects == nullptr),
                       "view %d: depth_keys, ids and rects go together", v0 + v);
            GS_REQUIRE(((uintptr_t)w.rects & 7) == 0, "view %d: rects must be 8-byte aligned", v0 + v);
            GS_REQUIRE(w.tile_row_begin >= 0 && w.tile_row_end >= w.tile_row_begin, "view %d: bad tile row band [%d,%d)",
                       v0 + v, w.tile_row_begin, w.tile_row_end);
            a.v[v] = FwdView{w.viewmatrix, w.projmatrix, w.campos, (float4*)w.splat, w.radii, w.tiles_touched, w.depth_keys,
                             w.ids, (uint2*)w.rects, (float4*)w.sh_jac, w.H, w.W, w.tile_row_begin, w.tile_row_end};
        }
#define GS_LAUNCH_FWD_VIEWS(D_, J_)                                                                                      \
    hipLaunchKernelGGL((preprocess_fwd_views_kernel<D_, J_>), grid, block, 0, stream, N, K, nv, a, means3D, scales, rotations, \
                       cov3D_precomp, opacities, shs)
        switch (sh_degree * 2 + (jac ? 1 : 0)) {
            case 0: GS_LAUNCH_FWD_VIEWS(0, false); break;
            case 1: GS_LAUNCH_FWD_VIEWS(0, true); break;
            case 2: GS_LAUNCH_FWD_VIEWS(1, false); break;
            case 3: GS_LAUNCH_FWD_VIEWS(1, true); break;
            case 4: GS_LAUNCH_FWD_VIEWS(2, false); break;
            case 5: GS_LAUNCH_FWD_VIEWS(2, true); break;
            case 6: GS_LAUNCH_FWD_VIEWS(3, false); break;
            default: GS_LAUNCH_FWD_VIEWS(3, true); break;
        }
#undef GS_LAUNCH_FWD_VIEWS
        GS_CHECK_LAUNCH("preprocess_fwd_views");
    }
    return GS_OK;
}

extern "C" int gs_preprocess_bwd(int N, int K, int sh_degree, int H, int W, const float* viewmatrix,
                                 const float* projmatrix, const float* campos, const float* means3D,
                                 const float* scales, const float* rotations, const float* cov3D_precomp,
                                 const float* opacities, const float* shs, const float* colors_precomp,
                                 const int32_t* radii, const float* grad_rows, float* g_means3D, float* g_means2D,
                                 float* g_scales, float* g_rotations, float* g_cov3D, float* g_opacities,
                                 float* g_shs, float* g_colors, int accumulate, const float* sh_jac,
                                 void* stream_) {
    if (N == 0) return GS_OK;
    int rc = check_common(N, K, sh_degree, H, W, means3D, scales, rotations, cov3D_precomp, shs, colors_precomp);
    if (rc) return rc;
    GS_REQUIRE(viewmatrix && projmatrix && campos && opacities && radii && grad_rows, "null argument");
    GS_REQUIRE(((uintptr_t)grad_rows & 15) == 0, "grad_rows must be 16-byte aligned");
    GS_REQUIRE(!rotations || ((uintptr_t)rotations & 15) == 0, "rotations must be 16-byte aligned");
    GS_REQUIRE(!g_rotations || ((uintptr_t)g_rotations & 15) == 0, "g_rotations must be 16-byte aligned");
    GS_REQUIRE(((uintptr_t)sh_jac & 15) == 0, "sh_jac must be 16-byte aligned");
    if (N == 0) return GS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const bool vec4 = shs && ((K * 3) % 4 == 0) && (((uintptr_t)shs & 15) == 0) && (((uintptr_t)g_shs & 15) == 0);
    const int mode = vec4 ? (K * 3 <= BWD_STAGE_MAX_ROW ? 2 : 1) : 0;
    const int deg = shs ? sh_degree : 0;
    const float4* rows = (const float4*)grad_rows;
    switch (deg) {
        case 0: return launch_bwd<0>(mode, N, K, H, W, viewmatrix, projmatrix, campos, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, radii, rows, g_means3D, g_means2D, g_scales, g_rotations, g_cov3D, g_opacities, g_shs, g_colors, accumulate != 0, (const float4*)sh_jac, stream);
        case 1: return launch_bwd<1>(mode, N, K, H, W, viewmatrix, projmatrix, campos, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, radii, rows, g_means3D, g_means2D, g_scales, g_rotations, g_cov3D, g_opacities, g_shs, g_colors, accumulate != 0, (const float4*)sh_jac, stream);
        case 2: return launch_bwd<2>(mode, N, K, H, W, viewmatrix, projmatrix, campos, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, radii, rows, g_means3D, g_means2D, g_scales, g_rotations, g_cov3D, g_opacities, g_shs, g_colors, accumulate != 0, (const float4*)sh_jac, stream);
        default: return launch_bwd<3>(mode, N, K, H, W, viewmatrix, projmatrix, campos, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, radii, rows, g_means3D, g_means2D, g_scales, g_rotations, g_cov3D, g_opacities, g_shs, g_colors, accumulate != 0, (const float4*)sh_jac, stream);
    }
}

extern "C" int gs_preprocess_bwd_views(int n_views, const GsBwdView* views, int N, int K, int sh_degree,
                                       const float* means3D, const float* scales, const float* rotations,
                                       const float* cov3D_precomp, const float* opacities, const float* shs,
                                       const float* colors_precomp, float* g_means3D, float* g_scales,
                                       float* g_rotations, float* g_cov3D, float* g_opacities, float* g_shs,
                                       float* g_colors, int accumulate, void* stream_) {
    GS_REQUIRE(n_views >= 0 && (n_views == 0 || views), "n_views = %d with views = %p", n_views, (const void*)views);
    if (N == 0 || n_views == 0) return GS_OK;
    const bool vec4 = shs && ((K * 3) % 4 == 0) && (((uintptr_t)shs & 15) == 0) && (((uintptr_t)g_shs & 15) == 0);
    bool fused = vec4 && K * 3 <= BWD_STAGE_MAX_ROW && n_views > 1;
    for (int v = 0; v < n_views; ++v) fused = fused && views[v].sh_jac != nullptr;
    if (!fused) {     // the same sums view by view
        for (int v = 0; v < n_views; ++v) {
            const GsBwdView& w = views[v];
            const int rc = gs_preprocess_bwd(N, K, sh_degree, w.H, w.W, w.viewmatrix, w.projmatrix, w.campos, means3D, scales,
                                             rotations, cov3D_precomp, opacities, shs, colors_precomp, w.radii, w.grad_rows,
                                             g_means3D, w.g_means2D, g_scales, g_rotations, g_cov3D, g_opacities, g_shs,
                                             g_colors, (accumulate != 0 || v > 0) ? 1 : 0, w.sh_jac, stream_);
            if (rc) return rc;
        }
        return GS_OK;
    }
    GS_REQUIRE(!rotations || ((uintptr_t)rotations & 15) == 0, "rotations must be 16-byte aligned");
    GS_REQUIRE(!g_rotations || ((uintptr_t)g_rotations & 15) == 0, "g_rotations must be 16-byte aligned");
    GS_REQUIRE(opacities, "null argument");
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid(gs_div_up(N, 256)), block(256);
    for (int v0 = 0; v0 < n_views; v0 += GS_MAX_BWD_VIEWS) {
        const int nv = n_views - v0 < GS_MAX_BWD_VIEWS ? n_views - v0 : GS_MAX_BWD_VIEWS;
        BwdViews a = {};
        for (int v = 0; v < nv; ++v) {
            const GsBwdView& w = views[v0 + v];
            const int rc = check_common(N, K, sh_degree, w.H, w.W, means3D, scales, rotations, cov3D_precomp, shs, colors_precomp);
            if (rc) return rc;
            GS_REQUIRE(w.viewmatrix && w.projmatrix && w.campos && w.radii && w.grad_rows, "view %d: null argument", v0 + v);
            GS_REQUIRE(((uintptr_t)w.grad_rows & 15) == 0 && ((uintptr_t)w.sh_jac & 15) == 0,
                       "view %d: grad_rows and sh_jac must be 16-byte aligned", v0 + v);
            a.v[v] = BwdView{w.viewmatrix, w.projmatrix, w.campos, w.radii, (const float4*)w.grad_rows,
                             (const float4*)w.sh_jac, w.g_means2D, w.H, w.W};
        }
        const bool acc = accumulate != 0 || v0 > 0;
#define GS_LAUNCH_BWD_VIEWS(D_)                                                                                          \
    hipLaunchKernelGGL((preprocess_bwd_views_kernel<D_>), grid, block, 0, stream, N, K, nv, a, means3D, scales, rotations, \
                       cov3D_precomp, opacities, shs, g_means3D, g_scales, g_rotations, g_cov3D, g_opacities, g_shs, acc)
        switch (sh_degree) {
            case 0: GS_LAUNCH_BWD_VIEWS(0); break;
            case 1: GS_LAUNCH_BWD_VIEWS(1); break;
            case 2: GS_LAUNCH_BWD_VIEWS(2); break;
            default: GS_LAUNCH_BWD_VIEWS(3); break;
        }
#undef GS_LAUNCH_BWD_VIEWS
        GS_CHECK_LAUNCH("preprocess_bwd_views");
    }
    return GS_OK;
}
