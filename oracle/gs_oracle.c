/*
 * gs_oracle.c -- CPU restatement of the reference's built-in PyTorch rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or
 * call it.  The product path (libgsplat_hip.so) never links or falls back to it.
 *
 * What it restates (file:line in /root/reference):
 *   gaussian_renderer/__init__.py:76-183   GaussianRasterizer.forward
 *   gaussian_renderer/__init__.py:185-208  compute_colors_from_sh
 *   gaussian_renderer/__init__.py:210-264  compute_cov3D
 *   gaussian_renderer/__init__.py:266-289  cov_from_upper_triangular
 *   gaussian_renderer/__init__.py:291-334  project_cov3D_to_2D
 *   gaussian_renderer/__init__.py:336-442  splat_gaussians (the per-Gaussian loop)
 *   utils/sh_utils.py:9-26, 40-96          SH constants and eval_sh
 * The backward pass is the reverse-mode derivative of exactly those lines (the
 * reference only has it through torch autograd).
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against the
 * golden vectors in tests/golden/ (.npz files), which were produced by importing and
 * running the reference itself (tests/golden/gen_golden.py).
 *
 * Parts with no counterpart in the reference ("parity unpinned" there, pinned
 * HIP <-> this file bit-exactly): integer radii, tiles_touched, the 64-bit
 * tile|depth keys, their stable sort and the per-tile ranges.
 *
 * Algorithm shape: global depth order (descending, ties by ascending index),
 * then per Gaussian a hard 3-sigma integer bounding box, exactly as the
 * reference loops -- not the tile pipeline the HIP kernels use.  For speed the
 * image is cut into independent bands of rows (OpenMP); every pixel still sees
 * the Gaussians in the reference's order.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * All arithmetic is IEEE fp32 without contraction, so the integer outputs of
 * gso_preprocess are reproducible bit for bit by the HIP preprocess kernel.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float gso_f32; /* the 32-bit float behind the depth key, whatever `float` means below */
#ifdef GSO_REAL_DOUBLE
/*
 * Arbiter build (libgs_oracle_f64.so, oracle/oracle64.py): the same source with every float -- arithmetic and
 * interface arrays alike -- a double.  The f-suffixed constants keep their float values, so this is the SAME function
 * of the same inputs evaluated (nearly) exactly: what the fp32 arithmetic above is an approximation of.  Used by the
 * randomised parity test to tell an ill-conditioned case (the fp32 oracle itself is far from this) from a wrong result.
 */
#define float double
#define sqrtf sqrt
#define expf exp
#define ceilf ceil
#endif

#define TILE 16
#define BAND 8

/* utils/sh_utils.py:9-26 */
static const float C0 = 0.28209479177387814f;
static const float C1 = 0.4886025119029199f;
static const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.3153915652525205f,
                            -1.0925484305920792f, 0.5462742152960396f};
static const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                            0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                            -0.5900435899266435f};

int gso_version(void) { return 1; }

int gso_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void gso_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/*
 * Stop threshold of the HIP path's switchable deviation (NOT in the reference, which never stops: :429-440).
 * A pixel whose transmittance 1 - alpha_acc has dropped to <= t_eps composites nothing further, in the forward
 * and in the derivative alike (the backward is then the exact gradient of the truncated forward).  0 (the
 * default) restates the reference: at 0 a skipped Gaussian has alpha = w * 0 anyway.  Kept as a double so that
 * the float64 build compares against the same number.
 */
static double g_t_eps = 0.0;
void gso_set_t_eps(double t) { g_t_eps = t > 0.0 ? t : 0.0; }
double gso_get_t_eps(void) { return g_t_eps; }

/* SH basis of eval_sh (utils/sh_utils.py:64-94) for direction (x,y,z); nb = (deg+1)^2 */
static void sh_basis(int deg, float x, float y, float z, float *b) {
    b[0] = C0;
    if (deg > 0) {
        b[1] = -C1 * y;
        b[2] = C1 * z;
        b[3] = -C1 * x;
    }
    if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = C2[0] * xy;
        b[5] = C2[1] * yz;
        b[6] = C2[2] * (2.0f * zz - xx - yy);
        b[7] = C2[3] * xz;
        b[8] = C2[4] * (xx - yy);
        if (deg > 2) {
            b[9] = C3[0] * y * (3.0f * xx - yy);
            b[10] = C3[1] * xy * z;
            b[11] = C3[2] * y * (4.0f * zz - xx - yy);
            b[12] = C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            b[13] = C3[4] * x * (4.0f * zz - xx - yy);
            b[14] = C3[5] * z * (xx - yy);
            b[15] = C3[6] * x * (xx - 3.0f * yy);
        }
    }
}

/* d basis_k / d (x,y,z) */
static void sh_basis_grad(int deg, float x, float y, float z, float (*g)[3]) {
    g[0][0] = g[0][1] = g[0][2] = 0.0f;
    if (deg > 0) {
        g[1][0] = 0; g[1][1] = -C1; g[1][2] = 0;
        g[2][0] = 0; g[2][1] = 0; g[2][2] = C1;
        g[3][0] = -C1; g[3][1] = 0; g[3][2] = 0;
    }
    if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        g[4][0] = C2[0] * y; g[4][1] = C2[0] * x; g[4][2] = 0;
        g[5][0] = 0; g[5][1] = C2[1] * z; g[5][2] = C2[1] * y;
        g[6][0] = C2[2] * -2.0f * x; g[6][1] = C2[2] * -2.0f * y; g[6][2] = C2[2] * 4.0f * z;
        g[7][0] = C2[3] * z; g[7][1] = 0; g[7][2] = C2[3] * x;
        g[8][0] = C2[4] * 2.0f * x; g[8][1] = C2[4] * -2.0f * y; g[8][2] = 0;
        if (deg > 2) {
            g[9][0] = C3[0] * 6.0f * xy; g[9][1] = C3[0] * (3.0f * xx - 3.0f * yy); g[9][2] = 0;
            g[10][0] = C3[1] * yz; g[10][1] = C3[1] * xz; g[10][2] = C3[1] * xy;
            g[11][0] = C3[2] * -2.0f * xy; g[11][1] = C3[2] * (4.0f * zz - xx - 3.0f * yy);
            g[11][2] = C3[2] * 8.0f * yz;
            g[12][0] = C3[3] * -6.0f * xz; g[12][1] = C3[3] * -6.0f * yz;
            g[12][2] = C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
            g[13][0] = C3[4] * (4.0f * zz - 3.0f * xx - yy); g[13][1] = C3[4] * -2.0f * xy;
            g[13][2] = C3[4] * 8.0f * xz;
            g[14][0] = C3[5] * 2.0f * xz; g[14][1] = C3[5] * -2.0f * yz; g[14][2] = C3[5] * (xx - yy);
            g[15][0] = C3[6] * (3.0f * xx - 3.0f * yy); g[15][1] = C3[6] * -6.0f * xy; g[15][2] = 0;
        }
    }
}

/* rotation matrix of compute_cov3D (:222-252); q = (r,x,y,z), not normalised here */
static void quat_to_R(const float *q, float R[3][3]) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.0f - 2.0f * (y * y + z * z);
    R[0][1] = 2.0f * (x * y - r * z);
    R[0][2] = 2.0f * (x * z + r * y);
    R[1][0] = 2.0f * (x * y + r * z);
    R[1][1] = 1.0f - 2.0f * (x * x + z * z);
    R[1][2] = 2.0f * (y * z - r * x);
    R[2][0] = 2.0f * (x * z - r * y);
    R[2][1] = 2.0f * (y * z + r * x);
    R[2][2] = 1.0f - 2.0f * (x * x + y * y);
}

/* Sigma = (R S)(R S)^T (:254-262) or the symmetric fill of :266-289 */
static void cov3d_of(const float *scales, const float *rots, const float *cov6, int i, float S[3][3]) {
    if (cov6) {
        const float *c = cov6 + 6 * (size_t)i;
        S[0][0] = c[0]; S[0][1] = c[1]; S[0][2] = c[2];
        S[1][0] = c[1]; S[1][1] = c[3]; S[1][2] = c[4];
        S[2][0] = c[2]; S[2][1] = c[4]; S[2][2] = c[5];
        return;
    }
    float R[3][3], M[3][3];
    quat_to_R(rots + 4 * (size_t)i, R);
    const float *s = scales + 3 * (size_t)i;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) M[a][b] = R[a][b] * s[b];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            S[a][b] = M[a][0] * M[b][0] + M[a][1] * M[b][1] + M[a][2] * M[b][2];
}

/* depth -> 32-bit key whose ascending order is descending depth (ties keep index order) */
static uint32_t depth_key_desc(float z) {
    uint32_t u;
    gso_f32 zf = (gso_f32)z;
    zf = zf + 0.0f; /* -0 -> +0 so that equal depths give equal keys */
    memcpy(&u, &zf, 4);
    uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}

/*
 * Per-Gaussian stage: projection (:116-133), covariance (:136-149), colours
 * (:108-113), and the per-Gaussian tests of splat_gaussians (:368-402):
 * the +-50 px guard, det < 1e-7, the 3-sigma integer box.
 * rect = (x_min, y_min, x_max, y_max), all zero when the Gaussian is skipped.
 * clampmask bit c set <=> colour channel c passes gradient (0 <= v <= 1, :206).
 */
int gso_preprocess(int N, int K, int deg, int H, int W, const float *view, const float *proj,
                   const float *campos, const float *means3D, const float *scales,
                   const float *rots, const float *cov6, const float *shs,
                   const float *colors_precomp, float *xy, float *depth, float *cov2d,
                   float *conic, float *rgb, int32_t *rect, int32_t *radii,
                   uint32_t *tiles_touched, uint8_t *clampmask) {
    if (N < 0 || deg < 0 || deg > 3 || (shs && (deg + 1) * (deg + 1) > K)) return -1;
    if (!shs == !colors_precomp) return -2;
    if (!cov6 && (!scales || !rots)) return -3;
    const float focal_x = proj[0] * (float)W / 2.0f;  /* :318 */
    const float focal_y = proj[5] * (float)H / 2.0f;  /* :319 */
    const float fW = (float)W, fH = (float)H;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1],
                    pz = means3D[3 * (size_t)i + 2];
        /* :119-125  out[r] = sum_c M[r][c] * ph[c], ph = (p, 1) */
        float cam[3], clip[4];
        for (int r = 0; r < 3; ++r)
            cam[r] = view[4 * r] * px + view[4 * r + 1] * py + view[4 * r + 2] * pz + view[4 * r + 3];
        for (int r = 0; r < 4; ++r)
            clip[r] = proj[4 * r] * px + proj[4 * r + 1] * py + proj[4 * r + 2] * pz + proj[4 * r + 3];
        const float wden = clip[3] + 1e-7f;            /* :128 */
        const float ndcx = clip[0] / wden, ndcy = clip[1] / wden;
        const float mx = ((ndcx + 1.0f) * fW - 1.0f) * 0.5f;  /* :132 */
        const float my = ((ndcy + 1.0f) * fH - 1.0f) * 0.5f;  /* :133 */
        xy[2 * (size_t)i] = mx;
        xy[2 * (size_t)i + 1] = my;
        depth[i] = cam[2];

        float S[3][3];
        cov3d_of(scales, rots, cov6, i, S);
        /* :315-332 */
        const float x = cam[0], y = cam[1], z = cam[2];
        const float j00 = focal_x / (z + 1e-7f);
        const float j02 = -x * focal_x / (z * z + 1e-7f);
        const float j11 = focal_y / (z + 1e-7f);
        const float j12 = -y * focal_y / (z * z + 1e-7f);
        const float t00 = j00 * S[0][0] + j02 * S[2][0];
        const float t01 = j00 * S[0][1] + j02 * S[2][1];
        const float t02 = j00 * S[0][2] + j02 * S[2][2];
        const float t11 = j11 * S[1][1] + j12 * S[2][1];
        const float t12 = j11 * S[1][2] + j12 * S[2][2];
        const float c00 = (t00 * j00 + t02 * j02) + 0.3f;
        const float c01 = t01 * j11 + t02 * j12;
        const float c11 = (t11 * j11 + t12 * j12) + 0.3f;
        cov2d[3 * (size_t)i] = c00;
        cov2d[3 * (size_t)i + 1] = c01;
        cov2d[3 * (size_t)i + 2] = c11;

        /* colours */
        uint8_t cm = 7;
        if (colors_precomp) {
            for (int c = 0; c < 3; ++c) rgb[3 * (size_t)i + c] = colors_precomp[3 * (size_t)i + c];
        } else {
            float dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2]; /* :199 */
            float nrm = sqrtf(dx * dx + dy * dy + dz * dz) + 1e-7f;             /* :200 */
            dx = dx / nrm; dy = dy / nrm; dz = dz / nrm;
            float b[16];
            sh_basis(deg, dx, dy, dz, b);
            const int nb = (deg + 1) * (deg + 1);
            const float *sh = shs + (size_t)i * K * 3;
            cm = 0;
            for (int c = 0; c < 3; ++c) {
                float v = 0.0f;
                for (int k = 0; k < nb; ++k) v += b[k] * sh[3 * k + c];
                v = v + 0.5f;                                   /* :206 */
                if (v >= 0.0f && v <= 1.0f) cm |= (uint8_t)(1u << c);
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                rgb[3 * (size_t)i + c] = v;
            }
        }
        clampmask[i] = cm;

        /* per-Gaussian tests of splat_gaussians */
        int drawn = 1;
        if (mx < -50.0f || mx > fW + 50.0f || my < -50.0f || my > fH + 50.0f) drawn = 0; /* :369-375 */
        const float det = c00 * c11 - c01 * c01;                                         /* :379 */
        if (det < 1e-7f) drawn = 0;                                                      /* :380 */
        const float inv_det = 1.0f / det;                                                /* :384 */
        conic[3 * (size_t)i] = c11 * inv_det;
        conic[3 * (size_t)i + 1] = -c01 * inv_det;
        conic[3 * (size_t)i + 2] = c00 * inv_det;
        const float sx = sqrtf(c00) * 3.0f, sy = sqrtf(c11) * 3.0f;                      /* :393-394 */
        /* Python max(0,v) = v if v > 0 else 0 ; min(W,v) = v if v < W else W ; int() truncates */
        float lo, hi;
        int xmin, xmax, ymin, ymax;
        lo = mx - sx;       lo = (lo > 0.0f) ? lo : 0.0f;
        hi = mx + sx + 1.0f; hi = (hi < fW) ? hi : fW;
        xmin = (int)lo; xmax = (int)hi;
        lo = my - sy;       lo = (lo > 0.0f) ? lo : 0.0f;
        hi = my + sy + 1.0f; hi = (hi < fH) ? hi : fH;
        ymin = (int)lo; ymax = (int)hi;
        if (xmin >= xmax || ymin >= ymax) drawn = 0;                                     /* :401 */
        if (!drawn) {
            rect[4 * (size_t)i] = rect[4 * (size_t)i + 1] = rect[4 * (size_t)i + 2] =
                rect[4 * (size_t)i + 3] = 0;
            radii[i] = 0;
            tiles_touched[i] = 0;
        } else {
            rect[4 * (size_t)i] = xmin;
            rect[4 * (size_t)i + 1] = ymin;
            rect[4 * (size_t)i + 2] = xmax;
            rect[4 * (size_t)i + 3] = ymax;
            const float m = (sx > sy) ? sx : sy;
            radii[i] = (m >= 1.0f && m < 32767.0f) ? (int)ceilf(m) : (m >= 32767.0f ? 32767 : 1);
            const int tx0 = xmin / TILE, tx1 = (xmax - 1) / TILE + 1;
            const int ty0 = ymin / TILE, ty1 = (ymax - 1) / TILE + 1;
            tiles_touched[i] = (uint32_t)((tx1 - tx0) * (ty1 - ty0));
        }
    }
    return 0;
}

/* ---- the integer stage (no counterpart in the reference) ------------------------------- */

/* exclusive scan of tiles_touched; returns D */
int64_t gso_scan(int N, const uint32_t *tiles_touched, uint32_t *offsets) {
    uint64_t run = 0;
    for (int i = 0; i < N; ++i) {
        offsets[i] = (uint32_t)run;
        run += tiles_touched[i];
    }
    return (int64_t)run;
}

/* key = tile_id << 32 | depth_key_desc; value = Gaussian index; emitted in index, row, col order */
int gso_duplicate_with_keys(int N, int W, const int32_t *rect, const float *depth,
                            const uint32_t *offsets, const uint32_t *tiles_touched, uint64_t *keys,
                            uint32_t *vals) {
    const int tiles_x = (W + TILE - 1) / TILE;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        if (!tiles_touched[i]) continue;
        const int32_t *r = rect + 4 * (size_t)i;
        const int tx0 = r[0] / TILE, tx1 = (r[2] - 1) / TILE + 1;
        const int ty0 = r[1] / TILE, ty1 = (r[3] - 1) / TILE + 1;
        const uint32_t dk = depth_key_desc(depth[i]);
        size_t o = offsets[i];
        for (int ty = ty0; ty < ty1; ++ty)
            for (int tx = tx0; tx < tx1; ++tx) {
                keys[o] = ((uint64_t)(uint32_t)(ty * tiles_x + tx) << 32) | dk;
                vals[o] = (uint32_t)i;
                ++o;
            }
    }
    return 0;
}

/* stable LSD radix sort of (key,val) pairs on the low `bits` bits, 8 bits a pass */
int gso_sort_pairs(int64_t D, int bits, uint64_t *keys, uint32_t *vals) {
    if (D <= 0) return 0;
    uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)D);
    uint32_t *v2 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)D);
    if (!k2 || !v2) { free(k2); free(v2); return -1; }
    uint64_t *ks = keys, *kd = k2;
    uint32_t *vs = vals, *vd = v2;
    for (int shift = 0; shift < bits; shift += 8) {
        size_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        for (int64_t i = 0; i < D; ++i) cnt[((ks[i] >> shift) & 0xFF) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < D; ++i) {
            size_t p = cnt[(ks[i] >> shift) & 0xFF]++;
            kd[p] = ks[i];
            vd[p] = vs[i];
        }
        uint64_t *tk = ks; ks = kd; kd = tk;
        uint32_t *tv = vs; vs = vd; vd = tv;
    }
    if (ks != keys) {
        memcpy(keys, ks, sizeof(uint64_t) * (size_t)D);
        memcpy(vals, vs, sizeof(uint32_t) * (size_t)D);
    }
    free(k2);
    free(v2);
    return 0;
}

/* ranges[2*t], ranges[2*t+1] = [start, end) of tile t in the sorted list (0,0 if empty) */
int gso_tile_ranges(int64_t D, int num_tiles, const uint64_t *keys, uint32_t *ranges) {
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)num_tiles);
    for (int64_t i = 0; i < D; ++i) {
        uint32_t t = (uint32_t)(keys[i] >> 32);
        if ((int)t >= num_tiles) return -1;
        if (i == 0 || (uint32_t)(keys[i - 1] >> 32) != t) ranges[2 * t] = (uint32_t)i;
        if (i == D - 1 || (uint32_t)(keys[i + 1] >> 32) != t) ranges[2 * t + 1] = (uint32_t)(i + 1);
    }
    return 0;
}

/* ---- global order + row bands ------------------------------------------------------------ */

typedef struct { uint32_t key; int32_t idx; } ord_t;

/* the reference's argsort(depths, descending=True) (:153) over the drawn Gaussians; ties by index */
static int32_t *global_order(int N, const float *depth, const uint32_t *tiles_touched, int *V_out) {
    int V = 0;
    for (int i = 0; i < N; ++i) V += tiles_touched[i] != 0;
    ord_t *a = (ord_t *)malloc(sizeof(ord_t) * (size_t)(V ? V : 1));
    ord_t *b = (ord_t *)malloc(sizeof(ord_t) * (size_t)(V ? V : 1));
    int32_t *out = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V ? V : 1));
    int n = 0;
    for (int i = 0; i < N; ++i)
        if (tiles_touched[i]) { a[n].key = depth_key_desc(depth[i]); a[n].idx = i; ++n; }
    for (int shift = 0; shift < 32; shift += 8) {
        size_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        for (int i = 0; i < V; ++i) cnt[((a[i].key >> shift) & 0xFF) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (int i = 0; i < V; ++i) b[cnt[(a[i].key >> shift) & 0xFF]++] = a[i];
        ord_t *t = a; a = b; b = t;
    }
    for (int i = 0; i < V; ++i) out[i] = a[i].idx;
    free(a);
    free(b);
    *V_out = V;
    return out;
}

typedef struct { int nb; size_t *start; int32_t *list; } bands_t;

static int build_bands(int H, int V, const int32_t *order, const int32_t *rect, bands_t *B) {
    const int nb = (H + BAND - 1) / BAND;
    B->nb = nb;
    B->start = (size_t *)calloc((size_t)nb + 1, sizeof(size_t));
    if (!B->start) return -1;
    for (int s = 0; s < V; ++s) {
        const int32_t *r = rect + 4 * (size_t)order[s];
        for (int b = r[1] / BAND; b <= (r[3] - 1) / BAND; ++b) B->start[b + 1]++;
    }
    for (int b = 0; b < nb; ++b) B->start[b + 1] += B->start[b];
    B->list = (int32_t *)malloc(sizeof(int32_t) * (B->start[nb] ? B->start[nb] : 1));
    size_t *fill = (size_t *)malloc(sizeof(size_t) * (size_t)nb);
    if (!B->list || !fill) return -1;
    memcpy(fill, B->start, sizeof(size_t) * (size_t)nb);
    for (int s = 0; s < V; ++s) {
        const int32_t *r = rect + 4 * (size_t)order[s];
        for (int b = r[1] / BAND; b <= (r[3] - 1) / BAND; ++b) B->list[fill[b]++] = order[s];
    }
    free(fill);
    return 0;
}

static void free_bands(bands_t *B) { free(B->start); free(B->list); }

/*
 * splat_gaussians (:336-442): image starts at bg (:361); per Gaussian in far-to-near order
 *   power = -0.5 (a dx^2 + (b+c) dx dy + d dy^2), weight = exp(power) * opacity      (:417-426)
 *   alpha = weight * (1 - A); image += alpha*colour; A += alpha; depth += alpha*z    (:429-440)
 * over the integer box only.  out_image (3,H,W), out_depth (H,W), out_alpha (H,W).
 */
int gso_render_forward(int N, int H, int W, const float *bg, const float *xy, const float *depth,
                       const float *conic, const float *rgb, const float *opacities,
                       const int32_t *rect, const uint32_t *tiles_touched, float *out_image,
                       float *out_depth, float *out_alpha) {
    int V;
    int32_t *order = global_order(N, depth, tiles_touched, &V);
    bands_t B;
    if (build_bands(H, V, order, rect, &B)) return -1;
    const size_t HW = (size_t)H * W;
    const float t_eps = (float)g_t_eps;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B.nb; ++b) {
        const int y0 = b * BAND, y1 = (y0 + BAND < H) ? y0 + BAND : H;
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t p = (size_t)y * W + x;
                out_image[p] = bg[0]; out_image[HW + p] = bg[1]; out_image[2 * HW + p] = bg[2];
                out_depth[p] = 0.0f; out_alpha[p] = 0.0f;
            }
        for (size_t e = B.start[b]; e < B.start[b + 1]; ++e) {
            const int i = B.list[e];
            const int32_t *r = rect + 4 * (size_t)i;
            const float mx = xy[2 * (size_t)i], my = xy[2 * (size_t)i + 1];
            const float ca = conic[3 * (size_t)i], cb = conic[3 * (size_t)i + 1],
                        cd = conic[3 * (size_t)i + 2];
            const float bc = cb + cb;
            const float op = opacities[i], z = depth[i];
            const float cr = rgb[3 * (size_t)i], cg = rgb[3 * (size_t)i + 1], cbl = rgb[3 * (size_t)i + 2];
            const int ya = r[1] > y0 ? r[1] : y0, yb = r[3] < y1 ? r[3] : y1;
            for (int y = ya; y < yb; ++y) {
                const float dy = (float)y - my;
                for (int x = r[0]; x < r[2]; ++x) {
                    const float dx = (float)x - mx;
                    const float power = -((ca * dx * dx + bc * dx * dy) + cd * dy * dy) * 0.5f;
                    const float w = expf(power) * op;
                    const size_t p = (size_t)y * W + x;
                    if (t_eps > 0.0f && 1.0f - out_alpha[p] <= t_eps) continue; /* stopped (deviation, see gso_set_t_eps) */
                    const float alpha = w * (1.0f - out_alpha[p]);
                    out_image[p] += alpha * cr;
                    out_image[HW + p] += alpha * cg;
                    out_image[2 * HW + p] += alpha * cbl;
                    out_alpha[p] += alpha;
                    out_depth[p] += alpha * z;
                }
            }
        }
    }
    free_bands(&B);
    free(order);
    return 0;
}

static inline void atomic_add_d(double *p, double v) {
#pragma omp atomic
    *p += v;
}

/*
 * Reverse-mode derivative of splat_gaussians.  Per pixel, with Abar the adjoint of the running
 * alpha_acc (initially G_A):
 *   d_alpha = c.G_C + z G_Z + Abar ; d_w = d_alpha (1 - A_prev) ; Abar -= w d_alpha
 *   d_colour += alpha G_C ; d_z += alpha G_Z ; d_opacity += exp(power) d_w ; d_power = w d_w
 * Outputs (per Gaussian, zero where not drawn):
 *   d_xy (N,2)  d loss / d screen mean in pixels
 *   d_conic (N,3): d/da, d/d(b) == d/d(c) (each off-diagonal entry gets this), d/dd
 *   d_rgb (N,3), d_opacity (N), d_depth (N)
 */
int gso_render_backward(int N, int H, int W, const float *xy, const float *depth,
                        const float *conic, const float *rgb, const float *opacities,
                        const int32_t *rect, const uint32_t *tiles_touched, const float *G_C,
                        const float *G_Z, const float *G_A, float *d_xy, float *d_conic,
                        float *d_rgb, float *d_opacity, float *d_depth) {
    int V;
    int32_t *order = global_order(N, depth, tiles_touched, &V);
    bands_t B;
    if (build_bands(H, V, order, rect, &B)) return -1;
    const size_t HW = (size_t)H * W;
    double *acc = (double *)calloc((size_t)N * 10 + 1, sizeof(double));
    if (!acc) return -1;
    int fail = 0;
    const float t_eps = (float)g_t_eps;
    const float SKIPPED = 2.0f; /* alpha_acc never exceeds 1: marks a (pixel, Gaussian) visit behind the stop */
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B.nb; ++b) {
        const int y0 = b * BAND, y1 = (y0 + BAND < H) ? y0 + BAND : H;
        size_t evals = 0;
        for (size_t e = B.start[b]; e < B.start[b + 1]; ++e) {
            const int32_t *r = rect + 4 * (size_t)B.list[e];
            const int ya = r[1] > y0 ? r[1] : y0, yb = r[3] < y1 ? r[3] : y1;
            evals += (size_t)(yb - ya) * (size_t)(r[2] - r[0]);
        }
        float *Aprev = (float *)malloc(sizeof(float) * (evals ? evals : 1));
        float *A = (float *)calloc((size_t)BAND * W, sizeof(float));
        float *Abar = (float *)malloc(sizeof(float) * (size_t)BAND * W);
        if (!Aprev || !A || !Abar) { fail = 1; free(Aprev); free(A); free(Abar); continue; }
        /* forward replay, remembering alpha_acc before each update (what autograd saved) */
        size_t sp = 0;
        for (size_t e = B.start[b]; e < B.start[b + 1]; ++e) {
            const int i = B.list[e];
            const int32_t *r = rect + 4 * (size_t)i;
            const float mx = xy[2 * (size_t)i], my = xy[2 * (size_t)i + 1];
            const float ca = conic[3 * (size_t)i], cb = conic[3 * (size_t)i + 1],
                        cd = conic[3 * (size_t)i + 2];
            const float bc = cb + cb, op = opacities[i];
            const int ya = r[1] > y0 ? r[1] : y0, yb = r[3] < y1 ? r[3] : y1;
            for (int y = ya; y < yb; ++y) {
                const float dy = (float)y - my;
                for (int x = r[0]; x < r[2]; ++x) {
                    const float dx = (float)x - mx;
                    const float power = -((ca * dx * dx + bc * dx * dy) + cd * dy * dy) * 0.5f;
                    const float w = expf(power) * op;
                    float *a = &A[(size_t)(y - y0) * W + x];
                    if (t_eps > 0.0f && 1.0f - *a <= t_eps) { Aprev[sp++] = SKIPPED; continue; }
                    Aprev[sp++] = *a;
                    *a += w * (1.0f - *a);
                }
            }
        }
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < W; ++x) Abar[(size_t)(y - y0) * W + x] = G_A[(size_t)y * W + x];
        /* reverse sweep */
        for (size_t e = B.start[b + 1]; e-- > B.start[b];) {
            const int i = B.list[e];
            const int32_t *r = rect + 4 * (size_t)i;
            const float mx = xy[2 * (size_t)i], my = xy[2 * (size_t)i + 1];
            const float ca = conic[3 * (size_t)i], cb = conic[3 * (size_t)i + 1],
                        cd = conic[3 * (size_t)i + 2];
            const float bc = cb + cb, op = opacities[i], z = depth[i];
            const float cr = rgb[3 * (size_t)i], cg = rgb[3 * (size_t)i + 1], cbl = rgb[3 * (size_t)i + 2];
            const int ya = r[1] > y0 ? r[1] : y0, yb = r[3] < y1 ? r[3] : y1;
            double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int y = yb; y-- > ya;) {
                const float dy = (float)y - my;
                for (int x = r[2]; x-- > r[0];) {
                    const float dx = (float)x - mx;
                    const float power = -((ca * dx * dx + bc * dx * dy) + cd * dy * dy) * 0.5f;
                    const float g = expf(power);
                    const float w = g * op;
                    const float ap = Aprev[--sp];
                    if (ap == SKIPPED) continue;
                    const float T = 1.0f - ap;
                    const float alpha = w * T;
                    const size_t p = (size_t)y * W + x;
                    const float gr = G_C[p], gg = G_C[HW + p], gb = G_C[2 * HW + p], gz = G_Z[p];
                    float *ab = &Abar[(size_t)(y - y0) * W + x];
                    const float d_alpha = (cr * gr + cg * gg + cbl * gb) + z * gz + *ab;
                    const float d_w = d_alpha * T;
                    *ab = *ab - w * d_alpha;
                    const float d_power = w * d_w;
                    s[0] += (double)((ca * dx + 0.5f * bc * dy) * d_power); /* d mx */
                    s[1] += (double)((cd * dy + 0.5f * bc * dx) * d_power); /* d my */
                    s[2] += (double)(-0.5f * dx * dx * d_power);             /* d a  */
                    s[3] += (double)(-0.5f * dx * dy * d_power);             /* d b (= d c) */
                    s[4] += (double)(-0.5f * dy * dy * d_power);             /* d d  */
                    s[5] += (double)(g * d_w);                               /* d opacity */
                    s[6] += (double)(alpha * gr);
                    s[7] += (double)(alpha * gg);
                    s[8] += (double)(alpha * gb);
                    s[9] += (double)(alpha * gz);
                }
            }
            for (int k = 0; k < 10; ++k) atomic_add_d(&acc[(size_t)i * 10 + k], s[k]);
        }
        free(Aprev); free(A); free(Abar);
    }
    for (int i = 0; i < N; ++i) {
        const double *s = acc + (size_t)i * 10;
        d_xy[2 * (size_t)i] = (float)s[0];
        d_xy[2 * (size_t)i + 1] = (float)s[1];
        d_conic[3 * (size_t)i] = (float)s[2];
        d_conic[3 * (size_t)i + 1] = (float)s[3];
        d_conic[3 * (size_t)i + 2] = (float)s[4];
        d_opacity[i] = (float)s[5];
        d_rgb[3 * (size_t)i] = (float)s[6];
        d_rgb[3 * (size_t)i + 1] = (float)s[7];
        d_rgb[3 * (size_t)i + 2] = (float)s[8];
        d_depth[i] = (float)s[9];
    }
    free(acc);
    free_bands(&B);
    free(order);
    return fail ? -1 : 0;
}

/*
 * Reverse-mode derivative of the per-Gaussian stage: 2x2 inverse (:384-389), J Sigma J^T + 0.3 I
 * (:315-332), Sigma(scales, rotations) (:222-262) or the 6-vector (:266-289), the two
 * homogeneous transforms and the NDC->pixel map (:119-133), SH colours with the two-sided clamp
 * (:199-206).  Any output pointer may be NULL.  Skipped Gaussians get zeros.
 */
int gso_preprocess_backward(int N, int K, int deg, int H, int W, const float *view,
                            const float *proj, const float *campos, const float *means3D,
                            const float *scales, const float *rots, const float *cov6,
                            const float *shs, const uint32_t *tiles_touched,
                            const uint8_t *clampmask, const float *cov2d, const float *d_xy,
                            const float *d_conic, const float *d_rgb, const float *d_depth,
                            float *g_means3D, float *g_scales, float *g_rots, float *g_cov6,
                            float *g_shs, float *g_colors_precomp) {
    const float focal_x = proj[0] * (float)W / 2.0f;
    const float focal_y = proj[5] * (float)H / 2.0f;
    const int nb = (deg + 1) * (deg + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float gm[3] = {0, 0, 0};
        if (g_scales) g_scales[3 * (size_t)i] = g_scales[3 * (size_t)i + 1] = g_scales[3 * (size_t)i + 2] = 0;
        if (g_rots) for (int k = 0; k < 4; ++k) g_rots[4 * (size_t)i + k] = 0;
        if (g_cov6) for (int k = 0; k < 6; ++k) g_cov6[6 * (size_t)i + k] = 0;
        if (g_shs) for (int k = 0; k < 3 * K; ++k) g_shs[(size_t)i * K * 3 + k] = 0;
        if (g_colors_precomp) for (int k = 0; k < 3; ++k) g_colors_precomp[3 * (size_t)i + k] = 0;
        if (tiles_touched[i]) {
            const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1],
                        pz = means3D[3 * (size_t)i + 2];
            float cam[3], clip[4];
            for (int r = 0; r < 3; ++r)
                cam[r] = view[4 * r] * px + view[4 * r + 1] * py + view[4 * r + 2] * pz + view[4 * r + 3];
            for (int r = 0; r < 4; ++r)
                clip[r] = proj[4 * r] * px + proj[4 * r + 1] * py + proj[4 * r + 2] * pz + proj[4 * r + 3];
            /* colours */
            const float *dc = d_rgb + 3 * (size_t)i;
            if (g_colors_precomp) {
                for (int c = 0; c < 3; ++c) g_colors_precomp[3 * (size_t)i + c] = dc[c];
            } else if (shs) {
                float vx = px - campos[0], vy = py - campos[1], vz = pz - campos[2];
                const float n = sqrtf(vx * vx + vy * vy + vz * vz);
                const float ne = n + 1e-7f;
                const float dx = vx / ne, dy = vy / ne, dz = vz / ne;
                float b[16], bg[16][3];
                sh_basis(deg, dx, dy, dz, b);
                sh_basis_grad(deg, dx, dy, dz, bg);
                const float *sh = shs + (size_t)i * K * 3;
                float ddir[3] = {0, 0, 0};
                for (int c = 0; c < 3; ++c) {
                    const float g = (clampmask[i] >> c) & 1 ? dc[c] : 0.0f;
                    for (int k = 0; k < nb; ++k) {
                        if (g_shs) g_shs[(size_t)i * K * 3 + 3 * k + c] = b[k] * g;
                        const float sg = sh[3 * k + c] * g;
                        ddir[0] += bg[k][0] * sg;
                        ddir[1] += bg[k][1] * sg;
                        ddir[2] += bg[k][2] * sg;
                    }
                }
                /* dir = v / (|v| + 1e-7) */
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
            /* screen mean: mx = ((ndcx+1) W - 1)/2, ndc = clip.xy / (clip.w + 1e-7) */
            {
                const float wden = clip[3] + 1e-7f;
                const float dnx = 0.5f * (float)W * d_xy[2 * (size_t)i];
                const float dny = 0.5f * (float)H * d_xy[2 * (size_t)i + 1];
                const float dcx = dnx / wden, dcy = dny / wden;
                const float dcw = -(dnx * clip[0] + dny * clip[1]) / (wden * wden);
                for (int c = 0; c < 3; ++c)
                    gm[c] += proj[c] * dcx + proj[4 + c] * dcy + proj[12 + c] * dcw;
            }
            /* conic -> cov2D */
            const float c00 = cov2d[3 * (size_t)i], c01 = cov2d[3 * (size_t)i + 1], c11 = cov2d[3 * (size_t)i + 2];
            const float det = c00 * c11 - c01 * c01;
            const float inv_det = 1.0f / det;
            const float da = d_conic[3 * (size_t)i], db = d_conic[3 * (size_t)i + 1], dd = d_conic[3 * (size_t)i + 2];
            const float d_inv_det = da * c11 - db * c01 - db * c01 + dd * c00;
            const float d_det = -inv_det * inv_det * d_inv_det;
            float G2[2][2]; /* d loss / d cov2D entries */
            G2[0][0] = dd * inv_det + d_det * c11;
            G2[1][1] = da * inv_det + d_det * c00;
            G2[0][1] = -db * inv_det - d_det * c01;
            G2[1][0] = -db * inv_det - d_det * c01;
            /* cov2D = J S J^T */
            float S[3][3];
            cov3d_of(scales, rots, cov6, i, S);
            const float x = cam[0], y = cam[1], z = cam[2];
            const float zi = z + 1e-7f, z2 = z * z + 1e-7f;
            float J[2][3] = {{focal_x / zi, 0.0f, -x * focal_x / z2}, {0.0f, focal_y / zi, -y * focal_y / z2}};
            float dS[3][3];
            for (int a = 0; a < 3; ++a)
                for (int b2 = 0; b2 < 3; ++b2) {
                    float v = 0;
                    for (int r = 0; r < 2; ++r)
                        for (int c = 0; c < 2; ++c) v += J[r][a] * G2[r][c] * J[c][b2];
                    dS[a][b2] = v;
                }
            float JS[2][3];
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 3; ++c) JS[r][c] = J[r][0] * S[0][c] + J[r][1] * S[1][c] + J[r][2] * S[2][c];
            float dJ[2][3];
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 3; ++c)
                    dJ[r][c] = (G2[r][0] + G2[0][r]) * JS[0][c] + (G2[r][1] + G2[1][r]) * JS[1][c];
            float dcam[3];
            dcam[0] = dJ[0][2] * (-focal_x / z2);
            dcam[1] = dJ[1][2] * (-focal_y / z2);
            dcam[2] = dJ[0][0] * (-focal_x / (zi * zi)) + dJ[1][1] * (-focal_y / (zi * zi)) +
                      dJ[0][2] * (x * focal_x * 2.0f * z / (z2 * z2)) +
                      dJ[1][2] * (y * focal_y * 2.0f * z / (z2 * z2)) + d_depth[i];
            for (int c = 0; c < 3; ++c)
                gm[c] += view[c] * dcam[0] + view[4 + c] * dcam[1] + view[8 + c] * dcam[2];
            if (cov6) {
                if (g_cov6) {
                    float *g = g_cov6 + 6 * (size_t)i;
                    g[0] = dS[0][0]; g[1] = dS[0][1] + dS[1][0]; g[2] = dS[0][2] + dS[2][0];
                    g[3] = dS[1][1]; g[4] = dS[1][2] + dS[2][1]; g[5] = dS[2][2];
                }
            } else {
                float R[3][3], M[3][3], dM[3][3];
                const float *q = rots + 4 * (size_t)i;
                const float *s = scales + 3 * (size_t)i;
                quat_to_R(q, R);
                for (int a = 0; a < 3; ++a)
                    for (int b2 = 0; b2 < 3; ++b2) M[a][b2] = R[a][b2] * s[b2];
                for (int a = 0; a < 3; ++a)
                    for (int b2 = 0; b2 < 3; ++b2) {
                        float v = 0;
                        for (int k = 0; k < 3; ++k) v += (dS[a][k] + dS[k][a]) * M[k][b2];
                        dM[a][b2] = v;
                    }
                float dR[3][3];
                for (int b2 = 0; b2 < 3; ++b2) {
                    float v = 0;
                    for (int a = 0; a < 3; ++a) { v += dM[a][b2] * R[a][b2]; dR[a][b2] = dM[a][b2] * s[b2]; }
                    if (g_scales) g_scales[3 * (size_t)i + b2] = v;
                }
                if (g_rots) {
                    const float r = q[0], qx = q[1], qy = q[2], qz = q[3];
                    float *g = g_rots + 4 * (size_t)i;
                    g[0] = 2.0f * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
                    g[1] = 2.0f * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2.0f * qx * dR[1][1] - r * dR[1][2] + qz * dR[2][0] + r * dR[2][1] - 2.0f * qx * dR[2][2]);
                    g[2] = 2.0f * (-2.0f * qy * dR[0][0] + qx * dR[0][1] + r * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] - r * dR[2][0] + qz * dR[2][1] - 2.0f * qy * dR[2][2]);
                    g[3] = 2.0f * (-2.0f * qz * dR[0][0] - r * dR[0][1] + qx * dR[0][2] + r * dR[1][0] - 2.0f * qz * dR[1][1] + qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
                }
            }
        }
        if (g_means3D) for (int c = 0; c < 3; ++c) g_means3D[3 * (size_t)i + c] = gm[c];
    }
    return 0;
}
