// Fused Adam step over all per-Gaussian parameter groups + densification statistics, one launch
// (the step right after the backward in every iteration).
//
// Replaces, in the reference: torch.optim.Adam(l, lr=0.0, eps=1e-15).step() over the six groups
// xyz / f_dc / f_rest / opacity / scaling / rotation (scene/gaussian_model.py:264-297, train.py:217),
// max_radii2D[vis] = max(max_radii2D[vis], radii[vis]) (train.py:184-187) and
// add_densification_stats (scene/gaussian_model.py:701-713).  The arithmetic is the one torch's own GPU
// kernels perform for Adam (no weight decay, no amsgrad, not maximize), rounding for rounding -- the
// forms were determined against torch 2.10 (ROCm) op by op (tools/adam_forms_gpu.py) and the result
// is bit-identical to torch.optim.Adam on the same GPU (its default there, the foreach path; the
// single-tensor path multiplies by a rounded reciprocal instead of dividing; tests/test_gpu_adam.py):
//     m  = fma(1 - beta1, g - m, m)                      exp_avg.lerp_(grad, 1 - beta1)
//     v  = fma(1 - beta2, g * g, v * beta2)              exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
//     d  = sqrt(v) / sqrt(1 - beta2^t) + eps             _foreach_sqrt, _foreach_div_(., [bc2_sqrt]), _foreach_add_(., eps)
//     p  = fma(-(lr / (1 - beta1^t)), m / d, p)          param.addcdiv_(exp_avg, denom, value=-step_size)
// with the scalars computed in double on the host exactly as torch's Python does and rounded to
// fp32 once.  Contraction is off: every fma is explicit; sqrt and the divisions are IEEE (sqrtf, not
// __fsqrt_rn: the latter compiles to a bare v_sqrt_f32, one ulp off for ~20 % of inputs).
//
// HBM-bound: per parameter 16 B read (p, g, m, v) + 12 B written.  A gradient may be a strided view
// (row stride != row length): the f_dc / f_rest gradients are the two column ranges of the
// rasterizer's (N,K,3) SH gradient, read in place instead of through the copies autograd's cat
// backward would make.
#include "gsplat_common.h"
#include "../../include/gsplat.h"
#pragma clang fp contract(off)

namespace {

constexpr int AD_MAX_GROUPS = 8;
constexpr int AD_THREADS = 256;
#ifndef AD_PER_THREAD_DEF
#define AD_PER_THREAD_DEF 8
#endif
constexpr int AD_PER_THREAD = AD_PER_THREAD_DEF;
constexpr int AD_V4 = AD_PER_THREAD / 4;    // float4s per lane in the vector path
constexpr int AD_BLOCK_ELEMS = AD_THREADS * AD_PER_THREAD;   // 2048 parameters per workgroup

struct AdGroup {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n_elems;        // rows * row_len
    long long g_row_stride;   // floats between rows of g
    unsigned long long magic; // floor(2^64 / row_len) + 1: e / row_len == umul64hi(e, magic) for e < 2^40
    int row_len;
    float neg_step_size;      // -(lr / (1 - beta1^t))
    int vec4;                 // p, m, v 16-byte aligned and n_elems % 4 == 0: 1 = g too (contiguous), 2 = g strided
    int pad;
};

struct AdArgs {
    AdGroup grp[AD_MAX_GROUPS];
    long long block_begin[AD_MAX_GROUPS + 1];   // first workgroup of each group; [ngroups] = first stats workgroup
    int ngroups;
    float one_minus_beta1, beta2, one_minus_beta2, bc2_sqrt, eps;
    // densification statistics (thread per Gaussian), all optional
    long long N;
    int num_views;
    const float* means2D_grad;   // (views, N, 3)
    const int32_t* radii;        // (views, N)
    float* xyz_gradient_accum;   // (N,)
    float* denom;                // (N,)
    float* max_radii2D;          // (N,)
};

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const AdArgs& a, float neg_step) {
    m = __fmaf_rn(a.one_minus_beta1, g - m, m);
    v = __fmaf_rn(a.one_minus_beta2, g * g, v * a.beta2);
    const float d = __fdiv_rn(sqrtf(v), a.bc2_sqrt) + a.eps;
    p = __fmaf_rn(neg_step, __fdiv_rn(m, d), p);
}

__global__ __launch_bounds__(AD_THREADS) void adam_densify_kernel(const AdArgs a) {
    const long long blk = blockIdx.x;
    if (blk >= a.block_begin[a.ngroups]) {
        // ---- densification statistics of Gaussian i over the step's views (in view order, as the
        //      reference's one-view-per-iteration loop would add them)
        const long long i = (blk - a.block_begin[a.ngroups]) * AD_THREADS + threadIdx.x;
        if (i >= a.N) return;
        float acc = a.xyz_gradient_accum ? a.xyz_gradient_accum[i] : 0.0f;
        float den = a.denom ? a.denom[i] : 0.0f;
        float mr = a.max_radii2D ? a.max_radii2D[i] : 0.0f;
        bool seen = false;
        for (int v = 0; v < a.num_views; ++v) {
            const long long k = (long long)v * a.N + i;
            const int r = a.radii[k];
            if (r > 0) {
                seen = true;
                if (a.means2D_grad) {
                    const float gx = a.means2D_grad[3 * k], gy = a.means2D_grad[3 * k + 1];
                    acc += sqrtf(gx * gx + gy * gy);
                }
                den += 1.0f;
                mr = fmaxf(mr, (float)r);
            }
        }
        if (seen) {
            if (a.xyz_gradient_accum && a.means2D_grad) a.xyz_gradient_accum[i] = acc;
            if (a.denom && a.means2D_grad) a.denom[i] = den;
            if (a.max_radii2D) a.max_radii2D[i] = mr;
        }
        return;
    }
    int gi = 0;
#pragma unroll
    for (int k = 1; k < AD_MAX_GROUPS; ++k) gi += (k < a.ngroups && blk >= a.block_begin[k]) ? 1 : 0;
    const AdGroup& G = a.grp[gi];
    const long long base = (blk - a.block_begin[gi]) * AD_BLOCK_ELEMS;
    const float neg_step = G.neg_step_size;
    if (G.vec4) {
        // two float4 per lane, all loads issued before the first use.  vec4 == 2: the gradient is a
        // strided view, fetched as four dwords (a float4 of parameters may straddle two gradient rows)
        float4 p[AD_V4], g[AD_V4], m[AD_V4], v[AD_V4];
        long long e[AD_V4];
#pragma unroll
        for (int k = 0; k < AD_V4; ++k) {
            e[k] = base + ((long long)k * AD_THREADS + threadIdx.x) * 4;
            if (e[k] < G.n_elems) {
                p[k] = *reinterpret_cast<const float4*>(G.p + e[k]);
                if (G.vec4 == 1) {
                    g[k] = *reinterpret_cast<const float4*>(G.g + e[k]);
                } else {
                    const unsigned long long row = G.magic ? __umul64hi((unsigned long long)e[k], G.magic) : (unsigned long long)e[k];
                    int col = (int)(e[k] - (long long)row * G.row_len);
                    const float* gp = G.g + (long long)row * G.g_row_stride;
                    float t[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        t[c] = gp[col];
                        if (++col == G.row_len) { col = 0; gp += G.g_row_stride; }
                    }
                    g[k] = make_float4(t[0], t[1], t[2], t[3]);
                }
                m[k] = *reinterpret_cast<const float4*>(G.m + e[k]);
                v[k] = *reinterpret_cast<const float4*>(G.v + e[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < AD_V4; ++k) {
            if (e[k] < G.n_elems) {
                adam_update(p[k].x, g[k].x, m[k].x, v[k].x, a, neg_step);
                adam_update(p[k].y, g[k].y, m[k].y, v[k].y, a, neg_step);
                adam_update(p[k].z, g[k].z, m[k].z, v[k].z, a, neg_step);
                adam_update(p[k].w, g[k].w, m[k].w, v[k].w, a, neg_step);
                typedef float f4v __attribute__((ext_vector_type(4)));
                // streamed once per step, far larger than the caches: keep them out of the way of the reads
                __builtin_nontemporal_store(f4v{p[k].x, p[k].y, p[k].z, p[k].w}, reinterpret_cast<f4v*>(G.p + e[k]));
                __builtin_nontemporal_store(f4v{m[k].x, m[k].y, m[k].z, m[k].w}, reinterpret_cast<f4v*>(G.m + e[k]));
                __builtin_nontemporal_store(f4v{v[k].x, v[k].y, v[k].z, v[k].w}, reinterpret_cast<f4v*>(G.v + e[k]));
            }
        }
        return;
    }
    // scalar path: consecutive lanes take consecutive parameters (coalesced dwords); the gradient of
    // parameter e lives at row (e / row_len) * g_row_stride + e % row_len
    float p[AD_PER_THREAD], g[AD_PER_THREAD], m[AD_PER_THREAD], v[AD_PER_THREAD];
#pragma unroll
    for (int k = 0; k < AD_PER_THREAD; ++k) {
        const long long e = base + (long long)k * AD_THREADS + threadIdx.x;
        if (e < G.n_elems) {
            const unsigned long long row = G.magic ? __umul64hi((unsigned long long)e, G.magic) : (unsigned long long)e;
            const long long col = e - (long long)row * G.row_len;
            p[k] = G.p[e];
            g[k] = G.g[(long long)row * G.g_row_stride + col];
            m[k] = G.m[e];
            v[k] = G.v[e];
        }
    }
#pragma unroll
    for (int k = 0; k < AD_PER_THREAD; ++k) {
        const long long e = base + (long long)k * AD_THREADS + threadIdx.x;
        if (e < G.n_elems) {
            adam_update(p[k], g[k], m[k], v[k], a, neg_step);
            G.p[e] = p[k];
            G.m[e] = m[k];
            G.v[e] = v[k];
        }
    }
}

}  // namespace

extern "C" int gs_adam_densify_step(int num_groups, const GsAdamGroup* groups, double beta1, double beta2, double eps,
                                    long long step, long long N, int num_views, const float* means2D_grad,
                                    const int32_t* radii, float* xyz_gradient_accum, float* denom,
                                    float* max_radii2D, void* stream_) {
    GS_REQUIRE(num_groups >= 0 && num_groups <= AD_MAX_GROUPS, "num_groups %d outside 0..%d", num_groups, AD_MAX_GROUPS);
    GS_REQUIRE(num_groups == 0 || groups, "groups is null");
    GS_REQUIRE(step >= 1, "step must be >= 1 (the count after this update, as torch's state['step'])");
    GS_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "betas outside [0,1)");
    GS_REQUIRE(N >= 0 && num_views >= 0, "negative size");
    const bool stats = N > 0 && num_views > 0 && radii != nullptr;
    GS_REQUIRE(!(num_views > 0 && N > 0 && !radii && (xyz_gradient_accum || denom || max_radii2D)),
               "densification statistics need radii");
    AdArgs a{};
    a.ngroups = num_groups;
    // torch/optim/adam.py (_single_tensor_adam): Python doubles, rounded to fp32 when they meet the tensor
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    a.one_minus_beta1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.one_minus_beta2 = (float)(1.0 - beta2);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = (float)eps;
    long long blocks = 0;
    for (int k = 0; k < num_groups; ++k) {
        const GsAdamGroup& s = groups[k];
        GS_REQUIRE(s.rows >= 0 && s.row_len >= 1, "group %d: bad shape %lld x %d", k, s.rows, s.row_len);
        const long long n = s.rows * (long long)s.row_len;
        GS_REQUIRE(n == 0 || (s.param && s.grad && s.exp_avg && s.exp_avg_sq), "group %d: null pointer", k);
        GS_REQUIRE(n < (1ll << 40) / s.row_len, "group %d too large", k);
        const long long stride = s.grad_row_stride > 0 ? s.grad_row_stride : s.row_len;
        GS_REQUIRE(stride >= s.row_len, "group %d: grad_row_stride %lld < row_len %d", k, stride, s.row_len);
        AdGroup& d = a.grp[k];
        d.p = s.param; d.g = s.grad; d.m = s.exp_avg; d.v = s.exp_avg_sq;
        d.n_elems = n;
        d.g_row_stride = stride;
        d.row_len = s.row_len;
        d.magic = s.row_len == 1 ? 0ull : ~0ull / (unsigned long long)s.row_len + 1ull;   // 0: row_len 1, row = e
        d.neg_step_size = (float)(-(s.lr / bc1));
        const uintptr_t al = (uintptr_t)s.param | (uintptr_t)s.exp_avg | (uintptr_t)s.exp_avg_sq;
        d.vec4 = ((al & 15) != 0 || n % 4 != 0) ? 0 : (stride == s.row_len && ((uintptr_t)s.grad & 15) == 0) ? 1 : 2;
        a.block_begin[k] = blocks;
        blocks += (n + AD_BLOCK_ELEMS - 1) / AD_BLOCK_ELEMS;
    }
    for (int k = num_groups; k <= AD_MAX_GROUPS; ++k) a.block_begin[k] = blocks;
    if (stats) {
        a.N = N; a.num_views = num_views; a.means2D_grad = means2D_grad; a.radii = radii;
        a.xyz_gradient_accum = xyz_gradient_accum; a.denom = denom; a.max_radii2D = max_radii2D;
        blocks += (N + AD_THREADS - 1) / AD_THREADS;
    }
    if (blocks == 0) return GS_OK;
    GS_REQUIRE(blocks < (1ll << 31), "too many workgroups (%lld)", blocks);
    hipLaunchKernelGGL(adam_densify_kernel, dim3((unsigned)blocks), dim3(AD_THREADS), 0, (hipStream_t)stream_, a);
    GS_CHECK_LAUNCH("adam_densify");
    return GS_OK;
}
