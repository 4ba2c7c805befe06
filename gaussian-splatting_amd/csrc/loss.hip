// Fused training loss: w_l1 * L1 + w_ssim * mean SSIM, value and d/d image in three launches.
//
// Restates reference utils/loss_utils.py: l1_loss :10-21, gaussian :38-53 / create_window :56-71
// (11x11 window, sigma 1.5), ssim :74-115 (five zero-padded per-channel convolutions, C1 = 0.01^2,
// C2 = 0.03^2, mean of the map), combined as train.py:141-146 does (w_l1 = 1 - lambda, w_ssim =
// +lambda: the reference's "1 - ssim_loss" is SSIM itself).  The backward is the derivative of those
// lines (autograd in the reference): with S = A1 A2 / (B1 B2) per pixel,
//   dS/dmu1 = 2 mu2 (A2 - A1) / (B1 B2) - 2 mu1 S (B2 - B1) / (B1 B2),  dS/dE11 = -S / B2,
//   dS/dE12 = 2 A1 / (B1 B2),
//   dL/dx = w_ssim / M * (blur(dS/dmu1) + 2 x blur(dS/dE11) + y blur(dS/dE12)) + w_l1 / M * sign(x - y)
// (the window is symmetric, so the adjoint of the zero-padded correlation is the same blur).
//
// The 2-D window is separable: every 32x32 tile stages its 42x42 halo in LDS, blurs rows, then
// columns, each lane producing 4 adjacent outputs per pass from a sliding window held in registers.
// Tiny next to the rasterizer: 8 B read + 12 B written per pixel-channel
// forward, 20 B read + 4 B written backward.  Per-workgroup partial sums and a one-workgroup
// final reduction keep the loss value bit-reproducible (no float atomics).
#include "gsplat_common.h"

namespace {

constexpr int LT = 32;              // tile width
constexpr int LTY = 32;             // tile height (256 lanes, 4 pixels each)
constexpr int LR = 5;               // window radius
constexpr int LH = LT + 2 * LR;     // tile + halo, columns
constexpr int LHY = LTY + 2 * LR;   // tile + halo, rows
constexpr int LP = LH + 1;          // padded LDS row
constexpr int LQ = 4;               // outputs per lane in the row pass (sliding window in registers)
constexpr int LQY = 4;              // outputs per lane in the column pass
constexpr int LTHREADS = 256;
constexpr int LNLOAD = (LHY * LH + LTHREADS - 1) / LTHREADS;   // halo elements per lane (7)

// The reference's gaussian(11, 1.5), float32, bit for bit (tests/golden/loss/_window_1d.npz): the
// values sum to 1 - 3.1e-8, and sigma = E[x^2] - mu^2 cancels so badly that a window summing to
// 1 + 1e-7 instead moves the SSIM of a smooth image by 7e-6.
__device__ constexpr float LW[11] = {0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f,
                                     0x1.b43c3ep-3f,  0x1.10656p-2f,  0x1.b43c3ep-3f, 0x1.bff0fep-4f,
                                     0x1.26eb18p-5f,  0x1.f1fe02p-8f, 0x1.0d956cp-10f};
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

__device__ __forceinline__ float block_sum(float v, float* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += lds[w];
    __syncthreads();
    return s;
}

// Both passes keep a sliding window in registers: a lane produces LQ adjacent outputs from
// LQ + 10 inputs, so the LDS is read 3.5 (rows) / 3.5 (columns) times per output instead of 11.

// dmaps: (3, C, H, W) = dS/dmu1, dS/dE11, dS/dE12 ; partial: (nblocks, 2) = sum |x-y|, sum S
__global__ __launch_bounds__(LTHREADS) void l1_ssim_fwd_kernel(int H, int W, const float* __restrict__ img,
                                                               const float* __restrict__ gt,
                                                               float* __restrict__ dmaps,
                                                               float* __restrict__ partial) {
    __shared__ float sx[LHY][LP], sy[LHY][LP];
    __shared__ float hb[5][LHY][LT + 1];
    __shared__ float red[LTHREADS / 64];
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LTY, ch = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float* px = img + ch * plane;
    const float* py = gt + ch * plane;
    {   // halo tile -> LDS; all of a lane's loads are issued before the first LDS write (one HBM round trip, not 7)
        float vx[LNLOAD], vy[LNLOAD];
#pragma unroll
        for (int i = 0; i < LNLOAD; ++i) {
            const int e = threadIdx.x + i * LTHREADS, r = e / LH, c = e % LH;
            const int yy = y0 + r - LR, xx = x0 + c - LR;
            const bool in = e < LHY * LH && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const size_t o = (size_t)(in ? yy : 0) * W + (in ? xx : 0);
            vx[i] = px[o];
            vy[i] = py[o];
            vx[i] = in ? vx[i] : 0.0f;     // conv2d zero padding
            vy[i] = in ? vy[i] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < LNLOAD; ++i) {
            const int e = threadIdx.x + i * LTHREADS, r = e / LH, c = e % LH;
            if (e < LHY * LH) {
                sx[r][c] = vx[i];
                sy[r][c] = vy[i];
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < LHY * (LT / LQ); e += LTHREADS) {   // blur rows
        const int r = e / (LT / LQ), c0 = (e % (LT / LQ)) * LQ;
        float a[LQ + 10], b[LQ + 10], aa[LQ + 10], bb[LQ + 10], ab[LQ + 10];
#pragma unroll
        for (int k = 0; k < LQ + 10; ++k) {
            a[k] = sx[r][c0 + k];
            b[k] = sy[r][c0 + k];
            aa[k] = a[k] * a[k];       // products once per input, not once per tap
            bb[k] = b[k] * b[k];
            ab[k] = a[k] * b[k];
        }
#pragma unroll
        for (int o = 0; o < LQ; ++o) {
            float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float g = LW[k];
                m1 += g * a[o + k];
                m2 += g * b[o + k];
                e11 += g * aa[o + k];
                e22 += g * bb[o + k];
                e12 += g * ab[o + k];
            }
            hb[0][r][c0 + o] = m1; hb[1][r][c0 + o] = m2; hb[2][r][c0 + o] = e11; hb[3][r][c0 + o] = e22;
            hb[4][r][c0 + o] = e12;
        }
    }
    __syncthreads();
    // blur columns: lane = (column, group of LQY rows)
    const int tx = threadIdx.x % LT, r0 = (threadIdx.x / LT) * LQY;
    float q[LQY][5];
#pragma unroll
    for (int o = 0; o < LQY; ++o)
#pragma unroll
        for (int m = 0; m < 5; ++m) q[o][m] = 0.f;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        float col[LQY + 10];
#pragma unroll
        for (int k = 0; k < LQY + 10; ++k) col[k] = hb[m][r0 + k][tx];
#pragma unroll
        for (int o = 0; o < LQY; ++o)
#pragma unroll
            for (int k = 0; k < 11; ++k) q[o][m] += LW[k] * col[o + k];
    }
    float l1 = 0.f, Ssum = 0.f;
    const int xx = x0 + tx;
    const size_t CHW = gridDim.z * plane;
#pragma unroll
    for (int o = 0; o < LQY; ++o) {
        const int yy = y0 + r0 + o;
        if (yy < H && xx < W) {
            const float mu1 = q[o][0], mu2 = q[o][1];
            const float s11 = q[o][2] - mu1 * mu1, s22 = q[o][3] - mu2 * mu2, s12 = q[o][4] - mu1 * mu2;
            const float A1 = 2.0f * mu1 * mu2 + SSIM_C1, A2 = 2.0f * s12 + SSIM_C2;
            const float B1 = mu1 * mu1 + mu2 * mu2 + SSIM_C1, B2 = s11 + s22 + SSIM_C2;
            const float inv = 1.0f / (B1 * B2);
            const float S = A1 * A2 * inv;
            const size_t off = ch * plane + (size_t)yy * W + xx;
            dmaps[off] = 2.0f * mu2 * (A2 - A1) * inv - 2.0f * mu1 * S * (B2 - B1) * inv;
            dmaps[CHW + off] = -S / B2;
            dmaps[2 * CHW + off] = 2.0f * A1 * inv;
            Ssum += S;
            l1 += fabsf(sx[r0 + o + LR][tx + LR] - sy[r0 + o + LR][tx + LR]);
        }
    }
    const float sl1 = block_sum(l1, red), sS = block_sum(Ssum, red);
    if (threadIdx.x == 0) {
        const size_t bi = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * bi] = sl1;
        partial[2 * bi + 1] = sS;
    }
}

// one workgroup: loss_out = (loss, l1, ssim)
__global__ __launch_bounds__(1024) void l1_ssim_reduce_kernel(const float* __restrict__ partial, int nblocks, float M,
                                                              float w_l1, float w_ssim, float* __restrict__ loss_out) {
    __shared__ float red[16];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 1024) {
        const float2 p = reinterpret_cast<const float2*>(partial)[i];
        a += p.x;
        b += p.y;
    }
    const float l1 = block_sum(a, red) / M, ss = block_sum(b, red) / M;
    if (threadIdx.x == 0) {
        loss_out[0] = w_l1 * l1 + w_ssim * ss;
        loss_out[1] = l1;
        loss_out[2] = ss;
    }
}

__global__ __launch_bounds__(LTHREADS) void l1_ssim_bwd_kernel(int H, int W, const float* __restrict__ img,
                                                               const float* __restrict__ gt,
                                                               const float* __restrict__ dmaps, float w_l1_M,
                                                               float w_ssim_M, float* __restrict__ grad) {
    __shared__ float sd[3][LHY][LP];
    __shared__ float hb[3][LHY][LT + 1];
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LTY, ch = blockIdx.z;
    const size_t plane = (size_t)H * W, CHW = gridDim.z * plane;
    {   // as in the forward: every load in flight before the first LDS write
        float v[LNLOAD][3];
#pragma unroll
        for (int i = 0; i < LNLOAD; ++i) {
            const int e = threadIdx.x + i * LTHREADS, r = e / LH, c = e % LH;
            const int yy = y0 + r - LR, xx = x0 + c - LR;
            const bool in = e < LHY * LH && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const size_t o = ch * plane + (size_t)(in ? yy : 0) * W + (in ? xx : 0);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                v[i][a] = dmaps[a * CHW + o];
                v[i][a] = in ? v[i][a] : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < LNLOAD; ++i) {
            const int e = threadIdx.x + i * LTHREADS, r = e / LH, c = e % LH;
            if (e < LHY * LH) {
#pragma unroll
                for (int a = 0; a < 3; ++a) sd[a][r][c] = v[i][a];
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < LHY * (LT / LQ); e += LTHREADS) {
        const int r = e / (LT / LQ), c0 = (e % (LT / LQ)) * LQ;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v[LQ + 10];
#pragma unroll
            for (int k = 0; k < LQ + 10; ++k) v[k] = sd[a][r][c0 + k];
#pragma unroll
            for (int o = 0; o < LQ; ++o) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) s += LW[k] * v[o + k];
                hb[a][r][c0 + o] = s;
            }
        }
    }
    __syncthreads();
    const int tx = threadIdx.x % LT, r0 = (threadIdx.x / LT) * LQY;
    float q[LQY][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float col[LQY + 10];
#pragma unroll
        for (int k = 0; k < LQY + 10; ++k) col[k] = hb[a][r0 + k][tx];
#pragma unroll
        for (int o = 0; o < LQY; ++o) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) s += LW[k] * col[o + k];
            q[o][a] = s;
        }
    }
    const int xx = x0 + tx;
#pragma unroll
    for (int o = 0; o < LQY; ++o) {
        const int yy = y0 + r0 + o;
        if (yy < H && xx < W) {
            const size_t off = ch * plane + (size_t)yy * W + xx;
            const float x = img[off], y = gt[off], d = x - y;
            const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);   // torch.abs: gradient 0 at 0
            grad[off] = w_ssim_M * (q[o][0] + 2.0f * x * q[o][1] + y * q[o][2]) + w_l1_M * sgn;
        }
    }
}

inline size_t loss_nblocks(int C, int H, int W) {
    return (size_t)C * gs_div_up(H, LTY) * gs_div_up(W, LT);
}

}  // namespace

extern "C" size_t gs_l1_ssim_workspace_bytes(int C, int H, int W) {
    if (C <= 0 || H <= 0 || W <= 0) return 0;
    return (((3 * (size_t)C * H * W + 1) & ~(size_t)1) + 2 * loss_nblocks(C, H, W)) * sizeof(float);
}

// loss_out (device, 3 floats) = (w_l1 * L1 + w_ssim * SSIM, L1, SSIM); grad (C,H,W, may be NULL) = d loss / d img.
extern "C" int gs_l1_ssim(int C, int H, int W, const float* img, const float* gt, float w_l1, float w_ssim,
                          float* loss_out, float* grad, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GS_REQUIRE(C > 0 && C <= 65535 && H > 0 && W > 0, "bad image shape (%d,%d,%d)", C, H, W);
    GS_REQUIRE(gs_div_up(H, LTY) <= 65535, "image too tall");
    GS_REQUIRE(img && gt && loss_out && workspace, "null argument");
    GS_REQUIRE(workspace_bytes >= gs_l1_ssim_workspace_bytes(C, H, W), "loss workspace too small");
    GS_REQUIRE(((uintptr_t)workspace & 7) == 0, "loss workspace must be 8-byte aligned");
    float* dmaps = (float*)workspace;
    float* partial = dmaps + ((3 * (size_t)C * H * W + 1) & ~(size_t)1);   // 8-byte aligned pairs
    const dim3 grid(gs_div_up(W, LT), gs_div_up(H, LTY), C), block(LTHREADS);
    const float M = (float)((double)C * H * W);
    hipLaunchKernelGGL(l1_ssim_fwd_kernel, grid, block, 0, stream, H, W, img, gt, dmaps, partial);
    GS_CHECK_LAUNCH("l1_ssim_fwd");
    hipLaunchKernelGGL(l1_ssim_reduce_kernel, dim3(1), dim3(1024), 0, stream, partial, (int)loss_nblocks(C, H, W), M,
                       w_l1, w_ssim, loss_out);
    GS_CHECK_LAUNCH("l1_ssim_reduce");
    if (grad) {
        hipLaunchKernelGGL(l1_ssim_bwd_kernel, grid, block, 0, stream, H, W, img, gt, dmaps, w_l1 / M, w_ssim / M, grad);
        GS_CHECK_LAUNCH("l1_ssim_bwd");
    }
    return GS_OK;
}
