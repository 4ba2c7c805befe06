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
// The 2-D window is separable: every 16x16 tile stages its 26x26 halo in LDS, blurs rows, then
// columns.  HBM-bound and tiny next to the rasterizer: 8 B read + 12 B written per pixel-channel
// forward, 20 B read + 4 B written backward.  Per-workgroup partial sums and a one-workgroup
// final reduction keep the loss value bit-reproducible (no float atomics).
#include "gsplat_common.h"

namespace {

constexpr int LT = 16;              // tile edge
constexpr int LR = 5;               // window radius
constexpr int LH = LT + 2 * LR;     // 26: tile + halo
constexpr int LP = LH + 1;          // padded LDS row

// The reference's gaussian(11, 1.5), float32, bit for bit (tests/golden/loss/_window_1d.npz): the
// values sum to 1 - 3.1e-8, and sigma = E[x^2] - mu^2 cancels so badly that a window summing to
// 1 + 1e-7 instead moves the SSIM of a smooth image by 7e-6.
__device__ constexpr float LW[11] = {0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f,
                                     0x1.b43c3ep-3f,  0x1.10656p-2f,  0x1.b43c3ep-3f, 0x1.bff0fep-4f,
                                     0x1.26eb18p-5f,  0x1.f1fe02p-8f, 0x1.0d956cp-10f};
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

__device__ __forceinline__ float block_sum(float v, float* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    const float s = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    __syncthreads();
    return s;
}

// dmaps: (3, C, H, W) = dS/dmu1, dS/dE11, dS/dE12 ; partial: (nblocks, 2) = sum |x-y|, sum S
__global__ __launch_bounds__(LT * LT) void l1_ssim_fwd_kernel(int H, int W, const float* __restrict__ img,
                                                              const float* __restrict__ gt,
                                                              float* __restrict__ dmaps,
                                                              float* __restrict__ partial) {
    __shared__ float sx[LH][LP], sy[LH][LP];
    __shared__ float hb[5][LH][LT];
    __shared__ float red[4];
    const int tx = threadIdx.x % LT, ty = threadIdx.x / LT;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, ch = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float* px = img + ch * plane;
    const float* py = gt + ch * plane;
    for (int e = threadIdx.x; e < LH * LH; e += LT * LT) {
        const int r = e / LH, c = e % LH;
        const int yy = y0 + r - LR, xx = x0 + c - LR;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        sx[r][c] = in ? px[(size_t)yy * W + xx] : 0.0f;     // conv2d zero padding
        sy[r][c] = in ? py[(size_t)yy * W + xx] : 0.0f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < LH * LT; e += LT * LT) {   // blur rows
        const int r = e / LT, c = e % LT;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float a = sx[r][c + k], b = sy[r][c + k], g = LW[k];
            m1 += g * a;
            m2 += g * b;
            e11 += g * (a * a);
            e22 += g * (b * b);
            e12 += g * (a * b);
        }
        hb[0][r][c] = m1; hb[1][r][c] = m2; hb[2][r][c] = e11; hb[3][r][c] = e22; hb[4][r][c] = e12;
    }
    __syncthreads();
    float q[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) {                            // blur columns
        const float g = LW[k];
#pragma unroll
        for (int a = 0; a < 5; ++a) q[a] += g * hb[a][ty + k][tx];
    }
    const int yy = y0 + ty, xx = x0 + tx;
    const bool in = yy < H && xx < W;
    float l1 = 0.f, S = 0.f;
    if (in) {
        const float mu1 = q[0], mu2 = q[1];
        const float s11 = q[2] - mu1 * mu1, s22 = q[3] - mu2 * mu2, s12 = q[4] - mu1 * mu2;
        const float A1 = 2.0f * mu1 * mu2 + SSIM_C1, A2 = 2.0f * s12 + SSIM_C2;
        const float B1 = mu1 * mu1 + mu2 * mu2 + SSIM_C1, B2 = s11 + s22 + SSIM_C2;
        const float inv = 1.0f / (B1 * B2);
        S = A1 * A2 * inv;
        const size_t o = ch * plane + (size_t)yy * W + xx, CHW = gridDim.z * plane;
        dmaps[o] = 2.0f * mu2 * (A2 - A1) * inv - 2.0f * mu1 * S * (B2 - B1) * inv;
        dmaps[CHW + o] = -S / B2;
        dmaps[2 * CHW + o] = 2.0f * A1 * inv;
        l1 = fabsf(sx[ty + LR][tx + LR] - sy[ty + LR][tx + LR]);
    }
    const float sl1 = block_sum(l1, red), sS = block_sum(S, red);
    if (threadIdx.x == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * b] = sl1;
        partial[2 * b + 1] = sS;
    }
}

// one workgroup: loss_out = (loss, l1, ssim)
__global__ __launch_bounds__(256) void l1_ssim_reduce_kernel(const float* __restrict__ partial, int nblocks, float M,
                                                             float w_l1, float w_ssim, float* __restrict__ loss_out) {
    __shared__ float red[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        a += partial[2 * i];
        b += partial[2 * i + 1];
    }
    const float l1 = block_sum(a, red) / M, ss = block_sum(b, red) / M;
    if (threadIdx.x == 0) {
        loss_out[0] = w_l1 * l1 + w_ssim * ss;
        loss_out[1] = l1;
        loss_out[2] = ss;
    }
}

__global__ __launch_bounds__(LT * LT) void l1_ssim_bwd_kernel(int H, int W, const float* __restrict__ img,
                                                              const float* __restrict__ gt,
                                                              const float* __restrict__ dmaps, float w_l1_M,
                                                              float w_ssim_M, float* __restrict__ grad) {
    __shared__ float sd[3][LH][LP];
    __shared__ float hb[3][LH][LT];
    const int tx = threadIdx.x % LT, ty = threadIdx.x / LT;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, ch = blockIdx.z;
    const size_t plane = (size_t)H * W, CHW = gridDim.z * plane;
    for (int e = threadIdx.x; e < LH * LH; e += LT * LT) {
        const int r = e / LH, c = e % LH;
        const int yy = y0 + r - LR, xx = x0 + c - LR;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const size_t o = ch * plane + (size_t)(in ? yy : 0) * W + (in ? xx : 0);
#pragma unroll
        for (int a = 0; a < 3; ++a) sd[a][r][c] = in ? dmaps[a * CHW + o] : 0.0f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < LH * LT; e += LT * LT) {
        const int r = e / LT, c = e % LT;
        float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float g = LW[k];
#pragma unroll
            for (int a = 0; a < 3; ++a) s[a] += g * sd[a][r][c + k];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) hb[a][r][c] = s[a];
    }
    __syncthreads();
    float q[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const float g = LW[k];
#pragma unroll
        for (int a = 0; a < 3; ++a) q[a] += g * hb[a][ty + k][tx];
    }
    const int yy = y0 + ty, xx = x0 + tx;
    if (yy < H && xx < W) {
        const size_t o = ch * plane + (size_t)yy * W + xx;
        const float x = img[o], y = gt[o], d = x - y;
        const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);   // torch.abs: gradient 0 at 0
        grad[o] = w_ssim_M * (q[0] + 2.0f * x * q[1] + y * q[2]) + w_l1_M * sgn;
    }
}

inline size_t loss_nblocks(int C, int H, int W) {
    return (size_t)C * gs_div_up(H, LT) * gs_div_up(W, LT);
}

}  // namespace

extern "C" size_t gs_l1_ssim_workspace_bytes(int C, int H, int W) {
    if (C <= 0 || H <= 0 || W <= 0) return 0;
    return (3 * (size_t)C * H * W + 2 * loss_nblocks(C, H, W)) * sizeof(float);
}

// loss_out (device, 3 floats) = (w_l1 * L1 + w_ssim * SSIM, L1, SSIM); grad (C,H,W, may be NULL) = d loss / d img.
extern "C" int gs_l1_ssim(int C, int H, int W, const float* img, const float* gt, float w_l1, float w_ssim,
                          float* loss_out, float* grad, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GS_REQUIRE(C > 0 && C <= 65535 && H > 0 && W > 0, "bad image shape (%d,%d,%d)", C, H, W);
    GS_REQUIRE(gs_div_up(H, LT) <= 65535, "image too tall");
    GS_REQUIRE(img && gt && loss_out && workspace, "null argument");
    GS_REQUIRE(workspace_bytes >= gs_l1_ssim_workspace_bytes(C, H, W), "loss workspace too small");
    float* dmaps = (float*)workspace;
    float* partial = dmaps + 3 * (size_t)C * H * W;
    const dim3 grid(gs_div_up(W, LT), gs_div_up(H, LT), C), block(LT * LT);
    const float M = (float)((double)C * H * W);
    hipLaunchKernelGGL(l1_ssim_fwd_kernel, grid, block, 0, stream, H, W, img, gt, dmaps, partial);
    GS_CHECK_LAUNCH("l1_ssim_fwd");
    hipLaunchKernelGGL(l1_ssim_reduce_kernel, dim3(1), dim3(256), 0, stream, partial, (int)loss_nblocks(C, H, W), M,
                       w_l1, w_ssim, loss_out);
    GS_CHECK_LAUNCH("l1_ssim_reduce");
    if (grad) {
        hipLaunchKernelGGL(l1_ssim_bwd_kernel, grid, block, 0, stream, H, W, img, gt, dmaps, w_l1 / M, w_ssim / M, grad);
        GS_CHECK_LAUNCH("l1_ssim_bwd");
    }
    return GS_OK;
}
