// Standalone check of fp32 sqrt / divide rounding on gfx950 for the optimizer kernel's formulas.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#pragma clang fp contract(off)
__global__ void k(const float* in, float* out, float bc, float eps) {
    const float g = in[threadIdx.x];
    const float m = __fmaf_rn(0.1f, g, 0.0f);
    const float v = __fmaf_rn(0.001f, g * g, 0.0f);
    const float s = sqrtf(v);
    const float d1 = __fdiv_rn(s, bc);
    const float d = d1 + eps;
    const float q = __fdiv_rn(m, d);
    float* o = out + 8 * threadIdx.x;
    o[0] = m; o[1] = v; o[2] = s; o[3] = d1; o[4] = d; o[5] = q; o[6] = s / bc; o[7] = m / d;
}
int main() {
    const float h[3] = {0.0013986037f, 0.0016567738f, 0.00015035813f};
    float *din, *dout, out[24];
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(out));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    const float bc = (float)sqrt(1.0 - 0.999), eps = 1e-15f;
    hipLaunchKernelGGL(k, dim3(1), dim3(3), 0, 0, din, dout, bc, eps);
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    for (int i = 0; i < 3; ++i) {
        const float g = h[i];
        const float m = fmaf(0.1f, g, 0.0f), v = fmaf(0.001f, g * g, 0.0f), s = sqrtf(v), d1 = s / bc, d = d1 + eps, q = m / d;
        printf("g %.9g\n  gpu m %.9g v %.9g s %.9g d1 %.9g d %.9g q %.9g | %.9g %.9g\n  cpu m %.9g v %.9g s %.9g d1 %.9g d %.9g q %.9g\n",
               g, out[8*i], out[8*i+1], out[8*i+2], out[8*i+3], out[8*i+4], out[8*i+5], out[8*i+6], out[8*i+7], m, v, s, d1, d, q);
    }
    printf("bc %.9g\n", bc);
    return 0;
}
