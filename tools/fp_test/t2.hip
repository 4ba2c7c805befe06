// Calls gs_adam_densify_step on three values and prints what it computed (p0 = 0, step size 1 -> p = -m/d).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "../../include/gsplat.h"
int main() {
    const int n = 3;
    const float h[n] = {0.0013986037f, 0.0016567738f, 0.00015035813f};
    float *p, *g, *m, *v, out[n];
    (void)hipMalloc(&p, 64); (void)hipMalloc(&g, 64); (void)hipMalloc(&m, 64); (void)hipMalloc(&v, 64);
    (void)hipMemset(p, 0, 64); (void)hipMemset(m, 0, 64); (void)hipMemset(v, 0, 64);
    (void)hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
    GsAdamGroup grp = {p, g, m, v, n, 0, 1, 0, 1.0 - pow(0.9, 1.0)};
    int rc = gs_adam_densify_step(1, &grp, 0.9, 0.999, 1e-15, 1, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    (void)hipDeviceSynchronize();
    printf("rc %d\n", rc);
    (void)hipMemcpy(out, p, sizeof(out), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("p %.9g ", out[i]);
    (void)hipMemcpy(out, m, sizeof(out), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("m %.9g ", out[i]);
    (void)hipMemcpy(out, v, sizeof(out), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("v %.9g ", out[i]);
    printf("\n");
    return 0;
}
