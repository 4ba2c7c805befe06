// Standalone check of the 16-lane row reduce-scatter used by render_bwd (see render.hip).
//   hipcc --offload-arch=gfx950 -O3 t3.hip -o t3 && ./t3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ float row_reduce_scatter10(const float (&v)[10], int t) {
    float o0, o1, o2, o3, o4, p0, p1, p2;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %6, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %7, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(p0), "=&v"(p1), "=&v"(p2)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
          "v"(v[9]));
    return t == 0 ? p0 : (t == 1 ? p1 : p2);
}

__global__ void k(const float* in, float* out) {
    float v[10];
    for (int i = 0; i < 10; ++i) v[i] = in[i * 64 + threadIdx.x];
    out[threadIdx.x] = row_reduce_scatter10(v, threadIdx.x & 3);
}

int main() {
    float h[640], *d, *o, r[64];
    for (int i = 0; i < 10; ++i) for (int l = 0; l < 64; ++l) h[i * 64 + l] = (float)((i + 1) * 1000 + ((l * 7 + i * 3) % 64));
    if (hipMalloc(&d, sizeof(h)) != hipSuccess || hipMalloc(&o, 256) != hipSuccess) return 2;
    if (hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return 2;
    k<<<1, 64>>>(d, o);
    if (hipMemcpy(r, o, 256, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l >> 4, t = l & 3, bank = (l >> 2) & 3;
        int vi = -1;   // the table in render.hip
        if (t == 0) vi = ((bank & 1) ? 2 : 0) + 5 * (bank >> 1);
        else if (t == 1) vi = ((bank & 1) ? 3 : 1) + 5 * (bank >> 1);
        else if (t == 2) vi = 4 + 5 * (bank >> 1);
        if (vi < 0) continue;
        double s = 0;
        for (int q = 0; q < 16; ++q) s += h[vi * 64 + row * 16 + q];
        if (r[l] != (float)s) { printf("lane %d: got %.0f want %.0f (v%d of row %d)\n", l, r[l], s, vi, row); bad = 1; }
    }
    printf(bad ? "FAIL\n" : "PASS\n");
    return bad;
}
