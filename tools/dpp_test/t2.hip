// Standalone check of the select-free reduce-scatter used by render_bwd (see render.hip).
//   hipcc --offload-arch=gfx950 -O3 t2.hip -o t2 && ./t2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

// 10 values per lane -> every lane of quad Q holds the 64-lane total of one value:
//   lane (bits 5..2) -> value: see the table printed by main().
__device__ __forceinline__ float wave_reduce_scatter10(float (&v)[10]) {
    float o0, o1, o2, o3, o4, p0, p1, p2;
    asm volatile(
        "s_nop 1\n\t"
        // step 1, lane^8: banks 0,1 keep v[j], banks 2,3 keep v[j+5]
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
        // step 2, lane^4: banks 0,2 keep (o0,o1), banks 1,3 keep (o2,o3); o4 everywhere
        "v_add_f32_dpp %5, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %6, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %7, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(p0), "=&v"(p1), "=&v"(p2)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
          "v"(v[9]));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    // step 3, lane^16: even rows keep p0, odd rows keep p1; p2 everywhere
    const u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
    const float q0 = __uint_as_float(a.x) + __uint_as_float(a.y);
    const u2 b = __builtin_amdgcn_permlane16_swap(__float_as_uint(p2), __float_as_uint(p2), false, false);
    const float q1 = __uint_as_float(b.x) + __uint_as_float(b.y);
    // step 4, lane^32: lanes 0..31 keep q0, lanes 32..63 keep q1
    const u2 c = __builtin_amdgcn_permlane32_swap(__float_as_uint(q0), __float_as_uint(q1), false, false);
    float r = __uint_as_float(c.x) + __uint_as_float(c.y);
    // step 5: sum over the quad
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(r));
    return r;
}

__global__ void k(const float* in, float* out) {
    float v[10];
    for (int i = 0; i < 10; ++i) v[i] = in[i * 64 + threadIdx.x];
    out[threadIdx.x] = wave_reduce_scatter10(v);
}
int main() {
    float h[640], *d, *o, r[64];
    for (int i = 0; i < 10; ++i) for (int l = 0; l < 64; ++l) h[i*64+l] = (float)((i+1) * 1000 + ((l * 7 + i * 3) % 64));
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1,64>>>(d, o);
    hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
    double s[10];
    for (int i = 0; i < 10; ++i) { s[i] = 0; for (int l = 0; l < 64; ++l) s[i] += h[i*64+l]; printf("expect v%d = %.0f\n", i, s[i]); }
    // expected map: value -> first lane
    const int lane_of[10] = {0, 16, 4, 20, 32, 8, 24, 12, 28, 40};
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        int which = -1;
        for (int i = 0; i < 10; ++i) if (r[l] == (float)s[i]) which = i;
        printf("lane %2d: %.0f (v%d)\n", l, r[l], which);
    }
    for (int i = 0; i < 10; ++i) if (r[lane_of[i]] != (float)s[i]) { printf("MISMATCH v%d at lane %d\n", i, lane_of[i]); bad = 1; }
    printf(bad ? "FAIL\n" : "PASS\n");
    return bad;
}
