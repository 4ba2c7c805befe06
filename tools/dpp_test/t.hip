#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ float wave_reduce_scatter10(float (&v)[10]) {
    float t0, out;
    const uint64_t m0 = 0xAAAAAAAAAAAAAAAAull, m1 = 0xCCCCCCCCCCCCCCCCull;
    asm volatile(
        "s_nop 1\n\t"
        // step A
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %9, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32 %0, %0, %5, %12\n\t"
        "v_cndmask_b32 %1, %1, %6, %12\n\t"
        "v_cndmask_b32 %2, %2, %7, %12\n\t"
        "v_cndmask_b32 %3, %3, %8, %12\n\t"
        "v_cndmask_b32 %4, %4, %9, %12\n\t"
        // step B: (a0,a3) (a1,a4) a2
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32 %0, %0, %3, %13\n\t"
        "v_cndmask_b32 %1, %1, %4, %13\n\t"
        // step C: c1 = b2 + b2[l^8]; c0 = (bit3 ? b1 : b0) + same of lane^8
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %10, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %10, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        // step D
        "v_add_f32_dpp %11, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %11, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "s_nop 0"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
          "+v"(v[8]), "+v"(v[9]), "=&v"(t0), "=&v"(out)
        : "s"(m0), "s"(m1));
    // fold the four 16-lane rows lane-wise (row_bcast would mix lanes that carry different values)
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(out), __float_as_uint(out), false, false);
    const float s = __uint_as_float(r.x) + __uint_as_float(r.y);
    const u2 q = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(q.x) + __uint_as_float(q.y);
}

__global__ void k(const float* in, float* out) {
    float v[10];
    for (int i = 0; i < 10; ++i) v[i] = in[i * 64 + threadIdx.x];
    out[threadIdx.x] = wave_reduce_scatter10(v);
}
int main() {
    float h[640], *d, *o, r[64];
    for (int i = 0; i < 10; ++i) for (int l = 0; l < 64; ++l) h[i*64+l] = (float)((i+1) * 1000 + l);
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1,64>>>(d, o);
    hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 10; ++i) { double s = 0; for (int l = 0; l < 64; ++l) s += h[i*64+l]; printf("expect v%d = %.0f\n", i, s); }
    for (int l = 0; l < 64; ++l) printf("lane %d: %.0f\n", l, r[l]);
    return 0;
}
