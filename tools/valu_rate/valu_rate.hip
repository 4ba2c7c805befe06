// Micro-benchmark: how many wave64 v_fma_f32 does one SIMD of an MI355X issue per cycle, as a function of the waves
// resident on it?  (Is the VALU issue peak one instruction per 4 cycles or per 2?)  Build: make -C tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a2), "v"(b2))
typedef float f2 __attribute__((ext_vector_type(2)));

// 128 v_fma_f32 per iteration, 8 independent chains
__global__ void __launch_bounds__(256) fma_kernel(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { FMA(x0); FMA(x1); FMA(x2); FMA(x3); FMA(x4); FMA(x5); FMA(x6); FMA(x7); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// 128 v_pk_fma_f32 per iteration (two FMAs per lane each), 8 independent chains
__global__ void __launch_bounds__(256) pkfma_kernel(float* out, int iters, float a, float b) {
    f2 a2 = {a, a}, b2 = {b, b};
    f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { PKFMA(x0); PKFMA(x1); PKFMA(x2); PKFMA(x3); PKFMA(x4); PKFMA(x5); PKFMA(x6); PKFMA(x7); }
    }
    f2 t = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = t.x + t.y;
}

#define KERNEL1(name, ASM)                                                                                      \
    __global__ void __launch_bounds__(256) name(float* out, int iters, float a, float b) {                      \
        float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6,    \
              x7 = x0 + 7;                                                                                       \
        for (int i = 0; i < iters; ++i) {                                                                        \
            _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                                     \
                asm volatile(ASM : "+v"(x0) : "v"(a)); asm volatile(ASM : "+v"(x1) : "v"(a));                  \
                asm volatile(ASM : "+v"(x2) : "v"(a)); asm volatile(ASM : "+v"(x3) : "v"(a));                  \
                asm volatile(ASM : "+v"(x4) : "v"(a)); asm volatile(ASM : "+v"(x5) : "v"(a));                  \
                asm volatile(ASM : "+v"(x6) : "v"(a)); asm volatile(ASM : "+v"(x7) : "v"(a));                  \
            }                                                                                                    \
        }                                                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                      \
    }
KERNEL1(dpp_kernel, "v_add_f32_dpp %0, %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf")
KERNEL1(dpp_quad_kernel, "v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL1(dpp_shr_kernel, "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL1(dpp_mirror_kernel, "v_add_f32_dpp %0, %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf")
KERNEL1(dpp_bank_kernel, "v_add_f32_dpp %0, %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3")
KERNEL1(mov_dpp_kernel, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL1(add_kernel, "v_add_f32 %0, %0, %1")
KERNEL1(exp_kernel, "v_exp_f32 %0, %0")
KERNEL1(rcp_kernel, "v_rcp_f32 %0, %0")
KERNEL1(mul_kernel, "v_mul_f32 %0, %0, %1")
KERNEL1(pkmax_kernel, "v_pk_max_u16 %0, %0, %1")
KERNEL1(cndmask_kernel, "v_cndmask_b32 %0, %0, %1, vcc")

// v_cndmask with an SGPR-pair mask (e64) and v_cmp writing VCC: the selects and compares of the render loops
__global__ void __launch_bounds__(256) cndmask_sgpr_kernel(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const unsigned long long m = __ballot((threadIdx.x & 1) == (b > 0.f));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#define CM(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(m))
            CM(x0); CM(x1); CM(x2); CM(x3); CM(x4); CM(x5); CM(x6); CM(x7);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void __launch_bounds__(256) cmp_kernel(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#define CP(x) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(x), "v"(a) : "vcc")
            CP(x0); CP(x1); CP(x2); CP(x3); CP(x4); CP(x5); CP(x6); CP(x7);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void __launch_bounds__(256) cndmask_vcc_kernel(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        asm volatile("s_mov_b64 vcc, 0x55555555" : : : "vcc");
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#define CV(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : )
            CV(x0); CV(x1); CV(x2); CV(x3); CV(x4); CV(x5); CV(x6); CV(x7);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// one v_cndmask_b32_e32 (VCC mask) among seven v_mul_f32: its cost inside a mixed stream
__global__ void __launch_bounds__(256) mixed_cndmask_kernel(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        asm volatile("s_mov_b64 vcc, 0x55555555" : : : "vcc");
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#define ML(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a))
            ML(x0); ML(x1); ML(x2); CV(x3); ML(x4); ML(x5); ML(x6); ML(x7);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// the same with the e64 form (mask in an SGPR pair)
__global__ void __launch_bounds__(256) mixed_cndmask64_kernel(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const unsigned long long m = __ballot((threadIdx.x & 1) == (b > 0.f));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { ML(x0); ML(x1); ML(x2); CM(x3); ML(x4); ML(x5); ML(x6); ML(x7); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", p.name, cus, ghz);
    float* out;
    CHECK(hipMalloc(&out, (size_t)cus * 16 * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 5000;
    struct { const char* name; void (*k)(float*, int, float, float); int flops; } K[] = {
        {"v_fma_f32      ", fma_kernel, 2}, {"v_pk_fma_f32   ", pkfma_kernel, 4}, {"v_mul_f32      ", mul_kernel, 1},
        {"v_add_f32_dpp ror4", dpp_kernel, 1}, {"v_add_f32_dpp quad_perm", dpp_quad_kernel, 1},
        {"v_add_f32_dpp row_shr:1", dpp_shr_kernel, 1}, {"v_add_f32_dpp half_mirror", dpp_mirror_kernel, 1},
        {"v_add_f32_dpp ror8 bank3", dpp_bank_kernel, 1}, {"v_mov_b32_dpp quad_perm", mov_dpp_kernel, 1},
        {"v_add_f32      ", add_kernel, 1}, {"v_exp_f32      ", exp_kernel, 1},   {"v_rcp_f32      ", rcp_kernel, 1},
        {"v_pk_max_u16   ", pkmax_kernel, 1}, {"v_cndmask_b32 vcc (old)", cndmask_kernel, 1}, {"v_cndmask_b32_e32 vcc set", cndmask_vcc_kernel, 1},
        {"v_cndmask_b32_e64 sgpr", cndmask_sgpr_kernel, 1}, {"v_cmp_gt_f32 vcc", cmp_kernel, 1},
        {"7 v_mul + 1 v_cndmask_e32 vcc (per instruction)", mixed_cndmask_kernel, 1},
        {"7 v_mul + 1 v_cndmask_e64 sgpr (per instruction)", mixed_cndmask64_kernel, 1}};
    for (auto& kk : K)
        for (int wg_per_cu : {1, 6}) {        // 256-thread workgroups = 1 wave per SIMD each
            const int grid = cus * wg_per_cu;
            kk.k<<<grid, 256>>>(out, 100, 1.0001f, 0.5f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            kk.k<<<grid, 256>>>(out, iters, 1.0001f, 0.5f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double insts_per_simd = (double)iters * 128 * wg_per_cu;      // wave-instructions per SIMD
            const double cycles = ms * 1e-3 * ghz * 1e9;
            printf("%s waves/SIMD %d: %7.3f ms, %.2f cycles (at %.1f GHz) per wave64 instruction per SIMD, %.0f G wave-inst/s\n",
                   kk.name, wg_per_cu, ms, cycles / insts_per_simd, ghz, insts_per_simd * cus * 4 / (ms * 1e-3) / 1e9);
        }
    return 0;
}
