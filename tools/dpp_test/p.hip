#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out) {
    float x = (float)threadIdx.x;
    // rows: fold 64 -> every lane holds sum over l, l^16, l^32, l^48
    u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float s = __uint_as_float(r.x) + __uint_as_float(r.y);
    u2 q = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    out[threadIdx.x] = __uint_as_float(q.x) + __uint_as_float(q.y);
}
int main() { float *o, r[64]; hipMalloc(&o, 256); k<<<1,64>>>(o); hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("%d:%.0f ", l, r[l]); printf("\n"); return 0; }
