// Is 1.f / a reproduced bit for bit by v_rcp_f32 followed by two Newton steps (4 FMAs) instead
// of the three (6 FMAs) of the compiler's division expansion?  Exhaustive over every float
// with 5e-8 <= |a| < 1e18.
//   hipcc -O2 --offload-arch=gfx950 rcp_steps.hip -o rcp_steps && ./rcp_steps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void check(uint32_t lo, uint32_t hi, unsigned long long *bad) {
    unsigned long long n4 = 0, n2 = 0, n4b = 0;
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * blockDim.x) {
        for (int sgn = 0; sgn < 2; sgn++) {
            const float a = __uint_as_float((uint32_t)b | (sgn ? 0x80000000u : 0u));
            const float ref = 1.f / a;
            const float r0 = __builtin_amdgcn_rcpf(a);
            const float e0 = __builtin_fmaf(-a, r0, 1.f);
            const float r1 = __builtin_fmaf(e0, r0, r0);
            const float e1 = __builtin_fmaf(-a, r1, 1.f);
            const float r2 = __builtin_fmaf(e1, r1, r1);
            // variant: second step refines with r0 * (1 + e0 + e0^2)
            const float r2b = __builtin_fmaf(e1, r0, r1);
            n2 += (__float_as_uint(r1) != __float_as_uint(ref));
            n4 += (__float_as_uint(r2) != __float_as_uint(ref));
            n4b += (__float_as_uint(r2b) != __float_as_uint(ref));
        }
    }
    atomicAdd(&bad[0], n2);
    atomicAdd(&bad[1], n4);
    atomicAdd(&bad[2], n4b);
}

int main() {
    float lo_f = 5.0e-8f, hi_f = 1.0e18f;
    uint32_t lo, hi;
    memcpy(&lo, &lo_f, 4);
    memcpy(&hi, &hi_f, 4);
    unsigned long long *d, h[3] = {0, 0, 0};
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, lo, hi, d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("values %llu x 2 signs; mismatches with 1.f/a: one step %llu, two steps %llu, two steps (second with r0) %llu\n",
           (unsigned long long)(hi - lo), h[0], h[1], h[2]);
    return 0;
}
