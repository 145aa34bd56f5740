// Which short sequences reproduce the correctly rounded sqrtf(x) bit for bit?  Exhaustive
// over every float in [lo, hi].
//   hipcc -O2 --offload-arch=gfx950 sqrt_steps.hip -o sqrt_steps && ./sqrt_steps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void check(uint32_t lo, uint32_t hi, unsigned long long *bad) {
    unsigned long long n[4] = {0, 0, 0, 0};
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const uint32_t ref = __float_as_uint(__builtin_sqrtf(x));
        const float y = __builtin_amdgcn_rsqf(x);
        // A: Markstein, one coupled refinement and a final correction
        {
            float g = x * y, h = 0.5f * y;
            const float r = __builtin_fmaf(-g, h, 0.5f);
            g = __builtin_fmaf(g, r, g);
            h = __builtin_fmaf(h, r, h);
            const float d = __builtin_fmaf(-g, g, x);
            n[0] += __float_as_uint(__builtin_fmaf(d, h, g)) != ref;
        }
        // B: the final correction only
        {
            const float g = x * y, h = 0.5f * y;
            const float d = __builtin_fmaf(-g, g, x);
            n[1] += __float_as_uint(__builtin_fmaf(d, h, g)) != ref;
        }
        // C: v_sqrt_f32 seed, correction with half the reciprocal square root
        {
            const float g = __builtin_amdgcn_sqrtf(x), h = 0.5f * y;
            const float d = __builtin_fmaf(-g, g, x);
            n[2] += __float_as_uint(__builtin_fmaf(d, h, g)) != ref;
        }
        // D: B applied twice
        {
            float g = x * y;
            const float h = 0.5f * y;
            float d = __builtin_fmaf(-g, g, x);
            g = __builtin_fmaf(d, h, g);
            d = __builtin_fmaf(-g, g, x);
            n[3] += __float_as_uint(__builtin_fmaf(d, h, g)) != ref;
        }
    }
    for (int k = 0; k < 4; k++) atomicAdd(&bad[k], n[k]);
}

int main() {
    float lo_f = 1.0e-30f, hi_f = 1.0e30f;
    uint32_t lo, hi;
    memcpy(&lo, &lo_f, 4);
    memcpy(&hi, &hi_f, 4);
    unsigned long long *d, h[4] = {0, 0, 0, 0};
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, lo, hi, d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%u values; mismatches with sqrtf: A (refine + correct) %llu, B (correct) %llu, C (v_sqrt seed + correct) %llu, D (correct twice) %llu\n",
           hi - lo + 1, h[0], h[1], h[2], h[3]);
    return 0;
}
