// Issue rate of the vector instructions the shading kernel is made of, on one SIMD, for
// 1..8 resident waves: cycles per wave-instruction = resident waves x elapsed shader cycles /
// instructions issued by one wave.  Eight independent dependency chains per wave.
//   hipcc -O2 --offload-arch=gfx950 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 2000

#define BODY8(INS)                                                                       \
    INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

template <int OP>
__global__ __launch_bounds__(512) void rate_kernel(unsigned long long *out, float seed) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    float a[8];
    v2f p[8];
    uint32_t u[8];
    for (int k = 0; k < 8; k++) {
        a[k] = seed + (float)(threadIdx.x + k) * 1e-3f;
        p[k] = (v2f){a[k], a[k] * 0.5f};
        u[k] = threadIdx.x * 2654435761u + k;
    }
    const float c = seed * 0.999f + 1e-3f;
    asm volatile("s_mov_b64 vcc, 0x55\n s_mov_b64 s[20:21], 0x33" : : : "vcc", "s20", "s21");
    const v2f pc = (v2f){c, c};
    const uint32_t uc = 0xD256D193u;
    uint32_t uc2 = uc + threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(c));
                if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
                if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
                if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[k]) : "v"(pc));
                if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(uc));
                if (OP == 7) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[k]) : "v"(uc));
                if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
                if (OP == 9) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
                if (OP == 10) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[k]) : "v"(uc));
                if (OP == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(uc));
                if (OP == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(c) : "vcc");
                if (OP == 13) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[k]), "v"(c) : "s20", "s21");
                if (OP == 14) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long *)&p[k]) : "v"(u[k]), "v"(uc) : "vcc");
                if (OP == 15) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[k]) : "v"(uc));
                if (OP == 16) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[k]) : "s"(pc));
                if (OP == 17) asm volatile("v_mov_b32 %0, %1" : "+v"(u[k]) : "v"(uc));
                if (OP == 18) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(u[k]) : "s20");
                if (OP == 19) asm volatile("s_and_b64 s[20:21], s[20:21], exec" : : : "s20", "s21", "scc");
                if (OP == 20) asm volatile("s_nop 0");
                if (OP == 21) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
                if (OP == 22) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (OP == 24) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u[k]) : "v"(uc));
                if (OP == 25) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[k]) : "v"(uc), "v"(threadIdx.x));
                if (OP == 26) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_add_f32 %1, %1, %3" : "+v"(u[k]), "+v"(a[k]) : "v"(uc), "v"(c));
                if (OP == 27) asm volatile("v_cmp_lt_f32 vcc, %1, %3\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(u[k]) : "v"(a[k]), "v"(uc), "v"(c) : "vcc");
                if (OP == 28) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (OP == 29) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(u[k]) : "v"(uc));
                if (OP == 30) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[k]) : "v"(uc));
                if (OP == 31) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (OP == 32) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (OP == 33) asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
                if (OP == 34) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c) : "vcc");
                if (OP == 35) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(a[k]), "v"(uc) : "vcc");
                if (OP == 36) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_add_f32 %3, %3, %4" : "+v"(u[k]), "+v"(u[(k + 1) & 7]), "+v"(uc2), "+v"(a[k]) : "v"(c));
                if (OP == 37) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %5, %5, %2, vcc\n v_add_f32 %3, %3, %4" : "+v"(u[k]), "+v"(u[(k + 1) & 7]), "+v"(uc2), "+v"(a[k]) : "v"(c), "v"(u[(k + 2) & 7]));
                if (OP == 38) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_add_f32 %3, %3, %4\n v_cndmask_b32 %1, %1, %2, vcc\n v_add_f32 %3, %3, %4" : "+v"(u[k]), "+v"(u[(k + 1) & 7]), "+v"(uc2), "+v"(a[k]) : "v"(c));
                if (OP == 39) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n s_nop 0\n v_cndmask_b32 %1, %1, %2, vcc\n s_nop 0" : "+v"(u[k]), "+v"(u[(k + 1) & 7]), "+v"(uc2));
                if (OP == 40) asm volatile("v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc" : "+v"(u[k]), "+v"(u[(k + 1) & 7]), "+v"(uc2));
                if (OP == 41) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]" : "+v"(u[k]), "+v"(u[(k + 1) & 7]), "+v"(uc2));
                if (OP == 42) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[k]) : "v"(uc) : "vcc");
                if (OP == 43) asm volatile("v_cmp_lt_f32 vcc, %1, %3\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(u[k]) : "v"(a[k]), "v"(uc), "v"(c) : "vcc");
                if (OP == 44) asm volatile("v_cmp_lt_f32 vcc, %1, %3\n v_add_f32 %1, %1, %3\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(u[k]), "+v"(a[k]) : "v"(uc), "v"(c) : "vcc");
                if (OP == 45) asm volatile("v_cmp_lt_f32 s[20:21], %1, %3\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(u[k]) : "v"(a[k]), "v"(uc), "v"(c) : "s20", "s21");
                if (OP == 46) asm volatile("v_cmp_lt_f32 vcc, %1, %3\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc\n v_add_f32 %1, %1, %3" : "+v"(u[k]), "+v"(a[k]) : "v"(uc), "v"(c) : "vcc");
                if (OP == 47) asm volatile("s_nop 1\n v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (OP == 23) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(uc));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    uint32_t x = 0;
    for (int k = 0; k < 8; k++) {
        s += a[k] + p[k].x + p[k].y;
        x ^= u[k];
    }
    if (s == 12345.678f && x == 77u) out[1023] = 1; // keep the results alive
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char *NAMES[] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32",
                              "v_mul_lo_u32", "v_mul_hi_u32", "v_rcp_f32", "v_sqrt_f32", "v_xor_b32", "v_cndmask_b32",
                              "v_cmp_lt_f32 vcc", "v_cmp_lt_f32 sgpr", "v_mad_u64_u32", "v_mul_u32_u24",
                              "v_pk_mul_f32 sgpr", "v_mov_b32", "v_readfirstlane", "s_and_b64", "s_nop 0", "v_exp_f32",
                              "v_max_f32", "v_add_u32", "v_cndmask sgpr", "v_cndmask dst!=src", "cndmask+add (2)", "cmp+cndmask (2)", "v_min_f32", "v_bfi_b32", "v_and_b32", "v_sub_f32", "mul+add dep (2)", "pk_mul+pk_add dep (2)", "cmp+add (2)", "v_cmp_class_f32", "2 cndmask + add (3)", "3 cndmask + add (4)", "cndmask add cndmask add (4)", "cndmask nop cndmask nop (4)", "2 cndmask_e64 vcc (2)", "cndmask vcc, cndmask sgpr (2)", "v_addc_co_u32", "cmp nop1 cndmask (3)", "cmp add cndmask (3)", "cmp_sgpr nop1 cndmask_e64 (3)", "cmp nop1 cndmask add (4)", "nop1 add (2)"};

template <int OP>
void run(unsigned long long *d_out, int n_cu) {
    printf("%-30s", NAMES[OP]);
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        // one block per CU: waves_per_simd * 4 waves
        const int threads = waves_per_simd * 4 * 64;
        if (threads > 512) { // two blocks per CU
            hipLaunchKernelGGL(rate_kernel<OP>, dim3(n_cu * (threads / 512)), dim3(512), 0, 0, d_out, 1.0f);
        } else {
            hipLaunchKernelGGL(rate_kernel<OP>, dim3(n_cu), dim3(threads), 0, 0, d_out, 1.0f);
        }
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        if (threads > 512)
            hipLaunchKernelGGL(rate_kernel<OP>, dim3(n_cu * (threads / 512)), dim3(512), 0, 0, d_out, 1.0f);
        else
            hipLaunchKernelGGL(rate_kernel<OP>, dim3(n_cu), dim3(threads), 0, 0, d_out, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD per nanosecond -> SIMD cycles per wave-instruction at 2.4 GHz
        const double per_simd = (double)ITER * 32 * waves_per_simd;
        const double wall_cyc = (double)ms * 1e-3 * 2.4e9 / per_simd;
        std::vector<unsigned long long> h(n_cu * 8 * 2);
        hipMemcpy(h.data(), d_out, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double cyc = 0;
        int n = threads > 512 ? 8 : threads / 64;
        for (int k = 0; k < n; k++) cyc += (double)h[k];
        cyc /= n;
        printf("  %dw: %6.2f", waves_per_simd, cyc * waves_per_simd / (double)(ITER * 32) / waves_per_simd * 1.0);
        printf("/%5.2f/%5.2f", cyc / (double)(ITER * 32) / waves_per_simd, wall_cyc);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("%s, %d CUs, clock %d kHz; columns: waves per SIMD: cycles per instruction seen by a wave / per SIMD issue slot / the same from wall time at 2.4 GHz\n",
           prop.name, prop.multiProcessorCount, prop.clockRate);
    unsigned long long *d_out;
    hipMalloc(&d_out, 1 << 20);
    const int n_cu = prop.multiProcessorCount;
    run<0>(d_out, n_cu); run<1>(d_out, n_cu); run<2>(d_out, n_cu); run<3>(d_out, n_cu); run<4>(d_out, n_cu);
    run<5>(d_out, n_cu); run<16>(d_out, n_cu); run<6>(d_out, n_cu); run<7>(d_out, n_cu); run<14>(d_out, n_cu);
    run<15>(d_out, n_cu); run<8>(d_out, n_cu); run<9>(d_out, n_cu); run<21>(d_out, n_cu); run<10>(d_out, n_cu);
    run<23>(d_out, n_cu); run<22>(d_out, n_cu); run<11>(d_out, n_cu); run<12>(d_out, n_cu); run<13>(d_out, n_cu);
    run<24>(d_out, n_cu); run<25>(d_out, n_cu); run<26>(d_out, n_cu); run<27>(d_out, n_cu); run<28>(d_out, n_cu);
    run<29>(d_out, n_cu); run<30>(d_out, n_cu); run<31>(d_out, n_cu); run<32>(d_out, n_cu); run<33>(d_out, n_cu);
    run<34>(d_out, n_cu); run<35>(d_out, n_cu);
    run<36>(d_out, n_cu); run<37>(d_out, n_cu); run<38>(d_out, n_cu); run<39>(d_out, n_cu); run<40>(d_out, n_cu); run<41>(d_out, n_cu); run<42>(d_out, n_cu);
    run<43>(d_out, n_cu); run<44>(d_out, n_cu); run<45>(d_out, n_cu); run<46>(d_out, n_cu); run<47>(d_out, n_cu);
    run<17>(d_out, n_cu); run<18>(d_out, n_cu); run<19>(d_out, n_cu); run<20>(d_out, n_cu);
    hipFree(d_out);
    return 0;
}
