// Microbenchmark: issue rate of the vector instructions the decode kernel is made of, at 8 waves per SIMD (gfx950).
// Each kernel runs ITER iterations of 16 independent instances of one instruction per lane; cycles per wave-instruction per SIMD =
// elapsed_cycles * SIMDs_busy / (waves_per_simd * ITER * 16).   build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 4096
#define OPS 16

#define KERNEL(NAME, ASM)                                                                            \
    __global__ void __launch_bounds__(256) NAME(unsigned *out, unsigned seed)                        \
    {                                                                                                \
        unsigned a[OPS], b = seed + threadIdx.x, c = seed * 3u + 1u;                                 \
        for (int i = 0; i < OPS; ++i) a[i] = threadIdx.x * 7u + i;                                   \
        for (int it = 0; it < ITER; ++it) {                                                          \
            _Pragma("unroll") for (int i = 0; i < OPS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                            \
        unsigned r = 0;                                                                              \
        for (int i = 0; i < OPS; ++i) r ^= a[i];                                                     \
        if (r == 0x12345678u) out[0] = r;                                                            \
    }

KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
KERNEL(k_pk_lshr, "v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]")
KERNEL(k_pk_max, "v_pk_max_u16 %0, %0, %1")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x30")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_pk_fma_f32_nop, "v_mul_f32 %0, %0, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_mov, "v_mov_b32 %0, %1")

#define KERNEL64(NAME, ASM)                                                                          \
    __global__ void __launch_bounds__(256) NAME(unsigned *out, unsigned seed)                        \
    {                                                                                                \
        double a[OPS], b = 1.0 + 1e-9 * (seed + threadIdx.x), c = 1e-12 * seed;                      \
        for (int i = 0; i < OPS; ++i) a[i] = 1.0 + 1e-6 * (threadIdx.x + i);                         \
        for (int it = 0; it < ITER; ++it) {                                                          \
            _Pragma("unroll") for (int i = 0; i < OPS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                            \
        double r = 0;                                                                                \
        for (int i = 0; i < OPS; ++i) r += a[i];                                                     \
        if (r == 0.12345) out[0] = 1;                                                                \
    }
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL64(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL64(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL64(k_cvt_like_min_f64, "v_min_f64 %0, %0, %1")

typedef void (*kern_t)(unsigned *, unsigned);

int main()
{
    unsigned *d;
    hipMalloc(&d, 64);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double clk_hz = p.clockRate * 1e3;
    struct K { const char *name; kern_t f; };
    std::vector<K> ks = {{"v_and_b32", k_and}, {"v_add_u32", k_add_u32}, {"v_pk_add_u16", k_pk_add_u16}, {"v_pk_sub_i16", k_pk_sub_i16},
                         {"v_pk_lshrrev_b16", k_pk_lshr}, {"v_pk_max_u16", k_pk_max}, {"v_bitop3_b32", k_bitop3}, {"v_and_or_b32", k_and_or},
                         {"v_or3_b32", k_or3}, {"v_perm_b32", k_perm}, {"v_lshl_or_b32", k_lshl_or}, {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_pk_fma_f32_nop},
                         {"v_cndmask_b32", k_cndmask}, {"v_mov_b32", k_mov}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64},
                         {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64}, {"v_min_f64", k_cvt_like_min_f64}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("device %s, %d CUs, clockRate %.0f MHz; 8 waves per SIMD (8 blocks of 256 threads per CU)\n", p.gcnArchName, cus, clk_hz / 1e6);
    for (auto &k : ks) {
        const int blocks = cus * 8;                       // 8 x 4 waves per CU = 8 waves per SIMD
        hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 1u);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 2u + rep);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double wave_instr_per_simd = 8.0 * ITER * OPS;          // 8 waves per SIMD
        printf("%-18s %8.3f ms  -> %.2f cycles per wave64 instruction per SIMD at %.0f MHz (%.2f at 2400 MHz)\n", k.name, best,
               best * 1e-3 * clk_hz / wave_instr_per_simd, clk_hz / 1e6, best * 1e-3 * 2.4e9 / wave_instr_per_simd);
    }
    return 0;
}
