// Micro-benchmark: can scalar stores carry the back-pointer bit-planes?  Per iteration a wave does 6 x (v_cmp_gt_f64 into an
// SGPR pair, v_max_f64) and stores the 6 masks with 3 s_store_dwordx4; compared with the same loop without the stores and
// with the v_addc_co_u32 accumulation the kernels use today.  Prints cycles per iteration per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITERS 4096

template <int MODE>
__global__ void __launch_bounds__(256) bench(double *out, unsigned long long *masks, int n)
{
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = -0.5, d = 0.25, e = 0.125, f = 3.0, g = 7.0;
    int bits = 0;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned long long *mine = masks + (size_t)__builtin_amdgcn_readfirstlane(wave) * 8 * 64;   // 64 iterations' worth, reused
    for (int i = 0; i < n; ++i) {
        unsigned long long *p = mine + (i & 63) * 8;
        if (MODE == 0 || MODE == 1) {
            asm volatile(
                "v_cmp_gt_f64_e64 s[60:61], %1, %0\n v_max_f64 %0, %0, %1\n"
                "v_cmp_gt_f64_e64 s[62:63], %3, %2\n v_max_f64 %2, %2, %3\n"
                "v_cmp_gt_f64_e64 s[64:65], %5, %4\n v_max_f64 %4, %4, %5\n"
                "v_cmp_gt_f64_e64 s[66:67], %1, %4\n v_max_f64 %4, %4, %1\n"
                "v_cmp_gt_f64_e64 s[68:69], %3, %0\n v_max_f64 %0, %0, %3\n"
                "v_cmp_gt_f64_e64 s[70:71], %5, %2\n v_max_f64 %2, %2, %5\n"
                : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f)
                :: "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71");
            if (MODE == 1)
                asm volatile("s_store_dwordx4 s[60:63], %0, 0x0\n s_store_dwordx4 s[64:67], %0, 0x10\n s_store_dwordx4 s[68:71], %0, 0x20\n"
                             :: "s"(p) : "memory");
        } else {
            asm volatile(
                "v_cmp_gt_f64_e32 vcc, %1, %0\n v_max_f64 %0, %0, %1\n v_addc_co_u32_e32 %6, vcc, %6, %6, vcc\n"
                "v_cmp_gt_f64_e32 vcc, %3, %2\n v_max_f64 %2, %2, %3\n v_addc_co_u32_e32 %6, vcc, %6, %6, vcc\n"
                "v_cmp_gt_f64_e32 vcc, %5, %4\n v_max_f64 %4, %4, %5\n v_addc_co_u32_e32 %6, vcc, %6, %6, vcc\n"
                "v_cmp_gt_f64_e32 vcc, %1, %4\n v_max_f64 %4, %4, %1\n v_addc_co_u32_e32 %6, vcc, %6, %6, vcc\n"
                "v_cmp_gt_f64_e32 vcc, %3, %0\n v_max_f64 %0, %0, %3\n v_addc_co_u32_e32 %6, vcc, %6, %6, vcc\n"
                "v_cmp_gt_f64_e32 vcc, %5, %2\n v_max_f64 %2, %2, %5\n v_addc_co_u32_e32 %6, vcc, %6, %6, vcc\n"
                : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(bits) :: "vcc");
        }
    }
    if (MODE == 1) asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + bits;
}

static int g_waves = 3;
template <int MODE> static void run(const char *name, double *d_out, unsigned long long *d_masks)
{
    const int grid = 256 * g_waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<MODE>, dim3(grid), dim3(256), 0, 0, d_out, d_masks, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(bench<MODE>, dim3(grid), dim3(256), 0, 0, d_out, d_masks, ITERS);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms -> %.1f cycles per iteration per SIMD (%d waves/SIMD)\n", name, ms,
           ms * 1e-3 * 2.4e9 / ((double)g_waves * ITERS), g_waves);
}

int main(int argc, char **argv)
{
    if (argc > 1) g_waves = atoi(argv[1]);
    double *d_out; unsigned long long *d_masks;
    (void)hipMalloc(&d_out, sizeof(double) * 256 * 4 * 256);
    (void)hipMalloc(&d_masks, sizeof(unsigned long long) * 8 * 64 * 256 * 4 * 4);
    run<0>("6 x (cmp -> sgpr, max)", d_out, d_masks);
    run<1>("6 x (cmp -> sgpr, max) + 3 s_store_dwordx4", d_out, d_masks);
    run<2>("6 x (cmp -> vcc, max, addc)", d_out, d_masks);
    // check that the stored masks are what a vector load sees
    unsigned long long h[8];
    (void)hipMemcpy(h, d_masks, sizeof h, hipMemcpyDeviceToHost);
    printf("first masks: %016llx %016llx %016llx\n", h[0], h[1], h[5]);
    return 0;
}
