// Micro-benchmark: issue cost of the VALU instructions the column kernel is made of, 4 waves per SIMD,
// every CU busy.  Prints cycles per wave-instruction per SIMD (kernel time x clock / instructions per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 4096
#define UNROLL 16

template <int OP>
__global__ void __launch_bounds__(256) bench(double *out, int n)
{
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = -0.5, d = 0.25;
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");
    int ia = threadIdx.x, ib = 3, ic = 5, id = 7, ie = 11;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == 0) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(c) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(b)); }
            if (OP == 1) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(ie)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ib) : "v"(ie)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ic) : "v"(ie)); }
            if (OP == 2) { asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ia) : "v"(ie)); asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ib) : "v"(ie)); asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ic) : "v"(ie)); }
            if (OP == 3) { asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(c) : "v"(b)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(d) : "v"(b)); }
            if (OP == 4) { asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(c), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(d), "v"(b) : "vcc"); }
            if (OP == 5) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia) : "v"(ie)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(ib) : "v"(ie)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(ic) : "v"(ie)); }
            if (OP == 6) { asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ia) : "v"(ie)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ib) : "v"(ie)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ic) : "v"(ie)); }
            if (OP == 7) { asm volatile("v_cmp_gt_f64 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(ia), "+v"(ib) : "v"(ie) : "vcc"); }
            if (OP == 10) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(ia) : "v"(ie) : "s20", "s21"); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(ib) : "v"(ie) : "s20", "s21"); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(ic) : "v"(ie) : "s20", "s21"); }
            if (OP == 11) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(ia) : "v"(id), "v"(ie)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(ib) : "v"(id), "v"(ie)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(ic) : "v"(id), "v"(ie)); }
            if (OP == 12) { asm volatile("v_max_f64 %0, %0, %3\n v_cmp_eq_f64 vcc, %0, %3\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(a), "+v"(ia) : "v"(ie), "v"(b) : "vcc"); }
            if (OP == 13) { asm volatile("v_cmp_gt_f64 vcc, %0, %4\n s_nop 1\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, vcc" : "+v"(a), "+v"(ia), "+v"(ib), "+v"(ic) : "v"(b), "v"(ie) : "vcc"); }
            if (OP == 14) { asm volatile("v_cmp_gt_f64_e64 s[20:21], %0, %4\n s_nop 1\n v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n v_cndmask_b32_e64 %2, %2, %5, s[20:21]\n v_cndmask_b32_e64 %3, %3, %5, s[20:21]" : "+v"(a), "+v"(ia), "+v"(ib), "+v"(ic) : "v"(b), "v"(ie) : "s20", "s21"); }
            if (OP == 15) { asm volatile("v_cmp_gt_f64_e64 s[20:21], %0, %4\n v_cmp_gt_f64_e64 s[22:23], %6, %4\n v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n v_cndmask_b32_e64 %2, %2, %5, s[20:21]\n v_cndmask_b32_e64 %3, %3, %5, s[22:23]\n v_cndmask_b32_e64 %7, %7, %5, s[22:23]" : "+v"(a), "+v"(ia), "+v"(ib), "+v"(ic) : "v"(b), "v"(ie), "v"(c), "v"(id) : "s20", "s21", "s22", "s23"); }
            if (OP == 20) { asm volatile("v_cmp_gt_f64_e32 vcc, %2, %0\n v_max_f64 %0, %0, %2\n v_addc_co_u32_e32 %1, vcc, %1, %1, vcc" : "+v"(a), "+v"(ia) : "v"(b) : "vcc"); }
            if (OP == 21) { asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_2" : "=v"(ia) : "v"(id), "v"(ie)); asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_2" : "=v"(ib) : "v"(id), "v"(ie)); asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_2" : "=v"(ic) : "v"(id), "v"(ie)); }
            if (OP == 22) { asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(ia) :: "vcc"); asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(ib) :: "vcc"); asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(ic) :: "vcc"); }
            if (OP == 23) { asm volatile("v_cmp_gt_f64_e32 vcc, %2, %0\n v_max_f64 %0, %0, %2\n v_addc_co_u32_e32 %1, vcc, %1, %1, vcc" : "+v"(a), "+v"(ia) : "v"(b) : "vcc"); asm volatile("v_cmp_gt_f64_e32 vcc, %2, %0\n v_max_f64 %0, %0, %2\n v_addc_co_u32_e32 %1, vcc, %1, %1, vcc" : "+v"(c), "+v"(ib) : "v"(b) : "vcc"); asm volatile("v_cmp_gt_f64_e32 vcc, %2, %0\n v_max_f64 %0, %0, %2\n v_addc_co_u32_e32 %1, vcc, %1, %1, vcc" : "+v"(d), "+v"(ic) : "v"(b) : "vcc"); }
            if (OP == 24) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b)); }
            if (OP == 8) { asm volatile("v_mov_b32 %0, %1" : "=v"(ia) : "v"(ie)); asm volatile("v_mov_b32 %0, %1" : "=v"(ib) : "v"(ie)); asm volatile("v_mov_b32 %0, %1" : "=v"(ic) : "v"(ie)); }
            if (OP == 9) { asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(ia) : "v"(ie)); asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(ib) : "v"(ie)); asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(ic) : "v"(ie)); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d + ia + ib + ic + id + ie;
}

static int g_waves = 4;
template <int OP> static void run(const char *name, double *d_out, double clock_ghz)
{
    const int grid = 256 * g_waves;      // g_waves blocks of 4 waves per CU = g_waves waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(256), 0, 0, d_out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(256), 0, 0, d_out, ITERS);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)g_waves * ITERS * UNROLL * 3;      // g_waves waves per SIMD, 3 instructions per unroll step
    printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, ms,
           ms * 1e-3 * clock_ghz * 1e9 / insts_per_simd, clock_ghz);
}

int main(int argc, char **argv)
{
    if (argc > 1) g_waves = atoi(argv[1]);
    printf("%d waves per SIMD\n", g_waves);
    double *d_out;
    hipMalloc(&d_out, sizeof(double) * 256 * 4 * 256);
    const double ghz = 2.4;
    run<0>("v_add_f64", d_out, ghz);
    run<3>("v_max_f64", d_out, ghz);
    run<4>("v_cmp_gt_f64", d_out, ghz);
    run<1>("v_cndmask_b32 (vcc)", d_out, ghz);
    run<5>("v_add_u32", d_out, ghz);
    run<2>("v_mov_b32_dpp wave_shr:1", d_out, ghz);
    run<6>("v_mov_b32_dpp row_shr:1", d_out, ghz);
    run<7>("cmp_gt_f64+nop+2 cndmask (/3)", d_out, ghz);
    run<10>("v_cndmask_b32_e64 sgpr mask", d_out, ghz);
    run<11>("v_cndmask_b32 vcc, dst!=src", d_out, ghz);
    run<12>("max_f64+cmp_eq+cndmask (/3)", d_out, ghz);
    run<13>("cmp vcc + 3 cndmask vcc (/3 => x4/3)", d_out, ghz);
    run<14>("cmp s[20:21] + 3 cndmask e64 (x4/3)", d_out, ghz);
    run<15>("2 cmp sgpr + 4 cndmask e64 (x2)", d_out, ghz);
    run<20>("relax: cmp,max,addc one chain", d_out, ghz);
    run<23>("relax x3 independent chains", d_out, ghz);
    run<21>("v_add_u32_sdwa", d_out, ghz);
    run<22>("v_addc_co_u32 vcc in/out", d_out, ghz);
    run<24>("v_add_f64 dependent chain", d_out, ghz);
    run<8>("v_mov_b32", d_out, ghz);
    run<9>("v_lshl_or_b32", d_out, ghz);
    return 0;
}
