// Micro-benchmark: does the issue cost of v_add_f64 / v_max_f64 / v_cmp_gt_f64 depend on which VGPR banks the operands sit in,
// on the number of independent chains, or on the number of waves per SIMD?  Explicit registers throughout.
// Prints cycles per wave-instruction per SIMD at the clock given as argv[2] (default 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITERS 2048

// 24 instructions per loop body
#define REP4(x) x x x x
#define BODY(txt) asm volatile(REP4(txt) ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","vcc","s20","s21","s22","s23")

template <int OP>
__global__ void __launch_bounds__(256) bench(double *out, int n)
{
    asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n"
                 "v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n"
                 "v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0x3ff00000\n"
                 "v_mov_b32 v28, 0\n v_mov_b32 v29, 0x3ff00000\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0x3ff00000\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n"
                 "v_mov_b32 v34, 0\n v_mov_b32 v35, 0x3ff00000\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0x3ff00000\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0x3ff00000\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3ff00000\n"
                 ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41");
    for (int i = 0; i < n; ++i) {
        // 6 independent chains, dst = src0, src1 = v[40:41] (bank 0/1); dst banks alternate 2,0,2,0...
        if (OP == 0) BODY("v_add_f64 v[10:11], v[10:11], v[40:41]\n v_add_f64 v[12:13], v[12:13], v[40:41]\n v_add_f64 v[14:15], v[14:15], v[40:41]\n v_add_f64 v[16:17], v[16:17], v[40:41]\n v_add_f64 v[18:19], v[18:19], v[40:41]\n v_add_f64 v[20:21], v[20:21], v[40:41]\n");
        // all three operands distinct registers, banks: dst 2, src0 0, src1 2
        if (OP == 1) BODY("v_add_f64 v[10:11], v[12:13], v[14:15]\n v_add_f64 v[16:17], v[20:21], v[22:23]\n v_add_f64 v[18:19], v[24:25], v[26:27]\n v_add_f64 v[28:29], v[32:33], v[34:35]\n v_add_f64 v[30:31], v[36:37], v[38:39]\n v_add_f64 v[40:41], v[12:13], v[22:23]\n");
        // src0 and src1 in the same bank pair (both start at bank 0)
        if (OP == 2) BODY("v_add_f64 v[10:11], v[12:13], v[16:17]\n v_add_f64 v[14:15], v[20:21], v[24:25]\n v_add_f64 v[18:19], v[28:29], v[32:33]\n v_add_f64 v[22:23], v[36:37], v[40:41]\n v_add_f64 v[26:27], v[12:13], v[24:25]\n v_add_f64 v[30:31], v[20:21], v[32:33]\n");
        // src0 and src1 in different bank pairs (0 and 2)
        if (OP == 3) BODY("v_add_f64 v[10:11], v[12:13], v[18:19]\n v_add_f64 v[14:15], v[20:21], v[26:27]\n v_add_f64 v[22:23], v[28:29], v[34:35]\n v_add_f64 v[30:31], v[36:37], v[38:39]\n v_add_f64 v[10:11], v[16:17], v[18:19]\n v_add_f64 v[14:15], v[24:25], v[26:27]\n");
        // one source an SGPR pair
        if (OP == 5) BODY("v_add_f64 v[10:11], v[10:11], s[20:21]\n v_add_f64 v[12:13], v[12:13], s[20:21]\n v_add_f64 v[14:15], v[14:15], s[20:21]\n v_add_f64 v[16:17], v[16:17], s[20:21]\n v_add_f64 v[18:19], v[18:19], s[20:21]\n v_add_f64 v[20:21], v[20:21], s[20:21]\n");
        // same source twice
        if (OP == 6) BODY("v_add_f64 v[10:11], v[40:41], v[40:41]\n v_add_f64 v[12:13], v[40:41], v[40:41]\n v_add_f64 v[14:15], v[40:41], v[40:41]\n v_add_f64 v[16:17], v[40:41], v[40:41]\n v_add_f64 v[18:19], v[40:41], v[40:41]\n v_add_f64 v[20:21], v[40:41], v[40:41]\n");
        // max, cmp (vcc), cmp (sgpr), fma, mul, add_f32 for reference
        if (OP == 7) BODY("v_max_f64 v[10:11], v[10:11], v[40:41]\n v_max_f64 v[12:13], v[12:13], v[40:41]\n v_max_f64 v[14:15], v[14:15], v[40:41]\n v_max_f64 v[16:17], v[16:17], v[40:41]\n v_max_f64 v[18:19], v[18:19], v[40:41]\n v_max_f64 v[20:21], v[20:21], v[40:41]\n");
        if (OP == 8) BODY("v_cmp_gt_f64 vcc, v[10:11], v[40:41]\n v_cmp_gt_f64 vcc, v[12:13], v[40:41]\n v_cmp_gt_f64 vcc, v[14:15], v[40:41]\n v_cmp_gt_f64 vcc, v[16:17], v[40:41]\n v_cmp_gt_f64 vcc, v[18:19], v[40:41]\n v_cmp_gt_f64 vcc, v[20:21], v[40:41]\n");
        if (OP == 9) BODY("v_cmp_gt_f64 s[20:21], v[10:11], v[40:41]\n v_cmp_gt_f64 s[22:23], v[12:13], v[40:41]\n v_cmp_gt_f64 s[20:21], v[14:15], v[40:41]\n v_cmp_gt_f64 s[22:23], v[16:17], v[40:41]\n v_cmp_gt_f64 s[20:21], v[18:19], v[40:41]\n v_cmp_gt_f64 s[22:23], v[20:21], v[40:41]\n");
        if (OP == 10) BODY("v_fma_f64 v[10:11], v[10:11], v[40:41], v[38:39]\n v_fma_f64 v[12:13], v[12:13], v[40:41], v[38:39]\n v_fma_f64 v[14:15], v[14:15], v[40:41], v[38:39]\n v_fma_f64 v[16:17], v[16:17], v[40:41], v[38:39]\n v_fma_f64 v[18:19], v[18:19], v[40:41], v[38:39]\n v_fma_f64 v[20:21], v[20:21], v[40:41], v[38:39]\n");
        if (OP == 11) BODY("v_add_f32 v10, v10, v40\n v_add_f32 v12, v12, v40\n v_add_f32 v14, v14, v40\n v_add_f32 v16, v16, v40\n v_add_f32 v18, v18, v40\n v_add_f32 v20, v20, v40\n");
        if (OP == 12) BODY("v_pk_add_f32 v[10:11], v[10:11], v[40:41]\n v_pk_add_f32 v[12:13], v[12:13], v[40:41]\n v_pk_add_f32 v[14:15], v[14:15], v[40:41]\n v_pk_add_f32 v[16:17], v[16:17], v[40:41]\n v_pk_add_f32 v[18:19], v[18:19], v[40:41]\n v_pk_add_f32 v[20:21], v[20:21], v[40:41]\n");
        // relaxation as shipped before (vcc + addc) and as masks
        if (OP == 13) BODY("v_cmp_gt_f64 vcc, v[40:41], v[10:11]\n v_max_f64 v[10:11], v[10:11], v[40:41]\n v_addc_co_u32 v30, vcc, v30, v30, vcc\n v_cmp_gt_f64 vcc, v[38:39], v[12:13]\n v_max_f64 v[12:13], v[12:13], v[38:39]\n v_addc_co_u32 v30, vcc, v30, v30, vcc\n");
        if (OP == 14) BODY("v_cmp_gt_f64 s[20:21], v[40:41], v[10:11]\n v_max_f64 v[10:11], v[10:11], v[40:41]\n v_add_f64 v[30:31], v[32:33], v[34:35]\n v_cmp_gt_f64 s[22:23], v[38:39], v[12:13]\n v_max_f64 v[12:13], v[12:13], v[38:39]\n v_add_f64 v[28:29], v[32:33], v[36:37]\n");
        // 32-bit integer add for reference
        if (OP == 15) BODY("v_add_u32 v10, v10, v40\n v_add_u32 v12, v12, v40\n v_add_u32 v14, v14, v40\n v_add_u32 v16, v16, v40\n v_add_u32 v18, v18, v40\n v_add_u32 v20, v20, v40\n");
        // f64 add with the constant operand inline (1.0)
        if (OP == 16) BODY("v_add_f64 v[10:11], v[10:11], 1.0\n v_add_f64 v[12:13], v[12:13], 1.0\n v_add_f64 v[14:15], v[14:15], 1.0\n v_add_f64 v[16:17], v[16:17], 1.0\n v_add_f64 v[18:19], v[18:19], 1.0\n v_add_f64 v[20:21], v[20:21], 1.0\n");
    }
    double r;
    asm volatile("v_add_f64 %0, v[10:11], v[12:13]" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static int g_waves = 3;
static double g_ghz = 2.4;
template <int OP> static void run(const char *name, double *d_out)
{
    const int grid = 256 * g_waves;
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
    const double insts_per_simd = (double)g_waves * ITERS * 24;
    printf("%-44s %8.3f ms  %.2f cycles/inst\n", name, ms, ms * 1e-3 * g_ghz * 1e9 / insts_per_simd);
}

int main(int argc, char **argv)
{
    if (argc > 1) g_waves = atoi(argv[1]);
    if (argc > 2) g_ghz = atof(argv[2]);
    printf("%d waves per SIMD, %.2f GHz assumed\n", g_waves, g_ghz);
    double *d_out;
    hipMalloc(&d_out, sizeof(double) * 256 * 16 * 256);
    run<11>("v_add_f32", d_out);
    run<15>("v_add_u32", d_out);
    run<12>("v_pk_add_f32", d_out);
    run<0>("v_add_f64 d=s0, s1 shared", d_out);
    run<1>("v_add_f64 3 distinct regs (banks 2,0,2)", d_out);
    run<2>("v_add_f64 srcs same bank pair", d_out);
    run<3>("v_add_f64 srcs different bank pairs", d_out);
    run<5>("v_add_f64 src1 sgpr", d_out);
    run<6>("v_add_f64 src0 == src1", d_out);
    run<16>("v_add_f64 inline constant", d_out);
    run<7>("v_max_f64", d_out);
    run<8>("v_cmp_gt_f64 vcc", d_out);
    run<9>("v_cmp_gt_f64 sgpr", d_out);
    run<10>("v_fma_f64", d_out);
    run<13>("relax vcc+addc (3 inst)", d_out);
    run<14>("relax sgpr + indep add (3 inst)", d_out);
    return 0;
}
