// Micro-benchmark: the shader clock the chip actually sustains under a given vector-instruction load.
// s_memtime ticks once per shader cycle (MI355X_MICROARCH.md, per-instruction table), s_memrealtime at a constant rate;
// every wavefront records both around its loop, the host adds the wall time of the launch from HIP events:
//   clock = d(memtime) / wall time of the wavefront's loop (its d(memrealtime) at the constant rate the first, idle run calibrates),
//   cycles per wave-instruction and SIMD in TRUE shader cycles = d(memtime) / (instructions per wave x waves per SIMD).
// Usage: clock_probe [waves per SIMD = 3] [iterations = 40000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define BODY(txt) asm volatile(REP4(txt) ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v40","v41","vcc","s20","s21","s22","s23")

template <int OP>
__global__ void __launch_bounds__(256) probe(unsigned long long *out, int n)
{
    asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n"
                 "v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n"
                 "v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3ff00000\n"
                 ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v40","v41");
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int i = 0; i < n; ++i) {
        if (OP == 0) BODY("v_add_f64 v[10:11], v[10:11], v[40:41]\n v_add_f64 v[12:13], v[12:13], v[40:41]\n v_add_f64 v[14:15], v[14:15], v[40:41]\n v_add_f64 v[16:17], v[16:17], v[40:41]\n v_add_f64 v[18:19], v[18:19], v[40:41]\n v_add_f64 v[20:21], v[20:21], v[40:41]\n");
        if (OP == 1) BODY("v_add_f32 v10, v10, v40\n v_add_f32 v12, v12, v40\n v_add_f32 v14, v14, v40\n v_add_f32 v16, v16, v40\n v_add_f32 v18, v18, v40\n v_add_f32 v20, v20, v40\n");
        if (OP == 2) BODY("s_sleep 1\n s_sleep 1\n s_sleep 1\n s_sleep 1\n s_sleep 1\n s_sleep 1\n");
        if (OP == 3) BODY("v_cmp_gt_f64 s[20:21], v[40:41], v[10:11]\n v_max_f64 v[10:11], v[10:11], v[40:41]\n v_add_f64 v[12:13], v[14:15], v[16:17]\n v_cmp_gt_f64 s[22:23], v[40:41], v[18:19]\n v_max_f64 v[18:19], v[18:19], v[40:41]\n v_add_f64 v[20:21], v[14:15], v[16:17]\n");
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    double r;
    asm volatile("v_add_f64 %0, v[10:11], v[12:13]" : "=v"(r));
    if (threadIdx.x % 64 == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
        if (r == 12345.678) out[0] = 0;
    }
}

static int g_waves = 3, g_iters = 40000;
static double g_real_hz = 1e8;

template <int OP> static void run(const char *name, unsigned long long *d_out, bool calibrate)
{
    const int grid = 256 * g_waves, n_waves = grid * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // dynamic LDS sized so that exactly g_waves workgroups fit a compute unit: one wave of each on every SIMD, evenly
    const size_t lds = (160 * 1024 / g_waves) & ~(size_t)1023;
    hipFuncSetAttribute((const void *)probe<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), lds, 0, d_out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), lds, 0, d_out, g_iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * n_waves);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double st = 0, sr = 0;
    for (int w = 0; w < n_waves; ++w) { st += (double)h[2 * w]; sr += (double)h[2 * w + 1]; }
    st /= n_waves; sr /= n_waves;
    if (calibrate) g_real_hz = sr / (ms * 1e-3);          // every wave runs for (nearly) the whole launch
    const double insts = (double)g_iters * 24;
    printf("%-34s launch %8.3f ms  memtime %.4g ticks  realtime %.4g ticks (%.1f MHz)  => shader clock %.0f MHz (vs events: %.0f)  "
           "%.2f true cycles per inst and SIMD (%.2f at 2.4 GHz nominal)\n", name, ms, st, sr, g_real_hz / 1e6,
           st / (sr / g_real_hz) / 1e6, st / (ms * 1e-3) / 1e6, st / (insts * g_waves), ms * 1e-3 * 2.4e9 / (insts * g_waves));
}

int main(int argc, char **argv)
{
    if (argc > 1) g_waves = atoi(argv[1]);
    if (argc > 2) g_iters = atoi(argv[2]);
    printf("%d waves per SIMD, %d iterations of 24 instructions\n", g_waves, g_iters);
    unsigned long long *d_out;
    hipMalloc(&d_out, 8 * 2 * 256 * 8 * 4);
    run<2>("s_sleep (idle, calibrates realtime)", d_out, true);
    run<1>("v_add_f32", d_out, false);
    run<0>("v_add_f64", d_out, false);
    run<3>("f64 cmp/max/add mix", d_out, false);
    run<0>("v_add_f64 (again)", d_out, false);
    return 0;
}
