// forward_generic.h -- sum-product twin of the generic Viterbi kernel (Model.log_probability).
//
// Restates HiddenMarkovModel._forward / _vl_log_probability
// (/root/reference/pomegranate/hmm.pyx:1371-1484, 1300-1313) with pair_lse (utils.pyx:72-90).
// Same wave/LDS organisation as viterbi_generic.h.  Floating point: the log-sum-exp folds run in the
// chunked A/B order and use the device libm, so results agree with the reference to rounding
// (tests allow 1e-9 relative; the north star asks for 1e-4).
#pragma once
#include "viterbi_generic.h"

#ifndef ADVNTR_LSE2_DEFINED
#define ADVNTR_LSE2_DEFINED
__device__ __forceinline__ double lse2(double x, double y)
{
    // branch-free form of the same expression: hi + log(exp(lo - hi) + 1) with hi/lo picked exactly as the
    // reference's (x > y) test does; the -inf / +inf cases are selects, so a wavefront never diverges here
    const bool xg = x > y;
    const double hi = xg ? x : y, lo = xg ? y : x;
    double r = hi + log(exp(lo - hi) + 1.0);
    r = (lo == -INFINITY) ? hi : r;
    r = (x == INFINITY || y == INFINITY) ? INFINITY : r;
    return r;
}
#endif

__device__ __forceinline__ void forward_silent_pass(const DevModel &M, double *cur, int t, int lane)
{
    const int P = M.P, m = M.m;
    for (int c = 0; c < M.n_chunks; ++c) {
        const int base = P + c * ADV_WAVE;
        const int l = base + lane;
        const bool active = l < m;
        double accE = -INFINITY, accS = -INFINITY;
        int qb = 0, qe = 0, nsrc = -1;
        double nlp = 0.0;
        const bool fixed_start = (t == 0 && l == M.start);
        if (active) {
            const int ls = l - P;
            qb = M.s_mid[ls];
            qe = M.s_ptr[ls + 1];
            if (fixed_start) {
                qb = qe;
            } else {
                for (int q = M.s_ptr[ls]; q < qb; ++q) {
                    const int s = M.s_src[q];
                    const double v = cur[s] + M.s_logp[q];
                    if (s < P) accE = lse2(accE, v); else accS = lse2(accS, v);
                }
            }
            if (qb < qe) { nsrc = M.s_src[qb]; nlp = M.s_logp[qb]; }
        }
        double val = fixed_start ? 0.0 : lse2(accE, accS);
        const int jn = __builtin_amdgcn_readfirstlane(min(ADV_WAVE, m - base) - 1);
        if (__ballot(nsrc >= 0)) {
            for (int j = 0; j < jn; ++j) {
                const double vj = readlane_f64(val, j);
                if (nsrc == base + j) {
                    accS = lse2(accS, vj + nlp);
                    val = lse2(accE, accS);
                    ++qb;
                    if (qb < qe) { nsrc = M.s_src[qb]; nlp = M.s_logp[qb]; } else nsrc = -1;
                }
            }
        }
        if (active) cur[l] = val;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(ADV_WAVE) forward_generic_kernel(BatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds_rows[];
    const int lane = threadIdx.x;
    for (;;) {
        const int it = next_read(a, lane);
        if (it >= a.n_reads) break;
        const int r = __builtin_amdgcn_readfirstlane(a.order ? a.order[it] : it);
        const DevModel M = a.models[__builtin_amdgcn_readfirstlane(a.read_model[r])];
        const uint8_t *seq = a.bases + a.read_off[r];
        const int n = __builtin_amdgcn_readfirstlane((int)(a.read_off[r + 1] - a.read_off[r]));
        const int P = M.P;
        double *prev = lds_rows, *cur = lds_rows + a.m_max;
        for (int l = lane; l < P; l += ADV_WAVE) cur[l] = -INFINITY;
        __syncthreads();
        forward_silent_pass(M, cur, 0, lane);
        for (int t = 1; t <= n; ++t) {
            double *tmp = prev; prev = cur; cur = tmp;
            const int x = seq[t - 1];
            for (int l = lane; l < P; l += ADV_WAVE) {
                double acc = -INFINITY;
                for (int k = M.e_ptr[l]; k < M.e_ptr[l + 1]; ++k) acc = lse2(acc, prev[M.e_src[k]] + M.e_logp[k]);
                cur[l] = acc + M.emis[4 * l + x];
            }
            __syncthreads();
            forward_silent_pass(M, cur, t, lane);
        }
        double lp;
        if (M.finite) {
            lp = cur[M.end];
        } else {
            // hmm.pyx:1308-1310: fold over the emitting states in index order; lane-strided partial folds
            double acc = -INFINITY;
            for (int l = lane; l < P; l += ADV_WAVE) acc = lse2(acc, cur[l]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc = lse2(acc, __shfl_xor(acc, o, 64));
            lp = acc;
        }
        if (lane == 0) a.out_logp[r] = lp;
        __syncthreads();
    }
    (void)0;
}
