// forward_rows.h -- sum-product (Model.log_probability) in the row-blocked layout of viterbi_rows.h, for the short reads of
// a large batch (G reads per wavefront, R rows per lane).  The recurrence is forward_columns.h's: LINEAR domain,
// probabilities times 16^row (the class and emission tables are exponentiated in LDS when a model is staged, emissions
// times the per-row scale 16; a sweep of at most 155 rows neither overflows nor underflows), the last row is captured
// as scaled probabilities, taken back to the log domain after the sweep and handed to col_tail_forward.  No back-pointers,
// no traceback.  Results agree with the reference's pair_lse folding to rounding (tests: 1e-9 relative; measured 3e-14).
#pragma once
#include "viterbi_rows.h"
#include "forward_columns.h"

template <int R, int G>
__device__ __forceinline__ void rows_sweep_fwd(const LdsTables &L, const int NC, const int s_end,
                                               const uint8_t *__restrict__ seq, const int n, const int lp, const int lane,
                                               double *__restrict__ rown, const unsigned cap_base)
{
    constexpr int W = 64 / G;
    double I[R], M[R], B[R], er[R];
    unsigned xp[(R + 3) / 4];
#pragma unroll
    for (int q = 0; q < (R + 3) / 4; ++q) xp[q] = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        I[k] = M[k] = B[k] = er[k] = 0.0;
        const int t = R * lp + k + 1;
        xp[k / 4] |= (unsigned)((t <= n) ? 8 * (int)seq[t - 1] : 32) << (8 * (k % 4));     // rows past the read: emission 0
    }
    auto xof = [&](const int k) { return (xp[k / 4] >> (8 * (k % 4))) & 0xffu; };
    const int kcap = (n >= 1 && (n - 1) / R == lp) ? (n - 1) - lp * R : -1;
    const bool first_lane = lp == 0;
    const bool fix = G == 2 && lane == 32;
    double nI = 0.0, nM = 0.0, nB = 0.0;
    unsigned pa = L.pinfo + (unsigned)(64 - lp) * 16u;
    uint2 meta = lds_uint2(pa + 8u);
    const unsigned cap_lane = cap_base + (unsigned)(W - lp) * 3u;
    int sstep = 0;
    auto step = [&]() {
        pa += 16u;
        const uint2 meta_next = lds_uint2(pa + 8u);
        LdsClass *T = (LdsClass *)(size_t)(meta.x & 0xffffu);
        const double iI = T->iI, iM = T->iM, iD = T->iD, mI = T->mI, mM = T->mM, mD = T->mD, dI = T->dI, dM = T->dM, dD = T->dD;
        const unsigned eM0 = meta.y & 0xffffu, eI0 = meta.y >> 16;
        const unsigned fl = meta.x >> 16;
        const bool anysink = __ballot((fl & COL_FLAG_SINK) != 0) != 0;
        const bool anyfeed = __ballot((fl & COL_FLAG_FEED) != 0) != 0;
        // row-0 forward value of b_c and the entry term of M_c of this lane's column (only group-first lanes use them)
        const int cq = min(max(sstep - lp, 0), NC - 1);
        const double v0 = *(LdsDouble *)(size_t)(L.fwd_lin + (unsigned)cq * 16u);
        const double mX = *(LdsDouble *)(size_t)(L.fwd_lin + (unsigned)cq * 16u + 8u);
        double dgI = nI, dgM = nM, dgB = nB;
        double upI = 0.0, upM = 0.0, upB = 0.0;
        double *capq = rown + 3 * sstep + cap_lane;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double eI = *(LdsDouble *)(size_t)(eI0 + xof(k));
            const double eM = *(LdsDouble *)(size_t)(eM0 + xof(k));
            double accM = dgI * mI + dgM * mM;
            if (k == 0) accM = accM + (first_lane ? mX : 0.0);
            const double vM = (accM + dgB * mD) * eM;
            if (k == 0) {
                nI = rows_shift<G>(I[R - 1], nI);            // group-first lanes keep 0 for the whole sweep
                nM = rows_shift<G>(M[R - 1], nM);
                nB = rows_shift<G>(B[R - 1], v0);
                if (G == 2) nB = fix ? v0 : nB;              // (I and M arrive as 0 from the padding lane)
                upI = nI; upM = nM; upB = nB;
            }
            const double oI = I[k], oM = M[k], oB = B[k];
            const double vI = ((upI * iI + upM * iM) + upB * iD) * eI;
            double vB = (oI * dI + oM * dM) + oB * dD;
            if (anysink) {
                asm volatile("; fan-in column" ::);
                const bool sk = (fl & COL_FLAG_SINK) != 0;
                vB = sk ? er[k] : vB;
                er[k] = sk ? 0.0 : er[k];
            }
            I[k] = vI; M[k] = vM; B[k] = vB;
            if (kcap == k) { capq[0] = vI; capq[1] = vM; capq[2] = vB; }
            upI = vI; upM = vM; upB = vB;
            dgI = oI; dgM = oM; dgB = oB;
        }
        if (anyfeed) {
            asm volatile("; feeder column" ::);
            const double erw = (fl & COL_FLAG_FEED) ? T->erw : 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) er[k] = er[k] + B[k] * erw;
        }
        ++sstep;
        meta = meta_next;
    };
    int s = 0;
    for (; s < s_end; s += 2) { step(); step(); }
    if (s == s_end) step();
}

template <int R, int G>
__global__ void __launch_bounds__(COL_WAVES * 64, ROWS_WAVES_PER_SIMD)
forward_rows_kernel(ColArgs g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int W = 64 / G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t gw = (int64_t)blockIdx.x * COL_WAVES + wave;
    int32_t *tile_slot = (int32_t *)lds;
    uint8_t *tables = lds + 16;
    double *rown = g.rown + gw * g.rown_stride;
    const int grp = lane / W, lp = lane - grp * W;
    int cur_model = -1;
    bool padded = false;
    LdsTables L{};
    const ColProgram *cp = nullptr;
    DevModel M{};
    for (;;) {
        __syncthreads();
        if (tid == 0) *tile_slot = atomicAdd(g.tile_counter, 1);
        __syncthreads();
        const int ti = __builtin_amdgcn_readfirstlane(*tile_slot);
        if (ti >= g.n_tiles) break;
        const ColTile tile = g.tiles[ti];
        if (tile.model != cur_model) {
            cur_model = tile.model;
            M = g.a.models[cur_model];
            cp = M.cols;
            padded = stage_model<1>(cp, tables, g.lds_tables, g.lds_level, L, tid);
            // to the linear domain, in place (forward_columns.h): transition classes -> probabilities, emission records ->
            // probability times the per-row scale 16, row-0 / entry terms into a table of their own behind the staged ones
            double *cls = (double *)L.classes, *em = (double *)L.emis;
            for (int i = tid; i < cp->n_tclass * (int)(sizeof(ColClass) / 8); i += COL_WAVES * 64) cls[i] = exp(cls[i]);
            for (int i = tid; i < cp->n_eclass * COL_EMIS_STRIDE; i += COL_WAVES * 64) em[i] = exp(em[i]) * 16.0;
            double *lin = (double *)(tables + ((g.lds_tables + 15) & ~15));
            const double *fw = (const double *)((const uint8_t *)cp + cp->off_fwd);
            for (int i = tid; i < 2 * cp->n_cols; i += COL_WAVES * 64) lin[i] = exp(fw[i]);
            L.fwd_lin = lds_addr(lin);
            __syncthreads();
        }
        const int NC = __builtin_amdgcn_readfirstlane(cp->n_cols);
        const int64_t row_doubles = 3 * (int64_t)(NC + 2 * W) + COL_MAX_TAIL;
        for (int j = wave * G; j < tile.count; j += COL_WAVES * G) {
            const bool have = j + grp < tile.count;
            const int r = have ? g.a.order[tile.first + j + grp] : 0;
            const uint8_t *seq = g.a.bases + g.a.read_off[r];
            const int n = have ? (int)(g.a.read_off[r + 1] - g.a.read_off[r]) : 0;
            int nmax = n;
#pragma unroll
            for (int o = 32; o >= W; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
            nmax = __builtin_amdgcn_readfirstlane(nmax);
            if (!padded || nmax > R * (W - (G == 2 ? 1 : 0))) {                       // the host never routes such a tile here
                if (have && lp == 0) g.a.out_logp[r] = __longlong_as_double(0x7ff8000000000000ll);
                continue;
            }
            rows_sweep_fwd<R, G>(L, NC, NC - 1 + (nmax - 1) / R, seq, n, lp, lane, rown, (unsigned)(grp * row_doubles));
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
#pragma unroll 1
            for (int q = 0; q < G; ++q) {
                if (j + q >= tile.count) break;
                const int rq = __builtin_amdgcn_readfirstlane(g.a.order[tile.first + j + q]);
                const int nq = __builtin_amdgcn_readfirstlane((int)(g.a.read_off[rq + 1] - g.a.read_off[rq]));
                double *final_row = rown + q * row_doubles + 3 * W;
                col_row_to_log(final_row, NC, nq, 0.0, lane);
                __threadfence_block();
                __builtin_amdgcn_wave_barrier();
                const double logp = col_tail_forward(cp, final_row, NC, lane);
                if (lane == 0) g.a.out_logp[rq] = logp;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}
