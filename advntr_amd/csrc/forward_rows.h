// forward_rows.h -- sum-product (Model.log_probability) in the row-blocked layout of viterbi_rows.h, for the short reads of
// a large batch (G reads per wavefront, R rows per lane).  The recurrence is forward_columns.h's: LINEAR domain,
// probabilities times 16^row (the class and emission tables are exponentiated in LDS when a model is staged, emissions
// times the per-row scale 16; a sweep of at most 155 rows neither overflows nor underflows), the last row is captured
// as scaled probabilities, taken back to the log domain after the sweep and handed to col_tail_forward.  No back-pointers,
// no traceback.  Results agree with the reference's pair_lse folding to rounding (tests: 1e-9 relative; measured 3e-14).
#pragma once
#include "viterbi_rows.h"
#include "forward_columns.h"

// Tail states (the silent states evaluated at the last row only) on the captured row as the sweep left it: scaled
// probabilities.  col_tail_forward folds log-domain values with pair_lse, an exp and a log per edge and per step of the
// wave reduction, after every one of the 3 NC row values was taken to the log domain; here a tail state is the plain sum of
// value x transition probability over its in-edges, and ONE logarithm is taken at the end (cycle counters in the kernel: the
// log-domain version was 19 % of a wavefront's time).  Tail values of earlier tail states are kept in the same scale.
__device__ __forceinline__ double rows_tail_forward_linear(const ColProgram *__restrict__ cp, double *__restrict__ row,
                                                           const int NC, const int rows, const int lane)
{
    const ColFinishTables F = col_finish_tables(cp);
    double *tailv = row + 3 * NC;
    const int n_tail = F.n_tail, end_tail = F.end_tail;
    double result = 0.0;
    int e1 = F.tptr[0];
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < n_tail; ++i) {
        const int e0 = e1;
        e1 = F.tptr[i + 1];
        double sum = 0.0;
        for (int e = e0 + lane; e < e1; e += 64) {
            const TailEdge ed = F.edges[e];
            const double v = ed.loc >= 0 ? row[(ed.loc >> 2) * 3 + (ed.loc & 3)] : tailv[-ed.loc - 1];
            sum = fma(v, exp(ed.logp), sum);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) tailv[i] = sum;
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        if (i == end_tail) result = sum;
    }
    return log(result) - (double)rows * 2.772588722239781;          // rows * log(16)
}

// The sweep mirrors rows_sweep (viterbi_rows.h) in its bookkeeping: emission {M, I} pairs from the symbol-major pair table
// (one address add and one 16-byte LDS read per cell, all requested at the top of the step), the two per-step flags as bytes
// of the padded info word, ping-pong registers for the values shifted in from the neighbouring lane, and the row-0 terms
// {b_c(0), entry term of M_c} from a table padded like the info table (one running address, no clamping).  The
// multiply-adds are explicit fma()s: the build contracts nothing by itself (-ffp-contract=off, for the Viterbi kernels and
// the host builder), and 11 fused operations per cell instead of 17 separate ones is a third of this kernel's arithmetic.
template <int R, int G>
__device__ __forceinline__ void rows_sweep_fwd(const LdsTables &L, const int NC, const int s_end,
                                               const uint8_t *__restrict__ seq, const int n, const int lp, const int lane,
                                               double *__restrict__ rown, const unsigned cap_base)
{
    constexpr int W = 64 / G;
    double I[R], M[R], B[R], er[R];
    unsigned esym[R];                  // LDS address of the emission-pair row of the base in the lane's kth row
#pragma unroll
    for (int k = 0; k < R; ++k) {
        I[k] = M[k] = B[k] = er[k] = 0.0;
        const int t = R * lp + k + 1;
        esym[k] = L.epair_base + (unsigned)((t <= n) ? (int)seq[t - 1] : 4) * L.epair_sym_stride;     // rows past the read: emission 0
    }
    const int kcap = (n >= 1 && (n - 1) / R == lp) ? (n - 1) - lp * R : -1;
    const bool first_lane = lp == 0;
    const bool fix = G == 2 && lane == 32;
    double nI = 0.0, nM = 0.0, nB = 0.0, qI = 0.0, qM = 0.0;      // group-first lanes keep 0 for the whole sweep
    unsigned pa = L.pinfo + (unsigned)(64 - lp) * 16u;
    unsigned pf = L.fwd_lin + (unsigned)(64 - lp) * 16u;
    uint2 meta = lds_uint2(pa + 8u);
    const unsigned cap_lane = cap_base + (unsigned)(W - lp) * 3u;
    int sstep = 0;
    auto step = [&](double &nI, double &nM, double &qI, double &qM) {
        const adv_f64x2 f0 = *(LdsDouble2 *)(size_t)pf;          // {row-0 value of b_c, entry term of M_c} of the lane's column
        pa += 16u; pf += 16u;
        const uint2 meta_next = lds_uint2(pa + 8u);
        LdsClass *T = (LdsClass *)(size_t)(meta.x & 0xffffu);
        const double iI = T->iI, iM = T->iM, iD = T->iD, mI = T->mI, mM = T->mM, mD = T->mD, dI = T->dI, dM = T->dM, dD = T->dD;
        const double erw_c = T->erw;
        const unsigned epo = meta.y;
        const bool on_sink = (meta.x >> 24) != 0u;
        const unsigned feedb = (meta.x >> 16) & 0xffu;
        const bool on_feed = feedb != 0u;
        unsigned long long sinkmask = __ballot(on_sink);
        const bool anyfeed = __ballot(on_feed) != 0;
        adv_f64x2 e_next = *(LdsDouble2 *)(size_t)(esym[0] + epo);      // one cell ahead (all R up front spills in this kernel)
        const double v0 = f0.x, mX = f0.y;
        double dgI = nI, dgM = nM, dgB = nB;
        double upI = 0.0, upM = 0.0, upB = 0.0;
        double *capq = rown + 3 * sstep + cap_lane;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double eM = e_next.x, eI = e_next.y;
            if (k + 1 < R) e_next = *(LdsDouble2 *)(size_t)(esym[k + 1] + epo);
            double accM = fma(dgI, mI, dgM * mM);
            if (k == 0) accM = accM + (first_lane ? mX : 0.0);
            const double vM = fma(dgB, mD, accM) * eM;
            const double oI = I[k], oM = M[k], oB = B[k];
            double vB = fma(oB, dD, fma(oI, dI, oM * dM));
            if (k == 0) {
                qI = rows_shift<G>(I[R - 1], qI);
                qM = rows_shift<G>(M[R - 1], qM);
                nB = rows_shift<G>(B[R - 1], v0);
                if (G == 2) nB = fix ? v0 : nB;              // (I and M arrive as 0 from the padding lane)
                upI = qI; upM = qM; upB = nB;
            }
            const double vI = fma(upB, iD, fma(upI, iI, upM * iM)) * eI;
            asm volatile("" : "+s"(sinkmask));               // re-tested as a scalar at every cell (viterbi_rows.h)
            if (sinkmask != 0ull) {
                asm volatile("; fan-in column" ::);
                vB = on_sink ? er[k] : vB;
                er[k] = on_sink ? 0.0 : er[k];
            }
            I[k] = vI; M[k] = vM; B[k] = vB;
            if (kcap == k) { capq[0] = vI; capq[1] = vM; capq[2] = vB; }
            upI = vI; upM = vM; upB = vB;
            dgI = oI; dgM = oM; dgB = oB;
        }
        if (anyfeed) {
            asm volatile("; feeder column" ::);
            const double erw = on_feed ? erw_c : 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) er[k] = fma(B[k], erw, er[k]);
        }
        ++sstep;
        meta = meta_next;
    };
    int s = 0;
    for (; s < s_end; s += 2) { step(nI, nM, qI, qM); step(qI, qM, nI, nM); }
    if (s == s_end) step(nI, nM, qI, qM);
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
            padded = stage_model<1, true>(cp, tables, g.lds_tables, g.lds_level, L, tid);
            // to the linear domain, in place (forward_columns.h): transition classes -> probabilities, emission pairs ->
            // probability times the per-row scale 16 (the fifth symbol row, -inf, becomes 0), row-0 / entry terms into a table of
            // their own behind the staged ones, padded like the info table (64 records in front, clamped at either end)
            double *cls = (double *)L.classes;
            double *ep = (double *)(tables + (cp->off_epair - cp->off_class));
            for (int i = tid; i < cp->n_tclass * (int)(sizeof(ColClass) / 8); i += COL_WAVES * 64) cls[i] = exp(cls[i]);
            for (int i = tid; i < COL_EPAIR_SYMBOLS * cp->n_epair * 2; i += COL_WAVES * 64) ep[i] = exp(ep[i]) * 16.0;
            double *lin = (double *)(tables + ((g.lds_tables + 15) & ~15));
            const double *fw = (const double *)((const uint8_t *)cp + cp->off_fwd);
            const int ncol = cp->n_cols;
            for (int i = tid; i < 2 * (ncol + 128); i += COL_WAVES * 64) {
                const int c = min(max((i >> 1) - 64, 0), ncol - 1);
                lin[i] = exp(fw[2 * c + (i & 1)]);
            }
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
                const double logp = rows_tail_forward_linear(cp, final_row, NC, nq, lane);
                if (lane == 0) g.a.out_logp[rq] = logp;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}
