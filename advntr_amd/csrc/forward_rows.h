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
// tw: the in-edge weights exp(transition log-probability) of the model, made once when the model was staged (LDS), or null:
// then every read exponentiates them again -- a thousand exponentials per read on a model whose tail states collect an edge
// from every column, 3 % of the kernel.
__device__ __forceinline__ double rows_tail_forward_linear(const ColProgram *__restrict__ cp, double *__restrict__ row,
                                                           double *__restrict__ tailv, const int rows, const int lane,
                                                           const double *__restrict__ tw)
{
    const ColFinishTables F = col_finish_tables(cp);
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
            sum = fma(v, tw ? tw[e] : exp(ed.logp), sum);
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
// Back-to-back sweeps as in rows_sweep (viterbi_rows.h): a lane group takes up to ROWS_DEPTH reads one behind the
// other along the step axis (`queue` = the lane's rows of the reads after the first, rows_pack_read; `depth` = reads per
// group), so the W - 1 steps of filling and draining the lanes are paid once per sweep instead of once per read.  Nothing is
// reset between reads: column 0 of a column program takes nothing from a previous column (probability 0 here), the fan-in
// accumulators are 0 after the last sink; the info and row-0 tables hold copies of columns 0 .. 32 behind the last column, the
// table pointers step back by NC records once every lane of the group is on its next read.  Row n of read k lands at
// rown[cap_base + 3 (W + k NC + c)].  The plain stretches between two reads run the step without any of this.
template <int R, int G>
__device__ __forceinline__ void rows_sweep_fwd(const LdsTables &L, const int NC, const int s_end,
                                               const uint8_t *__restrict__ seq, const int n, const int lp, const int lane,
                                               double *__restrict__ rown, const unsigned cap_base,
                                               unsigned long long queue = 0ull, const int depth = 1)
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
    int kcap = (n >= 1 && (n - 1) / R == lp) ? (n - 1) - lp * R : 7;          // slot of the read's last row (7: not in this lane)
    const bool first_lane = lp == 0;
    const bool fix = G == 2 && lane == 32;
    double nI = 0.0, nM = 0.0, nB = 0.0, qI = 0.0, qM = 0.0;      // group-first lanes keep 0 for the whole sweep
    unsigned pa = L.pinfo + (unsigned)(64 - lp) * 16u;
    unsigned pf = L.fwd_lin + (unsigned)(64 - lp) * 16u;
    uint2 meta = lds_uint2(pa + 8u);
    const unsigned cap_lane = cap_base + (unsigned)(W - lp) * 3u;
    int sstep = 0;
    auto step = [&](auto WIN, double &nI, double &nM, double &qI, double &qM, const int uw, const bool more) {
        constexpr int win_kind = decltype(WIN)::value;
        const adv_f64x2 f0 = *(LdsDouble2 *)(size_t)pf;          // {row-0 value of b_c, entry term of M_c} of the lane's column
        pa += 16u; pf += 16u;
        const uint2 meta_next = lds_uint2(pa + 8u);
        // the column's transition class record, 96 bytes as six 16-byte words (ColClass)
        LdsDouble2 *T2 = (LdsDouble2 *)(size_t)(meta.x & 0xffffu);
        const adv_f64x2 t0 = T2[0], t1 = T2[1], t2 = T2[2], t3 = T2[3], t4 = T2[4], t5 = T2[5];
        const double iI = t0.x, iM = t0.y, iD = t1.x, mI = t1.y, mM = t2.x, mD = t3.x, dI = t3.y, dM = t4.x, dD = t4.y;
        const double erw_c = t5.x;
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
        if (win_kind != 0) {
            // lane uw of a group is on its read's last column on this step: it takes its rows of the next read out of the queue
            const bool mine = lp == uw;
            if (more) {
                if (mine) {
                    const unsigned w = (unsigned)queue;
#pragma unroll
                    for (int k = 0; k < R; ++k) esym[k] = L.epair_base + __umul24((w >> (3 * k)) & 7u, L.epair_sym_stride);
                    kcap = (int)((w >> (3 * R)) & 7u);
                    queue >>= 3 * R + 3;
                }
                if (uw == W) { pa -= 16u * (unsigned)NC; pf -= 16u * (unsigned)NC; }     // every lane of the group is on its next read
            } else {
                kcap = mine ? 7 : kcap;          // behind the last read nothing is captured: the lanes walk on over copied columns
            }
        }
        ++sstep;
        meta = meta_next;
    };
    using Plain = std::integral_constant<int, 0>;
    using Window = std::integral_constant<int, 1>;
    int s = 0;
    for (int k = 0; k < depth; ++k) {
        const int first_turn = (k + 1) * NC - 1;                  // the step on which lane 0 is on its last column
        for (const int s1 = s + ((first_turn - s) & ~1); s < s1; s += 2) {
            step(Plain{}, nI, nM, qI, qM, 0, false);
            step(Plain{}, qI, qM, nI, nM, 0, false);
        }
        const bool more = k + 1 < depth;
        for (const int s1 = more ? first_turn + W + 1 : s_end + 1; s < s1; s += 2) {     // (a step too many does nothing)
            step(Window{}, nI, nM, qI, qM, s - first_turn, more);
            step(Window{}, qI, qM, nI, nM, s + 1 - first_turn, more);
        }
    }
}

#ifndef FWD_WAVES_PER_SIMD
#define FWD_WAVES_PER_SIMD ROWS_WAVES_PER_SIMD
#endif
template <int R, int G>
__global__ void __launch_bounds__(COL_WAVES * 64, FWD_WAVES_PER_SIMD)
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
    double *tailw = nullptr;
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
            padded = stage_model<1, true, true>(cp, tables, g.lds_tables, g.lds_level, L, tid);
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
                int c = (i >> 1) - 64;
                // behind the last column: copies of columns 0 .. 32, like the padded info table (back-to-back sweeps)
                if (c >= ncol && c - ncol <= 32 && c - ncol < ncol) c -= ncol;
                c = min(max(c, 0), ncol - 1);
                lin[i] = exp(fw[2 * c + (i & 1)]);
            }
            L.fwd_lin = lds_addr(lin);
            // the tail states' in-edge weights in the linear domain, behind the row-0 table, when the launch left room for them
            {
                const ColFinishTables F = col_finish_tables(cp);
                const int n_te = F.tptr[F.n_tail];
                tailw = nullptr;
                if (n_te <= g.fwd_tailw_cap) {
                    tailw = lin + 2 * (ncol + 128);
                    for (int e = tid; e < n_te; e += COL_WAVES * 64) tailw[e] = exp(F.edges[e].logp);
                }
            }
            __syncthreads();
        }
        const int NC = __builtin_amdgcn_readfirstlane(cp->n_cols);
        // per lane group: the row-n values of its reads one behind the other (padded by W columns either side); the tail values
        // of the read being finished sit behind the groups -- the layout of viterbi_rows_kernel, whose scratch this is
        const int dmax = NC >= ROWS_STREAM_MIN_COLS ? g.rows_depth : 1;
        const int64_t grp_doubles = 3 * ((int64_t)dmax * NC + 2 * W);
        double *tailv = rown + G * grp_doubles;
        for (int j0 = 0; j0 < tile.count; j0 += COL_WAVES * G * dmax) {
            const int jw = j0 + wave * G;
            if (jw >= tile.count) break;
            const int depth = min(dmax, (tile.count - jw + COL_WAVES * G - 1) / (COL_WAVES * G));
            // the reads of this sweep: lane k * G + q fetches what read k of lane group q is
            int fr = -1, fn = 0;
            long long fo = 0;
            {
                const int k = lane / G, q = lane - k * G;
                const int idx = jw + k * COL_WAVES * G + q;
                if (k < depth && idx < tile.count) {
                    fr = g.a.order[tile.first + idx];
                    fo = g.a.read_off[fr];
                    fn = (int)(g.a.read_off[fr + 1] - fo);
                }
            }
            int nbad = (!padded || fn > R * (W - (G == 2 ? 1 : 0))) ? 1 : 0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) nbad |= __shfl_xor(nbad, o, 64);
            if (__builtin_amdgcn_readfirstlane(nbad)) {                    // the host never routes such a tile here
                if (fr >= 0) g.a.out_logp[fr] = __longlong_as_double(0x7ff8000000000000ll);
                continue;
            }
            unsigned long long queue = 0ull;
            for (int k = depth - 1; k >= 1; --k) {
                const int src = k * G + grp;
                const int nk = __shfl(fn, src, 64);
                const long long ok = ((long long)__shfl((int)(fo >> 32), src, 64) << 32) | (unsigned)__shfl((int)fo, src, 64);
                queue = (queue << (3 * R + 3)) | rows_pack_read<R>(g.a.bases + ok, nk, lp);
            }
            const int n = __shfl(fn, grp, 64);
            const uint8_t *seq = g.a.bases + (((long long)__shfl((int)(fo >> 32), grp, 64) << 32) | (unsigned)__shfl((int)fo, grp, 64));
            int nlast = 0;
#pragma unroll
            for (int q = 0; q < G; ++q) nlast = max(nlast, __builtin_amdgcn_readlane(fn, (depth - 1) * G + q));
            const int s_end = depth * NC - 1 + (max(nlast, 1) - 1) / R;
            rows_sweep_fwd<R, G>(L, NC, s_end, seq, n, lp, lane, rown, (unsigned)(grp * grp_doubles), queue, depth);
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
#pragma unroll 1
            for (int k = 0; k < depth; ++k) {
#pragma unroll 1
                for (int q = 0; q < G; ++q) {
                    if (jw + k * COL_WAVES * G + q >= tile.count) break;
                    const int src = k * G + q;
                    const int rq = __builtin_amdgcn_readlane(fr, src), nq = __builtin_amdgcn_readlane(fn, src);
                    const double logp = rows_tail_forward_linear(cp, rown + q * grp_doubles + 3 * (W + (int64_t)k * NC), tailv, nq, lane, tailw);
                    if (lane == 0) g.a.out_logp[rq] = logp;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
}
