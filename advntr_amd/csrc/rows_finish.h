// rows_finish.h -- everything after a back-to-back sweep of viterbi_rows_kernel, for ALL reads of the sweep at once.
//
// The reads a wavefront has just swept (up to ROWS_DEPTH per lane group, G groups: 8 or 16) share one model.  Finishing them
// one after the other (rounds 1-4: tail states, traceback, summary per read) is a chain of dependent round trips to L2 during
// which the wavefront issues next to nothing -- 19 % of the S300 launch and 8 % of the REF150 launch for 9 % / 3 % of their
// vector instructions (scripts/budget_finish.sh, profiles/r05_finish_budget.json).  Here the same work is laid out so that
// the round trips of different reads are in flight together:
//
//   rows_tails   the tail states' maxima of all reads in one pass over the tail edges: an edge record is loaded once and
//                relaxed against every read's captured row; the chain of dependent tail states is walked once, not per read;
//   rows_finish_lanes   tracebacks and path summaries DEFERRED until the wavefront has swept 64 reads (their back-pointer slabs
//                stay in HBM meanwhile), then one read per LANE: every lane walks its own path cell by cell -- the plain
//                serial traceback of hmm.pyx:2107-2136 on the column layout -- and counts what advntr/hmm_utils.py:155-286
//                derives from the path as it goes (no second pass, no reversed-path buffer unless paths are asked for).  A step
//                costs the wavefront ~50 vector instructions for 64 reads instead of ~57 for one gather of one read, and the
//                ~300 dependent round trips of the longest path are paid once per 64 reads instead of ~28 per read.
//
// (A first round-5 version kept the wave-cooperative walk and ran 2, 4 or 8 of them in lock step: bit-exact, and no faster --
// S300 2.64 / 2.71 / 2.74 ms against 2.64: the scalar state of several walks does not fit the scalar registers next to the
// kernel's own, and what the lock step saved in round trips it spent in v_readlane / v_writelane.)
//
// Results are identical to the per-read route by construction (same candidates, same order, same first-maximum rule); the
// parity tests and fuzz scripts compare them with the oracle and with the anti-diagonal kernel, which still finishes per read.
#pragma once
#include "viterbi_columns.h"

#define ROWS_FINISH_MAXQ (ROWS_DEPTH * ROWS_MAX_GROUPS)       // reads of one sweep at most
#define ROWS_PEND_READS 64                                    // reads a wavefront sweeps before it finishes them, one per lane
#define ROWS_PEND_SLABS (ROWS_PEND_READS / (ROWS_DEPTH * 2))  // ... = back-pointer slabs (and fan-in winner blocks) per wavefront
#define ROWS_PEND_INTS (ROWS_PEND_READS * 8)                  // their descriptors (RowsPend) and
#define ROWS_TAILLOC_INTS (ROWS_PEND_READS * COL_MAX_TAIL)    // tail winners, at the end of the wavefront's `aux` scratch

__device__ __forceinline__ double wave_max_f64_raw(double v)
{
    // wave_max_f64 without fmax()'s canonicalising self-maxima (no NaN reaches the tail: sums of finite values and -inf)
    auto mx = [](double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    v = mx(v, wave_dpp_f64<0xB1>(v));
    v = mx(v, wave_dpp_f64<0x4E>(v));
    v = mx(v, wave_dpp_f64<0x141>(v));
    v = mx(v, wave_dpp_f64<0x140>(v));
    v = mx(v, wave_dpp_f64<0x142, 0xA>(v));
    v = mx(v, wave_dpp_f64<0x143, 0xC>(v));
    const long long b = __double_as_longlong(v);
    return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(b >> 32), 63) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)b, 63));
}

// Tail states at the last row for NQ reads of one model (col_tail for many reads).  roff[q]: where read q's captured row
// starts, in doubles from `rown`; tailv_all / tailloc_all: NQ x COL_MAX_TAIL scratch of the wave (tail values; the `loc` of the
// winning in-edge -- what the traceback needs of it).  logp[q] = value of the model's end state.
template <int NQ>
__device__ __forceinline__ void rows_tails(const ColFinishTables &F, const double *__restrict__ rown, const unsigned (&roff)[NQ],
                                           double *__restrict__ tailv_all, int32_t *__restrict__ tailloc_all, const int lane,
                                           double (&logp)[NQ])
{
    const int n_tail = F.n_tail, end_tail = F.end_tail;
    int e1 = F.tptr[0];
#pragma unroll
    for (int q = 0; q < NQ; ++q) logp[q] = -INFINITY;
    for (int i = 0; i < n_tail; ++i) {
        const int e0 = e1;
        e1 = F.tptr[i + 1];
        double best[NQ];
        int rank[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { best[q] = -INFINITY; rank[q] = 0x7fffffff; }
        for (int eb = e0; eb < e1; eb += 64) {
            const int e = eb + lane;
            const TailEdge ed = F.edges[min(e, e1 - 1)];
            const bool in = e < e1, isrow = ed.loc >= 0;
            const unsigned off_row = (unsigned)(ed.loc >> 2) * 3u + (unsigned)(ed.loc & 3), off_tail = (unsigned)(-ed.loc - 1);
            double v[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                v[q] = isrow ? rown[roff[q] + off_row] : tailv_all[q * COL_MAX_TAIL + off_tail];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const double cand = v[q] + ed.logp;
                const bool take = in && cand > best[q];
                best[q] = take ? cand : best[q];
                rank[q] = take ? e : rank[q];
            }
        }
        double top[NQ];
        int first[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            top[q] = wave_max_f64_raw(best[q]);
            first[q] = wave_min_i32(best[q] == top[q] ? rank[q] : 0x7fffffff);
        }
        int loc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) loc[q] = F.edges_u[first[q] != 0x7fffffff ? first[q] : e0].loc;      // (scalar loads, all in flight)
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) { tailv_all[q * COL_MAX_TAIL + i] = top[q]; tailloc_all[q * COL_MAX_TAIL + i] = loc[q]; }
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        if (i == end_tail) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) logp[q] = top[q];
        }
    }
}

// One read waiting for its traceback: what the sweep knew about it (32 bytes, written by the lane that held it in the sweep).
struct RowsPend {
    int32_t r, n, off_lo, off_hi;     // read index, length, offset of its bases
    int32_t model, nc;                // index into BatchArgs::models, columns of that model
    int32_t key;                      // slab | position in its lane group's back-to-back sweep << 4 | lane group << 8 |
                                      // fan-in winner block << 12 | present << 24 | has a path (log-probability > -inf) << 25
    int32_t pad;
};
static_assert(sizeof(RowsPend) == 32, "RowsPend is stored as two 16-byte words");

typedef __attribute__((address_space(1))) const int GlobalInt;

// Deferred finish, one read per lane.  Lane p takes pending read p (slot = sweep * NQ + position in the sweep): its tail chain
// (the winners rows_tails recorded), then cell by cell back through the trellis along the back-pointer masks of its sweep's
// slab -- M_c(t) <- (t-1, c-1), I_c(t) <- (t-1, c), b_c(t) <- (t, c-1), a fan-in sink <- the winner the sweep recorded --, then
// the row-0 silent chain.  Every visited state adds to the lane's summary counters through the per-column class words
// (ColCls: no dependent lookup by state index), with the running base-pair count of hmm_utils.py:171 read off the row: an
// emitting state visited in row t is the t-th emitting state of the path.  (Models whose class words do not mark exactly the
// emitting states as emitting never come here: engine.hip routes them to the anti-diagonal kernel.)
template <int R, int G>
__device__ __forceinline__ void rows_finish_lanes(const ColArgs &g, const uint32_t flags, const int n_slots,
                                                  const unsigned *__restrict__ bp_wave, const int32_t *__restrict__ aux_wave,
                                                  int32_t *__restrict__ rev_wave, const int lane)
{
    constexpr int W = 64 / G, WORDS = (R + 4) / 5;
    const bool want_summary = g.a.out_summary && !(flags & 4u), want_path = g.a.out_path && (flags & 1u);
    if (!want_summary && !want_path) return;
    const RowsPend *pend = (const RowsPend *)(aux_wave + (g.aux_stride - ROWS_PEND_INTS - ROWS_TAILLOC_INTS));
    const int32_t *tailloc = aux_wave + (g.aux_stride - ROWS_TAILLOC_INTS) + lane * COL_MAX_TAIL;
    if (lane >= n_slots) return;
    const int4 d0 = ((const int4 *)(pend + lane))[0], d1 = ((const int4 *)(pend + lane))[1];
    const int key = d1.z;
    if (!((key >> 24) & 1)) return;
    const int r = d0.x, n = d0.y, NC = d1.y;
    const uint8_t *seq = g.a.bases + (((long long)d0.w << 32) | (unsigned)d0.z);
    const DevModel *Mp = g.a.models + d1.x;
    const int P = Mp->P, start_state = Mp->start, m = Mp->m;
    const uint16_t *sclass = Mp->sclass;
    const uint8_t *cpb = (const uint8_t *)Mp->cols;
    const ColProgram *cp = (const ColProgram *)cpb;
    const int end_tail = cp->end_tail, n_tail = cp->n_tail;
    const int32_t *tstate = (const int32_t *)(cpb + cp->off_tail_state);
    const int32_t *pred0 = (const int32_t *)(cpb + cp->off_pred0);
    const ColState *state = (const ColState *)(cpb + cp->off_state);
    const uint2 *colcls = (const uint2 *)(cpb + cp->off_colcls);
    const int slab = key & 15, kq = (key >> 4) & 15, gq = (key >> 8) & 15, sinkblock = (key >> 12) & 0xfff;
    const unsigned *bp = bp_wave + (size_t)slab * (size_t)(g.rows_slab_bytes / 4);
    const int32_t *sinks = aux_wave + COL_MAX_TAIL + (int64_t)slab * g.rows_sink_slab + (int64_t)sinkblock * COL_MAX_SINKS * g.sink_stride;
    int32_t *rev = rev_wave + (int64_t)lane * g.a.path_cap;
    const int kNC = kq * NC, lane0 = gq * W;
    int len = 0;
    int matches = 0, rep_bp = 0, left_bp = 0, right_bp = 0, lm = 0, rm = 0, starts = 0, ends = 0;
    int first_start = -1, last_start = -1, first_end = -1, last_end = -1;
    // a visited state: class word, base pairs emitted up to and including it.  The walk runs against the path, so the first
    // start / end seen is the path's last one.
    auto visit = [&](const unsigned cls, const int cur_bp) {
        const bool emit = (cls & SC_EMIT) != 0;
        matches += (cls & SC_MATCH) ? 1 : 0;
        rep_bp += (emit && !(cls & SC_FIX)) ? 1 : 0;
        left_bp += (emit && (cls & SC_SUFFIX)) ? 1 : 0;
        right_bp += (emit && (cls & SC_PREFIX)) ? 1 : 0;
        const int seq_idx = cur_bp - (emit ? 1 : 0);
        if (!(cls & SC_SKIP) && (cls & SC_MATCH) && (cls & SC_BASE_VALID) && seq_idx >= 0 && seq_idx < n) {
            const bool hit = seq[seq_idx] == ((cls >> SC_BASE_SHIFT) & 3u);
            lm += (hit && (cls & SC_SUFFIX)) ? 1 : 0;
            rm += (hit && (cls & SC_PREFIX)) ? 1 : 0;
        }
        if ((cls & SC_UNIT_START) && n - cur_bp >= 3) { ++starts; if (last_start < 0) last_start = cur_bp; first_start = cur_bp; }
        if ((cls & SC_UNIT_END) && cur_bp >= 3) { ++ends; if (last_end < 0) last_end = cur_bp; first_end = cur_bp; }
    };
    bool failed = false;
    if ((key >> 25) & 1) {
        const int len_max = n + m;          // the reference's own path buffer (hmm.pyx:1953): a longer path is refused
        // ---- tail states (row n); the first one is the model's end state, which the summaries leave out
        int c = 0, slot = 0;
        {
            int ti = end_tail;
            for (int it = 0;; ++it) {
                if (it >= n_tail) { failed = true; break; }
                const int st = tstate[ti];
                if (want_path) rev[len] = st;
                if (len > 0 && want_summary && sclass) visit(sclass[st], n);
                ++len;
                const int loc = tailloc[ti];
                if (loc < 0) { ti = -loc - 1; continue; }
                c = loc >> 2;
                slot = loc & 3;
                break;
            }
        }
        // ---- the trellis
        int t = n, s0 = -1;
        while (!failed && t >= 1) {
            if (len > len_max) { failed = true; break; }
            const int tr = t - 1, lp = tr / R, kk = tr - lp * R, ln = lane0 + lp;
            const int grp = slot == 1 ? 0 : (slot == 0 ? 1 : 2);           // masks of a cell in relaxation order: M, I, b
            const unsigned *cell = bp + ((unsigned)(max(c, 0) + kNC + lp) * (unsigned)(64 * WORDS) + (unsigned)(kk * 12 + grp * 4 + (ln >> 5)));
            // (written with scalar stores, which do not pass through the vector L1: agent-scope loads go around it)
            const unsigned aw = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned bw = __hip_atomic_load(cell + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c = max(c, -1);                              // (column -1 is the dummy in front of the tables; no path goes there)
            const uint2 cc = colcls[c + 1];              // class words of the column's I, M, b states and its flags
            if (want_path) rev[len] = slot == 1 ? state[c + 1].sM : (slot == 0 ? state[c + 1].sI : state[c + 1].sB);
            ++len;
            if (want_summary) visit(slot == 1 ? (cc.x >> 16) : (slot == 0 ? (cc.x & 0xffffu) : (cc.y & 0xffffu)), t);
            const unsigned fl = cc.y >> 16;
            if (slot == 2 && (fl & COL_FLAG_SINK)) {
                // a fan-in sink: its predecessor is the winner the sweep recorded for this row (a feeder's b cell), not a pointer
                c = sinks[((fl >> 4) & 15) * g.sink_stride + t] - kNC;
                continue;
            }
            const unsigned a = (aw >> (ln & 31)) & 1u, b = (bw >> (ln & 31)) & 1u;      // a: the 2nd candidate won, b: the last one did
            if (slot == 1) {                         // M_c(t) <- [I, M (row 1: the entry edge), b](t - 1, c - 1)
                if (t == 1 && !b && a) { s0 = state[c + 1].sX; t = 0; break; }
                slot = b ? 2 : (int)a;
                --t; --c;
            } else if (slot == 0) {                  // I_c(t) <- [I, M, b](t - 1, c)
                slot = b ? 2 : (int)a;
                --t;
            } else {                                 // b_c(t) <- [I, M, b](t, c - 1)
                slot = b ? 2 : (int)a;
                --c;
            }
        }
        // ---- row 0: the silent chain back to the model's start state (left out of the summaries like the end state)
        if (!failed) {
            if (s0 < 0) s0 = state[c + 1].sB;
            while (s0 != start_state) {
                if (len > len_max || s0 < P) { failed = true; break; }
                if (want_path) rev[len] = s0;
                if (want_summary && sclass) visit(sclass[s0], 0);
                ++len;
                s0 = pred0[s0 - P];
            }
            if (!failed) { if (want_path) rev[len] = start_state; ++len; }
        }
        if (len > len_max) failed = true;
    }
    const int plen = failed ? -2 : len;
    if (want_summary) {
        int4 *o = (int4 *)(g.a.out_summary + (int64_t)r * 8);
        if (plen > 0) {
            int delta = 0;
            if (first_start >= 0 && first_end >= 0 && first_end < first_start && last_start > last_end) delta = 1;
            o[0] = make_int4((starts > ends ? starts : ends) + delta, matches, rep_bp, left_bp);
            o[1] = make_int4(right_bp, lm, rm, plen);
        } else {
            o[0] = make_int4(0, 0, 0, 0);
            o[1] = make_int4(0, 0, 0, plen);
        }
    }
    if (want_path) {
        const int64_t o0 = g.a.out_path_off[r];
        const int capo = (int)(g.a.out_path_off[r + 1] - o0);
        int olen = plen;
        if (plen > capo) olen = -2;
        if (olen > 0)
            for (int i = 0; i < plen; ++i) g.a.out_path[o0 + i] = rev[plen - 1 - i];
        g.a.out_path_len[r] = olen;
    }
}
