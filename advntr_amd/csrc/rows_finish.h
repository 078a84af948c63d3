// rows_finish.h -- everything after a back-to-back sweep of viterbi_rows_kernel, for ALL reads of the sweep at once.
//
// The reads a wavefront has just swept (up to ROWS_DEPTH per lane group, G groups: 8 or 16) share one model.  Finishing them
// one after the other (rounds 1-4: tail states, traceback, summary per read) is a chain of dependent round trips to L2 during
// which the wavefront issues next to nothing -- 19 % of the S300 launch and 8 % of the REF150 launch for 9 % / 3 % of their
// vector instructions (scripts/budget_finish.sh, profiles/r05_finish_budget.json).  Here the same work is laid out so that
// the round trips of different reads are in flight together:
//
//   rows_finish_lanes   everything after the sweep DEFERRED until the wavefront has swept 64 reads (their back-pointer slabs,
//                captured last rows and fan-in winners stay in HBM meanwhile), then one read per LANE: every lane evaluates its
//                read's tail states edge by edge (hmm.pyx:2065-2083 for those states) and walks its own path back through the
//                trellis -- the plain serial traceback of hmm.pyx:2107-2136 on the column layout, ROWS_WALK_K cells of the
//                current run fetched per round trip -- counting what advntr/hmm_utils.py:155-286 derives from the path as it
//                goes (no second pass, no cross-lane reduction, no reversed-path buffer unless paths are asked for).  The
//                dependent round trips of the longest path are paid once per 64 reads instead of ~28 per read.
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

// One read waiting for its finish: what the sweep knew about it (32 bytes, written by the lane that held it in the sweep).
struct RowsPend {
    int32_t r, n, off_lo, off_hi;     // read index, length, offset of its bases
    int32_t model, nc;                // index into BatchArgs::models, columns of that model
    int32_t key;                      // slab | position in its lane group's back-to-back sweep << 4 | lane group << 8 |
                                      // fan-in winner block << 12 | present << 24
    int32_t row_off;                  // where its captured last row starts, in doubles from the wavefront's row scratch
};
static_assert(sizeof(RowsPend) == 32, "RowsPend is stored as two 16-byte words");

#ifndef ROWS_WALK_K
#define ROWS_WALK_K 1                 // cells of the current run a lane fetches per round trip (measured: 1 -> S300 2.50 ms, 2 -> 2.52, 4 ->
                                      // 2.58: at the end of a launch every wavefront walks at once and the masks come from HBM a 128-byte line
                                      // per cell -- fetching ahead along a run only adds lines)
#endif

// Deferred finish, one read per lane.  Lane p takes pending read p (slot = sweep * NQ + position in the sweep):
//   1. its tail states at the last row, in index order, each the first maximum over its in-edges in the reference's order
//      (strict '>': the serial loop IS that order); values and the winners' `loc`s go to the lane's 16 scratch slots;
//   2. the path: the tail chain, then cell by cell back through the trellis along the back-pointer masks of its sweep's slab
//      -- M_c(t) <- (t-1, c-1), I_c(t) <- (t-1, c), b_c(t) <- (t, c-1), a fan-in sink <- the winner the sweep recorded --,
//      then the row-0 silent chain.  A path is made of runs (match diagonals, insert columns, delete rows), so a lane fetches
//      the masks of the next ROWS_WALK_K cells ALONG ITS CURRENT RUN in one round trip and uses as many as the pointers
//      confirm;
//   3. every visited state adds to the lane's summary counters through the per-column class words (ColCls: no dependent
//      lookup by state index), with the running base-pair count of hmm_utils.py:171 read off the row: an emitting state visited
//      in row t is the t-th emitting state of the path.  (Models whose class words do not mark exactly the emitting states as
//      emitting never come here: engine.hip routes them to the anti-diagonal kernel.)
template <int R, int G>
__device__ __forceinline__ void rows_finish_lanes(const ColArgs &g, const uint32_t flags, const int n_slots,
                                                  const unsigned *__restrict__ bp_wave, const double *__restrict__ rown_wave,
                                                  int32_t *__restrict__ aux_wave, int32_t *__restrict__ rev_wave, const int lane)
{
    constexpr int W = 64 / G, WORDS = (R + 4) / 5, K = ROWS_WALK_K;
    const bool want_summary = g.a.out_summary && !(flags & 4u), want_path = g.a.out_path && (flags & 1u);
    const RowsPend *pend = (const RowsPend *)(aux_wave + (g.aux_stride - ROWS_PEND_INTS - ROWS_TAILLOC_INTS));
    int32_t *tailloc = aux_wave + (g.aux_stride - ROWS_TAILLOC_INTS) + lane * COL_MAX_TAIL;
    double *tailv = (double *)rown_wave + (g.rown_stride - ROWS_PEND_READS * COL_MAX_TAIL) + lane * COL_MAX_TAIL;
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
    const int32_t *tptr = (const int32_t *)(cpb + cp->off_tail_ptr);
    const int32_t *tstate = (const int32_t *)(cpb + cp->off_tail_state);
    const TailEdge *edges = (const TailEdge *)(cpb + cp->off_tail_edge);
    const int32_t *pred0 = (const int32_t *)(cpb + cp->off_pred0);
    const ColState *state = (const ColState *)(cpb + cp->off_state);
    const uint2 *colcls = (const uint2 *)(cpb + cp->off_colcls);
    const int slab = key & 15, kq = (key >> 4) & 15, gq = (key >> 8) & 15, sinkblock = (key >> 12) & 0xfff;
    // ---- 1. tail states
    double logp = -INFINITY;
    {
        const double *row = rown_wave + (unsigned)d1.w;
        int e1 = tptr[0];
        for (int i = 0; i < n_tail; ++i) {
            const int e0 = e1;
            e1 = tptr[i + 1];
            double best = -INFINITY;
            int wloc = 0;
#pragma unroll 4
            for (int e = e0; e < e1; ++e) {
                const TailEdge ed = edges[e];
                const double v = ed.loc >= 0 ? row[(ed.loc >> 2) * 3 + (ed.loc & 3)] : tailv[-ed.loc - 1];
                const double cand = v + ed.logp;
                if (cand > best) { best = cand; wloc = ed.loc; }
            }
            tailv[i] = best;
            tailloc[i] = wloc;
            __threadfence_block();               // (this lane reads both back: later tail states, the tail chain below)
            if (i == end_tail) logp = best;
        }
    }
    g.a.out_logp[r] = logp;
    if (!want_summary && !want_path) return;
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
    if (logp != -INFINITY) {
        const int len_max = n + m;          // the reference's own path buffer (hmm.pyx:1953): a longer path is refused
        // ---- 2a. tail states (row n); the first one is the model's end state, which the summaries leave out
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
        // ---- 2b. the trellis
        int t = n, s0 = -1;
        while (!failed && t >= 1) {
            if (len > len_max) { failed = true; break; }
            // the next K cells along the run this cell belongs to: M (t - j, c - j), I (t - j, c), b (t, c - j)
            const int dt = slot != 2 ? 1 : 0, dc = slot != 0 ? 1 : 0;
            const int grp = slot == 1 ? 0 : (slot == 0 ? 1 : 2);           // masks of a cell in relaxation order: M, I, b
            unsigned aw[K], bw[K];
            uint2 cw[K];
            int lnj[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int tj = max(t - j * dt, 1), cj = max(c - j * dc, 0);
                const int tr = tj - 1, lp = tr / R, kk = tr - lp * R;
                lnj[j] = lane0 + lp;
                const unsigned *cell = bp + ((unsigned)(cj + kNC + lp) * (unsigned)(64 * WORDS) + (unsigned)(kk * 12 + grp * 4 + (lnj[j] >> 5)));
                // (written with scalar stores, which do not pass through the vector L1: agent-scope loads go around it)
                aw[j] = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bw[j] = __hip_atomic_load(cell + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cw[j] = colcls[cj + 1];          // class words of the column's I, M, b states and its flags
            }
            bool on_run = true;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (!on_run || t < 1 || c < 0 || len > len_max) break;
                if (want_path) rev[len] = slot == 1 ? state[c + 1].sM : (slot == 0 ? state[c + 1].sI : state[c + 1].sB);
                ++len;
                if (want_summary) visit(slot == 1 ? (cw[j].x >> 16) : (slot == 0 ? (cw[j].x & 0xffffu) : (cw[j].y & 0xffffu)), t);
                const unsigned fl = cw[j].y >> 16;
                if (slot == 2 && (fl & COL_FLAG_SINK)) {
                    // a fan-in sink: its predecessor is the winner the sweep recorded for this row (a feeder's b cell), not a pointer
                    c = sinks[((fl >> 4) & 15) * g.sink_stride + t] - kNC;
                    on_run = false;
                    break;
                }
                const unsigned a = (aw[j] >> (lnj[j] & 31)) & 1u, b = (bw[j] >> (lnj[j] & 31)) & 1u;   // a: the 2nd candidate won, b: the last one did
                const int from = b ? 2 : (int)a;         // the predecessor's kind: 0 I, 1 M, 2 b
                if (slot == 1 && t == 1 && from == 1) { s0 = state[c + 1].sX; t = 0; on_run = false; break; }     // row 1: the entry edge
                t -= dt;
                c -= dc;
                on_run = from == slot;
                slot = from;
            }
            if (c < 0) failed = true;            // (no path leaves the tables to the left)
        }
        // ---- 2c. row 0: the silent chain back to the model's start state (left out of the summaries like the end state)
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
