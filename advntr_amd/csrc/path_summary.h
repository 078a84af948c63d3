// path_summary.h -- wave-cooperative Viterbi-path summaries on device.
//
// What the reference derives from state NAMES along vpath[1:-1] (advntr/hmm_utils.py:155-286) is
// computed here from a per-state class word (ADVNTR_SC_* bits, include/advntr_hip.h) in one forward
// sweep over the path, 64 path entries per step, with ballot/popcount prefix sums for the running
// base-pair counter.  Output: the 8 int32 of ADVNTR_SUM_*.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SC_EMIT 0x0001
#define SC_MATCH 0x0002
#define SC_SUFFIX 0x0004
#define SC_PREFIX 0x0008
#define SC_UNIT_START 0x0010
#define SC_UNIT_END 0x0020
#define SC_SKIP 0x0040
#define SC_FIX 0x0080
#define SC_BASE_SHIFT 8
#define SC_BASE_VALID 0x0400

// Sum over the 64 lanes, the same value returned to all of them.  DPP adds (row_shr 1, 2, 4, 8 inside the rows of 16 lanes,
// then row_bcast 15 / 31 across rows): the finish phase is one dependent instruction stream, and six rounds of
// ds_bpermute + wait per sum were most of the summary's time.
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);       // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);       // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);       // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);       // row_shr:8  -> lane 15 of a row holds the row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);       // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);       // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __builtin_amdgcn_readlane(v, 63);
}

// rev[0..len) holds the path reversed (rev[0] = model end ... rev[len-1] = model start).
// Forward position i (0..len-1) is rev[len-1-i]; the reference drops positions 0 and len-1.
__device__ inline void summarize_path(const int32_t *rev, int len, const uint16_t *__restrict__ sclass,
                                      const uint8_t *__restrict__ seq, int n, int32_t *out, int lane)
{
    int cb = 0;                       // emitting states seen so far (reference: current_bp / seq_index)
    int starts = 0, ends = 0;
    int first_start = -1, last_start = -1, first_end = -1, last_end = -1;
    int matches = 0, rep_bp = 0, left_bp = 0, right_bp = 0, lm = 0, rm = 0;
    const unsigned long long le_mask = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);

    for (int b = 1; b < len - 1; b += 64) {
        const int i = b + lane;
        const bool valid = i < len - 1;
        const int st = valid ? rev[len - 1 - i] : 0;
        const unsigned cls = (valid && sclass) ? sclass[st] : 0u;
        const bool emit = (cls & SC_EMIT) != 0;
        const unsigned long long bal = __ballot(emit);
        const int cur_bp = cb + __popcll(bal & le_mask);          // hmm_utils.py:171-172
        const int seq_idx = cur_bp - (emit ? 1 : 0);              // hmm_utils.py:250-251

        matches += (cls & SC_MATCH) ? 1 : 0;                       // :191-197
        rep_bp += (emit && !(cls & SC_FIX)) ? 1 : 0;               // :200-206
        left_bp += (emit && (cls & SC_SUFFIX)) ? 1 : 0;            // :271-277, :246-247
        right_bp += (emit && (cls & SC_PREFIX)) ? 1 : 0;           // :280-286, :238-239

        const bool is_start = (cls & SC_UNIT_START) && (n - cur_bp >= 3);   // :173
        const bool is_end = (cls & SC_UNIT_END) && (cur_bp >= 3);           // :178
        const unsigned long long bs = __ballot(is_start), be = __ballot(is_end);
        if (bs) {
            const int fl = __ffsll((long long)bs) - 1, ll = 63 - __clzll((long long)bs);
            const int vf = __shfl(cur_bp, fl, 64), vl = __shfl(cur_bp, ll, 64);
            if (first_start < 0) first_start = vf;
            last_start = vl;
            starts += __popcll(bs);
        }
        if (be) {
            const int fl = __ffsll((long long)be) - 1, ll = 63 - __clzll((long long)be);
            const int vf = __shfl(cur_bp, fl, 64), vl = __shfl(cur_bp, ll, 64);
            if (first_end < 0) first_end = vf;
            last_end = vl;
            ends += __popcll(be);
        }
        // :227-249 -- states whose name holds 'start'/'end' are skipped; M states of a flank block are
        // compared with the flank base they were built from.
        if (valid && !(cls & SC_SKIP) && (cls & SC_MATCH) && (cls & SC_BASE_VALID) && seq_idx < n) {
            const bool hit = seq[seq_idx] == ((cls >> SC_BASE_SHIFT) & 3u);
            if (hit && (cls & SC_PREFIX)) rm += 1;
            if (hit && (cls & SC_SUFFIX)) lm += 1;
        }
        cb += __popcll(bal);
    }
    matches = wave_sum_i32(matches);
    rep_bp = wave_sum_i32(rep_bp);
    left_bp = wave_sum_i32(left_bp);
    right_bp = wave_sum_i32(right_bp);
    lm = wave_sum_i32(lm);
    rm = wave_sum_i32(rm);
    int delta = 0;                                                 // :183-186
    if (first_start >= 0 && first_end >= 0 && first_end < first_start && last_start > last_end) delta = 1;
    if (lane == 0) {
        out[0] = (starts > ends ? starts : ends) + delta;
        out[1] = matches;
        out[2] = rep_bp;
        out[3] = left_bp;
        out[4] = right_bp;
        out[5] = lm;
        out[6] = rm;
        out[7] = len;
    }
}

// The same summary, gathered WHILE the wave-cooperative traceback walks the path (viterbi_columns.h: col_traceback_walk)
// instead of in a second pass over the reversed path -- for the kernels whose walk knows the trellis row of every state it
// visits (the row-blocked kernels).  An emitting state visited in row t is the t-th emitting state of the path, so the
// running base-pair count of hmm_utils.py:171 is the row itself; that holds when the class words mark exactly the emitting
// states as emitting (what advntr's names do; engine.hip keeps other models away from these kernels).  With it the walk
// needs no reversed-path buffer unless the caller asked for paths: the buffer's stores sat in the same counter as the
// back-pointer gathers, so every round of the walk also waited for the previous round's stores to reach memory
// (REF150 launch 8.62 -> 8.38 ms, S300 2.64 -> 2.54 ms from leaving the stores out alone).
//
// A round of the walk hands over up to 64 visited states (lane i: state, base pairs emitted up to and including it); their
// class words and read bases are loaded at once and accumulated one round later, behind the next round's back-pointer
// gather, so that the loads never stand between two gathers.  The walk runs against the path: of the repeat-unit starts /
// ends seen, the FIRST is the path's last.
template <bool WIDE>          // WIDE: reads of any length (six 32-bit counters per lane); otherwise the counters travel in pairs
struct PathSummaryAcc {
    const uint16_t *__restrict__ sclass;
    const uint8_t *__restrict__ seq;
    int n;
    // per lane: matches | repeat bp << 16, left bp | right bp << 16, left | right flank matches << 16 (no total exceeds the
    // path length, which the reference bounds by n + m: engine.hip sends a model here only when that stays below 65 536);
    // WIDE: the high halves in registers of their own
    unsigned acc_mr = 0u, acc_lr = 0u, acc_fm = 0u, acc_r = 0u, acc_rt = 0u, acc_rm = 0u;
    int starts = 0, ends = 0, first_start = -1, last_start = -1, first_end = -1, last_end = -1;       // wave-uniform
    // handed over, not yet accumulated
    unsigned p_cls = 0u, p_base = 0xffu;
    int p_bp = 0;
    bool p_valid = false;

    __device__ __forceinline__ PathSummaryAcc(const uint16_t *sc, const uint8_t *sq, const int len) : sclass(sc), seq(sq), n(len) {}

    // lane-wise: this lane visited state `st` with `cur_bp` base pairs emitted up to and including it (valid: whether it visited
    // one at all; emitting: whether the state emits -- a match / insert cell -- which says where the base it is compared with
    // sits: seq_index = base pairs before the state, hmm_utils.py:250-251).  The previous visit must have been flushed.
    __device__ __forceinline__ void issue(const bool valid, const int st, const int cur_bp, const bool emitting)
    {
        p_valid = valid && sclass != nullptr;
        p_bp = cur_bp;
        p_cls = p_valid ? (unsigned)sclass[st] : 0u;
        const int seq_idx = cur_bp - (emitting ? 1 : 0);
        p_base = (p_valid && seq_idx >= 0 && seq_idx < n) ? (unsigned)seq[seq_idx] : 0xffu;
    }
    __device__ __forceinline__ void visit(const bool valid, const int st, const int cur_bp, const bool emitting)
    {
        flush();
        issue(valid, st, cur_bp, emitting);
    }
    // `ordered_behind`: a value the previous visit's accumulation must not be moved in front of (the back-pointer bits the
    // round's gather returned: the class words were requested before that gather and have arrived with it)
    __device__ __forceinline__ void flush(const int ordered_behind = 0)
    {
        asm volatile("" : "+v"(p_cls), "+v"(p_base) : "v"(ordered_behind));
        // class bits (include/advntr_hip.h): 0 EMIT, 1 MATCH, 2 SUFFIX, 3 PREFIX, 4 UNIT_START, 5 UNIT_END, 6 SKIP, 7 FIX,
        // 8-9 flank base, 10 BASE_VALID.  Everything below is a few bit operations on the class word (0 for a lane that
        // visited nothing): the counters of a pair sit 16 bits apart
        const unsigned cls = p_valid ? p_cls : 0u;
        const unsigned emit = cls & 1u;
        const unsigned rep = emit & ~(cls >> 7);                                  // emitting and not in a flank   :200-206
        const unsigned left = emit & (cls >> 2), right = emit & (cls >> 3);      // :271-286
        // :227-249 -- states whose name holds 'start' / 'end' are skipped; an M state of a flank block is compared with the
        // flank base it was built from
        const unsigned cmp = (cls >> 1) & (cls >> 10) & ~(cls >> 6) & 1u;
        const unsigned hit = (p_base == ((cls >> SC_BASE_SHIFT) & 3u)) ? cmp : 0u;
        const unsigned lhit = hit & (cls >> 2), rhit = hit & (cls >> 3);
        if (WIDE) {
            acc_mr += (cls >> 1) & 1u; acc_r += rep; acc_lr += left; acc_rt += right; acc_fm += lhit; acc_rm += rhit;
        } else {
            acc_mr += ((cls >> 1) & 1u) | rep << 16;
            acc_lr += left | right << 16;
            acc_fm += lhit | rhit << 16;
        }
        const unsigned long long bs = __ballot((cls & SC_UNIT_START) && (n - p_bp >= 3));     // :173
        const unsigned long long be = __ballot((cls & SC_UNIT_END) && (p_bp >= 3));           // :178
        if (bs) {
            const int near = __builtin_amdgcn_readlane(p_bp, __ffsll((long long)bs) - 1);     // lowest lane = latest on the path
            const int far = __builtin_amdgcn_readlane(p_bp, 63 - __clzll((long long)bs));
            if (last_start < 0) last_start = near;
            first_start = far;
            starts += __popcll(bs);
        }
        if (be) {
            const int near = __builtin_amdgcn_readlane(p_bp, __ffsll((long long)be) - 1);
            const int far = __builtin_amdgcn_readlane(p_bp, 63 - __clzll((long long)be));
            if (last_end < 0) last_end = near;
            first_end = far;
            ends += __popcll(be);
        }
        p_valid = false;
        p_cls = 0u;
    }
    // the 8-int record of a path of `len` states
    __device__ __forceinline__ void write(int32_t *out, const int len, const int lane)
    {
        flush();
        const unsigned mr = (unsigned)wave_sum_i32((int)acc_mr), lr = (unsigned)wave_sum_i32((int)acc_lr), fm = (unsigned)wave_sum_i32((int)acc_fm);
        int delta = 0;                                                 // :183-186
        if (first_start >= 0 && first_end >= 0 && first_end < first_start && last_start > last_end) delta = 1;
        if (WIDE) {
            const int rep = wave_sum_i32((int)acc_r), right = wave_sum_i32((int)acc_rt), rmatch = wave_sum_i32((int)acc_rm);
            if (lane == 0) {
                out[0] = (starts > ends ? starts : ends) + delta;
                out[1] = (int)mr; out[2] = rep; out[3] = (int)lr; out[4] = right; out[5] = (int)fm; out[6] = rmatch; out[7] = len;
            }
        } else if (lane == 0) {
            out[0] = (starts > ends ? starts : ends) + delta;
            out[1] = (int)(mr & 0xffffu);
            out[2] = (int)(mr >> 16);
            out[3] = (int)(lr & 0xffffu);
            out[4] = (int)(lr >> 16);
            out[5] = (int)(fm & 0xffffu);
            out[6] = (int)(fm >> 16);
            out[7] = len;
        }
    }
};
// (the walks of the other kernels keep the reversed path and summarise it afterwards)
struct NoPathSummaryAcc {
    __device__ __forceinline__ void visit(bool, int, int, bool) {}
    __device__ __forceinline__ void issue(bool, int, int, bool) {}
    __device__ __forceinline__ void flush(int = 0) {}
};
