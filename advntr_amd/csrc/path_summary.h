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

// Where a traceback leaves the path it walks (reversed: entry 0 = the model's end state): plain device memory, or -- the
// row-blocked short-read kernel -- the wavefront's own piece of LDS for the first REV_LDS_ENTRIES states (16 bits each: engine.hip
// sends a model there only when its state indices fit) with device memory behind it.  A store to device memory counts in the same
// counter as the traceback's back-pointer gathers, so every round of the walk also waited for the previous round's stores to
// reach memory (a build without the stores: REF150 launch 8.62 -> 8.38 ms, S300 2.64 -> 2.54 ms); an LDS store does not, and the
// summary pass that follows reads the path back from LDS as well.
#ifndef REV_LDS_ENTRIES
#define REV_LDS_ENTRIES 512       // (a build with 64 runs the parity suite through the spill-over into device memory)
#endif
struct RevLds {
    __attribute__((address_space(3))) unsigned short *lds;
    int32_t *mem;
};
__device__ __forceinline__ void rev_put(int32_t *rev, const int pos, const int val) { rev[pos] = val; }
__device__ __forceinline__ int rev_get(const int32_t *rev, const int pos) { return rev[pos]; }
__device__ __forceinline__ void rev_put(const RevLds &rev, const int pos, const int val)
{
    if (pos < REV_LDS_ENTRIES) rev.lds[pos] = (unsigned short)val;
    else rev.mem[pos] = val;
}
__device__ __forceinline__ int rev_get(const RevLds &rev, const int pos)
{
    return pos < REV_LDS_ENTRIES ? (int)rev.lds[pos] : rev.mem[pos];
}

// rev[0..len) holds the path reversed (rev[0] = model end ... rev[len-1] = model start).
// Forward position i (0..len-1) is rev[len-1-i]; the reference drops positions 0 and len-1.
template <class Rev>
__device__ inline void summarize_path(const Rev &rev, int len, const uint16_t *__restrict__ sclass,
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
        const int st = valid ? rev_get(rev, len - 1 - i) : 0;
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
