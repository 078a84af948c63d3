// flank_align.h -- local alignment of a short flanking region against long reads on the GPU.
//
// The step upstream of the long-read scoring path: the reference decides whether a PacBio read spans a VNTR by
// locally aligning the locus's two 100-base flanks to it (Bio.pairwise2.align.localms(read, flank, 1, -1, -1, -1),
// /root/reference/advntr/vntr_finder.py:324-365) and uses, of the first alignment it gets back, the score and the
// `begin` coordinate.  PARITY UNPINNED: biopython is a third-party dependency that is absent here (not vendored in
// the reference, not installed in the image), so nothing below could be checked against it.  What is implemented
// is the published algorithm as the reference parameterises it -- Smith-Waterman, match +1, mismatch -1, linear gap
// -1 per base -- and, for ties, the conventions of pairwise2 as far as they are documented: among the cells holding
// the best score the LAST one in row-major order (read position, then flank position) is where the first alignment
// ends; walking back, a horizontal step (gap in the read) is preferred to a diagonal one, a diagonal one to a
// vertical one; the walk stops at the first cell whose score is not positive and `begin` is the larger of the two
// start indices.  The CPU checker (flank_align_oracle.c, test infrastructure) restates the same with a full score
// matrix and an explicit walk.
//
// Kernel: one (read, flank) pair per wavefront; lane <-> flank position (K = 2 chunks of 64 columns: flanks of up to
// 128 bases), step s <-> anti-diagonal, lane j works on read position i = s - j -- the same anti-diagonal scheme as the
// Viterbi kernel (viterbi_columns.h): "up" is the lane's own previous value, "left" and "diagonal" are the neighbouring
// lane's values of the previous two steps (one DPP shift per step).  Instead of back-pointers every cell carries the
// start coordinates of the path the walk-back would take (i0 << 8 | j0), so no matrix is stored and no traceback runs:
// int32 scores, 1 byte per base streamed once (HBM-read bound by design; in practice VALU-bound like the Viterbi sweep).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FA_WAVES 4
#define FA_K 2                      // 64-column chunks per flank

struct FaArgs {
    const uint8_t *bases;           // reads: codes 0..3, anything else matches nothing
    const int64_t *read_off;
    const uint8_t *flank_bases;
    const int32_t *flank_off;
    const int32_t *pair_read, *pair_flank;   // pair_read in [n_reads, 2 n_reads): the reverse complement of read pair_read - n_reads
    int32_t n_pairs, n_reads;
    int32_t *out_score, *out_begin, *out_end;
    const int32_t *order;           // pairs by read length, longest first: the wavefronts take them from a counter (`next`),
    int32_t *next;                  // so that the launch ends evenly whatever the mix of lengths
};

__device__ __forceinline__ int fa_shr1(int v, int fill)            // lane i <- v[i-1]; lane 0 <- fill
{
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int fa_shr1_from(int v, int prev_chunk)  // lane 0 <- prev_chunk[63]
{
    const int r = __builtin_amdgcn_mov_dpp(prev_chunk, 0x13C, 0xf, 0xf, false);
    return __builtin_amdgcn_update_dpp(r, v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ int fa_rol1(int v)                       // lane i <- v[i+1], lane 63 <- v[0]
{
    return __builtin_amdgcn_mov_dpp(v, 0x134, 0xf, 0xf, false);
}

// 64 bases of a read as the sweep wants them: lane l <- read[at + l], symbols outside ACGT (N = 4 from this repo's hosts,
// 254 / 255 from advntr_encode_ascii) and positions past the read's end -> 4, which no flank symbol equals (flank N is 5,
// padding 255)
// rev: the read's reverse complement, made here from the uploaded read (check_if_pacbio_read_spans_vntr tests both strands of
// every read, vntr_finder.py:367-371: no second copy of the reads is made anywhere)
__device__ __forceinline__ int fa_window(const uint8_t *__restrict__ read, const int n, const int at, const int lane, const bool rev)
{
    const int i = at + lane;
    const int c = i < n ? (int)read[rev ? n - 1 - i : i] : 4;
    return c > 3 ? 4 : (rev ? 3 - c : c);
}

// One (read, flank) pair per wavefront.  The READ BASES travel through the lanes like the scores do: lane j works on read
// position i = s - j at step s, so the base it needs is the one its left neighbour used a step earlier -- one DPP shift per
// chunk and step -- and the base that enters at lane 0 comes out of a 64-base window register (lane l <- read[64 w + l],
// loaded once per 64 steps, one window ahead, rotated by one lane per step).  No per-lane load, no address arithmetic and no
// clamping inside the sweep (the first version fetched read[clamp(i + 2)] per lane, chunk and step: the sweep then ran at six
// cycles per vector instruction, waiting for those loads).  Rows before the read need no masking (every input of such a cell is
// zero and its base matches nothing); rows past the read's end occur only in the last lf - 1 steps, which run in a loop of
// their own; columns past the flank's end are masked with a loop-invariant lane mask.  The running best is ONE register per
// chunk, score << 23 | (read position + 1): a later row wins a tie by being the larger number.
__global__ void __launch_bounds__(FA_WAVES * 64) flank_align_kernel(FaArgs a)
{
    const int lane = threadIdx.x & 63;
    for (;;) {
        int q = 0;
        if (lane == 0) q = atomicAdd(a.next, 1);
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= a.n_pairs) break;
        const int p = a.order[q];
        const int r_in = a.pair_read[p], f = a.pair_flank[p];
        const bool rev = r_in >= a.n_reads;
        const int r = rev ? r_in - a.n_reads : r_in;
        const uint8_t *read = a.bases + a.read_off[r];
        const int n = __builtin_amdgcn_readfirstlane((int)(a.read_off[r + 1] - a.read_off[r]));
        const uint8_t *flank = a.flank_bases + a.flank_off[f];
        const int lf = __builtin_amdgcn_readfirstlane(a.flank_off[f + 1] - a.flank_off[f]);
        int b[FA_K], H[FA_K], S[FA_K], dH[FA_K], dS[FA_K], bestpk[FA_K], bS[FA_K], ai[FA_K], ij[FA_K];
        bool colok[FA_K];
#pragma unroll
        for (int k = 0; k < FA_K; ++k) {
            const int j = 64 * k + lane;
            b[k] = j < lf ? (int)flank[j] : 255;
            if (b[k] == 4) b[k] = 5;                                   // a non-ACGT flank symbol matches nothing, not even a read's N
            colok[k] = j < lf;
            H[k] = S[k] = dH[k] = dS[k] = bS[k] = 0;
            bestpk[k] = 0x007fffff;                                    // (score 0 never beats it; reads stay far below 2^23 bases)
            ai[k] = 4;                                                 // rows before the read
            ij[k] = ((0 - j) << 8) | j;                                // (i << 8) | j of step 0; + 256 per step
        }
        int ip1 = 1 - lane;                                            // i + 1 of chunk 0 at step 0 (chunk k: - 64 k)
        int win = fa_window(read, n, 0, lane, rev), winn = fa_window(read, n, 64, lane, rev);
        auto step = [&](const int s, const bool tail) {
            if ((s & 63) == 0 && s > 0) { win = winn; winn = fa_window(read, n, s + 64, lane, rev); }
            // the bases move on by one lane; lane 0 of chunk 0 takes read[s] from the window
            const int a_hi = FA_K > 1 ? fa_shr1_from(ai[FA_K - 1], ai[0]) : 0;
            ai[0] = fa_shr1(ai[0], win);
            if (FA_K > 1) ai[1] = a_hi;
            win = fa_rol1(win);
            int nH[FA_K], nS[FA_K];
#pragma unroll
            for (int k = FA_K - 1; k >= 0; --k) {
                // left neighbour's values of the previous step = (i, j-1); what was shifted in one step earlier = (i-1, j-1)
                const int lH = k == 0 ? fa_shr1(H[0], 0) : fa_shr1_from(H[k], H[k - 1]);
                const int lS = k == 0 ? fa_shr1(S[0], 0) : fa_shr1_from(S[k], S[k - 1]);
                const int m = ai[k] == b[k] ? 1 : -1;
                const int d = dH[k] + m, u = H[k] - 1, l = lH - 1;
                int h = max(max(d, u), max(l, 0));
                // the start the walk-back reaches: horizontal first, then diagonal (a diagonal step out of a cell whose
                // score is not positive begins the alignment here), then vertical.  (Starts of cells with score 0 are never
                // read: whoever takes one has seen a positive score there.)
                const int st_d = dH[k] > 0 ? dS[k] : ij[k];
                const int st_du = d == h ? st_d : S[k];
                const int st = l == h ? lS : st_du;
                bool ok = colok[k];
                if (tail) ok = ok && (ip1 - 64 * k) <= n;              // rows past the read's end
                h = ok ? h : 0;
                // score << 23 | (row + 1): later rows win ties; a row before the read has a negative row number and a score
                // of zero -- a negative key, which never wins the signed comparison
                const int pk = (h << 23) | (ip1 - 64 * k);
                const bool upd = pk > bestpk[k];
                bestpk[k] = upd ? pk : bestpk[k];
                bS[k] = upd ? st : bS[k];
                dH[k] = lH;
                dS[k] = lS;
                nH[k] = h;
                nS[k] = st;
                ij[k] += 256;
            }
#pragma unroll
            for (int k = 0; k < FA_K; ++k) { H[k] = nH[k]; S[k] = nS[k]; }
            ++ip1;
        };
        const int s_end = n + lf - 2;                                  // last step with an active cell
        int s = 0;
        for (; s + 1 < n; s += 2) { step(s, false); step(s + 1, false); }      // every lane's row lies inside the read (or before it)
        for (; s <= s_end; ++s) step(s, true);
        // best cell of the wave: score, then read position, then flank position -- the last best cell in row-major order.
        // Position and column travel as ONE key ((read position + 1) << 8 | flank position)
        int sc = 0, key = 0, st = 0;
#pragma unroll
        for (int k = 0; k < FA_K; ++k) {
            const int best = bestpk[k] >> 23, bi1 = best > 0 ? (bestpk[k] & 0x7fffff) : 0;
            const int kk = (bi1 << 8) | (64 * k + lane);
            if (best > sc || (best == sc && best > 0 && kk > key)) { sc = best; key = kk; st = bS[k]; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int osc = __shfl_xor(sc, o, 64), okey = __shfl_xor(key, o, 64), ost = __shfl_xor(st, o, 64);
            if (osc > sc || (osc == sc && osc > 0 && okey > key)) { sc = osc; key = okey; st = ost; }
        }
        const int ei = sc > 0 ? (key >> 8) - 1 : -1;
        if (lane == 0) {
            a.out_score[p] = sc;
            a.out_begin[p] = sc > 0 ? max(st >> 8, st & 0xff) : -1;
            a.out_end[p] = ei;
        }
    }
}
