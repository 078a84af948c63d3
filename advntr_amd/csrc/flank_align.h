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
    const int32_t *pair_read, *pair_flank;
    int32_t n_pairs;
    int32_t *out_score, *out_begin, *out_end;
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

// read codes above 3 -> 4 (in place, before the alignment kernel)
__global__ void __launch_bounds__(256) fa_clamp_codes_kernel(uint8_t *bases, const int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4; i < n; i += (int64_t)gridDim.x * 1024)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (i + q < n && bases[i + q] > 3) bases[i + q] = 4;
}

__global__ void __launch_bounds__(FA_WAVES * 64) flank_align_kernel(FaArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int p = blockIdx.x * FA_WAVES + wave; p < a.n_pairs; p += gridDim.x * FA_WAVES) {
        const int r = a.pair_read[p], f = a.pair_flank[p];
        const uint8_t *read = a.bases + a.read_off[r];
        const int n = __builtin_amdgcn_readfirstlane((int)(a.read_off[r + 1] - a.read_off[r]));
        const uint8_t *flank = a.flank_bases + a.flank_off[f];
        const int lf = __builtin_amdgcn_readfirstlane(a.flank_off[f + 1] - a.flank_off[f]);
        int b[FA_K], H[FA_K], S[FA_K], dH[FA_K], dS[FA_K], best[FA_K], bi[FA_K], bS[FA_K], a0[FA_K], a1[FA_K];
#pragma unroll
        for (int k = 0; k < FA_K; ++k) {
            const int j = 64 * k + lane;
            b[k] = j < lf ? (int)flank[j] : 255;
            if (b[k] == 4) b[k] = 5;                                   // a non-ACGT flank symbol matches nothing, not even a read's N
            H[k] = S[k] = dH[k] = dS[k] = best[k] = bS[k] = 0;
            bi[k] = -1;
            // read bases of steps 0 and 1 (row i = s - j; rows outside the read are masked below)
            a0[k] = n > 0 ? (int)read[min(max(0 - j, 0), n - 1)] : 4;
            a1[k] = n > 0 ? (int)read[min(max(1 - j, 0), n - 1)] : 4;
        }
        const int s_end = n + lf - 2;                                  // last step with an active cell
        for (int s = 0; s <= s_end; ++s) {
            int nH[FA_K], nS[FA_K];
#pragma unroll
            for (int k = FA_K - 1; k >= 0; --k) {
                const int j = 64 * k + lane, i = s - j;
                // prefetch the base of step s + 2
                const int a2 = n > 0 ? (int)read[min(max(i + 2, 0), n - 1)] : 4;
                const int ai = a0[k];
                a0[k] = a1[k];
                a1[k] = a2;
                // left neighbour's values of the previous step = (i, j-1); what was shifted in one step earlier = (i-1, j-1)
                const int lH = k == 0 ? fa_shr1(H[0], 0) : fa_shr1_from(H[k], H[k - 1]);
                const int lS = k == 0 ? fa_shr1(S[0], 0) : fa_shr1_from(S[k], S[k - 1]);
                const bool active = i >= 0 && i < n && j < lf;
                const int m = ai == b[k] ? 1 : -1;           // codes > 3 never compare equal (host: read N = 4, flank N = 5, pad = 255)
                const int d = dH[k] + m, u = H[k] - 1, l = lH - 1;
                int h = max(max(d, u), max(l, 0));
                // the start the walk-back reaches: horizontal first, then diagonal (a diagonal step out of a cell whose
                // score is not positive begins the alignment here), then vertical
                int st = l == h ? lS : (d == h ? (dH[k] > 0 ? dS[k] : ((i << 8) | j)) : S[k]);
                if (!active || h <= 0) { h = 0; st = 0; }
                const bool upd = h > 0 && h >= best[k];                              // later rows win ties
                best[k] = upd ? h : best[k];
                bi[k] = upd ? i : bi[k];
                bS[k] = upd ? st : bS[k];
                dH[k] = lH;
                dS[k] = lS;
                nH[k] = h;
                nS[k] = st;
            }
#pragma unroll
            for (int k = 0; k < FA_K; ++k) { H[k] = nH[k]; S[k] = nS[k]; }
        }
        // best cell of the wave: score, then read position, then flank position -- the last best cell in row-major order.
        // Position and column travel as ONE key ((read position + 1) << 8 | flank position: reads of up to 8 M bases): with
        // the two as separate variables the compiler (ROCm 7.2) dropped the update of the read position in the per-lane
        // step when the second chunk won, and a flank longer than 64 bases whose best cells tie across the chunks got the
        // first chunk's cell (found by scripts/fuzz_flank_align.py)
        int sc = 0, key = 0, st = 0;
#pragma unroll
        for (int k = 0; k < FA_K; ++k) {
            const int kk = ((bi[k] + 1) << 8) | (64 * k + lane);
            if (best[k] > sc || (best[k] == sc && best[k] > 0 && kk > key)) { sc = best[k]; key = kk; st = bS[k]; }
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
