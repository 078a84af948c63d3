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

// ---- two passes per (read, flank) pair ------------------------------------------------------------------------------
// Pass 1 finds the best local score and the LAST best cell in row-major order with a sweep that carries nothing but scores:
// a flank has at most 128 bases, so a score fits 16 bits and the two 64-column chunks of a lane travel in the two halves of
// ONE register (v_pk_add_i16 / v_pk_max_i16 / v_pk_min_u16 / v_pk_mad_i16): 19 vector instructions per step of 128 cells where
// the sweep that also carried the start coordinates took 56.
//   * the read bases move through the lanes as before (one DPP shift per step, lane 0 fed from a 64-base window register),
//     both chunks in one register: the base that leaves lane 63 of the low half enters lane 0 of the high half (the value
//     shifted left by 16 and rotated by one lane is the `old` operand of the shift);
//   * match / mismatch of both halves without a compare: x = bases ^ flank, t = min_u16(x, 1), m = 1 - 2 t;
//   * the running best is one 32-bit key per chunk, score << 24 | (read position + 1), built with one v_perm_b32 and kept with
//     one v_max_u32: a later row wins a tie by being the larger number;
//   * nothing is masked: columns past the flank's end and rows past the read's end only ever see mismatches, so a cell there
//     stays below the best real cell it descends from, and rows before the read are all-zero; the lanes past the flank are left
//     out of the final reduction.
// Pass 2 recovers `begin` for that cell: the walk-back of a local alignment that ends at (ei, ej) with score sc stays within
// 2 (ej + 1) rows of ei (every prefix of the path has a positive score, so mismatches + gaps < matches <= ej + 1), and on the
// cells of that path a sweep that starts from zeros at the top of this window computes the values of the full matrix -- off the
// path it can only be lower, which never turns a failed test of the walk-back's preference order (horizontal, diagonal,
// vertical) into a successful one.  So the sweep that carries the start coordinates (the one this kernel used to run over
// the whole read) runs over those rows only and is read out at (ei, ej): a few percent of a 10 kb read.
typedef short fa_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short fa_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int fa_pk_add(int a, int b) { return __builtin_bit_cast(int, (fa_s16x2)(__builtin_bit_cast(fa_s16x2, a) + __builtin_bit_cast(fa_s16x2, b))); }
__device__ __forceinline__ int fa_pk_max(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(fa_s16x2, a), __builtin_bit_cast(fa_s16x2, b))); }
// (as inline assembly: given min(x, 1) the compiler rewrites the match / mismatch arithmetic below into two 16-bit compares,
// two selects and a byte permute -- five instructions for the two this takes with the multiply-add that follows)
__device__ __forceinline__ int fa_pk_minu(int a, int b)
{
    int r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int fa_pk_mad(int a, int b, int c)          // (one v_pk_mad_i16: left to itself the compiler emits a shift and a subtraction)
{
    int r;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// both halves one lane on: lane i <- v[i-1]; lane 0: high half <- low half of lane 63, low half <- lane0_lo (lane 0's, high half 0)
__device__ __forceinline__ int fa_pk_shr1(int v, int lane0_lo)
{
    // (low half of the lane below) << 16 in ONE instruction: a 24-bit multiply by 65536 of the rotated register (the high half
    // leaves through the top) -- instead of a shift and a DPP move: 6.75 -> 6.1 ms per 16 000 alignments.  Hand-written DPP: the
    // two wait states between a vector write of `v` and its DPP read are spelled out (the compiler does not look into the asm).
    int old;
    asm volatile("s_nop 1\n\tv_mul_u32_u24_dpp %0, %1, %2 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(old) : "v"(v), "v"(65536));
    old |= lane0_lo;
    return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);         // wave_shr:1, lane 0 keeps `old`
}

__global__ void __launch_bounds__(FA_WAVES * 64) flank_align_kernel(FaArgs a)
{
    static_assert(FA_K == 2, "two chunks of 64 columns: the halves of a packed register");
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
        int b[FA_K];
#pragma unroll
        for (int k = 0; k < FA_K; ++k) {
            const int j = 64 * k + lane;
            b[k] = j < lf ? (int)flank[j] : 255;
            if (b[k] == 4) b[k] = 5;                                   // a non-ACGT flank symbol matches nothing, not even a read's N
        }
        // ---------------------------------------------------------------- pass 1: score and last best cell
        int sc = 0, key = 0;
        if (n > 0 && lf > 0) {
            const int b_pk = b[0] | (b[1] << 16);
            const int one = 0x00010001, minus1 = (int)0xffffffff, minus2 = (int)0xfffefffe;
            int ai = 0x00040004;                                       // rows before the read match nothing
            int H = 0, dH = 0;
            unsigned best0 = 0, best1 = 0;
            int row = 1 - lane;                                        // read position + 1 of chunk 0 (chunk 1: - 64)
            int win = fa_window(read, n, 0, lane, rev), winn = fa_window(read, n, 64, lane, rev);
            const int s_end = n + lf - 2;                              // last step with a cell of the matrix
            auto step = [&]() {
                ai = fa_pk_shr1(ai, win);
                win = fa_rol1(win);
                const int t = fa_pk_minu(ai ^ b_pk, one);              // 0: the bases are equal, 1: they differ
                const int m = fa_pk_mad(t, minus2, one);               // +1 / -1
                const int lH = fa_pk_shr1(H, 0);                       // (i, j-1); what it was a step earlier is (i-1, j-1)
                const int d = fa_pk_add(dH, m);
                const int g = fa_pk_add(fa_pk_max(H, lH), minus1);     // the better of the two gap moves
                const int h = fa_pk_max(fa_pk_max(d, g), 0);
                // score << 24 | (read position + 1): byte 0 of the low half / byte 2 of the high half on top of the row's three bytes
                best0 = max(best0, __builtin_amdgcn_perm((unsigned)h, (unsigned)row, 0x04020100u));
                best1 = max(best1, __builtin_amdgcn_perm((unsigned)h, (unsigned)row, 0x06020100u));
                dH = lH;
                H = h;
                ++row;
            };
            for (int s0 = 0; s0 <= s_end; s0 += 64) {
                if (s0 > 0) { win = winn; winn = fa_window(read, n, s0 + 64, lane, rev); }
                const int cnt = min(64, s_end + 1 - s0);
                if (cnt == 64) {
#pragma unroll 4
                    for (int u = 0; u < 64; ++u) step();
                } else {
                    for (int u = 0; u < cnt; ++u) step();
                }
            }
            // best cell of the wave: score, then read position, then flank position -- the last best cell in row-major order.
            // Position and column travel as ONE key ((read position + 1) << 8 | flank position)
#pragma unroll
            for (int k = 0; k < FA_K; ++k) {
                const unsigned bk = k == 0 ? best0 : best1;
                const int best = (int)(bk >> 24);
                // (sign-extended 24-bit row field: rows before the read are negative, and then the score is 0)
                const int bi1 = best > 0 ? (((int)(bk << 8)) >> 8) - 64 * k : 0;
                const int kk = (bi1 << 8) | (64 * k + lane);
                if (64 * k + lane < lf && (best > sc || (best == sc && best > 0 && kk > key))) { sc = best; key = kk; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int osc = __shfl_xor(sc, o, 64), okey = __shfl_xor(key, o, 64);
                if (osc > sc || (osc == sc && osc > 0 && okey > key)) { sc = osc; key = okey; }
            }
        }
        sc = __builtin_amdgcn_readfirstlane(sc);
        key = __builtin_amdgcn_readfirstlane(key);
        const int ei = sc > 0 ? (key >> 8) - 1 : -1, ej = key & 0xff;
        // ---------------------------------------------------------------- pass 2: the start of the walk-back from (ei, ej)
        int begin = -1;
        if (sc > 0) {
            const int r0 = max(0, ei - 2 * (ej + 1) - 1);              // first row of the window
            int H[FA_K], S[FA_K], dH[FA_K], dS[FA_K], ai[FA_K], ij[FA_K];
#pragma unroll
            for (int k = 0; k < FA_K; ++k) {
                const int j = 64 * k + lane;
                H[k] = S[k] = dH[k] = dS[k] = 0;
                ai[k] = 4;
                ij[k] = ((r0 - j) << 8) | j;                           // (i << 8) | j of step 0; + 256 per step
            }
            int win = fa_window(read, n, r0, lane, rev), winn = fa_window(read, n, r0 + 64, lane, rev);
            const int s_cap = (ei - r0) + ej;                          // the step on which lane ej works on row ei
            for (int s = 0; s <= s_cap; ++s) {
                if ((s & 63) == 0 && s > 0) { win = winn; winn = fa_window(read, n, r0 + s + 64, lane, rev); }
                const int a_hi = fa_shr1_from(ai[1], ai[0]);
                ai[0] = fa_shr1(ai[0], win);
                ai[1] = a_hi;
                win = fa_rol1(win);
                int nH[FA_K], nS[FA_K];
#pragma unroll
                for (int k = FA_K - 1; k >= 0; --k) {
                    // left neighbour's values of the previous step = (i, j-1); what was shifted in one step earlier = (i-1, j-1)
                    const int lH = k == 0 ? fa_shr1(H[0], 0) : fa_shr1_from(H[k], H[k - 1]);
                    const int lS = k == 0 ? fa_shr1(S[0], 0) : fa_shr1_from(S[k], S[k - 1]);
                    const int m = ai[k] == b[k] ? 1 : -1;
                    const int d = dH[k] + m, u = H[k] - 1, l = lH - 1;
                    const int h = max(max(d, u), max(l, 0));
                    // the start the walk-back reaches: horizontal first, then diagonal (a diagonal step out of a cell whose
                    // score is not positive begins the alignment here), then vertical.  (Starts of cells with score 0 are never
                    // read: whoever takes one has seen a positive score there.)
                    const int st_d = dH[k] > 0 ? dS[k] : ij[k];
                    const int st_du = d == h ? st_d : S[k];
                    const int st = l == h ? lS : st_du;
                    dH[k] = lH;
                    dS[k] = lS;
                    nH[k] = h;
                    nS[k] = st;
                    ij[k] += 256;
                }
#pragma unroll
                for (int k = 0; k < FA_K; ++k) { H[k] = nH[k]; S[k] = nS[k]; }
            }
            const int st = __builtin_amdgcn_readlane(ej >= 64 ? S[1] : S[0], ej & 63);
            begin = max(st >> 8, st & 0xff);
        }
        // (three lanes, one result each -- not `if (lane == 0)`: with the same condition at both ends of the loop body the
        // compiler threads lane 0 from this store straight into the next iteration's atomic and builds a lane-divergent loop
        // around the body, in which the other lanes take the same pair again and again)
        if (lane < 3) {
            int32_t *dst = lane == 0 ? a.out_score : (lane == 1 ? a.out_begin : a.out_end);
            dst[p] = lane == 0 ? sc : (lane == 1 ? begin : ei);
        }
    }
}
