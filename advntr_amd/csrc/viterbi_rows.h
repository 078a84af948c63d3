// viterbi_rows.h -- row-blocked variant of the column-program Viterbi sweep for short reads.
//
// Same recurrence, same arithmetic and same comparison order as viterbi_columns.h (and therefore as the reference,
// /root/reference/pomegranate/hmm.pyx:2026-2083); what changes is the mapping of the trellis onto the wavefront:
//
//   * a lane owns R CONSECUTIVE rows (t = R*l + 1 .. R*l + R) instead of one row per 64-row chunk, and at step s works
//     on column c = s - l for all of them.  The previous row of a lane's 2nd..Rth row is the lane's own register, so a
//     step needs ONE cross-lane shift (3 fp64 = 6 DPP moves) for R cells instead of one (6-12 moves) per cell, and the
//     column's info word and 88-B transition class are read from LDS once per R cells (the anti-diagonal kernel is
//     co-limited by VALU issue and LDS reads);
//   * the skew of the systolic sweep is one column per LANE, not per row: G reads share a wavefront, each in a group of
//     W = 64/G lanes, and a 150-base read costs (NC + 29) steps of 5 cells on 32 lanes: 88 % of the lane-steps do useful
//     work instead of 70 %.  G = 2: `wave_shr:1` crosses from the first group into the second; the first group's last
//     lane is kept a padding lane whose rows stay at -inf, which is the row-0 value of I and M, and b takes one select.
//     G = 4: a group is a DPP row of 16 lanes, `row_shr:1` never crosses groups and lanes 0/16/32/48 keep the `old`
//     operand = the row-0 boundary, so there is no fix-up at all;
//   * the six comparison outcomes of a cell are stored as the lane masks the comparisons produce, with scalar stores
//     (below): no vector instruction is spent on packing back-pointers.
//
// Reads of a tile go G at a time to a wavefront; after the sweep the wave finishes them one by one (tail states,
// cooperative traceback, path summary) with the code of viterbi_columns.h.
#pragma once
#include <type_traits>
#include "viterbi_columns.h"

// Three instantiations cover the short reads of a large batch (engine.hip routes by length):
//   <5, 2>: 125-155 bases, 5 rows per lane, 2 reads per wavefront (32 lanes each, the 32nd a padding lane)
//   <4, 2>:  65-124 bases, 4 rows per lane, 2 reads per wavefront
//   <4, 4>:   1-64  bases, 4 rows per lane, 4 reads per wavefront (a DPP row of 16 lanes each: `row_shr:1` never crosses
//             groups and lanes 0/16/32/48 keep the `old` operand = the row-0 boundary, so no fix-up at all)
// Measured on the bench workload (100 k reads of 150 bases, the 488-column REF150 model; kernel ms per launch; the anti-diagonal
// kernel: 16.5):  <5, 2> at 3 waves/SIMD 11.3 (no spill inside the sweep loop); at 4 waves/SIMD (128 VGPRs) 12.6:
// the loop then reloads spilled values, and on gfx9 a vector-memory load waits behind the back-pointer stores in the
// same counter;  <10, 4> needs 256 VGPRs (2 waves/SIMD) 13.2, with 3 waves it spills 22.3.
#define ROWS_CONFIGS 3
struct RowsConfig { int R, G, max_read; };
// longest read of a configuration: G = 2 keeps the group's last lane a padding lane (see rows_sweep)
static const RowsConfig rows_configs[ROWS_CONFIGS] = {{5, 2, 5 * 31}, {4, 2, 4 * 31}, {4, 4, 4 * 16}};
#ifndef ROWS_WAVES_PER_SIMD
#define ROWS_WAVES_PER_SIMD 3
#endif
#define ROWS_MAX_READ 155           // longest read any configuration takes
#define ROWS_MAX_GROUPS 4
#define ROWS_STASH_BYTES (COL_WAVES * ROWS_DEPTH * ROWS_MAX_GROUPS * 16)     // LDS behind the tables (viterbi_rows_kernel)
#define ROWS_REV_BYTES (COL_WAVES * REV_LDS_ENTRIES * 2)                    // ... and behind that the wavefronts' reversed paths (path_summary.h: RevLds)
#define ROWS_TAIL_LDS_BYTES (COL_WAVES * COL_MAX_TAIL * 12)                 // ... and the tail states' values and winners of the read being finished
// Back-to-back sweeps: a wavefront's lane groups take up to ROWS_DEPTH reads each, one behind the other ALONG THE STEP AXIS.
// When a lane has done the last column of its group's kth read it starts column 0 of read k + 1 on the next step -- the
// neighbouring lanes follow one step later each, exactly as at the start of a sweep -- so the W - 1 steps a sweep spends
// filling and draining its pipeline are paid once per ROWS_DEPTH reads instead of once per read (REF150, 488 columns: 6 %
// of the steps; the 99 columns of the metric's ~300-state shape: 23 %).  A lane's rows of the queued reads wait packed in
// one 64-bit register (3 bits per row + 3 bits for the slot of the read's last row): nothing else is carried.
#ifndef ROWS_DEPTH
#define ROWS_DEPTH 4
#endif
#define ROWS_STREAM_MIN_COLS 64      // narrower models sweep their reads one at a time (the wrap window must not lap itself)
// tile sizes (engine.hip): full depth until this share of a batch's reads is left, then half depth, then single sweeps
#ifndef ROWS_TAIL_HALF_PCT
#define ROWS_TAIL_HALF_PCT 25
#endif
#ifndef ROWS_TAIL_SINGLE_PCT
#define ROWS_TAIL_SINGLE_PCT 10
#endif
#ifndef ROWS_LONG_R
#define ROWS_LONG_R 5                // rows per lane of the tiled kernel for longer reads: row tiles of 320 rows (a read's last
                                     // tile runs with as few rows per lane as cover it: viterbi_rows_long_kernel)
#endif

template <int G>
__device__ __forceinline__ int rows_shr1(const int old, const int src)
{
    // lane i <- src[i-1] inside the read's lane group; the group's first lane keeps `old`
    if (G >= 4) return __builtin_amdgcn_update_dpp(old, src, 0x111, 0xf, 0xf, false);       // row_shr:1 (16-lane rows)
    return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false);                   // wave_shr:1
}
template <int G>
__device__ __forceinline__ double rows_shift(const double v, const double inject)
{
    const int lo = rows_shr1<G>(__double2loint(inject), __double2loint(v));
    const int hi = rows_shr1<G>(__double2hiint(inject), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rows_rol1(const double v)          // lane i <- v[i+1], lane 63 <- v[0]
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x134, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x134, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Back-pointers of the row-blocked sweeps.  The six comparison outcomes of a cell leave the wavefront as the 64-bit LANE
// MASKS the comparisons produce (v_cmp_gt_f64 into an SGPR pair), written with scalar stores: 48 bytes per cell of a step,
// R cells per step, steps 256 * ceil(R / 5) bytes apart.  Shifting the outcomes into a per-lane word instead costs one
// v_addc_co_u32 per comparison on a kernel that is bound by vector-instruction issue (SQ_INSTS_VALU per launch of the
// bench workload 5.34 G -> 4.54 G, 10.0 -> 9.2 ms on the same box); the scalar stores issue from the scalar port.  The
// traceback reads bit `lane` of the cell's six masks.
typedef unsigned long long adv_u64x2 __attribute__((ext_vector_type(2)));
// Two relaxations of one state -- if (cand > best) best = cand, strict '>' keeps the first maximum as the reference does
// (hmm.pyx:2039,2060,2080) -- and their masks to byte offset OFF of the step's slab in one 16-byte store.
__device__ __forceinline__ void relax_mask2(double &best, const double cand2, const double cand3, const unsigned *bps, const int OFF)
{
    unsigned long long m2, m3;
    asm("v_cmp_gt_f64_e64 %1, %2, %0\n\t"
        "v_max_f64 %0, %0, %2"
        : "+v"(best), "=s"(m2)
        : "v"(cand2));
    asm("v_cmp_gt_f64_e64 %1, %2, %0\n\t"
        "v_max_f64 %0, %0, %2"
        : "+v"(best), "=s"(m3)
        : "v"(cand3));
    adv_u64x2 m;
    m.x = m2; m.y = m3;
    asm volatile("s_store_dwordx4 %0, %1, %2" : : "s"(m), "s"(bps), "i"(OFF) : "memory");
}
// Scalar stores sit in the scalar data cache until written back; the traceback reads the masks with loads that bypass the
// vector L1 (rows_bp_at), which may still hold lines of the previous reads' slab.
__device__ __forceinline__ void rows_bp_publish()
{
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
// back-pointer bits of state st (0 = I, 1 = M, 2 = b) of cell (t, c), in the byte layout bp_ptr_* decode: R rows per lane,
// lane lane0 + (t - 1) / R works on column c at step c + (t - 1) / R; the two masks of a state are 16 bytes of one line
// (ln: the lane that worked on the row, k: the row's slot in it; lp: its position in its lane group)
template <int WORDS = 1>
__device__ __forceinline__ int rows_bp_at_split(const unsigned *__restrict__ bpw, const int ln, const int k, const int cc, const int st,
                                                const int lp_in = -1)
{
    const int lp = lp_in < 0 ? ln : lp_in;
    // masks of a cell in relaxation order: aM bM | aI bI | aB bB (a: the 2nd candidate won, b: the last one did)
    const int grp = st == 1 ? 0 : (st == 0 ? 1 : 2);
    const unsigned *cell = bpw + (int64_t)(cc + lp) * (64 * WORDS) + k * 12 + grp * 4;
    const int sh = ln & 31;
    // both masks of the state (16 bytes, one line) in ONE load; the masks were written with scalar stores, which do not pass
    // through the vector L1: read around it (sc1)
    uint4 w;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(cell) : "memory");
    const unsigned a = ((ln & 32 ? w.y : w.x) >> sh) & 1u;
    const unsigned b = ((ln & 32 ? w.w : w.z) >> sh) & 1u;
    return (int)((a << 1 | b) << (st == 1 ? 2 : (st == 0 ? 4 : 0)));
}

template <int R>
__device__ __forceinline__ int rows_bp_at(const unsigned *__restrict__ bpw, const int lane0, const int tt, const int cc, const int st)
{
    const int lp = (tt - 1) / R, k = (tt - 1) - lp * R;
    return rows_bp_at_split<(R + 4) / 5>(bpw, lane0 + lp, k, cc, st, lp);
}

// TILED (G = 1, reads longer than 64 R rows): the sweep covers rows row0+1 .. row0+n of a longer read; `seam` (tiles after
// the first) is the previous tile's last row, captured at seam[3 * (c + 64) + {0, 1, 2}], and takes the place of row 0.
// `queue`, `depth` (untiled sweeps): the lane's rows of the reads that follow the first one back to back (rows_queue_push),
// depth = reads per lane group in this sweep.  Read k occupies steps k * NC + lane .. k * NC + lane + NC - 1 of a lane, its
// row-n values land at rown[cap_base + 3 * (W + k * NC + c)] and its fan-in winners in the kth block of `aux`.
template <int R, int G, bool TILED = false>
__device__ __forceinline__ void rows_sweep(const LdsTables &L, const int NC, const int s_end,
                                           const uint8_t *__restrict__ seq, const int n, const int lp, const int lane,
                                           unsigned *__restrict__ bpw, double *__restrict__ rown, const unsigned cap_base,
                                           int32_t *__restrict__ aux, const unsigned sink_base, const int sink_stride,
                                           const int row0 = 0, const double *__restrict__ seam = nullptr,
                                           unsigned long long queue = 0ull, const int depth = 1)
{
    constexpr int W = 64 / G, WORDS = (R + 4) / 5;
    constexpr bool STREAM = !TILED;
    double I[R], M[R], B[R], er[R];
    unsigned esym[R];                  // LDS address of the emission-pair row of the base in the lane's kth row
#pragma unroll
    for (int k = 0; k < R; ++k) {
        I[k] = M[k] = B[k] = er[k] = -INFINITY;
        const int t = R * lp + k + 1;
        // rows past the read take the pair table's 5th symbol row, -inf: they stay at -inf in every state, so what the
        // group's last lane hands to the next group's first (G = 2: wave_shr crosses the boundary) is the row-0 value
        // of I and M already
        esym[k] = L.epair_base + (unsigned)((t <= n) ? (int)seq[t - 1] : 4) * L.epair_sym_stride;
    }
    // the lane that holds the read's last row, and the slot it sits in (7: not in this lane) -- kept as one LANE MASK per slot
    // (wave-uniform: a scalar register pair that the capture branch takes as its exec mask; the masks change only when a lane
    // group moves on to its next read, so no per-step compare is spent on them)
    int kcap = (n >= 1 && (n - 1) / R == lp) ? (n - 1) - lp * R : 7;
    unsigned long long capm[R];
#pragma unroll
    for (int k = 0; k < R; ++k) capm[k] = __builtin_amdgcn_ballot_w64(kcap == k);
    const bool first_lane = lp == 0 && (!TILED || row0 == 0);       // owner of the read's first row: entry edges
    const bool fix = G == 2 && lane == 32;
    // Row above the lane's first row (the neighbouring lane's last row at the previous step), shifted in with DPP.  A
    // step evaluates M of the lane's first row FIRST, from the values shifted in one step earlier (its diagonal
    // inputs), and only then shifts -- in place: the group-first lanes of nI / nM, which the shift never writes, hold the
    // row-0 value -inf for the whole sweep.
    double nI = -INFINITY, nM = -INFINITY, nB = -INFINITY;
    // (untiled sweeps ping-pong between two register pairs for the shifted I / M values: a step reads the pair the previous
    // step shifted into and shifts into the other one, whose group-first lanes also hold -inf for good -- no copy is needed
    // to keep the diagonal inputs alive across the in-place DPP shift)
    double qI = -INFINITY, qM = -INFINITY;
    unsigned pa = L.pinfo + (unsigned)(64 - lp) * 16u;          // padded info record of column c = -lp (64 dummies in front)
    uint2 meta = lds_uint2(pa + 8u);
    double v0b = *(LdsDouble *)(size_t)pa;
    // row-n capture of column c at rown[cap_off + 3 * c]; back-pointer words at bpw[bp_off + 64 * WORDS * s]: 32-bit
    // offsets from wave-uniform bases (scalar base + vector offset addressing, no 64-bit pointer arithmetic per step)
    // (the step index is wave-uniform: the per-step parts of these offsets are scalar arithmetic, the per-lane parts constants)
    const unsigned cap_lane = cap_base + (unsigned)(W - lp) * 3u;
    int sstep = 0;                                              // step index (scalar)
    // TILED, tiles after the first: lane l holds the seam values of column 64 * (s / 64) + ((l + s) % 64): a window of 64
    // columns, reloaded every 64 steps and rotated one lane per step (DPP wave_rol:1), so that lane 0 -- the only lane
    // whose shifted-in values come from the seam -- always holds the values of the column it works on
    const bool seamed = TILED && seam != nullptr;
    double wI = -INFINITY, wM = -INFINITY, wB = -INFINITY;
    unsigned win0 = 4u * (sink_base + (unsigned)(row0 + R * lp + 1));
    // (a register of its own: as one of eight kernel-argument words loaded together it is spilled and reloaded with them)
    unsigned sink_bytes = 4u * (unsigned)sink_stride;
    asm volatile("s_mov_b32 %0, %0" : "+s"(sink_bytes));
    // Back-to-back reads.  A lane that has done its read's last column starts column 0 of its group's next read on the next
    // step and takes its rows of that read out of the queue -- lane 0 first, the others one step later each, exactly as at the
    // start of a sweep: a WINDOW of W steps per read in which `uw`, the column lane 0 of a group is on next, says whose turn it
    // is (lane uw is on its last column).  A lane just keeps walking: what follows the last column in the padded info table
    // are copies of columns 0 .. W (stage_model<1, true, true>), so the table pointer goes on while the lanes change reads one
    // after the other; once all of them have, the pointers step back by NC records together.  Behind the LAST read the lanes
    // walk on as well, for the W - 1 steps at most until the sweep's last lane is done: they compute values nobody reads (no
    // capture; fan-in "winners" go to a spare block).  The steps between the windows carry none of this: the sweep is a
    // sequence of plain stretches (two steps per loop iteration, as ever) and windows (their own copies of the step: WIN = 1
    // between two reads, WIN = 2 behind the last one).
    auto step = [&](auto WIN, double &nI, double &nM, double &qI, double &qM, const int uw, const bool more) {
        constexpr int win_kind = decltype(WIN)::value;
        pa += 16u;
        const uint2 meta_next = lds_uint2(pa + 8u);
        const double v0b_next = *(LdsDouble *)(size_t)pa;
        // the column's transition class record, 96 bytes as six 16-byte words (ColClass)
        LdsDouble2 *T2 = (LdsDouble2 *)(size_t)(meta.x & 0xffffu);
        const adv_f64x2 t0 = T2[0], t1 = T2[1], t2 = T2[2], t3 = T2[3], t4 = T2[4], t5 = T2[5];
        const double iI = t0.x, iM = t0.y, iD = t1.x, mI = t1.y, mM = t2.x, mD = t3.x, dI = t3.y, dM = t4.x, dD = t4.y;
        const unsigned epo = meta.y;                 // this column's offset inside a symbol row of the emission pair table
        // flags of the lane's column (stage_model<1, true>): byte 3 = fan-in sink, byte 2 = feeder | fed sink's index << 1
        const bool on_sink = (meta.x >> 24) != 0u;
        const unsigned feedb = (meta.x >> 16) & 0xffu;
        const bool on_feed = feedb != 0u;
        unsigned long long sinkmask = __ballot(on_sink);                    // wave-uniform, rare
        // Fan-in (hmm.pyx order: the first maximum over the feeders, in column order): every row keeps the running
        // maximum `er`; a feeder that beats it writes its column straight into the sink's back-pointer slot of that row
        // (later winners overwrite earlier ones), so no winner register is carried.  Lanes that are not on a feeder
        // column carry weight -inf and never win.
        const double erw_c = t5.x, mX = t2.y;         // read with the rest of the record: one LDS round trip per step
        const bool anyfeed = __ballot(on_feed) != 0;      // wave-uniform: false while the wave is in a flank
        const unsigned *bps = bpw + (int64_t)sstep * (64 * WORDS);    // the step's back-pointer slab (scalar address)
        double *capq = rown + 3 * sstep + cap_lane;          // where this column's row-n values go (lane holding the last row)
        // emission log-probs {M, I} of the lane's rows: one 16-byte read per cell, all requested before the first cell so
        // that no LDS wait sits between the scalar stores of the step (they count in the same counter)
        adv_f64x2 ev[R];
#pragma unroll
        for (int k = 0; k < R; ++k) ev[k] = *(LdsDouble2 *)(size_t)(esym[k] + epo);
        {
            double dgI = nI, dgM = nM, dgB = nB;        // (t-1, c-1) of the lane's first row: shifted in at the previous step
            double upI = 0.0, upM = 0.0, upB = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double eM = ev[k].x, eI = ev[k].y;
                // M_c(t) <- [I_{c-1}, M_{c-1}, X, b_{c-1}](t-1); X exists for row 1 only and takes the M candidate's place
                // there.  Its inputs -- the previous values of the row above -- die here.
                double vM = (dgI + mI) + eM;
                double cM = dgM + mM;
                if (k == 0) cM = first_lane ? mX : cM;
                const double cM2 = cM + eM, cM3 = (dgB + mD) + eM;
                // b_c(t) <- [I_{c-1}, M_{c-1}, b_{c-1}](t): the lane's own previous values
                const double oI = I[k], oM = M[k], oB = B[k];
                double vB = oI + dI;
                const double cB2 = oM + dM, cB3 = oB + dD;
                if (k == 0) {
                    // row above the lane's first row, same column: the neighbouring lane's last row of the previous step
                    if (TILED && seamed) {
                        if ((sstep & 63) == 0) {
                            const int cw = min(sstep + lane, NC - 1) + 64;
                            wI = seam[3 * cw]; wM = seam[3 * cw + 1]; wB = seam[3 * cw + 2];
                        }
                        nI = rows_shift<G>(I[R - 1], wI);
                        nM = rows_shift<G>(M[R - 1], wM);
                        nB = rows_shift<G>(B[R - 1], wB);
                        wI = rows_rol1(wI); wM = rows_rol1(wM); wB = rows_rol1(wB);
                    } else {
                        qI = rows_shift<G>(I[R - 1], qI);
                        qM = rows_shift<G>(M[R - 1], qM);
                        nB = rows_shift<G>(B[R - 1], v0b);               // row 0 is read independent (host precomputed)
                    }
                    if (G == 2) nB = fix ? *(LdsDouble *)(size_t)(pa - 16u) : nB;    // (I and M arrive as -inf from the padding lane)
                    upI = TILED && seamed ? nI : qI; upM = TILED && seamed ? nM : qM; upB = nB;
                }
                // I_c(t) <- [I_c, M_c, b_c](t-1): the values the row above just got
                double vI = (upI + iI) + eI;
                const double cI2 = (upM + iM) + eI, cI3 = (upB + iD) + eI;
                // relaxations in the order M, I, b; the two masks of a state leave in one 16-byte scalar store
                relax_mask2(vM, cM2, cM3, bps, 48 * k);
                relax_mask2(vI, cI2, cI3, bps, 48 * k + 16);
                relax_mask2(vB, cB2, cB3, bps, 48 * k + 32);
                // a wave-uniform branch per cell, not four selects; the mask is re-tested as a scalar every time (as a bool
                // the compiler carries the condition as a lane mask and rebuilds it with two vector instructions per cell)
                asm volatile("" : "+s"(sinkmask));
                if (sinkmask != 0ull) {
                    asm volatile("; fan-in column" ::);
                    vB = on_sink ? er[k] : vB;
                    er[k] = on_sink ? -INFINITY : er[k];
                }
                I[k] = vI; M[k] = vM; B[k] = vB;
                if (__builtin_amdgcn_inverse_ballot_w64(capm[k])) { capq[0] = vI; capq[1] = vM; capq[2] = vB; }
                upI = vI; upM = vM; upB = vB;
                dgI = oI; dgM = oM; dgB = oB;
            }
        }
        if (anyfeed) {                 // after the cells: the accumulators take the rows' final b values of this column
            asm volatile("; feeder column" ::);
            const double erw = on_feed ? erw_c : -INFINITY;
            const unsigned win = win0 + (feedb >> 1) * sink_bytes;      // byte offset of the row's slot
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double cand = B[k] + erw;
                const bool won = cand > er[k];
                asm("v_max_f64 %0, %0, %1" : "+v"(er[k]) : "v"(cand));        // (fmax() adds two canonicalising self-maxima)
                if (won) *(int32_t *)((char *)aux + (win + 4u * k)) = sstep - lp;      // this lane's column
            }
        }
        if (win_kind != 0) {
            const bool mine = lp == uw;
            // fan-in winners of the next read go to the next block of `aux`; behind the last read's block lies one nobody
            // reads, for what the lanes "win" while they walk on past the end of the sweep
            win0 = mine ? win0 + (unsigned)COL_MAX_SINKS * sink_bytes : win0;
            if (more) {
                if (mine) {
                    const unsigned w = (unsigned)queue;
#pragma unroll
                    for (int k = 0; k < R; ++k) esym[k] = L.epair_base + __umul24((w >> (3 * k)) & 7u, L.epair_sym_stride);
                    kcap = (int)((w >> (3 * R)) & 7u);
                    queue >>= 3 * R + 3;
                }
#pragma unroll
                for (int k = 0; k < R; ++k) capm[k] = __builtin_amdgcn_ballot_w64(kcap == k);
                if (uw == W) pa -= 16u * (unsigned)NC;           // every lane of the group is on its next read now
            } else {
                // behind the last read: nothing is captured any more.  The rows stay what they were -- the lanes compute
                // values nobody reads on the copied columns for the few steps until the sweep's last lane is done
                const unsigned long long gone = __builtin_amdgcn_ballot_w64(mine);
#pragma unroll
                for (int k = 0; k < R; ++k) capm[k] &= ~gone;
            }
        }
        ++sstep;
        meta = meta_next;
        v0b = v0b_next;
    };
    // two steps per loop iteration: the rotation of the loop-carried row values (a cell's old values stay live for the
    // row below while its new ones are produced) becomes register renaming instead of ~3 moves per cell
    using Plain = std::integral_constant<int, 0>;
    using Window = std::integral_constant<int, 1>;
    int s = 0;
    if (TILED && seamed) {              // the seam variant shifts in place (its `old` operand is the seam window)
        for (; s < s_end; s += 2) { step(Plain{}, nI, nM, nI, nM, 0, false); step(Plain{}, nI, nM, nI, nM, 0, false); }
        if (s == s_end) step(Plain{}, nI, nM, nI, nM, 0, false);
    } else if (!STREAM) {
        for (; s < s_end; s += 2) { step(Plain{}, nI, nM, qI, qM, 0, false); step(Plain{}, qI, qM, nI, nM, 0, false); }
        if (s == s_end) step(Plain{}, nI, nM, qI, qM, 0, false);
    } else {
        // everything runs in pairs of steps; an odd step of a plain stretch is left to the window that follows, where it does
        // nothing (no lane answers to uw = -1)
        for (int k = 0; k < depth; ++k) {
            const int first_turn = (k + 1) * NC - 1;                  // the step on which lane 0 is on its last column
            for (const int s1 = s + ((first_turn - s) & ~1); s < s1; s += 2) {
                step(Plain{}, nI, nM, qI, qM, 0, false);
                step(Plain{}, qI, qM, nI, nM, 0, false);
            }
            const bool more = k + 1 < depth;
            for (const int s1 = more ? first_turn + W + 1 : s_end + 1; s < s1; s += 2) {     // (a step too many does nothing either)
                step(Window{}, nI, nM, qI, qM, s - first_turn, more);
                step(Window{}, qI, qM, nI, nM, s + 1 - first_turn, more);
            }
        }
    }
}

// The last row tile of a long read with rl <= RMAX rows per lane (rl is wave-uniform: one of RMAX copies of the sweep runs)
template <int RMAX>
__device__ __forceinline__ void rows_sweep_last(const int rl, const LdsTables &L, const int NC, const uint8_t *__restrict__ seq,
                                                const int nt, const int lane, unsigned *__restrict__ bpw,
                                                double *__restrict__ rown, const unsigned cap_base, int32_t *__restrict__ aux,
                                                const int sink_stride, const int row0, const double *__restrict__ seam)
{
    if constexpr (RMAX > 1) {
        if (rl < RMAX) { rows_sweep_last<RMAX - 1>(rl, L, NC, seq, nt, lane, bpw, rown, cap_base, aux, sink_stride, row0, seam); return; }
    }
    rows_sweep<RMAX, 1, true>(L, NC, NC - 1 + (nt - 1) / RMAX, seq, nt, lane, lane, bpw, rown, cap_base, aux, (unsigned)COL_MAX_TAIL,
                              sink_stride, row0, seam);
}

// A lane's rows of one queued read, as rows_sweep unpacks them when the lane gets there: 3 bits per row (base code, 4 = row
// past the read) and the slot of the read's last row (7 = not in this lane).
template <int R>
__device__ __forceinline__ unsigned rows_pack_read(const uint8_t *__restrict__ seq, const int n, const int lp)
{
    unsigned w = 0;
    unsigned char b[R];
#pragma unroll
    for (int k = 0; k < R; ++k) b[k] = seq[max(min(R * lp + k + 1, n), 1) - 1];      // (all R loads in flight together)
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int t = R * lp + k + 1;
        w |= (unsigned)((t <= n) ? (int)b[k] : 4) << (3 * k);
    }
    const int kc = (n >= 1 && (n - 1) / R == lp) ? (n - 1) - lp * R : 7;
    return w | (unsigned)kc << (3 * R);
}

// tail states, traceback (row-blocked back-pointer layout), summary and outputs of one read of the group; col0 = the steps
// the read's sweep began after (k * NC for the kth read of a back-to-back sweep)
template <int R, class Rev>
__device__ __forceinline__ void rows_finish_read(const ColArgs &g, const uint32_t flags, const ColProgram *__restrict__ cp,
                                                 const LdsTables &L, const DevModel &M, const int r,
                                                 const uint8_t *__restrict__ seq, const int n, double *final_row,
                                                 double *tailv, const unsigned *__restrict__ bpw, const int lane0,
                                                 int32_t *__restrict__ tailwin, const int32_t *__restrict__ sinkbp,
                                                 const Rev &rev, const int lane, const int col0)
{
    const int NC = cp->n_cols;
    // (ADVNTR_BUDGET_*: builds that leave one piece of the finish phase out -- wrong results, made only by scripts/budget_finish.sh
    // to price the pieces with SQ_INSTS_VALU; never defined in the shipped library)
#ifdef ADVNTR_BUDGET_NO_TAIL
    const double logp = final_row[3 * (NC - 1) + 2];
#else
    const double logp = col_tail(cp, final_row, tailwin, NC, lane, tailv);
#endif
    if (lane == 0) g.a.out_logp[r] = logp;
    int len = 0;
#ifndef ADVNTR_BUDGET_NO_TRACEBACK
    if (logp != -INFINITY) {
        auto bp_at = [&](int tt, int cc, int st) -> int { return rows_bp_at<R>(bpw, lane0, tt, cc, st); };
        len = col_traceback_walk(cp, L, n, M.start, M.P, bp_at, g.sink_stride, tailwin, sinkbp, rev, g.a.path_cap, lane, 0,
                                 1 << 30, col0);
        len = __builtin_amdgcn_readfirstlane(len);
    }
#endif
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
#ifdef ADVNTR_BUDGET_NO_SUMMARY
    if (lane == 0 && g.a.out_summary) g.a.out_summary[(int64_t)r * 8 + 7] = len;
#else
    col_emit_outputs(g, flags, M, r, seq, n, rev, len, lane);
#endif
}

// Measurement build only (scripts/build_variant.sh wgclocks -DADVNTR_WG_CLOCKS, scripts/wg_clocks.py): every workgroup leaves when
// it started, when it took its last tile, when it ended and how many tiles it ran at the head of its first wavefront's row
// scratch (s_memrealtime, 100 MHz) -- how evenly the dynamic dequeue ends a launch.  The shipped build compiles none of it.
#ifdef ADVNTR_WG_CLOCKS
#define WG_CLOCKS_BEGIN()                                                                                                          \
    const unsigned long long wgc_t0 = __builtin_amdgcn_s_memrealtime();                                                            \
    unsigned long long wgc_last = wgc_t0;                                                                                          \
    int wgc_tiles = 0
#define WG_CLOCKS_TILE() (++wgc_tiles, wgc_last = __builtin_amdgcn_s_memrealtime())
#define WG_CLOCKS_END(g, tid)                                                                                                      \
    if ((tid) == 0) {                                                                                                              \
        unsigned long long *wgc = (unsigned long long *)((g).rown + (int64_t)blockIdx.x * COL_WAVES * (g).rown_stride);            \
        wgc[0] = wgc_t0, wgc[1] = __builtin_amdgcn_s_memrealtime(), wgc[2] = wgc_last, wgc[3] = (unsigned long long)wgc_tiles;     \
    }
#else
#define WG_CLOCKS_BEGIN() ((void)0)
#define WG_CLOCKS_TILE() ((void)0)
#define WG_CLOCKS_END(g, tid) ((void)0)
#endif

template <int R, int G>
__global__ void __launch_bounds__(COL_WAVES * 64, ROWS_WAVES_PER_SIMD)
viterbi_rows_kernel(ColArgs g, uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int W = 64 / G, WORDS = (R + 4) / 5;
    static_assert(3 * (3 * R + 3) <= 64 && ROWS_DEPTH <= 4, "queued reads of a lane: one 64-bit register");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t gw = (int64_t)blockIdx.x * COL_WAVES + wave;
    int32_t *tile_slot = (int32_t *)lds;
    uint8_t *tables = lds + 16;
    // behind the tables: what the reads of a wavefront's sweep are (index, length, offset of the bases), kept out of the
    // registers while the sweep runs
    int4 *stash = (int4 *)(lds + 16 + g.lds_tables) + wave * (ROWS_DEPTH * ROWS_MAX_GROUPS);
    unsigned *bpw = (unsigned *)(g.bp + gw * g.bp_stride);
    double *rown = g.rown + gw * g.rown_stride;
    int32_t *aux = g.aux + gw * g.aux_stride;
    // the tail states' values and winning in-edges of the read being finished: in LDS as well (col_tail writes one of each per tail
    // state and waits for it -- a round trip to HBM per tail state and read when they sat in the wavefront's scratch there)
    double *tailv = (double *)(lds + 16 + g.lds_tables + ROWS_STASH_BYTES + ROWS_REV_BYTES) + wave * COL_MAX_TAIL;
    int32_t *tailwin = (int32_t *)(lds + 16 + g.lds_tables + ROWS_STASH_BYTES + ROWS_REV_BYTES + COL_WAVES * COL_MAX_TAIL * 8) + wave * COL_MAX_TAIL;
    // the reversed path of the read being finished: its first REV_LDS_ENTRIES states in this wavefront's piece of LDS (behind the
    // stash), the rest -- longer paths are rare -- in the wavefront's path scratch
    const RevLds rev{(__attribute__((address_space(3))) unsigned short *)(size_t)lds_addr(lds + 16 + g.lds_tables + ROWS_STASH_BYTES) +
                         wave * REV_LDS_ENTRIES,
                     g.a.path_scratch + gw * g.a.path_cap};
    const int grp = lane / W, lp = lane - grp * W;
    int cur_model = -1;
    bool padded = false;
    LdsTables L{};
    const ColProgram *cp = nullptr;
    DevModel M{};
    WG_CLOCKS_BEGIN();
    for (;;) {
        __syncthreads();
        if (tid == 0) *tile_slot = atomicAdd(g.tile_counter, 1);
        __syncthreads();
        const int ti = __builtin_amdgcn_readfirstlane(*tile_slot);
        if (ti >= g.n_tiles) {
            WG_CLOCKS_END(g, tid);
            break;
        }
        WG_CLOCKS_TILE();
        const ColTile tile = g.tiles[ti];
        if (tile.model != cur_model) {
            cur_model = tile.model;
            M = g.a.models[cur_model];
            cp = M.cols;
            padded = stage_model<1, true, true>(cp, tables, g.lds_tables, g.lds_level, L, tid);
        }
        const int NC = __builtin_amdgcn_readfirstlane(cp->n_cols);
        // per lane group: the row-n values of its reads one behind the other (padded by W columns either side: lanes run
        // ahead of and past their reads); the tail values of the read being finished sit behind the groups
        const int dmax = NC >= ROWS_STREAM_MIN_COLS ? g.rows_depth : 1;       // (the deepest tile of the launch: the scratch is laid out for it)
        const int64_t grp_doubles = 3 * ((int64_t)dmax * NC + 2 * W);
        // a round: every wavefront takes up to dmax reads per lane group; read (k, group) of wave w is the tile's read
        // j0 + (k * COL_WAVES + w) * G + group
        for (int j0 = 0; j0 < tile.count; j0 += COL_WAVES * G * dmax) {
            const int jw = j0 + wave * G;
            if (jw >= tile.count) break;
            const int depth = min(dmax, (tile.count - jw + COL_WAVES * G - 1) / (COL_WAVES * G));
            if (!padded) {                                                 // the host never routes such a tile here
                for (int k = 0; k < depth; ++k) {
                    const int idx = jw + k * COL_WAVES * G + grp;
                    if (idx < tile.count && lp == 0) g.a.out_logp[g.a.order[tile.first + idx]] = __longlong_as_double(0x7ff8000000000000ll);
                }
                continue;
            }
            // the reads of this sweep: lane k * G + q fetches what read k of lane group q is (index, length, where its bases
            // are); the sweep's lanes take their rows from there, and so does the finish phase afterwards
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
            int nbad = fn > R * (W - (G == 2 ? 1 : 0));
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) nbad |= __shfl_xor(nbad, o, 64);
            if (__builtin_amdgcn_readfirstlane(nbad)) {                    // the host never routes such a tile here
                if (fr >= 0) g.a.out_logp[fr] = __longlong_as_double(0x7ff8000000000000ll);
                continue;
            }
            // this lane's rows of its group's reads: the first one unpacked, the others queued (last one pushed first)
            unsigned long long queue = 0ull;
            for (int k = depth - 1; k >= 1; --k) {
                const int src = k * G + grp;
                const int nk = __shfl(fn, src, 64);
                const long long ok = ((long long)__shfl((int)(fo >> 32), src, 64) << 32) | (unsigned)__shfl((int)fo, src, 64);
                queue = (queue << (3 * R + 3)) | rows_pack_read<R>(g.a.bases + ok, nk, lp);
            }
            const int n = __shfl(fn, grp, 64);
            const uint8_t *seq = g.a.bases + (((long long)__shfl((int)(fo >> 32), grp, 64) << 32) | (unsigned)__shfl((int)fo, grp, 64));
            // steps: the last read's longest
            int nlast = 0;
#pragma unroll
            for (int q = 0; q < G; ++q) nlast = max(nlast, __builtin_amdgcn_readlane(fn, (depth - 1) * G + q));
            if (lane < ROWS_DEPTH * G) stash[lane] = make_int4(fr, fn, (int)fo, (int)(fo >> 32));
            const unsigned cap_base = (unsigned)(grp * grp_doubles);
            const unsigned sink_base = (unsigned)(COL_MAX_TAIL + grp * (dmax + 1) * COL_MAX_SINKS * g.sink_stride);
            const int s_end = depth * NC - 1 + (max(nlast, 1) - 1) / R;
            rows_sweep<R, G>(L, NC, s_end, seq, n, lp, lane, bpw, rown, cap_base, aux, sink_base, g.sink_stride, 0, nullptr,
                             queue, depth);
            rows_bp_publish();
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            int4 what = make_int4(0, 0, 0, 0);
            if (lane < ROWS_DEPTH * G) what = stash[lane];
#pragma unroll 1
            for (int k = 0; k < depth; ++k) {
#pragma unroll 1
                for (int q = 0; q < G; ++q) {
                    if (jw + k * COL_WAVES * G + q >= tile.count) break;
                    const int src = k * G + q;
                    const int rq = __builtin_amdgcn_readlane(what.x, src), nq = __builtin_amdgcn_readlane(what.y, src);
                    const uint8_t *sq = g.a.bases + (((long long)__builtin_amdgcn_readlane(what.w, src) << 32) |
                                                     (unsigned)__builtin_amdgcn_readlane(what.z, src));
                    rows_finish_read<R>(g, flags, cp, L, M, rq, sq, nq, rown + q * grp_doubles + 3 * (W + (int64_t)k * NC), tailv,
                                        bpw + (int64_t)k * NC * (64 * WORDS), q * W, tailwin,
                                        aux + COL_MAX_TAIL + (int64_t)(q * (dmax + 1) + k) * COL_MAX_SINKS * g.sink_stride, rev, lane, k * NC);
                }
            }
        }
    }
}

// Reads longer than the single-sweep kernels take (156 bases up to COL_MAX_LONG_READ), one per wavefront, in row tiles of
// 64 R rows: the last row of a tile ("seam", in the per-wave row buffers, ping-pong) is what the next tile's first lane
// shifts in instead of row 0.  One back-pointer slab per tile.
template <int R>
__global__ void __launch_bounds__(COL_WAVES * 64, ROWS_WAVES_PER_SIMD)
viterbi_rows_long_kernel(ColArgs g, uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int W = 64, RT = 64 * R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t gw = (int64_t)blockIdx.x * COL_WAVES + wave;
    int32_t *tile_slot = (int32_t *)lds;
    uint8_t *tables = lds + 16;
    unsigned *bpw = (unsigned *)(g.bp + gw * g.bp_stride);
    double *rown = g.rown + gw * g.rown_stride;
    int32_t *aux = g.aux + gw * g.aux_stride;
    int32_t *tailwin = aux;
    int32_t *rev = g.a.path_scratch + gw * g.a.path_cap;
    int cur_model = -1;
    bool padded = false;
    LdsTables L{};
    const ColProgram *cp = nullptr;
    DevModel M{};
    WG_CLOCKS_BEGIN();
    for (;;) {
        __syncthreads();
        if (tid == 0) *tile_slot = atomicAdd(g.tile_counter, 1);
        __syncthreads();
        const int ti = __builtin_amdgcn_readfirstlane(*tile_slot);
        if (ti >= g.n_tiles) {
            WG_CLOCKS_END(g, tid);
            break;
        }
        WG_CLOCKS_TILE();
        const ColTile tile = g.tiles[ti];
        if (tile.model != cur_model) {
            cur_model = tile.model;
            M = g.a.models[cur_model];
            cp = M.cols;
            padded = stage_model<1, true>(cp, tables, g.lds_tables, g.lds_level, L, tid);
        }
        const int NC = __builtin_amdgcn_readfirstlane(cp->n_cols);
        const int64_t row_doubles = 3 * (int64_t)(NC + 2 * W) + COL_MAX_TAIL;
        const int64_t slab = (int64_t)(NC + W) * 64;                  // back-pointer dwords per row tile
        for (int j = wave; j < tile.count; j += COL_WAVES) {
            const int r = __builtin_amdgcn_readfirstlane(g.a.order[tile.first + j]);
            const uint8_t *seq = g.a.bases + g.a.read_off[r];
            const int n = __builtin_amdgcn_readfirstlane((int)(g.a.read_off[r + 1] - g.a.read_off[r]));
            if (!padded) {                                            // the host never routes such a read here
                if (lane == 0) g.a.out_logp[r] = __longlong_as_double(0x7ff8000000000000ll);
                continue;
            }
            const int n_tiles = (n + RT - 1) / RT;
            // The last tile of a read holds n - (n_tiles - 1) RT rows, anything from 1 to RT: it is swept with as few rows per
            // lane as cover it (rl = ceil(rows / 64): the step costs what its rows cost, and a tile costs NC + 63 steps whatever it
            // holds -- with R rows per lane throughout, a read paid for ceil(n / RT) full tiles).  The back-pointers of that tile
            // are laid out for rl rows per lane; the traceback is told.
            const int rl = __builtin_amdgcn_readfirstlane((max(n - (n_tiles - 1) * RT, 1) + 63) >> 6);
            for (int i = 0; i < n_tiles; ++i) {
                const int row0 = i * RT, nt = min(RT, n - row0);
                const unsigned cap = (unsigned)(((i + 1) & 1) * row_doubles);
                const double *seam = i > 0 ? rown + (i & 1) * row_doubles : nullptr;
                if (i + 1 < n_tiles || rl == R)
                    rows_sweep<R, 1, true>(L, NC, NC - 1 + (nt - 1) / R, seq + row0, nt, lane, lane, bpw + i * slab, rown, cap, aux,
                                           (unsigned)COL_MAX_TAIL, g.sink_stride, row0, seam);
                else
                    rows_sweep_last<R - 1>(rl, L, NC, seq + row0, nt, lane, bpw + i * slab, rown, cap, aux, g.sink_stride, row0, seam);
                __threadfence_block();
                __builtin_amdgcn_wave_barrier();
            }
            rows_bp_publish();
            double *final_row = rown + (n_tiles & 1) * row_doubles + 3 * W;
            const double logp = col_tail(cp, final_row, tailwin, NC, lane);
            if (lane == 0) g.a.out_logp[r] = logp;
            int len = 0;
            if (logp != -INFINITY) {
                // x / rl for x < 400 as a multiply and a shift (rl = 1 .. 5)
                const unsigned rl_magic = rl == 1 ? 65536u : (rl == 2 ? 32768u : (rl == 3 ? 21846u : (rl == 4 ? 16384u : 13108u)));
                auto bp_at = [&](int tt, int cc, int st) -> int {
                    const int tl = (tt - 1) / RT;
                    if (tl + 1 < n_tiles) return rows_bp_at<R>(bpw + tl * slab, 0, tt - tl * RT, cc, st);
                    const int x = tt - tl * RT - 1, lp = (int)(((unsigned)x * rl_magic) >> 16);
                    return rows_bp_at_split(bpw + tl * slab, lp, x - lp * rl, cc, st);
                };
                len = col_traceback_walk(cp, L, n, M.start, M.P, bp_at, g.sink_stride, tailwin, aux + COL_MAX_TAIL, rev,
                                         g.a.path_cap, lane, 0, 1 << 30);
                len = __builtin_amdgcn_readfirstlane(len);
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            col_emit_outputs(g, flags, M, r, seq, n, rev, len, lane);
        }
    }
}
