// viterbi_columns.h -- anti-diagonal Viterbi kernel for models with a column program.
//
// One read per wavefront, lane <-> read position (row t = 64*chunk + lane + 1), up to 4 chunks of 64 rows
// held entirely in VGPRs.  At step s every lane advances to column c = s - t of its row, so the 64 lanes
// of a chunk sit on one anti-diagonal of the trellis and all three data dependencies of the profile-HMM
// stencil are either the lane's own registers (same row, previous column) or the neighbouring lane's
// values of the previous step (previous row), fetched with a wavefront shift (DPP wave_shr:1).  There
// is no same-row serial chain left: the delete chain that serialises a row-major sweep runs ALONG the
// step axis here.  Model parameters are staged once per workgroup in LDS as small class tables
// (column_program.h); per step a lane reads its column's 16-B info word, the 88-B transition class and
// two emission log-probs.  The arithmetic is the reference's, operation for operation:
// (v + t) + e in fp64, candidates compared in the reference's in-edge order with strict '>'
// (/root/reference/pomegranate/hmm.pyx:2026-2083), so scores and back-pointers are bit-identical.
//
// Back-pointers: one byte per (row, column) cell = the outcomes of the cell's six comparisons (relax_bit), written
// in diagonal-major order so that a chunk stores 64 consecutive bytes per step.  The last row's values are
// also written to a small per-wave buffer from which the "tail" states (prefix_end_prefix, model end:
// fan-in from every match state) are evaluated once, wave-parallel, after the sweep.  The wave then walks the
// pointers back cooperatively (col_traceback) and summarises the path (path_summary.h).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <vector>
#include "column_program.h"
#include "viterbi_generic.h"

#define COL_WAVES 4                 // wavefronts per workgroup (all on one model at a time)
#ifndef COL_UNROLL2_MAX_K
#define COL_UNROLL2_MAX_K 3          // sweeps with at most this many chunks run two steps per loop iteration
#endif
#define COL_MAX_TAIL 16
#define COL_TILE_READS 16
// Long reads (> COL_MAX_READ rows) are swept in row tiles of 64*COL_LONG_K rows.  Measured on the PacBio-size bench
// (scripts/pacbio_bench.py 40000: 1440-column model, reads of 900-1500 bases), once LDS no longer limited the
// resident workgroups (ColArgs::lds_level): 192-row tiles at 4 waves/SIMD 265 k reads/s, 128-row tiles at 4 waves
// 254 k, 192-row tiles at 3 waves 229 k, 256-row tiles at 2 waves 161 k.
#ifndef COL_LONG_K
#define COL_LONG_K 3
#endif
#ifndef COL_LONG_WAVES
#define COL_LONG_WAVES 4
#endif
#ifndef COL_LONG4_WAVES
#define COL_LONG4_WAVES 2
#endif
#ifndef COL_MIN_WAVES_PER_SIMD
#define COL_MIN_WAVES_PER_SIMD 4
#endif

struct ColTile {
    int32_t model, first, count, pad;   // reads order[first .. first+count)
};

struct ColArgs {
    BatchArgs a;
    const ColTile *tiles;
    int32_t n_tiles;
    int32_t *tile_counter;
    double *rown;            // per wave: rown_stride doubles  (3*NC row-n values + COL_MAX_TAIL tail values)
    int64_t rown_stride;
    int32_t *aux;            // per wave: aux_stride ints (tail winners + sink back-pointers)
    int64_t aux_stride;
    uint8_t *bp;             // per wave: bp_stride bytes
    int64_t bp_stride;
    int32_t lds_tables;      // bytes of LDS reserved for the tables
    int32_t sink_stride;     // ints per fan-in state in the sink back-pointer array (n_max + 1)
    int32_t ring;            // stream kernel: back-pointer slabs per wave (row tiles kept for the traceback)
    int32_t rows_depth;      // row-blocked kernels: reads per lane group of the deepest tile (back-to-back sweeps, viterbi_rows.h)
    int32_t fwd_tailw_cap;   // forward_rows_kernel: tail-edge weights (exp of the transition log-probabilities) that fit the LDS
                             // behind the row-0 table; 0: none, the weights are exponentiated per read
    int32_t lds_level;       // which tables of the column program are staged in LDS: 2 = all; 1 = all but the traceback's
                             // column->state table; 0 = only classes and emissions (+ the padded info copy the sweep
                             // indexes).  The rest is read from the model blob in HBM/L2, which leaves room for more
                             // resident workgroups per CU on wide models (PacBio: > 1000 columns)
};

__device__ __forceinline__ int dpp_wave_shr1(int old, int src)
{
    // lane i <- src[i-1]; lane 0 keeps `old` (bound_ctrl = 0)
    return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ double shift_up1(double v, double inject)
{
    const int lo = dpp_wave_shr1(__double2loint(inject), __double2loint(v));
    const int hi = dpp_wave_shr1(__double2hiint(inject), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bcast63(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// One Viterbi relaxation: if (cand > best) best = cand  -- strict '>' keeps the first maximum, as the reference
// does (hmm.pyx:2039,2060,2080) -- and the outcome of the comparison is shifted into `bits` (bits = 2*bits + won).
// A cell's six comparisons leave six bits = its back-pointer byte (layout below), so no per-pointer selects and
// no packing are needed.  Three instructions: v_cmp_gt_f64 into VCC, v_max_f64 for the value (equal to the select:
// no NaNs occur and max(a,b) of equal values is that value), v_addc_co_u32 bits+bits+VCC.  Written as assembly
// because the compiler lowers the C form to a compare, two or three VOP3 selects per relaxation and an or/shift
// chain per cell (SQ_INSTS_VALU per launch 13.3 G -> see profiles).
//   byte = aI<<5 | bI<<4 | aM<<3 | bM<<2 | aB<<1 | bB      (a: the 2nd candidate won, b: the last one did)
__device__ __forceinline__ void relax_bit(double &best, int &bits, const double cand)
{
    asm("v_cmp_gt_f64_e32 vcc, %2, %0\n\t"
        "v_max_f64 %0, %0, %2\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
        : "+v"(best), "+v"(bits)
        : "v"(cand)
        : "vcc");
}
// the first relaxation of a cell starts the byte (bits = won): saves the move that would clear it
__device__ __forceinline__ void relax_bit_first(double &best, int &bits, const double cand)
{
    asm("v_cmp_gt_f64_e32 vcc, %2, %0\n\t"
        "v_max_f64 %0, %0, %2\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, %3, vcc"
        : "+v"(best), "=v"(bits)
        : "v"(cand), "v"(0)
        : "vcc");
}
// pointers out of a back-pointer byte: 0/1/2(/3) = index of the winning candidate in evaluation order
// (M: 0 = I, 1 = M -- or, in a read's first row, the entry edge --, 3 = b of the previous column)
__device__ __forceinline__ int bp_ptr_I(const int byte) { return (byte & 0x10) ? 2 : ((byte >> 5) & 1); }
__device__ __forceinline__ int bp_ptr_M(const int byte) { return (byte & 0x04) ? 3 : ((byte >> 3) & 1); }
__device__ __forceinline__ int bp_ptr_B(const int byte) { return (byte & 0x01) ? 2 : ((byte >> 1) & 1); }

#ifndef ADVNTR_LSE2_DEFINED
#define ADVNTR_LSE2_DEFINED
// pair_lse of the reference (utils.pyx:72-90)
__device__ __forceinline__ double lse2(double x, double y)
{
    // branch-free form of the same expression: hi + log(exp(lo - hi) + 1) with hi/lo picked exactly as the
    // reference's (x > y) test does; the -inf / +inf cases are selects, so a wavefront never diverges here
    const bool xg = x > y;
    const double hi = xg ? x : y, lo = xg ? y : x;
    double r = hi + log(exp(lo - hi) + 1.0);
    r = (lo == -INFINITY) ? hi : r;
    r = (x == INFINITY || y == INFINITY) ? INFINITY : r;
    return r;
}
#endif

// LDS byte address of a pointer into the workgroup's shared memory, and typed reads at such an address.  The sweep's
// info words carry ready LDS addresses of the column's transition class and emission records (16 bits each: the class
// and emission tables sit at the start of the table area, far below 64 KiB), so a lane reaches its parameters with
// an `and`/`add` instead of an unpack-multiply-add chain per table.
typedef __attribute__((address_space(3))) const ColClass LdsClass;
typedef __attribute__((address_space(3))) const double LdsDouble;
typedef double adv_f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const adv_f64x2 LdsDouble2;
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}

typedef __attribute__((address_space(3))) const unsigned long long LdsU64;
__device__ __forceinline__ uint2 lds_uint2(const unsigned addr)
{
    const unsigned long long w = *(LdsU64 *)(size_t)addr;
    return make_uint2((unsigned)w, (unsigned)(w >> 32));
}

struct LdsTables {
    unsigned class_base, emis_base;   // LDS byte addresses of the two tables
    unsigned epair_base, epair_sym_stride;   // emission pair table (row-blocked Viterbi sweep): base and bytes per symbol row
    unsigned fwd_lin;                 // sum-product kernel: LDS byte address of {row-0 value of b_c, entry term of M_c} per
                                      // column, in the linear domain (forward_columns.h)
    unsigned pinfo;                   // LDS byte address of the padded info copy (record c + 64K; 0 if it does not fit):
                                      // {v0b, class address | flags << 16, emM address | emI address << 16}
    const ColClass *classes;
    const double *emis;
    const ColInfo *info0;     // original table, index c + 1 (LDS or, for wide models, the model blob in HBM/L2)
    const ColState *state;
};

template <int K>
struct ColRegs {
    double I[K], M[K], B[K], pI[K], pM[K], pB[K], er[K];
    int erwin[K], x[K];
    uint2 meta[K];             // this step's column info words (class / emission ids / flags), loaded a step ahead
    double v0b0;               // chunk 0: row-0 value of this step's column
    double n0I, n0M;           // chunk 0, first tile: shifted I/M values; lane 0 holds the row-0 value -inf for the
                               // whole sweep (wave_shr never writes lane 0), so the shift needs no fill move
    int sflag[K];              // stream mode: bit 0 = first row of a read, bit 1 = last row, bits 8.. = capture slot
    // row-tiling only: values of the previous tile's last row for 64 columns (lane i <-> column cb+i), and
    // the next 64 (prefetched)
    double sI, sM, sB, tI, tM, tB;
};

// Per-tile sweep context.  A read longer than 64*K rows is processed in row tiles: the last row of a tile
// ("seam", 3 fp64 per column in HBM) is the injected previous row of the next tile.
struct TileCtx {
    int NC;
    int n_tile;                // rows in this tile (1 .. 64K)
    int row0;                  // sink back-pointer row of the tile's first row minus one (index = row0 + t)
    int sink_stride;           // ints per sink in sinkbp
    uint8_t *bp;               // this tile's back-pointer slab
    double *cap;               // where the tile's last row goes (next seam, or the row-n buffer)
    const double *seam;        // previous tile's last row (nullptr for the first tile)
    int32_t *sinkbp;
    // stream mode (several reads packed along the row axis)
    double *seam_out;          // last row of a full tile -> next tile's seam
    int64_t cap_stride;        // doubles between the capture buffers of the reads that end in this tile
    unsigned hasfirst, haslast; // bit k: chunk k holds a first / last row of some read
    const double *fwd;         // sum-product kernel: per column {row-0 forward value of b_c, entry term of M_c} (log domain)
    double seam_off;           // sum-product kernel: log-domain offset of this row tile (its seam row was divided by
                               // exp(seam_off) when it was loaded); 0 for the first tile
};

__device__ __forceinline__ double shift_up1_from(double v, double prev_chunk)
{
    // lane i <- v[i-1]; lane 0 <- prev_chunk[63]   (wave_ror:1 feeds the `old` operand of wave_shr:1)
    // (every lane of a rotate has a source, so the move needs no `old` value: mov_dpp instead of update_dpp saves
    // the v_mov that would initialise it)
    const int rlo = __builtin_amdgcn_mov_dpp(__double2loint(prev_chunk), 0x13C, 0xf, 0xf, false);
    const int rhi = __builtin_amdgcn_mov_dpp(__double2hiint(prev_chunk), 0x13C, 0xf, 0xf, false);
    const int lo = dpp_wave_shr1(rlo, __double2loint(v));
    const int hi = dpp_wave_shr1(rhi, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// One trellis cell per lane of chunk k.  CHECKED = clamp the column index (used when chunk ranges cannot
// be split into branch-free phases); otherwise the info table is padded with 64*K dummy columns per side.
// MODE 0 = first (or only) row tile of one read: row 0 comes from the host-precomputed v0b, entry edges X are live
//          in row 1 (chunk 0, lane 0);
// MODE 1 = continuation tile of a long read: row 0 of the tile is the previous tile's last row (seam);
// MODE 2 = stream tile: several reads packed back to back along the row axis, first/last rows flagged per lane.
template <int K, bool CHECKED, int MODE, bool FWD = false>
__device__ __forceinline__ void col_cell(ColRegs<K> &R, const int k, const LdsTables &L, const TileCtx &C, const int s,
                                         const int lane, const double injI, const double injM, const double injB)
{
    constexpr int TPAD = 64 * K;
    const int NC = C.NC;
    const int t = 64 * k + lane + 1;
    const int c = s - t;
    const int cc = CHECKED ? min(max(c + 1, 0), NC + 1) : c + TPAD;
    // meta.x = class LDS address | flags << 16;  meta.y = emM record LDS address | emI record LDS address << 16
    uint2 meta;
    double v0b = 0.0;
    if (CHECKED) {
        const ColInfo inf = L.info0[cc];
        meta = make_uint2((L.class_base + (unsigned)inf.tclass * (unsigned)sizeof(ColClass)) | ((unsigned)inf.flags << 16),
                          (L.emis_base + (unsigned)inf.emM * (COL_EMIS_STRIDE * 8u)) |
                              ((L.emis_base + (unsigned)inf.emI * (COL_EMIS_STRIDE * 8u)) << 16));
        v0b = inf.v0b;
    } else {
        // the info word was loaded during the previous step (software pipelining of the dependent LDS chain
        // info -> class record); fetch the next column's now
        meta = R.meta[k];
        R.meta[k] = lds_uint2(L.pinfo + (unsigned)(cc + 1) * 16u + 8u);
        if (MODE == 0 && k == 0) { v0b = R.v0b0; R.v0b0 = *(LdsDouble *)(size_t)(L.pinfo + (unsigned)(cc + 1) * 16u); }
    }
    LdsClass *T = (LdsClass *)(size_t)(meta.x & 0xffffu);
    double fwd_mX = 0.0;
    if (FWD && MODE == 0 && k == 0) {            // row 0 / entry terms of the sum-product recursion (lane 0 only)
        const int cq = min(max(c, 0), NC - 1);
        v0b = *(LdsDouble *)(size_t)(L.fwd_lin + (unsigned)cq * 16u);
        fwd_mX = *(LdsDouble *)(size_t)(L.fwd_lin + (unsigned)cq * 16u + 8u);
    }
    // previous row, same column: the neighbouring lane's values of the previous step
    double nI, nM, nB;
    if (k == 0) {
        if (MODE == 0) {
            nI = R.n0I = shift_up1(R.I[0], R.n0I);
            nM = R.n0M = shift_up1(R.M[0], R.n0M);
            nB = shift_up1(R.B[0], v0b);          // row 0 is read independent (host precomputed)
        } else {
            nI = shift_up1(R.I[0], injI);
            nM = shift_up1(R.M[0], injM);
            nB = shift_up1(R.B[0], injB);
        }
    } else {
        nI = shift_up1_from(R.I[k], R.I[k - 1]);
        nM = shift_up1_from(R.M[k], R.M[k - 1]);
        nB = shift_up1_from(R.B[k], R.B[k - 1]);
    }
    const bool first_row = MODE == 2 ? (R.sflag[k] & 1) != 0 : false;
    const bool chunk_first = MODE == 2 && ((C.hasfirst >> k) & 1u);          // wave-uniform
    if (chunk_first) {                            // a read starts in this chunk: its row 0 is the model's
        const double v0c = CHECKED ? v0b : *(LdsDouble *)(size_t)(L.pinfo + (unsigned)cc * 16u);
        nI = first_row ? -INFINITY : nI;
        nM = first_row ? -INFINITY : nM;
        nB = first_row ? v0c : nB;
    }
    const double eI = *(LdsDouble *)(size_t)((meta.y >> 16) + R.x[k]);          // R.x = 8 * base code
    const double eM = *(LdsDouble *)(size_t)((meta.y & 0xffffu) + R.x[k]);
    double vI, vM, vB;
    int bits;                     // the cell's back-pointer byte, one comparison outcome per bit (relax_bit)
    if (FWD) {
        bits = 0;
        // sum-product in the LINEAR domain (forward_columns.h): the class and emission tables in LDS hold probabilities
        // (emissions times the per-row scale 16), values are probabilities times 16^row (times the tile's offset), so a
        // cell is a dozen multiply-adds instead of the fourteen exp/log calls of pair_lse folding
        // (explicit fma(): the build contracts nothing by itself, -ffp-contract=off)
        vI = fma(nB, T->iD, fma(nI, T->iI, nM * T->iM)) * eI;
        double accM = fma(R.pI[k], T->mI, R.pM[k] * T->mM);
        if (MODE == 0 && k == 0) accM = accM + ((t == 1) ? fwd_mX : 0.0);
        vM = fma(R.pB[k], T->mD, accM) * eM;
        vB = fma(R.B[k], T->dD, fma(R.I[k], T->dI, R.M[k] * T->dM));
        const unsigned flf = meta.x >> 16;
        if (__ballot((flf & 3u) != 0)) {
            if (flf & COL_FLAG_SINK) { vB = R.er[k]; R.er[k] = 0.0; }
            if (flf & COL_FLAG_FEED) R.er[k] = fma(vB, T->erw, R.er[k]);
        }
    } else {
    // I_c(t) <- [I_c, M_c, b_c](t-1)
    vI = (nI + T->iI) + eI;
    relax_bit_first(vI, bits, (nM + T->iM) + eI);
    relax_bit(vI, bits, (nB + T->iD) + eI);
    // M_c(t) <- [I_{c-1}, M_{c-1}, X, b_{c-1}](t-1); the entry edge X only exists for row 1 (chunk 0, lane 0)
    vM = (R.pI[k] + T->mI) + eM;
    // In a read's first row the I and M candidates are -inf (row 0 holds silent states only), so the entry edge X --
    // third in the reference's order, ahead of b -- can take the M candidate's place there without changing any
    // comparison: one select instead of a fourth relaxation, and pointer 1 in row 1 reads "entry edge".
    double cM = R.pM[k] + T->mM;
    if (MODE == 0 && k == 0) cM = (t == 1) ? T->mX : cM;
    else if (chunk_first) cM = first_row ? T->mX : cM;
    relax_bit(vM, bits, cM + eM);
    relax_bit(vM, bits, (R.pB[k] + T->mD) + eM);
    // b_c(t) <- [I_{c-1}, M_{c-1}, b_{c-1}](t)  (own values of the previous step)
    vB = R.I[k] + T->dI;
    relax_bit(vB, bits, R.M[k] + T->dM);
    relax_bit(vB, bits, R.B[k] + T->dD);
    const unsigned fl = meta.x >> 16;
    if (__ballot((fl & 3u) != 0)) {                                  // wave-uniform skip
        if (fl & COL_FLAG_SINK) {
            vB = R.er[k];                                            // fan-in column: the traceback reads the winner
                                                                     // from sinkbp whatever the byte's B bits say
            if (t <= C.n_tile) C.sinkbp[((fl >> 4) & 15u) * C.sink_stride + C.row0 + t] = R.erwin[k];   // padding rows own no slot
            R.er[k] = -INFINITY;
        }
        if (fl & COL_FLAG_FEED) {
            const double cand = vB + T->erw;
            if (cand > R.er[k]) { R.er[k] = cand; R.erwin[k] = c; }
        }
    }
    }
    R.pI[k] = nI; R.pM[k] = nM; R.pB[k] = nB;
    R.I[k] = vI; R.M[k] = vM; R.B[k] = vB;
    if (!FWD) C.bp[(int64_t)(s - 1) * TPAD + (t - 1)] = (uint8_t)bits;
    if (MODE == 2) {
        if (k == K - 1) {                                            // last row of a full tile -> next tile's seam
            if (lane == 63 && c >= 0 && c < NC) {
                C.seam_out[c * 3 + 0] = vI;
                C.seam_out[c * 3 + 1] = vM;
                C.seam_out[c * 3 + 2] = vB;
            }
        }
        if ((C.haslast >> k) & 1u) {                                 // wave-uniform: a read ends in this chunk
            if ((R.sflag[k] & 2) && c >= 0 && c < NC) {
                double *cap = C.cap + (int64_t)(R.sflag[k] >> 8) * C.cap_stride;
                cap[c * 3 + 0] = vI;
                cap[c * 3 + 1] = vM;
                cap[c * 3 + 2] = vB;
            }
        }
    } else if (MODE == 0 ? (k == K - 1) : true) {                    // single tile: row n lives in the last chunk
        if (t == C.n_tile && c >= 0 && c < NC) {
            // (the sum-product kernel captures probabilities here and takes their logarithms after the sweep, wave-parallel:
            // a log inside this branch would run on one active lane at every step)
            C.cap[c * 3 + 0] = vI;
            C.cap[c * 3 + 1] = vM;
            C.cap[c * 3 + 2] = vB;
        }
    }
}

// seam values for the column lane 0 works on at step s (row-tiled reads only): registers hold 64 columns,
// the next 64 are prefetched one block ahead
template <int K, bool FWD = false>
__device__ __forceinline__ void seam_fetch(ColRegs<K> &R, const TileCtx &C, const int s, const int lane, double &injI,
                                           double &injM, double &injB)
{
    const int j = (s - 1) & 63;
    if (j == 0) {
        R.sI = R.tI; R.sM = R.tM; R.sB = R.tB;
        const int cn = min(s - 1 + 64 + lane, C.NC - 1);
        R.tI = C.seam[cn * 3 + 0];
        R.tM = C.seam[cn * 3 + 1];
        R.tB = C.seam[cn * 3 + 2];
        if (FWD) {              // the seam is kept in the log domain; the tile works on exp(seam - seam_off) <= 1
            R.tI = exp(R.tI - C.seam_off); R.tM = exp(R.tM - C.seam_off); R.tB = exp(R.tB - C.seam_off);
        }
    }
    injI = readlane_f64(R.sI, j);
    injM = readlane_f64(R.sM, j);
    injB = readlane_f64(R.sB, j);
}

template <int K, int KLO, int KHI, int MODE, bool FWD = false>
__device__ __forceinline__ void col_phase(ColRegs<K> &R, const int s0, const int s1, const LdsTables &L, const TileCtx &C,
                                          const int lane)
{
    constexpr int TPAD = 64 * K;
#pragma unroll
    for (int k = KHI; k >= KLO; --k) {              // info words of the phase's first step
        const int cc = s0 - (64 * k + lane + 1) + TPAD;
        R.meta[k] = lds_uint2(L.pinfo + (unsigned)cc * 16u + 8u);
        if (MODE == 0 && k == 0) R.v0b0 = *(LdsDouble *)(size_t)(L.pinfo + (unsigned)cc * 16u);
    }
    auto step = [&](const int s) {
        double injI = 0, injM = 0, injB = 0;
        if (MODE != 0) seam_fetch<K, FWD>(R, C, s, lane, injI, injM, injB);
#pragma unroll
        for (int k = KHI; k >= KLO; --k) col_cell<K, false, MODE, FWD>(R, k, L, C, s, lane, injI, injM, injB);
    };
    // two steps per iteration: the "previous step" copies (pI/pM/pB <- shifted values) become register renames and
    // the loop overhead halves (-4.6 % kernel time at K = 3); at K = 4 the doubled body only adds spills
    if (COL_UNROLL2_MAX_K >= K) {
        int s = s0;
        for (; s < s1; s += 2) { step(s); step(s + 1); }
        if (s == s1) step(s);
    } else {
        for (int s = s0; s <= s1; ++s) step(s);
    }
}

template <int K, int MODE, bool FWD = false>
__device__ __forceinline__ void col_sweep(const LdsTables &L, const bool padded, const TileCtx &C,
                                          const uint8_t *__restrict__ seq_tile, const int lane,
                                          const int *slot_x = nullptr, const int *slot_flag = nullptr)
{
    const int NC = __builtin_amdgcn_readfirstlane(C.NC), n = __builtin_amdgcn_readfirstlane(C.n_tile);   // scalar loop bounds
    ColRegs<K> R;
    const double NONE = FWD ? 0.0 : -INFINITY;      // "impossible": probability 0 in the linear sum-product sweep
#pragma unroll
    for (int k = 0; k < K; ++k) {
        R.I[k] = R.M[k] = R.B[k] = R.pI[k] = R.pM[k] = R.pB[k] = R.er[k] = NONE;
        R.erwin[k] = 0;
        const int t = 64 * k + lane + 1;
        if (MODE == 2) {
            R.x[k] = slot_x[k] * 8;
            R.sflag[k] = slot_flag[k];
        } else {
            R.x[k] = (t <= n) ? 8 * (int)seq_tile[t - 1] : 0;
            R.sflag[k] = 0;
        }
    }
    R.sI = R.sM = R.sB = R.tI = R.tM = R.tB = NONE;
    R.n0I = R.n0M = NONE;
    if (MODE != 0) {                                // columns 0..63 of the seam; seam_fetch rotates at s = 1
        const int cn = min(lane, NC - 1);
        R.tI = C.seam[cn * 3 + 0];
        R.tM = C.seam[cn * 3 + 1];
        R.tB = C.seam[cn * 3 + 2];
        if (FWD) { R.tI = exp(R.tI - C.seam_off); R.tM = exp(R.tM - C.seam_off); R.tB = exp(R.tB - C.seam_off); }
    }
    const int s_end = n + NC - 1;
    if (padded && NC + 63 >= 64 * (K - 1) + 1) {
        // chunk k is busy for steps [64k+1, 64k+64+NC-1]: ramp-up phases, a branch-free steady state with all
        // chunks in one basic block (independent dependency chains interleave), ramp-down phases
        if (K >= 2) col_phase<K, 0, 0, MODE, FWD>(R, 1, min(s_end, 64), L, C, lane);
        if (K >= 3) col_phase<K, 0, (K >= 3 ? 1 : 0), MODE, FWD>(R, 65, min(s_end, 128), L, C, lane);
        if (K >= 4) col_phase<K, 0, (K >= 4 ? 2 : 0), MODE, FWD>(R, 129, min(s_end, 192), L, C, lane);
        col_phase<K, 0, K - 1, MODE, FWD>(R, 64 * (K - 1) + 1, min(s_end, NC + 63), L, C, lane);
        if (K >= 2) col_phase<K, (K >= 2 ? 1 : 0), K - 1, MODE, FWD>(R, NC + 64, min(s_end, NC + 127), L, C, lane);
        if (K >= 3) col_phase<K, (K >= 3 ? 2 : 0), K - 1, MODE, FWD>(R, NC + 128, min(s_end, NC + 191), L, C, lane);
        if (K >= 4) col_phase<K, (K >= 4 ? 3 : 0), K - 1, MODE, FWD>(R, NC + 192, min(s_end, NC + 255), L, C, lane);
    } else {
        for (int s = 1; s <= s_end; ++s) {
            double injI = 0, injM = 0, injB = 0;
            if (MODE != 0) seam_fetch<K, FWD>(R, C, s, lane, injI, injM, injB);
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                if (s < 64 * k + 1 || s > 64 * k + 64 + NC - 1) continue;      // wave-uniform
                col_cell<K, true, MODE, FWD>(R, k, L, C, s, lane, injI, injM, injB);
            }
        }
    }
}

// The finish phase of a read (tail states, traceback) is a chain of small dependent reads of the model blob.  The blob is
// read-only for the kernel and the addresses are wave-uniform, so they go through the scalar cache as constant-address-
// space loads (s_load) -- as plain pointers the compiler issues a vector load and a full wait for each (cp->n_tail,
// tptr[i], tstate[ti], edges[...], pred0[...]: 30-40 round trips per read).
#define ADV_CONST_AS __attribute__((address_space(4)))
struct ColFinishTables {
    int n_cols, n_tail, end_tail;
    const ADV_CONST_AS int32_t *tptr, *tstate, *pred0;
    const ADV_CONST_AS TailEdge *edges_u;      // wave-uniform index
    const TailEdge *edges;                     // lane-indexed
};
__device__ __forceinline__ ColFinishTables col_finish_tables(const ColProgram *__restrict__ cp)
{
    const unsigned long long v = (unsigned long long)cp;
    const unsigned long long b = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    const ADV_CONST_AS ColProgram *h = (const ADV_CONST_AS ColProgram *)b;
    ColFinishTables F;
    F.n_cols = h->n_cols; F.n_tail = h->n_tail; F.end_tail = h->end_tail;
    F.tptr = (const ADV_CONST_AS int32_t *)(b + (unsigned)h->off_tail_ptr);
    F.tstate = (const ADV_CONST_AS int32_t *)(b + (unsigned)h->off_tail_state);
    F.pred0 = (const ADV_CONST_AS int32_t *)(b + (unsigned)h->off_pred0);
    F.edges_u = (const ADV_CONST_AS TailEdge *)(b + (unsigned)h->off_tail_edge);
    F.edges = (const TailEdge *)(b + (unsigned)h->off_tail_edge);
    return F;
}

// Reductions over the 64 lanes with DPP moves (no LDS round trips): butterflies inside the rows of 16 lanes, then the row
// results passed on to the next row's lanes (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3): lane 63 holds the
// result, returned wave-uniform.
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ double wave_dpp_f64(const double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, CTRL, ROWS, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), CTRL, ROWS, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_max_f64(double v)
{
    v = fmax(v, wave_dpp_f64<0xB1>(v));            // quad_perm [1,0,3,2]
    v = fmax(v, wave_dpp_f64<0x4E>(v));            // quad_perm [2,3,0,1]
    v = fmax(v, wave_dpp_f64<0x141>(v));           // row_half_mirror
    v = fmax(v, wave_dpp_f64<0x140>(v));           // row_mirror
    v = fmax(v, wave_dpp_f64<0x142, 0xA>(v));      // row_bcast15
    v = fmax(v, wave_dpp_f64<0x143, 0xC>(v));      // row_bcast31
    const long long b = __double_as_longlong(v);
    return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(b >> 32), 63) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)b, 63));
}
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// Tail states at the last row: first maximum over the reference-order in-edge list, wave-parallel: a lane takes the edges
// e0 + lane, e0 + lane + 64, ... in order (four at a time: the loads of the edge records, then those of the values they point
// at, are in flight together), the maximum over the lanes is found first and then the lowest edge number among the lanes that
// hold it.  (tail values go to `tailv`; by default right behind the row)
__device__ __forceinline__ double col_tail(const ColProgram *__restrict__ cp, double *__restrict__ rown,
                                           int32_t *__restrict__ tailwin, const int NC, const int lane,
                                           double *__restrict__ tailv = nullptr)
{
    const ColFinishTables F = col_finish_tables(cp);
    if (tailv == nullptr) tailv = rown + 3 * NC;
    double result = -INFINITY;
    const int n_tail = F.n_tail, end_tail = F.end_tail;
    int e1 = F.tptr[0];
    for (int i = 0; i < n_tail; ++i) {
        double best = -INFINITY;
        int rank = 0x7fffffff;
        const int e0 = e1;
        e1 = F.tptr[i + 1];
        for (int eb = e0; eb < e1; eb += 256) {
            TailEdge ed[4];
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ed[u] = F.edges[min(eb + 64 * u + lane, e1 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ed[u].loc >= 0 ? rown[(ed[u].loc >> 2) * 3 + (ed[u].loc & 3)] : tailv[-ed[u].loc - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = eb + 64 * u + lane;
                const double cand = v[u] + ed[u].logp;
                if (e < e1 && cand > best) { best = cand; rank = e; }
            }
        }
        const double top = wave_max_f64(best);
        const int first = wave_min_i32(best == top ? rank : 0x7fffffff);
        if (lane == 0) { tailv[i] = top; tailwin[i] = first; }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        if (i == end_tail) result = top;
    }
    return result;
}

// Wave-cooperative traceback.  The walk position (t, c, slot, len) is wave-uniform.  Viterbi paths of reads
// are dominated by match->match moves, i.e. runs along a trellis diagonal: for those, the 64 lanes gather the
// back-pointer bytes of (t-i, c-i), i = 0..63, in one round trip, a ballot gives the length of the M->M run
// and the run's states are written in parallel.  Everything else advances one cell at a time (broadcast load).
// bp_at(t, c, st) -> back-pointer byte of cell (t, c), of which the walk uses the two bits of state st (0 = I, 1 = M, 2 = b):
// the layout of the back-pointer store belongs to the sweep that wrote it
template <class BpAt, class Rev>
__device__ __forceinline__ int col_traceback_walk(const ColProgram *__restrict__ cp, const LdsTables &L, const int n,
                                                  const int start_state, const int P, const BpAt &bp_at, const int sink_stride,
                                                  const int32_t *__restrict__ tailwin, const int32_t *__restrict__ sinkbp,
                                                  const Rev &rev, const int cap, const int lane,
                                                  const int U0, const int W, const int sink_col0 = 0)
{
    const ColFinishTables F = col_finish_tables(cp);
    int len = 0;
    int ti = F.end_tail, t = n, c = 0, slot = 0;
    // tail states (all in row n)
    for (;;) {
        if (len >= cap - 2) return -2;
        if (lane == 0) rev_put(rev, len, F.tstate[ti]);
        ++len;
        const int loc = F.edges_u[__builtin_amdgcn_readfirstlane(tailwin[ti])].loc;
        if (loc < 0) { ti = -loc - 1; continue; }
        c = loc >> 2;
        slot = loc & 3;
        break;
    }
    int s0 = -1;           // row-0 silent state to continue from
    while (t >= 1) {
        if (len >= cap - 66) return -2;
        if (slot == 1) {
            // diagonal gather: lane i looks at the M cell (t-i, c-i)
            const int tt = t - lane, cc = c - lane;
            const bool valid = tt >= 1 && cc >= 1;
            const int byte = valid ? bp_at(tt, cc, 1) : 0xff;
            const unsigned long long mm = __ballot(valid && tt > 1 && bp_ptr_M(byte) == 1);      // row 1: 1 = entry edge
            // run = number of leading lanes whose pointer is "M of the previous column"; the cell after the run
            // (lane `run`) is an M cell too (reached through an M pointer) unless it is invalid
            const int run = (~mm == 0ull) ? 64 : (__ffsll((long long)~mm) - 1);
            const int cells = min(run + 1, 64);                       // M cells visited, lanes 0..cells-1
            if (lane < cells) rev_put(rev, len + lane, L.state[cc + 1].sM);
            len += cells;
            if (run >= 64) { t -= 64; c -= 64; continue; }            // still on the diagonal: gather again
            // leave through the pointer of the last visited cell (lane `run`)
            const int lastbyte = __builtin_amdgcn_readlane(byte, run);
            int p = bp_ptr_M(lastbyte);
            if (p == 1) p = 2;                                         // only a first-row cell ends a run on pointer 1
            const ColState cs = L.state[c - run + 1];
            t -= run + 1;
            c -= run;
            if (p == 2) { s0 = cs.sX; break; }                        // entry edge from a row-0-only state
            c -= 1;
            slot = p == 3 ? 2 : p;                                     // p == 0 -> I, 3 -> b   (1 cannot occur here)
            continue;
        }
        const ColState cs = L.state[c + 1];
        if (slot == 0) {
            // insert self-loop runs (reads that do not belong to the locus sit in one insert state for most
            // of their length): vertical gather, lane i looks at the I cell (t-i, c)
            const int tt = t - lane;
            const bool valid = tt >= 1;
            const int byte = valid ? bp_at(tt, c, 0) : 0xff;
            const unsigned long long ii = __ballot(valid && bp_ptr_I(byte) == 0);
            const int run = (~ii == 0ull) ? 64 : (__ffsll((long long)~ii) - 1);
            const int cells = min(run + 1, 64);
            if (lane < cells) rev_put(rev, len + lane, cs.sI);
            len += cells;
            if (run >= 64) { t -= 64; continue; }
            slot = bp_ptr_I(__builtin_amdgcn_readlane(byte, run));      // 1 -> M, 2 -> b of the same column
            t -= run + 1;
            continue;
        }
        // b runs (deletions, and the silent columns between repeat units: 11 of the 18 moves of a 150-base read on the bench
        // model, in runs of three or four): horizontal gather, lane i < COL_B_RUN looks at the b cell (t, c-i) -- every cell of
        // it is a line of its own in the back-pointer store, so the gather is kept as short as the runs are.  A fan-in sink
        // column ends a run: its predecessor is the winner the sweep recorded, not a pointer
        {
            constexpr int COL_B_RUN = 8;
            const int cc = c - lane;
            const bool valid = cc >= 0 && lane < COL_B_RUN;
            const unsigned fl = valid ? L.info0[cc + 1].flags : 0u;
            const bool sink = (fl & COL_FLAG_SINK) != 0;
            const int byte = (valid && !sink) ? bp_at(t, cc, 2) : 0xff;
            const unsigned long long bb = __ballot(valid && !sink && bp_ptr_B(byte) == 2) | (~0ull << COL_B_RUN);
            const int run = (~bb == 0ull) ? COL_B_RUN : (__ffsll((long long)~bb) - 1);
            const int cells = min(run + 1, COL_B_RUN);
            if (lane < cells) rev_put(rev, len + lane, L.state[max(cc, 0) + 1].sB);
            len += cells;
            if (run >= COL_B_RUN) { c -= COL_B_RUN; continue; }
            const unsigned flr = (unsigned)__builtin_amdgcn_readlane((int)fl, run);
            if (flr & COL_FLAG_SINK) {
                c = __builtin_amdgcn_readfirstlane(sinkbp[((flr >> 4) & 15) * sink_stride + ((U0 + t - 1) % W) + 1]) - sink_col0;   // fan-in winner
            } else {
                slot = bp_ptr_B(__builtin_amdgcn_readlane(byte, run));         // 0 -> I, 1 -> M of the previous column
                c -= run + 1;
            }
        }
    }
    if (s0 < 0) s0 = L.state[c + 1].sB;                    // arrived in row 0 on the backbone
    while (s0 != start_state) {
        if (len >= cap - 2 || s0 < P) return -2;
        if (lane == 0) rev_put(rev, len, s0);
        ++len;
        s0 = F.pred0[s0 - P];
    }
    if (lane == 0) rev_put(rev, len, start_state);
    ++len;
    return len;
}

template <int K>
__device__ __forceinline__ int col_traceback(const ColProgram *__restrict__ cp, const LdsTables &L, const int n,
                                             const int start_state, const int P, const uint8_t *__restrict__ bp,
                                             const int64_t slab, const int sink_stride,
                                             const int32_t *__restrict__ tailwin, const int32_t *__restrict__ sinkbp,
                                             int32_t *__restrict__ rev, const int cap, const int lane,
                                             const int U0 = 0, const int ring = 1 << 30, const int W = 1 << 30)
{
    constexpr int TPAD = 64 * K;
    // read row tt is stream row U0+tt-1; row tiles of 64K rows, slabs reused modulo `ring`
    auto bp_at = [&](int tt, int cc, int) -> int {
        const int u = U0 + tt - 1;
        const int tile = u / TPAD, lt = u - tile * TPAD + 1;
        return bp[(tile % ring) * slab + (int64_t)(lt + cc - 1) * TPAD + (lt - 1)];
    };
    return col_traceback_walk(cp, L, n, start_state, P, bp_at, sink_stride, tailwin, sinkbp, rev, cap, lane, U0, W);
}

// Copy a model's class tables into LDS and build the padded info table (64*K dummy columns on either side, so a
// lane can index it with c + 64*K without clamping; c runs from 1-64K to 64K+NC-2).  Returns whether the padded
// copy fits.
// PAIR: the padded info records carry the column's emission PAIR offset (row-blocked Viterbi sweep) instead of the two
// emission record addresses
// WRAPCOPY: the 33 records behind the last column repeat columns 0 .. 32 instead of being dummies (back-to-back sweeps of
// viterbi_rows.h: a lane that has done the last column of a read walks straight on into column 0 of the next one)
template <int K, bool PAIR = false, bool WRAPCOPY = false>
__device__ __forceinline__ bool stage_model(const ColProgram *__restrict__ cp, uint8_t *tables, const int lds_tables,
                                            const int lds_level, LdsTables &L, const int tid)
{
    __syncthreads();
    const uint4 *src = (const uint4 *)((const uint8_t *)cp + cp->off_class);
    uint4 *dst = (uint4 *)tables;
    const int staged = lds_level >= 2 ? cp->lds_bytes
                     : (((lds_level == 1 ? cp->off_state : cp->off_info) - cp->off_class + 15) & ~15);
    for (int i = tid; i < staged / 16; i += COL_WAVES * 64) dst[i] = src[i];
    L.classes = (const ColClass *)tables;
    L.emis = (const double *)(tables + (cp->off_emis - cp->off_class));
    L.class_base = lds_addr(L.classes);
    L.emis_base = lds_addr(L.emis);
    L.epair_base = lds_addr(tables + (cp->off_epair - cp->off_class));
    L.epair_sym_stride = (unsigned)cp->n_epair * 16u;
    L.info0 = lds_level >= 1 ? (const ColInfo *)(tables + (cp->off_info - cp->off_class))
                             : (const ColInfo *)((const uint8_t *)cp + cp->off_info);
    L.pinfo = 0;
    L.state = lds_level >= 2 ? (const ColState *)(tables + (cp->off_state - cp->off_class))
                             : (const ColState *)((const uint8_t *)cp + cp->off_state);
    // (the 16-bit address fields need the class and emission tables below 64 KiB of LDS)
    const bool padded = (size_t)staged + (size_t)(cp->n_cols + 128 * K) * sizeof(ColInfo) <= (size_t)lds_tables &&
                        L.epair_base + (unsigned)COL_EPAIR_SYMBOLS * L.epair_sym_stride <= 0x10000u;
    if (padded) {
        __syncthreads();
        uint4 *pinfo = (uint4 *)(tables + staged);
        const int ncol = cp->n_cols;
        const uint16_t *pair_of_col = (const uint16_t *)((const uint8_t *)cp + cp->off_pair_of_col);
        for (int i = tid; i < ncol + 128 * K; i += COL_WAVES * 64) {
            const int c = i - 64 * K;
            int ci = (c >= 0 && c < ncol) ? c + 1 : 0;
            if (WRAPCOPY && c >= ncol && c - ncol <= 32 && c - ncol < ncol) ci = c - ncol + 1;
            const ColInfo inf = L.info0[ci];
            uint4 w;
            const unsigned long long vb = (unsigned long long)__double_as_longlong(inf.v0b);
            w.x = (unsigned)vb;
            w.y = (unsigned)(vb >> 32);
            w.z = (L.class_base + (unsigned)inf.tclass * (unsigned)sizeof(ColClass)) | ((unsigned)inf.flags << 16);
            w.w = (L.emis_base + (unsigned)inf.emM * (COL_EMIS_STRIDE * 8u)) |
                  ((L.emis_base + (unsigned)inf.emI * (COL_EMIS_STRIDE * 8u)) << 16);
            if (PAIR) {
                // row-blocked Viterbi sweep: offset inside a symbol row of the pair table; the two flags the sweep tests every
                // step sit in bytes of their own (a byte compare each): byte 2 = feeder flag | fed sink's index << 1, byte 3 =
                // sink flag
                w.w = (unsigned)pair_of_col[ci] * 16u;
                w.z = (L.class_base + (unsigned)inf.tclass * (unsigned)sizeof(ColClass)) |
                      ((inf.flags & COL_FLAG_FEED) ? (1u | (((unsigned)inf.flags >> 8) & 15u) << 1) << 16 : 0u) |
                      ((inf.flags & COL_FLAG_SINK) ? 1u << 24 : 0u);
            }
            pinfo[i] = w;
        }
        L.pinfo = lds_addr(pinfo);
    }
    __syncthreads();
    return padded;
}

// summary record and (optionally) the path of one read, from the reversed path in `rev`
template <class Rev>
__device__ __forceinline__ void col_emit_outputs(const ColArgs &g, const uint32_t flags, const DevModel &M, const int r,
                                                 const uint8_t *__restrict__ seq, const int n, const Rev &rev,
                                                 const int len_walked, const int lane)
{
    // the reference's path buffer holds n + m entries (hmm.pyx:1953, written without a bound check): a longer path is
    // refused here, the same way by every kernel
    const int len = len_walked > n + M.m ? -2 : len_walked;
    if (g.a.out_summary && !(flags & 4u)) {
        int32_t *out = g.a.out_summary + (int64_t)r * 8;
        if (len > 0) summarize_path(rev, len, M.sclass, seq, n, out, lane);
        else if (lane < 8) out[lane] = (lane == 7) ? len : 0;
    }
    if (g.a.out_path && (flags & 1u)) {
        const int64_t o0 = g.a.out_path_off[r];
        const int cap = (int)(g.a.out_path_off[r + 1] - o0);
        int olen = len;
        if (len > cap) olen = -2;
        if (olen > 0)
            for (int i = lane; i < len; i += 64) g.a.out_path[o0 + i] = rev_get(rev, len - 1 - i);
        if (lane == 0) g.a.out_path_len[r] = olen;
    }
    __builtin_amdgcn_wave_barrier();
}

// Everything after the sweep of one read: tail states, traceback, summaries, outputs.
template <int K>
__device__ __forceinline__ void col_finish_read(const ColArgs &g, const uint32_t flags, const ColProgram *__restrict__ cp,
                                                const LdsTables &L, const DevModel &M, const int r,
                                                const uint8_t *__restrict__ seq, const int n, double *final_row,
                                                const uint8_t *__restrict__ bp, const int64_t slab,
                                                int32_t *__restrict__ tailwin, const int32_t *__restrict__ sinkbp,
                                                int32_t *__restrict__ rev, const int lane, const int U0, const int ring,
                                                const int W)
{
    const int NC = cp->n_cols;
    const double logp = col_tail(cp, final_row, tailwin, NC, lane);
    if (lane == 0) g.a.out_logp[r] = logp;
    int len = 0;
    if (logp != -INFINITY) {
        len = col_traceback<K>(cp, L, n, M.start, M.P, bp, slab, g.sink_stride, tailwin, sinkbp, rev, g.a.path_cap, lane,
                               U0, ring, W);
        len = __builtin_amdgcn_readfirstlane(len);
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    col_emit_outputs(g, flags, M, r, seq, n, rev, len, lane);
}

// LONG = reads longer than 64*K rows, processed in row tiles of 64*K rows (K = 4).
template <int K, bool LONG>
__global__ void __launch_bounds__(COL_WAVES * 64, (K >= 4 ? (LONG ? COL_LONG4_WAVES : 3) : (LONG ? COL_LONG_WAVES : COL_MIN_WAVES_PER_SIMD)))
viterbi_columns_kernel(ColArgs g, uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int TPAD = 64 * K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t gw = (int64_t)blockIdx.x * COL_WAVES + wave;
    int32_t *tile_slot = (int32_t *)lds;                    // first 16 B: dequeued tile index
    uint8_t *tables = lds + 16;
    uint8_t *bp = g.bp + gw * g.bp_stride;
    double *rown = g.rown + gw * g.rown_stride;
    int32_t *aux = g.aux + gw * g.aux_stride;
    int32_t *tailwin = aux, *sinkbp = aux + COL_MAX_TAIL;
    int32_t *rev = g.a.path_scratch + gw * g.a.path_cap;
    int cur_model = -1;
    bool padded = false;
    LdsTables L{};
    const ColProgram *cp = nullptr;
    DevModel M{};

    for (;;) {
        __syncthreads();
        if (tid == 0) *tile_slot = atomicAdd(g.tile_counter, 1);
        __syncthreads();
        const int ti = __builtin_amdgcn_readfirstlane(*tile_slot);    // wave-uniform => scalar loop bounds downstream
        if (ti >= g.n_tiles) break;
        const ColTile tile = g.tiles[ti];
        if (tile.model != cur_model) {                      // (re)stage the model's class tables in LDS
            cur_model = tile.model;
            M = g.a.models[cur_model];
            cp = M.cols;
            padded = stage_model<K>(cp, tables, g.lds_tables, g.lds_level, L, tid);
        }
        const int NC = cp->n_cols;
        const int64_t slab = (int64_t)(TPAD + NC) * TPAD;      // back-pointer bytes per row tile
        for (int j = wave; j < tile.count; j += COL_WAVES) {
            const int r = __builtin_amdgcn_readfirstlane(g.a.order[tile.first + j]);
            const uint8_t *seq = g.a.bases + g.a.read_off[r];
            const int n = __builtin_amdgcn_readfirstlane((int)(g.a.read_off[r + 1] - g.a.read_off[r]));
            TileCtx C;
            C.NC = NC; C.sink_stride = g.sink_stride; C.sinkbp = sinkbp;
            double *final_row = rown;
            if (!LONG) {
                C.n_tile = n; C.row0 = 0; C.bp = bp; C.cap = rown; C.seam = nullptr;
                col_sweep<K, 0>(L, padded, C, seq, lane);
            } else {
                // seam ping-pong: tile i writes its last row to buffer (i+1)&1 and reads buffer i&1
                double *buf[2] = {rown, rown + 3 * (int64_t)NC + COL_MAX_TAIL};
                const int n_tiles = (n + TPAD - 1) / TPAD;
                for (int i = 0; i < n_tiles; ++i) {
                    C.row0 = i * TPAD;
                    C.n_tile = min(TPAD, n - C.row0);
                    C.bp = bp + i * slab;
                    C.cap = buf[(i + 1) & 1];
                    C.seam = buf[i & 1];
                    if (i == 0) col_sweep<K, 0>(L, padded, C, seq, lane);
                    else col_sweep<K, 1>(L, padded, C, seq + i * TPAD, lane);
                    __threadfence_block();
                    __builtin_amdgcn_wave_barrier();
                }
                final_row = buf[n_tiles & 1];
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            col_finish_read<K>(g, flags, cp, L, M, r, seq, n, final_row, bp, slab, tailwin, sinkbp, rev, lane, 0, 1 << 30,
                               1 << 30);
        }
    }
}


