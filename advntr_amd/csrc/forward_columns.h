// forward_columns.h -- sum-product (Model.log_probability) on the column program; split from viterbi_columns.h,
// whose sweep (col_sweep<K, MODE, FWD = true>) it reuses.
#pragma once
#include "viterbi_columns.h"

// ------------------------------------------------------------------------------------------------
// Sum-product (Model.log_probability) on the column program: the same sweep with pair_lse instead of max, no
// back-pointers, no traceback.  Tail states: fold over the emitting-sourced in-edges, fold over the silent-sourced
// ones, lse of the two (hmm.pyx:1446-1480), wave-parallel (the fold order inside each group differs from the
// reference's, so results agree to rounding; tests allow 1e-9 relative, the north star 1e-4).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double col_tail_forward(const ColProgram *__restrict__ cp, double *__restrict__ rown, const int NC,
                                                   const int lane)
{
    const uint8_t *base = (const uint8_t *)cp;
    const int32_t *tptr = (const int32_t *)(base + cp->off_tail_ptr);
    const TailEdge *edges = (const TailEdge *)(base + cp->off_tail_edge);
    double *tailv = rown + 3 * NC;
    double result = -INFINITY;
    for (int i = 0; i < cp->n_tail; ++i) {
        double pe = -INFINITY, ps = -INFINITY;
        for (int e = tptr[i] + lane; e < tptr[i + 1]; e += 64) {
            const TailEdge ed = edges[e];
            if (ed.loc >= 0) {
                const double v = rown[(ed.loc >> 2) * 3 + (ed.loc & 3)] + ed.logp;
                if ((ed.loc & 3) < 2) pe = lse2(pe, v); else ps = lse2(ps, v);
            } else {
                ps = lse2(ps, tailv[-ed.loc - 1] + ed.logp);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            pe = lse2(pe, __shfl_xor(pe, o, 64));
            ps = lse2(ps, __shfl_xor(ps, o, 64));
        }
        const double v = lse2(pe, ps);
        if (lane == 0) tailv[i] = v;
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        if (i == cp->end_tail) result = v;
    }
    return result;
}

// The last row of a tile was captured as probabilities * 16^rows * exp(-seam_off): back to the log domain.
__device__ __forceinline__ void col_row_to_log(double *__restrict__ row, const int NC, const int rows, const double seam_off,
                                               const int lane)
{
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    const double back = seam_off - (double)rows * 2.772588722239781;          // rows * log(16)
    for (int q = lane; q < 3 * NC; q += 64) row[q] = log(row[q]) + back;
}

template <int K, bool LONG>
__global__ void __launch_bounds__(COL_WAVES * 64, (K >= 4 ? (LONG ? COL_LONG4_WAVES : 3) : (LONG ? COL_LONG_WAVES : COL_MIN_WAVES_PER_SIMD)))
forward_columns_kernel(ColArgs g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int TPAD = 64 * K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t gw = (int64_t)blockIdx.x * COL_WAVES + wave;
    int32_t *tile_slot = (int32_t *)lds;
    uint8_t *tables = lds + 16;
    double *rown = g.rown + gw * g.rown_stride;
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
        if (tile.model != cur_model) {
            cur_model = tile.model;
            M = g.a.models[cur_model];
            cp = M.cols;
            padded = stage_model<K>(cp, tables, g.lds_tables, g.lds_level, L, tid);
            // to the linear domain, in place: transition classes -> probabilities, emission records -> probability times
            // the per-row scale 16 (an exact power of two; with it a row tile of <= 192 rows neither overflows, (0.97 *
            // 16)^192 = 1e228, nor underflows on any read, (0.25 * 0.02 * 16)^192 = 2e-211), and the row-0 / entry terms
            // of every column into an LDS table of their own behind the staged tables
            double *cls = (double *)L.classes, *em = (double *)L.emis;
            for (int i = tid; i < cp->n_tclass * (int)(sizeof(ColClass) / 8); i += COL_WAVES * 64) cls[i] = exp(cls[i]);
            for (int i = tid; i < cp->n_eclass * COL_EMIS_STRIDE; i += COL_WAVES * 64) em[i] = exp(em[i]) * 16.0;
            double *lin = (double *)(tables + ((g.lds_tables + 15) & ~15));
            const double *fw = (const double *)((const uint8_t *)cp + cp->off_fwd);
            for (int i = tid; i < 2 * cp->n_cols; i += COL_WAVES * 64) lin[i] = exp(fw[i]);
            L.fwd_lin = lds_addr(lin);
            __syncthreads();
        }
        const int NC = cp->n_cols;
        for (int j = wave; j < tile.count; j += COL_WAVES) {
            const int r = __builtin_amdgcn_readfirstlane(g.a.order[tile.first + j]);
            const uint8_t *seq = g.a.bases + g.a.read_off[r];
            const int n = __builtin_amdgcn_readfirstlane((int)(g.a.read_off[r + 1] - g.a.read_off[r]));
            TileCtx C;
            C.NC = NC; C.sink_stride = 0; C.sinkbp = nullptr; C.bp = nullptr;
            C.fwd = (const double *)((const uint8_t *)cp + cp->off_fwd);
            C.seam_off = 0.0;
            double *final_row = rown;
            if (!LONG) {
                C.n_tile = n; C.row0 = 0; C.cap = rown; C.seam = nullptr;
                col_sweep<K, 0, true>(L, padded, C, seq, lane);
                col_row_to_log(C.cap, NC, n, 0.0, lane);
            } else {
                double *buf[2] = {rown, rown + 3 * (int64_t)NC + COL_MAX_TAIL};
                const int n_tiles = (n + TPAD - 1) / TPAD;
                for (int i = 0; i < n_tiles; ++i) {
                    C.row0 = i * TPAD;
                    C.n_tile = min(TPAD, n - C.row0);
                    C.cap = buf[(i + 1) & 1];
                    C.seam = buf[i & 1];
                    C.seam_off = 0.0;
                    if (i > 0) {                       // renormalise: the tile works relative to its seam row's maximum
                        double mx = -INFINITY;
                        for (int q = lane; q < 3 * NC; q += 64) mx = fmax(mx, C.seam[q]);
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
                        C.seam_off = mx == -INFINITY ? 0.0 : mx;
                    }
                    if (i == 0) col_sweep<K, 0, true>(L, padded, C, seq, lane);
                    else col_sweep<K, 1, true>(L, padded, C, seq + i * TPAD, lane);
                    col_row_to_log(C.cap, NC, C.n_tile, C.seam_off, lane);
                    __threadfence_block();
                    __builtin_amdgcn_wave_barrier();
                }
                final_row = buf[n_tiles & 1];
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            const double logp = col_tail_forward(cp, final_row, NC, lane);
            if (lane == 0) g.a.out_logp[r] = logp;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

