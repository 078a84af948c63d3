// viterbi_columns_stream.h -- the experimental stream-packed variant of the column kernel (ADVNTR_FLAG_STREAM): parity-
// tested and used as a third implementation by the differential fuzz, slower than the bucketed kernel, not the default.
#pragma once
#include "viterbi_columns.h"

// ------------------------------------------------------------------------------------------------
// Stream kernel: a wavefront packs its reads back to back along the row axis and sweeps the stream in row
// tiles of 64*K rows, so every lane carries a useful row (a 150-base read otherwise fills 150 of 192 lanes).
// Read boundaries inside a tile are per-lane flags (first row: previous row := the model's row 0, entry edges
// live; last row: captured for the tail states); a read that straddles two tiles continues through the seam
// row exactly like a long read.  Back-pointer slabs form a ring so a straddling read can still be traced back.
// ------------------------------------------------------------------------------------------------
#ifndef COL_STREAM_WAVES_PER_SIMD
#define COL_STREAM_WAVES_PER_SIMD 4
#endif
#define COL_STREAM_K 3           // chunks per row tile of the stream kernel (192 rows)
#define COL_STREAM_READS 16      // reads per wavefront per tile of work
#define COL_STREAM_CAPS 4        // reads that may END inside one row tile (capture buffers)

struct StreamRead {
    int32_t r, n, U, pad;        // read index, length, stream row of its first base
};

template <int K>
__global__ void __launch_bounds__(COL_WAVES * 64, COL_STREAM_WAVES_PER_SIMD)
viterbi_columns_stream_kernel(ColArgs g, uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int TPAD = 64 * K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t gw = (int64_t)blockIdx.x * COL_WAVES + wave;
    int32_t *tile_slot = (int32_t *)lds;
    uint8_t *tables = lds + 16;
    StreamRead *mine = (StreamRead *)(lds + 16 + g.lds_tables) + wave * COL_STREAM_READS;
    uint8_t *bp = g.bp + gw * g.bp_stride;
    double *rbase = g.rown + gw * g.rown_stride;
    int32_t *aux = g.aux + gw * g.aux_stride;
    int32_t *tailwin = aux, *sinkbp = aux + COL_MAX_TAIL;
    int32_t *rev = g.a.path_scratch + gw * g.a.path_cap;
    const int ring = g.ring, W = g.ring * TPAD;
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
        }
        const int NC = cp->n_cols;
        const int64_t slab = (int64_t)(TPAD + NC) * TPAD;
        const int64_t cap_stride = 3 * (int64_t)NC + COL_MAX_TAIL;
        double *seam_buf[2] = {rbase, rbase + 3 * (int64_t)NC};
        double *capbuf = rbase + 6 * (int64_t)NC;

        // ---- lay this wave's reads out along the stream (wave-uniform arithmetic, lane 0 writes LDS)
        int nr = 0, U = 0, last_te = -1, cnt = 0;
        for (int j = wave; j < tile.count && nr < COL_STREAM_READS; j += COL_WAVES) {
            const int r = __builtin_amdgcn_readfirstlane(g.a.order[tile.first + j]);
            const int n = __builtin_amdgcn_readfirstlane((int)(g.a.read_off[r + 1] - g.a.read_off[r]));
            int te = (U + n - 1) / TPAD;
            if (te == last_te && cnt >= COL_STREAM_CAPS) {          // too many reads would end in that tile
                U = (te + 1) * TPAD;
                te = (U + n - 1) / TPAD;
            }
            if (te != last_te) { last_te = te; cnt = 0; }
            ++cnt;
            if (lane == 0) mine[nr] = StreamRead{r, n, U, 0};
            U += n;
            ++nr;
        }
        const int Utot = U;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();

        const int n_tiles = (Utot + TPAD - 1) / TPAD;
        for (int i = 0; i < n_tiles; ++i) {
            const int u0 = i * TPAD;
            int slot_x[K], slot_flag[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { slot_x[k] = 0; slot_flag[k] = 0; }
            int ends[COL_STREAM_CAPS];
#pragma unroll
            for (int e = 0; e < COL_STREAM_CAPS; ++e) ends[e] = -1;
            int ncap = 0;
            for (int q = 0; q < nr; ++q) {
                const StreamRead rd = mine[q];
                const int rU = __builtin_amdgcn_readfirstlane(rd.U), rn = __builtin_amdgcn_readfirstlane(rd.n);
                const int e = rU + rn - 1;
                if (e < u0 || rU >= u0 + TPAD) continue;
                const bool ends_here = e < u0 + TPAD;
                const uint8_t *seq = g.a.bases + g.a.read_off[__builtin_amdgcn_readfirstlane(rd.r)];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int u = u0 + 64 * k + lane;
                    if (u >= rU && u <= e) {
                        slot_x[k] = seq[u - rU];
                        slot_flag[k] = (u == rU ? 1 : 0) | (u == e ? 2 : 0) | (ncap << 8);
                    }
                }
                if (ends_here) {
#pragma unroll
                    for (int c2 = 0; c2 < COL_STREAM_CAPS; ++c2)
                        if (c2 == ncap) ends[c2] = q;
                    ++ncap;
                }
            }
            unsigned hasfirst = 0, haslast = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (__ballot(slot_flag[k] & 1)) hasfirst |= 1u << k;
                if (__ballot(slot_flag[k] & 2)) haslast |= 1u << k;
            }
            TileCtx C;
            C.NC = NC; C.sink_stride = g.sink_stride; C.sinkbp = sinkbp;
            C.n_tile = min(TPAD, Utot - u0);
            C.row0 = u0 % W;
            C.bp = bp + (i % ring) * slab;
            C.cap = capbuf; C.cap_stride = cap_stride;
            C.seam = seam_buf[i & 1]; C.seam_out = seam_buf[(i + 1) & 1];
            C.hasfirst = hasfirst; C.haslast = haslast;
            col_sweep<K, 2>(L, padded, C, nullptr, lane, slot_x, slot_flag);
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < COL_STREAM_CAPS; ++e) {
                if (ends[e] < 0) continue;
                const StreamRead rd = mine[ends[e]];
                const int r = __builtin_amdgcn_readfirstlane(rd.r), n = __builtin_amdgcn_readfirstlane(rd.n);
                const uint8_t *seq = g.a.bases + g.a.read_off[r];
                col_finish_read<K>(g, flags, cp, L, M, r, seq, n, capbuf + e * cap_stride, bp, slab, tailwin, sinkbp, rev,
                                   lane, __builtin_amdgcn_readfirstlane(rd.U), ring, W);
            }
        }
    }
}

