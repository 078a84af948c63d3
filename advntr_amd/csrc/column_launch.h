// column_launch.h -- host side of the column-kernel launches (bucketed, stream, forward).
#pragma once
#include "viterbi_columns.h"
#include "forward_columns.h"
#include "viterbi_columns_stream.h"
#include "viterbi_rows.h"
#include "forward_rows.h"

// ------------------------------------------------------------------------------------------------
// host side of the launch
// ------------------------------------------------------------------------------------------------
struct ColumnLaunch {
    int grid = 0;
    int waves_per_block = COL_WAVES;
    int nc_max = 0;
    int sink_stride = COL_MAX_READ + 1;
    size_t lds_bytes = 0;
    size_t lds_core_bytes = 0, lds_min_bytes = 0;     // footprints of staging levels 1 and 0 (batch_build)
    int lds_level = 2;
    int64_t bp_stride = 0, rown_stride = 0, aux_stride = 0;
    bool stream = false;                    // all column reads go through the stream kernel (tiles[0])
    int ring = 2;
    int rows_depth = 1;                     // reads per lane group of the deepest row-blocked tile
    double useful_cells[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // per tile list (row-blocked kernels only): trellis cells of the reads,
    double swept_cells[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};    // and cell slots the sweeps' lane-steps offer (advntr_batch_info)
    int reserve_workgroups = 0;             // resident workgroup slots the NEXT pass leaves unclaimed (a multi-GPU run's result
                                            // gather runs beside it: abi_comm.h); consumed by that pass (advntr_batch_run)
    std::vector<ColTile> tiles[9];          // per chunk count K = 1..4, [4] = row-tiled long reads, [5..7] = row-blocked kernels,
                                            // [8] = row-blocked kernel for reads of more than 155 bases (row tiles)
    ColTile *d_tiles[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int32_t *d_tile_counters = nullptr;     // 9 counters
    double *d_rown = nullptr;
    int32_t *d_aux = nullptr;
    uint8_t *d_bp = nullptr;
};

// Workgroups of one launch: the resident set, minus the slots a multi-GPU run's gather asked the NEXT pass to leave free
// (abi_comm.h) -- honoured only by a launch that would otherwise fill the device (a small grid leaves room by itself, and
// must not be cut down to a single workgroup), and never more than 1/16 of the grid.
static inline int launch_grid(const ColumnLaunch &cl, int n_tiles)
{
    int reserve = 0;
    if (cl.reserve_workgroups > 0 && n_tiles >= cl.grid && cl.grid >= 64) reserve = std::min(cl.reserve_workgroups, cl.grid / 16);
    return std::min(std::max(1, cl.grid - reserve), n_tiles);
}

template <int K, bool LONG>
static inline void column_launch_k(const ColumnLaunch &cl, const BatchArgs &a, uint32_t flags, hipStream_t stream)
{
    const int slot = LONG ? 4 : K - 1;
    if (cl.tiles[slot].empty()) return;
    ColArgs g{};
    g.a = a;
    g.tiles = cl.d_tiles[slot];
    g.n_tiles = (int32_t)cl.tiles[slot].size();
    g.tile_counter = cl.d_tile_counters + slot;
    g.rown = cl.d_rown; g.rown_stride = cl.rown_stride;
    g.aux = cl.d_aux; g.aux_stride = cl.aux_stride;
    g.bp = cl.d_bp; g.bp_stride = cl.bp_stride;
    g.lds_tables = (int32_t)cl.lds_bytes;
    g.lds_level = cl.lds_level;
    g.sink_stride = cl.sink_stride;
    const int grid = launch_grid(cl, g.n_tiles);
    if (cl.lds_bytes + 16 > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)viterbi_columns_kernel<K, LONG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(cl.lds_bytes + 16));
    hipLaunchKernelGGL((viterbi_columns_kernel<K, LONG>), dim3(grid), dim3(COL_WAVES * 64), cl.lds_bytes + 16, stream, g, flags);
}

// short reads, G per wavefront (viterbi_rows.h); cfg = index into rows_configs, tile list 5 + cfg
template <int R, int G>
static inline void column_launch_rows(const ColumnLaunch &cl, const BatchArgs &a, uint32_t flags, hipStream_t stream, const int cfg)
{
    const int slot = 5 + cfg;
    if (cl.tiles[slot].empty()) return;
    ColArgs g{};
    g.a = a;
    g.tiles = cl.d_tiles[slot];
    g.n_tiles = (int32_t)cl.tiles[slot].size();
    g.tile_counter = cl.d_tile_counters + slot;
    g.rown = cl.d_rown; g.rown_stride = cl.rown_stride;
    g.aux = cl.d_aux; g.aux_stride = cl.aux_stride;
    g.bp = cl.d_bp; g.bp_stride = cl.bp_stride;
    g.lds_tables = (int32_t)cl.lds_bytes;
    g.lds_level = cl.lds_level;
    g.sink_stride = cl.sink_stride;
    g.rows_depth = cl.rows_depth;
    const int grid = launch_grid(cl, g.n_tiles);
    const size_t lds = cl.lds_bytes + 16 + ROWS_STASH_BYTES + ROWS_REV_BYTES + ROWS_TAIL_LDS_BYTES;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)viterbi_rows_kernel<R, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((viterbi_rows_kernel<R, G>), dim3(grid), dim3(COL_WAVES * 64), lds, stream, g, flags);
}

// reads of more than 155 bases, one per wavefront, row tiles of 64 * ROWS_LONG_R rows (viterbi_rows.h); tile list 8
static inline void column_launch_rows_long(const ColumnLaunch &cl, const BatchArgs &a, uint32_t flags, hipStream_t stream)
{
    const int slot = 8;
    if (cl.tiles[slot].empty()) return;
    ColArgs g{};
    g.a = a;
    g.tiles = cl.d_tiles[slot];
    g.n_tiles = (int32_t)cl.tiles[slot].size();
    g.tile_counter = cl.d_tile_counters + slot;
    g.rown = cl.d_rown; g.rown_stride = cl.rown_stride;
    g.aux = cl.d_aux; g.aux_stride = cl.aux_stride;
    g.bp = cl.d_bp; g.bp_stride = cl.bp_stride;
    g.lds_tables = (int32_t)cl.lds_bytes;
    g.lds_level = cl.lds_level;
    g.sink_stride = cl.sink_stride;
    const int grid = launch_grid(cl, g.n_tiles);
    if (cl.lds_bytes + 16 > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)viterbi_rows_long_kernel<ROWS_LONG_R>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(cl.lds_bytes + 16));
    hipLaunchKernelGGL((viterbi_rows_long_kernel<ROWS_LONG_R>), dim3(grid), dim3(COL_WAVES * 64), cl.lds_bytes + 16, stream, g, flags);
}

template <int K>
static inline void column_launch_stream(const ColumnLaunch &cl, const BatchArgs &a, uint32_t flags, hipStream_t stream)
{
    if (cl.tiles[0].empty()) return;
    ColArgs g{};
    g.a = a;
    g.tiles = cl.d_tiles[0];
    g.n_tiles = (int32_t)cl.tiles[0].size();
    g.tile_counter = cl.d_tile_counters;
    g.rown = cl.d_rown; g.rown_stride = cl.rown_stride;
    g.aux = cl.d_aux; g.aux_stride = cl.aux_stride;
    g.bp = cl.d_bp; g.bp_stride = cl.bp_stride;
    g.lds_tables = (int32_t)cl.lds_bytes;
    g.lds_level = cl.lds_level;
    g.sink_stride = cl.sink_stride;
    g.ring = cl.ring;
    const int grid = launch_grid(cl, g.n_tiles);
    const size_t lds = cl.lds_bytes + 16 + COL_WAVES * COL_STREAM_READS * sizeof(StreamRead);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)viterbi_columns_stream_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((viterbi_columns_stream_kernel<K>), dim3(grid), dim3(COL_WAVES * 64), lds, stream, g, flags);
}

// LDS bytes of a sum-product launch whose tables take `tables` bytes: + tile slot + the linear row-0 table
static inline size_t forward_lds_bytes(const size_t tables, const int nc_max)
{
    return ((tables + 15) & ~size_t(15)) + 16 + 16 * ((size_t)nc_max + 128) + 16;     // (row-blocked kernels: padded like the info table)
}

// slot: which tile list (0..3 = reads of 1..4 chunks, 4 = long reads).  The sum-product sweep works in the linear domain
// with a per-row scale of 16, which bounds a row tile to 192 rows: reads of 193-256 rows (slot 3) go through the
// row-tiled kernel like the long ones.
template <int K, bool LONG>
static inline hipError_t column_launch_fwd(const ColumnLaunch &cl, const BatchArgs &a, hipStream_t stream, const int slot)
{
    if (cl.tiles[slot].empty()) return hipSuccess;
    ColArgs g{};
    g.a = a;
    g.tiles = cl.d_tiles[slot];
    g.n_tiles = (int32_t)cl.tiles[slot].size();
    g.tile_counter = cl.d_tile_counters + slot;
    g.rown = cl.d_rown; g.rown_stride = cl.rown_stride;
    g.lds_tables = (int32_t)cl.lds_bytes;
    g.lds_level = cl.lds_level;
    const int grid = launch_grid(cl, g.n_tiles);
    const size_t lds = forward_lds_bytes(cl.lds_bytes, cl.nc_max);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void *)forward_columns_kernel<K, LONG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL((forward_columns_kernel<K, LONG>), dim3(grid), dim3(COL_WAVES * 64), lds, stream, g);
    return hipSuccess;
}

// forward_rows_kernel keeps, behind the row-0 table, the tail states' in-edge weights in the linear domain (two fan-in states
// with an edge from every column, a few more): 16 bytes per column and then some -- when that still fits a compute unit
static inline int forward_rows_tailw_cap(const size_t tables, const int nc_max)
{
    const size_t want = 2 * ((size_t)nc_max + 64);
    return forward_lds_bytes(tables, nc_max) + 8 * want <= 160 * 1024 ? (int)want : 0;
}

// sum-product on the row-blocked layout: the short reads of a large batch (tile list 5 + cfg)
template <int R, int G>
static inline hipError_t column_launch_fwd_rows(const ColumnLaunch &cl, const BatchArgs &a, hipStream_t stream, const int cfg)
{
    const int slot = 5 + cfg;
    if (cl.tiles[slot].empty()) return hipSuccess;
    ColArgs g{};
    g.a = a;
    g.tiles = cl.d_tiles[slot];
    g.n_tiles = (int32_t)cl.tiles[slot].size();
    g.tile_counter = cl.d_tile_counters + slot;
    g.rown = cl.d_rown; g.rown_stride = cl.rown_stride;
    g.lds_tables = (int32_t)cl.lds_bytes;
    g.lds_level = cl.lds_level;
    g.rows_depth = cl.rows_depth;
    g.fwd_tailw_cap = forward_rows_tailw_cap(cl.lds_bytes, cl.nc_max);
    const int grid = launch_grid(cl, g.n_tiles);
    const size_t lds = forward_lds_bytes(cl.lds_bytes, cl.nc_max) + 8 * (size_t)g.fwd_tailw_cap;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void *)forward_rows_kernel<R, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL((forward_rows_kernel<R, G>), dim3(grid), dim3(COL_WAVES * 64), lds, stream, g);
    return hipSuccess;
}
