// viterbi_generic.h -- generic-CSR Viterbi kernel: any baked HMM, one read per wavefront.
//
// Restates HiddenMarkovModel._viterbi (/root/reference/pomegranate/hmm.pyx:1970-2136) for a 64-lane
// wavefront.  Two trellis rows live in LDS (fp64, 16 B per state); emitting states are strided over the
// lanes; silent states are walked 64 at a time: a parallel "A" pass over sources outside the chunk and
// a register-only serial "B" pass (v_readlane broadcast of each finished state) for sources inside it,
// which is what makes the hundreds-long same-row delete chains of a profile HMM cheap.  Ties resolve as
// in the reference (first maximum in its evaluation order) through the per-edge ordinal.
// Back-pointers: one ordinal per trellis cell (1 byte, 2 when a fan-in exceeds 255), written coalesced
// to a per-workgroup scratch slab in HBM; lane 0 walks them back, then the wave summarises the path.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "device_model.h"
#include "path_summary.h"

struct BatchArgs {
    const DevModel *models;
    const uint8_t *bases;
    const int64_t *read_off;
    const int32_t *read_model;
    int32_t n_reads;
    double *out_logp;
    int32_t *out_summary;        // n_reads * 8 or nullptr
    int32_t *out_path;           // or nullptr
    const int64_t *out_path_off; // n_reads+1 (capacities)
    int32_t *out_path_len;
    uint8_t *bp_scratch;         // grid * bp_stride bytes
    int64_t bp_stride;
    int32_t *path_scratch;       // grid * path_cap ints
    int32_t path_cap;
    int32_t m_max;               // LDS row pitch (states)
    const int32_t *order;        // read processing order (heaviest first) or nullptr
    int32_t *counter;            // device-wide dequeue head, zeroed before every launch
};

// One returning atomicAdd per read: dynamic dequeue keeps the 256 CUs busy when reads/models differ in
// cost (MI355X_MICROARCH "dequeue" row: ~0.25-1.1 us, noise next to a >100 us read).
__device__ __forceinline__ int next_read(const BatchArgs &a, int lane)
{
    int it = 0;
    if (lane == 0) it = atomicAdd(a.counter, 1);
    return __builtin_amdgcn_readfirstlane(it);
}

__device__ __forceinline__ double readlane_f64(double v, int j)
{
    // j is wave-uniform: two v_readlane_b32 into SGPRs
    int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void bp_store(uint8_t *row, int bpw, int l, int ord)
{
    if (bpw == 1) row[l] = (uint8_t)ord;
    else ((uint16_t *)row)[l] = (uint16_t)ord;
}

__device__ __forceinline__ int bp_load(const uint8_t *row, int bpw, int l)
{
    return bpw == 1 ? (int)row[l] : (int)((const uint16_t *)row)[l];
}

// Silent states of row t.  cur[] holds this row's emitting values already.
__device__ __forceinline__ void silent_pass(const DevModel &M, double *cur, uint8_t *bprow, int t, int lane)
{
    const int P = M.P, m = M.m;
    for (int c = 0; c < M.n_chunks; ++c) {
        const int base = P + c * ADV_WAVE;
        const int l = base + lane;
        const bool active = l < m;
        double best = -INFINITY;
        int bord = 0;
        int qb = 0, qe = 0;             // B list cursor
        int nsrc = -1, nord = 0;
        double nlp = 0.0;
        const bool fixed_start = (t == 0 && l == M.start);   // hmm.pyx:2006-2008
        if (active) {
            const int ls = l - P;
            const int qa = M.s_ptr[ls];
            qb = M.s_mid[ls];
            qe = M.s_ptr[ls + 1];
            if (fixed_start) {
                best = 0.0;
                qb = qe;
            } else {
                for (int q = qa; q < qb; ++q) {                         // list order: strict '>' suffices
                    const double cand = cur[M.s_src[q]] + M.s_logp[q];
                    if (cand > best) { best = cand; bord = M.s_ord[q]; }
                }
            }
            if (qb < qe) { nsrc = M.s_src[qb]; nlp = M.s_logp[qb]; nord = M.s_ord[qb]; }
        }
        // serial part: state base+j is final once sources base..base+j-1 have been broadcast
        const int jn = __builtin_amdgcn_readfirstlane(min(ADV_WAVE, m - base) - 1);
        const unsigned long long anyB = __ballot(nsrc >= 0);
        if (anyB) {
            for (int j = 0; j < jn; ++j) {
                const double vj = readlane_f64(best, j);
                if (nsrc == base + j) {
                    const double cand = vj + nlp;
                    if (cand > best || (cand == best && nord < bord)) { best = cand; bord = nord; }
                    ++qb;
                    if (qb < qe) { nsrc = M.s_src[qb]; nlp = M.s_logp[qb]; nord = M.s_ord[qb]; }
                    else nsrc = -1;
                }
            }
        }
        if (active) {
            cur[l] = best;
            bp_store(bprow, M.bp_width, l, bord);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(ADV_WAVE) viterbi_generic_kernel(BatchArgs a, uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) double lds_rows[];
    const int lane = threadIdx.x;
    uint8_t *bp = a.bp_scratch + (int64_t)blockIdx.x * a.bp_stride;
    int32_t *rev = a.path_scratch + (int64_t)blockIdx.x * a.path_cap;

    for (;;) {
        const int it = next_read(a, lane);
        if (it >= a.n_reads) break;
        const int r = __builtin_amdgcn_readfirstlane(a.order ? a.order[it] : it);
        const DevModel M = a.models[__builtin_amdgcn_readfirstlane(a.read_model[r])];
        const uint8_t *seq = a.bases + a.read_off[r];
        const int n = __builtin_amdgcn_readfirstlane((int)(a.read_off[r + 1] - a.read_off[r]));
        const int m = M.m, P = M.P, bpw = M.bp_width;
        const int64_t row_bytes = (int64_t)m * bpw;
        double *prev = lds_rows, *cur = lds_rows + a.m_max;

        // ---- row 0: only the silent closure of the start state is alive (hmm.pyx:1999-2023)
        for (int l = lane; l < P; l += ADV_WAVE) cur[l] = -INFINITY;
        __syncthreads();
        silent_pass(M, cur, bp, 0, lane);

        for (int t = 1; t <= n; ++t) {
            double *tmp = prev; prev = cur; cur = tmp;
            const int x = seq[t - 1];
            uint8_t *bprow = bp + (int64_t)t * row_bytes;
            // ---- emitting states (hmm.pyx:2026-2042): (v + t) + e, strict '>'
            for (int l = lane; l < P; l += ADV_WAVE) {
                const double e = M.emis[4 * l + x];
                const int k0 = M.e_ptr[l], k1 = M.e_ptr[l + 1];
                double best = -INFINITY;
                int bord = 0;
                for (int k = k0; k < k1; ++k) {
                    const double cand = prev[M.e_src[k]] + M.e_logp[k] + e;
                    if (cand > best) { best = cand; bord = k - k0; }
                }
                cur[l] = best;
                bp_store(bprow, bpw, l, bord);
            }
            __syncthreads();
            silent_pass(M, cur, bprow, t, lane);
        }

        // ---- final score (hmm.pyx:2089-2098)
        int end_state = M.end;
        double logp;
        if (M.finite) {
            logp = cur[M.end];
        } else {
            // first maximum over all states in index order
            double bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int l = lane; l < m; l += ADV_WAVE) {
                const double v = cur[l];
                if (v > bv) { bv = v; bi = l; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            logp = bv;
            end_state = (bv == -INFINITY) ? -1 : bi;
        }
        if (lane == 0) a.out_logp[r] = logp;

        // ---- traceback (hmm.pyx:2109-2130), lane 0; the slab was written by this wave
        __threadfence_block();
        __syncthreads();
        int len = 0;
        if (logp != -INFINITY) {
            if (lane == 0) {
                int px = n, py = end_state;
                bool overflow = false;
                while (px != 0 || py != M.start) {
                    if (len >= a.path_cap - 1) { overflow = true; break; }
                    rev[len++] = py;
                    const int ord = bp_load(bp + (int64_t)px * row_bytes, bpw, py);
                    if (py < P) { py = M.e_src[M.e_ptr[py] + ord]; px -= 1; }
                    else py = M.r_src[M.r_ptr[py - P] + ord];
                }
                if (overflow) len = -2;
                else rev[len++] = py;
            }
            len = __shfl(len, 0, 64);
            if (len > n + M.m) len = -2;          // beyond the reference's own path buffer (hmm.pyx:1953): refused by every kernel
        }
        __threadfence_block();
        __syncthreads();

        if (a.out_summary && !(flags & 4u)) {
            int32_t *out = a.out_summary + (int64_t)r * 8;
            if (len > 0) summarize_path(rev, len, M.sclass, seq, n, out, lane);
            else if (lane < 8) out[lane] = (lane == 7) ? len : 0;
        }
        if (a.out_path && (flags & 1u)) {
            const int64_t o0 = a.out_path_off[r];
            const int cap = (int)(a.out_path_off[r + 1] - o0);
            int olen = len;
            if (len > cap) olen = -2;
            if (olen > 0)
                for (int i = lane; i < len; i += ADV_WAVE) a.out_path[o0 + i] = rev[len - 1 - i];
            if (lane == 0) a.out_path_len[r] = olen;
        }
        __syncthreads();
    }
}
