// column_program.h -- host-side compiler from a baked CSR model to the "column program" that the
// anti-diagonal Viterbi kernel (viterbi_columns.h) executes.
//
// A flank-repeats-flank read matcher (/root/reference/advntr/hmm_utils.py:553-595) is, after bake(),
// one long chain: its silent states (delete states and connectors) are in topological index order and
// every emitting state hangs off that chain.  Column c = silent "backbone" state b_c plus at most one
// insert-like state I_c (self loop) and one match-like state M_c, with the profile-HMM stencil
//     I_c(t) = e + max[ I_c(t-1), M_c(t-1), b_c(t-1) ]                       (previous row, same column)
//     M_c(t) = e + max[ I_{c-1}(t-1), M_{c-1}(t-1), X, b_{c-1}(t-1) ]        (previous row, previous column;
//                                                   X = an entry edge from a row-0-only silent state)
//     b_c(t) =     max[ I_{c-1}(t), M_{c-1}(t), b_{c-1}(t) ]                 (same row, previous column)
// plus "feed/sink" pairs for silent fan-in across columns (end_repeating_pattern_match <- every
// unit_end_k) and a short "tail" of silent states that only matter in the last row (prefix_end_prefix,
// the model end: fan-in from every match state) which the kernel evaluates from a row-n buffer.
// Cell (t,c) depends only on anti-diagonals t+c-1 and t+c-2, so 64 rows advance in lock step.
//
// Nothing here trusts names: the layout is derived from the graph, and EVERY in-edge of every state must
// be claimed by the stencil in the reference's evaluation order (strict '>' => first maximum wins,
// hmm.pyx:2039,2060,2080); otherwise the model simply has no column program and runs on the generic
// kernel.  Transition/emission parameters are de-duplicated bit-exactly into small class tables so the
// whole program of a 1413-state model is ~20 KB of LDS.
#pragma once
#include <stdint.h>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define COL_FLAG_FEED 1u      // b_c feeds the running accumulator of the next sink
#define COL_FLAG_SINK 2u      // b_c := accumulator (fan-in from earlier backbone states)
#define COL_MAX_SINKS 4
#define COL_MAX_READ 256      // rows handled in registers: 4 chunks of 64 lanes (one row tile)
#define COL_MAX_LONG_READ (1 << 17)  // longer reads are row-tiled: 256 rows per tile with a seam row in HBM.  The bound is a
                                     // constant, not a structure: 131 072 bases is ten times what the reference can hand
                                     // over (reads are trimmed to the VNTR + 100-base flanks, VNTRs are capped at 10 kb:
                                     // vntr_finder.py:362,415, models.py:164,174) and what the GPU suite covers (100 000)

struct ColClass {             // 12 doubles = 96 B: six 16-byte LDS words.  (Rounds 1-3: 11 doubles = 88 B, an odd multiple of 8 B that
                              // kept the 8-byte reads of the anti-diagonal kernel off each other's banks; the row-blocked sweeps read
                              // the whole record every step, and as six ds_read_b128 -- 4 LDS cycles each -- instead of five
                              // ds_read2_b64 + one ds_read_b64 -- 8 each -- they run 2.6 % (Viterbi, REF150) and 2.9 % (sum-product)
                              // faster: HISTORY.md, round 4)
    double iI, iM, iD, mI;    // I_c <- I_c, M_c, b_c (prev row);   M_c <- I_{c-1}
    double mM, mX, mD, dI;    // M_c <- M_{c-1}, X (row 0 only, value incl. source), b_{c-1};  b_c <- I_{c-1}
    double dM, dD, erw;       // b_c <- M_{c-1}, b_{c-1};  feed weight
    double pad16;             // (112-byte records -- 7 slots, coprime with the 16 slots of an LDS line -- measured: no difference)
};
#define COL_EMIS_STRIDE 5     // doubles per emission class (4 used): 40-B records, bank-conflict free below 32 classes

struct ColInfo {              // 16 B per column (with one dummy column at either end)
    double v0b;               // row-0 value of b_c (read independent)
    uint16_t tclass, emM, emI, flags;   // flags: COL_FLAG_* | sink index << 4 (sink column) | fed sink's index << 8 (feeder)
};

struct ColState {             // 16 B per column: state indices for the traceback
    int32_t sI, sM, sB, sX;
};

struct TailEdge {             // in-edge of a tail state, reference evaluation order
    int32_t loc;              // >= 0: column*4 + slot (0 I, 1 M, 2 b);  < 0: -(tail index)-1
    int32_t src_state;
    double logp;
};

// device header; arrays follow at the byte offsets
struct ColProgram {
    int32_t n_cols, n_tclass, n_eclass, n_tail, n_sinks, end_tail, m, P;
    int32_t off_class, off_emis, off_info, off_state, off_pred0, off_tail_ptr, off_tail_state, off_tail_edge;
    int32_t off_v0, lds_bytes, off_fwd, pad1;     // off_fwd: n_cols x {fwd row-0 value of b_c, fwd entry term of M_c}
    int32_t off_epair, n_epair, off_pair_of_col, pad2;   // emission pair table (LDS-resident, between emis and info) and the
                                                         // pair class of every column (n_cols + 2 x uint16, read at staging)
};
#define COL_EPAIR_SYMBOLS 5   // A, C, G, T and the "row past the read" symbol (-inf)

struct ColProgramHost {
    bool valid = false;
    std::string why;                       // why there is no program (diagnostics)
    int32_t n_cols = 0, n_sinks = 0, m = 0, P = 0;
    std::vector<ColClass> classes;
    std::vector<double> emis;              // n_eclass * COL_EMIS_STRIDE
    // Emission PAIR table of the row-blocked sweep: a column's match and insert emissions of one symbol side by side, symbol
    // major -- epair[(s * n_epair + p) * 2 + {0: M, 1: I}] -- so that a lane reaches both with ONE address (per-row symbol
    // base + per-step pair offset, a plain add) and ONE 16-byte LDS read instead of two SDWA adds and two 8-byte reads
    std::vector<double> epair;             // COL_EPAIR_SYMBOLS * n_epair * 2
    std::vector<uint16_t> pair_of_col;     // n_cols + 2
    int32_t n_epair = 0;
    std::vector<ColInfo> info;             // n_cols + 2
    std::vector<ColState> state;           // n_cols + 2
    std::vector<int32_t> pred0;            // per silent state: row-0 predecessor state or -1
    std::vector<double> v0;                // per silent state: row-0 value
    std::vector<int32_t> tail_state, tail_ptr;
    std::vector<TailEdge> tail_edges;
    int32_t end_tail = -1;
    // sum-product (forward) twin of v0b / mX: row-0 forward values and the entry term v0f[x] + logp per column
    std::vector<double> fwd;               // n_cols * 2
    std::vector<double> fv0;               // per silent state

    void reset()
    {
        valid = false;
        why.clear();
        n_cols = n_sinks = m = P = 0;
        classes.clear(); emis.clear(); epair.clear(); pair_of_col.clear(); n_epair = 0;
        info.clear(); state.clear(); pred0.clear(); v0.clear();
        tail_state.clear(); tail_ptr.clear(); tail_edges.clear();
        end_tail = -1;
        fwd.clear(); fv0.clear();
    }

    size_t lds_bytes() const
    {
        return classes.size() * sizeof(ColClass) + emis.size() * sizeof(double) + epair.size() * sizeof(double) +
               info.size() * sizeof(ColInfo) + state.size() * sizeof(ColState);
    }

    std::vector<uint8_t> serialize() const
    {
        std::vector<uint8_t> out;
        serialize_into(out);
        return out;
    }

    // (the caller's buffer keeps its capacity across models: no allocation in the steady state of a bulk upload)
    void serialize_into(std::vector<uint8_t> &out) const
    {
        out.assign(sizeof(ColProgram), 0);
        auto add = [&](const void *p, size_t bytes) -> int32_t {
            size_t off = (out.size() + 15) & ~size_t(15);
            out.resize(off + bytes + 16, 0);
            if (bytes) memcpy(out.data() + off, p, bytes);
            return (int32_t)off;
        };
        ColProgram h{};
        h.n_cols = n_cols; h.n_tclass = (int32_t)classes.size(); h.n_eclass = (int32_t)(emis.size() / COL_EMIS_STRIDE);
        h.n_tail = (int32_t)tail_state.size(); h.n_sinks = n_sinks; h.end_tail = end_tail; h.m = m; h.P = P;
        // the LDS-resident tables are contiguous and in this order
        h.off_class = add(classes.data(), classes.size() * sizeof(ColClass));
        h.off_emis = add(emis.data(), emis.size() * sizeof(double));
        h.off_epair = add(epair.data(), epair.size() * sizeof(double));
        h.n_epair = n_epair;
        h.off_info = add(info.data(), info.size() * sizeof(ColInfo));
        h.off_state = add(state.data(), state.size() * sizeof(ColState));
        h.lds_bytes = (int32_t)(((out.size() + 15) & ~size_t(15)) - (size_t)h.off_class);
        h.off_pred0 = add(pred0.data(), pred0.size() * sizeof(int32_t));
        h.off_v0 = add(v0.data(), v0.size() * sizeof(double));
        h.off_tail_ptr = add(tail_ptr.data(), tail_ptr.size() * sizeof(int32_t));
        h.off_tail_state = add(tail_state.data(), tail_state.size() * sizeof(int32_t));
        h.off_tail_edge = add(tail_edges.data(), tail_edges.size() * sizeof(TailEdge));
        h.off_fwd = add(fwd.data(), fwd.size() * sizeof(double));
        h.off_pair_of_col = add(pair_of_col.data(), pair_of_col.size() * sizeof(uint16_t));
        memcpy(out.data(), &h, sizeof h);
    }
};

namespace colprog_detail {

struct Key96 {
    uint64_t w[11];
    bool operator<(const Key96 &o) const { return memcmp(w, o.w, sizeof w) < 0; }
};
struct Key32 {
    uint64_t w[4];
    bool operator<(const Key32 &o) const { return memcmp(w, o.w, sizeof w) < 0; }
};

// The compiler's temporaries, kept per thread across models (a bulk upload compiles thousands of them: from the general
// allocator these were ~3 000 small allocations per model -- the successor lists alone one per state).
struct Scratch {
    std::vector<int> outdeg, out_ptr, out_dst, pos, backbone, colI, colM, colOf, slotOf, feed_sink, ks, tail_idx;
    std::vector<char> selfloop, live, dead, tail;
    std::vector<ColClass> percol;
    std::vector<ColState> st;
    std::vector<uint16_t> flags;
    std::vector<double> fwd_mx;
    std::vector<int32_t> chash, ehash, phash;            // open addressing: index of the class / emission row / pair, -1 = empty
    std::vector<std::pair<int, int>> pairs;
};

static inline uint64_t hash_words(const void *p, int n_words)
{
    uint64_t h = 0x9e3779b97f4a7c15ull, w;
    for (int i = 0; i < n_words; ++i) {
        memcpy(&w, (const char *)p + 8 * i, 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 29;
    }
    return h;
}

}  // namespace colprog_detail

// Model view the builder needs (engine.hip's advntr_hmm satisfies it; tests can pass their own).
template <class Model>
static inline bool build_column_program(const Model &H, ColProgramHost &out)
{
    using namespace colprog_detail;
    const double NINF = -INFINITY;
    const int m = H.m, P = H.P, S = m - P;
    const auto &in_ptr = H.in_ptr;
    const auto &in_src = H.in_src;
    const auto &in_logp = H.in_logp;
    out.reset();                          // keeps the vectors' capacity: `out` may be a per-thread scratch object
    out.m = m; out.P = P;
    auto fail = [&](const std::string &why) { out.valid = false; out.why = why; return false; };
    if (!H.finite) return fail("model has no end state in-edges (infinite model)");
    if (S < 2 || P < 1) return fail("too few states");
    if (m > 65000) return fail("too many states");

    static thread_local Scratch scratch;
    Scratch &W = scratch;
    // out-degree, self loops
    std::vector<int> &outdeg = W.outdeg;
    std::vector<char> &selfloop = W.selfloop;
    outdeg.assign(m, 0);
    selfloop.assign(m, 0);
    for (int l = 0; l < m; ++l)
        for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k) {
            outdeg[in_src[k]]++;
            if (in_src[k] == l) selfloop[l] = 1;
        }
    for (int l = P; l < m; ++l)
        if (selfloop[l]) return fail("silent self loop");

    // reference evaluation order of a silent state's in-edges (emitting-sourced, then silent ki<l)
    auto rlist = [&](int l, std::vector<int> &ks) {
        ks.clear();
        for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k)
            if (in_src[k] < P) ks.push_back(k);
        for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k)
            if (in_src[k] >= P && in_src[k] < l) ks.push_back(k);
    };
    // silent edges from ki >= l never fire in the reference (hmm.pyx:2069); they only matter if present
    for (int l = P; l < m; ++l)
        for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k)
            if (in_src[k] >= l) return fail("silent edge against the topological order");

    // ---- row 0 (read independent): hmm.pyx:1999-2023
    out.v0.assign(S, NINF);
    out.pred0.assign(S, -1);
    out.v0[H.start - P] = 0.0;
    for (int l = P; l < m; ++l) {
        if (l == H.start) continue;
        double best = NINF;
        int bs = -1;
        for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k) {
            const int ki = in_src[k];
            if (ki < P || ki >= l) continue;
            const double cand = out.v0[ki - P] + in_logp[k];
            if (cand > best) { best = cand; bs = ki; }
        }
        out.v0[l - P] = best;
        out.pred0[l - P] = bs;
    }

    // ---- row 0 of the forward (sum-product) recursion: hmm.pyx:1403-1427, pair_lse utils.pyx:72-90
    auto lse2h = [](double x, double y) {
        if (x == INFINITY || y == INFINITY) return (double)INFINITY;
        if (x == -INFINITY) return y;
        if (y == -INFINITY) return x;
        if (x > y) return x + std::log(std::exp(y - x) + 1);
        return y + std::log(std::exp(x - y) + 1);
    };
    out.fv0.assign(S, NINF);
    out.fv0[H.start - P] = 0.0;
    for (int l = P; l < m; ++l) {
        if (l == H.start) continue;
        double lp = NINF;
        for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k) {
            const int ki = in_src[k];
            if (ki < P || ki >= l) continue;
            lp = lse2h(lp, out.fv0[ki - P] + in_logp[k]);
        }
        out.fv0[l - P] = lp;
    }

    // ---- reachable from an emitting state (=> may be alive in rows >= 1)
    std::vector<char> &live = W.live;
    live.assign(m, 0);
    for (int l = 0; l < P; ++l) live[l] = 1;
    for (int l = P; l < m; ++l)
        for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k)
            if (live[in_src[k]]) live[l] = 1;

    // ---- dead ends and tail: silent states whose every out-edge leads to the end through silent states only
    // (successor lists as one CSR: out_dst[out_ptr[l] .. out_ptr[l + 1]) in increasing destination order)
    std::vector<int> &out_ptr = W.out_ptr, &out_dst = W.out_dst;
    out_ptr.assign(m + 1, 0);
    for (int l = 0; l < m; ++l) out_ptr[l + 1] = out_ptr[l] + outdeg[l];
    out_dst.resize((size_t)out_ptr[m]);
    {
        std::vector<int> &fill = W.ks;
        fill.assign(out_ptr.begin(), out_ptr.end() - 1);
        for (int l = 0; l < m; ++l)
            for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k) out_dst[(size_t)fill[in_src[k]]++] = l;
    }
    std::vector<char> &dead = W.dead, &tail = W.tail;
    dead.assign(m, 0);
    tail.assign(m, 0);
    tail[H.end] = 1;
    if (H.end < P) return fail("end state is emitting");
    for (int l = m - 1; l >= P; --l) {
        if (l == H.end) continue;
        if (outdeg[l] == 0) { dead[l] = 1; continue; }
        bool all = true;
        for (int q = out_ptr[l]; q < out_ptr[l + 1]; ++q) {
            const int d = out_dst[(size_t)q];
            if (!(d >= P && (tail[d] || dead[d]))) all = false;
        }
        if (all) tail[l] = 1;
    }
    if (outdeg[H.end] != 0) return fail("end state has out-edges");

    // ---- backbone columns
    std::vector<int> &pos = W.pos, &backbone = W.backbone;
    pos.assign(m, -1);
    backbone.clear();
    for (int l = P; l < m; ++l)
        if (!dead[l] && !tail[l]) { pos[l] = (int)backbone.size(); backbone.push_back(l); }
    const int NC = (int)backbone.size();
    if (NC < 2) return fail("no backbone");
    if (backbone[0] != H.start && out.v0[backbone[0] - P] != NINF) { /* fine: any order */ }
    for (int l = P; l < m; ++l)
        if (dead[l])
            for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k) (void)k;   // edges into dead ends are ignored

    // ---- emitting states -> (column, slot)
    std::vector<int> &colI = W.colI, &colM = W.colM, &colOf = W.colOf, &slotOf = W.slotOf;
    colI.assign(NC, -1);
    colM.assign(NC, -1);
    colOf.assign(m, -1);
    slotOf.assign(m, -1);
    for (int u = 0; u < P; ++u) {
        int bmax = -1;
        for (int k = in_ptr[u]; k < in_ptr[u + 1]; ++k) {
            const int s = in_src[k];
            if (s >= P) {
                if (pos[s] < 0) return fail("emitting state fed by a tail/dead silent state");
                bmax = std::max(bmax, s);
            }
        }
        if (bmax < 0) return fail("emitting state without a silent predecessor");
        const bool isI = selfloop[u];
        const int c = pos[bmax] + (isI ? 0 : 1);
        if (c >= NC) return fail("match state beyond the last column");
        if (isI) {
            if (colI[c] >= 0) return fail("two insert-like states in one column");
            colI[c] = u;
        } else {
            if (colM[c] >= 0) return fail("two match-like states in one column");
            colM[c] = u;
        }
        colOf[u] = c;
        slotOf[u] = isI ? 0 : 1;
    }
    for (int c = 0; c < NC; ++c) { colOf[backbone[c]] = c; slotOf[backbone[c]] = 2; }

    // ---- per column parameters, validating every in-edge against the stencil order
    std::vector<ColClass> &percol = W.percol;
    std::vector<ColState> &st = W.st;
    std::vector<uint16_t> &flags = W.flags;
    std::vector<int> &feed_sink = W.feed_sink;
    std::vector<double> &fwd_mx = W.fwd_mx;
    percol.assign(NC, ColClass{});
    st.assign(NC, ColState{});
    flags.assign(NC, 0);
    feed_sink.assign(NC, -1);
    fwd_mx.assign(NC, NINF);
    std::vector<int> &ks = W.ks;
    int n_sinks = 0;
    for (int c = 0; c < NC; ++c) {
        ColClass &T = percol[c];
        T.iI = T.iM = T.iD = T.mI = T.mM = T.mX = T.mD = T.dI = T.dM = T.dD = T.erw = NINF;
        st[c].sI = colI[c]; st[c].sM = colM[c]; st[c].sB = backbone[c]; st[c].sX = -1;
    }
    for (int c = 0; c < NC; ++c) {
        ColClass &T = percol[c];
        // I slot: [I_c, M_c, b_c] at the previous row
        if (colI[c] >= 0) {
            const int u = colI[c];
            int stage = -1;
            for (int k = in_ptr[u]; k < in_ptr[u + 1]; ++k) {
                const int s = in_src[k];
                int w;
                if (s == u) w = 0;
                else if (s == colM[c]) w = 1;
                else if (s == backbone[c]) w = 2;
                else return fail("insert-like state has an in-edge outside the stencil");
                if (w <= stage) return fail("insert-like state: in-edge order differs from [I,M,b]");
                stage = w;
                (w == 0 ? T.iI : w == 1 ? T.iM : T.iD) = in_logp[k];
            }
        }
        // M slot: [I_{c-1}, M_{c-1}, X, b_{c-1}] at the previous row
        if (colM[c] >= 0) {
            const int u = colM[c];
            if (c == 0) return fail("match-like state in column 0");
            int stage = -1;
            for (int k = in_ptr[u]; k < in_ptr[u + 1]; ++k) {
                const int s = in_src[k];
                int w;
                if (s == colI[c - 1] && s >= 0) w = 0;
                else if (s == colM[c - 1] && s >= 0) w = 1;
                else if (s == backbone[c - 1]) w = 3;
                else if (s >= P && pos[s] >= 0 && !live[s]) w = 2;      // entry edge from a row-0-only state
                else return fail("match-like state has an in-edge outside the stencil");
                if (w <= stage) return fail("match-like state: in-edge order differs from [I,M,X,b]");
                stage = w;
                if (w == 0) T.mI = in_logp[k];
                else if (w == 1) T.mM = in_logp[k];
                else if (w == 3) T.mD = in_logp[k];
                else { T.mX = out.v0[s - P] + in_logp[k]; st[c].sX = s; fwd_mx[c] = out.fv0[s - P] + in_logp[k]; }   // (v + t), e is added on device
            }
        }
        // backbone state
        {
            const int l = backbone[c];
            rlist(l, ks);
            bool stencil = true;
            int stage = -1;
            for (int k : ks) {
                const int s = in_src[k];
                int w;
                if (c > 0 && s == colI[c - 1] && s >= 0) w = 0;
                else if (c > 0 && s == colM[c - 1] && s >= 0) w = 1;
                else if (c > 0 && s == backbone[c - 1]) w = 2;
                else { stencil = false; break; }
                if (w <= stage) { stencil = false; break; }
                stage = w;
            }
            if (stencil) {
                for (int k : ks) {
                    const int s = in_src[k];
                    if (s == colI[c - 1]) T.dI = in_logp[k];
                    else if (s == colM[c - 1]) T.dM = in_logp[k];
                    else T.dD = in_logp[k];
                }
            } else {
                // sink: every source is an earlier backbone state, in increasing column order
                if (n_sinks >= COL_MAX_SINKS) return fail("too many fan-in states");
                int lastpos = -1;
                for (int k : ks) {
                    const int s = in_src[k];
                    if (s < P || pos[s] < 0 || pos[s] >= c) return fail("silent fan-in from a non-backbone state");
                    if (pos[s] <= lastpos) return fail("silent fan-in not in column order");
                    lastpos = pos[s];
                    if (flags[pos[s]] & COL_FLAG_FEED) return fail("backbone state feeds two fan-in states");
                    flags[pos[s]] |= COL_FLAG_FEED | (uint16_t)(n_sinks << 8);
                    feed_sink[pos[s]] = n_sinks;
                    percol[pos[s]].erw = in_logp[k];
                }
                flags[c] |= COL_FLAG_SINK | (uint16_t)(n_sinks << 4);
                n_sinks++;
            }
        }
    }
    // Column 0 takes nothing from a previous column (no match-like state, backbone state without in-edges in rows >= 1).  The
    // back-to-back sweeps of viterbi_rows.h rely on it: a lane that starts column 0 of its next read still holds the last
    // column's values of the previous one, and these -inf transitions are what keeps them out.
    {
        const ColClass &T0 = percol[0];
        if (T0.mI != NINF || T0.mM != NINF || T0.mX != NINF || T0.mD != NINF || T0.dI != NINF || T0.dM != NINF || T0.dD != NINF ||
            (flags[0] & COL_FLAG_SINK))
            return fail("column 0 has a predecessor column");
    }
    // feeds and sinks must not interleave: between a feeder and its sink there is no other sink, and no
    // feeder of a different sink
    {
        int cur = -1;
        for (int c = 0; c < NC; ++c) {
            if (flags[c] & COL_FLAG_SINK) {
                const int sidx = (flags[c] >> 4) & 15;
                if (cur != -1 && cur != sidx) return fail("interleaved fan-in ranges");
                cur = -1;
            }
            if (flags[c] & COL_FLAG_FEED) {
                if (cur != -1 && cur != feed_sink[c]) return fail("interleaved fan-in ranges");
                cur = feed_sink[c];
            }
        }
        if (cur != -1) return fail("feeder without a sink");
    }
    // a feeder that is also claimed by the next column's stencil (b_{c-1} edge) would be double counted:
    // the sink takes ONLY the accumulator, so check sinks have no stencil edges (true by construction).

    // ---- tail states (evaluated at the last row only), index order
    std::vector<int> &tail_idx = W.tail_idx;             // per state: its index among the tail states
    tail_idx.assign(m, 0);
    out.tail_ptr.push_back(0);
    for (int l = P; l < m; ++l) {
        if (!tail[l]) continue;
        tail_idx[l] = (int)out.tail_state.size();
        out.tail_state.push_back(l);
        rlist(l, ks);
        for (int k : ks) {
            const int s = in_src[k];
            TailEdge e;
            e.src_state = s;
            e.logp = in_logp[k];
            if (s >= P && tail[s]) e.loc = -(tail_idx[s]) - 1;
            else if (s >= P && dead[s]) return fail("tail state fed by a dead end");
            else e.loc = colOf[s] * 4 + slotOf[s];
            out.tail_edges.push_back(e);
        }
        out.tail_ptr.push_back((int32_t)out.tail_edges.size());
    }
    out.end_tail = tail_idx[H.end];

    // ---- class tables (bit-exact de-duplication)
    // (ids in order of first appearance; open addressing on the bit patterns, at most 2 (NC + 2) + 2 entries per table)
    size_t hsize = 64;
    while (hsize < 4 * ((size_t)NC + 4)) hsize *= 2;
    const size_t hmask = hsize - 1;
    W.chash.assign(hsize, -1);
    W.ehash.assign(hsize, -1);
    static_assert(sizeof(Key96) == 88 && sizeof(Key96) <= sizeof(ColClass), "class key: the first 88 bytes of a ColClass");
    auto class_of = [&](const ColClass &T) {
        size_t h = (size_t)hash_words(&T, 11) & hmask;
        for (;; h = (h + 1) & hmask) {
            const int32_t id = W.chash[h];
            if (id < 0) break;
            if (memcmp(&out.classes[(size_t)id], &T, sizeof(Key96)) == 0) return (int)id;
        }
        const int id = (int)out.classes.size();
        out.classes.push_back(T);
        W.chash[h] = id;
        return id;
    };
    auto eclass_of = [&](const double *e4) {
        size_t h = (size_t)hash_words(e4, 4) & hmask;
        for (;; h = (h + 1) & hmask) {
            const int32_t id = W.ehash[h];
            if (id < 0) break;
            if (memcmp(&out.emis[(size_t)id * COL_EMIS_STRIDE], e4, 32) == 0) return (int)id;
        }
        const int id = (int)(out.emis.size() / COL_EMIS_STRIDE);
        out.emis.insert(out.emis.end(), e4, e4 + 4);
        out.emis.push_back(NINF);        // 5th slot: the emission of a padding row (viterbi_rows.h)
        W.ehash[h] = id;
        return id;
    };
    ColClass none{};
    none.iI = none.iM = none.iD = none.mI = none.mM = none.mX = none.mD = none.dI = none.dM = none.dD = none.erw = NINF;
    const int none_class = class_of(none);
    const double noe[4] = {NINF, NINF, NINF, NINF};
    const int none_emis = eclass_of(noe);
    out.info.resize(NC + 2);
    out.state.resize(NC + 2);
    for (int cc = 0; cc < NC + 2; ++cc) {
        ColInfo &I = out.info[cc];
        ColState &Sx = out.state[cc];
        if (cc == 0 || cc == NC + 1) {
            I.v0b = NINF; I.tclass = (uint16_t)none_class; I.emM = I.emI = (uint16_t)none_emis; I.flags = 0;
            Sx.sI = Sx.sM = Sx.sB = Sx.sX = -1;
            continue;
        }
        const int c = cc - 1;
        I.v0b = out.v0[backbone[c] - P];
        I.tclass = (uint16_t)class_of(percol[c]);
        I.emM = (uint16_t)(colM[c] >= 0 ? eclass_of(&H.emis[(size_t)colM[c] * 4]) : none_emis);
        I.emI = (uint16_t)(colI[c] >= 0 ? eclass_of(&H.emis[(size_t)colI[c] * 4]) : none_emis);
        I.flags = flags[c];
        Sx = st[c];
    }
    if (out.classes.size() > 60000 || out.emis.size() / COL_EMIS_STRIDE > 60000) return fail("class table overflow");
    {   // emission pair classes: distinct (emM, emI) combinations of the columns
        std::vector<std::pair<int, int>> &pairs = W.pairs;
        pairs.clear();
        W.phash.assign(hsize, -1);
        out.pair_of_col.resize(NC + 2);
        for (int cc = 0; cc < NC + 2; ++cc) {
            const std::pair<int, int> key(out.info[cc].emM, out.info[cc].emI);
            size_t h = (size_t)(((uint64_t)key.first << 16 | (uint64_t)key.second) * 0x9e3779b97f4a7c15ull >> 40) & hmask;
            int found = -1;
            for (;; h = (h + 1) & hmask) {
                const int32_t id = W.phash[h];
                if (id < 0) break;
                if (pairs[(size_t)id] == key) { found = id; break; }
            }
            if (found < 0) { found = (int)pairs.size(); pairs.push_back(key); W.phash[h] = found; }
            out.pair_of_col[cc] = (uint16_t)found;
        }
        out.n_epair = (int32_t)pairs.size();
        out.epair.assign((size_t)COL_EPAIR_SYMBOLS * pairs.size() * 2, NINF);
        for (int sy = 0; sy < COL_EPAIR_SYMBOLS; ++sy)
            for (size_t q = 0; q < pairs.size(); ++q) {
                out.epair[((size_t)sy * pairs.size() + q) * 2 + 0] = out.emis[(size_t)pairs[q].first * COL_EMIS_STRIDE + sy];
                out.epair[((size_t)sy * pairs.size() + q) * 2 + 1] = out.emis[(size_t)pairs[q].second * COL_EMIS_STRIDE + sy];
            }
    }
    out.fwd.resize((size_t)NC * 2);
    for (int c = 0; c < NC; ++c) {
        out.fwd[2 * c] = out.fv0[backbone[c] - P];
        out.fwd[2 * c + 1] = fwd_mx[c];
    }
    out.n_cols = NC;
    out.n_sinks = n_sinks;
    // What has to sit in LDS is the class / emission tables and the sweep's padded copy of the column-info table; the
    // traceback's state table and the unpadded info table are read from the model blob when they do not fit (staging
    // levels 1 and 0, engine.hip).  (The sum-product sweep needs 16 B per column more; advntr_forward_batch says so when a
    // model is too wide for it.)
    {
        const size_t tables = out.classes.size() * sizeof(ColClass) + (out.emis.size() + out.epair.size()) * sizeof(double);
        if (tables + (size_t)(NC + 128 * 4) * sizeof(ColInfo) > 150 * 1024)
            return fail("column program larger than the LDS of a CU (class / emission tables + 16 B per column)");
    }
    // the sweep addresses class and emission records through 16-bit LDS addresses (tables start 16 B into LDS)
    if (out.classes.size() * sizeof(ColClass) + (out.emis.size() + out.epair.size()) * sizeof(double) + 64 > 0x10000)
        return fail("class + emission tables larger than 64 KiB of LDS");
    out.valid = true;
    return true;
}
