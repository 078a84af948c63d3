// Native construction of adVNTR's read-matcher model (host C++, no device code).
//
// What it replaces on the reference side:
//   advntr/profile_hmm.py:13-161      build_profile_hmm_pseudocounts_for_alignment  -> Profile / estimate_profile
//   advntr/hmm_utils.py:290-420       get_prefix_matcher_hmm / get_suffix_matcher_hmm -> flank_block
//   advntr/hmm_utils.py:424-497       get_constant_number_of_repeats_matcher_hmm     -> repeat_block
//   advntr/hmm_utils.py:501-549       get_variable_number_of_repeats_matcher_hmm     -> open_repeat_block
//   advntr/hmm_utils.py:553-595       get_read_matcher_model                         -> read_matcher
//   pomegranate/hmm.pyx:673-1123      bake(merge=None)                               -> Net::bake
//   pomegranate/hmm.pyx:492-514       dense_transition_matrix                        -> Net::probability_rows
//   pomegranate/hmm.pyx:3146-3238     from_matrix                                    -> rebuild_from_rows
//   pomegranate/hmm.pyx:584-615       concatenate                                    -> Net::append
//
// The reference reaches the final model through three bakes, two dense m x m probability matrices and two
// from_matrix rebuilds, all in Python on networkx (0.8-1.0 s per locus).  The result depends on that route:
// the order in which edges enter the graph becomes the CSR in-edge order (= the Viterbi tie-break), the
// log -> exp -> log round trips decide the last bits of every parameter, and from_matrix's end-edge quirk
// shapes the tail of the model.  This builder takes the same route on a small insertion-ordered graph with
// SPARSE probability rows (no m x m matrix), so it lands on the same arrays in well under a millisecond.
//
// exp(): the reference exponentiates with numpy.exp, whose fp64 kernel is SIMD-dispatched and differs from
// libm's exp in the last bit for a few percent of arguments.  The caller may therefore pass the exp to use
// (the Python host passes numpy.exp, which makes the parameters bit-identical to the reference's on the same
// machine); with none given, libm exp is used (<= 1 ulp from it per round trip).
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/advntr_hip.h"

namespace mb {

typedef void (*ExpFn)(const double *in, double *out, int64_t n, void *user);

// log(x) as libm gives it, remembered per argument: a model's eighteen thousand logarithms are taken of a few hundred distinct
// probabilities (the transition classes of the flanks, the profile columns repeated in every copy), and the value libm returns
// for an argument is the value it returns the next time.  Direct-mapped, per thread.
static inline double log_or_ninf(double x)            // utils.pyx:64-70
{
    if (!(x > 0)) return -std::numeric_limits<double>::infinity();
    struct Entry { uint64_t key; double val; };
    static thread_local Entry memo[1024] = {};        // (key 0 = the bits of +0.0, which never gets here)
    uint64_t bits;
    memcpy(&bits, &x, 8);
    Entry &e = memo[(bits ^ (bits >> 17) ^ (bits >> 41)) & 1023u];
    if (e.key != bits) { e.key = bits; e.val = std::log(x); }
    return e.val;
}

static inline int base_code(char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

struct Arc { int to; double logp; };
struct Vertex { std::string name; int emission; };   // emission: row of Net::emissions, -1 = silent

typedef std::vector<std::vector<std::pair<int, double>>> Rows;   // sparse probability rows, columns ascending

// Insertion-ordered directed graph + the baked numbering of it.
struct Net {
    std::string name;
    std::vector<Vertex> v;
    std::vector<std::vector<Arc>> out;
    std::vector<std::array<double, 4>> emissions;     // log-probabilities of A,C,G,T
    int start = -1, end = -1;
    // after bake()
    std::vector<int> order, pos;
    int silent_start = 0, start_index = -1, end_index = -1;

    explicit Net(const std::string &nm) : name(nm)
    {
        start = add_vertex(nm + "-start", -1);
        end = add_vertex(nm + "-end", -1);
    }
    int add_vertex(const std::string &nm, int emission)
    {
        v.push_back(Vertex{nm, emission});
        out.emplace_back();
        out.back().reserve(4);                       // (a state of these models has three or four successors: one allocation)
        return (int)v.size() - 1;
    }
    int add_emission(const std::array<double, 4> &prob)
    {
        std::array<double, 4> lp;
        for (int i = 0; i < 4; ++i) lp[i] = log_or_ninf(prob[i]) + 0.0;   // + log(state weight 1), hmm.pyx:928-930
        emissions.push_back(lp);
        return (int)emissions.size() - 1;
    }
    // add_transition: a repeated (a,b) keeps its first position and takes the new value
    void arc(int a, int b, double prob)
    {
        const double lp = log_or_ninf(prob);
        for (Arc &e : out[a])
            if (e.to == b) { e.logp = lp; return; }
        out[a].push_back(Arc{b, lp});
    }
    void arc_new(int a, int b, double prob) { out[a].push_back(Arc{b, log_or_ninf(prob)}); }
    size_t n_arcs() const
    {
        size_t n = 0;
        for (const auto &o : out) n += o.size();
        return n;
    }

    // concatenate(other): vertices and arcs of `other` follow ours, end -> other.start with probability 1
    void append(const Net &other)
    {
        const int shift = (int)v.size(), eshift = (int)emissions.size();
        for (const Vertex &x : other.v) v.push_back(Vertex{x.name, x.emission < 0 ? -1 : x.emission + eshift});
        for (const auto &o : other.out) {
            out.emplace_back(o);
            for (Arc &e : out.back()) e.to += shift;
        }
        emissions.insert(emissions.end(), other.emissions.begin(), other.emissions.end());
        arc(end, other.start + shift, 1.0);
        end = other.end + shift;
    }

    // bake(merge=None): emitting vertices sorted by name, then the silent ones name-sorted and put in the
    // reversed post-order of an iterative DFS that stacks successors in adjacency order (networkx 1.11)
    void bake()
    {
        const int n = (int)v.size();
        std::vector<int> emitting, silent;
        for (int i = 0; i < n; ++i) (v[i].emission >= 0 ? emitting : silent).push_back(i);
        auto by_name = [&](int a, int b) { return v[a].name < v[b].name; };
        // (a net rebuilt from rows lists its emitting states in the previous bake's order: sorted already)
        if (!std::is_sorted(emitting.begin(), emitting.end(), by_name)) std::stable_sort(emitting.begin(), emitting.end(), by_name);
        if (!std::is_sorted(silent.begin(), silent.end(), by_name)) std::stable_sort(silent.begin(), silent.end(), by_name);
        std::vector<char> seen(n, 0), done(n, 0);
        std::vector<int> post, stack;
        post.reserve(silent.size());
        for (int root : silent) {
            if (done[root]) continue;
            stack.assign(1, root);
            while (!stack.empty()) {
                const int w = stack.back();
                if (done[w]) { stack.pop_back(); continue; }
                seen[w] = 1;
                const size_t before = stack.size();
                for (const Arc &e : out[w]) {
                    if (v[e.to].emission >= 0 || done[e.to]) continue;
                    if (seen[e.to]) throw std::runtime_error("silent states form a cycle");
                    stack.push_back(e.to);
                }
                if (stack.size() == before) { done[w] = 1; post.push_back(w); stack.pop_back(); }
            }
        }
        order = emitting;
        order.insert(order.end(), post.rbegin(), post.rend());
        silent_start = (int)emitting.size();
        pos.assign(n, -1);
        for (int i = 0; i < n; ++i) pos[order[i]] = i;
        start_index = pos[start];
        end_index = pos[end];
    }

    // dense_transition_matrix, sparsely: row i = out-arcs of baked state i as (baked column, exp(logp))
    Rows probability_rows(ExpFn exp_fn, void *user, int extra_rows) const
    {
        const int n = (int)v.size();
        std::vector<double> lp, p;
        lp.reserve(n_arcs());
        for (int i = 0; i < n; ++i)
            for (const Arc &e : out[order[i]]) lp.push_back(e.logp);
        p.resize(lp.size());
        {   // exp over the DISTINCT arguments only (a few hundred of several thousand; the caller's exp -- numpy's loop -- gives
            // an argument the same value wherever it stands in the array, which the bit-identical goldens already rest on)
            std::vector<uint64_t> keys(2048, ~0ull);                 // open addressing on the bit patterns (NaN never occurs)
            std::vector<int32_t> slot_of(2048, -1), which(lp.size());
            std::vector<double> uniq, uexp;
            bool fits = true;
            for (size_t k = 0; k < lp.size() && fits; ++k) {
                uint64_t bits;
                memcpy(&bits, &lp[k], 8);
                size_t h = (size_t)((bits ^ (bits >> 19) ^ (bits >> 43)) & 2047u);
                while (keys[h] != ~0ull && keys[h] != bits) h = (h + 1) & 2047u;
                if (keys[h] == ~0ull) {
                    if (uniq.size() >= 1536) { fits = false; break; }
                    keys[h] = bits;
                    slot_of[h] = (int32_t)uniq.size();
                    uniq.push_back(lp[k]);
                }
                which[k] = slot_of[h];
            }
            if (fits) {
                uexp.resize(uniq.size());
                if (exp_fn) exp_fn(uniq.data(), uexp.data(), (int64_t)uniq.size(), user);
                else
                    for (size_t k = 0; k < uniq.size(); ++k) uexp[k] = std::exp(uniq[k]);
                for (size_t k = 0; k < lp.size(); ++k) p[k] = uexp[(size_t)which[k]];
            } else if (exp_fn) exp_fn(lp.data(), p.data(), (int64_t)lp.size(), user);
            else
                for (size_t k = 0; k < lp.size(); ++k) p[k] = std::exp(lp[k]);
        }
        Rows rows(n + extra_rows);
        size_t k = 0;
        for (int i = 0; i < n; ++i) {
            auto &r = rows[i];
            r.reserve(out[order[i]].size() + 1);
            for (const Arc &e : out[order[i]]) r.emplace_back(pos[e.to], p[k++]);
            std::sort(r.begin(), r.end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.first < b.first; });
        }
        return rows;
    }
};

static inline void row_set(std::vector<std::pair<int, double>> &r, int col, double val)
{
    auto it = std::lower_bound(r.begin(), r.end(), col, [](const std::pair<int, double> &a, int c) { return a.first < c; });
    if (it != r.end() && it->first == col) it->second = val;
    else r.insert(it, std::make_pair(col, val));
}
static inline int row_last_nonzero(const std::vector<std::pair<int, double>> &r)
{
    for (size_t k = r.size(); k-- > 0;)
        if (r[k].second != 0) return r[k].first;
    throw std::runtime_error("state without successors");
}

// from_matrix(rows, distributions, starts = e_{start_col}, ends = e_{end_col}, state_names) + bake.  `from` supplies
// names and emissions of the first from.v.size() states (in baked order); `extra` names the silent ones after them.
static Net rebuild_from_rows(const Net &from, const Rows &rows, const std::vector<std::string> &extra,
                             int start_col, const std::string &name)
{
    Net g(name);
    const int n = (int)rows.size(), base = 2;
    g.emissions = from.emissions;
    for (int i = 0; i < (int)from.v.size(); ++i) {
        const Vertex &x = from.v[from.order[i]];
        g.add_vertex(x.name, x.emission);
    }
    for (const std::string &nm : extra) g.add_vertex(nm, -1);
    g.arc_new(g.start, base + start_col, 1.0);
    for (int i = 0; i < n; ++i)
        for (const auto &cv : rows[i])
            if (cv.second != 0) g.arc_new(base + i, base + cv.first, cv.second);
    // hmm.pyx:3231-3235: the end edge leaves states[j] with the inner loop's stale j = n-1, whatever `ends` marks
    g.arc(base + n - 1, g.end, 1.0);
    g.bake();
    return g;
}

// ---- a-6: profile parameters from aligned repeat units ---------------------------------------------------
struct Profile {
    int L = 0;
    // successor slots of a source at position k: 0 = I_k, 1 = M_{k+1}, 2 = D_{k+1}, 3 = unit_end
    std::vector<std::array<double, 4>> from_I, from_M, from_D;   // index k (from_M/from_D: 1..L; from_I: 0..L)
    std::array<double, 4> from_unit_start{};                     // slots 0 = I0, 1 = M1, 2 = D1
    std::vector<std::array<double, 4>> emit_I, emit_M;           // probabilities of A,C,G,T
};

static Profile estimate_profile(const std::vector<std::string> &alignment, double error_rate)
{
    const int rows = (int)alignment.size();
    if (rows == 0) throw std::invalid_argument("no repeat units");
    const int width = (int)alignment[0].size();
    for (const std::string &r : alignment)
        if ((int)r.size() != width)
            throw std::invalid_argument("repeat units of different lengths need a multiple alignment first (the reference "
                                        "shells out to muscle, profile_hmm.py:166-171); pass aligned rows");
    const double pseu = (rows / 4.0) * (error_rate / 10);
    const double gap_cut = 0.5 * rows;
    std::vector<char> insert_col(width, 0);
    int L = 0;
    for (int c = 0; c < width; ++c) {
        double gaps = 0;
        for (const std::string &r : alignment) gaps += r[c] == '-' ? 1.0 : 0.0;
        insert_col[c] = gaps >= gap_cut;
        L += !insert_col[c];
    }
    if (L == 0) throw std::invalid_argument("alignment has no match column");
    typedef std::array<long, 4> Cnt;
    std::vector<Cnt> eI(L + 1, Cnt{}), eM(L + 1, Cnt{}), tI(L + 1, Cnt{}), tM(L + 1, Cnt{}), tD(L + 1, Cnt{});
    Cnt tS{};
    for (const std::string &r : alignment) {
        int k = 1;
        char prev_kind = 'S';
        int prev_k = 0;
        auto step = [&](char kind, int at) {
            // slot of (kind, at) seen from (prev_kind, prev_k)
            const int slot = kind == 'I' ? 0 : (kind == 'M' ? 1 : 2);
            Cnt &src = prev_kind == 'S' ? tS : (prev_kind == 'I' ? tI[prev_k] : (prev_kind == 'M' ? tM[prev_k] : tD[prev_k]));
            src[slot] += 1;
            prev_kind = kind;
            prev_k = at;
        };
        for (int c = 0; c < width; ++c) {
            const char ch = r[c];
            if (!insert_col[c]) {
                if (ch == '-') step('D', k);
                else {
                    const int b = base_code(ch);
                    if (b < 0) throw std::invalid_argument(std::string("symbol '") + ch + "' in a repeat unit is not one of ACGT");
                    eM[k][b] += 1;
                    step('M', k);
                }
                ++k;
            } else if (ch != '-') {
                const int b = base_code(ch);
                if (b < 0) throw std::invalid_argument(std::string("symbol '") + ch + "' in a repeat unit is not one of ACGT");
                eI[k - 1][b] += 1;
                step('I', k - 1);
            }
        }
        Cnt &src = prev_kind == 'I' ? tI[prev_k] : (prev_kind == 'M' ? tM[prev_k] : tD[prev_k]);
        src[3] += 1;
    }
    auto emission = [&](const Cnt &cnt) {
        std::array<double, 4> e;
        long total = 0;
        for (long c : cnt) total += c;
        if (total > 0) {
            double sub_total = 0;
            for (int i = 0; i < 4; ++i) { e[i] = (1.0 * cnt[i]) / total + pseu; sub_total += e[i]; }
            for (int i = 0; i < 4; ++i) e[i] = e[i] / sub_total;
        } else
            e.fill(1.0 / 4);
        return e;
    };
    // the successors a source owns: {I_k, M_{k+1}, D_{k+1}} (3) before the last position, {I_L, unit_end} (2) at it
    auto transition = [&](const Cnt &cnt, bool last) {
        std::array<double, 4> t{};
        long total = 0;
        for (long c : cnt) total += c;
        const int k = last ? 2 : 3;
        for (int s = 0; s < 4; ++s) {
            const bool owned = last ? (s == 0 || s == 3) : (s != 3);
            if (!owned) { t[s] = 0; continue; }
            if (total > 0) {
                const double p = 1.0 * cnt[s] / total;
                t[s] = (p + pseu) / (1 + pseu * k);
            } else
                t[s] = last ? 1.0 / 2 : 1.0 / 3;
        }
        return t;
    };
    Profile P;
    P.L = L;
    P.from_unit_start = transition(tS, false);
    P.from_I.resize(L + 1); P.from_M.resize(L + 1); P.from_D.resize(L + 1);
    P.emit_I.resize(L + 1); P.emit_M.resize(L + 1);
    for (int k = 0; k <= L; ++k) {
        P.from_I[k] = transition(tI[k], k == L);
        P.emit_I[k] = emission(eI[k]);
        if (k >= 1) {
            P.from_M[k] = transition(tM[k], k == L);
            P.from_D[k] = transition(tD[k], k == L);
            P.emit_M[k] = emission(eM[k]);
        }
    }
    return P;
}

// ---- a-5: the three blocks ---------------------------------------------------------------------------------
static Net flank_block(const std::string &pattern, const std::string &tag, const std::string &model_name,
                       bool enter_anywhere, bool early_exit, double max_error_rate)
{
    Net g(model_name);
    const int F = (int)pattern.size();
    if (F == 0) throw std::invalid_argument("empty flanking region");
    const int uniform = g.add_emission({0.25, 0.25, 0.25, 0.25});
    int match_emission[4];
    for (int b = 0; b < 4; ++b) {
        std::array<double, 4> p = {0.01, 0.01, 0.01, 0.01};
        p[b] = 0.97;
        match_emission[b] = g.add_emission(p);
    }
    std::vector<int> ins(F + 1), mat(F), del(F);
    for (int i = 0; i <= F; ++i) ins[i] = g.add_vertex("I" + std::to_string(i) + "_" + tag, uniform);
    for (int i = 0; i < F; ++i) {
        const int b = base_code(pattern[i]);
        if (b < 0) throw std::invalid_argument(std::string("symbol '") + pattern[i] + "' in a flanking region is not one of ACGT");
        mat[i] = g.add_vertex("M" + std::to_string(i + 1) + "_" + tag, match_emission[b]);
    }
    for (int i = 0; i < F; ++i) del[i] = g.add_vertex("D" + std::to_string(i + 1) + "_" + tag, -1);
    const int unit_start = g.add_vertex(tag + "_start_" + tag, -1);
    const int unit_end = g.add_vertex(tag + "_end_" + tag, -1);
    const int last = F - 1;
    g.arc(g.start, unit_start, 1);
    g.arc(unit_end, g.end, 1);
    const double insert_error = max_error_rate * 2 / 5;
    const double delete_error = max_error_rate * 1 / 5;
    const double stay = 1 - insert_error - delete_error;
    if (enter_anywhere) {
        g.arc(unit_start, del[0], delete_error);
        g.arc(unit_start, ins[0], insert_error);
        for (int i = 0; i < F; ++i) g.arc(unit_start, mat[i], (1 - insert_error - delete_error) / F);
    } else {
        g.arc(unit_start, mat[0], stay);
        g.arc(unit_start, del[0], delete_error);
        g.arc(unit_start, ins[0], insert_error);
    }
    g.arc(ins[0], ins[0], insert_error);
    g.arc(ins[0], del[0], delete_error);
    g.arc(ins[0], mat[0], stay);
    g.arc(del[last], unit_end, 1 - insert_error);
    g.arc(del[last], ins[last + 1], insert_error);
    g.arc(mat[last], unit_end, 1 - insert_error);
    g.arc(mat[last], ins[last + 1], insert_error);
    g.arc(ins[last + 1], ins[last + 1], insert_error);
    g.arc(ins[last + 1], unit_end, 1 - insert_error);
    for (int i = 0; i < F; ++i) {
        g.arc(mat[i], ins[i + 1], insert_error);
        g.arc(del[i], ins[i + 1], insert_error);
        g.arc(ins[i + 1], ins[i + 1], insert_error);
        if (i < F - 1) {
            g.arc(ins[i + 1], mat[i + 1], stay);
            g.arc(ins[i + 1], del[i + 1], delete_error);
            if (early_exit) {
                g.arc(mat[i], mat[i + 1], 1 - insert_error - delete_error - 0.01);
                g.arc(mat[i], del[i + 1], delete_error);
                g.arc(mat[i], unit_end, 0.01);
            } else {
                g.arc(mat[i], mat[i + 1], stay);
                g.arc(mat[i], del[i + 1], delete_error);
            }
            g.arc(del[i], del[i + 1], delete_error);
            g.arc(del[i], mat[i + 1], stay);
        }
    }
    g.bake();
    return g;
}

static Net repeat_block(const Profile &P, int copies)
{
    Net g("Repeating Pattern Matcher HMM Model");
    const int L = P.L;
    if (copies < 1) throw std::invalid_argument("copies must be >= 1");
    int last_end = -1;
    for (int rep = 0; rep < copies; ++rep) {
        const std::string tag = "_" + std::to_string(rep);
        std::vector<int> ins(L + 1), mat(L + 1), del(L + 1);          // mat/del indexed 1..L
        for (int i = 0; i <= L; ++i) ins[i] = g.add_vertex("I" + std::to_string(i) + tag, g.add_emission(P.emit_I[i]));
        for (int i = 1; i <= L; ++i) mat[i] = g.add_vertex("M" + std::to_string(i) + tag, g.add_emission(P.emit_M[i]));
        for (int i = 1; i <= L; ++i) del[i] = g.add_vertex("D" + std::to_string(i) + tag, -1);
        const int unit_start = g.add_vertex("unit_start" + tag, -1);
        const int unit_end = g.add_vertex("unit_end" + tag, -1);
        if (rep > 0) g.arc(last_end, unit_start, 1);
        else g.arc(g.start, unit_start, 1);
        if (rep == copies - 1) g.arc(unit_end, g.end, 1);
        g.arc(unit_start, mat[1], P.from_unit_start[1]);
        g.arc(unit_start, del[1], P.from_unit_start[2]);
        g.arc(unit_start, ins[0], P.from_unit_start[0]);
        g.arc(ins[0], ins[0], P.from_I[0][0]);
        g.arc(ins[0], del[1], P.from_I[0][2]);
        g.arc(ins[0], mat[1], P.from_I[0][1]);
        g.arc(del[L], unit_end, P.from_D[L][3]);
        g.arc(del[L], ins[L], P.from_D[L][0]);
        g.arc(mat[L], unit_end, P.from_M[L][3]);
        g.arc(mat[L], ins[L], P.from_M[L][0]);
        g.arc(ins[L], ins[L], P.from_I[L][0]);
        g.arc(ins[L], unit_end, P.from_I[L][3]);
        for (int i = 1; i <= L; ++i) {
            g.arc(mat[i], ins[i], P.from_M[i][0]);
            g.arc(del[i], ins[i], P.from_D[i][0]);
            g.arc(ins[i], ins[i], P.from_I[i][0]);
            if (i < L) {
                g.arc(ins[i], mat[i + 1], P.from_I[i][1]);
                g.arc(ins[i], del[i + 1], P.from_I[i][2]);
                g.arc(mat[i], mat[i + 1], P.from_M[i][1]);
                g.arc(mat[i], del[i + 1], P.from_M[i][2]);
                g.arc(del[i], mat[i + 1], P.from_D[i][1]);
                g.arc(del[i], del[i + 1], P.from_D[i][2]);
            }
        }
        last_end = unit_end;
    }
    g.bake();
    return g;
}

static inline bool starts_with(const std::string &s, const char *p) { return s.compare(0, std::char_traits<char>::length(p), p) == 0; }
static inline bool ends_with(const std::string &s, const char *p)
{
    const size_t n = std::char_traits<char>::length(p);
    return s.size() >= n && s.compare(s.size() - n, n, p) == 0;
}

// every copy may be the last one: unit_end_k -> {its successor, end_repeating_pattern_match} 0.5 / 0.5
static Net open_repeat_block(const Net &fixed, ExpFn exp_fn, void *user)
{
    const int count = (int)fixed.v.size();
    Rows rows = fixed.probability_rows(exp_fn, user, 2);
    const int start_rep = count, end_rep = count + 1;
    const int first_unit_start = row_last_nonzero(rows[fixed.start_index]);
    row_set(rows[fixed.start_index], first_unit_start, 0.0);
    row_set(rows[fixed.start_index], start_rep, 1);
    row_set(rows[start_rep], first_unit_start, 1);
    for (int i = 0; i < count; ++i) {
        if (!starts_with(fixed.v[fixed.order[i]].name, "unit_end")) continue;
        const int next_state = row_last_nonzero(rows[i]);
        row_set(rows[i], next_state, 0.5);
        row_set(rows[i], end_rep, 0.5);
    }
    row_set(rows[end_rep], fixed.end_index, 1);
    return rebuild_from_rows(fixed, rows, {"start_repeating_pattern_match", "end_repeating_pattern_match"},
                             fixed.start_index, "Repeat Matcher HMM Model");
}

static Net read_matcher(const std::string &left, const std::string &right, const std::vector<std::string> &aligned_repeats,
                        int copies, double max_error_rate, ExpFn exp_fn, void *user)
{
    Net model = flank_block(left, "suffix", "Suffix Matcher HMM Model", true, false, max_error_rate);
    const Profile P = estimate_profile(aligned_repeats, max_error_rate);
    const Net repeats = open_repeat_block(repeat_block(P, copies), exp_fn, user);
    const Net right_block = flank_block(right, "prefix", "Prefix Matcher HMM Model", false, true, max_error_rate);
    model.append(repeats);
    model.append(right_block);
    model.bake();

    const int n = (int)model.v.size();
    Rows rows = model.probability_rows(exp_fn, user, 0);
    std::vector<int> first_repeat_matches, repeat_match_states;
    int suffix_start = -1;
    for (int i = 0; i < n; ++i) {
        const std::string &nm = model.v[model.order[i]].name;
        const size_t us = nm.rfind('_');
        const std::string tail = us == std::string::npos ? nm : nm.substr(us + 1);
        if (nm[0] == 'M' && tail == "0") first_repeat_matches.push_back(i);
        if (nm[0] == 'M' && tail != "prefix" && tail != "suffix") repeat_match_states.push_back(i);
        if (nm == "suffix_start_suffix") suffix_start = i;
    }
    auto &start_row = rows[model.start_index];
    row_set(start_row, suffix_start, 0.3);
    for (int idx : first_repeat_matches) row_set(start_row, idx, 0.7 / first_repeat_matches.size());
    for (int idx : repeat_match_states) {
        const double to_end = 0.7 / repeat_match_states.size();
        const double total = 1 + to_end;
        for (auto &cv : rows[idx])
            if (cv.second != 0) cv.second = cv.second / total;
        row_set(rows[idx], model.end_index, to_end / total);
    }
    return rebuild_from_rows(model, rows, {}, model.start_index, "Read Matcher");
}

// ---- the arrays advntr_hmm_create takes ----------------------------------------------------------------------
struct Built {
    int32_t m = 0, silent_start = 0, start_index = 0, end_index = 0;
    std::vector<int32_t> in_ptr, in_src;
    std::vector<double> in_logp, emis;
    std::vector<uint16_t> state_class;
    std::string names;                                   // '\n'-joined, baked order
};


// the string tests of advntr/hmm_utils.py:116-286 as class bits (same table as the Python host's state_class_from_name)
static uint16_t classify(const std::string &nm)
{
    // the states of a read matcher are M<k>_<tag>, I<k>_<tag>, D<k>_<tag> by the thousand and a few dozen connectors: the
    // former need two looks at the name, the latter take the general tests below
    if (nm.size() >= 3 && (nm[0] == 'M' || nm[0] == 'I' || nm[0] == 'D') && nm[1] >= '0' && nm[1] <= '9') {
        uint16_t c = nm[0] == 'D' ? 0 : ADVNTR_SC_EMIT;
        if (nm[0] == 'M') c |= ADVNTR_SC_MATCH;
        const size_t us = nm.find('_');
        if (us != std::string::npos && nm.find('_', us + 1) == std::string::npos) {
            const size_t tl = nm.size() - us - 1;
            const char *t = nm.c_str() + us + 1;
            if (tl == 6 && memcmp(t, "suffix", 6) == 0) return c | ADVNTR_SC_SUFFIX | ADVNTR_SC_FIX;
            if (tl == 6 && memcmp(t, "prefix", 6) == 0) return c | ADVNTR_SC_PREFIX | ADVNTR_SC_FIX;
            bool digits = tl > 0;
            for (size_t i = 0; i < tl; ++i) digits = digits && t[i] >= '0' && t[i] <= '9';
            if (digits) return c;                      // a repeat-copy state: none of the other words occur in its name
        }
    }
    uint16_t c = 0;
    if (nm[0] == 'M' || nm[0] == 'I' || starts_with(nm, "start_random_matches") || starts_with(nm, "end_random_matches")) c |= ADVNTR_SC_EMIT;
    if (nm[0] == 'M') c |= ADVNTR_SC_MATCH;
    if (ends_with(nm, "suffix")) c |= ADVNTR_SC_SUFFIX;
    if (ends_with(nm, "prefix")) c |= ADVNTR_SC_PREFIX;
    if (starts_with(nm, "unit_start")) c |= ADVNTR_SC_UNIT_START;
    if (starts_with(nm, "unit_end")) c |= ADVNTR_SC_UNIT_END;
    if (nm.find("start") != std::string::npos || nm.find("end") != std::string::npos) c |= ADVNTR_SC_SKIP;
    if (ends_with(nm, "fix")) c |= ADVNTR_SC_FIX;
    return c;
}

static Built export_baked(const Net &g, const std::string &left, const std::string &right)
{
    Built B;
    const int n = (int)g.v.size();
    B.m = n;
    B.silent_start = g.silent_start;
    B.start_index = g.start_index;
    B.end_index = g.end_index;
    // CSR in-edges: arcs in graph order (vertex insertion order, then adjacency order), stable by destination
    B.in_ptr.assign(n + 1, 0);
    for (int a = 0; a < n; ++a)
        for (const Arc &e : g.out[a]) B.in_ptr[g.pos[e.to] + 1] += 1;
    for (int i = 0; i < n; ++i) B.in_ptr[i + 1] += B.in_ptr[i];
    const int E = B.in_ptr[n];
    B.in_src.resize(E);
    B.in_logp.resize(E);
    std::vector<int32_t> fill(B.in_ptr.begin(), B.in_ptr.end() - 1);
    for (int a = 0; a < n; ++a)
        for (const Arc &e : g.out[a]) {
            const int k = fill[g.pos[e.to]]++;
            B.in_src[k] = g.pos[a];
            B.in_logp[k] = e.logp;
        }
    B.emis.resize((size_t)g.silent_start * 4);
    B.state_class.resize(n);
    for (int i = 0; i < n; ++i) {
        const Vertex &x = g.v[g.order[i]];
        if (i < g.silent_start)
            for (int b = 0; b < 4; ++b) B.emis[(size_t)i * 4 + b] = g.emissions[x.emission][b];
        uint16_t c = classify(x.name);
        if ((c & ADVNTR_SC_MATCH) && (c & (ADVNTR_SC_SUFFIX | ADVNTR_SC_PREFIX))) {
            const std::string &flank = (c & ADVNTR_SC_SUFFIX) ? left : right;
            const int at = std::atoi(x.name.c_str() + 1) - 1;
            if (at >= 0 && at < (int)flank.size() && base_code(flank[at]) >= 0)
                c |= ADVNTR_SC_BASE_VALID | (uint16_t)(base_code(flank[at]) << ADVNTR_SC_BASE_SHIFT);
        }
        B.state_class[i] = c;
        if (i) B.names.push_back('\n');
        B.names += x.name;
    }
    return B;
}

static Built build_read_matcher(const std::string &left, const std::string &right, const std::vector<std::string> &aligned_repeats,
                                int copies, double max_error_rate, ExpFn exp_fn, void *user)
{
    return export_baked(read_matcher(left, right, aligned_repeats, copies, max_error_rate, exp_fn, user), left, right);
}

}  // namespace mb
