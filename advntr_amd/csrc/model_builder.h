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
#include <memory>
#include <memory_resource>
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
// for an argument is the value it returns the next time.  Direct-mapped; one table per thread (it sits in the build's Arena:
// a thread_local of its own costs a __tls_get_addr call per logarithm inside a shared library).
struct LogMemo {
    struct Entry { uint64_t key; double val; };
    Entry memo[1024] = {};                              // (key 0 = the bits of +0.0, which never gets here)
    double operator()(double x)                         // utils.pyx:64-70 (log_or_ninf)
    {
        if (!(x > 0)) return -std::numeric_limits<double>::infinity();
        uint64_t bits;
        memcpy(&bits, &x, 8);
        Entry &e = memo[(bits ^ (bits >> 17) ^ (bits >> 41)) & 1023u];
        if (e.key != bits) { e.key = bits; e.val = std::log(x); }
        return e.val;
    }
};

static inline int base_code(char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// A state name, zero-padded in place: no allocation per state, a copy is a memcpy, and memcmp over the whole buffer orders
// two names exactly as std::string / Python compare them (a name that is a prefix of another sorts first: 0 < any character).
// The longest names are the start / end states of the blocks ("Repeating Pattern Matcher HMM Model-start": 41 characters).
struct Name {
    static constexpr int CAP = 48;
    char s[CAP];
    int len = 0;
    Name() { memset(s, 0, CAP); }
    explicit Name(const char *z) { memset(s, 0, CAP); put(z); }
    int size() const { return len; }
    Name &put(const char *z) { return put(z, strlen(z)); }
    Name &put(const Name &o) { return put(o.s, (size_t)o.len); }
    Name &put(const char *z, size_t n)
    {
        if ((size_t)len + n >= (size_t)CAP) throw std::length_error("state name too long");
        memcpy(s + len, z, n);
        len += (int)n;
        return *this;
    }
    Name &put(char c) { const char z[2] = {c, 0}; return put(z); }
    Name &put(int x)
    {
        char z[16];
        int k = 15;
        z[k] = 0;
        unsigned u = x < 0 ? 0u - (unsigned)x : (unsigned)x;
        do { z[--k] = (char)('0' + u % 10); u /= 10; } while (u);
        if (x < 0) z[--k] = '-';
        return put(z + k);
    }
    bool operator<(const Name &o) const { return memcmp(s, o.s, CAP) < 0; }
    uint64_t key() const                               // the first eight characters, ordered like the name
    {
        uint64_t k;
        memcpy(&k, s, 8);
        return __builtin_bswap64(k);
    }
    bool operator==(const char *z) const { return strncmp(s, z, CAP) == 0; }
    bool starts_with(const char *z) const { return strncmp(s, z, strlen(z)) == 0; }
};

// The memory of one build.  A model passes through six nets and two sets of probability rows on its way, about a megabyte of
// arrays that live for a fraction of a millisecond; from the general allocator that is page faults and trimming on every locus
// (a third of the build's time, measured).  Everything the builder allocates besides its result therefore comes from a
// per-thread bump allocator over a buffer that is touched once and rewound per locus (what does not fit goes upstream).
// State names are written once into it and then only pointed at: the nets share them.
#ifndef ADVNTR_ARENA_PLAIN
struct Arena {
    std::unique_ptr<unsigned char[]> buffer;
    std::pmr::monotonic_buffer_resource res;
    LogMemo log;
    explicit Arena(size_t bytes = (size_t)6 << 20) : buffer(new unsigned char[bytes]), res(buffer.get(), bytes) {}
    const Name *keep(const Name &nm)
    {
        Name *slot = (Name *)res.allocate(sizeof(Name), alignof(Name));
        *slot = nm;
        return slot;
    }
    void rewind() { res.release(); }
};
#else
// Sanitizer builds (tests/test_native_sanitizers.py): every array a block of its own from the general allocator, so that an
// access past its end is one AddressSanitizer sees -- inside the bump buffer it would land in a neighbour unnoticed.
struct Arena {
    struct Plain : std::pmr::memory_resource {
        void *do_allocate(size_t bytes, size_t align) override { return std::pmr::new_delete_resource()->allocate(bytes, align); }
        void do_deallocate(void *p, size_t bytes, size_t align) override { std::pmr::new_delete_resource()->deallocate(p, bytes, align); }
        bool do_is_equal(const std::pmr::memory_resource &o) const noexcept override { return this == &o; }
    } res;
    std::vector<std::unique_ptr<Name>> names;
    LogMemo log;
    explicit Arena(size_t = 0) {}
    const Name *keep(const Name &nm)
    {
        names.emplace_back(new Name(nm));
        return names.back().get();
    }
    void rewind() { names.clear(); }
};
#endif
template <class T> using Vec = std::pmr::vector<T>;

struct Arc { int to; int next; double logp; };       // next: the vertex's following arc in the pool (-1 = last)
struct Vertex {
    const Name *name;
    uint64_t key;                                    // name->key(): most comparisons of a sort end here
    int emission;                                    // row of Net::emissions, -1 = silent
};
static inline bool name_less(const Vertex &a, const Vertex &b)
{
    return a.key != b.key ? a.key < b.key : *a.name < *b.name;
}

// Sparse probability rows, columns ascending, in ONE pool: a row owns a slice with a little slack (the edits of the callers add
// a column or two to most rows) and moves to the end of the pool when it outgrows it.
struct Rows {
    struct Cell { int col; double val; };
    Vec<int> off, cnt, cap;
    Vec<Cell> pool;
    explicit Rows(std::pmr::memory_resource *mr) : off(mr), cnt(mr), cap(mr), pool(mr) {}
    int n() const { return (int)off.size(); }
    Cell *row(int i) { return pool.data() + off[i]; }
    const Cell *row(int i) const { return pool.data() + off[i]; }
    void add_row(int capacity)
    {
        off.push_back((int)pool.size());
        cnt.push_back(0);
        cap.push_back(capacity);
        pool.resize(pool.size() + (size_t)capacity);
    }
    void push(int i, int col, double val) { pool[(size_t)off[i] + cnt[i]++] = Cell{col, val}; }    // (within the capacity)
    void set(int i, int col, double val)
    {
        Cell *r = row(i);
        int lo = 0, hi = cnt[i];
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (r[mid].col < col) lo = mid + 1;
            else hi = mid;
        }
        if (lo < cnt[i] && r[lo].col == col) { r[lo].val = val; return; }
        if (cnt[i] == cap[i]) {
            const int grown = std::max(4, 2 * cap[i]), at = (int)pool.size();
            pool.resize(pool.size() + (size_t)grown);
            std::copy(pool.begin() + off[i], pool.begin() + off[i] + cnt[i], pool.begin() + at);
            off[i] = at;
            cap[i] = grown;
            r = row(i);
        }
        for (int k = cnt[i]; k > lo; --k) r[k] = r[k - 1];
        r[lo] = Cell{col, val};
        ++cnt[i];
    }
    int last_nonzero(int i) const
    {
        const Cell *r = row(i);
        for (int k = cnt[i]; k-- > 0;)
            if (r[k].val != 0) return r[k].col;
        throw std::runtime_error("state without successors");
    }
};

// Insertion-ordered directed graph + the baked numbering of it.  The successors of a vertex are a chain through one pool of
// arcs (in the order they were added -- the order the reference's graph library hands them out in).
struct Net {
    Arena *arena;
    Vec<Vertex> v;
    Vec<int> head, tail;                              // first / last arc of a vertex in `arcs` (-1 = none)
    Vec<Arc> arcs;
    Vec<std::array<double, 4>> emissions;             // log-probabilities of A,C,G,T
    int start = -1, end = -1;
    // after bake()
    Vec<int> order, pos;
    int silent_start = 0, start_index = -1, end_index = -1;

    Net(Arena &ar, const char *nm, size_t vertices, size_t emission_rows)
        : arena(&ar), v(&ar.res), head(&ar.res), tail(&ar.res), arcs(&ar.res), emissions(&ar.res), order(&ar.res), pos(&ar.res)
    {
        v.reserve(vertices + 4);
        head.reserve(vertices + 4);
        tail.reserve(vertices + 4);
        arcs.reserve(4 * vertices + 8);
        emissions.reserve(emission_rows);
        start = add_vertex(Name(nm).put("-start"), -1);
        end = add_vertex(Name(nm).put("-end"), -1);
    }
    // a copy whose arrays live in another arena (its names stay where they are: that arena must outlive the copy)
    Net(const Net &o, Arena &into)
        : arena(&into), v(o.v, &into.res), head(o.head, &into.res), tail(o.tail, &into.res), arcs(o.arcs, &into.res),
          emissions(o.emissions, &into.res), start(o.start), end(o.end), order(o.order, &into.res), pos(o.pos, &into.res),
          silent_start(o.silent_start), start_index(o.start_index), end_index(o.end_index)
    {
    }
    int add_vertex(const Name &nm, int emission) { return add_vertex(arena->keep(nm), emission); }
    int add_vertex(const Name *kept, int emission)     // a name the arena holds already
    {
        v.push_back(Vertex{kept, kept->key(), emission});
        head.push_back(-1);
        tail.push_back(-1);
        return (int)v.size() - 1;
    }
    int add_emission(const std::array<double, 4> &prob)
    {
        std::array<double, 4> lp;
        for (int i = 0; i < 4; ++i) lp[i] = arena->log(prob[i]) + 0.0;   // + log(state weight 1), hmm.pyx:928-930
        emissions.push_back(lp);
        return (int)emissions.size() - 1;
    }
    void link(int a, int b, double lp)
    {
        const int k = (int)arcs.size();
        arcs.push_back(Arc{b, -1, lp});
        if (tail[a] < 0) head[a] = k;
        else arcs[tail[a]].next = k;
        tail[a] = k;
    }
    // add_transition: a repeated (a,b) keeps its first position and takes the new value
    void arc(int a, int b, double prob)
    {
        const double lp = arena->log(prob);
        for (int k = head[a]; k >= 0; k = arcs[k].next)
            if (arcs[k].to == b) { arcs[k].logp = lp; return; }
        link(a, b, lp);
    }
    void arc_new(int a, int b, double prob) { link(a, b, arena->log(prob)); }       // (a, b) known not to be there yet
    size_t n_arcs() const { return arcs.size(); }
    int degree(int a) const
    {
        int d = 0;
        for (int k = head[a]; k >= 0; k = arcs[k].next) ++d;
        return d;
    }

    // concatenate(other): vertices and arcs of `other` follow ours, end -> other.start with probability 1
    void append(const Net &other)
    {
        const int shift = (int)v.size(), eshift = (int)emissions.size(), ashift = (int)arcs.size();
        v.reserve(v.size() + other.v.size());
        for (const Vertex &x : other.v) v.push_back(Vertex{x.name, x.key, x.emission < 0 ? -1 : x.emission + eshift});
        for (int h : other.head) head.push_back(h < 0 ? -1 : h + ashift);
        for (int t : other.tail) tail.push_back(t < 0 ? -1 : t + ashift);
        arcs.reserve(arcs.size() + other.arcs.size() + 2);
        for (const Arc &e : other.arcs) arcs.push_back(Arc{e.to + shift, e.next < 0 ? -1 : e.next + ashift, e.logp});
        emissions.insert(emissions.end(), other.emissions.begin(), other.emissions.end());
        arc(end, other.start + shift, 1.0);
        end = other.end + shift;
    }

    // bake(merge=None): emitting vertices sorted by name, then the silent ones name-sorted and put in the
    // reversed post-order of an iterative DFS that stacks successors in adjacency order (networkx 1.11)
    void bake()
    {
        const int n = (int)v.size();
        std::pmr::memory_resource *mr = &arena->res;
        Vec<int> emitting(mr), silent(mr);
        emitting.reserve(n);
        silent.reserve(n);
        for (int i = 0; i < n; ++i) (v[i].emission >= 0 ? emitting : silent).push_back(i);
        auto by_name = [&](int a, int b) { return name_less(v[a], v[b]); };
        // (a net rebuilt from rows lists its emitting states in the previous bake's order: sorted already)
        if (!std::is_sorted(emitting.begin(), emitting.end(), by_name)) std::stable_sort(emitting.begin(), emitting.end(), by_name);
        if (!std::is_sorted(silent.begin(), silent.end(), by_name)) std::stable_sort(silent.begin(), silent.end(), by_name);
        Vec<char> seen(n, 0, mr), done(n, 0, mr);
        Vec<int> post(mr), stack(mr);
        post.reserve(silent.size());
        stack.reserve(64);
        for (int root : silent) {
            if (done[root]) continue;
            stack.assign(1, root);
            while (!stack.empty()) {
                const int w = stack.back();
                if (done[w]) { stack.pop_back(); continue; }
                seen[w] = 1;
                const size_t before = stack.size();
                for (int k = head[w]; k >= 0; k = arcs[k].next) {
                    const int to = arcs[k].to;
                    if (v[to].emission >= 0 || done[to]) continue;
                    if (seen[to]) throw std::runtime_error("silent states form a cycle");
                    stack.push_back(to);
                }
                if (stack.size() == before) { done[w] = 1; post.push_back(w); stack.pop_back(); }
            }
        }
        order = std::move(emitting);
        silent_start = (int)order.size();
        order.insert(order.end(), post.rbegin(), post.rend());
        pos.assign(n, -1);
        for (int i = 0; i < n; ++i) pos[order[i]] = i;
        start_index = pos[start];
        end_index = pos[end];
    }

    // dense_transition_matrix, sparsely: row i = out-arcs of baked state i as (baked column, exp(logp))
    Rows probability_rows(ExpFn exp_fn, void *user, int extra_rows) const
    {
        const int n = (int)v.size();
        std::pmr::memory_resource *mr = &arena->res;
        Vec<double> lp(mr), p(mr);
        lp.reserve(n_arcs());
        for (int i = 0; i < n; ++i)
            for (int k = head[order[i]]; k >= 0; k = arcs[k].next) lp.push_back(arcs[k].logp);
        p.resize(lp.size());
        {   // exp over the DISTINCT arguments only (a few hundred of several thousand; the caller's exp -- numpy's loop -- gives
            // an argument the same value wherever it stands in the array, which the bit-identical goldens already rest on)
            Vec<uint64_t> keys(2048, ~0ull, mr);                     // open addressing on the bit patterns (NaN never occurs)
            Vec<int32_t> slot_of(2048, -1, mr), which(lp.size(), 0, mr);
            Vec<double> uniq(mr), uexp(mr);
            uniq.reserve(1536);
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
        Rows rows(mr);
        rows.off.reserve(n + extra_rows);
        rows.cnt.reserve(n + extra_rows);
        rows.cap.reserve(n + extra_rows);
        rows.pool.reserve(lp.size() + 2 * (size_t)(n + extra_rows) + 64);
        size_t k = 0;
        for (int i = 0; i < n; ++i) {
            const int u = order[i];
            rows.add_row(degree(u) + 2);
            for (int a = head[u]; a >= 0; a = arcs[a].next) rows.push(i, pos[arcs[a].to], p[k++]);
            Rows::Cell *r = rows.row(i);
            const int c = rows.cnt[i];
            if (c <= 8) {                                // (three or four successors as a rule)
                for (int x = 1; x < c; ++x) {
                    const Rows::Cell cell = r[x];
                    int y = x;
                    while (y > 0 && r[y - 1].col > cell.col) { r[y] = r[y - 1]; --y; }
                    r[y] = cell;
                }
            } else
                std::sort(r, r + c, [](const Rows::Cell &a, const Rows::Cell &b) { return a.col < b.col; });
        }
        for (int i = 0; i < extra_rows; ++i) rows.add_row(2);
        return rows;
    }
};

// from_matrix(rows, distributions, starts = e_{start_col}, ends = e_{end_col}, state_names) + bake.  `from` supplies
// names and emissions of the first from.v.size() states (in baked order); `extra` names the silent ones after them.
static Net rebuild_from_rows(const Net &from, const Rows &rows, const std::vector<const char *> &extra,
                             int start_col, const char *name)
{
    const int n = rows.n(), base = 2;
    Net g(*from.arena, name, (size_t)n, 0);
    g.arcs.reserve(rows.pool.size() + 8);
    g.emissions = from.emissions;
    for (int i = 0; i < (int)from.v.size(); ++i) {
        const Vertex &x = from.v[from.order[i]];
        g.add_vertex(x.name, x.emission);
    }
    for (const char *nm : extra) g.add_vertex(Name(nm), -1);
    g.arc_new(g.start, base + start_col, 1.0);
    for (int i = 0; i < n; ++i) {
        const Rows::Cell *r = rows.row(i);
        for (int k = 0; k < rows.cnt[i]; ++k)
            if (r[k].val != 0) g.arc_new(base + i, base + r[k].col, r[k].val);
    }
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
static Net build_flank_block(Arena &pool, const std::string &pattern, const char *tag, const char *model_name,
                             bool enter_anywhere, bool early_exit, double max_error_rate)
{
    const int F = (int)pattern.size();
    if (F == 0) throw std::invalid_argument("empty flanking region");
    Net g(pool, model_name, 3 * (size_t)F + 3, 5);
    const int uniform = g.add_emission({0.25, 0.25, 0.25, 0.25});
    int match_emission[4];
    for (int b = 0; b < 4; ++b) {
        std::array<double, 4> p = {0.01, 0.01, 0.01, 0.01};
        p[b] = 0.97;
        match_emission[b] = g.add_emission(p);
    }
    const Name suffix = Name("_").put(tag);
    Vec<int> ins(F + 1, 0, &pool.res), mat(F, 0, &pool.res), del(F, 0, &pool.res);
    for (int i = 0; i <= F; ++i) ins[i] = g.add_vertex(Name("I").put(i).put(suffix), uniform);
    for (int i = 0; i < F; ++i) {
        const int b = base_code(pattern[i]);
        if (b < 0) throw std::invalid_argument(std::string("symbol '") + pattern[i] + "' in a flanking region is not one of ACGT");
        mat[i] = g.add_vertex(Name("M").put(i + 1).put(suffix), match_emission[b]);
    }
    for (int i = 0; i < F; ++i) del[i] = g.add_vertex(Name("D").put(i + 1).put(suffix), -1);
    const int unit_start = g.add_vertex(Name(tag).put("_start_").put(tag), -1);
    const int unit_end = g.add_vertex(Name(tag).put("_end_").put(tag), -1);
    const int last = F - 1;
    g.arc(g.start, unit_start, 1);
    g.arc(unit_end, g.end, 1);
    const double insert_error = max_error_rate * 2 / 5;
    const double delete_error = max_error_rate * 1 / 5;
    const double stay = 1 - insert_error - delete_error;
    if (enter_anywhere) {
        g.arc(unit_start, del[0], delete_error);
        g.arc(unit_start, ins[0], insert_error);
        for (int i = 0; i < F; ++i) g.arc_new(unit_start, mat[i], (1 - insert_error - delete_error) / F);   // (F distinct targets)
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

// A flank block is the same for every flanking region of one length but for WHICH of the four match emissions each M state
// carries (the names, the arcs and hence the baked order do not look at the bases): the loci of a database share a handful of
// flank lengths, so each thread keeps the blocks it has built (baked, in an arena of their own that is never rewound) and a
// locus takes a copy with its bases' emissions filled in.
struct FlankBlocks {
    struct Entry {
        int F;
        const char *tag;
        bool enter_anywhere, early_exit;
        double rate;
        Net net;
    };
    static constexpr size_t MAX_ENTRIES = 16;
    static constexpr int MAX_LENGTH = 1024;
    Arena arena{(size_t)2 << 20};
    std::vector<Entry> entries;
};

static Net flank_block(Arena &pool, const std::string &pattern, const char *tag, const char *model_name,
                       bool enter_anywhere, bool early_exit, double max_error_rate)
{
    const int F = (int)pattern.size();
    if (F == 0) throw std::invalid_argument("empty flanking region");
    for (int i = 0; i < F; ++i)
        if (base_code(pattern[i]) < 0)
            throw std::invalid_argument(std::string("symbol '") + pattern[i] + "' in a flanking region is not one of ACGT");
    static thread_local FlankBlocks kept;
    const Net *base = nullptr;
    for (const FlankBlocks::Entry &e : kept.entries)
        if (e.F == F && e.enter_anywhere == enter_anywhere && e.early_exit == early_exit && e.rate == max_error_rate &&
            strcmp(e.tag, tag) == 0) {
            base = &e.net;
            break;
        }
    if (!base) {
        if (F > FlankBlocks::MAX_LENGTH || kept.entries.size() >= FlankBlocks::MAX_ENTRIES)
            return build_flank_block(pool, pattern, tag, model_name, enter_anywhere, early_exit, max_error_rate);
        kept.entries.push_back(FlankBlocks::Entry{F, tag, enter_anywhere, early_exit, max_error_rate,
                                                  build_flank_block(kept.arena, std::string((size_t)F, 'A'), tag, model_name,
                                                                    enter_anywhere, early_exit, max_error_rate)});
        base = &kept.entries.back().net;
    }
    Net g(*base, pool);
    // vertices: start, end, I0..IF, M1..MF, ...; emissions: the uniform one, then the match emission of A, C, G, T
    for (int i = 0; i < F; ++i) g.v[(size_t)(2 + F + 1 + i)].emission = 1 + base_code(pattern[i]);
    return g;
}

static Net repeat_block(Arena &pool, const Profile &P, int copies)
{
    const int L = P.L;
    if (copies < 1) throw std::invalid_argument("copies must be >= 1");
    Net g(pool, "Repeating Pattern Matcher HMM Model", (size_t)copies * (3 * (size_t)L + 3), (size_t)copies * (2 * (size_t)L + 1));
    int last_end = -1;
    Vec<int> ins(L + 1, 0, &pool.res), mat(L + 1, 0, &pool.res), del(L + 1, 0, &pool.res);       // mat/del indexed 1..L
    for (int rep = 0; rep < copies; ++rep) {
        const Name tag = Name("_").put(rep);
        for (int i = 0; i <= L; ++i) ins[i] = g.add_vertex(Name("I").put(i).put(tag), g.add_emission(P.emit_I[i]));
        for (int i = 1; i <= L; ++i) mat[i] = g.add_vertex(Name("M").put(i).put(tag), g.add_emission(P.emit_M[i]));
        for (int i = 1; i <= L; ++i) del[i] = g.add_vertex(Name("D").put(i).put(tag), -1);
        const int unit_start = g.add_vertex(Name("unit_start").put(tag), -1);
        const int unit_end = g.add_vertex(Name("unit_end").put(tag), -1);
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

// every copy may be the last one: unit_end_k -> {its successor, end_repeating_pattern_match} 0.5 / 0.5
static Net open_repeat_block(const Net &fixed, ExpFn exp_fn, void *user)
{
    const int count = (int)fixed.v.size();
    Rows rows = fixed.probability_rows(exp_fn, user, 2);
    const int start_rep = count, end_rep = count + 1;
    const int first_unit_start = rows.last_nonzero(fixed.start_index);
    rows.set(fixed.start_index, first_unit_start, 0.0);
    rows.set(fixed.start_index, start_rep, 1);
    rows.set(start_rep, first_unit_start, 1);
    for (int i = 0; i < count; ++i) {
        if (!fixed.v[fixed.order[i]].name->starts_with("unit_end")) continue;
        const int next_state = rows.last_nonzero(i);
        rows.set(i, next_state, 0.5);
        rows.set(i, end_rep, 0.5);
    }
    rows.set(end_rep, fixed.end_index, 1);
    return rebuild_from_rows(fixed, rows, {"start_repeating_pattern_match", "end_repeating_pattern_match"},
                             fixed.start_index, "Repeat Matcher HMM Model");
}

static Net read_matcher(Arena &pool, const std::string &left, const std::string &right,
                        const std::vector<std::string> &aligned_repeats, int copies, double max_error_rate, ExpFn exp_fn, void *user)
{
    Net model = flank_block(pool, left, "suffix", "Suffix Matcher HMM Model", true, false, max_error_rate);
    const Profile P = estimate_profile(aligned_repeats, max_error_rate);
    const Net repeats = open_repeat_block(repeat_block(pool, P, copies), exp_fn, user);
    const Net right_block = flank_block(pool, right, "prefix", "Prefix Matcher HMM Model", false, true, max_error_rate);
    model.v.reserve(model.v.size() + repeats.v.size() + right_block.v.size());
    model.arcs.reserve(model.arcs.size() + repeats.arcs.size() + right_block.arcs.size() + 4);
    model.append(repeats);
    model.append(right_block);
    model.bake();

    const int n = (int)model.v.size();
    Rows rows = model.probability_rows(exp_fn, user, 0);
    Vec<int> first_repeat_matches(&pool.res), repeat_match_states(&pool.res);
    int suffix_start = -1;
    for (int i = 0; i < n; ++i) {
        const Name &nm = *model.v[model.order[i]].name;
        if (nm.s[0] == 'M') {                          // the part after the last '_' (the whole name when there is none)
            const int len = nm.size();
            int us = len;
            while (us > 0 && nm.s[us - 1] != '_') --us;
            const char *tail = nm.s + us;
            if (strcmp(tail, "0") == 0) first_repeat_matches.push_back(i);
            if (strcmp(tail, "prefix") != 0 && strcmp(tail, "suffix") != 0) repeat_match_states.push_back(i);
        } else if (nm == "suffix_start_suffix")
            suffix_start = i;
    }
    rows.set(model.start_index, suffix_start, 0.3);
    for (int idx : first_repeat_matches) rows.set(model.start_index, idx, 0.7 / first_repeat_matches.size());
    for (int idx : repeat_match_states) {
        const double to_end = 0.7 / repeat_match_states.size();
        const double total = 1 + to_end;
        Rows::Cell *r = rows.row(idx);
        for (int k = 0; k < rows.cnt[idx]; ++k)
            if (r[k].val != 0) r[k].val = r[k].val / total;
        rows.set(idx, model.end_index, to_end / total);
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


static inline bool starts_with(const std::string &s, const char *p) { return s.compare(0, std::char_traits<char>::length(p), p) == 0; }
static inline bool ends_with(const std::string &s, const char *p)
{
    const size_t n = std::char_traits<char>::length(p);
    return s.size() >= n && s.compare(s.size() - n, n, p) == 0;
}

// the string tests of advntr/hmm_utils.py:116-286 as class bits (same table as the Python host's state_class_from_name)
static uint16_t classify(const char *name, size_t len)
{
    // the states of a read matcher are M<k>_<tag>, I<k>_<tag>, D<k>_<tag> by the thousand and a few dozen connectors: the
    // former need two looks at the name, the latter take the general tests below
    if (len >= 3 && (name[0] == 'M' || name[0] == 'I' || name[0] == 'D') && name[1] >= '0' && name[1] <= '9') {
        uint16_t c = name[0] == 'D' ? 0 : ADVNTR_SC_EMIT;
        if (name[0] == 'M') c |= ADVNTR_SC_MATCH;
        const char *us = (const char *)memchr(name, '_', len);
        if (us && !memchr(us + 1, '_', len - (size_t)(us + 1 - name))) {
            const size_t tl = len - (size_t)(us - name) - 1;
            const char *t = us + 1;
            if (tl == 6 && memcmp(t, "suffix", 6) == 0) return c | ADVNTR_SC_SUFFIX | ADVNTR_SC_FIX;
            if (tl == 6 && memcmp(t, "prefix", 6) == 0) return c | ADVNTR_SC_PREFIX | ADVNTR_SC_FIX;
            bool digits = tl > 0;
            for (size_t i = 0; i < tl; ++i) digits = digits && t[i] >= '0' && t[i] <= '9';
            if (digits) return c;                      // a repeat-copy state: none of the other words occur in its name
        }
    }
    const std::string nm(name, len);
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
    for (const Arc &e : g.arcs) B.in_ptr[g.pos[e.to] + 1] += 1;
    for (int i = 0; i < n; ++i) B.in_ptr[i + 1] += B.in_ptr[i];
    const int E = B.in_ptr[n];
    B.in_src.resize(E);
    B.in_logp.resize(E);
    std::vector<int32_t> fill(B.in_ptr.begin(), B.in_ptr.end() - 1);
    for (int a = 0; a < n; ++a)
        for (int q = g.head[a]; q >= 0; q = g.arcs[q].next) {
            const Arc &e = g.arcs[q];
            const int k = fill[g.pos[e.to]]++;
            B.in_src[k] = g.pos[a];
            B.in_logp[k] = e.logp;
        }
    B.emis.resize((size_t)g.silent_start * 4);
    B.state_class.resize(n);
    B.names.reserve((size_t)n * 12);
    for (int i = 0; i < n; ++i) {
        const Vertex &x = g.v[g.order[i]];
        const size_t len = (size_t)x.name->size();
        if (i < g.silent_start)
            for (int b = 0; b < 4; ++b) B.emis[(size_t)i * 4 + b] = g.emissions[x.emission][b];
        uint16_t c = classify(x.name->s, len);
        if ((c & ADVNTR_SC_MATCH) && (c & (ADVNTR_SC_SUFFIX | ADVNTR_SC_PREFIX))) {
            const std::string &flank = (c & ADVNTR_SC_SUFFIX) ? left : right;
            const int at = std::atoi(x.name->s + 1) - 1;
            if (at >= 0 && at < (int)flank.size() && base_code(flank[at]) >= 0)
                c |= ADVNTR_SC_BASE_VALID | (uint16_t)(base_code(flank[at]) << ADVNTR_SC_BASE_SHIFT);
        }
        B.state_class[i] = c;
        if (i) B.names.push_back('\n');
        B.names.append(x.name->s, len);
    }
    return B;
}

static Built build_read_matcher(const std::string &left, const std::string &right, const std::vector<std::string> &aligned_repeats,
                                int copies, double max_error_rate, ExpFn exp_fn, void *user)
{
    static thread_local Arena arena;
    struct Rewind {                                     // (also when a block throws)
        Arena &a;
        ~Rewind() { a.rewind(); }
    } rewind{arena};
    return export_baked(read_matcher(arena, left, right, aligned_repeats, copies, max_error_rate, exp_fn, user), left, right);
}

}  // namespace mb
