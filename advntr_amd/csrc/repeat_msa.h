// Multiple alignment of the repeat units of one locus (host C++) -- the step the reference delegates to the external
// program `muscle` (advntr/profile_hmm.py:166-171) before estimating the profile parameters.
//
// PARITY UNPINNED: muscle is not in this image and its output is not reproduced here.  What the profile estimator
// needs from an alignment is modest -- which columns are match columns (< 50 % gaps), the per-column base counts and
// the M/I/D walk of every row (profile_hmm.py:13-53) -- and the repeat units of a VNTR are near-identical copies of
// one pattern, so a deterministic progressive alignment serves: the distinct units are added, most frequent first,
// to a growing column profile by global dynamic programming on integer scores.  Units of equal length that differ
// only by substitutions stay gap-free (a substitution costs less than a gap pair), which is the case every golden
// locus and synthetic workload of this repo is in; for other loci the parameters may differ from a muscle-based
// build in the columns muscle would have gapped differently.  The builder only uses this when the caller asks for
// it (ADVNTR_BUILD_ALIGN_REPEATS); without the flag unequal-length units are an error.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace msa {

static inline int sym(char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// Scores are per existing row (summed over the rows of a column, so they stay integers):
//   base on base: +2 equal, -1 different; base on a row that has a gap there: -1
//   '-' for the new unit in a column: -2 per row that has a base there (free against rows that also have a gap)
//   a new column for the new unit: -2 per existing row
struct Aligner {
    std::vector<std::array<long, 5>> col;   // counts of A,C,G,T,'-' per column
    std::vector<std::string> rows;          // aligned distinct units
    std::vector<long> weight;               // multiplicity of each
    long total = 0;

    void add(const std::string &s, long mult)
    {
        const int n = (int)s.size();
        if (rows.empty()) {
            col.assign(n, std::array<long, 5>{});
            for (int i = 0; i < n; ++i) col[i][sym(s[i])] += mult;
            rows.push_back(s);
            weight.push_back(mult);
            total = mult;
            return;
        }
        const int W = (int)col.size();
        // dp[i][j]: best score of s[0..i) against columns [0..j); move: 0 diagonal, 1 skip column, 2 new column
        std::vector<long> dp((size_t)(n + 1) * (W + 1));
        std::vector<uint8_t> mv((size_t)(n + 1) * (W + 1), 0);
        auto at = [&](int i, int j) -> size_t { return (size_t)i * (W + 1) + j; };
        auto skip_cost = [&](int j) { return -2 * (total - col[j][4]); };
        dp[at(0, 0)] = 0;
        for (int j = 1; j <= W; ++j) { dp[at(0, j)] = dp[at(0, j - 1)] + skip_cost(j - 1); mv[at(0, j)] = 1; }
        for (int i = 1; i <= n; ++i) {
            dp[at(i, 0)] = dp[at(i - 1, 0)] - 2 * total;
            mv[at(i, 0)] = 2;
            const int b = sym(s[i - 1]);
            for (int j = 1; j <= W; ++j) {
                const auto &c = col[j - 1];
                const long on = 3 * c[b] - (total - c[4]) - c[4];          // +2 equal, -1 different, -1 on gap rows
                long best = dp[at(i - 1, j - 1)] + on;
                uint8_t m = 0;
                const long sk = dp[at(i, j - 1)] + skip_cost(j - 1);
                if (sk > best) { best = sk; m = 1; }
                const long nw = dp[at(i - 1, j)] - 2 * total;
                if (nw > best) { best = nw; m = 2; }
                dp[at(i, j)] = best;
                mv[at(i, j)] = m;
            }
        }
        // trace back: the new row over the old columns plus the positions of new columns
        std::string row;
        std::vector<int> ops;                                               // per output column, right to left
        for (int i = n, j = W; i > 0 || j > 0;) {
            const uint8_t m = mv[at(i, j)];
            ops.push_back(m);
            if (m == 0) { --i; --j; } else if (m == 1) { --j; } else { --i; }
        }
        std::reverse(ops.begin(), ops.end());
        std::vector<std::array<long, 5>> ncol;
        std::vector<std::string> nrows(rows.size());
        int i = 0, j = 0;
        for (int m : ops) {
            if (m == 2) {                                                   // new column: old rows get a gap
                std::array<long, 5> c{};
                c[4] = total;
                c[sym(s[i])] += mult;
                ncol.push_back(c);
                for (auto &r : nrows) r.push_back('-');
                row.push_back(s[i++]);
            } else {
                std::array<long, 5> c = col[j];
                for (size_t r = 0; r < rows.size(); ++r) nrows[r].push_back(rows[r][j]);
                if (m == 0) { c[sym(s[i])] += mult; row.push_back(s[i++]); }
                else { c[4] += mult; row.push_back('-'); }
                ncol.push_back(c);
                ++j;
            }
        }
        col.swap(ncol);
        rows.swap(nrows);
        rows.push_back(row);
        weight.push_back(mult);
        total += mult;
    }
};

// Returns the aligned rows in the order of `units`.
static std::vector<std::string> align_units(const std::vector<std::string> &units)
{
    if (units.empty()) return {};
    std::map<std::string, std::pair<long, int>> distinct;                   // unit -> (multiplicity, first index)
    for (size_t k = 0; k < units.size(); ++k) {
        if (units[k].empty()) throw std::invalid_argument("empty repeat unit");
        for (char c : units[k])
            if (sym(c) < 0) throw std::invalid_argument(std::string("symbol '") + c + "' in a repeat unit is not one of ACGT");
        auto it = distinct.find(units[k]);
        if (it == distinct.end()) distinct.emplace(units[k], std::make_pair(1L, (int)k));
        else it->second.first += 1;
    }
    // modal length (by multiplicity; ties -> the shorter)
    std::map<size_t, long> by_len;
    for (const auto &d : distinct) by_len[d.first.size()] += d.second.first;
    size_t modal = 0;
    long modal_n = -1;
    for (const auto &b : by_len)
        if (b.second > modal_n) { modal = b.first; modal_n = b.second; }
    struct Item { const std::string *s; long mult; int first; long off; };
    std::vector<Item> order;
    for (const auto &d : distinct) {
        const long off = (long)d.first.size() - (long)modal;
        order.push_back(Item{&d.first, d.second.first, d.second.second, off < 0 ? -off : off});
    }
    std::sort(order.begin(), order.end(), [](const Item &a, const Item &b) {
        if (a.off != b.off) return a.off < b.off;                           // modal-length units first
        if (a.mult != b.mult) return a.mult > b.mult;                       // then the most frequent
        return a.first < b.first;
    });
    Aligner A;
    std::map<std::string, size_t> where;
    for (const Item &it : order) {
        A.add(*it.s, it.mult);
        where[*it.s] = A.rows.size() - 1;
    }
    std::vector<std::string> out;
    out.reserve(units.size());
    for (const std::string &u : units) out.push_back(A.rows[where[u]]);
    return out;
}

}  // namespace msa
