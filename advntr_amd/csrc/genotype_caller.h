// genotype_caller.h -- host C++: RU-count genotypes from the kernels' summary records, many loci at once.
//
// The step that follows scoring and recruitment in the Illumina path of the reference:
//   VNTRFinder.find_repeat_count_from_alignment_file, after read selection   /root/reference/advntr/vntr_finder.py:807-887
//   VNTRFinder.read_flanks_repeats_with_confidence                           /root/reference/advntr/vntr_finder.py:311-322
//   get_flanking_regions_matching_rate (its final division)                  /root/reference/advntr/hmm_utils.py:252-268
//   VNTRFinder.find_genotype_based_on_observed_repeats / get_conditional_likelihood   vntr_finder.py:473-532
// The reference runs it in Python per locus; at model-database scale (6 719 loci) that loop was a third of the end-to-end
// time of this build (0.95 s next to 0.17 s of kernels), so it runs here on host threads, straight on the 8-int records
// the device summariser writes (path_summary.h) -- no Viterbi path, no Python object per read.
//
// Arithmetic follows the reference operation for operation so that the probabilities come out bit-equal
// (tests/test_genotype_native.py against the goldens the reference's own methods produced): Python's `x ** k` on floats
// is C pow(x, (double)k); numpy.prod of a list is a left-to-right product; dict and Counter orders are insertion orders;
// sorted(..., reverse=True) is stable.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace gt {

struct Result {
    int32_t a = -1, b = -1;          // genotype as the reference returns it (order of its candidate pair); -1, -1 = None
    double max_prob = 1e-20;
    int32_t recruited = 0, spanning = 0, flanking = 0;
};

// vntr_finder.py:473-483
static inline double conditional_likelihood(int ck, int ci, int cj, double r, double r_e)
{
    if (ck == ci && ci == cj) return 1 - r;
    if (cj == 0) return 0.5 * (1 - r);
    if (ck == ci) return 0.5 * ((1 - r) + std::pow(r_e, (double)std::abs(ck - cj)));
    if (ck == cj) return 0.5 * ((1 - r) + std::pow(r_e, (double)std::abs(ck - ci)));
    return 0.5 * (std::pow(r_e, (double)std::abs(ck - ci)) + std::pow(r_e, (double)std::abs(ck - cj)));
}

// vntr_finder.py:485-532.  observed = RU counts in the order the reference would hold them.
static inline void genotype_from_observed(const std::vector<int> &observed, bool haploid, Result &out)
{
    // ru_counts: dict in first-occurrence order
    std::vector<std::pair<int, int>> counts;
    for (int cn : observed) {
        bool found = false;
        for (auto &kv : counts)
            if (kv.first == cn) { kv.second++; found = true; break; }
        if (!found) counts.emplace_back(cn, 1);
    }
    double priors;
    if (counts.size() < 2) {
        priors = 0.5;
        // ru_counts[0] = 1 (a dict: an observed 0 is overwritten, not added)
        bool found = false;
        for (auto &kv : counts)
            if (kv.first == 0) { kv.second = 1; found = true; }
        if (!found) counts.emplace_back(0, 1);
    } else {
        const double n = (double)counts.size();
        priors = 1.0 / (n * (n - 1) / 2);
    }
    std::stable_sort(counts.begin(), counts.end(),
                     [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.second > y.second; });
    const double r = 0.03;
    const double r_e = r / (2 + r);
    const int K = (int)counts.size();
    // posteriors in the insertion order of the reference's dict: (i, j >= i) lexicographic; each the left-to-right product
    // over the ck != 0 of likelihood ** occ, times the prior
    std::vector<double> post;
    std::vector<std::pair<int, int>> keys;
    bool any_ck = false;
    for (const auto &kv : counts) any_ck |= kv.first != 0;
    double total = 0.0;
    if (any_ck) {
        for (int i = 0; i < K; ++i)
            for (int j = i; j < K; ++j) {
                if (haploid && i != j) continue;
                const int ci = counts[i].first, cj = counts[j].first;
                // (a pair that already has an entry -- possible only when the same count appears twice, which a dict
                // rules out -- would extend that entry; keys are unique here)
                double prod = 0.0;
                bool first = true;
                for (const auto &kv : counts) {
                    if (kv.first == 0) continue;
                    const double term = std::pow(conditional_likelihood(kv.first, ci, cj, r, r_e), (double)kv.second);
                    prod = first ? term : prod * term;
                    first = false;
                }
                keys.emplace_back(ci, cj);
                post.push_back(prod * priors);
            }
        for (double p : post) total = total + p;          // sum(): 0 + p0 + p1 + ...
    }
    out.a = out.b = -1;
    out.max_prob = 1e-20;
    for (size_t k = 0; k < post.size(); ++k) {
        const double q = post[k] / total;                 // 0/0 = nan compares false, as numpy's scalar division does
        if (q > out.max_prob) { out.max_prob = q; out.a = keys[k].first; out.b = keys[k].second; }
    }
}

// hmm_utils.py:252-268 + vntr_finder.py:311-322 on a summary record {RU, matches, repeat_bp, left_bp, right_bp, left_match,
// right_match, path_len}
static inline bool read_spans_with_confidence(const int32_t *s, int min_left, int min_right)
{
    const double right_rate = s[4] != 0 ? (double)s[6] / (double)s[4] : 1.0;
    const double left_rate = s[3] != 0 ? (double)s[5] / (double)s[3] : 1.0;
    if (std::min(right_rate, left_rate) < 0.95) return false;
    return s[3] > min_left && s[4] > min_right;
}

// vntr_finder.py:807-887 without the coverage estimate (the caller handles average_coverage)
static inline void repeat_count_from_selected(const int32_t *summ, int64_t n, bool accuracy_filter, bool haploid, int min_left,
                                              int min_right, Result &out)
{
    std::vector<int> covered, flanking;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t *s = summ + 8 * i;
        if (read_spans_with_confidence(s, min_left, min_right)) covered.push_back(s[0]);
        else if (!accuracy_filter) flanking.push_back(s[0]);
    }
    out.recruited = (int32_t)n;
    out.flanking = (int32_t)flanking.size();
    std::sort(flanking.begin(), flanking.end());
    const int min_valid = covered.empty() ? 0 : *std::max_element(covered.begin(), covered.end());
    std::vector<int> max_flanking;
    if (!flanking.empty()) {
        const int mx = flanking.back();
        for (int v : flanking)
            if (v == mx && v >= min_valid) max_flanking.push_back(v);
    }
    if (max_flanking.size() < 5) max_flanking.clear();
    if (accuracy_filter) {
        // Counter(covered).most_common(): counts descending, first occurrence first among equals
        std::vector<std::pair<int, int>> cnt;
        for (int v : covered) {
            bool found = false;
            for (auto &kv : cnt)
                if (kv.first == v) { kv.second++; found = true; break; }
            if (!found) cnt.emplace_back(v, 1);
        }
        std::stable_sort(cnt.begin(), cnt.end(),
                         [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.second > y.second; });
        std::vector<int> modified;
        for (const auto &kv : cnt)
            if (kv.second >= 3) modified.insert(modified.end(), (size_t)kv.second, kv.first);
        covered.swap(modified);
        max_flanking.clear();
    }
    out.spanning = (int32_t)covered.size();
    covered.insert(covered.end(), max_flanking.begin(), max_flanking.end());
    genotype_from_observed(covered, haploid, out);
}

// vntr_finder.py:534-585 after the scoring loop: observed = the RU counts of the spanning reads in scoring order
static inline void dominant_copy_numbers(const int32_t *ru, int64_t n, bool accuracy_filter, bool haploid, Result &out)
{
    if (n < 1) { out.a = out.b = -1; out.max_prob = 0.0; return; }            // "There is no spanning read": (None, 0)
    std::vector<int> observed(ru, ru + n);
    if (accuracy_filter) {
        std::vector<std::pair<int, int>> cnt;
        for (int v : observed) {
            bool found = false;
            for (auto &kv : cnt)
                if (kv.first == v) { kv.second++; found = true; break; }
            if (!found) cnt.emplace_back(v, 1);
        }
        std::stable_sort(cnt.begin(), cnt.end(),
                         [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.second > y.second; });
        std::vector<int> modified;
        for (const auto &kv : cnt)
            if (kv.second >= 3) modified.insert(modified.end(), (size_t)kv.second, kv.first);
        observed.swap(modified);
    }
    out.recruited = out.spanning = (int32_t)n;
    genotype_from_observed(observed, haploid, out);
}

}  // namespace gt
