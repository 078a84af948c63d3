// Address/UB-sanitised run of the product's host-only C++ (the native model builder and the repeat aligner): GPU
// sanitizers are not available on the pool, so the host code is exercised here under -fsanitize=address,undefined.
// Built and run by tests/test_native_sanitizers.py with plain g++ (no HIP).
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "../../advntr_amd/csrc/model_builder.h"
#include "../../advntr_amd/csrc/repeat_msa.h"
#include "../../advntr_amd/csrc/column_program.h"

// what build_column_program reads of a model (engine.hip: advntr_hmm)
struct HostModel {
    int m, P, start, end, finite;
    std::vector<int32_t> in_ptr, in_src;
    std::vector<double> in_logp, emis;
};

static std::string dna(std::mt19937 &rng, int n)
{
    std::string s(n, 'A');
    for (char &c : s) c = "ACGT"[rng() & 3];
    return s;
}

int main()
{
    std::mt19937 rng(12345);
    long checksum = 0;
    for (int trial = 0; trial < 60; ++trial) {
        const int L = 1 + (int)(rng() % 40), flank = 1 + (int)(rng() % 120), copies = 1 + (int)(rng() % 8);
        const std::string base = dna(rng, L);
        std::vector<std::string> units;
        const int n_units = 1 + (int)(rng() % 7);
        for (int u = 0; u < n_units; ++u) {
            std::string s = base;
            for (int e = (int)(rng() % 3); e > 0 && !s.empty(); --e) {
                const size_t at = rng() % s.size();
                switch (rng() % 3) {
                    case 0: s[at] = "ACGT"[rng() & 3]; break;
                    case 1: if (s.size() > 1) s.erase(at, 1); break;
                    default: s.insert(at, 1, "ACGT"[rng() & 3]);
                }
            }
            units.push_back(s);
        }
        const std::vector<std::string> rows = msa::align_units(units);
        for (size_t i = 0; i < rows.size(); ++i) {
            std::string back;
            for (char c : rows[i]) if (c != '-') back.push_back(c);
            if (back != units[i] || rows[i].size() != rows[0].size()) { std::printf("aligner broke unit %zu\n", i); return 1; }
        }
        try {
            const mb::Built b = mb::build_read_matcher(dna(rng, flank), dna(rng, flank), rows, copies, trial % 2 ? 0.05 : 0.3, nullptr, nullptr);
            checksum += b.m + b.in_ptr.back() + (long)b.names.size();
            if ((int)b.in_src.size() != b.in_ptr.back() || (int)b.state_class.size() != b.m) { std::printf("inconsistent model\n"); return 1; }
            // the column-program compiler on the built model, and its serialisation
            HostModel H{b.m, b.silent_start, b.start_index, b.end_index, 0, b.in_ptr, b.in_src, b.in_logp, b.emis};
            H.finite = (b.in_ptr[b.end_index + 1] - b.in_ptr[b.end_index]) != 0;
            ColProgramHost prog;
            build_column_program(H, prog);
            if (!prog.valid) { std::printf("no column program: %s\n", prog.why.c_str()); return 1; }
            checksum += (long)prog.serialize().size() + prog.n_cols;
        } catch (const std::exception &e) {
            // alignments whose every column is an insert column have no profile: a clean error, not a crash
            checksum += 1;
        }
    }
    // error paths
    try { mb::build_read_matcher("ACGT", "ACNT", {"ACG"}, 2, 0.05, nullptr, nullptr); return 1; } catch (const std::exception &) {}
    try { mb::build_read_matcher("", "ACGT", {"ACG"}, 2, 0.05, nullptr, nullptr); return 1; } catch (const std::exception &) {}
    try { mb::build_read_matcher("ACGT", "ACGT", {"ACG", "AC"}, 2, 0.05, nullptr, nullptr); return 1; } catch (const std::exception &) {}
    try { msa::align_units({"ACG", ""}); return 1; } catch (const std::exception &) {}
    std::printf("ok %ld\n", checksum);
    return 0;
}
