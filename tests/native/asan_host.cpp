// Address/UB-sanitised run of the product's host-only C++ (the native model builder, the repeat aligner, the column-program
// compiler, and the host text / piece-cutting entry points of the C ABI): GPU
// sanitizers are not available on the pool, so the host code is exercised here under -fsanitize=address,undefined.
// Built and run by tests/test_native_sanitizers.py with plain g++ (no HIP).
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstring>
#include <thread>

#include "../../include/advntr_hip.h"
#include "../../include/advntr_pyhost.h"
#include "../../advntr_amd/csrc/model_builder.h"
#include "../../advntr_amd/csrc/repeat_msa.h"
#include "../../advntr_amd/csrc/column_program.h"

// the host-only entry points of the C ABI (text handling, piece cutting, genotype caller) live in abi_genotype.h, which
// engine.hip includes behind its own `fail` and `host_cpu_limit`: the two are stood in for here, the code under test is the
// shipped header
static int fail(int code, const char *, ...) { return code; }
static int host_cpu_limit() { return 4; }
#include "../../advntr_amd/csrc/abi_genotype.h"

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
    // ---- host text handling and piece cutting (abi_genotype.h): against scalar restatements, ragged sizes, several threads
    for (int trial = 0; trial < 40; ++trial) {
        const int n_reads = 1 + (int)(rng() % 300);
        std::string text;
        std::vector<int64_t> start, end, off(1, 0);
        for (int r = 0; r < n_reads; ++r) {
            text += ">r" + std::to_string(r) + "\n";
            const int n = (int)(rng() % 400);                       // (empty reads included)
            start.push_back((int64_t)text.size());
            for (int i = 0; i < n; ++i) {
                const unsigned x = rng() % 1000;
                text.push_back(x < 960 ? "ACGTacgt"[x % 8] : (x < 985 ? "Nn"[x & 1] : "-*RX"[x & 3]));
            }
            end.push_back((int64_t)text.size());
            off.push_back(off.back() + n);
            if (r + 1 < n_reads || (rng() & 1)) text.push_back('\n');
        }
        // line index
        int64_t n_lines = 0;
        std::vector<int64_t> lines(2 * n_reads + 2);
        if (advntr_line_index(text.data(), (int64_t)text.size(), 1 + (int)(rng() % 5), lines.data(), (int64_t)lines.size(), &n_lines) != ADVNTR_OK) return 1;
        if (n_lines != 2 * n_reads) { std::printf("line index: %lld lines of %d\n", (long long)n_lines, 2 * n_reads); return 1; }
        for (int r = 0; r < n_reads; ++r)
            if (lines[2 * r + 1] != start[r]) { std::printf("line index: read %d\n", r); return 1; }
        if (advntr_line_index(text.data(), (int64_t)text.size(), 2, lines.data(), 1, &n_lines) != ADVNTR_ERR_TOO_LARGE) return 1;
        // encoding out of spans and out of separate texts, both case modes
        for (uint32_t flags : {0u, (uint32_t)ADVNTR_ENCODE_CASE_SENSITIVE}) {
            std::vector<uint8_t> codes((size_t)off.back() + 1, 77), codes2((size_t)off.back() + 1, 77), bad(n_reads, 9), bad2(n_reads, 9);
            if (advntr_encode_spans(text.data(), start.data(), end.data(), n_reads, flags, 1 + (int)(rng() % 5), off.data(), codes.data(), bad.data()) != ADVNTR_OK) return 1;
            std::vector<const char *> ptrs;
            for (int r = 0; r < n_reads; ++r) ptrs.push_back(text.data() + start[r]);
            if (advntr_encode_texts(ptrs.data(), n_reads, flags, 3, off.data(), codes2.data(), bad2.data()) != ADVNTR_OK) return 1;
            for (int r = 0; r < n_reads; ++r) {
                uint8_t want_bad = 0;
                for (int64_t i = start[r]; i < end[r]; ++i) {
                    char ch = text[(size_t)i];
                    if (!flags && ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
                    const uint8_t want = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : ch == 'N' ? 254 : 255;
                    if (want == 254 && want_bad < 1) want_bad = 1;
                    if (want == 255) want_bad = 2;
                    const size_t o = (size_t)(off[r] + i - start[r]);
                    if (codes[o] != want || codes2[o] != want) { std::printf("encode: read %d\n", r); return 1; }
                }
                if (bad[r] != want_bad || bad2[r] != want_bad) { std::printf("encode: bad flag of read %d\n", r); return 1; }
            }
            if (codes[(size_t)off.back()] != 77) { std::printf("encode wrote past its output\n"); return 1; }
            checksum += codes.empty() ? 0 : codes[0];
        }
        // a span that does not match its output slot is an error, not a write past the slot
        if (n_reads > 1 && end[0] > start[0]) {
            std::vector<int64_t> off_bad(off);
            off_bad[1] -= 1;
            std::vector<uint8_t> codes((size_t)off.back() + 1);
            if (advntr_encode_spans(text.data(), start.data(), end.data(), n_reads, 0, 2, off_bad.data(), codes.data(), nullptr) != ADVNTR_ERR_ARG) return 1;
        }
        // pieces of encoded reads, forward and reverse-complemented
        std::vector<uint8_t> codes((size_t)off.back() + 1);
        advntr_encode_spans(text.data(), start.data(), end.data(), n_reads, 0, 1, off.data(), codes.data(), nullptr);
        const int n_pieces = (int)(rng() % 200);
        std::vector<int32_t> piece_read;
        std::vector<int64_t> begin, stop, out_off(1, 0);
        std::vector<uint8_t> reverse;
        for (int p = 0; p < n_pieces; ++p) {
            const int r = (int)(rng() % n_reads);
            const int64_t n = off[r + 1] - off[r], b = n ? (int64_t)(rng() % (n + 1)) : 0, e = b + (n - b ? (int64_t)(rng() % (n - b + 1)) : 0);
            piece_read.push_back(r); begin.push_back(b); stop.push_back(e); reverse.push_back((uint8_t)(rng() & 1));
            out_off.push_back(out_off.back() + (e - b));
        }
        std::vector<uint8_t> cut((size_t)out_off.back() + 1, 77);
        if (advntr_cut_pieces(codes.data(), off.data(), n_reads, piece_read.data(), begin.data(), stop.data(), reverse.data(), n_pieces,
                              1 + (int)(rng() % 4), out_off.data(), cut.data()) != ADVNTR_OK) return 1;
        for (int p = 0; p < n_pieces; ++p)
            for (int64_t i = 0; i < stop[p] - begin[p]; ++i) {
                const uint8_t c = reverse[p] ? codes[(size_t)(off[piece_read[p]] + stop[p] - 1 - i)] : codes[(size_t)(off[piece_read[p]] + begin[p] + i)];
                const uint8_t want = c < 4 ? (uint8_t)(reverse[p] ? 3 - c : c) : (uint8_t)255;
                if (cut[(size_t)(out_off[p] + i)] != want) { std::printf("cut_pieces: piece %d\n", p); return 1; }
            }
        if (cut[(size_t)out_off.back()] != 77) { std::printf("cut_pieces wrote past its output\n"); return 1; }
        if (n_pieces) {
            stop[0] = off[piece_read[0] + 1] - off[piece_read[0]] + 1;       // past the end of its read
            if (advntr_cut_pieces(codes.data(), off.data(), n_reads, piece_read.data(), begin.data(), stop.data(), reverse.data(), n_pieces,
                                  2, out_off.data(), cut.data()) != ADVNTR_ERR_ARG) return 1;
            piece_read[0] = n_reads;
            if (advntr_cut_pieces(codes.data(), off.data(), n_reads, piece_read.data(), begin.data(), stop.data(), reverse.data(), n_pieces,
                                  2, out_off.data(), cut.data()) != ADVNTR_ERR_ARG) return 1;
        }
    }
    // the CPython helper in a process that is not an interpreter: an error code, no crash
    {
        const char *texts[1];
        int64_t lengths[1];
        if (advntr_pylist_texts(nullptr, texts, lengths, 1) != -1) return 1;
    }
    std::printf("ok %ld\n", checksum);
    return 0;
}
