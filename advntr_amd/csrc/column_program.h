// column_program.h -- placeholder until the anti-diagonal kernel lands.
#pragma once
#include <stdint.h>
#include <vector>
struct advntr_hmm;
struct ColProgram { int32_t n_cols; };
struct ColProgramHost {
    bool valid = false;
    int32_t n_cols = 0;
    std::vector<uint8_t> serialize() const { return {}; }
};
static inline void build_column_program(const advntr_hmm &, ColProgramHost &out) { out.valid = false; }
