// device_model.h -- device-side view of a baked profile HMM (shared by host code and kernels).
//
// Layout in HBM (one allocation per model, all arrays 16-byte aligned inside it):
//   emitting states l < P   : CSR in-edges exactly as pomegranate's bake() leaves them
//                             (e_ptr/e_src/e_logp, order = graph.edges_iter(), hmm.pyx:994-1011)
//   silent states   l >= P  : per state an "A" list (sources outside the state's 64-state chunk:
//                             emitting states and silent states of earlier chunks, list order kept)
//                             followed by a "B" list (sources inside the chunk, sorted by source);
//                             every entry carries its ordinal in the reference's evaluation order
//                             (emitting-sourced pass hmm.pyx:2044-2063, then silent-sourced pass
//                             hmm.pyx:2065-2083 with the ki<l filter) so ties resolve identically.
//   r_ptr/r_src             : that evaluation order as a plain list, for the traceback.
#pragma once
#include <stdint.h>

#define ADV_WAVE 64

struct ColProgram;   // anti-diagonal kernel's column program (column_program.h)

struct DevModel {
    int32_t m, P, start, end, finite;
    int32_t bp_width;            // 1 or 2 bytes per trellis cell (2 when some fan-in > 255)
    int32_t n_chunks;            // ceil((m-P)/64)
    int32_t max_indeg;
    const int32_t *e_ptr;        // P+1
    const int32_t *e_src;
    const double *e_logp;
    const double *emis;          // P*4
    const int32_t *s_ptr;        // S+1   start of A list
    const int32_t *s_mid;        // S     start of B list
    const int32_t *s_src;
    const int32_t *s_ord;
    const double *s_logp;
    const int32_t *r_ptr;        // S+1
    const int32_t *r_src;
    const double *r_logp;        // reference-order log-probs (forward kernel)
    const uint16_t *sclass;      // m
    const ColProgram *cols;      // nullptr when the model has no column program
};
