// oracle/colprog_check.cpp -- TEST INFRASTRUCTURE (CPU): checks the product's host-side column-program
// compiler (advntr_amd/csrc/column_program.h) without a GPU.
//
// It compiles a baked CSR model into a column program with the product's own builder and then
// evaluates that program with a plain row-major scalar loop that mirrors, statement for statement,
// what one lane of viterbi_columns.h does per trellis cell (same candidate order, same (v+t)+e
// association, same back-pointer codes, same tail evaluation and traceback).  tests/test_colprog_cpu.py
// compares its scores and paths with the oracle on the goldens, so a builder bug shows up on CPU.
// Only tests load this library; it is never linked into libadvntr_hip.so.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../advntr_amd/csrc/column_program.h"

struct CheckModel {
    int m, P, start, end, finite;
    std::vector<int32_t> in_ptr, in_src;
    std::vector<double> in_logp, emis;
};

struct Handle {
    CheckModel M;
    ColProgramHost prog;
};

extern "C" void *colprog_create(int m, int P, int start, int end, int n_edges, const int32_t *in_ptr,
                                const int32_t *in_src, const double *in_logp, const double *emis)
{
    Handle *h = new Handle();
    h->M.m = m; h->M.P = P; h->M.start = start; h->M.end = end;
    h->M.in_ptr.assign(in_ptr, in_ptr + m + 1);
    h->M.in_src.assign(in_src, in_src + n_edges);
    h->M.in_logp.assign(in_logp, in_logp + n_edges);
    h->M.emis.assign(emis, emis + (size_t)P * 4);
    h->M.finite = (in_ptr[end + 1] - in_ptr[end]) != 0;
    build_column_program(h->M, h->prog);
    return h;
}

extern "C" void colprog_destroy(void *p) { delete (Handle *)p; }
extern "C" int colprog_valid(void *p) { return ((Handle *)p)->prog.valid ? 1 : 0; }
extern "C" const char *colprog_why(void *p) { return ((Handle *)p)->prog.why.c_str(); }
extern "C" void colprog_stats(void *p, int *out)
{
    const ColProgramHost &g = ((Handle *)p)->prog;
    out[0] = g.n_cols; out[1] = (int)g.classes.size(); out[2] = (int)(g.emis.size() / COL_EMIS_STRIDE);
    out[3] = (int)g.tail_state.size(); out[4] = g.n_sinks; out[5] = (int)g.serialize().size();
    out[6] = ((const ColProgram *)g.serialize().data())->lds_bytes;
}

// returns logp; path (reversed order like the kernel's scratch) into rev, *len = length (0 impossible)
extern "C" double colprog_viterbi(void *p, const uint8_t *seq, int n, int32_t *path, int cap, int *len_out)
{
    const Handle *h = (const Handle *)p;
    const ColProgramHost &g = h->prog;
    const int NC = g.n_cols, P = h->M.P;
    const double NINF = -INFINITY;
    *len_out = 0;
    if (!g.valid || n < 1) return NAN;
    std::vector<double> pI(NC, NINF), pM(NC, NINF), pB(NC), cI(NC), cM(NC), cB(NC);
    std::vector<uint8_t> bp((size_t)(n + 1) * NC, 0);
    std::vector<int32_t> sinkbp((size_t)COL_MAX_SINKS * (n + 1), 0);
    for (int c = 0; c < NC; ++c) pB[c] = g.info[c + 1].v0b;            // row 0
    for (int t = 1; t <= n; ++t) {
        const int x = seq[t - 1];
        double er = NINF;
        int erwin = 0;
        for (int c = 0; c < NC; ++c) {
            const ColInfo &inf = g.info[c + 1];
            const ColClass &T = g.classes[inf.tclass];
            const double eI = g.emis[inf.emI * COL_EMIS_STRIDE + x], eM = g.emis[inf.emM * COL_EMIS_STRIDE + x];
            const double nI = pI[c], nM = pM[c], nB = pB[c];
            const double qI = c ? pI[c - 1] : NINF, qM = c ? pM[c - 1] : NINF, qB = c ? pB[c - 1] : NINF;
            const double oI = c ? cI[c - 1] : NINF, oM = c ? cM[c - 1] : NINF, oB = c ? cB[c - 1] : NINF;
            double vI = (nI + T.iI) + eI; int pi = 0;
            { const double c1 = (nM + T.iM) + eI, c2 = (nB + T.iD) + eI;
              if (c1 > vI) { vI = c1; pi = 1; } if (c2 > vI) { vI = c2; pi = 2; } }
            double vM = (qI + T.mI) + eM; int pm = 0;
            { const double c1 = (qM + T.mM) + eM, c2 = ((t == 1) ? T.mX : NINF) + eM, c3 = (qB + T.mD) + eM;
              if (c1 > vM) { vM = c1; pm = 1; } if (c2 > vM) { vM = c2; pm = 2; } if (c3 > vM) { vM = c3; pm = 3; } }
            double vB = oI + T.dI; int pb = 0;
            { const double c1 = oM + T.dM, c2 = oB + T.dD;
              if (c1 > vB) { vB = c1; pb = 1; } if (c2 > vB) { vB = c2; pb = 2; } }
            const unsigned fl = inf.flags;
            if (fl & COL_FLAG_SINK) { vB = er; pb = 3; sinkbp[((fl >> 4) & 15) * (n + 1) + t] = erwin; er = NINF; }
            if (fl & COL_FLAG_FEED) { const double cand = vB + T.erw; if (cand > er) { er = cand; erwin = c; } }
            cI[c] = vI; cM[c] = vM; cB[c] = vB;
            bp[(size_t)t * NC + c] = (uint8_t)(pi | (pm << 2) | (pb << 4));
        }
        pI.swap(cI); pM.swap(cM); pB.swap(cB);
    }
    // tail at row n (p* hold row n now)
    std::vector<double> tailv(g.tail_state.size(), NINF);
    std::vector<int> tailwin(g.tail_state.size(), 0);
    for (size_t i = 0; i < g.tail_state.size(); ++i) {
        double best = NINF; int rank = 0x7fffffff;
        for (int e = g.tail_ptr[i]; e < g.tail_ptr[i + 1]; ++e) {
            const TailEdge &ed = g.tail_edges[e];
            double v;
            if (ed.loc >= 0) { const int c = ed.loc >> 2, sl = ed.loc & 3; v = sl == 0 ? pI[c] : sl == 1 ? pM[c] : pB[c]; }
            else v = tailv[-ed.loc - 1];
            const double cand = v + ed.logp;
            if (cand > best) { best = cand; rank = e; }
        }
        tailv[i] = best; tailwin[i] = rank;
    }
    const double logp = tailv[g.end_tail];
    if (logp == NINF) return logp;
    int len = 0, ti = g.end_tail, t = n, c = 0, slot = 0;
    for (;;) {
        if (len >= cap - 2) { *len_out = -2; return logp; }
        path[len++] = g.tail_state[ti];
        const TailEdge &ed = g.tail_edges[tailwin[ti]];
        if (ed.loc < 0) { ti = -ed.loc - 1; continue; }
        c = ed.loc >> 2; slot = ed.loc & 3; break;
    }
    int s0 = -1;
    while (t >= 1) {
        if (len >= cap - 2) { *len_out = -2; return logp; }
        const ColState &cs = g.state[c + 1];
        path[len++] = slot == 0 ? cs.sI : slot == 1 ? cs.sM : cs.sB;
        const int byte = bp[(size_t)t * NC + c];
        if (slot == 0) { slot = byte & 3; t -= 1; }
        else if (slot == 1) {
            const int q = (byte >> 2) & 3; t -= 1;
            if (q == 2) { s0 = cs.sX; break; }
            c -= 1; slot = q == 3 ? 2 : q;
        } else {
            const int q = (byte >> 4) & 3;
            if (q == 3) c = sinkbp[((g.info[c + 1].flags >> 4) & 15) * (n + 1) + t];
            else { c -= 1; slot = q; }
        }
    }
    if (s0 < 0) s0 = g.state[c + 1].sB;
    while (s0 != h->M.start) {
        if (len >= cap - 2 || s0 < P) { *len_out = -2; return logp; }
        path[len++] = s0;
        s0 = g.pred0[s0 - P];
    }
    path[len++] = h->M.start;
    *len_out = len;
    return logp;
}

// Sum-product twin: evaluates the column program with pair_lse exactly as the forward kernel does (same fold order).
static double lse2c(double x, double y)
{
    if (x == INFINITY || y == INFINITY) return INFINITY;
    if (x == -INFINITY) return y;
    if (y == -INFINITY) return x;
    if (x > y) return x + std::log(std::exp(y - x) + 1);
    return y + std::log(std::exp(x - y) + 1);
}

extern "C" double colprog_forward(void *p, const uint8_t *seq, int n)
{
    const Handle *h = (const Handle *)p;
    const ColProgramHost &g = h->prog;
    const int NC = g.n_cols;
    const double NINF = -INFINITY;
    if (!g.valid || n < 1) return NAN;
    std::vector<double> pI(NC, NINF), pM(NC, NINF), pB(NC), cI(NC), cM(NC), cB(NC);
    for (int c = 0; c < NC; ++c) pB[c] = g.fwd[2 * c];
    for (int t = 1; t <= n; ++t) {
        const int x = seq[t - 1];
        double er = NINF;
        for (int c = 0; c < NC; ++c) {
            const ColInfo &inf = g.info[c + 1];
            const ColClass &T = g.classes[inf.tclass];
            const double eI = g.emis[inf.emI * COL_EMIS_STRIDE + x], eM = g.emis[inf.emM * COL_EMIS_STRIDE + x];
            const double qI = c ? pI[c - 1] : NINF, qM = c ? pM[c - 1] : NINF, qB = c ? pB[c - 1] : NINF;
            const double oI = c ? cI[c - 1] : NINF, oM = c ? cM[c - 1] : NINF, oB = c ? cB[c - 1] : NINF;
            double vI = lse2c(lse2c(pI[c] + T.iI, pM[c] + T.iM), pB[c] + T.iD) + eI;
            double vM = lse2c(lse2c(lse2c(qI + T.mI, qM + T.mM), (t == 1) ? g.fwd[2 * c + 1] : NINF), qB + T.mD) + eM;
            double vB = lse2c(lse2c(oI + T.dI, oM + T.dM), oB + T.dD);
            const unsigned fl = inf.flags;
            if (fl & COL_FLAG_SINK) { vB = er; er = NINF; }
            if (fl & COL_FLAG_FEED) er = lse2c(er, vB + T.erw);
            cI[c] = vI; cM[c] = vM; cB[c] = vB;
        }
        pI.swap(cI); pM.swap(cM); pB.swap(cB);
    }
    std::vector<double> tailv(g.tail_state.size(), NINF);
    for (size_t i = 0; i < g.tail_state.size(); ++i) {
        double pe = NINF, ps = NINF;                  // emitting-sourced fold, then silent-sourced fold (hmm.pyx:1446-1480)
        for (int e = g.tail_ptr[i]; e < g.tail_ptr[i + 1]; ++e) {
            const TailEdge &ed = g.tail_edges[e];
            if (ed.loc >= 0) {
                const int c = ed.loc >> 2, sl = ed.loc & 3;
                const double v = (sl == 0 ? pI[c] : sl == 1 ? pM[c] : pB[c]) + ed.logp;
                if (sl < 2) pe = lse2c(pe, v); else ps = lse2c(ps, v);
            } else ps = lse2c(ps, tailv[-ed.loc - 1] + ed.logp);
        }
        tailv[i] = lse2c(pe, ps);
    }
    return tailv[g.end_tail];
}
