/*
 * oracle/viterbi_oracle.c -- CPU restatement of the reference's profile-HMM scoring path.
 *
 * TEST INFRASTRUCTURE.  This file is the *checker*, not the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The shipped
 * path (advntr_amd/) never links, imports or falls back to anything in oracle/.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks every function here against
 * the fixtures under tests/golden/, which were produced by running the reference itself
 * (vendored pomegranate 0.6.1 + advntr/hmm_utils.py) in the build container
 * (tests/golden/make_golden.py).  The reference's own tests hold no Viterbi score
 * vectors, so those captured outputs are the pin.
 *
 * Restated algorithms (paths relative to /root/reference):
 *   oracle_viterbi   <- HiddenMarkovModel._viterbi   pomegranate/hmm.pyx:1970-2136
 *   oracle_forward   <- HiddenMarkovModel._forward / _vl_log_probability
 *                                                     pomegranate/hmm.pyx:1371-1484, 1300-1313
 *   lse2             <- pair_lse                       pomegranate/utils.pyx:72-90
 *   oracle_build_csr <- the CSR fill of bake()         pomegranate/hmm.pyx:970-1011
 *
 * Same data layout as the reference: full (n+1) x m tables calloc'd per call (hmm.pyx:1977-1980),
 * CSR in-edges in graph.edges_iter() order, fp64 throughout, strict '>' so the first maximum in
 * in-edge order wins, and the (v + t) + e association of hmm.pyx:2036-2037.
 */
#include <malloc.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int m;             /* number of states (emitting first, then silent; hmm.pyx:887)   */
    int silent_start;  /* index of first silent state = number of emitting states       */
    int start_index;
    int end_index;
    int finite;        /* 1 iff the end state has in-edges (hmm.pyx:977-980)            */
    int n_edges;
    int *in_ptr;       /* m+1 */
    int *in_src;       /* n_edges */
    double *in_logp;   /* n_edges */
    double *emis;      /* silent_start x 4 log-probabilities, symbol codes A,C,G,T=0..3 */
} oracle_model;

/* hmm.pyx:970-1011: count in-edges per destination, prefix-sum, then drop every edge of the
 * edges_iter() stream into the first free slot of its destination's segment (= stable by dst). */
oracle_model *oracle_model_create(int m, int silent_start, int start_index, int end_index, int n_edges,
                                  const int *edge_src, const int *edge_dst, const double *edge_logp,
                                  const double *emis_logp)
{
    oracle_model *M = (oracle_model *)calloc(1, sizeof(oracle_model));
    M->m = m; M->silent_start = silent_start; M->start_index = start_index; M->end_index = end_index;
    M->n_edges = n_edges;
    M->in_ptr = (int *)calloc((size_t)m + 1, sizeof(int));
    M->in_src = (int *)malloc(sizeof(int) * (size_t)(n_edges > 0 ? n_edges : 1));
    M->in_logp = (double *)malloc(sizeof(double) * (size_t)(n_edges > 0 ? n_edges : 1));
    M->emis = (double *)malloc(sizeof(double) * 4 * (size_t)(silent_start > 0 ? silent_start : 1));
    memcpy(M->emis, emis_logp, sizeof(double) * 4 * (size_t)silent_start);
    for (int k = 0; k < n_edges; ++k) M->in_ptr[edge_dst[k] + 1] += 1;
    M->finite = M->in_ptr[end_index + 1] != 0;
    for (int i = 1; i <= m; ++i) M->in_ptr[i] += M->in_ptr[i - 1];
    int *fill = (int *)calloc((size_t)m, sizeof(int));
    for (int k = 0; k < n_edges; ++k) {
        int b = edge_dst[k];
        int pos = M->in_ptr[b] + fill[b]++;
        M->in_src[pos] = edge_src[k];
        M->in_logp[pos] = edge_logp[k];
    }
    free(fill);
    return M;
}

void oracle_model_destroy(oracle_model *M)
{
    if (!M) return;
    free(M->in_ptr); free(M->in_src); free(M->in_logp); free(M->emis); free(M);
}

int oracle_model_csr(const oracle_model *M, int *in_ptr, int *in_src, double *in_logp)
{
    memcpy(in_ptr, M->in_ptr, sizeof(int) * ((size_t)M->m + 1));
    memcpy(in_src, M->in_src, sizeof(int) * (size_t)M->n_edges);
    memcpy(in_logp, M->in_logp, sizeof(double) * (size_t)M->n_edges);
    return M->finite;
}

/* hmm.pyx:1970-2136.  seq = symbol codes 0..3.  path (capacity path_cap ints) receives the state
 * indices from model start to model end; *path_len = its length, 0 when the sequence is impossible.
 * The reference writes into a fixed n+m buffer (hmm.pyx:1953); we report -2 in *path_len when the
 * true path would not fit path_cap instead of overrunning. */
double oracle_viterbi(const oracle_model *M, const uint8_t *seq, int n, int *path, int path_cap, int *path_len)
{
    const int m = M->m, p = M->silent_start;
    const int *in_edges = M->in_ptr;
    int *tbx = (int *)calloc((size_t)(n + 1) * m, sizeof(int));
    int *tby = (int *)calloc((size_t)(n + 1) * m, sizeof(int));
    double *v = (double *)calloc((size_t)(n + 1) * m, sizeof(double));
    double *e = (double *)calloc((size_t)n * p + 1, sizeof(double));
    double slp, logp;
    int end_index;

    *path_len = 0;
    /* emission table, state weight 1 => + log(1) = +0.0 (hmm.pyx:1990-1997, :928-930) */
    for (int l = 0; l < p; ++l)
        for (int i = 0; i < n; ++i)
            e[(size_t)l * n + i] = M->emis[4 * l + seq[i]] + 0.0;

    for (int i = 0; i < m; ++i) v[i] = -INFINITY;
    v[M->start_index] = 0;

    for (int l = p; l < m; ++l) {                                   /* hmm.pyx:2003-2023 */
        if (l == M->start_index) continue;
        for (int k = in_edges[l]; k < in_edges[l + 1]; ++k) {
            int ki = M->in_src[k];
            if (ki < p || ki >= l) continue;
            slp = v[ki] + M->in_logp[k];
            if (slp > v[l]) { v[l] = slp; tbx[l] = 0; tby[l] = ki; }
        }
    }

    for (int i = 0; i < n; ++i) {
        double *vp = v + (size_t)i * m, *vc = v + (size_t)(i + 1) * m;
        int *bx = tbx + (size_t)(i + 1) * m, *by = tby + (size_t)(i + 1) * m;
        for (int l = 0; l < p; ++l) {                               /* hmm.pyx:2026-2042 */
            vc[l] = -INFINITY;
            for (int k = in_edges[l]; k < in_edges[l + 1]; ++k) {
                int ki = M->in_src[k];
                slp = vp[ki] + M->in_logp[k] + e[(size_t)l * n + i];
                if (slp > vc[l]) { vc[l] = slp; bx[l] = i; by[l] = ki; }
            }
        }
        for (int l = p; l < m; ++l) {                               /* hmm.pyx:2044-2063 */
            vc[l] = -INFINITY;
            for (int k = in_edges[l]; k < in_edges[l + 1]; ++k) {
                int ki = M->in_src[k];
                if (ki >= p) continue;
                slp = vc[ki] + M->in_logp[k];
                if (slp > vc[l]) { vc[l] = slp; bx[l] = i + 1; by[l] = ki; }
            }
        }
        for (int l = p; l < m; ++l) {                               /* hmm.pyx:2065-2083 */
            for (int k = in_edges[l]; k < in_edges[l + 1]; ++k) {
                int ki = M->in_src[k];
                if (ki < p || ki >= l) continue;
                slp = vc[ki] + M->in_logp[k];
                if (slp > vc[l]) { vc[l] = slp; bx[l] = i + 1; by[l] = ki; }
            }
        }
    }

    if (M->finite == 1) {                                           /* hmm.pyx:2089-2098 */
        logp = v[(size_t)n * m + M->end_index];
        end_index = M->end_index;
    } else {
        end_index = -1;
        logp = -INFINITY;
        for (int i = 0; i < m; ++i)
            if (v[(size_t)n * m + i] > logp) { logp = v[(size_t)n * m + i]; end_index = i; }
    }

    if (logp != -INFINITY) {                                        /* hmm.pyx:2109-2130 */
        int px = n, py = end_index, length = 0, overflow = 0;
        while (px != 0 || py != M->start_index) {
            if (length < path_cap) path[length] = py; else overflow = 1;
            length += 1;
            int npx = tbx[(size_t)px * m + py];
            py = tby[(size_t)px * m + py];
            px = npx;
        }
        if (length < path_cap) path[length] = py; else overflow = 1;
        if (overflow) {
            *path_len = -2;
        } else {
            for (int i = 0; i < (length + 1) / 2; ++i) {
                int t = path[i]; path[i] = path[length - i]; path[length - i] = t;
            }
            *path_len = length + 1;
        }
    }
    free(tbx); free(tby); free(v); free(e);
    return logp;
}

/* utils.pyx:72-90 */
static double lse2(double x, double y)
{
    if (x == INFINITY || y == INFINITY) return INFINITY;
    if (x == -INFINITY) return y;
    if (y == -INFINITY) return x;
    if (x > y) return x + log(exp(y - x) + 1);
    return y + log(exp(x - y) + 1);
}

/* hmm.pyx:1371-1484 + 1300-1313 */
double oracle_forward(const oracle_model *M, const uint8_t *seq, int n)
{
    const int m = M->m, p = M->silent_start;
    const int *in_edges = M->in_ptr;
    double *f = (double *)calloc((size_t)m * (n + 1), sizeof(double));
    double lp;
    for (int i = 0; i < m; ++i) f[i] = -INFINITY;
    f[M->start_index] = 0.;
    for (int l = p; l < m; ++l) {
        if (l == M->start_index) continue;
        lp = -INFINITY;
        for (int k = in_edges[l]; k < in_edges[l + 1]; ++k) {
            int ki = M->in_src[k];
            if (ki < p || ki >= l) continue;
            lp = lse2(lp, f[ki] + M->in_logp[k]);
        }
        f[l] = lp;
    }
    for (int i = 0; i < n; ++i) {
        double *fp = f + (size_t)i * m, *fc = f + (size_t)(i + 1) * m;
        for (int l = 0; l < p; ++l) {
            lp = -INFINITY;
            for (int k = in_edges[l]; k < in_edges[l + 1]; ++k)
                lp = lse2(lp, fp[M->in_src[k]] + M->in_logp[k]);
            fc[l] = lp + (M->emis[4 * l + seq[i]] + 0.0);
        }
        for (int l = p; l < m; ++l) {
            lp = -INFINITY;
            for (int k = in_edges[l]; k < in_edges[l + 1]; ++k) {
                int ki = M->in_src[k];
                if (ki >= p) continue;
                lp = lse2(lp, fc[ki] + M->in_logp[k]);
            }
            fc[l] = lp;
        }
        for (int l = p; l < m; ++l) {
            lp = -INFINITY;
            for (int k = in_edges[l]; k < in_edges[l + 1]; ++k) {
                int ki = M->in_src[k];
                if (ki < p || ki >= l) continue;
                lp = lse2(lp, fc[ki] + M->in_logp[k]);
            }
            fc[l] = lse2(fc[l], lp);
        }
    }
    if (M->finite == 1) {
        lp = f[(size_t)n * m + M->end_index];
    } else {
        lp = -INFINITY;
        for (int i = 0; i < p; ++i) lp = lse2(lp, f[(size_t)n * m + i]);
    }
    free(f);
    return lp;
}

/* Batch driver used by bench.py's cpu_baseline leg: scores reads [0,n_reads) of one model, one
 * oracle_viterbi call each (fresh tables per call, exactly like the reference), 1 thread. */
void oracle_viterbi_many(const oracle_model *M, const uint8_t *bases, const int64_t *off, int n_reads,
                         double *out_logp, int *scratch_path, int path_cap, int *out_path_len)
{
    for (int r = 0; r < n_reads; ++r) {
        int len = 0;
        out_logp[r] = oracle_viterbi(M, bases + off[r], (int)(off[r + 1] - off[r]), scratch_path, path_cap, &len);
        if (out_path_len) out_path_len[r] = len;
    }
}


/* The same loop on n_threads host threads (reads are independent; the reference itself is single-threaded on this
 * path -- this only serves bench.py's "all host cores" baseline figure).  Each thread owns its path scratch. */
typedef struct {
    const oracle_model *M;
    const uint8_t *bases;
    const int64_t *off;
    int n_reads, path_cap, n_threads, tid;
    double *out_logp;
} many_job;

static void *many_worker(void *arg)
{
    many_job *j = (many_job *)arg;
    int *scratch = (int *)malloc(sizeof(int) * (size_t)j->path_cap);
    for (int r = j->tid; r < j->n_reads; r += j->n_threads) {
        int len = 0;
        j->out_logp[r] = oracle_viterbi(j->M, j->bases + j->off[r], (int)(j->off[r + 1] - j->off[r]), scratch, j->path_cap, &len);
    }
    free(scratch);
    return NULL;
}

void oracle_viterbi_many_mt(const oracle_model *M, const uint8_t *bases, const int64_t *off, int n_reads,
                            double *out_logp, int path_cap, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    /* the per-call tables (several MB, calloc'd as the reference does) would otherwise be mmap'd and unmapped on every
     * call, which serialises the threads in the kernel's address-space lock; keep them on the per-thread heaps */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    many_job *jobs = (many_job *)malloc(sizeof(many_job) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        jobs[t] = (many_job){M, bases, off, n_reads, path_cap, n_threads, t, out_logp};
        pthread_create(&th[t], NULL, many_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}
