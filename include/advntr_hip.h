/*
 * advntr_hip.h -- C ABI of the MI355X-native profile-HMM scoring engine (libadvntr_hip.so).
 *
 * This is the drop-in boundary for ONE path of adVNTR: the read-vs-locus-model Viterbi scoring
 * loop that the reference runs through its vendored Cython pomegranate
 *   HiddenMarkovModel.viterbi / _viterbi            /root/reference/pomegranate/hmm.pyx:1911-2136
 *   HiddenMarkovModel.log_probability / _forward    /root/reference/pomegranate/hmm.pyx:1258-1313, 1371-1484
 * as driven per read by advntr/vntr_finder.py:239,242,553,738, plus the Viterbi-path summaries that
 * every caller derives from the returned path (advntr/hmm_utils.py:155-286).
 *
 * Plain C: pointers and sizes only, caller owns every buffer, status codes instead of exceptions
 * (the reference raises ValueError / returns (-inf, None); see per-function notes).  No torch types.
 * The reference-side binding (ctypes) is shown in INTEGRATION.md; advntr_amd/_lib.py is that binding.
 *
 * Symbol codes: A,C,G,T = 0,1,2,3 (the reference maps symbols through model.keymap, hmm.pyx:57-81,
 * 1072-1080; any code > 3 is the "Symbol not defined" ValueError case -> ADVNTR_ERR_SYMBOL).
 */
#ifndef ADVNTR_HIP_H
#define ADVNTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADVNTR_OK               0
#define ADVNTR_ERR_ARG         -1   /* bad argument (null pointer, negative size, index out of range)   */
#define ADVNTR_ERR_SYMBOL      -2   /* a read holds a base code > 3  (reference: ValueError hmm.pyx:72,79) */
#define ADVNTR_ERR_DEVICE      -3   /* HIP runtime error; text in advntr_last_error()                    */
#define ADVNTR_ERR_TOO_LARGE   -4   /* model does not fit the on-chip trellis rows       */
#define ADVNTR_ERR_UNSUPPORTED -5

/* flags for advntr_viterbi_batch / advntr_batch_run */
#define ADVNTR_FLAG_PATH          1u  /* also return the Viterbi paths (state indices, start..end)      */
#define ADVNTR_FLAG_FORCE_GENERIC 2u  /* use the generic-CSR kernel even if a model has a column program */
#define ADVNTR_FLAG_NO_SUMMARY    4u  /* skip the path summaries (out_summary untouched)                */
#define ADVNTR_FLAG_STREAM        8u  /* experimental: pack several reads per wavefront along the row axis (stream kernel)
                                         instead of one read per wavefront bucketed by length */
#define ADVNTR_FLAG_ANTIDIAGONAL 16u  /* keep short reads of a large batch on the one-read-per-wavefront kernel instead of
                                         the row-blocked one (several reads per wavefront); results are identical */
#define ADVNTR_FLAG_BOTH_STRANDS 32u  /* the n_reads given are forward strands; the engine also scores their reverse
                                         complements (made on the device) as calls n_reads .. 2*n_reads-1, call n_reads+i
                                         against read i's model: every output array then holds 2*n_reads entries
                                         (process_unmapped_read scores both strands, vntr_finder.py:239-242) */

#define ADVNTR_FLAG_DEEP_TILES   64u  /* short reads go to the row-blocked kernels in tiles of full back-to-back depth (several
                                         reads per lane group, one behind the other) whatever the batch size and to the very
                                         end of the batch; by default small batches and the tail of a large one get shallower
                                         tiles so that the work spreads over the CUs.  Results are identical */

#define ADVNTR_FLAG_SECOND_QUEUE 128u /* the batch's stream is of a second class (a stream priority of its own): it never shares
                                         a hardware queue with the stream of a batch created without the flag.  Streams of one
                                         class are spread over a few hardware queues by the runtime, and two that land on the
                                         same one run their kernels strictly one after the other; a caller that keeps two
                                         copies of a batch so that the next pass starts while the previous one's last
                                         workgroups drain creates one copy with the flag and one without.  Results identical */

/* out_summary layout: ADVNTR_SUMMARY_INTS int32 per read (hmm_utils.py line numbers in brackets) */
#define ADVNTR_SUMMARY_INTS   8
#define ADVNTR_SUM_RU         0   /* get_number_of_repeats_in_vpath            [155-188] */
#define ADVNTR_SUM_MATCHES    1   /* get_number_of_matches_in_vpath            [191-197] */
#define ADVNTR_SUM_REPEAT_BP  2   /* get_number_of_repeat_bp_matches_in_vpath  [200-206] */
#define ADVNTR_SUM_LEFT_BP    3   /* get_left_flanking_region_size_in_vpath    [271-277]; also the left denominator of [209-268] */
#define ADVNTR_SUM_RIGHT_BP   4   /* get_right_flanking_region_size_in_vpath   [280-286]; right denominator */
#define ADVNTR_SUM_LEFT_MATCH 5   /* left_flanking_matches of get_flanking_regions_matching_rate [209-268] */
#define ADVNTR_SUM_RIGHT_MATCH 6  /* right_flanking_matches                                               */
#define ADVNTR_SUM_PATH_LEN   7   /* number of states on the path incl. model start/end; 0 = impossible read */

/* state_class bits (one uint16 per state; 0 for all = "no name information") */
#define ADVNTR_SC_EMIT        0x0001  /* is_emitting_state: name starts with M or I          [122-126] */
#define ADVNTR_SC_MATCH       0x0002  /* is_match_state: name starts with M                  [116-119] */
#define ADVNTR_SC_SUFFIX      0x0004  /* name ends with 'suffix' (left flank block)                    */
#define ADVNTR_SC_PREFIX      0x0008  /* name ends with 'prefix' (right flank block)                   */
#define ADVNTR_SC_UNIT_START  0x0010  /* name starts with 'unit_start'                                 */
#define ADVNTR_SC_UNIT_END    0x0020  /* name starts with 'unit_end'                                   */
#define ADVNTR_SC_SKIP        0x0040  /* 'start' or 'end' occurs in the name (skipped at [227-228])    */
#define ADVNTR_SC_FIX         0x0080  /* name ends with 'fix' ([204])                                  */
#define ADVNTR_SC_BASE_SHIFT  8       /* bits 8-9: flank base an M*_suffix / M*_prefix state is compared
                                         with at [236,248]; bit 10 set when that base is known         */
#define ADVNTR_SC_BASE_VALID  0x0400

typedef struct advntr_hmm advntr_hmm;       /* a baked model resident on the current device            */
typedef struct advntr_batch advntr_batch;   /* a device-resident batch of reads bound to models        */

/* ---- device ------------------------------------------------------------------------------- */
int advntr_device_count(void);
int advntr_set_device(int device);           /* per process: one process per GPU                        */
const char *advntr_last_error(void);         /* thread-local message of the last failing call           */
const char *advntr_version(void);
int advntr_host_threads(void);               /* CPUs the bulk host-side calls may use: hardware threads cut down to the CPU quota of
                                              * the process's control group; ADVNTR_HOST_THREADS overrides                       */
void advntr_trim(void);                      /* release the cached device buffers (batches reuse them between calls) */

/* ---- model (replaces the malloc'd CSR owned by a baked HiddenMarkovModel, hmm.pyx:935-1023) ---
 * States are ordered emitting-first (index < silent_start) then silent in topological order, as bake()
 * leaves them (hmm.pyx:850-887).  in_ptr/in_src/in_logp are bake()'s in_edge_count / in_transitions /
 * in_transition_log_probabilities: in-edges of state l are in_src[in_ptr[l] .. in_ptr[l+1]) in
 * graph.edges_iter() order -- that order decides ties (strict '>', hmm.pyx:2039,2060,2080).
 * emis_logp: silent_start x 4 log-probabilities (DiscreteDistribution.bake, distributions.pyx:1366-1385).
 * state_class: m entries of ADVNTR_SC_* bits or NULL.  Returns NULL on error.                        */
advntr_hmm *advntr_hmm_create(int32_t m, int32_t silent_start, int32_t start_index, int32_t end_index,
                              int32_t n_edges, const int32_t *in_ptr, const int32_t *in_src,
                              const double *in_logp, const double *emis_logp, const uint16_t *state_class);
void advntr_hmm_destroy(advntr_hmm *model);  /* replaces free_bake_buffers, hmm.pyx:332-346.  A model should outlive the
                                              * batches made from it; one destroyed earlier is only marked and is released
                                              * by the last of its batches (advntr_batch_destroy, after that batch's stream
                                              * has been synchronised) -- its device memory is reused by later uploads      */
/* 1 if the model was recognised as a flank-repeats-flank read matcher (hmm_utils.py:553-595) and has
 * a column program for the anti-diagonal kernel; 0 if it runs on the generic-CSR kernel.              */
int advntr_hmm_has_column_program(const advntr_hmm *model);
int advntr_hmm_info(const advntr_hmm *model, int32_t *m, int32_t *silent_start, int32_t *n_edges,
                    int32_t *n_columns);

/* ---- one-shot scoring from host buffers (replaces N calls of Model.viterbi) -------------------
 * bases: concatenated base codes; read r is bases[read_off[r] .. read_off[r+1]); read_model[r] indexes
 * models[].  out_logp[n_reads] (fp64; -inf for an impossible read, hmm.pyx:2100-2105).
 * out_summary[n_reads][ADVNTR_SUMMARY_INTS] or NULL.  With ADVNTR_FLAG_PATH: read r's path goes to
 * out_path[out_path_off[r] .. out_path_off[r+1]) (capacity), its length to out_path_len[r]
 * (0 = impossible, -2 = capacity too small).  The reference's own capacity is n+m (hmm.pyx:1953).
 * Returns ADVNTR_OK or an error; ADVNTR_ERR_SYMBOL is raised before anything is launched.             */
int advntr_viterbi_batch(advntr_hmm *const *models, int32_t n_models, const uint8_t *bases,
                         const int64_t *read_off, const int32_t *read_model, int32_t n_reads,
                         double *out_logp, int32_t *out_summary, int32_t *out_path,
                         const int64_t *out_path_off, int32_t *out_path_len, uint32_t flags);

/* sum-product twin (Model.log_probability, hmm.pyx:1258-1313) */
int advntr_forward_batch(advntr_hmm *const *models, int32_t n_models, const uint8_t *bases,
                         const int64_t *read_off, const int32_t *read_model, int32_t n_reads,
                         double *out_logp, uint32_t flags);

/* ---- device-resident batches (what bench.py and the multi-GPU driver use) ---------------------
 * create: uploads reads once; run: launches the kernels on the engine's stream, results stay in HBM;
 * fetch: copies results back.  run_timed brackets `iters` runs with HIP events on the launch stream
 * and returns the mean kernel-region milliseconds per run.                                           */
advntr_batch *advntr_batch_create(advntr_hmm *const *models, int32_t n_models, const uint8_t *bases,
                                  const int64_t *read_off, const int32_t *read_model, int32_t n_reads,
                                  uint32_t flags);
void advntr_batch_destroy(advntr_batch *batch);
int advntr_batch_run(advntr_batch *batch);
int advntr_batch_sync(advntr_batch *batch);
/* The NEXT advntr_batch_run leaves up to n_workgroups resident workgroup slots unclaimed (launches that would fill the
 * device only, at most 1/16 of a launch): what advntr_comm_gather_results_start asks for with peers, so that RCCL's
 * kernels run beside the pass -- exported so that a one-GPU rehearsal of a rank's share runs with the launch parameters
 * of the multi-GPU job (bench.py --emulate-ranks).  The request covers one pass.                                      */
int advntr_batch_reserve_next(advntr_batch *batch, int32_t n_workgroups);
int advntr_batch_run_timed(advntr_batch *batch, int32_t iters, float *ms_per_run);
/* Model.log_probability over a resident batch (hmm.pyx:1258-1313): the sum-product kernels on the uploaded reads,
 * results in the batch's logp array (fetch with advntr_batch_fetch; summaries untouched); _timed as run_timed.     */
int advntr_batch_forward(advntr_batch *batch);
int advntr_batch_forward_timed(advntr_batch *batch, int32_t iters, float *ms_per_run);
int advntr_batch_fetch(advntr_batch *batch, double *out_logp, int32_t *out_summary);
int advntr_batch_fetch_paths(advntr_batch *batch, int32_t *out_path, const int64_t *out_path_off,
                             int32_t *out_path_len);
/* The callers' keep / discard rule on the results as they are in HBM after advntr_batch_run -- what VNTRFinder does with
 * every scored read (vntr_finder.py): the strand with the larger log-probability (process_unmapped_read :242-246: the reverse
 * one iff logp < rev_logp; only with ADVNTR_FLAG_BOTH_STRANDS), recruit_read (:179-190: flank match rate >= 0.9, then
 * logp > scaled_score[model] * read_length for a locus with a trained score -- scaled_score NULL, NaN or 0: none,
 * get_min_score_to_select_a_read :174-177 -- else matches >= 0.9 * read_length and logp > -read_length), and
 * repeat_bp > min_repeat_bp (:251).  n_keep = survivors; fetch_recruited returns them IN READ ORDER: index of the (forward)
 * read, log-probability and summary record of the chosen strand, 1 where that strand is the reverse one (any output may be
 * NULL).  Only the survivors' records cross PCIe.                                                                          */
int advntr_batch_recruit(advntr_batch *batch, const double *scaled_score, int32_t min_repeat_bp, int32_t *n_keep);
int advntr_batch_fetch_recruited(advntr_batch *batch, int32_t *out_index, double *out_logp, int32_t *out_summary,
                                 uint8_t *out_reversed);
/* device addresses of the result arrays (fp64 logp[n_reads], int32 summary[n_reads][8]) so that a
 * multi-GPU driver can hand them to RCCL without a host round trip; valid until batch_destroy.      */
int advntr_batch_result_ptrs(advntr_batch *batch, void **d_logp, void **d_summary);
/* scratch / trellis bytes this batch holds in HBM (for DESIGN.md's layout accounting) */
int64_t advntr_batch_device_bytes(const advntr_batch *batch);
/* which kernels advntr_batch_run launches for this batch: NUL-terminated text, one line "kernel_name reads tiles useful"
 * per launch, in launch order (what `rocprofv3 --kernel-trace` will show); useful = share of the launch's lane-steps that
 * work on a cell of a read's trellis, in per mille (the rest is pipeline fill / drain and padding rows; -1: not accounted).  The routing depends only on read lengths,
 * the models' tables and the flags given at creation -- never on the environment.                             */
int advntr_batch_info(const advntr_batch *batch, char *buf, int32_t capacity);

/* ---- keyword prefilter (the stage upstream of the scoring path) ------------------------------------
 * Replaces the scan loop of the reference's adVNTR-Filtering binary (/root/reference/filtering/main.cc:
 * automaton build :56-157, per-read matching :247-283).  Keywords: concatenated base codes 0..3 with offsets
 * (any length >= 1; up to eight distinct lengths of <= 29 bases, longer ones share one slot: they are found through
 * their 29-base prefix and verified on the hits), kw_vntr[w] = caller's VNTR index of keyword w (the same string may belong to several VNTRs).
 * scan(): reads as base codes 0..3, 4 = any other symbol (resets a match, main.cc:44-55); returns unordered
 * (read, vntr, count) records with count >= 1 -- a read/vntr pair may be split over several records, sum them.
 * The order-dependent selection and printing (main.cc:286-331) is host logic (advntr_amd/filtering.py).
 * Returns ADVNTR_OK, or ADVNTR_ERR_TOO_LARGE with *n_out = records needed when capacity is too small.      */
typedef struct advntr_kwfilter advntr_kwfilter;
advntr_kwfilter *advntr_kwfilter_create(const uint8_t *kw_bases, const int64_t *kw_off, const int32_t *kw_vntr,
                                        int32_t n_keywords);
void advntr_kwfilter_destroy(advntr_kwfilter *filter);
/* scan_text: the reads are spans [span_start[r], span_end[r]) of the text of a FASTA file (its sequence lines, found with
 * advntr_line_index); the text is uploaded as it is and mapped to base codes on the device the way the reference's
 * char_to_num does (main.cc:44-55: upper-case A,C,G,T; every other byte resets a match).                          */
int advntr_kwfilter_scan_text(advntr_kwfilter *filter, const char *text, int64_t n_bytes, const int64_t *span_start,
                              const int64_t *span_end, int32_t n_reads, int32_t *out_read, int32_t *out_vntr,
                              int32_t *out_count, int64_t capacity, int64_t *n_out, float *kernel_ms);
int advntr_kwfilter_scan(advntr_kwfilter *filter, const uint8_t *bases, const int64_t *read_off, int32_t n_reads,
                         int32_t *out_read, int32_t *out_vntr, int32_t *out_count, int64_t capacity, int64_t *n_out,
                         float *kernel_ms);

/* ---- native model builder (the step that feeds advntr_hmm_create) ---------------------------------------
 * Replaces, for the read-matcher model family, the Python construction route of the reference:
 * get_read_matcher_model and its sub-builders (advntr/hmm_utils.py:290-595), the profile parameters
 * (advntr/profile_hmm.py:13-161) and the pomegranate calls underneath (bake(merge=None) hmm.pyx:673-1123,
 * dense_transition_matrix :492-514, from_matrix :3146-3238, concatenate :584-615) -- 0.8-1.0 s per locus there.
 * Builds n_loci models on n_threads host threads (<= 0: the host's cores, at most 32).  Locus i: left/right flanking regions as
 * NUL-terminated ACGT strings (already trimmed to the wanted length), aligned repeat units
 * repeats[repeat_off[i] .. repeat_off[i+1]) (equal-length rows over ACGT and '-'), copies[i] repeat copies.
 * exp_fn: the exponential applied to log-probabilities where the reference calls numpy.exp (hmm.pyx:514); NULL =
 * libm exp.  It is called from the worker threads, one call at a time -- unless ADVNTR_BUILD_EXP_STRIDED_LOOP says that
 * exp_fn is really a strided inner loop `void loop(char **args, const intptr_t *n, const intptr_t *steps, void *user)`
 * (args = {in, out}, steps in bytes: the form of a NumPy ufunc inner loop), which must be thread-safe and is called from
 * all worker threads concurrently (a Python callback is serialised by the interpreter lock; numpy's own loop is not).
 * flags: ADVNTR_BUILD_ALIGN_REPEATS aligns
 * repeat units of unequal length with the built-in progressive aligner (the reference shells out to `muscle` there,
 * profile_hmm.py:166-171; parity with muscle is NOT claimed, see csrc/repeat_msa.h) -- without it they are an error.
 * out[i] receives the model or NULL; returns ADVNTR_OK or the first error (advntr_last_error names the locus).       */
typedef struct advntr_built advntr_built;
typedef void (*advntr_exp_fn)(const double *in, double *out, int64_t n, void *user);
#define ADVNTR_BUILD_ALIGN_REPEATS 0x1u
#define ADVNTR_BUILD_EXP_STRIDED_LOOP 0x2u
int advntr_build_read_matchers(int32_t n_loci, const char *const *left_flank, const char *const *right_flank,
                               const char *const *repeats, const int32_t *repeat_off, const int32_t *copies,
                               double max_error_rate, advntr_exp_fn exp_fn, void *user, int32_t n_threads,
                               uint32_t flags, advntr_built **out);
/* The aligner on its own: n units (ACGT) -> n rows of *width characters over ACGT and '-', written back to back into
 * out (capacity bytes).  ADVNTR_ERR_TOO_LARGE with *width set when capacity < n * *width.                          */
int advntr_align_repeats(const char *const *units, int32_t n, char *out, int64_t capacity, int32_t *width);
/* info[6] = m, silent_start, start_index, end_index, n_edges, bytes of the '\n'-joined state names (no NUL) */
int advntr_built_info(const advntr_built *built, int32_t *info);
/* the same for n models at once: info[6 i .. 6 i + 5] (a NULL model: six -1) */
int advntr_built_info_many(const advntr_built *const *built, int32_t n, int32_t *info);
/* copy out the arrays advntr_hmm_create takes (any pointer may be NULL to skip it) */
int advntr_built_export(const advntr_built *built, int32_t *in_ptr, int32_t *in_src, double *in_logp,
                        double *emis_logp, uint16_t *state_class, char *names);
/* advntr_hmm_create on the built arrays (current device) */
advntr_hmm *advntr_built_upload(const advntr_built *built);
/* The same for n models at once: kernel-side tables and column programs are prepared on n_threads host threads
 * (<= 0: the host's cores, at most 32), then ONE device allocation and ONE host-to-device copy carry all of them (a model database of
 * thousands of loci otherwise pays a hipMalloc + synchronous copy per model).  The models share that allocation; it
 * is released with the last of them (advntr_hmm_destroy each, as usual).  out[i] = model or NULL on error.          */
int advntr_built_upload_many(const advntr_built *const *built, int32_t n, int32_t n_threads, advntr_hmm **out);
void advntr_built_destroy(advntr_built *built);

/* ---- flank alignment for long reads (the stage upstream of the PacBio scoring path) ---------------------------
 * Replaces the two Bio.pairwise2.align.localms(read, flank, 1, -1, -1, -1) calls per read and strand of
 * VNTRFinder.check_if_flanking_regions_align_to_str (/root/reference/advntr/vntr_finder.py:324-365): local alignment
 * (match +1, mismatch -1, gap -1 per base) of flank pair_flank[p] (<= 128 bases) against read pair_read[p], n_pairs at
 * once, one per wavefront.  Reads / flanks as base codes 0..3 (anything else matches nothing), concatenated with
 * offsets.  pair_read[p] in [n_reads, 2 n_reads) stands for the REVERSE COMPLEMENT of read pair_read[p] - n_reads, made on
 * the device from the uploaded read (check_if_pacbio_read_spans_vntr tests both strands, :367-371); coordinates are then
 * positions in the reverse-complemented read.  out_score[p] = best local score (0 = no positive-scoring alignment), out_begin[p] = the `begin` of the
 * first alignment pairwise2 would return (max of the two start indices; -1 if none), out_end[p] = read index of its
 * last aligned base.  PARITY UNPINNED with respect to biopython (absent from the image), see csrc/flank_align.h.     */
int advntr_flank_align(const uint8_t *bases, const int64_t *read_off, int32_t n_reads, const uint8_t *flank_bases,
                       const int32_t *flank_off, int32_t n_flanks, const int32_t *pair_read, const int32_t *pair_flank,
                       int32_t n_pairs, int32_t *out_score, int32_t *out_begin, int32_t *out_end, float *kernel_ms);

/* ---- read encoding (host threads) ---------------------------------------------------------------------------
 * ASCII reads (concatenated, read r = ascii[read_off[r] .. read_off[r+1])) -> the base codes the scoring calls take:
 * A,C,G,T in either case -> 0..3, N/n -> 254, anything else -> 255; out_bad[r] = 0 for a clean read, 1 when it holds
 * N (the reference skips such reads before scoring, vntr_finder.py:237), 2 when it holds any other symbol (the
 * reference's Model.viterbi raises ValueError on those, hmm.pyx:72,79).                                          */
int advntr_encode_ascii(const char *ascii, const int64_t *read_off, int32_t n_reads, int32_t n_threads,
                        uint8_t *out_codes, uint8_t *out_bad);
/* The same for reads that are spans of a larger text (the sequence lines of a FASTA file): read r =
 * ascii[span_start[r] .. span_end[r]) -> out_codes[out_off[r] .. out_off[r+1]).  ADVNTR_ENCODE_CASE_SENSITIVE: only upper-
 * case A,C,G,T,N are symbols (what the reference's filter does, filtering/main.cc:44-55: lower case matches nothing).
 * advntr_line_index: start offsets of the lines of a text (a final line without newline counts), for the two-line FASTA
 * records that binary reads (main.cc:247-252); returns ADVNTR_ERR_TOO_LARGE with *n_lines set when capacity is short. */
#define ADVNTR_ENCODE_CASE_SENSITIVE 1u
int advntr_encode_spans(const char *ascii, const int64_t *span_start, const int64_t *span_end, int32_t n_reads,
                        uint32_t flags, int32_t n_threads, const int64_t *out_off, uint8_t *out_codes, uint8_t *out_bad);
int advntr_line_index(const char *text, int64_t n_bytes, int32_t n_threads, int64_t *line_start, int64_t capacity,
                      int64_t *n_lines);
/* The same for reads held as separate texts (one pointer per read, e.g. the buffers of the host language's strings: long
 * reads are not joined into one text first): read r = texts[r][0 .. out_off[r+1] - out_off[r]).                    */
int advntr_encode_texts(const char *const *texts, int32_t n_reads, uint32_t flags, int32_t n_threads,
                        const int64_t *out_off, uint8_t *out_codes, uint8_t *out_bad);

/* Pieces of encoded reads, as reads of their own: piece p = codes[read_off[r] + begin[p] .. read_off[r] + end[p]) of read
 * r = piece_read[p] (0 <= begin <= end <= length of the read), reverse-complemented when reverse[p] != 0, written to
 * out_codes[out_off[p] .. out_off[p+1]) -- the trimming of spanning long reads (read[left_begin : right_begin + flank size] of
 * the strand that spans, /root/reference/advntr/vntr_finder.py:338-356) done on the codes the flank alignment was fed with,
 * instead of slicing, upper-casing, complementing and re-encoding strings in the host language.  Codes above 3 (N, other
 * symbols) come out as 255, what the scoring calls reject.  Host threads, no GPU.                                        */
int advntr_cut_pieces(const uint8_t *codes, const int64_t *read_off, int32_t n_reads, const int32_t *piece_read,
                      const int64_t *begin, const int64_t *end, const uint8_t *reverse, int64_t n_pieces, int32_t n_threads,
                      const int64_t *out_off, uint8_t *out_codes);

/* ---- genotype caller on the summary records (the step downstream of scoring; host threads, no GPU) -----------
 * Replaces, for many loci at once, the Illumina aggregation of VNTRFinder.find_repeat_count_from_alignment_file after
 * read selection (/root/reference/advntr/vntr_finder.py:807-887: spanning / flanking split by
 * read_flanks_repeats_with_confidence :311-322, the >= 5 agreeing flanking reads rule, the accuracy filter's >= 3
 * spanning reads per RU count) and the maximum-likelihood diploid / haploid call of
 * find_genotype_based_on_observed_repeats (:473-532).  summaries: the ADVNTR_SUMMARY_INTS records of the SELECTED
 * (recruited, > 2 repeat bases) reads grouped by locus, locus i = records locus_off[i] .. locus_off[i+1].
 * out_genotype[i] = the two RU counts in the reference's order, or -1, -1 for None; out_prob[i] = its probability
 * (1e-20 when None; bit-equal to the reference's arithmetic); out_counts[i] (may be NULL) = recruited, spanning,
 * flanking read counts of its GenotypeResult.  The coverage-based estimate (average_coverage) stays with the caller. */
#define ADVNTR_GENOTYPE_ACCURACY_FILTER 1u
#define ADVNTR_GENOTYPE_HAPLOID         2u
int advntr_genotype_illumina(const int32_t *summaries, const int64_t *locus_off, int32_t n_loci, uint32_t flags,
                             int32_t min_left_flank, int32_t min_right_flank, int32_t n_threads,
                             int32_t *out_genotype, double *out_prob, int32_t *out_counts);
/* The PacBio counterpart: the tail of VNTRFinder.get_dominant_copy_numbers_from_spanning_reads
 * (/root/reference/advntr/vntr_finder.py:568-580) for many loci at once.  ru_counts: the RU count of every spanning read
 * (ADVNTR_SUM_RU of its summary record) in the order the reads were scored, locus i = ru_counts[locus_off[i] ..
 * locus_off[i+1]); with ADVNTR_GENOTYPE_ACCURACY_FILTER only RU counts seen in >= 3 reads survive
 * (Counter.most_common order, :571-575); then find_genotype_based_on_observed_repeats (:485-532).  A locus without reads
 * gets -1, -1 and probability 0 (the reference returns (None, 0), :535-537).  Outputs as above.                       */
int advntr_genotype_observed(const int32_t *ru_counts, const int64_t *locus_off, int32_t n_loci, uint32_t flags,
                             int32_t n_threads, int32_t *out_genotype, double *out_prob);

/* ---- multi-GPU: the gather of the result records over RCCL / xGMI --------------------------------------------
 * One process per GPU; whole loci (with all their reads) are assigned to ranks (advntr_amd/sharding.py), so scoring
 * needs no exchange at all and the ONLY collective of the path is the final gather of the per-read records (fp64
 * log-probability + ADVNTR_SUMMARY_INTS x int32) to a root rank.  The reference has no counterpart: it scores loci
 * serially (/root/reference/advntr/genome_analyzer.py:280-297) and its optional worker processes append to a
 * multiprocessing.Manager().list() (/root/reference/advntr/vntr_finder.py:425-427).
 * Rank 0 makes the 128-byte id (ncclGetUniqueId) and the host hands it to the other ranks by any means (a file, a
 * socket: advntr_amd/comm.py); every rank then calls advntr_comm_create on its own device.  librccl.so is loaded on
 * first use.  counts[] arrays have one entry per rank and must be the same on every rank.                        */
typedef struct advntr_comm advntr_comm;
/* ADVNTR_OK if librccl (ADVNTR_RCCL_LIB, or the loader's librccl.so) loads with every entry point this file needs; no GPU
 * call, no collective.  ncclCommInitRank only returns when EVERY rank has called it, so the ranks exchange this answer
 * first (advntr_amd/comm.py) and nobody enters the collective when somebody cannot.                               */
int advntr_comm_available(void);
int advntr_comm_unique_id(uint8_t *id128);
advntr_comm *advntr_comm_create(int32_t rank, int32_t world, const uint8_t *id128);   /* NULL on error */
void advntr_comm_destroy(advntr_comm *comm);
int advntr_comm_info(const advntr_comm *comm, int32_t *rank, int32_t *world);
int advntr_comm_allgather_i64(advntr_comm *comm, int64_t mine, int64_t *out_world);
int advntr_comm_allreduce_max_f64(advntr_comm *comm, double *inout);
int advntr_comm_barrier(advntr_comm *comm);
/* gather-v of the batch's records as they are after the launches queued so far: returns at once (the records are
 * copied to staging buffers on the batch's own stream, so the next advntr_batch_run may follow immediately and
 * overlaps the transfer); counts[r] = reads of rank r's batch.  _finish waits; on the root out_logp / out_summary
 * (host, sum(counts) records in rank order, either may be NULL) receive everything.                              */
int advntr_comm_gather_results_start(advntr_comm *comm, advntr_batch *batch, int32_t root, const int64_t *counts);
int advntr_comm_gather_results_finish(advntr_comm *comm, double *out_logp, int32_t *out_summary);
/* milliseconds the gather finished last took on the communicator's stream, from the staging of its pass's records to
 * the last record's arrival (HIP events): the transfer alone if it ran beside the next pass's kernels, about a pass if it
 * had to wait for them (with peers the batch's later launches leave 8 workgroup slots unclaimed for RCCL's kernels) */
int advntr_comm_last_gather_ms(const advntr_comm *comm, float *ms);
/* ragged gather of host byte strings through the devices (the genotype driver's per-locus result rows):
 * counts[r] = bytes of rank r; dst (root only) receives them rank after rank.                                   */
int advntr_comm_gather_bytes(advntr_comm *comm, int32_t root, const void *src, const int64_t *counts, void *dst);

#ifdef __cplusplus
}
#endif
#endif /* ADVNTR_HIP_H */
