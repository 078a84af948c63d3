// abi_recruit.h -- the callers' keep / discard rule on the device (included by engine.hip).
//
// What VNTRFinder does with every scored read before anything else looks at it (/root/reference/advntr/vntr_finder.py):
//   process_unmapped_read :235-254   the strand with the larger log-probability is kept (`if logp < rev_logp`: the reverse one),
//                                    then recruit_read, then `repeat_bps > min_repeat_bp_to_add_read` selects the read;
//   recruit_read          :179-190   flank match rate (hmm_utils.get_flanking_regions_matching_rate :209-268: the smaller of
//                                    matches / bases of the two flanks, 1 for a flank the read does not touch) >= 0.9, and
//                                    either logp > scaled_score * read_length (a locus with a trained score) or
//                                    matches >= 0.9 * read_length and logp > -read_length (a locus without one).
// All of it is a dozen operations on the 40-byte record the scoring kernels leave in HBM, so it runs there: one thread per
// read, the survivors compacted IN READ ORDER (the genotype caller's tie-breaks depend on the order of a locus's reads) by a
// count per wavefront + one scan + a scatter, and only the survivors' records cross PCIe -- instead of every call's record
// followed by the same rule in numpy (advntr_amd/vntr_finder.py: recruit_mask), which was a quarter of the end-to-end time.
// The arithmetic is the reference's: double division and comparison, `>=` / `>` as written there.
#pragma once

struct RecruitArgs {
    const double *logp;          // per call
    const int32_t *summary;      // per call, 8 ints
    const int64_t *read_off;     // n_fwd + 1 (forward reads)
    const int32_t *read_model;   // per forward read
    const double *scaled_score;  // per model; NaN or 0: the locus has no trained score (get_min_score_to_select_a_read :174-177)
    int32_t n_fwd;               // forward reads; with both_strands call n_fwd + i is read i's reverse complement
    int32_t both_strands;
    int32_t min_repeat_bp;       // keep reads with repeat_bp > this (min_repeat_bp_to_add_read, vntr_finder.py:58)
    uint8_t *choice;             // per forward read: bit 0 keep, bit 1 reverse strand chosen
    int32_t *wave_count;         // kept reads per group of 64 reads
};

__device__ __forceinline__ bool recruit_rule(const double logp, const int32_t *__restrict__ s, const double n, const double scaled)
{
    if (s[ADVNTR_SUM_PATH_LEN] <= 2) return false;                 // impossible read: (-inf, None) in the reference
    const double lb = (double)s[ADVNTR_SUM_LEFT_BP], rb = (double)s[ADVNTR_SUM_RIGHT_BP];
    const double left = lb != 0.0 ? (double)s[ADVNTR_SUM_LEFT_MATCH] / lb : 1.0;
    const double right = rb != 0.0 ? (double)s[ADVNTR_SUM_RIGHT_MATCH] / rb : 1.0;
    if ((right < left ? right : left) < 0.90) return false;
    const bool has_score = !(scaled != scaled) && scaled != 0.0;
    if (has_score) return logp > scaled * n;
    return (double)s[ADVNTR_SUM_MATCHES] >= 0.9 * n && logp > -n;
}

__global__ void __launch_bounds__(256) recruit_mark_kernel(RecruitArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (i < a.n_fwd) {
        int call = i;
        bool rev = false;
        if (a.both_strands && a.logp[i] < a.logp[a.n_fwd + i]) { call = a.n_fwd + i; rev = true; }
        const int32_t *s = a.summary + (int64_t)call * ADVNTR_SUMMARY_INTS;
        const double n = (double)(a.read_off[i + 1] - a.read_off[i]);
        keep = recruit_rule(a.logp[call], s, n, a.scaled_score ? a.scaled_score[a.read_model[i]] : __longlong_as_double(0x7ff8000000000000ll)) &&
               s[ADVNTR_SUM_REPEAT_BP] > a.min_repeat_bp;
        a.choice[i] = (uint8_t)((keep ? 1 : 0) | (rev ? 2 : 0));
    }
    const unsigned long long kept = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && i < a.n_fwd) a.wave_count[i >> 6] = __popcll(kept);
}

// exclusive scan of the per-wavefront counts, one workgroup (n_waves is reads / 64: tens of thousands at most); total -> out_total
__global__ void __launch_bounds__(1024) recruit_scan_kernel(int32_t *wave_count, const int n_waves, int32_t *out_total)
{
    __shared__ int32_t part[1024];
    const int tid = threadIdx.x;
    const int per = (n_waves + 1023) / 1024, lo = tid * per, hi = min(n_waves, lo + per);
    int sum = 0;
    for (int j = lo; j < hi; ++j) sum += wave_count[j];
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int j = lo; j < hi; ++j) { const int c = wave_count[j]; wave_count[j] = run; run += c; }
    if (tid == 1023) *out_total = part[1023];
}

__global__ void __launch_bounds__(256) recruit_gather_kernel(RecruitArgs a, int32_t *out_index, double *out_logp, int32_t *out_summary,
                                                             uint8_t *out_reversed)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint8_t ch = i < a.n_fwd ? a.choice[i] : (uint8_t)0;
    const unsigned long long kept = __ballot(ch & 1);
    if (!(ch & 1)) return;
    const int lane = threadIdx.x & 63;
    const int pos = a.wave_count[i >> 6] + __popcll(kept & ((1ull << lane) - 1ull));
    const int call = (ch & 2) ? a.n_fwd + i : i;
    out_index[pos] = i;
    out_logp[pos] = a.logp[call];
    out_reversed[pos] = (uint8_t)((ch >> 1) & 1);
    const int4 *s = (const int4 *)(a.summary + (int64_t)call * ADVNTR_SUMMARY_INTS);
    int4 *o = (int4 *)(out_summary + (int64_t)pos * ADVNTR_SUMMARY_INTS);
    o[0] = s[0];
    o[1] = s[1];
}

// Apply the rule to the batch's results as they are in HBM (after advntr_batch_run, same stream); n_keep = survivors.
// scaled_score: one per model of the batch or NULL (no locus has a trained score).
extern "C" int advntr_batch_recruit(advntr_batch *B, const double *scaled_score, int32_t min_repeat_bp, int32_t *n_keep)
{
    if (!B || !n_keep) return fail(ADVNTR_ERR_ARG, "advntr_batch_recruit: bad argument");
    if (B->flags & ADVNTR_FLAG_NO_SUMMARY) return fail(ADVNTR_ERR_ARG, "advntr_batch_recruit: the batch was created without summaries");
    const bool both = (B->flags & ADVNTR_FLAG_BOTH_STRANDS) != 0;
    const int n_fwd = both ? B->n_reads / 2 : B->n_reads;
    *n_keep = 0;
    B->n_recruited = 0;
    if (n_fwd == 0) return ADVNTR_OK;
    const int n_waves = (n_fwd + 63) / 64;
    int rc;
    if (!B->d_rec_ready) {
        // all or nothing: `d_rec_ready` is set behind the last allocation, so a call after a failed one allocates what is
        // missing instead of launching the kernels on null pointers (what a failed call did get stays with the batch's blocks)
        if (!B->d_rec_choice && (rc = B->dmalloc(&B->d_rec_choice, (size_t)n_fwd))) return rc;
        if (!B->d_rec_count && (rc = B->dmalloc(&B->d_rec_count, (size_t)n_waves + 1))) return rc;
        if (!B->d_rec_scaled && (rc = B->dmalloc(&B->d_rec_scaled, B->models.size()))) return rc;
        if (!B->d_rec_index && (rc = B->dmalloc(&B->d_rec_index, (size_t)n_fwd))) return rc;
        if (!B->d_rec_logp && (rc = B->dmalloc(&B->d_rec_logp, (size_t)n_fwd))) return rc;
        if (!B->d_rec_summary && (rc = B->dmalloc(&B->d_rec_summary, (size_t)n_fwd * ADVNTR_SUMMARY_INTS))) return rc;
        if (!B->d_rec_reversed && (rc = B->dmalloc(&B->d_rec_reversed, (size_t)n_fwd))) return rc;
        B->d_rec_ready = true;
    }
    RecruitArgs a{};
    a.logp = B->d_logp; a.summary = B->d_summary; a.read_off = B->d_read_off; a.read_model = B->d_read_model;
    a.scaled_score = nullptr;
    if (scaled_score) {
        HIP_TRY(hipMemcpyAsync(B->d_rec_scaled, scaled_score, B->models.size() * sizeof(double), hipMemcpyHostToDevice, B->stream));
        a.scaled_score = B->d_rec_scaled;
    }
    a.n_fwd = n_fwd; a.both_strands = both ? 1 : 0; a.min_repeat_bp = min_repeat_bp;
    a.choice = B->d_rec_choice; a.wave_count = B->d_rec_count;
    const int grid = (n_fwd + 255) / 256;
    hipLaunchKernelGGL(recruit_mark_kernel, dim3(grid), dim3(256), 0, B->stream, a);
    hipLaunchKernelGGL(recruit_scan_kernel, dim3(1), dim3(1024), 0, B->stream, B->d_rec_count, n_waves, B->d_rec_count + n_waves);
    hipLaunchKernelGGL(recruit_gather_kernel, dim3(grid), dim3(256), 0, B->stream, a, B->d_rec_index, B->d_rec_logp, B->d_rec_summary,
                       B->d_rec_reversed);
    HIP_TRY(hipGetLastError());
    int32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, B->d_rec_count + n_waves, sizeof total, hipMemcpyDeviceToHost, B->stream));
    HIP_TRY(hipStreamSynchronize(B->stream));
    B->n_recruited = total;
    *n_keep = total;
    return ADVNTR_OK;
}

// The survivors of the last advntr_batch_recruit, in read order: index of the forward read, the chosen strand's
// log-probability and summary record, whether that strand is the reverse one.  Any output may be NULL.
extern "C" int advntr_batch_fetch_recruited(advntr_batch *B, int32_t *out_index, double *out_logp, int32_t *out_summary,
                                            uint8_t *out_reversed)
{
    if (!B) return fail(ADVNTR_ERR_ARG, "advntr_batch_fetch_recruited: null batch");
    const size_t n = (size_t)B->n_recruited;
    if (!n) return ADVNTR_OK;
    if (out_index) HIP_TRY(hipMemcpyAsync(out_index, B->d_rec_index, n * sizeof(int32_t), hipMemcpyDeviceToHost, B->stream));
    if (out_logp) HIP_TRY(hipMemcpyAsync(out_logp, B->d_rec_logp, n * sizeof(double), hipMemcpyDeviceToHost, B->stream));
    if (out_summary)
        HIP_TRY(hipMemcpyAsync(out_summary, B->d_rec_summary, n * ADVNTR_SUMMARY_INTS * sizeof(int32_t), hipMemcpyDeviceToHost, B->stream));
    if (out_reversed) HIP_TRY(hipMemcpyAsync(out_reversed, B->d_rec_reversed, n, hipMemcpyDeviceToHost, B->stream));
    HIP_TRY(hipStreamSynchronize(B->stream));
    return ADVNTR_OK;
}
