// keyword_filter.h -- multi-keyword read prefilter on the GPU.
//
// Replaces the scan loop of the reference's Aho-Corasick filter (/root/reference/filtering/main.cc:247-283):
// for every read, count per VNTR how many (position, keyword) matches occur.  With keywords of a few fixed
// lengths (adVNTR cuts 15-mers, vntr_finder.py:140-153 / genome_analyzer.py:180) the automaton is equivalent to
// exact k-mer lookup; keywords of more than 29 bases (the long-read mode's two 80-base flanks per VNTR,
// vntr_finder.py:151-152) are found through their 29-base prefix and verified base by base on the (rare) hits: a thread slides a 2-bit-packed window over its read (any symbol other than A,C,G,T -- code 4
// -- restarts the window, like symbol 4 in main.cc:44-55), tests a 64 KiB bit-set staged in LDS and, on a set bit,
// probes an open-addressing table of packed keywords in HBM/L2.  Hits are rare; they are tallied in four
// per-thread (vntr, count) slots, overflow goes out as single events.  Everything that depends on read ORDER
// (the 6000-read cap, sorting, the 2000+1 print quirk: main.cc:286-331) stays on the host, which gets
// (read, vntr, count) triples.  The kernel streams 1 byte per base once: HBM-read bound by design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KWF_MAX_LENGTHS 8
#define KWF_BITSET_BITS (1u << 20)          // 128 KiB of LDS: one 1024-thread workgroup per CU, two bits per keyword
#define KWF_EMPTY 0xffffffffffffffffull
#define KWF_SLOTS 4
#ifndef KWF_BATCH
#define KWF_BATCH 4                          // positions whose fingerprint loads are in flight together (4 or 8)
#endif

#define KWF_MAX_PACKED 29                    // bases in a 64-bit key (58 bits + a 6-bit tag)
#define KWF_LONG_TAG 30                      // tag of the 29-base prefix of a longer keyword

struct KwfLongRec {                          // a keyword of more than 29 bases: what follows its 29-base prefix
    uint32_t rest_off, rest_len;             // long_bases[rest_off .. rest_off + rest_len)
    int32_t owner, pad;
};

struct KwfDevice {
    int32_t n_lengths;
    int32_t length[KWF_MAX_LENGTHS];         // window length in bases (29 for the prefixes of longer keywords)
    int32_t tag[KWF_MAX_LENGTHS];            // key tag: the length itself, or KWF_LONG_TAG
    uint64_t mask[KWF_MAX_LENGTHS];          // 2*L low bits
    uint64_t table_mask;                     // slots - 1 (one table for all lengths; the length is mixed into the key hash)
    const uint64_t *keys;                    // slots, KWF_EMPTY = free; key = packed bases | length << 58
    const uint32_t *vals;                    // slots: first index into ids[] | count << 24
    const int32_t *ids;                      // vntr index per (keyword string, owner) pair
    const uint32_t *bitset;                  // KWF_BITSET_BITS / 32 words
    const uint4 *fp_buckets;                 // second-level filter: buckets of eight 16-bit fingerprints (0 = free), ONE 16-byte
    uint32_t bucket_mask;                    // load per tested position, no probe chain; 2 MiB (L2-resident) at ~320 k keywords
    const KwfLongRec *long_recs;             // keywords longer than 29 bases (the reference's long-read mode cuts two
    const uint8_t *long_bases;               // 80-base flanks per VNTR, vntr_finder.py:151-152), chained off their prefix
    // keyword sets of ONE length of at most 16 bases (adVNTR's 15-mers): the filters of keyword_filter_short_kernel -- a
    // blocked Bloom filter of KWF_BITSET_BITS bits for the LDS (two bits of one 32-bit word per keyword) and a second one of
    // ~16 bits per keyword in L2 (three bits of one word); 0 words: not applicable (several lengths, or longer keywords)
    const uint32_t *short_bloom;
    const uint32_t *short_l2;
    uint32_t short_l2_mask;                  // words - 1
    int32_t short_ok;
    // ... and their exact table: {32 bits of packed bases, value} in ONE 8-byte entry, open addressing from a slot the filters'
    // hash gives; value = KWF_SHORT_ONE | VNTR index for a keyword of one VNTR (no second look-up), first index into ids[] |
    // count << 24 for a shared one, KWF_SHORT_EMPTY for a free slot
    const uint2 *short_table;
    uint32_t short_table_mask;
};

struct KwfArgs {
    KwfDevice f;
    const uint8_t *bases;                    // codes 0..3, anything above = any other symbol; or (ASCII kernel) the text itself
    const int64_t *span_start, *span_end;    // read r = bases[span_start[r] .. span_end[r])
    int32_t n_reads;
    int32_t *out_read, *out_vntr, *out_count;
    unsigned long long *n_out;               // atomic cursor
    int64_t capacity;
};

// 32-bit mixing of the 64-bit packed key (two 32-bit multiplies; host and device must agree: the table and the
// bit-set are built on the host with the same function).  Returns 64 bits: low half -> table slot, high -> bit-set.
__host__ __device__ __forceinline__ uint64_t kwf_hash(uint64_t k)
{
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t x = lo ^ (hi * 0x9E3779B1u);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13;
    uint32_t y = (x ^ hi) * 0xC2B2AE35u;
    y ^= y >> 16;
    return ((uint64_t)y << 40) | x;
}

// Hashes of a keyword of at most 16 bases (32 bits of packed bases) for the blocked Bloom filters of
// keyword_filter_short_kernel: 24-bit multiplies only (full rate on the vector ALU; a 32-bit multiply is a quarter-rate
// instruction), as selective as murmur's finaliser on the 317 k-keyword set.  First level (LDS, 32 768 words): word = low 15
// bits, two bits inside the word from the next 2 x 5 bits.  Second level (L2): a second hash of the first, two bits in a word.
// (The multiplies take the low 24 bits of their operands: bits of a window register above the keyword's 2 L never matter.)
__host__ __device__ __forceinline__ uint32_t kwf_h24(uint32_t k)
{
    // (one round is as selective as two, and as murmur's finaliser, on the 317 k-keyword set: 21.5 % of random 15-mers and of
    // keywords with one base changed pass the first level either way)
    const uint32_t a = k & 0xffffffu, b = (k >> 6) & 0xffffffu;
    uint32_t x = a * 0x9E3779u + b * 0x85EBCBu;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t kwf_h24b(uint32_t y)
{
    uint32_t z = (y & 0xffffffu) * 0x85EBCBu + (y >> 9);
    z ^= z >> 15;
    return z;
}
#define KWF_SHORT_EMPTY 0xffffffffu
#define KWF_SHORT_ONE 0x80000000u
__host__ __device__ __forceinline__ uint32_t kwf_short_slot(uint32_t z) { return (z & 0xffffffu) * 0x9E3779u + (z >> 8); }
#define KWF_BLOOM_WORDS (KWF_BITSET_BITS / 32)         // 32 768 words of 32 bits
__host__ __device__ __forceinline__ uint32_t kwf_bloom_mask1(uint32_t y) { return (1u << ((y >> 15) & 31u)) | (1u << ((y >> 20) & 31u)); }
__host__ __device__ __forceinline__ uint32_t kwf_bloom_mask2(uint32_t z, int word_bits)
{
    return (1u << ((z >> word_bits) & 31u)) | (1u << ((z >> (word_bits + 5 > 27 ? 27 : word_bits + 5)) & 31u));
}

// fingerprint of a key (odd, never 0 and never the overflow marker) and its bucket in the fingerprint table
#define KWF_FP_OVERFLOW 0xffffu              // in a bucket's last slot: more keys hashed here than fit -> "maybe" for every key
__host__ __device__ __forceinline__ uint16_t kwf_fp(uint64_t h)
{
    const uint16_t f = (uint16_t)(((h >> 24) & 0xffffu) | 1u);
    return f == KWF_FP_OVERFLOW ? (uint16_t)0xfffdu : f;
}
__host__ __device__ __forceinline__ uint32_t kwf_fp_bucket(uint64_t h, uint32_t mask) { return (uint32_t)(h >> 8) & mask; }

// true if the key's bucket holds its fingerprint (or has overflowed): one 16-byte load, compared in registers
__device__ __forceinline__ bool kwf_fp_maybe(const uint4 b, const uint64_t h)
{
    const unsigned want = kwf_fp(h), w2 = want | (want << 16);
    unsigned hit = 0;
    const unsigned words[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned x = words[i] ^ w2;                             // a 16-bit half is zero where a slot equals the fingerprint
        hit |= ((x & 0xffffu) == 0u) | ((x >> 16) == 0u);
    }
    return hit != 0u || (b.w >> 16) == KWF_FP_OVERFLOW;
}

__device__ __forceinline__ void kwf_emit(const KwfArgs &a, int read, int vntr, int count)
{
    const unsigned long long pos = atomicAdd(a.n_out, 1ull);
    if ((int64_t)pos < a.capacity) {
        a.out_read[pos] = read;
        a.out_vntr[pos] = vntr;
        a.out_count[pos] = count;
    }
}

// tally one table hit (keyword string found): +1 for every VNTR that owns the string
__device__ __forceinline__ void kwf_tally(const KwfArgs &a, const uint32_t v, const int r, int (&svid)[KWF_SLOTS],
                                          int (&scnt)[KWF_SLOTS])
{
    const int first = (int)(v & 0xffffffu), cnt = (int)(v >> 24);
    for (int q = 0; q < cnt; ++q) {
        const int vid = a.f.ids[first + q];
        bool done = false;
#pragma unroll
        for (int s = 0; s < KWF_SLOTS; ++s) {
            if (!done && (svid[s] == vid || svid[s] < 0)) { svid[s] = vid; scnt[s] += 1; done = true; }
        }
        if (!done) kwf_emit(a, r, vid, 1);      // more than 4 VNTRs in one read: single events
    }
}

// ASCII text -> the code of a base: upper-case A, C, G, T -> 0..3, anything else -> 4 (the reference's char_to_num,
// filtering/main.cc:44-55, is case sensitive)
__device__ __forceinline__ unsigned kwf_code_of_ascii(const unsigned c)
{
    const unsigned t = c - 0x41u;                                     // 'A' 0, 'C' 2, 'G' 6, 'T' 19
    const bool valid = t < 20u && ((0x80045u >> t) & 1u);
    const unsigned x = (c >> 1) & 3u;                                 // A 0, C 1, G 3, T 2
    return valid ? (x ^ (x >> 1)) : 4u;
}

// one owner's count
__device__ __forceinline__ void kwf_tally_one(const KwfArgs &a, const int vid, const int r, int (&svid)[KWF_SLOTS],
                                              int (&scnt)[KWF_SLOTS])
{
    bool done = false;
#pragma unroll
    for (int s = 0; s < KWF_SLOTS; ++s) {
        if (!done && (svid[s] == vid || svid[s] < 0)) { svid[s] = vid; scnt[s] += 1; done = true; }
    }
    if (!done) kwf_emit(a, r, vid, 1);
}

// the window ending at read position `pos` equals the 29-base prefix of one or more long keywords: compare what follows
template <bool ASCII>
__device__ __forceinline__ void kwf_tally_long(const KwfArgs &a, const uint32_t v, const int r, const uint8_t *__restrict__ seq,
                                               const int pos, const int n, int (&svid)[KWF_SLOTS], int (&scnt)[KWF_SLOTS])
{
    const int first = (int)(v & 0xffffffu), cnt = (int)(v >> 24);
    for (int q = 0; q < cnt; ++q) {
        const KwfLongRec rec = a.f.long_recs[first + q];
        if (pos + 1 + (int64_t)rec.rest_len > n) continue;
        const uint8_t *want = a.f.long_bases + rec.rest_off;
        bool same = true;
        for (uint32_t i = 0; i < rec.rest_len && same; ++i)
            same = (ASCII ? kwf_code_of_ascii(seq[pos + 1 + i]) : (unsigned)seq[pos + 1 + i]) == (unsigned)want[i];
        if (same) kwf_tally_one(a, rec.owner, r, svid, scnt);
    }
}

template <bool ASCII>
__device__ __forceinline__ void kwf_probe(const KwfArgs &a, const uint64_t key, uint64_t slot, uint64_t k, const int r,
                                          const uint8_t *__restrict__ seq, const int pos, const int n,
                                          int (&svid)[KWF_SLOTS], int (&scnt)[KWF_SLOTS])
{
    for (uint64_t probes = 0; probes <= a.f.table_mask; ++probes) {
        if (k == KWF_EMPTY) return;
        if (k == key) {
            if ((key >> 58) == KWF_LONG_TAG) kwf_tally_long<ASCII>(a, a.f.vals[slot], r, seq, pos, n, svid, scnt);
            else kwf_tally(a, a.f.vals[slot], r, svid, scnt);
            return;
        }
        slot = (slot + 1) & a.f.table_mask;
        k = a.f.keys[slot];
    }
}

#define KWF_BLOCK 1024      // 16 waves share one 128 KiB bit-set (one workgroup per CU)
// ASCII: the reads are spans of the uploaded FASTA text and are mapped to codes on the fly (no encoded copy exists
// anywhere: the host only indexes the lines)
// (Tried and measured on the 2 M-read bench, kernel ms: this structure 1.50; the same with all eight positions of a word in
// one batch 1.50; a branch-free version -- every position hashed and tested, hits handled by one out-of-line function,
// 8 k instead of 44 k instructions -- 2.12: skipping the work of positions that cannot match beats the smaller code.)
template <bool ASCII>
__global__ void __launch_bounds__(KWF_BLOCK) keyword_filter_kernel(KwfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t bits[];
    for (int i = threadIdx.x; i < (int)(KWF_BITSET_BITS / 32); i += KWF_BLOCK) bits[i] = a.f.bitset[i];
    __syncthreads();
    const bool single = a.f.n_lengths == 1;
    const int L0 = a.f.length[0];
    const uint64_t mask0 = a.f.mask[0], tag0 = (uint64_t)a.f.tag[0] << 58;
    for (int r = blockIdx.x * KWF_BLOCK + threadIdx.x; r < a.n_reads; r += gridDim.x * KWF_BLOCK) {
        const uint8_t *seq = a.bases + a.span_start[r];
        const int n = (int)(a.span_end[r] - a.span_start[r]);
        int svid[KWF_SLOTS], scnt[KWF_SLOTS];
#pragma unroll
        for (int s = 0; s < KWF_SLOTS; ++s) { svid[s] = -1; scnt[s] = 0; }
        uint64_t win = 0;
        int run = 0;                             // valid bases in the window
        for (int pb = 0; pb < n; pb += 64) {     // one 64-byte sector of the read per outer step: every byte of the
                                                 // read is fetched from HBM once (8-byte steps re-fetched evicted sectors)
            uint64_t sector[8];
            if (pb + 64 <= n) {
                __builtin_memcpy(sector, seq + pb, 64);
            } else {
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    sector[w] = 0;
                    if (pb + 8 * w + 8 <= n) __builtin_memcpy(&sector[w], seq + pb + 8 * w, 8);
                    else for (int q = 0; pb + 8 * w + q < n; ++q) sector[w] |= (uint64_t)seq[pb + 8 * w + q] << (8 * q);
                }
            }
#pragma unroll
          for (int wi = 0; wi < 8; ++wi) {
            const int p0 = pb + 8 * wi;
            if (p0 >= n) break;
            const uint64_t word = sector[wi];
            if (single) {
                // per half-word (KWF_BATCH positions): phase 1 keys + LDS bit-set test; phase 2 the bucket loads of the
                // survivors issued together (memory-level parallelism instead of one dependent L2 round trip per
                // base); phase 3 full-key probes of the (rare) fingerprint matches
#pragma unroll
                for (int half = 0; half < 8 / KWF_BATCH; ++half) {
                    uint64_t key[KWF_BATCH], hh[KWF_BATCH];
                    uint4 f0[KWF_BATCH];
                    unsigned live = 0;
#pragma unroll
                    for (int q = 0; q < KWF_BATCH; ++q) {
                        const int qq = half * KWF_BATCH + q;
                        unsigned c = (unsigned)(word >> (8 * qq)) & 0xffu;
                        if (ASCII) c = kwf_code_of_ascii(c);
                        const bool inside = p0 + qq < n;
                        if (!inside || c > 3u) { run = 0; win = 0; continue; }
                        win = (win << 2) | c;
                        ++run;
                        if (run < L0) continue;
                        key[q] = (win & mask0) | tag0;
                        const uint64_t h = kwf_hash(key[q]);
                        const unsigned b = (unsigned)(h >> 40) & (KWF_BITSET_BITS - 1);
                        const unsigned b2 = (unsigned)(h >> 4) & (KWF_BITSET_BITS - 1);
                        if (((bits[b >> 5] >> (b & 31)) & (bits[b2 >> 5] >> (b2 & 31))) & 1u) { live |= 1u << q; hh[q] = h; }
                    }
#pragma unroll
                    for (int q = 0; q < KWF_BATCH; ++q)
                        if (live & (1u << q)) f0[q] = a.f.fp_buckets[kwf_fp_bucket(hh[q], a.f.bucket_mask)];
#pragma unroll
                    for (int q = 0; q < KWF_BATCH; ++q)
                        if ((live & (1u << q)) && kwf_fp_maybe(f0[q], hh[q])) {
                            const uint64_t slot = (hh[q] & 0xffffffffull) & a.f.table_mask;
                            kwf_probe<ASCII>(a, key[q], slot, a.f.keys[slot], r, seq, p0 + half * KWF_BATCH + q, n, svid, scnt);
                        }
                }
            } else {
                for (int q = 0; q < 8 && p0 + q < n; ++q) {
                    unsigned c = (unsigned)(word >> (8 * q)) & 0xffu;
                    if (ASCII) c = kwf_code_of_ascii(c);
                    if (c > 3u) { run = 0; win = 0; continue; }
                    win = (win << 2) | c;
                    ++run;
                    for (int li = 0; li < a.f.n_lengths; ++li) {
                        const int L = a.f.length[li];
                        if (run < L) continue;
                        const uint64_t key = (win & a.f.mask[li]) | ((uint64_t)a.f.tag[li] << 58);
                        const uint64_t h = kwf_hash(key);
                        const unsigned b = (unsigned)(h >> 40) & (KWF_BITSET_BITS - 1);
                        const unsigned b2 = (unsigned)(h >> 4) & (KWF_BITSET_BITS - 1);
                        if (!(((bits[b >> 5] >> (b & 31)) & (bits[b2 >> 5] >> (b2 & 31))) & 1u)) continue;
                        if (!kwf_fp_maybe(a.f.fp_buckets[kwf_fp_bucket(h, a.f.bucket_mask)], h)) continue;
                        const uint64_t slot = (h & 0xffffffffull) & a.f.table_mask;
                        kwf_probe<ASCII>(a, key, slot, a.f.keys[slot], r, seq, p0 + q, n, svid, scnt);
                    }
                }
            }
          }
        }
        // wave-aggregated output: one atomic per wavefront and slot instead of one per record (a single device-wide
        // counter saturates at ~88 atomics/us, MI355X_MICROARCH "dequeue" row)
#pragma unroll
        for (int s = 0; s < KWF_SLOTS; ++s) {
            const bool need = svid[s] >= 0;
            const unsigned long long m = __ballot(need);
            if (m == 0ull) continue;
            const int lane = threadIdx.x & 63;
            const int leader = __ffsll((long long)m) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(a.n_out, (unsigned long long)__popcll(m));
            base = __shfl(base, leader, 64);
            if (need) {
                const unsigned long long pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if ((int64_t)pos < a.capacity) {
                    a.out_read[pos] = r;
                    a.out_vntr[pos] = svid[s];
                    a.out_count[pos] = scnt[s];
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Keyword sets of one length of at most 16 bases (adVNTR's 15-mers: the prefilter of every Illumina run).
// The general kernel above spends ~105 vector and ~50 scalar instructions per read position (profiles/r02_filter_pmc.json):
// a 64-bit window kept base by base through per-base branches, a 64-bit hash with three quarter-rate multiplies, two LDS
// reads, a 16-byte fingerprint bucket compared in registers.  For keywords that fit 32 bits all of it gets cheaper:
//   * a 64-byte sector of the read becomes four words of 16 two-bit bases and a 64-bit "good base" mask with word-parallel
//     arithmetic; the window ending at a base is ONE funnel shift (v_alignbit) of two neighbouring words and a mask, and which
//     windows exist at all (no symbol outside ACGT inside, none before the read's start or past its end) is a handful of
//     shift-and-and steps per 16 bases -- no per-base branch, no run counter;
//   * the first-level filter is a BLOCKED Bloom filter (both bits of a keyword in one 32-bit LDS word: one read) on a hash
//     made of 24-bit multiplies; 21 % of the windows pass it with 317 k keywords in 1 Mbit (a 5 % filter would need two
//     compute units' LDS: splitting the keywords over groups of workgroups was built and measured -- every group still has to
//     walk every read, the wavefront-level work doubles and the kernel got slower, 2.8 against 1.5 ms);
//   * the second level is the same test on a 16-bits-per-keyword filter in L2, four windows' loads in flight together: one
//     4-byte gather and a mask compare instead of a 16-byte bucket and eight fingerprint compares; 0.2 % of all windows go on
//     to the exact table.
// Everything behind that (exact table, tallies, wave-aggregated output) is the general kernel's.
// Round 4: the levels behind the first are WAVE-COOPERATIVE.  A lane's survivors of the LDS filter used to be looked up by
// that lane, four at a time, until the lane with the most survivors was through (3.4 of 16 windows survive on average, the
// busiest of 64 lanes has 8: the wavefront executed the second level for 512 slots to serve 220), and a read cut from a locus
// -- two dozen true hits -- kept its wavefront's other 63 lanes waiting while it walked the exact table: the levels behind
// the first were 60 % of the kernel's vector instructions (SQ_INSTS_VALU 68 per position-wave, 27 of them first level and
// decoding).  Now the survivors of a 16-base word of all 64 reads go into a queue of the wavefront in LDS (an exclusive prefix
// sum of the lanes' survivor counts by ballots and mbcnt: no LDS, no DPP chain), and the wavefront drains it 64 entries at a
// time: one L2-filter gather per lane and round, the exact table for the few that pass, a hit written out as a (read, VNTR, 1)
// record under one atomic per wavefront and round -- the host sums the records of a (read, VNTR) pair as it always did.
// All loops are wave-uniform (bounds from the longest read of the 64), so that the cooperative steps run converged.
// WIDE: keywords of 15 or 16 bases -- the window needs no mask before it is hashed.
#define KWF_QCAP 384                         // queue entries per wavefront (a word of 64 reads has 220 survivors on average, 1 024
                                             // at most: a fuller queue is drained in passes)
#define KWF_SHORT_LDS_BYTES (KWF_BITSET_BITS / 8 + (KWF_BLOCK / 64) * KWF_QCAP * 5)
// (dynamic LDS of keyword_filter_short_kernel: the first-level filter + the wavefronts' survivor queues -- KWF_QCAP entries of 5
// bytes each -- must fit the 160 KiB of a compute unit, or the launch fails at run time)
static_assert(KWF_SHORT_LDS_BYTES <= 160 * 1024, "keyword_filter_short_kernel: first-level filter + queues exceed the LDS of a CU");
template <bool ASCII, bool WIDE>
__global__ void __launch_bounds__(KWF_BLOCK) keyword_filter_short_kernel(KwfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t bits[];
    {
        const uint4 *src = (const uint4 *)a.f.short_bloom;
        uint4 *dst = (uint4 *)bits;
        for (int i = threadIdx.x; i < (int)(KWF_BLOOM_WORDS / 4); i += KWF_BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *qwin = bits + KWF_BLOOM_WORDS + wave * KWF_QCAP;
    uint8_t *qsrc = (uint8_t *)(bits + KWF_BLOOM_WORDS + (KWF_BLOCK / 64) * KWF_QCAP) + wave * KWF_QCAP;
    const int L0 = a.f.length[0];
    const uint32_t mask0 = L0 >= 16 ? 0xffffffffu : ((1u << (2 * L0)) - 1u);
    const uint32_t l2_mask = a.f.short_l2_mask;
    const int l2_bits = 32 - __builtin_clz(l2_mask);                 // bits of the word index (l2_mask = words - 1 >= 32767)
    for (int r0 = blockIdx.x * KWF_BLOCK + wave * 64; r0 < a.n_reads; r0 += gridDim.x * KWF_BLOCK) {
        const int r = r0 + lane;
        const bool have = r < a.n_reads;
        const uint8_t *seq = a.bases + (have ? a.span_start[r] : 0);
        const int n = have ? (int)(a.span_end[r] - a.span_start[r]) : 0;
        int nmax = n;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
        nmax = __builtin_amdgcn_readfirstlane(nmax);
        int svid[KWF_SLOTS], scnt[KWF_SLOTS];
#pragma unroll
        for (int s = 0; s < KWF_SLOTS; ++s) { svid[s] = -1; scnt[s] = 0; }
        uint32_t prevP = 0, prevG = 0;           // the 16 bases before the current word, and which of them were good
        for (int pb = 0; pb < nmax; pb += 64) {  // one 64-byte sector of every read per outer step
            uint32_t d[16];
            if (pb + 64 <= n) {
                __builtin_memcpy(d, seq + pb, 64);
            } else {
#pragma unroll
                for (int w = 0; w < 16; ++w) {
                    d[w] = 0;
                    if (pb + 4 * w + 4 <= n) __builtin_memcpy(&d[w], seq + pb + 4 * w, 4);
                    else for (int q = 0; pb + 4 * w + q < n; ++q) d[w] |= (uint32_t)seq[pb + 4 * w + q] << (8 * q);
                }
            }
            const int left = n - pb;                                  // bases of this lane's read in this sector and beyond (<= 0: none)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (16 * j >= nmax - pb) break;                       // (wave-uniform)
                // 16 bases -> P (two bits per base, first base in the top bits) and G (one bit per base: a base A, C, G, T)
                uint32_t P = 0, G = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t w = d[4 * j + i];
                    uint32_t code, wrong;                             // per byte: the base code; non-zero where the byte is no base
                    if (ASCII) {
                        const uint32_t x = (w >> 1) & 0x03030303u;
                        code = x ^ ((x >> 1) & 0x01010101u);          // A 0, C 1, G 2, T 3 (kwf_code_of_ascii)
                        const uint32_t lo = code & 0x01010101u, hi = (code >> 1) & 0x01010101u, hl = hi & lo;
                        // the upper-case letter that has this code: 'A' + {0, 2, 6, 19}; anything else in the text is no base
                        const uint32_t want = 0x41414141u + (lo << 1) + (hi << 2) + (hi << 1) + (hl << 3) + (hl << 1) + hl;
                        wrong = w ^ want;
                    } else {
                        code = w & 0x03030303u;
                        wrong = w & 0xfcfcfcfcu;
                    }
                    // 0x80 in every byte of `wrong` that is zero (exact per byte)
                    const uint32_t good = ~(((wrong & 0x7f7f7f7fu) + 0x7f7f7f7fu) | wrong | 0x7f7f7f7fu);
                    const uint32_t p8 = ((code << 6) | (code >> 4) | (code >> 14) | (code >> 24)) & 0xffu;
                    const uint32_t g4 = ((good >> 4) | (good >> 13) | (good >> 22) | (good >> 31)) & 0xfu;
                    P = (P << 8) | p8;
                    G = (G << 4) | g4;
                }
                const int here = left - 16 * j;                       // bases of the read from this word on
                if (here <= 0) G = 0;                                 // past the read's end: no base
                else if (here < 16) G &= 0xffffu << (16 - here);
                // windows: bit b of `ok` <-> the window that ends at base 15 - b exists (L0 good bases, in the read)
                const uint32_t c0 = (prevG << 16) | (G & 0xffffu);
                const uint32_t c1 = c0 & (c0 >> 1), c2 = c1 & (c1 >> 2), c3 = c2 & (c2 >> 4);
                uint32_t ok = 0xffffu;
                int done = 0;
                if (L0 & 16) { ok &= c3 & (c3 >> 8); done = 16; }
                if (L0 & 8) { ok &= c3 >> done; done += 8; }
                if (L0 & 4) { ok &= c2 >> done; done += 4; }
                if (L0 & 2) { ok &= c1 >> done; done += 2; }
                if (L0 & 1) ok &= c0 >> done;
                ok &= 0xffffu;
                if (__ballot(ok != 0u) != 0ull) {
                    // first level, the 16 windows of the word, no branch: bit e of `live` <-> the window that ends at base e
                    // passed the LDS filter
                    uint32_t live = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        // (no mask: the hash takes bits 0 .. 29 of the window, and a keyword of 16 bases is hashed without its
                        // first base on the host as well)
                        const uint32_t win = __builtin_amdgcn_alignbit(prevP, P, 2 * (15 - e));
                        const uint32_t y = kwf_h24(WIDE ? win : (win & mask0));
                        const uint32_t m = kwf_bloom_mask1(y);
                        live |= (uint32_t)((bits[y & (KWF_BLOOM_WORDS - 1)] & m) == m) << e;
                    }
                    live &= __builtin_bitreverse32(ok) >> 16;
                    // where this lane's survivors go in the wavefront's queue: exclusive prefix sum of the lanes' counts, bit by
                    // bit of the count (a count is at most 16: five ballots, each lane counts the set bits below it)
                    const int cnt = __popc(live);
                    int before = 0, total = 0;
#pragma unroll
                    for (int b = 0; b < 5; ++b) {
                        const unsigned long long m = __ballot((cnt >> b) & 1);
                        before += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) << b;
                        total += __popcll(m) << b;
                    }
                    for (int q0 = 0; q0 < total; q0 += KWF_QCAP) {            // (one pass unless more than KWF_QCAP windows survived)
                        {
                            uint32_t lv = live;
                            int pos = before - q0;
                            while (lv != 0u) {
                                const int e = __builtin_ctz(lv);
                                lv &= lv - 1u;
                                if (pos >= 0 && pos < KWF_QCAP) {
                                    qwin[pos] = __builtin_amdgcn_alignbit(prevP, P, 2 * (15 - e)) & mask0;
                                    qsrc[pos] = (uint8_t)lane;
                                }
                                ++pos;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const int queued = min(KWF_QCAP, total - q0);
                        // four rounds of 64 entries at a time: their L2-filter gathers are in flight together (a round is a chain of
                        // dependent memory accesses -- queue, gather, test --, and a wavefront has three others beside it to hide it)
                        for (int i0 = 0; i0 < queued; i0 += 4 * 64) {
                            uint32_t kk[4], zz[4], w2[4];
                            int src[4];
                            bool act[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int i = i0 + 64 * u + lane;
                                act[u] = i < queued;
                                kk[u] = act[u] ? qwin[i] : 0u;
                                src[u] = act[u] ? (int)qsrc[i] : lane;
                                zz[u] = kwf_h24b(kwf_h24(kk[u]));
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) w2[u] = act[u] ? a.f.short_l2[zz[u] & l2_mask] : 0u;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (i0 + 64 * u >= queued) break;                              // (wave-uniform)
                                const uint32_t m2 = kwf_bloom_mask2(zz[u], l2_bits);
                                const bool cand = act[u] && (w2[u] & m2) == m2;
                                if (__ballot(cand) == 0ull) continue;
                                // the exact table for what is left (0.2 % of all windows, and the true hits): one 8-byte entry per
                                // look-up, the VNTR in the entry itself when the keyword has one owner
                                uint32_t val = KWF_SHORT_EMPTY;
                                if (cand) {
                                    uint32_t at = kwf_short_slot(zz[u]) & a.f.short_table_mask;
                                    uint2 en = a.f.short_table[at];
                                    for (uint32_t probes = 0; probes <= a.f.short_table_mask && en.y != KWF_SHORT_EMPTY; ++probes) {
                                        if (en.x == kk[u]) { val = en.y; break; }
                                        at = (at + 1u) & a.f.short_table_mask;
                                        en = a.f.short_table[at];
                                    }
                                }
                                // a hit goes to the lane that owns the read (its four (VNTR, count) slots, flushed once per read):
                                // the wavefront walks the lanes that found one -- one or two a round, two dozen over a read
                                // that comes from a locus
                                unsigned long long mh = __ballot(val != KWF_SHORT_EMPTY);
                                while (mh != 0ull) {
                                    const int lh = __builtin_amdgcn_readfirstlane(__ffsll((long long)mh) - 1);
                                    mh &= mh - 1ull;
                                    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)val, lh);
                                    const int owner = __builtin_amdgcn_readlane(src[u], lh);
                                    if (v & KWF_SHORT_ONE) {
                                        if (lane == owner) kwf_tally_one(a, (int)(v & 0xffffffu), r, svid, scnt);
                                    } else {                                            // a keyword string that several VNTRs share
                                        const int first = (int)(v & 0xffffffu), cntv = (int)(v >> 24);
                                        for (int q = 0; q < cntv; ++q) {
                                            const int vid = a.f.ids[first + q];
                                            if (lane == owner) kwf_tally_one(a, vid, r, svid, scnt);
                                        }
                                    }
                                }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                prevP = P;
                prevG = G & 0xffffu;
            }
        }
        // wave-aggregated output: one atomic per wavefront and slot (a single device-wide counter saturates at ~88 atomics/us)
#pragma unroll
        for (int s = 0; s < KWF_SLOTS; ++s) {
            const bool need = svid[s] >= 0;
            const unsigned long long m = __ballot(need);
            if (m == 0ull) continue;
            const int leader = __ffsll((long long)m) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(a.n_out, (unsigned long long)__popcll(m));
            base = __shfl(base, leader, 64);
            if (need) {
                const unsigned long long pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if ((int64_t)pos < a.capacity) {
                    a.out_read[pos] = r;
                    a.out_vntr[pos] = svid[s];
                    a.out_count[pos] = scnt[s];
                }
            }
        }
    }
}
