/* TEST INFRASTRUCTURE -- CPU restatement of the local alignment the reference asks biopython for when it looks for the
 * flanking regions of a VNTR in a long read:  pairwise2.align.localms(read, flank, 1, -1, -1, -1)
 * (/root/reference/advntr/vntr_finder.py:328,345), of which it uses alignments[0][2] (score) and alignments[0][3] (begin).
 *
 * PARITY UNPINNED: biopython (a third-party dependency of the reference; any 1.7x release provides Bio.pairwise2) is
 * neither vendored in /root/reference nor installed in this image, so this file restates the published algorithm and
 * could not be checked against the library:
 *   - Smith-Waterman with match +1, mismatch -1 and gap open = extend = -1 (a gap of length g costs g);
 *   - pairwise2 collects the cells holding the best score in row-major order and recovers alignments starting from the
 *     LAST of them; walking back it tries, at each cell, a horizontal step (gap in sequence A, the read) first, then a
 *     diagonal one, then a vertical one, and takes the first that is consistent with the score matrix;
 *   - the walk ends at the first cell whose score is <= 0 and `begin` = max(row, col) of that cell.
 * Full (n+1) x (m+1) matrix and an explicit walk -- deliberately not the formulation of the GPU kernel, which carries
 * the start coordinates forward instead (csrc/flank_align.h).  Symbols other than A,C,G,T (codes > 3) match nothing. */
#include <stdint.h>
#include <stdlib.h>

void oracle_flank_align(const uint8_t *read, int n, const uint8_t *flank, int m, int *out_score, int *out_begin, int *out_end)
{
    const size_t W = (size_t)m + 1;
    int *H = (int *)calloc((size_t)(n + 1) * W, sizeof(int));
    int best = 0, br = -1, bc = -1;
    for (int r = 1; r <= n; ++r) {
        for (int c = 1; c <= m; ++c) {
            const int match = (read[r - 1] == flank[c - 1] && read[r - 1] < 4) ? 1 : -1;
            int h = H[(size_t)(r - 1) * W + (c - 1)] + match;
            const int up = H[(size_t)(r - 1) * W + c] - 1, left = H[(size_t)r * W + (c - 1)] - 1;
            if (up > h) h = up;
            if (left > h) h = left;
            if (h < 0) h = 0;
            H[(size_t)r * W + c] = h;
            if (h > 0 && h >= best) { best = h; br = r; bc = c; }          /* row-major: the last best cell wins */
        }
    }
    *out_score = best;
    *out_begin = -1;
    *out_end = best > 0 ? br - 1 : -1;
    if (best > 0) {
        int r = br, c = bc;
        for (;;) {
            const int h = H[(size_t)r * W + c];
            if (h <= 0) break;
            const int match = (read[r - 1] == flank[c - 1] && read[r - 1] < 4) ? 1 : -1;
            if (H[(size_t)r * W + (c - 1)] - 1 == h) c -= 1;                                   /* gap in the read */
            else if (H[(size_t)(r - 1) * W + (c - 1)] + match == h) { r -= 1; c -= 1; }
            else r -= 1;                                                                        /* gap in the flank */
        }
        *out_begin = r > c ? r : c;
    }
    free(H);
}
