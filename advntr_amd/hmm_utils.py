"""Read-matcher model construction and Viterbi-path summaries -- host-side mirror of the functions of
/root/reference/advntr/hmm_utils.py that sit on the scoring path (same names, arguments and results):

  get_read_matcher_model (and the sub-builders it drives)      hmm_utils.py:290-595
  get_number_of_repeats_in_vpath ... get_right_flanking_region_size_in_vpath   hmm_utils.py:155-286
  extract_repeating_segments_from_read / get_multiple_alignment_of_viterbi_paths   hmm_utils.py:23-103
  get_repeating_pattern_lengths / get_repeat_segments_from_visited_states_and_region   hmm_utils.py:129-152

Read-matcher models come from the library's native builder (csrc/model_builder.h behind
advntr_build_read_matchers); the edge insertion order of the reference's route -- it becomes the in-edge order of
the baked CSR and therefore the Viterbi tie-break -- is reproduced there.  The path summaries accept the reference's vpath lists but evaluate on
the ADVNTR_SC_* class words with numpy prefix sums -- the same formulation the device summariser uses
(csrc/path_summary.h); inside `viterbi_batch` they are produced on the GPU and these host versions are
only for callers that hold a vpath.
"""
import numpy as np

from . import _lib, settings
from .pomegranate import DiscreteDistribution, State, state_class_from_name
from .pomegranate import HiddenMarkovModel as Model


def get_read_matcher_model(left_flanking_region, right_flanking_region, patterns, copies=1, vpaths=None, exp="numpy"):
    """hmm_utils.py:553-595 (and everything it drives, :290-549), built by the library's C++ builder
    (csrc/model_builder.h, ~1 ms per locus; the reference spends 0.8-1.0 s here).  The result is the baked model the
    reference's three bake() / two from_matrix() round trips leave: same state order, same CSR in-edge order, same
    log-probabilities (tests/test_builder_golden.py)."""
    # with vpaths the aligned repeat units come from the paths (hmm_utils.py:427-429) instead of `patterns`
    rows = get_multiple_alignment_of_repeats_from_reads(vpaths) if vpaths else patterns
    return build_read_matcher_models([(left_flanking_region, right_flanking_region, rows, copies)],
                                     threads=1, exp=exp)[0]


def build_read_matcher_models(loci, threads=0, exp="numpy", align=None):
    """Many loci at once on host threads: loci = [(left_flank, right_flank, aligned_repeat_units, copies), ...]
    -> list of baked models (advntr_build_read_matchers).  align (default settings.ALIGN_REPEATS): align repeat
    units of unequal length with the library's own aligner instead of refusing them."""
    loci = list(loci)
    align = settings.ALIGN_REPEATS if align is None else align
    built = _lib.build_read_matchers([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci],
                                     [int(l[3]) for l in loci], settings.MAX_ERROR_RATE, exp=exp, threads=threads,
                                     align=align)
    return [Model._from_built(b, 'Read Matcher') for b in built]


# ------------------------------------------------------------------------------------------------
# Viterbi-path summaries (host versions over class words; the batch path computes them on the GPU)
# ------------------------------------------------------------------------------------------------
def is_match_state(state_name):
    return state_name.startswith('M')


def is_emitting_state(state_name):
    return bool(state_class_from_name(state_name) & _lib.SC_EMIT)


def _classes(vpath):
    return np.array([state_class_from_name(state.name) for _, state in vpath[1:-1]], dtype=np.int64)


def summarize_classes(cls, n_bases=None):
    """(ru, matches, repeat_bp, left_bp, right_bp) from the class words of vpath[1:-1]."""
    cls = np.asarray(cls, dtype=np.int64)
    emit = (cls & _lib.SC_EMIT) != 0
    cur_bp = np.cumsum(emit)
    read_length = int(cur_bp[-1]) if len(cls) else 0
    if n_bases is not None:
        read_length = n_bases
    is_start = ((cls & _lib.SC_UNIT_START) != 0) & (read_length - cur_bp >= 3)
    is_end = ((cls & _lib.SC_UNIT_END) != 0) & (cur_bp >= 3)
    starts, ends = int(is_start.sum()), int(is_end.sum())
    delta = 0
    if starts and ends:
        sbp, ebp = cur_bp[is_start], cur_bp[is_end]
        if ebp[0] < sbp[0] and sbp[-1] > ebp[-1]:
            delta = 1
    ru = max(starts, ends) + delta
    matches = int(((cls & _lib.SC_MATCH) != 0).sum())
    repeat_bp = int((emit & ((cls & _lib.SC_FIX) == 0)).sum())
    left_bp = int((emit & ((cls & _lib.SC_SUFFIX) != 0)).sum())
    right_bp = int((emit & ((cls & _lib.SC_PREFIX) != 0)).sum())
    return ru, matches, repeat_bp, left_bp, right_bp


def get_number_of_repeats_in_vpath(vpath):
    return summarize_classes(_classes(vpath))[0]


def get_number_of_matches_in_vpath(vpath):
    return summarize_classes(_classes(vpath))[1]


def get_number_of_repeat_bp_matches_in_vpath(vpath):
    return summarize_classes(_classes(vpath))[2]


def get_left_flanking_region_size_in_vpath(vpath):
    return summarize_classes(_classes(vpath))[3]


def get_right_flanking_region_size_in_vpath(vpath):
    return summarize_classes(_classes(vpath))[4]


def flanking_rate_from_counts(left_matches, left_bp, right_matches, right_bp, accuracy_filter=False):
    """The final division of get_flanking_regions_matching_rate (hmm_utils.py:252-268)."""
    dflt = 0.00001 if accuracy_filter else 1
    right_rate = float(right_matches) / right_bp if right_bp != 0 else dflt
    left_rate = float(left_matches) / left_bp if left_bp != 0 else dflt
    return min(right_rate, left_rate)


def get_flanking_regions_matching_rate(vpath, sequence, left_flank, right_flank, accuracy_filter=False,
                                       verbose=False):
    names = [state.name for _, state in vpath[1:-1]]
    cls = _classes(vpath)
    emit = (cls & _lib.SC_EMIT) != 0
    seq_index = np.cumsum(emit) - emit
    max_hmm_index = -1
    for i, nm in enumerate(names):
        if 'suffix_end_suffix' in nm:
            max_hmm_index = int(names[i - 1 if i else 0].split("_")[0][1:])
            break
    lm = lb = rm = rb = 0
    for i, nm in enumerate(names):
        c = cls[i]
        if c & _lib.SC_SKIP:
            continue
        h = int(nm.split("_")[0][1:])
        if c & _lib.SC_PREFIX:
            if (c & _lib.SC_MATCH) and sequence[seq_index[i]] == right_flank[h - 1]:
                rm += 1
            rb += 1 if emit[i] else 0
        if c & _lib.SC_SUFFIX:
            if (c & _lib.SC_MATCH) and sequence[seq_index[i]] == left_flank[-(max_hmm_index - h + 1)]:
                lm += 1
            lb += 1 if emit[i] else 0
    return flanking_rate_from_counts(lm, lb, rm, rb, accuracy_filter)


# ------------------------------------------------------------------------------------------------
# Repeat-unit extraction from a Viterbi path and the column-wise merge of per-unit paths
# (hmm_utils.py:23-103, 129-152) -- the two functions the reference's own tests/test_hmm_utils.py exercises
# ------------------------------------------------------------------------------------------------
def _unit_spans(visited_states):
    """(state index of unit_start, state index of the next unit_end, bases consumed before each) for every repeat
    unit that is closed by a unit_end on the path."""
    spans = []
    bases = 0
    open_at = None
    for i, name in enumerate(visited_states):
        if name.startswith('unit_end') and open_at is not None:
            spans.append((open_at[0], i, open_at[1], bases))
        if name.startswith('unit_start'):
            open_at = (i, bases)
        if is_emitting_state(name):
            bases += 1
    return spans


def extract_repeating_segments_from_read(sequence, visited_states):
    repeats, vpaths = [], []
    for s_idx, e_idx, b0, b1 in _unit_spans(visited_states):
        repeats.append(sequence[b0:b1])
        vpaths.append(list(visited_states[s_idx + 1:e_idx]))
    return repeats, vpaths


def get_repeating_pattern_lengths(visited_states):
    return [b1 - b0 for _, _, b0, b1 in _unit_spans(visited_states)]


def get_repeat_segments_from_visited_states_and_region(visited_states, region):
    segments, added = [], 0
    for length in get_repeating_pattern_lengths(visited_states):
        segments.append(region[added:added + length])
        added += length
    return segments


def get_multiple_alignment_of_viterbi_paths(repeats_sequences, repeats_visited_states):
    """Align repeat units column-wise by their profile positions: column order M0,I0,M1,I1,...; a position appears
    as many times as the unit that visits it most often; units that skip a column get '-'."""
    width = {}
    top = 0
    per_unit = []
    for states in repeats_visited_states:
        keys = [s.split('_')[0] for s in states]
        per_unit.append(keys)
        count = {}
        for k in keys:
            count[k] = count.get(k, 0) + 1
        for k, v in count.items():
            top = max(top, int(k[1:]))
            width[k] = max(width.get(k, v), v) if k in width else v
    columns = []
    for i in range(top + 1):
        for kind in ('M', 'I'):
            key = '%s%d' % (kind, i)
            columns.extend([key] * width.get(key, 0))
    rows = []
    for seq, keys in zip(repeats_sequences, per_unit):
        remaining = list(keys)
        pos, row = 0, []
        for col in columns:
            if col in remaining:
                # the reference marks EVERY occurrence of the column's state as used at once (hmm_utils.py:58-61)
                remaining = ['DELETED' if k == col else k for k in remaining]
                row.append(seq[pos])
                pos += 1
            else:
                row.append('-')
        rows.append(''.join(row))
    return rows


def get_multiple_alignment_of_repeats_from_reads(sequence_vpath_list):
    seqs, paths = [], []
    for sequence, vpath in sequence_vpath_list:
        names = [state.name for _, state in vpath[1:-1]]
        r, p = extract_repeating_segments_from_read(sequence, names)
        seqs += r
        paths += p
    return get_multiple_alignment_of_viterbi_paths(seqs, paths)


def path_to_alignment(x, y, path):
    for i, (_, state) in enumerate(path[1:-1]):
        if state.name.startswith('D'):
            y = y[:i] + '-' + y[i:]
        elif state.name.startswith('I'):
            x = x[:i] + '-' + x[i:]
    return x, y


def get_emitted_basepair_from_visited_states(state, visited_states, sequence):
    """hmm_utils.py:106-113: the read base emitted at the first occurrence of `state` on the path."""
    at = 0
    for name in visited_states:
        if name == state:
            return sequence[at]
        if is_emitting_state(name):
            at += 1
    return None


# ------------------------------------------------------------------------------------------------
# Repeat finder for a reference region (hmm_utils.py:598-680): used when a VNTR is added to the model database
# ------------------------------------------------------------------------------------------------
def build_reference_repeat_finder_hmm(patterns, copies=1):
    """`copies` profile units of the first pattern (match emits 0.97 / 0.01, fixed 0.98 / 0.01 / 0.01 transitions) between
    two random-sequence states; baked with bake()'s default merge='All' like the reference.  This model has no column
    program (its random-match states emit): it is scored by the generic-CSR kernel."""
    pattern = patterns[0]
    L = len(pattern)
    model = Model(name="HMM Model")
    uniform = DiscreteDistribution({'A': 0.25, 'C': 0.25, 'G': 0.25, 'T': 0.25})
    start_random = State(uniform, name='start_random_matches')
    end_random = State(uniform, name='end_random_matches')
    model.add_states([start_random, end_random])
    add = model.add_transition
    last_end = None
    for repeat in range(copies):
        ins = [State(uniform, name='I%s_%s' % (i, repeat)) for i in range(L + 1)]
        mat = []
        for i in range(L):
            dist = dict({'A': 0.01, 'C': 0.01, 'G': 0.01, 'T': 0.01})
            dist[pattern[i]] = 0.97
            mat.append(State(DiscreteDistribution(dist), name='M%s_%s' % (str(i + 1), repeat)))
        dele = [State(None, name='D%s_%s' % (str(i + 1), repeat)) for i in range(L)]
        unit_start = State(None, name='unit_start_%s' % repeat)
        unit_end = State(None, name='unit_end_%s' % repeat)
        model.add_states(ins + mat + dele + [unit_start, unit_end])
        last = L - 1
        if repeat > 0:
            add(last_end, unit_start, 0.5)
        else:
            add(model.start, unit_start, 0.5)
            add(model.start, start_random, 0.5)
            add(start_random, unit_start, 0.5)
            add(start_random, start_random, 0.5)
        add(unit_end, end_random, 0.5)
        if repeat == copies - 1:
            add(unit_end, model.end, 0.5)
            add(end_random, end_random, 0.5)
            add(end_random, model.end, 0.5)
        add(unit_start, mat[0], 0.98)
        add(unit_start, dele[0], 0.01)
        add(unit_start, ins[0], 0.01)
        add(ins[0], ins[0], 0.01)
        add(ins[0], dele[0], 0.01)
        add(ins[0], mat[0], 0.98)
        add(dele[last], unit_end, 0.99)
        add(dele[last], ins[last + 1], 0.01)
        add(mat[last], unit_end, 0.99)
        add(mat[last], ins[last + 1], 0.01)
        add(ins[last + 1], ins[last + 1], 0.01)
        add(ins[last + 1], unit_end, 0.99)
        for i in range(L):
            add(mat[i], ins[i + 1], 0.01)
            add(dele[i], ins[i + 1], 0.01)
            add(ins[i + 1], ins[i + 1], 0.01)
            if i < L - 1:
                add(ins[i + 1], mat[i + 1], 0.98)
                add(ins[i + 1], dele[i + 1], 0.01)
                add(mat[i], mat[i + 1], 0.98)
                add(mat[i], dele[i + 1], 0.01)
                add(dele[i], dele[i + 1], 0.01)
                add(dele[i], mat[i + 1], 0.98)
        last_end = unit_end
    model.bake()
    return model


def find_repeat_segments(pattern, estimated_repeats, region_in_ref):
    """ReferenceVNTR.find_repeat_segments (reference_vntr.py:80-87): Viterbi path of the region through the repeat finder,
    cut at the unit boundaries."""
    model = build_reference_repeat_finder_hmm([pattern], copies=estimated_repeats)
    logp, path = model.viterbi(region_in_ref)
    visited_states = [state.name for _, state in path[1:-1]]
    return get_repeat_segments_from_visited_states_and_region(visited_states, region_in_ref)
