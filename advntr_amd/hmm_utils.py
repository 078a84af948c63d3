"""Read-matcher model construction and Viterbi-path summaries -- host-side mirror of the functions of
/root/reference/advntr/hmm_utils.py that sit on the scoring path (same names, arguments and results):

  get_prefix_matcher_hmm / get_suffix_matcher_hmm              hmm_utils.py:290-353 / 357-420
  get_constant_number_of_repeats_matcher_hmm                   hmm_utils.py:424-497
  get_variable_number_of_repeats_matcher_hmm                   hmm_utils.py:501-549
  get_read_matcher_model                                       hmm_utils.py:553-595
  get_number_of_repeats_in_vpath ... get_right_flanking_region_size_in_vpath   hmm_utils.py:155-286
  extract_repeating_segments_from_read / get_multiple_alignment_of_viterbi_paths   hmm_utils.py:23-103
  get_repeating_pattern_lengths / get_repeat_segments_from_visited_states_and_region   hmm_utils.py:129-152

Models are assembled through advntr_amd.pomegranate (the engine-backed mirror of the vendored
pomegranate) with transitions inserted in the reference's order, because that order becomes the in-edge
order of the baked CSR and therefore the Viterbi tie-break.  The two flank blocks share one builder; the
only differences between them (entry into every match state for the left flank, the 0.01 early exit for
the right flank) are parameters.  The path summaries accept the reference's vpath lists but evaluate on
the ADVNTR_SC_* class words with numpy prefix sums -- the same formulation the device summariser uses
(csrc/path_summary.h); inside `viterbi_batch` they are produced on the GPU and these host versions are
only for callers that hold a vpath.
"""
import numpy as np

from . import _lib, settings
from .pomegranate import DiscreteDistribution, State, state_class_from_name
from .pomegranate import HiddenMarkovModel as Model
from .profile_hmm import build_profile_hmm_for_repeats, build_profile_hmm_pseudocounts_for_alignment

_BASE = {c: i for i, c in enumerate("ACGT")}


# ------------------------------------------------------------------------------------------------
# flank blocks
# ------------------------------------------------------------------------------------------------
def _flank_matcher(pattern, hmm_name, model_name, enter_anywhere, early_exit):
    model = Model(name=model_name)
    F = len(pattern)
    uniform = DiscreteDistribution({'A': 0.25, 'C': 0.25, 'G': 0.25, 'T': 0.25})
    ins = [State(uniform, name='I%s_%s' % (i, hmm_name)) for i in range(F + 1)]
    mat = []
    for i in range(F):
        dist = dict({'A': 0.01, 'C': 0.01, 'G': 0.01, 'T': 0.01})
        dist[pattern[i]] = 0.97
        mat.append(State(DiscreteDistribution(dist), name='M%s_%s' % (str(i + 1), hmm_name)))
    dele = [State(None, name='D%s_%s' % (str(i + 1), hmm_name)) for i in range(F)]
    unit_start = State(None, name='%s_start_%s' % (hmm_name, hmm_name))
    unit_end = State(None, name='%s_end_%s' % (hmm_name, hmm_name))
    model.add_states(ins + mat + dele + [unit_start, unit_end])
    last = F - 1
    add = model.add_transition

    add(model.start, unit_start, 1)
    add(unit_end, model.end, 1)
    insert_error = settings.MAX_ERROR_RATE * 2 / 5
    delete_error = settings.MAX_ERROR_RATE * 1 / 5
    stay = 1 - insert_error - delete_error
    if enter_anywhere:                       # left flank: a read may start anywhere inside it
        add(unit_start, dele[0], delete_error)
        add(unit_start, ins[0], insert_error)
        for i in range(F):
            add(unit_start, mat[i], (1 - insert_error - delete_error) / F)
    else:
        add(unit_start, mat[0], stay)
        add(unit_start, dele[0], delete_error)
        add(unit_start, ins[0], insert_error)
    add(ins[0], ins[0], insert_error)
    add(ins[0], dele[0], delete_error)
    add(ins[0], mat[0], stay)
    add(dele[last], unit_end, 1 - insert_error)
    add(dele[last], ins[last + 1], insert_error)
    add(mat[last], unit_end, 1 - insert_error)
    add(mat[last], ins[last + 1], insert_error)
    add(ins[last + 1], ins[last + 1], insert_error)
    add(ins[last + 1], unit_end, 1 - insert_error)
    for i in range(F):
        add(mat[i], ins[i + 1], insert_error)
        add(dele[i], ins[i + 1], insert_error)
        add(ins[i + 1], ins[i + 1], insert_error)
        if i < F - 1:
            add(ins[i + 1], mat[i + 1], stay)
            add(ins[i + 1], dele[i + 1], delete_error)
            if early_exit:                   # right flank: a read may end anywhere inside it
                add(mat[i], mat[i + 1], 1 - insert_error - delete_error - 0.01)
                add(mat[i], dele[i + 1], delete_error)
                add(mat[i], unit_end, 0.01)
            else:
                add(mat[i], mat[i + 1], stay)
                add(mat[i], dele[i + 1], delete_error)
            add(dele[i], dele[i + 1], delete_error)
            add(dele[i], mat[i + 1], stay)
    model.bake(merge=None)
    return model


def get_prefix_matcher_hmm(pattern):
    return _flank_matcher(pattern, 'prefix', "Prefix Matcher HMM Model", enter_anywhere=False, early_exit=True)


def get_suffix_matcher_hmm(pattern):
    return _flank_matcher(pattern, 'suffix', "Suffix Matcher HMM Model", enter_anywhere=True, early_exit=False)


# ------------------------------------------------------------------------------------------------
# repeat block
# ------------------------------------------------------------------------------------------------
def get_constant_number_of_repeats_matcher_hmm(patterns, copies, vpaths=None):
    model = Model(name="Repeating Pattern Matcher HMM Model")
    if vpaths:          # re-estimation from observed paths (hmm_utils.py:427-429): the units the paths cut out of the reads,
        alignment = get_multiple_alignment_of_repeats_from_reads(vpaths)     # aligned column-wise by profile position
        transitions, emissions = build_profile_hmm_pseudocounts_for_alignment(settings.MAX_ERROR_RATE, alignment)
    else:
        transitions, emissions = build_profile_hmm_for_repeats(patterns, settings.MAX_ERROR_RATE)
    L = len([k for k in emissions.keys() if k.startswith('M')])
    add = model.add_transition
    last_end = None
    for repeat in range(copies):
        ins = [State(DiscreteDistribution(emissions['I%s' % i]), name='I%s_%s' % (i, repeat)) for i in range(L + 1)]
        mat = [State(DiscreteDistribution(emissions['M%s' % i]), name='M%s_%s' % (str(i), repeat))
               for i in range(1, L + 1)]
        dele = [State(None, name='D%s_%s' % (str(i), repeat)) for i in range(1, L + 1)]
        unit_start = State(None, name='unit_start_%s' % repeat)
        unit_end = State(None, name='unit_end_%s' % repeat)
        model.add_states(ins + mat + dele + [unit_start, unit_end])
        n = L - 1
        t = transitions
        if repeat > 0:
            add(last_end, unit_start, 1)
        else:
            add(model.start, unit_start, 1)
        if repeat == copies - 1:
            add(unit_end, model.end, 1)
        add(unit_start, mat[0], t['unit_start']['M1'])
        add(unit_start, dele[0], t['unit_start']['D1'])
        add(unit_start, ins[0], t['unit_start']['I0'])
        add(ins[0], ins[0], t['I0']['I0'])
        add(ins[0], dele[0], t['I0']['D1'])
        add(ins[0], mat[0], t['I0']['M1'])
        add(dele[n], unit_end, t['D%s' % (n + 1)]['unit_end'])
        add(dele[n], ins[n + 1], t['D%s' % (n + 1)]['I%s' % (n + 1)])
        add(mat[n], unit_end, t['M%s' % (n + 1)]['unit_end'])
        add(mat[n], ins[n + 1], t['M%s' % (n + 1)]['I%s' % (n + 1)])
        add(ins[n + 1], ins[n + 1], t['I%s' % (n + 1)]['I%s' % (n + 1)])
        add(ins[n + 1], unit_end, t['I%s' % (n + 1)]['unit_end'])
        for i in range(1, L + 1):
            add(mat[i - 1], ins[i], t['M%s' % i]['I%s' % i])
            add(dele[i - 1], ins[i], t['D%s' % i]['I%s' % i])
            add(ins[i], ins[i], t['I%s' % i]['I%s' % i])
            if i < L:
                add(ins[i], mat[i], t['I%s' % i]['M%s' % (i + 1)])
                add(ins[i], dele[i], t['I%s' % i]['D%s' % (i + 1)])
                add(mat[i - 1], mat[i], t['M%s' % i]['M%s' % (i + 1)])
                add(mat[i - 1], dele[i], t['M%s' % i]['D%s' % (i + 1)])
                add(dele[i - 1], mat[i], t['D%s' % i]['M%s' % (i + 1)])
                add(dele[i - 1], dele[i], t['D%s' % i]['D%s' % (i + 1)])
        last_end = unit_end
    model.bake(merge=None)
    return model


def _rebuild_from_matrix(model, mat, states, name):
    n = len(states)
    starts = np.zeros(n)
    starts[model.start_index] = 1.0
    ends = np.zeros(n)
    ends[model.end_index] = 1.0
    new_model = Model.from_matrix(mat, [s.distribution for s in states], starts, ends, name=name,
                                  state_names=[s.name for s in states], merge=None)
    new_model.bake(merge=None)
    return new_model


def get_variable_number_of_repeats_matcher_hmm(patterns, copies=1, vpaths=None):
    model = get_constant_number_of_repeats_matcher_hmm(patterns, copies, vpaths)
    mat = model.dense_transition_matrix()
    states = list(model.states) + [State(None, name='start_repeating_pattern_match'),
                                   State(None, name='end_repeating_pattern_match')]
    count = len(mat)
    start_rep, end_rep = count, count + 1
    mat = np.c_[mat, np.zeros(count), np.zeros(count)]
    mat = np.r_[mat, [np.zeros(count + 2)]]
    mat = np.r_[mat, [np.zeros(count + 2)]]
    unit_ends = [i for i, s in enumerate(model.states) if s.name.startswith('unit_end')]

    first_unit_start = int(np.flatnonzero(mat[model.start_index] != 0)[-1])
    mat[model.start_index][first_unit_start] = 0.0
    mat[model.start_index][start_rep] = 1
    mat[start_rep][first_unit_start] = 1
    for unit_end in unit_ends:                       # each copy may be the last one
        next_state = int(np.flatnonzero(mat[unit_end] != 0)[-1])
        mat[unit_end][next_state] = 0.5
        mat[unit_end][end_rep] = 0.5
    mat[end_rep][model.end_index] = 1
    return _rebuild_from_matrix(model, mat, states, 'Repeat Matcher HMM Model')


def get_read_matcher_model(left_flanking_region, right_flanking_region, patterns, copies=1, vpaths=None, native=True,
                           exp="numpy"):
    """hmm_utils.py:553-595.  native=True (default) builds the model in the library's C++ builder
    (csrc/model_builder.h, ~1 ms); native=False assembles it call by call through advntr_amd.pomegranate, the way the
    reference does through its pomegranate (~50 ms).  Both give the same arrays (tests/test_native_builder.py)."""
    if native:
        # with vpaths the aligned repeat units come from the paths (hmm_utils.py:427-429) instead of `patterns`
        rows = get_multiple_alignment_of_repeats_from_reads(vpaths) if vpaths else patterns
        return build_read_matcher_models([(left_flanking_region, right_flanking_region, rows, copies)],
                                         threads=1, exp=exp)[0]
    return _get_read_matcher_model_stepwise(left_flanking_region, right_flanking_region, patterns, copies, vpaths)


def build_read_matcher_models(loci, threads=0, exp="numpy", align=None):
    """Many loci at once on host threads: loci = [(left_flank, right_flank, aligned_repeat_units, copies), ...]
    -> list of baked models (advntr_build_read_matchers).  align (default settings.ALIGN_REPEATS): align repeat
    units of unequal length with the library's own aligner instead of refusing them."""
    loci = list(loci)
    align = settings.ALIGN_REPEATS if align is None else align
    built = _lib.build_read_matchers([l[0] for l in loci], [l[1] for l in loci], [list(l[2]) for l in loci],
                                     [int(l[3]) for l in loci], settings.MAX_ERROR_RATE, exp=exp, threads=threads,
                                     align=align)
    return [Model._from_built(b, 'Read Matcher') for b in built]


def _get_read_matcher_model_stepwise(left_flanking_region, right_flanking_region, patterns, copies=1, vpaths=None):
    model = get_suffix_matcher_hmm(left_flanking_region)
    repeats_matcher = get_variable_number_of_repeats_matcher_hmm(patterns, copies, vpaths)
    right_flanking_matcher = get_prefix_matcher_hmm(right_flanking_region)
    model.concatenate(repeats_matcher)
    model.concatenate(right_flanking_matcher)
    model.bake(merge=None)

    mat = model.dense_transition_matrix()
    first_repeat_matches, repeat_match_states, suffix_start = [], [], None
    for i, state in enumerate(model.states):
        if state.name[0] == 'M' and state.name.split('_')[-1] == '0':
            first_repeat_matches.append(i)
        if state.name[0] == 'M' and state.name.split('_')[-1] not in ['prefix', 'suffix']:
            repeat_match_states.append(i)
        if state.name == 'suffix_start_suffix':
            suffix_start = i
    mat[model.start_index][suffix_start] = 0.3        # reads starting in the left flank
    for idx in first_repeat_matches:                  # reads starting inside the first repeat unit
        mat[model.start_index][idx] = 0.7 / len(first_repeat_matches)
    for idx in repeat_match_states:                   # reads ending inside the repeats
        to_end = 0.7 / len(repeat_match_states)
        total = 1 + to_end
        row = mat[idx]
        nz = row != 0
        row[nz] = row[nz] / total
        mat[idx][model.end_index] = to_end / total
    new_model = _rebuild_from_matrix(model, mat, model.states, 'Read Matcher')

    # which flank base each M*_suffix / M*_prefix state stands for (used by the flank match-rate summary)
    bases = {}
    for i, ch in enumerate(left_flanking_region):
        if ch in _BASE:
            bases['M%d_suffix' % (i + 1)] = _BASE[ch]
    for i, ch in enumerate(right_flanking_region):
        if ch in _BASE:
            bases['M%d_prefix' % (i + 1)] = _BASE[ch]
    new_model.set_flank_bases(bases)
    return new_model


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
