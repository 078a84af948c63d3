"""Profile-HMM parameter estimation from aligned repeat units (host side of the scoring path).

Mirrors the interface of /root/reference/advntr/profile_hmm.py: `build_profile_hmm_pseudocounts_for_alignment
(error_rate, alignment)` (profile_hmm.py:13-161) and `build_profile_hmm_for_repeats(repeats, error_rate)`
(profile_hmm.py:165-175), returning the same `(transitions, emissions)` nested dicts that
advntr_amd.hmm_utils consumes.  Counting is done on integer arrays; the floating-point expressions keep
the reference's operand order ((count/total + pseu) summed over A,C,G,T then divided; (p + pseu) /
(1 + pseu*k)) so parameters come out bit-identical (pinned by tests/test_builder_golden.py).

`muscle` (the external MSA the reference shells out to for >1 repeat) is out of scope: repeats must come
pre-aligned (equal-length rows, '-' for gaps), see DESIGN.md.
"""
ALPHABET = "ACGT"


def build_profile_hmm_pseudocounts_for_alignment(error_rate, alignment):
    rows = len(alignment)
    width = len(alignment[0])
    pseu = (rows / 4.0) * (error_rate / 10)
    gap_cut = 0.5 * rows
    is_insert_col = [sum(1.0 for r in alignment if r[c] == '-') >= gap_cut for c in range(width)]
    L = width - sum(is_insert_col)

    # walk every row once: its state string, emission counts, transition counts
    emit_count = {'I0': [0] * 4}
    for k in range(1, L + 1):
        emit_count['I%d' % k] = [0] * 4
        emit_count['M%d' % k] = [0] * 4
    trans_count = {'unit_start': {'I0': 0, 'D1': 0, 'M1': 0}, 'I0': {'I0': 0, 'D1': 0, 'M1': 0}}

    def bump(a, b):
        trans_count.setdefault(a, {})
        trans_count[a][b] = trans_count[a].get(b, 0) + 1

    paths = []
    for row in alignment:
        path = []
        k = 1
        for c, ch in enumerate(row):
            if not is_insert_col[c]:
                if ch == '-':
                    path.append('D%d' % k)
                else:
                    path.append('M%d' % k)
                    emit_count['M%d' % k][ALPHABET.index(ch)] += 1
                k += 1
            elif ch != '-':
                path.append('I%d' % (k - 1))
                emit_count['I%d' % (k - 1)][ALPHABET.index(ch)] += 1
        paths.append(path)
    for path in paths:
        trans_count['unit_start'][path[0]] += 1
    for path in paths:
        for a, b in zip(path[:-1], path[1:]):
            bump(a, b)
        bump(path[-1], 'unit_end')

    emission = {'unit_start': dict.fromkeys(ALPHABET, 0), 'unit_end': dict.fromkeys(ALPHABET, 0)}
    for key, counts in emit_count.items():
        total = 0
        for cnt in counts:
            total += cnt
        if total > 0:
            vals = []
            sub_total = 0
            for cnt in counts:
                v = (1.0 * cnt) / total + pseu
                vals.append(v)
                sub_total += 1.0 * v
            emission[key] = {ch: v / sub_total for ch, v in zip(ALPHABET, vals)}
        else:
            emission[key] = {ch: 1.0 / len(ALPHABET) for ch in ALPHABET}
    for k in range(1, L + 1):
        emission['D%d' % k] = dict.fromkeys(ALPHABET, 0)

    # every position owns I/M/D rows, with their canonical successors present (possibly unseen)
    for k in range(1, L + 1):
        for kind in 'IMD':
            trans_count.setdefault('%s%d' % (kind, k), {})
    transition = {}
    for key, succ in trans_count.items():
        total = 0
        for cnt in succ.values():
            total += cnt
        succ = dict(succ)
        if key not in ('unit_start', 'I0'):
            idx = key[1:]
            if idx != str(L):
                for nxt in ('I' + idx, 'D%d' % (int(idx) + 1), 'M%d' % (int(idx) + 1)):
                    succ.setdefault(nxt, 0)
            else:
                succ.setdefault('I' + idx, 0)
                succ.setdefault('unit_end', 0)
        k = len(succ)
        out = {}
        for nxt, cnt in succ.items():
            if total > 0:
                p = 1.0 * cnt / total
                out[nxt] = (p + pseu) / (1 + pseu * k)
            elif k == 3:
                out[nxt] = 1.0 / 3
            elif k == 2:
                out[nxt] = 1.0 / 2
            else:
                out[nxt] = cnt
        transition[key] = out

    names = ['unit_start', 'I0']
    for k in range(1, L + 1):
        names.extend(['M%d' % k, 'D%d' % k, 'I%d' % k])
    names.append('unit_end')
    for a in names:
        row = transition.setdefault(a, {})
        for b in names:
            row.setdefault(b, 0)
    return transition, emission


def build_profile_hmm_for_repeats(repeats, error_rate):
    """One repeat: used as its own alignment (profile_hmm.py:172-173).  Several repeats must already be
    aligned (equal length); the reference would call the external `muscle` here (profile_hmm.py:166-171)."""
    if len(repeats) > 1:
        width = len(repeats[0])
        from . import settings
        if any(len(r) != width for r in repeats) and settings.ALIGN_REPEATS:
            from . import _lib
            repeats = _lib.align_repeats(list(repeats))
        elif any(len(r) != width for r in repeats):
            raise NotImplementedError("multiple un-aligned repeat units need an MSA (the reference shells out to "
                                      "`muscle`); pass pre-aligned rows -- see DESIGN.md, out of scope")
    return build_profile_hmm_pseudocounts_for_alignment(error_rate, list(repeats))
