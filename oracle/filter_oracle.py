"""CPU restatement of the reference's keyword prefilter, /root/reference/filtering/main.cc (Aho-Corasick
multi-keyword matching of reads against per-VNTR keyword sets).  TEST INFRASTRUCTURE: only tests/ and the
cpu_baseline leg of scripts/filter_bench.py may import it.

Parity status: PINNED -- tests/test_filter_oracle.py checks it against tests/golden/filter_*.json.gz, the stdout
of the reference binary itself (oracle/_ref/adVNTR-Filtering, compiled from main.cc by oracle/Makefile and run by
tests/golden/make_filter_golden.py), and, when oracle/_ref exists, against fresh runs of that binary.

What the automaton computes, restated without the automaton: the goto/failure machine of main.cc:56-157 reports,
at every read position, every keyword that ends there (main.cc:266-281); that is exact substring matching.  Any
character other than A,C,G,T maps to symbol 4 (main.cc:44-55), which no keyword contains, so it resets the match.
"""
import collections


def parse_keywords(text):
    """main.cc:176-216: one line per VNTR: id then keywords; duplicate tokens on a line collapse (std::set,
    sorted); parsing stops at the first line without tokens.  Returns (vntr_ids, words, word_vntr)."""
    vntr_ids, words, word_vntr = [], [], []
    for line in text.split("\n"):
        tokens = line.split()
        if len(tokens) < 1:
            break
        vid = _atoi(tokens[0])
        vntr_ids.append(vid)
        for tok in sorted(set(tokens[1:])):
            word_vntr.append(vid)
            words.append(tok)
    return vntr_ids, words, word_vntr


def _atoi(s):
    n, i, sign = 0, 0, 1
    if i < len(s) and s[i] in "+-":
        sign = -1 if s[i] == "-" else 1
        i += 1
    while i < len(s) and s[i].isdigit():
        n = n * 10 + ord(s[i]) - 48
        i += 1
    return sign * n


def count_matches(seq, by_len, word_vntr):
    """occurrences per vntr id in one read (main.cc:262-282): +1 for every (position, keyword index) match."""
    counts = collections.OrderedDict()
    for L, table in by_len.items():
        for i in range(L, len(seq) + 1):
            hit = table.get(seq[i - L:i])
            if hit:
                for w in hit:
                    v = word_vntr[w]
                    counts[v] = counts.get(v, 0) + 1
    return counts


def index_words(words):
    by_len = {}
    for w, s in enumerate(words):
        if not s or any(ch not in "ACGT" for ch in s):
            continue      # a keyword holding another symbol can still match in the reference only if the read holds
                          # the same symbol mapped to 4; keywords come from reference sequence, tests keep them ACGT
        by_len.setdefault(len(s), {}).setdefault(s, []).append(w)
    return by_len


def run_filter(fasta_text, keywords_text, min_matches=5, max_reads=2000):
    """Whole program: returns the stdout text of adVNTR-Filtering (main.cc:220-331)."""
    vntr_ids, words, word_vntr = parse_keywords(keywords_text)
    by_len = index_words(words)
    vntr_read_list = {}           # vid -> {name: occurrence}   (std::map: iterates by name)
    read_sequences = {}
    lines = fasta_text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    for k in range(0, len(lines) - 1, 2):
        name, seq = lines[k][1:], lines[k + 1]
        counts = count_matches(seq, by_len, word_vntr)
        for vid in sorted(counts):                                    # std::map<int,...> order
            occ = counts[vid]
            lst = vntr_read_list.setdefault(vid, {})
            if len(lst) > max_reads * 3:                               # main.cc:290
                continue
            if occ >= min_matches:
                lst[name] = occ
                read_sequences[name] = seq
    out = []
    filtered = set()
    acc = {}                      # vntr_filtered_reads: persists across duplicate ids (main.cc:304-312)
    for vid in vntr_ids:
        vec = acc.setdefault(vid, [])
        for name in sorted(vntr_read_list.get(vid, {})):
            vec.append((vntr_read_list[vid][name], name))
        result_size = min(len(vec), max_reads)
        line = "%d %d" % (vid, result_size)
        if vec:
            vec.sort(reverse=True)                                     # sort(rbegin, rend): descending pairs
            for j, (occ, name) in enumerate(vec):
                filtered.add(name)
                line += " " + name
                if j >= max_reads:                                     # main.cc:321 (prints max_reads + 1 names)
                    break
        out.append(line)
    for name in sorted(filtered):
        out.append("%s %s" % (name, read_sequences[name]))
    return "\n".join(out) + "\n"


def parse_output(text):
    """The consumer's view (genome_analyzer.py:187-197): vid -> set(read names), and the (name, seq) list."""
    vntr_read_ids, reads = {}, []
    for line in text.split("\n"):
        parts = line.split()
        if len(parts) < 2:
            continue
        if parts[0].isdigit() and parts[1].isdigit():
            vntr_read_ids[int(parts[0])] = set(parts[2:])
        else:
            reads.append((parts[0], parts[1]))
    return vntr_read_ids, reads
