"""Host-side mirror of the pomegranate surface adVNTR uses, backed by the HIP engine.

Drop-in for the names `advntr/hmm_utils.py` and `advntr/vntr_finder.py` import from the vendored
pomegranate 0.6.1 (`HiddenMarkovModel`, `State`, `DiscreteDistribution`): same constructor arguments,
method names, return shapes and error behaviour for the model-construction and scoring calls on
adVNTR's path.  Reference lines (relative to /root/reference/pomegranate):

  State                      base.pyx:362-458
  DiscreteDistribution       distributions.pyx:1256-1545   (only dict construction / keys / log_probability)
  HiddenMarkovModel.__init__ hmm.pyx:191-232      add_state(s) hmm.pyx:348-391
  add_transition             hmm.pyx:392-434      concatenate  hmm.pyx:584-615
  dense_transition_matrix    hmm.pyx:492-514      bake         hmm.pyx:673-1123 (merge=None semantics)
  from_matrix                hmm.pyx:3146-3238    viterbi      hmm.pyx:1911-1967
  log_probability            hmm.pyx:1258-1298

What is NOT here: everything adVNTR never calls (fit, backward, sampling, plotting, other
distributions) -- see DESIGN.md "out of scope".  bake() supports merge=None (the read matchers) and the
default merge="All" / "Partial" (hmm.pyx:725-823; the repeat finder of hmm_utils.py:598-680 is baked that way).

The graph container is a plain insertion-ordered adjacency map.  bake() derives the same state
numbering and in-edge order as the reference does on networkx 1.11 under Python >= 3.7 (emitting states
sorted by name, silent states name-sorted then DFS-topologically sorted, in-edges in edge-insertion
order), because that order decides Viterbi ties.  Scoring itself never happens on the host: viterbi()
and log_probability() call the C ABI of libadvntr_hip.so and fail loudly without it.
"""
import math
from operator import attrgetter

import numpy as np

from . import _lib

NEGINF = float("-inf")


def _log(x):
    """utils.pyx:64-70: C log, -inf for 0."""
    return math.log(x) if x > 0 else NEGINF


class DiscreteDistribution(object):
    """Symbol -> probability table (distributions.pyx:1256-1545)."""

    def __init__(self, characters, frozen=False):
        self.name = "DiscreteDistribution"
        self.frozen = frozen
        self.d = 1
        self.dist = dict(characters)
        self.log_dist = {key: _log(value) for key, value in self.dist.items()}
        self.parameters = [self.dist]

    def keys(self):
        return tuple(self.dist.keys())

    def items(self):
        return tuple(self.dist.items())

    def log_probability(self, symbol):
        return self.log_dist.get(symbol, NEGINF)

    def probability(self, symbol):
        return self.dist.get(symbol, 0.0)

    def copy(self):
        return DiscreteDistribution(dict(self.dist), self.frozen)


class State(object):
    """base.pyx:362-458: distribution (None => silent), name, weight (default 1)."""

    def __init__(self, distribution, name=None, weight=None):
        import uuid
        self.distribution = distribution
        self.name = name or str(uuid.uuid4())
        self.weight = weight or 1.

    def is_silent(self):
        return self.distribution is None

    def tie(self, state):
        state.distribution = self.distribution

    def copy(self):
        return State(self.distribution.copy() if self.distribution is not None else None, self.name)

    def __repr__(self):
        return "State(%r)" % (self.name,)


class _Graph(object):
    """Insertion-ordered directed graph: exactly the operations bake()/concatenate() need."""

    def __init__(self):
        self.succ = {}

    def add_node(self, n):
        if n not in self.succ:
            self.succ[n] = {}

    def add_edge(self, a, b, logp):
        self.add_node(a)
        self.add_node(b)
        self.succ[a][b] = logp          # re-adding keeps the edge's first position (dict update)

    def nodes(self):
        return list(self.succ)

    def remove_node(self, n):
        del self.succ[n]
        for nbrs in self.succ.values():
            nbrs.pop(n, None)

    def remove_edge(self, a, b):
        del self.succ[a][b]

    def edges(self):
        for a, nbrs in self.succ.items():
            for b, lp in nbrs.items():
                yield a, b, lp

    def n_edges(self):
        return sum(len(v) for v in self.succ.values())

    @staticmethod
    def union(g, h):
        if set(g.succ) & set(h.succ):
            raise ValueError("The node sets of G and H are not disjoint.")
        r = _Graph()
        for src in (g, h):
            for n in src.succ:
                r.add_node(n)
            for a, b, lp in src.edges():
                r.add_edge(a, b, lp)
        return r


def _topological_sort(succ_of, nbunch):
    """Iterative DFS, reversed post-order, successors stacked in adjacency order -- the algorithm of
    networkx 1.11's topological_sort(G, nbunch), which bake() applies to the silent sub-graph
    (hmm.pyx:870-874)."""
    seen, explored, order = set(), set(), []
    for v in nbunch:
        if v in explored:
            continue
        fringe = [v]
        while fringe:
            w = fringe[-1]
            if w in explored:
                fringe.pop()
                continue
            seen.add(w)
            new_nodes = []
            for n in succ_of[w]:
                if n not in explored:
                    if n in seen:
                        raise ValueError("Graph contains a cycle.")
                    new_nodes.append(n)
            if new_nodes:
                fringe.extend(new_nodes)
            else:
                explored.add(w)
                order.append(w)
                fringe.pop()
    order.reverse()
    return order


def state_class_from_name(name, expected_base=None):
    """ADVNTR_SC_* bits from a state name, following the string tests of advntr/hmm_utils.py:116-286."""
    c = 0
    if name.startswith("M") or name.startswith("I") or name.startswith("start_random_matches") \
            or name.startswith("end_random_matches"):
        c |= _lib.SC_EMIT
    if name.startswith("M"):
        c |= _lib.SC_MATCH
    if name.endswith("suffix"):
        c |= _lib.SC_SUFFIX
    if name.endswith("prefix"):
        c |= _lib.SC_PREFIX
    if name.startswith("unit_start"):
        c |= _lib.SC_UNIT_START
    if name.startswith("unit_end"):
        c |= _lib.SC_UNIT_END
    if "start" in name or "end" in name:
        c |= _lib.SC_SKIP
    if name.endswith("fix"):
        c |= _lib.SC_FIX
    if expected_base is not None:
        c |= _lib.SC_BASE_VALID | (int(expected_base) << _lib.SC_BASE_SHIFT)
    return c


class HiddenMarkovModel(object):
    def __init__(self, name=None, start=None, end=None):
        self.name = str(name) or str(id(self))
        self.model = "HiddenMarkovModel"
        self.graph = _Graph()
        self.start = start or State(None, name=self.name + "-start")
        self.end = end or State(None, name=self.name + "-end")
        self.d = 0
        self.n_edges = 0
        self.n_states = 0
        self.graph.add_node(self.start)
        self.graph.add_node(self.end)
        self.states = []
        self.start_index = self.end_index = self.silent_start = -1
        self.keymap = None
        self._device = None
        self._flank_bases = None        # optional: state name -> expected base, set by hmm_utils builders

    # ---- construction ----------------------------------------------------------------------
    def add_state(self, state):
        self.graph.add_node(state)

    def add_states(self, *states):
        for state in states:
            if isinstance(state, list):
                for s in state:
                    self.add_state(s)
            else:
                self.add_state(state)

    def add_transition(self, a, b, probability, pseudocount=None, group=None):
        self.graph.add_edge(a, b, _log(probability))

    def state_count(self):
        return self.n_states

    def edge_count(self):
        return self.n_edges

    def concatenate(self, other, suffix='', prefix=''):
        other.name = "{}{}{}".format(prefix, other.name, suffix)
        for state in other.states:
            state.name = "{}{}{}".format(prefix, state.name, suffix)
        self.graph = _Graph.union(self.graph, other.graph)
        self.add_transition(self.end, other.start, 1.00)
        self.end = other.end

    # ---- bake ------------------------------------------------------------------------------
    def bake(self, verbose=False, merge="All"):
        merge = merge.lower() if merge else None
        if merge not in (None, "none", "partial", "all"):
            raise ValueError("merge must be None, 'Partial' or 'All'")
        if merge in ("partial", "all"):
            self._merge(merge)
        states = self.graph.nodes()
        self.n_states = len(states)
        self.n_edges = self.graph.n_edges()
        silent_states = [s for s in states if s.is_silent()]
        normal_states = [s for s in states if not s.is_silent()]
        normal_states = sorted(normal_states, key=attrgetter('name'))          # hmm.pyx:861
        silent_states = sorted(silent_states, key=attrgetter('name'))          # hmm.pyx:862
        silent_set = set(silent_states)
        sub_succ = {s: [t for t in self.graph.succ[s] if t in silent_set] for s in silent_states}
        silent_sorted = _topological_sort(sub_succ, silent_states)             # hmm.pyx:870-874
        self.silent_start = len(normal_states)
        self.states = normal_states + silent_sorted
        indices = {s: i for i, s in enumerate(self.states)}
        n = self.n_states

        # CSR fill in edge-insertion order (hmm.pyx:970-1023)
        m = self.n_edges
        src = np.empty(m, np.int32)
        dst = np.empty(m, np.int32)
        lp = np.empty(m, np.float64)
        for k, (a, b, p) in enumerate(self.graph.edges()):
            src[k], dst[k], lp[k] = indices[a], indices[b], p
        order_in = np.argsort(dst, kind="stable")
        self._in_ptr = np.zeros(n + 1, np.int32)
        np.cumsum(np.bincount(dst, minlength=n), out=self._in_ptr[1:])
        self._in_src = np.ascontiguousarray(src[order_in])
        self._in_logp = np.ascontiguousarray(lp[order_in])
        order_out = np.argsort(src, kind="stable")
        self._out_ptr = np.zeros(n + 1, np.int32)
        np.cumsum(np.bincount(src, minlength=n), out=self._out_ptr[1:])
        self._out_dst = np.ascontiguousarray(dst[order_out])
        self._out_logp = np.ascontiguousarray(lp[order_out])
        self.finite = int(self._in_ptr[indices[self.end] + 1] - self._in_ptr[indices[self.end]] != 0)

        dist = None
        for state in self.states:
            if not state.is_silent():
                dist = state.distribution
                break
        if dist is None:
            raise ValueError("model has no emitting state")
        if not isinstance(dist, DiscreteDistribution):
            raise NotImplementedError("only DiscreteDistribution emissions are on adVNTR's path")
        self.d = 1
        keys = []
        for state in self.states[:self.silent_start]:
            keys.extend(state.distribution.keys())
        extra = set(keys) - set("ACGT")
        if extra:
            raise NotImplementedError("the engine scores the 4-letter DNA alphabet; got symbols %r" % sorted(extra))
        self.keymap = [{c: i for i, c in enumerate("ACGT")}]
        self._emis = np.full((self.silent_start, 4), NEGINF, np.float64)
        for i, state in enumerate(self.states[:self.silent_start]):
            w = _log(state.weight)                                             # hmm.pyx:928-930
            for j, c in enumerate("ACGT"):
                self._emis[i, j] = state.distribution.log_probability(c) + w   # hmm.pyx:1996-1997
        self.start_index = indices[self.start]
        self.end_index = indices[self.end]
        self._release_device()

    def _merge(self, merge):
        """The graph rewriting bake() does before numbering the states when merge is 'All' (the default of bake() and what
        hmm_utils.build_reference_repeat_finder_hmm and from_json get) or 'Partial' (hmm.pyx:725-823): orphan removal,
        normalisation of out-edges, folding of silent states that have a probability-1 transition."""
        g = self.graph
        if merge == "all":
            # hmm.pyx:725-757.  The two counters are allocated once for the initial node count and keep accumulating over
            # the passes while the node positions shift after every removal -- mirrored as is.
            n0 = len(g.nodes())
            in_count, out_count = [0] * n0, [0] * n0
            while True:
                removed = 0
                pre = g.nodes()
                index = {st: i for i, st in enumerate(pre)}
                for a, b, _ in list(g.edges()):
                    out_count[index[a]] += 1
                    in_count[index[b]] += 1
                for i, st in enumerate(pre):
                    if st is self.start or st is self.end:
                        continue
                    if in_count[i] == 0 or out_count[i] == 0:
                        removed += 1
                        g.remove_node(st)
                if removed == 0:
                    break
        # hmm.pyx:760-776: out-edges that do not sum to 1 (to 8 decimals) are normalised, except those of the end state
        for st in g.nodes():
            total = round(sum(np.e ** lp for lp in g.succ[st].values()), 8)
            if total != 1. and st is not self.end:
                for b in g.succ[st]:
                    g.succ[st][b] = g.succ[st][b] - _log(total)
        # hmm.pyx:779-814: a silent state with a probability-1 edge to b ('Partial': to a silent b) hands its in-edges to b
        while True:
            merged = 0
            for a, b, lp in list(g.edges()):
                if a not in g.succ or b not in g.succ:
                    continue
                if a is self.start or b is self.end:
                    continue
                lp = g.succ[a].get(b, lp)                 # the edge's current value (normalisation shares the record)
                if lp == 0.0 and a.is_silent() and (merge == "all" or b.is_silent()):
                    for x, y, d in list(g.edges()):
                        if y is a:
                            merged += 1
                            g.remove_edge(x, y)
                            g.add_edge(x, b, d)
                    g.remove_node(a)
            if merged == 0:
                break

    def dense_transition_matrix(self):
        m = len(self.states)
        # exp() only where there is an edge: exp(-inf) = 0 everywhere else (numpy's exp is an elementwise function, so
        # the finite entries come out as they would from exponentiating the full matrix; pinned by the golden models)
        out = np.zeros((m, m))
        src = np.repeat(np.arange(m), np.diff(self._out_ptr))
        out[src, self._out_dst] = np.exp(self._out_logp)
        return out

    @classmethod
    def from_matrix(cls, transition_probabilities, distributions, starts, ends=None, state_names=None,
                    name=None, verbose=False, merge='All'):
        model = cls(name=name)
        state_names = state_names or ["s{}".format(i) for i in range(len(distributions))]
        states = [State(distribution, name=sname) for sname, distribution in zip(state_names, distributions)]
        n = len(states)
        for state in states:
            model.add_state(state)
        for i, prob in enumerate(starts):
            if prob != 0:
                model.add_transition(model.start, states[i], prob)
        # same edge insertion order as the reference's double loop (row-major over non-zero entries); after
        # that loop the reference's inner variable j is left at the last column index
        tp = np.asarray(transition_probabilities, dtype=np.float64)
        rows, cols = np.nonzero(tp)
        vals = tp[rows, cols]
        for i, jj, prob in zip(rows.tolist(), cols.tolist(), vals.tolist()):
            model.add_transition(states[i], states[jj], prob)
        j = tp.shape[1] - 1 if tp.ndim == 2 and tp.shape[1] else 0
        if ends is not None:
            for i, prob in enumerate(ends):
                if prob != 0:
                    # hmm.pyx:3231-3235 adds the edge from states[j] with the stale inner-loop j (= n-1), not
                    # states[i]; adVNTR's models depend on that (SURVEY 8a-3 quirk i), so it is reproduced.
                    model.add_transition(states[j], model.end, prob)
        model.bake(verbose=verbose, merge=merge)
        return model

    @classmethod
    def _from_built(cls, built, name):
        """A baked model around a model of the native builder (_lib.BuiltModel)."""
        return _BuiltHiddenMarkovModel(built, name)

    # ---- JSON (hmm.pyx:3023-3143; the format of the reference's stored HMMs, vntr_finder.py:124-137) ---------
    def to_json(self, separators=(',', ' : '), indent=4):
        import json
        if self.d == 0:
            raise ValueError("must bake model before serialising it")

        def state_json(st):
            dist = None
            if not st.is_silent():
                dist = {'class': 'Distribution', 'name': 'DiscreteDistribution',
                        'parameters': [{str(k): v for k, v in st.distribution.dist.items()}], 'frozen': st.distribution.frozen}
            return {'class': 'State', 'distribution': dist, 'name': st.name, 'weight': st.weight}
        index = {st: i for i, st in enumerate(self.states)}
        if self.graph is not None:
            triples = [(index[a], index[b], lp) for a, b, lp in self.graph.edges()]
        else:                                   # a model of the native builder: edges in CSR order
            a = self.baked_arrays()
            dst = np.repeat(np.arange(a["m"]), np.diff(a["in_ptr"]))
            triples = list(zip(a["in_src"].tolist(), dst.tolist(), a["in_logp"].tolist()))
        edges = [(s_, e_, math.e ** lp, math.e ** lp, None) for s_, e_, lp in triples]
        model = {'class': 'HiddenMarkovModel', 'name': self.name, 'start': state_json(self.start), 'end': state_json(self.end),
                 'states': [state_json(st) for st in self.states], 'end_index': self.end_index,
                 'start_index': self.start_index, 'silent_index': self.silent_start, 'edges': edges,
                 'distribution ties': []}
        return json.dumps(model, separators=separators, indent=indent)

    @classmethod
    def from_json(cls, s, verbose=False):
        """A model from the reference's JSON (a string, or the name of a file holding one).  As in the reference the
        loaded graph is baked with the DEFAULT merge ('All'): the probabilities go through one more log, orphan states
        (among them the fresh model's own start and end) disappear, probability-1 silent states are folded."""
        import json
        try:
            d = json.loads(s)
        except ValueError:
            try:
                with open(s, 'r') as infile:
                    d = json.load(infile)
            except (IOError, OSError, ValueError):
                raise IOError("String must be properly formatted JSON or filename of properly formatted JSON.")
        model = cls(str(d['name']))
        states = []
        for j in d['states']:
            if j['class'] != 'State':
                raise IOError("State object attempting to decode {} object".format(j['class']))
            dist = None
            if j['distribution'] is not None:
                if j['distribution'].get('name') != 'DiscreteDistribution':
                    raise NotImplementedError("only DiscreteDistribution emissions are on adVNTR's path")
                dist = DiscreteDistribution(j['distribution']['parameters'][0], j['distribution'].get('frozen', False))
            states.append(State(dist, str(j['name']), j['weight']))
        for i, j in d.get('distribution ties', []):
            states[i].tie(states[j])
        model.add_states(states)
        model.start = states[d['start_index']]
        model.end = states[d['end_index']]
        for start, end, probability, pseudocount, group in d['edges']:
            model.add_transition(states[start], states[end], probability, pseudocount, group)
        model.bake(verbose=verbose)
        return model

    # ---- device residency --------------------------------------------------------------------
    def baked_arrays(self):
        """The arrays the C ABI takes (advntr_hmm_create)."""
        if self.d == 0:
            raise ValueError("must bake model first")
        return dict(m=len(self.states), silent_start=self.silent_start, start_index=self.start_index,
                    end_index=self.end_index, in_ptr=self._in_ptr, in_src=self._in_src, in_logp=self._in_logp,
                    emis_logp=self._emis, state_class=self.state_classes())

    def state_classes(self):
        fb = self._flank_bases or {}
        return np.array([state_class_from_name(s.name, fb.get(s.name)) for s in self.states], dtype=np.uint16)

    def set_flank_bases(self, mapping):
        """state name -> base code (0..3) that an M*_suffix / M*_prefix state is compared with in
        get_flanking_regions_matching_rate (hmm_utils.py:236,248)."""
        self._flank_bases = dict(mapping)
        self._release_device()

    def _release_device(self):
        if self._device is not None:
            self._device.close()
            self._device = None

    def device_model(self):
        if self._device is None:
            a = self.baked_arrays()
            self._device = _lib.DeviceModel(a["m"], a["silent_start"], a["start_index"], a["end_index"],
                                            a["in_ptr"], a["in_src"], a["in_logp"], a["emis_logp"], a["state_class"])
        return self._device

    # ---- scoring (HIP engine only) -----------------------------------------------------------
    def viterbi(self, sequence):
        """(logp, [(state_index, State), ...]) or (-inf, None); ValueError on a non-ACGT symbol."""
        if self.d == 0:
            raise ValueError("must bake model before using Viterbi algorithm")
        logp, _, paths = self.viterbi_batch([sequence], want_paths=True)
        path = paths[0]
        lp = float(logp[0])
        if path is None or not (lp > NEGINF):
            return lp, None
        return lp, [(i, self.states[i]) for i in path]

    def viterbi_batch(self, sequences, want_paths=False, want_summary=True, flags=0):
        if self.d == 0:
            raise ValueError("must bake model before using Viterbi algorithm")
        bases, off = _lib.encode_reads(sequences)
        dm = self.device_model()
        return _lib.viterbi_batch([dm], bases, off, np.zeros(len(sequences), np.int32), flags=flags,
                                  want_paths=want_paths, want_summary=want_summary)

    def log_probability(self, sequence, check_input=True):
        if self.d == 0:
            raise ValueError("must bake model before computing probability")
        bases, off = _lib.encode_reads([sequence])
        return float(_lib.forward_batch([self.device_model()], bases, off, np.zeros(1, np.int32))[0])

    def log_probability_batch(self, sequences):
        bases, off = _lib.encode_reads(sequences)
        return _lib.forward_batch([self.device_model()], bases, off, np.zeros(len(sequences), np.int32))


class _BuiltHiddenMarkovModel(HiddenMarkovModel):
    """Same attributes as a model after bake(), backed by the native builder's arrays (csrc/model_builder.h).  The
    host copies (CSR arrays, State list) are only materialised when something asks for them: scoring needs neither,
    the model goes from the builder straight to the device (advntr_built_upload).  `states` carry the reference's
    names (all any consumer of a vpath reads) and, for emitting states, a distribution made from the baked
    log-probabilities."""

    def __init__(self, built, name):
        # (the attributes HiddenMarkovModel.__init__ sets, without the construction graph and the two State objects it makes:
        # thousands of these are made per run; `start` / `end` appear with `states`)
        self.name = str(name)
        self.model = "HiddenMarkovModel"
        self._start = self._end = None
        self._device = None
        self._flank_bases = None
        self.graph = None                      # the construction graph stays inside the native builder
        self._built = built
        self._arrays = None
        self._states = None
        self.n_states, self.n_edges = built.m, built.n_edges
        self.silent_start, self.start_index, self.end_index = built.silent_start, built.start_index, built.end_index
        self.d = 1
        self.keymap = [{c: i for i, c in enumerate("ACGT")}]

    @property
    def start(self):
        if self._start is None:
            self.states
        return self._start

    @start.setter
    def start(self, value):
        self._start = value

    @property
    def end(self):
        if self._end is None:
            self.states
        return self._end

    @end.setter
    def end(self, value):
        self._end = value

    @property
    def states(self):
        if self._states is None:
            emis = self.baked_arrays()["emis_logp"]
            out = []
            for i, nm in enumerate(self._built.names()):
                dist = None
                if i < self.silent_start:
                    dist = DiscreteDistribution({c: math.exp(emis[i, j]) for j, c in enumerate("ACGT")})
                    dist.log_dist = {c: float(emis[i, j]) for j, c in enumerate("ACGT")}
                out.append(State(dist, name=nm))
            self._states = out
            self.start, self.end = out[self.start_index], out[self.end_index]
        return self._states

    @states.setter
    def states(self, value):
        self._states = value or None

    def bake(self, verbose=False, merge="All"):
        pass                                   # already baked

    def baked_arrays(self):
        if self._arrays is None:
            self._arrays = self._built.arrays()
            a = self._arrays
            self._in_ptr, self._in_src, self._in_logp, self._emis = a["in_ptr"], a["in_src"], a["in_logp"], a["emis_logp"]
            self.finite = int(a["in_ptr"][self.end_index + 1] != a["in_ptr"][self.end_index])
        return self._arrays

    def state_classes(self):
        return self.baked_arrays()["state_class"]

    def set_flank_bases(self, mapping):
        raise NotImplementedError("the native builder already encodes the flank bases in the state classes")

    def _merge(self, merge):
        """The graph rewriting bake() does before numbering the states when merge is 'All' (the default of bake() and what
        hmm_utils.build_reference_repeat_finder_hmm and from_json get) or 'Partial' (hmm.pyx:725-823): orphan removal,
        normalisation of out-edges, folding of silent states that have a probability-1 transition."""
        g = self.graph
        if merge == "all":
            # hmm.pyx:725-757.  The two counters are allocated once for the initial node count and keep accumulating over
            # the passes while the node positions shift after every removal -- mirrored as is.
            n0 = len(g.nodes())
            in_count, out_count = [0] * n0, [0] * n0
            while True:
                removed = 0
                pre = g.nodes()
                index = {st: i for i, st in enumerate(pre)}
                for a, b, _ in list(g.edges()):
                    out_count[index[a]] += 1
                    in_count[index[b]] += 1
                for i, st in enumerate(pre):
                    if st is self.start or st is self.end:
                        continue
                    if in_count[i] == 0 or out_count[i] == 0:
                        removed += 1
                        g.remove_node(st)
                if removed == 0:
                    break
        # hmm.pyx:760-776: out-edges that do not sum to 1 (to 8 decimals) are normalised, except those of the end state
        for st in g.nodes():
            total = round(sum(np.e ** lp for lp in g.succ[st].values()), 8)
            if total != 1. and st is not self.end:
                for b in g.succ[st]:
                    g.succ[st][b] = g.succ[st][b] - _log(total)
        # hmm.pyx:779-814: a silent state with a probability-1 edge to b ('Partial': to a silent b) hands its in-edges to b
        while True:
            merged = 0
            for a, b, lp in list(g.edges()):
                if a not in g.succ or b not in g.succ:
                    continue
                if a is self.start or b is self.end:
                    continue
                lp = g.succ[a].get(b, lp)                 # the edge's current value (normalisation shares the record)
                if lp == 0.0 and a.is_silent() and (merge == "all" or b.is_silent()):
                    for x, y, d in list(g.edges()):
                        if y is a:
                            merged += 1
                            g.remove_edge(x, y)
                            g.add_edge(x, b, d)
                    g.remove_node(a)
            if merged == 0:
                break

    def dense_transition_matrix(self):
        a = self.baked_arrays()
        m = a["m"]
        out = np.zeros((m, m))
        dst = np.repeat(np.arange(m), np.diff(a["in_ptr"]))
        out[a["in_src"], dst] = np.exp(a["in_logp"])
        return out

    def device_model(self):
        if self._device is None:
            self._device = self._built.upload()
        return self._device


def device_models(models, threads=0):
    """The device handles of a list of baked models; the ones that came from the native builder and are not on the
    device yet go up together (one allocation, one copy: _lib.upload_built_models)."""
    todo = [m for m in models if isinstance(m, _BuiltHiddenMarkovModel) and m._device is None]
    if len(todo) > 1:
        for m, dm in zip(todo, _lib.upload_built_models([m._built for m in todo], threads)):
            m._device = dm
    return [m.device_model() for m in models]
