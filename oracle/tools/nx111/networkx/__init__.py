"""Restatement of the networkx==1.11 DiGraph subset used by the reference's pomegranate.

TEST INFRASTRUCTURE, build container only (see oracle/tools/build_reference.py).

The reference pins networkx==1.11 (setup.py:19, requirements.txt:6); this image
ships networkx 3.4 whose API (no edges_iter / .edge, generator nodes(), Kahn-style
topological_sort without nbunch) is incompatible with pomegranate/hmm.pyx:199-1011.
networkx 1.11 is absent from /root/reference, so -- as the task prescribes for an
absent third-party dependency -- its published algorithm is restated here:

  * DiGraph = three dict-of-dicts (node, succ(=adj=edge), pred); on Python >= 3.7
    these iterate in insertion order, which is what makes hmm.pyx's
    `graph.edges_iter()` (hmm.pyx:970,994) and hence the in-edge order of the
    baked CSR deterministic;
  * add_edge on an existing edge updates the data dict in place and keeps the
    edge's original position (networkx 1.11 digraph.py add_edge);
  * subgraph(nbunch): node order = nbunch order, successor order = self order;
  * union(G, H): nodes of G, edges of G, nodes of H, edges of H, fresh data dicts;
  * topological_sort(G, nbunch): the 1.11 non-recursive DFS (dag.py), post-order
    reversed, successors pushed in adjacency order.

Only what hmm.pyx calls is provided.  Nothing in the shipped package imports this.
"""
__version__ = "1.11-restated"


class NetworkXError(Exception):
    pass


class NetworkXUnfeasible(NetworkXError):
    pass


class DiGraph(object):
    def __init__(self):
        self.graph = {}
        self.node = {}
        self.adj = {}
        self.pred = {}
        self.succ = self.adj
        self.edge = self.adj
        self.name = ''

    def is_directed(self):
        return True

    def is_multigraph(self):
        return False

    def __iter__(self):
        return iter(self.node)

    def __contains__(self, n):
        try:
            return n in self.node
        except TypeError:
            return False

    def __len__(self):
        return len(self.node)

    def __getitem__(self, n):
        return self.adj[n]

    def add_node(self, n, attr_dict=None, **attr):
        if attr_dict is None:
            attr_dict = attr
        else:
            attr_dict.update(attr)
        if n not in self.succ:
            self.succ[n] = {}
            self.pred[n] = {}
            self.node[n] = attr_dict
        else:
            self.node[n].update(attr_dict)

    def add_nodes_from(self, nodes, **attr):
        for n in nodes:
            if n not in self.succ:
                self.succ[n] = {}
                self.pred[n] = {}
                self.node[n] = attr.copy()
            else:
                self.node[n].update(attr)

    def remove_node(self, n):
        try:
            nbrs = self.succ[n]
            del self.node[n]
        except KeyError:
            raise NetworkXError("The node %s is not in the digraph." % (n,))
        for u in nbrs:
            del self.pred[u][n]
        del self.succ[n]
        for u in self.pred[n]:
            del self.succ[u][n]
        del self.pred[n]

    def add_edge(self, u, v, attr_dict=None, **attr):
        if attr_dict is None:
            attr_dict = attr
        else:
            attr_dict.update(attr)
        if u not in self.succ:
            self.succ[u] = {}
            self.pred[u] = {}
            self.node[u] = {}
        if v not in self.succ:
            self.succ[v] = {}
            self.pred[v] = {}
            self.node[v] = {}
        datadict = self.adj[u].get(v, {})
        datadict.update(attr_dict)
        self.succ[u][v] = datadict
        self.pred[v][u] = datadict

    def add_edges_from(self, ebunch, attr_dict=None, **attr):
        if attr_dict is None:
            attr_dict = attr
        else:
            attr_dict.update(attr)
        for e in ebunch:
            ne = len(e)
            if ne == 3:
                u, v, dd = e
            elif ne == 2:
                u, v = e
                dd = {}
            else:
                raise NetworkXError("Edge tuple %s must be a 2-tuple or 3-tuple." % (e,))
            if u not in self.succ:
                self.succ[u] = {}
                self.pred[u] = {}
                self.node[u] = {}
            if v not in self.succ:
                self.succ[v] = {}
                self.pred[v] = {}
                self.node[v] = {}
            datadict = self.adj[u].get(v, {})
            datadict.update(attr_dict)
            datadict.update(dd)
            self.succ[u][v] = datadict
            self.pred[v][u] = datadict

    def remove_edge(self, u, v):
        try:
            del self.succ[u][v]
            del self.pred[v][u]
        except KeyError:
            raise NetworkXError("The edge %s-%s not in graph." % (u, v))

    def nodes_iter(self, data=False):
        if data:
            return iter(self.node.items())
        return iter(self.node)

    def nodes(self, data=False):
        return list(self.nodes_iter(data))

    def nbunch_iter(self, nbunch=None):
        if nbunch is None:
            return iter(self.adj.keys())
        if nbunch in self:
            return iter([nbunch])
        adj = self.adj
        return (n for n in nbunch if n in adj)

    def edges_iter(self, nbunch=None, data=False):
        if nbunch is None:
            nodes_nbrs = self.adj.items()
        else:
            nodes_nbrs = ((n, self.adj[n]) for n in self.nbunch_iter(nbunch))
        if data is True:
            for n, nbrs in nodes_nbrs:
                for nbr, ddict in nbrs.items():
                    yield (n, nbr, ddict)
        else:
            for n, nbrs in nodes_nbrs:
                for nbr in nbrs:
                    yield (n, nbr)

    def edges(self, nbunch=None, data=False):
        return list(self.edges_iter(nbunch, data))

    def subgraph(self, nbunch):
        bunch = self.nbunch_iter(nbunch)
        H = self.__class__()
        for n in bunch:
            H.node[n] = self.node[n]
        H_succ = H.succ
        H_pred = H.pred
        self_succ = self.succ
        for n in H:
            H_succ[n] = {}
            H_pred[n] = {}
        for u in H_succ:
            Hnbrs = H_succ[u]
            for v, datadict in self_succ[u].items():
                if v in H_succ:
                    Hnbrs[v] = datadict
                    H_pred[v][u] = datadict
        H.graph = self.graph
        return H


def union(G, H, rename=(None, None), name=None):
    R = G.__class__()
    if name is None:
        name = "union( %s, %s )" % (G.name, H.name)
    R.name = name
    if set(G) & set(H):
        raise NetworkXError('The node sets of G and H are not disjoint.')
    G_edges = G.edges_iter(data=True)
    H_edges = H.edges_iter(data=True)
    R.add_nodes_from(G)
    R.add_edges_from(G_edges)
    R.add_nodes_from(H)
    R.add_edges_from(H_edges)
    R.node.update(G.node)
    R.node.update(H.node)
    R.graph.update(G.graph)
    R.graph.update(H.graph)
    return R


def topological_sort(G, nbunch=None, reverse=False):
    if not G.is_directed():
        raise NetworkXError("Topological sort not defined on undirected graphs.")
    seen = set()
    order = []
    explored = set()
    if nbunch is None:
        nbunch = G.nodes_iter()
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
            for n in G[w]:
                if n not in explored:
                    if n in seen:
                        raise NetworkXUnfeasible("Graph contains a cycle.")
                    new_nodes.append(n)
            if new_nodes:
                fringe.extend(new_nodes)
            else:
                explored.add(w)
                order.append(w)
                fringe.pop()
    if reverse:
        return order
    return list(reversed(order))
