"""tests/golden/ref_shim.py against the PUBLISHED examples of the libraries it stands in for.

The goldens under tests/golden/ are outputs of the reference's own code executed on top of the shim (dgl and
torch_cluster are not installable in the build image), so the shim must be pinned by something other than the oracle
it helps to pin.  These tests replay the worked examples of the libraries' own documentation -- inputs and printed
outputs quoted below -- through the shim:

  torch_cluster README (github.com/rusty1s/pytorch_cluster, sections "KNN-Graph", "Radius-Graph", "knn", "radius")
  DGL API reference (docs.dgl.ai/generated/): dgl.batch, dgl.readout_nodes, dgl.DGLGraph.multi_update_all,
      dgl.DGLGraph.apply_edges, dgl.DGLGraph.local_scope, dgl.function.u_sub_v / copy_e / sum / mean,
      dgl.DGLGraph.remove_edges / add_edges

(the build container has no network: the examples are quoted from those pages as published for torch_cluster 1.6 and
DGL 1.x/2.x; the README examples are two-dimensional, the shim's distance is written for 3-D points, so a zero z
column is appended).  Runs without /root/reference: only ref_shim.install() needs the reference tree."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import ref_shim as S  # noqa: E402


def _pad3(x):
    return torch.cat([x, torch.zeros(x.shape[0], 1)], dim=1)


# ---- torch_cluster README -------------------------------------------------------------------------------------------
X4 = _pad3(torch.tensor([[-1.0, -1.0], [-1.0, 1.0], [1.0, -1.0], [1.0, 1.0]]))
Y2 = _pad3(torch.tensor([[-1.0, 0.0], [1.0, 0.0]]))
GRAPH_EXPECT = torch.tensor([[1, 2, 0, 3, 0, 3, 1, 2],
                             [0, 0, 1, 1, 2, 2, 3, 3]])
ASSIGN_EXPECT = torch.tensor([[0, 0, 1, 1],
                              [0, 1, 2, 3]])


def test_knn_graph_readme_example():
    """README "KNN-Graph": knn_graph(x, k=2, batch=[0,0,0,0], loop=False) -> row 0 = neighbour (source), row 1 =
    the point it belongs to (target), targets ascending."""
    out = S.tc_knn_graph(X4, k=2, batch=torch.zeros(4, dtype=torch.int64), loop=False)
    assert torch.equal(out, GRAPH_EXPECT)


def test_radius_graph_readme_example():
    """README "Radius-Graph": radius_graph(x, r=2.5, batch=[0,0,0,0], loop=False): the diagonal partner at distance
    2.83 is outside, the two side neighbours at 2.0 inside."""
    out = S.tc_radius_graph(X4, r=2.5, batch=torch.zeros(4, dtype=torch.int64), loop=False)
    assert torch.equal(out, GRAPH_EXPECT)


def test_knn_readme_example():
    """README "knn": knn(x, y, 2, batch_x, batch_y) -> row 0 = index into y (the query), row 1 = index into x."""
    out = S.tc_knn(X4, Y2, 2, torch.zeros(4, dtype=torch.int64), torch.zeros(2, dtype=torch.int64))
    assert torch.equal(out, ASSIGN_EXPECT)


def test_radius_readme_example():
    """README "radius": radius(x, y, 1.5, batch_x, batch_y) -> same index convention as knn."""
    out = S.tc_radius(X4, Y2, 1.5, torch.zeros(4, dtype=torch.int64), torch.zeros(2, dtype=torch.int64))
    assert torch.equal(out, ASSIGN_EXPECT)


def test_cluster_ops_respect_batch_vectors_and_limits():
    """Documented arguments the reference relies on (dynamics_gvp.py:194-211): only points of the same example are
    neighbours (batch / batch_x / batch_y), at most max_num_neighbors per query, no self loops with loop=False."""
    x = torch.cat([X4, X4 + 0.25])
    b = torch.tensor([0, 0, 0, 0, 1, 1, 1, 1])
    e = S.tc_radius_graph(x, r=2.5, batch=b, max_num_neighbors=200)
    assert torch.equal(b[e[0]], b[e[1]]) and e.shape[1] == 16 and bool((e[0] != e[1]).all())
    e = S.tc_radius_graph(x, r=2.5, batch=b, max_num_neighbors=1)
    assert e.shape[1] == 8 and torch.equal(e[1], torch.arange(8))
    a = S.tc_radius(x, Y2, 1.5, b, torch.tensor([0, 1]))
    assert torch.equal(b[a[1]], torch.tensor([0, 1])[a[0]])
    k = S.tc_knn(x, Y2, 3, b, torch.tensor([1, 1]))
    assert bool((b[k[1]] == 1).all()) and k.shape[1] == 6


# ---- DGL API reference ----------------------------------------------------------------------------------------------
def _homograph(u, v, n):
    return S.heterograph({("_N", "_E", "_N"): (u, v)}, num_nodes_dict={"_N": n})


def test_batch_doc_example():
    """dgl.batch: g1 = graph(([0,1,2],[1,2,3])), g2 = graph(([0,0,0,1],[0,1,2,0])) ->
    batch_size 2, batch_num_nodes [4,3], batch_num_edges [3,4], edges ([0,1,2,4,4,4,5],[1,2,3,4,5,6,4]);
    node features are concatenated in list order."""
    g1 = _homograph([0, 1, 2], [1, 2, 3], 4)
    g2 = _homograph([0, 0, 0, 1], [0, 1, 2, 0], 3)
    g1.nodes["_N"].data["h"] = torch.tensor([1.0, 2.0, 3.0, 4.0])
    g2.nodes["_N"].data["h"] = torch.tensor([5.0, 6.0, 7.0])
    bg = S.batch([g1, g2])
    assert bg.batch_size == 2
    assert bg.batch_num_nodes("_N").tolist() == [4, 3] and bg.batch_num_edges("_E").tolist() == [3, 4]
    u, v = bg.edges(form="uv", etype="_E")
    assert u.tolist() == [0, 1, 2, 4, 4, 4, 5] and v.tolist() == [1, 2, 3, 4, 5, 6, 4]
    assert bg.nodes["_N"].data["h"].tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0]
    # dgl.unbatch doc example: the inverse
    a, b = S.unbatch(bg)
    assert a.num_nodes("_N") == 4 and b.num_nodes("_N") == 3
    assert [t.tolist() for t in b.edges(form="uv", etype="_E")] == [[0, 0, 0, 1], [0, 1, 2, 0]]
    assert b.nodes["_N"].data["h"].tolist() == [5.0, 6.0, 7.0]


def test_readout_nodes_doc_example():
    """dgl.readout_nodes: g1.ndata['h'] = [1,2], g2.ndata['h'] = [1,2,3]; the page prints the sums [3., 6.] of the
    batched graph; op='mean' (what the reference calls, pharmacodiff.py:96) divides by the node counts."""
    g1 = _homograph([0, 1], [1, 0], 2)
    g2 = _homograph([0, 1], [1, 2], 3)
    g1.nodes["_N"].data["h"] = torch.tensor([1.0, 2.0])
    g2.nodes["_N"].data["h"] = torch.tensor([1.0, 2.0, 3.0])
    bg = S.batch([g1, g2])
    out = S.readout_nodes(bg, "h", ntype="_N", op="mean")
    assert torch.allclose(out, torch.tensor([3.0 / 2, 6.0 / 3]))


def _follow_graph():
    # dgl.DGLGraph.multi_update_all doc example
    g = S.heterograph({("user", "follows", "user"): ([0, 1], [1, 1]), ("game", "attracts", "user"): ([0], [1])},
                      num_nodes_dict={"user": 2, "game": 1})
    g.nodes["user"].data["h"] = torch.tensor([[1.0], [2.0]])
    g.nodes["game"].data["h"] = torch.tensor([[1.0]])
    return g


def test_multi_update_all_doc_example():
    """dgl.DGLGraph.multi_update_all: {'follows': (copy_u('h','m'), sum('m','h')), 'attracts': (copy_u('h','m'),
    sum('m','h'))}, cross reducer "sum" -> user h = [[0.],[4.]]: per-relation reduce, then the cross-type sum; the
    node without in-edges gets ZERO.  The reference sends edge data (copy_e), so the source feature is first copied
    onto the edges with apply_edges -- the edge UDF sees edges.src, as documented for dgl.DGLGraph.apply_edges."""
    g = _follow_graph()
    for et in ("follows", "attracts"):
        g.apply_edges(lambda edges: {"m_e": edges.src["h"]}, etype=et)
    fn = sys.modules.get("dgl.function")
    copy_e, fsum, fmean = S._CopyE, S._fn_sum, S._fn_mean
    g.multi_update_all({"follows": (copy_e("m_e", "m"), fsum("m", "h")),
                        "attracts": (copy_e("m_e", "m"), fsum("m", "h"))}, "sum")
    assert g.nodes["user"].data["h"].tolist() == [[0.0], [4.0]]
    # fn.mean as the per-relation reducer (gvp.py:381): follows -> mean(1, 2) = 1.5, attracts -> 1; cross sum 2.5
    g = _follow_graph()
    for et in ("follows", "attracts"):
        g.apply_edges(lambda edges: {"m_e": edges.src["h"]}, etype=et)
    g.multi_update_all({"follows": (copy_e("m_e", "m"), fmean("m", "h")),
                        "attracts": (copy_e("m_e", "m"), fmean("m", "h"))}, "sum")
    assert g.nodes["user"].data["h"].tolist() == [[0.0], [2.5]]
    assert fn is None or fn.copy_e is S._CopyE


def test_u_sub_v_is_source_minus_destination():
    """dgl.function.u_sub_v(lhs_field, rhs_field, out): "u" is the source node, "v" the destination:
    out = u[lhs] - v[rhs] on every edge (the reference's x_diff, gvp.py:474)."""
    g = S.heterograph({("a", "ab", "b"): ([0, 1, 1], [0, 0, 1])}, num_nodes_dict={"a": 2, "b": 2})
    g.nodes["a"].data["x"] = torch.tensor([[1.0, 0.0, 0.0], [0.0, 2.0, 0.0]])
    g.nodes["b"].data["x"] = torch.tensor([[0.0, 0.0, 5.0], [1.0, 1.0, 1.0]])
    g.apply_edges(S._USubV("x", "x", "x_diff"), etype="ab")
    assert g.edges["ab"].data["x_diff"].tolist() == [[1.0, 0.0, -5.0], [0.0, 2.0, -5.0], [-1.0, 1.0, -1.0]]


def test_apply_edges_udf_doc_example():
    """dgl.DGLGraph.apply_edges (heterograph example): ('user','plays','game'): ([0,1,1,2],[0,0,2,1]),
    edata h = ones(4,5); apply_edges(lambda edges: {'h': edges.data['h'] * 2}) -> all 2."""
    g = S.heterograph({("user", "plays", "game"): ([0, 1, 1, 2], [0, 0, 2, 1])}, num_nodes_dict={"user": 3, "game": 3})
    g.edges[("user", "plays", "game")].data["h"] = torch.ones(4, 5)
    g.apply_edges(lambda edges: {"h": edges.data["h"] * 2}, etype="plays")
    assert torch.equal(g.edges["plays"].data["h"], 2 * torch.ones(4, 5))
    # the UDF's EdgeBatch exposes canonical_etype and the gathered src / dst frames (gvp.py:540-547)
    seen = {}
    g.nodes["user"].data["u"] = torch.tensor([10.0, 20.0, 30.0])
    g.nodes["game"].data["v"] = torch.tensor([1.0, 2.0, 3.0])

    def udf(edges):
        seen["cet"] = edges.canonical_etype
        return {"s": edges.src["u"] + edges.dst["v"]}
    g.apply_edges(udf, etype="plays")
    assert seen["cet"] == ("user", "plays", "game")
    assert g.edges["plays"].data["s"].tolist() == [11.0, 21.0, 23.0, 32.0]


def test_local_scope_doc_example():
    """dgl.DGLGraph.local_scope: features created inside the scope are gone afterwards ("'h' in g.edata -> False"),
    features that existed before are restored; structure changes are NOT reverted (the reference removes its dynamic
    edges by hand, dynamics_gvp.py:183)."""
    g = _homograph([0, 1, 1], [0, 0, 2], 3)
    g.nodes["_N"].data["keep"] = torch.zeros(3)
    with g.local_scope():
        g.edges["_E"].data["h"] = torch.ones(3, 3)
        g.nodes["_N"].data["keep"] = torch.ones(3)
        g.add_edges([2], [1], etype="_E")
    assert "h" not in g.edges["_E"].data
    assert g.nodes["_N"].data["keep"].tolist() == [0.0, 0.0, 0.0]
    assert g.num_edges("_E") == 4


def test_add_and_remove_edges_doc_examples():
    """dgl.DGLGraph.add_edges appends (edge ids keep their order); remove_edges(eids) drops those ids and their
    features; edges(form='eid') lists 0..E-1 (dynamics_gvp.py:238-240 removes all ids of an edge type)."""
    g = S.heterograph({("user", "plays", "game"): ([0, 1, 1, 2], [0, 0, 2, 1])}, num_nodes_dict={"user": 3, "game": 3})
    g.edges["plays"].data["w"] = torch.tensor([0.0, 1.0, 2.0, 3.0])
    g.remove_edges(torch.tensor([0, 1]), etype="plays")            # doc example: edges (1,2),(2,1) remain
    u, v = g.edges(form="uv", etype="plays")
    assert u.tolist() == [1, 2] and v.tolist() == [2, 1] and g.edges["plays"].data["w"].tolist() == [2.0, 3.0]
    g.add_edges(torch.tensor([0]), torch.tensor([2]), etype="plays")
    u, v = g.edges(form="uv", etype="plays")
    assert u.tolist() == [1, 2, 0] and v.tolist() == [2, 1, 2]
    g.remove_edges(g.edges(form="eid", etype="plays"), etype="plays")
    assert g.num_edges("plays") == 0


def test_set_batch_num_edges_is_what_batch_num_edges_returns():
    """dgl.DGLGraph.set_batch_num_edges / batch_num_edges: the graph reports the caller's bookkeeping verbatim (DGL
    stores it without validation against the structure beyond the total) -- this is how the reference's per-graph
    edge counts, including the ones it derives from the wrong index row (dynamics_gvp.py:220), reach gvp.py:506."""
    g = S.batch([_homograph([0], [1], 2), _homograph([0, 1], [1, 0], 2)])
    g.set_batch_num_edges({("_N", "_E", "_N"): torch.tensor([3, 0])})
    assert g.batch_num_edges(("_N", "_E", "_N")).tolist() == [3, 0]
