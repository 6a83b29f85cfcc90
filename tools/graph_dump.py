#!/usr/bin/env python3
"""Read back the captured step's hipGraphs: every node (kernel name / memset target) and every edge the replay honours.

    python tools/graph_dump.py OUT_PREFIX [--memset-nodes] [--config dp|single] [--poe label|paired|cluster]

``--memset-nodes`` re-enables round 2's hipMemsetAsync calls on the captured path (SPV_MEMSET_NODES=1, csrc/spv_abi.hip: zero_fill)
so that the graph that intermittently produced NaN gradients can be inspected edge by edge.  Writes, per
graph k = 1 (forward + loss + decoder half of the backward pass, or the whole step) and, for ``dp``, k = 2 (encoder half):
    OUT_PREFIX_graphK.json   {"nodes": [{"id", "type", "name" | "memset": {dst, bytes}}], "edges": [[from, to], ...]}
    OUT_PREFIX_graphK.dot    hipGraphDebugDotPrint's own rendering (when the runtime writes one)
tools/graph_check.py (and tests/test_graph_edges.py on the committed copies under profiles/) verify that every fill reaches its consumer."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}


class KernelNodeParams(C.Structure):   # hipKernelNodeParams
    _fields_ = [("blockDim", C.c_uint * 3), ("extra", C.c_void_p), ("func", C.c_void_p), ("gridDim", C.c_uint * 3), ("kernelParams", C.c_void_p),
                ("sharedMemBytes", C.c_uint)]


class MemsetParams(C.Structure):       # hipMemsetParams
    _fields_ = [("dst", C.c_void_p), ("elementSize", C.c_uint), ("height", C.c_size_t), ("pitch", C.c_size_t), ("value", C.c_uint), ("width", C.c_size_t)]


def read_graph(hip, graph: int) -> dict:
    g = C.c_void_p(graph)
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(g, None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(g, nodes, C.byref(n)) == 0
    ids = {int(nodes[i]): i for i in range(n.value)}
    out = []
    hip.hipKernelNameRefByPtr.restype = C.c_char_p
    for i in range(n.value):
        t = C.c_int(-1)
        assert hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t)) == 0
        d = {"id": i, "type": NODE_TYPES.get(t.value, str(t.value))}
        if t.value == 0:
            kp = KernelNodeParams()
            if hip.hipGraphKernelNodeGetParams(C.c_void_p(nodes[i]), C.byref(kp)) == 0:
                nm = hip.hipKernelNameRefByPtr(C.c_void_p(kp.func), None)
                d["name"] = nm.decode() if nm else hex(kp.func or 0)
                d["grid"], d["block"] = list(kp.gridDim), list(kp.blockDim)
        elif t.value == 2:
            mp = MemsetParams()
            if hip.hipGraphMemsetNodeGetParams(C.c_void_p(nodes[i]), C.byref(mp)) == 0:
                d["memset"] = {"dst": mp.dst, "bytes": int(mp.width) * int(mp.elementSize) * max(int(mp.height), 1), "value": mp.value}
        out.append(d)
    ne = C.c_size_t(0)
    assert hip.hipGraphGetEdges(g, None, None, C.byref(ne)) == 0
    fr, to = (C.c_void_p * ne.value)(), (C.c_void_p * ne.value)()
    if ne.value:
        assert hip.hipGraphGetEdges(g, fr, to, C.byref(ne)) == 0
    edges = [[ids[int(fr[i])], ids[int(to[i])]] for i in range(ne.value)]
    return {"nodes": out, "edges": edges}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("prefix")
    ap.add_argument("--memset-nodes", action="store_true")
    ap.add_argument("--config", default="dp", choices=["dp", "single"])
    ap.add_argument("--poe", default="label", choices=["label", "paired", "cluster"])
    ap.add_argument("--genes", type=int, default=600)
    ap.add_argument("--cells", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=128)
    a = ap.parse_args()
    if a.memset_nodes:
        os.environ["SPV_MEMSET_NODES"] = "1"   # read once, at the first fill: set before the library is used
    os.environ["SPV_GRAPH_KEEP"] = "1"
    import numpy as np
    import torch
    from spvipes_amd.data import MinibatchSampler, make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    dev = torch.device("cuda:0")
    groups = [make_synthetic_group(g, a.cells, a.genes, dev) for g in range(2)]
    torch.manual_seed(0)
    kw = dict(n_hidden=a.hidden, n_dimensions_shared=10, n_dimensions_private=5, precision="bf16")
    if a.poe == "label":
        module = spVIPESmodule({0: a.genes, 1: a.genes}, use_labels=True, **kw).to(dev)
        trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups], overlap_allreduce=(a.config == "dp"))
    else:
        import scipy.sparse as sp
        rng = np.random.default_rng(0)
        n = a.cells
        i, j, v = np.repeat(np.arange(n), 3), rng.integers(0, n, size=3 * n), rng.random(3 * n).astype(np.float32) + 0.1
        P = sp.coo_matrix((v, (i, j)), shape=(n, n)).tocsr()
        module = spVIPESmodule({0: a.genes, 1: a.genes}, use_labels=False, transport_plan=P, pair_data=(a.poe == "paired"), **kw).to(dev)
        trainer = Trainer(module, [g.counts for g in groups], components=[g.labels for g in groups] if a.poe == "cluster" else None,
                          overlap_allreduce=(a.config == "dp"))
    module.train()
    sampler = MinibatchSampler([a.cells, a.cells], a.batch, dev, seed=0)
    rows = next(iter(sampler.epoch()))
    trainer.capture(rows, warmup=2)
    for _ in range(3):
        lo = trainer.step(rows, kl_weight=1.0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(trainer.fp.grad).all()) and bool(torch.isfinite(lo.loss))
    hip = C.CDLL("libamdhip64.so")
    for k, g in ((1, trainer.graph), (2, trainer.graph2)):
        if g is None:
            continue
        raw = g.raw_cuda_graph()
        info = read_graph(hip, raw)
        info["what"] = f"graph {k} of a {a.config} step, {a.poe} PoE, memset_nodes={bool(a.memset_nodes)}, B {a.batch} x G {a.genes}, H {a.hidden}"
        with open(f"{a.prefix}_graph{k}.json", "w") as f:
            json.dump(info, f, indent=0)
        rc = hip.hipGraphDebugDotPrint(C.c_void_p(raw), f"{a.prefix}_graph{k}.dot".encode(), C.c_uint(1))
        kinds = {}
        for nd in info["nodes"]:
            kinds[nd["type"]] = kinds.get(nd["type"], 0) + 1
        print(f"graph {k}: {len(info['nodes'])} nodes {kinds}, {len(info['edges'])} edges; hipGraphDebugDotPrint rc {rc}", flush=True)


if __name__ == "__main__":
    main()
