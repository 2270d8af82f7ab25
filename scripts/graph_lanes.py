#!/usr/bin/env python
"""Which lane (internal stream of the hipGraph executor, at most DEBUG_HIP_FORCE_GRAPH_QUEUES = 4) every node of the
captured step runs on.  Input: the DOT file ROCm writes with DEBUG_HIP_GRAPH_DOT_PRINT=1 (graph_<pid>_dot_print_*):
it carries the executor's StreamId per node.  Prints the nodes in capture order with their lane and dependencies, and
the places where a node waits behind a weight-gradient kernel of the same lane that it does not depend on."""
import re
import sys


def short(name):
    m = re.match(r"_Z\d+([a-z_0-9]+?)I", name) or re.match(r"_Z\d+([a-z_0-9]+)", name)
    n = m.group(1) if m else name[:24]
    t = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)", name)
    if t and "gemm" in n:
        n += "<%s,%s,%s>" % t.groups()
    return n


def main():
    txt = open(sys.argv[1]).read()
    nodes = {}
    for m in re.finditer(r'"graph_1_node_(\d+)"\[[^\]]*?label="(\d+)\n([^\n]+)\nStreamId:(\d+)', txt):
        nodes[int(m.group(1))] = (short(m.group(3)), int(m.group(4)))
    edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_1_node_(\d+)"\s*->\s*"graph_1_node_(\d+)"', txt)]
    parents = {}
    for a, b in edges:
        parents.setdefault(b, []).append(a)
    print("%d nodes, %d edges, lanes used: %s" % (len(nodes), len(edges), sorted({l for _, l in nodes.values()})))
    last_on_lane = {}
    heavy = ("wgrad", "wg9")
    for i in sorted(nodes):
        name, lane = nodes[i]
        prev = last_on_lane.get(lane)
        note = ""
        if prev is not None and prev not in _ancestors(i, parents) and any(h in nodes[prev][0] for h in heavy) \
                and not any(h in name for h in heavy + ("finalize",)):
            note = "   <-- queued behind unrelated %s (node %d)" % (nodes[prev][0], prev)
        print("%4d lane %d  %-34s parents %s%s" % (i, lane, name, sorted(parents.get(i, [])), note))
        last_on_lane[lane] = i


_anc_cache = {}


def _ancestors(i, parents):
    if i in _anc_cache:
        return _anc_cache[i]
    out = set()
    for p in parents.get(i, []):
        out.add(p)
        out |= _ancestors(p, parents)
    _anc_cache[i] = out
    return out


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    main()
