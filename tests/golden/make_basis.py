#!/usr/bin/env python3
"""Initial bases of our own for the golden cases with --basis-name (tests/golden/long_*.fco).

The reference's append_edge() never checks MAXEDGES (codec/wfalib.c:253-273): the sixth and later edges of a
label run on into the row of the next label / state, whose own edges are then sorted in among them
(fa_wfa_append_edge, DevFrame.bx).  The reference's data/medium.fco and data/large.fco rely on that (up to 8
edges per label in the file, lists of up to 33 entries in memory).  These generated bases do the same on
purpose, so that the GPU box -- which has no /root/reference/data -- can pin the behaviour against streams of
the real reference:

    long_a.fco   20 states, 1 .. 8 edges per label, lines of both labels interleaved
    long_b.fco   60 states, <= 5 edges per label (no list runs on; more states than DevFrame's own rows hold)
    long_c.fco   40 states, every third with up to 12 edges per label: lists that run across several rows

ASCII format of input/read.c:219-340: magic, number of states N (state 0, the constant, is implicit), N
use-as-domain flags, N final distributions, then per state its number, `label domain weight' lines and -1.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def make_basis(path, n, seed, max_edges, aux=(), dense_every=1):
    rng = np.random.default_rng(seed)
    lines = ["Fiasco", str(n), ""]
    flags = [0 if s in aux else 1 for s in range(1, n + 1)]
    lines.append(" ".join(map(str, flags)))
    lines.append("")
    final = [0.5, 0.5] + [float(v) for v in rng.uniform(-0.5, 0.5, n - 2)]
    lines.append(" ".join("%.6e" % v for v in final))
    lines.append("")
    for s in range(1, n + 1):
        lines.append("%d" % s)
        if s == 1:          # x and y ramps like the built-in small basis: something sensible to combine
            edges = [(0, 2, 0.5), (1, 2, 0.5), (1, 0, 0.5)]
        elif s == 2:
            edges = [(0, 1, 1.0), (1, 1, 1.0)]
        else:
            edges = []
            for label in (0, 1):
                # the last states keep short lists: every list must end inside the rows of the basis
                # (and with dense_every > 1 only every dense_every-th state has long lists: the short rows in
                # between take up what runs over)
                short = s > n - 3 or s % dense_every != 0
                k = int(rng.integers(1, (2 if short else max_edges) + 1))
                doms = rng.integers(0, n + 1, k)
                w = rng.uniform(-1.0, 1.0, k)
                w *= 0.9 / max(np.abs(w).sum(), 0.9)            # a contraction: images stay bounded
                edges += [(label, int(d), float(x)) for d, x in zip(doms, w)]
            order = rng.permutation(len(edges))
            edges = [edges[i] for i in order]
        for label, dom, wt in edges:
            lines.append("%d %d %.6e" % (label, dom, wt))
        lines.append("-1")
    lines.append("-1")
    open(path, "w").write("\n".join(lines) + "\n")


def main():
    make_basis(os.path.join(HERE, "long_a.fco"), 20, 101, 8, aux=(7,))
    make_basis(os.path.join(HERE, "long_b.fco"), 60, 102, 5, aux=(11, 40))
    make_basis(os.path.join(HERE, "long_c.fco"), 40, 103, 12, dense_every=3)


if __name__ == "__main__":
    main()
