"""CPU: the compare-exchange networks of the ORCA neighbour selection (csrc/orca_sortnet.h) are what tools/gen_sortnet.py generates,
and they are correct: the 8-key network sorts every 0-1 input, the merge leaves the ten smallest of a sorted 10-list and a sorted
8-chunk in order (0-1 principle for merging networks: every pair of SORTED 0-1 inputs), and -- end to end, on random 64-bit keys -- chunked
sort + merge + left-over insertion yields the same ten keys as one-at-a-time insertion (the kernel's previous form)."""
import importlib.util
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_sortnet", os.path.join(ROOT, "tools", "gen_sortnet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_networks_are_correct_and_the_header_is_the_generated_one():
    g = _gen()
    g.check_sort8()
    comps, out = g.merge_network(10, 8, 10)
    g.check_merge(comps, out, 10, 8, 10)
    assert out == [("L", i) for i in range(10)]
    header = open(os.path.join(ROOT, "social_navigation_pyenvs_amd", "csrc", "orca_sortnet.h")).read()
    in_header = re.findall(r"CE\((\w)\[(\d+)\], (\w)\[(\d+)\]\)", header)
    want = [("C", str(a), "C", str(b)) for a, b in g.SORT8] + [(x[0], str(x[1]), y[0], str(y[1])) for x, y in comps]
    assert in_header == want


def test_chunked_selection_equals_insertion_on_random_keys():
    g = _gen()
    comps, _ = g.merge_network(10, 8, 10)
    rng = np.random.default_rng(3)
    SENT = np.uint64(0x7F7FFFFFFFFFFFFF)
    for rows in (8, 9, 16, 25, 26, 31, 40, 64):
        for _ in range(200):
            keys = rng.integers(0, 2 ** 62, rows, dtype=np.uint64)
            keys[rng.uniform(size=rows) < 0.3] = SENT                      # out of range / myself
            if rng.uniform() < 0.3:
                keys[rng.integers(0, rows)] = keys[rng.integers(0, rows)]  # a tie
            want = np.sort(np.concatenate([keys, np.full(10, SENT)]))[:10]
            C = {("C", i): keys[i] for i in range(8)}
            for a, b in g.SORT8:
                if C[("C", a)] > C[("C", b)]:
                    C[("C", a)], C[("C", b)] = C[("C", b)], C[("C", a)]
            L = [C[("C", i)] for i in range(8)] + [SENT, SENT]
            b0 = 8
            while b0 + 8 <= rows:
                v = {("C", i): keys[b0 + i] for i in range(8)}
                for a, b in g.SORT8:
                    if v[("C", a)] > v[("C", b)]:
                        v[("C", a)], v[("C", b)] = v[("C", b)], v[("C", a)]
                v.update({("L", i): L[i] for i in range(10)})
                for x, y in comps:
                    if v[x] > v[y]:
                        v[x], v[y] = v[y], v[x]
                L = [v[("L", i)] for i in range(10)]
                b0 += 8
            for k in keys[b0:]:                                             # the rows left over: insertion
                x = k
                for s in range(10):
                    lo, hi = min(L[s], x), max(L[s], x)
                    L[s], x = lo, hi
            np.testing.assert_array_equal(np.array(L, dtype=np.uint64), want)
