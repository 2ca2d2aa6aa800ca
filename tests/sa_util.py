"""Test helper: suffix array of a prepared text by numpy prefix doubling (virtual smallest end
marker), the order the reference's test sorter produces (bwt_qsufsort.c:176-240)."""
import numpy as np


def suffix_array(text):
    t = np.asarray(text, dtype=np.int64)
    n = len(t)
    rank = t + 1
    k = 1
    while True:
        r2 = np.zeros(n, dtype=np.int64)
        r2[:n - k] = rank[k:]
        order = np.lexsort((r2, rank))
        kr, k2 = rank[order], r2[order]
        change = np.ones(n, dtype=np.int64)
        change[1:] = (kr[1:] != kr[:-1]) | (k2[1:] != k2[:-1])
        newrank = np.empty(n, dtype=np.int64)
        newrank[order] = np.cumsum(change)
        rank = newrank
        if rank.max() == n:
            return order.astype(np.int64)
        k *= 2
