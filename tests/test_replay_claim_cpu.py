"""The claim behind topk_tie_replay_kernel (gorse_amd/csrc/topk_mfma.hip), checked on the CPU with Go's container/heap
rules restated in pure Python (same rules as gorse_amd/csrc/goheap.hpp / common/heap/pq.go):

  Bruteforce's queue (bruteforce.go:46-53: Push, then Pop once it holds more than k) ends in the same heap ARRAY
  when only a superset of the accepted vectors is pushed literally and every run of `gap` strictly-rejected vectors in
  between is replaced by T^gap, T = "push +inf, pop" -- T being evaluated by cycle detection, because with equal
  weights in the heap it is a permutation with a short period rather than the identity.
"""
import heapq

import numpy as np

INF = float("inf")


class GoMaxHeap:  # heap.NewPriorityQueue(true): less(i, j) = w[i] > w[j]
    def __init__(self):
        self.v, self.w = [], []

    def less(self, i, j):
        return self.w[i] > self.w[j]

    def swap(self, i, j):
        self.v[i], self.v[j] = self.v[j], self.v[i]
        self.w[i], self.w[j] = self.w[j], self.w[i]

    def up(self, j):
        while True:
            i = (j - 1) // 2
            if j == 0 or i == j or not self.less(j, i):
                break
            self.swap(i, j)
            j = i

    def down(self, i, n):
        while True:
            j1 = 2 * i + 1
            if j1 >= n:
                break
            j = j1
            if j1 + 1 < n and self.less(j1 + 1, j1):
                j = j1 + 1
            if not self.less(j, i):
                break
            self.swap(i, j)
            i = j

    def push(self, v, w):
        self.v.append(v)
        self.w.append(w)
        self.up(len(self.v) - 1)

    def pop(self):
        n = len(self.v) - 1
        self.swap(0, n)
        self.down(0, n)
        self.v.pop()
        self.w.pop()


def literal(dist, k):
    h = GoMaxHeap()
    for i, x in enumerate(dist):
        h.push(i, x)
        if len(h.v) > k:
            h.pop()
    return list(h.v)


def t_pow(h, gap):
    """exactly the control flow of t_pow in topk_tie_replay_kernel"""
    def apply():
        h.push(-1, INF)
        h.pop()
    steps = 0
    while steps < gap and steps < 16:
        snap = list(h.v)
        apply()
        steps += 1
        if h.v == snap:
            return True
    if steps == gap:
        return True
    snap = list(h.v)
    period, closed = 0, False
    while period < 64:
        apply()
        period += 1
        steps += 1
        if steps == gap:
            return True
        if h.v == snap:
            closed = True
            break
    if not closed:
        return False
    for _ in range((gap - steps) % period):
        apply()
    return True


def replay(dist, k, recorded):
    """the control flow of topk_tie_replay_kernel: unrecorded vectors and recorded ones that are strictly worse than
    the root are T applications, batched until the next vector the heap really takes"""
    h = GoMaxHeap()
    prev, pend = -1, 0
    for i in list(recorded) + [len(dist)]:
        gap = i - prev - 1
        if gap > 0:
            assert len(h.v) == k
        pend += gap
        prev = i
        if i == len(dist):
            break
        if len(h.v) == k and dist[i] > h.w[0]:
            pend += 1
            continue
        if pend > 0:
            if not t_pow(h, pend):
                return None
            pend = 0
        h.push(i, dist[i])
        if len(h.v) > k:
            h.pop()
    if pend > 0 and not t_pow(h, pend):
        return None
    return list(h.v)


def test_replay_with_gaps_equals_the_literal_heap():
    rng = np.random.default_rng(1)
    undecided = 0
    for trial in range(250):
        n = int(rng.integers(50, 3000))
        k = int(rng.integers(1, 60))
        levels = int(rng.integers(2, 80))  # few distinct weights: ties everywhere
        dist = [float(x) for x in rng.integers(0, levels, n)]
        # superset of the accepted pushes: d_i <= running k-th smallest, plus a few strictly rejected ones
        rec, best = [], []
        for i, x in enumerate(dist):
            if len(best) < k or x <= -best[0]:
                rec.append(i)
            elif rng.random() < 0.03:
                rec.append(i)
            if len(best) < k:
                heapq.heappush(best, -x)
            elif x < -best[0]:
                heapq.heapreplace(best, -x)
        got = replay(dist, k, rec)
        if got is None:
            undecided += 1
            continue
        assert got == literal(dist, k), (trial, n, k, levels)
    assert undecided <= 2  # the kernel hands such queries to the literal scan


def test_t_is_a_short_cycle():
    # pre-period and period of T stay far below the bounds the kernel searches (16 and 64)
    rng = np.random.default_rng(5)
    worst_pre = worst_per = 0
    for trial in range(400):
        k = int(rng.integers(1, 200))
        levels = int(rng.integers(1, 12))
        h = GoMaxHeap()
        for i in range(k + int(rng.integers(0, 200))):
            h.push(i, float(rng.integers(0, levels)))
            if len(h.v) > k:
                h.pop()
        if len(h.v) < k:
            continue
        seen, t = {}, 0
        while tuple(h.v) not in seen:
            seen[tuple(h.v)] = t
            h.push(-1, INF)
            h.pop()
            t += 1
        worst_pre = max(worst_pre, seen[tuple(h.v)])
        worst_per = max(worst_per, t - seen[tuple(h.v)])
    assert worst_pre <= 12 and worst_per <= 32, (worst_pre, worst_per)


def t_is_identity(h):
    """the criterion of t_is_identity in topk_tie_replay_kernel: T = push(+inf), pop leaves the array of a FULL heap as it is"""
    n = len(h.v)
    if n < 1:
        return False
    child = (n - 1) // 2
    e = h.w[child]
    if not e < INF:
        return False
    while child > 0:
        parent = (child - 1) // 2
        wp = h.w[parent]
        if not wp > e or not wp < INF:
            return False
        if child % 2 == 0 and not wp > h.w[child - 1]:
            return False
        child = parent
    return True


def test_identity_criterion_of_T_is_sound():
    """Whenever the criterion holds, the literal T leaves values and weights where they were; and it is not vacuous: it
    holds for most heaps without ties, fails for some with ties, and among the failures T really does move elements."""
    rng = np.random.default_rng(3)
    held = moved_when_rejected = rejected = 0
    for trial in range(4000):
        k = int(rng.integers(1, 40))
        levels = int(rng.integers(1, 6))  # few distinct weights: many ties
        h = GoMaxHeap()
        for i in range(k + int(rng.integers(0, 30))):
            h.push(i, float(rng.integers(0, levels if trial % 2 else 1000)))
            if len(h.v) > k:
                h.pop()
        if len(h.v) < k:
            continue
        v0, w0 = list(h.v), list(h.w)
        ok = t_is_identity(h)
        h.push(-1, INF)
        h.pop()
        if ok:
            held += 1
            assert h.v == v0 and h.w == w0
        else:
            rejected += 1
            moved_when_rejected += h.v != v0
    assert held > 1000 and rejected > 100 and moved_when_rejected > 20
