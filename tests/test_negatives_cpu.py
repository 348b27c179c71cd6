"""dataset.SampleUserNegatives restated (oracle/gorse_oracle.c orc_sample_user_negatives) against a pure-Python transcription of
RandomGenerator.SampleInt32 (common/util/random.go:108-132) on the same per-user Philox stream: the rejection branch (draw
order, distinctness, both exclusion sets) and the enumerate-when-dense branch (random.go:115-121: all remaining items,
ascending), duplicates inside the feedback rows counting once (mapset semantics)."""
import numpy as np

from oracle import oracle as orc

NEG_STREAM = 0x6E6567
M32 = 0xFFFFFFFF


class Philox:  # Philox4x32-10 keyed like csrc/common.hpp (seed -> key, (sample lo, sample hi, block, epoch) -> counter)
    def __init__(self, seed, epoch, sample):
        self.c = [sample & M32, (sample >> 32) & M32, 0, epoch & M32]
        self.k = [seed & M32, (seed >> 32) & M32]
        self.buf, self.pos = [], 4

    def block(self):
        a0, a1, a2, a3 = self.c
        x0, x1 = self.k
        for _ in range(10):
            p0, p1 = 0xD2511F53 * a0, 0xCD9E8D57 * a2
            a0, a1, a2, a3 = ((p1 >> 32) ^ a1 ^ x0) & M32, p1 & M32, ((p0 >> 32) ^ a3 ^ x1) & M32, p0 & M32
            x0, x1 = (x0 + 0x9E3779B9) & M32, (x1 + 0xBB67AE85) & M32
        self.buf, self.pos = [a0, a1, a2, a3], 0
        self.c[2] = (self.c[2] + 1) & M32

    def int31(self):
        if self.pos == 4:
            self.block()
        v = self.buf[self.pos]
        self.pos += 1
        return v >> 1

    def int31n(self, n):  # Go math/rand (*Rand).Int31n
        if n & (n - 1) == 0:
            return self.int31() & (n - 1)
        mx = (1 << 31) - 1 - (1 << 31) % n
        v = self.int31()
        while v > mx:
            v = self.int31()
        return v % n


def sample_int32(rng, low, high, n, *exclude):  # random.go:108-132
    ex = set().union(*exclude)
    out = []
    if n >= (high - low) - len(ex):
        for i in range(low, high):
            if i not in ex:
                out.append(i)
                ex.add(i)
    else:
        while len(out) < n:
            v = rng.int31n(high - low) + low
            if v not in ex:
                out.append(v)
                ex.add(v)
    return out


def csr(rows):
    ptr = np.zeros(len(rows) + 1, np.int64)
    np.cumsum([len(r) for r in rows], out=ptr[1:])
    idx = np.array([x for r in rows for x in r], np.int32)
    return ptr, idx


def test_oracle_negatives_equal_the_transcription():
    o = orc.Oracle()
    rng = np.random.default_rng(8)
    for I, n in ((50, 10), (37, 30), (64, 20), (12, 12), (1000, 99)):
        U = 40
        train = [list(rng.integers(0, I, int(rng.integers(0, max(2, I // 2))))) for _ in range(U)]  # duplicates happen
        test = [list(rng.integers(0, I, int(rng.integers(0, 4)))) for _ in range(U)]
        train[3], test[3] = list(range(I)), []  # nothing left for this user
        test[5] = list(train[5][:2]) + test[5]  # test items that are also train items count once
        tp, ti = csr(train)
        sp, si = csr(test)
        neg, ln = o.sample_user_negatives(U, I, tp, ti, sp, si, n, seed=0)
        dense = sparse = 0
        for u in range(U):
            exp = sample_int32(Philox(0, NEG_STREAM, u), 0, I, n, set(test[u]), set(train[u]))
            assert ln[u] == len(exp) and list(neg[u, :ln[u]]) == exp, (I, n, u)
            assert (neg[u, ln[u]:] == -1).all()
            dense += n >= I - len(set(test[u]) | set(train[u]))
            sparse += n < I - len(set(test[u]) | set(train[u]))
        if (I, n) in ((37, 30), (12, 12)):
            assert dense > 0
        if (I, n) != (12, 12):
            assert sparse > 0
