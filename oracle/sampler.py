"""BPR triplet samplers, restated on the CPU.  TEST INFRASTRUCTURE.

RefSampler   -- bit-exact restatement of elliot/dataset/samplers/custom_sampler.py:14-46
                (global legacy np.random MT19937 seeded 42, np.random.randint's masked rejection
                over 32-bit outputs, draw order u -> position of i -> j repeated while j in pos(u)).
                Pinned against the reference's own class in oracle/gen_golden.py.
philox_sample -- restatement of the DEVICE sampler's algorithm (elliot_amd/csrc/el_bpr.hip:
                k_bpr_sample): same distribution as the reference, Philox4x32-10 bit stream.
"""
import numpy as np

# ------------------------------------------------------------------------------------------
# MT19937 (Matsumoto & Nishimura 1998), as seeded by np.random.seed(int)  (init_genrand)
# ------------------------------------------------------------------------------------------
class MT19937:
    N, M = 624, 397

    def __init__(self, seed):
        mt = np.empty(self.N, dtype=np.uint64)
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, self.N):
            mt[i] = (1812433253 * (int(mt[i - 1]) ^ (int(mt[i - 1]) >> 30)) + i) & 0xFFFFFFFF
        self.mt = mt.astype(np.uint32)
        self.pos = self.N
        self.buf = None

    def _twist(self):
        mt = [int(x) for x in self.mt]
        N, M = self.N, self.M
        for kk in range(N):
            y = (mt[kk] & 0x80000000) | (mt[(kk + 1) % N] & 0x7FFFFFFF)
            v = mt[(kk + M) % N] ^ (y >> 1)
            if y & 1:
                v ^= 0x9908B0DF
            mt[kk] = v
        self.mt = np.array(mt, dtype=np.uint32)
        y = self.mt.astype(np.uint64)
        y ^= y >> np.uint64(11)
        y ^= (y << np.uint64(7)) & np.uint64(0x9D2C5680)
        y ^= (y << np.uint64(15)) & np.uint64(0xEFC60000)
        y ^= y >> np.uint64(18)
        self.buf = [int(v) & 0xFFFFFFFF for v in y]
        self.pos = 0

    def next_u32(self):
        if self.pos >= self.N:
            self._twist()
        v = self.buf[self.pos]
        self.pos += 1
        return v

    def randint(self, n):
        """np.random.randint(n) (legacy RandomState, default dtype): 0 draws when n == 1, otherwise
        masked rejection over successive 32-bit outputs (numpy/random/_bounded_integers: range < 2^32)."""
        rng = n - 1
        if rng == 0:
            return 0
        mask = rng
        mask |= mask >> 1
        mask |= mask >> 2
        mask |= mask >> 4
        mask |= mask >> 8
        mask |= mask >> 16
        while True:
            v = self.next_u32() & mask
            if v <= rng:
                return v


class RefSampler:
    """custom_sampler.Sampler (custom_sampler.py:14-46).

    ui_lists[u] must be the reference's per-user list ``list(set(i_train_dict[u]))`` IN ITS ORDER
    (custom_sampler.py:21: CPython set order, not sorted)."""

    def __init__(self, ui_lists, n_items, seed=42):
        self.rng = MT19937(seed)               # np.random.seed(42), custom_sampler.py:15
        self.ui = [list(map(int, l)) for l in ui_lists]
        self.ui_set = [set(l) for l in self.ui]
        self.n_users = len(self.ui)
        self.n_items = int(n_items)

    def sample(self):
        r = self.rng.randint
        u = r(self.n_users)                    # :32
        ui = self.ui[u]
        lui = len(ui)
        if lui == self.n_items:                # :35-36 (recursive result is discarded by the reference)
            self.sample()
        i = ui[r(lui)]                         # :37
        j = r(self.n_items)                    # :39
        while j in self.ui_set[u]:             # :40-41
            j = r(self.n_items)
        return u, i, j

    def step(self, events, batch_size):        # :44-46
        for start in range(0, events, batch_size):
            n = min(start + batch_size, events) - start
            t = np.array([self.sample() for _ in range(n)], dtype=np.int64)
            yield t[:, 0:1], t[:, 1:2], t[:, 2:3]


class RefPointwiseSampler:
    """pointwise_pos_neg_sampler.Sampler (pointwise_pos_neg_sampler.py:14-50): NumPy's legacy MT19937 (`np.random.seed(42)`,
    restated by MT19937 above) interleaved with Python's own generator (`random.seed(42)`, `random.getrandbits(1)`: the
    standard library's -- it is CPython, not the reference, so it is used as is).  ui_lists as for RefSampler."""

    def __init__(self, ui_lists, n_items):
        import random
        self.rng = MT19937(42)                 # :16
        self.py = random.Random(42)            # :17
        self.ui = [list(map(int, l)) for l in ui_lists]
        self.n_users = len(self.ui)
        self.n_items = int(n_items)

    def sample(self):
        r = self.rng.randint
        u = r(self.n_users)                    # :34
        ui = self.ui[u]
        lui = len(ui)
        if lui == self.n_items:                # :37-38
            self.sample()
        b = self.py.getrandbits(1)             # :39
        if b:
            i = ui[r(lui)]                     # :41
        else:
            i = r(self.n_items)                # :43
            while i in ui:                     # :44-45
                i = r(self.n_items)
        return u, i, b

    def step(self, events, batch_size):        # :48-50
        for start in range(0, events, batch_size):
            n = min(start + batch_size, events) - start
            t = np.array([self.sample() for _ in range(n)], dtype=np.int64)
            yield t[:, 0], t[:, 1], t[:, 2]


# ------------------------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al. 2011) -- the device sampler's bit stream
# ------------------------------------------------------------------------------------------
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> 32, p0 & 0xFFFFFFFF, p1 >> 32, p1 & 0xFFFFFFFF
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & 0xFFFFFFFF, lo1, (hi0 ^ c3 ^ k1) & 0xFFFFFFFF, lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


class _PhiloxStream:
    def __init__(self, n, seed):
        self.n_lo, self.n_hi = n & 0xFFFFFFFF, (n >> 32) & 0xFFFFFFFF
        self.k0, self.k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
        self.a = 0
        self.w = []

    def next(self):
        if not self.w:
            self.w = list(philox4x32_10(self.n_lo, self.n_hi, self.a, 0, self.k0, self.k1))
            self.a += 1
        return self.w.pop(0)

    def bounded(self, n):
        m = n - 1
        m |= m >> 1
        m |= m >> 2
        m |= m >> 4
        m |= m >> 8
        m |= m >> 16
        while True:
            v = self.next() & m
            if v < n:
                return v


def philox_sample(indptr, indices, n_users, n_items, seed, first_sample, n, item_lo=0, item_hi=None):
    """Same distribution as custom_sampler.py:31-42; stream layout of k_bpr_sample."""
    if item_hi is None:
        item_hi = n_items
    out = np.empty((n, 3), dtype=np.int32)
    rows = [set(map(int, indices[indptr[u]:indptr[u + 1]])) for u in range(n_users)]
    for t in range(n):
        ps = _PhiloxStream(first_sample + t, seed)
        while True:
            u = ps.bounded(n_users)
            r0, r1 = int(indptr[u]), int(indptr[u + 1])
            lui = r1 - r0
            if lui <= 0 or lui >= n_items:
                continue
            i = int(indices[r0 + ps.bounded(lui)])
            j = -1
            for _ in range(4096):
                c = item_lo + ps.bounded(item_hi - item_lo)
                if c not in rows[u]:
                    j = c
                    break
            if j < 0:
                continue
            out[t] = (u, i, j)
            break
    return out[:, 0].copy(), out[:, 1].copy(), out[:, 2].copy()
