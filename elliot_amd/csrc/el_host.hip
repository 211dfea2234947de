// Host-side data plane: the two sequential integer loops between a ratings file and the CSR the kernels
// consume, in C instead of per-user / per-row Python.  No HIP call in this file; the functions work without a GPU.
//
//   el_host_split_flags    per-user train / test flags exactly as elliot/splitter/base_splitter.py:256-274 draws them
//                          (np.random.seed(seed) once -- process_splitting :73 --, then per user, in groupby order, a list
//                          [0]*train + [1]*test shuffled by the legacy np.random.shuffle: Fisher-Yates from the top,
//                          j = masked-rejection draw from the MT19937 32-bit stream)
//   el_host_negative_sample  per-user uniform negatives of the evaluation protocol exactly as NegativeSampler.
//                          sample_by_random_uniform draws them (elliot/negative_sampling/negative_sampling.py:95-105):
//                          random.sample(range(n_candidates), num) on the `random` module's MT19937 stream, candidates = the
//                          items in neither train nor test, ascending
//   el_host_pyset_order    iteration order of a CPython set after inserting the given non-negative ints in order: the
//                          private item ids of the reference are the positions in `list({k for a in train_dict.values() for k
//                          in a.keys()})` (elliot/dataset/dataset.py:202, :211-214)
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "el_common.h"

namespace {

// MT19937, the generator behind the legacy numpy.random module (init_genrand seeding = np.random.seed(int))
struct Mt {
    uint32_t key[624];
    int pos;
    explicit Mt(uint32_t seed) {
        for (int i = 0; i < 624; ++i) {
            key[i] = seed;
            seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    void gen() {
        const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
        int i = 0;
        for (; i < 624 - 397; ++i) {
            const uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        }
        for (; i < 623; ++i) {
            const uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        }
        const uint32_t y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        pos = 0;
    }
    uint32_t next() {
        if (pos == 624) gen();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    // numpy random_interval / legacy rk_interval: uniform on [0, max], max <= 0xffffffff
    uint32_t interval(uint32_t max) {
        if (max == 0) return 0;
        uint32_t mask = max;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        uint32_t v;
        while ((v = (next() & mask)) > max) {
        }
        return v;
    }
};

}  // namespace

static int split_flags_on(Mt& mt, const int64_t* seg_len, int64_t n_seg, int mode, double param, int32_t n_folds, int8_t* flags) {
    int64_t off = 0;
    for (int32_t fold = 0; fold < n_folds; ++fold)          // base_splitter.py:266-267: folds outside, users inside, one stream
    for (int64_t s = 0; s < n_seg; ++s) {
        const int64_t n = seg_len[s];
        EL_REQUIRE(n >= 0 && n <= 0xffffffffll, "el_host_split_flags: a segment longer than 2^32 - 1 rows");
        int64_t train;
        if (mode == 0)
            train = (int64_t)floor((double)n * (1.0 - param));         // base_splitter.py:257
        else
            train = n - (int64_t)param;                                // :277-279 (leave-n-out)
        if (train < 0) train = 0;
        if (train > n) train = n;
        int8_t* x = flags + off;
        memset(x, 0, (size_t)train);
        memset(x + train, 1, (size_t)(n - train));
        for (int64_t i = n - 1; i >= 1; --i) {                         // legacy shuffle of a sequence
            const uint32_t j = mt.interval((uint32_t)i);
            const int8_t t = x[i];
            x[i] = x[j];
            x[j] = t;
        }
        off += n;
    }
    return 0;
}

extern "C" int el_host_split_flags(const int64_t* seg_len, int64_t n_seg, int mode, double param, uint32_t seed, int32_t n_folds, int8_t* flags) {
    EL_REQUIRE(seg_len != nullptr && flags != nullptr && n_seg >= 0, "el_host_split_flags: null argument");
    EL_REQUIRE(mode == 0 || mode == 1, "el_host_split_flags: mode 0 = random_subsampling(test_ratio), 1 = leave_n_out(n)");
    EL_REQUIRE(n_folds >= 1, "el_host_split_flags: n_folds >= 1");
    Mt mt(seed);
    return split_flags_on(mt, seg_len, n_seg, mode, param, n_folds, flags);
}

// The same draws on a generator state the CALLER carries (624 key words + position, as np.random.get_state()[1:3]): the reference
// seeds np.random ONCE per Splitter.process_splitting (base_splitter.py:73) and every level of the train / validation / test
// hierarchy -- the test split, then the validation split of each test fold's train part (:86-98) -- continues that one stream.
extern "C" int el_host_split_flags_state(const int64_t* seg_len, int64_t n_seg, int mode, double param, uint32_t* np_state625,
                                         int32_t n_folds, int8_t* flags) {
    EL_REQUIRE(seg_len != nullptr && flags != nullptr && n_seg >= 0 && np_state625 != nullptr, "el_host_split_flags_state: null argument");
    EL_REQUIRE(mode == 0 || mode == 1, "el_host_split_flags_state: mode 0 = random_subsampling(test_ratio), 1 = leave_n_out(n)");
    EL_REQUIRE(n_folds >= 1, "el_host_split_flags_state: n_folds >= 1");
    EL_REQUIRE(np_state625[624] <= 624u, "el_host_split_flags_state: position word out of range");
    Mt mt(0u);
    memcpy(mt.key, np_state625, sizeof(mt.key));
    mt.pos = (int)np_state625[624];
    const int rc = split_flags_on(mt, seg_len, n_seg, mode, param, n_folds, flags);
    memcpy(np_state625, mt.key, sizeof(mt.key));
    np_state625[624] = (uint32_t)mt.pos;
    return rc;
}

// CPython set (Objects/setobject.c, 3.7 - 3.12): open addressing, LINEAR_PROBES = 9 slots after the home slot when they fit
// below the end of the table, then i = 5 i + 1 + perturb with perturb >>= 5; growth when fill * 5 >= mask * 3 to the first
// power of two above used * 4 (used * 2 beyond 50 000); a resize re-inserts in table order; iteration is table order.
// hash(int) = the int for 0 <= x < 2^61 - 1.  `out` receives the distinct keys in iteration order; *n_out their number.
extern "C" int el_host_pyset_order(const int64_t* keys, int64_t n, int64_t* out, int64_t* n_out) {
    EL_REQUIRE(n_out != nullptr && (n == 0 || (keys != nullptr && out != nullptr)), "el_host_pyset_order: null argument");
    const int LINEAR_PROBES = 9;
    size_t mask = 7, fill = 0;
    int64_t* tab = (int64_t*)malloc((mask + 1) * sizeof(int64_t));   // key + 1, 0 = empty
    EL_REQUIRE(tab != nullptr, "el_host_pyset_order: out of memory");
    memset(tab, 0, (mask + 1) * sizeof(int64_t));
    auto insert_clean = [](int64_t* t, size_t m, int64_t key) {
        size_t perturb = (size_t)key, i = (size_t)key & m;
        for (;;) {
            if (t[i] == 0) {
                t[i] = key + 1;
                return;
            }
            if (i + LINEAR_PROBES <= m) {
                for (int j = 1; j <= LINEAR_PROBES; ++j)
                    if (t[i + j] == 0) {
                        t[i + j] = key + 1;
                        return;
                    }
            }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & m;
        }
    };
    for (int64_t e = 0; e < n; ++e) {
        const int64_t key = keys[e];
        if (key < 0 || key >= 2305843009213693951ll) {
            free(tab);
            EL_REQUIRE(false, "el_host_pyset_order: keys must be ints in [0, 2^61 - 1)");
        }
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        bool placed = false, present = false;
        while (!placed && !present) {
            const int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
            for (int j = 0; j <= probes; ++j) {
                const int64_t v = tab[i + j];
                if (v == 0) {
                    tab[i + j] = key + 1;
                    placed = true;
                    break;
                }
                if (v == key + 1) {
                    present = true;
                    break;
                }
            }
            if (!placed && !present) {
                perturb >>= 5;
                i = (i * 5 + 1 + perturb) & mask;
            }
        }
        if (!placed) continue;
        ++fill;
        if (fill * 5 < mask * 3) continue;
        const size_t minused = fill > 50000 ? fill * 2 : fill * 4;
        size_t newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        int64_t* nt = (int64_t*)malloc(newsize * sizeof(int64_t));
        if (nt == nullptr) {
            free(tab);
            EL_REQUIRE(false, "el_host_pyset_order: out of memory");
        }
        memset(nt, 0, newsize * sizeof(int64_t));
        for (size_t s = 0; s <= mask; ++s)
            if (tab[s] != 0) insert_clean(nt, newsize - 1, tab[s] - 1);
        free(tab);
        tab = nt;
        mask = newsize - 1;
    }
    int64_t m = 0;
    for (size_t s = 0; s <= mask; ++s)
        if (tab[s] != 0) out[m++] = tab[s] - 1;
    *n_out = m;
    free(tab);
    return 0;
}

// Python's random.Random on a caller-held state (random.getstate()[1]: 624 key words + position).  getrandbits(k), k <= 32 =
// genrand_uint32() >> (32 - k); _randbelow_with_getrandbits(n): k = n.bit_length(), redraw while r >= n.
namespace {
struct PyRandom {
    uint32_t* key;
    uint32_t* posp;
    uint32_t next() {
        if (*posp >= 624) {
            const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
            int i = 0;
            for (; i < 624 - 397; ++i) {
                const uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
                key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
            }
            for (; i < 623; ++i) {
                const uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
                key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
            }
            const uint32_t y = (key[623] & UP) | (key[0] & LO);
            key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
            *posp = 0;
        }
        uint32_t y = key[(*posp)++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    uint32_t below(uint32_t n) {                 // n >= 1
        int k = 32 - __builtin_clz(n);
        uint32_t r;
        do {
            r = next() >> (32 - k);
        } while (r >= n);
        return r;
    }
};
}  // namespace

// excl = per user the sorted, duplicate-free private ids of the items in train or test.  out[u * num + s] = the s-th sampled
// negative of user u (sample order).  setsize = random.sample's switch between its pool and its selection-set algorithm
// (21, + 4 ** ceil(log(3 num, 4)) when num > 5: computed by the caller with Python's own math.log).
extern "C" int el_host_negative_sample(const int64_t* excl_indptr, const int32_t* excl_indices, int64_t n_users, int64_t n_items,
                                       int32_t num, int64_t setsize, uint32_t* py_state625, int32_t* out) {
    EL_REQUIRE(excl_indptr != nullptr && py_state625 != nullptr && (out != nullptr || n_users * num == 0), "el_host_negative_sample: null argument");
    EL_REQUIRE(n_items >= 1 && n_items < 0x7fffffffll && num >= 0, "el_host_negative_sample: bad sizes");
    PyRandom rng{py_state625, py_state625 + 624};
    int32_t* pool = nullptr;
    int64_t pool_cap = 0;
    for (int64_t u = 0; u < n_users; ++u) {
        const int32_t* e = excl_indices + excl_indptr[u];
        const int64_t m = excl_indptr[u + 1] - excl_indptr[u];
        const int64_t n = n_items - m;                                 // candidates
        if (num > n) {
            free(pool);
            EL_REQUIRE(false, "el_host_negative_sample: a user has fewer candidate negatives than num_items (random.sample raises ValueError)");
        }
        // j-th candidate (ascending) = j + #{s : e[s] - s <= j}
        auto cand = [&](int64_t j) -> int32_t {
            int64_t lo = 0, hi = m;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)e[mid] - mid <= j)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            return (int32_t)(j + lo);
        };
        int32_t* o = out + u * (int64_t)num;
        if (n <= setsize) {
            if (n > pool_cap) {
                free(pool);
                pool_cap = n > 4096 ? n : 4096;
                pool = (int32_t*)malloc((size_t)pool_cap * sizeof(int32_t));
                EL_REQUIRE(pool != nullptr, "el_host_negative_sample: out of memory");
            }
            for (int64_t j = 0, s = 0, c = 0; c < n_items && j < n; ++c) {   // candidates, ascending
                if (s < m && e[s] == c) {
                    ++s;
                    continue;
                }
                pool[j++] = (int32_t)c;
            }
            for (int32_t i = 0; i < num; ++i) {
                const uint32_t j = rng.below((uint32_t)(n - i));
                o[i] = pool[j];
                pool[j] = pool[n - i - 1];
            }
        } else {
            // selection set: positions drawn so far (num is small -- 99 in the protocol -- a scan of the sorted-by-time list)
            static thread_local int64_t* sel = nullptr;
            static thread_local int32_t sel_cap = 0;
            if (num > sel_cap) {
                free(sel);
                sel_cap = num;
                sel = (int64_t*)malloc((size_t)sel_cap * sizeof(int64_t));
                EL_REQUIRE(sel != nullptr, "el_host_negative_sample: out of memory");
            }
            for (int32_t i = 0; i < num; ++i) {
                uint32_t j;
                bool again;
                do {
                    j = rng.below((uint32_t)n);
                    again = false;
                    for (int32_t q = 0; q < i; ++q)
                        if (sel[q] == (int64_t)j) {
                            again = true;
                            break;
                        }
                } while (again);
                sel[i] = j;
                o[i] = cand(j);
            }
        }
    }
    free(pool);
    return 0;
}
