"""NGCF's dataset object behind the reference's `utility.load_data` surface (NGCF_SPEX/code/utility/load_data.py).

`Data(path, batch_size)` reads `<path>/rec/{train.txt,test.txt,negative.txt}` (:29-31: `uid item item ...` per line;
negative.txt `uid neg ...`) and exposes what the drivers and `utility.batch_test` touch: n_users, n_items, n_train,
n_test, exist_users, all_items, neg_item, R, train_items, test_set, get_adj_mat(), load_train_data(),
print_statistics(), get_num_users_items(), negative_pool().

What changes underneath
  * adjacency (create_adj_mat, :122-166): the three matrices are built straight into CSR by spex_amd.graph (one
    vectorised sort, the reference's arithmetic — D^-1 (A + I) in float64, D^-1 A in float32) instead of dok -> lil
    slicing -> dok (9.7 s on Epinion2).  No `s_*_adj_mat.npz` cache is written: the reference's cache is keyed by the
    directory only and silently serves a stale matrix after the data changes.
  * training samples (load_train_data / train_sample, :13-23,176-195): per user 5 x |positives| DISTINCT negatives,
    `random.sample` from (all items - positives).  The reference fans the users out to a multiprocessing.Pool whose
    workers each start from the parent's `random` state, so its stream depends on how the pool happens to chunk the
    users; drawn serially from the global `random` the negatives are a pure function of `random.seed` — and equal the
    reference's whenever its pool is a serial map (that is how the test fixtures were minted).
"""
import random

import numpy as np
import scipy.sparse as sp
import torch

from spex_amd.dropin.ngcf.ngcf_parser import parse_known
from spex_amd.graph import ngcf_adjacency

args = parse_known()          # parsed at import like the reference (:11); only print_statistics reads it
TRAIN_USER_BLOCK = 256        # load_train_data walks the users in blocks of 256 and DROPS the ragged tail (:177,181)


def _int_fields(line):
    return [int(tok) for tok in line.strip("\n").split(" ")]


def train_sample(u, hu, ai):
    """One user's samples (:13-23): [users, items, labels] with 5 x |pos| distinct negatives first, then the positives.
    The candidate order fed to random.sample is the iteration order of `ai - set(pos)`, as in the reference."""
    pos = hu[u]
    n_neg = 5 * len(pos)
    candidates = tuple(ai - set(pos))            # random.sample(set) == random.sample(tuple(set)) on Python <= 3.10
    items = random.sample(candidates, n_neg)
    items.extend(pos)
    return [[u] * (n_neg + len(pos)), items, [0] * n_neg + [1] * len(pos)]


# ------------------------------------------------------------------------------------------------------------------------------
# train_sample() for a whole epoch WITHOUT a Python-level draw per negative (0.9 s per Epinion2 epoch — five times the GPU time of the
# epoch's 4 700 training steps).  What `random.sample(tuple(all_items - set(pos)), 5 |pos|)` does on CPython 3.10 (Lib/random.py):
#   * the population: the set difference iterates in ascending item order (small ints hash to themselves and the result's table is
#     larger than the largest item) — checked on a probe below, not assumed;
#   * n > setsize (= 21 + 4 ** ceil(log(3k, 4)) for k > 5): k times `j = randbelow(n)`, redrawn while j was selected before;
#     randbelow(n) = getrandbits(n.bit_length()) redrawn while >= n;  getrandbits(b <= 32) = one MT19937 output >> (32 - b).
#     So the user's negatives are the first k DISTINCT values < n of the stream's outputs >> (32 - b): one vectorised pass;
#   * n <= setsize (users with more than ~270 items; 63 of Epinion2's 3 185): the pool-swap form, whose threshold changes every draw —
#     replayed draw by draw on the same stream.
# The stream is NumPy's MT19937 started from `random.getstate()` (the same generator, the same state layout); afterwards `random` is
# left exactly where a draw-by-draw run would have left it.
_MT_BLOCK = 1 << 18


def _blocked_replay_applies(all_items, train_items):
    """Does `tuple(all_items - set(pos))` iterate in ascending item order for every user?  Small non-negative ints hash to themselves,
    so a set iterates ascending as long as its hash table has more slots than its largest member — and the table of the difference
    grows with the number of members: the user with the MOST positives has the smallest one.  That user's tuple is checked, not
    assumed; anything unexpected (another Python, other item types) keeps the draw-by-draw loop."""
    import sys
    if sys.version_info >= (3, 11) or not isinstance(all_items, set) or not all_items or not train_items:
        return False
    try:
        if min(all_items) < 0 or max(all_items) >= (1 << 31):
            return False
        heavy = max(train_items, key=lambda u: len(train_items[u]))
        probe = tuple(all_items - set(train_items[heavy]))
    except TypeError:
        return False
    return all(type(x) is int for x in probe[:64]) and list(probe) == sorted(probe)


class _PyMTStream:
    """The outputs `random`'s generator is about to produce, block by block, with the generator itself untouched until sync()."""

    def __init__(self):
        st = random.getstate()
        self._gauss = st[2]
        self._bg = np.random.MT19937()
        self._bg.state = {"bit_generator": "MT19937", "state": {"key": np.array(st[1][:-1], dtype=np.uint32), "pos": int(st[1][-1])}}
        self._next_block()

    def _next_block(self):
        self._start = self._bg.state                                  # (a dict holding its own copy of the key)
        self.buf = self._bg.random_raw(_MT_BLOCK).astype(np.uint32)
        self.cur = 0

    def take(self, m):
        """The next m outputs WITHOUT consuming them (may cross into a new block: the unconsumed tail is carried over)."""
        while self.cur + m > len(self.buf):
            tail = self.buf[self.cur:]
            # restart the bookkeeping at the first unconsumed output: state at block start advanced by `cur`
            bg = np.random.MT19937(); bg.state = self._start
            if self.cur:
                bg.random_raw(self.cur)
            self._start = bg.state
            self.buf = np.concatenate([tail, self._bg.random_raw(_MT_BLOCK).astype(np.uint32)])
            self.cur = 0
        return self.buf[self.cur: self.cur + m]

    def consume(self, m):
        self.cur += m

    def sync(self):
        """Put `random` where the stream stands."""
        bg = np.random.MT19937(); bg.state = self._start
        if self.cur:
            bg.random_raw(self.cur)
        st = bg.state["state"]
        random.setstate((3, tuple(int(x) for x in st["key"]) + (int(st["pos"]),), self._gauss))


def _sample_epoch_blocked(users, train_items, all_items):
    from math import ceil, log
    stream = _PyMTStream()
    items_sorted = np.fromiter(sorted(all_items), dtype=np.int64, count=len(all_items))     # the population's order (ids may have gaps)
    n_all = len(items_sorted)
    dense = bool(n_all and items_sorted[-1] == n_all - 1)                # ids 0 .. n-1 without gaps: an item is its own rank
    total = 6 * sum(len(train_items[u]) for u in users)
    out_u, out_v, out_r = np.empty(total, np.int64), np.empty(total, np.int64), np.zeros(total, np.float32)
    first_at = np.empty(n_all, dtype=np.int64)                           # scratch: where a rank was first seen among a user's draws
    w_ = 0
    for u in users:
        pos = train_items[u]
        n_pos = len(pos)
        k = 5 * n_pos
        pos_arr = np.asarray(pos, dtype=np.int64)
        # the population — all items minus the user's, ascending — is never built: its j-th member is the item of rank
        # j + #{positives' ranks <= that rank}
        ranks = np.sort(pos_arr) if dense else np.searchsorted(items_sorted, np.sort(pos_arr))
        p_adj = ranks - np.arange(n_pos)
        if n_pos > 1 and (np.diff(p_adj) < 0).any():                     # (a positive listed twice: set(pos) has fewer members)
            ranks = np.unique(ranks)
            p_adj = ranks - np.arange(len(ranks))
        n = n_all - len(p_adj)
        if k > n:
            raise ValueError("Sample larger than population or is negative")      # (random.sample's own error)
        setsize = 21 + (4 ** ceil(log(k * 3, 4)) if k > 5 else 0)
        if n <= setsize:
            # the pool-swap form, draw by draw on the stream (the threshold n - i changes with every draw): j = randbelow(n - i),
            # result[i] = pool[j], pool[j] = pool[n - i - 1]
            pool = np.arange(n)
            pool = pool + np.searchsorted(p_adj, pool, side="right")
            pool = (pool if dense else items_sorted[pool]).tolist()
            neg = [0] * k
            m = 2 * k + 64
            w = stream.take(m).tolist()
            p_ = 0
            for i in range(k):
                t = n - i
                sh = 32 - t.bit_length()
                while True:
                    if p_ >= m:
                        m *= 2
                        w = stream.take(m).tolist()
                    r = w[p_] >> sh
                    p_ += 1
                    if r < t:
                        break
                neg[i] = pool[r]
                pool[r] = pool[t - 1]
            stream.consume(p_)
            out_v[w_: w_ + k] = neg
        else:
            shift = 32 - int(n).bit_length()
            m = k + (k >> 1) + 64
            while True:
                r = stream.take(m) >> np.uint32(shift)
                ok = np.flatnonzero(r < n)
                vals = r[ok].astype(np.int64)
                idx = np.arange(len(vals))
                first_at[vals[::-1]] = idx[::-1]                        # (assigned back to front: the FIRST occurrence's index stays)
                fresh = np.flatnonzero(first_at[vals] == idx)           # the accepted draws that are not repeats, in order
                if len(fresh) >= k:
                    fresh = fresh[:k]
                    j = vals[fresh]
                    j = j + np.searchsorted(p_adj, j, side="right")
                    out_v[w_: w_ + k] = j if dense else items_sorted[j]
                    stream.consume(int(ok[fresh[-1]]) + 1)
                    break
                m *= 2
        out_u[w_: w_ + k + n_pos] = u
        out_v[w_ + k: w_ + k + n_pos] = pos_arr
        out_r[w_ + k: w_ + k + n_pos] = 1.0
        w_ += k + n_pos
    stream.sync()
    return out_u, out_v, out_r


class Data(object):
    def __init__(self, path, batch_size):
        self.path, self.batch_size = path, batch_size
        self.neg_pools, self.neg_item = {}, {}
        self.exist_users, self.train_items, self.test_set = [], {}, {}
        self.n_train = self.n_test = 0
        max_user = max_item = 0
        items_seen = set()
        with open(path + "/rec/train.txt") as fh:
            for line in fh:
                row = _int_fields(line)                       # a line without items raises, like :48-50
                uid, its = row[0], row[1:]
                max_item = max(max_item, max(its))
                max_user = max(max_user, uid)
                items_seen.update(its)
                self.exist_users.append(uid)
                self.train_items[uid] = its
                self.n_train += len(its)
        with open(path + "/rec/test.txt") as fh:
            for line in fh:
                try:
                    row = _int_fields(line)
                    uid, its = row[0], row[1:]
                    max_item = max(max_item, max(its))
                except Exception:                             # malformed / empty lines are skipped (:57-60,98-101)
                    continue
                self.n_test += len(its)
                self.test_set[uid] = its
        with open(path + "/rec/negative.txt") as fh:
            for line in fh:
                try:
                    row = _int_fields(line)
                except Exception:
                    continue
                self.neg_item[row[0]] = row[1:]
        self.n_users, self.n_items = max_user + 1, max_item + 1
        self.all_items = items_seen
        uu = np.repeat(np.fromiter(self.train_items.keys(), np.int64, len(self.train_items)),
                       [len(v) for v in self.train_items.values()])
        ii = np.fromiter((i for v in self.train_items.values() for i in v), np.int64, self.n_train)
        self._pairs = (uu, ii)
        R = sp.csr_matrix((np.ones(len(uu), np.float32), (uu, ii)), shape=(self.n_users, self.n_items))
        R.data[:] = 1.0                                        # a repeated pair is one interaction (dok assignment, :92)
        self.R = R
        self.R_Item_Interacts = sp.dok_matrix((self.n_items, self.n_items), dtype=np.float32)   # (:82, never filled)

    # ------------------------------------------------------------------ adjacency (:107-166)
    def get_adj_mat(self):
        return self.create_adj_mat()

    def create_adj_mat(self):
        """(A, D^-1 (A + I), D^-1 A) over n_users + n_items nodes as scipy CSR, values as the reference computes them."""
        n = self.n_users + self.n_items
        out = []
        for kind in ("plain", "norm", "mean"):
            rowptr, col, val = ngcf_adjacency(self._pairs[0], self._pairs[1], self.n_users, self.n_items, kind)
            out.append(sp.csr_matrix((val, col, rowptr), shape=(n, n)))
        from spex_amd.dropin import sparse_hook
        if sparse_hook.installed():        # an unchanged driver's own torch.sparse.mm(adj, x) then lands on spex_spmm_f32
            for m in out:
                sparse_hook.register_adjacency(m)
        return tuple(out)

    # ------------------------------------------------------------------ training samples (:168-195)
    def negative_pool(self):
        for u, its in self.train_items.items():
            free = list(set(range(self.n_items)) - set(its))
            self.neg_pools[u] = [random.choice(free) for _ in range(100)]

    def sample_epoch(self, fast=True):
        """The epoch's (users, items, labels) as int64 / int64 / float32 arrays, in the reference's order: users in
        file order, whole blocks of 256 users only, per user the negatives then the positives.
        fast: the blocked replay of the `random` stream (_sample_epoch_blocked: the same negatives, the same generator state
        afterwards, ~6 x faster); False — or a Python whose random.sample / set order the replay does not know — the loop below."""
        users = list(self.train_items.keys())
        users = users[: len(users) // TRAIN_USER_BLOCK * TRAIN_USER_BLOCK]
        if fast and _blocked_replay_applies(self.all_items, self.train_items):
            return _sample_epoch_blocked(users, self.train_items, self.all_items)
        us, vs, rs = [], [], []
        for u in users:
            a, b, c = train_sample(u, self.train_items, self.all_items)
            us.extend(a); vs.extend(b); rs.extend(c)
        return np.asarray(us, np.int64), np.asarray(vs, np.int64), np.asarray(rs, np.float32)

    def load_train_data(self):
        us, vs, rs = self.sample_epoch()
        ds = torch.utils.data.TensorDataset(torch.from_numpy(us), torch.from_numpy(vs), torch.from_numpy(rs))
        return torch.utils.data.DataLoader(ds, batch_size=self.batch_size, shuffle=True)

    # ------------------------------------------------------------------ small accessors
    def get_num_users_items(self):
        return self.n_users, self.n_items

    def print_statistics(self):
        print(args.dataset)
        print("use:", self.n_users)
        print("item:", self.n_items)
        print("----------------")
