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

    def sample_epoch(self):
        """The epoch's (users, items, labels) as int64 / int64 / float32 arrays, in the reference's order: users in
        file order, whole blocks of 256 users only, per user the negatives then the positives."""
        users = list(self.train_items.keys())
        users = users[: len(users) // TRAIN_USER_BLOCK * TRAIN_USER_BLOCK]
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
