"""Dataset + graph + negative sampler behind the reference's `utility1.dataloader` surface.

Same classes, attributes and file formats as LightGCN_SPEX/code/utility1/dataloader.py (Loader :65-238,
LightTrainData :241-277), so main_rec.py / main_auto_expert_s.py use it unchanged:

    dataset = dataloader.Loader(args)                       # main_rec.py:18
    LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)   # main_rec.py:19
    dataset.getSparseGraph()                                # model.py:38

What differs is how the work is done: the interaction lists are parsed and turned into CSR with vectorised NumPy
(no per-pair dok assignment, no lil slicing), the adjacency lives in HBM behind a libspexhip graph handle instead of a
torch sparse tensor, and `ng_sample()` replays NumPy's global RNG stream in bulk — it returns exactly the negatives
the reference's Python loop would draw for the same seed, in a fraction of a second instead of ~27 s per epoch.
"""
import os

import numpy as np
import scipy.sparse as sp
import torch
from torch.utils.data import Dataset

from spex_amd.graph import SpexGraph, lightgcn_norm_adj, row_block


class BasicDataset(Dataset):
    """Interface of the reference's BasicDataset (dataloader.py:10-63)."""

    def __init__(self):
        pass

    n_users = property(lambda self: self._not_impl())
    m_items = property(lambda self: self._not_impl())
    trainDataSize = property(lambda self: self._not_impl())
    testDict = property(lambda self: self._not_impl())
    allPos = property(lambda self: self._not_impl())

    def _not_impl(self):
        raise NotImplementedError

    def getUserItemFeedback(self, users, items):
        raise NotImplementedError

    def getUserPosItems(self, users):
        raise NotImplementedError

    def getUserNegItems(self, users):
        raise NotImplementedError

    def getSparseGraph(self):
        raise NotImplementedError


def _read_pairs(path):
    """`u i [r]` per line, space separated (written by data_process_rec.py:401-416)."""
    import pandas as pd
    df = pd.read_csv(path, sep=" ", header=None, usecols=[0, 1], dtype=np.int32)
    return df[0].to_numpy(), df[1].to_numpy()


class Loader(BasicDataset):
    """Reads `<data_path><ds>/rec/<ds>.{train.rating,test.rating,test.negative}` and owns the normalised graph."""

    def __init__(self, config):
        super().__init__()
        dataset = config.dataset
        root = getattr(config, "data_path", "../data/")
        self.path = os.path.join(root, dataset) + "/"
        self.split = config.A_split
        self.folds = config.a_fold
        self.mode_dict = {"train": 0, "test": 1}
        self.mode = self.mode_dict["train"]
        self.traindataSize = 0
        self.testDataSize = 0

        rec = os.path.join(self.path, "rec")
        self.trainUser, self.trainItem = _read_pairs(os.path.join(rec, f"{dataset}.train.rating"))
        self.n_user = int(self.trainUser.max()) + 1
        self.m_item = int(self.trainItem.max()) + 1
        self.trainUniqueUsers = np.unique(self.trainUser)
        self.rec_train_data = np.stack([self.trainUser, self.trainItem], 1).tolist()

        ones = np.ones(len(self.trainUser), np.float32)
        shape = (self.n_user + 1, self.m_item)          # +1: pad row used by the trust paths (model.py:32)
        coo = sp.coo_matrix((ones, (self.trainUser, self.trainItem)), shape=shape)
        coo.sum_duplicates()
        coo.data[:] = 1.0
        self.train_mat = coo.todok()
        self.UserItemNet = sp.csr_matrix((np.ones(len(self.trainUser)), (self.trainUser, self.trainItem)), shape=shape)
        self.users_D = np.asarray(self.UserItemNet.sum(axis=1)).squeeze()
        self.users_D[self.users_D == 0.0] = 1.0
        self.items_D = np.asarray(self.UserItemNet.sum(axis=0)).squeeze()
        self.items_D[self.items_D == 0.0] = 1.0

        self.testRatings = self.load_test_rating_as_dict(os.path.join(rec, f"{dataset}.test.rating"))
        self.testNegatives = self.load_test_negative_as_dict(os.path.join(rec, f"{dataset}.test.negative"))

        self.Graph = None
        self.adj_csr = None
        print(dataset)
        print("use:", self.n_user)
        print("item:", self.m_item)
        print("----------------")

    n_users = property(lambda self: self.n_user)
    m_items = property(lambda self: self.m_item)
    trainDataSize = property(lambda self: self.traindataSize)

    @staticmethod
    def load_test_rating_as_dict(filename):
        """{user: [item]} — a later line for the same user replaces an earlier one (dataloader.py:139-148), which is
        how the 99 negative rows in front of each positive row drop out."""
        out = {}
        with open(filename) as f:
            for line in f:
                a = line.split()
                if a:
                    out[int(a[0])] = [int(a[1])]
        return out

    @staticmethod
    def load_test_negative_as_dict(filename):
        out = {}
        with open(filename) as f:
            for line in f:
                a = line.split()
                if a:
                    out[int(a[0])] = [int(x) for x in a[1:]]
        return out

    def build_adjacency(self):
        """Host CSR of D^-1/2 [[0,R],[R^T,0]] D^-1/2, fp32 (reference: dataloader.py:197-212)."""
        if self.adj_csr is None:
            self.adj_csr = lightgcn_norm_adj(self.trainUser, self.trainItem, self.n_user, self.m_item)
        return self.adj_csr

    def getSparseGraph(self):
        """The propagation operator on the GPU.  A SpexGraph, or with --A_split a list of `a_fold` row-block graphs
        (dataloader.py:167-177)."""
        if self.Graph is None:
            rowptr, col, val = self.build_adjacency()
            n = len(rowptr) - 1
            if self.split:
                fold_len = n // self.folds
                bounds = [i * fold_len for i in range(self.folds)] + [n]
                # edge ids = positions in the unsplit matrix: a dropout mask indexed by them addresses the same entries
                # in the row blocks, in the unsplit graph and in the transposed blocks the backward pass uses
                eid = np.arange(len(col), dtype=np.int32)
                self.fold_bounds = bounds
                self.Graph = []
                for i in range(self.folds):
                    r, c, v, e = row_block(rowptr, col, val, bounds[i], bounds[i + 1], eid)
                    self.Graph.append(SpexGraph(r, c, v, n_cols=n, edge_id=e))
            else:
                self.Graph = SpexGraph(rowptr, col, val)
                print("self.Graph:", self.Graph.size())
        return self.Graph

    def getUserItemFeedback(self, users, items):
        return np.array(self.UserItemNet[users, items]).astype("uint8").reshape((-1,))


class LightTrainData(Dataset):
    """Positives + 5 sampled negatives each (dataloader.py:241-277).  `ng_sample()` consumes NumPy's global RNG
    exactly as the reference loop does — one `np.random.randint(num_item)` per attempt, redrawn while the pair is a
    training interaction — but evaluates the whole stream at once."""

    def __init__(self, features, num_item, train_mat=None):
        super().__init__()
        self.features_ps = features
        self.num_item = num_item
        self.train_mat = train_mat
        self.num_ng = 5
        self.labels = [0 for _ in range(len(features))]
        ps = np.asarray(features, dtype=np.int64).reshape(-1, 2) if len(features) else np.zeros((0, 2), np.int64)
        self._ps = ps
        if train_mat is not None:
            coo = sp.coo_matrix(train_mat)
            self._train_keys = np.unique(coo.row.astype(np.int64) * np.int64(num_item) + coo.col.astype(np.int64))
        else:
            self._train_keys = np.zeros(0, np.int64)
        self.users_fill = self.items_fill = self.labels_fill_np = None

    def _in_train(self, u, j):
        key = u * np.int64(self.num_item) + j
        pos = np.searchsorted(self._train_keys, key)
        pos[pos == len(self._train_keys)] = 0
        return (self._train_keys[pos] == key) if len(self._train_keys) else np.zeros(len(key), bool)

    def ng_sample_device(self, device="cuda", seed=None):
        """Same distribution as ng_sample(), drawn on the GPU (spex_sample_negatives: one thread per negative, Philox
        draws, binary search in the user's sorted item list).  Not NumPy's stream: the negatives for a given seed
        differ from the reference's, their distribution does not.  `seed` defaults to a draw from NumPy's global RNG so
        that np.random.seed(...) still makes runs reproducible."""
        import torch
        from spex_amd import ops
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2654435761 % (2 ** 63)
        dev = torch.device(device)
        if getattr(self, "_dev_csr", None) is None or self._dev_csr[0].device != dev:
            n_rows = int(self._ps[:, 0].max()) + 1 if len(self._ps) else 0
            if len(self._train_keys):
                n_rows = max(n_rows, int(self._train_keys[-1] // self.num_item) + 1)
            users = (self._train_keys // self.num_item).astype(np.int64)
            rowptr = np.zeros(n_rows + 1, np.int64)
            np.cumsum(np.bincount(users, minlength=n_rows), out=rowptr[1:])
            items = (self._train_keys % self.num_item).astype(np.int32)          # keys are sorted: rows come out sorted
            self._dev_csr = (torch.from_numpy(rowptr.astype(np.int32)).to(dev), torch.from_numpy(items).to(dev),
                             torch.from_numpy(np.ascontiguousarray(self._ps[:, 0])).to(dev))
        rowptr, items, pos_user = self._dev_csr
        neg = ops.sample_negatives(rowptr, items, pos_user, self.num_ng, self.num_item, seed).cpu().numpy()
        P, S = len(self._ps), len(self._ps) * self.num_ng
        self.users_fill = np.concatenate([self._ps[:, 0], np.repeat(self._ps[:, 0], self.num_ng)])
        self.items_fill = np.concatenate([self._ps[:, 1], neg])
        self.labels_fill_np = np.concatenate([np.ones(P, np.int64), np.zeros(S, np.int64)])
        self._features_ng = None

    def ng_sample(self, block=8192):
        """Slot k (positive k // num_ng, user u_k) takes the first not-yet-consumed stream value j with (u_k, j) not a
        training pair; rejected values are consumed too.  The stream is processed in blocks: inside a block the
        mapping position -> slot is found by fixed-point iteration (a position's slot is its index minus the
        rejections before it; rejections are sparse, so this settles in a few passes), the running slot offset is
        carried from block to block, and exactly as many values are drawn as the reference would draw."""
        P = len(self._ps)
        S = P * self.num_ng
        slot_user = np.repeat(self._ps[:, 0], self.num_ng)
        neg_items = np.empty(S, np.int64)
        k = 0                                           # next slot to fill
        while k < S:
            stream = np.random.randint(self.num_item, size=S - k).astype(np.int64)   # never more than still needed
            pos = 0
            while pos < len(stream):
                vals = stream[pos:pos + block]          # len(vals) <= remaining slots, so slots never overrun
                n = len(vals)
                rej = np.zeros(n, bool)
                while True:
                    slot = k + np.arange(n) - (np.cumsum(rej) - rej)
                    new = self._in_train(slot_user[slot], vals)
                    if np.array_equal(new, rej):
                        break
                    rej = new
                acc = vals[~rej]
                neg_items[k:k + len(acc)] = acc
                k += len(acc)
                pos += n
        self.users_fill = np.concatenate([self._ps[:, 0], slot_user])
        self.items_fill = np.concatenate([self._ps[:, 1], neg_items])
        self.labels_fill_np = np.concatenate([np.ones(P, np.int64), np.zeros(S, np.int64)])
        self._features_ng = None

    @property
    def features_ng(self):
        if self._features_ng is None:
            P = len(self._ps)
            self._features_ng = np.stack([self.users_fill[P:], self.items_fill[P:]], 1).tolist()
        return self._features_ng

    @property
    def features_fill(self):
        return np.stack([self.users_fill, self.items_fill], 1).tolist()

    @property
    def labels_fill(self):
        return self.labels_fill_np.tolist()

    def __len__(self):
        return (self.num_ng + 1) * len(self.labels)

    def __getitem__(self, idx):
        return int(self.users_fill[idx]), int(self.items_fill[idx]), int(self.labels_fill_np[idx])

    def __getitems__(self, indices):
        """Batched fetch used by torch's DataLoader: one fancy-index per field instead of 256 __getitem__ calls.  Returns
        the three fields as int64 tensors; DataLoader's default collate stacks them into one [3, B] tensor, which the
        caller's `user, item, label = data` (main_rec.py:32) unpacks into the same three [B] int64 tensors the reference's
        per-sample collate produces — without building and re-parsing 256 Python tuples per step."""
        ix = np.asarray(indices)
        return [torch.from_numpy(self.users_fill[ix]), torch.from_numpy(self.items_fill[ix]),
                torch.from_numpy(self.labels_fill_np[ix])]
