"""Padded trust-path batches behind the reference's `utility2.utils.Data` (LightGCN_SPEX/code/utility2/utils.py:3-51):
paths are right-padded with the index `n_node` (the embedding table's extra pad row, model.py:32) and carry a 0/1
mask; `generate_batch` / `get_slice` keep the reference's slicing (ragged last slice)."""
import numpy as np


class Data:
    def __init__(self, data, n_node, shuffle=False, graph=None, test=False):
        paths = data[0]
        self.n_node = n_node
        lens = np.fromiter((len(p) for p in paths), dtype=np.int64, count=len(paths))
        self.len_max = int(lens.max()) if len(paths) else 0
        self.inputs = np.full((len(paths), self.len_max), n_node, dtype=np.int64)
        for r, p in enumerate(paths):
            self.inputs[r, :len(p)] = p
        self.mask = (np.arange(self.len_max)[None, :] < lens[:, None]).astype(np.int64)
        self.targets = np.asarray(data[1])
        self.length = len(paths)
        self.shuffle, self.graph, self.test = shuffle, graph, test
        if test:
            self.neg = np.asarray(data[2])

    def generate_batch(self, batch_size):
        if self.shuffle:
            order = np.arange(self.length)
            np.random.shuffle(order)
            self.inputs, self.mask, self.targets = self.inputs[order], self.mask[order], self.targets[order]
            if self.test:
                self.neg = self.neg[order]
        n_batch = -(-self.length // batch_size)
        return [np.arange(k * batch_size, min((k + 1) * batch_size, self.length)) for k in range(n_batch)]

    def get_slice(self, i):
        if self.test:
            return self.inputs[i], self.mask[i], self.targets[i], self.neg[i]
        return self.inputs[i], self.mask[i], self.targets[i]

    def data_masks(self, all_usr_pois, item_tail):
        lens = [len(p) for p in all_usr_pois]
        len_max = max(lens)
        return ([p + item_tail * (len_max - le) for p, le in zip(all_usr_pois, lens)],
                [[1] * le + [0] * (len_max - le) for le in lens], len_max)
