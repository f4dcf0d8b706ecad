"""Padded trust-path batches behind the reference's `utility2.utils.Data` (LightGCN_SPEX/code/utility2/utils.py:3-51).

A trust sample is a path of user ids with a target user; the model consumes them as a dense [n, len_max] index matrix
right-padded with `n_node` (the embedding table's extra pad row, model.py:32) plus a 0/1 mask of the real positions.
The reference pads with Python list arithmetic per path; here the matrix is filled once with NumPy.  Public attributes
and method contracts are the reference's (inputs, mask, targets, neg, length, len_max; ragged last slice).
"""
import numpy as np


def _pad_paths(paths, pad_value):
    """(index matrix [n, len_max] filled with pad_value, 0/1 mask, len_max) for a list of variable-length paths."""
    sizes = np.array([len(p) for p in paths], dtype=np.int64)
    width = int(sizes.max()) if sizes.size else 0
    table = np.full((sizes.size, width), pad_value, dtype=np.int64)
    real = np.arange(width)[None, :] < sizes[:, None]
    if sizes.size:
        table[real] = np.concatenate([np.asarray(p, dtype=np.int64) for p in paths]) if sizes.sum() else []
    return table, real.astype(np.int64), width


class Data:
    _FIELDS = ("inputs", "mask", "targets")

    def __init__(self, data, n_node, shuffle=False, graph=None, test=False):
        self.n_node, self.shuffle, self.graph, self.test = n_node, shuffle, graph, test
        self.inputs, self.mask, self.len_max = _pad_paths(data[0], n_node)
        self.targets = np.asarray(data[1])
        self.length = self.inputs.shape[0]
        if test:
            self.neg = np.asarray(data[2])

    def _fields(self):
        return self._FIELDS + (("neg",) if self.test else ())

    def generate_batch(self, batch_size):
        """Index slices of `batch_size` samples (the last one ragged); with shuffle, the samples are permuted first
        through NumPy's global RNG like the reference does."""
        if self.shuffle:
            perm = np.arange(self.length)
            np.random.shuffle(perm)
            for name in self._fields():
                setattr(self, name, getattr(self, name)[perm])
        starts = range(0, self.length, batch_size)
        return [np.arange(s, min(s + batch_size, self.length)) for s in starts]

    def get_slice(self, i):
        return tuple(getattr(self, name)[i] for name in self._fields())

    def data_masks(self, all_usr_pois, item_tail):
        """The reference's list-of-lists form of the same padding (kept for callers that use it directly)."""
        table, real, width = _pad_paths(all_usr_pois, item_tail[0])
        return table.tolist(), real.tolist(), width
