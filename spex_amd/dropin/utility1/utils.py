"""Seeding and batching helpers (reference: LightGCN_SPEX/code/utility1/utils.py)."""
import os
import random

import numpy as np
import torch


def set_seed(seed):
    """Same sources, same order as utils.py:7-14 (torch CPU + all GPUs, Python, NumPy, PYTHONHASHSEED)."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def minibatch(*tensors, batch_size=256):
    n = len(tensors[0])
    for i in range(0, n, batch_size):
        yield tensors[0][i:i + batch_size] if len(tensors) == 1 else tuple(x[i:i + batch_size] for x in tensors)


def getFileName(dataset, n_layers, latent_dim, model_name="lgn"):
    return f"checkpoints/{model_name}-{dataset}-{n_layers}-{latent_dim}.pth.tar"
