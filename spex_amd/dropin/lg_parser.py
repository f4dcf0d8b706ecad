"""Command-line flags of the LightGCN drivers — same names, types and defaults as the reference's
LightGCN_SPEX/code/lg_parser.py:3-23, so `main_rec.py --dataset epinion2 --recdim 64 --layer 3 ...` parses unchanged.
Table-driven; `parse_args_r(argv)` additionally accepts an explicit argv for tests.
"""
import argparse

# (flag, type, default, help)
_FLAGS = [
    ("cuda_id", str, "0", "which GPU (exported as CUDA_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES by the driver)"),
    ("data_path", str, "../data/", "input data root"),
    ("dataset", str, "twitter", "epinion2 | weibo | twitter"),
    ("nb_heads", int, 3, "attention heads of the trust-path head"),
    ("recdim", int, 64, "embedding size"),
    ("layer", int, 3, "number of propagation layers"),
    ("lr", float, 0.001, "learning rate"),
    ("dropout", int, 0, "edge dropout on/off"),
    ("keepprob", float, 0.6, "edge keep probability when dropout is on"),
    ("a_fold", int, 100, "row blocks when A_split is on"),
    ("epochs", int, 50, "training epochs"),
    ("seed", int, 2020, "random seed"),
    ("A_split", int, 0, "propagate the adjacency as a_fold row blocks"),
    ("batch_size", int, 256, "parsed for compatibility (the driver batches 256)"),
    ("batchSize", int, 256, "trust-path batch size"),
    ("hiddenSize", int, 64, "hidden size of the trust-path head"),
    ("act", int, 1, "activation selector"),
]


def build_parser():
    p = argparse.ArgumentParser(description="LightGCN-SPEX on MI355X")
    for name, typ, default, text in _FLAGS:
        p.add_argument("--" + name, type=typ, default=default, help=text)
    p.add_argument("--nonhybrid", action="store_true", help="only use the global preference to predict")
    return p


def parse_args_r(argv=None):
    return build_parser().parse_args(argv)
