"""Command-line flags of the NGCF drivers — same names, types and defaults as the reference's
NGCF_SPEX/code/ngcf_parser.py:3-26, so `main_rec.py --dataset epinion2 --embed_size 64 --layer_size [64] ...` parses
unchanged.  Table-driven; `parse_args(argv)` additionally accepts an explicit argv (tests), and unknown flags of a
launcher are ignored by `parse_known()`."""
import argparse

# (flag, type or None for nargs='?' strings, default, help)
_FLAGS = [
    ("cuda_id", str, "0", "which GPU"),
    ("nb_heads", int, 3, "attention heads of the trust-path head"),
    ("data_path", None, "../Data/", "input data root"),
    ("dataset", None, "epinion2", "epinion2 | twitter | weibo"),
    ("verbose", int, 1, "evaluation interval"),
    ("epoch", int, 50, "training epochs"),
    ("embed_size", int, 64, "embedding size"),
    ("layer_size", None, "[64]", "output size of every propagation layer"),
    ("batch_size", int, 256, "batch size"),
    ("hidden_size", int, 64, "hidden size of the trust-path head"),
    ("regs", None, "[1e-5]", "regularisation (parsed, unused by the BCE loss)"),
    ("lr", float, 0.001, "learning rate"),
    ("adj_type", None, "norm", "plain | norm | mean adjacency"),
    ("alg_type", None, "ngcf", "graph convolution type"),
    ("mess_dropout", None, "[0.1]", "message dropout ratio per layer"),
    ("Ks", None, "[10,20,50]", "cut-offs of recall / ndcg"),
    ("test_flag", None, "part", "part | full ranking"),
    ("act", int, 0, "activation selector"),
]


def build_parser():
    p = argparse.ArgumentParser(description="NGCF-SPEX on MI355X")
    for name, typ, default, text in _FLAGS:
        if typ is None:
            p.add_argument("--" + name, nargs="?", default=default, help=text)
        else:
            p.add_argument("--" + name, type=typ, default=default, help=text)
    p.add_argument("--nonhybrid", action="store_true", help="only use the global preference to predict")
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def parse_known(argv=None):
    return build_parser().parse_known_args(argv)[0]
