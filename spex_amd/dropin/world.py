"""`world.config` view of the parsed flags.

BASELINE.json's north star names `world.config`; the reference has no `world` module (only dead code mentions it,
LightGCN_SPEX/code/utility1/utils.py:16-33) — the name comes from upstream LightGCN-PyTorch.  This alias exposes the
lg_parser Namespace under the upstream keys so either spelling works.
"""
import sys

from lg_parser import build_parser

# the launcher's argv may carry driver-only flags: ignore what we do not know
args, _ = build_parser().parse_known_args(sys.argv[1:])
config = {
    "latent_dim_rec": args.recdim, "lightGCN_n_layers": args.layer, "lr": args.lr, "dropout": args.dropout,
    "keep_prob": args.keepprob, "A_n_fold": args.a_fold, "A_split": bool(args.A_split), "bpr_batch_size": args.batch_size,
    "decay": 1e-4, "pretrain": 0, "test_u_batch_size": 100,
}
dataset = args.dataset
seed = args.seed
model_name = "lgn"
TRAIN_epochs = args.epochs
