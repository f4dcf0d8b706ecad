"""An NGCF driver that keeps the MODEL IN THE DRIVER, as NGCF_SPEX/code/main_rec.py does (its Model_Wrapper is defined at :36-113
and calls `torch.sparse.mm(self.norm_adj.to(self.device), ego)` at :76) — the case in which nothing of libspexhip is named by
the driver at all.  Run through the launcher,

    python -m spex_amd.dropin tests/drivers/ngcf_unchanged_driver.py --data_path <root> --dataset small --epoch 3

its data / sampler / evaluation come from the drop-in `utility` package and its sparse product from the operator hook
(spex_amd/dropin/sparse_hook.py): the adjacency the drop-in Data handed out is recognised when the driver moves it to the
device, the per-call upload is dropped and the product (and its autograd backward) runs on spex_spmm_f32.  Everything else of
the model is the driver's own torch code (embeddings, the two Linear layers per propagation layer, LeakyReLU, dropout, row
normalisation, the concatenation, dot + BCE), written here in the reference's structure and layer names.

Test-only switch (the parity test of the hook): SPEX_TEST_COUNTER_DROPOUT=<seed> replaces the driver's nn.Dropout modules by the
counter-based masks the G12-NGCF goldens were minted with (oracle/gen_golden.py does the same to the reference's model), so
the run is a function of the seeds and can be compared with the golden.  The last line printed reports how many products went
through the hook.
"""
from ngcf_parser import parse_args

cli = parse_args()

import os
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from utility.batch_test import args, data_generator, test
from utility.helper import *  # noqa: F401,F403  (trans_to_cuda)


def to_torch_sparse(mat):
    coo = mat.tocoo().astype(np.float32)
    index = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
    return torch.sparse_coo_tensor(index, torch.from_numpy(coo.data), torch.Size(coo.shape))


class Model_Wrapper(nn.Module):
    """The driver's own NGCF: user / item embeddings, per layer a graph-convolution Linear and a bi-interaction Linear."""

    def __init__(self, data_config, device):
        super().__init__()
        self.device = device
        self.n_users, self.n_items = data_config["n_users"], data_config["n_items"]
        sizes = [args.embed_size] + list(eval(args.layer_size))
        drops = list(eval(args.mess_dropout))
        self.norm_adj = to_torch_sparse(data_config["norm_adj"]).float()          # a CPU sparse tensor, moved per forward
        self.dropout_list, self.GC_Linear_list, self.Bi_Linear_list = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for k in range(len(sizes) - 1):
            self.GC_Linear_list.append(nn.Linear(sizes[k], sizes[k + 1]))
            self.Bi_Linear_list.append(nn.Linear(sizes[k], sizes[k + 1]))
            self.dropout_list.append(nn.Dropout(drops[k]))
        self.user_embedding = nn.Embedding(self.n_users + 1, args.embed_size)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        self.item_embedding = nn.Embedding(self.n_items, args.embed_size)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self.bce = nn.BCEWithLogitsLoss()

    def forward(self, user, item, labels_list, flag):
        ego = torch.cat((self.user_embedding.weight[:-1], self.item_embedding.weight), dim=0)
        layers = [ego]
        for k in range(len(self.GC_Linear_list)):
            side = torch.sparse.mm(self.norm_adj.to(self.device), ego)              # <- the operator the hook takes over
            ego = F.leaky_relu(self.GC_Linear_list[k](side)) + F.leaky_relu(self.Bi_Linear_list[k](ego * side))
            ego = self.dropout_list[k](ego)
            layers.append(F.normalize(ego, p=2, dim=1))
        table = torch.cat(layers, dim=1)
        users, items = torch.split(table, [self.n_users, self.n_items], dim=0)
        if flag == 1:
            return users, items
        score = (users[trans_to_cuda(user)] * items[trans_to_cuda(item)]).sum(dim=1)
        return self.bce(score, trans_to_cuda(labels_list))


class CounterDropout(nn.Module):
    """at::dropout's arithmetic (x * keep / (1 - p)) with the keep mask from the counter-based generator (seed, step, layer)."""

    def __init__(self, p, layer, seed, clock):
        super().__init__()
        self.p, self.layer, self.seed, self.clock = p, layer, seed, clock

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        from oracle.oracle import message_keep_mask              # test infrastructure (this driver lives under tests/)
        keep = message_keep_mask(x.shape[0], x.shape[1], self.p, self.seed, self.clock["step"], self.layer)
        noise = torch.from_numpy(keep.astype(np.float32)).div_(1 - self.p).to(x.device)
        return x * noise


def run():
    torch.manual_seed(2020); random.seed(2020); np.random.seed(2020)
    data_generator.print_statistics()
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    plain_adj, norm_adj, mean_adj = data_generator.get_adj_mat()
    net = Model_Wrapper({"n_users": data_generator.n_users, "n_items": data_generator.n_items, "norm_adj": norm_adj}, dev).to(dev)
    clock = {"step": 0}
    if os.environ.get("SPEX_TEST_COUNTER_DROPOUT"):
        for k, p in enumerate(eval(args.mess_dropout)):
            net.dropout_list[k] = CounterDropout(p, k, int(os.environ["SPEX_TEST_COUNTER_DROPOUT"]), clock)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr)
    for epoch in range(args.epoch):
        total = 0.0
        for user, item, labels in data_generator.load_train_data():
            net.train()
            opt.zero_grad()
            loss = net(user=trans_to_cuda(user), item=trans_to_cuda(item), labels_list=trans_to_cuda(labels), flag=0)
            loss.backward()
            opt.step()
            total += loss.item()
            clock["step"] += 1
        net.eval()
        ret = test(net, list(data_generator.test_set.keys()), drop_flag=True)
        print("epoch %d loss %.5f recall=%s ndcg=%s" % (epoch, total, ret["recall"].round(4).tolist(), ret["ndcg"].round(4).tolist()))
    try:
        from spex_amd.dropin import sparse_hook
        print("sparse_hook", sparse_hook.installed(), sparse_hook.stats)
    except ImportError:
        pass


if __name__ == "__main__":
    run()
