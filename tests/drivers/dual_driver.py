"""Dual-task (recommendation + trust-path) driver for the drop-in launcher test.

It touches the package exactly where LightGCN_SPEX/code/main_auto_expert_s.py does — the same imports by the same
names, Loader -> LightTrainData -> DataLoader(256, shuffle) -> pickled trust paths wrapped in utility2.utils.Data ->
model_expert_s.LightGCN(args, dataset).to(device) -> Adam; per batch the paths of the batch's users (at most three
trust batches' worth) go in with the rec batch at flag=0 and the two losses are combined with the learned task
weights; evaluation is rec_test + trust_test5 — but is written as small functions for the test's needs (both trust
pickles come from --data_path; the reference hard-codes '../data/' for the test split, :41).
Run it as `python -m spex_amd.dropin tests/drivers/dual_driver.py --dataset tiny --data_path <root> --epochs 2`.
"""
from lg_parser import parse_args_r

cli = parse_args_r()

import pickle
import random
from collections import defaultdict

import numpy as np
import torch
from torch.utils.data import DataLoader

import utility1.dataloader as dataloader
import utility1.model_expert_s as model
import utility1.utils as utils
from utility1.batch_test import rec_test
from utility1.dataloader import LightTrainData
from utility2.batch_test_gnn import trust_test5
from utility2.utils import Data


def load_trust(split):
    with open("%s%s/trust/%s" % (cli.data_path, cli.dataset, split), "rb") as fh:
        return pickle.load(fh)


def paths_by_first_user(paths):
    index = defaultdict(list)
    for k, nodes in enumerate(paths):
        index[nodes[0]].append(k)
    return index


def weighted_loss(net, rec_loss, trust_loss, n_rec_samples, n_paths):
    """main_auto_expert_s.py:76-82: homoscedastic-uncertainty weighting of the two task losses."""
    w = net.task_weights
    return (torch.exp(-2 * w[0]) * rec_loss + torch.exp(-2 * w[1]) * trust_loss
            + 2 * (5 + 1) * n_rec_samples * w[0] + n_paths * w[1])


def run():
    utils.set_seed(cli.seed)
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    rec_data = dataloader.Loader(cli)
    rec_batches = DataLoader(LightTrainData(rec_data.rec_train_data, rec_data.m_item, rec_data.train_mat),
                             batch_size=256, shuffle=True)
    raw_train = load_trust("train.txt")
    by_user = paths_by_first_user(raw_train[0])
    trust_train = Data(raw_train, rec_data.n_users, shuffle=False)
    trust_test = Data(load_trust("test2.txt"), rec_data.n_users, shuffle=False, test=True)
    cap = 3 * max(1, len(raw_train[0]) // len(rec_batches))
    net = model.LightGCN(cli, rec_data).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=cli.lr)

    for epoch in range(cli.epochs):
        rec_batches.dataset.ng_sample()
        net.train()
        sums = [0.0, 0.0]
        for user, item, label in rec_batches:
            opt.zero_grad()
            chosen = [k for u in set(user.numpy().tolist()) for k in by_user[u]]
            if len(chosen) > cap:
                chosen = random.sample(chosen, cap)
            l_rec, l_trust = net(users=user.to(dev), items=item.to(dev), labels=label.to(dev),
                                 slice_indices=np.array(chosen, dtype=int), trust_data=trust_train, flag=0)
            weighted_loss(net, l_rec, l_trust, len(user), len(chosen)).backward()
            sums[0] += l_rec.item()
            sums[1] += l_trust.item()
            opt.step()
        print("%d,%.5f,%.5f" % (epoch, sums[0], sums[1]))
        net.eval()
        with torch.no_grad():
            r = rec_test(net, rec_data.testRatings, rec_data.testNegatives)
            print("Rec:  Epoch %d : recall=%s ndcg=%s" % (epoch, r["recall"].round(4).tolist(), r["ndcg"].round(4).tolist()))
            print("Trust:Epoch %d : recall=[%.4f, %.4f, %.4f],  ndcg=[%.4f, %.4f, %.4f]"
                  % ((epoch,) + tuple(trust_test5(net, trust_test))))


if __name__ == "__main__":
    run()
