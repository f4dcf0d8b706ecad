"""Test driver written against the reference's import names and call order for the dual-task (recommendation +
trust-path) run, LightGCN_SPEX/code/main_auto_expert_s.py:2-160 — what an unmodified reference driver looks like to
the drop-in modules (both trust pickles are read from --data_path; the reference hard-codes '../data/' for one, :41).
Run through `python -m spex_amd.dropin`.
"""
from lg_parser import parse_args_r

args = parse_args_r()

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

utils.set_seed(args.seed)
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
dataset = dataloader.Loader(args)
train_dataset = LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
train_loader = DataLoader(train_dataset, batch_size=256, shuffle=True)
train_data2 = pickle.load(open(args.data_path + args.dataset + "/trust/train.txt", "rb"))
test_data2 = pickle.load(open(args.data_path + args.dataset + "/trust/test2.txt", "rb"))
user_path_indx = defaultdict(list)
path = train_data2[0]
for i, p in zip(range(len(path)), path):
    user_path_indx[p[0]].append(i)
train_data2 = Data(train_data2, dataset.n_users, shuffle=False)
test_data2 = Data(test_data2, dataset.n_users, shuffle=False, test=True)
trust_batch_size = max(1, len(path) // len(train_loader))

Recmodel = model.LightGCN(args, dataset).to(device)
optimizer = torch.optim.Adam(Recmodel.parameters(), lr=args.lr)


def Train(epoch):
    train_loader.dataset.ng_sample()
    Recmodel.train()
    total1 = total2 = 0.0
    for data in train_loader:
        optimizer.zero_grad()
        user, item, label = data
        path_index = []
        for u in set(user.numpy().tolist()):
            path_index.extend(user_path_indx[u])
        if len(path_index) > trust_batch_size * 3:
            path_index = random.sample(path_index, trust_batch_size * 3)
        loss1, loss2 = Recmodel(users=user.to(device), items=item.to(device), labels=label.to(device),
                                slice_indices=np.array(list(path_index), dtype=int), trust_data=train_data2, flag=0)
        T, n_rec, T_rec = len(path_index), 5, len(user)
        precision1 = torch.exp(-2 * Recmodel.task_weights[0])
        precision2 = torch.exp(-2 * Recmodel.task_weights[1])
        loss = (precision1 * loss1 + precision2 * loss2 + 2 * (n_rec + 1) * T_rec * Recmodel.task_weights[0]
                + T * Recmodel.task_weights[1])
        loss.backward()
        total1 += loss1.item()
        total2 += loss2.item()
        optimizer.step()
    print("%d,%.5f,%.5f" % (epoch, total1, total2))


def Test(epoch):
    Recmodel.eval()
    with torch.no_grad():
        ret = rec_test(Recmodel, dataset.testRatings, dataset.testNegatives)
        print("Rec:  Epoch %d : recall=%s ndcg=%s" % (epoch, ret["recall"].round(4).tolist(), ret["ndcg"].round(4).tolist()))
        r = trust_test5(Recmodel, test_data2)
        print("Trust:Epoch %d : recall=[%.4f, %.4f, %.4f],  ndcg=[%.4f, %.4f, %.4f]" % ((epoch,) + tuple(r)))


if __name__ == "__main__":
    for epoch in range(args.epochs):
        Train(epoch)
        Test(epoch)
