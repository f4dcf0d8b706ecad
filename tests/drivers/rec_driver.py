"""Test driver written against the reference's import names and call order (LightGCN_SPEX/code/main_rec.py:2-37,50):
it is what an unmodified reference driver looks like to the drop-in modules.  Run through `python -m spex_amd.dropin`.
"""
from lg_parser import parse_args_r

args = parse_args_r()

import torch
from torch.utils.data import DataLoader

import utility1.dataloader as dataloader
import utility1.model as model
import utility1.utils as utils
from utility1.batch_test import test
from utility1.dataloader import LightTrainData

utils.set_seed(args.seed)
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
dataset = dataloader.Loader(args)
train_dataset = LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
train_loader = DataLoader(train_dataset, batch_size=256, shuffle=True)
Recmodel = model.LightGCN(args, dataset).to(device)
optimizer = torch.optim.Adam(Recmodel.parameters(), lr=args.lr)


def run_epoch(epoch):
    train_loader.dataset.ng_sample()
    Recmodel.train()
    running = 0.0
    for user, item, label in train_loader:
        optimizer.zero_grad()
        loss = Recmodel(users=user.to(device), items=item.to(device), labels=label.to(device), flag=0)
        loss.backward()
        optimizer.step()
        running += loss.item()
    print("%d,%.5f" % (epoch, running))


def evaluate(epoch):
    Recmodel.eval()
    with torch.no_grad():
        ret = test(Recmodel, dataset.testRatings, dataset.testNegatives)
    print("Rec:  Epoch %d : recall=%s ndcg=%s" % (epoch, ret["recall"].round(4).tolist(), ret["ndcg"].round(4).tolist()))
    return ret


if __name__ == "__main__":
    for epoch in range(args.epochs):
        run_epoch(epoch)
        evaluate(epoch)
