"""Recommendation driver for the drop-in launcher test.

It touches the package exactly where LightGCN_SPEX/code/main_rec.py does — the same imports by the same names,
set_seed -> Loader(args) -> LightTrainData -> DataLoader(256, shuffle) -> model.LightGCN(args, dataset).to(device) ->
Adam; per epoch ng_sample(), one forward(flag=0) / backward / step per batch with `loss.item()` accumulated, then
test() under eval() and no_grad() — written as two small functions.
Run it as `python -m spex_amd.dropin tests/drivers/rec_driver.py --dataset tiny --data_path <root> --epochs 2`.
"""
from lg_parser import parse_args_r

cli = parse_args_r()

import torch
from torch.utils.data import DataLoader

import utility1.dataloader as dataloader
import utility1.model as model
import utility1.utils as utils
from utility1.batch_test import test
from utility1.dataloader import LightTrainData

utils.set_seed(cli.seed)
dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
rec_data = dataloader.Loader(cli)
train_loader = DataLoader(LightTrainData(rec_data.rec_train_data, rec_data.m_item, rec_data.train_mat), batch_size=256,
                          shuffle=True)
Recmodel = model.LightGCN(cli, rec_data).to(dev)
opt = torch.optim.Adam(Recmodel.parameters(), lr=cli.lr)


def run_epoch(epoch):
    train_loader.dataset.ng_sample()
    Recmodel.train()
    seen = 0.0
    for user, item, label in train_loader:
        opt.zero_grad()
        batch_loss = Recmodel(users=user.to(dev), items=item.to(dev), labels=label.to(dev), flag=0)
        batch_loss.backward()
        opt.step()
        seen += batch_loss.item()
    print("%d,%.5f" % (epoch, seen))


def evaluate(epoch):
    Recmodel.eval()
    with torch.no_grad():
        r = test(Recmodel, rec_data.testRatings, rec_data.testNegatives)
    print("Rec:  Epoch %d : recall=%s ndcg=%s" % (epoch, r["recall"].round(4).tolist(), r["ndcg"].round(4).tolist()))
    return r


if __name__ == "__main__":
    for epoch in range(cli.epochs):
        run_epoch(epoch)
        evaluate(epoch)
