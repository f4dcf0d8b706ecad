"""NGCF driver for the drop-in launcher test.

It touches the package exactly where NGCF_SPEX/code/main_rec.py does — `from ngcf_parser import parse_args`,
`from utility.helper import *`, `from utility.batch_test import test, data_generator, args`, the three adjacency
matrices from `data_generator.get_adj_mat()`, a `Model_Wrapper(data_config=..., device=...)`, torch Adam, per epoch
`data_generator.load_train_data()` + the shuffled batches + `test(model, users_to_test, drop_flag=True)` — with the one
change a maintainer makes to run on libspexhip: the model class comes from spex_amd.ngcf instead of being defined in
the driver (main_rec.py:36-113).
Run it as `python -m spex_amd.dropin tests/drivers/ngcf_driver.py --data_path <root> --dataset small --epoch 3`.
"""
from ngcf_parser import parse_args

cli = parse_args()

import random

import numpy as np
import torch

from spex_amd.ngcf import Model_Wrapper
from utility.batch_test import args, data_generator, test
from utility.helper import *  # noqa: F401,F403  (trans_to_cuda)


def seed_everything(seed):
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)


def run():
    seed_everything(2020)
    data_generator.print_statistics()
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    plain_adj, norm_adj, mean_adj = data_generator.get_adj_mat()
    config = {"n_users": data_generator.n_users, "n_items": data_generator.n_items, "norm_adj": norm_adj}
    net = Model_Wrapper(data_config=config, device=dev).to(dev)
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
        net.eval()
        ret = test(net, list(data_generator.test_set.keys()), drop_flag=True)
        print("epoch %d loss %.5f recall=%s ndcg=%s" % (epoch, total, ret["recall"].round(4).tolist(), ret["ndcg"].round(4).tolist()))


if __name__ == "__main__":
    run()
