"""The helpers of NGCF_SPEX/code/utility/helper.py the drivers use: device moves (:57-69) and early stopping (:38-55)."""
import os

import torch


def trans_to_cuda(variable):
    return variable.cuda() if torch.cuda.is_available() else variable


def trans_to_cpu(variable):
    return variable.cpu() if torch.cuda.is_available() else variable


def ensureDir(dir_path):
    os.makedirs(os.path.dirname(dir_path), exist_ok=True)


def early_stopping(log_value, best_value, stopping_step, expected_order="acc", flag_step=100):
    """Returns (best_value, stopping_step, should_stop): the counter resets whenever the value does not get worse."""
    if expected_order not in ("acc", "dec"):
        raise AssertionError(expected_order)
    better = log_value >= best_value if expected_order == "acc" else log_value <= best_value
    if better:
        return log_value, 0, False
    stopping_step += 1
    stop = stopping_step >= flag_step
    if stop:
        print("Early stopping is trigger at step: {} log:{}".format(flag_step, log_value))
    return best_value, stopping_step, stop
