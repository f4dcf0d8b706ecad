"""Device helpers with the reference's names (LightGCN_SPEX/code/utility1/gpuutil.py)."""
import torch


def trans_to_cuda(variable):
    return variable.cuda() if torch.cuda.is_available() else variable


def trans_to_cpu(variable):
    return variable.cpu() if torch.cuda.is_available() else variable
