"""Device helpers with the reference's names (LightGCN_SPEX/code/utility1/gpuutil.py): a tensor or module goes to the
GPU / back to the host when a GPU is present and is returned untouched otherwise."""
import torch


def _moved(obj, target):
    return obj.to(target) if torch.cuda.is_available() else obj


def trans_to_cuda(variable):
    return _moved(variable, "cuda")


def trans_to_cpu(variable):
    return _moved(variable, "cpu")
