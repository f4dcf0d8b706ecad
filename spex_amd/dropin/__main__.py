"""Run an unmodified reference driver against the MI355X modules.

    cd <reference>/LightGCN_SPEX/code && python -m spex_amd.dropin main_rec.py --dataset epinion2 --recdim 64 --layer 3

The driver's own imports (`from lg_parser import parse_args_r`, `import utility1.dataloader as dataloader`,
`import utility1.model as model`, `from utility1.batch_test import test`, main_rec.py:2-13) then resolve to
spex_amd/dropin/ because it is placed ahead of the script's directory on sys.path; anything this package does not
provide still resolves to the driver's own directory.  A driver that imports `ngcf_parser` (NGCF_SPEX/code/main_*.py)
gets spex_amd/dropin/ngcf/ instead — its `utility` package is NGCF's (load_data, batch_test, helper, ...), not the
LightGCN alias of the same name — and, because NGCF's model class is defined inside the driver itself, the operator hook of
spex_amd/dropin/sparse_hook.py: the driver's own `torch.sparse.mm(self.norm_adj.to(self.device), ego)` (main_rec.py:76) then
runs on spex_spmm_f32 (SPEX_SPARSE_MM_HOOK=0 turns the hook off).

Before the script starts, the heavy imports it will make anyway (torch, numpy, scipy, pandas) are done here and frozen
out of Python's cyclic garbage collector: a full collection over their import-time objects takes ~40 ms and otherwise
lands several times in every epoch of the driver's 4 906-step loop (tools/stall_probe.py).
"""
import os
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        sys.exit("usage: python -m spex_amd.dropin <main script> [script flags...]")
    script = os.path.abspath(sys.argv[1])
    here = os.path.dirname(os.path.abspath(__file__))
    repo_root = os.path.dirname(os.path.dirname(here))
    sys.argv = [script] + sys.argv[2:]
    with open(script) as fh:
        ngcf = "ngcf_parser" in fh.read()
    root = os.path.join(here, "ngcf") if ngcf else here
    sys.path[:] = [root] + [p for p in sys.path if p not in ("", here, root)] + [os.path.dirname(script)]
    if repo_root not in sys.path:
        sys.path.insert(1, repo_root)
    import gc
    import numpy, pandas, scipy.sparse, torch, torch.utils.data   # noqa: F401,E401  (what the drivers and the modules import)
    if ngcf and os.environ.get("SPEX_SPARSE_MM_HOOK", "1") != "0":
        # NGCF's model lives in the driver (main_rec.py:36-113): its torch.sparse.mm(norm_adj.to(device), ego) lands on the HIP
        # SpMM through the operator hook, with the driver file untouched (spex_amd/dropin/sparse_hook.py)
        from spex_amd.dropin import sparse_hook
        sparse_hook.install()
    gc.collect()
    gc.freeze()
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
