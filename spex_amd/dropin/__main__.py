"""Run an unmodified reference driver against the MI355X modules.

    cd <reference>/LightGCN_SPEX/code && python -m spex_amd.dropin main_rec.py --dataset epinion2 --recdim 64 --layer 3

The driver's own imports (`from lg_parser import parse_args_r`, `import utility1.dataloader as dataloader`,
`import utility1.model as model`, `from utility1.batch_test import test`, main_rec.py:2-13) then resolve to
spex_amd/dropin/ because it is placed ahead of the script's directory on sys.path; anything this package does not
provide (e.g. utility2 for the dual-task drivers) still resolves to the driver's own directory.
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
    sys.path[:] = [here] + [p for p in sys.path if p not in ("", here)] + [os.path.dirname(script)]
    if repo_root not in sys.path:
        sys.path.insert(1, repo_root)
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
