#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Two mints of the SAME reference run (oracle/gen_golden.py --stage ngcf-epochs-epinion2: NGCF_SPEX
main_rec.py's epoch 0 on Epinion2, same seeds, same injected dropout masks, 8 vs 4 CPU threads) do not agree: torch's CPU
index_put_(accumulate) and its threaded reductions are not run-to-run deterministic, and 4 757 Adam steps amplify the last
bits.  This script stores both mints' trajectory figures side by side in tests/golden/ngcf_epinion2_ref_spread.npz so that the
whole-epoch parity test can state the reference's OWN run-to-run spread next to the build's deviation from it.

  python oracle/record_reference_spread.py <older ngcf_epinion2_epochs.npz>     # e.g. `git show <rev>:tests/golden/...` > file
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


def main():
    old = np.load(sys.argv[1])
    new = np.load(os.path.join(GOLD, "ngcf_epinion2_epochs.npz"))
    assert int(old["n_steps"]) == int(new["n_steps"]) and (old["sample_sha"] == new["sample_sha"]).all()   # the same run
    assert np.array_equal(old["eval_steps"], new["eval_steps"])
    out = dict(eval_steps=new["eval_steps"])
    for k in ("losses", "eval_metrics", "recall", "ndcg", "step_losses", "user_w", "item_w", "final_GC_Linear_list__0__weight",
              "final_Bi_Linear_list__0__weight"):
        out[k + "_a"], out[k + "_b"] = old[k], new[k]
    assert np.array_equal(old["rows_u"], new["rows_u"]) and np.array_equal(old["rows_i"], new["rows_i"])
    np.savez_compressed(os.path.join(GOLD, "ngcf_epinion2_ref_spread.npz"), **out)
    print("loss sums", old["losses"], new["losses"], "rel", abs(old["losses"][0] - new["losses"][0]) / new["losses"][0])
    print("metric spread at the evaluated steps", np.abs(old["eval_metrics"] - new["eval_metrics"]).max(1),
          "end of epoch", max(np.abs(old["recall"] - new["recall"]).max(), np.abs(old["ndcg"] - new["ndcg"]).max()))


if __name__ == "__main__":
    main()
