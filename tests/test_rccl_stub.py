"""The native exchange's arithmetic without a wire (VERDICT r03 missing #1): libspexhip bound to a RECORDING stand-in for RCCL
(tests/stubs/rccl_record_stub.c through SPEX_RCCL_LIB) in a child process (the binding happens once per process), which asserts
the exact nccl* call sequence — see tests/drivers/rccl_stub_driver.py.  RCCL itself does not run here; multi-GPU row 8e stays
"unmeasured on hardware"."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(mode, tmp_path):
    so = str(tmp_path / "librccl_record_stub.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Werror", "-o", so, os.path.join(HERE, "stubs", "rccl_record_stub.c")], check=True)
    env = dict(os.environ, SPEX_RCCL_LIB=so)
    env.pop("SPEX_COMM_NO_SHORTCUT", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "drivers", "rccl_stub_driver.py"), mode], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rccl stub (%s): ok" % mode in r.stdout


def test_native_exchange_call_sequence_world_2_4_8_without_a_wire(tmp_path):
    """CPU: every rank of world 2 / 4 / 8 with uneven rows_per_rank (a peer without rows, a rank without rows): one group of
    world - 1 sends + receives, counts = rows_per_rank[q] * d, receive offsets = q * max_rows * d, the caller's stream; the
    equal-shard form = one ncclAllGather; the all-reduce in place; a failing ncclSend / ncclRecv closes the group."""
    _run("cpu", tmp_path)


@pytest.mark.gpu
def test_partitioned_step_call_sequence_world_4_rank_2_on_the_recording_stub(tmp_path):
    """GPU (the step's kernels run for real, the wire is absent): every exchange IN PLACE (sent from the rank's own slot of the table
    it is received around, the two tables alternating), every call on the caller's stream, both exchange forms, a switch of the form
    between steps; per step one all-reduce and — fast path — 2L - 1 exchanges (none in front of the backward's first product: it is
    the push; the first backward exchange is the push target's table), — deterministic mode / fast=False — 2L.  The one-call
    partitioned DUAL-TASK step alike: the first exchange into the kept table (E^0), ONE all-reduce of 4B rows, no gate-gradient
    collective."""
    _run("gpu", tmp_path)
