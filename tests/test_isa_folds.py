"""ISA-level guard for the two lock-free in-kernel folds (CPU test: hipcc cross-compiles gfx950 without a GPU).

Both folds publish a share with agent-scope (sc1, write-through) stores and then take a ticket with a relaxed RMW; the LAST
arriver folds every share.  That is only correct if each publisher's stores are ACKNOWLEDGED before its ticket RMW is issued:
an explicit `s_waitcnt vmcnt(0)` (a workgroup-scope release fence emits no vmcnt wait on gfx950; round 3 shipped without it in
trust.hip and with an accidental one in spmm.hip — VERDICT r03 weak #1).  This test compiles the two files with the Makefile's
flags and asserts the wait in the text of the compiled kernels, so a compiler or source change cannot silently remove it.

Semantics at stake: the logits over the whole user table (reference utility1/model_expert_s.py:128-148) and the SpMM row sums
(utility1/model.py:83-92)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spex_amd", "csrc")


def _makefile_var(name):
    for line in open(os.path.join(CSRC, "Makefile")):
        m = re.match(r"%s\s*\?=\s*(.*)" % name, line)
        if m:
            return m.group(1).strip()
    raise AssertionError(name)


def _compile_to_isa(src, tmp_path):
    hipcc = _makefile_var("HIPCC")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not present")
    flags = _makefile_var("FLAGS").replace("$(ARCH)", _makefile_var("ARCH")).split()
    out = str(tmp_path / (src + ".s"))
    subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out], check=True, cwd=CSRC)
    return open(out).read().splitlines()


def _kernels(lines):
    """{mangled name: [instruction lines]} for every function in the listing."""
    out, name, body = {}, None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            out[name] = body
        elif name is not None:
            if ln.startswith(".Lfunc_end"):
                name = None
            else:
                body.append(ln.strip())
    return out


_SC1_STORE = re.compile(r"^global_store_dword\w*\s.*\bsc1\b")
_VM0 = re.compile(r"^s_waitcnt\b.*vmcnt\(0\)")
_TICKET = re.compile(r"^global_atomic_(cmpswap|add)_x2\b")


def _check_publish_then_ticket(body, need_barrier):
    """Between the last sc1 store textually in front of the first ticket RMW and that RMW: an s_waitcnt vmcnt(0) that follows
    every such store (and, when the ticket is taken by one thread for the workgroup, an s_barrier after the wait)."""
    first_rmw = next(i for i, ins in enumerate(body) if _TICKET.match(ins))
    stores = [i for i in range(first_rmw) if _SC1_STORE.match(body[i])]
    assert stores, "no agent-scope publishing store in front of the ticket"
    last_store = stores[-1]
    between = body[last_store + 1:first_rmw]
    waits = [k for k, ins in enumerate(between) if _VM0.match(ins)]
    assert waits, "no s_waitcnt vmcnt(0) between the publishing stores and the ticket RMW:\n" + "\n".join(between)
    if need_barrier:
        bars = [k for k, ins in enumerate(between) if ins.startswith("s_barrier")]
        assert bars and bars[-1] > waits[0], "no s_barrier between the stores' acknowledgement and the ticket"
    # the wait must come BEFORE the first ticket access of any kind (the relaxed load of the ticket word included)
    first_ticket_load = next((k for k, ins in enumerate(between) if re.match(r"^global_load_dwordx2\b.*\bsc1\b", ins)), len(between))
    assert waits[0] < first_ticket_load, "the ticket word is read before the publishing stores are acknowledged"
    return last_store, first_rmw


def test_trust_split_kernel_waits_for_its_stores_before_the_ticket(tmp_path):
    ks = _kernels(_compile_to_isa("trust.hip", tmp_path))
    split = [k for k in ks if "trust_path_split_kernel" in k]
    assert len(split) == 1, split
    body = ks[split[0]]
    last_store, rmw = _check_publish_then_ticket(body, need_barrier=True)
    # the share is three kinds of sc1 stores (raw scores in the sweep, part_da2, part_ms): all textually in front of the wait
    n_sc1 = sum(1 for i in range(rmw) if _SC1_STORE.match(body[i]))
    assert n_sc1 >= 4, n_sc1
    # kernels without a ticket must not have grown one (the fold is the split form's only)
    for k, b in ks.items():
        if "trust_path_split_kernel" not in k:
            assert not any(_TICKET.match(i) for i in b), k


def test_spmm_hub_fold_waits_for_its_partial_row_before_the_ticket(tmp_path):
    ks = _kernels(_compile_to_isa("spmm.hip", tmp_path))
    folded = {k: b for k, b in ks.items() if any(_TICKET.match(i) for i in b)}
    # spmm_chunk_kernel<EPI, MASKED, ROWIDS, FOLD>: 12 instantiations with the fold (unmasked and edge-dropout alike), 6 edge-dropout
    # ones without (graphs that have no hub), and nothing else in the file takes a ticket
    with_fold = [k for k in ks if re.search(r"spmm_chunk_kernelILi\dELb[01]ELb[01]ELb1EE", k)]
    without = [k for k in ks if re.search(r"spmm_chunk_kernelILi\dELb[01]ELb[01]ELb0EE", k)]
    assert len(with_fold) == 12 and len(without) == 6, (with_fold, without)
    assert all(re.search(r"spmm_chunk_kernelILi\dELb1E", k) for k in without), "an unmasked instantiation lost its fold"
    assert sorted(folded) == sorted(with_fold), sorted(set(folded) ^ set(with_fold))
    for k, body in folded.items():
        _check_publish_then_ticket(body, need_barrier=False)
