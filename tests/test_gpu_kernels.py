"""GPU parity tests proper: every libspexhip kernel, called through the C ABI (ctypes via spex_amd), against the CPU
oracle on the same seeded inputs and against the golden vectors minted from the reference.

Tolerances: the north star asks per-layer embeddings within 1e-5 relative in fp32; the SpMM's fmaf chain is the
oracle's, so rows handled by one wave are required to match BIT-FOR-BIT, and only rows cut into segments (> 64
entries on the tuned d = 64 / 128 / 256 kernel, > 128 on the generic one) may differ by the re-association of their partial sums
(<= 1e-6 relative here).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def random_csr(rng, n_rows, n_cols, degrees):
    rowptr = np.zeros(n_rows + 1, np.int64)
    cols = []
    for r in range(n_rows):
        k = min(int(degrees[r]), n_cols)
        cols.append(np.sort(rng.choice(n_cols, k, replace=False)))
        rowptr[r + 1] = rowptr[r] + k
    col = np.concatenate(cols).astype(np.int32) if cols else np.zeros(0, np.int32)
    val = rng.normal(size=len(col)).astype(np.float32)
    return rowptr.astype(np.int32), col, val


@pytest.fixture(scope="module")
def G():
    from spex_amd.graph import SpexGraph
    return SpexGraph


# ---------------------------------------------------------------------------------------------- SpMM
def test_library_is_the_hip_build():
    from spex_amd import _lib
    assert _lib.load().spex_version() == 5


@pytest.mark.parametrize("d", [64, 32, 100, 128, 256, 1])
def test_spmm_matches_oracle_random_graph(G, oracle, d):
    rng = np.random.default_rng(d)
    n_rows, n_cols = 700, 500
    deg = rng.integers(0, 60, n_rows)
    deg[::50] = 0                      # empty rows
    deg[7], deg[300] = 480, 129        # long rows -> segments (+ ragged last segment)
    deg[301] = 128                     # exactly at the threshold: still one wave
    rowptr, col, val = random_csr(rng, n_rows, n_cols, deg)
    X = rng.normal(size=(n_cols, d)).astype(np.float32)
    g = G(rowptr, col, val, n_cols=n_cols)
    assert g.n_long_rows == 2 and g.n_segments == 8 + 3          # segments of 64 entries (kSegLen)
    Y = g.spmm(t(X)).cpu().numpy()
    ref = oracle.spmm(rowptr, col, val, X)
    short = np.diff(rowptr) <= (64 if d in (64, 128, 256) else 128)  # rows owned by a single wave
    assert np.array_equal(Y[short], ref[short])                    # bit-exact fmaf chain
    assert rel_err(Y[~short], ref[~short]) <= 1e-6
    assert np.all(Y[deg == 0] == 0)


def test_spmm_folds_hub_rows_inside_the_launch(G, oracle, monkeypatch):
    """Rows beyond 1 024 entries (hubs: 64-entry segments on many waves).  The d == 64 kernel folds them INSIDE the launch — 16
    adjacent segments per workgroup summed through LDS, one partial row per group, the hub's last group to arrive adds the groups in
    order — instead of leaving them to spmm_long_fixup_kernel (SPEX_HUB_FOLD=0, and the wide kernels).  Several hubs whose groups
    straddle workgroups (1 025 entries = 17 segments; 2 000; 6 812; 20 000 = 313 segments in 20 groups), every epilogue form:
    against the oracle / the fp64 sum, against the fix-up form, and bit for bit against itself over repeated launches on the same
    handle (the tickets carry a per-launch tag; nothing is reset in between).  Edge-dropout launches fold the same way."""
    rng = np.random.default_rng(11)
    n_rows, n_cols = 900, 24000
    deg = rng.integers(0, 50, n_rows)
    deg[3], deg[4], deg[400], deg[401], deg[899] = 1025, 2000, 6812, 20000, 1100
    deg[5] = 1024                                   # the longest row that still folds inside ONE workgroup
    rowptr, col, val = random_csr(rng, n_rows, n_cols, deg)
    X = rng.normal(size=(n_cols, 64)).astype(np.float32)
    add, acc = (rng.normal(size=(n_rows, 64)).astype(np.float32) for _ in range(2))
    g = G(rowptr, col, val, n_cols=n_cols)
    ref = oracle.spmm(rowptr, col, val, X)
    hubs = np.diff(rowptr) > 1024

    def forms():
        out = [g.spmm(t(X)).clone()]
        Y, A = torch.empty(n_rows, 64, device=DEV), torch.empty(n_rows, 64, device=DEV)
        g.spmm(t(X), Y=Y, acc_in=t(acc), acc_out=A, acc_div=4.0)
        out += [Y.clone(), A.clone()]
        Y2 = torch.empty(n_rows, 64, device=DEV)
        g.spmm(t(X), Y=Y2, add_in=t(add), add_div=3.0)
        out.append(Y2.clone())
        return out
    monkeypatch.setenv("SPEX_HUB_FOLD", "0")
    fix = forms()
    monkeypatch.delenv("SPEX_HUB_FOLD")
    runs = [forms() for _ in range(10)]
    torch.cuda.synchronize()
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)
    y = runs[0][0].cpu().numpy()
    # (hub rows: up to 20 000 N(0, 1) terms in fp32 — the oracle's one sequential chain and the kernel's segment / group sums round
    #  differently by a few 1e-6 of the largest output; the fp64 sum is the arbiter: the kernel is no further from it than the oracle)
    exact = np.zeros((n_rows, 64))
    for r in np.nonzero(hubs)[0]:
        sl = slice(rowptr[r], rowptr[r + 1])
        exact[r] = val[sl].astype(np.float64) @ X[col[sl]].astype(np.float64)
    assert rel_err(y[hubs], exact[hubs]) <= max(2e-6, 1.5 * rel_err(ref[hubs], exact[hubs]))
    assert rel_err(y[~hubs], ref[~hubs]) <= 3e-6            # (incl. the 1 024-entry row: 16 segment sums re-associated)
    for a, b in zip(runs[0], fix):                                   # the two forms differ in the association of a hub's sum only
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.array_equal(a[~hubs], b[~hubs])
        assert rel_err(a[hubs], b[hubs]) <= 1e-5
    assert rel_err(runs[0][2].cpu().numpy(), (acc + ref) / np.float32(4.0)) <= 1e-5
    assert rel_err(runs[0][3].cpu().numpy(), ref + add / np.float32(3.0)) <= 1e-5
    # edge dropout (reference README.md:119-123 / utility1/model.py:46-55): the masked launch folds its hubs in the launch too
    # (spmm_chunk_kernel<.., MASKED, .., FOLD> — taken only on a graph that has hubs); every epilogue form: fold vs fix-up vs oracle
    keep = (rng.random(len(col)) < 0.4).astype(np.uint8)
    g.set_edge_mask(1, t(keep), 0.4, 0)
    masked = [forms() for _ in range(4)]
    monkeypatch.setenv("SPEX_HUB_FOLD", "0")
    masked_fix = forms()
    monkeypatch.delenv("SPEX_HUB_FOLD")
    g.set_edge_mask(0)
    for m in masked[1:]:
        for a, b in zip(masked[0], m):
            assert torch.equal(a, b)
    vm = np.where(keep != 0, val / np.float32(0.4), np.float32(0.0)).astype(np.float32)
    ref_m = oracle.spmm(rowptr, col, vm, X)
    exact_m = np.zeros((n_rows, 64))
    for r in np.nonzero(hubs)[0]:
        sl = slice(rowptr[r], rowptr[r + 1])
        exact_m[r] = vm[sl].astype(np.float64) @ X[col[sl]].astype(np.float64)
    ym = masked[0][0].cpu().numpy()
    assert rel_err(ym[hubs], exact_m[hubs]) <= max(2e-6, 1.5 * rel_err(ref_m[hubs], exact_m[hubs]))
    assert rel_err(ym[~hubs], ref_m[~hubs]) <= 3e-6
    for a, b in zip(masked[0], masked_fix):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.array_equal(a[~hubs], b[~hubs])
        assert rel_err(a[hubs], b[hubs]) <= 1e-5
    assert rel_err(masked[0][2].cpu().numpy(), (acc + ref_m) / np.float32(4.0)) <= 1e-5
    assert rel_err(masked[0][3].cpu().numpy(), ref_m + add / np.float32(3.0)) <= 1e-5


def test_hub_fold_survives_replay_from_a_captured_graph(G):
    """A launch replayed from a captured HIP graph carries the SAME ticket tag every time: the hub's last group puts the ticket back
    to 0, so every replay folds again (without that, the second replay would never see a 'last' group and the hub rows would keep
    their first result).  Capture one product on a graph with hubs, change X between replays, compare with eager launches."""
    rng = np.random.default_rng(12)
    n_rows, n_cols = 300, 9000
    deg = rng.integers(1, 40, n_rows)
    deg[2], deg[150] = 3000, 8000
    rowptr, col, val = random_csr(rng, n_rows, n_cols, deg)
    g = G(rowptr, col, val, n_cols=n_cols)
    X, Y = torch.zeros(n_cols, 64, device=DEV), torch.zeros(n_rows, 64, device=DEV)
    xs = [t(rng.normal(size=(n_cols, 64)).astype(np.float32)) for _ in range(3)]
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        g.spmm(X, Y=Y)
    got = []
    for x in xs:
        X.copy_(x)
        cg.replay()
        got.append(Y.clone())
    torch.cuda.synchronize()
    for x, y in zip(xs, got):
        assert torch.equal(g.spmm(x), y)


def test_spmm_fused_epilogues(G, oracle):
    rng = np.random.default_rng(1)
    n = 400
    rowptr, col, val = random_csr(rng, n, n, rng.integers(0, 40, n))
    X, add, acc = (rng.normal(size=(n, 64)).astype(np.float32) for _ in range(3))
    g = G(rowptr, col, val)
    y = oracle.spmm(rowptr, col, val, X)
    Y, A = torch.empty(n, 64, device=DEV), torch.empty(n, 64, device=DEV)
    g.spmm(t(X), Y=Y, add_in=t(add), add_div=3.0, acc_in=t(acc), acc_out=A, acc_div=4.0)
    y2 = y + add / np.float32(3.0)
    assert np.array_equal(Y.cpu().numpy(), y2)
    assert np.array_equal(A.cpu().numpy(), (acc + y2) / np.float32(4.0))
    # acc only (last layer), in place
    A2 = t(acc)
    g.spmm(t(X), acc_in=A2, acc_out=A2, acc_div=1.0)
    assert np.array_equal(A2.cpu().numpy(), acc + y)


def test_spmm_edge_cases(G, oracle):
    from spex_amd._lib import SpexError
    # empty matrix, matrix without entries, single huge row
    g0 = G(np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), n_cols=5)
    assert g0.spmm(torch.zeros(5, 64, device=DEV)).shape == (0, 64)
    g1 = G(np.zeros(11, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), n_cols=3)
    assert torch.all(g1.spmm(torch.ones(3, 64, device=DEV)) == 0)
    rng = np.random.default_rng(2)
    rowptr, col, val = random_csr(rng, 3, 5000, [0, 5000, 1])
    X = rng.normal(size=(5000, 64)).astype(np.float32)
    g2 = G(rowptr, col, val, n_cols=5000)
    got, ref = g2.spmm(t(X)).cpu().numpy(), oracle.spmm(rowptr, col, val, X)
    # a 5000-term sum of random-sign products: the segmented order and the sequential order are both ~sqrt(n) eps
    # away from the exact sum; require the north-star 1e-5 and that the GPU is no further from fp64 than the CPU is
    assert rel_err(got, ref) <= 1e-5
    import scipy.sparse as sp
    exact = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(3, 5000)) @ X.astype(np.float64)
    assert rel_err(got, exact) <= 2.0 * rel_err(ref, exact) + 1e-7
    # NaN / Inf in an unrelated source row must not leak (no 0 * Inf from padded lanes)
    X[4999] = np.inf
    rowptr, col, val = random_csr(rng, 50, 4999, rng.integers(1, 30, 50))
    g3 = G(rowptr, col, val, n_cols=5000)
    assert np.isfinite(g3.spmm(t(X)).cpu().numpy()).all()
    with pytest.raises(ValueError):
        g3.spmm(torch.zeros(10, 64, device=DEV))                   # wrong shape is caught on the host
    with pytest.raises(SpexError):
        Xa = t(X[:50].copy())
        G(*random_csr(rng, 50, 50, [1] * 50)).spmm(Xa, Y=Xa)       # aliasing refused


def test_spmm_linearity_and_transpose_identity_at_epinion2_size(G, golden, epinion2):
    """Size-independent properties at the full benchmark size: A(ax + by) = aAx + bAy and <Ax, y> = <x, A^T y>."""
    from spex_amd.graph import lightgcn_norm_adj
    tr = epinion2["train"]
    rowptr, col, val = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
    g = G(rowptr, col, val)
    gen = torch.Generator(device=DEV).manual_seed(0)
    x, y = (torch.randn(15593, 64, device=DEV, generator=gen) for _ in range(2))
    lhs = g.spmm(2.0 * x + 0.5 * y)
    rhs = 2.0 * g.spmm(x) + 0.5 * g.spmm(y)
    assert (lhs - rhs).abs().max().item() <= 1e-5 * rhs.abs().max().item()
    a = (g.spmm(x).double() * y.double()).sum().item()
    b = (x.double() * g.spmm(y).double()).sum().item()       # A symmetric
    assert abs(a - b) <= 1e-6 * max(abs(a), 1.0)
    # rows of A_hat sum to <= sqrt(deg) bound and A 1 is what the host CSR says
    ones = g.spmm(torch.ones(15593, 64, device=DEV))[:, 0].cpu().numpy()
    host = np.add.reduceat(np.r_[val, 0.0].astype(np.float64), np.minimum(rowptr[:-1], len(val)))
    host[np.diff(rowptr) == 0] = 0
    assert np.allclose(ones, host, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------- propagation vs goldens
def _epinion2(golden, epinion2):
    from spex_amd.datasets import epinion2_tables
    from spex_amd.graph import lightgcn_norm_adj
    g = golden("lightgcn_epinion2")
    tr = epinion2["train"]
    csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
    uw, iw = epinion2_tables(3186, 12407)
    return g, csr, np.concatenate([uw, iw])


def test_g2_propagation_tiny_bit_exact_vs_reference(G, golden):
    g = golden("lightgcn_tiny")
    gr = G(g["rowptr"], g["col"], g["val"])
    n, d = g["E0"].shape
    layers = torch.empty(3, n, d, device=DEV)
    out = gr.propagate(t(g["E0"]), 3, layers_out=layers)
    for l in range(3):
        assert np.array_equal(layers[l].cpu().numpy(), g[f"E{l + 1}"])     # == torch.sparse.mm on CPU, bit for bit
    assert np.array_equal(out.cpu().numpy(), g["light_out"])
    out2 = gr.propagate(t(g["E0"]), 3)                                        # ping-pong workspace path
    assert np.array_equal(out2.cpu().numpy(), g["light_out"])


def test_g2_propagation_epinion2_vs_reference(G, golden, epinion2, oracle):
    g, csr, E0 = _epinion2(golden, epinion2)
    gr = G(*csr)
    layers = torch.empty(3, *E0.shape, device=DEV)
    out = gr.propagate(t(E0), 3, layers_out=layers).cpu().numpy()
    rows = g["sample_rows"]
    L = layers.cpu().numpy()
    for l in range(3):
        assert rel_err(L[l][rows], g[f"E{l + 1}_rows"]) <= 1e-5                    # the north-star gate
        assert np.allclose(L[l].astype(np.float64).sum(0), g[f"E{l + 1}_colsum"], rtol=1e-5, atol=1e-6)
        assert np.isclose(np.sqrt((L[l].astype(np.float64) ** 2).sum()), g[f"E{l + 1}_fro"], rtol=1e-6)
    assert rel_err(out[rows], g["light_out_rows"]) <= 1e-5
    # and against the oracle on every row: bit-exact where one wave owns the row
    ref, ref_layers = oracle.propagate_mean(*csr, E0, 3, n_threads=8, return_layers=True)
    short = np.diff(csr[0]) <= 64
    assert np.array_equal(L[0][short], ref_layers[0][short])
    assert rel_err(out, ref) <= 1e-6


def test_g3_g4_training_steps_vs_reference(G, golden, epinion2):
    """Loss, d loss / d E0 and the tables after 1, 2, 5 Adam steps, through the autograd-free stepper."""
    from spex_amd.trainer import LightGCNStepper
    for ds in ("tiny", "epinion2"):
        if ds == "tiny":
            g = golden("lightgcn_tiny")
            csr, E0, n_u = (g["rowptr"], g["col"], g["val"]), g["E0"], int(g["n_user"]) + 1
        else:
            g, csr, E0 = _epinion2(golden, epinion2)
            n_u = 3186
        st = LightGCNStepper(G(*csr), t(E0.copy()), n_u, n_layers=3, lr=1e-3)
        for s in range(5):
            u, i, y = t(g["batch_users"][s]), t(g["batch_items"][s]), t(g["batch_labels"][s].astype(np.float32))
            loss = st.step_bce(u, i, y).item()
            assert abs(loss - float(g["g4_losses"][s])) <= 2e-6
            if s == 0:
                grad = st.grad_E0.cpu().numpy()
                want = g["g3_grad"] if ds == "tiny" else g["g3_grad_rows"]
                got = grad if ds == "tiny" else grad[g["sample_rows"]]
                assert rel_err(got, want) <= 1e-5
                assert np.isclose(np.sqrt((grad.astype(np.float64) ** 2).sum()), g["g3_grad_fro"], rtol=1e-5)
            if s + 1 in (1, 2, 5):
                W = st.E0.cpu().numpy()
                want = g[f"g4_w_step{s + 1}"] if ds == "tiny" else g[f"g4_w_step{s + 1}_rows"]
                got = W if ds == "tiny" else W[g["sample_rows"]]
                assert rel_err(got, want) <= 5e-6
                assert np.allclose(W.astype(np.float64).sum(0), g[f"g4_w_step{s + 1}_colsum"], rtol=1e-5, atol=1e-6)


def test_g9_dropout_injected_mask_and_backward_consistency(G, golden, oracle):
    from spex_amd.graph import csr_transpose
    g = golden("lightgcn_tiny")
    csr = (g["rowptr"], g["col"], g["val"])
    keep_prob = float(g["g9_keep"])
    keep = oracle.dropout_keep_mask(g["g9_rand"], keep_prob)
    gr = G(*csr)
    gr.set_edge_mask(1, t(keep.astype(np.uint8)), keep_prob, 0)
    out = gr.propagate(t(g["E0"]), 3).cpu().numpy()
    assert rel_err(out, g["g9_light_out"]) <= 1e-6                          # vs the reference with the same mask
    assert np.array_equal(out, oracle.propagate_mean_masked(*csr, keep, keep_prob, g["E0"], 3))
    # the transposed handle with the edge-id permutation applies the SAME mask: <A'x, y> == <x, A'^T y>
    t_rowptr, t_col, t_val, eid = csr_transpose(*csr, len(csr[0]) - 1)
    gt = G(t_rowptr, t_col, t_val, edge_id=eid)
    gt.set_edge_mask(1, t(keep.astype(np.uint8)), keep_prob, 0)
    gen = torch.Generator(device=DEV).manual_seed(1)
    x, y = (torch.randn(len(csr[0]) - 1, 64, device=DEV, generator=gen) for _ in range(2))
    a = (gr.spmm(x).double() * y.double()).sum().item()
    b = (x.double() * gt.spmm(y).double()).sum().item()
    scale = (gr.spmm(x.abs()).double() * y.abs().double()).sum().item()
    assert abs(a - b) <= 1e-6 * scale                                       # fp32 products, fp64 reduction
    # and it is a genuinely different operator from the unmasked one
    gr.set_edge_mask(0)
    assert abs((gr.spmm(x).double() * y.double()).sum().item() - a) > 1e-3 * scale
    gr.set_edge_mask(1, t(keep.astype(np.uint8)), keep_prob, 0)
    # sampled (philox) mode: same seed -> same mask in A and A^T; keep rate ~ keep_prob; E[A'] = A
    gr.set_edge_mask(2, None, keep_prob, 1234)
    gt.set_edge_mask(2, None, keep_prob, 1234)
    a = (gr.spmm(x).double() * y.double()).sum().item()
    b = (x.double() * gt.spmm(y).double()).sum().item()
    assert abs(a - b) <= 1e-6 * scale
    gr.set_edge_mask(0)
    assert np.array_equal(gr.propagate(t(g["E0"]), 3).cpu().numpy(), g["light_out"])


def test_dropout_sampled_mask_statistics(G):
    n = 4096
    rowptr = np.arange(n + 1, dtype=np.int32) * 8
    col = (np.arange(n * 8) % n).reshape(n, 8)
    col.sort(axis=1)
    # make columns distinct within a row
    col = (np.arange(8)[None, :] * 511 + np.arange(n)[:, None]) % n
    col.sort(axis=1)
    g = G(rowptr, col.reshape(-1).astype(np.int32), np.ones(n * 8, np.float32))
    ones = torch.ones(n, 64, device=DEV)
    for keep_prob in (0.3, 0.6):
        g.set_edge_mask(2, None, keep_prob, 99)
        y = g.spmm(ones)[:, 0].cpu().numpy() * keep_prob          # = kept entries per row
        assert np.allclose(y, np.round(y), atol=1e-4)
        rate = y.sum() / (n * 8)
        assert abs(rate - keep_prob) < 0.01
        g.set_edge_mask(2, None, keep_prob, 100)
        y2 = g.spmm(ones)[:, 0].cpu().numpy() * keep_prob
        assert (y != y2).mean() > 0.3                               # a different seed is a different mask


def test_dropout_stand_in_rows_do_not_leak(G, oracle):
    """A dropped entry keeps its slot with value 0 and gathers a stand-in row; that row must be one the wave's kept
    entries fetch anyway, never an unrelated row (0 * Inf would be NaN).  Row 0 of the table is referenced by nobody
    and holds Inf; entry 0 of most rows is dropped."""
    rng = np.random.default_rng(5)
    n = 600
    deg = rng.integers(1, 90, n)
    deg[10], deg[11] = 700, 2000
    rowptr, col, val = random_csr(rng, n, n - 1, deg)
    col = (col + 1).astype(np.int32)                                 # nobody reads source row 0
    X = rng.normal(size=(n, 64)).astype(np.float32)
    X[0] = np.inf
    keep = rng.random(len(col)) < 0.5
    keep[rowptr[:-1]] = False                                        # first entry of every row dropped
    g = G(rowptr, col, val, n_cols=n)
    g.set_edge_mask(1, t(keep.astype(np.uint8)), 0.5, 0)
    got = g.spmm(t(X)).cpu().numpy()
    assert np.isfinite(got).all()
    ref = oracle.spmm(rowptr, col, np.where(keep, val / np.float32(0.5), np.float32(0)).astype(np.float32), np.where(np.isfinite(X), X, np.float32(0)))
    assert rel_err(got, ref) <= 1e-5


# ---------------------------------------------------------------------------------------------- scoring kernels
def test_score_bce_vs_oracle(oracle):
    from spex_amd import ops
    rng = np.random.default_rng(3)
    U, I, B = 300, 700, 1000
    users, items = rng.normal(size=(U, 64)).astype(np.float32) * 0.3, rng.normal(size=(I, 64)).astype(np.float32) * 0.3
    u, i = rng.integers(0, U, B), rng.integers(0, I, B)
    u[:50] = 7                                                       # duplicates accumulate
    y = (rng.random(B) < 0.2).astype(np.float32)
    gamma_o, loss_o, gu_o, gi_o = oracle.score_bce(users, items, u, i, y, want_grad=True)
    gu, gi = torch.zeros(U, 64, device=DEV), torch.zeros(I, 64, device=DEV)
    gamma, loss_sum = ops.score_bce(t(users), t(items), t(u), t(i), t(y), gu, gi, 1.0 / B)
    assert rel_err(gamma.cpu().numpy(), gamma_o) <= 1e-6
    assert abs(loss_sum.item() / B - float(loss_o)) <= 1e-6
    assert rel_err(gu.cpu().numpy(), gu_o) <= 1e-5 and rel_err(gi.cpu().numpy(), gi_o) <= 1e-5
    # scores only, indices handed over on the host as int64 (main_rec.py:33-34 / batch_test.py:33)
    gamma2, none = ops.score_bce(t(users), t(items), torch.from_numpy(u), torch.from_numpy(i))
    assert none is None and torch.equal(gamma2, gamma)
    # out-of-range index: skipped, flagged as NaN, nothing else disturbed
    bad = u.copy(); bad[3] = U + 5
    gamma3, _ = ops.score_bce(t(users), t(items), t(bad), t(i))
    g3 = gamma3.cpu().numpy()
    assert np.isnan(g3[3]) and np.array_equal(np.delete(g3, 3), np.delete(gamma.cpu().numpy(), 3))


@pytest.mark.parametrize("grouped", [False, True])
def test_bpr_kernels_vs_closed_form(oracle, grouped):
    from spex_amd import ops
    rng = np.random.default_rng(5)
    U, I, T = 200, 400, 3000
    Ut, It = rng.normal(size=(U, 64)).astype(np.float32) * 0.2, rng.normal(size=(I, 64)).astype(np.float32) * 0.2
    u, p, n = rng.integers(0, U, T), rng.integers(0, I, T), rng.integers(0, I, T)
    loss_o, Un, In = oracle.bpr_sgd(Ut, It, Ut, It, u, p, n, lr=0.05, reg=1e-3)
    Uw, Iw = t(Ut.copy()), t(It.copy())
    loss = ops.bpr_sgd_step(t(Ut), t(It), Uw, Iw, t(u), t(p), t(n), lr=0.05, reg=1e-3, grouped=grouped)
    assert abs(loss.item() / T - loss_o) <= 1e-6
    assert rel_err(Uw.cpu().numpy(), Un) <= 1e-5 and rel_err(Iw.cpu().numpy(), In) <= 1e-5
    # gradient form: finite-difference free check through the closed form (lr=-1 on a zero table == gradient)
    _, dU, dI = oracle.bpr_sgd(Ut, It, np.zeros_like(Ut), np.zeros_like(It), u, p, n, lr=-1.0, reg=0.0)
    gu, gi = torch.zeros(U, 64, device=DEV), torch.zeros(I, 64, device=DEV)
    ops.bpr_loss_grad(t(Ut), t(It), t(u), t(p), t(n), gu, gi, 1.0 / T, grouped=grouped)
    assert rel_err(gu.cpu().numpy(), dU) <= 1e-5 and rel_err(gi.cpu().numpy(), dI) <= 1e-5


@pytest.mark.parametrize("grouped", [False, True])
def test_bpr_kernel_run_accumulation_large_batch(oracle, grouped):
    """Large batches give a wave several consecutive triples; runs of equal user / positive item are accumulated in
    registers and flushed once.  Sampler order (5 negatives per pair, pairs sorted by user), a shuffled copy, and a few
    out-of-range triples (skipped): all must match the closed form."""
    from spex_amd import ops
    rng = np.random.default_rng(17)
    U, I, P = 300, 500, 60000
    pu = np.sort(rng.integers(0, U, P))
    pi = rng.integers(0, I, P)
    u, p = np.repeat(pu, 5), np.repeat(pi, 5)
    n = rng.integers(0, I, 5 * P)
    Ut, It = rng.normal(size=(U, 64)).astype(np.float32) * 0.2, rng.normal(size=(I, 64)).astype(np.float32) * 0.2
    for order in (np.arange(5 * P), rng.permutation(5 * P)):
        uu, pp, nn = u[order], p[order], n[order]
        loss_o, Un, In = oracle.bpr_sgd(Ut, It, Ut, It, uu, pp, nn, lr=0.5, reg=1e-3)
        Uw, Iw = t(Ut.copy()), t(It.copy())
        loss = ops.bpr_sgd_step(t(Ut), t(It), Uw, Iw, t(uu), t(pp), t(nn), lr=0.5, reg=1e-3, grouped=grouped)
        assert abs(loss.item() / len(uu) - loss_o) <= 2e-6
        assert rel_err(Uw.cpu().numpy(), Un) <= 1e-5 and rel_err(Iw.cpu().numpy(), In) <= 1e-5
        _, dU, dI = oracle.bpr_sgd(Ut, It, np.zeros_like(Ut), np.zeros_like(It), uu, pp, nn, lr=-1.0, reg=0.0)
        gu, gi = torch.zeros(U, 64, device=DEV), torch.zeros(I, 64, device=DEV)
        ops.bpr_loss_grad(t(Ut), t(It), t(uu), t(pp), t(nn), gu, gi, 1.0 / len(uu), grouped=grouped)
        assert rel_err(gu.cpu().numpy(), dU) <= 1e-5 and rel_err(gi.cpu().numpy(), dI) <= 1e-5
    # out-of-range indices are skipped, the rest of the wave's run is unaffected
    bad = u.copy()
    bad[[7, 8, 100000, len(bad) - 1]] = U + 5
    keep = bad < U
    _, Un, In = oracle.bpr_sgd(Ut, It, Ut, It, u[keep], p[keep], n[keep], lr=0.5 * keep.sum() / len(u), reg=0.0)
    Uw, Iw = t(Ut.copy()), t(It.copy())
    ops.bpr_sgd_step(t(Ut), t(It), Uw, Iw, t(bad), t(p), t(n), lr=0.5, reg=0.0, grouped=grouped)
    assert rel_err(Uw.cpu().numpy(), Un) <= 1e-5 and rel_err(Iw.cpu().numpy(), In) <= 1e-5


def test_grouped_bpr_hot_rows_many_buckets_and_dispatch(oracle):
    """The LDS-bucketed BPR form at its edges: (a) every triple on the same user and a handful of items — one bucket cut
    into many slices, the atomic flush path; (b) tables of 25 000 rows (391 buckets, most of them with a single
    workgroup: the plain-store flush) with Zipf-distributed items; (c) the automatic choice: atomic form below
    GROUPED_BPR_MIN_TRIPLES, grouped form from there up, and the in-place (hogwild) call never grouped."""
    from spex_amd import ops
    rng = np.random.default_rng(23)
    # (a)
    U, I, T = 70, 130, 50000
    Ut, It = rng.normal(size=(U, 64)).astype(np.float32) * 0.2, rng.normal(size=(I, 64)).astype(np.float32) * 0.2
    u, p, n = np.full(T, 5), rng.integers(0, 3, T), rng.integers(100, 104, T)
    loss_o, Un, In = oracle.bpr_sgd(Ut, It, Ut, It, u, p, n, lr=0.3, reg=0.0)
    Uw, Iw = t(Ut.copy()), t(It.copy())
    loss = ops.bpr_sgd_step(t(Ut), t(It), Uw, Iw, t(u), t(p), t(n), lr=0.3, reg=0.0, grouped=True)
    assert abs(loss.item() / T - loss_o) <= 2e-6
    assert rel_err(Uw.cpu().numpy(), Un) <= 2e-5 and rel_err(Iw.cpu().numpy(), In) <= 2e-5
    # (b)
    U, I, T = 5000, 20000, 300000
    Ut, It = rng.normal(size=(U, 64)).astype(np.float32) * 0.1, rng.normal(size=(I, 64)).astype(np.float32) * 0.1
    u = rng.integers(0, U, T)
    p = np.minimum((rng.zipf(1.3, T) - 1), I - 1)
    n = rng.integers(0, I, T)
    loss_o, Un, In = oracle.bpr_sgd(Ut, It, Ut, It, u, p, n, lr=0.5, reg=1e-3)
    Uw, Iw = t(Ut.copy()), t(It.copy())
    loss = ops.bpr_sgd_step(t(Ut), t(It), Uw, Iw, t(u), t(p), t(n), lr=0.5, reg=1e-3)          # auto: grouped
    assert abs(loss.item() / T - loss_o) <= 2e-6
    assert rel_err(Uw.cpu().numpy(), Un) <= 1e-5 and rel_err(Iw.cpu().numpy(), In) <= 1e-5
    untouched = np.setdiff1d(np.arange(I), np.union1d(p, n))
    assert np.array_equal(Iw.cpu().numpy()[untouched], It[untouched])                         # rows nobody names are not written
    # (c)
    assert ops._lib.load().spex_bpr_grouped_workspace_bytes(1000, 8192 * 64, 1) == 0           # too many rows: not offered
    tab = t(Ut.copy())
    with pytest.raises(ValueError):
        ops.bpr_sgd_step(tab, t(It), tab, t(It.copy()), t(u), t(p), t(n), lr=0.1, grouped=True)  # in place: atomic form only


def test_adam_kernel_vs_oracle(oracle):
    from spex_amd import ops
    rng = np.random.default_rng(6)
    n = 64 * 1001 + 3
    p, g, m, v = (rng.normal(size=n).astype(np.float32) for _ in range(4))
    v = np.abs(v)
    pp, mm, vv = t(p), t(m), t(v)
    for step in (1, 2, 17):
        oracle.adam_step(p, g, m, v, step, lr=1e-3)
        ops.adam_step(pp, t(g), mm, vv, step, lr=1e-3)
        assert rel_err(pp.cpu().numpy(), p) <= 1e-6 and rel_err(mm.cpu().numpy(), m) <= 1e-6
        assert rel_err(vv.cpu().numpy(), v) <= 1e-6
    # the same pass can clear one more buffer of the same length (the next step's gradient accumulation table)
    z = torch.ones(n, device=DEV)
    oracle.adam_step(p, g, m, v, 18, lr=1e-3)
    ops.adam_step(pp, t(g), mm, vv, 18, lr=1e-3, zero=z)
    assert rel_err(pp.cpu().numpy(), p) <= 1e-6 and torch.count_nonzero(z).item() == 0


# ---------------------------------------------------------------------------------------------- NGCF + gate
@pytest.mark.parametrize("ds", ["tiny", "epinion2"])
def test_g7_ngcf_forward_vs_reference(G, golden, epinion2, ds):
    from spex_amd import ops
    from spex_amd.graph import ngcf_norm_adj
    from spex_amd.datasets import epinion2_tables
    g = golden(f"ngcf_{ds}")
    if ds == "tiny":
        csr, uw, iw = (g["rowptr"], g["col"], g["val"]), g["user_w"], g["item_w"]
    else:
        tr = epinion2["train"]
        csr = ngcf_norm_adj(tr[:, 0], tr[:, 1], int(g["n_users"]), int(g["n_items"]))
        uw, iw = epinion2_tables(int(g["n_users"]) + 1, int(g["n_items"]))
    ego = t(np.concatenate([uw[:-1], iw]))
    side = G(*csr).spmm(ego)
    out = ops.ngcf_layer(ego, side, t(g["W_gc"]), t(g["b_gc"]), t(g["W_bi"]), t(g["b_bi"])).cpu().numpy()
    want = g["all_emb"] if ds == "tiny" else g["all_emb_rows"]
    got = out if ds == "tiny" else out[g["sample_rows"]]
    assert rel_err(got, want) <= 1e-5
    assert np.allclose(out.astype(np.float64).sum(0), g["all_emb_colsum"], rtol=1e-4, atol=1e-5)
    # loss on the concatenated tables (ld = 128): NGCF main_rec.py:89-100
    n_u = int(g["n_users"])
    o = t(out)
    gamma, loss_sum = ops.score_bce(o[:n_u], o[n_u:], t(g["batch_users"]), t(g["batch_items"]), t(g["batch_labels"]))
    assert abs(loss_sum.item() / 256 - float(g["loss"])) <= 2e-6


def test_g8_expert_gate_vs_reference(G, golden):
    from spex_amd import ops
    g = golden("lightgcn_tiny")
    n_u = int(g["n_user"]) + 1
    E0 = t(g["E0"])
    out = G(g["rowptr"], g["col"], g["val"]).propagate(E0, 3)
    mu = ops.expert_gate(E0[:n_u].contiguous(), out[:n_u].contiguous(), t(g["g8_att_exp1"]))
    mi = ops.expert_gate(E0[n_u:].contiguous(), out[n_u:].contiguous(), t(g["g8_att_exp2"]))
    gamma, _ = ops.score_bce(mu, mi, t(g["batch_users"][0]), t(g["batch_items"][0]))
    assert rel_err(gamma.cpu().numpy(), g["g8_gamma"]) <= 1e-5


def test_expert_gate_backward_vs_torch_autograd():
    """The gate's backward kernel against torch autograd of the reference expression (model_expert_s.py:156-161)."""
    from spex_amd import ops
    rng = np.random.default_rng(31)
    for n, d in ((3186, 64), (257, 32), (5, 100)):
        raw, prop = (rng.normal(size=(n, d)).astype(np.float32) for _ in range(2))
        att = (rng.normal(size=(2 * d, 2)) * 0.3).astype(np.float32)
        gm = rng.normal(size=(n, d)).astype(np.float32)
        R, P, A = (torch.tensor(x, device=DEV, requires_grad=True) for x in (raw, prop, att))
        (ops.expert_gate_autograd(R, P, A) * t(gm)).sum().backward()
        R2, P2, A2 = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (raw, prop, att))
        w = torch.softmax(torch.cat([R2, P2], 1) @ A2, 1)
        ((R2 * w[:, :1] + P2 * w[:, 1:]) * torch.from_numpy(gm).double()).sum().backward()
        for got, want in ((R, R2), (P, P2), (A, A2)):
            assert (got.grad.cpu().double() - want.grad).abs().max().item() <= 1e-5 * want.grad.abs().max().item()


def test_expert_gate_row_form_matches_the_dense_gate():
    """spex_expert_gate_rows_f32 / _bwd_f32 (the gate at a batch's rows, slot by slot) against the dense kernels: the
    gated rows are bit-identical; scattering a per-slot gradient into a dense table and running the dense backward gives
    the same d raw / d prop tables and gate-matrix gradients (repeated rows, both experts)."""
    from spex_amd import ops
    rng = np.random.default_rng(77)
    n_u, n_i, d, B = 300, 500, 64, 256
    N = n_u + n_i
    raw, prop = (t(rng.normal(size=(N, d)).astype(np.float32)) for _ in range(2))
    att_u, att_i = (t((rng.normal(size=(2 * d, 2)) * 0.3).astype(np.float32)) for _ in range(2))
    users = torch.from_numpy(rng.integers(0, 40, B)).to(DEV)                 # many repeated users
    items = torch.from_numpy(rng.integers(0, n_i, B)).to(DEV)
    mixed_dense = torch.cat([ops.expert_gate(raw[:n_u].contiguous(), prop[:n_u].contiguous(), att_u),
                             ops.expert_gate(raw[n_u:].contiguous(), prop[n_u:].contiguous(), att_i)])
    rows = torch.cat([users, items + n_u])
    got = ops.expert_gate_rows(raw, prop, att_u, att_i, users, items, n_u)
    assert torch.equal(got, mixed_dense[rows])
    g_slots = t(rng.normal(size=(2 * B, d)).astype(np.float32))
    g_prop, g_raw = torch.zeros_like(raw), torch.zeros_like(raw)
    ga_u, ga_i = torch.zeros_like(att_u), torch.zeros_like(att_i)
    d_prop_c = ops.expert_gate_rows_bwd(raw, prop, att_u, att_i, users, items, n_u, g_slots, g_prop, g_raw, ga_u, ga_i)
    # reference: the same through autograd of the dense gate with the slot gradients scattered (summed) into a dense table
    R, P, AU, AI = (x.clone().requires_grad_(True) for x in (raw, prop, att_u, att_i))
    mixed = torch.cat([ops.expert_gate_autograd(R[:n_u], P[:n_u], AU), ops.expert_gate_autograd(R[n_u:], P[n_u:], AI)])
    (mixed[rows] * g_slots).sum().backward()
    for got_t, want in ((g_raw, R.grad), (g_prop, P.grad), (ga_u, AU.grad), (ga_i, AI.grad)):
        assert (got_t - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    dense_from_slots = torch.zeros_like(raw).index_add_(0, rows, d_prop_c)
    assert (dense_from_slots - P.grad).abs().max().item() <= 2e-5 * P.grad.abs().max().item()


# ---------------------------------------------------------------------------------------------- BASELINE configs 3-5 shapes
@pytest.mark.parametrize("name,n_users,n_items,n_edges", [("weibo-like", 6812, 20000, 400000),
                                                          ("twitter-like", 8930, 20000, 400000)])
def test_synthetic_weibo_twitter_graphs_vs_oracle(G, oracle, name, n_users, n_items, n_edges):
    """Weibo / Twitter raw data are not in the reference (README.md:81): heavy-tailed synthetic graphs with their
    published user counts (Trust_SPEX/code/main_trust.py:41-44).  LightGCN propagation (config 3/5 shape) and the NGCF
    layer (config 4 shape) against the oracle; hubs here exceed 1024 entries, so the global-scratch path runs too."""
    from spex_amd import ops
    from spex_amd.datasets import synthetic_interactions, xavier_uniform_np
    from spex_amd.graph import lightgcn_norm_adj, ngcf_norm_adj
    u, i = synthetic_interactions(n_users, n_items, n_edges, seed=7)
    u, i = u.numpy(), i.numpy()
    csr = lightgcn_norm_adj(u, i, n_users, n_items)
    deg = np.diff(csr[0])
    assert deg.max() > 1024 and (deg == 0).sum() > 0          # hubs and empty rows are both present
    rng = np.random.default_rng(1)
    E0 = xavier_uniform_np(len(deg), 64, rng)
    g = G(*csr)
    out = g.propagate(t(E0), 3).cpu().numpy()
    ref = oracle.propagate_mean(*csr, E0, 3, n_threads=8)
    assert rel_err(out, ref) <= 1e-5
    single_wave = deg <= 64
    y = g.spmm(t(E0)).cpu().numpy()
    assert np.array_equal(y[single_wave], oracle.spmm(*csr, E0)[single_wave])
    # NGCF layer on D^-1 (A + I)
    ncsr = ngcf_norm_adj(u, i, n_users, n_items)
    W1, W2 = (rng.normal(size=(64, 64)).astype(np.float32) * 0.1 for _ in range(2))
    b1, b2 = (rng.normal(size=64).astype(np.float32) * 0.1 for _ in range(2))
    uw, iw = xavier_uniform_np(n_users + 1, 64, rng), xavier_uniform_np(n_items, 64, rng)
    ego = t(np.concatenate([uw[:-1], iw]))
    got = ops.ngcf_layer(ego, G(*ncsr).spmm(ego), t(W1), t(b1), t(W2), t(b2)).cpu().numpy()
    want = oracle.ngcf_forward(*ncsr, uw, iw, W1, b1, W2, b2)
    assert rel_err(got, want) <= 1e-5


def test_timer_hook_brackets_the_main_kernel(G, golden):
    g = golden("lightgcn_tiny")
    gr = G(g["rowptr"], g["col"], g["val"])
    X = t(g["E0"])
    gr.attach_timer(8, every=2)
    for _ in range(9):
        gr.spmm(X)
    ms = gr.read_timer()
    assert len(ms) == 5 and (ms > 0).all() and (ms < 5.0).all()       # launches 0,2,4,6,8
    assert len(gr.read_timer()) == 0                                   # reset by the read
    gr.detach_timer()
    gr.spmm(X)


def test_gpu_negative_sampler(epinion2):
    """spex_sample_negatives: never a training pair, reproducible per seed, uniform over each user's non-interacted
    items (chi-square on a small catalogue), wired into LightTrainData."""
    from spex_amd import ops
    rng = np.random.default_rng(0)
    U, I, num_ng = 64, 40, 5
    dense = rng.random((U, I)) < 0.4
    dense[:, 0] = True
    dense[7, :] = True
    dense[7, 33] = False                                   # a user with a single admissible item
    rowptr = np.r_[0, np.cumsum(dense.sum(1))].astype(np.int32)
    items = np.concatenate([np.nonzero(r)[0] for r in dense]).astype(np.int32)
    reps = 4000
    pos_user = np.repeat(np.arange(U), reps).astype(np.int64)
    neg = ops.sample_negatives(t(rowptr), t(items), t(pos_user), num_ng, I, seed=123).cpu().numpy().reshape(-1, num_ng)
    assert not dense[pos_user[:, None].repeat(num_ng, 1), neg].any()
    assert (neg[pos_user == 7] == 33).all()
    again = ops.sample_negatives(t(rowptr), t(items), t(pos_user), num_ng, I, seed=123).cpu().numpy().reshape(-1, num_ng)
    other = ops.sample_negatives(t(rowptr), t(items), t(pos_user), num_ng, I, seed=124).cpu().numpy().reshape(-1, num_ng)
    assert np.array_equal(neg, again) and (neg != other).mean() > 0.5
    for u in (0, 1, 30):                                   # chi-square against the uniform law over admissible items
        adm = np.nonzero(~dense[u])[0]
        cnt = np.bincount(neg[pos_user == u].reshape(-1), minlength=I)[adm]
        exp = cnt.sum() / len(adm)
        chi2 = ((cnt - exp) ** 2 / exp).sum()
        assert chi2 < 3.0 * len(adm), (u, chi2)            # dof = len(adm)-1; 3x is far out in the tail
    # through the dataset class on Epinion2
    import scipy.sparse as sp
    import utility1.dataloader as dl
    tr = epinion2["train"]
    mat = sp.coo_matrix((np.ones(len(tr), np.float32), (tr[:, 0], tr[:, 1])), shape=(3186, 12407)).tocsr()
    td = dl.LightTrainData(tr.tolist(), 12407, mat)
    np.random.seed(2020)
    td.ng_sample_device()
    assert len(td) == 6 * len(tr) and td.labels_fill_np.sum() == len(tr)
    ng = np.asarray(td.features_ng)
    assert np.asarray(mat[ng[:, 0], ng[:, 1]]).sum() == 0            # no negative is a training pair
    assert np.array_equal(ng[:, 0], np.repeat(tr[:, 0], 5))
    first = td.items_fill.copy()
    np.random.seed(2020)
    td.ng_sample_device()
    assert np.array_equal(first, td.items_fill)                       # np.random.seed still pins the run


def test_spmm_task_builder_fuzz(G, oracle):
    """Random CSR shapes around every boundary of the task table (chunk = 16, task = 64, in-workgroup rows <= 1024, hub
    segments of 128): each row either bit-identical to the oracle (<= 64 entries) or within re-association error; all
    three epilogue forms; source tables on both sides of the 16 MiB packing switch."""
    rng = np.random.default_rng(42)
    special = [0, 1, 15, 16, 17, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 1500, 2049]
    for trial in range(12):
        n_rows = int(rng.integers(1, 400))
        # trials 2-4: a source table above 16 MiB switches graph creation to the HBM-resident layout (adjacent-row
        # tasks, no per-entry row ids, round-robin XCD placement)
        n_cols = 70000 if trial in (2, 3, 4) else int(rng.integers(2100, 4000))
        deg = rng.integers(0, 70, n_rows)
        k = min(n_rows, int(rng.integers(0, 12)))
        deg[rng.choice(n_rows, k, replace=False)] = rng.choice(special, k)
        if trial == 0:
            deg[:] = 0                                         # nothing stored at all
        if trial == 1:
            deg[:] = 64                                        # every task exactly full
        rowptr, col, val = random_csr(rng, n_rows, n_cols, deg)
        X = rng.normal(size=(n_cols, 64)).astype(np.float32)
        add, acc = (rng.normal(size=(n_rows, 64)).astype(np.float32) for _ in range(2))
        g = G(rowptr, col, val, n_cols=n_cols)
        ref = oracle.spmm(rowptr, col, val, X)
        exact = np.diff(rowptr) <= 64
        tol = lambda a, b: rel_err(a, b) <= 1e-5 if len(a) else True   # rows of up to 2049 random-sign terms
        y = g.spmm(t(X)).cpu().numpy()
        assert np.array_equal(y[exact], ref[exact]) and tol(y[~exact], ref[~exact]), trial
        Y, A = torch.empty(n_rows, 64, device=DEV), t(acc)
        g.spmm(t(X), Y=Y, acc_in=A, acc_out=A, acc_div=4.0)
        assert np.array_equal(Y.cpu().numpy(), y), trial
        want = (acc + ref) / np.float32(4.0)
        a = A.cpu().numpy()
        assert np.array_equal(a[exact], want[exact]) and tol(a[~exact], want[~exact]), trial
        Y2 = g.spmm(t(X), add_in=t(add), add_div=3.0).cpu().numpy()
        want2 = ref + add / np.float32(3.0)
        assert np.array_equal(Y2[exact], want2[exact]) and tol(Y2[~exact], want2[~exact]), trial


# ---------------------------------------------------------------------------------------------- learned edge values
@pytest.mark.parametrize("d", [64, 128, 20, 3])
def test_sddmm_and_set_values_vs_oracle(G, oracle, d):
    """values refreshed on the device, SpMM with them on A and (through the edge ids) on A^T, and the
    SDDMM that is the SpMM's gradient w.r.t. the values."""
    from spex_amd.graph import csr_transpose
    rng = np.random.default_rng(100 + d)
    n_rows, n_cols = 900, 2500
    deg = rng.integers(0, 70, n_rows)
    deg[5], deg[6], deg[40], deg[41], deg[899] = 0, 650, 130, 1500, 1025     # incl. hub rows (> 1024: global-scratch path)
    rowptr, col, val = random_csr(rng, n_rows, n_cols, deg)
    nnz = len(col)
    g = G(rowptr, col, val, n_cols=n_cols)
    t_rowptr, t_col, t_val, perm = csr_transpose(rowptr, col, val, n_cols)
    gt = G(t_rowptr, t_col, t_val, n_cols=n_rows, edge_id=perm)
    A = rng.normal(size=(n_rows, d)).astype(np.float32)
    B = rng.normal(size=(n_cols, d)).astype(np.float32)
    ref = oracle.sddmm(rowptr, col, A, B, np.float64)
    got = g.sddmm(t(A), t(B)).cpu().numpy()
    scale = np.sqrt(d) * 4
    assert np.abs(got - ref).max() <= 2e-6 * scale
    # the transposed handle writes the same array (edge-id order), from the swapped operands
    got_t = gt.sddmm(t(B), t(A)).cpu().numpy()
    assert np.abs(got_t - ref).max() <= 2e-6 * scale
    # new values: the SpMM on both handles now uses them
    new = rng.normal(size=nnz).astype(np.float32)
    g.set_values(t(new))
    gt.set_values(t(new))
    Y = g.spmm(t(B)).cpu().numpy()
    assert rel_err(Y, oracle.spmm(rowptr, col, new, B)) <= 3e-6          # (1 500-term rows: re-associated partial sums)
    Yt = gt.spmm(t(A)).cpu().numpy()
    assert rel_err(Yt, oracle.spmm(t_rowptr, t_col, new[perm], A)) <= 3e-6
    with pytest.raises(ValueError):
        g.set_values(t(new[:-1]))


def test_edge_softmax_forward_backward_vs_oracle(G, oracle):
    from spex_amd.graph import csr_transpose
    rng = np.random.default_rng(21)
    n_rows, n_cols = 1200, 2000
    deg = rng.integers(0, 50, n_rows)
    deg[0], deg[1], deg[2], deg[3] = 0, 1, 16, 17
    deg[9], deg[10], deg[700], deg[1199] = 777, 1500, 1024, 1025       # in-tile rows up to 1024 entries, hub rows beyond
    rowptr, col, val = random_csr(rng, n_rows, n_cols, deg)
    g = G(rowptr, col, val, n_cols=n_cols)
    v = (rng.normal(size=len(col)) * 3).astype(np.float32)
    y = g.edge_softmax(t(v))
    ref = oracle.edge_softmax(rowptr, v, np.float64)
    assert np.abs(y.cpu().numpy() - ref).max() <= 1e-6
    sums = np.add.reduceat(np.r_[y.cpu().numpy(), 0.0], np.minimum(rowptr[:-1], len(col)))[deg > 0]
    assert np.allclose(sums, 1.0, atol=1e-5)
    gy = rng.normal(size=len(col)).astype(np.float32)
    gx = g.edge_softmax_bwd(y, t(gy)).cpu().numpy()
    assert np.abs(gx - oracle.edge_softmax_bwd(rowptr, ref, gy, np.float64)).max() <= 2e-6
    # softmax over the rows of A^T through the transposed handle: per-edge arrays stay in A's entry order
    t_rowptr, t_col, t_val, perm = csr_transpose(rowptr, col, val, n_cols)
    gt = G(t_rowptr, t_col, t_val, n_cols=n_rows, edge_id=perm)
    yt = gt.edge_softmax(t(v)).cpu().numpy()
    ref_t = np.empty(len(col))
    ref_t[perm] = oracle.edge_softmax(t_rowptr, v[perm], np.float64)
    assert np.abs(yt - ref_t).max() <= 1e-6
    # in place
    buf = t(v)
    g.edge_softmax(buf, out=buf)
    assert np.array_equal(buf.cpu().numpy(), y.cpu().numpy())


def test_learned_spmm_autograd_matches_torch_sparse(G):
    """The two autograd Functions end to end against torch's own sparse autograd on the same device."""
    from spex_amd import ops
    from spex_amd.graph import csr_transpose
    rng = np.random.default_rng(8)
    n_rows, n_cols, d = 500, 400, 64
    rowptr, col, val = random_csr(rng, n_rows, n_cols, rng.integers(1, 40, n_rows))
    g = G(rowptr, col, val, n_cols=n_cols)
    t_rowptr, t_col, t_val, perm = csr_transpose(rowptr, col, val, n_cols)
    gt = G(t_rowptr, t_col, t_val, n_cols=n_rows, edge_id=perm)
    p = torch.tensor(rng.normal(size=len(col)).astype(np.float32), device=DEV, requires_grad=True)
    X = torch.tensor(rng.normal(size=(n_cols, d)).astype(np.float32), device=DEV, requires_grad=True)
    W = torch.tensor(rng.normal(size=(n_rows, d)).astype(np.float32), device=DEV)
    loss = (ops.spmm_learned(ops.edge_softmax(torch.exp(torch.sigmoid(p)), g), X, g, gt) * W).sum()
    loss.backward()
    rows = np.repeat(np.arange(n_rows), np.diff(rowptr))
    idx = torch.from_numpy(np.stack([rows, col.astype(np.int64)])).to(DEV)
    p2 = p.detach().double().requires_grad_()
    X2 = X.detach().double().requires_grad_()
    S = torch.sparse.softmax(torch.sparse_coo_tensor(idx, torch.exp(torch.sigmoid(p2)), (n_rows, n_cols)), dim=1)
    loss2 = (torch.sparse.mm(S, X2) * W.double()).sum()
    loss2.backward()
    assert abs(loss.item() - loss2.item()) <= 1e-5 * max(1.0, abs(loss2.item()))
    assert (p.grad.double() - p2.grad).abs().max().item() <= 1e-5 * p2.grad.abs().max().item()
    assert (X.grad.double() - X2.grad).abs().max().item() <= 1e-5 * X2.grad.abs().max().item()


@pytest.mark.parametrize("d", [128, 256])
@pytest.mark.parametrize("n_cols", [900, 70000])           # <= 16 MiB table: bin-packed tasks with row ids; above: adjacent rows
def test_wide_embeddings_take_the_tuned_kernel(G, oracle, d, n_cols):
    """d = 128 / 256: the chunk kernel with 2 / 4 columns per lane — every row form (empty, single-wave, combined in the
    workgroup, hub), the three epilogues and the propagation wrapper."""
    rng = np.random.default_rng(d + n_cols)
    n_rows = 400
    deg = rng.integers(0, 64, n_rows)
    deg[[0, 1, 2, 3, 4, 5, 6, 7]] = [0, 1, 64, 65, 700, 1024, 1025, 1500][: 8]
    deg = np.minimum(deg, n_cols)
    rowptr, col, val = random_csr(rng, n_rows, n_cols, deg)
    g = G(rowptr, col, val, n_cols=n_cols)
    X = rng.normal(size=(n_cols, d)).astype(np.float32)
    add, acc = (rng.normal(size=(n_rows, d)).astype(np.float32) for _ in range(2))
    y = oracle.spmm(rowptr, col, val, X)
    exact = np.diff(rowptr) <= 64          # longer rows re-associate <= 16 partial sums of random-sign terms: <= 3e-6
    got = g.spmm(t(X)).cpu().numpy()
    assert np.array_equal(got[exact], y[exact]) and rel_err(got, y) <= 3e-6
    Y, A = torch.empty(n_rows, d, device=DEV), t(acc)
    g.spmm(t(X), Y=Y, acc_in=A, acc_out=A, acc_div=4.0)
    assert np.array_equal(Y.cpu().numpy(), got)
    want = (acc + y) / np.float32(4.0)
    assert np.array_equal(A.cpu().numpy()[exact], want[exact]) and rel_err(A.cpu().numpy(), want) <= 3e-6
    Y2 = g.spmm(t(X), add_in=t(add), add_div=3.0).cpu().numpy()
    want2 = y + add / np.float32(3.0)
    assert np.array_equal(Y2[exact], want2[exact]) and rel_err(Y2, want2) <= 3e-6
    # edge dropout on the wide kernel (injected keep mask; the oracle applies the same mask)
    keep = rng.random(len(col)) < 0.7
    g.set_edge_mask(1, t(keep.astype(np.uint8)), 0.7, 0)
    got_m = g.spmm(t(X)).cpu().numpy()
    g.set_edge_mask(0)
    want_m = oracle.spmm_masked(rowptr, col, val, keep, 0.7, X)
    assert np.array_equal(got_m[exact], want_m[exact]) and rel_err(got_m, want_m) <= 3e-6
    if n_cols == 900:
        # square graph: whole propagation (ping-pong workspace, running mean) at this width
        rp, cc, vv = random_csr(rng, 900, 900, np.minimum(rng.integers(0, 90, 900) + (np.arange(900) == 5) * 800, 900))
        vv = (vv * 0.1).astype(np.float32)
        gs = G(rp, cc, vv)
        E0 = rng.normal(size=(900, d)).astype(np.float32)
        out = gs.propagate(t(E0), 3).cpu().numpy()
        assert rel_err(out, oracle.propagate_mean(rp, cc, vv, E0, 3)) <= 1e-6
        back = gs.propagate_bwd(t(E0), 3).cpu().numpy()          # (A need not be symmetric: this is A's own operator)
        cur = E0 / np.float32(4.0)
        acc_ref = cur.copy()
        for _ in range(3):
            acc_ref = cur + oracle.spmm(rp, cc, vv, acc_ref)
        assert rel_err(back, acc_ref) <= 1e-5


# ---------------------------------------------------------------------------------------------- trust-path attention
@pytest.mark.parametrize("H,n_heads", [(64, 3), (64, 1), (32, 2), (100, 1)])
def test_path_attention_kernels_vs_closed_form(H, n_heads):
    """forward and backward of the path attention layer (all heads in one launch) against the CPU
    closed form in fp64 (oracle/trust_oracle.py, itself held against the reference's loops in test_host_logic.py)."""
    from oracle.trust_oracle import path_attention as ref
    from spex_amd import ops
    rng = np.random.default_rng(H + n_heads)
    B, L, U = 37, 6, 90
    seq_l = rng.integers(1, L + 1, B)
    seq_l[:3] = [1, 2, L]
    seq = np.full((B, L), U)
    for p in range(B):
        seq[p, :seq_l[p]] = rng.choice(U, seq_l[p], replace=True)       # users repeat within and across paths
    emb = (rng.normal(size=(U + 1, H)) * 0.5).astype(np.float32)
    a = (rng.normal(size=(n_heads, 2 * H)) * 0.3).astype(np.float32)
    gout = rng.normal(size=(B, L, n_heads * H)).astype(np.float32)
    # gathered, positional (the three in_att heads of model_expert_s.py)
    E = torch.tensor(emb, device=DEV, requires_grad=True)
    A = torch.tensor(a, device=DEV, requires_grad=True)
    out = ops.path_attention(E, t(seq), t(seq_l), A, True)
    (out * t(gout)).sum().backward()
    E64 = torch.tensor(emb, dtype=torch.float64, requires_grad=True)
    A64 = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    want = torch.cat([ref(E64, torch.from_numpy(seq), torch.from_numpy(seq_l), A64[h], True) for h in range(n_heads)], dim=2)
    (want * torch.from_numpy(gout).double()).sum().backward()
    assert (out.detach().cpu().double() - want.detach()).abs().max().item() <= 2e-6 * want.abs().max().item()
    assert (E.grad.cpu().double() - E64.grad).abs().max().item() <= 1e-5 * E64.grad.abs().max().item()
    assert (A.grad.cpu().double() - A64.grad).abs().max().item() <= 1e-5 * A64.grad.abs().max().item() + 1e-6
    assert A.grad[:, :H].abs().max().item() == 0.0                       # a1 cancels in the softmax: exactly no gradient
    # dense, no offsets (out_att)
    Xd = (rng.normal(size=(B, L, H))).astype(np.float32)
    X = torch.tensor(Xd, device=DEV, requires_grad=True)
    A1 = torch.tensor(a[:1], device=DEV, requires_grad=True)
    out2 = ops.path_attention(X, None, t(seq_l), A1, False)
    (out2 * t(gout[..., :H])).sum().backward()
    X64 = torch.tensor(Xd, dtype=torch.float64, requires_grad=True)
    A164 = torch.tensor(a[0], dtype=torch.float64, requires_grad=True)
    want2 = ref(None, X64, torch.from_numpy(seq_l), A164, False)
    (want2 * torch.from_numpy(gout[..., :H]).double()).sum().backward()
    assert (out2.detach().cpu().double() - want2.detach()).abs().max().item() <= 2e-6 * want2.abs().max().item()
    assert (X.grad.cpu().double() - X64.grad).abs().max().item() <= 1e-5 * X64.grad.abs().max().item()
    assert (A1.grad[0].cpu().double() - A164.grad).abs().max().item() <= 1e-5 * A164.grad.abs().max().item() + 1e-6
    # an out-of-range index is never gathered (row of zeros in, nothing written back)
    bad = seq.copy()
    bad[0, 0] = U + 7
    ops.path_attention(t(emb), t(bad), t(seq_l), t(a), True)
    torch.cuda.synchronize()


def test_spmm_properties_on_an_hbm_resident_graph(G, oracle):
    """Epinion2 x 67 (1.04 M nodes, 28 M stored entries: the layout without per-entry row ids, round-robin XCD
    placement): linearity, the symmetric operator's inner-product identity, and sampled rows against the CPU oracle."""
    from spex_amd.datasets import scaled_graph
    rowptr, col, val, _ = scaled_graph(20, device=DEV)
    n = len(rowptr) - 1
    g = G(rowptr, col, val)
    gen = torch.Generator(device=DEV).manual_seed(1)
    x, y = (torch.randn(n, 64, device=DEV, generator=gen) for _ in range(2))
    ax, ay = g.spmm(x), g.spmm(y)
    lhs = g.spmm(2.0 * x + 0.5 * y)
    rhs = 2.0 * ax + 0.5 * ay
    assert (lhs - rhs).abs().max().item() <= 1e-5 * rhs.abs().max().item()
    a = (ax.double() * y.double()).sum().item()
    b = (x.double() * ay.double()).sum().item()
    assert abs(a - b) <= 1e-6 * max(abs(a), abs(b), 1.0)
    rows = np.unique(np.r_[0, n - 1, np.argmax(np.diff(rowptr)), np.random.default_rng(0).integers(0, n, 500)])
    sub_ptr = np.zeros(len(rows) + 1, np.int64)
    sub_ptr[1:] = np.cumsum(rowptr[rows + 1] - rowptr[rows])
    idx = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in rows])
    ref = oracle.spmm(sub_ptr.astype(np.int32), col[idx], val[idx], x.cpu().numpy())
    got = ax[torch.from_numpy(rows).to(DEV)].cpu().numpy()
    short = (rowptr[rows + 1] - rowptr[rows]) <= 64
    assert np.array_equal(got[short], ref[short])                     # single-wave rows: the oracle's fmaf chain, bit for bit
    assert rel_err(got, ref) <= 2e-6


@pytest.mark.parametrize("with_base,d", [(True, 64), (False, 64), (True, 100), (False, 32)])
def test_attn_fuse_kernels_vs_torch_autograd(with_base, d):
    """Diffnet++ node-level attention fusion (Model.py:308-345): forward and every gradient against the tensor-op
    expression in fp64."""
    import torch.nn.functional as F
    from spex_amd import ops
    rng = np.random.default_rng(d + with_base)
    n = 1500
    U = rng.normal(size=(n, d)).astype(np.float32) if with_base else None
    X1, X2, gm = (rng.normal(size=(n, d)).astype(np.float32) for _ in range(3))
    nw = (d if with_base else 0) + d
    p1, p2 = ((rng.normal(size=nw + 3) * 0.3).astype(np.float32) for _ in range(2))
    c1, c2, bc, mc = (0.7, 0.3, 0.5, 0.5) if with_base else (1.0, 1.0, 0.0, 1.0)
    mk = lambda a, dt, dev: None if a is None else torch.tensor(a, dtype=dt, device=dev, requires_grad=True)
    got = [mk(a, torch.float32, DEV) for a in (U, X1, X2, p1, p2)]
    out = ops.attn_fuse(*got, c1, c2, bc, mc)
    (out * t(gm)).sum().backward()
    ref = [mk(a, torch.float64, "cpu") for a in (U, X1, X2, p1, p2)]
    Ur, X1r, X2r, p1r, p2r = ref

    def e(X, p, c):
        inp = X if Ur is None else torch.cat([Ur, X], 1)
        tt = torch.tanh(inp @ p[:nw] + p[nw])
        return torch.exp(F.leaky_relu(p[nw + 1] * tt + p[nw + 2], 0.2)) + c
    e1, e2 = e(X1r, p1r, c1), e(X2r, p2r, c2)
    want = mc * ((e1 / (e1 + e2))[:, None] * X1r + (e2 / (e1 + e2))[:, None] * X2r)
    if Ur is not None:
        want = want + bc * Ur
    (want * torch.from_numpy(gm).double()).sum().backward()
    assert (out.detach().cpu().double() - want.detach()).abs().max().item() <= 2e-6 * want.abs().max().item()
    for gt, rf in zip(got, ref):
        if gt is not None:
            err = (gt.grad.cpu().double() - rf.grad).abs().max().item()
            assert err <= 2e-5 * rf.grad.abs().max().item() + 1e-6, err


def test_spmm_owned_rows_is_the_row_list_product_on_the_owner_and_zero_elsewhere(G, golden, epinion2, oracle):
    """spex_spmm_owned_rows_f32 (the partitioned steps' last forward layer, utility1/model.py:91-97 at the rows of :115-116): on a row
    block of Epinion2's adjacency the owned slots hold exactly what the row-list kernel computes for the local row (same segments,
    same order), with the layer tables added in layer order and the raw row beside it; slots of other ranks' positions, negative
    and too large positions are ZERO; plus hub rows (> 1 024 entries) and an empty row against the oracle."""
    from spex_amd import ops
    g_, csr, E0 = _epinion2(golden, epinion2)
    rowptr, col, val = csr
    n = len(rowptr) - 1
    r0, r1 = 4000, 9000                                                   # a block holding user AND item rows
    lrp = (rowptr[r0:r1 + 1] - rowptr[r0]).astype(np.int64)
    sl = slice(rowptr[r0], rowptr[r1])
    blk = G(lrp, col[sl], val[sl], n_cols=n)
    rng = np.random.default_rng(41)
    X = t(E0)
    acc, a2, a3, raw = (t(rng.normal(size=(r1 - r0, 64)).astype(np.float32)) for _ in range(4))
    pos = torch.from_numpy(np.r_[rng.integers(0, n, 500), [r0, r1 - 1, r0 - 1, r1, -5, n + 3, r0, r0]]).to(DEV)
    own = ((pos >= r0) & (pos < r1)).cpu().numpy()
    assert own.sum() > 100 and (~own).sum() > 100
    loc = (pos - r0).clamp(0, r1 - r0 - 1)
    for tables in ((None, None), (a2, None), (a2, a3)):
        full = torch.empty(r1 - r0, 64, device=DEV)
        blk.spmm_rows(X, torch.arange(r1 - r0, device=DEV), Y=full)       # the row-list kernel at every local row
        run = acc.clone()
        for tb in tables:
            if tb is not None:
                run = run + tb
        want = ((run + full) / 4.0)[loc]
        out_p, out_r = torch.full((len(pos), 64), 7.0, device=DEV), torch.full((len(pos), 64), 7.0, device=DEV)
        ops.spmm_owned_rows(blk, X, pos, r0, acc, 4.0, out_p, raw=raw, out_raw=out_r, acc2=tables[0], acc3=tables[1])
        assert torch.equal(out_p[own], want[own]) and torch.equal(out_r[own], raw[loc][own])
        assert float(out_p[~own].abs().sum()) == 0.0 and float(out_r[~own].abs().sum()) == 0.0
    out_p = torch.full((len(pos), 64), 7.0, device=DEV)
    ops.spmm_owned_rows(blk, X, pos, r0, acc, 1.0, out_p)                  # no raw rows wanted
    assert torch.equal(out_p[own], (acc + full)[loc][own])
    # under edge dropout (a mask on the handle): the masked whole-block product's rows (model.py:46-64)
    for mask in ((2, None, 0.3, (5 << 32) | 7), (1, torch.from_numpy((rng.random(len(col)) < 0.4).astype(np.uint8)).to(DEV), 0.4, 0)):
        blk_e = G(lrp, col[sl], val[sl], n_cols=n, edge_id=np.arange(rowptr[r0], rowptr[r1], dtype=np.int32))   # global edge ids
        blk_e.set_edge_mask(*mask)
        masked = blk_e.spmm(X)
        out_m = torch.empty(len(pos), 64, device=DEV)
        ops.spmm_owned_rows(blk_e, X, pos, r0, acc, 4.0, out_m)
        assert rel_err(out_m[own].cpu().numpy(), ((acc + masked) / 4.0)[loc][own].cpu().numpy()) <= 2e-6
        assert float(out_m[~own].abs().sum()) == 0.0
        assert (masked - full).abs().max().item() > 1e-3                   # (the mask mattered)
        blk_e.set_edge_mask(0)
    # hub rows, an empty row
    deg = rng.integers(0, 50, 300)
    deg[3], deg[4], deg[5] = 0, 1500, 2600
    rp2, c2, v2 = random_csr(rng, 300, 3000, deg)
    g2 = G(rp2, c2, v2, n_cols=3000)
    X2 = rng.normal(size=(3000, 64)).astype(np.float32)
    acc2_ = rng.normal(size=(300, 64)).astype(np.float32)
    want2 = (acc2_ + oracle.spmm(rp2, c2, v2, X2)) / 3.0
    pos2 = torch.tensor([1003, 1004, 1005, 1299, 1300, 999, 1004], device=DEV)
    out2 = torch.empty(7, 64, device=DEV)
    ops.spmm_owned_rows(g2, t(X2), pos2, 1000, t(acc2_), 3.0, out2)
    got2 = out2.cpu().numpy()
    for k, r in enumerate([3, 4, 5, 299, None, None, 4]):
        if r is None:
            assert not got2[k].any()
        else:
            assert rel_err(got2[k], want2[r]) <= 3e-6, (k, r)


def test_spmm_rowlist_is_the_full_product_at_the_listed_rows(G, golden, epinion2):
    """spex_spmm_rowlist_f32: bit-identical to the full launch at the listed rows (every row of Epinion2 has <= 1024
    entries: same segments, same order), other rows untouched; plus a graph with hub rows, empty rows and bad indices."""
    g_, csr, E0 = _epinion2(golden, epinion2)
    g = G(*csr)
    rng = np.random.default_rng(4)
    X, acc = t(E0), t(rng.normal(size=E0.shape).astype(np.float32))
    full = torch.empty_like(X)
    full_acc = torch.empty_like(X)
    g.spmm(X, Y=full, acc_in=acc, acc_out=full_acc, acc_div=4.0)
    users = torch.from_numpy(np.r_[rng.integers(0, 3185, 300), np.argsort(-np.diff(csr[0])[:3186])[:8]]).to(DEV)
    items = torch.from_numpy(rng.integers(0, 12407, 300)).to(DEV)
    Y = torch.full_like(X, 7.0)
    A = acc.clone()
    g.spmm_rows(X, users, items, 0, 3186, Y=Y, acc_in=acc, acc_out=A, acc_div=4.0)     # (users repeat: out of place!)
    rows = torch.cat([users, items + 3186])
    assert torch.equal(Y[rows], full[rows]) and torch.equal(A[rows], full_acc[rows])
    untouched = torch.ones(len(E0), dtype=torch.bool, device=DEV)
    untouched[rows] = False
    assert (Y[untouched] == 7.0).all() and torch.equal(A[untouched], acc[untouched])
    # hub rows (> 1024 entries), an empty row, out-of-range indices
    deg = rng.integers(0, 50, 300)
    deg[3], deg[4], deg[5] = 0, 1500, 2600
    rowptr, col, val = random_csr(rng, 300, 3000, deg)
    g2 = G(rowptr, col, val, n_cols=3000)
    X2 = t(rng.normal(size=(3000, 64)).astype(np.float32))
    want = g2.spmm(X2)
    idx = torch.tensor([3, 4, 5, 7, 299, 300, -1, 4], device=DEV)
    got = torch.zeros(300, 64, device=DEV)
    g2.spmm_rows(X2, idx, Y=got)
    ok = torch.tensor([3, 4, 5, 7, 299], device=DEV)
    assert rel_err(got[ok].cpu().numpy(), want[ok].cpu().numpy()) <= 3e-6
    assert torch.equal(got[torch.tensor([3, 7, 299], device=DEV)], want[torch.tensor([3, 7, 299], device=DEV)])


def test_lightgcn_batch_kernel_equals_the_three_launch_sequence(G, golden, epinion2, oracle):
    """spex_lightgcn_batch_f32 (last layer at the batch's rows + layer mean + scores + BCE + gradient rows + push-form first
    backward product, one launch) against (a) the sequence it replaces — spmm_rows, score_bce with per-sample rows,
    spmm_push_batch — and (b) the oracle's loss / dense gradient on the same inputs.  Epinion2 (hub rows of ~1 000 entries
    among the batch's rows, repeated users and items), B = 256 as in the reference driver; a second batch carries an
    out-of-range index, which is skipped."""
    from spex_amd import ops
    g_, csr, E0 = _epinion2(golden, epinion2)
    g = G(*csr)                                                          # symmetric: A^T == A
    n, n_u, L = len(E0), 3186, 3
    rng = np.random.default_rng(12)
    X, run = t(E0), t((rng.normal(size=E0.shape) * 0.05).astype(np.float32))
    deg = np.diff(csr[0])
    users = rng.integers(0, 3185, 256)
    items = rng.integers(0, 12407, 256)
    users[:4] = np.argsort(-deg[:n_u])[:4]                               # the heaviest user rows
    items[:4] = np.argsort(-deg[n_u:])[:4]                               # the heaviest item rows (~1 000 entries)
    users[10:14] = users[0]                                              # repeats
    items[20:30] = items[1]
    labels = (rng.random(256) < 1 / 6).astype(np.float32)
    u_d, i_d, y_d = t(users), t(items), t(labels)
    # (a) the three-launch sequence
    lo = run.clone()
    g.spmm_rows(X, u_d, i_d, 0, n_u, acc_in=run, acc_out=lo, acc_div=float(L + 1))
    slots = torch.zeros(512, 64, device=DEV)
    loss_a = torch.zeros(1, device=DEV)
    g_out_a = torch.zeros(n, 64, device=DEV)
    ops.score_bce(lo[:n_u], lo[n_u:], u_d, i_d, y_d, loss_sum=loss_a, grad_users=g_out_a[:n_u], grad_items=g_out_a[n_u:],
                  grad_scale=1.0 / 256, grad_slots=slots)
    G_a = torch.zeros(n, 64, device=DEV)
    ops.spmm_push_batch(g, u_d, i_d, n_u, slots, G_a, add=slots, scale=1.0 / (L + 1))
    # the fused launch
    loss_b, g_out_b, G_b = torch.zeros(1, device=DEV), torch.zeros(n, 64, device=DEV), torch.zeros(n, 64, device=DEV)
    ops.lightgcn_batch(g, X, run, float(L + 1), u_d, i_d, y_d, n_u, 1.0 / 256, 1.0 / (L + 1), loss_b, g_out_b, G_b)
    assert abs(loss_a.item() - loss_b.item()) <= 1e-5 * abs(loss_a.item())
    assert rel_err(g_out_b.cpu().numpy(), g_out_a.cpu().numpy()) <= 2e-6       # float atomics: order only
    assert rel_err(G_b.cpu().numpy(), G_a.cpu().numpy()) <= 2e-6
    # (b) the oracle: light rows, loss, dense gradient, then (g + A^T g) / (L + 1) in pull form
    light = (run.cpu().numpy() + oracle.spmm(*csr, E0)) / np.float32(L + 1)
    x = np.einsum("bd,bd->b", light[users].astype(np.float64), light[items + n_u].astype(np.float64))
    want_loss = np.sum(np.maximum(x, 0) - x * labels + np.log1p(np.exp(-np.abs(x))))
    assert abs(loss_b.item() - want_loss) <= 2e-6 * abs(want_loss) + 1e-6
    dg = ((1.0 / (1.0 + np.exp(-x)) - labels) / 256.0)
    want_g = np.zeros((n, 64))
    np.add.at(want_g, users, dg[:, None] * light[items + n_u])
    np.add.at(want_g, items + n_u, dg[:, None] * light[users])
    assert rel_err(g_out_b.cpu().numpy(), want_g) <= 5e-6
    want_G = (want_g + oracle.spmm(*csr, want_g.astype(np.float32)).astype(np.float64)) / (L + 1)
    assert rel_err(G_b.cpu().numpy(), want_G) <= 1e-5
    # an out-of-range index skips its sample, everything else is unchanged
    bad_u = users.copy()
    bad_u[7] = 999999
    ok = np.ones(256, bool); ok[7] = False
    loss_c, g_out_c, G_c = torch.zeros(1, device=DEV), torch.zeros(n, 64, device=DEV), torch.zeros(n, 64, device=DEV)
    ops.lightgcn_batch(g, X, run, float(L + 1), t(bad_u), i_d, y_d, n_u, 1.0 / 256, 1.0 / (L + 1), loss_c, g_out_c, G_c)
    want_gc = np.zeros((n, 64))
    np.add.at(want_gc, users[ok], dg[ok, None] * light[items[ok] + n_u])
    np.add.at(want_gc, items[ok] + n_u, dg[ok, None] * light[users[ok]])
    assert rel_err(g_out_c.cpu().numpy(), want_gc) <= 5e-6
    assert torch.isfinite(G_c).all()
    # per-sample losses instead of the accumulated sum (what the one-call step uses: its Adam pass adds them up in order)
    per = torch.full((256,), 7.0, device=DEV)
    ops.lightgcn_batch(g, X, run, float(L + 1), t(bad_u), i_d, y_d, n_u, 1.0 / 256, 1.0 / (L + 1), None, torch.zeros_like(g_out_c),
                       torch.zeros_like(G_c), loss_per_sample=per)
    want_per = np.maximum(x, 0) - x * labels + np.log1p(np.exp(-np.abs(x)))
    want_per[7] = 0.0
    assert np.abs(per.cpu().numpy() - want_per).max() <= 2e-6


def test_lightgcn_batch_kernel_hub_rows_and_a_non_symmetric_matrix(G, oracle):
    """The same launch on a NON-symmetric square matrix with rows beyond 1 024 entries (16 virtual waves chaining several
    segments each), an empty row and a one-entry row among the batch's rows.  The first backward product is A^T g: in push form
    it walks the rows of A itself ((A^T g)[c] = sum_r A[r, c] g[r]), so the result must equal the oracle's PULL-form product on
    the transposed CSR — what autograd computes for `torch.sparse.mm(A, x)` (round 2 pushed over the rows of the transposed
    handle, i.e. computed A g: right only for a symmetric matrix).  Also against the three-launch sequence."""
    from spex_amd import ops
    rng = np.random.default_rng(5)
    n, n_u, L = 3000, 1000, 2
    deg = rng.integers(0, 50, n)
    deg[[2, 1500, 2999]] = [1500, 2600, 1100]
    deg[7], deg[1200] = 0, 1
    rowptr, col, val = random_csr(rng, n, n, deg)
    val *= 0.05
    t_csr = oracle.csr_transpose(rowptr, col, val, n)
    g = G(rowptr, col, val)
    X = t((rng.normal(size=(n, 64)) * 0.3).astype(np.float32))
    run = t((rng.normal(size=(n, 64)) * 0.3).astype(np.float32))
    users = rng.integers(0, n_u, 64)
    items = rng.integers(0, n - n_u, 64)
    users[:3] = [2, 7, 2]
    items[:3] = [1500 - n_u, 2999 - n_u, 1200 - n_u]
    labels = (rng.random(64) < 0.3).astype(np.float32)
    u_d, i_d, y_d = t(users), t(items), t(labels)
    lo = run.clone()
    g.spmm_rows(X, u_d, i_d, 0, n_u, acc_in=run, acc_out=lo, acc_div=float(L + 1))
    slots = torch.zeros(128, 64, device=DEV)
    loss_a, g_out_a, G_a = torch.zeros(1, device=DEV), torch.zeros(n, 64, device=DEV), torch.zeros(n, 64, device=DEV)
    ops.score_bce(lo[:n_u], lo[n_u:], u_d, i_d, y_d, loss_sum=loss_a, grad_users=g_out_a[:n_u], grad_items=g_out_a[n_u:],
                  grad_scale=1.0 / 64, grad_slots=slots)
    ops.spmm_push_batch(g, u_d, i_d, n_u, slots, G_a, add=slots, scale=1.0 / (L + 1))
    loss_b, g_out_b, G_b = torch.zeros(1, device=DEV), torch.zeros(n, 64, device=DEV), torch.zeros(n, 64, device=DEV)
    ops.lightgcn_batch(g, X, run, float(L + 1), u_d, i_d, y_d, n_u, 1.0 / 64, 1.0 / (L + 1), loss_b, g_out_b, G_b)
    assert abs(loss_a.item() - loss_b.item()) <= 1e-5 * abs(loss_a.item())
    assert rel_err(g_out_b.cpu().numpy(), g_out_a.cpu().numpy()) <= 3e-6
    assert rel_err(G_b.cpu().numpy(), G_a.cpu().numpy()) <= 3e-6
    gd = g_out_b.cpu().numpy()
    # out[c] += val[e] * g[r] over the entries e = (r, c) of A  <=>  out = A^T g: the oracle's pull form on the transposed CSR
    want_G = (gd.astype(np.float64) + oracle.spmm(*t_csr[:3], gd).astype(np.float64)) / (L + 1)
    assert rel_err(G_b.cpu().numpy(), want_G) <= 1e-5
    not_sym = (gd.astype(np.float64) + oracle.spmm(rowptr, col, val, gd).astype(np.float64)) / (L + 1)
    assert rel_err(not_sym, want_G) > 1e-2                                   # the matrix really is not symmetric


def test_gated_batch_forward_equals_the_three_launch_sequence(G, golden, epinion2):
    """spex_gated_batch_fwd_f32 (dual-task rec branch: last layer at the batch's rows + layer mean + expert gate + scores + BCE +
    per-sample gradient rows, one launch) against spmm_rows -> expert_gate_rows -> score_bce(grad_slots) on Epinion2 with hub
    rows and repeats in the batch."""
    from spex_amd import ops
    g_, csr, E0 = _epinion2(golden, epinion2)
    g = G(*csr)
    n, n_u, L, B = len(E0), 3186, 3, 256
    rng = np.random.default_rng(21)
    X, run, raw = t(E0), t((rng.normal(size=E0.shape) * 0.05).astype(np.float32)), t((rng.normal(size=E0.shape) * 0.1).astype(np.float32))
    att_u, att_i = t((rng.normal(size=(128, 2)) * 0.5).astype(np.float32)), t((rng.normal(size=(128, 2)) * 0.5).astype(np.float32))
    deg = np.diff(csr[0])
    users, items = rng.integers(0, 3185, B), rng.integers(0, 12407, B)
    users[:4] = np.argsort(-deg[:n_u])[:4]
    items[:4] = np.argsort(-deg[n_u:])[:4]
    users[10:14] = users[0]
    labels = (rng.random(B) < 1 / 6).astype(np.float32)
    u_d, i_d, y_d = t(users), t(items), t(labels)
    # three launches
    lo_a = torch.zeros(n, 64, device=DEV)
    g.spmm_rows(X, u_d, i_d, 0, n_u, acc_in=run, acc_out=lo_a, acc_div=float(L + 1))
    mixed = ops.expert_gate_rows(raw, lo_a, att_u, att_i, u_d, i_d, n_u)
    slots_a, loss_a = torch.zeros(2 * B, 64, device=DEV), torch.zeros(1, device=DEV)
    ar = torch.arange(B, device=DEV)
    ops.score_bce(mixed[:B], mixed[B:], ar, ar, y_d, None, None, 1.0 / B, loss_sum=loss_a, grad_slots=slots_a, want_gamma=False)
    # one launch
    lo_b, slots_b, loss_b = torch.zeros(n, 64, device=DEV), torch.full((2 * B, 64), 7.0, device=DEV), torch.zeros(1, device=DEV)
    ops.gated_batch_fwd(g, X, run, float(L + 1), raw, att_u, att_i, u_d, i_d, y_d, n_u, 1.0 / B, loss_b, lo_b, slots_b)
    rows = torch.cat([u_d, i_d + n_u])
    assert torch.equal(lo_b[rows], lo_a[rows])                              # the row-list kernel's segments and order
    assert abs(loss_a.item() - loss_b.item()) <= 1e-5 * abs(loss_a.item())
    assert rel_err(slots_b.cpu().numpy(), slots_a.cpu().numpy()) <= 2e-6


def _gated_middle_three_launches(ops, g, X, run, L, raw, att_u, att_i, u_d, i_d, y_d, n_u):
    """gated_batch_fwd -> expert_gate_rows_bwd -> spmm_push_batch: the sequence spex_gated_batch_f32 replaces."""
    n, B = X.shape[0], u_d.numel()
    z = lambda *s: torch.zeros(*s, device=DEV)
    lo, slots, loss = z(n, 64), z(2 * B, 64), z(1)
    ops.gated_batch_fwd(g, X, run, float(L + 1), raw, att_u, att_i, u_d, i_d, y_d, n_u, 1.0 / B, loss, lo, slots)
    g_prop, g_raw, G_, ga_u, ga_i = z(n, 64), z(n, 64), z(n, 64), z(128, 2), z(128, 2)
    d_prop = ops.expert_gate_rows_bwd(raw, lo, att_u, att_i, u_d, i_d, n_u, slots, g_prop, g_raw, ga_u, ga_i)
    ops.spmm_push_batch(g, u_d, i_d, n_u, d_prop, G_, add=d_prop, scale=1.0 / (L + 1))
    return loss, g_prop, G_, g_raw, ga_u, ga_i


def _check_gated_middle(ops, g, X, run, L, raw, att_u, att_i, u_d, i_d, y_d, n_u, tol):
    want = _gated_middle_three_launches(ops, g, X, run, L, raw, att_u, att_i, u_d, i_d, y_d, n_u)
    n, B = X.shape[0], u_d.numel()
    z = lambda *s: torch.zeros(*s, device=DEV)
    for copies in (1, 64):                     # (one copy of the gate gradients: every sample adds to the same words)
        got = (z(1), z(n, 64), z(n, 64), z(n, 64), z(copies, 2, 128, 2))
        ops.gated_batch(g, X, run, float(L + 1), raw, att_u, att_i, u_d, i_d, y_d, n_u, 1.0 / B, 1.0 / (L + 1), *got)
        assert abs(want[0].item() - got[0].item()) <= 1e-5 * abs(want[0].item()) + 1e-7
        g_att = got[4].sum(0)
        for nm, a, b in zip(("g_prop", "G", "g_raw", "g_att_u", "g_att_i"), want[1:], got[1:4] + (g_att[0], g_att[1])):
            assert rel_err(b.cpu().numpy(), a.cpu().numpy()) <= tol, (nm, copies)


def test_gated_batch_one_launch_equals_forward_gate_backward_push(G, golden, epinion2):
    """spex_gated_batch_f32 (the dual-task rec branch's whole batch-sized middle in one launch) against spex_gated_batch_fwd_f32 ->
    spex_expert_gate_rows_bwd_f32 -> spex_spmm_push_batch_f32 on Epinion2 with hub rows (shared by several workgroups) and
    repeated users: dense d light, the pushed table, d raw, both gate gradients, the loss."""
    from spex_amd import ops
    g_, csr, E0 = _epinion2(golden, epinion2)
    g = G(*csr)
    n_u, L, B = 3186, 3, 256
    rng = np.random.default_rng(22)
    X, run, raw = t(E0), t((rng.normal(size=E0.shape) * 0.05).astype(np.float32)), t((rng.normal(size=E0.shape) * 0.1).astype(np.float32))
    att_u, att_i = t((rng.normal(size=(128, 2)) * 0.5).astype(np.float32)), t((rng.normal(size=(128, 2)) * 0.5).astype(np.float32))
    deg = np.diff(csr[0])
    users, items = rng.integers(0, 3185, B), rng.integers(0, 12407, B)
    users[:4] = np.argsort(-deg[:n_u])[:4]
    items[:4] = np.argsort(-deg[n_u:])[:4]
    users[10:14] = users[0]
    labels = (rng.random(B) < 1 / 6).astype(np.float32)
    _check_gated_middle(ops, g, X, run, L, raw, att_u, att_i, t(users), t(items), t(labels), n_u, 3e-6)


@pytest.mark.parametrize("B", [1, 3, 17])
def test_batch_kernels_small_batches_and_degenerate_rows(G, oracle, B):
    """spex_lightgcn_batch_f32 and spex_gated_batch_fwd_f32 on batches of 1 / 3 / 17 samples whose rows include an EMPTY row, a
    one-entry row and a row of exactly 64 and of 65 entries (segment boundary), against the multi-launch sequences."""
    from spex_amd import ops
    rng = np.random.default_rng(100 + B)
    n, n_u, L = 600, 200, 3
    deg = rng.integers(1, 30, n)
    deg[0], deg[1], deg[2], deg[3] = 0, 1, 64, 65
    deg[n_u], deg[n_u + 1], deg[n_u + 2] = 0, 65, 64
    rowptr, col, val = random_csr(rng, n, n, deg)
    val *= 0.1
    g = G(rowptr, col, val)
    X = t((rng.normal(size=(n, 64)) * 0.3).astype(np.float32))
    run = t((rng.normal(size=(n, 64)) * 0.3).astype(np.float32))
    raw = t((rng.normal(size=(n, 64)) * 0.3).astype(np.float32))
    att_u, att_i = t((rng.normal(size=(128, 2)) * 0.5).astype(np.float32)), t((rng.normal(size=(128, 2)) * 0.5).astype(np.float32))
    users = np.array(([0, 1, 2, 3] + list(rng.integers(0, n_u, 32)))[:B])
    items = np.array(([0, 1, 2, 5] + list(rng.integers(0, n - n_u, 32)))[:B])
    labels = (rng.random(B) < 0.4).astype(np.float32)
    u_d, i_d, y_d = t(users), t(items), t(labels)
    lo = run.clone()
    g.spmm_rows(X, u_d, i_d, 0, n_u, acc_in=run, acc_out=lo, acc_div=float(L + 1))
    # LightGCN form
    slots, loss_a = torch.zeros(2 * B, 64, device=DEV), torch.zeros(1, device=DEV)
    g_out_a, G_a = torch.zeros(n, 64, device=DEV), torch.zeros(n, 64, device=DEV)
    ops.score_bce(lo[:n_u], lo[n_u:], u_d, i_d, y_d, loss_sum=loss_a, grad_users=g_out_a[:n_u], grad_items=g_out_a[n_u:],
                  grad_scale=1.0 / B, grad_slots=slots)
    ops.spmm_push_batch(g, u_d, i_d, n_u, slots, G_a, add=slots, scale=1.0 / (L + 1))
    loss_b, g_out_b, G_b = torch.zeros(1, device=DEV), torch.zeros(n, 64, device=DEV), torch.zeros(n, 64, device=DEV)
    ops.lightgcn_batch(g, X, run, float(L + 1), u_d, i_d, y_d, n_u, 1.0 / B, 1.0 / (L + 1), loss_b, g_out_b, G_b)
    assert abs(loss_a.item() - loss_b.item()) <= 1e-5 * abs(loss_a.item()) + 1e-7
    assert rel_err(g_out_b.cpu().numpy(), g_out_a.cpu().numpy()) <= 3e-6
    assert rel_err(G_b.cpu().numpy(), G_a.cpu().numpy()) <= 3e-6
    # gated form
    mixed = ops.expert_gate_rows(raw, lo, att_u, att_i, u_d, i_d, n_u)
    slots_a, loss_c = torch.zeros(2 * B, 64, device=DEV), torch.zeros(1, device=DEV)
    ar = torch.arange(B, device=DEV)
    ops.score_bce(mixed[:B], mixed[B:], ar, ar, y_d, None, None, 1.0 / B, loss_sum=loss_c, grad_slots=slots_a, want_gamma=False)
    lo_b, slots_b, loss_d = torch.zeros(n, 64, device=DEV), torch.full((2 * B, 64), 7.0, device=DEV), torch.zeros(1, device=DEV)
    ops.gated_batch_fwd(g, X, run, float(L + 1), raw, att_u, att_i, u_d, i_d, y_d, n_u, 1.0 / B, loss_d, lo_b, slots_b)
    rows = torch.cat([u_d, i_d + n_u])
    assert torch.equal(lo_b[rows], lo[rows])
    assert abs(loss_c.item() - loss_d.item()) <= 1e-5 * abs(loss_c.item()) + 1e-7
    assert rel_err(slots_b.cpu().numpy(), slots_a.cpu().numpy()) <= 3e-6
    # gated form with the gate's backward and the push in the same launch
    _check_gated_middle(ops, g, X, run, L, raw, att_u, att_i, u_d, i_d, y_d, n_u, 3e-6)


def test_one_handle_driven_from_two_streams(G):
    """A graph with hub rows (> 1024 entries: their segment sums go through the handle's scratch buffer) driven from two
    streams in alternation: the library orders each launch behind the scratch's previous user, so every product equals the
    one computed alone on the default stream (bit for bit: the kernel is deterministic)."""
    rng = np.random.default_rng(77)
    deg = rng.integers(0, 60, 2000)
    deg[[3, 500, 1999]] = [3000, 2500, 1100]
    rowptr, col, val = random_csr(rng, 2000, 4000, deg)
    g = G(rowptr, col, val, n_cols=4000)
    Xs = [t(rng.normal(size=(4000, 64)).astype(np.float32)) for _ in range(6)]
    want = [g.spmm(X).clone() for X in Xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty(2000, 64, device=DEV) for _ in Xs]
    for rep in range(20):
        for k, X in enumerate(Xs):
            with torch.cuda.stream(streams[k & 1]):
                g.spmm(X, Y=outs[k])
    torch.cuda.synchronize()
    for k in range(len(Xs)):
        assert torch.equal(outs[k], want[k]), k


def test_one_handle_driven_from_two_host_threads(G):
    """The same handle from two HOST threads, each on its own stream (ctypes releases the GIL inside the library call): the launch
    that writes the handle's scratch holds the handle's mutex from the ordering decision until its kernels are queued, so the two
    threads' launches interleave in some order but never overlap on the scratch — every product equals the serial one bit for bit
    (hub rows folded in the launch: partial rows + arrival counters are the shared state)."""
    import threading
    rng = np.random.default_rng(78)
    deg = rng.integers(0, 60, 2000)
    deg[[3, 500, 1999]] = [3000, 2500, 1100]
    rowptr, col, val = random_csr(rng, 2000, 4000, deg)
    g = G(rowptr, col, val, n_cols=4000)
    Xs = [t(rng.normal(size=(4000, 64)).astype(np.float32)) for _ in range(2)]
    want = [g.spmm(X).clone() for X in Xs]
    torch.cuda.synchronize()
    outs = [[torch.empty(2000, 64, device=DEV) for _ in range(150)] for _ in Xs]
    errs = []

    def worker(k):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for rep in range(150):
                    g.spmm(Xs[k], Y=outs[k][rep])
            st.synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for k in range(2):
        for rep in range(150):
            assert torch.equal(outs[k][rep], want[k]), (k, rep)


# ---------------------------------------------------------------------------------------------- row-sparse backward pieces
def test_unique_rows_and_push_form_spmm_vs_oracle(G, oracle):
    """spex_unique_rows_i32 + spex_spmm_push_rows_f32: the distinct rows of a batch with repeats (and an out-of-range index),
    then out += scale * (A^T scatter(src) + scatter(add)) over those rows — against the oracle's pull-form product with the
    transposed matrix on the same row-sparse input.  A hub row (1 500 entries) and an empty row are among the listed rows;
    both source conventions (table-indexed / compact) are used."""
    from spex_amd import ops
    rng = np.random.default_rng(31)
    n = 900
    deg = rng.integers(0, 40, n)
    deg[5], deg[17] = 1500, 0
    rowptr, col, val = random_csr(rng, n, n, np.minimum(deg, n))
    g = G(rowptr, col, val)
    ua = rng.integers(0, 400, 256); ub = rng.integers(0, 500, 256)
    ua[:3] = [5, 5, 17]                                   # hub twice, the empty row
    ua[9] = 5000                                          # out of range: ignored
    rows = ops.UniqueRows(n, 512, DEV).update(t(ua), t(ub), 0, 400)
    want_rows = np.unique(np.concatenate([ua[ua < n], ub + 400]))
    cnt = int(rows.count.item())
    got_rows = np.sort(rows.list.cpu().numpy()[:cnt])
    assert np.array_equal(got_rows, want_rows)
    rows.update(t(ua), t(ub), 0, 400)                     # a second epoch on the same stamp table gives the same set
    assert int(rows.count.item()) == cnt
    listed = rows.list.cpu().numpy()[:cnt]
    src = np.zeros((n, 64), np.float32)
    src[want_rows] = rng.normal(size=(len(want_rows), 64)).astype(np.float32)
    t_csr = oracle.csr_transpose(rowptr, col, val, n)
    want = np.float32(0.25) * (oracle.spmm(*t_csr, src) + src)
    out = torch.zeros(n, 64, device=DEV)
    ops.spmm_push_rows(g, rows, t(src), out, True, add=t(src), add_indexed=True, scale=0.25)
    assert rel_err(out.cpu().numpy(), want) <= 2e-6
    compact = np.zeros((512, 64), np.float32)
    compact[:cnt] = src[listed]
    out2 = torch.zeros(n, 64, device=DEV)
    ops.spmm_push_rows(g, rows, t(compact), out2, False)
    assert rel_err(out2.cpu().numpy(), oracle.spmm(*t_csr, src)) <= 2e-6
    # the batch-driven form: no list, every slot pushes its own source row (rows named twice get two contributions)
    ua_ok = np.where(ua < n, ua, 5)
    slot_rows = np.concatenate([ua_ok, ub + 400])
    per_slot = rng.normal(size=(512, 64)).astype(np.float32)
    dense = np.zeros((n, 64), np.float32)
    np.add.at(dense, slot_rows, per_slot)
    out3 = torch.zeros(n, 64, device=DEV)
    ops.spmm_push_batch(g, t(ua_ok), t(ub), 400, t(per_slot), out3, add=t(per_slot), scale=0.25)
    assert rel_err(out3.cpu().numpy(), np.float32(0.25) * (oracle.spmm(*t_csr, dense) + dense)) <= 2e-6


@pytest.mark.parametrize("L", [1, 2, 3])
def test_one_call_bpr_step_equals_propagate_then_bpr(G, golden, epinion2, oracle, L):
    """spex_lightgcn_step_bpr_f32 (layer 1 with the running sum fused, later layers plain, the layer mean formed by the BPR kernel
    at its triples' rows) against spex_propagate_f32 followed by spex_bpr_sgd_step_f32 on Epinion2: same loss sum, same updated
    table (float-atomic updates in both forms: order only), and against the oracle's closed-form BPR-SGD step."""
    from spex_amd import ops
    from spex_amd.trainer import LightGCNStepper
    g_, csr, E0 = _epinion2(golden, epinion2)
    n_u = 3186
    rng = np.random.default_rng(40 + L)
    u, p, n = rng.integers(0, 3185, 2048), rng.integers(0, 12407, 2048), rng.integers(0, 12407, 2048)
    u[:64] = u[0]                                                        # a hot user row
    ud, pd_, nd = t(u), t(p), t(n)
    g = G(*csr)
    st = LightGCNStepper(g, t(E0), n_u, n_layers=L, lr=0.05)
    loss_new = st.step_bpr_sgd(ud, pd_, nd).item()
    lo = g.propagate(t(E0), L)
    U_w = t(E0)
    loss_old = ops.bpr_sgd_step(lo[:n_u], lo[n_u:], U_w[:n_u], U_w[n_u:], ud, pd_, nd, 0.05, 0.0, grouped=False).item()
    assert abs(loss_new - loss_old) <= 1e-6 * abs(loss_old)
    assert rel_err(st.E0.cpu().numpy(), U_w.cpu().numpy()) <= 2e-6
    lo_h = oracle.propagate_mean(*csr, E0, L)
    loss_o, Un, In = oracle.bpr_sgd(lo_h[:n_u], lo_h[n_u:], E0[:n_u], E0[n_u:], u, p, n, lr=0.05, reg=0.0)
    assert abs(loss_new / 2048 - loss_o) <= 2e-6
    assert np.abs(st.E0.cpu().numpy() - np.concatenate([Un, In])).max() <= 1e-5
