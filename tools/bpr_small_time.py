#!/usr/bin/env python3
"""Small-batch BPR kernel latency: with / without the loss accumulation, T = 256 ... 16384 (Epinion2-sized tables)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spex_amd import _lib
from spex_amd.graph import _ptr, _stream

dev = torch.device("cuda:0")
U, I = 3186, 12407
Ur, Ir = torch.rand(U, 64, device=dev), torch.rand(I, 64, device=dev)
Uw, Iw = Ur.clone(), Ir.clone()
loss = torch.zeros(1, device=dev)
for T in (256, 1024, 2048, 4096, 16384):
    u = torch.randint(0, U, (T,), device=dev); p = torch.randint(0, I, (T,), device=dev); n = torch.randint(0, I, (T,), device=dev)
    for name, lp in (("with loss", _ptr(loss)), ("no loss", None)):
        def fn():
            _lib.call("spex_bpr_sgd_step_f32", _ptr(Ur), _ptr(Ir), _ptr(Uw), _ptr(Iw), U, I, _ptr(u), _ptr(p), _ptr(n), T, 64,
                      1e-6, 0.0, lp, _stream())
        for _ in range(10): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200): fn()
        e.record(); e.synchronize()
        print("T=%6d %-10s %.2f us" % (T, name, s.elapsed_time(e) / 200 * 1e3))
