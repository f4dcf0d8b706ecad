"""spex_amd — MI355X-native implementation of the SPEX LightGCN / NGCF graph-convolution + negative-sampled scoring
hot path.  HIP kernels behind a C ABI (include/spex_hip.h, spex_amd/csrc/), a thin torch-facing layer (graph, ops,
trainer, dist) and drop-in modules with the reference's Python surface (spex_amd/dropin/).

Importing the package is cheap and GPU-free; the HIP library is loaded on first use and its absence is an error —
there is no CPU fallback anywhere in this package.
"""
__version__ = "0.1.0"
