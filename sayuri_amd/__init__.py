"""sayuri_amd -- MI355X-native NN-evaluation backend for Sayuri's self-play hot path.

Only what the path needs lives here: `csrc/hip` (hand-written gfx950 kernels + the C-ABI
of include/sayuri_hip.h), `csrc/host` (C++ host side mirroring the reference's
NetworkForwardPipe plugin interface) and thin Python bindings used by tests and bench.py.
"""
__version__ = "0.1.0"
