"""MI355X-native ICL forward/generate hot path (SALMONN / Qwen2-Audio few-shot speech+text).

Sub-packages: ``csrc`` (HIP kernels + C-ABI), ``runtime`` (ctypes binding and the engines that chain
the kernels), ``models`` / ``inference`` / ``config`` / ``data`` / ``utils`` (host-side mirror of
the reference's plugin surface and CLI for this path).
"""
__version__ = "0.1.0"
