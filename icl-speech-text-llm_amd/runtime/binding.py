"""ctypes binding of libicl_hip.so (include/icl_hip.h).

PyTorch-ROCm is plumbing here: it owns device memory (``tensor.data_ptr()``) and the HIP stream
(``torch.cuda.current_stream().cuda_stream``); every arithmetic op of the hot path goes through the
C-ABI below.  The library is loaded from the in-tree ``lib/`` directory only; if it is missing the
import of this module fails loudly (no CPU / eager fallback exists in the product path).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "lib", "libicl_hip.so"))

ICL_BF16, ICL_F32 = 0, 1
EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_SWIGLU = 1, 2, 4, 8
ABI_VERSION = 5
SAMPLE_TOP_K_MAX = 1024      # SAMPLE_CAP of csrc/sampling.hip: candidate-list size of icl_sample_eos


# bench.py sets this to a list to time every GEMM launch with HIP events on the launch stream:
# entries are (tile, split_k, algorithmic_flops, start_event, end_event, (M, N, K, batch))
GEMM_PROFILE = None


class IclError(RuntimeError):
    """Raised when a libicl_hip entry point returns a negative code."""


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("R", c_void_p),
        ("workspace", c_void_p),
        ("lda", c_int64), ("ldw", c_int64), ("ldc", c_int64), ("ldr", c_int64),
        ("strideA", c_int64), ("strideC", c_int64), ("strideR", c_int64),
        ("M", c_int32), ("N", c_int32), ("K", c_int32), ("batch", c_int32), ("epilogue", c_int32),
        ("out_dtype", c_int32), ("res_dtype", c_int32), ("split_k", c_int32), ("tile", c_int32),
    ]


class AttnArgs(Structure):
    _fields_ = [
        ("Q", c_void_p), ("K", c_void_p), ("V", c_void_p), ("O", c_void_p),
        ("cu_seqlens", c_void_p), ("kv_lens", c_void_p), ("rel_bias", c_void_p), ("rel_gate", c_void_p),
        ("ldq", c_int64), ("ldk", c_int64), ("ldv", c_int64), ("ldo", c_int64),
        ("n_seqs", c_int32), ("max_seqlen", c_int32), ("n_heads", c_int32), ("head_dim", c_int32),
        ("causal", c_int32), ("rel_span", c_int32), ("scale", c_float), ("reserved", c_int32),
        ("kv_seq_stride", c_int64), ("kv_head_stride", c_int64),
    ]


# name -> (restype, argtypes); mirrors include/icl_hip.h one to one
_SIGNATURES = {
    "icl_abi_version": (c_int, []),
    "icl_last_error": (c_char_p, []),
    "icl_device_cu_count": (c_int, []),
    "icl_gemm_bf16": (c_int, [POINTER(GemmArgs), c_void_p]),
    "icl_gemm_rmsnorm_bf16": (c_int, [POINTER(GemmArgs), c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "icl_gemm_select_tile": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "icl_attn_fwd_bf16": (c_int, [POINTER(AttnArgs), c_void_p]),
    "icl_attn_decode_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                     c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "icl_attn_decode_rope_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float,
                                          c_void_p]),
    "icl_layernorm": (c_int, [c_void_p, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64,
                              c_void_p, c_int64, c_int32, c_int32, c_float, c_int32, c_int32, c_void_p]),
    "icl_rmsnorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_float,
                            c_int32, c_int32, c_void_p]),
    "icl_rope_kv_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "icl_gemm_rope_kv_bf16": (c_int, [POINTER(GemmArgs), c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "icl_pack_decode_weights": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "icl_embed_gather_interleave": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                            c_int32, c_int32, c_void_p]),
    "icl_argmax_eos": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                               c_int32, c_int32, c_void_p, c_void_p]),
    "icl_sample_eos": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int32, c_int32,
                               c_float, c_float, c_int32, c_float, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                               c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "icl_beam_step": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_float,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "icl_kv_copy_spans_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                       c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "icl_logmel_whisper": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_void_p,
                                   c_void_p, c_int64, c_void_p, c_void_p]),
    "icl_spec_to_xt": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int64, c_void_p]),
    "icl_fbank_kaldi": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_float, c_float,
                                c_void_p, c_void_p]),
    "icl_qformer_window_xattn": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                         c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "icl_beats_gate": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                               c_void_p]),
    "icl_axpby_cast": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32, c_float, c_void_p,
                               c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "icl_lora_down_bf16": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32, c_float, c_int32,
                                   c_void_p]),
    "icl_beats_patchify": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "icl_beats_posconv_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                       c_void_p]),
    "icl_gather_rows_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "icl_cross_entropy": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load_library() -> ctypes.CDLL:
    """Load lib/libicl_hip.so (once) and type every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ICL_LIB_PATH") or LIB_PATH        # A/B runs against another build of the same ABI
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C icl-speech-text-llm_amd/csrc`). There is no CPU fallback for the HIP path.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    ver = lib.icl_abi_version()
    if ver != ABI_VERSION:
        raise ImportError(f"libicl_hip ABI version {ver} != binding version {ABI_VERSION}")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().icl_last_error()
        raise IclError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def _eos_pair(eos_id):
    """int | (int,) | (int, int) -> (eos_id, eos_id2); -1 = unused (HF takes one id or a list; two cover the reference's models)."""
    if isinstance(eos_id, (tuple, list)):
        ids = [int(e) for e in eos_id]
        if not 1 <= len(ids) <= 2:
            raise ValueError(f"one or two EOS ids are supported, not {ids}")
        return ids[0], (ids[1] if len(ids) == 2 and ids[1] != ids[0] else -1)
    return int(eos_id), -1


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return ICL_BF16
    if t.dtype == torch.float32:
        return ICL_F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise IclError("libicl_hip operands must live on the GPU (no CPU fallback in the product path)")


# ------------------------------------------------------------------------------------------------
# thin wrappers (shapes are taken from the tensors; leading dimensions from strides)
# ------------------------------------------------------------------------------------------------
def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias=None, residual=None, gelu=False,
         swiglu=False, split_k: int = 1, workspace=None, tile: int = 0, M=None, K=None, lda=None,
         batch: int = 1, stride_a: int = 0, stride_c: int = 0, stride_r: int = 0, rope=None, N=None) -> torch.Tensor:
    """out = epilogue(a @ w.T).  a: bf16 [M,K] (row stride lda), w: bf16 [N,K], out: bf16|f32 [M,N'].

    ``rope`` = (k_off, v_off, cos, sin, pos, seq_ids, kcache, vcache, n_heads, head_dim, max_len[, kv_rows_to_c]) runs the QKV projection
    with RoPE + KV-cache append fused into its epilogue (icl_gemm_rope_kv_bf16; see ``rope_fusable``)."""
    _require_gpu(a, w, out, bias, residual, workspace)
    lib = load_library()
    g = GemmArgs()
    N = w.shape[0] if N is None else N      # tile 5 takes the decode-packed copy (rows padded to 16): pass the true N
    g.A, g.W, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.bias, g.R, g.workspace = _ptr(bias), _ptr(residual), _ptr(workspace)
    g.lda = a.stride(-2) if lda is None else lda
    g.ldw = w.stride(0)
    g.ldc = out.stride(-2)
    g.ldr = residual.stride(-2) if residual is not None else 0
    g.strideA, g.strideC, g.strideR = stride_a, stride_c, stride_r
    g.M = a.shape[-2] if M is None else M
    g.N = N
    g.K = w.shape[1] if K is None else K
    g.batch = batch
    epi = 0
    if bias is not None:
        assert bias.dtype == torch.float32
        epi |= EPI_BIAS
    if gelu:
        epi |= EPI_GELU
    if residual is not None:
        epi |= EPI_RESIDUAL
    if swiglu:
        epi |= EPI_SWIGLU
    g.epilogue = epi
    g.out_dtype = _dt(out)
    g.res_dtype = _dt(residual) if residual is not None else ICL_F32
    g.split_k = split_k
    if tile == 0:  # resolve the library's auto choice here so a profiler hook knows which kernel ran
        tile = lib.icl_gemm_select_tile(g.M, N, g.K, batch, split_k)
    g.tile = tile
    if split_k > 1 and workspace is not None:
        assert workspace.dtype == torch.float32 and workspace.numel() >= split_k * g.M * N
    if rope is not None:
        k_off, v_off, cos, sin, pos, seq_ids, kcache, vcache, n_heads, head_dim, max_len = rope[:11]
        kv_rows_to_c = int(rope[11]) if len(rope) > 11 else 1
        _require_gpu(cos, sin, pos, seq_ids, kcache, vcache)

        def launch():
            _check(lib.icl_gemm_rope_kv_bf16(ctypes.byref(g), k_off, v_off, cos.data_ptr(), sin.data_ptr(), pos.data_ptr(),
                                             _ptr(seq_ids), _ptr(kcache), _ptr(vcache), n_heads, head_dim, max_len,
                                             kv_rows_to_c, _stream()), "icl_gemm_rope_kv_bf16")
    else:
        def launch():
            _check(lib.icl_gemm_bf16(ctypes.byref(g), _stream()), "icl_gemm_bf16")
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        GEMM_PROFILE.append((tile, split_k, 2.0 * g.M * N * g.K * batch, e0, e1, (g.M, N, g.K, batch)))
        return out
    launch()
    return out


def gemm_rmsnorm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, gamma: torch.Tensor, eps: float, xn: torch.Tensor, *,
                 residual=None, split_k: int = 1, workspace=None, tile: int = 0, N=None, K=None) -> torch.Tensor:
    """out = residual + a @ w.T (f32) and xn[:, :N] = bf16(rmsnorm(out) * gamma): the decode form of a projection back into the
    residual stream followed by the next RMSNorm (icl_gemm_rmsnorm_bf16: with split_k > 1 one kernel reduces, adds, stores and
    normalises)."""
    _require_gpu(a, w, out, gamma, xn, residual, workspace)
    lib = load_library()
    g = GemmArgs()
    N = w.shape[0] if N is None else N
    g.A, g.W, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.bias, g.R, g.workspace = 0, _ptr(residual), _ptr(workspace)
    g.lda, g.ldw, g.ldc = a.stride(-2), w.stride(0), out.stride(-2)
    g.ldr = residual.stride(-2) if residual is not None else 0
    g.strideA = g.strideC = g.strideR = 0
    g.M, g.N, g.K, g.batch = a.shape[-2], N, (w.shape[1] if K is None else K), 1
    g.epilogue = EPI_RESIDUAL if residual is not None else 0
    g.out_dtype, g.res_dtype, g.split_k = _dt(out), ICL_F32, split_k
    g.tile = tile if tile else lib.icl_gemm_select_tile(g.M, N, g.K, 1, split_k)
    assert out.dtype == torch.float32 and xn.dtype == torch.bfloat16 and gamma.dtype == torch.float32
    if split_k > 1 and workspace is not None:
        assert workspace.dtype == torch.float32 and workspace.numel() >= split_k * g.M * N
    _check(lib.icl_gemm_rmsnorm_bf16(ctypes.byref(g), gamma.data_ptr(), eps, xn.data_ptr(), xn.stride(0), _stream()),
           "icl_gemm_rmsnorm_bf16")
    return out


def pack_decode_weights(w: torch.Tensor, K=None) -> torch.Tensor:
    """Decode-packed copy of a row-major bf16 weight [N, >=K] for ``gemm(..., tile=5, N=N)`` (icl_pack_decode_weights)."""
    _require_gpu(w)
    assert w.dtype == torch.bfloat16 and w.dim() == 2
    N, K = w.shape[0], (w.shape[1] if K is None else K)
    out = torch.empty((N + 15) // 16 * 16, K, dtype=torch.bfloat16, device=w.device)
    _check(load_library().icl_pack_decode_weights(w.data_ptr(), w.stride(0), N, K, out.data_ptr(), _stream()),
           "icl_pack_decode_weights")
    return out


def rope_fusable(M: int, n_heads: int, head_dim: int, K: int) -> bool:
    """True when the QKV projection [M, 3*n_heads*head_dim] x K runs on the 256x256 tile with the fused RoPE epilogue."""
    hd = n_heads * head_dim
    return (head_dim == 128 and hd % 256 == 0 and K >= 128 and
            load_library().icl_gemm_select_tile(M, 3 * hd, K, 1, 1) == 3)


def attn_fwd(q, k, v, out, cu_seqlens, max_seqlen: int, n_heads: int, head_dim: int, scale: float, *,
             causal=False, kv_lens=None, rel_bias=None, rel_gate=None, rel_span: int = 0, kv_cache_max_len: int = 0):
    """``kv_cache_max_len`` > 0: k / v are KV-cache tensors [n_seqs, n_heads, max_len, head_dim] (read in place)."""
    _require_gpu(q, k, v, out, cu_seqlens, kv_lens, rel_bias, rel_gate)
    lib = load_library()
    a = AttnArgs()
    a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.cu_seqlens, a.kv_lens = cu_seqlens.data_ptr(), _ptr(kv_lens)
    a.rel_bias, a.rel_gate = _ptr(rel_bias), _ptr(rel_gate)
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.reserved = 0
    if kv_cache_max_len > 0:
        a.ldk = a.ldv = head_dim
        a.kv_seq_stride, a.kv_head_stride = n_heads * kv_cache_max_len * head_dim, kv_cache_max_len * head_dim
    else:
        a.kv_seq_stride = a.kv_head_stride = 0
    a.n_seqs = cu_seqlens.numel() - 1
    a.max_seqlen, a.n_heads, a.head_dim = max_seqlen, n_heads, head_dim
    a.causal, a.rel_span, a.scale = int(causal), rel_span, scale
    assert cu_seqlens.dtype == torch.int32
    _check(lib.icl_attn_fwd_bf16(ctypes.byref(a), _stream()), "icl_attn_fwd_bf16")
    return out


def attn_decode(q, kcache, vcache, out, lens, n_heads: int, head_dim: int, max_len: int, scale: float):
    _require_gpu(q, kcache, vcache, out, lens)
    assert lens.dtype == torch.int32
    _check(load_library().icl_attn_decode_bf16(q.data_ptr(), q.stride(0), kcache.data_ptr(), vcache.data_ptr(),
                                               out.data_ptr(), out.stride(0), lens.data_ptr(), q.shape[0],
                                               n_heads, head_dim, max_len, scale, _stream()),
           "icl_attn_decode_bf16")
    return out


def attn_decode_rope(qkv, k_off: int, v_off: int, cos, sin, pos, seq_ids, kcache, vcache, out, lens, n_heads: int,
                     head_dim: int, max_len: int, scale: float):
    """One decode step's RoPE + cache append + attention in one launch (icl_attn_decode_rope_bf16): bit-identical to
    ``rope_kv`` followed by ``attn_decode``; ``qkv`` is left as the projection wrote it."""
    _require_gpu(qkv, cos, sin, pos, seq_ids, kcache, vcache, out, lens)
    assert lens.dtype == torch.int32 and pos.dtype == torch.int32 and qkv.dtype == torch.bfloat16
    _check(load_library().icl_attn_decode_rope_bf16(qkv.data_ptr(), qkv.stride(0), k_off, v_off, cos.data_ptr(), sin.data_ptr(),
                                                    pos.data_ptr(), _ptr(seq_ids), kcache.data_ptr(), vcache.data_ptr(),
                                                    out.data_ptr(), out.stride(0), lens.data_ptr(), qkv.shape[0], n_heads,
                                                    head_dim, max_len, scale, _stream()), "icl_attn_decode_rope_bf16")
    return out


def layernorm(x, gamma, beta, out, eps: float, *, res=None, alpha: float = 1.0, out2=None, M=None, N=None):
    _require_gpu(x, gamma, beta, out, res, out2)
    M = x.shape[0] if M is None else M
    N = x.shape[1] if N is None else N
    if res is not None:
        assert res.dtype == x.dtype and res.stride(0) == x.stride(0)
    _check(load_library().icl_layernorm(x.data_ptr(), x.stride(0), _ptr(res), alpha, gamma.data_ptr(),
                                        beta.data_ptr(), out.data_ptr(), out.stride(0), _ptr(out2),
                                        out2.stride(0) if out2 is not None else 0, M, N, eps, _dt(x), _dt(out),
                                        _stream()), "icl_layernorm")
    return out


def rmsnorm(x, gamma, out, eps: float, *, M=None, N=None):
    _require_gpu(x, gamma, out)
    M = x.shape[0] if M is None else M
    N = x.shape[1] if N is None else N
    _check(load_library().icl_rmsnorm(x.data_ptr(), x.stride(0), gamma.data_ptr(), out.data_ptr(), out.stride(0),
                                      M, N, eps, _dt(x), _dt(out), _stream()), "icl_rmsnorm")
    return out


def rope_kv(qkv, k_off: int, v_off: int, cos, sin, pos, seq_ids, kcache, vcache, n_heads: int, head_dim: int,
            max_len: int, M=None):
    _require_gpu(qkv, cos, sin, pos, seq_ids, kcache, vcache)
    M = qkv.shape[0] if M is None else M
    _check(load_library().icl_rope_kv_bf16(qkv.data_ptr(), qkv.stride(0), k_off, v_off, cos.data_ptr(),
                                           sin.data_ptr(), pos.data_ptr(), _ptr(seq_ids), _ptr(kcache),
                                           _ptr(vcache), M, n_heads, head_dim, max_len, _stream()),
           "icl_rope_kv_bf16")


def embed_gather_interleave(src_idx, table, speech, out):
    _require_gpu(src_idx, table, speech, out)
    assert src_idx.dtype == torch.int32 and table.dtype == torch.bfloat16 and out.dtype == torch.float32
    _check(load_library().icl_embed_gather_interleave(src_idx.data_ptr(), table.data_ptr(), _ptr(speech),
                                                      out.data_ptr(), src_idx.numel(), table.shape[1],
                                                      table.shape[0], 0 if speech is None else speech.shape[0],
                                                      _stream()), "icl_embed_gather_interleave")
    return out


def argmax_eos(logits, eos_id, pad_id: int, finished, out_tokens, step: int, next_ids, V=None):
    _require_gpu(logits, finished, out_tokens, next_ids)
    e1, e2 = _eos_pair(eos_id)
    _check(load_library().icl_argmax_eos(logits.data_ptr(), logits.stride(0), logits.shape[0],
                                         logits.shape[1] if V is None else V, e1, e2, pad_id, finished.data_ptr(),
                                         out_tokens.data_ptr(), out_tokens.stride(0), step, next_ids.data_ptr(),
                                         _stream()), "icl_argmax_eos")


def sample_eos(logits, work, uniforms, eos_id, pad_id: int, finished, out_tokens, step: int, next_ids, *,
               temperature: float = 1.0, top_k: int = 50, top_p: float = 1.0, repetition_penalty: float = 1.0, V=None,
               debug=None):
    """Sampled decode tail: tokens generated so far (``out_tokens[:, :step]``) feed the repetition penalty; ``uniforms`` f32 [B]
    are the draws; ``debug`` = (ids int32 [B,cap], probs f32 [B,cap], count int32 [B]) receives the kept distribution."""
    _require_gpu(logits, work, uniforms, finished, out_tokens, next_ids)
    assert logits.dtype == torch.float32 and work.dtype == torch.float32 and uniforms.dtype == torch.float32
    V = logits.shape[1] if V is None else V
    e1, e2 = _eos_pair(eos_id)
    dbg = (None, None, None, 0)
    if debug is not None:
        _require_gpu(*debug)
        dbg = (debug[0].data_ptr(), debug[1].data_ptr(), debug[2].data_ptr(), debug[0].shape[1])
    _check(load_library().icl_sample_eos(logits.data_ptr(), logits.stride(0), logits.shape[0], V, work.data_ptr(),
                                         work.stride(0), out_tokens.data_ptr(), out_tokens.stride(0), step,
                                         repetition_penalty, temperature, top_k, top_p, uniforms.data_ptr(), e1, e2, pad_id,
                                         finished.data_ptr(), out_tokens.data_ptr(), out_tokens.stride(0), step,
                                         next_ids.data_ptr(), dbg[0], dbg[1], dbg[2], dbg[3], _stream()), "icl_sample_eos")


class BeamState:
    """Device-resident state of ``icl_beam_step`` for B rows x K beams x T new tokens (see include/icl_hip.h)."""

    def __init__(self, get, B: int, K: int, T: int, pad_id: int):
        f32, i32 = torch.float32, torch.int32
        self.B, self.K, self.T = B, K, T
        self.run_score, self.fin_score = get("gen_beam_run_score", (B, K), f32), get("gen_beam_fin_score", (B, K), f32)
        self.run_seq, self.fin_seq = get("gen_beam_run_seq", (B, K, T), i32), get("gen_beam_fin_seq", (B, K, T), i32)
        self.fin_len, self.fin_flag = get("gen_beam_fin_len", (B, K), i32), get("gen_beam_fin_flag", (B, K), i32)
        self.unsat = get("gen_beam_unsat", (B,), i32)
        self.next_ids, self.parent = get("gen_beam_next", (B * K,), i32), get("gen_beam_parent", (B * K,), i32)
        self.run_score.fill_(-1.0e9)
        self.run_score[:, 0] = 0.0
        self.fin_score.fill_(-1.0e9)
        self.run_seq.fill_(pad_id)
        self.fin_seq.fill_(pad_id)
        self.fin_len.zero_()
        self.fin_flag.zero_()
        self.unsat.fill_(1)


def beam_step(logits, state: BeamState, step: int, eos_id, length_penalty: float, V=None, repetition_penalty: float = 1.0):
    """One beam-search step over ``logits`` f32 [B, V] (step 0: the prompt's distribution, shared by the K beams) or
    [B * K, V] (row b * K + k = running beam k of row b)."""
    _require_gpu(logits, state.run_score)
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    rows = logits.shape[0] // state.B
    assert rows * state.B == logits.shape[0] and rows in (1, state.K)
    e1, e2 = _eos_pair(eos_id)
    _check(load_library().icl_beam_step(logits.data_ptr(), logits.stride(0), rows, state.B,
                                        logits.shape[1] if V is None else V, state.K, state.T, step, e1, e2,
                                        float(length_penalty), float(repetition_penalty), state.run_score.data_ptr(),
                                        state.run_seq.data_ptr(),
                                        state.fin_score.data_ptr(), state.fin_seq.data_ptr(), state.fin_len.data_ptr(),
                                        state.fin_flag.data_ptr(), state.unsat.data_ptr(), state.next_ids.data_ptr(),
                                        state.parent.data_ptr(), _stream()), "icl_beam_step")


def kv_copy_spans(src, dst, n_rows: int, *, src_seq=None, src_t0=None, dst_seq=None, dst_t0=None, n_t=None, n_fixed: int = 0):
    """src / dst: bf16 [layers][seqs][heads][positions][head_dim] (any strides on the first three dims); copies, per row r,
    layer and head, ``n_t[r]`` (or ``n_fixed``) positions from (src_seq[r], src_t0[r]) to (dst_seq[r], dst_t0[r])."""
    _require_gpu(src, dst, src_seq, src_t0, dst_seq, dst_t0, n_t)
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.bfloat16 and src.dim() == 5 and dst.dim() == 5
    assert src.stride(4) == 1 and dst.stride(4) == 1 and src.stride(3) == src.shape[4] and dst.stride(3) == dst.shape[4]
    assert src.shape[0] == dst.shape[0] and src.shape[2] == dst.shape[2] and src.shape[4] == dst.shape[4]
    for t in (src_seq, src_t0, dst_seq, dst_t0, n_t):
        assert t is None or (t.dtype == torch.int32 and t.numel() >= n_rows)
    _check(load_library().icl_kv_copy_spans_bf16(src.data_ptr(), dst.data_ptr(), src.stride(0), src.stride(1), src.stride(2),
                                                 dst.stride(0), dst.stride(1), dst.stride(2), _ptr(src_seq), _ptr(src_t0),
                                                 _ptr(dst_seq), _ptr(dst_t0), _ptr(n_t), n_fixed, n_rows, src.shape[0],
                                                 src.shape[2], src.shape[4], src.shape[1], dst.shape[1], src.shape[3],
                                                 dst.shape[3], _stream()), "icl_kv_copy_spans_bf16")


def logmel_whisper(wav, wav_lens, mel_filters, n_mel: int, spec, xt, workspace):
    _require_gpu(wav, wav_lens, mel_filters, spec, xt, workspace)
    assert wav.dtype == torch.float32 and mel_filters.dtype == torch.float64 and wav_lens.dtype == torch.int32
    _check(load_library().icl_logmel_whisper(wav.data_ptr(), wav.stride(0), wav_lens.data_ptr(),
                                             mel_filters.data_ptr(), n_mel, wav.shape[0], _ptr(spec), _ptr(xt),
                                             xt.stride(-2) if xt is not None else 0, workspace.data_ptr(),
                                             _stream()), "icl_logmel_whisper")


def spec_to_xt(spec, xt):
    _require_gpu(spec, xt)
    assert spec.dtype == torch.float32 and spec.is_contiguous()
    _check(load_library().icl_spec_to_xt(spec.data_ptr(), spec.shape[1], spec.shape[0], xt.data_ptr(),
                                         xt.stride(-2), _stream()), "icl_spec_to_xt")


def fbank_kaldi(wav, wav_lens, mel_banks, max_frames: int, mean: float, std: float, out):
    _require_gpu(wav, wav_lens, mel_banks, out)
    assert wav.dtype == torch.float32 and mel_banks.dtype == torch.float64 and wav_lens.dtype == torch.int32
    _check(load_library().icl_fbank_kaldi(wav.data_ptr(), wav.stride(0), wav_lens.data_ptr(), mel_banks.data_ptr(),
                                          wav.shape[0], max_frames, mean, std, out.data_ptr(), _stream()),
           "icl_fbank_kaldi")


def qformer_window_xattn(q, kv, v_off: int, out, n_audio: int, win_per_audio: int, win: int, rows_per_audio: int,
                         n_heads: int, scale: float):
    _require_gpu(q, kv, out)
    _check(load_library().icl_qformer_window_xattn(q.data_ptr(), q.stride(0), kv.data_ptr(), kv.stride(0), v_off,
                                                   out.data_ptr(), out.stride(0), n_audio, win_per_audio, win,
                                                   rows_per_audio, n_heads, scale, _stream()),
           "icl_qformer_window_xattn")


def beats_gate(qkv, grep_w, grep_b, grep_a, gate, n_heads: int, M=None):
    _require_gpu(qkv, grep_w, grep_b, grep_a, gate)
    M = qkv.shape[0] if M is None else M
    _check(load_library().icl_beats_gate(qkv.data_ptr(), qkv.stride(0), grep_w.data_ptr(), grep_b.data_ptr(),
                                         grep_a.data_ptr(), gate.data_ptr(), M, n_heads, _stream()),
           "icl_beats_gate")


def axpby_cast(x, out, alpha: float = 1.0, add=None):
    _require_gpu(x, out, add)
    M, N = x.shape
    _check(load_library().icl_axpby_cast(x.data_ptr(), x.stride(0), _dt(x), _ptr(add),
                                         add.stride(0) if add is not None else 0,
                                         _dt(add) if add is not None else ICL_F32, alpha, out.data_ptr(),
                                         out.stride(0), _dt(out), M, N, _stream()), "icl_axpby_cast")
    return out


def lora_down(x, K0: int, a, r_total: int, scale: float, M=None):
    _require_gpu(x, a)
    M = x.shape[0] if M is None else M
    _check(load_library().icl_lora_down_bf16(x.data_ptr(), x.stride(0), K0, a.data_ptr(), a.stride(0), r_total,
                                             scale, M, _stream()), "icl_lora_down_bf16")


def device_cu_count() -> int:
    return load_library().icl_device_cu_count()


def beats_patchify(fbank, cu_rows, total_rows: int, out):
    _require_gpu(fbank, cu_rows, out)
    _check(load_library().icl_beats_patchify(fbank.data_ptr(), fbank.shape[1], cu_rows.data_ptr(), fbank.shape[0],
                                             total_rows, out.data_ptr(), _stream()), "icl_beats_patchify")


def beats_posconv_pack(x, cu_rows, valid_rows, n_audio: int, total_rows: int, groups: int, xg):
    _require_gpu(x, cu_rows, valid_rows, xg)
    _check(load_library().icl_beats_posconv_pack(x.data_ptr(), cu_rows.data_ptr(), valid_rows.data_ptr(), n_audio,
                                                 total_rows, x.shape[1], groups, xg.data_ptr(), _stream()),
           "icl_beats_posconv_pack")


def gather_rows(src, idx, out, N=None):
    _require_gpu(src, idx, out)
    assert src.dtype == torch.float32 and out.dtype == torch.float32 and idx.dtype == torch.int32
    _check(load_library().icl_gather_rows_f32(src.data_ptr(), src.stride(0), idx.data_ptr(), out.data_ptr(),
                                              out.stride(0), idx.numel(), src.shape[1] if N is None else N,
                                              _stream()), "icl_gather_rows_f32")
    return out


def cross_entropy(logits, labels, row_loss, mean_loss, V=None):
    _require_gpu(logits, labels, row_loss, mean_loss)
    assert logits.dtype == torch.float32 and labels.dtype == torch.int32
    _check(load_library().icl_cross_entropy(logits.data_ptr(), logits.stride(0), labels.data_ptr(), labels.numel(),
                                            logits.shape[1] if V is None else V, row_loss.data_ptr(),
                                            mean_loss.data_ptr(), _stream()), "icl_cross_entropy")
    return mean_loss
