"""Architecture descriptions of the networks on the ICL hot path.

The reference never states these numbers in-tree: they live in the checkpoints / HF configs that
``SALMONN.from_config`` loads (models/custom_salmon.py:64-97).  Defaults are the BASELINE.json C2
configuration (Whisper-large-v2 + BEATs iter3+ + 2-layer window Q-Former + Llama-2-7B);
``tiny_*`` constructors give the miniature shapes used by the parity tests.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Optional


@dataclass(frozen=True)
class WhisperCfg:
    d_model: int = 1280
    n_layers: int = 32
    n_heads: int = 20
    ffn: int = 5120
    n_mels: int = 80
    n_ctx: int = 1500  # positions after the stride-2 conv (3000 mel frames in)


@dataclass(frozen=True)
class BeatsCfg:
    embed: int = 512          # patch-embedding channels
    d_model: int = 768
    n_layers: int = 12
    n_heads: int = 12
    ffn: int = 3072
    conv_pos: int = 128
    conv_groups: int = 16
    num_buckets: int = 320
    max_distance: int = 800
    fbank_mean: float = 15.41663
    fbank_std: float = 6.55582

    @property
    def deep_norm_alpha(self) -> float:
        return (2.0 * self.n_layers) ** 0.25


@dataclass(frozen=True)
class QFormerCfg:
    hidden: int = 768
    n_layers: int = 2
    n_heads: int = 12
    ffn: int = 3072
    enc_width: int = 2048     # whisper d_model + beats d_model
    n_query: int = 1
    ln_eps: float = 1e-12
    second_per_window: float = 0.333333
    second_stride: float = 0.333333


@dataclass(frozen=True)
class LlamaCfg:
    hidden: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    ffn: int = 11008
    vocab: int = 32001        # 32000 + the [PAD] token SALMONN adds (SURVEY.md §7 "quirks")
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_pos: int = 4096       # Llama-2 (max_position_embeddings); HF-folder ingestion overrides it from config.json
    bos_id: int = 1
    eos_id: int = 2
    pad_id: int = 32000
    lora_rank: int = 8        # 0 = no LoRA
    lora_alpha: float = 32.0  # config/inference_config.py:36 (class default is 16, custom_salmon.py:45)
    lora_targets: tuple = ("q_proj", "v_proj")   # peft's Llama default (SALMONN); Qwen path: ("q_proj", "k_proj"), custom_qwen.py:75
    qkv_bias: bool = False    # Qwen2 attention has q/k/v biases

    @property
    def head_dim(self) -> int:
        return self.hidden // self.n_heads

    @property
    def lora_scale(self) -> float:
        return self.lora_alpha / self.lora_rank if self.lora_rank else 0.0


@dataclass(frozen=True)
class SalmonnCfg:
    whisper: WhisperCfg = field(default_factory=WhisperCfg)
    beats: Optional[BeatsCfg] = field(default_factory=BeatsCfg)
    qformer: QFormerCfg = field(default_factory=QFormerCfg)
    llama: LlamaCfg = field(default_factory=LlamaCfg)

    @staticmethod
    def llama2_7b() -> "SalmonnCfg":
        return SalmonnCfg()

    @staticmethod
    def llama2_13b() -> "SalmonnCfg":
        return SalmonnCfg(llama=LlamaCfg(hidden=5120, n_layers=40, n_heads=40, ffn=13824))

    @staticmethod
    def whisper_tiny_7b() -> "SalmonnCfg":
        """BASELINE.json configs[0]: Whisper-tiny + Llama-2-7B (text_only: the encoder is never run)."""
        w = WhisperCfg(d_model=384, n_layers=4, n_heads=6, ffn=1536)
        return SalmonnCfg(whisper=w, beats=None, qformer=QFormerCfg(enc_width=384))

    @staticmethod
    def tiny(use_beats: bool = True, lora: bool = True, vocab: int = 260) -> "SalmonnCfg":
        """Miniature shapes for parity tests (all HIP-path alignment rules hold: head_dim 64/128, K % 64)."""
        w = WhisperCfg(d_model=128, n_layers=2, n_heads=2, ffn=256)
        b = BeatsCfg(embed=64, d_model=128, n_layers=2, n_heads=2, ffn=256) if use_beats else None
        q = QFormerCfg(hidden=128, n_layers=2, n_heads=2, ffn=256, enc_width=128 + (128 if use_beats else 0))
        l = LlamaCfg(hidden=256, n_layers=2, n_heads=2, ffn=512, vocab=vocab, max_pos=2048, pad_id=vocab - 1,
                     lora_rank=8 if lora else 0)
        return SalmonnCfg(whisper=w, beats=b, qformer=q, llama=l)


@dataclass(frozen=True)
class QwenAudioCfg:
    """Qwen2-Audio (models/custom_qwen.py:51-55): Whisper-style 128-mel audio tower + AvgPool(2) + ln_post, a
    Linear projector and a Qwen2 decoder.  LM dims follow the Qwen2-Audio-7B-Instruct checkpoint (hidden 4096, 32 layers,
    32 heads, FFN 11008, vocab 156032) [upstream config.json — to be confirmed when the checkpoint is reachable]."""
    audio: WhisperCfg = field(default_factory=lambda: WhisperCfg(n_mels=128))
    llm: LlamaCfg = field(default_factory=lambda: LlamaCfg(vocab=156032, rms_eps=1e-5, rope_theta=10000.0, max_pos=8192,
                                                           bos_id=151643, eos_id=151645, pad_id=151643, qkv_bias=True,
                                                           lora_targets=("q_proj", "k_proj")))
    audio_token_id: int = 151646

    @staticmethod
    def tiny(lora: bool = True, vocab: int = 300) -> "QwenAudioCfg":
        a = WhisperCfg(d_model=128, n_layers=2, n_heads=2, ffn=256, n_mels=128)
        l = LlamaCfg(hidden=256, n_layers=2, n_heads=2, ffn=512, vocab=vocab, max_pos=4096, bos_id=1, eos_id=2,
                     pad_id=vocab - 1, lora_rank=8 if lora else 0, qkv_bias=True, lora_targets=("q_proj", "k_proj"))
        return QwenAudioCfg(audio=a, llm=l, audio_token_id=vocab - 2)


def with_llama(cfg: SalmonnCfg, **kw) -> SalmonnCfg:
    return replace(cfg, llama=replace(cfg.llama, **kw))
