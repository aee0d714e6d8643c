"""`-m gpu`: the Qwen2-Audio path (SURVEY.md §8 row a7 / BASELINE config 4) at miniature dims against the oracle (which is
itself pinned to HF Qwen2AudioForConditionalGeneration by tests/golden/qwen2_audio_tiny.npz)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def model():
    from icl_speech_text_llm_amd.models.model_factory import ModelFactory
    return ModelFactory.create_model("qwen2", device="cuda", arch="tiny", model_path="none").eval()


def _oracle_sd(model):
    return {k: v.float().cpu() for k, v in model.model.state_dict().items()}


def _batch(model, secs=(2.5, 7.0)):
    p = model.input_processor
    audios = [np.clip(np.random.default_rng(10 + i).normal(0, 0.1, int(16000 * s)), -1, 1).astype(np.float32) for i, s in enumerate(secs)]
    conv = [{"role": "system", "content": "Classify the sentiment."},
            {"role": "user", "content": [{"type": "text", "text": "Here are few examples to learn from:\n"},
                                         {"type": "audio", "audio_url": "x"}, {"type": "text", "text": "Label: positive\n"},
                                         {"type": "text", "text": "\nNow analyze this input:\n"}, {"type": "audio", "audio_url": "y"}]}]
    text = p.apply_chat_template(conv, add_generation_prompt=True, tokenize=False)
    enc = p(text=text + "negative<|im_end|>", audios=audios, return_tensors="pt", sampling_rate=16000)
    prompt_len = p(text=text, audios=audios).input_ids.shape[1]
    return enc, prompt_len, audios


def test_audio_tower_matches_oracle(model):
    from oracle import audio_frontend as af, models as om
    enc, _, audios = _batch(model)
    feats, out_lens = model.runtime.encode_audio(input_features=enc.input_features, mel_lens=enc.feature_attention_mask.sum(-1).tolist())
    sd = _oracle_sd(model)
    spec = torch.stack([torch.from_numpy(af.whisper_logmel(a, n_mels=128)) for a in audios])
    assert float((enc.input_features - spec).abs().max()) < 1e-4          # K1 with 128 mel bins
    ref, ref_lens = om.qwen_audio_features(sd, spec, enc.feature_attention_mask.sum(-1).tolist(), model.cfg.audio.n_heads, rnd=om.bf16_round)
    assert out_lens == ref_lens
    for i, n in enumerate(out_lens):
        r = _rel(feats[i, :n], ref[i, :n])
        print(f"qwen audio tower[{i}]: rel {r:.2e}")
        assert r < 1.5e-3, i               # measured 5.5e-4 .. 6.9e-4


def test_forward_and_generate_match_oracle(model):
    from oracle import audio_frontend as af, models as om
    enc, prompt_len, audios = _batch(model)
    batch = {"input_ids": enc.input_ids, "attention_mask": enc.attention_mask, "input_features": enc.input_features,
             "feature_attention_mask": enc.feature_attention_mask, "prompt_length": torch.tensor([prompt_len])}
    out = model.forward(batch)
    S = enc.input_ids.shape[1]
    assert out["logits"].shape == (1, S, model.cfg.llm.vocab)
    labels = out["labels"].cpu()
    assert (labels[0, :prompt_len] == -100).all() and torch.equal(labels[0, prompt_len:], enc.input_ids[0, prompt_len:])
    # oracle
    sd = _oracle_sd(model)
    spec = torch.stack([torch.from_numpy(af.whisper_logmel(a, n_mels=128)) for a in audios])
    feats, lens = om.qwen_audio_features(sd, spec, enc.feature_attention_mask.sum(-1).tolist(), model.cfg.audio.n_heads, rnd=om.bf16_round)
    lsd = {k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
    c = model.cfg.llm
    llm = om.LlamaOracle(lsd, c.n_heads, c.rms_eps, c.rope_theta, c.lora_scale, rnd=om.bf16_round)
    emb = llm.embed(enc.input_ids[0])
    pos = (enc.input_ids[0] == model.cfg.audio_token_id).nonzero().flatten()
    emb[pos] = torch.cat([feats[i, :n] for i, n in enumerate(lens)])
    ref_logits, ref_loss = llm.forward(emb[None], labels)
    r = _rel(out["logits"], ref_logits)
    print(f"qwen forward logits rel {r:.2e}")
    assert r < 4e-3                      # measured 2.0e-3
    assert abs(float(out["loss"]) - float(ref_loss)) < 5e-3 * max(1.0, abs(float(ref_loss)))
    # generation: prompt only
    gen_batch = {k: (v[:, :prompt_len] if k in ("input_ids", "attention_mask") else v) for k, v in batch.items()}
    texts = model.generate_output(gen_batch)
    assert isinstance(texts, list) and len(texts) == 1 and isinstance(texts[0], str)
    res = model.runtime.generate(model.runtime.segments_from_ids([enc.input_ids[0, :prompt_len].tolist()], lens),
                                 model.runtime.encode_audio(input_features=enc.input_features,
                                                            mel_lens=enc.feature_attention_mask.sum(-1).tolist())[0],
                                 max_new_tokens=5, want_first_logits=True, suppress_eos=True)
    _, first = llm.generate_greedy(emb[None, :prompt_len], 1, -1, c.pad_id, return_first_logits=True)
    e = float((res.first_logits.cpu() - first).abs().max())
    print(f"qwen first-step logits max abs {e:.2e}")
    assert e < 5e-3                      # measured 2.1e-3
    assert res.tokens.shape == (1, 5)
    # beam search through the plugin's batch keys (generation_config knobs; transformers _beam_search rules)
    # the reference's generate_output reads no knob from the batch dict (max_new_tokens=10 is hard-coded, the rest is the model's
    # generation config): keys such as MultiTaskModel's do not change the answer here either
    plain = model.generate_ids(dict(gen_batch))
    ignored = model.generate_ids(dict(gen_batch, num_beams=3, max_new_tokens=4, do_sample=True, temperature=0.3))
    assert torch.equal(plain.tokens, ignored.tokens)
    old_cfg = dict(model.generation_config)
    try:
        model.generation_config.update(num_beams=3, length_penalty=1.0, max_new_tokens=4)
        beams = model.generate_ids(dict(gen_batch))
    finally:
        model.generation_config.clear(); model.generation_config.update(old_cfg)
    want = llm.generate_beam(emb[None, :prompt_len], 4, c.eos_id, c.pad_id, 3, 1.0)
    print(f"qwen beams: gpu {beams.tokens.tolist()} oracle {want.tolist()}")
    assert beams.tokens[0, : want.shape[1]].tolist() == want[0].tolist()


def test_padded_batch_rows_are_packed(model):
    """Two rows of different length (right padded) in one batch: padding is stripped through attention_mask."""
    enc, prompt_len, _ = _batch(model)
    ids = enc.input_ids[0, :prompt_len]
    short = ids[: prompt_len - 4]
    # second row: text-only suffix of the prompt (no audio tokens) so the audio count still matches
    aid = model.cfg.audio_token_id
    tail = ids[(ids == aid).nonzero().flatten()[-1] + 1:]
    S = prompt_len
    row2 = torch.full((S,), model.cfg.llm.pad_id, dtype=torch.long)
    row2[: tail.numel()] = tail
    att = torch.ones(2, S, dtype=torch.long)
    att[1, tail.numel():] = 0
    batch = {"input_ids": torch.stack([ids, row2]), "attention_mask": att, "input_features": enc.input_features,
             "feature_attention_mask": enc.feature_attention_mask}
    out = model.generate_output(batch)
    assert len(out) == 2


def test_cli_qwen2_end_to_end(tmp_path):
    import json, os
    from icl_speech_text_llm_amd.inference.inference import main
    rc = main(["--peft_model_path", "", "--run_name", "q", "--dataset_type", "hvb", "--arch", "tiny", "--model_type", "qwen2",
               "--synthetic_items", "3", "--batch_size", "2", "--num_examples", "2", "--results_dir", str(tmp_path), "--device", "cuda"])
    assert rc == 0
    res = [f for f in os.listdir(tmp_path) if f.endswith("_results.json")]
    assert len(res) == 1 and len(json.load(open(tmp_path / res[0]))) == 3


def test_audio_longer_than_30_seconds_is_truncated_like_the_feature_extractor(model):
    """WhisperFeatureExtractor (inside the reference's Qwen2-Audio processor, models/custom_qwen.py:57) cuts every clip to 30 s =
    3000 mel frames = 750 audio tokens: a 31.5 s clip must expand to 750 placeholders and match the oracle on the first 30 s."""
    from oracle import audio_frontend as af, models as om
    enc, prompt_len, audios = _batch(model, secs=(31.5, 2.0))
    mel_lens = enc.feature_attention_mask.sum(-1).tolist()
    assert mel_lens[0] == 3000 and int((enc.input_ids[0] == model.cfg.audio_token_id).sum()) == 750 + ((200 - 1) // 2 + 1 - 2) // 2 + 1
    feats, out_lens = model.runtime.encode_audio(input_features=enc.input_features, mel_lens=mel_lens)
    raw, raw_lens = model.runtime.encode_audio(raw_wav=torch.nn.utils.rnn.pad_sequence([torch.from_numpy(a) for a in audios], batch_first=True),
                                               wav_lens=[len(a) for a in audios])
    assert out_lens == raw_lens == [750, 50]
    sd = _oracle_sd(model)
    spec = torch.stack([torch.from_numpy(af.whisper_logmel(a, n_mels=128)) for a in audios])
    ref, ref_lens = om.qwen_audio_features(sd, spec, mel_lens, model.cfg.audio.n_heads, rnd=om.bf16_round)
    assert ref_lens == out_lens
    for i, n in enumerate(out_lens):
        assert _rel(feats[i, :n], ref[i, :n]) < 1.5e-3 and _rel(raw[i, :n], ref[i, :n]) < 1.5e-3
    batch = {"input_ids": enc.input_ids[:, :prompt_len], "attention_mask": enc.attention_mask[:, :prompt_len],
             "input_features": enc.input_features, "feature_attention_mask": enc.feature_attention_mask}
    assert len(model.generate_output(batch)) == 1


def test_reference_customqwen_glue_end_to_end():
    """tests/golden/qwen_glue_e2e.npz: the REFERENCE's CustomQwen (unmodified) over HF Qwen2AudioForConditionalGeneration at the
    `tiny` dims on CPU — `forward` (labels = -100 up to prompt_length, loss, logits) and `generate_output` (generate(max_new_tokens=10),
    slice, decode).  This build's CustomQwen on the GPU, same weights (rebuilt from the seed: tests/golden/e2e_weights.py) and the
    same batch dict: same labels, loss within the bf16 tolerance, and the same ten generated ids (tightest greedy margin 12x the
    fp32-vs-bf16 oracle distance)."""
    import os
    import sys
    from icl_speech_text_llm_amd.models.model_factory import ModelFactory
    from icl_speech_text_llm_amd.runtime.qwen import normalize_qwen_keys
    gold = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, gold)
    from e2e_weights import qwen_batch, tiny_qwen_hf_model
    g = np.load(os.path.join(gold, "qwen_glue_e2e.npz"))
    cfg, hf = tiny_qwen_hf_model(int(g["weights_seed"]))
    model = ModelFactory.create_model("qwen2", device="cuda", arch="tiny", model_path="none", lora=False).eval()
    sd = normalize_qwen_keys({k: v.detach().clone() for k, v in hf.state_dict().items()})
    missing, unexpected = model.model.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "embed_positions" not in k] and not [k for k in unexpected if "embed_positions" not in k], (missing, unexpected)
    batch, prompt_len = qwen_batch(cfg)
    assert prompt_len == int(g["prompt_length"])
    out = model.forward(dict(batch))
    assert torch.equal(out["labels"][0].cpu(), torch.from_numpy(g["labels"]))
    assert abs(float(out["loss"]) - float(g["loss"])) < 2e-2 * max(1.0, abs(float(g["loss"])))
    r = _rel(out["logits"][0, -8:], torch.from_numpy(g["logits_tail"]))
    print(f"reference CustomQwen.forward vs GPU: loss {float(g['loss']):.4f} / {float(out['loss']):.4f}, last-8 logits rel {r:.2e}")
    assert r < 1e-2
    gen_batch = {k: (v[:, :prompt_len] if k in ("input_ids", "attention_mask") else v) for k, v in batch.items()}
    res = model.generate_ids(dict(gen_batch))
    print("reference generate_output ids", g["gen_ids"].tolist(), "| GPU", res.tokens[0].tolist())
    assert res.tokens[0].tolist() == g["gen_ids"].tolist()
