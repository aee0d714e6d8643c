#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ (run in the build container only; needs /root/reference and the
installed `transformers`).  Nothing here ships to the GPU box except the small .npz/.json outputs.

The reference has no tests and no fixtures of its own, and the arithmetic of its hot path lives in
third-party packages, so the oracle is pinned by vectors minted from
  (a) the installed transformers modules the reference calls (WhisperFeatureExtractor, WhisperEncoder,
      LlamaForCausalLM.forward/generate) and — as the architectural stand-in for SALMONN's Q-Former — Blip2QFormerModel,
      with seeded random weights at miniature dims;
  (b) the reference's OWN in-tree code, imported from /root/reference: CustomSALMONN.custom_prompt_wrap / forward /
      generate_output (models/custom_salmon.py) with the absent external `SALMONN.models.salmonn_org` package
      replaced by a minimal in-memory module (a tiny HF Llama + a byte tokenizer + a deterministic encode_speech),
      SalmonProcessor._format_default_prompt (data/model_processors.py) and clean_prediction (utils/evaluation_utils.py).
Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def g1_logmel():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor()
    out = {"mel_filters_T": fe.mel_filters.astype(np.float64)[::4]}
    for tag, L, seed in (("0p5s", 8000, 1234), ("7p3s", 116800, 1235), ("30s", 480000, 1236)):
        wav = np.clip(np.random.default_rng(seed).normal(0, 0.1, L), -1, 1).astype(np.float32)
        spec = fe(wav, sampling_rate=16000, return_tensors="np")["input_features"][0]
        out[f"spec_{tag}"] = spec[:, ::12].astype(np.float32)      # every 12th frame keeps the file small
        out[f"len_{tag}"], out[f"seed_{tag}"] = L, seed
    save("logmel.npz", **out)


def g2_whisper():
    from transformers import WhisperConfig
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    torch.manual_seed(0)
    cfg = WhisperConfig(d_model=32, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=64, num_mel_bins=80,
                        max_source_positions=1500)
    enc = WhisperEncoder(cfg).eval()
    with torch.no_grad():
        for p in enc.parameters():
            if p.requires_grad:
                p.copy_(torch.randn_like(p) * (0.15 if p.dim() > 1 else 0.3) + (1.0 if "layer_norm.weight" in "" else 0.0))
        c, t = torch.arange(80.0)[:, None], torch.arange(3000.0)[None, :]
        spec = (0.5 * torch.sin(0.01 * (c + 1.0) * t + c))[None]      # closed form: nothing to store
        out = enc(spec).last_hidden_state
    sd = {k: v for k, v in enc.state_dict().items() if k != "embed_positions.weight"}   # sinusoids are regenerated
    save("whisper_tiny.npz", out=out[0, ::25], **{"w:" + k: v for k, v in sd.items()})


def g3_qformer():
    from transformers import Blip2QFormerConfig, Blip2QFormerModel
    torch.manual_seed(1)
    cfg = Blip2QFormerConfig(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                             encoder_hidden_size=48, cross_attention_frequency=1, hidden_dropout_prob=0.0,
                             attention_probs_dropout_prob=0.0)
    m = Blip2QFormerModel(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * (0.2 if p.dim() > 1 else 0.3) + (1.0 if p.dim() == 1 and p.shape[0] == 32 and False else 0.0))
        query = torch.randn(5, 1, 32)
        enc = torch.randn(5, 17, 48)
        out = m(query_embeds=query, encoder_hidden_states=enc, encoder_attention_mask=torch.ones(5, 17, dtype=torch.long)).last_hidden_state
    # HF Blip2 names -> SALMONN Qformer.py (BERT) names
    ren = {}
    for k, v in m.state_dict().items():
        k2 = k.replace("layernorm.", "embeddings.LayerNorm.").replace(".attention.attention.", ".attention.self.")
        k2 = k2.replace(".crossattention.attention.", ".crossattention.self.")
        if k2.startswith("encoder.") or k2.startswith("embeddings."):
            ren["speech_Qformer.bert." + k2] = v
    save("qformer_tiny.npz", query=query, enc=enc, out=out, **{"w:" + k: v for k, v in ren.items()})


def _tiny_llama(seed=2, vocab=260):
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=2, vocab_size=vocab, rms_norm_eps=1e-5, max_position_embeddings=2048,
                      pad_token_id=vocab - 1, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.randn_like(p) * 0.3 if p.dim() > 1 else 1.0 + 0.1 * torch.randn_like(p))
    return m


def g4_llama():
    m = _tiny_llama()
    torch.manual_seed(3)
    emb = torch.randn(2, 23, 64) * 0.5
    with torch.no_grad():
        logits = m(inputs_embeds=emb).logits
        labels = torch.full((2, 23), -100)
        labels[:, -5:] = torch.randint(3, 250, (2, 5))
        loss = m(inputs_embeds=emb, labels=labels).loss
        free = m.generate(inputs_embeds=emb, attention_mask=torch.ones(2, 23, dtype=torch.long), max_new_tokens=10,
                          do_sample=False, num_beams=1, min_length=1, pad_token_id=259, eos_token_id=None)
        # declare the token row 0 emits at step 3 to be EOS: row 0 stops there and is pad-filled, row 1 continues
        eos_mid = int(free[0, 3])
        mid = m.generate(inputs_embeds=emb, attention_mask=torch.ones(2, 23, dtype=torch.long), max_new_tokens=10,
                         do_sample=False, num_beams=1, min_length=1, pad_token_id=259, eos_token_id=eos_mid)
        eos0 = int(free[0, 0])
        first = m.generate(inputs_embeds=emb[:1], attention_mask=torch.ones(1, 23, dtype=torch.long), max_new_tokens=10,
                           do_sample=False, num_beams=1, min_length=1, pad_token_id=259, eos_token_id=eos0)
    save("llama_tiny.npz", emb=emb, logits=logits, labels=labels, loss=loss, gen_free=free, eos_mid=eos_mid, gen_mid=mid,
         eos0=eos0, gen_eos0=first, **{"w:" + k: v for k, v in m.state_dict().items()})


def g5_reference_glue():
    """Runs the reference's CustomSALMONN (unmodified, imported from /root/reference) on a stub SALMONN."""
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    llama = _tiny_llama(seed=4)
    tok = ByteTokenizer(260)
    H = 64
    torch.manual_seed(5)
    proj = torch.randn(34, H) * 0.3

    class StubSALMONN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.llama_model = llama
            self.llama_tokenizer = tok

        @classmethod
        def from_config(cls, cfg):
            return cls()

        def encode_speech(self, spectrogram=None, raw_wav=None, audio_padding_mask=None):
            B = spectrogram.shape[0]
            feat = spectrogram.float().mean(dim=1)[:, :88 * 34].reshape(B, 88, 34)
            emb = torch.tanh(feat @ proj)
            return emb, torch.ones(B, 88, dtype=torch.long)

    for name in ("SALMONN", "SALMONN.models", "SALMONN.models.salmonn_org"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["SALMONN.models.salmonn_org"].SALMONN = StubSALMONN
    sys.path.insert(0, REF)
    from models.custom_salmon import CustomSALMONN as RefSALMONN      # the reference's own class
    from data.model_processors import SalmonProcessor as RefProcessor
    from data.master_config import DatasetType as RefDT, get_dataset_config as ref_cfg
    ref = RefSALMONN(lora=False, device=torch.device("cpu"))
    rproc = RefProcessor.__new__(RefProcessor)
    tmpl = ref_cfg(RefDT.VOXCELEB).prompt_template
    ex = [{"text": f"example sentence number {i} about things", "label": ["positive", "negative", "neutral"][i % 3]} for i in range(5)]
    torch.manual_seed(6)
    cases = {}
    for case, (mode, few, nspeech_ex) in {"text_only": ("text_only", "text", 0), "speech_text_ex": ("speech_only", "text", 0),
                                           "speech_speech_ex": ("speech_only", "speech", 2)}.items():
        exs = ex[:nspeech_ex] if few == "speech" else ex
        prompt = rproc._format_default_prompt(tmpl, "the query sentence to classify", exs, mode, few)
        samples = {"prompt": [prompt], "completion": ["positive"], "num_examples": torch.tensor([len(exs) if few == "speech" else 0])}
        if mode != "text_only":
            samples["spectrogram"] = torch.randn(1, 80, 3000) * 0.5
            samples["raw_wav"] = torch.zeros(1, 16000)
            samples["padding_mask"] = torch.zeros(1, 16000, dtype=torch.bool)
        if few == "speech":
            samples["example_spectrograms"] = torch.randn(1, nspeech_ex, 80, 3000) * 0.5
            samples["example_wavs"] = torch.zeros(1, nspeech_ex, 16000)
            samples["example_padding_masks"] = torch.zeros(1, nspeech_ex, 16000, dtype=torch.bool)
        with torch.no_grad():
            ref.batch_counter = 1   # skip the first-batch debug logging
            sp, sa, ee, ea = ref.get_speech_embeddings(dict(samples))
            wrapped, watts = ref.custom_prompt_wrap(sp, sa, samples["prompt"], samples["num_examples"], ee, ea)
            fwd = ref.forward(dict(samples))
            gen = ref.generate_output(dict(samples))
            gen_ids = ref.llama_model.generate(inputs_embeds=wrapped, attention_mask=watts, max_new_tokens=10, num_beams=1,
                                               do_sample=False, min_length=1, pad_token_id=tok.pad_token_id,
                                               eos_token_id=tok.eos_token_id)
            # the same call with the beam knobs the reference forwards to HF (custom_salmon.py:709-714)
            gen_b3 = ref.generate_output(dict(samples, num_beams=3, length_penalty=1.0))
            ids_b3 = ref.llama_model.generate(inputs_embeds=wrapped, attention_mask=watts, max_new_tokens=10, num_beams=3,
                                              do_sample=False, min_length=1, length_penalty=1.0, pad_token_id=tok.pad_token_id,
                                              eos_token_id=tok.eos_token_id)
        arrs = dict(wrapped=wrapped[0], logits_tail=fwd["logits"][0, -12:], labels=fwd["labels"][0], loss=fwd["loss"],
                    gen_ids=gen_ids[0], gen_ids_beams3=ids_b3[0])
        if sp is not None:
            arrs["speech"] = sp[0]
        if ee is not None:
            arrs["examples"] = torch.stack(ee[0])
        cases[case] = {"prompt": prompt, "completion": "positive", "generated_text": gen[0], "generated_text_beams3": gen_b3[0],
                       "num_examples": int(samples["num_examples"][0]), "S": int(wrapped.shape[1])}
        save(f"glue_{case}.npz", **arrs)
    # SQA: two audios (question + document) per query and per speech exemplar (custom_salmon.py:135-149,206-241,383-404,444-488)
    sqa_t = ref_cfg(RefDT.SQA).prompt_template
    sqa_ex = [{"question": f"what about item {i}", "document": f"the document number {i} says things", "completion": f"{i}.5 {i + 2}.25"}
              for i in range(2)]
    for case, (mode, few) in {"sqa_speech_text_ex": ("speech_only", "text"), "sqa_speech_speech_ex": ("speech_only", "speech"),
                              "sqa_speechtext_zero": ("speech_and_text", "zero")}.items():
        exs = None if few == "zero" else sqa_ex
        prompt = rproc._format_sqa_prompt(sqa_t, "the document text to search", exs, mode, "text" if few == "zero" else few,
                                          question="where is the answer")
        E = len(sqa_ex) if few == "speech" else 0
        samples = {"prompt": [prompt], "completion": ["3.5 4.75"], "num_examples": torch.tensor([E]),
                   "question_spectrogram": torch.randn(1, 80, 3000) * 0.5, "document_spectrogram": torch.randn(1, 80, 3000) * 0.5,
                   "question_raw_wav": torch.zeros(1, 8000), "document_raw_wav": torch.zeros(1, 12000),
                   "question_padding_mask": torch.zeros(1, 8000, dtype=torch.bool),
                   "document_padding_mask": torch.zeros(1, 12000, dtype=torch.bool)}
        if E:
            samples.update({"example_question_spectrograms": torch.randn(1, E, 80, 3000) * 0.5,
                            "example_document_spectrograms": torch.randn(1, E, 80, 3000) * 0.5,
                            "example_question_wavs": torch.zeros(1, E, 8000), "example_document_wavs": torch.zeros(1, E, 8000),
                            "example_question_padding_masks": torch.zeros(1, E, 8000, dtype=torch.bool),
                            "example_document_padding_masks": torch.zeros(1, E, 8000, dtype=torch.bool)})
        with torch.no_grad():
            ref.batch_counter = 1
            sp, sa, ee, ea = ref.get_speech_embeddings(dict(samples))
            wrapped, watts = ref.custom_prompt_wrap(sp, sa, samples["prompt"], samples["num_examples"], ee, ea)
            fwd = ref.forward(dict(samples))
            gen = ref.generate_output(dict(samples))
            gen_ids = ref.llama_model.generate(inputs_embeds=wrapped, attention_mask=watts, max_new_tokens=10, num_beams=1,
                                               do_sample=False, min_length=1, pad_token_id=tok.pad_token_id,
                                               eos_token_id=tok.eos_token_id)
        arrs = dict(wrapped=wrapped[0], logits_tail=fwd["logits"][0, -12:], labels=fwd["labels"][0], loss=fwd["loss"],
                    gen_ids=gen_ids[0], speech_q=sp[0][0], speech_d=sp[1][0])
        if ee is not None:
            arrs["examples_q"] = torch.stack([e[0] for e in ee[0]])
            arrs["examples_d"] = torch.stack([e[1] for e in ee[0]])
        cases[case] = {"prompt": prompt, "completion": "3.5 4.75", "generated_text": gen[0], "num_examples": E,
                       "S": int(wrapped.shape[1]), "sqa": True}
        save(f"glue_{case}.npz", **arrs)
    save("glue_llama.npz", **{"w:" + k: v for k, v in llama.state_dict().items()})
    with open(os.path.join(HERE, "glue_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    # prompt formatter goldens for all three tasks x modes (SURVEY.md A11 sizes)
    fmt = {}
    for dt in (RefDT.VOXCELEB, RefDT.HVB, RefDT.VOXPOPULI):
        t = ref_cfg(dt).prompt_template
        for mode in ("speech_only", "text_only", "speech_and_text"):
            for few in ("text", "speech"):
                fmt[f"{dt.value}|{mode}|{few}"] = rproc._format_default_prompt(t, "query text", ex[:3], mode, few)
        fmt[f"{dt.value}|speech_only|zero"] = rproc._format_default_prompt(t, "query text", None, "speech_only", "text")
    for mode in ("speech_only", "text_only", "speech_and_text"):
        for few in ("text", "speech"):
            fmt[f"sqa|{mode}|{few}"] = rproc._format_sqa_prompt(sqa_t, "document text", sqa_ex, mode, few, question="the question")
    fmt["sqa|speech_only|zero"] = rproc._format_sqa_prompt(sqa_t, "document text", None, "speech_only", "text", question="the question")
    with open(os.path.join(HERE, "format_prompt.json"), "w") as f:
        json.dump(fmt, f, indent=1)


def g7_qwen2_audio():
    """HF Qwen2AudioForConditionalGeneration (tiny dims, seeded weights): audio tower + masked scatter + Qwen2 LM."""
    from transformers import Qwen2AudioConfig, Qwen2AudioEncoderConfig, Qwen2AudioForConditionalGeneration, Qwen2Config
    torch.manual_seed(7)
    acfg = Qwen2AudioEncoderConfig(num_mel_bins=128, d_model=32, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=64,
                                   max_source_positions=1500)
    tcfg = Qwen2Config(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                       num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=4096,
                       tie_word_embeddings=False)
    cfg = Qwen2AudioConfig(audio_config=acfg, text_config=tcfg, audio_token_index=298)
    m = Qwen2AudioForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        from icl_speech_text_llm_amd.runtime.synth import whisper_sinusoids
        for n, p in m.named_parameters():
            if "embed_positions" in n:      # a plain loaded table in Qwen2AudioEncoder: set it to the closed form so the
                p.copy_(whisper_sinusoids(1500, 32))   # fixture need not store 1500x32 floats
                continue
            p.copy_(torch.randn_like(p) * (0.2 if p.dim() > 1 else 0.2) + (1.0 if ("norm" in n and n.endswith("weight")) else 0.0))
        c, t = torch.arange(128.0)[:, None], torch.arange(3000.0)[None, :]
        feats = torch.stack([0.5 * torch.sin(0.01 * (c + 1.0) * t + c), 0.4 * torch.cos(0.013 * (c + 2.0) * t)])
        mel_lens = [3000, 1234]
        fmask = torch.zeros(2, 3000, dtype=torch.long)
        for i, L in enumerate(mel_lens):
            fmask[i, :L] = 1
        outl = [((L - 1) // 2 + 1 - 2) // 2 + 1 for L in mel_lens]
        g = torch.Generator().manual_seed(8)
        txt = lambda k: torch.randint(3, 290, (k,), generator=g).tolist()
        ids = txt(10) + [298] * outl[0] + txt(5) + [298] * outl[1] + txt(6)
        input_ids = torch.tensor([ids])
        out = m(input_ids=input_ids, attention_mask=torch.ones_like(input_ids), input_features=feats, feature_attention_mask=fmask)
        logits = out.logits[0]
    sd = {k: v for k, v in m.state_dict().items() if "embed_positions" not in k}
    save("qwen2_audio_tiny.npz", input_ids=input_ids[0], mel_lens=mel_lens, logits_tail=logits[-24:], logits_strided=logits[::97],
         **{"w:" + k: v for k, v in sd.items()})


def g8_clean_prediction():
    sys.path.insert(0, REF)
    from utils.evaluation_utils import clean_prediction as ref_clean
    from data.master_config import DatasetType as RefDT
    raw = ["positive", " Positive.", "The sentiment is negative because", "neutral\nOutput: positive", "Postive", "", "  ",
           "acknowledge, thanks", "thanks,, statement_close,", ",question_check, foo", "Answer: statement_general (maybe)",
           "none", "None", "person, place", "place, person, xyz", "law\\n", "org;quant", "when , when", "ORG, NORP,",
           "negative positive", "it's neutral!", "123", "\\positive", "statement_open,\nthanks"]
    table = []
    for dt in (RefDT.VOXCELEB, RefDT.HVB, RefDT.VOXPOPULI):
        for r in raw:
            table.append({"dataset_type": dt.value, "raw": r, "cleaned": ref_clean(r, dt)})
    more = raw + ["alpha", "Beta.", "it is gamma", "alpha, beta", "joy", "Sadness!", "12.5 17.25", "3 4 5", "1.234 9",
                  "PER: 1.5 2.25; ORG: 3 4.125", "per: a b; LOC: 0.5 1", "LAW 1 2", "omega, psi", "phi"]
    for dt in (RefDT.VOXCELEB_GREEK, RefDT.HVB_GREEK, RefDT.VOXPOPULI_GREEK, RefDT.MELD_EMOTION, RefDT.MELD_EMOTION_GREEK,
               RefDT.VOXCELEB_SWAP, RefDT.HVB_SWAP, RefDT.SQA, RefDT.VOXPOPULI_NEL, RefDT.VP_NEL, RefDT.MELD, None):
        for r in more:
            table.append({"dataset_type": dt.value if dt else None, "raw": r, "cleaned": ref_clean(r, dt)})
    with open(os.path.join(HERE, "clean_prediction.json"), "w") as f:
        json.dump(table, f, indent=1)
    print(f"clean_prediction.json: {len(table)} rows")


def g9_metrics():
    """Seeded prediction tables scored by the reference's own evaluate_predictions / evaluate_vp_nel."""
    sys.path.insert(0, REF)
    import pandas as pd
    from utils.evaluation_utils import evaluate_predictions as ref_eval, evaluate_vp_nel as ref_nel
    from data.master_config import DatasetType as RefDT, get_dataset_config as ref_cfg, get_swap_config as ref_swap
    rng = np.random.default_rng(77)
    cases = []

    def noisy_single(label, labels):
        r = rng.random()
        if r < 0.45:
            return rng.choice([label, label.capitalize() + ".", f"The answer is {label}", f" {label}\nOutput: x"])
        if r < 0.8:
            return str(rng.choice(labels))
        return str(rng.choice(["unknown", "", "maybe", "123", "Postive"]))

    def noisy_multi(gold, labels):
        r = rng.random()
        if r < 0.35:
            return ", ".join(gold)
        if r < 0.7:
            k = int(rng.integers(1, 4))
            return ", ".join(rng.choice(labels, size=k, replace=False).tolist())
        if r < 0.85:
            return ",".join(gold[:1]) + ", foo (partial"
        return str(rng.choice(["none", "nothing here", "", "xyz, abc"]))

    singles = [RefDT.VOXCELEB, RefDT.VOXCELEB_GREEK, RefDT.VOXCELEB_SWAP, RefDT.MELD_EMOTION, RefDT.MELD_EMOTION_GREEK,
               RefDT.MELD, RefDT.MELD_GREEK]
    multis = [RefDT.HVB, RefDT.HVB_GREEK, RefDT.HVB_SWAP, RefDT.VOXPOPULI, RefDT.VOXPOPULI_GREEK, RefDT.VOXPOPULI_SWAP]
    for dt in singles + multis:
        cfg = ref_swap(dt) if dt in (RefDT.VOXCELEB_SWAP, RefDT.HVB_SWAP, RefDT.VOXPOPULI_SWAP) else ref_cfg(dt)
        labels = [l.lower() for l in cfg.valid_labels]
        for n in (1, 7, 48):
            preds = []
            for i in range(n):
                if dt in singles:
                    gold = str(rng.choice(labels + ["other"])) if rng.random() < 0.1 else str(rng.choice(labels))
                    preds.append({"text": f"t{i}", "true_label": gold, "predicted_label": noisy_single(gold, labels)})
                else:
                    k = int(rng.integers(1, 4))
                    gold = rng.choice(labels, size=k, replace=False).tolist()
                    if "VOXPOPULI" in dt.name and rng.random() < 0.2:
                        gold = ["none"]
                    if rng.random() < 0.05:
                        gold = ["bogus"]
                    preds.append({"text": f"t{i}", "true_label": ", ".join(gold), "predicted_label": noisy_multi(gold, labels)})
            cases.append({"dataset_type": dt.value, "predictions": preds,
                          "metrics": ref_eval([dict(p) for p in preds], dt)})
    for dt in (RefDT.MELD_EMOTION_SWAP, RefDT.VP_NEL, RefDT.VOXPOPULI_NEL):     # the reference's dead ends, kept as they are
        preds = [{"text": "t", "true_label": "joy", "predicted_label": "joy"}]
        cases.append({"dataset_type": dt.value, "predictions": preds, "metrics": ref_eval([dict(p) for p in preds], dt)})
    cases.append({"dataset_type": "voxceleb", "predictions": [], "metrics": ref_eval([], RefDT.VOXCELEB)})
    nel = []
    types = ["per", "org", "loc", "law"]
    for n in (1, 12):
        gt, pr = [], []
        for i in range(n):
            spans = [(str(rng.choice(types)), float(np.round(rng.uniform(0, 10), 2))) for _ in range(int(rng.integers(0, 4)))]
            g = "; ".join(f"{t.upper()}: {s} {np.round(s + rng.uniform(0.2, 2), 2)}" for t, s in spans)
            p = "; ".join(f"{(t if rng.random() < 0.8 else 'org').upper()}: {np.round(s + rng.normal(0, 0.15), 2)} "
                          f"{np.round(s + rng.uniform(0.2, 2), 2)}" for t, s in spans if rng.random() < 0.85)
            if rng.random() < 0.15:
                p += "; junk span"
            gt.append(g); pr.append(p)
        nel.append({"gt": gt, "pd": pr, "metrics": ref_nel(pd.DataFrame({"gt": gt, "pd": pr}), None)})
    with open(os.path.join(HERE, "metrics.json"), "w") as f:
        json.dump({"cases": cases, "vp_nel": nel}, f, indent=0, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    print(f"metrics.json: {len(cases)} + {len(nel)} cases, {os.path.getsize(os.path.join(HERE, 'metrics.json')) / 1024:.1f} KiB")


def g11_sampling():
    """The distribution HF's sample mode draws from, for the knobs the reference passes (custom_salmon.py:705-721)."""
    from transformers.generation.logits_process import (LogitsProcessorList, RepetitionPenaltyLogitsProcessor,
                                                        TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)
    g = torch.Generator().manual_seed(21)
    out = {}
    cases = [(0.8, 50, 0.9, 1.0), (0.8, 50, 0.9, 1.3), (1.0, 5, 0.5, 1.0), (0.3, 50, 0.95, 2.0), (1.5, 200, 0.99, 1.1),
             (0.8, 1, 0.9, 1.2), (0.7, 20, 1.0, 1.0)]
    for i, (temp, k, p, pen) in enumerate(cases):
        V = 777
        logits = torch.randn(3, V, generator=g) * 3.0
        logits[1, 100:110] = logits[1].max() + 0.25           # a tie at the top-k boundary and at the top
        prev = torch.randint(0, V, (3, 6), generator=g)
        prev[2, :3] = logits[2].topk(3).indices               # penalised tokens inside the nucleus
        procs = LogitsProcessorList()
        if pen != 1.0:
            procs.append(RepetitionPenaltyLogitsProcessor(penalty=pen))
        if temp != 1.0:
            procs.append(TemperatureLogitsWarper(temp))
        procs.append(TopKLogitsWarper(top_k=k, min_tokens_to_keep=1))
        if p < 1.0:
            procs.append(TopPLogitsWarper(top_p=p, min_tokens_to_keep=1))
        probs = torch.softmax(procs(prev, logits.clone()), dim=-1)
        out[f"logits_{i}"], out[f"prev_{i}"], out[f"probs_{i}"] = logits, prev, probs
        out[f"knobs_{i}"] = np.array([temp, k, p, pen], dtype=np.float64)
    save("sampling.npz", **out)


class ConversationDump:
    """Stand-in for the HF AutoProcessor inside the reference's QwenProcessor: renders the conversation structure verbatim."""

    def apply_chat_template(self, conversation, add_generation_prompt=True, tokenize=False):
        return json.dumps({"conversation": conversation, "add_generation_prompt": add_generation_prompt}, sort_keys=True)


def g12_qwen_prompts():
    """Conversation assembly of the reference's QwenProcessor (default + SQA) for every input / few-shot mode."""
    sys.path.insert(0, REF)
    from data.model_processors import QwenProcessor as RefQwen
    from data.master_config import DatasetType as RefDT
    proc = RefQwen(ConversationDump())
    ex = [{"text": f"example sentence {i}", "label": ["positive", "negative"][i % 2]} for i in range(3)]
    sqa_ex = [{"question": f"what about item {i}", "document": f"document {i} says things", "completion": f"{i}.5 {i + 2}.25",
               "answer": f"ans{i}"} for i in range(2)]
    out = {}
    for mode in ("speech_only", "text_only", "speech_and_text"):
        for few, exs in (("text", ex), ("speech", ex), ("zero", None)):
            fm = "text" if few == "zero" else few
            out[f"default|{mode}|{few}"] = proc.format_prompt("SYSTEM TEMPLATE", "query text", exs, mode, fm, RefDT.HVB)
            out[f"sqa|{mode}|{few}"] = proc.format_prompt("SYSTEM TEMPLATE", "document text", None if few == "zero" else sqa_ex,
                                                          mode, fm, RefDT.SQA, question="the question")
    with open(os.path.join(HERE, "qwen_prompts.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(f"qwen_prompts.json: {len(out)} prompts")


class FakeClock:
    """Deterministic stand-in for time.time(): advances by a fixed pattern per call."""

    def __init__(self):
        self.t, self.n = 1000.0, 0

    def __call__(self):
        self.n += 1
        self.t += 0.125 + 0.03125 * (self.n % 5)
        return self.t


PERF_UPDATES = [(0.5, 16, None, None), (0.25, 16, 1.5, 480), (0.75, 8, 0.5, 200), (0.125, 1, None, 31), (1.0, 16, 2.0, None),
                (0.3, 16, None, 512), (0.2, 3, 0.25, 90)]


def g13_performance_tracker():
    """The reference's PerformanceTracker under a fake clock: get_summary() and the periodic log lines."""
    sys.path.insert(0, REF)
    import logging
    import utils.performance_utils as ref
    out = {}
    for interval in (3, 100):
        clock = FakeClock()
        ref.time.time, lines = clock, []

        class L:
            def info(self, msg):
                lines.append(str(msg))
        try:
            tr = ref.PerformanceTracker(log_interval=interval, logger=L())
            for st, bs, loss, tok in PERF_UPDATES:
                tr.update(st, bs, loss=loss, token_count=tok)
            summary = tr.get_summary()
            tr.log_summary()
        finally:
            import time as _t
            ref.time.time = _t.time.__self__.time if hasattr(_t.time, "__self__") else _t.time
        out[str(interval)] = {"summary": summary, "lines": lines}
    with open(os.path.join(HERE, "performance_tracker.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("performance_tracker.json:", {k: len(v["lines"]) for k, v in out.items()})


DATASET_CASES = [   # (tasks, input_mode, fewshot_mode, num_examples, balance, interleave)
    (["voxceleb"], "speech_only", "text", 5, False, False),
    (["voxceleb"], "speech_only", "speech", 3, False, False),
    (["voxceleb"], "text_only", "text", 2, False, False),
    (["voxceleb"], "speech_and_text", "text", 0, False, False),
    (["voxceleb_greek"], "speech_only", "text", 5, False, False),
    (["voxceleb_swap"], "speech_only", "text", 4, False, False),
    (["hvb"], "speech_only", "text", 5, False, False),
    (["hvb_greek"], "text_only", "text", 3, False, False),
    (["hvb_swap"], "speech_only", "speech", 2, False, False),
    (["voxpopuli"], "speech_only", "text", 5, False, False),
    (["voxpopuli_greek"], "speech_and_text", "text", 3, False, False),
    (["voxpopuli_swap"], "speech_only", "text", 3, False, False),
    (["meld"], "speech_only", "text", 3, False, False),
    (["meld_greek"], "speech_only", "speech", 2, False, False),
    (["meld_emotion"], "speech_only", "text", 4, False, False),
    (["meld_emotion_greek"], "text_only", "text", 4, False, False),
    (["meld_emotion_swap"], "speech_only", "text", 2, False, False),
    (["sqa"], "speech_only", "text", 2, False, False),
    (["sqa"], "speech_only", "speech", 2, False, False),
    (["sqa"], "text_only", "text", 1, False, False),
    (["sqa"], "speech_and_text", "text", 0, False, False),
    (["vp_nel"], "speech_only", "text", 3, False, False),
    (["voxceleb", "hvb", "voxpopuli"], "speech_only", "text", 5, False, False),
    (["voxceleb", "hvb", "voxpopuli"], "speech_only", "text", 2, False, True),
    (["hvb", "voxceleb"], "text_only", "text", 1, True, True),
]
DATASET_SIZES = dict(n_items=4, n_lookup=5, n_fewshot=5, audio_seconds=(0.05, 0.15), seed=11)


def g10_dataset_items():
    """Items and batches of the reference's own dataset pipeline (DatasetFactory → InferenceDataset / MultiTaskInferenceDataset →
    SalmonProcessor) over seeded on-disk datasets written by this repo's ``write_synthetic_hf_datasets``."""
    import random
    import shutil
    import tempfile
    sys.path.insert(0, REF)
    from transformers import WhisperFeatureExtractor
    from icl_speech_text_llm_amd.data.synthetic_dataset import write_synthetic_hf_datasets
    from icl_speech_text_llm_amd.data.task_configs import DatasetType as MyDT
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    import data.master_config as mc
    import data.voxceleb_config as c1, data.hvb_config as c2, data.voxpopuli_config as c3, data.meld_emotion_config as c4
    from data.dataset_factory import DatasetFactory
    from data.model_processors import SalmonProcessor
    from utils.data_utils import load_dataset, clear_dataset_cache
    root = tempfile.mkdtemp(prefix="icl_golden_ds_")
    try:
        all_types = sorted({t for case in DATASET_CASES for t in case[0]})
        write_synthetic_hf_datasets(root, [MyDT(t) for t in all_types], **DATASET_SIZES)
        seen = set()
        for cfg in (list(mc.DATASET_CONFIGS.values()) + c1.VOXCELEB_SWAP_CONFIGS + c2.HVB_SWAP_CONFIGS + c3.VOXPOPULI_SWAP_CONFIGS
                    + c4.MELD_EMOTION_SWAP_CONFIGS):
            for d in (cfg.paths, cfg.audio_lookup_paths):
                if d is not None and id(d) not in seen:
                    seen.add(id(d))
                    for k in list(d):
                        d[k] = os.path.join(root, os.path.basename(d[k].rstrip("/")))
        proc = SalmonProcessor.__new__(SalmonProcessor)
        proc.tokenizer, proc.max_length, proc.batch_counter = ByteTokenizer(260), 128, 0
        proc.processor = WhisperFeatureExtractor()
        out = []
        for tasks, input_mode, fewshot_mode, k, balance, interleave in DATASET_CASES:
            clear_dataset_cache()
            dts = [mc.DatasetType(t) for t in tasks]
            random.seed(5)
            np.random.seed(6)
            rows = {dt: load_dataset(dt, split="test") for dt in dts}
            ds = DatasetFactory.create_dataset(dataset_type=dts, dataset=rows, processor=proc, is_training=False,
                                               input_mode=input_mode, fewshot_mode=fewshot_mode, num_examples=k,
                                               random_examples=False, model_type="salmonn", randomize_swap=False,
                                               balance_datasets=balance, interleave=interleave)
            items = [ds[i] for i in range(len(ds))]

            def L(x):
                return None if x is None else int(len(x))

            rec = []
            for it in items:
                r = {"prompt": it["prompt"], "completion": it["completion"], "text": it["text"],
                     "dataset_type": it["dataset_type"].value, "num_examples": int(it["num_examples"])}
                if "question" in it:
                    r.update({"question": it["question"], "unique_id": it["unique_id"],
                              "question_wav_length": int(it["question_wav_length"]),
                              "document_wav_length": int(it["document_wav_length"]),
                              "example_lengths": [[int(e["question"]["wav_length"]), int(e["document"]["wav_length"])]
                                                  for e in it["examples_speech"]],
                              "wav_sum": None if it["question_raw_wav"] is None else float(it["question_raw_wav"].double().sum()
                                                                                           + it["document_raw_wav"].double().sum())})
                else:
                    r.update({"wav_length": int(it["wav_length"]),
                              "example_lengths": [int(e["wav_length"]) for e in it["examples_speech"]],
                              "wav_sum": None if it["raw_wav"] is None else float(it["raw_wav"].double().sum())})
                rec.append(r)
            batch = proc.collate_batch(items[:3])
            shapes = {key: (list(v.shape) if hasattr(v, "shape") else len(v)) for key, v in batch.items()
                      if "spectrogram" not in key}
            out.append({"tasks": tasks, "input_mode": input_mode, "fewshot_mode": fewshot_mode, "num_examples": k,
                        "balance": balance, "interleave": interleave, "len": len(ds), "items": rec, "batch3": shapes})
        with open(os.path.join(HERE, "dataset_items.json"), "w") as f:
            json.dump({"sizes": {k: (list(v) if isinstance(v, tuple) else v) for k, v in DATASET_SIZES.items()}, "cases": out}, f, indent=0)
        print(f"dataset_items.json: {len(out)} cases, {os.path.getsize(os.path.join(HERE, 'dataset_items.json')) / 1024:.1f} KiB")
    finally:
        shutil.rmtree(root, ignore_errors=True)


def g14_reference_glue_sentencepiece():
    """The reference's CustomSALMONN glue again, but with a REAL sentencepiece Llama tokenizer (the reference takes
    `LlamaTokenizer.from_pretrained(llama_path, use_fast=False)` + a `[PAD]` token from the SALMONN object, custom_salmon.py:109):
    word-boundary pieces, the dummy-prefix space in front of every separately tokenised prompt part, byte fallback for '\n'.
    No Llama tokenizer files are reachable offline, so a 400-piece BPE model is trained here with sentencepiece on the task
    prompts (Llama's own trainer settings: bpe, byte_fallback, identity normalisation, dummy prefix) and committed as the
    fixture tests/golden/llama_spm/ ; what is pinned is the reference's splitting / per-part tokenisation / interleave / label
    logic under such a tokenizer, not a vocabulary."""
    import sentencepiece as spm
    from transformers import AutoTokenizer
    sys.path.insert(0, REF)
    from data.model_processors import SalmonProcessor as RefProcessor
    from data.master_config import DatasetType as RefDT, get_dataset_config as ref_cfg
    rproc = RefProcessor.__new__(RefProcessor)
    tmpl = ref_cfg(RefDT.VOXCELEB).prompt_template
    ex = [{"text": f"example sentence number {i} about things", "label": ["positive", "negative", "neutral"][i % 3]} for i in range(5)]
    d = os.path.join(HERE, "llama_spm")
    os.makedirs(d, exist_ok=True)
    corpus = os.path.join(d, "_corpus.txt")
    lines = [rproc._format_default_prompt(tmpl, "the query sentence to classify", ex, m, f)
             for m in ("speech_only", "text_only", "speech_and_text") for f in ("text", "speech")]
    with open(corpus, "w") as f:
        f.write("\n".join(l.replace("\n", " ") for l in lines * 20))
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(d, "tokenizer"), vocab_size=400, model_type="bpe",
                                   character_coverage=1.0, byte_fallback=True, unk_id=0, bos_id=1, eos_id=2, pad_id=-1,
                                   normalization_rule_name="identity", add_dummy_prefix=True, minloglevel=2)
    os.remove(corpus)
    os.remove(os.path.join(d, "tokenizer.vocab"))
    with open(os.path.join(d, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "LlamaTokenizer", "legacy": True, "add_bos_token": True, "add_eos_token": False}, f)
    tok = AutoTokenizer.from_pretrained(d, use_fast=False)
    tok.add_special_tokens({"pad_token": "[PAD]"})
    tok.padding_side = "right"
    assert len(tok) == 401
    llama = _tiny_llama(seed=14, vocab=401)
    H = 64
    torch.manual_seed(15)
    proj = torch.randn(34, H) * 0.3

    class StubSALMONN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.llama_model = llama
            self.llama_tokenizer = tok

        @classmethod
        def from_config(cls, cfg):
            return cls()

        def encode_speech(self, spectrogram=None, raw_wav=None, audio_padding_mask=None):
            B = spectrogram.shape[0]
            feat = spectrogram.float().mean(dim=1)[:, :88 * 34].reshape(B, 88, 34)
            return torch.tanh(feat @ proj), torch.ones(B, 88, dtype=torch.long)

    for name in ("SALMONN", "SALMONN.models", "SALMONN.models.salmonn_org"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["SALMONN.models.salmonn_org"].SALMONN = StubSALMONN
    sys.modules.pop("models.custom_salmon", None)
    from models.custom_salmon import CustomSALMONN as RefSALMONN
    ref = RefSALMONN(lora=False, device=torch.device("cpu"))
    torch.manual_seed(16)
    cases = {}
    for case, (mode, few, nspeech_ex) in {"text_only": ("text_only", "text", 0), "speech_text_ex": ("speech_only", "text", 0),
                                           "speech_speech_ex": ("speech_only", "speech", 2)}.items():
        exs = ex[:nspeech_ex] if few == "speech" else ex
        prompt = rproc._format_default_prompt(tmpl, "the query sentence to classify", exs, mode, few)
        samples = {"prompt": [prompt], "completion": ["positive"], "num_examples": torch.tensor([len(exs) if few == "speech" else 0])}
        if mode != "text_only":
            samples["spectrogram"] = torch.randn(1, 80, 3000) * 0.5
            samples["raw_wav"] = torch.zeros(1, 16000)
            samples["padding_mask"] = torch.zeros(1, 16000, dtype=torch.bool)
        if few == "speech":
            samples["example_spectrograms"] = torch.randn(1, nspeech_ex, 80, 3000) * 0.5
            samples["example_wavs"] = torch.zeros(1, nspeech_ex, 16000)
            samples["example_padding_masks"] = torch.zeros(1, nspeech_ex, 16000, dtype=torch.bool)
        with torch.no_grad():
            ref.batch_counter = 1
            sp, sa, ee, ea = ref.get_speech_embeddings(dict(samples))
            wrapped, watts = ref.custom_prompt_wrap(sp, sa, samples["prompt"], samples["num_examples"], ee, ea)
            fwd = ref.forward(dict(samples))
            gen = ref.generate_output(dict(samples))
            gen_ids = ref.llama_model.generate(inputs_embeds=wrapped, attention_mask=watts, max_new_tokens=10, num_beams=1,
                                               do_sample=False, min_length=1, pad_token_id=tok.pad_token_id,
                                               eos_token_id=tok.eos_token_id)
        arrs = dict(wrapped=wrapped[0], logits_tail=fwd["logits"][0, -12:], labels=fwd["labels"][0], loss=fwd["loss"],
                    gen_ids=gen_ids[0])
        if sp is not None:
            arrs["speech"] = sp[0]
        if ee is not None:
            arrs["examples"] = torch.stack(ee[0])
        cases[case] = {"prompt": prompt, "completion": "positive", "generated_text": gen[0],
                       "num_examples": int(samples["num_examples"][0]), "S": int(wrapped.shape[1])}
        save(f"glue_spm_{case}.npz", **arrs)
    save("glue_spm_llama.npz", **{"w:" + k: v for k, v in llama.state_dict().items()})
    with open(os.path.join(HERE, "glue_spm_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)


def _signature_table(fn):
    """[(name, kind, default repr | None)] of a callable's parameters, `self` / `cls` dropped."""
    import inspect
    out = []
    for name, prm in inspect.signature(fn).parameters.items():
        if name in ("self", "cls"):
            continue
        out.append([name, prm.kind.name, None if prm.default is inspect.Parameter.empty else repr(prm.default)])
    return out


def g16_beam_search():
    """HF beam search with inputs_embeds only, for the knobs the reference forwards (custom_salmon.py:709-714: num_beams,
    length_penalty, min_length=1; early_stopping stays at its default False) on the miniature Llama of ``llama_tiny.npz``."""
    m = _tiny_llama()
    torch.manual_seed(3)
    emb = torch.randn(2, 23, 64) * 0.5            # the same prompts as llama_tiny.npz (same seed, same order of draws)
    att = torch.ones(2, 23, dtype=torch.long)
    out = {}

    def run(name, e, num_beams, lp, eos, max_new=10):
        with torch.no_grad():
            r = m.generate(inputs_embeds=e, attention_mask=att[: e.shape[0]], max_new_tokens=max_new, do_sample=False,
                           num_beams=num_beams, min_length=1, length_penalty=lp, pad_token_id=259, eos_token_id=eos,
                           return_dict_in_generate=True, output_scores=True)
        out[name + "_seq"], out[name + "_score"] = r.sequences, r.sequences_scores
        out[name + "_knobs"] = np.array([num_beams, lp, -1 if eos is None else eos, max_new], dtype=np.float64)
        return r.sequences

    free = run("free4", emb, 4, 1.0, None)
    run("free3_lp2", emb, 3, 2.0, None)
    run("free2_lp0", emb, 2, 0.0, None)
    # an EOS the best hypothesis of row 0 meets at step 3 / at step 0; one both rows meet early; one single row
    run("eos_mid4", emb, 4, 1.0, int(free[0, 3]))
    run("eos_first4", emb, 4, 1.0, int(free[0, 0]))
    run("eos_mid4_lpneg", emb, 4, -1.0, int(free[1, 2]))
    run("eos_mid5_lp2", emb, 5, 2.0, int(free[1, 4]))
    run("one_row_eos3", emb[1:], 3, 1.0, int(free[1, 1]))
    run("short3", emb, 3, 1.0, int(free[0, 1]), max_new=3)
    # HF's list form of eos_token_id (Qwen2-Audio's generation_config names two): 3K continuations are kept per row
    two = [int(free[0, 3]), int(free[1, 2])]
    for name, nb, lp in (("two_eos4", 4, 1.0), ("two_eos2_lpneg", 2, -1.0), ("two_eos3_lp2", 3, 2.0)):
        with torch.no_grad():
            r = m.generate(inputs_embeds=emb, attention_mask=att, max_new_tokens=10, do_sample=False, num_beams=nb, min_length=1,
                           length_penalty=lp, pad_token_id=259, eos_token_id=two, return_dict_in_generate=True, output_scores=True)
        out[name + "_seq"], out[name + "_score"] = r.sequences, r.sequences_scores
        out[name + "_knobs"] = np.array([nb, lp, two[0], 10, two[1]], dtype=np.float64)
    with torch.no_grad():
        g2 = m.generate(inputs_embeds=emb, attention_mask=att, max_new_tokens=10, do_sample=False, num_beams=1, min_length=1,
                        pad_token_id=259, eos_token_id=[int(free[0, 3]), 221])      # 221: row 1's first greedy token
    out["two_eos_greedy_seq"], out["two_eos_greedy_eos"] = g2, np.array([int(free[0, 3]), 221])
    # beams under a repetition penalty (custom_salmon.py:713 forwards it next to num_beams): the processor acts on log-probabilities
    for name, nb, lp, pen, eos in (("rep3", 3, 1.0, 1.5, None), ("rep4_eos", 4, 1.0, 1.3, int(free[0, 3])), ("rep2_lp2", 2, 2.0, 2.0, None)):
        with torch.no_grad():
            r = m.generate(inputs_embeds=emb, attention_mask=att, max_new_tokens=10, do_sample=False, num_beams=nb, min_length=1,
                           length_penalty=lp, repetition_penalty=pen, pad_token_id=259, eos_token_id=eos,
                           return_dict_in_generate=True, output_scores=True)
        out[name + "_seq"], out[name + "_score"] = r.sequences, r.sequences_scores
        out[name + "_knobs"] = np.array([nb, lp, -1 if eos is None else eos, 10, -1, pen], dtype=np.float64)
    save("beam_tiny.npz", **out)


def g17_multi_task_wrapper():
    """The reference's MultiTaskModel wrapper logic (models/multi_task_model.py:52-149), imported unmodified over the stub SALMONN, with
    the inner model's forward / generate_output replaced by recorders: task switching, prompt-template substitution, per-task knobs."""
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    llama, tok = _tiny_llama(seed=4), ByteTokenizer(260)

    class StubSALMONN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.llama_model, self.llama_tokenizer = llama, tok

        @classmethod
        def from_config(cls, cfg):
            return cls()

    for name in ("SALMONN", "SALMONN.models", "SALMONN.models.salmonn_org"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["SALMONN.models.salmonn_org"].SALMONN = StubSALMONN
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.multi_task_model import MultiTaskModel as RefMT
    tasks = {"sentiment": {"prompt_template": "SENTIMENT> ", "max_new_tokens": 4, "num_beams": 3},
             "intent": {"prompt_template": "INTENT> ", "do_sample": True, "temperature": 0.5},
             "plain": {"max_new_tokens": 7}}
    mt = RefMT("salmonn", task_configs=tasks, default_task="intent", lora=False, device=torch.device("cpu"))
    mt.model.prompt_template = "BASE: "
    mt.model.batch_counter = 1
    seen = {}
    mt.model.forward = lambda samples: (seen.__setitem__("fwd", {k: (list(v) if isinstance(v, list) else v) for k, v in samples.items()}) or {"loss": 0.0})
    mt.model.generate_output = lambda samples: (seen.__setitem__("gen", dict(samples)) or ["out"] * len(samples["prompt"]))
    out = {"templates": {t: mt.get_task_prompt_template(t) for t in ("sentiment", "intent", "plain", "unknown")},
           "template_current": mt.get_task_prompt_template(), "set_task": {}}
    for t in ("sentiment", "nope", "plain"):
        out["set_task"][t] = [mt.set_task(t), mt.current_task]
    prompts = ["BASE: classify this", "BASE: and this BASE: twice", "no base here", "BASE: last"]
    r = mt.forward({"prompt": list(prompts), "task": ["sentiment", None, "intent", "plain"]})
    out["forward_with_tasks"] = {"prompts": seen["fwd"]["prompt"], "task_out": r["task"]}
    r = mt.forward({"prompt": list(prompts)})
    out["forward_without_tasks"] = {"prompts": seen["fwd"]["prompt"], "task_out": r["task"]}
    r = mt.forward({"prompt": list(prompts[:2]), "task": [None, None]})
    out["forward_all_none"] = {"prompts": seen["fwd"]["prompt"], "task_out": r["task"]}
    gen = {}
    for name, samples in (("task_sentiment", {"prompt": ["p"], "task": ["sentiment"]}),
                          ("task_intent_overrides_batch_keys", {"prompt": ["p", "q"], "task": ["intent", "sentiment"], "max_new_tokens": 99}),
                          ("task_plain", {"prompt": ["p"], "task": ["plain"]}),
                          ("task_unknown", {"prompt": ["p"], "task": ["unknown"], "temperature": 0.3}),
                          ("task_none", {"prompt": ["p"], "task": [None]}),
                          ("no_task_key", {"prompt": ["p"]})):
        res = mt.generate_output(samples)
        gen[name] = {"samples_after": {k: v for k, v in seen["gen"].items() if k != "prompt"}, "current_task": mt.current_task, "n_out": len(res)}
    out["generate_output"] = gen
    with open(os.path.join(HERE, "multi_task_wrapper.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("multi_task_wrapper.json written")


def g18_results_files_and_inference_config():
    """f1: what the reference's CLI leaves on disk — save_final_results (inference/inference.py:394-456: file names, the records as
    dumped, the metrics file; an error inside is logged, not raised) — and config/inference_config.py's get_inference_config."""
    import argparse
    import tempfile
    for name in ("SALMONN", "SALMONN.models", "SALMONN.models.salmonn_org", "peft"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["SALMONN.models.salmonn_org"].SALMONN = getattr(sys.modules["SALMONN.models.salmonn_org"], "SALMONN", object)
    for attr in ("LoraConfig", "get_peft_model", "TaskType"):
        setattr(sys.modules["peft"], attr, getattr(sys.modules["peft"], attr, object))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import inference.inference as ref_cli
    from config.inference_config import get_inference_config as ref_cfg
    from data.master_config import DatasetType as RefDT

    def rec(dt, text, true, pred):
        return {"text": text, "true_label": true, "predicted_label": pred, "dataset_type": dt}
    results = [rec("voxceleb", "a fine day", "positive", " Positive."), rec("voxceleb", "bad", "negative", "the sentiment is neutral"),
               rec("hvb", "thanks a lot", ["thanks", "statement_close"], "thanks, statement_close,"), rec("hvb", "is it?", ["question_check"], "none"),
               rec("voxpopuli", "in Paris", {"place": ["Paris"]}, "place"), rec("voxpopuli", "nothing", {}, "None")]
    out = {"save_final_results": {}}
    for name, ns in (("multi", dict(dataset_type="voxceleb-hvb-voxpopuli", output_suffix="")),
                     ("single_suffix", dict(dataset_type="voxceleb", output_suffix="v2")),
                     ("bad_dataset_type", dict(dataset_type="voxceleb-notadataset", output_suffix=""))):
        args = argparse.Namespace(run_name="run7", input_mode="speech_only", fewshot_mode="text", num_examples=5, **ns)
        with tempfile.TemporaryDirectory() as d:
            raised = None
            try:
                ref_cli.save_final_results([dict(r) for r in results], args, d)
            except Exception as e:      # the reference logs and swallows
                raised = f"{type(e).__name__}: {e}"
            files = {}
            for fn in sorted(os.listdir(d)):
                with open(os.path.join(d, fn)) as f:
                    files[fn] = json.load(f)
        out["save_final_results"][name] = {"raised": raised, "files": files}
    cfgs = {}
    for mt in ("salmonn", "qwen2"):
        for dt in (None,) + tuple(RefDT):
            cfgs[f"{mt}|{dt.value if dt else None}"] = ref_cfg(mt, dt)
    try:
        ref_cfg("nope")
        cfgs["nope"] = "no error"
    except Exception as e:
        cfgs["nope"] = f"{type(e).__name__}: {e}"
    out["get_inference_config"] = cfgs
    with open(os.path.join(HERE, "results_files.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)
    print("results_files.json:", {k: list(v) for k, v in out["save_final_results"].items()})


def g19_interactive():
    """inference/interactive_inference.py of the reference: its argparse table, and run_interactive_inference (:166-222) driven with
    the reference's own SalmonProcessor over a byte tokenizer and a recording model — what reaches generate_output."""
    import argparse
    for name in ("SALMONN", "SALMONN.models", "SALMONN.models.salmonn_org", "peft"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["SALMONN.models.salmonn_org"].SALMONN = getattr(sys.modules["SALMONN.models.salmonn_org"], "SALMONN", object)
    for attr in ("LoraConfig", "get_peft_model", "TaskType"):
        setattr(sys.modules["peft"], attr, getattr(sys.modules["peft"], attr, object))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    captured = []
    real_parse = argparse.ArgumentParser.parse_args
    argparse.ArgumentParser.parse_args = lambda self, *a, **k: captured.append(self) or argparse.Namespace()
    try:
        sys.modules.pop("inference.interactive_inference", None)
        import inference.interactive_inference as ref_ii
        ref_ii.parse_args()
    finally:
        argparse.ArgumentParser.parse_args = real_parse
    actions = []
    for a in captured[0]._actions:
        if not a.option_strings or a.dest == "help":
            continue
        default = "<cuda if available else cpu>" if a.dest == "device" else a.default
        actions.append({"flags": a.option_strings, "dest": a.dest, "action": type(a).__name__,
                        "type": getattr(a.type, "__name__", None), "default": default})
    from data.model_processors import SalmonProcessor as RefProcessor
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    proc = RefProcessor.__new__(RefProcessor)       # its __init__ downloads the Whisper feature extractor, which text-only never calls
    proc.tokenizer, proc.max_length, proc.batch_counter = ByteTokenizer(260), 128, 1
    seen = {}

    class Recorder:
        def eval(self):
            return self

        def generate_output(self, batch):
            seen["batch"] = batch
            return ["the answer"]
    args = argparse.Namespace(device="cpu", max_new_tokens=100, temperature=0.8)
    text = ref_ii.run_interactive_inference(Recorder(), proc, "What is the definition of positive?", args)
    b = seen["batch"]
    out = {"cli": actions, "returned": text,
           "batch_keys": sorted(b), "non_tensor": {k: (str(v) if k == "dataset_type" else v) for k, v in b.items() if not isinstance(v, torch.Tensor)},
           "tensor_shapes": {k: list(v.shape) for k, v in b.items() if isinstance(v, torch.Tensor)},
           "input_ids": b["input_ids"][0].tolist(), "num_examples": b["num_examples"].tolist()}
    with open(os.path.join(HERE, "interactive.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("interactive.json: batch keys", out["batch_keys"])


CLI_LOOP_PREDICTIONS = ["positive", " Negative", "neutral.", "thanks,,", "none", "Postive", "law\n", "place, x"]     # <= 9 bytes: 10 new tokens with the EOS
CLI_LOOP_SIZES = dict(n_items=5, n_lookup=5, n_fewshot=5, audio_seconds=(0.05, 0.15), seed=13)


def cli_loop_prediction(prompt: str) -> str:
    """The stand-in model of the CLI-loop golden: a fixed function of the prompt (both the reference run and ours use it)."""
    return CLI_LOOP_PREDICTIONS[len(prompt) % len(CLI_LOOP_PREDICTIONS)]


def g20_cli_loop():
    """The reference's run_inference (inference/inference.py:106-392), unmodified, over seeded on-disk datasets with a stand-in model
    (ModelFactory.create_model replaced: generate_output is `cli_loop_prediction` of each prompt, and the third batch raises): the
    records it assembles, their order, what --max_samples really limits, what a failing batch costs, and the files it writes."""
    import random
    import shutil
    import tempfile
    from unittest import mock
    for name in ("SALMONN", "SALMONN.models", "SALMONN.models.salmonn_org", "peft"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["SALMONN.models.salmonn_org"].SALMONN = getattr(sys.modules["SALMONN.models.salmonn_org"], "SALMONN", object)
    for attr in ("LoraConfig", "get_peft_model", "TaskType"):
        setattr(sys.modules["peft"], attr, getattr(sys.modules["peft"], attr, object))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from transformers import WhisperFeatureExtractor
    from icl_speech_text_llm_amd.data.synthetic_dataset import write_synthetic_hf_datasets
    from icl_speech_text_llm_amd.data.task_configs import DatasetType as MyDT
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    import data.master_config as mc
    import data.voxceleb_config as c1, data.hvb_config as c2, data.voxpopuli_config as c3, data.meld_emotion_config as c4
    import data.model_processors as rmp
    from utils.data_utils import clear_dataset_cache
    import inference.inference as ref_cli
    root = tempfile.mkdtemp(prefix="icl_golden_cli_")
    out = {"sizes": {k: (list(v) if isinstance(v, tuple) else v) for k, v in CLI_LOOP_SIZES.items()}, "predictions": CLI_LOOP_PREDICTIONS, "runs": {}}
    try:
        write_synthetic_hf_datasets(root, [MyDT("voxceleb"), MyDT("hvb"), MyDT("voxpopuli")], **CLI_LOOP_SIZES)
        seen = set()
        for cfg in (list(mc.DATASET_CONFIGS.values()) + c1.VOXCELEB_SWAP_CONFIGS + c2.HVB_SWAP_CONFIGS + c3.VOXPOPULI_SWAP_CONFIGS
                    + c4.MELD_EMOTION_SWAP_CONFIGS):
            for d in (cfg.paths, cfg.audio_lookup_paths):
                if d is not None and id(d) not in seen:
                    seen.add(id(d))
                    for k in list(d):
                        d[k] = os.path.join(root, os.path.basename(d[k].rstrip("/")))

        def proc_init(self, tokenizer, max_length=128):       # the reference's __init__ downloads the Whisper feature extractor
            self.processor, self.tokenizer, self.max_length, self.batch_counter = WhisperFeatureExtractor(), tokenizer, max_length, 1
        rmp.SalmonProcessor.__init__ = proc_init

        class StandIn:
            input_processor = None
            llama_tokenizer = ByteTokenizer(260)

            def __init__(self, fail_batch):
                self.calls, self.fail_batch = 0, fail_batch

            def to(self, *_a, **_k):
                return self

            def eval(self):
                return self

            def generate_output(self, batch):
                self.calls += 1
                if self.calls - 1 == self.fail_batch:
                    raise ValueError("injected failure")
                return [cli_loop_prediction(p) for p in batch["prompt"]]
        real_makedirs = os.makedirs
        for name, argv, fail in (
                ("multi_text_only", ["--dataset_type", "voxceleb-hvb", "--input_mode", "text_only", "--num_examples", "2", "--batch_size", "2"], 2),
                ("single_max_samples", ["--dataset_type", "voxpopuli", "--input_mode", "text_only", "--num_examples", "1", "--batch_size", "2",
                                        "--max_samples", "3"], -1),
                ("speech_batch1", ["--dataset_type", "voxceleb", "--input_mode", "speech_only", "--num_examples", "2", "--batch_size", "1",
                                   "--debug_samples", "3"], -1)):
            clear_dataset_cache()
            random.seed(5)
            np.random.seed(6)
            res_dir = tempfile.mkdtemp(prefix="icl_golden_cli_res_")
            model = StandIn(fail)
            with mock.patch.object(sys, "argv", ["inference.py", "--peft_model_path", "", "--run_name", "g20", "--device", "cpu",
                                                 "--num_workers", "0", "--split", "test"] + argv):
                args = ref_cli.parse_args()
            real_save = ref_cli.save_final_results
            with mock.patch.object(ref_cli.ModelFactory, "create_model", staticmethod(lambda **kw: model)), \
                    mock.patch.object(os, "makedirs", lambda p, *a, **k: real_makedirs(p, *a, **k) if not str(p).startswith("/data2") else None), \
                    mock.patch.object(ref_cli, "save_final_results", lambda results, a, d: real_save(results, a, res_dir)):
                ret = ref_cli.run_inference(args)
            files = {}
            for fn in sorted(os.listdir(res_dir)):
                with open(os.path.join(res_dir, fn)) as f:
                    files[fn] = json.load(f)
            shutil.rmtree(res_dir, ignore_errors=True)
            out["runs"][name] = {"argv": argv, "fail_batch": fail, "generate_calls": model.calls, "results": ret["results"],
                                 "performance_keys": sorted(ret["performance"]), "total_examples": ret["performance"].get("total_examples"),
                                 "files": files}
        # --peft_model_path: which load_state_dict receives which keys for the four checkpoint layouts (:157-177), and a missing file
        out["checkpoint_dispatch"] = {}
        layouts = {"model_state_dict": lambda sd: {"model_state_dict": sd, "epoch": 3}, "state_dict": lambda sd: {"state_dict": sd},
                   "model": lambda sd: {"model": sd}, "raw": lambda sd: sd}
        sd = {"speech_llama_proj.weight": torch.zeros(2, 2), "llama_model.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight": torch.ones(1, 2)}
        for lname, wrap in list(layouts.items()) + [("missing_file", None)]:
            clear_dataset_cache()
            random.seed(5)
            np.random.seed(6)
            res_dir = tempfile.mkdtemp(prefix="icl_golden_cli_res_")
            ck = os.path.join(res_dir, "ck.pt")
            if wrap is not None:
                torch.save(wrap(sd), ck)
            calls = []

            class Inner:
                def load_state_dict(self, state, strict=True):
                    calls.append(["model.salmonn.load_state_dict", sorted(state), strict])

            class Recording(StandIn):
                salmonn = Inner()

                def load_state_dict(self, state, strict=True):
                    calls.append(["model.load_state_dict", sorted(state), strict])
            model = Recording(-1)
            with mock.patch.object(sys, "argv", ["inference.py", "--peft_model_path", ck, "--run_name", "g20", "--device", "cpu",
                                                 "--num_workers", "0", "--split", "test", "--dataset_type", "voxceleb", "--input_mode",
                                                 "text_only", "--num_examples", "1", "--batch_size", "1", "--debug_samples", "1"]):
                args = ref_cli.parse_args()
            real_save = ref_cli.save_final_results
            error = None
            with mock.patch.object(ref_cli.ModelFactory, "create_model", staticmethod(lambda **kw: model)), \
                    mock.patch.object(os, "makedirs", lambda p, *a, **k: real_makedirs(p, *a, **k) if not str(p).startswith("/data2") else None), \
                    mock.patch.object(ref_cli, "save_final_results", lambda results, a, d: real_save(results, a, res_dir)):
                try:
                    ret = ref_cli.run_inference(args)
                except Exception as e:
                    error = f"{type(e).__name__}: {e}".replace(ck, "<ckpt>")
            shutil.rmtree(res_dir, ignore_errors=True)
            out["checkpoint_dispatch"][lname] = {"calls": calls, "error": error, "n_results": None if error else len(ret["results"])}
        with open(os.path.join(HERE, "cli_loop.json"), "w") as f:
            json.dump(out, f, indent=0, default=str)
        print("cli_loop.json:", {k: (v["generate_calls"], len(v["results"])) for k, v in out["runs"].items()})
    finally:
        shutil.rmtree(root, ignore_errors=True)


def g21_model_factory():
    """models/model_factory.py of the reference: error messages of create_model / from_config, the description records, the model
    list and clear_cache's return value (constructors are not reached by any of these calls)."""
    for name in ("SALMONN", "SALMONN.models", "SALMONN.models.salmonn_org", "peft"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["SALMONN.models.salmonn_org"].SALMONN = getattr(sys.modules["SALMONN.models.salmonn_org"], "SALMONN", object)
    for attr in ("LoraConfig", "get_peft_model", "TaskType"):
        setattr(sys.modules["peft"], attr, getattr(sys.modules["peft"], attr, object))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.model_factory import ModelFactory as RefFactory

    def err(fn):
        try:
            fn()
            return None
        except Exception as e:
            return f"{type(e).__name__}: {e}"
    out = {"create_model_unknown": err(lambda: RefFactory.create_model("Whisper")),
           "create_model_multi_without_tasks": err(lambda: RefFactory.create_model("salmonn", multi_task=True)),
           "create_model_multi_empty_tasks": err(lambda: RefFactory.create_model("QWEN2", multi_task=True, task_configs={})),
           "from_config_no_type": err(lambda: RefFactory.from_config({})),
           "from_config_multi_without_tasks": err(lambda: RefFactory.from_config({"model_type": "salmonn", "multi_task": True})),
           "from_config_unknown": err(lambda: RefFactory.from_config({"model_type": "gpt"})),
           "get_model_info": {t: RefFactory.get_model_info(t) for t in ("salmonn", "qwen2", "SALMONN")},
           "get_model_info_unknown": err(lambda: RefFactory.get_model_info("gpt")),
           "get_available_models": RefFactory.get_available_models(),
           "clear_cache": RefFactory.clear_cache(),
           "checkpoint_missing": err(lambda: RefFactory.get_model_from_checkpoint("/nonexistent/x.pt", "base", "salmonn"))}
    with open(os.path.join(HERE, "model_factory.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("model_factory.json:", {k: (v if isinstance(v, (str, int, type(None))) else "...") for k, v in out.items()})


E2E_SIZES = dict(n_items=3, n_lookup=4, n_fewshot=4, audio_seconds=(0.05, 0.1), seed=17)
E2E_ARGV = ["--dataset_type", "voxceleb-hvb", "--input_mode", "text_only", "--num_examples", "2", "--batch_size", "1"]


def g22_cli_end_to_end():
    """The reference END TO END: its CLI (run_inference) over its own CustomSALMONN (unmodified; the absent SALMONN package replaced by
    a stub that holds a two-layer HF Llama at the width of this build's `tiny` arch and the byte tokenizer) on seeded on-disk
    datasets, text_only mode — prompts, embeddings, HF generate, decoding, cleaning, scoring, files.  The Llama's seed is the
    first whose every greedy decision has a margin of more than 8x the distance between the fp32 and the bf16-rounding oracle on
    the same step, so that a bf16 implementation must reproduce every token (the test runs this build's CLI on the GPU)."""
    import random
    import shutil
    import tempfile
    from unittest import mock
    from transformers import LlamaConfig, LlamaForCausalLM
    for name in ("SALMONN", "SALMONN.models", "SALMONN.models.salmonn_org", "peft"):
        sys.modules.setdefault(name, types.ModuleType(name))
    for attr in ("LoraConfig", "get_peft_model", "TaskType"):
        setattr(sys.modules["peft"], attr, getattr(sys.modules["peft"], attr, object))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from transformers import WhisperFeatureExtractor
    from icl_speech_text_llm_amd.data.synthetic_dataset import write_synthetic_hf_datasets
    from icl_speech_text_llm_amd.data.task_configs import DatasetType as MyDT
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    from oracle import models as om
    import data.master_config as mc
    import data.voxceleb_config as c1, data.hvb_config as c2, data.voxpopuli_config as c3, data.meld_emotion_config as c4
    import data.model_processors as rmp
    from utils.data_utils import clear_dataset_cache
    tok = ByteTokenizer(260)
    holder = {}

    class StubSALMONN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.llama_model, self.llama_tokenizer = holder["llama"], tok

        @classmethod
        def from_config(cls, cfg):
            return cls()
    sys.modules["SALMONN.models.salmonn_org"].SALMONN = StubSALMONN
    for m in ("models.custom_salmon", "models.model_factory", "models.multi_task_model", "inference.inference"):
        sys.modules.pop(m, None)
    import inference.inference as ref_cli

    def proc_init(self, tokenizer, max_length=128):
        self.processor, self.tokenizer, self.max_length, self.batch_counter = WhisperFeatureExtractor(), tokenizer, max_length, 1
    rmp.SalmonProcessor.__init__ = proc_init
    root = tempfile.mkdtemp(prefix="icl_golden_e2e_")
    real_makedirs = os.makedirs
    try:
        write_synthetic_hf_datasets(root, [MyDT("voxceleb"), MyDT("hvb")], **E2E_SIZES)
        seen = set()
        for cfg in (list(mc.DATASET_CONFIGS.values()) + c1.VOXCELEB_SWAP_CONFIGS + c2.HVB_SWAP_CONFIGS + c3.VOXPOPULI_SWAP_CONFIGS
                    + c4.MELD_EMOTION_SWAP_CONFIGS):
            for d in (cfg.paths, cfg.audio_lookup_paths):
                if d is not None and id(d) not in seen:
                    seen.add(id(d))
                    for k in list(d):
                        d[k] = os.path.join(root, os.path.basename(d[k].rstrip("/")))
        chosen = None
        for seed in range(int(os.environ.get("G22_SEEDS", "200"))):
            torch.manual_seed(1000 + seed)
            cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                              vocab_size=260, rms_norm_eps=1e-5, max_position_embeddings=2048, pad_token_id=259, bos_token_id=1,
                              eos_token_id=2, tie_word_embeddings=False)
            llama = LlamaForCausalLM(cfg).eval()
            with torch.no_grad():
                for n, p in llama.named_parameters():
                    # bf16-representable weights; the layers' projections are small next to the embeddings / LM head, so that the
                    # bf16 rounding inside the layers moves the logits by far less than the typical top-1 / top-2 gap
                    std = 0.03 if "_proj" in n else 0.3
                    p.copy_((torch.randn_like(p) * std if p.dim() > 1 else 1.0 + 0.1 * torch.randn_like(p)).to(torch.bfloat16).float())
            holder["llama"] = llama
            captured = []
            real_generate = llama.generate

            def spy(*a, **k):
                out_ids = real_generate(*a, **k)
                captured.append((k["inputs_embeds"].detach().clone(), out_ids.detach().clone()))
                return out_ids
            llama.generate = spy
            clear_dataset_cache()
            random.seed(5)
            np.random.seed(6)
            res_dir = tempfile.mkdtemp(prefix="icl_golden_e2e_res_")
            with mock.patch.object(sys, "argv", ["inference.py", "--peft_model_path", "", "--run_name", "e2e", "--device", "cpu",
                                                 "--num_workers", "0", "--split", "test"] + E2E_ARGV):
                args = ref_cli.parse_args()
            real_save = ref_cli.save_final_results
            real_create = ref_cli.ModelFactory.create_model
            def create(**kw):
                model = real_create(**{**kw, "lora": False, "low_resource": False})
                # the CLI reads model.input_processor (:200), which the reference's CustomSALMONN does not define (its older
                # mlp_salmonn_old.py did: a WhisperFeatureExtractor); get_processor ignores the value for "salmonn"
                model.input_processor = None
                return model
            with mock.patch.object(ref_cli.ModelFactory, "create_model", staticmethod(create)), \
                    mock.patch.object(os, "makedirs", lambda p, *a, **k: real_makedirs(p, *a, **k) if not str(p).startswith("/data2") else None), \
                    mock.patch.object(ref_cli, "save_final_results", lambda results, a, d: real_save(results, a, res_dir)):
                ret = ref_cli.run_inference(args)
            files = {}
            for fn in sorted(os.listdir(res_dir)):
                with open(os.path.join(res_dir, fn)) as f:
                    files[fn] = json.load(f)
            shutil.rmtree(res_dir, ignore_errors=True)
            # margins: every greedy decision against the fp32 / bf16-rounding oracle distance at that step
            sd = {k: v.detach().clone() for k, v in llama.state_dict().items()}
            of, ob = om.LlamaOracle(sd, 2, 1e-5), om.LlamaOracle(sd, 2, 1e-5, rnd=om.bf16_round)
            worst = float("inf")
            for emb, ids in captured:
                ids = ids[:, : max(1, int((ids[0] != 259).sum()))]
                lf, lb = of.teacher_forced_logits(emb, ids)[0], ob.teacher_forced_logits(emb, ids)[0]
                for t in range(ids.shape[1]):
                    top2 = lf[t].topk(2)
                    if int(top2.indices[0]) != int(ids[0, t]) or int(lb[t].argmax()) != int(ids[0, t]):
                        if os.environ.get("G22_DEBUG"):
                            print("  mismatch at step", t, "hf", int(ids[0, t]), "fp32", int(top2.indices[0]), "bf16", int(lb[t].argmax()),
                                  "gap", float(top2.values[0] - top2.values[1]), "err", float((lf[t] - lb[t]).abs().max()), "emb", tuple(emb.shape), emb.dtype, "ids", ids.tolist())
                        worst = -1.0
                        break
                    worst = min(worst, float(top2.values[0] - top2.values[1]) / max(float((lf[t] - lb[t]).abs().max()), 1e-6))
                if worst < 0:
                    break
            print(f"seed {1000 + seed}: {len(ret['results'])} records, smallest margin / bf16 distance = {worst:.1f}")
            if worst > 8.0:
                chosen = (seed, llama, ret, files, worst)
                break
        assert chosen is not None, "no seed with decisive margins"
        seed, llama, ret, files, worst = chosen
        # bf16-representable values: the upper 16 bits are the whole number
        save("cli_e2e_llama.npz", **{"w16:" + k: (v.detach().contiguous().view(torch.int32) >> 16).to(torch.int16).numpy().view(np.uint16)
                                     for k, v in llama.state_dict().items()})
        with open(os.path.join(HERE, "cli_e2e.json"), "w") as f:
            json.dump({"sizes": {k: (list(v) if isinstance(v, tuple) else v) for k, v in E2E_SIZES.items()}, "argv": E2E_ARGV,
                       "llama_seed": 1000 + seed, "smallest_margin_over_bf16_distance": worst, "results": ret["results"], "files": files},
                      f, indent=0, default=str)
        print("cli_e2e.json:", len(ret["results"]), "records;", [r["predicted_label"] for r in ret["results"]])
    finally:
        shutil.rmtree(root, ignore_errors=True)


E2E_SPEECH_SIZES = dict(n_items=2, n_lookup=4, n_fewshot=4, audio_seconds=(0.4, 0.9), seed=19)
E2E_SPEECH_RUNS = {"speech_query_text_exemplars": ["--dataset_type", "voxceleb-hvb", "--input_mode", "speech_only", "--fewshot_mode", "text",
                                                   "--num_examples", "2", "--batch_size", "1"],
                   "speech_query_speech_exemplars": ["--dataset_type", "voxceleb", "--input_mode", "speech_only", "--fewshot_mode", "speech",
                                                     "--num_examples", "2", "--batch_size", "1"],
                   "sqa_two_audios_speech_exemplar": ["--dataset_type", "sqa", "--input_mode", "speech_only", "--fewshot_mode", "speech",
                                                      "--num_examples", "1", "--batch_size", "1"],
                   "speech_and_text_zero_shot": ["--dataset_type", "hvb", "--input_mode", "speech_and_text", "--fewshot_mode", "text",
                                                 "--num_examples", "0", "--batch_size", "1"]}


def g23_cli_end_to_end_speech():
    """g22 with audio: the reference's CLI over its own CustomSALMONN in speech_only mode (a speech query after two text exemplars;
    a speech query after two SPEECH exemplars).  Everything in the reference's repository runs unmodified — dataset items, the
    Whisper feature extractor, prompt split, `<Speech>` interleave of query and exemplar embeddings, HF generate, decoding, scoring —
    and the one thing it imports from the absent SALMONN package, `encode_speech`, is this repo's fp32 oracle of it (Whisper + BEATs
    + window-level Q-Former over `tests/golden/e2e_weights.py`'s miniature weights).  What the golden pins beyond the oracle is
    the integration: which audio goes where in which prompt, and that the whole chain yields these tokens."""
    import random
    import shutil
    import tempfile
    from unittest import mock
    from transformers import LlamaConfig, LlamaForCausalLM, WhisperFeatureExtractor
    for name in ("SALMONN", "SALMONN.models", "SALMONN.models.salmonn_org", "peft"):
        sys.modules.setdefault(name, types.ModuleType(name))
    for attr in ("LoraConfig", "get_peft_model", "TaskType"):
        setattr(sys.modules["peft"], attr, getattr(sys.modules["peft"], attr, object))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.path.insert(0, HERE)
    from e2e_weights import tiny_salmonn_weights
    from icl_speech_text_llm_amd.data.synthetic_dataset import write_synthetic_hf_datasets
    from icl_speech_text_llm_amd.data.task_configs import DatasetType as MyDT
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    from oracle import models as om
    import data.master_config as mc
    import data.voxceleb_config as c1, data.hvb_config as c2, data.voxpopuli_config as c3, data.meld_emotion_config as c4
    import data.model_processors as rmp
    from utils.data_utils import clear_dataset_cache
    tok = ByteTokenizer(260)
    holder = {}

    class StubSALMONN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.llama_model, self.llama_tokenizer = holder["llama"], tok

        @classmethod
        def from_config(cls, cfg):
            return cls()

        def encode_speech(self, spectrogram=None, raw_wav=None, audio_padding_mask=None):
            cfg, sd = holder["cfg"], holder["sd"]
            lens = ((~audio_padding_mask.bool()).sum(1).tolist() if audio_padding_mask is not None else [raw_wav.shape[1]] * raw_wav.shape[0])
            emb = om.salmonn_encode_speech(sd, spectrogram.float(), raw_wav.float(), lens, cfg.whisper.n_heads,
                                           beats_cfg=dict(n_heads=cfg.beats.n_heads), qformer_heads=cfg.qformer.n_heads)
            if holder.get("mute"):          # control run: does the answer depend on the audio at all?
                emb = torch.zeros_like(emb)
            return emb, torch.ones(emb.shape[:2], dtype=torch.long)
    sys.modules["SALMONN.models.salmonn_org"].SALMONN = StubSALMONN
    for m in ("models.custom_salmon", "models.model_factory", "models.multi_task_model", "inference.inference"):
        sys.modules.pop(m, None)
    import inference.inference as ref_cli

    def proc_init(self, tokenizer, max_length=128):
        self.processor, self.tokenizer, self.max_length, self.batch_counter = WhisperFeatureExtractor(), tokenizer, max_length, 1
    rmp.SalmonProcessor.__init__ = proc_init
    root = tempfile.mkdtemp(prefix="icl_golden_e2es_")
    real_makedirs = os.makedirs
    try:
        write_synthetic_hf_datasets(root, [MyDT("voxceleb"), MyDT("hvb"), MyDT("sqa")], **E2E_SPEECH_SIZES)
        seen = set()
        for cfg_ in (list(mc.DATASET_CONFIGS.values()) + c1.VOXCELEB_SWAP_CONFIGS + c2.HVB_SWAP_CONFIGS + c3.VOXPOPULI_SWAP_CONFIGS
                     + c4.MELD_EMOTION_SWAP_CONFIGS):
            for d in (cfg_.paths, cfg_.audio_lookup_paths):
                if d is not None and id(d) not in seen:
                    seen.add(id(d))
                    for k in list(d):
                        d[k] = os.path.join(root, os.path.basename(d[k].rstrip("/")))
        chosen = None
        # seeds 0..119 were searched once (9 minutes); 115 is the first above 12x (15.2x), so the search starts there
        for seed in range(int(os.environ.get("G23_FIRST_SEED", "115")), 400):
            cfg, sd = tiny_salmonn_weights(seed)
            lc = cfg.llama
            llama = LlamaForCausalLM(LlamaConfig(hidden_size=lc.hidden, intermediate_size=lc.ffn, num_hidden_layers=lc.n_layers,
                                                 num_attention_heads=lc.n_heads, num_key_value_heads=lc.n_heads, vocab_size=lc.vocab,
                                                 rms_norm_eps=lc.rms_eps, max_position_embeddings=lc.max_pos, pad_token_id=lc.pad_id,
                                                 bos_token_id=lc.bos_id, eos_token_id=lc.eos_id, tie_word_embeddings=False)).eval()
            lsd = {k[len("llama_model."):]: v for k, v in sd.items() if k.startswith("llama_model.")}
            missing, unexpected = llama.load_state_dict(lsd, strict=False)
            assert not unexpected and all("rotary" in m for m in missing), (missing, unexpected)
            holder.update(llama=llama, cfg=cfg, sd=sd)
            captured = []
            real_generate = llama.generate

            def spy(*a, **k):
                out_ids = real_generate(*a, **k)
                captured.append((k["inputs_embeds"].detach().clone(), out_ids.detach().clone()))
                return out_ids
            llama.generate = spy
            runs = {}
            for name, argv in list(E2E_SPEECH_RUNS.items()) + [("muted_control", E2E_SPEECH_RUNS["speech_query_text_exemplars"])]:
                holder["mute"] = name == "muted_control"
                if holder["mute"]:
                    n_real = len(captured)          # the control's decisions are not part of the golden: no margin asked of them
                clear_dataset_cache()
                random.seed(5)
                np.random.seed(6)
                res_dir = tempfile.mkdtemp(prefix="icl_golden_e2es_res_")
                with mock.patch.object(sys, "argv", ["inference.py", "--peft_model_path", "", "--run_name", "e2e", "--device", "cpu",
                                                     "--num_workers", "0", "--split", "test"] + argv):
                    args = ref_cli.parse_args()
                real_save = ref_cli.save_final_results
                real_create = ref_cli.ModelFactory.create_model

                def create(**kw):
                    model = real_create(**{**kw, "lora": False, "low_resource": False})
                    model.input_processor = None      # see g22
                    return model
                with mock.patch.object(ref_cli.ModelFactory, "create_model", staticmethod(create)), \
                        mock.patch.object(os, "makedirs", lambda p, *a, **k: real_makedirs(p, *a, **k) if not str(p).startswith("/data2") else None), \
                        mock.patch.object(ref_cli, "save_final_results", lambda results, a, d: real_save(results, a, res_dir)):
                    ret = ref_cli.run_inference(args)
                files = {}
                for fn in sorted(os.listdir(res_dir)):
                    with open(os.path.join(res_dir, fn)) as f:
                        files[fn] = json.load(f)
                shutil.rmtree(res_dir, ignore_errors=True)
                runs[name] = {"argv": argv, "results": ret["results"], "files": files}
            of, ob = om.LlamaOracle(lsd, lc.n_heads, lc.rms_eps), om.LlamaOracle(lsd, lc.n_heads, lc.rms_eps, rnd=om.bf16_round)
            worst = float("inf")
            for emb, ids in captured[:n_real]:
                ids = ids[:, : max(1, int((ids[0] != lc.pad_id).sum()))]
                lf, lb = of.teacher_forced_logits(emb, ids)[0], ob.teacher_forced_logits(emb, ids)[0]
                for t in range(ids.shape[1]):
                    top2 = lf[t].topk(2)
                    if int(top2.indices[0]) != int(ids[0, t]) or int(lb[t].argmax()) != int(ids[0, t]):
                        worst = -1.0
                        break
                    worst = min(worst, float(top2.values[0] - top2.values[1]) / max(float((lf[t] - lb[t]).abs().max()), 1e-6))
                if worst < 0:
                    break
            muted = runs.pop("muted_control")
            hears = [a["predicted_label"] != b["predicted_label"] for a, b in zip(runs["speech_query_text_exemplars"]["results"], muted["results"])]
            print(f"seed {seed}: {sum(len(r['results']) for r in runs.values())} records, smallest margin / bf16 distance = {worst:.1f}, "
                  f"answers that change when the audio is muted: {sum(hears)}/{len(hears)}")
            if worst > 12.0 and all(hears):
                chosen = (seed, runs, worst)
                break
        assert chosen is not None, "no seed with decisive margins"
        seed, runs, worst = chosen
        with open(os.path.join(HERE, "cli_e2e_speech.json"), "w") as f:
            json.dump({"sizes": {k: (list(v) if isinstance(v, tuple) else v) for k, v in E2E_SPEECH_SIZES.items()}, "weights_seed": seed,
                       "smallest_margin_over_bf16_distance": worst, "answers_change_when_audio_is_muted": True,
                       "runs": runs}, f, indent=0, default=str)
        print("cli_e2e_speech.json:", {k: [r["predicted_label"] for r in v["results"]] for k, v in runs.items()})
    finally:
        shutil.rmtree(root, ignore_errors=True)


def g24_qwen_glue_end_to_end():
    """The reference's CustomQwen (models/custom_qwen.py, unmodified: `forward` with its prompt_length label mask, `generate_output`
    with generate(max_new_tokens=10) + slice + decode) over HF Qwen2AudioForConditionalGeneration at this build's `tiny` dims
    (`from_pretrained` / `AutoProcessor.from_pretrained` replaced: no checkpoint is reachable; `peft` is an empty module, lora=False).
    Golden: labels, loss, logits at the last positions, and the ten generated ids — with a weights seed whose greedy margins are
    more than 12x the fp32-vs-bf16 oracle distance (audio tower + decoder), for the GPU test of this build's CustomQwen."""
    for name in ("peft",):
        sys.modules.setdefault(name, types.ModuleType(name))
    for attr in ("LoraConfig", "get_peft_model", "TaskType"):
        setattr(sys.modules["peft"], attr, getattr(sys.modules["peft"], attr, object))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.path.insert(0, HERE)
    from e2e_weights import qwen_batch, tiny_qwen_hf_model
    from icl_speech_text_llm_amd.runtime.qwen import normalize_qwen_keys
    from oracle import models as om
    sys.modules.pop("models.custom_qwen", None)
    import models.custom_qwen as rq
    holder = {}

    class Decoder:
        class tokenizer:
            @staticmethod
            def decode(ids, **k):
                return " ".join(str(int(i)) for i in ids)

        @staticmethod
        def batch_decode(ids, skip_special_tokens=True, clean_up_tokenization_spaces=False):
            return [" ".join(str(int(i)) for i in row) for row in ids]
    rq.Qwen2AudioForConditionalGeneration.from_pretrained = classmethod(lambda cls, *a, **k: holder["hf"])
    rq.AutoProcessor.from_pretrained = staticmethod(lambda *a, **k: Decoder())
    chosen = None
    for seed in range(int(os.environ.get("G24_FIRST_SEED", "0")), 300):
        cfg, hf = tiny_qwen_hf_model(seed)
        holder["hf"] = hf
        ref = rq.CustomQwen(model_path="none", lora=False, device=torch.device("cpu"), use_fp16=False)
        ref.batch_counter = 1
        batch, prompt_len = qwen_batch(cfg)
        with torch.no_grad():
            fwd = ref.forward(dict(batch))
            gen_batch = {k: (v[:, :prompt_len] if k in ("input_ids", "attention_mask") else v) for k, v in batch.items()}
            text = ref.generate_output(dict(gen_batch))
        gen_ids = [int(t) for t in text[0].split()]
        # margins against the fp32 / bf16-rounding oracles (audio tower + decoder)
        sd = normalize_qwen_keys({k: v.detach().clone() for k, v in hf.state_dict().items()})
        mel_lens = batch["feature_attention_mask"].sum(-1).tolist()
        lsd = {k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
        worst = float("inf")
        logits_by = {}
        for tag, rnd in (("fp32", None), ("bf16", om.bf16_round)):
            af, out_lens = om.qwen_audio_features(sd, batch["input_features"], mel_lens, cfg.audio.n_heads, rnd=rnd)
            llm = om.LlamaOracle(lsd, cfg.llm.n_heads, cfg.llm.rms_eps, cfg.llm.rope_theta, rnd=rnd)
            ids = batch["input_ids"][0, :prompt_len]
            emb = llm.embed(ids)
            emb[(ids == cfg.audio_token_id).nonzero().flatten()] = torch.cat([af[i, :n] for i, n in enumerate(out_lens)])
            logits_by[tag] = llm.teacher_forced_logits(emb[None], torch.tensor([gen_ids]))[0]
        lf, lb = logits_by["fp32"], logits_by["bf16"]
        for t, tok in enumerate(gen_ids):
            top2 = lf[t].topk(2)
            if int(top2.indices[0]) != tok or int(lb[t].argmax()) != tok:
                worst = -1.0
                break
            worst = min(worst, float(top2.values[0] - top2.values[1]) / max(float((lf[t] - lb[t]).abs().max()), 1e-6))
        print(f"seed {seed}: ids {gen_ids}, smallest margin / bf16 distance = {worst:.1f}")
        if worst > 12.0:
            chosen = (seed, fwd, gen_ids, worst, prompt_len)
            break
    assert chosen is not None
    seed, fwd, gen_ids, worst, prompt_len = chosen
    save("qwen_glue_e2e.npz", weights_seed=seed, margin=worst, prompt_length=prompt_len, labels=fwd["labels"][0], loss=fwd["loss"],
         logits_tail=fwd["logits"][0, -8:], gen_ids=torch.tensor(gen_ids))


def g15_boundary():
    """The plugin boundary as the reference declares it: inspect.signature of BaseModel's public methods,
    ModelFactory's static methods, CustomSALMONN / CustomQwen constructors and entry points, and the action table of the
    CLI's argparse parser (inference/inference.py:41-91).  Imports only: the absent third-party packages (SALMONN, peft) are
    empty in-memory modules, nothing is constructed or run."""
    import argparse
    import inspect
    for name in ("SALMONN", "SALMONN.models", "SALMONN.models.salmonn_org", "peft"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["SALMONN.models.salmonn_org"].SALMONN = getattr(sys.modules["SALMONN.models.salmonn_org"], "SALMONN", object)
    for attr in ("LoraConfig", "get_peft_model", "TaskType"):
        setattr(sys.modules["peft"], attr, getattr(sys.modules["peft"], attr, object))
    sys.path.insert(0, REF)
    for m in ("models.custom_salmon", "models.custom_qwen", "models.model_factory", "models.base_model"):
        sys.modules.pop(m, None)
    from models.base_model import BaseModel as RefBase
    from models.custom_salmon import CustomSALMONN as RefSALMONN
    from models.custom_qwen import CustomQwen as RefQwen
    from models.model_factory import ModelFactory as RefFactory

    def public_methods(cls, own_only=True):
        names = [n for n, v in vars(cls).items() if not n.startswith("_") or n == "__init__"] if own_only else dir(cls)
        out = {}
        for n in names:
            fn = inspect.getattr_static(cls, n)
            kind = "staticmethod" if isinstance(fn, staticmethod) else "classmethod" if isinstance(fn, classmethod) else "method"
            target = getattr(cls, n)
            if callable(target):
                out[n] = {"kind": kind, "abstract": bool(getattr(target, "__isabstractmethod__", False)),
                          "params": _signature_table(target)}
        return out

    table = {"BaseModel": public_methods(RefBase), "CustomSALMONN": public_methods(RefSALMONN),
             "CustomQwen": public_methods(RefQwen), "ModelFactory": public_methods(RefFactory)}
    # the CLI: capture the parser the reference's parse_args() builds
    captured = []
    real_parse = argparse.ArgumentParser.parse_args
    argparse.ArgumentParser.parse_args = lambda self, *a, **k: captured.append(self) or argparse.Namespace()
    try:
        sys.modules.pop("inference.inference", None)
        import inference.inference as ref_cli
        ref_cli.parse_args()
    finally:
        argparse.ArgumentParser.parse_args = real_parse
    actions = []
    for a in captured[0]._actions:
        if not a.option_strings or a.dest == "help":
            continue
        default = a.default
        if a.dest == "today":
            default = "<today %Y-%m-%d>"
        elif a.dest == "device":
            default = "<cuda if available else cpu>"
        actions.append({"flags": a.option_strings, "dest": a.dest, "action": type(a).__name__,
                        "type": getattr(a.type, "__name__", None), "default": default, "required": bool(a.required),
                        "choices": list(a.choices) if a.choices else None})
    table["inference_cli"] = actions
    with open(os.path.join(HERE, "boundary.json"), "w") as f:
        json.dump(table, f, indent=1)
    print("boundary.json:", {k: len(v) for k, v in table.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1:                  # e.g. `make_golden.py g14_reference_glue_sentencepiece`: regenerate one family only
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1_logmel()
    g2_whisper()
    g3_qformer()
    g4_llama()
    g5_reference_glue()
    g7_qwen2_audio()
    g8_clean_prediction()
    g9_metrics()
    g10_dataset_items()
    g11_sampling()
    g12_qwen_prompts()
    g13_performance_tracker()
    g14_reference_glue_sentencepiece()
    g15_boundary()
    g16_beam_search()
    g17_multi_task_wrapper()
    g18_results_files_and_inference_config()
    g19_interactive()
    g20_cli_loop()
    g21_model_factory()
    g22_cli_end_to_end()
    g23_cli_end_to_end_speech()
    g24_qwen_glue_end_to_end()
