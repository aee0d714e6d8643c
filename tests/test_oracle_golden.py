"""CPU: pins the oracle (oracle/*.py) to the golden vectors minted from the reference stack
(tests/golden/make_golden.py): installed-transformers modules and the reference's own in-tree glue."""
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    z = np.load(os.path.join(G, name))
    arrays = {k: z[k] for k in z.files if not k.startswith("w:")}
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    return arrays, sd


def test_logmel_matches_whisper_feature_extractor():
    from oracle import audio_frontend as af
    a, _ = _load("logmel.npz")
    assert np.abs(af.slaney_mel_filters(80).T[::4] - a["mel_filters_T"]).max() < 1e-12
    for tag in ("0p5s", "7p3s", "30s"):
        L, seed = int(a[f"len_{tag}"]), int(a[f"seed_{tag}"])
        wav = np.clip(np.random.default_rng(seed).normal(0, 0.1, L), -1, 1).astype(np.float32)
        got = af.whisper_logmel(wav)[:, ::12]
        # the reference computes the STFT in f32 (torch.stft); the oracle in f64: agreement to 2e-5 of a [-1, 2] range
        assert np.abs(got - a[f"spec_{tag}"]).max() < 2e-5, tag


def test_product_filter_tables_match_oracle():
    from icl_speech_text_llm_amd.runtime import audio_tables as at
    from oracle import audio_frontend as af
    assert np.abs(at.slaney_mel_filters(80) - af.slaney_mel_filters(80)).max() < 1e-12
    assert np.abs(at.slaney_mel_filters(128) - af.slaney_mel_filters(128)).max() < 1e-12
    assert np.abs(at.kaldi_mel_banks() - af.kaldi_mel_banks()).max() < 1e-12


def test_whisper_encoder_matches_hf():
    from icl_speech_text_llm_amd.runtime.synth import whisper_sinusoids
    from oracle import models as om
    a, sd = _load("whisper_tiny.npz")
    sd["embed_positions.weight"] = whisper_sinusoids(1500, 32)
    c, t = torch.arange(80.0)[:, None], torch.arange(3000.0)[None, :]
    spec = (0.5 * torch.sin(0.01 * (c + 1.0) * t + c))[None]
    out = om.whisper_encoder(sd, spec, n_heads=2)
    assert np.abs(out[0, ::25].numpy() - a["out"]).max() < 2e-4


def test_qformer_matches_blip2_qformer():
    from oracle import models as om
    a, sd = _load("qformer_tiny.npz")
    out = om.qformer(sd, torch.from_numpy(a["query"]), torch.from_numpy(a["enc"]), n_heads=2)
    assert np.abs(out.numpy() - a["out"]).max() < 2e-5


def test_llama_forward_loss_and_generate_match_hf():
    from oracle import models as om
    a, sd = _load("llama_tiny.npz")
    llm = om.LlamaOracle(sd, n_heads=2, rms_eps=1e-5)
    emb = torch.from_numpy(a["emb"])
    labels = torch.from_numpy(a["labels"])
    logits, loss = llm.forward(emb, labels)
    assert np.abs(logits.numpy() - a["logits"]).max() < 2e-4
    assert abs(float(loss) - float(a["loss"])) < 1e-4
    assert llm.generate_greedy(emb, 10, eos_id=-1, pad_id=259).tolist() == a["gen_free"].tolist()
    # one row hits EOS mid-way -> pad fill while the other continues
    assert llm.generate_greedy(emb, 10, eos_id=int(a["eos_mid"]), pad_id=259).tolist() == a["gen_mid"].tolist()
    # EOS at step 0 -> width-1 output (min_length=1 is a no-op with inputs_embeds)
    g0 = llm.generate_greedy(emb[:1], 10, eos_id=int(a["eos0"]), pad_id=259)
    assert g0.tolist() == a["gen_eos0"].tolist() and g0.shape == (1, 1)
    # eos_token_id as a list (generation_config.json of Qwen2-Audio names two ids): either one ends a row
    g = np.load(os.path.join(G, "beam_tiny.npz"))
    two = llm.generate_greedy(emb, 10, eos_id=[int(t) for t in g["two_eos_greedy_eos"]], pad_id=259)
    assert two.tolist() == g["two_eos_greedy_seq"].tolist()


BEAM_CASES = ["free4", "free3_lp2", "free2_lp0", "eos_mid4", "eos_first4", "eos_mid4_lpneg", "eos_mid5_lp2", "one_row_eos3",
              "short3", "two_eos4", "two_eos2_lpneg", "two_eos3_lp2", "rep3", "rep4_eos", "rep2_lp2"]


@pytest.mark.parametrize("case", BEAM_CASES)
def test_beam_search_matches_hf(case):
    """oracle.generate_beam == HF generate(inputs_embeds, num_beams=K, length_penalty=...) on the miniature Llama: returned
    hypothesis (pad-filled after its EOS) and its length-penalised score."""
    from oracle import models as om
    a, sd = _load("llama_tiny.npz")
    g = np.load(os.path.join(G, "beam_tiny.npz"))
    llm = om.LlamaOracle(sd, n_heads=2, rms_eps=1e-5)
    emb = torch.from_numpy(a["emb"])
    K, lp, eos, max_new = g[case + "_knobs"][:4]
    kn = g[case + "_knobs"]
    if len(kn) > 4 and kn[4] >= 0:                 # HF's list form of eos_token_id
        eos = [int(eos), int(kn[4])]
    pen = float(kn[5]) if len(kn) > 5 else 1.0
    want = g[case + "_seq"]
    e = emb[1:] if want.shape[0] == 1 else emb
    ids, score = llm.generate_beam(e, int(max_new), eos_id=eos if isinstance(eos, list) else int(eos), pad_id=259, num_beams=int(K), length_penalty=float(lp),
                                   return_scores=True, repetition_penalty=pen)
    assert ids.tolist() == want.tolist()
    assert np.abs(score.numpy() - g[case + "_score"]).max() < 2e-4 * max(1.0, float(np.abs(g[case + "_score"]).max()))


@pytest.mark.parametrize("case", ["text_only", "speech_text_ex", "speech_speech_ex", "sqa_speech_text_ex",
                                  "sqa_speech_speech_ex", "sqa_speechtext_zero"])
def test_prompt_wrap_labels_logits_generate_match_reference_glue(case):
    """Our host-side prompt logic + the Llama oracle reproduce what the REFERENCE's CustomSALMONN computed."""
    from icl_speech_text_llm_amd.models.custom_salmon import build_labels, interleave_plan, interleave_plan_sqa, split_prompt
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    from oracle import models as om
    meta = json.load(open(os.path.join(G, "glue_cases.json")))[case]
    a, _ = _load(f"glue_{case}.npz")
    _, sd = _load("glue_llama.npz")
    llm = om.LlamaOracle(sd, n_heads=2, rms_eps=1e-5)
    tok = ByteTokenizer(260)
    speech = torch.from_numpy(a["speech"]) if "speech" in a else None
    examples = torch.from_numpy(a["examples"]) if "examples" in a else None
    n_ex = meta["num_examples"]
    sqa = meta.get("sqa", False)
    if sqa:
        speech = {"speech_q": torch.from_numpy(a["speech_q"]), "speech_d": torch.from_numpy(a["speech_d"])}
        examples = {"example_q": torch.from_numpy(a["examples_q"]), "example_d": torch.from_numpy(a["examples_d"])} if "examples_q" in a else None
    parts = split_prompt(meta["prompt"], n_ex, examples is not None, is_sqa=sqa)
    pieces = []
    n_emb = None if examples is None else (len(examples["example_q"]) if sqa else len(examples))
    for kind, i in (interleave_plan_sqa if sqa else interleave_plan)(len(parts), n_ex, n_emb, speech is not None):
        if kind == "text":
            pieces.append(llm.embed(torch.tensor(tok.encode(parts[i], add_special_tokens=False), dtype=torch.long)))
        elif sqa:
            pieces.append(speech[kind] if kind.startswith("speech") else examples[kind][i])
        else:
            pieces.append(speech if kind == "speech" else examples[i])
    wrapped = torch.cat(pieces, 0)
    assert wrapped.shape[0] == meta["S"]
    assert torch.equal(wrapped, torch.from_numpy(a["wrapped"]))                       # exact: pure gather
    tgt = tok([meta["completion"]], padding="longest", return_tensors="pt", add_special_tokens=False)
    labels = build_labels(wrapped.shape[0], tgt.input_ids, tgt.attention_mask)
    assert torch.equal(labels[0], torch.from_numpy(a["labels"]))
    full = torch.cat([wrapped, llm.embed(tgt.input_ids[0])], 0)[None]
    logits, loss = llm.forward(full, labels)
    assert np.abs(logits[0, -12:].numpy() - a["logits_tail"]).max() < 5e-4
    assert abs(float(loss) - float(a["loss"])) < 2e-4
    ids = llm.generate_greedy(wrapped[None], 10, eos_id=tok.eos_token_id, pad_id=tok.pad_token_id)
    assert ids[0].tolist() == a["gen_ids"].tolist()
    assert tok.batch_decode(ids, skip_special_tokens=True)[0] == meta["generated_text"]
    if "gen_ids_beams3" in a:       # the reference's generate_output with num_beams=3 in the batch dict (custom_salmon.py:709-714)
        b3 = llm.generate_beam(wrapped[None], 10, eos_id=tok.eos_token_id, pad_id=tok.pad_token_id, num_beams=3, length_penalty=1.0)
        assert b3[0].tolist() == a["gen_ids_beams3"].tolist()[: b3.shape[1]]
        assert tok.batch_decode(b3, skip_special_tokens=True)[0] == meta["generated_text_beams3"]


@pytest.mark.parametrize("case", ["text_only", "speech_text_ex", "speech_speech_ex"])
def test_reference_glue_with_a_sentencepiece_llama_tokenizer(case):
    """The same pin as above under a REAL sentencepiece Llama tokenizer (`load_llama_tokenizer` on a folder holding
    `tokenizer.model`, + `[PAD]`, the reference's `LlamaTokenizer.from_pretrained(llama_path, use_fast=False)` path): word-boundary
    pieces, the dummy-prefix space in front of every separately tokenised part, byte fallback — the reference's CustomSALMONN
    tokenised each prompt part with it (tests/golden/make_golden.py::g14) and our split / per-part tokenisation / interleave /
    labels must land on exactly its wrapped sequence, labels, logits and greedy ids."""
    from icl_speech_text_llm_amd.models.custom_salmon import build_labels, interleave_plan, split_prompt
    from icl_speech_text_llm_amd.utils.tokenization import load_llama_tokenizer
    from oracle import models as om
    meta = json.load(open(os.path.join(G, "glue_spm_cases.json")))[case]
    a, _ = _load(f"glue_spm_{case}.npz")
    _, sd = _load("glue_spm_llama.npz")
    llm = om.LlamaOracle(sd, n_heads=2, rms_eps=1e-5)
    tok = load_llama_tokenizer(os.path.join(G, "llama_spm"), 401)
    assert len(tok) == 401 and tok.pad_token_id == 400 and type(tok).__name__.startswith("LlamaTokenizer")
    assert tok.convert_ids_to_tokens(tok("Output: positive", add_special_tokens=False)["input_ids"])[0] == "\u2581Output"
    speech = torch.from_numpy(a["speech"]) if "speech" in a else None
    examples = torch.from_numpy(a["examples"]) if "examples" in a else None
    n_ex = meta["num_examples"]
    parts = split_prompt(meta["prompt"], n_ex, examples is not None)
    pieces = []
    for kind, i in interleave_plan(len(parts), n_ex, None if examples is None else len(examples), speech is not None):
        if kind == "text":
            ids = tok(parts[i], padding="longest", return_tensors="pt", add_special_tokens=False)["input_ids"].reshape(-1)   # as CustomSALMONN._segments does
            pieces.append(llm.embed(ids.long()))
        else:
            pieces.append(speech if kind == "speech" else examples[i])
    wrapped = torch.cat(pieces, 0)
    assert wrapped.shape[0] == meta["S"]
    assert torch.equal(wrapped, torch.from_numpy(a["wrapped"]))
    tgt = tok([meta["completion"]], padding="longest", return_tensors="pt", add_special_tokens=False, return_attention_mask=True)
    labels = build_labels(wrapped.shape[0], tgt["input_ids"], tgt["attention_mask"])
    assert torch.equal(labels[0], torch.from_numpy(a["labels"]))
    logits, loss = llm.forward(torch.cat([wrapped, llm.embed(tgt["input_ids"][0])], 0)[None], labels)
    assert np.abs(logits[0, -12:].numpy() - a["logits_tail"]).max() < 5e-4
    assert abs(float(loss) - float(a["loss"])) < 2e-4
    ids = llm.generate_greedy(wrapped[None], 10, eos_id=tok.eos_token_id, pad_id=tok.pad_token_id)
    assert ids[0].tolist() == a["gen_ids"].tolist()
    assert tok.batch_decode(ids, skip_special_tokens=True)[0] == meta["generated_text"]


def test_qwen2_audio_matches_hf():
    """Oracle audio tower (key padding, AvgPool, ln_post, projector) + scatter + Qwen2 LM (QKV bias) vs HF
    Qwen2AudioForConditionalGeneration on a 2-audio prompt."""
    from icl_speech_text_llm_amd.runtime.qwen import normalize_qwen_keys
    from icl_speech_text_llm_amd.runtime.synth import whisper_sinusoids
    from oracle import models as om
    a, sd = _load("qwen2_audio_tiny.npz")
    sd = normalize_qwen_keys(sd)
    sd["audio_tower.embed_positions.weight"] = whisper_sinusoids(1500, 32)
    c, t = torch.arange(128.0)[:, None], torch.arange(3000.0)[None, :]
    feats = torch.stack([0.5 * torch.sin(0.01 * (c + 1.0) * t + c), 0.4 * torch.cos(0.013 * (c + 2.0) * t)])
    mel_lens = a["mel_lens"].tolist()
    af, out_lens = om.qwen_audio_features(sd, feats, mel_lens, n_heads=2)
    assert out_lens == [750, 308]
    lsd = {k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
    llm = om.LlamaOracle(lsd, n_heads=2, rms_eps=1e-5)
    ids = torch.from_numpy(a["input_ids"])
    emb = llm.embed(ids)
    pos = (ids == 298).nonzero().flatten()
    emb[pos] = torch.cat([af[0, :750], af[1, :308]])
    logits, _ = llm.forward(emb[None])
    assert np.abs(logits[0, -24:].numpy() - a["logits_tail"]).max() < 5e-4
    assert np.abs(logits[0, ::97].numpy() - a["logits_strided"]).max() < 5e-4


def test_format_prompt_matches_reference_formatter():
    from icl_speech_text_llm_amd.data.model_processors import SalmonProcessor
    from icl_speech_text_llm_amd.data.task_configs import DatasetType, get_dataset_config
    gold = json.load(open(os.path.join(G, "format_prompt.json")))
    proc = SalmonProcessor(tokenizer=None)
    ex = [{"text": f"example sentence number {i} about things", "label": ["positive", "negative", "neutral"][i % 3]} for i in range(5)]
    sqa_ex = [{"question": f"what about item {i}", "document": f"the document number {i} says things", "completion": f"{i}.5 {i + 2}.25"}
              for i in range(2)]
    assert sum(k.startswith("sqa|") for k in gold) == 7
    for key, want in gold.items():
        dt, mode, few = key.split("|")
        tmpl = get_dataset_config(DatasetType(dt)).prompt_template
        if dt == "sqa":
            got = proc.format_prompt(tmpl, "document text", None if few == "zero" else sqa_ex, input_mode=mode,
                                     fewshot_mode="text" if few == "zero" else few, dataset_type=DatasetType.SQA,
                                     question="the question")
        else:
            got = proc.format_prompt(tmpl, "query text", None if few == "zero" else ex[:3], input_mode=mode,
                                     fewshot_mode="text" if few == "zero" else few)
        assert got == want, key


def test_clean_prediction_matches_reference():
    from icl_speech_text_llm_amd.utils.evaluation_utils import clean_prediction
    table = json.load(open(os.path.join(G, "clean_prediction.json")))
    assert len(table) >= 60
    for row in table:
        assert clean_prediction(row["raw"], row["dataset_type"]) == row["cleaned"], row


def _same(a, b, path=""):
    if isinstance(a, dict):
        assert isinstance(b, dict) and set(a) == set(b), (path, a, b)
        for k in a:
            _same(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), (path, a, b)
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    elif isinstance(a, float) or isinstance(b, float):
        a, b = float(a), float(b)
        assert (a != a and b != b) or abs(a - b) <= 1e-12 * max(1.0, abs(a)), (path, a, b)
    else:
        assert a == b, (path, a, b)


def test_metrics_match_reference_evaluate_predictions():
    """f1: every score table the reference writes to {run}_metrics.json (utils/evaluation_utils.py:16-467), for all task
    families incl. greek / swap variants and the reference's dead ends (MELD_EMOTION_SWAP, VP_NEL, VOXPOPULI_NEL, empty)."""
    from icl_speech_text_llm_amd.utils.evaluation_utils import evaluate_predictions, evaluate_vp_nel
    g = json.load(open(os.path.join(G, "metrics.json")))
    assert len(g["cases"]) >= 40
    for case in g["cases"]:
        got = evaluate_predictions([dict(p) for p in case["predictions"]], case["dataset_type"])
        want = case["metrics"]
        if "error" in want and "error" in got and set(want) == set(got):     # exception texts are scikit-learn's own
            continue
        _same(json.loads(json.dumps(got)), want, case["dataset_type"])
    for case in g["vp_nel"]:
        _same(json.loads(json.dumps(evaluate_vp_nel(case["gt"], case["pd"]))), case["metrics"], "vp_nel")


def test_beats_oracle_self_consistency():
    """BEATs has no upstream source here (parity unpinned vs upstream): check the restated Kaldi fbank against an
    independent straight-line computation of the same published formula, and the bucket function's invariants."""
    from oracle import audio_frontend as af, models as om
    rng = np.random.default_rng(5)
    wav = np.clip(rng.normal(0, 0.1, 4000), -1, 1).astype(np.float32)
    fb = af.kaldi_fbank(wav)
    assert fb.shape == (af.kaldi_num_frames(4000), 128)
    x = wav.astype(np.float64) * 32768.0
    f = x[160 * 3:160 * 3 + 400].copy()
    f -= f.mean()
    f = np.concatenate([[f[0] * (1 - 0.97)], f[1:] - 0.97 * f[:-1]])
    f *= np.array([(0.5 - 0.5 * np.cos(2 * np.pi * n / 399)) ** 0.85 for n in range(400)])
    spec = np.abs(np.fft.rfft(np.pad(f, (0, 112)))) ** 2
    ref = (np.log(np.maximum(af.kaldi_mel_banks() @ spec, np.finfo(np.float32).eps)) - af.FBANK_MEAN) / (2 * af.FBANK_STD)
    assert np.abs(fb[3] - ref).max() < 1e-4
    b = om.beats_relative_buckets(torch.arange(-900, 901))
    assert int(b.min()) >= 0 and int(b.max()) < 320 and int(b[900]) == 0
    assert torch.equal(b[901:981], torch.arange(1, 81) + 160) and torch.equal(b[820:900].flip(0), torch.arange(1, 81))


def test_sampling_distribution_matches_hf_logits_processors():
    """f4: the kept-token distribution of the sampled decode tail (repetition penalty → temperature → top-k → top-p → softmax)
    against HF's own processors, incl. ties at the top-k boundary and penalised tokens inside the nucleus."""
    from oracle import models as om
    a, _ = _load("sampling.npz")
    n = sum(k.startswith("knobs_") for k in a)
    assert n >= 7
    for i in range(n):
        temp, k, p, pen = a[f"knobs_{i}"].tolist()
        for b in range(3):
            ids, probs = om.sample_filter(a[f"logits_{i}"][b], a[f"prev_{i}"][b].tolist(), pen, temp, int(k), p)
            want = a[f"probs_{i}"][b]
            kept = np.nonzero(want > 0)[0]
            # exact ties at a cut are broken by torch.sort's internal order in HF and by token id here: the kept sets may
            # then differ, but only by tokens of identical score — compare the multiset of (score, probability)
            assert len(ids) == len(kept), (i, b)
            lg = a[f"logits_{i}"][b]
            mine_only, hf_only = sorted(set(ids.tolist()) - set(kept.tolist())), sorted(set(kept.tolist()) - set(ids.tolist()))
            assert sorted(lg[mine_only].tolist()) == sorted(lg[hf_only].tolist()), (i, b)
            assert np.abs(np.sort(probs) - np.sort(want[kept])).max() < 2e-6, (i, b)
            both = [t for t in ids.tolist() if t in set(kept.tolist())]
            assert np.abs(probs[[ids.tolist().index(t) for t in both]] - want[both]).max() < 2e-6, (i, b)
            assert all(probs[j] >= probs[j + 1] - 1e-9 for j in range(len(probs) - 1))
            assert om.sample_pick(probs, 0.0) == 0 and om.sample_pick(probs, 0.999999) == len(probs) - 1


def test_qwen_conversation_assembly_matches_reference():
    """QwenProcessor.format_prompt (default and SQA): same conversation items in the same order as the reference's
    (data/model_processors.py:240-383), rendered through a processor stand-in that dumps the structure."""
    from icl_speech_text_llm_amd.data.model_processors import QwenProcessor
    from icl_speech_text_llm_amd.data.task_configs import DatasetType

    class Dump:
        def apply_chat_template(self, conversation, add_generation_prompt=True, tokenize=False):
            return json.dumps({"conversation": conversation, "add_generation_prompt": add_generation_prompt}, sort_keys=True)

    gold = json.load(open(os.path.join(G, "qwen_prompts.json")))
    proc = QwenProcessor(Dump())
    ex = [{"text": f"example sentence {i}", "label": ["positive", "negative"][i % 2]} for i in range(3)]
    sqa_ex = [{"question": f"what about item {i}", "document": f"document {i} says things", "completion": f"{i}.5 {i + 2}.25",
               "answer": f"ans{i}"} for i in range(2)]
    assert len(gold) == 18
    for key, want in gold.items():
        kind, mode, few = key.split("|")
        fm = "text" if few == "zero" else few
        if kind == "sqa":
            got = proc.format_prompt("SYSTEM TEMPLATE", "document text", None if few == "zero" else sqa_ex, mode, fm,
                                     DatasetType.SQA, question="the question")
        else:
            got = proc.format_prompt("SYSTEM TEMPLATE", "query text", None if few == "zero" else ex, mode, fm, DatasetType.HVB)
        assert got == want, key
