"""CPU: the C-ABI library loads and exports every symbol include/icl_hip.h declares (no compute without a GPU);
argument validation fails loudly; host-side plugin logic; world_size-2 gloo data-parallel sharding."""
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import icl_speech_text_llm_amd.runtime.binding as b
    lib = b.load_library()
    header = open(os.path.join(ROOT, "include", "icl_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(icl_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(b.EXPORTED_SYMBOLS), declared ^ set(b.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.icl_abi_version() == b.ABI_VERSION


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "icl-speech-text-llm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


def test_ops_fail_loudly_without_gpu_operands():
    import icl_speech_text_llm_amd.runtime.binding as b
    a = torch.zeros(8, 64, dtype=torch.bfloat16)
    with pytest.raises(b.IclError, match="no CPU fallback"):
        b.gemm(a, a, torch.zeros(8, 8))
    if not torch.cuda.is_available():
        from icl_speech_text_llm_amd.runtime.config import SalmonnCfg
        from icl_speech_text_llm_amd.runtime.salmonn import SalmonnRuntime
        with pytest.raises(b.IclError, match="needs a GPU"):
            SalmonnRuntime(SalmonnCfg.tiny(), {}, device="cpu")


def test_model_factory_contract():
    from icl_speech_text_llm_amd.models.model_factory import ModelFactory
    with pytest.raises(RuntimeError, match="Failed to create model: Unknown model type"):
        ModelFactory.create_model("nope")
    m = ModelFactory.create_model("salmonn", device="cpu", arch="tiny", low_resource=True, llama_path="x", lora_alpha=32,
                                  use_cache=True)   # unknown kwarg tolerated (reference model_factory.py:142)
    assert m.speech_placeholder == "<SpeechHere>" and m.batch_counter == 0
    assert hasattr(m, "input_processor") and hasattr(m, "llama_tokenizer") and hasattr(m, "salmonn")
    # peft-style keys load through the 4 checkpoint conventions of inference/inference.py:157-177
    from icl_speech_text_llm_amd.models.model_factory import load_finetuned_checkpoint
    key = "llama_model.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight"
    for wrap in (lambda d: {"model_state_dict": {"salmonn." + k: v for k, v in d.items()}}, lambda d: {"model": d},
                 lambda d: {"state_dict": {"salmonn." + k: v for k, v in d.items()}}, lambda d: {"salmonn." + k: v for k, v in d.items()}):
        val = torch.full((8, 256), float(torch.rand(())))
        load_finetuned_checkpoint(m, wrap({key: val}))
        got = m.salmonn.state_dict()["llama_model.model.layers.0.self_attn.q_proj.lora_A.weight"]
        assert torch.allclose(got.float(), val.to(torch.bfloat16).float())


def test_processor_batch_layout_and_dataset_schema():
    from torch.utils.data import DataLoader
    from icl_speech_text_llm_amd.data.model_processors import SalmonProcessor
    from icl_speech_text_llm_amd.data.synthetic_dataset import SyntheticICLDataset
    from icl_speech_text_llm_amd.data.task_configs import DatasetType
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    proc = SalmonProcessor(ByteTokenizer(260))
    ds = SyntheticICLDataset(proc, [DatasetType.VOXCELEB, DatasetType.HVB], n_items=2, fewshot_mode="speech",
                             audio_seconds=0.5, num_examples=2, vary_length=True)
    b = next(iter(DataLoader(ds, batch_size=2, collate_fn=proc.collate_batch)))
    assert b["input_ids"].shape == (2, 1, 128) and b["raw_wav"].dtype == torch.float32
    assert b["padding_mask"].dtype == torch.bool and b["padding_mask"].shape == b["raw_wav"].shape
    assert b["example_wavs"].shape[:2] == (2, 2) and b["example_padding_masks"].shape == b["example_wavs"].shape
    assert (~b["padding_mask"]).sum(1).tolist() == b["wav_lengths"].tolist()
    assert "spectrogram" not in b            # K1 runs on the GPU inside the model
    assert b["num_examples"].tolist() == [2, 2] and len(b["prompt"]) == 2 and "<Example1>" in b["prompt"][0]
    ds2 = SyntheticICLDataset(proc, [DatasetType.VOXCELEB, DatasetType.HVB, DatasetType.VOXPOPULI], n_items=2,
                              input_mode="text_only", interleave=True)
    assert [ds2[i]["dataset_type"].value for i in range(4)] == ["voxceleb", "hvb", "voxpopuli", "voxceleb"]


def test_byte_tokenizer_protocol():
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    tok = ByteTokenizer(32001)
    assert tok.pad_token_id == 32000 and tok.eos_token_id == 2 and len(tok) == 32001
    e = tok("", padding="longest", return_tensors="pt", add_special_tokens=False)
    assert e.input_ids.shape == (1, 0)
    e = tok(["ab", "abcd"], padding="longest", return_tensors="pt", add_special_tokens=False).to("cpu")
    assert e["attention_mask"].tolist() == [[1, 1, 0, 0], [1, 1, 1, 1]] and e.input_ids[0, 2].item() == 32000
    assert tok.batch_decode(torch.tensor([[100, 101, 2, 32000]]), skip_special_tokens=True) == ["ab"]
    assert tok("héllo", return_tensors="pt").input_ids[0, 0].item() == 1


def test_performance_tracker_definition():
    from icl_speech_text_llm_amd.utils.performance_utils import PerformanceTracker
    t = PerformanceTracker()
    t.start_time -= 2.0
    t.update(0.5, 4)
    t.update(0.25, 4)
    s = t.get_summary()
    assert s["total_examples"] == 8 and s["step_count"] == 2 and s["avg_step_time"] == "0.3750s"
    assert 3.5 < float(s["examples_per_second"]) < 4.01       # total_examples / wall time since construction


_DP_SCRIPT = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from icl_speech_text_llm_amd.inference.inference import run_inference, parse_args
from icl_speech_text_llm_amd.runtime.salmonn import GenerateResult
import icl_speech_text_llm_amd.models.custom_salmon as cs

if os.environ.get("DP_STANDIN") == "1":
    # the offline sub-word stand-in tokenizer (what the benchmark and the offline CLI run with): every process first
    # tokenises DIFFERENT text, so ids handed out in first-seen order (the round-2 form) would differ between ranks
    from icl_speech_text_llm_amd.utils.tokenization import SubwordStandInTokenizer
    def _standin(path, vocab_size=32001):
        tok = SubwordStandInTokenizer(32001)
        tok.encode(" ".join(f"warm{os.environ.get('RANK', '0')}x{i}" for i in range(50)))
        return tok
    cs.load_llama_tokenizer = _standin

def fake_generate_ids(self, samples, want_first_logits=False):
    # CPU stand-in for the HIP generate: a deterministic function of the prompt (tests the DP plumbing only)
    if os.environ.get("DP_DROP") == "1" and int(os.environ.get("RANK", "0")) == 0 and self.batch_counter == 1:
        drop = (1,)          # row 1 of rank 0's second batch "exceeds max_pos": the other row of the batch must survive
    else:
        drop = ()
    if os.environ.get("DP_FAIL") == "1" and int(os.environ.get("RANK", "0")) == 1 and self.batch_counter == 1:
        self.batch_counter += 1
        raise ValueError("injected failure of rank 1's second batch")
    tok, V = self.llama_tokenizer, max(self.cfg.llama.vocab, len(self.llama_tokenizer))
    rows = [tok(f"neutral {len(p) % 7}", add_special_tokens=False, return_tensors="pt")["input_ids"].reshape(-1).tolist()
            + [tok.eos_token_id] for p in samples["prompt"]]
    w = max(len(r) for r in rows)
    toks = torch.tensor([r + [tok.pad_token_id] * (w - len(r)) for r in rows], dtype=torch.int64)
    first = torch.stack([torch.arange(V, dtype=torch.float32) * 0.25 + (len(p) % 13) for p in samples["prompt"]])
    self.batch_counter += 1
    return GenerateResult(tokens=toks, first_logits=first, dropped=drop)
cs.CustomSALMONN.generate_ids = fake_generate_ids
args = parse_args(["--peft_model_path", "", "--run_name", "dp", "--dataset_type", "voxceleb-hvb", "--device", "cpu",
                   "--arch", "tiny", "--synthetic_items", "5", "--batch_size", "2", "--num_workers", os.environ.get("DP_WORKERS", "0"),
                   "--input_mode", os.environ.get("DP_INPUT_MODE", "text_only"), "--results_dir", sys.argv[2]])
out = run_inference(args)
if int(os.environ["RANK"]) == 0:
    json.dump({"n": len(out["results"]), "texts": [r["text"] for r in out["results"]],
               "preds": [r["predicted_label"] for r in out["results"]],
               "label_logits": out["label_logits"],
               "missing": out["performance"]["missing_indices"], "failed": out["performance"]["failed_batches"],
               "failed_rows": out["performance"]["failed_rows"]},
              open(os.path.join(sys.argv[2], "summary.json"), "w"))
"""


def _run_dp(script, out_dir, world, port, extra_env=None):
    if world == 1:
        env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", **(extra_env or {}))
        cmd = [sys.executable, str(script), ROOT, str(out_dir)]
    else:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        env.update(extra_env or {})
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(out_dir)]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return json.load(open(os.path.join(out_dir, "summary.json")))


def test_data_parallel_sharding_gloo_world2(tmp_path):
    """N>1 path on CPU: 2 ranks over gloo shard the utterances i = rank (mod 2); ONE fixed-shape all-gather carries (dataset
    index, generated ids, generated length, first-step label logits) and rank 0 decodes and re-orders by index.  Checked
    against the single-process run: same texts, same predictions, same label logits, in dataset order — and with a batch
    failing on rank 1 the surviving records keep their places and the lost indices are reported, not shifted over."""
    script = tmp_path / "dp.py"
    script.write_text(_DP_SCRIPT)
    for d in ("w1", "w2", "w2f", "w3", "w2d", "s1", "s2"):
        (tmp_path / d).mkdir()
    a = _run_dp(script, tmp_path / "w1", 1, 0)
    b = _run_dp(script, tmp_path / "w2", 2, 29611)
    assert a["n"] == b["n"] == 10 and a["missing"] == b["missing"] == [] and b["failed"] == 0
    assert a["texts"] == b["texts"] and a["preds"] == b["preds"]
    assert a["label_logits"][0] and a["label_logits"] == b["label_logits"]       # bf16 slice of the first-step logits, by label
    assert any(f.endswith("_metrics.json") for f in os.listdir(tmp_path / "w2"))
    d3 = _run_dp(script, tmp_path / "w3", 3, 29615)                 # 10 utterances over 3 ranks: 4 + 3 + 3, padded rows carry index -1
    assert d3["n"] == 10 and d3["missing"] == [] and d3["texts"] == a["texts"] and d3["preds"] == a["preds"]
    assert d3["label_logits"] == a["label_logits"]
    c = _run_dp(script, tmp_path / "w2f", 2, 29613, {"DP_FAIL": "1"})
    assert c["failed"] == 1 and c["missing"] == [5, 7] and c["n"] == 8     # rank 1 holds 1,3,5,7,9: its 2nd batch is (5, 7)
    keep = [i for i in range(10) if i not in (5, 7)]
    assert c["texts"] == [a["texts"][i] for i in keep] and c["preds"] == [a["preds"][i] for i in keep]
    # one ROW over max_pos (GenerateResult.dropped) costs that utterance only: rank 0 holds 0,2,4,6,8, its 2nd batch is (4, 6)
    e = _run_dp(script, tmp_path / "w2d", 2, 29617, {"DP_DROP": "1"})
    assert e["failed"] == 0 and e["failed_rows"] == 1 and e["missing"] == [6] and e["n"] == 9
    keep = [i for i in range(10) if i != 6]
    assert e["texts"] == [a["texts"][i] for i in keep] and e["preds"] == [a["preds"][i] for i in keep]
    assert e["label_logits"] == [a["label_logits"][i] for i in keep]
    # with the offline sub-word stand-in tokenizer ids must be the same function of the text in every process:
    # world 2 (each rank's tokenizer warmed on different text) == the single-process run, decoded text and label logits alike
    s1 = _run_dp(script, tmp_path / "s1", 1, 0, {"DP_STANDIN": "1"})
    s2 = _run_dp(script, tmp_path / "s2", 2, 29619, {"DP_STANDIN": "1"})
    assert s1["n"] == s2["n"] == 10 and s1["preds"] == s2["preds"] and s1["preds"][0].startswith("neutral")
    assert s1["label_logits"] == s2["label_logits"] and s1["label_logits"][0]
    # round 4: every rank feeds itself through ArenaBatchLoader WORKER PROCESSES (collate into shared slots; speech_only items carry
    # waveforms through the arenas) — same records as the in-process loader
    (tmp_path / "k0").mkdir()
    (tmp_path / "k2").mkdir()
    k0 = _run_dp(script, tmp_path / "k0", 2, 29621, {"DP_INPUT_MODE": "speech_only"})
    k2 = _run_dp(script, tmp_path / "k2", 2, 29623, {"DP_INPUT_MODE": "speech_only", "DP_WORKERS": "2"})
    assert k0["n"] == k2["n"] == 10 and k0["texts"] == k2["texts"] and k0["preds"] == k2["preds"] and k2["missing"] == []


_BENCH_DP_SCRIPT = r"""
import argparse, json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import bench
from icl_speech_text_llm_amd.models.model_factory import ModelFactory
from icl_speech_text_llm_amd.runtime.salmonn import GenerateResult
import icl_speech_text_llm_amd.models.custom_salmon as cs

def fake_generate_ids(self, samples, want_first_logits=False):     # CPU stand-in for the HIP generate (DP plumbing only)
    tok, V = self.llama_tokenizer, len(self.llama_tokenizer)
    n = len(samples["prompt"])
    toks = torch.tensor([[3 + (len(p) + t) % 200 for t in range(10)] for p in samples["prompt"]], dtype=torch.int64)
    first = torch.stack([torch.arange(V, dtype=torch.float32) * 0.5 + len(p) % 11 for p in samples["prompt"]])
    self.last_stage_seconds = {"speech_launch": 0.0, "segments": 0.0, "generate": 0.0}
    return GenerateResult(tokens=toks, first_logits=first)
cs.CustomSALMONN.generate_ids = fake_generate_ids
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
args = argparse.Namespace(plugin_batch=None, batch=3, plugin_workers=0, tiny=True, plugin_audio_seconds=0.5)
block = bench.through_plugin(args, "cpu", dist=dist, rank=rank, world=world, n_batches=2, warm=1, workers=0,
                             make_model=lambda: ModelFactory.create_model("salmonn", device="cpu", arch="tiny", ckpt_path="").eval())
if rank == 0:
    json.dump(block, open(os.path.join(sys.argv[2], "plugin.json"), "w"))
else:
    assert block is None
dist.destroy_process_group()
"""


def test_bench_through_plugin_leg_runs_on_every_rank_gloo_world2(tmp_path):
    """at N > 1 bench.py's through-plugin leg (DataLoader -> H2D -> generate -> pack -> ONE all_gather_into_tensor
    per batch, barriers on both sides) runs on EVERY rank on its own shard, and rank 0 reports the whole-node rate next to each
    rank's generate / between-batches times and host ceiling.  Two ranks over gloo with a CPU stand-in for the HIP generate."""
    script = tmp_path / "bench_dp.py"
    script.write_text(_BENCH_DP_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", str(script), ROOT, str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    b = json.load(open(tmp_path / "plugin.json"))
    assert b["n_ranks"] == 2 and b["batch_size_per_rank"] == 3 and b["batches_timed_per_rank"] == 2
    assert b["last_batch_indices_ok"] is True and "all_gather_into_tensor" in b["collective"]
    assert [m["rank"] for m in b["per_rank"]] == [0, 1] and all(m["utterances"] == 6 for m in b["per_rank"])
    assert all(len(m["generate_output_ms"]) == 2 and len(m["between_batches_ms"]) == 1 for m in b["per_rank"])
    assert b["utt_per_s"] > 0 and b["generate_output_ms"]["min"] <= b["generate_output_ms"]["max"]
    assert b["host_ceiling_utt_per_s_per_rank"]["min"] > 0 and b["host_cores"] >= 1 and b["host_threads_per_rank"] >= 1


def test_row_packer_roundtrip_and_result_gather_layout():
    """runtime/dp.py byte rows: every field 16-byte aligned, round trip exact for int64 / int32 / bf16 (odd vocabulary width),
    zero-width logits allowed."""
    import torch
    from icl_speech_text_llm_amd.runtime.dp import RowPacker, result_packer
    pk = result_packer(10, 32001)
    assert pk.row_bytes % 16 == 0 and all(f[3] % 16 == 0 for f in pk.fields)
    g = torch.Generator().manual_seed(0)
    idx = torch.arange(5, dtype=torch.int64) * 7
    ids = torch.randint(0, 32001, (5, 10), generator=g, dtype=torch.int32)
    ln = torch.randint(1, 11, (5,), generator=g, dtype=torch.int32)
    lg = torch.randn(5, 32001, generator=g).to(torch.bfloat16)
    got = pk.unpack(pk.pack(pk.alloc(5, "cpu"), index=idx, gen_ids=ids, gen_len=ln, first_logits=lg))
    assert torch.equal(got["index"], idx) and torch.equal(got["gen_ids"], ids) and torch.equal(got["gen_len"], ln)
    assert torch.equal(got["first_logits"], lg)
    pk0 = result_packer(4, 0)
    got0 = pk0.unpack(pk0.pack(pk0.alloc(2, "cpu"), index=idx[:2], gen_ids=ids[:2, :4], gen_len=ln[:2],
                               first_logits=torch.zeros(2, 0, dtype=torch.bfloat16)))
    assert got0["first_logits"].shape == (2, 0) and torch.equal(got0["gen_ids"], ids[:2, :4])


def test_dataset_pipeline_matches_reference_items(tmp_path):
    """f2: rows of on-disk HF datasets → few-shot prompt / completion / audio, for every task family, input mode and
    multi-task ordering, against items captured from the reference's own DatasetFactory + InferenceDataset +
    SalmonProcessor over the same seeded folders (tests/golden/make_golden.py::g10_dataset_items)."""
    import random
    import numpy as np
    from icl_speech_text_llm_amd.data import task_configs as tc
    from icl_speech_text_llm_amd.data.dataset_factory import DatasetFactory
    from icl_speech_text_llm_amd.data.model_processors import SalmonProcessor
    from icl_speech_text_llm_amd.data.synthetic_dataset import write_synthetic_hf_datasets
    from icl_speech_text_llm_amd.utils.data_utils import clear_dataset_cache, load_dataset
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dataset_items.json")))
    sizes = dict(g["sizes"], audio_seconds=tuple(g["sizes"]["audio_seconds"]))
    all_types = sorted({t for c in g["cases"] for t in c["tasks"]})
    try:
        write_synthetic_hf_datasets(str(tmp_path), [tc.DatasetType(t) for t in all_types], **sizes)
        proc = SalmonProcessor(ByteTokenizer(260))
        assert len(g["cases"]) >= 25
        for case in g["cases"]:
            clear_dataset_cache()
            dts = [tc.DatasetType(t) for t in case["tasks"]]
            random.seed(5)
            np.random.seed(6)
            rows = {dt: load_dataset(dt, split="test") for dt in dts}
            ds = DatasetFactory.create_dataset(dataset_type=dts, dataset=rows, processor=proc, is_training=False,
                                               input_mode=case["input_mode"], fewshot_mode=case["fewshot_mode"],
                                               num_examples=case["num_examples"], random_examples=False,
                                               randomize_swap=False, balance_datasets=case["balance"],
                                               interleave=case["interleave"])
            assert len(ds) == case["len"], case["tasks"]
            items = [ds[i] for i in range(len(ds))]
            for it, want in zip(items, case["items"]):
                tag = (case["tasks"], case["input_mode"], case["fewshot_mode"])
                assert it["prompt"] == want["prompt"], tag
                assert it["completion"] == want["completion"] and it["text"] == want["text"], tag
                assert it["dataset_type"].value == want["dataset_type"] and int(it["num_examples"]) == want["num_examples"], tag
                if "question" in want:
                    assert it["question"] == want["question"] and it["unique_id"] == want["unique_id"]
                    assert [it["question_wav_length"], it["document_wav_length"]] == [want["question_wav_length"], want["document_wav_length"]]
                    assert [[e["question"]["wav_length"], e["document"]["wav_length"]] for e in it["examples_speech"]] == want["example_lengths"]
                    s = None if it["question_raw_wav"] is None else float(it["question_raw_wav"].double().sum() + it["document_raw_wav"].double().sum())
                else:
                    assert it["wav_length"] == want["wav_length"], tag
                    assert [e["wav_length"] for e in it["examples_speech"]] == want["example_lengths"], tag
                    s = None if it["raw_wav"] is None else float(it["raw_wav"].double().sum())
                assert (s is None) == (want["wav_sum"] is None) and (s is None or abs(s - want["wav_sum"]) < 1e-9), tag
            batch = proc.collate_batch(items[:3])
            for key, shape in case["batch3"].items():          # every tensor the reference batches (spectrograms aside: GPU K1)
                got = batch[key]
                assert (list(got.shape) if hasattr(got, "shape") else len(got)) == shape, (key, case["tasks"])
    finally:
        tc.set_dataset_root(None)
        clear_dataset_cache()


def test_hf_folder_ingestion_yields_canonical_names(tmp_path):
    """f3 (host part): HF Llama / Whisper folders → exactly the parameter names the packers read (runtime/synth.py emits the
    same set), architecture from config.json, one zero [PAD] row appended to embed_tokens and lm_head."""
    from transformers import LlamaConfig, LlamaForCausalLM, WhisperConfig, WhisperModel
    from icl_speech_text_llm_amd.runtime import checkpoints as ck, synth
    from icl_speech_text_llm_amd.runtime.config import LlamaCfg, QFormerCfg, SalmonnCfg
    LlamaForCausalLM(LlamaConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                                 intermediate_size=512, vocab_size=259, tie_word_embeddings=False)).save_pretrained(tmp_path / "l")
    WhisperModel(WhisperConfig(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256, decoder_layers=1,
                               decoder_attention_heads=2, decoder_ffn_dim=64, vocab_size=64, pad_token_id=0, bos_token_id=1,
                               eos_token_id=2, decoder_start_token_id=1)).save_pretrained(tmp_path / "w")
    lc = ck.llama_cfg_from_hf(ck.read_config(str(tmp_path / "l")), LlamaCfg(lora_rank=0))
    wc = ck.whisper_cfg_from_hf(ck.read_config(str(tmp_path / "w")))
    assert (lc.hidden, lc.n_layers, lc.n_heads, lc.ffn, lc.vocab, lc.pad_id) == (256, 2, 2, 512, 260, 259)
    sd = ck.load_pretrained_parts(str(tmp_path / "l"), str(tmp_path / "w"), "", vocab=lc.vocab)
    cfg = SalmonnCfg(whisper=wc, beats=None, qformer=QFormerCfg(enc_width=128), llama=lc)
    want = {k: tuple(v.shape) for k, v in synth.salmonn_state(cfg, seed=0, device="cpu").items()
            if k.startswith(("llama_model.", "speech_encoder."))}
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    assert float(sd["llama_model.model.embed_tokens.weight"][259].abs().max()) == 0.0
    with pytest.raises(NotImplementedError, match="grouped-query"):
        ck.llama_cfg_from_hf({"hidden_size": 256, "num_hidden_layers": 1, "num_attention_heads": 4, "num_key_value_heads": 2,
                              "intermediate_size": 512, "vocab_size": 10}, LlamaCfg())


def test_performance_tracker_matches_reference(monkeypatch):
    """a9: PerformanceTracker.update / get_summary / log lines under a fake clock, against the reference's class
    (tests/golden/make_golden.py::g13_performance_tracker): same keys, same string formats, same definition of examples/s."""
    import time
    from icl_speech_text_llm_amd.utils.performance_utils import PerformanceTracker

    class FakeClock:
        def __init__(self):
            self.t, self.n = 1000.0, 0

        def __call__(self):
            self.n += 1
            self.t += 0.125 + 0.03125 * (self.n % 5)
            return self.t

    updates = [(0.5, 16, None, None), (0.25, 16, 1.5, 480), (0.75, 8, 0.5, 200), (0.125, 1, None, 31), (1.0, 16, 2.0, None),
               (0.3, 16, None, 512), (0.2, 3, 0.25, 90)]
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "performance_tracker.json")))
    for interval, want in gold.items():
        lines = []

        class L:
            def info(self, msg):
                lines.append(str(msg))
        monkeypatch.setattr(time, "time", FakeClock())
        tr = PerformanceTracker(log_interval=int(interval), logger=L())
        for st, bs, loss, tok in updates:
            tr.update(st, bs, loss=loss, token_count=tok)
        summary = tr.get_summary()
        tr.log_summary()
        monkeypatch.undo()
        assert summary == want["summary"] and lines == want["lines"], interval


def test_plugin_boundary_matches_the_reference_signatures():
    """b-1, mechanically: names, parameter order and defaults of BaseModel / ModelFactory / CustomSALMONN / CustomQwen methods and the
    CLI's flag table against inspect.signature / argparse actions dumped from the imported reference
    (tests/golden/make_golden.py::g15_boundary -> boundary.json; models/base_model.py:21-58, models/model_factory.py:29-37,
    inference/inference.py:41-91).  This surface may ADD (extra keyword parameters after the reference's, extra flags); every
    other difference must be listed in DELIBERATE below with its reason."""
    import argparse
    import inspect
    from icl_speech_text_llm_amd.inference import inference as cli
    from icl_speech_text_llm_amd.models.base_model import BaseModel
    from icl_speech_text_llm_amd.models.custom_qwen import CustomQwen
    from icl_speech_text_llm_amd.models.custom_salmon import CustomSALMONN
    from icl_speech_text_llm_amd.models.model_factory import ModelFactory
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "boundary.json")))
    DELIBERATE = {
        # ragged prompts batch on this path and results are batch-invariant: the default is a throughput choice (CLI docstring)
        ("cli", "batch_size", "default"): (1, None),        # None = auto per model and free HBM (inference.auto_batch_size)
        # workers share the node's cores with the other ranks: None = usable cores / local ranks - 1, at most 8 (default_num_workers)
        ("cli", "num_workers", "default"): (4, None),
        # argparse's type=bool turns ANY non-empty string into True ("--interleave False" is True in the reference); _bool parses
        # true/false/1/0 and agrees with the reference on every spelling of True
        ("cli", "randomize_swap", "type"): ("bool", "_bool"), ("cli", "balance_datasets", "type"): ("bool", "_bool"),
        ("cli", "interleave", "type"): ("bool", "_bool"),
    }
    problems = []

    def table(fn):
        return [[n, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
                for n, p in inspect.signature(fn).parameters.items() if n not in ("self", "cls")]

    for cname, cls in (("BaseModel", BaseModel), ("CustomSALMONN", CustomSALMONN), ("CustomQwen", CustomQwen), ("ModelFactory", ModelFactory)):
        for mname, g in gold[cname].items():
            if not hasattr(cls, mname):
                problems.append(f"{cname}.{mname}: missing")
                continue
            raw = inspect.getattr_static(cls, mname)
            kind = "staticmethod" if isinstance(raw, staticmethod) else "classmethod" if isinstance(raw, classmethod) else "method"
            if kind != g["kind"]:
                problems.append(f"{cname}.{mname}: {kind} != {g['kind']}")
            ours = table(getattr(cls, mname))
            names = [o[0] for o in ours]
            ref_named = [r for r in g["params"] if not r[1].startswith("VAR_")]
            for r in g["params"]:
                if r[1].startswith("VAR_"):
                    if not any(o[1] == r[1] for o in ours):
                        problems.append(f"{cname}.{mname}: no {r[1]} parameter")
                elif r[0] not in names:
                    problems.append(f"{cname}.{mname}: parameter {r[0]} missing")
                elif ours[names.index(r[0])][2] != r[2]:
                    problems.append(f"{cname}.{mname}({r[0]}): default {ours[names.index(r[0])][2]} != {r[2]}")
            if [n for n in names if n in {r[0] for r in ref_named}] != [r[0] for r in ref_named]:
                problems.append(f"{cname}.{mname}: parameter order differs")
            # positional compatibility: the reference's named parameters come first, in its order
            if names[:len(ref_named)] != [r[0] for r in ref_named] and [r[0] for r in ref_named] and not problems:
                problems.append(f"{cname}.{mname}: extra parameters precede the reference's")
    captured = []
    real = argparse.ArgumentParser.parse_args
    argparse.ArgumentParser.parse_args = lambda self, *a, **k: captured.append(self) or argparse.Namespace()
    try:
        cli.parse_args([])
    finally:
        argparse.ArgumentParser.parse_args = real
    acts = {a.dest: a for a in captured[0]._actions}
    for g in gold["inference_cli"]:
        a = acts.get(g["dest"])
        if a is None:
            problems.append(f"cli: flag {g['flags']} missing")
            continue
        mine = {"flags": list(a.option_strings), "action": type(a).__name__, "type": getattr(a.type, "__name__", None),
                "default": g["default"] if g["dest"] in ("today", "device") else a.default, "required": bool(a.required),
                "choices": list(a.choices) if a.choices else None}
        for field, val in mine.items():
            if val != g[field] and DELIBERATE.get(("cli", g["dest"], field)) != (g[field], val):
                problems.append(f"cli --{g['dest']}: {field} {val!r} != {g[field]!r}")
    assert not problems, "\n".join(problems)
    from icl_speech_text_llm_amd.inference.inference import _bool
    assert _bool("True") is True and _bool("true") is True and _bool("1") is True and _bool("False") is False
    assert ModelFactory.get_available_models() == ["salmonn", "qwen2"] and ModelFactory.clear_cache() == 0
    assert set(ModelFactory.get_model_info("SALMONN")) == {"name", "full_name", "description", "default_paths", "supports_lora",
                                                           "supports_speech", "supports_audio"}
    with pytest.raises(ValueError):
        ModelFactory.get_model_info("whisper")
    with pytest.raises(RuntimeError, match="Failed to load model from checkpoint"):
        ModelFactory.get_model_from_checkpoint("/nonexistent/ckpt.pth", "", "salmonn")


def test_workspace_capacity_is_bounded_and_tracks_generation():
    """buffers are keyed by name with a capacity — 200 distinct ragged row counts keep ONE allocation per
    name (bytes bounded by the largest request), growth bumps `generation` (captured graphs must be retired), and zero=True
    buffers are re-zeroed when their inner dimensions change (their never-written regions sit at fixed flat offsets only
    for fixed inner dims)."""
    import numpy as np
    import torch
    from icl_speech_text_llm_amd.runtime.engines import Workspace, KVCache
    ws = Workspace("cpu")
    rng = np.random.default_rng(0)
    sizes = rng.integers(1, 5000, 200).tolist()
    for m in sizes:
        t = ws.get("dc_qkv", (m, 96), torch.bfloat16)
        assert t.shape == (m, 96) and t.is_contiguous()
        ws.get("gen_h", (m, 32), torch.float32)
    assert max(sizes) * (96 * 2 + 32 * 4) <= ws.nbytes() <= int(max(sizes) * (96 * 2 + 32 * 4) * 1.0625) + 8   # 1/16 regrowth headroom
    gen = ws.generation
    assert 0 < gen <= 2 * 12          # ~ln(200) record highs per name, not one per shape
    # the bench's ragged batches: totals that creep up by a fraction of a percent reallocate once, not once per record
    ws2 = Workspace("cpu")
    for m in (47744, 47900, 47950, 48000, 48128, 48100):
        ws2.get("gen_h", (m, 8), torch.float32)
    assert ws2.generation == 0
    a = ws.get("dc_qkv", (17, 96), torch.bfloat16)
    b = ws.get("dc_qkv", (4000, 96), torch.bfloat16)
    assert a.data_ptr() == b.data_ptr() and ws.generation == gen          # within capacity: same storage, no bump
    ws.get("dc_qkv", (6000, 96), torch.bfloat16)
    assert ws.generation == gen + 1
    # only buffers a captured decode graph points into (dc_* / gen_* / kv_*) retire the graphs when they move;
    # an encoder or prefill buffer outgrowing its capacity (a longer clip, a longer prompt) does not
    ws.get("pf_qkv", (100, 96), torch.bfloat16)
    ws.get("pf_qkv", (9000, 96), torch.bfloat16)
    ws.get("wh_ff", (10, 8), torch.bfloat16)
    ws.get("wh_ff", (1000, 8), torch.bfloat16)
    assert ws.generation == gen + 1
    # zero=True: tail columns stay zero across row counts; a change of inner dims re-zeroes
    z = ws.get("xn", (8, 40), torch.float32, zero=True)
    z[:, :32] = 1.0
    z2 = ws.get("xn", (5, 40), torch.float32, zero=True)
    assert float(z2[:, 32:].abs().sum()) == 0.0 and float(z2[:, :32].sum()) == 5 * 32
    z3 = ws.get("xn", (4, 50), torch.float32, zero=True)
    assert float(z3.abs().sum()) == 0.0

    class _C:
        n_layers, n_heads, head_dim = 2, 4, 8
    kv1 = KVCache(_C, 3, 64, ws)
    p = kv1.k.data_ptr()
    kv2 = KVCache(_C, 2, 64, ws)
    assert kv2.k.data_ptr() == p and kv2.k.shape == (2, 2, 4, 64, 8)     # one K allocation, re-viewed per batch shape


def test_audio_cells_decode_without_a_datasets_backend(tmp_path):
    """f2 / f3 (real on-disk formats): an HF ``Audio``-typed column (``data/multi_task_dataset.py:135-158`` reads
    ``item["audio"]["array"]`` and relies on the datasets decoder) is read here even when no decoder backend is installed:
    the column is cast to decode=False and the stored RIFF/WAVE bytes are parsed by utils/audio_io.py — PCM16, IEEE float, stereo
    (averaged), 8 kHz (resampled to 16 kHz); decoded cells and bare lists pass through; a FLAC payload raises a clear error."""
    import io, struct, wave
    import numpy as np
    from icl_speech_text_llm_amd.utils.audio_io import decode_audio, undecoded_audio_columns, audio_backend_available
    rng = np.random.default_rng(3)
    x = np.clip(rng.normal(0, 0.1, 4000), -1, 1).astype(np.float32)

    def pcm16(sig, sr=16000, ch=1):
        b = io.BytesIO()
        with wave.open(b, "wb") as w:
            w.setnchannels(ch); w.setsampwidth(2); w.setframerate(sr)
            w.writeframes((np.round(sig * 32767).astype("<i2")).tobytes())
        return b.getvalue()

    got = decode_audio({"bytes": pcm16(x), "path": None})
    assert got.dtype == np.float32 and got.shape == x.shape and float(np.abs(got - x).max()) <= 1.0 / 32767 + 1e-7
    stereo = np.stack([x, -x * 0.5], 1).reshape(-1)
    got2 = decode_audio({"bytes": pcm16(stereo, ch=2), "path": None})
    assert got2.shape == x.shape and float(np.abs(got2 - 0.25 * x).max()) < 1e-4
    f32 = b"RIFF" + struct.pack("<I", 36 + 4 * x.size) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, 16000, 64000, 4, 32) + \
        b"data" + struct.pack("<I", 4 * x.size) + x.astype("<f4").tobytes()
    assert np.array_equal(decode_audio({"bytes": f32}), x)
    low = decode_audio({"array": x[:2000].tolist(), "sampling_rate": 8000})
    assert low.shape == (4000,)                                           # 8 kHz -> 16 kHz
    assert np.array_equal(decode_audio({"array": x, "sampling_rate": 16000}), x) and np.array_equal(decode_audio(x.tolist()), x)
    assert decode_audio(None) is None
    with pytest.raises(ValueError, match="FLAC"):
        decode_audio({"bytes": b"fLaC" + b"\0" * 64})
    # through an HF dataset folder with an Audio-typed column and the item pipeline
    import datasets
    # Audio leaves nested in a list-of-dicts column (few_shot_examples) are switched to decode=False as well
    nested = datasets.Dataset.from_dict(
        {"few_shot_examples": [[{"text": "a", "audio": {"bytes": pcm16(x[:800]), "path": None}}]],
         "audio": [{"bytes": pcm16(x), "path": None}]})

    def typed_as(decode):
        return datasets.Features({"few_shot_examples": [{"text": datasets.Value("string"),
                                                         "audio": datasets.Audio(sampling_rate=16000, decode=decode)}],
                                  "audio": datasets.Audio(sampling_rate=16000, decode=decode)})
    nested = nested.cast(typed_as(False)).cast(typed_as(True))          # what a real folder declares: decoding Audio leaves
    nested = undecoded_audio_columns(nested)
    if not audio_backend_available():
        row = nested[0]                                         # would raise (no decoder) if any Audio leaf still decoded
        assert decode_audio(row["few_shot_examples"][0]["audio"]).shape == (800,) and decode_audio(row["audio"]).shape == x.shape
    from icl_speech_text_llm_amd.data import task_configs as tc
    from icl_speech_text_llm_amd.data.dataset_factory import DatasetFactory
    from icl_speech_text_llm_amd.data.model_processors import SalmonProcessor
    from icl_speech_text_llm_amd.data.synthetic_dataset import write_synthetic_hf_datasets
    from icl_speech_text_llm_amd.utils.data_utils import clear_dataset_cache, load_dataset
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    try:
        write_synthetic_hf_datasets(str(tmp_path), [tc.DatasetType.VOXCELEB], n_items=3, n_lookup=8, n_fewshot=5,
                                    audio_seconds=(0.3, 0.6), splits=("test",))
        clear_dataset_cache()
        cfg = tc.get_dataset_config(tc.DatasetType.VOXCELEB)
        path = cfg.get_path(tc.DatasetSplit.TEST)
        plain = datasets.load_from_disk(path)
        want = [np.asarray(r["audio"]["array"], dtype=np.float32) for r in plain]
        typed = plain.map(lambda r: {"audio": {"bytes": pcm16(np.asarray(r["audio"]["array"], dtype=np.float32)), "path": None}})
        typed = typed.cast_column("audio", datasets.Audio(sampling_rate=16000, decode=False))
        typed = typed.cast_column("audio", datasets.Audio(sampling_rate=16000))          # what a real folder declares
        import shutil
        typed.save_to_disk(path + "_typed")                 # a dataset cannot overwrite the folder it was loaded from
        del plain, typed
        shutil.rmtree(path)
        os.rename(path + "_typed", path)
        clear_dataset_cache()
        rows = load_dataset(tc.DatasetType.VOXCELEB, split="test")
        if not audio_backend_available():
            assert rows.features["audio"].decode is False
        ds = DatasetFactory.create_dataset(dataset_type=[tc.DatasetType.VOXCELEB], dataset={tc.DatasetType.VOXCELEB: rows},
                                           processor=SalmonProcessor(ByteTokenizer(260)), is_training=False, input_mode="speech_only",
                                           fewshot_mode="text", num_examples=2, random_examples=False)
        by_len = {w.shape[0]: w for w in want}              # the multi-task wrapper orders items its own way: match by length
        assert len(by_len) == len(want) == len(ds)
        for i in range(len(ds)):
            wav = np.asarray(ds[i]["raw_wav"], dtype=np.float32).reshape(-1)
            ref = by_len[wav.shape[0]]
            assert float(np.abs(wav - ref).max()) <= 1.0 / 32767 + 1e-6
    finally:
        tc.set_dataset_root(None)
        clear_dataset_cache()


def test_inline_asm_in_csrc_is_limited_to_memory_and_wait_instructions():
    """hipcc's hazard recogniser does not look inside asm statements: a vector instruction written as inline asm can be scheduled
    right behind the MFMA that produces its operand and read it before the matrix pipe has written it (round 2 found a row
    maximum going stale that way).  Inline asm in the kernels is therefore limited to waits, LDS / global reads and
    empty scheduling pins; arithmetic goes through the compiler."""
    import glob
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "icl-speech-text-llm_amd", "csrc")
    allow = {"s_waitcnt", "ds_read_b128", "ds_read_b64_tr_b16", "global_load_dwordx4"}
    seen, bad = 0, []
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        for m in re.finditer(r'\basm\s*(?:volatile\s*)?\(\s*"([^"]*)"', open(f).read()):
            for ins in re.split(r"\\n\\t|\\n|;", m.group(1)):
                ins = ins.strip()
                if ins:
                    seen += 1
                    if ins.split()[0] not in allow:
                        bad.append((os.path.basename(f), ins[:60]))
    assert seen > 10 and not bad, bad


def test_qwen_generation_config_json_is_inherited(tmp_path):
    """The reference calls generate(max_new_tokens=10) and inherits the rest from the checkpoint folder's generation_config.json
    (models/custom_qwen.py:227-233, HF from_pretrained): the plugin reads the same file; implemented knobs only, EOS list kept."""
    from icl_speech_text_llm_amd.models.custom_qwen import _read_generation_config
    assert _read_generation_config("Qwen/Qwen2-Audio-7B-Instruct") == {} and _read_generation_config(None) == {}
    (tmp_path / "generation_config.json").write_text(json.dumps({
        "bos_token_id": 151643, "do_sample": True, "eos_token_id": [151645, 151643], "pad_token_id": 151643,
        "repetition_penalty": 1.1, "temperature": 0.7, "top_k": 20, "top_p": 0.5, "transformers_version": "4.38.1",
        "no_repeat_ngram_size": 0}))
    g = _read_generation_config(str(tmp_path))
    assert g == {"do_sample": True, "temperature": 0.7, "top_p": 0.5, "top_k": 20, "repetition_penalty": 1.1,
                 "eos_token_id": [151645, 151643], "pad_token_id": 151643}
    (tmp_path / "generation_config.json").write_text(json.dumps({"eos_token_id": [5, 6, 7], "num_beams": 4, "length_penalty": 0.5}))
    assert _read_generation_config(str(tmp_path)) == {"num_beams": 4, "length_penalty": 0.5, "eos_token_id": [5, 6]}
    from icl_speech_text_llm_amd.runtime.binding import _eos_pair
    assert _eos_pair(2) == (2, -1) and _eos_pair([151645, 151643]) == (151645, 151643) and _eos_pair((7, 7)) == (7, -1)
    with pytest.raises(ValueError):
        _eos_pair([1, 2, 3])


def test_multi_task_wrapper_matches_the_reference():
    """models/multi_task_model.py:52-149 — task switching, prompt-template substitution and per-task generation knobs — against
    tests/golden/multi_task_wrapper.json, recorded from the reference's own MultiTaskModel (imported over a stub SALMONN) with the
    inner model's forward / generate_output replaced by recorders, as here."""
    from icl_speech_text_llm_amd.models.multi_task_model import MultiTaskModel
    with open(os.path.join(os.path.dirname(__file__), "golden", "multi_task_wrapper.json")) as f:
        want = json.load(f)
    tasks = {"sentiment": {"prompt_template": "SENTIMENT> ", "max_new_tokens": 4, "num_beams": 3},
             "intent": {"prompt_template": "INTENT> ", "do_sample": True, "temperature": 0.5},
             "plain": {"max_new_tokens": 7}}
    mt = MultiTaskModel("salmonn", task_configs=tasks, default_task="intent", arch="tiny", llama_path="none", device="cpu")
    mt.model.prompt_template = "BASE: "
    mt.model.batch_counter = 1
    seen = {}
    mt.model.forward = lambda samples: (seen.__setitem__("fwd", {k: (list(v) if isinstance(v, list) else v) for k, v in samples.items()}) or {"loss": 0.0})
    mt.model.generate_output = lambda samples: (seen.__setitem__("gen", dict(samples)) or ["out"] * len(samples["prompt"]))
    assert {t: mt.get_task_prompt_template(t) for t in ("sentiment", "intent", "plain", "unknown")} == want["templates"]
    assert mt.get_task_prompt_template() == want["template_current"]
    for t in ("sentiment", "nope", "plain"):
        assert [mt.set_task(t), mt.current_task] == want["set_task"][t]
    prompts = ["BASE: classify this", "BASE: and this BASE: twice", "no base here", "BASE: last"]
    r = mt.forward({"prompt": list(prompts), "task": ["sentiment", None, "intent", "plain"]})
    assert {"prompts": seen["fwd"]["prompt"], "task_out": r["task"]} == want["forward_with_tasks"]
    r = mt.forward({"prompt": list(prompts)})
    assert {"prompts": seen["fwd"]["prompt"], "task_out": r["task"]} == want["forward_without_tasks"]
    r = mt.forward({"prompt": list(prompts[:2]), "task": [None, None]})
    assert {"prompts": seen["fwd"]["prompt"], "task_out": r["task"]} == want["forward_all_none"]
    for name, samples in (("task_sentiment", {"prompt": ["p"], "task": ["sentiment"]}),
                          ("task_intent_overrides_batch_keys", {"prompt": ["p", "q"], "task": ["intent", "sentiment"], "max_new_tokens": 99}),
                          ("task_plain", {"prompt": ["p"], "task": ["plain"]}),
                          ("task_unknown", {"prompt": ["p"], "task": ["unknown"], "temperature": 0.3}),
                          ("task_none", {"prompt": ["p"], "task": [None]}),
                          ("no_task_key", {"prompt": ["p"]})):
        res = mt.generate_output(samples)
        got = {"samples_after": {k: v for k, v in seen["gen"].items() if k != "prompt"}, "current_task": mt.current_task, "n_out": len(res)}
        assert got == want["generate_output"][name], name


def test_results_files_and_inference_config_match_the_reference(tmp_path):
    """f1: save_final_results (inference/inference.py:394-456) leaves the same files with the same contents as the reference did on
    the same records (tests/golden/results_files.json: multi-dataset run, single dataset with a suffix, and a bad dataset type —
    where the reference logs the error, keeps the results file it had already written and does NOT raise), and
    get_inference_config (config/inference_config.py) returns the same dictionaries."""
    import argparse
    from icl_speech_text_llm_amd.config.inference_config import get_inference_config
    from icl_speech_text_llm_amd.data.task_configs import DatasetType
    from icl_speech_text_llm_amd.inference.inference import save_final_results
    with open(os.path.join(os.path.dirname(__file__), "golden", "results_files.json")) as f:
        want = json.load(f)

    def rec(dt, text, true, pred):
        return {"text": text, "true_label": true, "predicted_label": pred, "dataset_type": dt}
    results = [rec("voxceleb", "a fine day", "positive", " Positive."), rec("voxceleb", "bad", "negative", "the sentiment is neutral"),
               rec("hvb", "thanks a lot", ["thanks", "statement_close"], "thanks, statement_close,"), rec("hvb", "is it?", ["question_check"], "none"),
               rec("voxpopuli", "in Paris", {"place": ["Paris"]}, "place"), rec("voxpopuli", "nothing", {}, "None")]
    for name, ns in (("multi", dict(dataset_type="voxceleb-hvb-voxpopuli", output_suffix="")),
                     ("single_suffix", dict(dataset_type="voxceleb", output_suffix="v2")),
                     ("bad_dataset_type", dict(dataset_type="voxceleb-notadataset", output_suffix=""))):
        args = argparse.Namespace(run_name="run7", input_mode="speech_only", fewshot_mode="text", num_examples=5, **ns)
        d = tmp_path / name
        d.mkdir()
        save_final_results([dict(r) for r in results], args, str(d))          # must not raise, as the reference does not
        files = {}
        for fn in sorted(os.listdir(d)):
            with open(d / fn) as f:
                files[fn] = json.load(f)
        assert sorted(files) == sorted(want["save_final_results"][name]["files"]), name
        for fn, content in want["save_final_results"][name]["files"].items():
            assert json.loads(json.dumps(files[fn], default=str)) == content, (name, fn)
    for key, cfg in want["get_inference_config"].items():
        if key == "nope":
            with pytest.raises(ValueError, match=cfg.split(": ", 1)[1]):
                get_inference_config("nope")
            continue
        mt, dt = key.split("|")
        got = get_inference_config(mt, None if dt == "None" else DatasetType(dt))
        assert json.loads(json.dumps(got, default=str)) == cfg, key


def test_interactive_inference_matches_the_reference(monkeypatch):
    """inference/interactive_inference.py: the reference's flags are all there with its defaults (peft_model_path aside: the authors'
    cluster path), and run_interactive_inference hands generate_output exactly the batch the reference's own function does
    (tests/golden/interactive.json, recorded from it): the per-item generation knobs do NOT survive collate_batch, so a query is
    answered with the model's defaults (10 greedy tokens) unless --apply_generation_flags (ours) is given."""
    import argparse
    from icl_speech_text_llm_amd.data.model_processors import SalmonProcessor
    from icl_speech_text_llm_amd.inference import interactive_inference as ii
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    with open(os.path.join(os.path.dirname(__file__), "golden", "interactive.json")) as f:
        want = json.load(f)
    captured = []
    monkeypatch.setattr(argparse.ArgumentParser, "parse_args", lambda self, *a, **k: captured.append(self) or argparse.Namespace())
    ii.parse_args([])
    monkeypatch.undo()
    ours = {a.dest: a for a in captured[0]._actions if a.option_strings and a.dest != "help"}
    for ref in want["cli"]:
        a = ours[ref["dest"]]
        assert a.option_strings == ref["flags"] and type(a).__name__ == ref["action"] and getattr(a.type, "__name__", None) == ref["type"]
        if ref["dest"] not in ("device", "peft_model_path"):
            assert a.default == ref["default"], ref["dest"]
    assert set(ours) - {r["dest"] for r in want["cli"]} == {"arch", "seed", "apply_generation_flags"}
    seen = {}

    class Recorder:
        def eval(self):
            return self

        def generate_output(self, batch):
            seen["batch"] = batch
            return ["the answer"]
    proc = SalmonProcessor(ByteTokenizer(260), max_length=128)
    args = argparse.Namespace(device="cpu", max_new_tokens=100, temperature=0.8)
    assert ii.run_interactive_inference(Recorder(), proc, "What is the definition of positive?", args) == want["returned"]
    b = seen["batch"]
    assert sorted(b) == want["batch_keys"]
    assert {k: (str(v) if k == "dataset_type" else v) for k, v in b.items() if not isinstance(v, torch.Tensor)} == want["non_tensor"]
    assert {k: list(v.shape) for k, v in b.items() if isinstance(v, torch.Tensor)} == want["tensor_shapes"]
    assert b["input_ids"][0].tolist() == want["input_ids"] and b["num_examples"].tolist() == want["num_examples"]
    args = argparse.Namespace(device="cpu", max_new_tokens=7, temperature=0.5, apply_generation_flags=True, seed=3)
    ii.run_interactive_inference(Recorder(), proc, "q", args)
    assert seen["batch"]["max_new_tokens"] == 7 and seen["batch"]["temperature"] == 0.5 and seen["batch"]["do_sample"] is True


def test_cli_loop_matches_the_reference_run_inference(tmp_path, monkeypatch):
    """The whole host loop of the CLI against the reference's own run_inference (inference/inference.py:106-392, run unmodified for
    tests/golden/cli_loop.json over the same seeded on-disk datasets with the same stand-in model): record assembly and order,
    what --max_samples really limits (whole batches: 3 -> 4 records at batch 2), a failing batch costing exactly its records,
    --debug_samples, and the two files left on disk."""
    import random
    import numpy as np
    import icl_speech_text_llm_amd.models.custom_salmon as cs
    from icl_speech_text_llm_amd.data.synthetic_dataset import write_synthetic_hf_datasets
    from icl_speech_text_llm_amd.data.task_configs import DatasetType
    from icl_speech_text_llm_amd.inference.inference import parse_args, run_inference
    from icl_speech_text_llm_amd.runtime.salmonn import GenerateResult
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    with open(os.path.join(os.path.dirname(__file__), "golden", "cli_loop.json")) as f:
        want = json.load(f)
    preds = want["predictions"]
    sizes = {k: (tuple(v) if isinstance(v, list) else v) for k, v in want["sizes"].items()}
    root = tmp_path / "ds"
    write_synthetic_hf_datasets(str(root), [DatasetType("voxceleb"), DatasetType("hvb"), DatasetType("voxpopuli")], **sizes)
    monkeypatch.setattr(cs, "load_llama_tokenizer", lambda path, vocab_size=260: ByteTokenizer(260))
    state = {}

    def fake_generate_ids(self, samples, want_first_logits=False):
        state["calls"] += 1
        if state["calls"] - 1 == state["fail"]:
            raise ValueError("injected failure")
        tok = self.llama_tokenizer
        rows = [tok(preds[len(p) % len(preds)], add_special_tokens=False, return_tensors="pt")["input_ids"].reshape(-1).tolist()
                + [tok.eos_token_id] for p in samples["prompt"]]
        w = max(len(r) for r in rows)
        toks = torch.tensor([r + [tok.pad_token_id] * (w - len(r)) for r in rows], dtype=torch.int64)
        V = max(self.cfg.llama.vocab, len(tok))
        return GenerateResult(tokens=toks, first_logits=torch.zeros(len(rows), V) if want_first_logits else None)
    monkeypatch.setattr(cs.CustomSALMONN, "generate_ids", fake_generate_ids)
    for name, run in want["runs"].items():
        state.update(calls=0, fail=run["fail_batch"])
        random.seed(5)
        np.random.seed(6)
        res_dir = tmp_path / name
        res_dir.mkdir()
        args = parse_args(["--peft_model_path", "", "--run_name", "g20", "--device", "cpu", "--num_workers", "0", "--split", "test",
                           "--arch", "tiny", "--dataset_root", str(root), "--results_dir", str(res_dir)] + run["argv"])
        ret = run_inference(args)
        assert state["calls"] == run["generate_calls"], name
        got = [{k: v for k, v in r.items() if k != "first_step_label_logits"} for r in ret["results"]]
        assert json.loads(json.dumps(got, default=str)) == run["results"], name
        for fn, content in run["files"].items():
            with open(res_dir / fn) as f:
                assert json.load(f) == content, (name, fn)
        assert set(run["performance_keys"]) <= set(ret["performance"]) and ret["performance"]["total_examples"] == run["total_examples"]


def test_model_factory_messages_match_the_reference():
    """models/model_factory.py: error messages of create_model / from_config / get_model_info / get_model_from_checkpoint, the
    description records, the model list and clear_cache's return value, as the reference's own class produced them
    (tests/golden/model_factory.json)."""
    from icl_speech_text_llm_amd.models.model_factory import ModelFactory
    with open(os.path.join(os.path.dirname(__file__), "golden", "model_factory.json")) as f:
        want = json.load(f)

    def err(fn):
        try:
            fn()
            return None
        except Exception as e:
            return f"{type(e).__name__}: {e}"
    got = {"create_model_unknown": err(lambda: ModelFactory.create_model("Whisper")),
           "create_model_multi_without_tasks": err(lambda: ModelFactory.create_model("salmonn", multi_task=True)),
           "create_model_multi_empty_tasks": err(lambda: ModelFactory.create_model("QWEN2", multi_task=True, task_configs={})),
           "from_config_no_type": err(lambda: ModelFactory.from_config({})),
           "from_config_multi_without_tasks": err(lambda: ModelFactory.from_config({"model_type": "salmonn", "multi_task": True})),
           "from_config_unknown": err(lambda: ModelFactory.from_config({"model_type": "gpt"})),
           "get_model_info": {t: ModelFactory.get_model_info(t) for t in ("salmonn", "qwen2", "SALMONN")},
           "get_model_info_unknown": err(lambda: ModelFactory.get_model_info("gpt")),
           "get_available_models": ModelFactory.get_available_models(),
           "clear_cache": ModelFactory.clear_cache(),
           "checkpoint_missing": err(lambda: ModelFactory.get_model_from_checkpoint("/nonexistent/x.pt", "base", "salmonn"))}
    for k in want:
        assert got[k] == want[k], (k, got[k], want[k])


def test_checkpoint_dispatch_matches_the_reference(tmp_path):
    """--peft_model_path (inference/inference.py:157-177): which load_state_dict receives the keys for the four checkpoint layouts,
    always strict=False, as recorded from the reference's run_inference (tests/golden/cli_loop.json, `checkpoint_dispatch`); a
    missing file fails the run with the reference's message."""
    from icl_speech_text_llm_amd.inference import inference as cli
    from icl_speech_text_llm_amd.models.model_factory import load_finetuned_checkpoint
    with open(os.path.join(os.path.dirname(__file__), "golden", "cli_loop.json")) as f:
        want = json.load(f)["checkpoint_dispatch"]
    sd = {"speech_llama_proj.weight": torch.zeros(2, 2),
          "llama_model.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight": torch.ones(1, 2)}
    layouts = {"model_state_dict": {"model_state_dict": sd, "epoch": 3}, "state_dict": {"state_dict": sd}, "model": {"model": sd}, "raw": sd}
    for name, ckpt in layouts.items():
        calls = []

        class Inner:
            def load_state_dict(self, state, strict=True):
                calls.append(["model.salmonn.load_state_dict", sorted(state), strict])

        class Recording:
            salmonn = Inner()

            def load_state_dict(self, state, strict=True):
                calls.append(["model.load_state_dict", sorted(state), strict])
        assert load_finetuned_checkpoint(Recording(), ckpt) == len(sd)
        assert calls == want[name]["calls"], name
    args = cli.parse_args(["--peft_model_path", str(tmp_path / "nope.pt"), "--run_name", "x", "--dataset_type", "voxceleb", "--device", "cpu",
                           "--arch", "tiny", "--synthetic_items", "1", "--num_workers", "0", "--input_mode", "text_only",
                           "--results_dir", str(tmp_path)])
    with pytest.raises(RuntimeError) as e:
        cli.run_inference(args)
    assert f"{type(e.value).__name__}: {e.value}".replace(str(tmp_path / "nope.pt"), "<ckpt>") == want["missing_file"]["error"]
