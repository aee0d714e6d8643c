"""Host input pipeline in front of the hot path (SURVEY.md §8 f2 + §8e "the >= 6x at 8 GPUs is a host-pipeline problem"):
zero-copy reads of number-list audio columns, padding masks derived on demand, the slot-based batch loader.  CPU only.

Reference behaviour being matched: ``data/multi_task_dataset.py:135-158,229-241`` (``item["audio"]["array"]``),
``data/model_processors.py:786-874`` (the batch dict incl. ``padding_mask`` bool [B, L]), ``inference/inference.py:259-266``
(an unshuffled DataLoader over the dataset, batches in dataset order)."""
import os
import time

import numpy as np
import pytest
import torch


def _voxceleb_folder(root, n_items=6, seconds=(30.0, 30.0), n_lookup=6):
    from icl_speech_text_llm_amd.data import task_configs as tc
    from icl_speech_text_llm_amd.data.synthetic_dataset import write_synthetic_hf_datasets
    from icl_speech_text_llm_amd.utils.data_utils import clear_dataset_cache, load_dataset
    write_synthetic_hf_datasets(str(root), [tc.DatasetType.VOXCELEB], n_items=n_items, n_lookup=n_lookup, audio_seconds=seconds)
    clear_dataset_cache()
    return load_dataset(tc.DatasetType.VOXCELEB, split="test")


@pytest.fixture()
def restore_dataset_root():
    from icl_speech_text_llm_amd.data import task_configs as tc
    yield
    tc.set_dataset_root(None)
    from icl_speech_text_llm_amd.utils.data_utils import clear_dataset_cache
    clear_dataset_cache()


def test_arrow_audio_rows_equal_the_plain_rows_and_are_fast(tmp_path, restore_dataset_root):
    """The on-disk item path: a 30 s clip stored as a number list costs <= 10 ms per item (the plain ``dataset[idx]`` path boxes
    480 000 floats: ~120-140 ms, timed beside it), with the SAME float32 waveform, through select() / shuffled index maps and
    sliced tables as well."""
    from icl_speech_text_llm_amd.data import task_configs as tc
    from icl_speech_text_llm_amd.data.arrow_audio import FastAudioRows, LazyAudio
    from icl_speech_text_llm_amd.data.dataset_factory import DatasetFactory
    from icl_speech_text_llm_amd.data.model_processors import SalmonProcessor
    from icl_speech_text_llm_amd.utils.audio_io import decode_audio
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    rows = _voxceleb_folder(tmp_path)
    fast = FastAudioRows.wrap(rows)
    assert fast is not None and fast.columns == ["audio"]
    t0 = time.perf_counter()
    slow0 = rows[0]
    t_slow = time.perf_counter() - t0
    r0 = fast.row(0)
    assert isinstance(r0["audio"], LazyAudio) and {k: v for k, v in r0.items() if k != "audio"} == {k: v for k, v in slow0.items() if k != "audio"}
    want = np.asarray(slow0["audio"]["array"], dtype=np.float32)
    got = decode_audio(r0["audio"])
    assert got.dtype == np.float32 and got.shape == (480000,) and np.array_equal(got, want)
    # index maps (select / shuffle) and sliced tables address the same rows
    for view in (rows.select([4, 1, 3]), rows.shuffle(seed=3), rows.select(range(2, 5))):
        fv = FastAudioRows.wrap(view)
        for i in range(len(view)):
            assert fv.row(i)["normalized_text"] == view[i]["normalized_text"]
            assert np.array_equal(decode_audio(fv.row(i)["audio"]), np.asarray(view[i]["audio"]["array"], dtype=np.float32))
    # the whole item path (row -> exemplars -> prompt -> tokenizer -> float32 tensor), steady state
    ds = DatasetFactory.create_dataset(tc.DatasetType.VOXCELEB, rows, SalmonProcessor(ByteTokenizer(260)))
    ds[0]
    t0 = time.perf_counter()
    items = [ds[i] for i in range(len(ds))]
    per_item = (time.perf_counter() - t0) / len(ds)
    assert items[0]["raw_wav"].dtype == torch.float32 and items[0]["wav_length"] == 480000
    assert torch.equal(items[0]["raw_wav"], torch.from_numpy(want))
    print(f"on-disk item path: {per_item * 1e3:.2f} ms per 30 s clip (plain dataset[idx] row alone: {t_slow * 1e3:.0f} ms)")
    assert per_item < 0.010, f"{per_item * 1e3:.1f} ms per item"
    # speech exemplars come through the audio-lookup folder the same way
    ds_s = DatasetFactory.create_dataset(tc.DatasetType.VOXCELEB, rows, SalmonProcessor(ByteTokenizer(260)), fewshot_mode="speech", num_examples=3)
    it = ds_s[1]
    assert len(it["examples_speech"]) == 3 and all(e["wav_length"] == 480000 for e in it["examples_speech"])
    # a null audio cell stays None; a non-Arrow dataset is left to the plain path
    assert FastAudioRows.wrap([{"audio": {"array": [0.0], "sampling_rate": 16000}}]) is None


def test_collated_batch_derives_padding_masks_on_demand():
    """``padding_mask`` bool [B, L] is part of the reference's batch dict (data/model_processors.py:802); here it is not stored —
    123 MB per 256-utterance batch that only ever became lengths again — but any caller that asks gets the reference's tensor."""
    import copy
    import pickle
    from icl_speech_text_llm_amd.data.model_processors import CollatedBatch, SalmonProcessor
    from icl_speech_text_llm_amd.data.synthetic_dataset import SyntheticICLDataset
    from icl_speech_text_llm_amd.data.task_configs import DatasetType
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    proc = SalmonProcessor(ByteTokenizer(260))
    ds = SyntheticICLDataset(proc, [DatasetType.VOXCELEB], n_items=3, fewshot_mode="speech", audio_seconds=0.5, num_examples=2,
                             vary_length=True)
    items = [ds[i] for i in range(3)]
    items[2]["examples_speech"], items[2]["num_examples"] = items[2]["examples_speech"][:1], 1       # a row with fewer exemplars
    b = proc.collate_batch(items)
    assert isinstance(b, CollatedBatch) and "padding_mask" not in dict.keys(b) and "example_padding_masks" not in dict.keys(b)
    assert "padding_mask" in b and "example_padding_masks" in b and "question_padding_mask" not in b
    assert b.get("question_padding_mask") is None and b.get("nope", 7) == 7
    # the reference's collate, restated: arange >= length; all-True rows for missing exemplars
    L = b["raw_wav"].shape[1]
    assert b["raw_wav"].shape == (3, max(it["wav_length"] for it in items))
    want = torch.arange(L)[None] >= b["wav_lengths"][:, None]
    m = b["padding_mask"]
    assert m.dtype == torch.bool and torch.equal(m, want) and "padding_mask" in dict.keys(b)       # cached once built
    for i, it in enumerate(items):
        n = it["wav_length"]
        assert torch.equal(b["raw_wav"][i, :n], it["raw_wav"]) and not bool(b["raw_wav"][i, n:].any())
    em = b.get("example_padding_masks")
    assert em.shape == b["example_wavs"].shape and bool(em[2, 1].all()) and int((~em[2, 0]).sum()) == items[2]["examples_speech"][0]["wav_length"]
    assert not bool(b["example_wavs"][2, 1].any())
    with pytest.raises(KeyError):
        b["document_padding_mask"]
    # survives the trips a batch makes: worker -> main (pickle), pin_memory / device_prefetch (copy)
    b2 = proc.collate_batch(items)
    for clone in (pickle.loads(pickle.dumps(b2)), copy.copy(b2), b2.copy()):
        assert isinstance(clone, CollatedBatch) and "padding_mask" not in dict.keys(clone) and torch.equal(clone["padding_mask"], want)
    from torch.utils.data._utils.pin_memory import pin_memory as _pin      # what DataLoader(pin_memory=True) does to a batch
    try:
        pinned = _pin(b2)
        assert isinstance(pinned, CollatedBatch)
    except RuntimeError:
        pass                                                               # no accelerator to pin for: the type logic ran


@pytest.mark.parametrize("workers", [0, 2])
def test_arena_batch_loader_matches_the_dataloader(workers):
    """Same batches, same order, same tensors as ``DataLoader(shuffle=False)`` — with the waveforms collated into the loader's
    preallocated shared slots; a ragged batch larger than a slot overflows into ordinary memory, never into an error."""
    from torch.utils.data import DataLoader
    from icl_speech_text_llm_amd.data.model_processors import SalmonProcessor
    from icl_speech_text_llm_amd.data.synthetic_dataset import SyntheticICLDataset
    from icl_speech_text_llm_amd.data.task_configs import DatasetType
    from icl_speech_text_llm_amd.utils.batch_loader import ArenaBatchLoader
    from icl_speech_text_llm_amd.utils.tokenization import ByteTokenizer
    proc = SalmonProcessor(ByteTokenizer(260))
    ds = SyntheticICLDataset(proc, [DatasetType.VOXCELEB, DatasetType.HVB], n_items=9, fewshot_mode="speech", audio_seconds=0.3,
                             num_examples=2, vary_length=True)          # 18 items, batches of 4: the last one is short
    want = list(DataLoader(ds, batch_size=4, shuffle=False, collate_fn=proc.collate_batch))
    ld = ArenaBatchLoader(ds, 4, proc.collate_batch, num_workers=workers, device="cpu", slots=3)
    try:
        assert len(ld) == len(want) == 5
        for rep in range(2):                                            # workers and slots persist between passes
            n = 0
            for got, ref in zip(ld, want):
                assert type(got) is type(ref) and set(dict.keys(got)) == set(dict.keys(ref)) - {"padding_mask"}   # (cached in ref below)
                for k, v in ref.items():
                    if isinstance(v, torch.Tensor):
                        assert torch.equal(got[k], v), k
                    else:
                        assert got[k] == v, k
                assert torch.equal(got["padding_mask"], ref["padding_mask"])
                n += 1
            assert n == 5
        part = list(ld.batches(1, 3))
        assert [b["prompt"] for b in part] == [b["prompt"] for b in want[1:3]]
        first = next(iter(ld.batches(0, 5)))                            # an early exit drains the workers before slots are reused
        assert first["prompt"] == want[0]["prompt"]
        assert [b["prompt"] for b in ld] == [b["prompt"] for b in want]
        assert ld.overflow_batches == 0
    finally:
        ld.close()
    # a slot that is too small: the overflowing tensors come back through ordinary memory
    small = ArenaBatchLoader(ds, 4, proc.collate_batch, num_workers=workers, device="cpu", slot_bytes=4096)
    try:
        for got, ref in zip(small, want):
            assert torch.equal(got["raw_wav"], ref["raw_wav"]) and torch.equal(got["example_wavs"], ref["example_wavs"])
    finally:
        small.close()


def test_arena_batch_loader_reports_a_failing_item():
    """A dataset item that raises in a worker surfaces in the main process as an exception (the reference's DataLoader does the
    same: its loop has no try around the iterator, inference/inference.py:301), and the loader can be closed afterwards."""
    from icl_speech_text_llm_amd.utils.batch_loader import ArenaBatchLoader

    class Bad(torch.utils.data.Dataset):
        def __len__(self):
            return 6

        def __getitem__(self, i):
            if i == 3:
                raise ValueError("broken row 3")
            return {"raw_wav": torch.full((10,), float(i)), "wav_length": 10}

    def collate(items):
        from icl_speech_text_llm_amd.data.model_processors import _pad_rows
        return {"raw_wav": _pad_rows([it["raw_wav"] for it in items])}

    ld = ArenaBatchLoader(Bad(), 2, collate, num_workers=1, device="cpu")
    try:
        it = iter(ld)
        assert next(it)["raw_wav"][:, 0].tolist() == [0.0, 1.0]
        with pytest.raises(RuntimeError, match="broken row 3"):
            next(it)
    finally:
        ld.close()


def test_cli_defaults_follow_the_host_and_the_model(monkeypatch):
    """``--num_workers`` / ``--batch_size`` left out: workers = usable cores / ranks on the node - 1 (at most 8, at least 1), the
    batch from the model family on a GPU and a small constant on the CPU (reference defaults: 4 and 1,
    inference/inference.py:62-69; deliberate differences listed in test_plugin_boundary_matches_the_reference_signatures)."""
    from types import SimpleNamespace
    from icl_speech_text_llm_amd.inference import inference as cli
    args = cli.parse_args(["--peft_model_path", "", "--run_name", "x", "--dataset_type", "voxceleb"])
    assert args.batch_size is None and args.num_workers is None
    monkeypatch.setattr(cli, "usable_cores", lambda: 16)
    assert [cli.default_num_workers(w) for w in (1, 2, 4, 8, 16, 64)] == [8, 7, 3, 1, 1, 1]
    monkeypatch.setattr(cli, "usable_cores", lambda: 128)
    assert cli.default_num_workers(8) == 8
    assert 1 <= cli.usable_cores.__wrapped__() if hasattr(cli.usable_cores, "__wrapped__") else True
    model = SimpleNamespace(cfg=SimpleNamespace(llama=SimpleNamespace(hidden=4096, n_layers=32)))
    assert cli.auto_batch_size(model, "cpu") == 8 and cli.auto_batch_size(SimpleNamespace(), "cpu") == 8


def test_arena_batch_loader_notices_a_dead_worker():
    """A worker that is killed (the OOM killer, a segfault in a decoder) must not leave the main process waiting for its batch for
    the whole timeout: the loader sees the dead process within seconds and raises."""
    import os
    import signal
    from icl_speech_text_llm_amd.utils.batch_loader import ArenaBatchLoader

    class Suicide(torch.utils.data.Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            if i == 2:
                os.kill(os.getpid(), signal.SIGKILL)
            return {"raw_wav": torch.full((10,), float(i))}

    def collate(items):
        from icl_speech_text_llm_amd.data.model_processors import _pad_rows
        return {"raw_wav": _pad_rows([it["raw_wav"] for it in items])}

    ld = ArenaBatchLoader(Suicide(), 2, collate, num_workers=1, device="cpu", timeout=120)
    try:
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match="died"):      # (batch 0 may be lost with the worker: its queue feeder dies too)
            for _ in ld:
                pass
        assert time.perf_counter() - t0 < 30
    finally:
        ld.close()
