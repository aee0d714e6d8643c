"""Inference CLI of the MI355X ICL path — same flags, loop and output files as the reference's
``inference/inference.py`` (:31-93 flags, :105-390 loop, :394-456 results/metrics JSON, :463-479 exit codes).

Differences, all additive:
  * ``--results_dir`` replaces the hard-coded ``/data2/.../metrics/{today}`` (:132);
  * datasets: ``--dataset_root DIR`` re-roots the reference's hard-coded cluster paths (data/*_config.py) to
    ``DIR/<same folder name>`` and runs the reference's item pipeline (``load_dataset`` → ``DatasetFactory`` →
    ``InferenceDataset`` / ``MultiTaskInferenceDataset``); ``--write_synthetic_datasets`` first fills DIR
    with seeded stand-in folders of the same column schema.  Without ``--dataset_root`` the CLI runs on the in-memory
    ``SyntheticICLDataset`` (same item schema and prompt templates, 30 s clips);
  * data-parallel inference: launched under ``torchrun`` (one process per GPU) the utterances are sharded
    ``i ≡ rank (mod world)`` and rank 0 gathers every rank's results before scoring (the reference's inference is
    single-process);
  * ``--batch_size`` defaults to AUTO, not the reference's 1 (its collate stacks un-padded prompts, :448-450, so 1 is all it can
    run): ragged prompts batch fine here and a row's speech embeddings, prefill and first-step logits do not depend on its
    batch (tested bit for bit; later tokens within the decode kernels' tolerance), so the default is a throughput choice made
    per model from the HBM that is free — 256 utterances for SALMONN-7B on an MI355X (the headline setting: ~145 utterances/s,
    ~140 GiB of HBM), 128 for SALMONN-13B, 64 for Qwen2-Audio (`auto_batch_size`); ``--batch_size 1`` reproduces the
    reference's loop shape;
  * ``--num_workers`` defaults to the cores this process may use divided by the ranks on the node, minus one for the main
    thread, at most 8 (the reference: 4); batches come from ``utils/batch_loader.ArenaBatchLoader`` — workers collate straight
    into preallocated pinned shared-memory slots and the H2D copy of batch i+1 runs under batch i's kernels — instead of
    ``DataLoader`` + ``pin_memory`` (same batches, same order: tests/test_host_pipeline.py).
The host loop is pinned to the reference's own ``run_inference`` by tests/golden/cli_loop.json (records, order, failing batch,
``--max_samples`` limiting whole batches, ``--debug_samples``, output files).
"""
from __future__ import annotations

import argparse
import datetime
import json
import logging
import os
import sys
import time
import traceback
from typing import Any, Dict, List

import torch
from torch.utils.data import Subset

from ..config.inference_config import get_inference_config
from ..data.model_processors import get_processor
from ..data.synthetic_dataset import SyntheticICLDataset
from ..data.dataset_factory import DatasetFactory
from ..data.task_configs import DatasetType, parse_dataset_types, set_dataset_root
from ..utils.batch_loader import ArenaBatchLoader
from ..utils.data_utils import load_dataset
from ..models.model_factory import ModelFactory, load_finetuned_checkpoint
from ..utils.evaluation_utils import clean_prediction, evaluate_predictions
from ..utils.performance_utils import PerformanceTracker
from ..runtime.binding import IclError
from ..runtime.dp import collective_device, gather_json_records, gather_results, shard_indices

logger = logging.getLogger(__name__)


def _bool(s) -> bool:
    return str(s).lower() in ("1", "true", "yes", "y")   # the reference's type=bool makes any non-empty string truthy (:84-90)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Run inference with ICL models (MI355X path)")
    p.add_argument("--peft_model_path", type=str, required=True, help="fine-tuned checkpoint ('' = none)")
    p.add_argument("--run_name", type=str, required=True)
    p.add_argument("--today", type=str, default=datetime.datetime.now().strftime("%Y-%m-%d"))
    p.add_argument("--output_suffix", type=str, default="")
    p.add_argument("--dataset_type", type=str, required=True, help="e.g. voxceleb or voxceleb-hvb-voxpopuli")
    p.add_argument("--split", type=str, default="test", choices=["train", "val", "test"])
    p.add_argument("--input_mode", type=str, default="speech_only", choices=["speech_only", "text_only", "speech_and_text"])
    p.add_argument("--fewshot_mode", type=str, default="text", choices=["text", "speech"])
    p.add_argument("--model_type", type=str, default="salmonn")
    p.add_argument("--batch_size", type=int, default=None,
                   help="utterances per generate call (default: auto from the model and the free HBM — 256 for SALMONN-7B on an "
                        "MI355X; reference default 1; results are batch-invariant)")
    p.add_argument("--num_examples", type=int, default=5)
    p.add_argument("--num_workers", type=int, default=None,
                   help="item-pipeline worker processes of this rank (default: usable cores / ranks on the node - 1, at most 8; reference: 4)")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--max_samples", type=int, default=None)
    p.add_argument("--save_per_dataset", action="store_true")
    p.add_argument("--fp16", action="store_true")
    p.add_argument("--bf16", action="store_true")
    p.add_argument("--optimize_batch_size", action="store_true")
    p.add_argument("--max_batch_size", type=int, default=32)
    p.add_argument("--compile", action="store_true")
    p.add_argument("--pin_memory", action="store_true", default=True)
    p.add_argument("--device", type=str, default="cuda" if torch.cuda.is_available() else "cpu")
    p.add_argument("--debug_samples", type=int, default=0)
    p.add_argument("--randomize_swap", type=_bool, default=False)
    p.add_argument("--balance_datasets", type=_bool, default=False)
    p.add_argument("--interleave", type=_bool, default=False)
    # additions
    p.add_argument("--results_dir", type=str, default=None, help="default: ./results/{today}")
    p.add_argument("--dataset_root", type=str, default=None, help="folder holding the tasks' HF datasets folders")
    p.add_argument("--write_synthetic_datasets", action="store_true",
                   help="fill --dataset_root with seeded synthetic folders of the reference's column schema first")
    p.add_argument("--synthetic_items", type=int, default=64, help="items per task of the synthetic dataset")
    p.add_argument("--arch", type=str, default=None, help="7b | 13b | tiny (default: inferred from llama_path)")
    p.add_argument("--max_new_tokens", type=int, default=10)
    return p.parse_args(argv)


def _dist_env():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return int(os.environ.get("RANK", "0")), world, int(os.environ.get("LOCAL_RANK", "0"))


def usable_cores() -> int:
    """Cores this process may run on: scheduler affinity, capped by a cgroup CPU quota when one is set."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def default_num_workers(world: int) -> int:
    """Item-pipeline workers per rank: the ranks of a node share its cores (8 ranks x 4 workers on 16 cores would oversubscribe
    them 2x), one core per rank stays with the main thread that drives the GPU."""
    return max(1, min(8, usable_cores() // max(1, world) - 1))


def auto_batch_size(model, device) -> int:
    """Utterances per generate call when ``--batch_size`` is not given: the widest batch of the model family that leaves the
    workspace (activations + K/V, measured per utterance at the BASELINE prompt lengths: 0.37 GiB SALMONN-7B, 0.68 GiB
    SALMONN-13B, 0.84 GiB Qwen2-Audio) plus two copies of the decoder weights inside 70 % of the HBM that is free now."""
    dev = torch.device(device)
    cfg = getattr(model, "cfg", None)
    lm = getattr(cfg, "llama", None) or getattr(cfg, "llm", None)
    if dev.type != "cuda" or lm is None:
        return 8
    is_qwen = hasattr(cfg, "audio_token_id")
    per_utt = 0.84 if is_qwen else 0.68 if lm.hidden > 4096 else 0.37
    weights = 2 * 12 * lm.hidden * lm.hidden * lm.n_layers * 2 / 2 ** 30 * 1.1
    free = torch.cuda.mem_get_info(dev)[0] / 2 ** 30 + torch.cuda.memory_reserved(dev) / 2 ** 30 - torch.cuda.memory_allocated(dev) / 2 ** 30
    bs = 64 if is_qwen else 256
    while bs > 1 and per_utt * bs + weights > 0.7 * free:
        bs //= 2
    return bs


def _tokenizer_of(model):
    tok = getattr(model, "llama_tokenizer", None)
    return tok if tok is not None else getattr(getattr(model, "input_processor", None), "tokenizer", None)


def _pad_token_id(model) -> int:
    tok = _tokenizer_of(model)
    pad = getattr(tok, "pad_token_id", None)
    return int(pad) if pad is not None else 0


def _label_token_ids(model, dataset_types):
    """First token id of every valid label of the run's tasks: the label-restricted slice of the first-step logits that the
    data-parallel gather carries.  Empty when no task has a closed label set."""
    from ..data.task_configs import get_dataset_config
    tok = _tokenizer_of(model)
    names: List[str] = []
    for dt in dataset_types:
        cfg = get_dataset_config(dt)
        for lab in (getattr(cfg, "valid_labels", None) or []):
            if lab not in names:
                names.append(lab)
    ids = []
    for lab in names:
        enc = tok(lab, add_special_tokens=False)["input_ids"]
        enc = enc.reshape(-1).tolist() if hasattr(enc, "reshape") else list(enc)
        ids.append(int(enc[0]) if enc else 0)
    return names, ids


def _generated_lengths(tokens: torch.Tensor, eos_id) -> torch.Tensor:
    """Tokens each row produced before its pad fill: up to and including the first EOS, else the full width."""
    n, w = tokens.shape
    out = torch.full((n,), w, dtype=torch.int32)
    if eos_id is not None and w:
        is_eos = tokens == int(eos_id)
        has = is_eos.any(1)
        out[has] = (is_eos.float().argmax(1)[has] + 1).to(torch.int32)
    return out


def run_inference(args) -> Dict[str, Any]:
    rank, world, local = _dist_env()
    dist = None
    if world > 1:
        import torch.distributed as dist
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            args.device = f"cuda:{local}"
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    try:
        dataset_types = parse_dataset_types(args.dataset_type)
        results_dir = args.results_dir or os.path.join("results", args.today)
        if rank == 0:
            os.makedirs(results_dir, exist_ok=True)
        tracker = PerformanceTracker(log_interval=10)
        config = get_inference_config(args.model_type)
        model_args = dict(config.get("model_args", {}))
        if args.arch:
            model_args["arch"] = args.arch
        model = ModelFactory.create_model(model_type=args.model_type, multi_task=False, device=args.device,
                                          low_resource=True, **model_args)
        if args.peft_model_path and args.peft_model_path.strip():
            if not os.path.exists(args.peft_model_path):      # utils/training_utils.py:91-92 of the reference (load_checkpoint)
                raise FileNotFoundError(f"Checkpoint file not found: {args.peft_model_path}")
            ckpt = torch.load(args.peft_model_path, map_location="cpu")
            n = load_finetuned_checkpoint(model, ckpt)
            logger.info("Updated %d parameters from %s", n, args.peft_model_path)
        else:
            logger.info("No checkpoint path provided, using base model without loading weights")
        model.to(args.device)
        if args.num_workers is None:
            args.num_workers = default_num_workers(int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        if args.batch_size is None:
            args.batch_size = auto_batch_size(model, args.device)
            logger.info("--batch_size not given: %d utterances per generate call (auto)", args.batch_size)
        if isinstance(getattr(model, "generation_config", None), dict):      # Qwen2-Audio: knobs live in the model's generation config
            model.generation_config["max_new_tokens"] = args.max_new_tokens       # (--max_new_tokens is this CLI's addition; 10 = reference)
        if args.model_type == "salmonn":
            processor = get_processor(args.model_type, model.input_processor, model.llama_tokenizer)
        else:   # the Qwen host processor computes its log-mel on the GPU: keep item processing in the main process
            processor = get_processor(args.model_type, model.input_processor)
            args.num_workers = 0
        if args.dataset_root:
            set_dataset_root(args.dataset_root)
            if args.write_synthetic_datasets:
                if rank == 0:
                    from ..data.synthetic_dataset import write_synthetic_hf_datasets
                    write_synthetic_hf_datasets(args.dataset_root, dataset_types, n_items=args.synthetic_items,
                                                n_lookup=max(8, args.num_examples + 3), n_fewshot=max(args.num_examples, 5),
                                                audio_seconds=(2.0, 8.0), splits=(args.split,))
                if dist is not None:
                    dist.barrier()
            rows = {}
            for dt in dataset_types:                                   # reference: inference.py:204-218
                full = load_dataset(dt, split=args.split)
                rows[dt] = full.select(range(args.debug_samples)) if args.debug_samples and args.debug_samples > 0 else full
                logger.info("Loaded dataset %s: %d examples", dt, len(rows[dt]))
            dataset = DatasetFactory.create_dataset(
                dataset_type=dataset_types, dataset=rows, processor=processor, is_training=False,
                input_mode=args.input_mode, fewshot_mode=args.fewshot_mode, num_examples=args.num_examples,
                random_examples=False, model_type=args.model_type, run_name=args.run_name,
                randomize_swap=args.randomize_swap, balance_datasets=args.balance_datasets, interleave=args.interleave)
        else:
            n_items = args.debug_samples if args.debug_samples and args.debug_samples > 0 else args.synthetic_items
            dataset = SyntheticICLDataset(processor, dataset_types, n_items=n_items, num_examples=args.num_examples,
                                          input_mode=args.input_mode, fewshot_mode=args.fewshot_mode, seed=1234,
                                          interleave=args.interleave)
        # --max_samples limits WHOLE batches in the reference (`if batch_idx * batch_size >= max_samples: break`, :301-303): 3 samples
        # at batch size 2 are 4 records (tests/golden/cli_loop.json, recorded from its run_inference)
        total = len(dataset)
        if args.max_samples is not None:
            total = min(total, -(-args.max_samples // args.batch_size) * args.batch_size)
        indices = shard_indices(total, rank, world)   # i ≡ rank (mod world), no padding duplicates
        loader = ArenaBatchLoader(Subset(dataset, indices), args.batch_size, processor.collate_batch, num_workers=args.num_workers,
                                  device=args.device, pin_memory=args.pin_memory and torch.cuda.is_available())
        model.eval()
        label_names, label_ids = _label_token_ids(model, dataset_types)
        pad_id = _pad_token_id(model)
        eos_id = getattr(_tokenizer_of(model), "eos_token_id", None)
        rows: Dict[int, Dict[str, Any]] = {}         # dataset index -> record (strings) of THIS rank
        ids_l, len_l, logit_l, idx_l = [], [], [], []
        failed_batches = failed_rows = 0
        fatal = None
        with torch.no_grad():
            for batch_idx, batch in enumerate(loader):      # batch i+1 is collated and copied to the device under batch i's kernels
                n_b = len(batch["prompt"])
                b_idx = indices[batch_idx * args.batch_size: batch_idx * args.batch_size + n_b]
                try:
                    batch["max_new_tokens"] = args.max_new_tokens      # SALMONN reads it from the batch dict (custom_salmon.py:708)
                    t0 = time.time()
                    res = model.generate_ids(batch, want_first_logits=True)
                    outputs = model.decode_ids(res.tokens)
                    dt = time.time() - t0
                    # rows the model could not run (prompt over max_pos) cost their own utterance only: they are left out of
                    # every list below and surface as missing indices, the other rows of the batch proceed
                    dropped = set(getattr(res, "dropped", ()) or ())
                    keep = [i for i in range(n_b) if i not in dropped]
                    if dropped:
                        failed_rows += len(dropped)
                        logger.error("Error processing dataset indices %s of batch %d: prompt + new tokens exceed the model's "
                                     "max_pos", [b_idx[i] for i in sorted(dropped)], batch_idx)
                    toks = torch.full((n_b, args.max_new_tokens), pad_id, dtype=torch.int32)
                    toks[:, :res.tokens.shape[1]] = res.tokens.to(torch.int32)
                    ids_l.append(toks[keep])
                    len_l.append(_generated_lengths(res.tokens, eos_id)[keep])
                    logit_l.append(res.first_logits[:, label_ids].to(torch.bfloat16).cpu()[keep])
                    idx_l.extend(b_idx[i] for i in keep)
                    for i in keep:
                        out, true_label, dt_i = outputs[i], batch["completion"][i], batch["dataset_type"][i]
                        rows[b_idx[i]] = {"text": batch["text"][i], "true_label": true_label,
                                          "predicted_label (cleaned)": clean_prediction(out, dt_i),
                                          "predicted_label": out.strip(), "dataset_type": DatasetType(dt_i).value}
                        if DatasetType(dt_i).value == "sqa":      # reference :358-361: SQA records carry the question text too
                            q = batch["question"]
                            rows[b_idx[i]]["question"] = q[i] if isinstance(q, list) else q
                    tracker.update(dt, len(keep))
                except (torch.cuda.OutOfMemoryError, IclError) as e:
                    # device-side failures are not "a bad sample": stop with a non-zero exit code — on EVERY rank (the others
                    # would otherwise wait in the result gather until the collective times out): leave the loop here, exchange
                    # the flag below, raise everywhere
                    fatal = e
                    logger.error("Device-side failure in batch %d (dataset indices %s): %s", batch_idx, b_idx, e)
                    break
                except Exception as e:   # a failed batch is logged and skipped, as in the reference (:370-373)
                    failed_batches += 1
                    logger.error("Error processing batch %d (dataset indices %s): %s", batch_idx, b_idx, e)
                    logger.debug(traceback.format_exc())
                    continue
        if world > 1:
            flag = torch.tensor([1.0 if fatal is not None else 0.0], device=collective_device(dist, args.device))
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if fatal is None and flag.item() > 0:
                fatal = RuntimeError("a device-side failure on another rank (see its log)")
        if fatal is not None:
            raise fatal
        perf = tracker.get_summary()
        perf["failed_batches"], perf["failed_rows"] = failed_batches, failed_rows
        if world > 1:
            # ONE fixed-shape all-gather of (index, gen_ids, gen_len, first-step label logits) + one of the UTF-8 string side;
            # rank 0 re-orders by DATASET INDEX, so a batch dropped on one rank shows up as missing indices, never as a shift.
            per_rank = -(-total // world)
            cat = lambda xs, shape, dt: torch.cat(xs) if xs else torch.zeros(shape, dtype=dt)     # noqa: E731
            got = gather_results(dist, args.device, torch.tensor(idx_l, dtype=torch.int64),
                                 cat(ids_l, (0, args.max_new_tokens), torch.int32), cat(len_l, (0,), torch.int32),
                                 cat(logit_l, (0, len(label_ids)), torch.bfloat16), per_rank)
            meta = gather_json_records(dist, args.device, [{k: rows[i][k] for k in ("text", "true_label", "dataset_type", "question")
                                                            if k in rows[i]} for i in idx_l], idx_l, per_rank)
            if rank == 0:
                texts = model.decode_ids(got["gen_ids"].cpu().to(torch.int64))
                rows = {}
                for r, i in enumerate(got["index"].tolist()):
                    m = meta[i]
                    rows[i] = {"text": m["text"], "true_label": m["true_label"],
                               "predicted_label (cleaned)": clean_prediction(texts[r], DatasetType(m["dataset_type"])),
                               "predicted_label": texts[r].strip(), "dataset_type": m["dataset_type"]}
                    if "question" in m:
                        rows[i]["question"] = m["question"]
                    if label_names:
                        rows[i]["first_step_label_logits"] = dict(zip(label_names, got["first_logits"][r].float().tolist()))
            counts = torch.tensor([perf["total_examples"], failed_batches], dtype=torch.float64,
                                  device=collective_device(dist, args.device))
            dist.all_reduce(counts)
            perf["total_examples_all_ranks"], perf["failed_batches"] = int(counts[0].item()), int(counts[1].item())
        elif label_names:
            ll = torch.cat(logit_l).float() if logit_l else torch.zeros(0, len(label_ids))
            for r, i in enumerate(idx_l):
                rows[i]["first_step_label_logits"] = dict(zip(label_names, ll[r].tolist()))
        results = [rows[i] for i in sorted(rows)]
        label_logits = [r.get("first_step_label_logits") for r in results]
        if rank == 0:
            missing = sorted(set(range(total)) - set(rows))
            perf["missing_indices"] = missing
            if missing:
                logger.error("%d of %d dataset indices have no result (failed batches): %s%s", len(missing), total,
                             missing[:20], " ..." if len(missing) > 20 else "")
            save_final_results(results, args, results_dir)
            tracker.log_summary()
        return {"results": results, "performance": perf, "label_logits": label_logits}
    except Exception as e:
        logger.error("Error during inference: %s", e)
        logger.debug(traceback.format_exc())
        raise RuntimeError(f"Inference failed: {e}") from e
    finally:
        if "loader" in locals():
            loader.close()
        if dist is not None and dist.is_initialized():
            dist.destroy_process_group()


def save_final_results(results, args, results_dir):
    """Reference :394-456: ``{run}_{datasets}_{input_mode}_{fewshot_mode}_{k}shots[_suffix]_results.json`` (the records as
    generated), then per dataset type the cleaned predictions + ``evaluate_predictions`` -> ``..._metrics.json``.  As in the
    reference an error in here is logged, not raised: what was written before it stays on disk and the run still succeeds."""
    try:
        clean = args.dataset_type.replace(" ", "") if "-" in args.dataset_type else args.dataset_type
        base = f"{args.run_name}_{clean}_{args.input_mode}_{args.fewshot_mode}_{args.num_examples}shots"
        if args.output_suffix:
            base += f"_{args.output_suffix}"
        # the reference's result records carry exactly five keys (:350-356); the first-step label logits that the data-parallel
        # gather brings along go to a file of their own, index-aligned with the results list
        label_logits = [r.pop("first_step_label_logits", None) for r in results]
        with open(os.path.join(results_dir, f"{base}_results.json"), "w") as f:
            json.dump(results, f, indent=2)
        if any(x is not None for x in label_logits):
            with open(os.path.join(results_dir, f"{base}_label_logits.json"), "w") as f:
                json.dump(label_logits, f)
        metrics = {}
        for dt in parse_dataset_types(args.dataset_type):
            rows = [r for r in results if r["dataset_type"] == dt.value]
            if rows:
                for r in rows:
                    r["predicted_label (cleaned)"] = clean_prediction(r["predicted_label"], dt)
                metrics[dt.value] = evaluate_predictions(rows, dt)
        with open(os.path.join(results_dir, f"{base}_metrics.json"), "w") as f:
            json.dump(metrics, f, indent=2)
        logger.info("Saved results and metrics under %s/%s_*.json", results_dir, base)
    except Exception as e:
        logger.error("Error saving final results: %s", e)
        logger.debug(traceback.format_exc())


def main(argv=None) -> int:
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    try:
        run_inference(parse_args(argv))
        logger.info("Inference completed successfully")
        return 0
    except Exception as e:
        logger.error("Inference failed: %s", e)
        return 1


if __name__ == "__main__":
    sys.exit(main())
