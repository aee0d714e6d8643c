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
    single-process); ``--batch_size`` defaults to 16 because ragged prompts batch fine on this path.
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
from torch.utils.data import DataLoader, Subset

from ..config.inference_config import get_inference_config
from ..data.model_processors import get_processor
from ..data.synthetic_dataset import SyntheticICLDataset
from ..data.dataset_factory import DatasetFactory
from ..data.task_configs import DatasetType, parse_dataset_types, set_dataset_root
from ..utils.data_utils import load_dataset
from ..models.model_factory import ModelFactory, load_finetuned_checkpoint
from ..utils.evaluation_utils import clean_prediction, evaluate_predictions
from ..utils.performance_utils import PerformanceTracker

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
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--num_examples", type=int, default=5)
    p.add_argument("--num_workers", type=int, default=4)
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
            ckpt = torch.load(args.peft_model_path, map_location="cpu")
            n = load_finetuned_checkpoint(model, ckpt)
            logger.info("Updated %d parameters from %s", n, args.peft_model_path)
        else:
            logger.info("No checkpoint path provided, using base model without loading weights")
        model.to(args.device)
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
        total = len(dataset) if args.max_samples is None else min(len(dataset), args.max_samples)
        indices = list(range(rank, total, world))     # shard by rank, no padding duplicates
        loader = DataLoader(Subset(dataset, indices), batch_size=args.batch_size, shuffle=False,
                            num_workers=args.num_workers, pin_memory=args.pin_memory and torch.cuda.is_available(),
                            collate_fn=processor.collate_batch)
        model.eval()
        results: List[Dict[str, Any]] = []
        with torch.no_grad():
            for batch_idx, batch in enumerate(loader):
                try:
                    batch = {k: (v.to(args.device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
                    batch["max_new_tokens"] = args.max_new_tokens
                    t0 = time.time()
                    outputs = model.generate_output(batch)
                    dt = time.time() - t0
                    for i, (out, true_label) in enumerate(zip(outputs, batch["completion"])):
                        dt_i = batch["dataset_type"][i]
                        results.append({"text": batch["text"][i], "true_label": true_label,
                                        "predicted_label (cleaned)": clean_prediction(out, dt_i),
                                        "predicted_label": out.strip(), "dataset_type": DatasetType(dt_i).value})
                    tracker.update(dt, len(batch["input_ids"]))
                except Exception as e:   # a failed batch is logged and skipped, as in the reference (:370-373)
                    logger.error("Error processing batch %d: %s", batch_idx, e)
                    logger.debug(traceback.format_exc())
                    continue
        perf = tracker.get_summary()
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, results)
            merged: List[Dict[str, Any]] = []
            for i in range(max(len(g) for g in gathered)):     # restore the unsharded dataset order
                for g in gathered:
                    if i < len(g):
                        merged.append(g[i])
            results = merged
            counts = torch.tensor([perf["total_examples"]], dtype=torch.float64,
                                  device=args.device if torch.cuda.is_available() else "cpu")
            dist.all_reduce(counts)
            perf["total_examples_all_ranks"] = int(counts.item())
        if rank == 0:
            save_final_results(results, args, results_dir)
            tracker.log_summary()
        return {"results": results, "performance": perf}
    except Exception as e:
        logger.error("Error during inference: %s", e)
        logger.debug(traceback.format_exc())
        raise RuntimeError(f"Inference failed: {e}") from e
    finally:
        if dist is not None and dist.is_initialized():
            dist.destroy_process_group()


def save_final_results(results, args, results_dir):
    base = f"{args.run_name}_{args.dataset_type.replace(' ', '')}_{args.input_mode}_{args.fewshot_mode}_{args.num_examples}shots"
    if args.output_suffix:
        base += f"_{args.output_suffix}"
    with open(os.path.join(results_dir, f"{base}_results.json"), "w") as f:
        json.dump(results, f, indent=2)
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
