"""Interactive text-query loop over the same generate path.

Mirror of the reference's inference/interactive_inference.py: same flags (:23-44), same two functions — ``setup_model(args)``
(:46-169: ModelFactory + optional fine-tuned checkpoint + processor) and ``run_interactive_inference(model, processor, query,
args)`` (:171-232: ``process_inputs`` of a text-only VOXCELEB-typed item with an empty completion, a one-item
``collate_batch``, then ``generate_output``) — and the same prompt loop in ``main`` (:234-281: first query from ``--query`` or
stdin, then until 'exit' / 'quit' / 'q').

The reference puts ``max_new_tokens`` / ``temperature`` / ``do_sample=True`` into the ITEM dict (:190-193); its processor's
``collate_batch`` carries only ``prompt`` / ``completion`` / ``text`` / ``dataset_type`` next to the tensors
(data/model_processors.py:871-874), so the knobs never reach ``generate_output`` and the reference answers every query with
10 greedy tokens whatever the flags say (pinned by tests/golden/interactive.json, recorded from the reference's own function).
This mirror does the same by default — same command line, same text; ``--apply_generation_flags`` (not a reference flag)
makes ``--max_new_tokens`` / ``--temperature`` effective, i.e. sampled generation as the flag help promises.

Differences, all on the host side: no CUDA_VISIBLE_DEVICES pinning and no low-memory device_map logic (one process per GPU
owns its whole card here); ``--compile`` is accepted and ignored (there is no tracing compiler in this build); ``--seed``
makes the sampled tokens reproducible (the kernel consumes host-supplied uniforms).  The model runs on the HIP library
only: without a GPU / without libicl_hip.so construction fails loudly, as everywhere else.
"""
from __future__ import annotations

import argparse
import logging
import sys
import traceback

import torch

from ..config.inference_config import get_inference_config
from ..data.model_processors import get_processor
from ..data.task_configs import DatasetType
from ..models.model_factory import ModelFactory, load_finetuned_checkpoint

logger = logging.getLogger(__name__)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Interactive inference with ICL models (MI355X build)")
    p.add_argument("--model_type", type=str, default="salmonn", help="Type of model to use")
    p.add_argument("--peft_model_path", type=str, default="", help="Path to the fine-tuned model ('' = none)")
    p.add_argument("--device", type=str, default="cuda" if torch.cuda.is_available() else "cpu")
    p.add_argument("--fp16", action="store_true", help="accepted for CLI compatibility (the path computes in bf16)")
    p.add_argument("--compile", action="store_true", help="accepted for CLI compatibility (ignored)")
    p.add_argument("--query", type=str, default="", help="Text query to process")
    p.add_argument("--max_new_tokens", type=int, default=100, help="Maximum number of tokens to generate")
    p.add_argument("--temperature", type=float, default=0.8, help="Temperature for sampling")
    p.add_argument("--arch", type=str, default=None, help="7b | 13b | tiny (default: inferred from llama_path)")
    p.add_argument("--seed", type=int, default=None, help="seed of the sampling generator (default: non-deterministic)")
    p.add_argument("--apply_generation_flags", action="store_true",
                   help="make --max_new_tokens / --temperature effective (the reference's collate_batch drops them: 10 greedy tokens)")
    return p.parse_args(argv)


def setup_model(args):
    config = get_inference_config(args.model_type)
    model_args = dict(config.get("model_args", {}))
    if getattr(args, "arch", None):
        model_args["arch"] = args.arch
    if str(args.device).startswith("cuda"):
        idx = int(str(args.device).split(":")[1]) if ":" in str(args.device) else 0
        args.device = f"cuda:{idx}"
        torch.cuda.set_device(idx)
        props = torch.cuda.get_device_properties(idx)
        logger.info("device %s: %s, %.1f GB", args.device, props.name, props.total_memory / 1e9)
    logger.info("Creating model of type %s", args.model_type)
    model = ModelFactory.create_model(model_type=args.model_type, multi_task=False, device=args.device, low_resource=True,
                                      **model_args)
    if args.peft_model_path and args.peft_model_path.strip():
        logger.info("Loading checkpoint from %s", args.peft_model_path)
        ckpt = torch.load(args.peft_model_path, map_location="cpu")
        n = load_finetuned_checkpoint(model, ckpt)
        logger.info("Updating %d keys from finetuned model", n)
    model.to(args.device)
    if args.model_type == "salmonn":
        processor = get_processor(args.model_type, model.input_processor, model.llama_tokenizer)
    else:
        processor = get_processor(args.model_type, model.input_processor)
    return model, processor


def run_interactive_inference(model, processor, query, args):
    logger.info("Processing query: %s", query)
    processed = processor.process_inputs(
        data={"prompt": query, "fewshot_mode": "text", "input_mode": "text_only", "completion": "", "audio": None,
              "examples_audio": None, "dataset_type": DatasetType.VOXCELEB},
        is_training=False)
    item = {"input_ids": processed["input_ids"].squeeze(0), "attention_mask": processed["attention_mask"].squeeze(0),
            "prompt": query, "text": query, "completion": "", "dataset_type": DatasetType.VOXCELEB,
            "max_new_tokens": args.max_new_tokens, "temperature": args.temperature, "do_sample": True, "num_examples": 0}
    batch = processor.collate_batch([item])
    batch = {k: (v.to(args.device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    if getattr(args, "apply_generation_flags", False):
        # generate_output reads the knobs from the batch dict (samples.get(...)); collate_batch has just dropped the item's copies
        batch["max_new_tokens"], batch["temperature"], batch["do_sample"] = args.max_new_tokens, args.temperature, True
        seed = getattr(args, "seed", None)
        if seed is not None:
            batch["generator"] = torch.Generator().manual_seed(int(seed))
    model.eval()
    with torch.no_grad():
        output = model.generate_output(batch)
    return output[0] if isinstance(output, list) else output


def main(argv=None) -> int:
    args = parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    try:
        model, processor = setup_model(args)
        query = args.query or input("Enter your query: ")
        print("\nModel output:")
        print(run_interactive_inference(model, processor, query, args))
        while True:
            try:
                query = input("\nEnter a new query (or 'exit' to quit): ")
            except EOFError:
                break
            if query.lower() in ("exit", "quit", "q"):
                break
            print("\nModel output:")
            print(run_interactive_inference(model, processor, query, args))
    except Exception as e:   # same contract as the reference: log, print the trace, exit code 1
        logger.error("Error: %s", e)
        traceback.print_exc()
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
