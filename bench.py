#!/usr/bin/env python3
"""bench.py — ICL utterances/s of the MI355X-native hot path (BASELINE.json metric).

One *step* = one pass of the whole hot path over one micro-batch of synthetic utterances:
K1 log-mel -> Whisper-large-v2 encoder ‖ BEATs -> LN/concat -> window Q-Former -> projector ->
prompt interleave -> Llama-2-7B prefill -> greedy decode to exactly 10 new tokens (EOS suppressed).
Workload at N=1: BASELINE.json configs[1] (C2: speech_only, 5 text exemplars, VOXCELEB;
prompt = 288 text tokens + 88 audio tokens = 376 positions).  Inputs (30 s / 16 kHz
waveforms and token ids) are resident in HBM before the timed region; weights are seeded random
N(0, 0.02^2) in bf16 (no checkpoints are reachable offline).

Multi-GPU: one process per GPU (torch.distributed over RCCL), utterances sharded by rank (weak scaling:
fixed per-GPU micro-batch); the only collective is one all-gather of the generated ids per step.

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` (dominant kernel: the
128x128 bf16 MFMA GEMM, timed live with HIP events on the launch stream) and, at N=1, `cpu_baseline`
(the fp32 CPU oracle on one utterance of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
S_TEXT, N_AUDIO_TOK, NEW_TOKENS = 288, 88, 10
SPEECH_AT = 280             # the <SpeechHere> slot sits near the end of the VOXCELEB prompt ("...\nOutput:")


def log(msg: str):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU cores this process may actually use: cgroup quota, else scheduler affinity, else cpu_count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, q // int(f2.read())))
            break
        except Exception:
            continue
    return n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128, help="utterances per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-profile", action="store_true")
    ap.add_argument("--no-graphs", action="store_true", help="run the decode loop eagerly (rocprofv3 --pmc crashes on HIP-graph capture)")
    ap.add_argument("--tiny", action="store_true", help="miniature model (smoke only; not a valid bench number)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c2s", "c4", "c5"],
                    help="BASELINE.md §4: c2 = headline (default); c2s = 5 speech exemplars; c4 = Qwen2-Audio HVB; "
                         "c5 = Llama2-13B VOXCELEB+HVB+VOXPOPULI round-robin")
    return ap.parse_args()


WORKLOADS = {   # name -> (description, text tokens per utterance (round-robin list), audios per utterance)
    "c2": ("C2: SALMONN (Whisper-large-v2 + BEATs + Llama2-7B) speech_only 5-shot VOXCELEB", [288], 1),
    "c2s": ("C2s: SALMONN-7B speech_only, 5 SPEECH exemplars, VOXCELEB", [200], 6),
    "c4": ("C4: Qwen2-Audio-7B speech_only 5-shot HVB", [514], 1),
    "c5": ("C5: SALMONN Llama2-13B, VOXCELEB+HVB+VOXPOPULI round-robin, 5 text exemplars", [288, 512, 320], 1),
}


def synth_utterances(first: int, count: int, vocab: int):
    """SURVEY.md §8d synthetic inputs: audio N(0,0.1^2) clipped, default_rng(1234+i); ids uniform [3,32000), default_rng(99+i)."""
    wav = np.empty((count, 480000), dtype=np.float32)
    ids = np.empty((count, S_TEXT), dtype=np.int64)
    for j in range(count):
        i = first + j
        wav[j] = np.clip(np.random.default_rng(1234 + i).normal(0.0, 0.1, 480000), -1.0, 1.0).astype(np.float32)
        ids[j] = np.random.default_rng(99 + i).integers(3, min(32000, vocab - 1), S_TEXT)
    return wav, ids


def synth_workload(first: int, count: int, vocab: int, text_lens, n_audio: int, audio_tokens: int):
    """Generalisation of synth_utterances for the non-headline workloads: per utterance `n_audio` clips and a prompt of
    text_lens[i % len] tokens with the audio slots spread through it (exemplars first, query slot 8 tokens before the end)."""
    from icl_speech_text_llm_amd.runtime.salmonn import speech_segment
    wav = np.empty((count * n_audio, 480000), dtype=np.float32)
    prompts = []
    for j in range(count):
        i = first + j
        for a in range(n_audio):
            wav[j * n_audio + a] = np.clip(np.random.default_rng(1234 + i * 7 + a).normal(0.0, 0.1, 480000), -1.0, 1.0)
        n = text_lens[i % len(text_lens)]
        ids = np.random.default_rng(99 + i).integers(3, min(32000, vocab - 1), n).tolist()
        cuts = [int((n - 8) * (k + 1) / n_audio) for k in range(n_audio)]
        segs, prev = [], 0
        for a, c in enumerate(cuts):
            segs.append(ids[prev:c])
            segs.append(speech_segment((j * n_audio + a) * audio_tokens, audio_tokens))
            prev = c
        segs.append(ids[prev:])
        prompts.append([sg for sg in segs if not (isinstance(sg, list) and not sg)])
    return wav, prompts


def build_prompts(ids: np.ndarray):
    from icl_speech_text_llm_amd.runtime.salmonn import speech_segment
    return [[row[:SPEECH_AT].tolist(), speech_segment(b * N_AUDIO_TOK, N_AUDIO_TOK), row[SPEECH_AT:].tolist()]
            for b, row in enumerate(ids)]


def cpu_baseline(cfg, sd_gpu, wav: np.ndarray, ids: np.ndarray, threads: int):
    """The oracle (fp32, batch 1, greedy 10 tokens) on ONE utterance of the same workload, host cores only."""
    from oracle import audio_frontend as af, models as om
    torch.set_num_threads(threads)
    log(f"cpu_baseline: copying {len(sd_gpu)} tensors to host fp32 ...")
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd_gpu.items()}
    log(f"cpu_baseline: running the fp32 oracle on 1 utterance with {threads} threads ...")
    t0 = time.perf_counter()
    spec = torch.from_numpy(af.whisper_logmel(wav))[None]
    emb = om.salmonn_encode_speech(sd, spec, torch.from_numpy(wav)[None], [wav.shape[0]], cfg.whisper.n_heads,
                                   use_beats=cfg.beats is not None,
                                   beats_cfg=dict(n_heads=cfg.beats.n_heads, num_buckets=cfg.beats.num_buckets,
                                                  max_distance=cfg.beats.max_distance) if cfg.beats else None,
                                   qformer_heads=cfg.qformer.n_heads)
    lsd = {k[len("llama_model."):]: v for k, v in sd.items() if k.startswith("llama_model.")}
    llm = om.LlamaOracle(lsd, cfg.llama.n_heads, cfg.llama.rms_eps, cfg.llama.rope_theta, cfg.llama.lora_scale)
    x = torch.cat([llm.embed(torch.from_numpy(ids[:SPEECH_AT])), emb[0], llm.embed(torch.from_numpy(ids[SPEECH_AT:]))])[None]
    log(f"cpu_baseline: speech encoders done at {time.perf_counter() - t0:.1f} s; Llama prefill + decode ...")
    out, first = llm.generate_greedy(x, NEW_TOKENS, eos_id=-1, pad_id=cfg.llama.pad_id, return_first_logits=True)
    dt = time.perf_counter() - t0
    return dt, out[0].tolist(), first[0]


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # ICL_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend — lets the N > 1 code path (sharding, gather, max-over-
    # ranks timing) be exercised on a one-GPU box; never used for reported numbers
    rehearsal = os.environ.get("ICL_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from icl_speech_text_llm_amd.runtime import binding as B, synth
    from icl_speech_text_llm_amd.runtime.config import SalmonnCfg
    from icl_speech_text_llm_amd.runtime.salmonn import SalmonnRuntime

    wl_desc, wl_text, wl_naudio = WORKLOADS[args.workload]
    is_qwen = args.workload == "c4"
    if is_qwen:
        from icl_speech_text_llm_amd.runtime.config import QwenAudioCfg
        from icl_speech_text_llm_amd.runtime.qwen import QwenAudioRuntime
        cfg = QwenAudioCfg.tiny() if args.tiny else QwenAudioCfg()
    else:
        cfg = SalmonnCfg.tiny() if args.tiny else (SalmonnCfg.llama2_13b() if args.workload == "c5" else SalmonnCfg.llama2_7b())
    log(f"building {'tiny ' if args.tiny else ''}{wl_desc} synthetic weights on {dev} ...")
    t_build = time.perf_counter()
    if is_qwen:
        sd = synth.qwen_audio_state(cfg, seed=0, device=dev, dtype=torch.bfloat16)
        rt = QwenAudioRuntime(cfg, dict(sd), device=dev)
    else:
        sd = synth.salmonn_state(cfg, seed=0, device=dev, dtype=torch.bfloat16)   # identical replica on every rank
        rt = SalmonnRuntime(cfg, dict(sd), device=dev)
    if args.no_graphs:
        rt.use_graphs = False
    want_cpu = (world == 1 and rank == 0 and not args.no_cpu_baseline and args.workload == "c2")
    if not want_cpu:
        del sd
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    log(f"model packed in {t_build:.1f} s ({torch.cuda.memory_allocated(dev) / 2**30:.1f} GiB allocated)")

    # ---- inputs, resident in HBM before the timed region ----------------------------------------
    Bm, total_steps = args.batch, args.warmup + args.steps
    vocab = cfg.llm.vocab if is_qwen else cfg.llama.vocab
    audio_tokens = 750 if is_qwen else N_AUDIO_TOK
    wavs, prompts = [], []
    lens = [480000] * (Bm * wl_naudio)
    for s in range(total_steps):
        first = (s * world + rank) * Bm           # utterance i belongs to rank (i // Bm) % world of step i // (Bm*world)
        if args.workload == "c2":
            w, ids = synth_utterances(first, Bm, vocab)
            prompts.append(build_prompts(ids))
            if s == 0:
                w0, ids0 = w[0].copy(), ids[0].copy()
        else:
            w, pr = synth_workload(first, Bm, vocab, wl_text, wl_naudio, audio_tokens)
            prompts.append(pr)
        wavs.append(torch.from_numpy(w).to(dev))
    gathered = torch.empty(world * Bm, NEW_TOKENS, dtype=torch.int32, device=dev) if world > 1 else None

    def step(s, gather=True):
        if is_qwen:
            speech, _ = rt.encode_audio(raw_wav=wavs[s], wav_lens=lens)
        else:
            speech = rt.encode_speech(wavs[s], lens)
        res = rt.generate(prompts[s], speech, max_new_tokens=NEW_TOKENS, suppress_eos=True)
        if world > 1 and gather:
            dist.all_gather_into_tensor(gathered, rt.ws.get("gen_tokens", (Bm, NEW_TOKENS), torch.int32))
        return res.tokens

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"inputs resident ({total_steps} x {Bm} utterances); warmup ...")
    first_tokens = None
    for s in range(args.warmup):
        toks = step(s)
        torch.cuda.synchronize()
        log(f"warmup step {s} done")
        if s == 0:
            first_tokens = toks[0].tolist()
    profile = None if args.no_gemm_profile else []
    barrier()
    B.GEMM_PROFILE = profile
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        toks = step(s)
        if first_tokens is None and s == 0:
            first_tokens = toks[0].tolist()
    barrier()
    elapsed = time.perf_counter() - t0
    B.GEMM_PROFILE = None
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (128x128 MFMA GEMM), from live HIP-event timings -------
    roof = None
    if profile:
        names = {1: "gemm_bf16_kernel<2,2,4,4> (128x128x64 tile)", 2: "gemm_bf16_kernel<2,2,2,2> (64x64x64 tile)",
                 3: "gemm256_bf16_kernel (256x256x64 rolling LDS-DMA pipeline)"}
        per_tile = {}
        for (tile, sk, f, e0, e1, shape) in profile:
            fl, tt, n = per_tile.get(tile, (0.0, 0.0, 0))
            per_tile[tile] = (fl + f, tt + e0.elapsed_time(e1) * 1e-3, n + 1)
        all_t = sum(v[1] for v in per_tile.values())
        dom = max(per_tile, key=lambda k: per_tile[k][1])          # dominant kernel = largest share of GPU time
        fl, tt, n = per_tile[dom]
        traffic = None
        try:   # HBM-side bytes per launch of this kernel from the committed PMC passes of this same command (profiles/)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            key = {1: "gemm_bf16_kernel<2, 2, 4, 4>", 2: "gemm_bf16_kernel<2, 2, 2, 2>", 3: "gemm256_bf16_kernel"}[dom]
            if f"(batch {Bm}" in pmc["command"] and not args.tiny:
                rows = [v for k, v in pmc["kernels"].items() if k == key or k.startswith(key + "<")]   # all instantiations
                traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rows) / sum(v["launches"] for v in rows))
        except Exception:
            traffic = None
        roof = {"bound": "mfma", "kernel": names.get(dom, str(dom)), "achieved": round(fl / tt / 1e12, 1),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / tt / 1e12 / PEAK_BF16_TFLOPS, 4),
                "traffic": traffic, "traffic_note": "bytes/launch = 2*FETCH_SIZE + WRITE_SIZE (KiB) from separate rocprofv3 --pmc "
                "passes of this command (profiles/r01_pmc_traffic.json); the counters sit at the L2<->fabric boundary and "
                "include Infinity-Cache hits" if traffic else None, "launches": n, "avg_launch_us": round(tt / n * 1e6, 2),
                "avg_launch_gflop": round(fl / n / 1e9, 3), "share_of_step_time": round(tt / elapsed, 3),
                "all_gemm_share_of_step_time": round(all_t / elapsed, 3),
                "other_gemm_kernels": {names.get(k, str(k)): {"achieved_tflops": round(v[0] / v[1] / 1e12, 1),
                                                             "share_of_step_time": round(v[1] / elapsed, 3), "launches": v[2]}
                                       for k, v in per_tile.items() if k != dom}}

    if profile and os.environ.get("ICL_BENCH_GEMM_SHAPES"):
        by_shape = {}
        for (tile, sk, f, e0, e1, shape) in profile:
            fl, tt, n = by_shape.get((tile, sk) + shape, (0.0, 0.0, 0))
            by_shape[(tile, sk) + shape] = (fl + f, tt + e0.elapsed_time(e1) * 1e-3, n + 1)
        for k, (fl, tt, n) in sorted(by_shape.items(), key=lambda kv: -kv[1][1])[:40]:
            log(f"gemm tile={k[0]} sk={k[1]} M={k[2]} N={k[3]} K={k[4]} batch={k[5]}: {n} launches, {tt / n * 1e6:9.1f} us avg, "
                f"{fl / tt / 1e12:7.1f} TF/s, {100 * tt / elapsed:5.1f}% of step time")
    # ---- per-phase timing: ONE extra, untimed step with HIP events at the phase boundaries -----------------
    phases = None
    if rank == 0 and args.workload == "c2" and not args.tiny:
        marks = {k: torch.cuda.Event(enable_timing=True) for k in ("enc_start", "prefill_start", "prefill_end", "decode_end")}
        rt.phase_marks = marks
        marks["enc_start"].record()
        step(total_steps - 1, gather=False)      # rank 0 only: no collective in here
        torch.cuda.synchronize()
        rt.phase_marks = None
        t_enc = marks["enc_start"].elapsed_time(marks["prefill_start"]) * 1e-3      # front-ends, encoders, Q-Former, prompt gather
        t_pre = marks["prefill_start"].elapsed_time(marks["prefill_end"]) * 1e-3
        t_dec = marks["prefill_end"].elapsed_time(marks["decode_end"]) * 1e-3
        S = wl_text[0] + wl_naudio * audio_tokens
        kv_bytes = sum(0.524288e6 * (S + t) for t in range(1, NEW_TOKENS)) * Bm       # K+V bf16, 32 layers x 4096, re-read per step
        phases = {
            "encoder": {"ms": round(t_enc * 1e3, 1), "tflop": round(2.647 * Bm, 1),
                        "mfma_frac": round(2.647e12 * Bm / t_enc / 2.5e15, 3)},
            "prefill": {"ms": round(t_pre * 1e3, 1), "tflop": round(4.907 * Bm, 1),
                        "mfma_frac": round(4.907e12 * Bm / t_pre / 2.5e15, 3)},
            "decode": {"ms": round(t_dec * 1e3, 1), "steps": NEW_TOKENS - 1,
                       "hbm_gb": round((13.48e9 * (NEW_TOKENS - 1) + kv_bytes) / 1e9, 1),
                       "hbm_frac": round((13.48e9 * (NEW_TOKENS - 1) + kv_bytes) / t_dec / 8.0e12, 3)},
            "note": "one untimed step after the timed region, HIP events at the phase boundaries; FLOPs / bytes per utterance from "
                    "SURVEY.md §8d (encoder 2.647 TFLOP, prefill 4.907 TFLOP at 376 positions, decode = 13.48 GB of weights per step "
                    "+ the K/V of every sequence); peaks 2.5 PFLOP/s bf16 dense and 8 TB/s",
        }
        log(f"phases: encoder {t_enc * 1e3:.1f} ms, prefill {t_pre * 1e3:.1f} ms, decode {t_dec * 1e3:.1f} ms")
    if rank == 0:
        n_utt = Bm * args.steps * world
        out = {
            "metric": "ICL utterances/sec (whole node), SALMONN+Llama2-7B 5-shot VOXCELEB" if args.workload == "c2" else
                      f"ICL utterances/sec (whole node), workload {args.workload}",
            "value": round(n_utt / elapsed, 3), "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl_desc if not args.tiny else "TINY smoke model (not a benchmark)",
                       "utterances_per_step_per_gpu": Bm,
                       "prompt_positions": [t + wl_naudio * audio_tokens for t in wl_text] if len(wl_text) > 1 else wl_text[0] + wl_naudio * audio_tokens,
                       "audio_seconds": 30, "new_tokens": NEW_TOKENS, "parallelism": f"dp{world}",
                       "weights": "seeded N(0,0.02^2) bf16, LoRA r=8 un-merged"},
            "roofline": roof, "phases": phases,
            "build_s": round(t_build, 1), "first_utterance_tokens": first_tokens,
        }
        if want_cpu:
            threads = host_cores()
            dt, cpu_tokens, cpu_first = cpu_baseline(cfg, sd, w0, ids0, threads)
            # full-size numerical check of the same utterance: GPU first-step logits vs the fp32 CPU oracle
            sp0 = rt.encode_speech(torch.from_numpy(w0)[None], [480000])
            g0 = rt.generate(build_prompts(ids0[None]), sp0, max_new_tokens=1, suppress_eos=True, want_first_logits=True)
            diff = (g0.first_logits[0].cpu() - cpu_first).abs()
            top2 = cpu_first.topk(2).values
            out["cpu_baseline"] = {"value": round(1.0 / dt, 5), "unit": "utterances/s", "cores": threads, "kind": "port",
                                   "sample": f"1 utterance of the same C2 workload (30 s audio, 376 positions, 10 greedy "
                                             f"tokens), fp32 torch-CPU oracle, {dt:.1f} s",
                                   "tokens": cpu_tokens, "tokens_match_gpu": cpu_tokens == first_tokens,
                                   "first_logits_max_abs_diff_gpu_vs_cpu": round(float(diff.max()), 5),
                                   "first_logits_rel_l2_diff": round(float(diff.norm() / cpu_first.norm()), 5),
                                   "cpu_logit_abs_max": round(float(cpu_first.abs().max()), 4),
                                   "cpu_top1_margin": round(float(top2[0] - top2[1]), 5)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
