#!/usr/bin/env python3
"""bench.py — ICL utterances/s of the MI355X-native hot path (BASELINE.json metric).

One *step* = one pass of the whole hot path over one micro-batch of synthetic utterances:
K1 log-mel -> Whisper-large-v2 encoder ‖ BEATs -> LN/concat -> window Q-Former -> projector ->
prompt interleave -> Llama-2-7B prefill -> greedy decode to exactly 10 new tokens (EOS suppressed).
Workload at N=1: BASELINE.json configs[1] (C2: speech_only, 5 text exemplars, VOXCELEB;
prompt = 288 text tokens + 88 audio tokens = 376 positions).  Inputs (30 s / 16 kHz
waveforms and token ids) are resident in HBM before the timed region; weights are seeded random
N(0, 0.02^2) in bf16 (no checkpoints are reachable offline).

Multi-GPU: one process per GPU over RCCL, utterances sharded by rank (weak scaling: fixed per-GPU
micro-batch), ONE fixed-shape all-gather per step carrying (utterance index, generated ids, generated length, first-step
logits bf16 [b, V]) of every rank.  `python bench.py --gpus N` (N > 1, not already under a launcher) starts the N ranks
itself through `python -m torch.distributed.run` as a CHILD process before anything touches the GPU and relays its output
and exit code; under an external launcher (WORLD_SIZE set) it is simply one rank.

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` (dominant kernel: the 256x256 bf16 MFMA
GEMM, timed live with HIP events on the launch stream inside the timed region) and, at N=1, `cpu_baseline` (the fp32 CPU
oracle on one utterance of the same workload) and `parity` (the same utterance at FULL size against the oracle with the
bf16 rounding hook, stage by stage, all 10 greedy decisions teacher-forced; plus a decisive-margin weight set on which
the ids must match exactly).  A parity bound exceeded makes the process exit 4 after printing the line.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
S_TEXT, N_AUDIO_TOK, NEW_TOKENS = 288, 88, 10
SPEECH_AT = 280             # the <SpeechHere> slot sits near the end of the VOXCELEB prompt ("...\nOutput:")

# full-size parity bounds (relative L2 vs the oracle with the bf16 rounding hook) = ~2x the values measured on MI355X
# (profiles/r02_bench_default.json); the north star's 1e-3 is a per-kernel figure, the chains below stack 32-64 layers
# Measured (r2, utterance 0): log-mel 4.7e-8, Whisper 4.2e-3, BEATs 1.4e-3, speech embeddings 2.9e-3, margin-weight logits 2.8e-3.
# Under the frozen N(0, 0.02^2) weights the 32-layer decoder amplifies rounding noise: the oracle's OWN two precisions
# (bf16 hook vs fp32) differ by 3.3e-2 in the first-step logits, the HIP path sits 2.7e-2 from the bf16 oracle — hence the
# loose decoder bounds on those weights and the tight one on the well-conditioned margin set.
PARITY_RATIO = 1.25     # where a pure-fp32 oracle run exists: rel(gpu, fp32) <= PARITY_RATIO * rel(bf16-rounding oracle, fp32) — the
                        # HIP path may not sit further from the exact arithmetic than the reference's own bf16 rounding points do
PARITY_BOUNDS = {"logmel": 1e-6, "whisper": 8e-3, "beats": 3e-3, "encode_speech": 6e-3, "prefill_last_hidden": 6e-2,
                 "first_step_logits": 6e-2, "decode_step_logits": 6e-2, "margin_step_logits": 6e-3}


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
    ap.add_argument("--batch", type=int, default=256, help="utterances per step per GPU (prefilled 128 at a time, decoded together)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle legs (cpu_baseline + parity)")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="BASELINE.md §3 in full: 16 utterances at 8 threads and at all cores (median), plus config 1 "
                         "(text_only, 160 positions); ~10+ minutes of host time, result also written to --cpu-baseline-out")
    ap.add_argument("--cpu-baseline-out", default=None)
    ap.add_argument("--no-gemm-profile", action="store_true", help="no per-launch HIP events in the timed region (roofline = null)")
    ap.add_argument("--no-graphs", action="store_true", help="run the decode loop eagerly (rocprofv3 --pmc crashes on HIP-graph capture)")
    ap.add_argument("--no-phases", action="store_true", help="skip the extra untimed per-phase step (profiling passes: fewer dispatches)")
    ap.add_argument("--tiny", action="store_true", help="miniature model (smoke only; not a valid bench number)")
    ap.add_argument("--through-plugin", action="store_true",
                    help="(default on the headline workload, at every N: with N > 1 each rank runs it on its own shard) also time the path through the reference-compatible plugin: "
                         "ModelFactory -> SalmonProcessor / DataLoader -> generate_output (H2D, tokenisation, batch_decode inside "
                         "the timed region; SURVEY.md §8d)")
    ap.add_argument("--no-through-plugin", action="store_true", help="skip the plugin-path leg")
    ap.add_argument("--plugin-workers", type=int, default=8, help="item-pipeline workers of the plugin-path leg (capped at cores per rank - 1)")
    ap.add_argument("--plugin-rows", type=int, default=0, help="distinct clips in the on-disk dataset of the plugin-path leg (default min(256, batch))")
    ap.add_argument("--plugin-batch", type=int, default=None, help="batch size of the plugin-path leg (default: --batch, the micro-batch of the runtime number)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the BASELINE.json configs 4 and 5 (1-GPU halves) that the default N = 1 run appends after the headline legs")
    ap.add_argument("--other-batch", type=int, default=128, help="micro-batch of the configs 4 / 5 legs")
    ap.add_argument("--workload", default="c2", choices=["c2", "c2s", "c4", "c5"],
                    help="BASELINE.md §4: c2 = headline (default); c2s = 5 speech exemplars; c4 = Qwen2-Audio HVB; "
                         "c5 = Llama2-13B VOXCELEB+HVB+VOXPOPULI round-robin")
    return ap.parse_args()


def launch_ranks(args) -> None:
    """`--gpus N` outside a launcher: start N ranks as a child `torch.distributed.run` and exit with its code.  Runs before
    any HIP call of this process (device_count() does not initialise the runtime on this image) — a process that has touched
    the GPU must never be replaced or forked into ranks."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if os.environ.get("ICL_BENCH_REHEARSAL") != "1":
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            print(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible", file=sys.stderr)
            sys.exit(2)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log(f"launching {args.gpus} ranks: {' '.join(cmd)}")
    sys.exit(subprocess.run(cmd, env=env).returncode)       # ranks inherit stdout: rank 0's JSON line passes straight through


WORKLOADS = {   # name -> (description, text tokens per utterance (round-robin list), audios per utterance)
    "c2": ("C2: SALMONN (Whisper-large-v2 + BEATs + Llama2-7B) speech_only 5-shot VOXCELEB", [288], 1),
    "c2s": ("C2s: SALMONN-7B speech_only, 5 SPEECH exemplars, VOXCELEB", [200], 6),
    "c4": ("C4: Qwen2-Audio-7B speech_only 5-shot HVB", [514], 1),
    "c5": ("C5: SALMONN Llama2-13B, VOXCELEB+HVB+VOXPOPULI round-robin, 5 text exemplars", [288, 512, 320], 1),
}


GEMM_KERNEL_NAMES = {1: "gemm_bf16_kernel<2,2,4,4> (128x128x64 tile)", 2: "gemm_bf16_kernel<2,2,2,2> (64x64x64 tile)",
                     3: "gemm256_bf16_kernel (256x256x64 rolling LDS-DMA pipeline)", 4: "gemm_skinny_kernel (decode, <= 8 rows)",
                     5: "gemm_m128_kernel (decode, 9..256 rows, decode-packed weights)", 6: "gemm_skinny_kernel (decode-packed weights)"}
TFLOP_PER_UTT = {"c2": 7.68, "c2s": 25.6, "c4": 19.2, "c5": 14.7}      # SURVEY.md §8d / BASELINE.md §4


def gemm_event_summary(profile, elapsed: float):
    """Live HIP-event timings of every GEMM launch of the timed region -> (dominant tile id, {tile: (flops, seconds, launches)})."""
    per_tile = {}
    for (tile, sk, f, e0, e1, shape) in profile:
        fl, tt, n = per_tile.get(tile, (0.0, 0.0, 0))
        per_tile[tile] = (fl + f, tt + e0.elapsed_time(e1) * 1e-3, n + 1)
    dom = max(per_tile, key=lambda k: per_tile[k][1])          # dominant kernel = largest share of GPU time
    return dom, per_tile


def workload_roofline(wl: str, utt_per_s: float, profile, elapsed: float):
    """The per-config roofline block of the non-headline workloads: whole-utterance MFMA fraction from the §8d FLOP count and
    the dominant GEMM kernel's own fraction from the live events of the timed steps."""
    tf = TFLOP_PER_UTT[wl]
    out = {"bound": "mfma", "tflop_per_utterance": tf, "achieved": round(utt_per_s * tf, 1), "peak": PEAK_BF16_TFLOPS,
           "unit": "TFLOP/s", "frac": round(utt_per_s * tf / PEAK_BF16_TFLOPS, 4), "mfma_roof_utt_per_s": round(PEAK_BF16_TFLOPS / tf, 1)}
    if profile:
        dom, per_tile = gemm_event_summary(profile, elapsed)
        fl, tt, n = per_tile[dom]
        out["dominant_kernel"] = {"kernel": GEMM_KERNEL_NAMES.get(dom, str(dom)), "achieved": round(fl / tt / 1e12, 1),
                                  "frac": round(fl / tt / 1e12 / PEAK_BF16_TFLOPS, 4), "launches": n,
                                  "avg_launch_us": round(tt / n * 1e6, 2), "avg_launch_gflop": round(fl / n / 1e9, 3),
                                  "share_of_step_time": round(tt / elapsed, 3)}
    return out


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


# ======================================================================================================================
# CPU legs (rank 0, N = 1 only): the oracle is the CHECKER and the reported baseline, never the thing measured above
# ======================================================================================================================
def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _maxabs(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.detach().float().cpu().reshape(-1) - b.detach().float().cpu().reshape(-1)).abs().max())


def _oracle_speech(sd, cfg, wav: np.ndarray, rnd):
    """Stage outputs of the oracle for one utterance: log-mel, Whisper out, BEATs out, speech embeddings [88, H]."""
    from oracle import audio_frontend as af, models as om
    n = wav.shape[0]
    spec = torch.from_numpy(af.whisper_logmel(wav))[None]
    wh = om.whisper_encoder(sd, spec, cfg.whisper.n_heads, "speech_encoder.", rnd=rnd)
    be = None
    if cfg.beats is not None:
        be, _ = om.beats_encoder(sd, torch.from_numpy(wav)[None], [n], prefix="beats.", n_heads=cfg.beats.n_heads,
                                 num_buckets=cfg.beats.num_buckets, max_distance=cfg.beats.max_distance, rnd=rnd)
    emb = om.salmonn_fuse_qformer(sd, wh, be, rnd=rnd, qformer_heads=cfg.qformer.n_heads)
    return spec[0], wh[0], (be[0] if be is not None else None), emb[0]


def _oracle_llm(sd, cfg, rnd, prefix="llama_model."):
    from oracle import models as om
    lsd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    return om.LlamaOracle(lsd, cfg.llama.n_heads, cfg.llama.rms_eps, cfg.llama.rope_theta, cfg.llama.lora_scale, rnd=rnd)


def _to_host_f32(sd_gpu):
    return {k: v.detach().to("cpu", torch.float32) for k, v in sd_gpu.items()}


def cpu_utterance(sd, cfg, wav: np.ndarray, ids: np.ndarray, threads: int):
    """The fp32 oracle (batch 1, greedy 10 tokens) on ONE C2 utterance; returns (seconds, tokens, first logits, stage outputs)."""
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    spec, wh, be, emb = _oracle_speech(sd, cfg, wav, None)
    t_enc = time.perf_counter() - t0
    llm = _oracle_llm(sd, cfg, None)
    x = torch.cat([llm.embed(torch.from_numpy(ids[:SPEECH_AT])), emb, llm.embed(torch.from_numpy(ids[SPEECH_AT:]))])[None]
    out, first = llm.generate_greedy(x, NEW_TOKENS, eos_id=-1, pad_id=cfg.llama.pad_id, return_first_logits=True)
    dt = time.perf_counter() - t0
    return dt, t_enc, out[0].tolist(), first[0], (spec, wh, be, emb)


def cpu_text_only(sd, cfg, n_tokens: int, threads: int, utt: int = 0):
    """Config 1 (BASELINE.json configs[0]): 0-shot text_only — no audio is encoded (data/model_processors.py:638), the prompt
    is `n_tokens` text positions through Llama-2-7B, batch 1, 10 greedy tokens.  Returns seconds."""
    torch.set_num_threads(threads)
    ids = np.random.default_rng(99 + utt).integers(3, min(32000, cfg.llama.vocab - 1), n_tokens)
    t0 = time.perf_counter()
    llm = _oracle_llm(sd, cfg, None)
    llm.generate_greedy(llm.embed(torch.from_numpy(ids))[None], NEW_TOKENS, eos_id=-1, pad_id=cfg.llama.pad_id)
    return time.perf_counter() - t0


def full_size_parity(cfg, sd_host, rt, dev, wav: np.ndarray, ids: np.ndarray, fp32_stages=None, fp32_first=None):
    """The SAME utterance through the HIP path and through the oracle with the bf16 rounding hook at FULL model size, stage by
    stage (contract: the reference's llama_model(...) / .generate(...) calls, models/custom_salmon.py:630-640,
    704-731).  All 10 greedy decisions are checked by teacher-forcing the oracle along the GPU's tokens."""
    from icl_speech_text_llm_amd.runtime.engines import _i32
    from oracle import models as om
    n = wav.shape[0]
    wav_d = torch.from_numpy(wav)[None].to(dev)
    xt, spec_g = rt.logmel(rt.ws, wav_d, _i32([n], dev), want_spec=True)
    spec_g = spec_g[0].clone()
    wh_g = rt.whisper.forward(rt.ws, xt).clone()
    be_g, cu, _ = rt.beats.forward(rt.ws, wav_d, [n], [n])
    be_g = be_g.clone()
    emb_g = rt.qformer.forward(rt.ws, wh_g, 1, be_g, cu).clone()
    gen = rt.generate(build_prompts(ids[None]), emb_g.view(1, N_AUDIO_TOK, -1), max_new_tokens=NEW_TOKENS, suppress_eos=True,
                      want_step_logits=True)
    last_g = rt.ws.get("gen_last", (1, cfg.llama.hidden), torch.float32).clone()
    toks = gen.tokens[0]
    step_g = gen.step_logits[:, 0].cpu()

    rnd = om.bf16_round_activations(sd_host)          # weights are bf16-exact already: round activations only
    t0 = time.perf_counter()
    spec_b, wh_b, be_b, emb_b = _oracle_speech(sd_host, cfg, wav, rnd)
    llm = _oracle_llm(sd_host, cfg, rnd)
    x = torch.cat([llm.embed(torch.from_numpy(ids[:SPEECH_AT])), emb_b, llm.embed(torch.from_numpy(ids[SPEECH_AT:]))])[None]
    cache: list = []
    T = x.shape[1]
    h = llm.forward_hidden(x.float(), torch.arange(T)[None], cache)
    last_b = h[0, -1]
    tf = [llm.logits(h[:, -1:])[0, 0]]
    for t in range(NEW_TOKENS - 1):
        e = llm.embed(toks[t:t + 1])[:, None]
        tf.append(llm.logits(llm.forward_hidden(e, torch.full((1, 1), T + t), cache))[0, 0])
    tf = torch.stack(tf)
    t_oracle = time.perf_counter() - t0
    stages = {
        "logmel": {"rel_l2": _rel(spec_g, spec_b), "max_abs": _maxabs(spec_g, spec_b)},
        "whisper": {"rel_l2": _rel(wh_g, wh_b)},
        "beats": {"rel_l2": _rel(be_g, be_b)},
        "encode_speech": {"rel_l2": _rel(emb_g, emb_b)},
        "prefill_last_hidden": {"rel_l2": _rel(last_g[0], last_b)},
        "first_step_logits": {"rel_l2": _rel(step_g[0], tf[0]), "max_abs": _maxabs(step_g[0], tf[0])},
    }
    if fp32_stages is not None:      # the same GPU outputs against the pure-fp32 oracle (the reference's CPU behaviour)
        _, f_wh, f_be, f_emb = fp32_stages
        for name, g_, b_, f_ in (("whisper", wh_g, wh_b, f_wh), ("beats", be_g, be_b, f_be), ("encode_speech", emb_g, emb_b, f_emb)):
            stages[name]["rel_l2_vs_fp32"] = _rel(g_, f_)
            stages[name]["bf16_oracle_vs_fp32_oracle_rel_l2"] = _rel(b_, f_)
        stages["first_step_logits"].update(rel_l2_vs_fp32=_rel(step_g[0], fp32_first), max_abs_vs_fp32=_maxabs(step_g[0], fp32_first),
                                           bf16_oracle_vs_fp32_oracle_rel_l2=_rel(tf[0], fp32_first))
    ratio_ok = True
    for name, v in stages.items():
        if "bf16_oracle_vs_fp32_oracle_rel_l2" in v:
            v["ratio_to_bf16_oracle_distance"] = v["rel_l2_vs_fp32"] / max(v["bf16_oracle_vs_fp32_oracle_rel_l2"], 1e-30)
            ratio_ok = ratio_ok and v["ratio_to_bf16_oracle_distance"] <= PARITY_RATIO
    rels, errs, margins, within, exact = [], [], [], 0, 0
    for t in range(NEW_TOKENS):
        err = _maxabs(step_g[t], tf[t])
        top2 = tf[t].topk(2)
        rels.append(_rel(step_g[t], tf[t])); errs.append(err); margins.append(float(top2.values[0] - top2.values[1]))
        within += int(float(top2.values[0] - tf[t, int(toks[t])]) <= 2 * err + 1e-6)
        exact += int(int(toks[t]) == int(top2.indices[0]))
    decode = {"rel_l2_max": max(rels), "max_abs_max": max(errs), "oracle_top1_margin_min": min(margins),
              "gpu_choice_is_oracle_argmax_within_2x_logit_error": f"{within}/{NEW_TOKENS}",
              "gpu_choice_equals_oracle_argmax": f"{exact}/{NEW_TOKENS}", "gpu_tokens": toks.tolist()}
    ok = within == NEW_TOKENS and ratio_ok
    for k, b in PARITY_BOUNDS.items():
        if k in stages:
            ok = ok and stages[k]["rel_l2"] <= b
    ok = ok and decode["rel_l2_max"] <= PARITY_BOUNDS["decode_step_logits"]
    for v in stages.values():
        for k in v:
            v[k] = float(f"{v[k]:.3e}")
    for k in ("rel_l2_max", "max_abs_max", "oracle_top1_margin_min"):
        decode[k] = float(f"{decode[k]:.3e}")
    return {"stages": stages, "decode_steps_teacher_forced": decode, "oracle_seconds": round(t_oracle, 1),
            "ratio_criterion": f"rel(gpu, fp32) <= {PARITY_RATIO} x rel(bf16 oracle, fp32) wherever an fp32 oracle run exists",
            "ratio_ok": bool(ratio_ok), "ok": bool(ok)}


def margin_parity(cfg, dev, ids: np.ndarray, speech_emb: torch.Tensor):
    """Token exactness where it is decidable: a Llama-2-7B-size decoder with decisive arg-max margins (synth margin=True) on the
    same prompt layout; the GPU's 10 greedy ids must EQUAL the bf16-rounding oracle's, and the oracle's own free-running ids
    must equal the designed successor chain."""
    from icl_speech_text_llm_amd.runtime import synth
    from icl_speech_text_llm_amd.runtime.salmonn import SalmonnRuntime
    from oracle import models as om
    msd = synth.salmonn_state(cfg, seed=1, device=dev, dtype=torch.bfloat16, parts=("llama",), margin=True)
    mrt = SalmonnRuntime(cfg, dict(msd), device=dev, parts=("llama",))
    gen = mrt.generate(build_prompts(ids[None]), speech_emb.view(1, N_AUDIO_TOK, -1), max_new_tokens=NEW_TOKENS,
                       suppress_eos=True, want_step_logits=True)
    toks, step_g = gen.tokens[0], gen.step_logits[:, 0].cpu()
    del mrt
    host = _to_host_f32(msd)
    del msd
    torch.cuda.empty_cache()
    chain, t = [], int(ids[-1])
    for _ in range(NEW_TOKENS):
        t = synth.margin_successor(host, t); chain.append(t)
    llm = _oracle_llm(host, cfg, om.bf16_round_activations(host))
    x = torch.cat([llm.embed(torch.from_numpy(ids[:SPEECH_AT])), speech_emb.float().cpu(),
                   llm.embed(torch.from_numpy(ids[SPEECH_AT:]))])[None]
    tf = llm.teacher_forced_logits(x, toks[None])[0]
    oracle_ids = tf.argmax(-1).tolist()
    rels = [_rel(step_g[t], tf[t]) for t in range(NEW_TOKENS)]
    errs = [_maxabs(step_g[t], tf[t]) for t in range(NEW_TOKENS)]
    margins = [float(v[0] - v[1]) for v in (tf[t].topk(2).values for t in range(NEW_TOKENS))]
    match = toks.tolist() == oracle_ids
    return {"tokens_match": bool(match), "gpu_tokens": toks.tolist(), "oracle_tokens": oracle_ids,
            "designed_successor_chain_matches": bool(chain == oracle_ids),
            "oracle_top1_margin_min": float(f"{min(margins):.3e}"), "step_logits_max_abs_err": float(f"{max(errs):.3e}"),
            "step_logits_rel_l2_max": float(f"{max(rels):.3e}"),
            "ok": bool(match and max(rels) <= PARITY_BOUNDS["margin_step_logits"])}


def cpu_baseline_full(sd, cfg, wavs, idss, threads_all: int):
    """BASELINE.md §3 / SURVEY.md §8d in full: 16 C2 utterances at 8 threads (the reference's own cap,
    utils/performance_utils.py:323-324) and at all host cores, median seconds per utterance; config 1 timed once per setting."""
    out = {}
    for thr in sorted({8, threads_all}):
        if thr > threads_all:
            continue
        times = []
        for i in range(len(wavs)):
            dt, _, _, _, _ = cpu_utterance(sd, cfg, wavs[i], idss[i], thr)
            times.append(dt)
            log(f"cpu_baseline_full: threads {thr} utterance {i}: {dt:.1f} s")
        c1 = cpu_text_only(sd, cfg, 160, thr)
        out[f"threads_{thr}"] = {"c2_utterances": len(times), "c2_seconds_median": round(float(np.median(times)), 2),
                                 "c2_seconds_min": round(min(times), 2), "c2_seconds_max": round(max(times), 2),
                                 "c2_utt_per_s": round(1.0 / float(np.median(times)), 5),
                                 "c1_text_only_160_seconds": round(c1, 2), "c1_utt_per_s": round(1.0 / c1, 4)}
    return out


# ======================================================================================================================
# the plugin-path number ("dataloader/log-mel included")
# ======================================================================================================================
def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else None


def through_plugin(args, dev, dist=None, rank: int = 0, world: int = 1, n_batches=None, warm: int = 2, make_model=None,
                   workers=None, gpu_rate=None):
    """ModelFactory.create_model -> on-disk HF folder -> load_dataset -> DatasetFactory / InferenceDataset -> SalmonProcessor ->
    ArenaBatchLoader(num_workers) -> model.generate_ids / decode_ids, timed the way the reference's loop counts examples
    (inference/inference.py:259-266,301-368; utils/performance_utils.py:96-122): the item pipeline (Arrow row -> few-shot prompt
    -> tokenizer -> float32 waveform), collation, H2D of the raw waveforms, the host prompt split + tokenisation, K1..K11 and
    batch_decode are all inside the timed region.

    LENGTH: the loader keeps ``workers + 2`` batches in flight, so a short run is fed from its start-up burst and says nothing
    about the host keeping up.  The timed region is therefore >= 30 s of GPU work and >= twice that depth
    (``n_batches`` = max(30 s x gpu_rate / batch, 2 x (workers + 2) + 2)), and ``host_ceiling_utt_per_s`` is the SAME loader
    run host-only (no model, no device) over the same number of batches after its own warm-up wave — a steady-state rate.
    ``host_bound`` = ceiling < 1.5 x the GPU-side rate.

    ``world > 1``: EVERY rank runs this leg on its own shard (dataset index i = rank mod world) with its own workers, and every
    batch ends with the ONE fixed-shape all_gather_into_tensor of (index, ids, length, bf16 first-step logits), bracketed by
    barriers, MAX over ranks.  Rank 0 returns the whole-node rate plus each rank's figures; the other ranks return None."""
    import math
    import shutil
    import tempfile
    from torch.utils.data import Subset
    from icl_speech_text_llm_amd.data.dataset_factory import DatasetFactory
    from icl_speech_text_llm_amd.data.model_processors import get_processor
    from icl_speech_text_llm_amd.data.synthetic_dataset import write_voxceleb_folder_arrow
    from icl_speech_text_llm_amd.data.task_configs import DatasetType, set_dataset_root
    from icl_speech_text_llm_amd.models.model_factory import ModelFactory
    from icl_speech_text_llm_amd.runtime import dp
    from icl_speech_text_llm_amd.utils.batch_loader import ArenaBatchLoader
    from icl_speech_text_llm_amd.utils.data_utils import clear_dataset_cache, load_dataset
    from icl_speech_text_llm_amd.utils.performance_utils import PerformanceTracker
    dev = torch.device(dev)
    on_gpu = dev.type == "cuda"
    bs = args.plugin_batch or args.batch
    workers = args.plugin_workers if workers is None else workers
    depth = workers + 2
    if n_batches is None:
        n_batches = max(int(math.ceil(30.0 * (gpu_rate or 145.0) / bs)), 2 * depth + 2)
    if make_model is None:
        model = ModelFactory.create_model("salmonn", device=str(dev), arch="tiny" if args.tiny else "7b", low_resource=True,
                                          llama_path="stand-in:subword", ckpt_path="", lora_alpha=32).eval()
    else:
        model = make_model()
    proc = get_processor("salmonn", model.input_processor, model.llama_tokenizer)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        sync()
    # ---- the dataset: an HF folder on local disk, read by the reference-compatible item pipeline ---------------------------
    secs = float(getattr(args, "plugin_audio_seconds", 30.0))
    n_rows = int(getattr(args, "plugin_rows", 0)) or min(256, max(bs, 8))      # distinct clips on disk; the run cycles through them
    root = os.environ.get("ICL_BENCH_DATA_ROOT") or os.path.join(tempfile.gettempdir(), f"icl_bench_data_{os.getuid()}_{n_rows}x{secs:g}s")
    t_write = time.perf_counter()
    if rank == 0:
        write_voxceleb_folder_arrow(root, n_rows, seconds=secs)
    barrier()
    set_dataset_root(root)
    clear_dataset_cache()
    rows = load_dataset(DatasetType.VOXCELEB, split="test")
    t_write = time.perf_counter() - t_write
    items = DatasetFactory.create_dataset(DatasetType.VOXCELEB, rows, proc, is_training=False, input_mode="speech_only",
                                          fewshot_mode="text", num_examples=5)
    per_rank = bs * (n_batches + warm)
    shard = dp.shard_indices(per_rank * world, rank, world)                    # i = rank (mod world), as the CLI shards
    ds = Subset(items, [i % len(items) for i in shard])

    # ---- host ceiling: the loader alone, host-only, steady state --------------------------------------------------------------
    barrier()
    host = ArenaBatchLoader(ds, bs, proc.collate_batch, num_workers=workers, device="cpu")
    stamps, n_host = [], 0
    for batch in host:
        n_host += len(batch["prompt"])
        stamps.append(time.perf_counter())
    host.close()
    skip = min(depth, max(0, len(stamps) - 2))               # the first wave was produced in parallel before anything was consumed
    host_rate = (len(stamps) - 1 - skip) * bs / max(stamps[-1] - stamps[skip], 1e-9)
    vocab = len(model.llama_tokenizer)
    if world > 1:
        packer = dp.result_packer(NEW_TOKENS, vocab)
        cdev = dp.collective_device(dist, dev)
        row_local = packer.alloc(bs, cdev)
        row_all = torch.empty(world * bs, packer.row_bytes, dtype=torch.uint8, device=cdev)
    tracker, done, prompt_tokens, stages, gaps, t_prev_end, t_start, t_end = None, 0, [], [], [], None, None, None
    pad_id, eos_id = model.llama_tokenizer.pad_token_id, model.llama_tokenizer.eos_token_id
    loader = ArenaBatchLoader(ds, bs, proc.collate_batch, num_workers=workers, device=dev)
    last_b = 0
    try:
        with torch.no_grad():
            for b_i, batch in enumerate(loader):       # as the CLI does: batch i+1 collated and copied under batch i's kernels
                if b_i == warm:
                    barrier()
                    tracker, t_start, t_prev_end = PerformanceTracker(log_interval=10 ** 9), time.perf_counter(), None
                batch["max_new_tokens"] = NEW_TOKENS
                t1 = time.perf_counter()
                if t_prev_end is not None and tracker is not None:
                    gaps.append(round((t1 - t_prev_end) * 1e3, 1))
                n_b = len(batch["prompt"])
                if world > 1:      # the data-parallel CLI's per-batch work: ids + logits as tensors, rank-local decode, one gather
                    res = model.generate_ids(batch, want_first_logits=True)
                    out = model.decode_ids(res.tokens)
                    ids = torch.full((n_b, NEW_TOKENS), pad_id, dtype=torch.int32)
                    ids[:, :res.tokens.shape[1]] = res.tokens.to(torch.int32)
                    is_eos = res.tokens == eos_id
                    glen = torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, torch.full((n_b,), res.tokens.shape[1])).to(torch.int32)
                    idx = torch.tensor(shard[b_i * bs: b_i * bs + n_b], dtype=torch.int64)
                    packer.pack(row_local, index=idx, gen_ids=ids, gen_len=glen, first_logits=res.first_logits)
                    dp.all_gather_rows(dist, row_local, out=row_all)
                else:
                    out = model.generate_output(batch)
                t_prev_end = time.perf_counter()
                if tracker is not None:
                    tracker.update(t_prev_end - t1, n_b)
                    done += len(out)
                    st = dict(getattr(model, "last_stage_seconds", {}), total=t_prev_end - t1)
                    stages.append({k: round(v * 1e3, 1) for k, v in st.items()})
                if b_i == 0:
                    prompt_tokens = [len(model.llama_tokenizer(p, add_special_tokens=False)["input_ids"]) for p in batch["prompt"][:4]]
                last_b = b_i
        barrier()
        t_end = time.perf_counter()
        overflow = loader.overflow_batches
        pinned = f"{loader.slots_pinned}/{depth}"
    finally:
        loader.close()
    dt = t_end - t_start
    summ = tracker.get_summary()
    gen_ms = [st["total"] for st in stages]
    my_rate = done / dt
    mine = {"rank": rank, "utterances": done, "seconds": round(dt, 3),
            "generate_output_ms": gen_ms, "between_batches_ms": gaps,
            "host_ceiling_utt_per_s": round(host_rate, 1), "host_stage_ms_last_batch": stages[-1] if stages else None,
            "examples_per_second_tracker": summ.get("examples_per_second")}
    note = ("ModelFactory -> HF folder on local disk (" + f"{n_rows} distinct {secs:g} s clips stored as Arrow number lists, cycled) -> load_dataset -> "
            "DatasetFactory / InferenceDataset -> SalmonProcessor -> ArenaBatchLoader -> generate_output; item pipeline, collation into "
            "pinned shared slots, H2D of raw audio, prompt split + tokenisation (stand-in sub-word tokenizer at ~3.9 chars per token: no "
            "Llama tokenizer files offline; prompt positions as listed, vs the frozen 376), K1..K11 and batch_decode inside the timed "
            "region; first batches excluded as warm-up; host_ceiling = the same loader host-only over the same batches, after its first "
            "wave; host_bound = ceiling < 1.5 x rate")
    if world == 1:
        del model
        if on_gpu:
            torch.cuda.empty_cache()
        steady = bs / (_median(gen_ms) * 1e-3) if gen_ms else None
        return {"utt_per_s": round(my_rate, 2), "examples_per_second_tracker": summ.get("examples_per_second"),
                "batch_size": bs, "batches_timed": len(gen_ms), "seconds_timed": round(dt, 1), "dataloader_workers": workers,
                "loader_batches_in_flight": depth, "dataset_rows_on_disk": n_rows, "dataset_write_and_load_s": round(t_write, 1),
                "host_ceiling_utt_per_s_per_rank": round(host_rate, 1), "host_ceiling_batches": len(stamps) - 1 - skip,
                "host_bound": bool(host_rate < 1.5 * my_rate), "host_ceiling_over_rate": round(host_rate / my_rate, 2),
                "host_stage_ms_last_batch": stages[-1] if stages else None,
                "generate_output_ms": {"min": min(gen_ms), "median": _median(gen_ms), "max": max(gen_ms)},
                "between_batches_ms": {"min": min(gaps), "median": _median(gaps), "max": max(gaps)} if gaps else None,
                "slot_overflow_batches": overflow, "slots_page_locked": pinned,
                "utt_per_s_steady": round(steady, 2) if steady else None,
                "prompt_positions_first_rows": [t + N_AUDIO_TOK for t in prompt_tokens], "note": note}
    # ---- world > 1: MAX over ranks, every rank's figures to rank 0 (objects, outside the timed region) ----------------------
    tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    ok_idx = None
    if rank == 0:
        got = packer.unpack(row_all)
        want = sorted(i for r in range(world) for i in dp.shard_indices(per_rank * world, r, world)[last_b * bs:(last_b + 1) * bs])
        ok_idx = sorted(got["index"].cpu().tolist()) == want
    del model
    if on_gpu:
        torch.cuda.empty_cache()
    if rank != 0:
        return None
    gen = [x for m in everyone for x in m["generate_output_ms"]]
    gap = [x for m in everyone for x in m["between_batches_ms"]]
    node_rate = sum(m["utterances"] for m in everyone) / float(tt.item())
    ceil_min = min(m["host_ceiling_utt_per_s"] for m in everyone)
    return {"utt_per_s": round(node_rate, 2), "n_ranks": world,
            "batch_size_per_rank": bs, "batches_timed_per_rank": len(gen_ms), "seconds_max_over_ranks": round(float(tt.item()), 3),
            "collective": f"all_gather_into_tensor per batch, {bs * packer.row_bytes} B per rank ({dist.get_backend()})",
            "last_batch_indices_ok": ok_idx,
            "generate_output_ms": {"min": min(gen), "median": _median(gen), "max": max(gen)},
            "between_batches_ms": {"min": min(gap), "median": _median(gap), "max": max(gap)} if gap else None,
            "host_ceiling_utt_per_s_per_rank": {"min": ceil_min, "max": max(m["host_ceiling_utt_per_s"] for m in everyone)},
            "host_bound": bool(ceil_min < 1.5 * node_rate / world), "host_ceiling_over_rate": round(ceil_min / (node_rate / world), 2),
            "loader_batches_in_flight": depth, "dataset_rows_on_disk": n_rows,
            "host_cores": host_cores(), "host_threads_per_rank": torch.get_num_threads(), "dataloader_workers_per_rank": workers,
            "per_rank": everyone, "prompt_positions_first_rows": [t + N_AUDIO_TOK for t in prompt_tokens],
            "note": note + "; world > 1: every rank runs its own workers on its own shard (i = rank mod world), all ranks' host-ceiling "
                           "runs share the node's cores at the same time, times are bracketed by barriers, utt_per_s = all ranks' "
                           "utterances / MAX over ranks of the wall time"}


def quick_workload(wl: str, dev, batch: int = 64, steps: int = 3, warmup: int = 1, num_beams: int = 1):
    """One of the non-headline workloads of BASELINE.md §4, in THIS process (a process that has initialised the GPU must not
    start GPU children): build the seeded model, keep `warmup + steps` micro-batches resident, two untimed passes for the decode
    graph, then time `steps` passes.  Same step definition as the headline loop (encoders -> prefill -> 10 greedy tokens)."""
    from icl_speech_text_llm_amd.runtime import binding as B, synth
    from icl_speech_text_llm_amd.runtime.config import QwenAudioCfg, SalmonnCfg
    wl_desc, wl_text, wl_naudio = WORKLOADS[wl]
    is_qwen = wl == "c4"
    t_build = time.perf_counter()
    if is_qwen:
        from icl_speech_text_llm_amd.runtime.qwen import QwenAudioRuntime
        cfg = QwenAudioCfg()
        rt = QwenAudioRuntime(cfg, synth.qwen_audio_state(cfg, seed=0, device=dev, dtype=torch.bfloat16), device=dev, consume=True)
        vocab, audio_tokens = cfg.llm.vocab, 750
    else:
        from icl_speech_text_llm_amd.runtime.salmonn import SalmonnRuntime
        cfg = SalmonnCfg.llama2_13b() if wl == "c5" else SalmonnCfg.llama2_7b()
        rt = SalmonnRuntime(cfg, synth.salmonn_state(cfg, seed=0, device=dev, dtype=torch.bfloat16), device=dev, consume=True)
        vocab, audio_tokens = cfg.llama.vocab, N_AUDIO_TOK
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    wavs, prompts = [], []
    lens = [480000] * (batch * wl_naudio)
    for s_ in range(warmup + steps):
        w, pr = synth_workload(s_ * batch, batch, vocab, wl_text, wl_naudio, audio_tokens)
        wavs.append(torch.from_numpy(w).to(dev))
        prompts.append(pr)

    def step(s_):
        speech = rt.encode_audio(raw_wav=wavs[s_], wav_lens=lens)[0] if is_qwen else rt.encode_speech(wavs[s_], lens)
        return rt.generate(prompts[s_], speech, max_new_tokens=NEW_TOKENS, suppress_eos=True, num_beams=num_beams).tokens
    for s_ in range(max(warmup, 2)):          # eager pass + capture pass of the decode graph, before the clock
        step(min(s_, warmup + steps - 1))
    torch.cuda.synchronize()
    profile = []
    B.GEMM_PROFILE = profile
    t0 = time.perf_counter()
    for s_ in range(warmup, warmup + steps):
        toks = step(s_)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    B.GEMM_PROFILE = None
    roof = workload_roofline(wl, batch * steps / dt, profile, dt) if num_beams == 1 else None
    if roof is not None and batch == 1:      # one utterance per step streams every weight once per phase and per token: HBM-bound
        roof["hbm"] = {"gb_per_utterance": 136.0, "peak_tb_s": 8.0, "frac": round(steps / dt * 136.0e9 / 8.0e12, 4),
                       "hbm_floor_ms_per_utterance": 17.0, "note": "SURVEY.md §8d: (1.48 + 13.48 + 9 x 13.48) GB of weights per utterance at micro-batch 1"}
    del profile
    out = {"workload": wl_desc + (f", beam search with {num_beams} beams" if num_beams > 1 else ""), "value": round(batch * steps / dt, 2), "unit": "utterances/s", "ms_per_step": round(dt / steps * 1e3, 1),
           "utterances_per_step": batch, "steps": steps,
           "prompt_positions": [t + wl_naudio * audio_tokens for t in wl_text] if len(wl_text) > 1 else wl_text[0] + wl_naudio * audio_tokens,
           "new_tokens": int(toks.shape[1]), "workspace_gib": round(rt.ws.nbytes() / 2 ** 30, 1), "build_s": round(t_build, 1),
           "roofline": roof}
    del rt, wavs
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        log(f"note: --gpus {args.gpus} but the launcher started {world} rank(s); n_gpus reports {world}")
    # host threads and DataLoader workers per rank are bounded BEFORE anything touches the GPU: 8 ranks share the node's cores
    cores = host_cores()
    cores_per_rank = max(1, cores // max(world, 1))
    torch.set_num_threads(cores_per_rank)       # at N = 1 too: torch's default is the machine's core count, not the cgroup's share
    plugin_workers = max(1, min(args.plugin_workers, cores_per_rank - 1))
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

    from icl_speech_text_llm_amd.runtime import binding as B, dp, synth
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
    wavs, prompts, firsts = [], [], []
    lens = [480000] * (Bm * wl_naudio)
    for s in range(total_steps):
        first = (s * world + rank) * Bm           # utterance i belongs to rank (i // Bm) % world of step i // (Bm*world)
        firsts.append(first)
        if args.workload == "c2":
            w, ids = synth_utterances(first, Bm, vocab)
            prompts.append(build_prompts(ids))
            if s == 0:
                w0, ids0 = w[0].copy(), ids[0].copy()
        else:
            w, pr = synth_workload(first, Bm, vocab, wl_text, wl_naudio, audio_tokens)
            prompts.append(pr)
        wavs.append(torch.from_numpy(w).to(dev))
    # §8e result row: (utterance index, gen_ids int32 [10], gen_len, first-step logits bf16 [V]); ONE all-gather per step
    packer = dp.result_packer(NEW_TOKENS, vocab)
    cdev = dp.collective_device(dist, dev) if world > 1 else dev
    row_local = packer.alloc(Bm, cdev) if world > 1 else None
    row_all = torch.empty(world * Bm, packer.row_bytes, dtype=torch.uint8, device=cdev) if world > 1 else None
    idx_dev = torch.arange(Bm, dtype=torch.int64, device=cdev)
    len_dev = torch.full((Bm,), NEW_TOKENS, dtype=torch.int32, device=cdev)

    def step(s, gather=True):
        if is_qwen:
            speech, _ = rt.encode_audio(raw_wav=wavs[s], wav_lens=lens)
        else:
            speech = rt.encode_speech(wavs[s], lens)
        res = rt.generate(prompts[s], speech, max_new_tokens=NEW_TOKENS, suppress_eos=True, want_first_logits=True)
        if world > 1 and gather:
            packer.pack(row_local, index=idx_dev + firsts[s], gen_ids=rt.ws.get("gen_tokens", (Bm, NEW_TOKENS), torch.int32),
                        gen_len=len_dev, first_logits=res.first_logits)
            dp.all_gather_rows(dist, row_local, out=row_all)
        return res.tokens

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"inputs resident ({total_steps} x {Bm} utterances); warmup ...")
    first_tokens = None
    # the decode loop is captured into a HIP graph on its second pass: make both passes happen BEFORE the timed region
    # whatever --warmup says (a capture inside the timed region would also be a step that is not like the others)
    for s in range(max(args.warmup, 0)):
        toks = step(s)
        torch.cuda.synchronize()
        log(f"warmup step {s} done")
        if s == 0:
            first_tokens = toks[0].tolist()
    for extra in range(max(0, 2 - args.warmup) if rt.use_graphs else 0):
        toks = step(0, gather=False)
        torch.cuda.synchronize()
        log(f"graph warm-up pass {extra} (untimed, not counted in --warmup) done")
        if first_tokens is None:
            first_tokens = toks[0].tolist()
    profile = None if args.no_gemm_profile else []
    barrier()
    B.GEMM_PROFILE = profile
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        toks = step(s)
    barrier()
    elapsed = time.perf_counter() - t0
    B.GEMM_PROFILE = None
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gather_info = None
    if world > 1 and rank == 0:
        got = packer.unpack(row_all)
        s_last = total_steps - 1
        want_idx = torch.cat([torch.arange(Bm) + (s_last * world + r) * Bm for r in range(world)])
        mine = rt.ws.get("gen_tokens", (Bm, NEW_TOKENS), torch.int32).cpu()
        gather_info = {"collective": "all_gather_into_tensor (one per step)", "bytes_per_rank_per_step": Bm * packer.row_bytes,
                       "fields": [f[0] for f in packer.fields], "logits": f"bf16 [{Bm}, {vocab}] per rank",
                       "last_step_indices_ok": bool(torch.equal(got["index"].cpu(), want_idx)),
                       "own_rows_roundtrip_ok": bool(torch.equal(got["gen_ids"][:Bm].cpu(), mine)),
                       "backend": dist.get_backend()}

    # ---- roofline of the dominant kernel, from live HIP-event timings -------------------------------
    roof = None
    if profile:
        names = GEMM_KERNEL_NAMES
        dom, per_tile = gemm_event_summary(profile, elapsed)
        all_t = sum(v[1] for v in per_tile.values())
        fl, tt, n = per_tile[dom]
        traffic, traffic_src = None, None
        for cand in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:   # HBM-side bytes per launch of this kernel from the committed PMC passes of this same command (profiles/)
                pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
                key = {1: "gemm_bf16_kernel<2, 2, 4, 4>", 2: "gemm_bf16_kernel<2, 2, 2, 2>", 3: "gemm256_bf16_kernel"}[dom]
                if f"(batch {Bm}" in pmc["command"] and not args.tiny:
                    # all instantiations; launches listed apart ("name [label]": the decode split-K launches, which run inside the
                    # captured decode graph and are not among the live-timed launches above) stay out of the average
                    rows = [v for k, v in pmc["kernels"].items() if (k == key or k.startswith(key + "<")) and "[" not in k]
                    traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rows) / sum(v["launches"] for v in rows))
                    traffic_src = cand
                    break
            except Exception:
                continue
        roof = {"bound": "mfma", "kernel": names.get(dom, str(dom)), "achieved": round(fl / tt / 1e12, 1),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / tt / 1e12 / PEAK_BF16_TFLOPS, 4),
                "traffic": traffic, "traffic_note": f"bytes/launch = 2*FETCH_SIZE + WRITE_SIZE (KiB) from separate rocprofv3 --pmc "
                f"passes of this command (profiles/{traffic_src}); the counters sit at the L2<->fabric boundary and "
                "include Infinity-Cache hits" if traffic else None, "launches": n, "avg_launch_us": round(tt / n * 1e6, 2),
                "avg_launch_gflop": round(fl / n / 1e9, 3), "share_of_step_time": round(tt / elapsed, 3),
                "clock_note": "peak is the nominal 2.4 GHz figure; SQ counters over the bench's own launches (round 3, profiles/r03_pmc_sq_*.json: "
                              "SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) read 75-80 % matrix-pipe busy at ~1.75 GHz on the K >= 4096 shapes, "
                              "51-63 % at ~1.9 GHz on the K = 1280 shapes: busy x clock / 2.4 GHz reproduces this fraction",
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
    if rank == 0 and args.workload == "c2" and not args.tiny and not args.no_phases:
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
    parity_failed = False
    want_plugin = args.workload == "c2" and not args.no_through_plugin and (args.through_plugin or not args.tiny)
    plugin_block = None
    if world > 1 and want_plugin:
        # every rank: hand the runtime leg's model and workspace back first (the plugin builds its own replica), then the
        # whole per-rank host pipeline + the per-batch gather, barriers on both sides
        hbm_peak = round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)
        ws_gib = round(rt.ws.nbytes() / 2 ** 30, 2)
        del rt
        torch.cuda.empty_cache()
        log(f"through-plugin leg on all {world} ranks ({plugin_workers} DataLoader workers, {cores_per_rank} host threads per rank) ...")
        # no try / except here: a rank that fails must take the job down (non-zero exit under the launcher), not leave the
        # others waiting in a collective
        plugin_block = through_plugin(args, dev, dist=dist, rank=rank, world=world, workers=plugin_workers,
                                      gpu_rate=Bm * args.steps / elapsed)
        rt = None
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
            "roofline": roof, "phases": phases, "gather": gather_info,
            "workspace_gib": round(rt.ws.nbytes() / 2 ** 30, 2) if rt is not None else ws_gib,
            "build_s": round(t_build, 1), "first_utterance_tokens": first_tokens,
            "host": {"cores": cores, "ranks": world, "threads_per_rank": torch.get_num_threads(),
                     "dataloader_workers_per_rank": plugin_workers},
        }
        out["hbm_peak_gib_timed_loop"] = round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1) if rt is not None else hbm_peak
        if plugin_block is not None:
            out["through_plugin"] = plugin_block
            log(f"through-plugin (all ranks): {json.dumps({k: v for k, v in plugin_block.items() if k not in ('per_rank', 'note')})}")
        if world == 1:
            # the side legs below build a second model (plugin) and a third set of decoder weights (margin parity) next to this
            # one: hand the 94 GiB of micro-batch-256 workspace back first (the parity legs run one utterance and re-grow
            # what they need)
            rt.ws.clear()
            torch.cuda.empty_cache()
        if world == 1 and want_plugin:
            log("through-plugin leg ...")
            try:
                out["through_plugin"] = through_plugin(args, dev, workers=plugin_workers, gpu_rate=n_utt / elapsed)
            except Exception as e:       # a reported side number must never cost the headline line
                out["through_plugin"] = {"error": f"{type(e).__name__}: {e}"}
            log(f"through-plugin: {out['through_plugin']}")
        if want_cpu:
            threads = host_cores()
            log(f"cpu legs: copying {len(sd)} tensors to host fp32 ...")
            sd_host = _to_host_f32(sd)
            del sd
            log(f"cpu_baseline: the fp32 oracle on 1 utterance with {threads} threads ...")
            dt, t_enc, cpu_tokens, cpu_first, fp32_stages = cpu_utterance(sd_host, cfg, w0, ids0, threads)
            log(f"cpu_baseline: {dt:.1f} s ({t_enc:.1f} s speech encoders)")
            cb = {"value": round(1.0 / dt, 5), "unit": "utterances/s", "cores": threads, "kind": "port",
                  "sample": f"1 utterance of the same C2 workload (30 s audio, 376 positions, 10 greedy tokens), fp32 torch-CPU "
                            f"oracle, batch 1, {dt:.1f} s", "utterances": 1, "tokens": cpu_tokens,
                  "tokens_match_gpu": cpu_tokens == first_tokens,
                  "tokens_note": "informational: under the frozen N(0, 0.02^2) weights the fp32 oracle's top-1 margins are ~0.01 logits "
                                 "against a bf16-vs-fp32 logit distance of ~0.2 (parity block: the oracle's own two precisions differ by "
                                 "3.3e-2 relative), so greedy ids of the two precisions may part at a near-tie; what is asserted is "
                                 "parity.decode_steps_teacher_forced (every GPU choice within 2x the logit error of the oracle's arg-max) "
                                 "and token-for-token equality on the decisive-margin weights (tokens_match_gpu_on_margin_weights)"}
            rec = os.path.join(ROOT, "profiles", "r02_cpu_baseline_full.json")
            if os.path.exists(rec):     # the full BASELINE.md §3 protocol, recorded once with --cpu-baseline-full on an MI355X host
                try:
                    cb["recorded_full_protocol"] = dict(json.load(open(rec)), source="profiles/r02_cpu_baseline_full.json")
                except Exception:
                    pass
            if args.cpu_baseline_full:
                ws_, is_ = synth_utterances(0, 16, vocab)
                full = cpu_baseline_full(sd_host, cfg, list(ws_), list(is_), threads)
                full["host_cores"] = threads
                cb["full_protocol"] = full
                if args.cpu_baseline_out:
                    os.makedirs(os.path.dirname(os.path.abspath(args.cpu_baseline_out)), exist_ok=True)
                    json.dump(full, open(args.cpu_baseline_out, "w"), indent=1)
            out["cpu_baseline"] = cb
            if not args.tiny:
                log("parity: the bf16-rounding oracle at full size, stage by stage + 10 teacher-forced decode steps ...")
                par = full_size_parity(cfg, sd_host, rt, dev, w0, ids0, fp32_stages, cpu_first)
                emb0 = rt.encode_speech(torch.from_numpy(w0)[None], [480000]).clone()[0]
                del sd_host
                log(f"parity: {json.dumps(par)}")
                log("parity: decisive-margin weight set (token exactness) ...")
                par["margin_weights"] = margin_parity(cfg, dev, ids0, emb0)
                log(f"parity(margin): {json.dumps(par['margin_weights'])}")
                par["bounds_rel_l2"] = PARITY_BOUNDS
                par["ok"] = bool(par["ok"] and par["margin_weights"]["ok"])
                cb["tokens_match_gpu_on_margin_weights"] = par["margin_weights"]["tokens_match"]
                out["parity"] = par
                parity_failed = not par["ok"]
        if world == 1 and args.workload == "c2" and not args.tiny and not args.no_other_workloads:
            # last: the parent's model, workspace and side models are released first (the children build their own)
            try:
                del rt
            except Exception:
                pass
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["other_workloads"] = {"note": "BASELINE.json configs 4 and 5 (the 1-GPU half of 5) on this box at micro-batch "
                                              f"{args.other_batch} (prefilled 128 sequences at a time, decoded together), 3 timed steps each after "
                                              "the graph warm-up, same step definition, each with its own roofline block (TFLOP per utterance "
                                              "from SURVEY.md §8d; dominant GEMM kernel from live HIP events of the timed steps); run after the "
                                              "headline legs in this process; c2_beams4 = the headline workload under num_beams=4 (64 prompts "
                                              "prefilled once, 256 beam sequences decoded); batch1 = the reference CLI's own operating point "
                                              "(--batch_size 1, inference/inference.py:62): one utterance per step, latency-bound"}
            for wl in ("c4", "c5", "c2_beams4", "batch1"):
                log(f"other workload {wl} ...")
                try:
                    out["other_workloads"][wl] = (quick_workload("c2", dev, num_beams=4) if wl == "c2_beams4" else
                                                  quick_workload("c2", dev, batch=1, steps=20, warmup=2) if wl == "batch1" else
                                                  quick_workload(wl, dev, batch=args.other_batch))
                except Exception as e:      # a reported side number must never cost the headline line
                    out["other_workloads"][wl] = {"error": f"{type(e).__name__}: {e}"}
                    torch.cuda.empty_cache()
                log(f"other workload {wl}: {out['other_workloads'][wl]}")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if parity_failed:
        print("bench.py: FULL-SIZE PARITY BOUND EXCEEDED (see \"parity\" in the JSON line)", file=sys.stderr, flush=True)
        sys.exit(4)


if __name__ == "__main__":
    main()
