"""Where does the HIP Llama chain leave the bf16-rounding oracle?  Layer-0 stage-by-stage comparison at miniature dims."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B, synth
from icl_speech_text_llm_amd.runtime.config import SalmonnCfg
from icl_speech_text_llm_amd.runtime.salmonn import SalmonnRuntime
from icl_speech_text_llm_amd.runtime.engines import _i32
from oracle import models as om
import torch.nn.functional as F

def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm())

cfg = SalmonnCfg.tiny(use_beats=True, lora=True)
sd = synth.salmonn_state(cfg, seed=0, jitter=True)
rt = SalmonnRuntime(cfg, dict(sd), device="cuda", parts=("llama",))
lsd = {k[len("llama_model."):]: v for k, v in sd.items() if k.startswith("llama_model.")}
ob = om.LlamaOracle(lsd, cfg.llama.n_heads, cfg.llama.rms_eps, cfg.llama.rope_theta, cfg.llama.lora_scale, rnd=om.bf16_round)
S = 155
ids = np.random.default_rng(99).integers(3, cfg.llama.vocab - 1, S).tolist()
c, w, ws = cfg.llama, rt.llama.w, rt.ws
hd, H, D, I = c.hidden, c.n_heads, c.head_dim, c.ffn
h = rt.llama.embed(ws, _i32(ids, "cuda"), None, name="dbg_h").clone()
x0 = ob.embed(torch.tensor(ids))[None]
print("embed", rel(h, x0[0]))
L = w.layers[0]
rnd = om.bf16_round
lp = "model.layers.0."
# oracle pieces
xn_o = ob._rms(x0, lp + "input_layernorm.weight")
q_o = ob._proj(xn_o, lp + "self_attn.q_proj"); k_o = ob._proj(xn_o, lp + "self_attn.k_proj"); v_o = rnd(ob._proj(xn_o, lp + "self_attn.v_proj"))
pos = torch.arange(S)[None]
qr_o = rnd(ob._rope(rnd(q_o).view(1, S, H, D), pos)); kr_o = rnd(ob._rope(rnd(k_o).view(1, S, H, D), pos))
mask = (torch.arange(S)[None, :] > torch.arange(S)[:, None])[None, None]
a_o = rnd(om._mha(qr_o.reshape(1, S, hd), kr_o.reshape(1, S, hd), v_o, H, D ** -0.5, mask=mask))
h1_o = x0 + ob._proj(a_o, lp + "self_attn.o_proj")
xn2_o = ob._rms(h1_o, lp + "post_attention_layernorm.weight")
g_o = om._lin(xn2_o, lsd, lp + "mlp.gate_proj", rnd); u_o = om._lin(xn2_o, lsd, lp + "mlp.up_proj", rnd)
act_o = rnd(F.silu(g_o) * u_o)
h2_o = h1_o + om._lin(act_o, lsd, lp + "mlp.down_proj", rnd)
# GPU pieces (same calls as LlamaHIP._layer, no cache)
BF16 = torch.bfloat16
xn = torch.zeros(S, w.k_aug, dtype=BF16, device="cuda")
B.rmsnorm(h, L.rms1, xn, c.rms_eps, N=hd)
print("rmsnorm (bf16 out) vs rnd(oracle)", rel(xn[:, :hd], rnd(xn_o[0])), "mismatching elems", float((xn[:, :hd].float().cpu() != rnd(xn_o[0])).float().mean()))
r2 = L.lora_a.shape[0]
B.gemm(xn, L.lora_a, xn[:, hd:hd + r2], K=hd, tile=2)
qkv = torch.empty(S, 3 * hd, dtype=BF16, device="cuda")
B.gemm(xn, L.wqkv, qkv, bias=L.bqkv, N=3 * hd, K=w.k_aug)
print("q (pre-rope) vs rnd(oracle)", rel(qkv[:, :hd], rnd(q_o[0])), " k", rel(qkv[:, hd:2 * hd], rnd(k_o[0])), " v", rel(qkv[:, 2 * hd:], v_o[0]))
posd, sid = _i32(list(range(S)), "cuda"), _i32([0] * S, "cuda")
B.rope_kv(qkv, hd, 2 * hd, w.rope_cos, w.rope_sin, posd, sid, None, None, H, D, 0, M=S)
print("q roped vs oracle", rel(qkv[:, :hd], qr_o.reshape(S, hd)), " k roped", rel(qkv[:, hd:2 * hd], kr_o.reshape(S, hd)))
att = torch.empty(S, hd, dtype=BF16, device="cuda")
cu = _i32([0, S], "cuda")
B.attn_fwd(qkv[:, :hd], qkv[:, hd:2 * hd], qkv[:, 2 * hd:], att, cu, S, H, D, D ** -0.5, causal=True)
print("attention out vs oracle", rel(att, a_o[0]), "mismatching elems", float((att.float().cpu() != a_o[0]).float().mean()))
h1 = h.clone()
B.gemm(att, L.wo, h1, residual=h1, N=hd, K=hd)
print("h after attention block", rel(h1, h1_o[0]))
B.rmsnorm(h1, L.rms2, xn, c.rms_eps, N=hd)
act = torch.empty(S, I, dtype=BF16, device="cuda")
B.gemm(xn, L.wgu, act, swiglu=True, K=hd, N=2 * I)
print("swiglu act vs oracle", rel(act, act_o[0]), "mismatching elems", float((act.float().cpu() != act_o[0]).float().mean()))
B.gemm(act, L.wdown, h1, residual=h1, N=hd, K=I)
print("h after layer 0", rel(h1, h2_o[0]))
