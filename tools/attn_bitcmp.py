"""Same inputs through two builds of the library (ICL_LIB_A / ICL_LIB_B): are the prefill-attention outputs bit-identical?
Covers D = 64 without bias (full tiles, ragged lengths, key padding, causal), D = 128 causal / key-padded, the gated bias.
usage: ICL_LIB=<lib.so> python tools/attn_bitcmp.py <out.pt>;  python tools/attn_bitcmp.py --cmp a.pt b.pt"""
import os, sys
import torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B

def run(lib, cases):
    B.LIB_PATH = lib
    B.load_library()
    outs = []
    for (lens, H, D, causal, kvl, seed) in cases:
        g = torch.Generator().manual_seed(seed)
        M = sum(lens)
        q, k, v = (torch.randn(M, H * D, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
        cu = [0]
        for n in lens:
            cu.append(cu[-1] + n)
        cu_t = torch.tensor(cu, dtype=torch.int32, device="cuda")
        out = torch.empty(M, H * D, dtype=torch.bfloat16, device="cuda")
        kw = {}
        if kvl == "bias":           # BEATs gated relative-position bias
            span = max(lens)
            kw = dict(rel_bias=torch.randn(H, 2 * span - 1, generator=g).cuda(), rel_gate=(torch.rand(M, H, generator=g) * 2).cuda(), rel_span=span)
        elif kvl is not None:
            kw["kv_lens"] = torch.tensor(kvl, dtype=torch.int32, device="cuda")
        B.attn_fwd(q, k, v, out, cu_t, max(lens), H, D, D ** -0.5, causal=causal, **kw)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    return outs

if __name__ == "__main__":
    cases = [([1500] * 4, 20, 64, False, None, 1), ([1496, 700, 64, 1, 129], 12, 64, False, None, 2),
             ([300, 130], 2, 64, False, [260, 100], 3), ([200, 77, 513], 4, 64, True, None, 4),
             ([32, 88, 88], 12, 64, False, None, 5), ([1500, 1500], 20, 64, False, [750, 1499], 6),
             # the generic kernel: decoder prefill (D = 128 causal; 1..6 tiles per block, ragged) and the gated-bias variant
             ([376, 373, 600, 64, 1, 129, 65], 8, 128, True, None, 7), ([128, 256, 192], 4, 128, False, [100, 256, 65], 8),
             ([1496, 700, 64, 1, 129, 191], 12, 64, False, "bias", 9), ([376] * 3, 32, 128, True, None, 10)]
    if sys.argv[1] == "--cmp":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        same = [bool(torch.equal(x, y)) for x, y in zip(a, b)]
        print(f"bitcmp {sys.argv[2]} vs {sys.argv[3]}: {'IDENTICAL' if all(same) else 'DIFFERENT'} {same}")
        sys.exit(0 if all(same) else 1)
    which = sys.argv[1]
    torch.save(run(os.environ["ICL_LIB"], cases), which)
