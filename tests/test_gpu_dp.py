"""`-m gpu`: the data-parallel gather over the RCCL backend (torch.distributed "nccl" on ROCm), as far as one GPU allows —
a world of ONE rank in a child process (RCCL refuses two ranks on one device; the world-2 / world-3 semantics are covered on
gloo in tests/test_host_and_abi.py, the 8-GPU run is the driver's).  What this pins: the device-side pack / all-gather /
unpack path (`collective_device` picks the GPU under nccl), bf16 logits travelling as raw bytes, padding rows dropped,
ordering by dataset index."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["ICL_ROOT"])
    from icl_speech_text_llm_amd.runtime import dp
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    dev = torch.device("cuda:0")
    assert dp.collective_device(dist, dev).type == "cuda"
    n, T, V = 5, 10, 32001
    g = torch.Generator().manual_seed(0)
    index = torch.tensor([40, 8, 16, 0, 24])                       # a shard in loader order, not sorted
    ids = torch.randint(0, 32000, (n, T), generator=g, dtype=torch.int32).to(dev)
    ln = torch.randint(1, T + 1, (n,), generator=g, dtype=torch.int32).to(dev)
    logits = torch.randn(n, V, generator=g).to(torch.bfloat16).to(dev)
    out = dp.gather_results(dist, dev, index, ids, ln, logits, rows_per_rank=8)      # 3 padding rows
    order = torch.argsort(index)
    assert out["index"].tolist() == sorted(index.tolist())
    assert out["gen_ids"].is_cuda and torch.equal(out["gen_ids"], ids[order.to(dev)])
    assert torch.equal(out["gen_len"], ln[order.to(dev)])
    assert out["first_logits"].dtype == torch.bfloat16 and torch.equal(out["first_logits"], logits[order.to(dev)])
    recs = dp.gather_json_records(dist, dev, [{"text": "caf\\u00e9 %d" % i} for i in index.tolist()], index.tolist(), rows_per_rank=8)
    assert sorted(recs) == sorted(index.tolist()) and recs[16]["text"] == "caf\\u00e9 16"
    dist.barrier()
    dist.destroy_process_group()
    print("rccl-gather-ok")
""")


def test_gather_over_rccl_world1(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", ICL_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl-gather-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
