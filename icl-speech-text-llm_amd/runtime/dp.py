"""Data-parallel plumbing of the ICL path: utterances are independent units, rank r of W takes the
dataset indices i ≡ r (mod W), every rank holds a full weight replica, and the ONLY collective is a fixed-shape
``all_gather_into_tensor`` of the per-utterance results (generated ids, their lengths, the first-step logits and the
dataset index each row belongs to).  The reference has no inference collective at all (its only ``torch.distributed``
use is DDP training, train/train.py:138,233-238) — this is new work, not a translation.

Nothing here touches HIP: with the ``nccl`` backend (= RCCL over xGMI on ROCm) the tensors live on the rank's GPU, with
``gloo`` (CPU tests, world_size 2) on the host.  Rows travel as ONE byte matrix per call so that a step costs one
collective whatever the number of fields; rows a rank does not have are marked with index −1, which is how a batch that
failed on one rank shows up at rank 0 as *missing indices* instead of silently shifting every later record.
"""
from __future__ import annotations

import json
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch


def shard_indices(total: int, rank: int, world: int) -> List[int]:
    """Dataset indices of rank ``rank``: i ≡ rank (mod world), no DistributedSampler-style padding duplicates."""
    return list(range(rank, total, world))


def collective_device(dist, fallback: torch.device) -> torch.device:
    """Device collectives must run on: the rank's GPU under nccl/RCCL, the host under gloo."""
    return torch.device(fallback) if dist.get_backend() == "nccl" else torch.device("cpu")


class RowPacker:
    """Fixed byte layout of one result row: named fields of (dtype, trailing shape), each 16-byte aligned.

    ``pack`` writes tensors [n, *shape] into a uint8 matrix [n_rows, row_bytes]; ``unpack`` returns views of a gathered
    matrix.  The layout is a pure function of the field list, so every rank computes the same one."""

    def __init__(self, fields: Sequence[Tuple[str, torch.dtype, Tuple[int, ...]]]):
        self.fields = []
        off = 0
        for name, dtype, shape in fields:
            n = 1
            for s in shape:
                n *= int(s)
            nbytes = n * torch.empty((), dtype=dtype).element_size()
            self.fields.append((name, dtype, tuple(int(s) for s in shape), off, nbytes))
            off += -(-nbytes // 16) * 16
        self.row_bytes = off

    def alloc(self, n_rows: int, device) -> torch.Tensor:
        return torch.zeros(n_rows, self.row_bytes, dtype=torch.uint8, device=device)

    def pack(self, buf: torch.Tensor, **tensors: torch.Tensor) -> torch.Tensor:
        for name, dtype, shape, off, nbytes in self.fields:
            t = tensors[name]
            n = t.shape[0]
            if n == 0 or nbytes == 0:
                continue
            src = t.to(device=buf.device, dtype=dtype).reshape(n, -1).contiguous()
            buf[:n, off:off + nbytes] = src.view(torch.uint8).reshape(n, nbytes)
        return buf

    def unpack(self, buf: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        n = buf.shape[0]
        for name, dtype, shape, off, nbytes in self.fields:
            if nbytes == 0:
                out[name] = torch.zeros((n,) + shape, dtype=dtype, device=buf.device)
                continue
            out[name] = buf[:, off:off + nbytes].contiguous().view(dtype).reshape((n,) + shape)
        return out


def result_packer(new_tokens: int, n_logits: int) -> RowPacker:
    """§8e row: dataset index, generated ids (pad-filled to ``new_tokens``), generated length, first-step logits in bf16
    (the full vocabulary row, or the label-restricted slice when ``n_logits`` is the number of label tokens)."""
    return RowPacker([("index", torch.int64, ()), ("gen_ids", torch.int32, (new_tokens,)), ("gen_len", torch.int32, ()),
                      ("first_logits", torch.bfloat16, (n_logits,))])


def all_gather_rows(dist, local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ONE ``all_gather_into_tensor`` of a fixed-shape row matrix: [n, row_bytes] per rank -> [world * n, row_bytes]."""
    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def gather_results(dist, device, index: torch.Tensor, gen_ids: torch.Tensor, gen_len: torch.Tensor,
                   first_logits: torch.Tensor, rows_per_rank: int) -> Dict[str, torch.Tensor]:
    """Gather every rank's result rows (padded to ``rows_per_rank`` with index −1) and return them ordered by dataset
    index with the padding dropped: {"index", "gen_ids", "gen_len", "first_logits"} on ``device``'s collective device."""
    dev = collective_device(dist, device)
    n = int(index.shape[0])
    if n > rows_per_rank:
        raise ValueError(f"{n} local rows > rows_per_rank {rows_per_rank}")
    pk = result_packer(gen_ids.shape[1], first_logits.shape[1])
    buf = pk.alloc(rows_per_rank, dev)
    idx = torch.full((rows_per_rank,), -1, dtype=torch.int64, device=dev)
    idx[:n] = index.to(dev)
    pk.pack(buf, index=idx, gen_ids=_pad_rows(gen_ids, rows_per_rank, dev), gen_len=_pad_rows(gen_len, rows_per_rank, dev),
            first_logits=_pad_rows(first_logits, rows_per_rank, dev))
    got = pk.unpack(all_gather_rows(dist, buf))
    keep = got["index"] >= 0
    order = torch.argsort(got["index"][keep], stable=True)
    return {k: v[keep][order] for k, v in got.items()}


def _pad_rows(t: torch.Tensor, n_rows: int, dev) -> torch.Tensor:
    t = t.to(dev)
    if t.shape[0] == n_rows:
        return t
    out = torch.zeros((n_rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
    out[:t.shape[0]] = t
    return out


def gather_json_records(dist, device, records: Sequence[Dict[str, Any]], index: Sequence[int], rows_per_rank: int
                        ) -> Dict[int, Dict[str, Any]]:
    """The string side of a result (text, true label, dataset type) as fixed-width UTF-8 rows — same one-collective shape as
    the numeric side, no pickled objects.  Returns {dataset index: record}."""
    dev = collective_device(dist, device)
    blobs = [json.dumps(r, ensure_ascii=False).encode("utf-8") for r in records]
    width = torch.tensor([max([len(b) for b in blobs], default=0)], dtype=torch.int64, device=dev)
    dist.all_reduce(width, op=dist.ReduceOp.MAX)
    W = max(16, -(-int(width.item()) // 16) * 16)
    pk = RowPacker([("index", torch.int64, ()), ("nbytes", torch.int32, ()), ("utf8", torch.uint8, (W,))])
    rows = torch.zeros(rows_per_rank, W, dtype=torch.uint8)
    for i, b in enumerate(blobs):
        rows[i, :len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
    idx = torch.full((rows_per_rank,), -1, dtype=torch.int64)
    idx[:len(blobs)] = torch.as_tensor(list(index), dtype=torch.int64)
    nb = torch.zeros(rows_per_rank, dtype=torch.int32)
    nb[:len(blobs)] = torch.tensor([len(b) for b in blobs], dtype=torch.int32)
    buf = pk.pack(pk.alloc(rows_per_rank, dev), index=idx, nbytes=nb, utf8=rows)
    got = pk.unpack(all_gather_rows(dist, buf))
    out: Dict[int, Dict[str, Any]] = {}
    gi, gn, gu = got["index"].cpu(), got["nbytes"].cpu(), got["utf8"].cpu()
    for r in range(gi.shape[0]):
        i = int(gi[r])
        if i >= 0:
            out[i] = json.loads(bytes(gu[r, :int(gn[r])].tolist()).decode("utf-8"))
    return out
