"""Multi-GPU plumbing of the rollout path: the path shards by evaluation chain, one process per GPU,
no data-path collective.  RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in CPU tests) is used for
 * ``shard_sequences``  - the reference's rank slicing of the evaluation chains (eval_utils.py:523-527),
 * ``reduce_metrics``   - ONE small all-reduce that replaces the reference's pickled ``gather_object`` of per-chain
                          result tuples (eval_utils.py:567-568),
 * ``all_gather_values`` - calibration deltas for the threshold solver (value_net.py:195-201).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collective_device(device=None):
    """Device the tensors of a collective must live on: RCCL ("nccl" on ROCm) only moves device memory - a CPU tensor handed to
    all_reduce / all_gather raises - while gloo (CPU tests) takes host tensors.  None = leave the tensor where it is."""
    if device is not None:
        return torch.device(device)
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return None


def shard_sequences(seqs: Sequence, rank: int, world: int) -> List:
    """eval_utils.py:523-527: ``assert NUM_SEQUENCES % device_num == 0``; rank r takes a contiguous block."""
    n = len(seqs)
    assert n % world == 0, f"number of sequences {n} must be divisible by the number of ranks {world}"
    per = n // world
    return list(seqs[rank * per:(rank + 1) * per])


METRIC_FIELDS = ["sum_success_len", "n_chains", "sr1", "sr2", "sr3", "sr4", "sr5", "sum_exit_plus1", "n_steps", "sum_llm_time"]


def pack_metrics(results: Sequence[int], exit_layers: Sequence[int], n_layers: int, llm_time: float = 0.0) -> torch.Tensor:
    """results: successful chain length (0..5) per evaluation chain; exit_layers: exit layer of every control step.
    Layout: METRIC_FIELDS followed by the exit histogram (n_layers bins).  (count_success / count_exit_ratio of
    eval_utils.py:53-118 become sums that reduce with one all_reduce(SUM).)"""
    t = torch.zeros(len(METRIC_FIELDS) + n_layers, dtype=torch.float64)
    t[0] = float(sum(results))
    t[1] = len(results)
    for k in range(1, 6):
        t[1 + k] = float(sum(1 for r in results if r >= k))
    t[7] = float(sum(e + 1 for e in exit_layers))
    t[8] = len(exit_layers)
    t[9] = llm_time
    for e in exit_layers:
        t[len(METRIC_FIELDS) + e] += 1
    return t


def reduce_metrics(packed: torch.Tensor, device=None, n_extra: int = 0) -> Dict[str, float]:
    """all_reduce(SUM) over ranks (a few hundred bytes: latency-bound, xGMI bandwidth is irrelevant) -> metrics of
    ``print_and_save`` (eval_utils.py:71-118): avg successful length, chain success rates 1-5, avg exit layer (+1)."""
    t = packed.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dev = collective_device(device)
        t = t.to(dev) if dev is not None else t
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t.cpu()
    extra = None
    if n_extra:                                                   # caller-defined sums appended to the packed vector
        extra = [float(v) for v in t[-n_extra:]]
        t = t[:-n_extra]
    n_chains, n_steps = max(float(t[1]), 1.0), max(float(t[8]), 1.0)
    out = {"avg_seq_len": float(t[0]) / n_chains, "n_chains": int(t[1]), "n_steps": int(t[8]),
           "avg_exit": float(t[7]) / n_steps, "llm_time": float(t[9]),
           "chain_sr": [float(t[1 + k]) / n_chains for k in range(1, 6)],
           "exit_hist": [int(v) for v in t[len(METRIC_FIELDS):]]}
    if extra is not None:
        out["extra"] = extra
    return out


def all_gather_values(values: torch.Tensor) -> torch.Tensor:
    """value_net.py:195-201: gather the (n_exit, n_local) delta matrices of all ranks along dim 1."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return values
    dev = collective_device()
    src = values.to(dev) if dev is not None else values
    parts = [torch.zeros_like(src) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, src.contiguous())
    return torch.cat(parts, dim=1).to(values.device)
