"""Multi-GPU job sharding: independent (main.r1cs, trusted set) jobs, one process per GPU, no
data-path collective; the only exchange is an all-reduce (MIN) of the 4-byte verdict/done word
(RCCL over xGMI on GPUs, gloo in the CPU tests).  SURVEY.md §8(e)."""


def assign(weights, n_ranks):
    """Longest-processing-time-first bin packing of job indices onto ranks (deterministic)."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    loads = [0] * n_ranks
    parts = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += weights[i]
    return [sorted(p) for p in parts]


def allreduce_verdict(local_all_good, dist, device="cuda"):
    """AND of the per-rank verdicts via a MIN all-reduce of one int32."""
    import torch
    word = torch.tensor([1 if local_all_good else 0], dtype=torch.int32, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(word, op=dist.ReduceOp.MIN)
    return bool(int(word.item()))
