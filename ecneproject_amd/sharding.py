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


def allreduce_words(flags, dist, device="cuda"):
    """AND of several per-rank booleans in ONE MIN all-reduce (one int32 per flag: [every job ran, every verdict sound])."""
    import torch
    word = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(word, op=dist.ReduceOp.MIN)
    return [bool(int(x)) for x in word.tolist()]


def max_over_ranks(x, dist, device="cuda"):
    """MAX all-reduce of one float64 (the timed region of the slowest rank is the job's time)."""
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_replica_steps(step, steps, warmup, dist, world, sync=lambda: None, device="cuda"):
    """bench.py's contract for ONE circuit on N GPUs (a single circuit does not shard: every rank solves its own replica, the only exchange
    is the MIN all-reduce of the done / verdict word inside `step`): `warmup` untimed steps, then exactly `steps` steps bracketed by a
    barrier + device synchronisation on both sides, the MAX over ranks of the elapsed time. `step()` returns (result, verdict_word).
    Returns (elapsed_max_s, elapsed_this_rank_s, results, words)."""
    import time
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    results, words = [], []
    for _ in range(steps):
        r, w = step()
        results.append(r)
        words.append(w)
    sync()
    mine = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = max_over_ranks(elapsed, dist, device)
    return elapsed, mine, results, words
