"""Batch front-end: a list of independent verification jobs (main .r1cs + trusted functions) run on the GPUs of one
node. Stands in for the reference's drivers around the solver path -- the CLI loop of src/Ecne.jl:9-37 and the REST
wrapper src/Server.jl:6-30 (whose call signature is stale) -- only as far as they feed SolveConstraintsSymbolic:
a job queue, nothing else.

One process per GPU (torch.distributed, backend "nccl" = RCCL; "gloo" in the CPU tests). Jobs shard at file
granularity: longest-processing-time-first packing by non-zero count (sharding.assign), every rank solves its share as
ONE batch launch (one workgroup(-group) per job, ecne_solve_batch) and the ranks meet in a single all-reduce (MIN) of the
4-byte verdict word -- there is no data-path collective.

    python -m ecneproject_amd.jobs jobs.json            # [{"r1cs": "...", "name": "...", "trusted": [["file", "Name"], ...], "secp_solve": false}, ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m ecneproject_amd.jobs jobs.json
"""
import json
import os
import sys
import time

from . import sharding


class Job:
    def __init__(self, r1cs, name=None, trusted=(), secp_solve=False):
        self.r1cs, self.name = r1cs, name or os.path.basename(r1cs)
        self.trusted = [tuple(t) for t in trusted]           # (file, function name)
        self.secp_solve = bool(secp_solve)

    def build(self, E):
        """readR1CS + the abstraction loop of solveWithTrustedFunctions (:513-544); returns (System, rows of the main file, nnz)"""
        main = E.R1CS(self.r1cs)
        fl = sorted(((n, E.R1CS(f)) for f, n in self.trusted), key=lambda x: -len(x[1]))      # :527
        s = E.System(main)
        for n, f in fl:
            s.abstract(f, n)
        return s, len(main), int(sum(main.info.nnz))


def job_weights(E, jobs):
    """non-zero count of every main file (header + one pass, on every rank: the packing must be the same everywhere)"""
    return [int(sum(E.R1CS(j.r1cs).info.nnz)) + 1 for j in jobs]


class Runner:
    """Builds this rank's share of the jobs once (parse, abstraction, upload happen at the first run) and solves it as
    batch launches. dist = torch.distributed or None (single process)."""

    def __init__(self, jobs, rank=0, world=1, device=0, dist=None, weights=None, E=None):
        if E is None:
            import ecneproject_amd as E
        self.E, self.jobs, self.rank, self.world, self.device, self.dist = E, list(jobs), rank, world, device, dist
        self.weights = weights if weights is not None else job_weights(E, self.jobs)
        self.mine = sharding.assign(self.weights, world)[rank]
        self.systems, self.rows_main = [], 0
        for i in self.mine:
            s, n, _ = self.jobs[i].build(E)
            self.systems.append(s)
            self.rows_main += n
        self.secp = [self.jobs[i].secp_solve for i in self.mine]
        # A share of more jobs than one launch holds (~248 single-workgroup jobs) is solved as several launches, one after the other, and
        # each takes as long as its longest job: the jobs are handed to the engine longest first -- by weight before the first pass, by the
        # jobs' own in-kernel clocks of the last pass afterwards -- so that the long ones share a launch (504 mid-depth circuits: 5.9 -> 3.x ms).
        self.order = sorted(range(len(self.mine)), key=lambda k: -self.weights[self.mine[k]])
        self._passes = 0

    def run(self, stream=None, fetch_states=False, device_for_word="cuda"):
        """one pass over this rank's jobs; returns (results of this rank, verdict word): the word is True iff EVERY job on EVERY rank ran
        (status 0) AND came back sound (function_good) -- what crosses RCCL for the config-5 DAG must be the AND of the verdicts, not "every
        job ran" (SURVEY.md 8e). self.all_ran keeps the weaker fact (every job on every rank reached a verdict) from the same all-reduce."""
        E = self.E
        if hasattr(self.systems[0] if self.systems else None, "set_secp_solve"):
            # one launch for the whole share: every system carries its own secp_solve (:511) -- set once (504 FFI calls per pass were a
            # quarter of a millisecond of a 2 ms pass)
            applied = tuple((id(s), bool(f)) for s, f in zip(self.systems, self.secp))
            if getattr(self, "_secp_applied", None) != applied:          # (re-applied whenever the systems / flags were replaced or extended)
                for s, f in zip(self.systems, self.secp):
                    s.set_secp_solve(f)
                self._secp_applied = applied
            res = [None] * len(self.systems)
            if self.systems:
                out = E.solve_batch([self.systems[k] for k in self.order], device=self.device, stream=stream, fetch_states=fetch_states)
                for k, r in zip(self.order, out):
                    res[k] = r
                if self._passes < 2:          # (the first two passes settle the order: reading 504 jobs' clocks costs more than their launches)
                    try:
                        t = [float(sum(list(r.summary.phase_ms)[:6])) for r in res]
                        self.order = sorted(range(len(res)), key=lambda k: -t[k])
                    except Exception:      # noqa: BLE001  (an engine without per-job clocks: the order by weight stays)
                        pass
                self._passes += 1
        else:
            res = [None] * len(self.systems)
            for flag in (False, True):                   # (an engine without per-system flags: one launch per flag value)
                idx = [k for k, f in enumerate(self.secp) if f == flag]
                if idx:
                    out = E.solve_batch([self.systems[k] for k in idx], secp_solve=flag, device=self.device, stream=stream,
                                        fetch_states=fetch_states)
                    for k, r in zip(idx, out):
                        res[k] = r
        ran = all(r.status == 0 for r in res)
        sound = all(r.status == 0 and bool(r.function_good) for r in res)
        if self.dist is not None:
            ran, sound = sharding.allreduce_words([ran, sound], self.dist, device=device_for_word)
        self.all_ran = ran
        return res, sound


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    import torch
    import torch.distributed as dist
    import ecneproject_amd as E
    with open(argv[0]) as f:
        jobs = [Job(j["r1cs"], j.get("name"), j.get("trusted", ()), j.get("secp_solve", False)) for j in json.load(f)]
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    r = Runner(jobs, rank, world, local, dist if world > 1 else None)
    t = time.perf_counter()
    res, ok = r.run(stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    for i, g in zip(r.mine, res):
        print(json.dumps({"job": jobs[i].name, "rank": rank, "status": g.status, "sound": bool(g.function_good),
                          "unique": int(g.summary.unique_nontrivial), "of": int(g.summary.n_nontrivial),
                          "targets": "%d/%d" % (g.summary.unique_targets, g.summary.n_targets), "device_ms": round(g.summary.device_ms, 3)}))
    if rank == 0:
        print(json.dumps({"jobs": len(jobs), "n_gpus": world, "all_ran": r.all_ran, "all_sound": ok, "wall_s": round(dt, 4)}))
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
