"""BASELINE.json config 4: the 67 ecne_circomlib_tests/*.r1cs files, sharded file-per-GPU.

One GPU:   python tests/tools/suite_bench.py [reps]
N GPUs:    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
               --master-port 29511 tests/tools/suite_bench.py [reps]
Every rank takes its share of the files (longest-processing-time-first packing by non-zero count,
ecneproject_amd/sharding.py), solves it as ONE batch launch on its GPU (one workgroup(-group) per
file) and the ranks meet in a single RCCL all-reduce (MIN) of the 4-byte verdict word. Rank 0 prints
the suite's wall time (max over ranks) next to the sequential oracle run file by file on one core."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
import torch.distributed as dist
import ecneproject_amd as E, fixtures, orc
from ecneproject_amd import sharding

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
torch.cuda.set_device(local_rank)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world)
rels = fixtures.circomlib_suite()
files = [E.R1CS(fixtures.path(r)) for r in rels]
weights = [int(sum(f.info.nnz)) + 1 for f in files]
mine = sharding.assign(weights, world)[rank]
systems = [E.System(files[i]) for i in mine]
rows = sum(len(f) for f in files)
stream = torch.cuda.current_stream().cuda_stream
E.solve_batch(systems, device=local_rank, stream=stream, fetch_states=False)          # upload + classify + warm-up
ts = []
for _ in range(reps):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = E.solve_batch(systems, device=local_rank, stream=stream, fetch_states=False) if systems else []
    ok = sharding.allreduce_verdict(all(r.status == 0 for r in res), dist)            # the done word
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ts.append(float(dt.item()))
good = torch.tensor([sum(int(r.function_good) for r in res)], dtype=torch.int64, device="cuda")
if world > 1:
    dist.all_reduce(good)
if rank == 0:
    t_cpu = 0.0
    for r in rels:
        o = orc.run(fixtures.path(r), want_states=False)
        t_cpu += o.summary.t_solve
    best = min(ts)
    print({"files": len(rels), "rows": rows, "n_gpus": world, "files_rank0": len(mine),
           "gpu_wall_ms_best": round(best * 1e3, 2), "gpu_constraints_per_s": round(rows / best),
           "cpu_solve_s_sum_1core": round(t_cpu, 3), "cpu_constraints_per_s": round(rows / t_cpu),
           "all_ok": ok, "verdicts_true": int(good.item())})
if world > 1:
    dist.destroy_process_group()
