"""The N>1 path on CPU: two gloo ranks shard a candidate list, each aligns its shard (with the
CPU oracle standing in for the GPU, which this container does not have) and rank 0 gathers the
fixed-size records — the same partition_pairs / pack / gather code bench.py runs over RCCL."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_a_balanced_cover():
    from lamp_amd.sharding import partition_pairs
    rng = np.random.default_rng(0)
    keys = list(range(40))
    sizes = {k: int(rng.integers(8000, 70000)) for k in keys}
    pairs = [(int(rng.integers(0, 40)), int(rng.integers(0, 40))) for _ in range(500)]
    for world in (1, 2, 4, 8):
        parts = partition_pairs(pairs, sizes, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(500))                       # every pair exactly once
        load = [sum(sizes[pairs[i][0]] * sizes[pairs[i][1]] for i in p) for p in parts]
        assert max(load) <= 1.15 * (sum(load) / world)         # LPT balance
        assert partition_pairs(pairs, sizes, world) == parts   # deterministic
    assert partition_pairs([], sizes, 4) == [[], [], [], []]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from lamp_amd import capi
    from lamp_amd.sharding import gather_records, pack_results, partition_pairs
    from lamp_amd.workload import Workload
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    work = Workload("C2", pairs=5, points=1500, keyframes=8)   # same list on every rank
    sizes = {k: len(v[0]) for k, v in work.keyframes.items()}
    mine = partition_pairs(work.pairs, sizes, world)[rank]
    results = []
    for i in mine:
        a, b = work.pairs[i]
        src, sn = work.keyframes[a]
        tgt, tn = work.keyframes[b]
        ref = O.perform_alignment(O.base_icp_params(init_method=O.INIT_IDENTITY, solver=O.SOLVER_GN),
                                  src, sn, tgt, tn, src, tn, O.pose_from_quat([1, 0, 0, 0], [0, 0, 0]))
        r = capi.Result()
        r.key_from, r.key_to, r.status, r.iterations = a, b, ref["status"], ref["iterations"]
        r.fitness = ref["fitness"]
        r.T_icp[:] = ref["T_icp"].reshape(-1).tolist()
        r.delta_t[:] = ref["delta_t"].tolist()
        results.append(r)
    rec = gather_records(pack_results(results), dist, rank, world)
    if rank == 0:
        np.save(out_path, rec)
    # the fixed-count gatherer the benchmark uses: same buffers on every call
    import torch
    from lamp_amd.sharding import RESULT_WIDTH, RecordGatherer
    g = RecordGatherer(dist, rank, world, 3, torch.device("cpu"))
    for call in range(2):
        mine3 = np.full((3, RESULT_WIDTH), 100.0 * call + 10.0 * rank) + np.arange(3)[:, None]
        allr = g.gather(mine3)
        if rank == 0:
            np.save(out_path + f".fixed{call}.npy", allr.copy())
        else:
            assert allr is None
    # strong scaling, as `bench.py --mode strong` runs it: ONE candidate list cut over the ranks,
    # every rank holds only the keyframes its own pairs touch, variable record counts per rank
    import torch
    from lamp_amd.sharding import shard_workload
    one = Workload("C3", pairs=7, points=1200, keyframes=6)
    mine, keys = shard_workload(one, world, rank)
    held = {k: one.keyframes[k] for k in keys}          # what this rank would register
    recs = []
    for i in mine:
        a, b = one.pairs[i]
        src, sn = held[a]                               # KeyError = a keyframe the shard lacks
        tgt, tn = held[b]
        ref = O.perform_alignment(O.base_icp_params(init_method=O.INIT_IDENTITY, solver=O.SOLVER_GN),
                                  src, sn, tgt, tn, src, tn, O.pose_from_quat([1, 0, 0, 0], [0, 0, 0]))
        r = capi.Result()
        r.key_from, r.key_to, r.status, r.iterations = a, b, ref["status"], ref["iterations"]
        r.fitness = ref["fitness"]
        recs.append(r)
    t = torch.zeros(world, dtype=torch.int64)
    t[rank] = len(mine)
    dist.all_reduce(t)
    counts = [int(v) for v in t]
    gs = RecordGatherer(dist, rank, world, counts, torch.device("cpu"))
    for call in range(2):                               # the same buffers on every step
        allr = gs.gather(pack_results(recs) if recs else np.zeros((0, RESULT_WIDTH)))
        if rank == 0:
            np.save(out_path + f".strong{call}.npy", allr.copy())
            np.save(out_path + ".strong_counts.npy", np.array(counts))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "gathered.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    rec = np.load(out)
    sys.path.insert(0, ROOT)
    from lamp_amd.workload import Workload
    work = Workload("C2", pairs=5, points=1500, keyframes=8)
    assert rec.shape == (5, 128)
    got = sorted((int(r[0]), int(r[1])) for r in rec)
    assert got == sorted(work.pairs)
    assert np.all(rec[:, 2] == 0) and np.all(rec[:, 4] > 0)
    for call in range(2):
        fixed = np.load(out + f".fixed{call}.npy")
        assert fixed.shape == (6, 128)
        assert fixed[:, 0].tolist() == [100.0 * call + 10.0 * r + k for r in range(2) for k in range(3)]
    # strong mode: every pair of the one list exactly once, whatever the per-rank counts were
    one = Workload("C3", pairs=7, points=1200, keyframes=6)
    counts = np.load(out + ".strong_counts.npy")
    assert counts.sum() == 7 and len(counts) == 2
    for call in range(2):
        rec = np.load(out + f".strong{call}.npy")
        assert rec.shape == (7, 128)
        assert sorted((int(r[0]), int(r[1])) for r in rec) == sorted(one.pairs)
        assert np.all(rec[:, 2] == 0) and np.all(rec[:, 4] > 0)
