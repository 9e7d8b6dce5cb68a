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
