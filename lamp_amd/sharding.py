"""Multi-GPU sharding of candidate pairs (one process per GPU) and the result gather.

Candidate pairs are independent (the reference already treats them so with its thread pool,
loop_closure/src/IcpLoopComputation.cc:327-356), so the data path needs no collective: every
rank aligns its own pairs on its own GPU.  The only exchange is the gather of fixed-size result
records to rank 0, done with torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).
"""
import numpy as np

RESULT_WIDTH = 128  # doubles per pair record


def partition_pairs(pairs, cloud_sizes, world):
    """Greedy longest-processing-time assignment of pairs to ranks by estimated cost
    n_src * n_tgt, with an affinity bonus for ranks that already hold one of the pair's
    keyframes (each keyframe is uploaded once per GPU that needs it).  Deterministic.
    Returns a list of index lists, one per rank."""
    cost = [float(cloud_sizes[a]) * float(cloud_sizes[b]) for a, b in pairs]
    order = sorted(range(len(pairs)), key=lambda i: (-cost[i], i))
    load = [0.0] * world
    held = [set() for _ in range(world)]
    out = [[] for _ in range(world)]
    mean = (sum(cost) / max(len(cost), 1)) or 1.0
    for i in order:
        a, b = pairs[i]
        best, best_score = 0, None
        for r in range(world):
            shared = (a in held[r]) + (b in held[r])
            score = load[r] - 0.25 * mean * shared
            if best_score is None or score < best_score:
                best, best_score = r, score
        out[best].append(i)
        load[best] += cost[i]
        held[best].update((a, b))
    for r in range(world):
        out[r].sort()
    return out


def shard_workload(work, world, rank):
    """One rank's share of ONE candidate list (strong scaling: the reference's pair-level
    parallelism, IcpLoopComputation.cc:327-365, across GPUs instead of pool threads): the indices
    of the pairs rank `rank` aligns and the keys of the keyframes it has to register for them.
    Every rank computes the same partition from the same list."""
    sizes = {k: len(v[0]) for k, v in work.keyframes.items()}
    mine = partition_pairs(work.pairs, sizes, world)[rank]
    keys = sorted({k for i in mine for k in work.pairs[i]})
    return mine, keys


def pack_results(results, n=None):
    """lgicp_result array (the ctypes array lgicp_submit filled, or a list of results) ->
    [n, RESULT_WIDTH] float64 (keys are exact up to 2^53)."""
    import ctypes
    if isinstance(results, ctypes.Array):
        n = len(results) if n is None else n
        rec = np.frombuffer(results, dtype=np.dtype(type(results)._type_), count=n)
        out = np.zeros((n, RESULT_WIDTH), np.float64)
        out[:, 0], out[:, 1] = rec["key_from"], rec["key_to"]
        out[:, 2], out[:, 3], out[:, 4] = rec["status"], rec["iterations"], rec["fitness"]
        out[:, 5:21] = rec["T_icp"]
        out[:, 21:30] = rec["delta_R"]
        out[:, 30:33] = rec["delta_t"]
        out[:, 33:69] = rec["covariance"]
        out[:, 69:105] = rec["information"]
        return out
    out = np.zeros((len(results), RESULT_WIDTH), np.float64)
    for i, r in enumerate(results):
        row = out[i]
        row[0], row[1], row[2], row[3] = r.key_from, r.key_to, r.status, r.iterations
        row[4] = r.fitness
        row[5:21] = r.T_icp[:]
        row[21:30] = r.delta_R[:]
        row[30:33] = r.delta_t[:]
        row[33:69] = r.covariance[:]
        row[69:105] = r.information[:]
    return out


def gather_records(records, dist, rank, world, device=None, equal_counts=False):
    """Gather per-rank [n_r, RESULT_WIDTH] arrays to rank 0 (ranks may hold different n_r;
    equal_counts=True skips the count exchange when every rank is known to hold the same n).
    Returns the concatenated array on rank 0, None elsewhere."""
    import torch
    if dist is None:
        return records
    dev = torch.device("cpu") if device is None else device
    if equal_counts:
        counts = torch.full((world,), records.shape[0], dtype=torch.int64)
    else:
        counts = torch.zeros(world, dtype=torch.int64, device=dev)
        counts[rank] = records.shape[0]
        dist.all_reduce(counts)
        counts = counts.cpu()
    n_max = int(counts.max().item())
    buf = torch.zeros((n_max, RESULT_WIDTH), dtype=torch.float64, device=dev)
    if records.shape[0]:
        buf[: records.shape[0]] = torch.from_numpy(np.ascontiguousarray(records)).to(dev)
    gathered = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if rank != 0:
        return None
    parts = [gathered[r][: int(counts[r].item())].cpu().numpy() for r in range(world)]
    return np.concatenate(parts, axis=0)


class RecordGatherer:
    """The same gather for a loop that sends the same number of records every time: staging
    buffers (pinned host, device, rank 0's landing area) are allocated once, the per-rank
    parts arrive in one contiguous device tensor and reach the host in a single copy.

    counts: records held by each rank (ranks may hold different numbers: every rank sends
    max(counts) rows, rank 0 keeps the first counts[r] of rank r's).  An int means every rank
    holds that many."""

    def __init__(self, dist, rank, world, counts, device):
        import torch
        if isinstance(counts, int):
            counts = [counts] * world
        if len(counts) != world:
            raise ValueError("one count per rank")
        self.dist, self.rank, self.world = dist, rank, world
        self.counts = [int(c) for c in counts]
        self.n = max(self.counts + [1])
        pin = device.type == "cuda"
        self.h_in = torch.zeros((self.n, RESULT_WIDTH), dtype=torch.float64, pin_memory=pin)
        self.d_in = torch.zeros((self.n, RESULT_WIDTH), dtype=torch.float64, device=device)
        self.d_all = self.h_all = None
        self._sent = None  # event after the last gather on this rank (device runs only)
        if rank == 0:
            self.d_all = torch.zeros((world, self.n, RESULT_WIDTH), dtype=torch.float64, device=device)
            self.h_all = torch.zeros((world, self.n, RESULT_WIDTH), dtype=torch.float64, pin_memory=pin)

    def gather(self, records):
        """records: [counts[rank], RESULT_WIDTH] float64.  Returns [sum(counts), RESULT_WIDTH] on
        rank 0 (rank order, then each rank's own order), None elsewhere."""
        import torch
        mine = self.counts[self.rank]
        if records.shape != (mine, RESULT_WIDTH):
            raise ValueError(f"expected {(mine, RESULT_WIDTH)} records, got {records.shape}")
        # the previous call's asynchronous upload and collective may still be reading the
        # staging buffers: wait for them on every rank before rewriting
        if self._sent is not None:
            self._sent.synchronize()
        np.copyto(self.h_in.numpy()[:mine], records)
        self.d_in.copy_(self.h_in, non_blocking=True)
        self.dist.gather(self.d_in, list(self.d_all.unbind(0)) if self.rank == 0 else None, dst=0)
        if self.d_in.is_cuda:
            self._sent = torch.cuda.Event()
            self._sent.record()
        if self.rank != 0:
            return None
        self.h_all.copy_(self.d_all, non_blocking=True)
        if self.d_all.is_cuda:
            torch.cuda.current_stream().synchronize()
        h = self.h_all.numpy()
        if all(c == self.n for c in self.counts):
            return h.reshape(self.world * self.n, RESULT_WIDTH)
        return np.concatenate([h[r, :self.counts[r]] for r in range(self.world)], axis=0)
