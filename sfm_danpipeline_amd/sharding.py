"""Multi-GPU partitioning of the hot path (one process per GPU).

* Matching: image pairs are independent units (reference src/Sfm.cpp:511-515) -> the pair list
  is dealt to ranks by estimated cost; no collective on the data path, the host concatenates
  per-rank match lists back into pair order, so results do not depend on the GPU count.
* Bundle adjustment: points (with all their observations) are block-partitioned, cameras and
  the shared focal replicated; the only exchange is the sum of the reduced camera system per
  LM iteration (`AllReduce`, RCCL over xGMI through torch.distributed).

Pure host logic (numpy / torch.distributed); exercised on CPU with the gloo backend in tests.
"""
import numpy as np


# ----------------------------------------------------------------------------- matching
def shard_pairs(pairs, n_rows, world):
    """Deterministic longest-processing-time split of the pair list by cost Nq*Nt.
    Returns a list (one entry per rank) of index arrays into `pairs`, each ascending."""
    pairs = np.asarray(pairs, np.int64).reshape(-1, 2)
    n_rows = np.asarray(n_rows, np.int64)
    cost = n_rows[pairs[:, 0]] * n_rows[pairs[:, 1]]
    order = np.lexsort((np.arange(len(pairs)), -cost))  # heaviest first, index as tie-break
    load = np.zeros(world, np.int64)
    owner = np.empty(len(pairs), np.int64)
    for p in order:
        r = int(np.argmin(load))  # first minimum: deterministic
        owner[p] = r
        load[r] += max(int(cost[p]), 1)
    return [np.nonzero(owner == r)[0] for r in range(world)]


def merge_pair_results(shards, per_rank, n_pairs):
    """Inverse of shard_pairs for results: per_rank[r] = (counts, q, t, d) of rank r's pairs in
    its shard order; returns (counts, q, t, d) in global pair order."""
    counts = np.zeros(n_pairs, np.int32)
    lists = [None] * n_pairs
    for idx, (cnt, q, t, d) in zip(shards, per_rank):
        off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        for k, p in enumerate(idx):
            counts[p] = cnt[k]
            lists[p] = (q[off[k]:off[k + 1]], t[off[k]:off[k + 1]], d[off[k]:off[k + 1]])
    qs = [l[0] for l in lists if l is not None]
    ts = [l[1] for l in lists if l is not None]
    ds = [l[2] for l in lists if l is not None]
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return counts, cat(qs, np.int32), cat(ts, np.int32), cat(ds, np.float32)


# ----------------------------------------------------------------------------- bundle adjustment
def point_block(n_pt, rank, world):
    """Contiguous block [lo, hi) of points owned by `rank`."""
    base, rem = divmod(int(n_pt), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_ba_problem(obs_cam, obs_pt, obs_xy, pts3, rank, world):
    """This rank's slice: its point block (re-indexed from 0) and the observations of those
    points, original observation order preserved.  Returns dict(lo, hi, obs_cam, obs_pt, obs_xy, pts)."""
    obs_cam = np.asarray(obs_cam, np.int32)
    obs_pt = np.asarray(obs_pt, np.int32)
    obs_xy = np.asarray(obs_xy, np.float64).reshape(-1, 2)
    pts3 = np.asarray(pts3, np.float64).reshape(-1, 3)
    lo, hi = point_block(pts3.shape[0], rank, world)
    sel = (obs_pt >= lo) & (obs_pt < hi)
    return dict(lo=lo, hi=hi, obs_cam=np.ascontiguousarray(obs_cam[sel]),
                obs_pt=np.ascontiguousarray(obs_pt[sel] - lo), obs_xy=np.ascontiguousarray(obs_xy[sel]),
                pts=np.ascontiguousarray(pts3[lo:hi]))


class TorchAllReduce:
    """Sum-all-reduce of a raw device (or host) float64 buffer through torch.distributed.
    On GPU the buffer is wrapped without a copy via __cuda_array_interface__ and reduced by
    RCCL (backend "nccl"); the tensor views are cached per (pointer, count)."""

    def __init__(self, group=None, device="cuda"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.device = torch, dist, group, device
        self._views = {}

    def _view(self, ptr, count):
        key = (ptr, count)
        v = self._views.get(key)
        if v is None:
            class _Raw:
                pass
            raw = _Raw()
            raw.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False),
                                            "version": 2, "strides": None}
            v = self.torch.as_tensor(raw, device=self.device)
            self._views[key] = v
        return v

    def __call__(self, ptr, count):
        self.dist.all_reduce(self._view(ptr, count), op=self.dist.ReduceOp.SUM, group=self.group)


class StagedAllReduce(TorchAllReduce):
    """The same sum through a HOST staging copy and whatever backend the process group has (gloo):
    for functional runs of the N>1 path where RCCL cannot be used -- two ranks sharing one device
    (a single-GPU box).  Synchronous; not a performance path."""

    def __call__(self, ptr, count):
        self.counts = getattr(self, "counts", [])
        self.counts.append(int(count))                  # (tests look at the sizes of the exchanges)
        v = self._view(ptr, count)
        self.torch.cuda.synchronize()
        h = v.cpu()
        self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
        v.copy_(h)
        self.torch.cuda.synchronize()


class InProcessRanks:
    """N LOGICAL ranks of the sharded bundle adjustment in one process on one device, a host thread per rank: the sum of the
    ranks' buffers is formed on the device, in rank order, by whichever thread arrives last at the meeting point -- every rank
    gets the same bits back, as after an RCCL all-reduce.  For functional runs of the world = 4 / 8 split on a one-GPU box
    (tests, bench.py's measured shard times); synchronous, not a performance path."""

    def __init__(self, world, device="cuda:0"):
        import threading
        import torch
        self.torch, self.world = torch, int(world)
        self._barrier = threading.Barrier(self.world)
        self._bufs = [None] * self.world
        self._views = TorchAllReduce(device=device)
        self.counts = [[] for _ in range(self.world)]  # (tests look at the sizes of the exchanges)

    def allreduce(self, rank, ctx):
        """The callable rank `rank` hands to BaProblem.set_allreduce; `ctx` is that rank's Context (its stream is drained
        before the buffers meet)."""
        def fn(ptr, count):
            ctx.synchronize()
            self._bufs[rank] = (int(ptr), int(count))
            self.counts[rank].append(int(count))
            if self._barrier.wait() == 0:  # (one thread of the party adds; which one does not matter: the order is the ranks')
                views = [self._views._view(p, c) for p, c in self._bufs]
                assert len({c for _, c in self._bufs}) == 1, f"ranks disagree on an exchange: {self._bufs}"
                tot = views[0].clone()
                for v in views[1:]:
                    tot += v
                for v in views:
                    v.copy_(tot)
                self.torch.cuda.synchronize()
            self._barrier.wait()
        return fn

    def run(self, target):
        """target(rank) on a thread per rank; returns the list of results in rank order, re-raises the first failure."""
        import threading
        out, err = [None] * self.world, [None] * self.world

        def body(r):
            try:
                out[r] = target(r)
            except BaseException as e:  # noqa: BLE001 -- a failed rank must not leave the others waiting
                err[r] = e
                self._barrier.abort()
        ts = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for e in err:
            if e is not None and not isinstance(e, __import__("threading").BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return out
