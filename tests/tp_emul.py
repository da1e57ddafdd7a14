"""In-process stand-in for a torch.distributed group, for the tensor-parallel GPU tests on a ONE-GPU box: the ranks are threads
that take turns (exactly one runs at a time, so the native library's per-device scratch buffers are never shared by two
half-finished op sequences), and a collective is the point where a rank hands the turn to the next one.  The last rank to
arrive computes the result for everybody.  Sums are taken in fp32 and rounded once (RCCL's ring adds fp16 pairwise: tests
compare within tolerance, not bitwise).  Test infrastructure only."""

import threading

import torch


class LocalGroup:
    def __init__(self, world):
        self.world = world
        self.cv = threading.Condition()
        self.turn = 0
        self.gen = 0
        self.slots = {}
        self.done = [False] * world
        self.failed = None

    def comm(self, rank):
        return _LocalComm(self, rank)

    # -- scheduling ---------------------------------------------------------------------------------------------------------
    def _wait_turn(self, rank):
        while self.turn != rank:
            if self.failed is not None:
                raise RuntimeError("another rank failed") from self.failed
            self.cv.wait(timeout=1.0)

    def _next_turn(self, rank):
        for step in range(1, self.world + 1):
            r = (rank + step) % self.world
            if not self.done[r]:
                self.turn = r
                break
        self.cv.notify_all()

    def run(self, fns):
        """fns[r]() is rank r's program; returns their results."""
        results = [None] * self.world

        def worker(r):
            try:
                with self.cv:
                    self._wait_turn(r)
                results[r] = fns[r]()
            except BaseException as e:               # noqa: BLE001 -- reported to the caller
                with self.cv:
                    if self.failed is None:
                        self.failed = e
            finally:
                with self.cv:
                    self.done[r] = True
                    self._next_turn(r)

        threads = [threading.Thread(target=worker, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if self.failed is not None:
            raise self.failed
        return results

    def collective(self, rank, tensor, finish):
        """Deposit `tensor`; when every rank has, `finish(list of tensors by rank)` runs once (on the last arriver)."""
        with self.cv:
            self.slots[rank] = tensor
            if len(self.slots) == self.world:
                torch.cuda.synchronize() if tensor.is_cuda else None
                finish([self.slots[r] for r in range(self.world)])
                self.slots = {}
                self.gen += 1
                self.turn = 0
                self.cv.notify_all()
            else:
                gen = self.gen
                self._next_turn(rank)
                while self.gen == gen:
                    if self.failed is not None:
                        raise RuntimeError("another rank failed") from self.failed
                    self.cv.wait(timeout=1.0)
            self._wait_turn(rank)


class _LocalComm:
    """The two calls exllama_amd.tp.TensorParallel makes on a torch.distributed module."""

    def __init__(self, group, rank):
        self.group, self.rank = group, rank
        self._gathered = None

    def all_reduce(self, t):
        def finish(ts):
            total = torch.stack([x.float() for x in ts]).sum(0).to(ts[0].dtype)
            for x in ts:
                x.copy_(total)
        self.group.collective(self.rank, t, finish)

    def all_gather(self, parts, t):
        box = {}

        def finish(ts):
            box["all"] = [x.clone() for x in ts]
            self.group._last_gather = box["all"]
        self.group.collective(self.rank, t, finish)
        for dst, src in zip(parts, self.group._last_gather):
            dst.copy_(src)
