"""leansdr_amd/shard.py — multi-GPU plumbing for independent captures.

The leandvb path shards by capture: every rank owns one GPU and one independent stream, there
is NO data-path collective.  torch.distributed is used only for (1) a barrier before/after the
timed region, (2) the max-over-ranks step time and (3) the sum of samples processed — three scalar
all-reduces per benchmark run, over **gloo** (CPU tensors, TCP on 127.0.0.1): the path has no exchange
step, so RCCL/xGMI is not initialised at all (north_star: "no RCCL").
"""
import os


class Shard:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            if backend not in (None, "gloo"):
                raise ValueError("leansdr_amd.shard: captures share nothing; only the gloo control plane is supported")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            self.device = torch.device("cpu")
            dist.init_process_group("gloo")
            self.dist = dist

    def device_index(self):
        """The GPU of this rank: its LOCAL_RANK, or entry LOCAL_RANK of LSDR_RANK_DEVICES ("0,0": two ranks on GPU 0 — how the
        N > 1 path is exercised, kernels and all, on a one-GPU box: tests/test_gpu_multirank.py)."""
        m = os.environ.get("LSDR_RANK_DEVICES")
        if m:
            ids = [int(v) for v in m.split(",") if v.strip() != ""]
            if self.local_rank < len(ids):
                return ids[self.local_rank]
        return self.local_rank

    def pin_to_gpu_numa(self, pci_bus_id):
        """Pin this process (and the worker threads it starts later) to the CPUs of the NUMA node its GPU hangs off: on an 8-GPU node
        with two sockets a capture's host threads — 16 per GPU in the c1 workload, 128 on the node — otherwise wander across both.
        Returns {"node": n, "cpus": k} or None when the node cannot be told (single-node boxes report −1) or the affinity cannot be set."""
        try:
            with open(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/numa_node") as f:
                node = int(f.read().strip())
            if node < 0:
                return None
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                cpus = set()
                for part in f.read().strip().split(","):
                    lo, _, hi = part.partition("-")
                    cpus.update(range(int(lo), int(hi or lo) + 1))
            allowed = cpus & set(os.sched_getaffinity(0))
            if not allowed:
                return None
            os.sched_setaffinity(0, allowed)
            return {"node": node, "cpus": len(allowed)}
        except (OSError, ValueError, AttributeError):
            return None

    def all_ranks_ok(self, ok):
        """(ranks that passed, ranks): every rank verifies its own captures and contributes its verdict."""
        failed = self.sum_over_ranks(0.0 if ok else 1.0)
        return self.world - int(round(failed)), self.world

    def capture_seed(self, base=1):
        """Every rank demodulates its own capture."""
        return base + self.rank

    def barrier(self):
        if self.dist is None:
            return
        import torch
        t = torch.zeros(1, device=self.device)
        self.dist.all_reduce(t)

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], device=self.device, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], device=self.device, dtype=torch.float64)
        self.dist.all_reduce(t)
        return float(t.item())

    def gather_ints(self, values):
        """Every rank's short list of integers, by rank (control plane only: seeds, counts)."""
        if self.dist is None:
            return [list(values)]
        import torch
        mine = torch.tensor([int(v) for v in values], device=self.device, dtype=torch.int64)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [[int(v) for v in t.tolist()] for t in out]

    def aggregate(self, units_this_rank, seconds_this_rank):
        """Whole-job throughput: units over all ranks ÷ the slowest rank's time."""
        total = self.sum_over_ranks(units_this_rank)
        dt = self.max_over_ranks(seconds_this_rank)
        return total, dt, total / dt

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
