"""Multi-GPU plumbing: one process per GPU, one independent image stream per rank.
The hot path has NO data-path collective — frames shard embarrassingly — so torch.distributed (RCCL on GPUs,
gloo in the CPU tests) is used only to agree on the timing (MAX over ranks) and to gather throughput counters."""
import os

import torch


def env_ranks():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend, world, rank, local_rank=0, force=False):
    """Returns the torch.distributed module (initialised) or None for a single process.  force: initialise at world size 1 too, so that
    the communicator, the barrier and the collectives below run exactly as at N > 1 (tests/test_gpu_bench.py drives the RCCL path
    that way on the 1-GPU box)."""
    if world <= 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    world = max(world, 1)
    # No device_id here: with it torch builds the RCCL communicator (and RCCL its streams) eagerly, BEFORE the pipeline creates
    # its lane streams, and the runtime's stream -> hardware-queue placement the lanes rely on would shift (NOTES.md §4.5).
    # The communicator is built lazily at the first collective instead: the barrier in front of the timed region.
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def barrier(dist, backend, local_rank=0):
    if dist is None:
        return
    if backend == "nccl":
        dist.barrier(device_ids=[local_rank])
    else:
        dist.barrier()


def agree_max(dist, value, device):
    """MAX of an integer over the ranks (e.g. the repeat count of the timed block: every rank must run the same number of steps)."""
    if dist is None:
        return value
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def stream_first_index(rank, ring):
    """First synthetic-frame index of this rank's stream: ranks never share frames."""
    return rank * ring


def reduce_run(dist, elapsed_s, counters, device):
    """elapsed_s: this rank's wall time of the timed region; counters: list of per-rank additive counters.
    Returns (max elapsed over ranks, element-wise sum of counters over ranks, per-rank counter rows)."""
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    c = torch.tensor(list(counters), dtype=torch.float64, device=device)
    if dist is None:
        return float(t.item()), c.cpu().tolist(), [c.cpu().tolist()]
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rows = [torch.zeros_like(c) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, c)
    total = torch.stack(rows).sum(0)
    return float(t.item()), total.cpu().tolist(), [r.cpu().tolist() for r in rows]


def bind_to_gpu_numa(local_rank, bind=True):
    """Multi-rank runs: keep this rank's host threads on the NUMA node its GPU hangs off (launch threads, the synthesis and parity
    pools), intersected with the CPUs the process may already use.  Best effort.  Returns a dict every rank can put into its
    counters row (bench.py's `per_rank`: the first real 8-GPU record then says where each rank ran): `pci_bus` of the GPU, its
    `numa_node` (-1: not exposed), `host_threads` the rank may run on afterwards, `bound` (1 when the affinity was narrowed) and a
    one-line `desc`."""
    info = {"pci_bus": -1, "numa_node": -1, "host_threads": len(os.sched_getaffinity(0)), "bound": 0, "desc": None}
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        info["pci_bus"] = int(p.pci_bus_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        info["numa_node"] = node
        if node < 0 or not bind:
            return info
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        mine = cpus & os.sched_getaffinity(0)
        if not mine or mine == os.sched_getaffinity(0):
            return info
        os.sched_setaffinity(0, mine)
        info.update(host_threads=len(mine), bound=1, desc="cuda:%d (%s) -> NUMA node %d, %d host threads" % (local_rank, bdf, node, len(mine)))
        return info
    except Exception:
        return info
