"""Multi-GPU partitioning of a batch of independent images.

Images are independent, so N GPUs are N replicas of the decoder, each working on a contiguous
block of the batch: image i goes to rank i // ceil(n / world).  There is no data-path exchange
and therefore no collective; torch.distributed is used only by the harness (barrier, max of
the per-rank wall time, optional gathering of checksums)."""
import math


def shard_bounds(n_images, rank, world):
    """[lo, hi) of the images rank `rank` decodes."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    per = math.ceil(n_images / world) if n_images else 0
    lo = min(n_images, rank * per)
    return lo, min(n_images, lo + per)


def shard(items, rank, world):
    lo, hi = shard_bounds(len(items), rank, world)
    return items[lo:hi]


def max_over_ranks(value, device=None):
    """Max of a python float over all ranks (the bench reports the slowest rank's time)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- host side of a rank: which cores its threads run on ------------------------------------
#
# With images sharded over the GPUs the data path has no exchange, but the ranks share the host: every
# rank parses, preprocesses and stages its own frames ("host preprocess + PCIe decide the
# 8-GPU curve").  Each rank therefore gets a share of the cores of its own GPU's NUMA node -- pinned
# staging memory is then allocated, written and read by the DMA engine next to that GPU -- and sizes
# its thread pool to that share.


def parse_cpu_list(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/nodeN/cpulist)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def cpu_share(available, node_cpus, ranks_on_node, index_on_node):
    """The cores of one rank: the rank's NUMA node's cores (as far as this process may use them) cut evenly
    among the ranks whose GPUs sit on that node; without NUMA information, `available` cut evenly instead.
    available: cores this process may run on; node_cpus: cores of the GPU's node, or None."""
    pool = sorted(set(available) & set(node_cpus)) if node_cpus else []
    if not pool:
        pool = sorted(available)
    ranks_on_node = max(1, ranks_on_node)
    per = max(1, len(pool) // ranks_on_node)
    lo = min(len(pool) - 1, (index_on_node % ranks_on_node) * per) if pool else 0
    return pool[lo:lo + per] or pool


def gpu_numa_node(pci_bus_id):
    """NUMA node of a GPU from sysfs, or None (single-node hosts report -1)."""
    try:
        with open(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/numa_node") as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except (OSError, ValueError):
        return None


def node_cpus(node):
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            return parse_cpu_list(f.read())
    except OSError:
        return None


def bind_rank_to_its_cores(local_rank, world, pci_bus_ids):
    """Restricts this process (and the threads it starts later) to its share of the host and returns
    (cores, numa node or None).  pci_bus_ids: bus id of every local GPU, index = local rank."""
    import os

    available = sorted(os.sched_getaffinity(0))
    nodes = [gpu_numa_node(b) for b in pci_bus_ids]
    mine = nodes[local_rank] if local_rank < len(nodes) else None
    if mine is None:
        share = cpu_share(available, None, world, local_rank)
    else:
        same = [r for r in range(min(world, len(nodes))) if nodes[r] == mine]
        share = cpu_share(available, node_cpus(mine), len(same), same.index(local_rank))
    if world > 1 and share:
        os.sched_setaffinity(0, share)
    return share, mine


def host_numa_nodes(available):
    """[(node, cores of it this process may use)] for every NUMA node that has such cores; [] without sysfs."""
    import glob
    import os
    import re

    nodes = []
    for path in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"), key=lambda p_: int(re.search(r"(\d+)$", p_).group(1))):
        node = int(re.search(r"(\d+)$", path).group(1))
        cpus = node_cpus(node) or []
        mine = sorted(set(cpus) & set(available))
        if mine:
            nodes.append((node, mine))
    return nodes


def bind_rank_round_robin(rank, world):
    """For a host whose GPUs are not there to ask (the one-card box that rehearses an N-rank run): ranks go round robin
    over the NUMA nodes this process may use, and the ranks of a node share its cores evenly.  Returns (cores, node or
    None, text that says what was done)."""
    import os

    available = sorted(os.sched_getaffinity(0))
    nodes = host_numa_nodes(available)
    if not nodes:
        share = cpu_share(available, None, world, rank)
        if world > 1 and share:
            os.sched_setaffinity(0, share)
        return share, None, f"no NUMA information: {len(available)} cores cut evenly among {world} ranks"
    node, cores = nodes[rank % len(nodes)]
    on_node = [r for r in range(world) if r % len(nodes) == rank % len(nodes)]
    share = cpu_share(cores, cores, len(on_node), on_node.index(rank))
    if share:
        os.sched_setaffinity(0, share)
    return share, node, (f"ranks round robin over the {len(nodes)} NUMA nodes this process may use "
                         f"({', '.join(str(len(c)) for _, c in nodes)} cores): rank {rank} on node {node}, {len(share)} cores")


def cpu_quota_cores():
    """Cores' worth of CPU time this process tree may use per second (cgroup v2 cpu.max), or None: no limit known."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        return None if quota == "max" else int(quota) / int(period)
    except (OSError, ValueError):
        return None
