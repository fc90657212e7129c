"""One process per GPU on a multi-socket host: keep a rank's host side next to its GPU.

A rank's page-locked ring (three slots of a contig's transfer form, 0.4-0.6 GB each) is read by its GPU's copy engine at ~55 GB/s and
written by its ingest threads (BGZF inflate, record walk, wire build).  Eight ranks on a two-socket node move 8 x 55 GB/s of wire reads:
if the pages sit on the other socket, all of that crosses the inter-socket links on top of PCIe.  `bind_rank` restricts the process to
its share of the CPUs of the GPU's NUMA node BEFORE the engine allocates pinned memory or starts threads -- first touch then places the
pages on that node, and the ingest threads (bam.usable_cpus: the affinity mask) shrink to the share instead of 8 ranks x all cores.

No libnuma: sysfs only; every step degrades to "not bound" with the reason in the returned dict (containers hide sysfs, single-node
hosts report node -1).  NC_NUMA_BIND=0 turns it off."""
from __future__ import annotations

import os


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


def gpu_pci_address(device_index):
    """'dddd:bb:dd.f' of a visible GPU, or None"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def gpu_numa_node(device_index, sysfs="/sys"):
    """NUMA node of the GPU's PCIe root, or -1 (unknown / single node)"""
    addr = gpu_pci_address(device_index)
    if not addr:
        return -1
    try:
        return int(open(os.path.join(sysfs, "bus/pci/devices", addr, "numa_node")).read())
    except (OSError, ValueError):
        return -1


def node_cpus(node, sysfs="/sys"):
    """CPUs of a NUMA node, the hardware threads of one core next to each other (a node lists cores 0-63 and then their SMT siblings 128-191:
    contiguous shares of THAT order would hand two ranks the two threads of the same cores)"""
    try:
        cpus = _parse_cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read())
    except (OSError, ValueError):
        return []

    def core_key(c):
        try:
            sib = _parse_cpulist(open(os.path.join(sysfs, "devices/system/cpu/cpu%d/topology/thread_siblings_list" % c)).read())
            return (min(sib), c)
        except (OSError, ValueError):
            return (c, c)
    return sorted(cpus, key=core_key)


def cpu_share(cpus, k, n):
    """the k-th of n contiguous shares of a CPU list (neighbouring CPUs share caches), never empty when cpus is not"""
    n = max(1, n)
    lo, hi = k * len(cpus) // n, (k + 1) * len(cpus) // n
    return cpus[lo:hi] or cpus[k % len(cpus):k % len(cpus) + 1]


def plan_binding(gpu_nodes, local_rank, allowed, cpus_of_node):
    """pure planning step (tested on CPU): gpu_nodes[i] = NUMA node of local GPU i, `allowed` = CPUs this process may use now.
    -> (cpus to bind to, note).  Ranks whose GPUs hang off one node split that node's allowed CPUs between them."""
    node = gpu_nodes[local_rank] if 0 <= local_rank < len(gpu_nodes) else -1
    if node < 0:
        return None, "the GPU's NUMA node is unknown (single-node host or hidden sysfs)"
    cpus = [c for c in cpus_of_node(node) if c in allowed]
    if not cpus:
        return None, "no allowed CPU on NUMA node %d" % node
    peers = [i for i, nd in enumerate(gpu_nodes) if nd == node]
    return cpu_share(cpus, peers.index(local_rank), len(peers)), "NUMA node %d, share %d of %d" % (node, peers.index(local_rank) + 1, len(peers))


def bind_rank(device_index, local_rank=None, local_world=None):
    """Bind this process (and every thread it starts later) to its share of the CPUs next to GPU `device_index`.  Call before the engine
    allocates page-locked memory.  -> dict(bound, cpus, node, note)"""
    if os.environ.get("NC_NUMA_BIND") == "0" or not hasattr(os, "sched_setaffinity"):
        return dict(bound=False, cpus=None, node=-1, note="off")
    local_rank = int(os.environ.get("LOCAL_RANK", 0)) if local_rank is None else local_rank
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", 1)) if local_world is None else local_world
    try:
        import torch
        n_dev = torch.cuda.device_count()
    except Exception:
        n_dev = 0
    # which GPU does each local rank drive?  one process per GPU: local rank i -> device i (engine.local_device), this one -> device_index
    nodes = [gpu_numa_node(i) for i in range(n_dev)]
    if not (0 <= device_index < n_dev):
        return dict(bound=False, cpus=None, node=-1, note="no such device")
    gpu_nodes = [nodes[i % n_dev] for i in range(max(local_world, local_rank + 1))]
    gpu_nodes[local_rank] = nodes[device_index]
    allowed = set(os.sched_getaffinity(0))
    cpus, note = plan_binding(gpu_nodes, local_rank, allowed, node_cpus)
    if not cpus:
        return dict(bound=False, cpus=None, node=nodes[device_index], note=note)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        return dict(bound=False, cpus=None, node=nodes[device_index], note="sched_setaffinity: %s" % e)
    return dict(bound=True, cpus=cpus, node=nodes[device_index], note=note)
