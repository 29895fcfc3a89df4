"""Host-side placement for one-process-per-GPU runs: the CPU threads of a rank (its Python thread, the MapPool loader and
writer threads it starts, the page-locked staging buffers they touch first) are kept on the NUMA node its GPU hangs off,
so that N ranks on one node do not queue on one socket's memory controller while the kernels are being measured.

    pin_to_gpu_node(local_rank, world) -> dict describing what was done (also when nothing could be done)

Linux sysfs only: /sys/bus/pci/devices/<bdf>/numa_node names the node, /sys/devices/system/node/node<N>/cpulist its cores.
When the platform reports no node (-1: single socket, or a VM) the cores this process may run on are cut into `world`
equal shares instead, so that ranks at least do not share cores.  Never raises: a rank that cannot be pinned runs unpinned."""
import os
from pathlib import Path


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_pci_address(local_rank, lib=None):
    """'0000:c1:00.0' of HIP device `local_rank` as libuncalled_hip.so numbers the devices (unc_device_pci_address: the ordinal a
    worker passes as --device, under whatever HIP_VISIBLE_DEVICES / ROCR filter is in force), or None when the library cannot
    say (then the caller falls back to equal core shares, and says so)."""
    try:
        import ctypes as C
        from . import capi
        L = lib or capi.load()
        buf = C.create_string_buffer(32)
        L.unc_device_pci_address.argtypes = [C.c_int, C.c_char_p, C.c_int]
        if L.unc_device_pci_address(int(local_rank), buf, 32) != 0:
            return None
        return buf.value.decode() or None
    except Exception:
        return None


def node_of(bdf, sysfs="/sys"):
    try:
        return int((Path(sysfs) / "bus" / "pci" / "devices" / bdf / "numa_node").read_text())
    except Exception:
        return -1


def cpus_of_node(node, sysfs="/sys"):
    try:
        return parse_cpulist((Path(sysfs) / "devices" / "system" / "node" / f"node{node}" / "cpulist").read_text())
    except Exception:
        return []


def plan(local_rank, world, allowed, bdf=None, sysfs="/sys"):
    """The cores rank `local_rank` of `world` should run on, given the cores it is allowed now: (cpus, how)"""
    allowed = sorted(allowed)
    node = node_of(bdf, sysfs) if bdf else -1
    if node >= 0:
        cpus = [c for c in cpus_of_node(node, sysfs) if c in set(allowed)]
        if cpus:
            return cpus, f"numa node {node} of {bdf}"
    if world > 1 and len(allowed) >= world:
        share = len(allowed) // world
        return allowed[local_rank * share:(local_rank + 1) * share], f"no numa node reported: share {local_rank} of {world} of the allowed cores"
    return allowed, "left as is"


def pin_to_gpu_node(local_rank, world, sysfs="/sys"):
    try:
        allowed = os.sched_getaffinity(0)
        bdf = gpu_pci_address(local_rank)
        cpus, how = plan(local_rank, world, allowed, bdf, sysfs)
        if set(cpus) != set(allowed):
            os.sched_setaffinity(0, cpus)
        if bdf is None:
            how += " (the GPU's PCI address could not be read)"
        return {"gpu": bdf, "cpus": len(cpus), "first_cpu": cpus[0] if cpus else None, "how": how}
    except Exception as e:      # placement is an optimisation, never a reason to stop
        return {"gpu": None, "cpus": None, "how": f"not pinned: {e!r}"}
