"""Host-thread placement for the one-process-per-GPU jobs: a rank's host threads -- the library's worker
pool, its staging copies, torch.distributed's helper threads -- belong on the NUMA node its GPU hangs off.  The reference's
`cards` fan-out (/root/reference/sift4g/src/main.cpp:254-262, database_alignment.cpp:97-104) runs every card from threads of one
process and leaves placement to the OS; with a process per GPU it can be stated.

Everything here reads sysfs only (the KFD topology), so it can run BEFORE anything initialises HIP, and it is best effort: on a
box without the files, or when the affinity mask may not be changed, the caller gets a description of why nothing was done.
"""
import os

_KFD = "/sys/class/kfd/kfd/topology/nodes"


def _props(path):
    out = {}
    try:
        with open(path) as f:
            for line in f:
                k, _, v = line.strip().partition(" ")
                if k:
                    out[k] = v
    except OSError:
        pass
    return out


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_nodes(kfd_root=_KFD):
    """NUMA node of every GPU in the KFD's enumeration order (= HIP's default device order) -> list of int (-1 = unknown).
    A GPU node's NUMA node is the CPU node its first io_link leads to; PCI sysfs is the fallback."""
    try:
        ids = sorted(int(d) for d in os.listdir(kfd_root) if d.isdigit())
    except OSError:
        return []
    nodes = {i: _props(os.path.join(kfd_root, str(i), "properties")) for i in ids}
    cpu_nodes = {i for i, p in nodes.items() if int(p.get("cpu_cores_count", "0") or 0) > 0}
    out = []
    for i in ids:
        p = nodes[i]
        if int(p.get("simd_count", "0") or 0) == 0:
            continue                                    # a CPU-only node
        numa = -1
        links = os.path.join(kfd_root, str(i), "io_links")
        try:
            for l in sorted(os.listdir(links), key=lambda x: int(x) if x.isdigit() else 1 << 30):
                to = int(_props(os.path.join(links, l, "properties")).get("node_to", "-1"))
                if to in cpu_nodes:
                    numa = sorted(cpu_nodes).index(to)  # KFD numbers CPU nodes in NUMA order
                    break
        except OSError:
            pass
        if numa < 0:
            try:
                dom, loc = int(p.get("domain", "0")), int(p.get("location_id", "0"))
                bdf = f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7}"
                with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
                    numa = int(f.read().strip())
            except (OSError, ValueError):
                numa = -1
        out.append(numa)
    return out


def visible_index(local_index):
    """physical GPU index behind HIP device `local_index` when a *_VISIBLE_DEVICES list renumbers the devices"""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            items = [x.strip() for x in v.split(",") if x.strip()]
            if local_index < len(items) and items[local_index].isdigit():
                return int(items[local_index])
            return None                                 # UUIDs or out of range: unknown
    return local_index


def pin_to_gpu(local_index, kfd_root=_KFD, node_root="/sys/devices/system/node"):
    """Restrict this process (and every thread it starts from now on) to the CPUs of the NUMA node of HIP device
    `local_index`.  Call before the first HIP call.  -> dict(describing what was done) for the bench line."""
    info = {"gpu": local_index, "pinned": False}
    if not hasattr(os, "sched_setaffinity"):
        return dict(info, reason="no sched_setaffinity on this platform")
    phys = visible_index(local_index)
    if phys is None:
        return dict(info, reason="device list renumbered by a *_VISIBLE_DEVICES value that is not a list of indices")
    numa = gpu_numa_nodes(kfd_root)
    if phys >= len(numa) or numa[phys] < 0:
        return dict(info, reason="no NUMA node known for this GPU (KFD topology not readable)")
    node = numa[phys]
    try:
        with open(os.path.join(node_root, f"node{node}", "cpulist")) as f:
            cpus = _parse_cpulist(f.read())
    except (OSError, ValueError):
        return dict(info, numa_node=node, reason="cpulist of the node not readable")
    allowed = os.sched_getaffinity(0)
    want = cpus & allowed
    if not want:
        return dict(info, numa_node=node, reason="none of the node's CPUs is in this process's affinity mask")
    if len(want) < 4 and len(allowed) > len(want):
        return dict(info, numa_node=node, reason=f"the node offers {len(want)} CPU(s) only: not worth leaving {len(allowed)} for")
    try:
        os.sched_setaffinity(0, want)
    except OSError as exc:
        return dict(info, numa_node=node, reason=f"sched_setaffinity refused: {exc}")
    return dict(info, pinned=True, numa_node=node, cpus=len(want))
