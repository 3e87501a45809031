"""Host placement of one rank of a multi-GPU run: the CPUs (and with them, by first touch, the memory) next to the rank's GPU.

A host-staged video stream moves every frame through pinned host memory twice (main.cpp:273-275 is the buffer this replaces);
at 8 GPUs x ~45 GB/s each way that is a host-DRAM / socket-interconnect load, and a rank whose pinned ring landed on the other
socket pays for it on every frame.  So a rank pins itself to the CPUs of its GPU's NUMA node BEFORE it allocates the ring
(hipHostMalloc populates the pages at once, on the node of the calling thread) and before it imports anything that starts
threads.  Nothing here needs HIP or torch: the GPU -> NUMA node map is read from sysfs,

  /sys/class/kfd/kfd/topology/nodes/<n>/properties   KFD's node list: GPU nodes in the order the ROCm runtime enumerates them
                                                      (simd_count > 0), each with its drm_render_minor
  /sys/class/drm/renderD<minor>/device/numa_node      the PCI device's NUMA node (-1: the platform does not say)
  /sys/class/drm/renderD<minor>/device/local_cpulist  the CPUs of that node
  /sys/bus/pci/devices/<domain:bus:dev.fn>/...        the same two files by PCI address (what the runtime reports for a device:
                                                      used to CONFIRM the guess once the device is open)

and the visible-device lists of the environment (ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES, plain
index lists) are applied the way the runtime applies them.  Every function takes the sysfs root as an argument: the tests
build a fake tree (tests/test_placement.py)."""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0,1,2,3,8,10,11}; '' -> set()"""
    out = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.update(range(int(a), int(b) + 1))
        else:
            out.add(int(part))
    return out


def format_cpulist(cpus):
    """{0,1,2,3,8} -> '0-3,8'"""
    cs = sorted(cpus)
    runs, i = [], 0
    while i < len(cs):
        j = i
        while j + 1 < len(cs) and cs[j + 1] == cs[j] + 1:
            j += 1
        runs.append(str(cs[i]) if i == j else f"{cs[i]}-{cs[j]}")
        i = j + 1
    return ",".join(runs)


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _props(text):
    d = {}
    for ln in (text or "").splitlines():
        kv = ln.split()
        if len(kv) == 2:
            try:
                d[kv[0]] = int(kv[1])
            except ValueError:
                d[kv[0]] = kv[1]
    return d


def _device_info(devdir):
    """numa node, local CPUs and PCI address of a PCI device directory (a renderD*/device link or a /sys/bus/pci/devices entry)"""
    node = _read(os.path.join(devdir, "numa_node"))
    cpus = _read(os.path.join(devdir, "local_cpulist"))
    try:
        pci = os.path.basename(os.path.realpath(devdir))
    except OSError:
        pci = None
    return {"numa_node": int(node.strip()) if node and node.strip().lstrip("-").isdigit() else None,
            "cpus": parse_cpulist(cpus) if cpus else set(), "pci": pci}


def gpu_nodes(sysfs="/sys"):
    """the GPUs in the order the ROCm runtime enumerates them (KFD topology order), each {kfd_node, render_minor, numa_node, cpus, pci}"""
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        ids = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    out = []
    for n in ids:
        p = _props(_read(os.path.join(base, str(n), "properties")))
        if p.get("simd_count", 0) <= 0:
            continue  # a CPU node
        minor = p.get("drm_render_minor", -1)
        ent = {"kfd_node": n, "render_minor": minor, "numa_node": None, "cpus": set(), "pci": None}
        if isinstance(minor, int) and minor >= 0:
            ent.update(_device_info(os.path.join(sysfs, "class", "drm", f"renderD{minor}", "device")))
        out.append(ent)
    return out


def _index_list(value):
    """'2,3' -> [2, 3]; anything else (UUIDs, empty) -> None: not interpreted"""
    if value is None:
        return None
    try:
        return [int(v) for v in value.split(",") if v.strip() != ""]
    except ValueError:
        return None


def visible_gpus(nodes, env):
    """apply ROCR_VISIBLE_DEVICES (the runtime's filter), then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (HIP's, indices into what
    the runtime left) -- index lists only; an entry out of range ends the list, as in the runtime"""
    cur = list(nodes)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        val = env.get(var)
        if var == "HIP_VISIBLE_DEVICES" and val is None:
            val = env.get("CUDA_VISIBLE_DEVICES")
        if val is None:
            continue
        idx = _index_list(val)
        if idx is None:
            return None  # a form this module does not interpret (UUIDs): no guess
        nxt = []
        for i in idx:
            if i < 0 or i >= len(cur):
                break
            nxt.append(cur[i])
        cur = nxt
    return cur


def plan(device_index, sysfs="/sys", env=None, allowed=None):
    """the placement of the rank that will open HIP device `device_index`: {"numa_node", "cpus" (sorted list, already
    intersected with the CPUs this process may use), "pci", "source"} -- or None when sysfs does not say (no KFD tree, NUMA node
    -1, an empty intersection): the rank then stays where the launcher put it"""
    env = os.environ if env is None else env
    if allowed is None:
        try:
            allowed = os.sched_getaffinity(0)
        except (AttributeError, OSError):
            allowed = None
    vis = visible_gpus(gpu_nodes(sysfs), env)
    if not vis or device_index < 0 or device_index >= len(vis):
        return None
    g = vis[device_index]
    return _finish(g, allowed, "kfd topology order + visible-device lists")


def plan_for_pci(pci_bus_id, sysfs="/sys", allowed=None):
    """the same from the PCI address the runtime reports for an OPEN device ('0000:c1:00.0'): the confirmation of plan()"""
    if allowed is None:
        try:
            allowed = os.sched_getaffinity(0)
        except (AttributeError, OSError):
            allowed = None
    devdir = os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower())
    if not os.path.isdir(devdir):
        return None
    g = _device_info(devdir)
    g["pci"] = pci_bus_id.lower()
    return _finish(g, allowed, "pci address of the open device")


def _finish(g, allowed, source):
    if g["numa_node"] is None or g["numa_node"] < 0 or not g["cpus"]:
        return None
    cpus = set(g["cpus"]) if allowed is None else set(g["cpus"]) & set(allowed)
    if not cpus:
        return None
    return {"numa_node": g["numa_node"], "cpus": sorted(cpus), "pci": g["pci"], "source": source}


def apply(p):
    """pin the calling thread (and every thread it starts from now on) to the plan's CPUs; returns True when done"""
    if not p:
        return False
    try:
        os.sched_setaffinity(0, p["cpus"])
        return True
    except (AttributeError, OSError):
        return False


def describe(p, applied):
    """what goes into the bench line"""
    if not p:
        return {"numa_node": None, "cpus": None, "applied": False, "note": "sysfs names no NUMA node for this GPU (or no CPU of it is ours)"}
    return {"numa_node": p["numa_node"], "cpus": format_cpulist(p["cpus"]), "n_cpus": len(p["cpus"]), "pci": p["pci"], "applied": bool(applied),
            "source": p["source"]}
