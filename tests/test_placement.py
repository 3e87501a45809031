"""Host placement of a multi-GPU rank (watermarking-gpu_amd/placement.py): from a faked sysfs tree -- two sockets, eight GPUs
in KFD order, two CPU nodes in front of them as on a real host -- the rank that opens HIP device r must pick the CPUs of THAT
GPU's NUMA node, under the visible-device lists the runtime honours, intersected with the CPUs the process may use; and leave
the process alone when sysfs does not say.  (CPU-only: nothing here touches HIP or torch.)"""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    spec = importlib.util.spec_from_file_location("wm_placement", os.path.join(ROOT, "watermarking-gpu_amd", "placement.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


PL = load()
# socket 0: CPUs 0-63 + 128-191 (SMT siblings), socket 1: 64-127 + 192-255; GPUs 0-3 on socket 0, 4-7 on socket 1 -- but the
# render minors are NOT in GPU order (as on real hosts), so a card-number shortcut would pick the wrong socket
NODE_CPUS = {0: "0-63,128-191", 1: "64-127,192-255"}
GPUS = [  # (kfd node, render minor, numa node, pci address)
    (2, 129, 0, "0000:05:00.0"), (3, 128, 0, "0000:15:00.0"), (4, 131, 0, "0000:65:00.0"), (5, 130, 0, "0000:75:00.0"),
    (6, 133, 1, "0000:85:00.0"), (7, 132, 1, "0000:95:00.0"), (8, 135, 1, "0000:e5:00.0"), (9, 134, 1, "0000:f5:00.0"),
]


def fake_sysfs(root, gpus=GPUS, numa_override=None):
    def w(path, text):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
    nodes = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    for n in (0, 1):  # the two CPU nodes come first and have no SIMDs
        w(os.path.join(nodes, str(n), "properties"), "cpu_cores_count 64\nsimd_count 0\ndrm_render_minor -1\n")
    for kfd, minor, numa, pci in gpus:
        w(os.path.join(nodes, str(kfd), "properties"), f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {minor}\nlocation_id 1234\n")
        dev = os.path.join(root, "bus", "pci", "devices", pci)
        node = numa if numa_override is None else numa_override
        w(os.path.join(dev, "numa_node"), f"{node}\n")
        w(os.path.join(dev, "local_cpulist"), (NODE_CPUS[numa] if node >= 0 else "0-255") + "\n")
        os.makedirs(os.path.join(root, "class", "drm", f"renderD{minor}"), exist_ok=True)
        os.symlink(dev, os.path.join(root, "class", "drm", f"renderD{minor}", "device"))
    return root


def test_cpulist_round_trip():
    assert PL.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert PL.parse_cpulist("") == set()
    assert PL.format_cpulist({0, 1, 2, 3, 8, 10, 11}) == "0-3,8,10-11"


def test_each_rank_gets_the_cpus_of_its_own_gpus_socket(tmp_path):
    sysfs = fake_sysfs(str(tmp_path))
    allowed = set(range(256))
    for r in range(8):
        p = PL.plan(r, sysfs=sysfs, env={}, allowed=allowed)
        want = 0 if r < 4 else 1
        assert p["numa_node"] == want and p["cpus"] == sorted(PL.parse_cpulist(NODE_CPUS[want])), (r, p)
        assert p["pci"] == GPUS[r][3]
        # the confirmation by PCI address (what the runtime reports for the open device) agrees
        q = PL.plan_for_pci(GPUS[r][3].upper(), sysfs=sysfs, allowed=allowed)
        assert q["numa_node"] == p["numa_node"] and q["cpus"] == p["cpus"]
    assert PL.plan(8, sysfs=sysfs, env={}, allowed=allowed) is None  # no such device


def test_visible_device_lists_are_applied_like_the_runtime_does(tmp_path):
    sysfs = fake_sysfs(str(tmp_path))
    allowed = set(range(256))
    # ROCR list first (the runtime's), HIP list indexes into what is left: device 0 of the process is GPU 5, device 1 is GPU 2
    env = {"ROCR_VISIBLE_DEVICES": "2,5,7", "HIP_VISIBLE_DEVICES": "1,0"}
    assert PL.plan(0, sysfs=sysfs, env=env, allowed=allowed)["pci"] == GPUS[5][3]
    assert PL.plan(0, sysfs=sysfs, env=env, allowed=allowed)["numa_node"] == 1
    assert PL.plan(1, sysfs=sysfs, env=env, allowed=allowed)["pci"] == GPUS[2][3]
    assert PL.plan(2, sysfs=sysfs, env=env, allowed=allowed) is None
    # CUDA_VISIBLE_DEVICES stands in for HIP_VISIBLE_DEVICES when that is unset; an out-of-range entry ends the list
    assert PL.plan(0, sysfs=sysfs, env={"CUDA_VISIBLE_DEVICES": "6"}, allowed=allowed)["numa_node"] == 1
    assert PL.plan(1, sysfs=sysfs, env={"HIP_VISIBLE_DEVICES": "3,9,1"}, allowed=allowed) is None
    # a UUID list is not interpreted: no guess rather than a wrong one
    assert PL.plan(0, sysfs=sysfs, env={"ROCR_VISIBLE_DEVICES": "GPU-abcdef0123456789"}, allowed=allowed) is None


def test_intersection_with_the_cpus_of_the_job(tmp_path):
    sysfs = fake_sysfs(str(tmp_path))
    # a job confined to 16 CPUs of socket 1: a rank on a socket-1 GPU keeps the 16, a rank on a socket-0 GPU is left alone
    allowed = set(range(64, 80))
    assert PL.plan(6, sysfs=sysfs, env={}, allowed=allowed)["cpus"] == list(range(64, 80))
    assert PL.plan(1, sysfs=sysfs, env={}, allowed=allowed) is None


def test_no_numa_information_means_no_pinning(tmp_path):
    assert PL.plan(0, sysfs=str(tmp_path / "empty"), env={}, allowed={0, 1}) is None          # no KFD tree (this container)
    sysfs = fake_sysfs(str(tmp_path / "flat"), numa_override=-1)                              # single-node platform: numa_node = -1
    assert PL.plan(0, sysfs=sysfs, env={}, allowed=set(range(256))) is None
    assert PL.apply(None) is False
    assert PL.describe(None, False)["applied"] is False


def test_apply_pins_the_calling_process(tmp_path):
    """in a child process (the affinity of the test runner stays untouched): a plan restricted to one allowed CPU is applied"""
    code = (
        "import importlib.util, os, sys\n"
        "spec = importlib.util.spec_from_file_location('p', sys.argv[1]); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        "cpu = sorted(os.sched_getaffinity(0))[-1]\n"
        "ok = m.apply({'numa_node': 0, 'cpus': [cpu], 'pci': None, 'source': 'test'})\n"
        "print(ok, os.sched_getaffinity(0) == {cpu})\n")
    out = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "watermarking-gpu_amd", "placement.py")], capture_output=True, text=True, timeout=60)
    assert out.stdout.split() == ["True", "True"], out.stdout + out.stderr


def test_bench_rank_prologue_uses_it_before_torch(tmp_path):
    """bench.py's rank prologue (place_rank) runs before torch is imported and reports what it did; with a faked sysfs root the
    rank of LOCAL_RANK 5 lands on socket 1"""
    sysfs = fake_sysfs(str(tmp_path))
    code = (
        "import os, sys, json\n"
        "sys.argv = ['bench.py']\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('bench', sys.argv[0] if False else os.path.join(%r, 'bench.py'))\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "r = b.place_rank(5, 8, sysfs=%r, allowed=set(range(256)), do_apply=False)\n"
        "print(json.dumps({'r': r, 'torch': 'torch' in sys.modules}))\n") % (ROOT, sysfs)
    env = {k: v for k, v in os.environ.items() if "VISIBLE_DEVICES" not in k}  # (this container exports an empty HIP_VISIBLE_DEVICES: no device)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=env)
    import json
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["torch"] is False
    assert rec["r"]["numa_node"] == 1 and rec["r"]["cpus"] == "64-127,192-255" and rec["r"]["applied"] is False


def test_cpp_worker_placement_reads_the_same_tree(tmp_path):
    """wm_stream's worker threads (csrc/app/placement.hpp) choose their CPUs from the PCI address of their device: the
    --placement-of mode runs that lookup without touching a GPU.  This process may use only a few CPUs (the container's), so
    the expected set is the device's local CPUs intersected with ours -- and a device on the other socket yields no plan."""
    import json
    exe = os.path.join(ROOT, "watermarking-gpu_amd", "wm_stream")
    if not os.path.exists(exe):
        pytest.skip("wm_stream not built")
    sysfs = fake_sysfs(str(tmp_path))
    ours = os.sched_getaffinity(0)
    for kfd, minor, numa, pci in GPUS:
        out = subprocess.run([exe, "--placement-of", pci.upper(), "--sysfs", sysfs], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        rec = json.loads(out.stdout.strip())
        want = PL.parse_cpulist(NODE_CPUS[numa]) & ours
        assert rec["numa_node"] == numa
        assert rec["valid"] == bool(want) and PL.parse_cpulist(rec["cpus"]) == want, (pci, rec, sorted(want))
    out = subprocess.run([exe, "--placement-of", "0000:aa:00.0", "--sysfs", sysfs], capture_output=True, text=True, timeout=60)
    assert json.loads(out.stdout.strip())["valid"] is False
