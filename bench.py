#!/usr/bin/env python3
"""bench.py -- frames/s of embed+detect (ME mask) at 3840x2160 on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (no launcher: for N > 1 this process starts the N ranks itself,
                                                            before anything touches the GPU, and relays rank 0's line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              (the driver's form: RANK / LOCAL_RANK / WORLD_SIZE from the env)

A "step" = one pass of the hot path over one batch of synthetic frames per GPU: for every frame
makeWatermark(x, x, ME) followed by detectWatermark(y, ME), frames resident in HBM
before the timed region.  Workload at N=1: BASELINE.json configs[2] (3840x2160, ME mask, f32 planes).
Frames are independent units, so N GPUs shard the stream frame-parallel (weak scaling: every rank gets
its own batch); the only collective is the RCCL gather of the per-frame detector scores (4 B/frame).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : dominant kernel, algorithmic bytes per launch / average launch duration measured with
                 hipEvents on the launch stream (wm_prof_*), against the 8 TB/s HBM peak
  kernels      : the same figure for every kernel of the path
  cpu_baseline : the CPU oracle (oracle/wm_oracle.c, kind "port") timed on this box's host cores on a
                 bounded sample of the same frames (rank 0, N=1 only)
  parity       : GPU vs oracle on those sample frames (max |d corr|, max rel |d a|)
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

# Algorithmic bytes of one LAUNCH of each kernel over F frames (DESIGN.md "bytes per kernel"): every plane the sweep
# needs, once; halos and scalar partials excluded.  es = bytes per pixel of the frame planes.  W is one plane shared by
# all frames of the batch, so it counts once per launch (the frame-fastest block order lets the hardware serve the
# other F-1 uses from L2) -- per FRAME the SURVEY.md section 8d figures (W counted in every sweep) are kept for the
# path-level number.
ALG_BYTES = {
    "k_gram": lambda es, F: es * F,                 # {x} per frame
    "k_me_stats": lambda es, F: es * F + 4,         # {x} per frame, W once
    "k_embed": lambda es, F: 2 * es * F + 4,        # {x (= base) -> y} per frame, W once
    "k_detect": lambda es, F: es * F + 4,           # {y} per frame, W once
    "k_nvf_stats": lambda es, F: es * F + 4,
}


def stream_leg(wm, synth, torch, dist, dev, dev_index, rank, world, R, Cc, nframes, seconds, F=8, S=4):
    """BASELINE.json configs[3]: a 3840x2160 u8 Y-plane stream, watermark_interval = 1, frame i of the
    stream on GPU i mod N, >= 512 distinct frames cycled from a ring.  Three parts, every one embed + detect (ME) per frame:
      resident : the ring lives in HBM (what the kernels can do; bound: HBM),
      staged   : the ring lives in pinned host memory; frames cross PCIe once each way (wm_embed stages in and out, wm_detect
                 reads the slot's device copy of the output, WM_MEM_SLOT_OUT),
      link     : embed only, host-staged -- the rate the host link sustains for one frame in and one frame out.
    Returns per-GPU-aggregated frames/s (summed over ranks)."""
    import ctypes as C
    import numpy as np
    L = wm.lib()
    ME = int(wm.MASK_TYPE.ME)
    n = R * Cc
    nb = max(S, nframes // F)          # batches in the ring
    nframes = nb * F
    eng = wm.Watermark.generated(R, Cc, synth.SEED, 3, 40.0, device=dev_index, nslots=S, max_frames=F)
    ring_dev = torch.empty((nframes, R, Cc), dtype=torch.uint8, device=dev)
    for b in range(nb):  # this rank's frames of the stream: global frame index rank + world * k
        ring_dev[b * F:(b + 1) * F] = synth.synth_frames_torch(R, Cc, F, dev, dtype="u8", first_frame=(rank + world * b) * F)
    hp = L.wm_host_alloc(nframes * n)
    assert hp, "pinned allocation for the host ring failed"
    ring_host = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(nframes, R, Cc))
    torch.from_numpy(ring_host).copy_(ring_dev)
    out_dev = [torch.empty((F, R, Cc), dtype=torch.uint8, device=dev) for _ in range(S)]
    out_hp = [L.wm_host_alloc(F * n) for _ in range(S)]
    a = [(C.c_float * F)() for _ in range(S)]
    corr = [(C.c_float * F)() for _ in range(S)]
    st = [(C.c_int * F)() for _ in range(S)]
    torch.cuda.synchronize()

    def plane(ptr, mem):
        return wm.wm_plane(ptr, R, Cc, 1, wm.WM_U8, mem, F, Cc, 0, n)
    p_dev = [plane(ring_dev[b * F].data_ptr(), wm.WM_MEM_DEVICE) for b in range(nb)]
    p_host = [plane(hp + b * F * n, wm.WM_MEM_HOST) for b in range(nb)]
    p_out_dev = [plane(o.data_ptr(), wm.WM_MEM_DEVICE) for o in out_dev]
    p_out_host = [plane(o, wm.WM_MEM_HOST) for o in out_hp]
    p_slot = plane(None, wm.WM_MEM_SLOT_OUT)

    def run(kind):
        busy = [False] * S
        done = 0

        def one(b):
            s = b % S
            if busy[s]:
                assert L.wm_sync(eng._ctx, s) == 0
            if kind == "resident":
                rc = L.wm_embed(eng._ctx, ME, C.byref(p_dev[b % nb]), C.byref(p_dev[b % nb]), C.byref(p_out_dev[s]), a[s], st[s], s)
                rc |= L.wm_detect(eng._ctx, ME, C.byref(p_out_dev[s]), corr[s], None, s)
            else:
                rc = L.wm_embed(eng._ctx, ME, C.byref(p_host[b % nb]), C.byref(p_host[b % nb]), C.byref(p_out_host[s]), a[s], st[s], s)
                if kind == "staged":
                    rc |= L.wm_detect(eng._ctx, ME, C.byref(p_slot), corr[s], None, s)
            assert rc == 0, wm.lib().wm_last_error(eng._ctx)
            busy[s] = True
        for b in range(S):  # warm-up
            one(b)
        for s in range(S):
            L.wm_sync(eng._ctx, s); busy[s] = False
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        b = 0
        while True:
            one(b)
            b += 1
            if b % nb == 0 and time.perf_counter() - t0 >= seconds:  # whole passes over the ring
                break
        for s in range(S):
            if busy[s]:
                assert L.wm_sync(eng._ctx, s) == 0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert all(v == 0 for sl in st for v in sl)
        return b * F / dt, corr[0][0]
    res = {}
    for kind in ("resident", "staged", "link"):
        fps, c0 = run(kind)
        res["per_rank_" + kind] = [fps]
        if world > 1:
            cd = dev if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor([fps], dtype=torch.float64, device=cd)
            every = torch.empty(world, dtype=torch.float64, device=cd)
            dist.all_gather_into_tensor(every, t)
            res["per_rank_" + kind] = [float(v) for v in every.cpu()]
            fps = float(every.sum().item())
        res[kind] = fps
        res["corr_" + kind] = c0
    eng.close()
    for o in out_hp:
        L.wm_host_free(o)
    L.wm_host_free(hp)
    return res, nframes, F, S


class GpuStateSampler:
    """engine clock, power and busy percentage of the GPU during a stretch of the bench, read from the amdgpu sysfs nodes in a
    background thread (every 20 ms): the record that says whether two boxes' different kernel times come with different
    clocks.  Every AMD card the job can see is sampled; the one that was busiest is reported (a one-GPU box may show the
    host's other cards).  Missing nodes (no permission, another driver) give None: the bench does not depend on it."""

    def __init__(self):
        import glob
        self.cards = []
        for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
            hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
            ent = {"busy": os.path.join(dev, "gpu_busy_percent"), "sclk": None, "power": None}
            for h in hw:
                for name, key in (("freq1_input", "sclk"), ("power1_average", "power"), ("power1_input", "power")):
                    if ent[key] is None and os.path.exists(os.path.join(h, name)):
                        ent[key] = os.path.join(h, name)
            if ent["sclk"] or os.path.exists(ent["busy"]):
                self.cards.append(ent)
        self.samples = [[] for _ in self.cards]
        self._stop = False
        self._th = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop:
            for k, c in enumerate(self.cards):
                self.samples[k].append((self._read(c["sclk"]) if c["sclk"] else None, self._read(c["power"]) if c["power"] else None,
                                        self._read(c["busy"])))
            time.sleep(0.02)

    def start(self):
        import threading
        if self.cards:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()

    def stop(self):
        self._stop = True
        if self._th:
            self._th.join(timeout=1.0)
        best, best_busy = None, -1.0
        for smp in self.samples:
            busy = [b for _, _, b in smp if b is not None]
            mb = sum(busy) / len(busy) if busy else 0.0
            if smp and mb > best_busy:
                best, best_busy = smp, mb
        if not best:
            return None
        clk = [c / 1e6 for c, _, _ in best if c]
        pw = [p / 1e6 for _, p, _ in best if p]
        return {"samples": len(best), "sclk_MHz": {"min": round(min(clk)), "mean": round(sum(clk) / len(clk)), "max": round(max(clk))} if clk else None,
                "power_W_mean": round(sum(pw) / len(pw), 1) if pw else None, "gpu_busy_percent_mean": round(best_busy, 1),
                "source": "amdgpu sysfs (hwmon freq1_input / power1_average, gpu_busy_percent), 20 ms period, during the sustained stretch"}


def box_record(pci):
    """What two boxes of the pool can differ in beyond clock and power (which GpuStateSampler samples): memory / compute partition
    modes, HBM occupancy, memory clock, temperatures, power cap, VBIOS, driver and firmware versions -- read from the amdgpu
    sysfs nodes of the card at PCI address `pci` (or the first card when unknown).  Missing nodes give None."""
    import glob
    import platform

    def rd(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except Exception:
            return None
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    dev = None
    for c in cards:
        try:
            if pci and os.path.basename(os.path.realpath(c)).lower() == pci.lower():
                dev = c
        except OSError:
            pass
    matched = dev is not None
    if dev is None and cards:
        dev = cards[0]
    rec = {"pci": pci, "sysfs_card": os.path.basename(os.path.dirname(dev)) if dev else None, "matched_by_pci": matched}
    if dev:
        for key in ("current_memory_partition", "current_compute_partition", "available_memory_partition", "mem_info_vram_used", "mem_info_vram_total",
                    "vbios_version", "mem_busy_percent", "current_link_speed", "current_link_width", "numa_node", "xgmi_hive_id"):
            rec[key] = rd(os.path.join(dev, key))

        def current(path):  # pp_dpm_*: the level marked with '*'
            t = rd(path)
            if not t:
                return None
            cur = [ln.split(":", 1)[1].replace("*", "").strip() for ln in t.splitlines() if "*" in ln and ":" in ln]
            return cur[0] if cur else None
        rec["mclk"] = current(os.path.join(dev, "pp_dpm_mclk"))
        rec["fclk"] = current(os.path.join(dev, "pp_dpm_fclk"))
        rec["sclk_level"] = current(os.path.join(dev, "pp_dpm_sclk"))
        temps, hw = {}, sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        for h in hw:
            for lab in glob.glob(os.path.join(h, "temp*_label")):
                v = rd(lab.replace("_label", "_input"))
                if v is not None:
                    try:
                        temps[rd(lab)] = round(int(v) / 1000.0, 1)
                    except ValueError:
                        pass
            cap = rd(os.path.join(h, "power1_cap"))
            if cap and "power_cap_W" not in rec:
                try:
                    rec["power_cap_W"] = round(int(cap) / 1e6, 1)
                except ValueError:
                    pass
        rec["temperatures_C"] = temps or None
        fw = {}
        for f in sorted(glob.glob(os.path.join(dev, "fw_version", "*_fw_version"))):
            v = rd(f)
            if v and v not in ("0x00000000", "0"):
                fw[os.path.basename(f).replace("_fw_version", "")] = v
        rec["firmware"] = fw or None
    rec["amdgpu_driver"] = rd("/sys/module/amdgpu/version")
    rec["rocm"] = rd("/opt/rocm/.info/version")
    rec["kernel"] = platform.release()
    return rec


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher: start the N rank processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, rendezvous on 127.0.0.1) from THIS process, which has not imported torch and never touches
    HIP -- nothing that has initialised the GPU is forked or re-executed.  Rank 0's stdout is relayed (its one JSON line),
    the other ranks' stdout goes to stderr; the first rank that fails takes the others down (exact PIDs) and its exit code
    becomes ours.  Returns the exit code."""
    import socket
    import subprocess
    import threading
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), WM_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, cores // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def relay():
        for ln in procs[0].stdout:  # the JSON line to stdout; library chatter (gloo prints there) to stderr
            dst = sys.stdout if ln.lstrip().startswith("{") else sys.stderr
            dst.write(ln)
            dst.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    limit = float(os.environ.get("WM_BENCH_SPAWN_TIMEOUT", "3000"))
    t0, rc = time.time(), 0
    alive = set(range(n))
    while alive and rc == 0:
        for r in sorted(alive):
            c = procs[r].poll()
            if c is not None:
                alive.discard(r)
                if c != 0:
                    print(f"bench.py: rank {r} exited with code {c}", file=sys.stderr)
                    rc = c if 0 < c < 256 else 1
        if time.time() - t0 > limit:
            print(f"bench.py: ranks still running after {limit:.0f} s, stopping them", file=sys.stderr)
            rc = 124
        time.sleep(0.05)
    for r in alive:  # a rank failed or the limit passed: stop the ranks this process started, by PID
        procs[r].terminate()
    for r in alive:
        try:
            procs[r].wait(timeout=10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    th.join(timeout=5)
    return rc


def _placement_module():
    """watermarking-gpu_amd/placement.py loaded by path: the package's __init__ imports numpy, and nothing that may start threads
    is imported before the rank has pinned itself"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wm_placement", os.path.join(ROOT, "watermarking-gpu_amd", "placement.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def place_rank(device_index, world, sysfs="/sys", allowed=None, do_apply=None):
    """The first thing a rank does (before torch / numpy are imported, before the pinned ring is allocated): find the NUMA node
    of the GPU it will open and -- in a multi-rank run -- pin itself to that node's CPUs, so that its threads and, by first
    touch, its pinned host buffers live next to its GPU (8 ranks x 45 GB/s each way is a host-memory load: a ring on the wrong
    socket crosses the socket interconnect twice per frame).  A one-rank run is only described, not pinned (its CPU baseline
    uses every core of the job).  Returns the record for the bench line."""
    pl = _placement_module()
    p = pl.plan(device_index, sysfs=sysfs, allowed=allowed)
    if do_apply is None:
        do_apply = world > 1 and os.environ.get("WM_BENCH_NO_PIN") != "1"
    applied = pl.apply(p) if do_apply else False
    return pl.describe(p, applied)


def confirm_placement(rec, pci_bus_id, world):
    """once the device is open: the runtime's PCI address for it against the sysfs guess made before (the KFD order and the
    visible-device lists decide which GPU 'device r' is; containers can hide devices in ways the lists do not show).  A
    different NUMA node re-pins the calling thread -- the one that allocates the pinned ring -- and says so in the record."""
    pl = _placement_module()
    q = pl.plan_for_pci(pci_bus_id) if pci_bus_id else None
    rec = dict(rec)
    rec["pci_of_open_device"] = pci_bus_id
    if q is None:
        rec["confirmed_by_pci"] = None
        return rec
    same = rec.get("numa_node") == q["numa_node"]
    rec["confirmed_by_pci"] = bool(same)
    if not same:
        do_apply = world > 1 and os.environ.get("WM_BENCH_NO_PIN") != "1"
        applied = pl.apply(q) if do_apply else False
        rec.update(pl.describe(q, applied))
        rec["confirmed_by_pci"] = False
        rec["pci_of_open_device"] = pci_bus_id
    return rec


def plumbing_only(args, torch, dist, rank, world):
    """--plumbing-only: what a multi-rank run does around the GPU work -- rendezvous, the ranks_seen all-reduce, one score
    gather re-sequenced into stream order, barrier, rank 0's single line -- over gloo on the CPU, with made-up scores."""
    frames_mod = importlib.import_module("watermarking-gpu_amd.frames")
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ranks_seen, B = 1, 6
    scores = torch.tensor([float(rank + world * k) for k in range(B)], dtype=torch.float32)  # frame i lives on rank i mod N
    if world > 1:
        ones = torch.ones(1)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        scores, _ = frames_mod.gather_scores(scores, B * world, rank, world)
        dist.barrier()
    ok = bool((scores == torch.arange(B * world, dtype=torch.float32)).all())
    if rank == 0:
        print(json.dumps({"metric": "plumbing-only (no GPU work, no rate)", "value": None, "n_gpus": world, "ranks_seen": ranks_seen,
                          "scores_in_stream_order": ok}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok and ranks_seen == world else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=2160)
    ap.add_argument("--cols", type=int, default=3840)
    ap.add_argument("--dtype", choices=["f32", "u8"], default="f32")
    ap.add_argument("--mask", choices=["ME", "NVF"], default="ME", help="mask of the timed loop (the metric is ME; NVF is BASELINE.json "
                    "configs[1] / [4]'s other mask: four sweeps per frame instead of five)")
    ap.add_argument("--frames-per-slot", type=int, default=16)
    ap.add_argument("--slots", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-call", action="store_true", help="skip the one-image-per-call leg (wm_single)")
    ap.add_argument("--no-stream", action="store_true", help="skip the video-stream leg (BASELINE.json configs[3])")
    ap.add_argument("--no-slot-out", action="store_true", help="skip the Gram hand-over leg (path_slot_out)")
    ap.add_argument("--no-membench", action="store_true", help="skip the pure store / copy / read yardstick (membench)")
    ap.add_argument("--stream-batch", type=int, default=8, help="frames per call of the stream leg")
    ap.add_argument("--stream-slots", type=int, default=4, help="slots (calls in flight) of the stream leg")
    ap.add_argument("--stream-frames", type=int, default=512, help="distinct u8 Y planes in the stream leg's ring, per NODE (shared out over the ranks)")
    ap.add_argument("--plumbing-only", action="store_true", help="rendezvous, rank count and score gather over gloo with no GPU work: "
                    "the CPU test of the launcher path (the line says so in `metric`)")
    ap.add_argument("--sustain-seconds", type=float, default=1.0, help="span of the sustained repeat of the timed loop (0 = skip)")
    ap.add_argument("--stream-seconds", type=float, default=1.0, help="timed span of every part of the stream leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work budget of the cpu_baseline sample")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (before torch is imported, let alone the GPU touched)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # host placement first: nothing that starts threads or allocates pinned memory has been imported yet
    placement = place_rank(0 if os.environ.get("WM_BENCH_ALL_ON_DEVICE0") == "1" else local_rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist

    if world != args.gpus and rank == 0:
        # a launcher decides the world size; --gpus only documents it
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); running with {world}", file=sys.stderr)
    if args.plumbing_only:
        return plumbing_only(args, torch, dist, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (the engine has no CPU fallback)"
    # rehearsal knobs (development only): WM_BENCH_ALL_ON_DEVICE0=1 puts every rank on GPU 0 and WM_BENCH_BACKEND=gloo
    # swaps RCCL for gloo, so the N>1 code path can be exercised on a one-GPU box
    dev_index = 0 if os.environ.get("WM_BENCH_ALL_ON_DEVICE0") == "1" else local_rank
    backend = os.environ.get("WM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        pci = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        pci = None
    placement = confirm_placement(placement, pci, world)
    # WM_BENCH_FORCE_DIST=1: take the multi-GPU code path (process group, RCCL score gather, barrier) with one rank too
    force_dist = world == 1 and os.environ.get("WM_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29655")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    ranks_seen = 1
    if world > 1 or force_dist:
        # did the communicator see every rank?  an all-reduce of ones on it answers that from the record
        ones = torch.ones(1, dtype=torch.float32, device=coll_dev)
        dist.all_reduce(ones)
        ranks_seen = int(round(float(ones.item())))
        assert ranks_seen == world, f"the {backend} communicator counts {ranks_seen} ranks, WORLD_SIZE is {world}"

    wm = importlib.import_module("watermarking-gpu_amd")
    synth = importlib.import_module("watermarking-gpu_amd.synth")
    frames_mod = importlib.import_module("watermarking-gpu_amd.frames")
    pending_gather, last_scores = [], [None]

    R, Cc = args.rows, args.cols
    F, S = args.frames_per_slot, args.slots
    B = F * S  # frames per step per GPU
    es = 4 if args.dtype == "f32" else 1
    N = R * Cc
    ME = int(wm.MASK_TYPE.ME) if args.mask == "ME" else int(wm.MASK_TYPE.NVF)  # (the mask of every leg below; named after the metric's)

    # ---- synthetic inputs, resident in HBM before the timed region ------------------------------------
    # W is generated ON each rank's GPU from the seed (wm_create_generated: element (r,c) depends on (seed, r, c) only, so
    # every GPU of the node holds the same matrix without a file, an upload or a broadcast)
    eng = wm.Watermark.generated(R, Cc, synth.SEED, 3, 40.0, device=dev_index, nslots=S, max_frames=F)
    xs = [synth.synth_frames_torch(R, Cc, F, dev, dtype=args.dtype, first_frame=(rank * S + s) * F) for s in range(S)]
    ys = [torch.empty_like(x) for x in xs]
    a_out = [(C.c_float * F)() for _ in range(S)]
    corr_out = [(C.c_float * F)() for _ in range(S)]
    st_e = [(C.c_int * F)() for _ in range(S)]
    st_d = [(C.c_int * F)() for _ in range(S)]
    scores_dev = torch.zeros(B, dtype=torch.float32, device=coll_dev)
    scores_pinned = torch.zeros(B, dtype=torch.float32).pin_memory()
    torch.cuda.synchronize()

    px = [wm.plane_of(x) for x in xs]
    py = [wm.plane_of(y) for y in ys]

    have_results = [False] * S

    def collect(sl):
        """wait for slot `sl`'s previous batch and take its detector scores (a no-op before the first enqueue)"""
        if not have_results[sl]:
            return
        eng.sync(sl)
        have_results[sl] = False
        if world > 1 or force_dist:
            scores_pinned[sl * F:(sl + 1) * F] = torch.frombuffer(corr_out[sl], dtype=torch.float32)

    def step():
        # One step = one batch of B = S*F frames.  The slots form a software pipeline across steps: a slot is only
        # synchronised right before it is re-armed, so the other slots keep the GPU busy meanwhile; the barrier that
        # closes the timed region drains all of them.  Per slot: embed of its F frames, then detect on the
        # watermarked frames (same stream => ordered).
        for sl in range(S):
            collect(sl)
            eng.embed_async(px[sl], px[sl], py[sl], ME, sl, a_out=a_out[sl], status_out=st_e[sl])
            eng.detect_async(py[sl], ME, sl, corr_out=corr_out[sl], status_out=st_d[sl])
            have_results[sl] = True
        if world > 1 or force_dist:
            # the path's only exchange: per-frame detector scores (of the batch just collected) to every rank -- RCCL
            # all-gather, 4 B/frame, re-sequenced into stream order (frame i lives on rank i mod N): frames.py
            scores_dev.copy_(scores_pinned, non_blocking=True)
            (recv, finish), work = frames_mod.gather_scores(scores_dev, B * world, rank, world, device=coll_dev, async_op=True, force_collective=force_dist)
            pending_gather.append((work, finish))
            if len(pending_gather) > 2:
                w0, f0 = pending_gather.pop(0)
                w0.wait()
                last_scores[0] = f0()

    def barrier():
        for sl in range(S):
            collect(sl)
        while pending_gather:
            w0, f0 = pending_gather.pop(0)
            w0.wait()
            last_scores[0] = f0()
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    per_rank_fps = [B * args.steps / dt]
    if world > 1 or force_dist:
        mine = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        every = torch.empty(world, dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(every, mine)
        per_rank_fps = [B * args.steps / float(v) for v in every.cpu()]
        tmax = mine.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    fps = world * B * args.steps / dt
    # every frame of the timed steps must have been solvable: a run over passthrough frames would time nothing
    for sl in range(S):
        assert all(v == 0 for v in st_e[sl]) and all(v == 0 for v in st_d[sl]), "unsolvable frames in the timed region"

    # ---- per-kernel durations with hipEvents on the launch stream: a separate pass, every launch on slot 0's stream and
    # bracketed by its own pair of events (with several slots in flight kernels of different streams overlap and an event pair
    # would time the overlap, not the kernel).  The steps are enqueued back to back and synchronised ONCE at the end: with a
    # host synchronisation per step the GPU idles between the steps and the first kernels behind every gap run 10-20 %
    # longer -- that is the cost of the gap, not of the kernel.  The pass comes RIGHT AFTER the timed steps: it describes the
    # kernels in the state the timed region ran in.  After a second of full load the board regulates harder (it sits at its
    # power limit) and the compute-dense sweeps take 10-25 % longer (k_gram 105 -> 123 us, k_detect 110 -> 136 us; the
    # memory-bound ones +1-3 %) while the frame rate holds: that second sample is reported as `kernels_after_sustained_load`
    def step_serial(sync=True):
        eng.embed_async(px[0], px[0], py[0], ME, 0, a_out=a_out[0], status_out=st_e[0])
        eng.detect_async(py[0], ME, 0, corr_out=corr_out[0], status_out=st_d[0])
        if sync:
            eng.sync(0)

    step_serial()
    eng.prof_enable(True)
    eng.prof_reset()
    prof_steps = max(1, min(args.steps, 10))
    torch.cuda.synchronize()
    t_ser = time.perf_counter()
    for _ in range(prof_steps):
        step_serial(sync=False)
    eng.sync(0)
    torch.cuda.synchronize()
    serial_step_us = 1e6 * (time.perf_counter() - t_ser) / prof_steps
    rep = eng.prof_report()
    eng.prof_enable(False)
    kernels = {}
    for name, (n, ms) in rep.items():
        avg_us = 1e3 * ms / n
        ent = {"launches": int(n), "avg_us": round(avg_us, 2), "frames_per_launch": F}
        if name in ALG_BYTES:
            byts = ALG_BYTES[name](es, F) * N
            ent["alg_bytes_per_launch"] = byts
            ent["achieved_GBs"] = round(byts / (avg_us * 1e-6) / 1e9, 1)
            ent["frac"] = round(ent["achieved_GBs"] / HBM_PEAK_GBS, 4)
        kernels[name] = ent
    # the same loop once more over >= --sustain-seconds (the contract's K steps last tens of milliseconds at this rate): same
    # step count on every rank (derived from the all-reduced time), same barriers, max over ranks
    sustained = None
    if args.sustain_seconds > 0:
        def timed_steps(n):
            barrier()
            t1 = time.perf_counter()
            for _ in range(n):
                step()
            barrier()
            d = time.perf_counter() - t1
            if world > 1 or force_dist:
                tm = torch.tensor([d], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                d = float(tm.item())
            return d
        n_sus = max(args.steps, int(1.1 * args.sustain_seconds / (dt / args.steps)) + 1)
        sampler = GpuStateSampler() if rank == 0 else None
        if sampler:
            sampler.start()
        dts = timed_steps(n_sus)
        if dts < args.sustain_seconds:  # the estimate fell short (the K-step figure was a slow sample): one more stretch, summed
            n2 = int(1.5 * (args.sustain_seconds - dts) / (dts / n_sus)) + 1
            dts += timed_steps(n2)
            n_sus += n2
        sustained = {"steps": n_sus, "seconds": round(dts, 3), "frames_per_s": round(world * B * n_sus / dts, 1),
                     "gpu_state": sampler.stop() if sampler else None}
    kernels_after = None
    if sustained is not None:
        eng.prof_enable(True)
        eng.prof_reset()
        for _ in range(5):
            step_serial(sync=False)
        eng.sync(0)
        torch.cuda.synchronize()
        kernels_after = {name: round(1e3 * ms / n, 2) for name, (n, ms) in eng.prof_report().items()}
        eng.prof_enable(False)
    # ---- the same loop with the opt-in Gram hand-over (wm.h wm_set_handover): the detector reads the slot's last embed output
    # (WM_MEM_SLOT_OUT) and k_embed has left y's tile-internal lag sums, so the detector's Gram sweep over y is not run (4 sweeps
    # + a seam pass per frame instead of 5).  Reported BESIDE the headline, never as it: the headline's detector takes a plane it
    # knows nothing about, as Watermark::detectWatermark does.
    slot_out = None
    if world == 1 and not force_dist and args.dtype == "f32" and not args.no_slot_out:
        indep = [list(c) for c in corr_out]
        eng.set_handover(True)
        p_slot = wm.wm_plane(None, R, Cc, 1, wm.WM_F32, wm.WM_MEM_SLOT_OUT, F, Cc, 0, N)

        def step_ho():
            for sl in range(S):
                collect(sl)
                eng.embed_async(px[sl], px[sl], py[sl], ME, sl, a_out=a_out[sl], status_out=st_e[sl])
                eng.detect_async(p_slot, ME, sl, corr_out=corr_out[sl], status_out=st_d[sl])
                have_results[sl] = True
        for _ in range(max(args.warmup, 2)):
            step_ho()
        barrier()
        n_ho = max(args.steps, int(0.5 / (dt / args.steps)) + 1)
        t1 = time.perf_counter()
        for _ in range(n_ho):
            step_ho()
        barrier()
        dt_ho = time.perf_counter() - t1
        diff = max(abs(a_ - b_) for ca, cb in zip(corr_out, indep) for a_, b_ in zip(ca, cb))
        for sl in range(S):
            assert all(v == 0 for v in st_e[sl]) and all(v == 0 for v in st_d[sl]), "unsolvable frames in the hand-over leg"
        eng.prof_enable(True)
        eng.prof_reset()
        for _ in range(5):
            eng.embed_async(px[0], px[0], py[0], ME, 0, a_out=a_out[0], status_out=st_e[0])
            eng.detect_async(p_slot, ME, 0, corr_out=corr_out[0], status_out=st_d[0])
        eng.sync(0)
        torch.cuda.synchronize()
        k_ho = {name: round(1e3 * ms / n, 2) for name, (n, ms) in eng.prof_report().items()}
        eng.prof_enable(False)
        eng.set_handover(False)
        assert "k_gram_ho" in k_ho, "the hand-over leg did not take the hand-over kernels"
        # ... and the same work through the one-call pair (wm.h wm_embed_detect = Watermark::makeAndDetectWatermark) on batches in
        # flight: the library knows the detector's input is the plane it has just written, so the hand-over needs no opt-in and
        # no promise from the caller
        L_ = wm.lib()

        def step_pair():
            for sl in range(S):
                collect(sl)
                rc_ = L_.wm_embed_detect(eng._ctx, ME, C.byref(px[sl]), C.byref(px[sl]), C.byref(py[sl]), a_out[sl], corr_out[sl], st_e[sl], sl)
                assert rc_ >= 0, f"wm_embed_detect: {rc_}"
                have_results[sl] = True
        for _ in range(2):
            step_pair()
        barrier()
        t2 = time.perf_counter()
        for _ in range(n_ho):
            step_pair()
        barrier()
        dt_pair = time.perf_counter() - t2
        diff_pair = max(abs(a_ - b_) for ca, cb in zip(corr_out, indep) for a_, b_ in zip(ca, cb))
        slot_out = {"frames_per_s": round(B * n_ho / dt_ho, 1), "steps": n_ho, "seconds": round(dt_ho, 3),
                    "vs_independent_calls": round(B * n_ho / dt_ho / (sustained["frames_per_s"] if sustained else fps), 4),
                    "vs_what": "the sustained figure of the independent-calls loop (same board state)" if sustained else "the timed steps",
                    "max_abs_score_difference_to_independent_calls": float(diff),
                    "one_call_pair_on_batches": {"frames_per_s": round(B * n_ho / dt_pair, 1), "seconds": round(dt_pair, 3),
                                                 "max_abs_score_difference_to_independent_calls": float(diff_pair),
                                                 "what": "wm_embed_detect per slot and batch (no wm_set_handover: the pair hands over by itself)"},
                    "kernels_avg_us": k_ho,
                    "what": "opt-in (wm_set_handover): wm_detect on WM_MEM_SLOT_OUT, the slot's last wm_embed output; k_embed accumulates "
                            "the lag sums of y inside its tiles, k_gram_ho adds strip seams, border frame and solve; the same exact "
                            "products in another f64 summation order (tests/test_gpu_handover.py)"}
    # ---- memory-system yardstick, independent of the engine's sweeps (wm.h wm_membench): one pure-store, one pure-copy and one
    # pure-read kernel over the bytes of one 16-frame k_embed launch's output (531 MB at 4K f32), ~0.3 s each.  k_embed is the one
    # sweep whose time differs from box to box at equal clocks and power: a box that writes slowly shows
    # here too, with nothing of the engine in the loop
    membench = None
    if rank == 0 and not args.no_membench:
        mb_bytes = int(F * es * N)
        membench = {"bytes_per_launch": mb_bytes, "what": "wm_membench: 16 B per lane, non-temporal stores, mean launch duration by events attached to the dispatch; "
                                                        "store / copy / read: 2048 blocks x 256 threads x 4 elements in flight (a grid like the sweeps'), *_best_shape: "
                                                        "65536 blocks x 1 element per thread (the fastest streaming shape on MI355X)"}
        for kind, name in ((0, "store"), (1, "copy"), (2, "read"), (3, "store_best_shape"), (4, "copy_best_shape"), (5, "read_best_shape")):
            us, nl = C.c_double(0.0), C.c_int(0)
            rc_mb = wm.lib().wm_membench(dev_index, kind, mb_bytes, 0.2, C.byref(us), C.byref(nl))
            moved = mb_bytes * (2 if kind % 3 == 1 else 1)
            membench[name] = {"mean_us": round(us.value, 2), "launches": nl.value, "GBs": round(moved / (us.value * 1e-6) / 1e9, 1) if rc_mb == 0 and us.value > 0 else None}
        if "k_embed" in kernels and membench["store"]["GBs"]:
            # k_embed's algorithmic bytes against what the same box does with a pure copy of as many bytes: the kernel's share
            membench["k_embed_vs_copy"] = round(kernels["k_embed"]["achieved_GBs"] / membench["copy"]["GBs"], 4) if membench["copy"]["GBs"] else None
    placements = [placement]
    if world > 1 or force_dist:
        # every rank's record as a fixed-size byte tensor through the same collective the rates use (all_gather_into_tensor on
        # the communicator's device): nothing here that the score gather has not exercised already
        PLEN = 1024
        raw = json.dumps(placement).encode()[:PLEN]
        buf = torch.zeros(PLEN, dtype=torch.uint8)
        buf[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
        every_b = torch.empty(world * PLEN, dtype=torch.uint8, device=coll_dev)
        dist.all_gather_into_tensor(every_b, buf.to(coll_dev))
        placements = []
        for r in range(world):
            chunk = bytes(every_b[r * PLEN:(r + 1) * PLEN].cpu().tolist()).rstrip(b"\0")
            try:
                placements.append(json.loads(chunk.decode()))
            except Exception:
                placements.append({"numa_node": None, "note": "record of this rank did not fit / parse"})
    dom = max((k for k in kernels if "achieved_GBs" in kernels[k]), key=lambda k: kernels[k]["avg_us"] * kernels[k]["launches"])
    # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in runs of their own,
    # tools/run_pmc.sh): a RECORDED figure read from profiles/pmc_traffic.json, not a measurement of this run
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get(f"{R}x{Cc}_{args.dtype}_F{F}" + ("_NVF" if args.mask == "NVF" else ""), {}).get(dom)
            if ent:
                traffic = ent["hbm_bytes_per_launch"]
                traffic_src = {"file": "profiles/pmc_traffic.json", "captured": tj.get("captured", "round 1"),
                               "note": "recorded PMC pass (2*FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md), not measured in this run"}
        except Exception:
            traffic = None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(kernels[dom]["achieved_GBs"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_launch": kernels[dom]["alg_bytes_per_launch"], "avg_launch_us": kernels[dom]["avg_us"],
                "timing": f"hipEvents around each launch on its stream, {prof_steps} steps enqueued back to back on one slot after the timed region"}
    # what actually limits each sweep: the HBM roofline is the contract's yardstick for all of them
    LIMITER = {"k_gram": "f32 frames: the strip march's request stream (5.8 TB/s for this shape with no arithmetic at all; the 13 exact f64 "
                         "lag products per pixel hide behind the loads: the kernel gains 5 % with every FMA removed, docs/history.md); "
                         "u8 frames: integer dot4 issue",
               "k_embed": "HBM (reads x, writes y; W from L2)", "k_me_stats": "HBM (reads x; W from L2)",
               "k_detect": "HBM (reads y; W from L2); vector issue close behind (~89 instructions per 4-pixel row and lane)",
               "k_nvf_stats": "vector issue beside HBM (the pinned NVF arithmetic: 17 sums + 3 correctly rounded quotients per pixel)"}
    roofline["limiter"] = LIMITER.get(dom, "HBM")
    # whole metric frame: embed-ME 3 sweeps {x};{x,W};{x,W->y} + detect-ME 2 sweeps {y};{y,W}
    # (NVF: no Gram sweep on the embed side -- embed 2 sweeps {x,W};{x,W->y} + detect 2 sweeps {y};{y,W}: SURVEY.md 8d)
    nsweep_x = 3 if args.mask == "ME" else 2  # sweeps of the embed
    frame_bytes = (((es) if args.mask == "ME" else 0) + (es + 4) + (es + 4 + es) + (es) + (es + 4)) * N
    path_gbs = fps / world * frame_bytes / 1e9
    # bytes that have to come from / go to HBM per frame: W is ONE plane shared by the F frames of a launch (the block
    # order lets L2 serve the other F-1 uses), so it counts 1/F per sweep that reads it
    frame_bytes_hbm = ((6 if args.mask == "ME" else 5) * es + 3 * 4.0 / F) * N
    path_hbm_gbs = fps / world * frame_bytes_hbm / 1e9

    out = {
        "metric": f"frames/sec embed+detect ({args.mask} mask) at {Cc}x{R}",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{Cc}x{R} {args.dtype} luminance frames, {args.mask} mask p=3 psnr=40, makeWatermark+detectWatermark per frame, "
                               f"frames resident in HBM" + (" (BASELINE.json configs[2])" if (R, Cc, args.mask, args.dtype) == (2160, 3840, "ME", "f32") else ""),
                   "frames_per_step_per_gpu": B, "slots": S, "frames_per_launch": F, "parallelism": f"frame-parallel x{world}"},
        # multi-GPU record: ranks the communicator counted (all-reduce of ones), every rank's own rate over the timed steps
        "ranks_seen": ranks_seen, "backend": (backend if (world > 1 or force_dist) else None),
        "per_rank_frames_per_s": [round(v, 1) for v in per_rank_fps],
        # host placement of every rank (NUMA node of its GPU, the CPUs it pinned itself to before allocating anything pinned;
        # watermarking-gpu_amd/placement.py).  A one-rank run is described, not pinned
        "placement": placements,
        "sustained": sustained,
        "box": box_record(pci) if rank == 0 else None,
        "membench": membench,
        "path_slot_out": slot_out,
        "roofline": roofline,
        "path": {"hbm_bytes_per_frame": int(frame_bytes_hbm), "achieved_GBs_per_gpu": round(path_hbm_gbs, 1),
                 "frac_of_hbm_peak": round(path_hbm_gbs / HBM_PEAK_GBS, 4),
                 "definition": f"{nsweep_x + 2} sweeps per frame, every frame plane once per sweep, W once per launch of F frames",
                 # SURVEY.md 8d's unit counts W in every sweep of every frame (36 N bytes at f32): L2 serves most of it,
                 # so this figure can exceed what HBM delivers and is NOT a roofline fraction
                 "survey_unit": {"bytes_per_frame": frame_bytes, "GBs_per_gpu": round(path_gbs, 1)},
                 "x_realtime_30fps": round(fps / 30.0, 1),
                 # the compulsory floor, every plane once per op (embed {x,W->y}, detect {y,W}); only a
                 # persistent single-launch design with grid barriers could approach it
                 "compulsory_bytes_per_frame": ((es + 4 + es) + (es + 4)) * N,
                 "frac_of_hbm_peak_compulsory": round(fps / world * ((es + 4 + es) + (es + 4)) * N / 1e9 / HBM_PEAK_GBS, 4)},
        "kernels": kernels,
        # cross-check of the per-kernel durations: the same serial steps timed on the host (enqueue of all steps -> one
        # synchronisation) against the sum of the five launches' event durations.  The events are attached to the dispatches
        # (hipExtLaunchKernelGGL start / stop); rocprofv3's kernel trace reports k_gram and k_detect 15-25 us shorter than these
        # events do, and its durations do NOT add up to the wall time of the serial step (profiles/r03_kernel_trace_summary.json)
        # the same launches sampled again right after the sustained stretch (average us per launch): the board's regulation
        # under a second of full load costs the compute-dense sweeps 10-25 %, the memory-bound ones 1-3 %
        "kernels_after_sustained_load": kernels_after,
        "serial_step_check": {"wall_us_per_step": round(serial_step_us, 1),
                              "sum_of_kernel_event_us": round(sum(v["avg_us"] * v["launches"] for v in kernels.values()) / prof_steps, 1)},
    }

    # ---- the video-stream configuration (BASELINE.json configs[3]): every rank runs its shard of the stream
    if not args.no_stream and (R, Cc) == (2160, 3840) and args.mask == "ME":
        sres, sn, sF, sS = stream_leg(wm, synth, torch, dist, dev, dev_index, rank, world, R, Cc, max(1, args.stream_frames // world), args.stream_seconds,
                                      F=args.stream_batch, S=args.stream_slots)
        yb = R * Cc  # bytes of a u8 Y plane
        out["stream"] = {
            "config": f"3840x2160 u8 Y planes, watermark_interval=1, {sn} distinct frames per GPU ({sn * world} per node) cycled from a ring, frame i -> GPU i mod {world}, "
                      f"{sF} frames per call x {sS} slots, embed + detect (ME) per frame (BASELINE.json configs[3])",
            "resident_frames_per_s": round(sres["resident"], 1), "resident_x_realtime_30fps": round(sres["resident"] / 30.0, 1),
            # HBM bytes per u8 frame: five sweeps {x};{x,W};{x,W->y};{y};{y,W} = 6 N + W once per launch
            "resident_frac_of_hbm_peak": round(sres["resident"] / world * (6 * yb + 3 * 4.0 * yb / sF) / 1e9 / HBM_PEAK_GBS, 4),
            "host_staged_frames_per_s": round(sres["staged"], 1),
            "host_staged_GBs_each_way_per_gpu": round(sres["staged"] / world * yb / 1e9, 2),
            "host_staged_GBs_each_way_by_rank": [round(v * yb / 1e9, 2) for v in sres["per_rank_staged"]],
            "host_link_GBs_each_way_by_rank": [round(v * yb / 1e9, 2) for v in sres["per_rank_link"]],
            "resident_frames_per_s_by_rank": [round(v, 1) for v in sres["per_rank_resident"]],
            "host_link_embed_only_frames_per_s": round(sres["link"], 1),
            "host_staged_frac_of_link_rate": round(sres["staged"] / sres["link"], 4),
            "note": "host-staged frames cross PCIe once each way: wm_embed stages the frame in and its output out, wm_detect reads the slot's device "
                    "copy of the output (WM_MEM_SLOT_OUT); the link rate is the embed-only rate of the same loop (one frame in, one out)",
            "detector_score_first_frame": {k[5:]: round(v, 6) for k, v in sres.items() if k.startswith("corr_")},
            "pinned_ring_MB_per_rank": round(sn * yb / 1e6 + sS * sF * yb / 1e6, 1),
        }

    # ---- one image per synchronous call: the reference's own call pattern (Watermark::makeWatermark, then
    # Watermark::detectWatermark, main.cpp:165-220), timed from C++ through include/Watermark.hpp by wm_single.  These
    # calls take the fused single-launch kernels (wm_k_fused.hip); the same calls on the batched sweeps beside it.
    if rank == 0 and world == 1 and not args.no_single_call and args.mask == "ME":
        import subprocess
        exe = os.path.join(ROOT, "watermarking-gpu_amd", "wm_single")

        def single(rows, cols, mask, fused=True, dtype=None):
            env = dict(os.environ, WM_FUSED="1" if fused else "0")
            p = subprocess.run([exe, str(rows), str(cols), "300", dtype or args.dtype, mask], env=env, capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                raise RuntimeError("wm_single failed: " + p.stderr[-500:])
            return json.loads(p.stdout.strip().splitlines()[-1])
        if os.path.exists(exe):
            sc = single(R, Cc, "ME")
            sw = single(R, Cc, "ME", fused=False)
            assert sc["fallbacks"] == 0, "a fused launch timed out during the single-call measurement"
            ent = {"api": "Watermark::makeWatermark + Watermark::detectWatermark (include/Watermark.hpp), one synchronous call each per frame, "
                          "timed in C++ (csrc/app/wm_single.cpp, 300 loops)",
                   "path": "fused single-launch kernels" if sc["fused"] else "batched sweeps (shape not fusable)",
                   "us_per_frame": sc["pair_us"], "frames_per_s": round(1e6 / sc["pair_us"], 1), "embed_us": sc["embed_us"], "detect_us": sc["detect_us"],
                   # the same pair as ONE call (wm.h wm_embed_detect / Watermark::makeAndDetectWatermark: both launches back to
                   # back, one wait) -- an addition to the reference's interface, reported beside the two-call figure, not as it
                   "one_call_pair_us": sc.get("pair_one_call_us"),
                   # the lead figure: the bytes the fused kernels have to move (embed {x, W -> y}, detect {y, W}: 20 N at f32 =
                   # SURVEY.md 8d's compulsory floor) against the HBM peak.  A single-call loop keeps x, W, y in the 256 MiB
                   # Infinity Cache, and ~14 of each ~30 us are hand-off waits: this path is latency-bound, the fraction says how
                   # far.  The sweeps' unit (36 N: what the same calls move on the batched path) is kept beside it, labelled
                   "frac": round(((es + 4 + es) + (es + 4)) * N / (sc["pair_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                   "frac_definition": "compulsory bytes per pair (20 N at f32) / time / 8 TB/s",
                   "frac_in_the_sweeps_unit_36N": round(frame_bytes / (sc["pair_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                   "workgroups": sc["workgroups"], "tile_rows": sc["tile_rows"],
                   "same_calls_on_the_sweeps": {"us_per_frame": sw["pair_us"], "embed_us": sw["embed_us"], "detect_us": sw["detect_us"]}}
            if (R, Cc) == (2160, 3840):
                # BASELINE.json configs[1]: 1920x1080 single image, NVF + ME masks, all four operations
                c1 = {m: single(1080, 1920, m, dtype="f32") for m in ("ME", "NVF")}
                ent["config1_1080p_f32"] = {m: {"embed_us": v["embed_us"], "detect_us": v["detect_us"], "pair_us": v["pair_us"], "one_call_pair_us": v.get("pair_one_call_us")} for m, v in c1.items()}
                # the other 4K single-image cases: NVF mask (f32) and a u8 Y plane (ME)
                # ... and the width of the reference's 4k_non_divisible sample (f32 planes of any width take the fused kernels)
                c2 = {"f32 NVF": single(R, Cc, "NVF", dtype="f32"), "u8 ME": single(R, Cc, "ME", dtype="u8"),
                      "f32 ME, 3838 columns": single(R, Cc - 2, "ME", dtype="f32")}
                ent["other_4k"] = {m: {"embed_us": v["embed_us"], "detect_us": v["detect_us"], "pair_us": v["pair_us"], "one_call_pair_us": v.get("pair_one_call_us")} for m, v in c2.items()}
            out["single_call"] = ent

    # ---- CPU baseline + parity on a bounded sample (rank 0, N=1 only) ------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        OM = O.MASK_ME if args.mask == "ME" else O.MASK_NVF
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        host_threads = cores
        # the box gives this job a CPU share (cgroup quota), usually far below the host's thread count: more OpenMP threads
        # than that only fight for the same cores (round 1 measured 0.5 frames/s on 256 threads against 42 on 16)
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
        except Exception:
            pass
        os.environ["OMP_NUM_THREADS"] = str(cores)
        # the oracle mallocs its temporaries per call like the code it restates; keep freed planes in the heap so the
        # baseline is not a page-fault benchmark, and pick the thread count that is fastest on this host
        try:
            libc = C.CDLL("libc.so.6")
            libc.mallopt(-3, 1 << 30)          # M_MMAP_THRESHOLD
            libc.mallopt(-1, (1 << 31) - 1)    # M_TRIM_THRESHOLD
        except Exception:
            pass
        xh = xs[0].cpu().numpy()
        yh = ys[0].cpu().numpy()
        W = eng.watermark()  # the matrix the GPU generated, downloaded: the oracle runs on the very same W

        def cpu_frame(f):
            if args.dtype == "f32":
                st, yo, ao = O.embed(xh[f], xh[f], W, mask=OM)
                st, co = O.detect(yo, W, mask=OM)
            else:
                st, yo, ao = O.embed_u8(xh[f], W, mask=OM)
                st, co = O.detect_u8(yo, W, mask=OM)
            return ao, co

        def timed_frame(f):
            t1 = time.perf_counter()
            r = cpu_frame(f)
            return time.perf_counter() - t1, r

        cpu_frame(0)  # warm-up (thread pool, page faults)
        tried = {}
        try:
            gomp = C.CDLL("libgomp.so.1")
            cands = sorted({c for c in (cores, cores // 2, 2 * cores, 16, 32) if 1 <= c <= min(host_threads, 2 * cores)}, reverse=True)
            for T in cands:
                # sustained rate, not the luckiest call: mean over >= 4 evaluations and >= 0.8 s (capped at 3 s)
                gomp.omp_set_num_threads(T)
                timed_frame(0)
                tt, nn = 0.0, 0
                while (nn < 4 or tt < 0.8) and tt < 3.0:
                    tt += timed_frame(nn % F)[0]
                    nn += 1
                tried[T] = tt / nn
            gomp.omp_set_num_threads(1)
            t_single = timed_frame(0)[0]
            best = min(tried, key=tried.get)
            gomp.omp_set_num_threads(best)
        except Exception:
            gomp, best, t_single = None, cores, None
            tried = {}
        t_one = timed_frame(0)[0]
        nsamp = int(max(2, min(1024, round(args.cpu_seconds / max(t_one, 1e-3)))))
        max_dcorr, max_da = 0.0, 0.0
        tcpu, tbest = 0.0, float("inf")
        for k in range(nsamp):
            f = k % F
            dt_k, (ao, co) = timed_frame(f)
            tcpu += dt_k
            tbest = min(tbest, dt_k)
            # parity: GPU strength vs oracle strength; GPU correlation vs the oracle's detector on the GPU's own output
            if k < F:
                if args.dtype == "f32":
                    st, cg = O.detect(yh[f], W, mask=OM)
                else:
                    st, cg = O.detect_u8(yh[f], W, mask=OM)
                max_dcorr = max(max_dcorr, abs(corr_out[0][f] - cg))
                max_da = max(max_da, abs(a_out[0][f] - ao) / abs(ao))
        out["cpu_baseline"] = {"value": round(nsamp / tcpu, 4), "unit": "frames/s", "cores": best, "kind": "port",
                               "sample": f"{nsamp} embed+detect ME evaluations over the benchmark's {Cc}x{R} {args.dtype} frames, "
                                         f"oracle/wm_oracle.c with OpenMP on {best} threads (CPU share of this job: {cores} of the host's {host_threads} hardware threads; the fastest of "
                                         f"{sorted(tried)} tried; {tcpu:.1f} s of CPU wall time)",
                               "best_single_evaluation": round(1.0 / tbest, 4),  # the mean (value) includes whatever else the host did
                               "single_thread_value": round(1.0 / t_single, 4) if t_single else None,
                               "frames_per_s_by_threads": {str(k): round(1.0 / v, 3) for k, v in sorted(tried.items())}}
        out["parity"] = {"frames": min(nsamp, F), "max_abs_dcorr_vs_oracle": max_dcorr, "max_rel_da_vs_oracle": max_da,
                         "tolerance": {"corr_abs": 1e-5, "a_rel": 1e-4}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
