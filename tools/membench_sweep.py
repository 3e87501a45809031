"""dev helper (GPU box): the pure store / copy / read yardstick (wm.h wm_membench) over grid sizes and elements in flight per thread,
one child process per shape (the shape is read from the environment once per process)
usage: python tools/membench_sweep.py [bytes]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 4 * 2160 * 3840
code = ("import sys, json, ctypes as C, importlib; sys.path.insert(0, %r); wm = importlib.import_module('watermarking-gpu_amd'); L = wm.lib(); out = {}\n"
        "for kind, name in ((0, 'store'), (1, 'copy'), (2, 'read')):\n"
        "    us, n = C.c_double(), C.c_int(); rc = L.wm_membench(0, kind, %d, 0.15, C.byref(us), C.byref(n)); out[name] = round(%d * (2 if kind == 1 else 1) / us.value / 1e3, 1) if rc == 0 and us.value > 0 else None\n"
        "print(json.dumps(out))\n") % (ROOT, nbytes, nbytes)
for blocks in (512, 1024, 2048, 4096, 8192, 16384, 65536):
    for unr in (1, 2, 4, 8):
        env = dict(os.environ, WM_MEMBENCH_BLOCKS=str(blocks), WM_MEMBENCH_UNROLL=str(unr))
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(f"blocks {blocks:6d} unroll {unr}: {p.stdout.strip() or p.stderr[-300:]}", flush=True)
