#!/usr/bin/env python3
"""A/B of two builds of libvqvae_hip.so on the whole forward, interleaved in ONE process is not possible (one library per
process), so: run bench.py for each library back to back, several rounds, and print ms_per_step of each run."""
import json, os, subprocess, sys
libs = sys.argv[1:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ, VQVAE_HIP_LIB_OVERRIDE=lib)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-power", "--no-other-workloads", "--steps", "20", "--min-seconds", "0.5"],
                             capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        k = d["kernels"]
        print(rnd, os.path.basename(lib), d["ms_per_step"], d["timing"]["ms_per_step_min"], {n: v["ms_per_step"] for n, v in k.items()}, d["roofline"]["avg_kernel_us"])
