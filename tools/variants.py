#!/usr/bin/env python3
"""Build named variants of libp2p_hip.so (tile shape, occupancy, ...) for A/B timing on the GPU box.

    python tools/variants.py build name=-DFLAG,-DFLAG ...     (here: hipcc cross-compiles without a GPU)
    python tools/variants.py run [bench args]                  (on the GPU box: bench every built variant)

Variants live in gpurun_variants/ (git-ignored, but they travel to the GPU box)."""
import glob
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "gpurun_variants")


def main():
    cmd = sys.argv[1]
    if cmd == "build":
        b = importlib.import_module("360-to-planer-images_amd._build")
        os.makedirs(VDIR, exist_ok=True)
        for spec in sys.argv[2:]:
            name, _, flags = spec.partition("=")
            out = os.path.join(VDIR, "libp2p_%s.so" % name)
            b.build(force=True, out=out, extra_flags=[f for f in flags.split(",") if f])
            print("built", out)
    elif cmd == "run":
        args = sys.argv[2:]
        sos = sorted(glob.glob(os.path.join(VDIR, "libp2p_*.so")))
        if sos:  # one untimed pass first: the variant that runs first otherwise meets a colder GPU (about +1 us)
            subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "600", "--warmup",
                            "300", "--counters", "none"] + args, env=dict(os.environ, P2P_LIB_PATH=sos[0]), capture_output=True)
        for so in sos:
            env = dict(os.environ, P2P_LIB_PATH=so)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "600",
                                "--warmup", "300", "--counters", "none"] + args, env=env, capture_output=True, text=True)
            try:
                j = json.loads(r.stdout.strip().splitlines()[-1])
                print("%-28s kernel %.1f us  frac %.3f" % (os.path.basename(so), j["roofline"]["kernel_ms_avg"] * 1e3,
                                                         j["roofline"]["frac"]), flush=True)
            except Exception:
                print(os.path.basename(so), "FAILED", r.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
