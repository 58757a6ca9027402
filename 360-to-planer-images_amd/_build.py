"""Build libp2p_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libp2p_hip.so")
# the plan pass and the view kernels are built once per tile shape (csrc/p2p_device.h: tile shapes): the *_w128.hip and
# *_band.hip files include their namesakes with the other shapes' constants
SOURCES = [os.path.join(CSRC, f) for f in ("p2p_views.hip", "p2p_plan.hip", "p2p_float.hip", "p2p_views_w128.hip", "p2p_plan_w128.hip",
                                            "p2p_float_w128.hip", "p2p_views_band.hip", "p2p_plan_band.hip", "p2p_maps.hip", "p2p_remap.hip", "p2p_lists.hip")]
# the host side (csrc/p2p_host.h lists the units): the C ABI's entry points + exception barrier, then what they call
HOST_SOURCES = [os.path.join(CSRC, f) for f in ("p2p_abi.cpp", "p2p_host_pool.cpp", "p2p_host_ctx.cpp", "p2p_host_plan.cpp",
                                                 "p2p_host_job.cpp", "p2p_host_oneshot.cpp")]
SOURCES += HOST_SOURCES
DEPS = SOURCES + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
    [os.path.join(HERE, "..", "include", "p2p_hip.h")]
# -ffp-contract=off: the coordinate maths must round exactly where NumPy rounds (no fused a*b+c
# unless written as fmaf).  IEEE divide / sqrt are hipcc's default for fp32.
# -amdgpu-atomic-optimizer-strategy=DPP: the default (iterative) strategy turns every LDS atomicMin/Max
# of the footprint reduction into a 64-iteration scalar loop; DPP makes it a 6-step wave reduction.
# -fvisibility=hidden: the dynamic symbol table holds the C ABI (P2P_EXPORT in p2p_abi.cpp) and nothing else.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-Wall", "-fvisibility=hidden",
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP"]


def build(force=False, verbose=False, out=None, extra_flags=None):
    """Build the library.  `out` / `extra_flags` make a variant next to the shipped one (tools/variants.py:
    tile shape, occupancy ... for A/B timing; load it with P2P_LIB_PATH)."""
    out = out or OUT
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = out + ".tmp.%d" % os.getpid()
    extra = list(extra_flags) if extra_flags is not None else os.environ.get("P2P_EXTRA_FLAGS", "").split()
    # every translation unit on its own (a few at a time: the three tile shapes of the view kernels are the long ones),
    # then one link -- a third of the time of one hipcc command over all sources
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    cflags = [f for f in FLAGS if f != "-shared"] + extra
    with tempfile.TemporaryDirectory(prefix="p2p_build_") as objdir:
        def compile_one(src):
            obj = os.path.join(objdir, os.path.basename(src) + ".o")
            cmd = [hipcc] + cflags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            return obj

        jobs = max(1, min(len(SOURCES), int(os.environ.get("P2P_BUILD_JOBS", "0")) or (os.cpu_count() or 2) // 2))
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            objs = list(ex.map(compile_one, SOURCES))
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", tmp] + objs
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
