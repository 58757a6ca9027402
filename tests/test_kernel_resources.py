"""What the compiler made of the view kernels that ship on the default paths: no scratch (a spilled register is a scratch
access per pair in kernels that write gigabytes), and register counts that keep the occupancy the LDS budget was chosen
for.  Read from the code objects inside the built libp2p_hip.so (llvm-objcopy -> clang-offload-bundler -> llvm-readelf
--notes): the shipped binary, not a recompilation.  No GPU needed."""
import importlib
import os
import re
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernels(tmp_path):
    build = importlib.import_module("360-to-planer-images_amd._build")
    so = build.OUT
    if not os.path.exists(so):
        build.build()
    fat = tmp_path / "fat.bin"
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so, str(tmp_path / "copy.so")], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = {}
    for i, a in enumerate(starts):
        b = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part = tmp_path / f"bundle{i}.bin"
        part.write_bytes(blob[a:b])
        co = tmp_path / f"code{i}.co"
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={part}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.(name|private_segment_fixed_size|vgpr_count|vgpr_spill_count):\s+(\S+)", line)
            if not m:
                continue
            if m.group(1) == "name":
                cur = out.setdefault(m.group(2), {})
            elif cur is not None:
                cur[m.group(1)] = int(m.group(2))
    return {k: v for k, v in out.items() if "vgpr_count" in v}


def _pick(kernels, *parts):
    hits = [(k, v) for k, v in kernels.items() if all(p in k for p in parts)]
    assert hits, (parts, sorted(kernels)[:5])
    return hits


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-readelf"), reason="no LLVM binutils")
def test_default_path_view_kernels_have_no_scratch_and_keep_their_occupancy(tmp_path):
    k = _kernels(tmp_path)
    # (mangled names: p2p::w64::remap_views_kernel<false> = _ZN3p2p3w6418remap_views_kernelILb0EEE...)
    budget = [
        (("3w64", "18remap_views_kernel"), 72),           # seven workgroups of four waves per CU
        (("4w128", "18remap_views_kernel"), 80),          # three workgroups of eight waves per CU
        (("4w64b", "23remap_views_band_kernel"), 96),     # the band shape: five workgroups of four waves per CU
        (("3w64", "25remap_views_gather_kernel"), 128),
        (("3w64", "23remap_views_rest_kernel"), 128),
        (("3w64", "24remap_views_table_kernel"), 96),    # (both border instances; round 5: 140 registers, three waves per SIMD)
        (("3w64", "15pair_ctx_kernel"), 64),
        # the float pixel path (BASELINE config 5's "fp16 pixel path"): six waves per SIMD, no scratch in either precision
        # and either tile shape (round 5: the f16 kernel kept two spilled registers, 12 bytes of scratch per lane)
        (("3w64", "18float_views_kernel"), 80),
        (("4w128", "18float_views_kernel"), 80),
    ]
    for parts, max_vgprs in budget:
        for name, r in _pick(k, *parts):
            assert r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0, (name, r)
            assert r["vgpr_count"] <= max_vgprs, (name, r, max_vgprs)
