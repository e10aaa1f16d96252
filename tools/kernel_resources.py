#!/usr/bin/env python3
"""Per-kernel resources of the code objects INSIDE a built libflashe_hip*.so: VGPRs, SGPRs, scratch and static LDS as the kernel
descriptors carry them (the numbers the hardware allocates by) -- not rocprofv3's per-dispatch VGPR_Count column, which reports the
same 52 / 112 for nearly every kernel of this library.

How: the .hip_fatbin section of the shared library is a sequence of clang offload bundles (one per translation unit); each bundle's
gfx950 entry is an ELF code object whose NT_AMDGPU_METADATA note (llvm-readelf --notes) lists every kernel with .vgpr_count,
.sgpr_count, .private_segment_fixed_size, .group_segment_fixed_size.  Names are demangled with c++filt so that they equal
rocprofv3's Kernel_Name column.

usage: kernel_resources.py [lib.so] [out.json]      (defaults: flashe_amd/libflashe_hip.so, stdout)"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """The gfx9xx ELF images of every bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(tmp, "unused.so")])
        blob = open(fat, "rb").read()
    at = blob.find(MAGIC)
    while at >= 0:
        (n,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                yield triple, blob[at + off:at + off + size]
        at = blob.find(MAGIC, at + len(MAGIC))


def kernels_of(image):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
    # the metadata prints as YAML: one "  - .agpr_count: ..." block per kernel under amdhsa.kernels
    for block in re.split(r"\n\s+- \.", notes.split("amdhsa.kernels:")[-1])[1:]:
        fields = dict(re.findall(r"\.?([a-z_]+):\s+'?([^\n']+)'?", "." + block))
        if "name" not in fields or "vgpr_count" not in fields:
            continue
        yield fields["name"], {"vgpr": int(fields["vgpr_count"]), "agpr": int(fields.get("agpr_count", 0)), "sgpr": int(fields["sgpr_count"]),
                               "scratch_bytes_per_lane": int(fields.get("private_segment_fixed_size", 0)),
                               "lds_bytes_static": int(fields.get("group_segment_fixed_size", 0)),
                               "max_workgroup": int(fields.get("max_flat_workgroup_size", 0)),
                               "vgpr_spills": int(fields.get("vgpr_spill_count", 0)), "sgpr_spills": int(fields.get("sgpr_spill_count", 0))}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return dict(zip(names, out))


def norm(name):
    """Kernel names compare equal whatever spacing / `void ` prefix the demangler at hand uses."""
    return re.sub(r"\s+", "", name.replace("void ", ""))


def resources(lib):
    found = {}
    for _triple, image in code_objects(lib):
        found.update(dict(kernels_of(image)))
    pretty = demangle(list(found))
    return {pretty[k]: dict(v, mangled=k) for k, v in found.items()}


def waves_per_simd(vgpr, agpr=0):
    alloc = -(-max(vgpr + agpr, 1) // 8) * 8
    return min(8, 512 // alloc)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "flashe_amd", "libflashe_hip.so")
    res = resources(lib)
    for v in res.values():
        v["waves_per_simd_by_vgpr"] = waves_per_simd(v["vgpr"], v["agpr"])
    text = json.dumps({"library": os.path.basename(lib), "kernels": dict(sorted(res.items()))}, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    else:
        print(text)
