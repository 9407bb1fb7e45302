#!/usr/bin/env python3
"""Register / LDS / spill figures of a kernel as the CODE OBJECT states them (the AMDGPU metadata note of the gfx950 ELF inside the
library's clang offload bundle) -- not what a profiler's CSV prints (rocprofv3's VGPR_Count column said 40 for a 79-VGPR kernel:
VERDICT r5, weak #10).

    python tools/code_object_regs.py [library.so | object.o] [kernel-name substring ...]

As a module: kernel_resources(path) -> {demangled kernel name: {"vgpr", "agpr", "sgpr", "sgpr_spill", "vgpr_spill", "lds", "scratch", "kernarg"}}."""
import os
import struct
import subprocess
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _device_elves(blob):
    """every hip*-amdgcn...gfx950 code object of every offload bundle in the file"""
    at = 0
    while True:
        i = blob.find(MAGIC, at)
        if i < 0:
            return
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        p = i + 32
        end = i + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + ts].decode(errors="replace")
            p += ts
            end = max(end, i + off + size)
            if "amdgcn" in triple and "gfx950" in triple and size:
                yield blob[i + off:i + off + size]
        at = max(end, i + 24)


def _notes(elf):
    """(name, type, desc) of every note of a 64-bit little-endian ELF"""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for k in range(shnum):
        sh = elf[shoff + k * shentsize: shoff + (k + 1) * shentsize]
        sh_type, = struct.unpack_from("<I", sh, 4)
        off, size = struct.unpack_from("<QQ", sh, 0x18)
        if sh_type != 7:   # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def _demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [o.split("(")[0].replace("void zoic::", "") for o in out[:len(names)]]
    except OSError:
        return names


def kernel_resources(path):
    blob = open(path, "rb").read()
    res = {}
    for elf in _device_elves(blob):
        for name, ntype, desc in _notes(elf):
            if name != b"AMDGPU" or ntype != 32:   # NT_AMDGPU_METADATA (code object v3+): a msgpack map
                continue
            meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            kernels = meta.get("amdhsa.kernels", [])
            short = _demangle([k[".name"] for k in kernels])
            for k, s in zip(kernels, short):
                res[s] = {"vgpr": k.get(".vgpr_count"), "agpr": k.get(".agpr_count", 0), "sgpr": k.get(".sgpr_count"),
                          "sgpr_spill": k.get(".sgpr_spill_count", 0), "vgpr_spill": k.get(".vgpr_spill_count", 0),
                          "lds": k.get(".group_segment_fixed_size"), "scratch": k.get(".private_segment_fixed_size"),
                          "kernarg": k.get(".kernarg_segment_size")}
    return res


if __name__ == "__main__":
    args = sys.argv[1:]
    path = os.path.join(ROOT, "zoic_amd", "libzoic_amd.so")
    if args and os.path.exists(args[0]):
        path = args.pop(0)
    for name, r in sorted(kernel_resources(path).items()):
        if not args or any(a in name for a in args):
            print("%-52s vgpr %3s sgpr %3s spill s%-3s v%-3s lds %5s scratch %s" % (name[:52], r["vgpr"], r["sgpr"], r["sgpr_spill"], r["vgpr_spill"], r["lds"], r["scratch"]))
