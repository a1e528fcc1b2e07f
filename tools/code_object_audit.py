"""Registers, scratch and LDS of every kernel in the built library, read from the gfx950 code objects inside
libcq_halo2.so (the clang offload bundles of its .hip_fatbin section, their AMDGPU metadata notes via llvm-readelf).
A kernel that starts to spill (scratch > 0) or crosses an occupancy step shows up here, not in a timing a week later.
   python3 tools/code_object_audit.py [substring ...]      # name, vgprs, sgprs, scratch bytes, LDS bytes, waves per SIMD"""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "sha2_on_cq_halo2_amd", "libcq_halo2.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path=LIB):
    """the gfx950 ELF images of every bundle in the file"""
    blob = open(path, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return out
        (count,) = struct.unpack_from("<Q", blob, pos + 24)
        p = pos + 32
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += 24


def kernels(path=LIB):
    """{demangled-ish kernel name: dict(vgpr, sgpr, scratch, lds, max_flat_workgroup_size)}"""
    res = {}
    for image in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(image)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for block in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            get = lambda key: re.search(r"\." + key + r":\s*(\S+)", block)
            name = get("name").group(1)
            res[name] = {"vgpr": int(get("vgpr_count").group(1)), "sgpr": int(get("sgpr_count").group(1)),
                         "scratch": int(get("private_segment_fixed_size").group(1)), "lds": int(get("group_segment_fixed_size").group(1)),
                         "threads": int(get("max_flat_workgroup_size").group(1))}
    return res


def waves_per_simd(vgpr):
    return min(8, 512 // max(8, (vgpr + 7) // 8 * 8))


if __name__ == "__main__":
    want = sys.argv[1:]
    ks = kernels()
    for name in sorted(ks):
        if want and not any(w in name for w in want):
            continue
        k = ks[name]
        print("%-90s vgpr %3d sgpr %3d scratch %4d lds %6d waves/SIMD %d" % (name[:90], k["vgpr"], k["sgpr"], k["scratch"], k["lds"], waves_per_simd(k["vgpr"])))
    print(len(ks), "kernels;", sum(1 for k in ks.values() if k["scratch"]), "with scratch")
