"""Registers, scratch and spills of the kernels in a built object (or in every build/gecm_kernels_*_p2.o): read from the
AMDGPU metadata note of the gfx950 code object inside the object's .hip_fatbin section.
usage: python tools/kernel_resources.py [kernel-name-substring] [object ...]"""
import os, re, subprocess, sys, tempfile, glob
LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def kernels(obj):
    """[(demangled-ish name, vgprs, agprs, sgprs, scratch bytes, vgpr spills)] of one host object."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + co], check=True, stderr=subprocess.DEVNULL)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        agpr = int(re.match(r"\s*(\d+)", blk).group(1))
        out.append((name, g("vgpr_count"), agpr, g("sgpr_count"), g("private_segment_fixed_size"), g("vgpr_spill_count")))
    return out


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    objs = sys.argv[2:] or sorted(glob.glob(os.path.join(ROOT, "avx-ecm_amd", "build", "gecm_kernels_*_p2.o")),
                                  key=lambda p: int(re.search(r"_(\d+)_p2", p).group(1)))
    for o in objs:
        for name, v, a, s, scr, sp in kernels(o):
            if pat in name:
                print("%-28s %-44s vgpr %3d agpr %3d sgpr %3d scratch %5d B spills %d" % (os.path.basename(o), name[:44], v, a, s, scr, sp))
