"""Register / LDS / spill figures of every kernel in libcnmf_hip.so (read from the code object's metadata notes):
    python tools/kernel_resources.py [substring ...]
Unbundles the gfx950 code object from the shared library's .hip_fatbin section (clang-offload-bundler) and prints
`llvm-readelf --notes` per kernel: vgpr / agpr / sgpr counts, spills, LDS bytes, waves per SIMD the counts allow."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    so = os.environ.get("CNMF_LIB_PATH", os.path.join(ROOT, "cnmf_amd", "libcnmf_hip.so"))
    pats = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
        co = os.path.join(tmp, "gfx950.co")
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
        names = subprocess.run(["c++filt"], input=notes, capture_output=True, text=True).stdout
    rows = []
    for blk in re.split(r"\n\s+- \.agpr_count:", names)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, "?"])[1]
        name = re.search(r"\.name:\s+(.*)", blk)
        name = name.group(1).strip().strip("'") if name else "?"
        if pats and not any(p in name for p in pats):
            continue
        v, a = get("vgpr_count"), get("agpr_count")
        tot = (int(v) if v.isdigit() else 0)
        waves = min(8, 512 // max(8, (tot + 7) // 8 * 8)) if tot else 0
        rows.append((name, v, a, get("sgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"),
                     get("group_segment_fixed_size"), get("private_segment_fixed_size"), waves))
    print("%-6s %-5s %-5s %-7s %-7s %-8s %-8s %-5s %s" % ("vgpr", "agpr", "sgpr", "vspill", "sspill", "lds", "scratch", "waves", "kernel"))
    for r in sorted(rows):
        print("%-6s %-5s %-5s %-7s %-7s %-8s %-8s %-5s %s" % (r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[0][:150]))


if __name__ == "__main__":
    main()
