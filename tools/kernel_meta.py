"""Per-kernel code-object metadata (registers, spills, scratch bytes) of a built library.

usage: python tools/kernel_meta.py [ddrl_amd/libddrl_pop.so]
The library's .hip_fatbin section holds one clang offload bundle per translation unit; each is unbundled for gfx950 and
its AMDGPU metadata notes are read with llvm-readelf.  tests/test_kernel_meta_cpu.py holds the product library to
"no kernel uses scratch memory, none spills vector registers" (by-value contexts and
widened scalar loads had put stack objects into the update kernels).
"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def tools_present():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump"))


def kernel_meta(so_path):
    """-> {kernel name: {"vgpr": n, "sgpr_spill": n, "vgpr_spill": n, "scratch": bytes, "lds": bytes}}"""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path, os.path.join(d, "unused")],
                       check=True, capture_output=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            part = os.path.join(d, "bundle%d.bin" % i)
            with open(part, "wb") as f:
                f.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, "dev%d.co" % i)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=" + TARGET, "--output=" + co], check=True, capture_output=True)
            if os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            cur = {}
            for ln in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", ln)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == "group_segment_fixed_size": cur = {"lds": int(v)}
                elif k == "name" and "lds" in cur and "name" not in cur: cur["name"] = v
                elif k == "private_segment_fixed_size": cur["scratch"] = int(v)
                elif k == "sgpr_spill_count": cur["sgpr_spill"] = int(v)
                elif k == "vgpr_count": cur["vgpr"] = int(v)
                elif k == "vgpr_spill_count":
                    cur["vgpr_spill"] = int(v)
                    if "name" in cur:
                        out[cur.pop("name")] = cur
                    cur = {}
    return out


def kernel_mfma_counts(so_path, name_filter):
    """-> {kernel name: number of v_mfma instructions in its code} for the kernels whose name contains name_filter
    (llvm-objdump -d of the gfx950 code objects).  A static schedule that LLVM left ROLLED shows up as a fraction of the
    fully unrolled count (and usually as scratch memory, which tests/test_kernel_meta_cpu.py checks separately)."""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path, os.path.join(d, "unused")],
                       check=True, capture_output=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            chunk = blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)]
            if name_filter.encode() not in chunk:
                continue
            part, co = os.path.join(d, "bundle%d.bin" % i), os.path.join(d, "dev%d.co" % i)
            with open(part, "wb") as f:
                f.write(chunk)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=" + TARGET, "--output=" + co], check=True, capture_output=True)
            if os.path.getsize(co) == 0:
                continue
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
            cur = None
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
                if m:
                    cur = m.group(1) if name_filter in m.group(1) else None
                    if cur:
                        out[cur] = 0
                elif cur and "v_mfma" in ln:
                    out[cur] += 1
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "ddrl_amd", "libddrl_pop.so")
    meta = kernel_meta(so)
    for name in sorted(meta):
        m = meta[name]
        print("%4d vgpr %4d sgpr-spill %3d vgpr-spill %4d scratch B %6d lds B  %s" % (m.get("vgpr", -1), m.get("sgpr_spill", -1),
              m.get("vgpr_spill", -1), m.get("scratch", -1), m.get("lds", -1), name[:110]))
    print(len(meta), "kernels")
