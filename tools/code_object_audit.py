#!/usr/bin/env python3
"""Per-kernel resource audit of libprag.so's gfx950 code objects (no GPU needed).

  python tools/code_object_audit.py [path/to/libprag.so]

Extracts the offload bundles with llvm-objdump, reads the AMDGPU metadata notes with llvm-readelf and prints, for
every kernel, VGPRs / spilled VGPRs / scratch bytes per lane (.private_segment_fixed_size) / LDS.  Exit code 1 when
any kernel uses scratch: a spilled value reloads behind `s_waitcnt vmcnt(0)` and drains whatever the kernel keeps
in flight (cdna_hip_programming.md section 5.7 item 4).  tests/test_abi_cpu.py runs the same audit."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("name", "private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count",
          "group_segment_fixed_size", "agpr_count", "sgpr_count")


def kernels(lib_path):
    """[{name, demangled, private_segment_fixed_size, vgpr_count, ...}] for every kernel of every gfx950 bundle."""
    tmp = tempfile.mkdtemp(prefix="prag_co_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)            # llvm-objdump writes the bundles next to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                   check=True, capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
                if m and m.group(1) in FIELDS:
                    cur[m.group(1)] = m.group(2)
                if re.match(r"\s+\.wavefront_size:", line):     # last field of a kernel record
                    if "name" in cur:
                        out.append(cur)
                    cur = {}
        names = [k["name"] for k in out]
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().splitlines() \
            if names and shutil.which("c++filt") else names
        for k, d in zip(out, dem):
            k["demangled"] = d
            for f in FIELDS[1:]:
                k[f] = int(k.get(f, 0))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "probing-rag_amd", "lib", "libprag.so")
    ks = kernels(lib)
    bad = 0
    for k in sorted(ks, key=lambda r: (-r["private_segment_fixed_size"], -r["vgpr_count"])):
        flag = "  <-- scratch" if k["private_segment_fixed_size"] else ""
        bad += bool(k["private_segment_fixed_size"])
        print(f"{k['vgpr_count']:4d} vgpr {k['vgpr_spill_count']:3d} spilled {k['private_segment_fixed_size']:4d} B scratch "
              f"{k['group_segment_fixed_size']:6d} B static LDS  {k['demangled'][:150]}{flag}")
    print(f"{len(ks)} kernels, {bad} with scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
