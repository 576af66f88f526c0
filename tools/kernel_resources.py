#!/usr/bin/env python3
"""Register / spill / LDS / kernel-argument figures of every kernel in a built libuavx.so, read from the code object's
own metadata (the `amdhsa.kernels` note the compiler writes; nothing is executed, no GPU needed).

    python tools/kernel_resources.py [path/to/libuavx.so] [--filter step_ex] [--json]

The library is a host ELF whose `.hip_fatbin` section holds one clang offload bundle per translation unit; each bundle holds
the gfx950 code object.  llvm-objcopy dumps the section, the bundles are cut apart at their magic string, clang-offload-bundler
extracts the device ELF and `llvm-readelf --notes` prints the metadata (YAML-like text, parsed below).

Waves per SIMD on gfx950 (512 VGPRs per lane and SIMD in blocks of 8, at most 8 wavefronts; 800 SGPRs per SIMD in blocks of 16
with 16 more per wavefront for the trap handler): min(8, 512 // roundup(vgpr, 8), 800 // (roundup(sgpr, 16) + 16)).
tests/test_host_cpu.py uses `kernel_table()` to keep the hot kernels free of spills.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("UAVX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "gym_uav_collision_avoidance_amd", "csrc", "libuavx.so")
BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "kernarg_segment_size", "max_flat_workgroup_size")


def demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True)
            return out.stdout.split("\n")[:len(names)]
        except (OSError, subprocess.CalledProcessError):
            continue
    return list(names)


def waves_per_simd(vgpr, sgpr):
    v = max(8, -(-vgpr // 8) * 8)
    s = max(16, -(-sgpr // 16) * 16) + 16      # + the trap handler's 16
    return max(1, min(8, 512 // v, 800 // s))


def device_objects(lib, workdir):
    """The gfx950 code objects inside `lib` (paths of extracted files)."""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(workdir, "unused.so")],
                   check=True, capture_output=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(BUNDLE_MAGIC), blob)]
    out = []
    for k, s in enumerate(starts):
        part = os.path.join(workdir, f"bundle{k}.bin")
        open(part, "wb").write(blob[s:(starts[k + 1] if k + 1 < len(starts) else len(blob))])
        listing = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", f"--input={part}"],
                                 capture_output=True, text=True, check=True).stdout.split()
        for target in listing:
            if "amdgcn" not in target:
                continue
            co = os.path.join(workdir, f"dev{k}.co")
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                            f"--targets={target}", f"--output={co}"], check=True, capture_output=True)
            out.append(co)
    return out


def kernel_table(lib=DEFAULT_LIB):
    """[{name (demangled), symbol, vgpr_count, sgpr_count, sgpr_spill_count, ..., waves_per_simd}] for every kernel of `lib`."""
    rows = []
    with tempfile.TemporaryDirectory() as wd:
        for co in device_objects(lib, wd):
            text = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            # one "  - .agpr_count: ..." item per kernel under amdhsa.kernels; items start with a line "  - .<key>:"
            body = text.split("amdhsa.kernels:", 1)[1] if "amdhsa.kernels:" in text else ""
            for item in re.split(r"\n  - (?=\.)", body)[1:]:
                item = item.split("\namdhsa.", 1)[0]
                m = re.search(r"^\s*\.name:\s+(\S+)", item, re.M)
                if not m:
                    continue
                row = {"symbol": m.group(1)}
                for f in FIELDS:
                    mm = re.search(rf"^\s*\.{f}:\s+(\d+)", item, re.M)
                    row[f] = int(mm.group(1)) if mm else 0
                rows.append(row)
    for row, name in zip(rows, demangle([r["symbol"] for r in rows])):
        row["name"] = re.sub(r"\(.*", "", name.replace("void ", "")).replace("uavx::", "")
        row["waves_per_simd"] = waves_per_simd(row["vgpr_count"] + row["agpr_count"], row["sgpr_count"])
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=DEFAULT_LIB)
    ap.add_argument("--filter", default="", help="only kernels whose demangled name contains this")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    rows = [r for r in kernel_table(a.lib) if a.filter in r["name"]]
    if a.json:
        json.dump(rows, sys.stdout, indent=1)
        return
    print(f"{'kernel':58s} vgpr sgpr s-spill v-spill scratch  lds kernarg waves/SIMD")
    for r in sorted(rows, key=lambda r: r["name"]):
        print(f"{r['name'][:58]:58s} {r['vgpr_count']:4d} {r['sgpr_count']:4d} {r['sgpr_spill_count']:7d} {r['vgpr_spill_count']:7d} "
              f"{r['private_segment_fixed_size']:7d} {r['group_segment_fixed_size']:4d} {r['kernarg_segment_size']:7d} {r['waves_per_simd']:6d}")


if __name__ == "__main__":
    main()
