#!/usr/bin/env python3
"""The gfx950 code objects inside the built csrc/*.o files: per-kernel register counts / spills and instruction counts.

    python tools/codeobj.py                     # table of every kernel: VGPRs, AGPRs, SGPRs, spills, LDS, scratch
    python tools/codeobj.py --grep 'v_pk_\\w+_f32'   # instructions matching a regex, per object and kernel
    python tools/codeobj.py --save FILE.json    # the table as JSON (before / after comparisons)

Recipe (no GPU needed): llvm-objcopy --dump-section .hip_fatbin -> clang-offload-bundler --unbundle -> llvm-readelf --notes /
llvm-objdump -d.  tests/test_host_logic.py uses packed_fp32_instructions() as a build-time check."""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tinydiffusionmodels_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
PACKED_FP32 = r"\bv_pk_(mul|add|fma)_f32\b"


def objects():
    return sorted(glob.glob(os.path.join(CSRC, "*.o")))


def unbundle(obj, tmpdir):
    """-> path of the gfx950 code object of one host object file, or None if it holds no device code."""
    base = os.path.join(tmpdir, os.path.basename(obj))
    fat, co = base + ".fatbin", base + ".co"
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return None
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={fat}", f"--output={co}"],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
        return None
    return co


def disassemble(co):
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout


def kernel_meta(co):
    out = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    rows, cur = [], None
    for line in out.splitlines():
        m = re.match(r"\s*(-\s+)?\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        if m.group(1) and m.group(2) in ("agpr_count", "args"):
            cur = {}
            rows.append(cur)
        if cur is not None and m.group(2) in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "name",
                                              "group_segment_fixed_size", "private_segment_fixed_size"):
            v = m.group(3).strip().strip("'")
            cur[m.group(2)] = int(v) if v.isdigit() else v
    return [r for r in rows if "name" in r and "vgpr_count" in r]


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def grep_instructions(regex):
    """{object: {kernel: [instruction lines]}} for every shipped code object."""
    rx = re.compile(regex)
    found = {}
    with tempfile.TemporaryDirectory() as td:
        for obj in objects():
            co = unbundle(obj, td)
            if co is None:
                continue
            cur = None
            for line in disassemble(co).splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    continue
                if rx.search(line):
                    found.setdefault(os.path.basename(obj), {}).setdefault(cur, []).append(line.strip().split("//")[0].strip())
    return found


def packed_fp32_instructions():
    return grep_instructions(PACKED_FP32)


def table():
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for obj in objects():
            co = unbundle(obj, td)
            if co is None:
                continue
            meta = kernel_meta(co)
            for r, d in zip(meta, demangle([r["name"] for r in meta])):
                r["object"], r["demangled"] = os.path.basename(obj), d
                rows.append(r)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grep")
    ap.add_argument("--save")
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    if a.grep:
        f = grep_instructions(a.grep)
        n = 0
        for o, ks in f.items():
            for k, lines in ks.items():
                n += len(lines)
                print(f"{o}: {demangle([k])[0][:110]}: {len(lines)}")
                for l in lines[:4]:
                    print("      " + l)
        print(f"total {n}")
        return 0
    rows = table()
    if a.save:
        json.dump(rows, open(a.save, "w"), indent=0)
    for r in rows:
        if a.filter in r["demangled"]:
            print(f'{r["object"]:18s} {r["demangled"][:96]:96s} v{r["vgpr_count"]:>4} a{r["agpr_count"]:>4} s{r["sgpr_count"]:>4} '
                  f'spill v{r.get("vgpr_spill_count", 0)}/s{r.get("sgpr_spill_count", 0)} lds {r.get("group_segment_fixed_size", 0):>6} '
                  f'scratch {r.get("private_segment_fixed_size", 0)}')
    return 0


if __name__ == "__main__":
    sys.exit(main())
