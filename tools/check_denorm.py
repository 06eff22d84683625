#!/usr/bin/env python3
"""Build-time guard for the integer-max3-on-f16-bit-patterns trick (sw_kernels.hip.h, ArithI16M3 / sw_scan_pkd / sw_scan_lists):
v_pk_maximum3_f16 orders raw integer bit patterns only while f16 DENORMALS ARE PRESERVED -- with B = 64 most stored values are
denormal patterns -- i.e. while every such kernel's descriptor says FLOAT_DENORM_MODE_16_64 = 3 (compute_pgm_rsrc1 bits 19:18).
A build with -fgpu-flush-denormals-to-zero / -ffast-math would clear it and be silently wrong until a test ran.

usage: tools/check_denorm.py <libs4g_hip.so>      (run by `make all`; exits non-zero naming the offending kernels)
Extracts the gfx950 code objects from the library's offload bundles (llvm-objdump --offloading, in a scratch directory) and
reads each kernel descriptor (<kernel>.kd, 64 bytes in .rodata) out of the ELF."""
import os, re, shutil, struct, subprocess, sys, tempfile

LLVM = os.environ.get("S4G_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
PACKED = re.compile(r"sw_scan_pk|sw_scan_lists|clk_probe|denorm_selftest")      # kernels whose arithmetic relies on the mode


def descriptors(elf):
    """-> {kernel name: compute_pgm_rsrc1} for every <name>.kd symbol of a code object"""
    secs = {}
    out = subprocess.run([f"{LLVM}/llvm-readelf", "-S", "-W", elf], capture_output=True, text=True, check=True).stdout
    for m in re.finditer(r"\[\s*(\d+)\]\s+(\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", out):
        secs[int(m.group(1))] = (int(m.group(3), 16), int(m.group(4), 16))       # index -> (address, file offset)
    data = open(elf, "rb").read()
    res = {}
    out = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "-W", elf], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        f = line.split(None, 7)
        if len(f) < 8 or not f[7].endswith(".kd") or not f[6].isdigit():
            continue
        addr, sec = int(f[1], 16), int(f[6])
        base, off = secs[sec]
        kd = data[off + addr - base: off + addr - base + 64]
        res[f[7][:-3]] = struct.unpack_from("<I", kd, 48)[0]
    return res


def main(lib):
    tmp = tempfile.mkdtemp(prefix="s4g_denorm_")
    try:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", copy], capture_output=True, check=True)
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f]
        if not objs:
            print(f"check_denorm: no device code objects found in {lib}", file=sys.stderr)
            return 2
        checked, bad = 0, []
        for o in objs:
            for name, rsrc1 in descriptors(o).items():
                if PACKED.search(name):
                    checked += 1
                    if (rsrc1 >> 18) & 3 != 3:
                        bad.append((name, (rsrc1 >> 18) & 3))
        if bad or not checked:
            for name, mode in bad:
                print(f"check_denorm: {name}: FLOAT_DENORM_MODE_16_64 = {mode}, must be 3 (f16 denormals preserved)", file=sys.stderr)
            if not checked:
                print("check_denorm: no packed kernel found: the name filter is stale", file=sys.stderr)
            return 1
        print(f"check_denorm: {checked} packed kernels, all with FLOAT_DENORM_MODE_16_64 = 3")
        return 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
