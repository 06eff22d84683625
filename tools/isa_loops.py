#!/usr/bin/env python3
"""Instruction mix of the basic blocks of one kernel in a hipcc -save-temps .s file.

    python tools/isa_loops.py build/isa/s4g_scan-hip-amdgcn-amd-amdhsa-gfx950.s sw_scan_pkc [min_instructions]

Prints every basic block with at least `min_instructions` (default 150) vector instructions: its label, the opcode
histogram, scratch accesses and s_nop count.  The column loops of the scan kernels are the large blocks."""
import collections
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*%s\w*:" % re.escape(name), l):
            start = i
            break
    if start is None:
        sys.exit(f"no function matching {name}")
    blocks, cur, label = [], [], lines[start].rstrip(":")
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        cur.append(t.split(";")[0].strip())
    blocks.append((label, cur))
    print(f"{lines[start].rstrip(':')}: {len(blocks)} blocks, {sum(len(b) for _, b in blocks)} instructions")
    for label, ins in blocks:
        ops = collections.Counter(i.split()[0] for i in ins)
        valu = sum(n for o, n in ops.items() if o.startswith("v_"))
        if valu < floor:
            continue
        scratch = sum(n for o, n in ops.items() if o.startswith("scratch_"))
        print(f"\n{label}: {len(ins)} instructions, {valu} vector, {ops.get('s_nop', 0)} s_nop, {scratch} scratch, "
              f"{sum(n for o, n in ops.items() if o.startswith('ds_'))} ds, {sum(n for o, n in ops.items() if o.startswith('global_'))} global")
        print("   " + ", ".join(f"{o} {n}" for o, n in ops.most_common() if o.startswith("v_")))
        branch = [i for i in ins if i.startswith(("s_cbranch", "s_branch"))]
        print("   branches: " + "; ".join(branch))


if __name__ == "__main__":
    main()
