"""Instruction mix of a kernel's basic blocks from hipcc's gfx950 assembly.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -I include --save-temps -c <src>.hip -o /tmp/x.o   (in a scratch dir)
    python3 tools/isa_mix.py <src>-hip-amdgcn-amd-amdhsa-gfx950.s <kernel name substring> [min block size]

Prints, for every basic block of at least `min` instructions, the opcode histogram, and the kernel's register / occupancy
line.  Used for DESIGN.md section 7's "what the accumulate kernel's 2 259 instructions per addition are" table.
"""
import collections
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for si, s in enumerate(starts):
        if want not in lines[s]:
            continue
        end = starts[si + 1] if si + 1 < len(starts) else len(lines)
        body = lines[s:end]
        print(lines[s].split(":")[0])
        blocks, cur = [], ["entry", []]
        blocks.append(cur)
        for l in body[1:]:
            l = l.strip()
            if re.match(r"^\.LBB\d+_\d+:", l):
                cur = [l.split(":")[0], []]
                blocks.append(cur)
            elif l and not l.startswith((";", ".")):
                cur[1].append(l.split()[0])
        total = sum(len(b[1]) for b in blocks)
        print(f"  {total} instructions in {len(blocks)} blocks")
        for name, ins in blocks:
            if len(ins) < floor:
                continue
            c = collections.Counter(ins)
            print(f"  {name}: {len(ins)} instructions")
            for op, k in c.most_common(14):
                print(f"      {k:5d}  {op}")
        for l in body:
            if re.search(r"; (NumVgprs|NumSgprs|Occupancy|ScratchSize|LDSByteSize)", l):
                print(" ", l.strip("; ").strip())


if __name__ == "__main__":
    main()
