"""Static load-pipeline audit of the library's gfx950 ISA.

    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o /tmp/cindm.s cindm_amd/csrc/cindm_hip.hip
    python tools/isa_audit.py /tmp/cindm.s                      rank every kernel by loads that are waited for where they are issued
    python tools/isa_audit.py /tmp/cindm.s dconv2_kernelILi3ELi4ELi0ELb0    list every s_waitcnt vmcnt(N) of one kernel (mangled-name substring)

For every `s_waitcnt vmcnt(N)` the YOUNGEST memory operation the wait covers is found (vmcnt retires in order on gfx9: waiting
for it drains every older one) and the MFMAs / instructions issued between that operation and the wait are counted.  A wait within
12 instructions of its load is an "immediate-wait load"; fewer than 20 MFMAs and 200 instructions is "short cover": an exposed L2 /
Infinity-Cache round trip unless something else hides it.  Stores count in vmcnt on gfx9 and are tracked, but only loads are ranked.
(A scratch translation unit that includes one header and instantiates one kernel compiles in 4 - 10 s and is the way to iterate.)"""
import re
import subprocess
import sys


def kernels(lines):
    for st, l in enumerate(lines):
        if re.match(r'^_ZN5cindm\w+:', l):
            en = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
            yield l.split(':')[0], [x.strip() for x in lines[st:en + 1] if x.strip() and not x.strip().startswith(';')]


def waits(body):
    """(instruction index, MFMAs so far, N, MFMAs of cover, instructions of cover, kind L|S, text of the op waited for)"""
    ops, nm = [], 0
    for i, t in enumerate(body):
        if 'v_mfma' in t:
            nm += 1
        if re.match(r'(global|buffer|flat)_(load|atomic)', t):
            ops.append((i, nm, 'L', t[:56]))
        elif re.match(r'(global|buffer|flat)_store', t):
            ops.append((i, nm, 'S', t[:56]))
        m = re.search(r'vmcnt\((\d+)\)', t)
        if t.startswith('s_waitcnt') and m:
            n = int(m.group(1))
            if len(ops) > n:
                y = ops[len(ops) - n - 1]
                yield i, nm, n, nm - y[1], i - y[0], y[2], y[3]
                ops = ops[len(ops) - n:]


def main():
    lines = open(sys.argv[1]).read().split('\n')
    if len(sys.argv) > 2:
        for name, body in kernels(lines):
            if sys.argv[2] in name:
                print(subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip())
                print("  idx   mfma#  vmcnt(N)  cover: mfma  instr   youngest operation waited for")
                for i, nm, n, cm, ci, kind, text in waits(body):
                    flag = "   <-- exposed" if kind == 'L' and cm < 20 and ci < 200 else ""
                    print(f"  {i:5d} {nm:5d}   {n:3d}      {cm:5d}  {ci:5d}   {text}{flag}")
        return
    rows = []
    for name, body in kernels(lines):
        imm = short = 0
        for i, nm, n, cm, ci, kind, text in waits(body):
            if kind != 'L':
                continue
            if ci < 12:
                imm += 1
            elif cm < 20 and ci < 200:
                short += 1
        rows.append((imm, short, sum('v_mfma' in t for t in body), name))
    rows.sort(reverse=True)
    for imm, short, nm, name in rows[:80]:
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:110]
        print(f"{imm:4d} immediate-wait loads  {short:4d} short-cover  {nm:5d} mfma  {dem}")


if __name__ == "__main__":
    main()
