"""Static load-pipeline audit of the library's gfx950 ISA.

    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o /tmp/cindm.s cindm_amd/csrc/cindm_hip.hip
    python tools/isa_audit.py /tmp/cindm.s                      rank every kernel by loads that are waited for where they are issued
    python tools/isa_audit.py /tmp/cindm.s dconv2_kernelILi3ELi4ELi0ELb0    list every s_waitcnt vmcnt(N) of one kernel (mangled-name substring)
    python tools/isa_audit.py /tmp/cindm.s --handover              flag-load-before-payload-load order of the in-launch hand-overs (exit 1 if violated)

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


def handover_order(body):
    """The acquire side of a flag + payload hand-over between workgroups (dconv2_kernel's y0 all-gather) is ordered
    by the COMPILER only: relaxed `sc1` flag loads in a poll loop, `asm volatile("" ::: "memory")`, then the `sc1` payload loads (VMEM
    issues in program order and the loop exit depends on the flag values).  This checks the emitted code: every 16-byte `sc1`
    payload load must come AFTER the end of the innermost loop that holds the 4-byte `sc1` flag loads.  Returns None when the kernel
    has no such pair (granule hand-overs carry their tag in the datum: 8-byte `sc1` loads, nothing to order), else
    (ok, flag indices, poll-loop end, first payload index)."""
    labels = {}
    for i, t in enumerate(body):
        m = re.match(r'^(\.LBB[0-9_]+):', t)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, t in enumerate(body):
        m = re.match(r's_cbranch_\w+\s+(\S+)', t) or re.match(r's_branch\s+(\S+)', t)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    flags = [i for i, t in enumerate(body) if re.match(r'(global|buffer|flat)_load_dword\s', t) and ' sc1' in t]
    payload = [i for i, t in enumerate(body) if re.match(r'(global|buffer|flat)_load_dwordx4\s', t) and ' sc1' in t]
    if not flags or not payload:
        return None
    ends = []
    for f in flags:
        inner = [lp for lp in loops if lp[0] <= f <= lp[1]]
        if not inner:
            return (False, flags, -1, min(payload))          # a flag load outside any loop: not a poll
        ends.append(min(inner, key=lambda lp: lp[1] - lp[0])[1])
    end = max(ends)
    after = [p_ for p_ in payload if p_ > min(flags)]
    first = min(after) if after else -1
    return (bool(after) and first > end and not any(min(flags) <= p_ <= end for p_ in payload), flags, end, first)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "--handover":
        lines = open(sys.argv[1]).read().split('\n')
        bad = 0
        for name, body in kernels(lines):
            if "dconv2_kernel" not in name and "attn1d_head_kernel" not in name and "dconv_kernel" not in name:
                continue
            r = handover_order(body)
            short = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:100]
            if r is None:
                print(f"{short:100s}  no flag + payload pair (granules only)")
            else:
                ok, flags, end, first = r
                bad += 0 if ok else 1
                print(f"{short:100s}  {'OK ' if ok else 'BAD'} {len(flags)} flag loads, poll loop ends at {end}, first payload load at {first}")
        sys.exit(1 if bad else 0)
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
