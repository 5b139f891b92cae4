# Round 6: what bounds dconv2_kernel's K loops?  Phase tables of the profiling build and of the three ablation builds of the same
# sources (cindm_amd/build.py: abl1 = no MFMAs -- the operands are still fetched and waited for --, abl2 = every stage re-reads stage
# 0's weight fragments (L2-hot), abl3 = both), printed side by side for three launches -> profiles/r06_ablation_kloops.txt
cd /root/repo; export TMPDIR=/tmp; out=gpurun_out/meas6; mkdir -p $out
for v in prof abl1 abl2 abl3; do
  CINDM_LIB_VARIANT=$v timeout 300 python tools/phase_table.py cfg2 40 > $out/phase_$v.txt 2> $out/phase_$v.err
done
python - > $out/r06_ablation_kloops.txt <<'PY'
import re
out = "/root/repo/gpurun_out/meas6"
names = {"prof": "everything in", "abl1": "no MFMAs", "abl2": "weights L2-hot", "abl3": "no MFMAs + L2-hot"}
print("# tools/r6_ablation.sh: dconv2_kernel phases (us, median over workgroups and waves) in the profiling build and the three ablation builds")
print("# (wrong results by design; abl1: the MFMAs are replaced by empty asm statements that keep every operand wait; abl2: load_b_tap reads stage 0 for every stage)")
for launch in ("mid_block1", "ups.0.0", "ups.1.0", "downs.2.1"):
    print(f"\n## {launch}")
    rows = {}
    for v in names:
        txt = open(f"{out}/phase_{v}.txt").read()
        m = re.search(r"#\s*\d+ (dconv2<[^>]*> " + re.escape(launch) + r"):[^\n]*\n((?:    [^\n]*\n)+)", txt)
        rows[v] = [(l[4:50].strip(), float(l[50:59])) for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith("phase") and "medians" not in l]
        head = m.group(1)
    print(f"# {head}")
    print(f"{'phase':46s} " + " ".join(f"{names[v]:>18s}" for v in names))
    for i, (lab, _) in enumerate(rows["prof"]):
        print(f"{lab:46s} " + " ".join(f"{rows[v][i][1]:18.2f}" for v in names))
for v in names:
    print(f"\n# {names[v]}: " + open(f"{out}/phase_{v}.txt").read().strip().split("\n")[-1])
PY
cat $out/r06_ablation_kloops.txt | head -30
