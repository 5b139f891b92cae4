# Round 6, experiment 4: what bounds dconv2_kernel's K loops?  Phase tables of the profiling build and of the three ablation builds
# (cindm_amd/build.py: abl1 = no MFMAs, abl2 = every stage re-reads stage 0's fragments (L2-hot), abl3 = both).
cd /root/repo; export TMPDIR=/tmp; out=gpurun_out/r6; mkdir -p $out
for v in prof abl1 abl2 abl3; do
CINDM_LIB_VARIANT=$v PHASE_OPTS=tune=304 timeout 300 python tools/phase_table.py cfg2 40 > $out/phase4_$v.txt 2> $out/phase4_$v.err
tail -1 $out/phase4_$v.txt
done
