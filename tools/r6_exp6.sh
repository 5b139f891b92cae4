# Round 6, experiment 6: dconv2's outputs written through (tune bit 2 = 4) -- does the end-of-kernel L2 write-back show in the gaps?
cd /root/repo; export TMPDIR=/tmp; out=gpurun_out/r6; mkdir -p $out
for t in 0; do python tools/ab1d.py tune 3 $t 600 cfg2 | grep us/step; done > $out/ab_tune6.txt 2>&1
cat $out/ab_tune6.txt
for t in 0; do
CINDM_LIB_VARIANT=prof PHASE_OPTS=tune=$t timeout 300 python tools/phase_table.py cfg2 40 > $out/phase6_cfg2_tune$t.txt 2> $out/phase6_$t.err
tail -1 $out/phase6_cfg2_tune$t.txt
done
python tools/ab1d.py tune 3 0 300 cfg3 | grep us/step
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
