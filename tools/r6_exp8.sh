# Round 6, experiment 8: where a kernel issues the next launch's L2 touches (tune bit 3 = 8: behind its LAST load request)
cd /root/repo; export TMPDIR=/tmp; out=gpurun_out/r6; mkdir -p $out
python tools/ab1d.py tune 0 8 600 cfg2 | grep us/step | tee $out/ab_tune8.txt
python tools/ab1d.py tune 0 8 300 cfg3 | grep us/step | tee -a $out/ab_tune8.txt
CINDM_LIB_VARIANT=prof PHASE_OPTS=tune=8 timeout 300 python tools/phase_table.py cfg2 40 > $out/phase8_cfg2_tune8.txt 2> $out/phase8.err; tail -1 $out/phase8_cfg2_tune8.txt
