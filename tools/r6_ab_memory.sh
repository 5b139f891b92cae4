# Round 6: the two memory-system changes of the 1-D step, one at a time and together, same process / same box (option "tune": bit 0 =
# round 5's L2 warm-up placement + regions, bit 1 = round 5's plain output stores; 0 = shipped) -> profiles/r06_ab_memory_system.txt
cd /root/repo; export TMPDIR=/tmp; out=gpurun_out/meas6; mkdir -p $out
(echo "# tools/r6_ab_memory.sh: python tools/ab1d.py tune 3 <v> 600 <workload>  (3 = round 5's choices; 2 = late warm-up only; 1 = write-through outputs only; 0 = both = shipped)"
for v in 2 1 0; do python tools/ab1d.py tune 3 $v 600 cfg2 | grep us/step; done
echo "# cfg3 (768 rows)"; python tools/ab1d.py tune 3 0 300 cfg3 | grep us/step) > $out/r06_ab_memory_system.txt 2>&1
cat $out/r06_ab_memory_system.txt
