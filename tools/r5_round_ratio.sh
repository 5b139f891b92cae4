# Same-box comparison of the round's first commit (in _before/, built from its own sources) with the current tree: config 5 step
# (alternating), one kernel trace each, the surrogate gradient call, and the 1-D headline step.   bash tools/r5_round_ratio.sh
cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r5ratio; mkdir -p $out
bash tools/r5_ab_trees.sh r5ratio 2>&1 | tail -8
for t in _before .; do (cd /root/repo/$t && timeout 600 python tools/bench_force.py 64 2 5 2>/dev/null | grep "design gradient" | sed "s#^#$t #"); done | tee $out/force.txt
for t in _before . _before .; do (cd /root/repo/$t && python tools/prof1d.py 256 600 2>/dev/null | sed "s#^#$t #"); done | tee $out/cfg2.txt
