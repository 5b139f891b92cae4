# Round 6 against round 5 on ONE box: _before/ = `git archive aff30e4` (round 5's final tree) with its own built library, . = this tree;
# alternating processes (tools/ab_trees_1d.sh).  -> gpurun_out/meas6/r06_round_ratio.txt
cd /root/repo; mkdir -p gpurun_out/meas6
(echo "# tools/r6_round_ratio.sh: round 5's final tree (_before, commit aff30e4) and this tree, alternating processes on one box; us per reverse step"
 echo "## cfg2 (256 rows)"; bash tools/ab_trees_1d.sh 4 600 cfg2
 echo "## cfg3 (768 rows)"; bash tools/ab_trees_1d.sh 3 300 cfg3) | tee gpurun_out/meas6/r06_round_ratio.txt
