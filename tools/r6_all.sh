cd /root/repo; bash tools/r6_ablation.sh > /dev/null 2>&1; bash tools/r6_ab_memory.sh > /dev/null 2>&1; bash tools/r6_measure.sh all 2>&1 | tail -40
