# Same-box A/B of two source trees of the 1-D path: the current one and _before/ (git archive of a commit + its own built library),
# alternating processes.   bash tools/ab_trees_1d.sh [reps] [steps] [cfg2 | cfg3]
reps=${1:-4}; steps=${2:-600}; wl=${3:-cfg2}
for r in $(seq 1 $reps); do
  for t in _before .; do
    (cd /root/repo/$t && echo -n "$t: " && python3 tools/ab1d.py tune 0 0 $steps $wl 2>/dev/null | grep us/step | tail -1)
  done
done
