# Same-box comparison of the current tree with the tree of the round's first commit (_before/, built by hand: see DESIGN.md section 5).
# Boxes differ by +-3 % in the reverse-step time: only ratios measured on ONE box say anything about a code change.
cd /root/repo
for w in ${1:-cfg2}; do
  if [ -d _before ]; then (cd _before && python tools/ab1d.py pingpong 1 1 600 $w 2>&1 | grep "us/step" | tail -n 2 | sed "s/^/before $w /"); fi
  python tools/ab1d.py pingpong 1 1 600 $w 2>&1 | grep "us/step" | tail -n 2 | sed "s/^/now    $w /"
done
