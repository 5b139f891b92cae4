# Same-box comparison of the current tree with the tree of the round's first commit in _before/ (git-ignored; made by hand:
#   git worktree add /tmp/before <first commit of the round> && mkdir _before && (cd /tmp/before && tar cf - .) | (cd _before && tar xf -)
#   (cd _before && python -m cindm_amd.build)        # its own library, built from its own sources
# DESIGN.md section 5 quotes the ratios; round 4's first commit is de7b64f).
# Boxes differ by +-3 % in the reverse-step time: only ratios measured on ONE box say anything about a code change.
cd /root/repo
for w in ${1:-cfg2}; do
  if [ -d _before ]; then (cd _before && python tools/ab1d.py pingpong 1 1 600 $w 2>&1 | grep "us/step" | tail -n 2 | sed "s/^/before $w /"); fi
  python tools/ab1d.py pingpong 1 1 600 $w 2>&1 | grep "us/step" | tail -n 2 | sed "s/^/now    $w /"
done
