# Same-box A/B of two source trees (the current one and _before/, each with its own built library): step time of config 5 in
# alternation and one kernel trace each -> gpurun_out/<tag>/{before,now}_launches.txt.   bash tools/r5_ab_trees.sh <tag>
tag=${1:-r5ab}; export TMPDIR=/tmp; out=/root/repo/gpurun_out/$tag; mkdir -p $out
for rep in 1 2; do
  for t in _before .; do
    (cd /root/repo/$t && python3 tools/prof2d.py 64 2 20 2>/dev/null | sed "s#^#$t #")
  done
done | tee $out/steps.txt
for t in _before .; do
  name=now; [ "$t" = _before ] && name=before
  cd /tmp; rm -rf /tmp/kt_$name
  rocprofv3 --kernel-trace -d /tmp/kt_$name -o t -- python3 /root/repo/$t/tools/prof2d.py 64 2 10 > /tmp/kt_$name.log 2>&1
  cd /root/repo
  python3 tools/trace_gaps.py $(find /tmp/kt_$name -name "*.db" | head -1) stem7 > $out/${name}_launches.txt
done
paste <(cut -c1-60,68-78 $out/before_launches.txt) <(cut -c68-78 $out/now_launches.txt) | awk 'NR>2{printf "%s %8.3f\n", $0, $NF/$(NF-1)}' > $out/ratio.txt
head -1 $out/before_launches.txt; head -1 $out/now_launches.txt
