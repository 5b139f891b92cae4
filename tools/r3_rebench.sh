cd /root/repo; out=gpurun_out/meas3; mkdir -p $out
for w in cfg2 cfg3 cfg4 cfg2-ddim250 cfg5; do python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err; cut -c1-120 $out/bench_$w.json; done
python bench.py --workload cfg5g --steps 1 --warmup 1 > $out/bench_cfg5g.json 2> $out/bench_cfg5g.err; cut -c1-120 $out/bench_cfg5g.json
