# Round measurement on the GPU box (run through gpurun): bench lines, rocprofv3 kernel-trace summaries and the two
# separate PMC passes per workload; results land in gpurun_out/ and are copied into profiles/ by hand.
#   usage: bash tools/gpu_round_measure.sh [cfg2|cfg5|all]
what=${1:-all}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/meas
if [ "$what" = cfg2 ] || [ "$what" = all ]; then
cd /tmp
rm -rf /tmp/k2; rocprofv3 --kernel-trace -d /tmp/k2 -o c2 -- python3 /root/repo/tools/prof1d.py 256 50 > /dev/null 2>&1
rm -rf /tmp/pmc2_fetch /tmp/pmc2_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc2_fetch -- python3 /root/repo/tools/prof1d.py 256 20 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc2_write -- python3 /root/repo/tools/prof1d.py 256 20 > /dev/null 2>&1
cd /root/repo
python3 tools/rocprof_summary.py $(find /tmp/k2 -name "*.db" | head -1) gpurun_out/meas/kstats_cfg2.txt > /dev/null
python3 tools/pmc_traffic.py /tmp/pmc2_fetch /tmp/pmc2_write 22 > gpurun_out/meas/pmc_traffic_cfg2.json
cp gpurun_out/meas/pmc_traffic_cfg2.json profiles/r02_pmc_traffic_cfg2.json      # bench.py reads the committed name
python bench.py > gpurun_out/meas/bench_cfg2.json 2> gpurun_out/meas/bench_cfg2.err
cat gpurun_out/meas/bench_cfg2.json
fi
if [ "$what" = cfg5 ] || [ "$what" = all ]; then
cd /tmp
rm -rf /tmp/k5; rocprofv3 --kernel-trace -d /tmp/k5 -o c5 -- python3 /root/repo/tools/prof2d.py 64 2 10 > /dev/null 2>&1
rm -rf /tmp/pmc5_fetch /tmp/pmc5_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc5_fetch -- python3 /root/repo/tools/prof2d.py 64 2 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc5_write -- python3 /root/repo/tools/prof2d.py 64 2 5 > /dev/null 2>&1
cd /root/repo
python3 tools/rocprof_summary.py $(find /tmp/k5 -name "*.db" | head -1) gpurun_out/meas/kstats_cfg5.txt > /dev/null
python3 tools/pmc_traffic.py /tmp/pmc5_fetch /tmp/pmc5_write 7 > gpurun_out/meas/pmc_traffic_cfg5.json
cp gpurun_out/meas/pmc_traffic_cfg5.json profiles/r02_pmc_traffic_cfg5.json      # bench.py reads the committed name
python bench.py --workload cfg5 > gpurun_out/meas/bench_cfg5.json 2> gpurun_out/meas/bench_cfg5.err
cat gpurun_out/meas/bench_cfg5.json
fi
