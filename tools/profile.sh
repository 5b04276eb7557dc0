cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --workload S2 --steps 3 --warmup 1 --cpu-poses 0 --no-roofline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v8 -- python bench.py --workload S2 --steps 10 --warmup 2 --cpu-poses 0 > gpurun_out/prof_v8.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc8a -- $B > gpurun_out/pmc8a.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc8f -- $B > gpurun_out/pmc8f.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc8w -- $B > gpurun_out/pmc8w.log 2>&1 && \
python tools/pmc_summary.py gpurun_out/pmc8_summary.csv gpurun_out/pmc8a gpurun_out/pmc8f gpurun_out/pmc8w > /dev/null && \
timeout -k 10 400 python bench.py > gpurun_out/bench_v8.log 2>&1 && \
timeout -k 10 300 python bench.py --workload S1 --steps 20 --warmup 3 > gpurun_out/bench_s1_v8.log 2>&1
echo rc=$?
tail -1 gpurun_out/bench_v8.log | cut -c1-600
find gpurun_out/prof_v8 -name "*kernel_stats.csv" | head -1 | xargs head -20
