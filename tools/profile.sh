# rocprofv3 passes behind profiles/rNN_*: run on the GPU box from the repo root (copy gpurun_out/TAG_* and gpurun_out/traffic.json into profiles/ afterwards),
#   gpurun -- 'bash tools/profile.sh TAG'
# kernel-trace/stats and the PMC passes are separate runs (the pool refuses them combined).
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --workload S2 --steps 6 --warmup 2 --cpu-poses 0 --no-roofline --no-extras"
O=gpurun_out/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload S2 --steps 20 --warmup 5 --cpu-poses 0 --no-extras > $O.stats.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_a -- $B > $O.pmc_a.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_b -- $B > $O.pmc_b.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- $B > $O.pmc_f.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- $B > $O.pmc_w.log 2>&1
echo rc=$?
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_busy_S2.csv $O/pmc_a $O/pmc_b > /dev/null
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_traffic_S2.csv $O/pmc_f $O/pmc_w > /dev/null
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_S2_kernel_stats.csv
tail -1 $O.stats.log | cut -c1-400
head -25 gpurun_out/${TAG}_S2_kernel_stats.csv
# counter traffic per launch by bench.py's kernel names, stamped with the build that was profiled (bench.py reports it for that build only)
BID=$(python3 -c "import sys; sys.path.insert(0, 'icm-slam_amd'); from icmslam_hip import _lib; print(_lib.load().icm_build_id().decode())")
cp profiles/traffic.json gpurun_out/traffic.json 2>/dev/null
python3 tools/pmc_summary.py --traffic gpurun_out/traffic.json S2 gpurun_out/${TAG}_pmc_traffic_S2.csv $BID > /dev/null
sed -i "s#profiles/${TAG}_pmc_traffic_S2.csv#profiles/${TAG}_pmc_traffic_S2.csv#" gpurun_out/traffic.json
echo "build $BID"
# one steady-state sweep on every queue (from the stats pass's kernel trace)
python3 tools/sweep_timeline.py $O/stats > gpurun_out/${TAG}_sweep_timeline_S2.txt 2>&1
head -40 gpurun_out/${TAG}_sweep_timeline_S2.txt
