cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "csr or two_ranks or sharded or sign or spmm or stable_grouping or full_size_configs_vs_oracle or hub" > gpurun_out/r5_t1.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5_t1.log
timeout 400 python tests/stress_csr_oversubscribed.py --procs 8 --iters 300 > gpurun_out/r5_oversub_new.txt 2>&1; echo "oversub new rc=$?"; tail -12 gpurun_out/r5_oversub_new.txt
timeout 400 python tests/stress_csr_oversubscribed.py --procs 8 --iters 300 --deadline 200 --lib tools/r4_lib/libsubgraph_sketch_r4.so > gpurun_out/r5_oversub_r4.txt 2>&1; echo "oversub r4 rc=$?"; tail -12 gpurun_out/r5_oversub_r4.txt
timeout 600 bash tools/kstats_cmd.sh r5csr python tools/probe_csr_large.py collab ppa citation2 > gpurun_out/r5_csr_kstats.txt 2>&1; cat gpurun_out/r5_csr_kstats.txt; grep -v "^$" gpurun_out/prof_r5csr_cmd.log | tail -12
