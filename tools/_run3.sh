cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r5_t2.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5_t2.log
hipcc --offload-arch=gfx950 -O3 -o /tmp/record_gather tools/micro/record_gather.hip && timeout 600 /tmp/record_gather > gpurun_out/r5_record_gather.txt 2>&1; cat gpurun_out/r5_record_gather.txt
timeout 900 bash tools/kstats_cmd.sh r5csr python $R/tools/probe_csr_large.py collab ppa citation2 > gpurun_out/r5_csr_kstats.txt 2>&1; cat gpurun_out/r5_csr_kstats.txt; grep -E "ok=" gpurun_out/prof_r5csr_cmd.log
timeout 300 python tests/stress_csr_oversubscribed.py --procs 8 --iters 60 --check-every 10 > gpurun_out/r5_oversub_new.txt 2>&1; echo "oversub new rc=$?"; grep -v amdgpu.ids gpurun_out/r5_oversub_new.txt | tail -20
SS_STRESS_LOG=mp_stress_new.txt timeout 900 bash tools/stress_multiproc.sh 10 20 > /dev/null 2>&1; echo "mp stress rc=$?"; grep -E "launch|summary|FAIL" gpurun_out/mp_stress_new.txt | tail -14
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_a.json 2> gpurun_out/r5_bench_a.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r5_bench_a.json
