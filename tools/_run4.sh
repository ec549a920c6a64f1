cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 600 -k "csr_is_reused or elph or deferred" > gpurun_out/r3j_tests.txt 2>&1; tail -8 gpurun_out/r3j_tests.txt
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-secondary --sustain-seconds 0 "$@" 2>/dev/null | tail -1 > gpurun_out/r3j_$name.json; python -c "import json; d=json.load(open('gpurun_out/r3j_$name.json')); print('$name', round(d['ms_per_step'],4), 'ms/step', {k: round(v['mean_launch_ms']*1e3,1) for k,v in d.get('kernels',{}).items() if isinstance(v,dict)})"; }
run elph2048 --api elph --batch 2048
SS_REUSE_CSR=0 run elph2048_noreuse --api elph --batch 2048
run elph65536 --api elph
run default
