cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 900 -k "csr or mega or hub or edge_cases or deferred_table or fused_hop or powerlaw or skew or full_size_configs_vs_oracle" > gpurun_out/r3c_tests1.txt 2>&1
tail -15 gpurun_out/r3c_tests1.txt
for t in "pl09 --graph powerlaw --alpha 0.9" "pl05 --graph powerlaw --alpha 0.5" "uni"; do set -- $t; tag=r3c_$1; shift; timeout 600 bash tools/kstats.sh $tag "$@" > gpurun_out/$tag.txt 2>&1; cat gpurun_out/$tag.txt; done
