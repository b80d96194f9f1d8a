cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for m in 1 2; do DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_t$m.so python tools/phase_timers.py $m 2>&1 | grep level | tee -a gpurun_out/${TAG}_phase_timers.txt; done
rm -f facebook360_dep_amd/libderp_var_t1.so facebook360_dep_amd/libderp_var_t2.so
for v in $PARITY_VARIANTS; do
  DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/${TAG}_parity_$v.txt 2>&1
  echo "$v parity: $(tail -1 gpurun_out/${TAG}_parity_$v.txt)"
done
VARIANTS_NO_PARITY=1 tools/variants.sh 2>&1 | tee gpurun_out/${TAG}_variants.txt
VARIANTS_NO_PARITY=1 tools/variants.sh 2>&1 | tee -a gpurun_out/${TAG}_variants.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -m gpu -k "levels_2_and_1 or clamped" -s 2>&1 | grep -E "oracle|differ|passed|failed|Error|assert" | tee gpurun_out/${TAG}_newtests.txt
