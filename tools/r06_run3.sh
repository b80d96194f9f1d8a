cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail > gpurun_out/r6_list_avail.txt 2>&1
for v in $PARITY_VARIANTS; do
  DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/${TAG}_parity_$v.txt 2>&1
  echo "$v parity: $(tail -1 gpurun_out/${TAG}_parity_$v.txt)"
done
VARIANTS_NO_PARITY=1 tools/variants.sh 2>&1 | tee gpurun_out/${TAG}_variants.txt
