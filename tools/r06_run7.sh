cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in $PARITY_VARIANTS; do
  DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/${TAG}_parity_$v.txt 2>&1
  echo "$v parity: $(tail -1 gpurun_out/${TAG}_parity_$v.txt)"
done
for rep in 1 2; do
for lib in facebook360_dep_amd/libderp_var_*.so; do
  name=$(basename $lib .so)
  DERP_LIB=$PWD/$lib timeout 600 python bench.py --frames 2 --steps 3 --warmup 1 --no-cpu-baseline --no-single-frame > /tmp/v.json 2>/tmp/v.err || { echo "$lib FAILED"; tail -3 /tmp/v.err; continue; }
  python - "$name" <<'PY'
import json, sys
d = json.load(open("/tmp/v.json"))
s = d["stage_ms_per_step"]
print("%-28s %7.1f Mpix/s %7.2f ms/frame  reproject %.2f  pp0 %.2f random %.1f" % (sys.argv[1], d["value"], d["ms_per_frame"], s["reproject"] / 2, d["roofline"]["kernel_ms"], s["random_proposals"] / 2))
PY
done; done 2>&1 | tee gpurun_out/${TAG}_variants.txt
