cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2000 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_gpu_tests.txt | tail -2
DERP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 > gpurun_out/${TAG}_plain_gpus2.json 2> gpurun_out/${TAG}_plain_gpus2.err; echo "gpus2 rc=$? stdout lines: $(wc -l < gpurun_out/${TAG}_plain_gpus2.json)"
DERP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 8 --backend gloo --steps 2 --warmup 1 > gpurun_out/${TAG}_plain_gpus8.json 2> gpurun_out/${TAG}_plain_gpus8.err; echo "gpus8 rc=$? stdout lines: $(wc -l < gpurun_out/${TAG}_plain_gpus8.json)"
python - <<'PY'
import json
for n in (2, 8):
    try:
        d = json.load(open("gpurun_out/%s_plain_gpus%d.json" % (__import__("os").environ["TAG"], n)))
        print(n, d["n_gpus"], d["value"], d["result_crc_matches_n1"], d.get("halo_transport_per_rank"), json.dumps(d.get("exchange"))[:400])
    except Exception as e:
        print(n, "failed", e)
PY
