cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in cur tb8 tb2; do
  DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_$v.so PMC_TIMEOUT=240 bash tools/pmc_run.sh r6n_$v "FETCH_SIZE" --no-single-frame --frames 1 > /dev/null 2>&1
  python - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/pmc_r6n_%s.json" % sys.argv[1]))
out = []
tot = 0
for k, v in d.items():
    f = v.get("FETCH_SIZE")
    if not f: continue
    tot += f["sum"]
    if "k_ping_pong(" in k or "k_random_proposals(" in k or "k_reproject" in k:
        out.append("%s level-0 %.1f GB (all launches %.1f GB)" % (k.split("(")[0].replace("derp::", ""), f["max"] * 2048 / 1e9, f["sum"] * 2048 / 1e9))
print(sys.argv[1], "| fetched per frame, all kernels: %.1f GB |" % (tot * 2048 / 1e9), " | ".join(out))
PY
done
