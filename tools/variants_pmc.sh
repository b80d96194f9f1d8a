#!/bin/bash
# Developer A/B with counters: for every facebook360_dep_amd/libderp_var_*.so one rocprofv3 --pmc pass per counter
# group over a 2-frame bench; prints the level-0 k_ping_pong and k_random_proposals dispatch (max) of each counter.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT"
for lib in ${VARIANT_LIBS:-facebook360_dep_amd/libderp_var_*.so}; do
  name=$(basename $lib .so)
  for g in 1 2; do
    eval "C=\$G$g"
    rm -rf /tmp/pmc_v
    DERP_LIB=$PWD/$lib timeout 900 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_v -o p -- python bench.py --frames 2 --steps 1 --warmup 0 --no-cpu-baseline --no-single-frame > /dev/null 2> gpurun_out/pmcv_${name}_$g.err
    python tools/pmc_summarize.py /tmp/pmc_v gpurun_out/pmcv_${name}_$g.json > /dev/null
  done
  python - $name <<'PY'
import json, sys
name = sys.argv[1]
row = {}
for g in (1, 2):
    d = json.load(open("gpurun_out/pmcv_%s_%d.json" % (name, g)))
    for k, v in d.items():
        short = "pp" if "k_ping_pong(" in k or "k_ping_pongE" in k else "rnd" if "k_random_proposals" in k else None
        if short and "commit" not in k:
            for c, x in v.items():
                row.setdefault(short, {})[c] = x["max"]
for short, r in row.items():
    wc = r.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-18s %-4s" % (name, short), " ".join("%s=%.3g" % (c.replace("SQ_", ""), v) for c, v in sorted(r.items())))
    print("%-18s %-4s shares of wave-cycles: VALU %.3f SCA %.3f LDS %.3f WAIT_ANY %.3f WAIT_INST %.3f" % (
        name, short, r.get("SQ_ACTIVE_INST_VALU", 0) / wc, r.get("SQ_ACTIVE_INST_SCA", 0) / wc, r.get("SQ_ACTIVE_INST_LDS", 0) / wc,
        r.get("SQ_WAIT_ANY", 0) / wc, r.get("SQ_WAIT_INST_ANY", 0) / wc))
PY
done
