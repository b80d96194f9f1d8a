#!/bin/bash
# BASELINE config 4 (24 x 4096^2, single frame, destinations batched under the table budget): bench line +
# HBM traffic passes -> gpurun_out/<tag>_* (tools/make_profiles.py <tag> cfg4 trims them into profiles/)
tag=${1:-r06cfg4}
ARGS="--config cfg4 --frames 1 --temporal 0 --no-cpu-baseline --no-single-frame"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 $ARGS > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py --steps 2 --warmup 1 $ARGS > /dev/null 2>&1
cp /tmp/prof_$tag/${tag}_kernel_stats.csv gpurun_out/${tag}_kernel_stats_full.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> gpurun_out/${tag}_pmc_$c.err
  python tools/pmc_summarize.py /tmp/pmc_$c gpurun_out/${tag}_pmc_$c.json
done
rm -rf /tmp/pmc_SQ
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_SQ -o p -- python bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> gpurun_out/${tag}_pmc_SQ_ISSUE.err
python tools/pmc_summarize.py /tmp/pmc_SQ gpurun_out/${tag}_pmc_SQ_ISSUE.json
pmc() {  # name, counters
  rm -rf /tmp/pmc_$1
  rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_$1 -o p -- python bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> gpurun_out/${tag}_pmc_$1.err
  python tools/pmc_summarize.py /tmp/pmc_$1 gpurun_out/${tag}_pmc_$1.json
}
# round 6: the counters tools/valu_model.py and the L2 hit rate need, so that config 4's bench line carries a full roofline
pmc SQ_INSTS "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"
pmc VALU_F32 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
pmc VALU_F64 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES"
pmc L2 "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
