#!/bin/bash
# Instruction-class calibration on the GPU box: cycles per wave64 instruction (tools/valu_ubench.hip) and which
# rocprofv3 SQ_INSTS_VALU_* counter sees which class (one dispatch per class under --pmc). -> gpurun_out/<tag>_ubench*
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=facebook360_dep_amd/bin/valu_ubench
$B > gpurun_out/${tag}_ubench.jsonl 2> gpurun_out/${tag}_ubench.err
rocprofv3 -L > gpurun_out/${tag}_counters_list.txt 2>&1
pmc() {
  rm -rf /tmp/ub_$1
  rocprofv3 --pmc $2 --output-format csv -d /tmp/ub_$1 -o p -- $B --classes > /dev/null 2> gpurun_out/${tag}_ubench_pmc_$1.err
  python tools/pmc_summarize.py /tmp/ub_$1 gpurun_out/${tag}_ubench_pmc_$1.json
}
pmc F32 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
pmc F64 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES"
# the same typed counters on the bench's kernels (2 frames: the level-0 launches are what is priced)
pmcb() {
  rm -rf /tmp/ubb_$1
  rocprofv3 --pmc $2 --output-format csv -d /tmp/ubb_$1 -o p -- python bench.py --steps 1 --warmup 0 --frames 2 --no-cpu-baseline --no-single-frame > /dev/null 2> gpurun_out/${tag}_pmc_$1.err
  python tools/pmc_summarize.py /tmp/ubb_$1 gpurun_out/${tag}_pmc_$1.json
}
pmcb VALU_F32 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
pmcb VALU_F64 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES"
