#!/bin/bash
# Collect the round's judged evidence on the GPU box into gpurun_out/ (tools/make_profiles.py then trims it
# into profiles/):
#   <tag>_bench.json                 the un-profiled bench line (default command)
#   <tag>_kernel_stats_full.csv      rocprofv3 --kernel-trace --stats of the same command
#   <tag>_pmc_<GROUP>.json           one rocprofv3 --pmc pass per counter group, summarised per kernel
# PMC passes run `bench.py --steps 1 --warmup 0` (every kernel of the sequence once per frame); counters
# never share a run with a trace domain.          usage: tools/profile_round.sh <tag> [bench args]
tag=${1:-r03}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-frame "$@" > gpurun_out/${tag}_bench_under_rocprof.json 2> /dev/null
cp /tmp/prof_$tag/${tag}_kernel_stats.csv gpurun_out/${tag}_kernel_stats_full.csv
pmc() {  # name, counters
  rm -rf /tmp/pmc_$1
  rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_$1 -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single-frame "${@:3}" > /dev/null 2> gpurun_out/${tag}_pmc_$1.err
  python tools/pmc_summarize.py /tmp/pmc_$1 gpurun_out/${tag}_pmc_$1.json
}
pmc FETCH_SIZE "FETCH_SIZE" "$@"
pmc WRITE_SIZE "WRITE_SIZE" "$@"
pmc SQ_ISSUE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "$@"
pmc SQ_INSTS "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "$@"
# typed VALU counters (tools/valu_model.py prices them per instruction class): the level-0 launches are what is
# priced, two frames have them
pmc VALU_F32 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" --frames 2 "$@"
pmc VALU_F64 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" --frames 2 "$@"
# L1 <-> L2 traffic (random proposals: every lane of a wave gathers from its own cache lines)
pmc L2 "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "$@"
