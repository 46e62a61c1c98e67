#!/bin/bash
# usage: tools/pmc_run.sh <tag> <bench args...>   -- two rocprofv3 --pmc passes + kernel trace, summarised to gpurun_out/<tag>_pmc{1,2}.json
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1 /tmp/pmc2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmc1 -o p --output-format csv -- python $R/bench.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/pmc2 -o p --output-format csv -- python $R/bench.py "$@" > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc1 $R/gpurun_out/${tag}_pmc1.json
python $R/tools/pmc_summary.py /tmp/pmc2 $R/gpurun_out/${tag}_pmc2.json
