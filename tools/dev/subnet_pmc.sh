#!/bin/bash
# dev: SQ counters of the fused subnet kernel (timing script, N = 8 only)
R=$PWD; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_subnet/sq -- python3 $R/tools/dev/subnet_time.py > $O/pmc_subnet.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc_subnet/sq2 -- python3 $R/tools/dev/subnet_time.py >> $O/pmc_subnet.log 2>&1
find $O/pmc_subnet -name '*counter_collection.csv' | head
