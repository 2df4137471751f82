set -e
export TMPDIR=$PWD/gpurun_out/tmp; mkdir -p $TMPDIR
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_pf1 -- python3 tools/pmc_prefill.py > gpurun_out/pmc_pf1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_pf2 -- python3 tools/pmc_prefill.py > gpurun_out/pmc_pf2.log 2>&1
rm -rf $TMPDIR
find gpurun_out/pmc_pf1 gpurun_out/pmc_pf2 -name "*counter_collection.csv" | head
