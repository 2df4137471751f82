#!/bin/bash
# Counter passes of a round, each a rocprofv3 run of its own (this pool: one --pmc set per run, no trace domains
# beside it).  Results land under gpurun_out/pmc/; summarise them afterwards, off the box, with
#   python tools/pmc_summarise.py gpurun_out/pmc/decode_info.json gpurun_out/pmc/fetch gpurun_out/pmc/write > profiles/pmc_traffic.json
#   python tools/pmc_summarise.py gpurun_out/pmc/decode_info_32_8.json gpurun_out/pmc/fetch32 gpurun_out/pmc/write32 > profiles/pmc_traffic_32_8.json
#   python tools/pmc_prefill_summarise.py gpurun_out/pmc/pf_info.json gpurun_out/pmc/pf1 gpurun_out/pmc/pf2 gpurun_out/pmc/pf_fetch gpurun_out/pmc/pf_write > profiles/rN_pmc_prefill.json
set -e
out=gpurun_out/pmc
rm -rf $out; mkdir -p $out/tmp
export TMPDIR=$PWD/$out/tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/pmc_decode.py 8 $out/decode_info.json > $out/fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/pmc_decode.py 8 > $out/write.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch32 -- python3 tools/pmc_decode.py 8 $out/decode_info_32_8.json 32,8 > $out/fetch32.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write32 -- python3 tools/pmc_decode.py 8 - 32,8 > $out/write32.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pf1 -- python3 tools/pmc_prefill.py $out/pf_info.json > $out/pf1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU --output-format csv -d $out/pf2 -- python3 tools/pmc_prefill.py > $out/pf2.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf_fetch -- python3 tools/pmc_prefill.py > $out/pf_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pf_write -- python3 tools/pmc_prefill.py > $out/pf_write.log 2>&1
rm -rf $out/tmp
# keep only the counter tables (the merge back is capped in size)
find $out -type f ! -name "*counter_collection.csv" ! -name "*.json" ! -name "*.log" -delete
find $out -name "*counter_collection.csv" | head
