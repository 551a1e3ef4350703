cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in v1 v2p; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_attn_${v}_a --output-format csv -- python $R/tools/attn_pmc_run.py $v > $R/gpurun_out/pmc_attn_${v}_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_attn_${v}_b --output-format csv -- python $R/tools/attn_pmc_run.py $v > $R/gpurun_out/pmc_attn_${v}_b.log 2>&1
done
cd $R
python tools/pmc_table.py gpurun_out/pmc_attn_v1_a gpurun_out/pmc_attn_v1_b --match=attn_mfma > gpurun_out/r03_attn_pmc_v1.json
python tools/pmc_table.py gpurun_out/pmc_attn_v2p_a gpurun_out/pmc_attn_v2p_b --match=attn_mfma > gpurun_out/r03_attn_pmc_v2p.json
cat gpurun_out/r03_attn_pmc_v1.json gpurun_out/r03_attn_pmc_v2p.json; tail -3 gpurun_out/pmc_attn_v2p_b.log
