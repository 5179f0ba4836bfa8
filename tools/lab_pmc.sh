# PMC of the lab kernels: bash tools/lab_pmc.sh <outdir> (env for tools/g8_lab.py is passed through)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); O=$R/${1:-gpurun_out/labpmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/p1 -o r -- python $R/tools/g8_lab.py > $O/run.txt 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/p2 -o r -- python $R/tools/g8_lab.py >> $O/run.txt 2>&1
cd $R
python tools/rocpd_pmc.py $(ls $O/p1/*results.db $O/p1/*/*results.db 2>/dev/null | head -1) $(ls $O/p2/*results.db $O/p2/*/*results.db 2>/dev/null | head -1) --out $O/pmc.csv
grep -i "gemm" $O/pmc.csv
