#!/bin/bash
# MFMA utilisation by counter (separate --pmc passes, kernel-trace only): run on the GPU box through gpurun.
#   tools/prof_mfma.sh <tag> [bench args...]
# Writes gpurun_out/mfma_<tag>/{a,b}/ ; tools/summarize_mfma.py condenses them into profiles/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r02}; shift
OUT=gpurun_out/mfma_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
run() {  # name, counters...
    local name=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- \
        python3 bench.py --no-cpu-baseline --no-latency --no-extra --steps 10 --warmup 3 "${BARGS[@]}" > "$OUT/$name.json" 2> "$OUT/$name.err"
    local rc=$?
    echo "pass $name rc=$rc" | tee -a "$OUT/passes.txt"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
BARGS=("$@")
run a SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA
run b SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES
run c SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
# round 4: what the wavefronts execute -- FP64 vector instructions by kind (x 64 lanes x 1 or 2 FLOPs), scalar and LDS
# instructions; and the LDS conflict counters in more detail
run d SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM
run e SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_INT32
exit 0
