#!/bin/bash
# PMC passes over csrc/tools/layer_bench (single-layer variants) — run on the MI355X box from the repo root.
set -o pipefail
TAG=${1:-lb}
OUT=gpurun_out/pmc_${TAG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
BIN=./celebrity_image_denoiser_amd/csrc/tools/layer_bench
rocprofv3 --kernel-trace --output-format csv -d "$OUT/p1" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -- $BIN 256 > "$OUT/p1.txt" 2>&1 || { echo p1 failed; tail -3 "$OUT/p1.txt"; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d "$OUT/p2" --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES -- $BIN 256 > "$OUT/p2.txt" 2>&1 || { echo p2 failed; tail -3 "$OUT/p2.txt"; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d "$OUT/p3" --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_DATA_FIFO_FULL -- $BIN 256 > "$OUT/p3.txt" 2>&1 || { echo p3 failed; tail -3 "$OUT/p3.txt"; exit 1; }
echo done
