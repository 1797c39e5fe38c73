#!/bin/bash
# Run on the MI355X box from the repo root:  bash profiles/collect.sh <tag>
# Writes raw rocprofv3 CSVs under gpurun_out/prof_<tag>/{stats,pmc_sq,pmc_fetch,pmc_write}; summarise with
# profiles/summarize.py.  Counter passes are separate runs with --kernel-trace only (never combined
# with other trace domains), as the MI355X guide prescribes.
set -o pipefail
TAG=${1:-run}
OUT=gpurun_out/prof_${TAG}
# Always start from an empty directory: round 2's "final" summary was condensed from a directory that still held an earlier
# run's CSVs (summarize.py then averaged two builds).  A stale tag is removed, never merged into.
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
# BENCH_ARGS: extra bench.py arguments (e.g. "--dtype f16 --batch-per-gpu 512" for BASELINE configs[4])
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS"
SHORT="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_stats.json" 2> "$OUT/stats.err" || { echo "stats pass failed"; tail -5 "$OUT/stats.err"; exit 1; }
echo "stats pass done"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_sq" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $SHORT > "$OUT/bench_pmc_sq.json" 2> "$OUT/pmc_sq.err" || { echo "pmc sq pass failed"; tail -5 "$OUT/pmc_sq.err"; exit 1; }
echo "pmc sq pass done"
# ALU accounting: how many vector instructions are issued beside the MFMAs.  The fp32 MFMA runs at the vector
# rate on the SIMD's ALUs, so (MFMA busy cycles + VALU issue cycles) / SIMD cycles is the ceiling argument of DESIGN.md section 2.
rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_alu" --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 -- $SHORT > "$OUT/bench_pmc_alu.json" 2> "$OUT/pmc_alu.err" || { echo "pmc alu pass failed"; tail -5 "$OUT/pmc_alu.err"; exit 1; }
echo "pmc alu pass done"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_fetch" --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- $SHORT > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || { echo "pmc fetch pass failed"; tail -5 "$OUT/pmc_fetch.err"; exit 1; }
echo "pmc fetch pass done"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_write" --pmc WRITE_SIZE -- $SHORT > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err" || { echo "pmc write pass failed"; tail -5 "$OUT/pmc_write.err"; exit 1; }
echo "pmc write pass done"
# optional diagnostic pass: EXTRA_PMC="<counters>" (e.g. the LDS set: SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
# SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL) -> $OUT/pmc_extra
if [ -n "$EXTRA_PMC" ]; then
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_extra" --pmc $EXTRA_PMC -- $SHORT > "$OUT/bench_pmc_extra.json" 2> "$OUT/pmc_extra.err" || { echo "pmc extra pass failed"; tail -5 "$OUT/pmc_extra.err"; exit 1; }
  echo "pmc extra pass done"
fi
find "$OUT" -name "*.csv" | head -40
