#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run via gpurun from the repo root):
#   kernel-trace stats, then PMC passes (each in its own run, as the gfx950 guide prescribes), plus the
#   dword-store / dword-load calibration kernels for FETCH_SIZE / WRITE_SIZE.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc; rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 $R/tools/pmc_calib.hip -o /tmp/pmc_calib || exit 1
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.json 2> $OUT/trace.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- $BENCH > $OUT/$c.json 2> $OUT/$c.err || exit 1
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/calib_$c -- /tmp/pmc_calib > $OUT/calib_$c.txt 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/lds -- $BENCH > $OUT/lds.json 2> $OUT/lds.err || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.json 2> $OUT/sq.err || exit 1
echo done
