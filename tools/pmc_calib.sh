#!/bin/bash
# PMC calibration on the GPU box (through gpurun from the repo root): tools/pmc_calib.sh TAG
# runs tools/ubench/pmc_calib.bin three times: plain (timings), under rocprofv3 --pmc FETCH_SIZE, under --pmc WRITE_SIZE
# (separate passes, counters never combined with other trace domains), then tools/pmc_calib_report.py writes the factors.
TAG=${1:-r04}; R=$(pwd); O=$R/gpurun_out/${TAG}_pmc_calib; mkdir -p $O
export TMPDIR=/tmp
$R/tools/ubench/pmc_calib.bin > $O/timings.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o $C -- $R/tools/ubench/pmc_calib.bin > $O/$C.log 2>&1)
done
python $R/tools/pmc_calib_report.py $O $O/pmc_calibration.json | tee $O/pmc_calibration.txt
