#!/bin/bash
tag=$1
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
F="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mfma f16 --patch 256 --no-series --no-kernel-timing"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_h_fetch -o r --output-format csv -- $F > $R/gpurun_out/${tag}_h_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_h_write -o r --output-format csv -- $F > $R/gpurun_out/${tag}_h_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_h_stats -o r --output-format csv -- $F > $R/gpurun_out/${tag}_h_stats.log 2>&1 || exit 1
cd $R
python3 tools/pmc_traffic.py gpurun_out/${tag}_h_fetch gpurun_out/${tag}_h_write > gpurun_out/${tag}_pmc_traffic_f16_256.json
find gpurun_out/${tag}_h_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_f16_256_kernel_stats.csv \;
rm -rf gpurun_out/${tag}_h_fetch gpurun_out/${tag}_h_write gpurun_out/${tag}_h_stats
echo done
