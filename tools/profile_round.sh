#!/bin/bash
# One GPU-box call that produces every rocprofv3 artefact profiles/README.md cites for a round:
#   tools/profile_round.sh r01f        (run from the repo root on the GPU box; outputs under gpurun_out/<tag>_*)
# rocprofv3 rules of this pool: program directly after `--`, PMC passes separate from each other and with
# --kernel-trace only.
tag=${1:-rXX}
only=${2:-all}        # "split": the four passes of the split-operand step; "c5trained": only the four passes of the config-5 everything-trained step (a change to the fp16 backward kernels)
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
if [ "$only" = "c5trained" ]; then
  G="python3 $R/tools/bench_c5_trained.py --no-fp32 --steps 2 --warmup 1"
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_g_stats -o r --output-format csv -- $G > $R/gpurun_out/${tag}_g_stats.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_g_fetch -o r --output-format csv -- $G > $R/gpurun_out/${tag}_g_fetch.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_g_write -o r --output-format csv -- $G > $R/gpurun_out/${tag}_g_write.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $R/gpurun_out/${tag}_g_mfma -o r --output-format csv -- $G > $R/gpurun_out/${tag}_g_mfma.log 2>&1 || exit 1
  cd $R
  python3 tools/pmc_traffic.py gpurun_out/${tag}_g_fetch gpurun_out/${tag}_g_write > gpurun_out/${tag}_pmc_traffic_c5trained.json
  python3 tools/pmc_mfma_util.py gpurun_out/${tag}_g_mfma > gpurun_out/${tag}_pmc_mfma_util_c5trained.json
  find gpurun_out/${tag}_g_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_bench_c5trained_kernel_stats.csv \;
  for d in g_stats g_fetch g_write g_mfma; do rm -rf gpurun_out/${tag}_$d; done
  echo done
  exit 0
fi
if [ "$only" = "split" ]; then
  # the headline's step on the opt-in split-operand convs (bench.py's series headline_split_128px alone)
  P="python3 $R/tools/bench_split.py"
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_s_stats -o r --output-format csv -- $P > $R/gpurun_out/${tag}_s_stats.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_s_fetch -o r --output-format csv -- $P > $R/gpurun_out/${tag}_s_fetch.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_s_write -o r --output-format csv -- $P > $R/gpurun_out/${tag}_s_write.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $R/gpurun_out/${tag}_s_mfma -o r --output-format csv -- $P > $R/gpurun_out/${tag}_s_mfma.log 2>&1 || exit 1
  cd $R
  python3 tools/pmc_traffic.py gpurun_out/${tag}_s_fetch gpurun_out/${tag}_s_write > gpurun_out/${tag}_pmc_traffic_split.json
  python3 tools/pmc_mfma_util.py gpurun_out/${tag}_s_mfma > gpurun_out/${tag}_pmc_mfma_util_split.json
  find gpurun_out/${tag}_s_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_bench_split_kernel_stats.csv \;
  for d in s_stats s_fetch s_write s_mfma; do rm -rf gpurun_out/${tag}_$d; done
  echo done
  exit 0
fi
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-series"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o r --output-format csv -- $B > $R/gpurun_out/${tag}_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_fetch -o r --output-format csv -- $B --no-kernel-timing > $R/gpurun_out/${tag}_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_write -o r --output-format csv -- $B --no-kernel-timing > $R/gpurun_out/${tag}_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $R/gpurun_out/${tag}_mfma -o r --output-format csv -- $B --no-kernel-timing > $R/gpurun_out/${tag}_mfma.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_trainf -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-f --no-series --no-kernel-timing > $R/gpurun_out/${tag}_trainf.log 2>&1 || exit 1
# the f-trained series and config 5 (fp16, 256 px): HBM bytes per launch of their dominant kernels
T="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-f --no-series --no-kernel-timing"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_tf_fetch -o r --output-format csv -- $T > $R/gpurun_out/${tag}_tf_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_tf_write -o r --output-format csv -- $T > $R/gpurun_out/${tag}_tf_write.log 2>&1 || exit 1
F="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mfma f16 --patch 256 --no-series --no-kernel-timing"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_h_fetch -o r --output-format csv -- $F > $R/gpurun_out/${tag}_h_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_h_write -o r --output-format csv -- $F > $R/gpurun_out/${tag}_h_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_h_stats -o r --output-format csv -- $F > $R/gpurun_out/${tag}_h_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $R/gpurun_out/${tag}_h_mfma -o r --output-format csv -- $F > $R/gpurun_out/${tag}_h_mfma.log 2>&1 || exit 1
# config 5 with everything trained on the fp16-MFMA gradient path (round 4)
G="python3 $R/tools/bench_c5_trained.py --no-fp32 --steps 2 --warmup 1"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_g_stats -o r --output-format csv -- $G > $R/gpurun_out/${tag}_g_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_g_fetch -o r --output-format csv -- $G > $R/gpurun_out/${tag}_g_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_g_write -o r --output-format csv -- $G > $R/gpurun_out/${tag}_g_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $R/gpurun_out/${tag}_g_mfma -o r --output-format csv -- $G > $R/gpurun_out/${tag}_g_mfma.log 2>&1 || exit 1
cd $R
python3 tools/pmc_traffic.py gpurun_out/${tag}_g_fetch gpurun_out/${tag}_g_write > gpurun_out/${tag}_pmc_traffic_c5trained.json
python3 tools/pmc_mfma_util.py gpurun_out/${tag}_mfma > gpurun_out/${tag}_pmc_mfma_util.json
python3 tools/pmc_mfma_util.py gpurun_out/${tag}_h_mfma > gpurun_out/${tag}_pmc_mfma_util_f16_256.json
python3 tools/pmc_mfma_util.py gpurun_out/${tag}_g_mfma > gpurun_out/${tag}_pmc_mfma_util_c5trained.json
python3 tools/pmc_traffic.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write > gpurun_out/${tag}_pmc_traffic.json
python3 tools/pmc_traffic.py gpurun_out/${tag}_tf_fetch gpurun_out/${tag}_tf_write > gpurun_out/${tag}_pmc_traffic_trainf.json
python3 tools/pmc_traffic.py gpurun_out/${tag}_h_fetch gpurun_out/${tag}_h_write > gpurun_out/${tag}_pmc_traffic_f16_256.json
for pair in stats:bench trainf:bench_trainf h_stats:bench_f16_256 g_stats:bench_c5trained; do
  find gpurun_out/${tag}_${pair%%:*} -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_${pair##*:}_kernel_stats.csv \;
done
# keep the summaries, drop the raw per-dispatch traces (hundreds of MB)
for d in stats fetch write mfma trainf tf_fetch tf_write h_fetch h_write h_stats h_mfma g_stats g_fetch g_write g_mfma; do rm -rf gpurun_out/${tag}_$d; done
echo done
