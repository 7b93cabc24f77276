#!/bin/bash
# Round-2 profile of the bench command on the GPU box (run from the repo root through gpurun):
#   kernel trace + stats, then the two HBM traffic counter passes (separate runs, --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/r02_prof $R/gpurun_out/r02_pmc
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02_prof_bench.json 2> $R/gpurun_out/r02_prof_bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r02_pmc/fetch --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r02_pmc/write --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R && python3 profiles/make_pmc_traffic.py gpurun_out/r02_pmc r02 1048576
cp profiles/r02_pmc_hbm_traffic.json gpurun_out/
ls gpurun_out/r02_prof/*/ | head
