#!/bin/bash
# Round-3 profile of the bench command on the GPU box (run from the repo root through gpurun):
#   1. the default bench line (with extra_configs), 2. the same bench under a one-rank RCCL process group
#   (torch.distributed.run --nproc-per-node 1), 3. kernel trace + stats of the default command (all configurations),
#   4. the two HBM traffic counter passes (separate runs, --kernel-trace only) of the C3 leg;
#   0. before them: smoke(), the small-batch latency table and the graphed training step (second session of the round).
R=/root/repo
cd $R
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/r03_smoke.log
python3 profiles/tools/small_batch_latency.py > gpurun_out/r03_small_batch_latency.log 2>&1; echo "latency rc $?"
python3 profiles/tools/train_graph_bench.py > gpurun_out/r03_train_graph_bench.log 2>&1; echo "train graph rc $?"
python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc $?"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r03_rccl_one_rank.json 2> gpurun_out/r03_rccl_one_rank.err; echo "rccl bench rc $?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r03_prof $R/gpurun_out/r03_pmc
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_prof --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r03_prof_bench.json 2> $R/gpurun_out/r03_prof_bench.err; echo "prof rc $?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r03_pmc/fetch --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1; echo "fetch rc $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r03_pmc/write --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1; echo "write rc $?"
cd $R && python3 profiles/make_pmc_traffic.py gpurun_out/r03_pmc r03 1048576
cp profiles/r03_pmc_hbm_traffic.json gpurun_out/
find gpurun_out/r03_prof -name "*kernel_stats.csv" | head -3
