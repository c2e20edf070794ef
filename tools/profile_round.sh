#!/bin/bash
# The measurement set of a round, on the GPU box (through gpurun): bench lines of every workload, the
# rocprofv3 kernel statistics of the bench commands, the two HBM-traffic PMC passes of the default one and the SQ
# counter passes of the decode kernels.
# usage: tools/profile_round.sh <tag> [bench|stats|pmc|sq|all]   (outputs under gpurun_out/<tag>_*;
#                                           tools/collect_profiles.py copies the summaries to profiles/; the parts fit
#                                           one gpurun call each)
set -e
tag=${1:-r06}
part=${2:-all}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
if [ $part = all ] || [ $part = bench ]; then
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python3 bench.py --workload encode > $out/${tag}_bench_encode.json 2>> $out/${tag}_bench.err
python3 bench.py --workload coresident > $out/${tag}_bench_coresident.json 2>> $out/${tag}_bench.err
python3 bench.py --workload adpcm > $out/${tag}_bench_adpcm.json 2>> $out/${tag}_bench.err
python3 bench.py --workload amvlib > $out/${tag}_bench_amvlib.json 2>> $out/${tag}_bench.err
python3 bench.py --width 320 --height 240 --frames 32000 --no-cpu-baseline > $out/${tag}_bench_decode320.json 2>> $out/${tag}_bench.err
python3 bench.py --frames 10000 --no-cpu-baseline > $out/${tag}_bench_decode10k.json 2>> $out/${tag}_bench.err
python3 bench.py --strong --frames 20000 --no-cpu-baseline > $out/${tag}_bench_strong_world1.json 2>> $out/${tag}_bench.err
python3 bench.py --stream mixed --no-secondary > $out/${tag}_bench_mixed.json 2>> $out/${tag}_bench.err
python3 bench.py --stream amv1 --frames 200000 --no-secondary > $out/${tag}_bench_amv1.json 2>> $out/${tag}_bench.err
# the line an N-rank run leaves, rehearsed at world size 1 under the launcher the driver uses (RCCL, one rank)
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --strong > $out/${tag}_bench_torchrun_world1.json 2>> $out/${tag}_bench.err
python3 bench.py --frames 1250 --no-cpu-baseline > $out/${tag}_bench_decode1250.json 2>> $out/${tag}_bench.err
# ... and with TWO real ranks on this one GPU: gloo instead of RCCL, both ranks on device 0 -- the launcher, the store, the side
# channel, the kernels, the exchange and the record of an N-rank run, under bench.py's own launcher and under the driver's
AMV_BENCH_BACKEND=gloo AMV_BENCH_ONE_DEVICE=1 timeout -k 10 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 > $out/${tag}_rehearsal_two_ranks_one_gpu_gloo.json 2>> $out/${tag}_bench.err
AMV_BENCH_BACKEND=gloo AMV_BENCH_ONE_DEVICE=1 timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 2 --steps 3 --warmup 1 > $out/${tag}_rehearsal_two_ranks_one_gpu_gloo_torchrun.json 2>> $out/${tag}_bench.err
python3 tools/time_kernels.py --frames 10000 --trace $out/${tag}_trace_10k.npy > $out/${tag}_trace_10k.json 2>> $out/${tag}_bench.err
python3 tools/time_kernels.py --frames 1250 --trace $out/${tag}_trace_1250.npy > $out/${tag}_trace_1250.json 2>> $out/${tag}_bench.err
echo "benches done"
fi
cd /tmp && export TMPDIR=/tmp
if [ $part = all ] || [ $part = stats ]; then
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o run --output-format csv -- python3 $root/bench.py --no-cpu-baseline --no-secondary > $out/${tag}_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats_encode -o run --output-format csv -- python3 $root/bench.py --workload encode --no-cpu-baseline > $out/${tag}_stats_encode.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats_coresident -o run --output-format csv -- python3 $root/bench.py --workload coresident --no-cpu-baseline > $out/${tag}_stats_coresident.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats_adpcm -o run --output-format csv -- python3 $root/bench.py --workload adpcm --no-cpu-baseline > $out/${tag}_stats_adpcm.log 2>&1
echo "stats done"
fi
if [ $part = all ] || [ $part = pmc ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o run --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o run --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $out/${tag}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch_encode -o run --output-format csv -- python3 $root/bench.py --workload encode --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch_encode.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write_encode -o run --output-format csv -- python3 $root/bench.py --workload encode --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write_encode.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch_decode320 -o run --output-format csv -- python3 $root/bench.py --width 320 --height 240 --frames 128000 --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch_decode320.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write_decode320 -o run --output-format csv -- python3 $root/bench.py --width 320 --height 240 --frames 128000 --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write_decode320.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch_adpcm -o run --output-format csv -- python3 $root/bench.py --workload adpcm --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch_adpcm.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write_adpcm -o run --output-format csv -- python3 $root/bench.py --workload adpcm --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write_adpcm.log 2>&1
echo "pmc done"
fi
if [ $part = all ] || [ $part = sq ]; then
bash $root/tools/pmc_sq.sh ${tag}sq 160000
bash $root/tools/pmc_sq_bench.sh ${tag}sqenc --workload encode
bash $root/tools/pmc_sq_bench.sh ${tag}sqadpcm --workload adpcm
fi
echo done
